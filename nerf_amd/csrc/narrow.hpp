// Narrow radiance fields (the distilled student of distill.py:193-195: width 16 or 32, depth 2, no
// skip connection): how their weights are laid out for the whole-ray kernel of narrow.hip.
//
// Such a model is 4,628 parameters: every matrix is one or two MFMA A operands per lane, so a
// wavefront keeps ALL weights in registers for the whole launch and there is no weight stream, no
// LDS ring and no barrier.  What the kernel spends its time on instead is the encoding (36 sines and
// cosines per sample against 9,024 FLOP of matrix work), so the encoder's slots are laid out for an
// even split over the two lane halves of a 32x32x16 MFMA's B operand with no padding pairs in front
// of real ones: a pair-slot POSITION pos = 4 u + q (k-unit u, slots 2 q and 2 q + 1 of each half)
// holds one sin/cos pair per lane half; a positional encoder's raw coordinates take positions 0 and 1
// of half 0 (x, y | z, 0).  PE(3,6): 18 pairs + raw = 10 positions (3 k-units) where the general
// kernels' layout evaluates 16.
//
// Included by mlp.hpp (after the layout helpers it uses); include that, not this file.
#pragma once

namespace nerfx {

constexpr int NARROW_MAX_UNITS = 4;              // encoder k-units (16 slots each)
constexpr int NARROW_MAX_POS = 4 * NARROW_MAX_UNITS;
constexpr int NARROW_CHAIN_UNITS = 2;            // chained k-units: widths up to 32

// Pair evaluated by lane half h at position pos (-1: raw coordinates or nothing).
__host__ __device__ inline int narrow_pair_index(int kind, int pos, int h) {
    if (kind == NERFX_ENC_POSITIONAL) return pos < 2 ? (h == 1 ? pos : -1) : 2 * pos - 2 + h;
    return 2 * pos + h;
}
// Positions an encoder with P pairs occupies.
__host__ __device__ inline int narrow_positions(int kind, int pairs) {
    if (kind == NERFX_ENC_POSITIONAL) return pairs <= 2 ? 2 : 2 + (pairs - 2 + 1) / 2;
    return (pairs + 1) / 2;
}
// Column of the reference encoding (features.py:32-36, :52) in element e of lane half h of k-unit u; -1: padding.
__host__ __device__ inline int narrow_slot_col(int kind, int n, int u, int h, int e) {
    const int pos = 4 * u + (e >> 1), which = e & 1;
    if (kind == NERFX_ENC_POSITIONAL && pos < 2 && h == 0) {
        const int c = 2 * pos + which;
        return c < 3 ? c : -1;
    }
    const int k = narrow_pair_index(kind, pos, h);
    const int pairs = kind == NERFX_ENC_POSITIONAL ? 3 * n : n;
    if (k < 0 || k >= pairs) return -1;
    if (kind == NERFX_ENC_POSITIONAL) { const int f = k / 3, c = k - 3 * f; return 3 + 6 * f + 3 * which + c; }
    return which * n + k;
}
// Row of a 32x32 accumulator tile held in register r of lane half h.
__host__ __device__ constexpr int acc_row(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }

// The packed block: [A operands, 1 KiB units of 64 lanes x 16 B][floats].
// The density head is a matrix product too (the matrix pipes are mostly idle, vector issue is what
// bounds the kernel): rows 0 and 1 of its units hold the weight vector split into a bf16 value and
// the bf16 value of what that rounding left, so the head sees its weights to 16 bits; its inputs are
// the bf16 activations every other layer consumes (what autocast feeds torch's Linear as well).
// Pair tables hold two positions per entry for packed arithmetic: entry (j, h) = positions 2 j, 2 j + 1
// of lane half h as (cx0, cx1, cy0, cy1), (cz0, cz1, 0, 0).
struct NarrowPlan {
    int ok;                                      // the model has this form
    int width, half, activ;
    int kx, nx, pos_x, ex_cols;                  // position encoder: kind, size, positions, reference width
    int kd, nd, pos_d, ed_cols;
    // units (1 KiB each) from the start of the block
    int u_l0, u_l1, u_sigma, u_feat, u_rgb0, u_rgb2, u_rgb0_enc, n_units;
    // floats after the units
    int f_tabx, f_tabd, f_bias, f_sigw, f_sigb, f_cmax, n_floats;
    int64_t bytes;
};
enum { NARROW_BIAS_L0 = 0, NARROW_BIAS_L1, NARROW_BIAS_FEAT, NARROW_BIAS_RGB0, NARROW_BIAS_RGB2, NARROW_BIASES };

inline NarrowPlan narrow_plan(const nerfx_model_desc_t& d) {
    NarrowPlan p{};
    const int px = enc_pairs(d.enc_x_kind, d.enc_x_n), pd = enc_pairs(d.enc_d_kind, d.enc_d_n);
    p.pos_x = narrow_positions(d.enc_x_kind, px);
    p.pos_d = narrow_positions(d.enc_d_kind, pd);
    p.ok = d.depth == 2 && !d.resid && (d.width == 16 || d.width == 32) && p.pos_x <= NARROW_MAX_POS && p.pos_d <= NARROW_MAX_POS;
    p.width = d.width; p.half = d.width / 2; p.activ = d.activ;
    p.kx = d.enc_x_kind; p.nx = d.enc_x_n; p.ex_cols = enc_width(d.enc_x_kind, d.enc_x_n);
    p.kd = d.enc_d_kind; p.nd = d.enc_d_n; p.ed_cols = enc_width(d.enc_d_kind, d.enc_d_n);
    int u = 0;
    p.u_l0 = u; u += NARROW_MAX_UNITS;
    p.u_l1 = u; u += NARROW_CHAIN_UNITS;
    p.u_sigma = u; u += NARROW_CHAIN_UNITS;
    p.u_feat = u; u += NARROW_CHAIN_UNITS;
    p.u_rgb0 = u; u += NARROW_CHAIN_UNITS;
    p.u_rgb2 = u; u += 1;
    p.u_rgb0_enc = u; u += NARROW_MAX_UNITS;
    p.n_units = u;
    int f = 0;
    p.f_tabx = f; f += (NARROW_MAX_POS / 2) * 2 * 8;
    p.f_tabd = f; f += (NARROW_MAX_POS / 2) * 2 * 8;
    p.f_bias = f; f += NARROW_BIASES * 32;
    p.f_sigw = f; f += 32;
    p.f_sigb = f; f += 2;
    p.f_cmax = f; f += 2;                        // largest |coefficient| of the position encoder's table (revolutions per unit)
    p.n_floats = f;
    p.bytes = (int64_t)p.n_units * UNIT_BYTES + (int64_t)p.n_floats * 4;
    return p;
}

}  // namespace nerfx
