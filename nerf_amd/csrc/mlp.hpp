// Layout shared by the weight packer (model.hip) and the fused MLP kernels: how a
// radiance field's Linear layers become one linear "unit stream".
//
// A UNIT is 1 KiB = 64 lanes x 16 B: the A operands (weights) one wavefront reads
// with a single ds_read_b128 for one 32-row output tile and one k-unit (16 input
// features in bf16, 8 in fp32).  Units are stored in exactly the order the kernel
// consumes them, so the kernel's LDS ring is a sliding window over a flat array
// filled by global_load_lds, and lane l always reads unit + 16 l.
//
// Order: layer -> tile group (NG output tiles computed together) -> segment
// (first the k-units fed by an encoder, then the k-units fed by the previous
// layer) -> k-unit -> tile in group.  Every segment is padded to whole PIECES of
// 8 units; the ring is filled and handed over piece by piece.
#pragma once
#include "common.hpp"

namespace nerfx {

constexpr int UNIT_BYTES = 1024;
constexpr int PIECE_UNITS = 8;                  // ring granularity: 8 KiB
// Ring geometry of one kernel configuration: WAVES wavefronts per workgroup share a ring of
// PIECES pieces, AHEAD of which are in flight ahead of the consumer; one hand-over covers
// at most SYNC = PIECES - AHEAD pieces.
template <int WAVES_, int PIECES_, int AHEAD_, int BLOCKS_PER_CU_>
struct RingGeometry {
    static constexpr int WAVES = WAVES_, PIECES = PIECES_, AHEAD = AHEAD_, SYNC = PIECES_ - AHEAD_;
    static constexpr int BLOCKS_PER_CU = BLOCKS_PER_CU_;
};
// bf16: one 8-wave workgroup per CU (two waves per SIMD, 256 registers each).  A ring of 8
// pieces (a power of two: slot arithmetic is a mask) with 4 in flight measured 2 % faster than
// 12 with 8 in flight: the kernel is bound by instructions issued per wavefront, and the scalar
// bookkeeping of the ring is a visible part of them.  6 in flight with hand-overs of 2 pieces
// (twice as many barriers, 1.5 x the time for a piece to land) measured 1 % slower for rendering
// and 11 % slower for a training step.  fp32: one 4-wave workgroup per CU (512
// registers).
using RingBf16 = RingGeometry<8, 8, 4, 1>;
using RingFp32 = RingGeometry<4, 16, 12, 1>;
// The two bf16 kernels that write a training tape (training forward, backward chain): a hand-over's
// wait also sits out every tape store the wave issued before the refill it waits for (one counter,
// retired in issue order), so how far the refills run ahead is how long a store may take.
#ifndef NERFX_TRAIN_RING
#define NERFX_TRAIN_RING 8, 4
#endif
#ifndef NERFX_BWD_RING
#define NERFX_BWD_RING 8, 4
#endif
using RingBf16Train = RingGeometry<8, NERFX_TRAIN_RING, 1>;
using RingBf16Bwd = RingGeometry<8, NERFX_BWD_RING, 1>;
// bf16 kernels keep each wave's positional-encoding B operands (ENC_BLOCK k-units) in LDS.
constexpr int STASH_BYTES_PER_WAVE = 4 * 1024;
constexpr int ENC_BLOCK = 4;                    // encoder k-units come in multiples of this
constexpr int MAX_LAYERS = 20;                  // 2*hd trunk + features + rgb.0 + rgb.2, depth <= 16
constexpr int BIAS_STRIDE = 256;                // floats of const block per layer
constexpr int MAX_TABLE = 1024;                 // floats per encoder table

// A third stream layout beside the two public precisions: bf16 weights cut for
// v_mfma_f32_16x16x32_bf16 (a unit = 16 output rows x 32 input features, lane l holds row l % 16
// and inputs 8 (l / 16) .. + 7), read by the inference kernel of the 256-wide skip models
// (mlp_fwd16_impl.hpp).  Same arithmetic as NERFX_PREC_BF16; the chip sustains a higher clock on
// this MFMA shape.  Functions below take it wherever they take a precision.
constexpr int NERFX_LAYOUT_M16 = 2;
// A fourth stream, for bf16 inference of the 256-wide skip models whose encoders are wider than the
// kernels' register cache (Fourier features): NERFX_PREC_BF16 units, but a layer with an encoder
// segment carries the encoder units of ALL its tile groups first, chunk by chunk (ENC_BLOCK k-units
// x every group), then the chained units group by group.  The kernel evaluates a chunk of phi once
// and multiplies it into every accumulator tile of the layer (linear_encmajor in
// mlp_fwd_impl.hpp); in the plain order a chunk is evaluated once per tile group.
constexpr int NERFX_STREAM_ENCMAJOR = 3;
constexpr int N_LAYOUTS = 4;
// precision / unit shape of a stream
__host__ __device__ constexpr int layout_prec(int layout) { return layout == NERFX_STREAM_ENCMAJOR ? NERFX_PREC_BF16 : layout; }

// slots (input features) per k-unit and k-units per 32-feature activation tile
__host__ __device__ constexpr int slots_per_unit(int prec) { return prec == NERFX_LAYOUT_M16 ? 32 : prec == NERFX_PREC_BF16 ? 16 : 8; }
__host__ __device__ constexpr int units_per_tile(int prec) { return prec == NERFX_LAYOUT_M16 ? 1 : prec == NERFX_PREC_BF16 ? 2 : 4; }
// output rows of one unit (= of one accumulator tile) and lanes that hold the same row group
__host__ __device__ constexpr int rows_per_tile(int prec) { return prec == NERFX_LAYOUT_M16 ? 16 : 32; }
// encoder k-units come in multiples of this (64 slots in both bf16 layouts)
__host__ __device__ constexpr int enc_block(int prec) { return prec == NERFX_LAYOUT_M16 ? 2 : 4; }

// Encoder slot order used inside the kernels.  A positional encoder first has its 3 raw
// coordinates (and one zero) in slots 0..3 -- a fixed place whatever the number of
// frequencies -- then, for both kinds, sin/cos pairs: pair-slot k = slot / 2 holds
// sin (even slot) and cos (odd slot) of the angle of pair k - raw_slots / 2.
// enc_slot_col returns the column of the REFERENCE layout (features.py:32-36, :52) held by
// `slot`, or -1 for padding.
__host__ __device__ constexpr int enc_raw_slots(int kind) { return kind == NERFX_ENC_POSITIONAL ? 4 : 0; }
__host__ __device__ inline int enc_slot_col(int kind, int n, int slot) {
    const int R = enc_raw_slots(kind);
    if (slot < R) return slot < 3 ? slot : -1;
    const int P = kind == NERFX_ENC_POSITIONAL ? 3 * n : n;
    const int q = (slot - R) >> 1, which = (slot - R) & 1;
    if (q >= P) return -1;
    if (kind == NERFX_ENC_POSITIONAL) { const int f = q / 3, c = q - 3 * f; return 3 + 6 * f + 3 * which + c; }
    return which * n + q;
}
// k-units an encoder occupies (padded with zero weights to a multiple of ENC_BLOCK).
__host__ __device__ inline int enc_units(int kind, int n, int prec) {
    const int slots = enc_raw_slots(kind) + (kind == NERFX_ENC_POSITIONAL ? 6 * n : 2 * n);
    const int units = (slots + slots_per_unit(prec) - 1) / slots_per_unit(prec);
    return (units + enc_block(prec) - 1) / enc_block(prec) * enc_block(prec);
}
// Entries of an encoder's pair-coefficient table (one float4 per pair-slot, covering the
// padded units of either precision): angle of pair-slot k = c[k] . v' with v' = v (positional,
// c = frequency on one axis) or 2 pi v (Fourier, c = b row).
__host__ __device__ inline int enc_pair_slots(int kind, int n) { return enc_units(kind, n, NERFX_PREC_BF16) * 8; }
// Encoder slot supplied by lane group h (lane / rows_per_tile) as element e of k-unit u.
__host__ __device__ inline int enc_unit_slot(int prec, int u, int h, int e) {
    const int spu = slots_per_unit(prec);
    return spu * u + (spu * rows_per_tile(prec) / 64) * h + e;
}
// Feature (row of the previous layer's 32x32 accumulator tiles) supplied by lane
// half h as element e of chained k-unit u: the accumulator registers of a
// v_mfma_f32_32x32x* result, reused in place as the next B operand.
__host__ __device__ inline int chain_unit_feature(int prec, int u, int h, int e) {
    // 16x16 accumulators: lane group h of tile T holds rows 16 T + 4 h + r; a k-unit is two tiles
    if (prec == NERFX_LAYOUT_M16) return 32 * u + 16 * (e >> 2) + 4 * h + (e & 3);
    if (prec == NERFX_PREC_BF16) return 16 * u + 8 * (e >> 2) + 4 * h + (e & 3);
    return 8 * u + 4 * h + e;
}

// Output tiles a layer computes at a time: bounded by the accumulator registers that
// fit beside the input and output fragments.
__host__ __device__ constexpr int group_tiles(int prec, int n_tiles) {
    const int cap = prec == NERFX_PREC_BF16 ? 2 : 4;       // (four 16-row tiles in the M16 layout)
    return n_tiles < cap ? n_tiles : cap;
}
// Whether the training tape of this precision / activation carries sign words (Tape::signs):
// bf16 with an activation whose slope depends on the sign of z only.
__host__ __device__ constexpr bool tape_has_signs(int prec, int act) {
    return prec == NERFX_PREC_BF16 && (act == NERFX_ACT_RELU || act == NERFX_ACT_LEAKY_RELU);
}
__host__ __device__ constexpr int pad_to_piece(int units) {
    return (units + PIECE_UNITS - 1) / PIECE_UNITS * PIECE_UNITS;
}

// Layouts whose streams carry rgb.0 with the `features` layer folded in (see LayerPlan::folded): all
// bf16 ones.  fp32 is the exact mode and keeps the reference's two layers: the folded weight rounds once
// more, and a 4,096-ray training step's gradients, sums that cancel to 1e-5 of their terms, showed it
// (9e-4 of a gradient's largest entry against a float64 evaluation, where the unfolded fp32 step has 2e-4).
__host__ __device__ constexpr bool folds_features(int layout) { return layout != NERFX_PREC_FP32; }

// One Linear layer as the kernels see it.
struct LayerPlan {
    int n_tiles;        // output tiles of rows_per_tile rows
    int out_dim;        // true rows
    int enc_units;      // leading k-units fed by an encoder (0 = none)
    int enc_kind, enc_n, enc_cols;   // that encoder and its width in the reference layout
    int chain_tiles;    // 32-feature input tiles fed by the previous layer (0 for the first)
    int chain_dim;      // true chained inputs
    int in_dim;         // leading dimension of the reference weight (enc_cols + chain_dim)
    int group;          // NG
    int enc_major;      // NERFX_STREAM_ENCMAJOR: all groups' encoder units first, chunk-major (see above)
    int enc_seg;        // units of one group's encoder segment (padded to pieces)
    int chain_seg;      // units of one group's chained segment (padded to pieces)
    int64_t unit0;      // first unit of this layer in the stream
    // Backward-stream layers carry W^T of forward layer `src_layer`: row i, input k is
    // W[k][src_col0 + i] of a weight with leading dimension src_ld.
    int transposed, src_layer, src_ld, src_col0;
    // `features` has no activation (model.py:51, :76), so the bf16 kernels do not evaluate it: it is folded
    // into rgb.0 when their streams are packed (folds_features) -- rgb.0(phi_d, W_f h + b_f) = W_d phi_d + (W_r W_f) h +
    // (W_r b_f + b_r).  `absent` marks the layer that is in no stream (and has no tape tiles),
    // `folded` the one whose weight is read from the folded copy (nerfx_model::fold_dev).  Training
    // differentiates the folded form and unfolds the two layers' gradients afterwards (mlp_wgrad.hip).
    int absent, folded;
};

struct ModelPlan {
    nerfx_model_desc_t desc;
    int prec;
    int wt;             // hidden tiles (power of two): 32 * wt >= width
    int rt;             // rgb.0 output tiles
    int hd;             // layers per trunk half
    int n_layers;       // 2*hd + 3
    LayerPlan layer[MAX_LAYERS];
    int64_t trunk_units;   // stream units up to and including the last trunk layer
    int64_t total_units;   // whole stream (M16 layout: with the padding of m16_static, if any)
    int m16_static;        // M16 layout: both stream lengths are whole ring lengths (see M16_RING_UNITS)
    // const block (floats): [layer biases | sigma w | sigma b | table x | table d]
    int sigw_off, sigb_off, tabx_off, tabd_off, const_floats;
    int foldb_off;         // const block: rgb.0's bias with the `features` layer folded in (M16 kernel)
    // Backward stream: the input-gradient chain rgb.2 -> rgb.0 -> features (fp32 only) -> trunk (last .. 1),
    // each layer the transposed chained block of its forward layer.
    int n_bwd;
    LayerPlan bwd[MAX_LAYERS];
    int64_t bwd_units;
    // Training tape tiles (see Tape in mlp_fwd_impl.hpp).  The pre-activation tape and the
    // gradient tape share one tile map: encoder slots of x and of d (32 per tile), then the output
    // tiles of every layer in the streams but rgb.2, then one "heads" tile (gradient tape only) whose positions
    // 0..2 are the gradients at rgb.2's outputs and position 3 the one at the density head's
    // pre-activation.
    int tape_ex_tile, tape_ed_tile, tape_z_tile[MAX_LAYERS], tape_heads_tile, tape_tiles;
};

__host__ __device__ inline int units_of(const LayerPlan& L) { return (L.n_tiles / L.group) * (L.enc_seg + L.chain_seg); }

// Fills `p` from a model description for stream layout `layout`; returns 0 or an error code.
int make_plan(const nerfx_model_desc_t& d, int layout, ModelPlan& p);
// Models the encoder-major stream is packed for: 256-wide (8 tiles), skip connection, depth 4 or 8,
// an encoder wider than ENC_BLOCK bf16 k-units.
inline bool encmajor_supported(const nerfx_model_desc_t& d) {
    return d.width > 128 && d.width <= 256 && d.resid && (d.depth == 4 || d.depth == 8) &&
           (enc_units(d.enc_x_kind, d.enc_x_n, NERFX_PREC_BF16) > ENC_BLOCK || enc_units(d.enc_d_kind, d.enc_d_n, NERFX_PREC_BF16) > ENC_BLOCK);
}
// The M16 kernel's weight ring, in 1 KiB units.  A stream whose length is a whole number of ring
// lengths puts every unit of a sample block at the same ring position in every block, which makes
// all of the kernel's ring arithmetic compile-time constants (static mode).  The depth-4 streams fit:
// 448 units up to the density head = 4 rings; 536 with the colour layers (rgb.0 with `features` folded
// in, rgb.2), padded to 560 = 5 rings.
constexpr int M16_RING_UNITS = 112;
constexpr int M16_STATIC_PAD = 24;              // padding units behind rgb.2 in static mode
__host__ __device__ constexpr int m16_trunk_units(int hd) { return 32 + (2 * hd - 1) * 128 + 32; }
__host__ __device__ constexpr int m16_total_units(int hd) { return m16_trunk_units(hd) + 80 + 8; }
__host__ __device__ constexpr bool m16_static_depth(int hd) {
    return m16_trunk_units(hd) % M16_RING_UNITS == 0 && (m16_total_units(hd) + M16_STATIC_PAD) % M16_RING_UNITS == 0;
}
static_assert(m16_static_depth(2), "the depth-4 model streams are whole ring lengths");
// Models the M16 inference kernel takes: 129..256 wide, skip connection, depth 4 or 8, encoders
// of at most 64 slots (positional up to 10 frequencies, Fourier up to 32 features).
inline bool m16_supported(const nerfx_model_desc_t& d) {
    return d.width > 128 && d.width <= 256 && d.resid && (d.depth == 4 || d.depth == 8) &&
           enc_units(d.enc_x_kind, d.enc_x_n, NERFX_LAYOUT_M16) <= enc_block(NERFX_LAYOUT_M16) &&
           enc_units(d.enc_d_kind, d.enc_d_n, NERFX_LAYOUT_M16) <= enc_block(NERFX_LAYOUT_M16);
}

// Samples per tape tile for n_points samples: whole workgroup blocks of either kernel family
// and whole sample chunks of the weight-gradient kernel.
inline uint64_t tape_m_pad(int64_t n_points) { return ((uint64_t)n_points + 255) / 256 * 256; }

}  // namespace nerfx

#include "narrow.hpp"

// The opaque handle of include/nerfx.h.
struct nerfx_model {
    nerfx::ModelPlan plan[nerfx::N_LAYOUTS];          // indexed by NERFX_PREC_* / NERFX_LAYOUT_M16
    void* stream_dev[nerfx::N_LAYOUTS];               // packed unit streams
    bool has_m16;                      // the M16 stream exists (see m16_supported)
    bool has_encmajor;                 // the encoder-major stream exists (see encmajor_supported)
    bool has_layout(int l) const { return l == nerfx::NERFX_LAYOUT_M16 ? has_m16 : l == nerfx::NERFX_STREAM_ENCMAJOR ? has_encmajor : true; }
    void* stream_bwd_dev[2];           // packed backward (transposed) unit streams
    nerfx::LayerPlan* plans_dev[nerfx::N_LAYOUTS];    // forward then backward layer plans (read by the pack kernel)
    void* wtasks_dev[2];               // weight-gradient task tables (mlp_wgrad.hip)
    int n_wtasks[2];
    float* const_dev;                  // const block (same for both precisions)
    // rgb.0 with `features` folded in (fp32, written by every refresh): weight [width / 2][in_dim], bias
    // [width / 2]; then what unfolding the gradients needs of the weights as they were at that refresh:
    // W_f [width][width], b_f [width], W_r's chained columns [width / 2][width]; then the folded layer's
    // gradient (weight, bias) as the weight-gradient kernels leave it.
    float* fold_dev;
    struct FoldSpots { float *w, *b, *wft, *bf, *wr, *gw, *gb; int64_t floats; };
    FoldSpots fold_spots() const {
        const nerfx::LayerPlan& R = plan[0].layer[2 * plan[0].hd + 1];
        const int64_t half = R.out_dim, width = R.chain_dim, ld = R.in_dim;
        FoldSpots s{};
        int64_t at = 0;
        auto take = [&](int64_t n) { float* q = fold_dev ? fold_dev + at : nullptr; at += n; return q; };
        s.w = take(half * ld);
        s.b = take(half);
        s.wft = take(width * width);
        s.bf = take(width);
        s.wr = take(half * width);
        s.gw = take(half * ld);
        s.gb = take(half);
        s.floats = at;
        return s;
    }
    // Two words of NAN_* bits (model.hip): refresh number e records in word e & 1 which outputs a NaN
    // parameter reaches and zeroes the other word for refresh e + 1.  Every refresh reads every
    // parameter (each forward stream holds all layers), so the word of the latest refresh describes
    // the current weights, whichever precision it packed.  nan_flag() is what the kernels are given.
    uint32_t* nan_flags_dev;
    uint32_t pack_epoch;
    const uint32_t* nan_flag() const { return nan_flags_dev + (pack_epoch & 1); }
    nerfx::NarrowPlan narrow;          // narrow.hpp: models the whole-ray kernel takes (narrow.ok), and their
    void* narrow_dev;                  // packed block
    int n_tensors;
    int device;
};
