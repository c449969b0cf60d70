// A whole render pass of a NARROW radiance field in one kernel (bf16 MFMA): sample positions ->
// encoding -> every layer -> density / colour heads -> alpha compositing, per ray, with nothing
// per-sample written to memory except the weights the API returns (reference renderer.py:88-97,
// :138-142, :203 with model.py:68-78 inside).  See narrow.hpp for the models it takes.
//
// A wavefront owns whole rays.  It walks a ray in blocks of 32 samples: sample s of the block
// is column s of every MFMA (lanes s and s + 32), all weights sit in registers as A operands,
// a layer's 32 x 32 accumulator tile becomes the next layer's B operand in place, the density
// head is one more product on the same operands, and the compositor's transmittance is a product scan over
// the 32 lanes added to the sum carried from the ray's earlier blocks.  phi(d) and its share of
// the colour layer (W_rgb0[:, :E_d] phi(d) + b) are formed once per ray and enter that layer as
// the accumulator's initial value; `features`, which has no activation, is multiplied into rgb.0 when
// the weights are packed.  No LDS ring, no barrier after the set-up, no HBM traffic for densities and
// colours: 8 bytes per sample (z value in, weight out) and 40 per ray instead of 16 more per sample
// through sigma / rgb.  The kernel is bound by vector-instruction issue (about 180 per block of 32
// samples with the colour head, 20 of them sines and cosines), not by the matrix pipes or memory.
#include <atomic>

#include "mlp_fwd_impl.hpp"

namespace nerfx {

// ---- packing -----------------------------------------------------------------------------
struct NarrowSources { const float* t[14]; };    // tensor list of a depth-2 model (nerfx_model_create's order)
enum { NS_TABX = 0, NS_TABD, NS_W0, NS_B0, NS_W1, NS_B1, NS_WS, NS_BS, NS_WF, NS_BF, NS_WR0, NS_BR0, NS_WR2, NS_BR2 };

__device__ __forceinline__ float narrow_clean(float v) { return v != v ? __builtin_bit_cast(float, 0x7FC00000u) : v; }

__global__ __launch_bounds__(256) void pack_narrow_kernel(const NarrowPlan p, const NarrowSources s, char* __restrict__ out) {
    const int n_slices = p.n_units * 64;
    const int W = p.width, H = p.half;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_slices + p.n_floats; i += gridDim.x * 256) {
        if (i < n_slices) {
            const int unit = i >> 6, lane = i & 63, row = lane & 31, h = lane >> 5;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = 0.0f;
                if (unit < p.u_l1) {                                             // layer 0: encoder -> W
                    const int col = narrow_slot_col(p.kx, p.nx, unit - p.u_l0, h, e);
                    if (row < W && col >= 0) x = s.t[NS_W0][row * p.ex_cols + col];
                } else if (unit >= p.u_rgb0_enc) {                               // rgb.0, the phi(d) columns
                    const int col = narrow_slot_col(p.kd, p.nd, unit - p.u_rgb0_enc, h, e);
                    if (row < H && col >= 0) x = s.t[NS_WR0][row * (p.ed_cols + W) + col];
                } else {
                    const int base = unit < p.u_sigma ? p.u_l1 : unit < p.u_feat ? p.u_sigma : unit < p.u_rgb0 ? p.u_feat : unit < p.u_rgb2 ? p.u_rgb0 : p.u_rgb2;
                    const int f = chain_unit_feature(NERFX_PREC_BF16, unit - base, h, e);
                    if (base == p.u_l1) { if (row < W && f < W) x = s.t[NS_W1][row * W + f]; }
                    else if (base == p.u_sigma) {                                // row 0: bf16(w), row 1: bf16(w - bf16(w))
                        if (row < 2 && f < W) {
                            const float w = narrow_clean(s.t[NS_WS][f]), hi = (float)(__bf16)w;
                            x = row == 0 ? hi : w - hi;
                        }
                    }
                    else if (base == p.u_feat) { x = 0.0f; }                     // (folded into rgb.0 below)
                    else if (base == p.u_rgb0) {
                        // `features` has no activation (model.py:51, :76), so rgb.0 sees W_rgb0[:, E_d:] (W_f h + b_f):
                        // the two matrices are multiplied here, once per refresh, instead of once per sample
                        if (row < H && f < W)
                            for (int j = 0; j < W; ++j)
                                x = fmaf(s.t[NS_WR0][row * (p.ed_cols + W) + p.ed_cols + j], s.t[NS_WF][j * W + f], x);
                    }
                    else { if (row < 3 && f < H) x = s.t[NS_WR2][row * H + f]; }
                }
                v[e] = narrow_clean(x);
            }
            __bf16 b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] = (__bf16)v[e];
            *reinterpret_cast<uint4*>(out + (size_t)i * 16) = *reinterpret_cast<uint4*>(b);
        } else {
            const int j = i - n_slices;
            float x = 0.0f;
            if (j < p.f_bias) {                                                  // pair tables, in revolutions (see narrow.hpp)
                const bool dirs = j >= p.f_tabd;
                const int q = j - (dirs ? p.f_tabd : p.f_tabx), slot = q & 7, h = (q >> 3) & 1;
                const int pos = 2 * (q >> 4) + (slot & 1), c = slot < 4 ? slot >> 1 : (slot < 6 ? 2 : 3);
                const int kind = dirs ? p.kd : p.kx, n = dirs ? p.nd : p.nx;
                const float* tab = s.t[dirs ? NS_TABD : NS_TABX];
                const int k = narrow_pair_index(kind, pos, h);
                if (k >= 0 && k < enc_pairs(kind, n) && c < 3) {
                    if (kind == NERFX_ENC_POSITIONAL) x = (k % 3 == c) ? tab[k / 3] * 0.15915494309189535f : 0.0f;
                    else x = tab[3 * k + c];                                     // 2 pi v . b is v . b revolutions
                }
            } else if (j < p.f_sigw) {
                const int q = j - p.f_bias, layer = q >> 5, h = (q >> 4) & 1, r = q & 15, row = acc_row(r, h);
                const int dims[NARROW_BIASES] = {W, W, W, H, 3};
                const int src[NARROW_BIASES] = {NS_B0, NS_B1, NS_BF, NS_BR0, NS_BR2};
                if (row < dims[layer]) x = s.t[src[layer]][row];
                if (layer == NARROW_BIAS_RGB0 && row < H)                        // + W_rgb0[:, E_d:] b_f (see the weights)
                    for (int jj = 0; jj < W; ++jj) x = fmaf(s.t[NS_WR0][row * (p.ed_cols + W) + p.ed_cols + jj], s.t[NS_BF][jj], x);
            } else if (j < p.f_sigb) {
                const int q = j - p.f_sigw, row = acc_row(q & 15, q >> 4);
                if (row < W) x = s.t[NS_WS][row];
            } else if (j == p.f_sigb) {
                x = s.t[NS_BS][0];
            } else if (j == p.f_cmax) {                                          // bounds the angles: see the kernel's WRAP
                const int entries = p.kx == NERFX_ENC_POSITIONAL ? p.nx : 3 * p.nx;
                for (int k = 0; k < entries; ++k) {
                    const float c = fabsf(s.t[NS_TABX][k]) * (p.kx == NERFX_ENC_POSITIONAL ? 0.15915494309189535f : 1.0f);
                    x = (c > x || c != c) ? c : x;                                // a NaN entry stays: the bound then fails
                }
            }
            reinterpret_cast<float*>(out + (size_t)n_slices * 16)[j] = narrow_clean(x);
        }
    }
}

int narrow_pack(const NarrowPlan& p, const float* const* tensors, void* dst, hipStream_t stream) {
    NarrowSources s{};
    for (int i = 0; i < 14; ++i) s.t[i] = tensors[i];
    hipLaunchKernelGGL(pack_narrow_kernel, dim3(8), dim3(256), 0, stream, p, s, (char*)dst);
    return launch_status("pack_narrow_kernel");
}

// ---- the render pass ------------------------------------------------------------------------
struct NarrowArgs {
    const char* pack;
    NarrowPlan p;
    const uint32_t* nan_flag;
    const float *ro, *rd, *t;                   // (B,3), (B,3), (B,n)
    float* t_out;                               // t == NULL: the z values are linspace(tn, tf, n) (ray.py:91-92), made
    float tn, tf;                               // here and written to t_out (B,n)
    int64_t n_rays;
    int n;
    float *weights, *rgb_map, *depth;           // (B,n); (B,3), (B) with the colour head
};

using P16 = Prec<NERFX_PREC_BF16>;
using lds_f4_t = const __attribute__((address_space(3))) f32x4*;

// ---- cross-lane steps on the data-parallel-primitive path (no LDS round trip) ---------------------
// Lane l gets `v` of the lane the control names; lanes whose source lies outside their row of 16 (or
// that the row mask leaves out) get `identity`.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
constexpr int DPP_ROW_SHR = 0x110, DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142;
// Inclusive sum over each half of the wavefront (lanes 0..31 and 32..63 separately); the half's total is
// in its last lane.
__device__ __forceinline__ float half_wave_sum_scan(float v) {
    v += dpp<DPP_ROW_SHR + 1>(0.0f, v);
    v += dpp<DPP_ROW_SHR + 2>(0.0f, v);
    v += dpp<DPP_ROW_SHR + 4>(0.0f, v);
    v += dpp<DPP_ROW_SHR + 8>(0.0f, v);
    v += dpp<DPP_ROW_BCAST15, 0xA>(0.0f, v);
    return v;
}

// B operands of an encoding of point (vx, vy, vz): PX pair-slot positions (compile time: straight-line
// code, so that the conversions pack in pairs), table in LDS.  Two positions per step, their angles formed
// with packed fp32 arithmetic.
// WRAP: reduce the angle to [0, 1) revolutions first; the hardware sine / cosine take +-256 revolutions as
// they are, so callers that have bounded the angles leave it out.
template <int PX, bool WRAP>
__device__ __forceinline__ void narrow_encode(lds_f4_t tab, bool positional, int h, float vx, float vy, float vz,
                                              bf16x8 (&B)[NARROW_MAX_UNITS]) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const f32x2 x2 = {vx, vx}, y2 = {vy, vy}, z2 = {vz, vz};
#pragma unroll
    for (int u = 0; u < NARROW_MAX_UNITS; ++u) {
        f32x8 v = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (4 * u < PX) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (4 * u + 2 * j < PX) {
                    const f32x4 a = tab[((2 * u + j) * 2 + h) * 2], b = tab[((2 * u + j) * 2 + h) * 2 + 1];
                    const f32x2 cx = {a[0], a[1]}, cy = {a[2], a[3]}, cz = {b[0], b[1]};
                    // (packed: 2.23 against 2.28 ms per frame with one fma chain per position)
                    const f32x2 rev = __builtin_elementwise_fma(z2, cz, __builtin_elementwise_fma(y2, cy, x2 * cx));
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if (4 * u + 2 * j + k < PX) {
                            const float r = WRAP ? __builtin_amdgcn_fractf(rev[k]) : rev[k];
                            v[4 * j + 2 * k] = __builtin_amdgcn_sinf(r); v[4 * j + 2 * k + 1] = __builtin_amdgcn_cosf(r);
                        }
                    }
                }
            }
            if (u == 0) {
                const bool raw = positional && h == 0;
                v[0] = raw ? vx : v[0]; v[1] = raw ? vy : v[1]; v[2] = raw ? vz : v[2]; v[3] = raw ? 0.0f : v[3];
            }
        }
        B[u] = __builtin_convertvector(v, bf16x8);
    }
}

// Accumulator registers 8 s .. 8 s + 7 -> B operand of chained k-unit s, through the hidden activation.
// Converted pair by pair: hipcc turns a whole-vector conversion of accumulator registers into one
// conversion and one byte permute per VALUE, a two-element one into one v_cvt_pk_bf16_f32 per PAIR.
template <int ACT>
__device__ __forceinline__ bf16x8 narrow_chain(const f32x16& acc, int s) {
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
    constexpr bool PACKED = ACT == NERFX_ACT_RELU || ACT < 0;       // ReLU after the conversion, on the packed pairs
    u32x4 w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float lo = acc[8 * s + 2 * q], hi = acc[8 * s + 2 * q + 1];
        const f32x2 v = {PACKED ? lo : activate<NERFX_PREC_BF16, ACT>(lo), PACKED ? hi : activate<NERFX_PREC_BF16, ACT>(hi)};
        w[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    }
    const bf16x8 b = __builtin_bit_cast(bf16x8, w);
    return ACT == NERFX_ACT_RELU ? P16::relu(b) : b;
}

__device__ __forceinline__ f32x16 narrow_bias(const __attribute__((address_space(3))) float* bias, int layer, int h) {
    f32x16 a;
    lds_f4_t p = (lds_f4_t)(bias + layer * 32 + h * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 q = p[g];
        a[4 * g] = q[0]; a[4 * g + 1] = q[1]; a[4 * g + 2] = q[2]; a[4 * g + 3] = q[3];
    }
    return a;
}

constexpr int NARROW_WAVES = 4;

// PX: pair-slot positions of the position encoder that are evaluated (>= the model's; table entries and
// weights beyond the model's are zero).
// GEN: the z values are the unperturbed coarse ones, formed in the kernel instead of read.
template <int ACT, bool COLOUR, int PX, bool GEN>
__global__ __launch_bounds__(64 * NARROW_WAVES, COLOUR ? 4 : 5) void narrow_render_kernel(const NarrowArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [floats | rgb.0's phi(d) units]
    const NarrowPlan& p = a.p;
    __attribute__((address_space(3))) float* fl = (__attribute__((address_space(3))) float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, col = lane & 31;
    const float* gfl = reinterpret_cast<const float*>(a.pack + (size_t)p.n_units * UNIT_BYTES);
    const uint32_t poison = a.nan_flag ? *a.nan_flag : 0u;
    const float nan = __builtin_bit_cast(float, 0x7FC00000u);
    for (int i = tid; i < p.n_floats; i += 64 * NARROW_WAVES) {
        float v = gfl[i];
        if ((poison & 1u) && i == p.f_sigb) v = nan;                // a NaN parameter reaches this head
        if ((poison & 2u) && i >= p.f_bias + NARROW_BIAS_RGB2 * 32 && i < p.f_bias + NARROW_BIAS_RGB2 * 32 + 32) v = nan;
        fl[i] = v;
    }
    const int fl_bytes = (p.n_floats * 4 + 15) / 16 * 16;
    constexpr int LDS_UNITS = 3 * NARROW_CHAIN_UNITS + 1 + NARROW_MAX_UNITS;    // density head, features, rgb.0, rgb.2, rgb.0's phi(d) part
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.pack + (size_t)p.u_sigma * UNIT_BYTES);
        __attribute__((address_space(3))) f32x4* dst = (__attribute__((address_space(3))) f32x4*)(smem + fl_bytes);
        for (int i = tid; i < LDS_UNITS * 64; i += 64 * NARROW_WAVES) dst[i] = src[i];
    }
    __syncthreads();
    const lds_f4_t tabx0 = (lds_f4_t)(fl + p.f_tabx), tabd0 = (lds_f4_t)(fl + p.f_tabd);
    const __attribute__((address_space(3))) float* bias0 = fl + p.f_bias;
    // this lane's A operands of the layers behind layer 1, read from LDS where they are used (one wide
    // read each, LDS is nearly idle) so that four wavefronts fit a SIMD; layers 0 and 1 stay in registers
    const __attribute__((address_space(3))) bf16x8* lds_a0 = (const __attribute__((address_space(3))) bf16x8*)(smem + fl_bytes) + lane;
    const int o_sigma = 0, o_rgb0 = (p.u_rgb0 - p.u_sigma) * 64,
              o_rgb2 = (p.u_rgb2 - p.u_sigma) * 64, o_rgb0_enc = (p.u_rgb0_enc - p.u_sigma) * 64;
    const bf16x8* A = reinterpret_cast<const bf16x8*>(a.pack) + lane;
    constexpr int UX = (PX + 3) / 4;
    bf16x8 A0[UX], A1[NARROW_CHAIN_UNITS];
#pragma unroll
    for (int u = 0; u < UX; ++u) A0[u] = A[(p.u_l0 + u) * 64];
#pragma unroll
    for (int u = 0; u < NARROW_CHAIN_UNITS; ++u) A1[u] = A[(p.u_l1 + u) * 64];
    const float sigb = fl[p.f_sigb] - 1.0f;                         // ShiftedSoftplus: softplus(x - 1), activations.py:33
    const float cmax = fl[p.f_cmax];
    const bool pos_x = p.kx == NERFX_ENC_POSITIONAL, pos_d = p.kd == NERFX_ENC_POSITIONAL;
    const int n = a.n, n_blocks = (n + 31) / 32;
    constexpr float LOG2E = 1.4426950408889634f;

    // A ray's origin, direction and first z values are requested while the ray before it is computed: a ray
    // is only two to six blocks long, and a round trip to memory takes about as long as two blocks.
    struct RayIn { float ox, oy, oz, dx, dy, dz, t_lo, t_hi, t0; };
    auto fetch = [&](int64_t ray) {
        RayIn r{};
        if (ray < a.n_rays) {
            r.ox = a.ro[3 * ray]; r.oy = a.ro[3 * ray + 1]; r.oz = a.ro[3 * ray + 2];
            r.dx = a.rd[3 * ray]; r.dy = a.rd[3 * ray + 1]; r.dz = a.rd[3 * ray + 2];
            if (GEN) {
                r.t_lo = a.tn; r.t_hi = a.tf;
                r.t0 = linspace_at(a.tn, a.tf, n, col < n ? col : n - 1);
            } else {
                r.t_lo = a.t[ray * n]; r.t_hi = a.t[ray * n + n - 1];
                r.t0 = col < n ? a.t[ray * n + col] : 0.0f;
            }
        }
        return r;
    };
    const int64_t stride = (int64_t)gridDim.x * NARROW_WAVES;
    RayIn ahead = fetch((int64_t)blockIdx.x * NARROW_WAVES + wave);
    for (int64_t ray = (int64_t)blockIdx.x * NARROW_WAVES + wave; ray < a.n_rays; ray += stride) {
        const RayIn in = ahead;
        ahead = fetch(ray + stride);
        const float ox = in.ox, oy = in.oy, oz = in.oz, dx = in.dx, dy = in.dy, dz = in.dz;
        const float dn = norm3(dx, dy, dz);
        const float* t_row = GEN ? nullptr : a.t + ray * n;
        float* w_row = a.weights + ray * n;
        float t_nxt = in.t0;
        // The largest angle of this ray, in revolutions: |c . x| <= cmax |x|_1 with |x|_1 <= |o|_1 + |t|_max |d|_1
        // (z values ascend; either end may be the larger).  Below the hardware's range no angle is wrapped.
        const float t_big = fmaxf(fabsf(in.t_lo), fabsf(in.t_hi));
        const bool bounded = cmax * (fabsf(ox) + fabsf(oy) + fabsf(oz) + t_big * (fabsf(dx) + fabsf(dy) + fabsf(dz))) < 250.0f;
        f32x16 yd;                                                  // W_rgb0[:, :E_d] phi(d) + b, once per ray
        if (COLOUR) {
            int fresh = 0;
            asm volatile("" : "+s"(fresh));
            bf16x8 Bd[NARROW_MAX_UNITS];
            narrow_encode<PX, true>(tabd0 + fresh, pos_d, h, dx, dy, dz, Bd);
            yd = narrow_bias(bias0 + fresh, NARROW_BIAS_RGB0, h);
#pragma unroll
            for (int u = 0; u < UX; ++u)
                yd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_a0[o_rgb0_enc + u * 64 + fresh], Bd[u], yd, 0, 0, 0);
        }
        float carry = 0.0f, sum_r = 0.0f, sum_g = 0.0f, sum_b = 0.0f, sum_d = 0.0f;
        auto blocks = [&](auto wrap_tag) {
          constexpr bool WRAP = decltype(wrap_tag)::value;
          for (int blk = 0; blk < n_blocks; ++blk) {
            // Tables, biases and the later layers' weights are re-read from LDS in every block (about 40
            // wide reads against ~250 vector instructions): hoisted out of the loops they would be 150
            // registers per lane.  The opaque zero keeps the compiler from hoisting them.
            int fresh = 0;
            asm volatile("" : "+s"(fresh));
            const lds_f4_t tabx = tabx0 + fresh;
            const __attribute__((address_space(3))) float* bias = bias0 + fresh;
            const __attribute__((address_space(3))) bf16x8* lds_a = lds_a0 + fresh;
            const int idx = 32 * blk + col;
            const float t = t_nxt;
            if (GEN) {
                t_nxt = linspace_at(a.tn, a.tf, n, idx + 32 < n ? idx + 32 : n - 1);
                if (idx < n && h == 0) __builtin_nontemporal_store(t, a.t_out + ray * n + idx);
            } else {
                t_nxt = idx + 32 < n ? t_row[idx + 32] : 0.0f;      // the next block's z values, ahead of their use
            }
            // ---- model.py:68-78 on the block's 32 samples
            bf16x8 Bx[NARROW_MAX_UNITS];
            narrow_encode<PX, WRAP>(tabx, pos_x, h, fmaf(t, dx, ox), fmaf(t, dy, oy), fmaf(t, dz, oz), Bx);
            f32x16 acc = narrow_bias(bias, NARROW_BIAS_L0, h);
#pragma unroll
            for (int u = 0; u < UX; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[u], Bx[u], acc, 0, 0, 0);
            bf16x8 Hc[NARROW_CHAIN_UNITS];
#pragma unroll
            for (int s = 0; s < NARROW_CHAIN_UNITS; ++s) Hc[s] = narrow_chain<ACT>(acc, s);
            acc = narrow_bias(bias, NARROW_BIAS_L1, h);
#pragma unroll
            for (int s = 0; s < NARROW_CHAIN_UNITS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[s], Hc[s], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NARROW_CHAIN_UNITS; ++s) Hc[s] = narrow_chain<ACT>(acc, s);
            // density head (sigma.0 + ShiftedSoftplus, model.py:75) as one more product: rows 0 and 1 are the two
            // halves of its weight vector (narrow.hpp); valid in lane half 0, which is the half that composites
            f32x16 head = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < NARROW_CHAIN_UNITS; ++s) head = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_a[o_sigma + s * 64], Hc[s], head, 0, 0, 0);
            const float sigma = P16::softplus((head[0] + head[1]) + sigb);
            float rgb[3] = {0.0f, 0.0f, 0.0f};
            if (COLOUR) {
                acc = yd;                                           // rgb.0 on features(h): one product with the pre-multiplied matrix
#pragma unroll
                for (int s = 0; s < NARROW_CHAIN_UNITS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_a[o_rgb0 + s * 64], Hc[s], acc, 0, 0, 0);
                Hc[0] = narrow_chain<ACT>(acc, 0);                  // at most 16 rows: one k-unit
                acc = narrow_bias(bias, NARROW_BIAS_RGB2, h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_a[o_rgb2], Hc[0], acc, 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 3; ++c)                         // WidenedSigmoid (activations.py:18-20); rows 0..2: lane half 0
                    rgb[c] = fmaf(1.002f, __builtin_amdgcn_rcpf(1.0f + __expf(-acc[c])), -0.001f);
            }
            // ---- renderer.py:50-57, :203 on the same 32 samples.  The transmittance is the reference's
            // running product of (1 - alpha + 1e-10) written as exp(-sum of sigma delta): the 1e-10 it
            // leaves out changes a weight by less than 1e-10, and the sum is a scan of plain additions.
            const bool valid = idx < n;
            float t_up = dpp<DPP_WAVE_SHL1>(0.0f, t);
            const float t_first_next = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_nxt), 0));
            if (col == 31) t_up = t_first_next;
            const float delta = (idx == n - 1 ? 1e10f : t_up - t) * dn;                 // ray.py:60-65
            const float od = valid ? sigma * delta : 0.0f;                              // optical depth of the sample
            const float inc = half_wave_sum_scan(od);
            const float alpha = 1.0f - __builtin_amdgcn_exp2f(-LOG2E * od);             // renderer.py:50
            // (the sum BEFORE the sample is the previous lane's inclusive sum, not inc - od: the last sample's
            // optical depth is ~1e10 and would swallow it)
            const float before = dpp<DPP_WAVE_SHR1>(0.0f, inc);
            const float T = __builtin_amdgcn_exp2f(-LOG2E * (carry + (col == 0 ? 0.0f : before)));   // renderer.py:51
            carry += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inc), 31));
            const float w = alpha * T;
            if (valid && h == 0) __builtin_nontemporal_store(w, w_row + idx);
            if (COLOUR) {
                sum_r = fmaf(w, rgb[0], sum_r); sum_g = fmaf(w, rgb[1], sum_g); sum_b = fmaf(w, rgb[2], sum_b);
                sum_d = fmaf(w, t, sum_d);
            }
          }
        };
        if (bounded) blocks(std::false_type{});
        else blocks(std::true_type{});
        if (COLOUR) {
            sum_r = half_wave_sum_scan(sum_r); sum_g = half_wave_sum_scan(sum_g);
            sum_b = half_wave_sum_scan(sum_b); sum_d = half_wave_sum_scan(sum_d);
            if (lane == 31) {
                a.rgb_map[3 * ray] = sum_r; a.rgb_map[3 * ray + 1] = sum_g; a.rgb_map[3 * ray + 2] = sum_b;
                a.depth[ray] = sum_d;
            }
        }
    }
}

static std::atomic<int> g_narrow{1};

template <bool COLOUR, bool GEN>
static int narrow_launch_act(const NarrowArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)(a.p.n_floats * 4 + 15) / 16 * 16 + (size_t)(3 * NARROW_CHAIN_UNITS + 1 + NARROW_MAX_UNITS) * UNIT_BYTES;
    int dev = 0, cus = 256;
    NERFX_HIP(hipGetDevice(&dev));
    NERFX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t want = (a.n_rays + NARROW_WAVES - 1) / NARROW_WAVES;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)cus * (COLOUR ? 4 : 5)));   // resident workgroups
    // positions evaluated: 10 covers positional encoders up to 6 frequencies, 16 everything the plan admits
#define NERFX_NARROW(ACT_)                                                                                                    \
    do {                                                                                                                      \
        if (a.p.pos_x <= 10 && a.p.pos_d <= 10) hipLaunchKernelGGL((narrow_render_kernel<ACT_, COLOUR, 10, GEN>), dim3(grid), dim3(64 * NARROW_WAVES), lds, stream, a); \
        else hipLaunchKernelGGL((narrow_render_kernel<ACT_, COLOUR, NARROW_MAX_POS, GEN>), dim3(grid), dim3(64 * NARROW_WAVES), lds, stream, a);      \
    } while (0)
    switch (a.p.activ) {
        case NERFX_ACT_RELU: NERFX_NARROW(NERFX_ACT_RELU); break;
        case NERFX_ACT_LEAKY_RELU: NERFX_NARROW(NERFX_ACT_LEAKY_RELU); break;
        default: NERFX_NARROW(NERFX_ACT_SILU); break;
    }
#undef NERFX_NARROW
    return launch_status("narrow_render_kernel");
}

bool narrow_enabled() { return g_narrow.load(std::memory_order_relaxed) != 0; }

// One pass of `model` over the z values `t` (B, n): weights (B, n) and, with `rgb_map` / `depth`, the colour head
// and the ray sums.  t == NULL (weights-only passes): the z values are linspace(tn, tf, n), written to t_out.
int narrow_render_pass(const nerfx_model* model, const float* ro, const float* rd, const float* t, float tn, float tf,
                       float* t_out, int64_t n_rays, int n, float* weights, float* rgb_map, float* depth, hipStream_t stream) {
    NERFX_REQUIRE(model && model->narrow.ok && model->narrow_dev, "not a narrow model");
    NERFX_REQUIRE(ro && rd && (t || (t_out && !rgb_map)) && weights && (!rgb_map == !depth) && n >= 1, "bad argument");
    if (n_rays == 0) return 0;
    NarrowArgs a{};
    a.pack = (const char*)model->narrow_dev;
    a.p = model->narrow;
    a.nan_flag = model->nan_flag();
    a.ro = ro; a.rd = rd; a.t = t; a.t_out = t_out; a.tn = tn; a.tf = tf; a.n_rays = n_rays; a.n = n;
    a.weights = weights; a.rgb_map = rgb_map; a.depth = depth;
    if (rgb_map) return narrow_launch_act<true, false>(a, stream);
    return t ? narrow_launch_act<false, false>(a, stream) : narrow_launch_act<false, true>(a, stream);
}

}  // namespace nerfx

// Diagnostic switch: keep narrow models on the general kernels (A/B timing, and tests that hold one
// path against the other).
extern "C" void nerfx_debug_set_narrow(int enable) { nerfx::g_narrow.store(enable ? 1 : 0, std::memory_order_relaxed); }
