// K2: the fused radiance-field MLP (encode -> trunk -> density head -> colour head).
//
// Orientation.  Every layer is computed TRANSPOSED, H^T = W . X^T, with the
// weights as the MFMA A operand (32 output features x k) and the samples as the
// 32 columns of the B operand.  A 32x32 accumulator tile then has its SAMPLE on
// the lane and its 16 FEATURES in registers, which is exactly what the next
// layer's B operand needs: activations never leave the register file between
// layers (v_mfma_f32_32x32x16_bf16 takes pairs of converted accumulator
// registers, v_mfma_f32_32x32x2_f32 takes the accumulator registers as they
// stand).  The price is a k order that is permuted per lane half; the weight
// packer (model.hip) bakes that permutation into the unit stream.
//
// Work split.  A wavefront owns 32 samples end to end; a workgroup is WAVES
// wavefronts that share the weights through LDS.  Weights arrive as a flat unit
// stream (mlp.hpp) through an LDS ring of 8 KiB pieces filled by global_load_lds,
// RING_AHEAD pieces ahead of the consumer.  A segment of k-units is handed over
// with one counted s_waitcnt + one s_barrier; workgroups are persistent and the
// stream wraps from one block of 32*WAVES samples into the next, so the ring never
// drains.
//
// The encoders are fused as producers of B operands: the sin/cos k-units of a
// sample are computed once per block into registers and consumed by layer 0, the
// skip layer and rgb.0; they are never stored.
#pragma once
#include "mlp.hpp"

namespace nerfx {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x8 = __attribute__((ext_vector_type(8))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using lds_bytes_t = __attribute__((address_space(3))) char*;          // explicit LDS pointers: the
using lds_floats_t = const __attribute__((address_space(3))) float*;  // ring must never go through flat loads
#define NERFX_LDS_VEC(T, p) (*reinterpret_cast<const __attribute__((address_space(3))) T*>(p))

struct MlpArgs {
    const char* stream;          // packed units
    const float* consts;         // const block (global)
    const float* ro;             // ray mode: (B,3); point mode: x (M,3)
    const float* rd;             // ray mode: (B,3); point mode: d (M,3)
    const float* t;              // ray mode: (B,n); point mode: NULL
    float* sigma;                // (M)
    float* rgb;                  // (M,3) or NULL (density only)
    uint32_t n_points;           // M
    uint32_t n_samples;          // samples per ray (ray mode)
    uint32_t n_blocks;
    uint32_t n_pieces;           // stream length for this launch, in pieces
    int hd, resid;
    int ex_kind, ex_n, ex_units, ed_kind, ed_n, ed_units;
    int sigw_off, sigb_off, tabx_off, tabd_off, const_floats;
};

// ---- LDS weight ring ---------------------------------------------------------------
template <int WAVES>
struct WeightRing {
    static constexpr int LPW = PIECE_UNITS / WAVES;         // global_load_lds per wave per piece
    static_assert(PIECE_UNITS % WAVES == 0, "a piece must split evenly over the waves");
    const char* src;             // this lane's source: stream + (wave * LPW) KiB + 16 lane
    lds_bytes_t lds;             // ring base
    uint32_t n_pieces, issue_piece, issue_slot;
    uint32_t piece;              // consumer position: ring slot of the current segment's first piece
    int wave, lane;

    __device__ __forceinline__ void issue() {
#pragma unroll
        for (int k = 0; k < LPW; ++k) {
            const char* g = src + ((size_t)issue_piece * PIECE_UNITS + k) * UNIT_BYTES;
            lds_bytes_t l = lds + (issue_slot * PIECE_UNITS + wave * LPW + k) * UNIT_BYTES;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                             (void __attribute__((address_space(3)))*)l, 16, 0, 0);
        }
        issue_piece = issue_piece + 1 == n_pieces ? 0 : issue_piece + 1;
        issue_slot = (issue_slot + 1) & (RING_PIECES - 1);
    }
    __device__ __forceinline__ void init(const char* s, lds_bytes_t l, uint32_t np, int w, int ln) {
        src = s + (size_t)(w * LPW) * UNIT_BYTES + ln * 16;
        lds = l; n_pieces = np; wave = w; lane = ln;
        issue_piece = 0; issue_slot = 0; piece = 0;
#pragma unroll
        for (int c = 0; c < RING_AHEAD; ++c) issue();
    }
    // Hand-over of the next N pieces: wait for this wave's share of them (everything but the
    // (RING_AHEAD - N) newest pieces has landed), meet the other waves -- their shares have
    // landed too and nobody reads anything older any more -- then refill the N slots that
    // fell out of the window.
    template <int N>
    __device__ __forceinline__ void acquire() {
        static_assert(N >= 1 && N <= SYNC_PIECES, "hand-over larger than the ring's slack");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING_AHEAD - N) * LPW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < N; ++i) issue();
    }
    template <int N>
    __device__ __forceinline__ void release() { piece = (piece + N) & (RING_PIECES - 1); }
    // Address of unit `rel` (compile-time) of the segment that starts at `piece`.
    __device__ __forceinline__ lds_bytes_t unit_ptr(int rel) const {
        const uint32_t slot = (piece + rel / PIECE_UNITS) & (RING_PIECES - 1);
        return lds + slot * (PIECE_UNITS * UNIT_BYTES) + (rel % PIECE_UNITS) * UNIT_BYTES + lane * 16;
    }
};

// ---- per-precision pieces ------------------------------------------------------------
template <int PREC> struct Prec;

template <> struct Prec<NERFX_PREC_BF16> {
    static constexpr int SPUH = 8;                          // elements a lane supplies per k-unit
    static constexpr int UPT = 2;
    using Frag = bf16x8;                                    // one B operand per k-unit
    static __device__ __forceinline__ Frag from_floats(const float (&v)[8]) {
        Frag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (__bf16)v[e];
        return f;
    }
    // accumulator registers 8s..8s+7 of tile `a` -> B operand of chained k-unit 2T+s
    static __device__ __forceinline__ Frag from_acc(const f32x16& a, int s) {
        Frag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (__bf16)a[8 * s + e];
        return f;
    }
    // ReLU on packed bf16: a negative float is a negative int16 (v_pk_max_i16 with 0)
    static constexpr bool PACKED_RELU = true;
    static __device__ __forceinline__ Frag relu(const Frag& f) {
        const s16x8 v = __builtin_bit_cast(s16x8, f);
        return __builtin_bit_cast(Frag, __builtin_elementwise_max(v, s16x8(0)));
    }
    static __device__ __forceinline__ void mma(f32x16& acc, lds_bytes_t unit, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(NERFX_LDS_VEC(bf16x8, unit), b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void sincos_pair(float angle, float& s, float& c) {
        float r = angle * 0.15915494309189535f;             // revolutions
        r = r - floorf(r);
        s = __builtin_amdgcn_sinf(r);
        c = __builtin_amdgcn_cosf(r);
    }
    static __device__ __forceinline__ float expneg(float x) { return __expf(-x); }
};

template <> struct Prec<NERFX_PREC_FP32> {
    static constexpr int SPUH = 4;
    static constexpr int UPT = 4;
    using Frag = f32x4;                                     // four B operands (k-steps of 2) per k-unit
    static __device__ __forceinline__ Frag from_floats(const float (&v)[4]) { return Frag{v[0], v[1], v[2], v[3]}; }
    // accumulator registers 4g..4g+3 of tile `a` -> B operands of chained k-unit 4T+g
    static __device__ __forceinline__ Frag from_acc(const f32x16& a, int g) {
        return Frag{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
    }
    static constexpr bool PACKED_RELU = false;
    static __device__ __forceinline__ Frag relu(const Frag& f) { return f; }
    static __device__ __forceinline__ void mma(f32x16& acc, lds_bytes_t unit, const Frag& b) {
        const f32x4 a = NERFX_LDS_VEC(f32x4, unit);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void sincos_pair(float angle, float& s, float& c) { sincosf(angle, &s, &c); }
    static __device__ __forceinline__ float expneg(float x) { return expf(-x); }
};

// B operand of encoder k-unit u for this lane's point (vx, vy, vz): see enc_unit_slot().
template <int PREC>
__device__ __forceinline__ typename Prec<PREC>::Frag encode_unit(int kind, int n, lds_floats_t table, int u,
                                                                 int h, float vx, float vy, float vz) {
    using P = Prec<PREC>;
    constexpr int SPUH = P::SPUH;
    const int pairs = enc_pairs(kind, n);
    float v[SPUH];
#pragma unroll
    for (int i = 0; i < SPUH / 2; ++i) {
        const int q = SPUH * u + (SPUH / 2) * h + i;         // pair index = slot / 2
        if (q < pairs) {
            P::sincos_pair(enc_angle(kind, table, q, vx, vy, vz), v[2 * i], v[2 * i + 1]);
        } else {
            const int c = 2 * (q - pairs);                   // raw coordinates follow the pairs
            const bool pe = kind == NERFX_ENC_POSITIONAL;
            v[2 * i] = (pe && c < 3) ? (c == 0 ? vx : (c == 1 ? vy : vz)) : 0.0f;
            v[2 * i + 1] = (pe && c + 1 < 3) ? (c + 1 == 1 ? vy : vz) : 0.0f;
        }
    }
    return P::from_floats(v);
}

// An encoder's k-units for one sample: the first ENC_BLOCK live in registers (that covers
// positional encodings up to 10 frequencies); wider encoders are re-evaluated on use.
template <int PREC>
struct Encoding {
    typename Prec<PREC>::Frag cached[ENC_BLOCK];
    int kind, n, units;
    lds_floats_t table;
    float vx, vy, vz;
    int h;
    template <bool BIGENC>
    __device__ __forceinline__ void prepare(int k, int nn, int nu, lds_floats_t tb, int hh, float x, float y, float z) {
        kind = k; n = nn; units = nu; table = tb; h = hh; vx = x; vy = y; vz = z;
        if (!BIGENC) {
#pragma unroll
            for (int u = 0; u < ENC_BLOCK; ++u) cached[u] = encode_unit<PREC>(kind, n, table, u, h, vx, vy, vz);
        }
    }
    __device__ __forceinline__ typename Prec<PREC>::Frag eval(int u) const {
        return encode_unit<PREC>(kind, n, table, u, h, vx, vy, vz);
    }
};

template <int PREC, int ACT>
__device__ __forceinline__ float activate(float x) {
    if (ACT == NERFX_ACT_RELU) return fmaxf(x, 0.0f);
    if (ACT == NERFX_ACT_LEAKY_RELU) return x > 0.0f ? x : 0.01f * x;
    if (ACT == NERFX_ACT_SILU) return x / (1.0f + Prec<PREC>::expneg(x));
    return x;                                                // ACT < 0: plain Linear
}

// ---- one Linear layer -------------------------------------------------------------------
// out = act(W . [encoder units | chained input] + b), computed NG output tiles at a time so
// that only NG accumulator tiles are live beside the input and output fragments.
//   hin  : B operands of the chained input (CT * UPT fragments used)
//   hout : receives the activated output as next-layer B operands (NT * UPT fragments)
//   with_enc: this layer's stream carries an encoder segment (wave-uniform)
//   SIGMA: also accumulate the density head's dot product on the fp32 activations
//   raw  : if non-null (NT == 1 only) receives the fp32 accumulator tile instead of hout
template <int PREC, int WAVES, int ACT, bool BIGENC, int NT, int CT, bool SIGMA, int HN_IN, int HN_OUT>
__device__ __forceinline__ float linear(WeightRing<WAVES>& ring, lds_floats_t bias, int h,
                                        const typename Prec<PREC>::Frag (&hin)[HN_IN],
                                        typename Prec<PREC>::Frag (&hout)[HN_OUT],
                                        bool with_enc, const Encoding<PREC>& enc, lds_floats_t sigw,
                                        f32x16* raw = nullptr) {
    using P = Prec<PREC>;
    constexpr int NG = group_tiles(PREC, NT);
    constexpr int KB_PER_PIECE = PIECE_UNITS / NG;              // k-blocks (NG units each) per piece
    constexpr int CHAIN_KU = CT * P::UPT;
    constexpr int CHAIN_PIECES = pad_to_piece(CHAIN_KU * NG) / PIECE_UNITS;
    constexpr int ENC_PIECES = pad_to_piece(ENC_BLOCK * NG) / PIECE_UNITS;   // per ENC_BLOCK k-units
    static_assert(CHAIN_KU <= HN_IN && NT * P::UPT <= HN_OUT, "fragment arrays too small");
    float dot = 0.0f;
#pragma unroll
    for (int grp = 0; grp < NT / NG; ++grp) {
        f32x16 acc[NG];
        // bias as the initial accumulator: register r of a tile is feature (r&3) + 8 (r>>2) + 4 h
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            lds_floats_t bp = bias + 32 * (grp * NG + t) + 4 * h;
            const f32x4 b0 = NERFX_LDS_VEC(f32x4, bp), b1 = NERFX_LDS_VEC(f32x4, bp + 8),
                        b2 = NERFX_LDS_VEC(f32x4, bp + 16), b3 = NERFX_LDS_VEC(f32x4, bp + 24);
            const f32x8 lo = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            const f32x8 hi = __builtin_shufflevector(b2, b3, 0, 1, 2, 3, 4, 5, 6, 7);
            acc[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        }
        if (with_enc) {
            if (!BIGENC) {                                         // register-resident encoding
                ring.template acquire<ENC_PIECES>();
#pragma unroll
                for (int u = 0; u < ENC_BLOCK; ++u)
#pragma unroll
                    for (int t = 0; t < NG; ++t) P::mma(acc[t], ring.unit_ptr(u * NG + t), enc.cached[u]);
                ring.template release<ENC_PIECES>();
            } else {
                for (int u0 = 0; u0 < enc.units; u0 += ENC_BLOCK) {
                    ring.template acquire<ENC_PIECES>();
#pragma unroll
                    for (int u = 0; u < ENC_BLOCK; ++u) {
                        const typename P::Frag b = enc.eval(u0 + u);
#pragma unroll
                        for (int t = 0; t < NG; ++t) P::mma(acc[t], ring.unit_ptr(u * NG + t), b);
                    }
                    ring.template release<ENC_PIECES>();
                }
            }
        }
        // chained input, handed over SYNC_PIECES pieces at a time
#pragma unroll
        for (int p0 = 0; p0 < CHAIN_PIECES; p0 += SYNC_PIECES) {
            constexpr int dummy = 0; (void)dummy;
            if (CHAIN_PIECES - p0 >= SYNC_PIECES) ring.template acquire<SYNC_PIECES>();
            else ring.template acquire<(CHAIN_PIECES % SYNC_PIECES) ? (CHAIN_PIECES % SYNC_PIECES) : SYNC_PIECES>();
#pragma unroll
            for (int kb = 0; kb < SYNC_PIECES * KB_PER_PIECE; ++kb) {
                const int ku = p0 * KB_PER_PIECE + kb;
                if (ku < CHAIN_KU) {
#pragma unroll
                    for (int t = 0; t < NG; ++t) P::mma(acc[t], ring.unit_ptr(kb * NG + t), hin[ku]);
                }
                if ((kb & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
            if (CHAIN_PIECES - p0 >= SYNC_PIECES) ring.template release<SYNC_PIECES>();
            else ring.template release<(CHAIN_PIECES % SYNC_PIECES) ? (CHAIN_PIECES % SYNC_PIECES) : SYNC_PIECES>();
        }
        // epilogue: activation, density dot, conversion to the next layer's operands
        constexpr bool packed_relu = ACT == NERFX_ACT_RELU && P::PACKED_RELU && !SIGMA;
        if (!packed_relu) {
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = activate<PREC, ACT>(acc[t][r]);
        }
        if (SIGMA) {
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = NERFX_LDS_VEC(f32x4, sigw + 32 * (grp * NG + t) + 8 * g + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dot = fmaf(acc[t][4 * g + i], w[i], dot);
                }
        }
        if (raw) {
            *raw = acc[0];
        } else {
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int s = 0; s < P::UPT; ++s) {
                    const typename P::Frag f = P::from_acc(acc[t], s);
                    hout[(grp * NG + t) * P::UPT + s] = packed_relu ? P::relu(f) : f;
                }
        }
    }
    return dot;
}

template <int PREC, int WT, int WAVES, int ACT, bool BIGENC>
__global__ __launch_bounds__(64 * WAVES) void mlp_fwd_kernel(const MlpArgs a) {
    using P = Prec<PREC>;
    using Frag = typename P::Frag;
    constexpr int RT = WT > 1 ? WT / 2 : 1;
    constexpr int HN = WT * P::UPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) float* cst = (__attribute__((address_space(3))) float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

    WeightRing<WAVES> ring;
    ring.init(a.stream, (lds_bytes_t)smem + a.const_floats * (int)sizeof(float), a.n_pieces, wave, lane);
    for (int i = tid; i < a.const_floats; i += 64 * WAVES) cst[i] = a.consts[i];
    __syncthreads();

    const lds_floats_t sigw = cst + a.sigw_off;
    const bool colour = a.rgb != nullptr;
    const int last = 2 * a.hd - 1;                                 // last trunk layer feeds the density head

    for (uint32_t blk = blockIdx.x; blk < a.n_blocks; blk += gridDim.x) {
        // ---- this lane's sample (both lane halves hold the same one) ----------------
        const uint32_t m = blk * (32 * WAVES) + wave * 32 + (lane & 31);
        const bool valid = m < a.n_points;
        const uint32_t mc = valid ? m : a.n_points - 1;
        float px, py, pz, dx, dy, dz;
        if (a.t) {                                                  // ray.py:132 / :226
            const uint32_t ray = mc / a.n_samples;
            const float tt = a.t[mc];
            dx = a.rd[3 * ray]; dy = a.rd[3 * ray + 1]; dz = a.rd[3 * ray + 2];
            px = __fadd_rn(a.ro[3 * ray], __fmul_rn(dx, tt));
            py = __fadd_rn(a.ro[3 * ray + 1], __fmul_rn(dy, tt));
            pz = __fadd_rn(a.ro[3 * ray + 2], __fmul_rn(dz, tt));
        } else {
            px = a.ro[3 * (size_t)mc]; py = a.ro[3 * (size_t)mc + 1]; pz = a.ro[3 * (size_t)mc + 2];
            dx = a.rd[3 * (size_t)mc]; dy = a.rd[3 * (size_t)mc + 1]; dz = a.rd[3 * (size_t)mc + 2];
        }
        // A non-finite input poisons every feature in the reference (model.py:68-78).
        const float probe = (px + py + pz) + (colour ? (dx + dy + dz) : 0.0f);
        const bool poisoned = !(fabsf(probe) <= 3.0e38f);

        Encoding<PREC> enc;
        enc.template prepare<BIGENC>(a.ex_kind, a.ex_n, a.ex_units, cst + a.tabx_off, h, px, py, pz);
        Frag A[HN], B[HN];                                        // ping-pong activation fragments
        float dot = 0.0f;
        // ---- fc_1 then fc_2; fc_2's first layer optionally sees [phi_x | h] (model.py:71-73).
        // Layer 0 writes A; the remaining 2 hd - 1 trunk layers alternate A -> B, B -> A, so the
        // last one always lands in B.
        if (last == 0) dot = linear<PREC, WAVES, ACT, BIGENC, WT, 0, true>(ring, cst, h, A, B, true, enc, sigw);
        else linear<PREC, WAVES, ACT, BIGENC, WT, 0, false>(ring, cst, h, B, A, true, enc, sigw);
        int li = 1;
        for (; li + 1 <= last; li += 2) {
            linear<PREC, WAVES, ACT, BIGENC, WT, WT, false>(ring, cst + li * BIAS_STRIDE, h, A, B,
                                                    li == a.hd && a.resid, enc, sigw);
            linear<PREC, WAVES, ACT, BIGENC, WT, WT, false>(ring, cst + (li + 1) * BIAS_STRIDE, h, B, A,
                                                    li + 1 == a.hd && a.resid, enc, sigw);
        }
        if (last > 0)
            dot = linear<PREC, WAVES, ACT, BIGENC, WT, WT, true>(ring, cst + last * BIAS_STRIDE, h, A, B,
                                                         last == a.hd && a.resid, enc, sigw);
        // ---- density head (model.py:75, activations.py:33-35) ------------------------------
        {
            dot += __shfl_xor(dot, 32, 64);
            const float sx = (dot + cst[a.sigb_off]) - 1.0f;
            float sg = log1pf(expf(-fabsf(sx))) + (sx >= 0.0f ? sx : 0.0f);
            if (poisoned) sg = __builtin_nanf("");
            if (valid && h == 0) a.sigma[m] = sg;
        }
        if (colour) {
            li = last + 1;
            // ---- features: Linear without activation (model.py:51, :76): B -> A ------------
            linear<PREC, WAVES, -1, BIGENC, WT, WT, false>(ring, cst + li * BIAS_STRIDE, h, B, A, false, enc, sigw);
            ++li;
            // ---- rgb.0 on [phi_d | features] (model.py:53, :76): A -> B ---------------------
            enc.template prepare<BIGENC>(a.ed_kind, a.ed_n, a.ed_units, cst + a.tabd_off, h, dx, dy, dz);
            linear<PREC, WAVES, ACT, BIGENC, RT, WT, false>(ring, cst + li * BIAS_STRIDE, h, A, B, true, enc, sigw);
            ++li;
            // ---- rgb.2 + widened sigmoid (model.py:54, activations.py:18-20) ----------------
            f32x16 out;
            linear<PREC, WAVES, -1, BIGENC, 1, RT, false>(ring, cst + li * BIAS_STRIDE, h, B, A, false, enc, sigw, &out);
            if (valid && h == 0) {
                // rows 0..2 of the output tile sit in accumulator registers 0..2 of lane half 0
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = 0.5f * (1.0f + 1.002f * tanhf(0.5f * out[c]));
                    if (poisoned) v = __builtin_nanf("");
                    a.rgb[3 * (size_t)m + c] = v;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PREC, int WT, int WAVES, int ACT, bool BIGENC>
static int launch_cfg(const MlpArgs& a, size_t lds_bytes, hipStream_t stream) {
    auto kern = mlp_fwd_kernel<PREC, WT, WAVES, ACT, BIGENC>;
    static thread_local size_t configured = 0;
    if (lds_bytes > configured) {
        NERFX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        configured = lds_bytes;
    }
    int dev = 0, cus = 256;
    NERFX_HIP(hipGetDevice(&dev));
    NERFX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const uint32_t grid = std::min<uint32_t>(a.n_blocks, (uint32_t)cus);   // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds_bytes, stream, a);
    return launch_status("mlp_fwd_kernel");
}

template <int PREC, int WAVES, int ACT, bool BIGENC>
static int launch_width(int wt, const MlpArgs& a, size_t lds, hipStream_t s) {
    switch (wt) {
        case 1: return launch_cfg<PREC, 1, WAVES, ACT, BIGENC>(a, lds, s);
        case 2: return launch_cfg<PREC, 2, WAVES, ACT, BIGENC>(a, lds, s);
        case 4: return launch_cfg<PREC, 4, WAVES, ACT, BIGENC>(a, lds, s);
        case 8: return launch_cfg<PREC, 8, WAVES, ACT, BIGENC>(a, lds, s);
    }
    return fail(NERFX_E_ARG, "unsupported tile count %d", wt);
}

template <int PREC, int WAVES, bool BIGENC>
static int launch_activation(int activ, int wt, const MlpArgs& a, size_t lds, hipStream_t s) {
    switch (activ) {
        case NERFX_ACT_RELU: return launch_width<PREC, WAVES, NERFX_ACT_RELU, BIGENC>(wt, a, lds, s);
        case NERFX_ACT_LEAKY_RELU: return launch_width<PREC, WAVES, NERFX_ACT_LEAKY_RELU, BIGENC>(wt, a, lds, s);
        case NERFX_ACT_SILU: return launch_width<PREC, WAVES, NERFX_ACT_SILU, BIGENC>(wt, a, lds, s);
    }
    return fail(NERFX_E_ARG, "unknown activation %d", activ);
}

}  // namespace nerfx
