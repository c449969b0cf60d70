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
// stream (mlp.hpp) through an LDS ring filled by global_load_lds, RING_AHEAD
// chunks ahead of the consumer; workgroups are persistent and the stream simply
// wraps from one 32*WAVES-sample block into the next, so the ring never drains.
//
// The encoders are fused as producers of B operands: a k-unit of sin/cos values
// is computed in registers right before the MFMAs that consume it (layer 0, the
// skip layer and rgb.0), never stored.
#include "mlp.hpp"

namespace nerfx {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using lds_bytes_t = __attribute__((address_space(3))) char*;          // explicit LDS pointers: the
using lds_floats_t = const __attribute__((address_space(3))) float*;  // ring must never go through flat loads

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
    uint32_t n_chunks;           // stream length for this launch, in chunks
    int hd, resid, activ;
    int ex_kind, ex_n, ex_units, ed_kind, ed_n, ed_units;
    int sigw_off, sigb_off, tabx_off, tabd_off, const_floats;
};

// ---- LDS weight ring -----------------------------------------------------------
template <int WAVES>
struct WeightRing {
    static constexpr int LPW = CHUNK_UNITS / WAVES;        // global_load_lds per wave per chunk
    static constexpr int RING_UNITS = RING_SLOTS * CHUNK_UNITS;
    static_assert(CHUNK_UNITS % WAVES == 0, "chunk must split evenly over the waves");
    const char* src;
    lds_bytes_t lds;
    uint32_t n_chunks, issue_chunk, issue_slot, pos;
    int wave, lane;

    __device__ __forceinline__ void issue() {
#pragma unroll
        for (int k = 0; k < LPW; ++k) {
            const uint32_t unit = wave * LPW + k;
            const char* g = src + ((size_t)issue_chunk * CHUNK_UNITS + unit) * UNIT_BYTES + lane * 16;
            lds_bytes_t l = lds + (issue_slot * CHUNK_UNITS + unit) * UNIT_BYTES;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                             (void __attribute__((address_space(3)))*)l, 16, 0, 0);
        }
        issue_chunk = issue_chunk + 1 == n_chunks ? 0 : issue_chunk + 1;
        issue_slot = (issue_slot + 1) & (RING_SLOTS - 1);
    }
    __device__ __forceinline__ void init(const char* s, lds_bytes_t l, uint32_t nch, int w, int ln) {
        src = s; lds = l; n_chunks = nch; wave = w; lane = ln;
        issue_chunk = 0; issue_slot = 0; pos = 0;
#pragma unroll
        for (int c = 0; c < RING_AHEAD; ++c) issue();
    }
    // Call before consuming the k-block at `pos`.  On a chunk boundary: wait for this
    // wave's share of the chunk, meet the other waves (their shares have landed and
    // they are done with the previous chunk), then refill the slot freed two chunks ago.
    __device__ __forceinline__ void step() {
        if ((pos & (CHUNK_UNITS - 1)) == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING_AHEAD - 1) * LPW) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue();
        }
    }
    __device__ __forceinline__ lds_bytes_t block_ptr() const { return lds + pos * UNIT_BYTES + lane * 16; }
    __device__ __forceinline__ void advance(int units) { pos = (pos + units) & (RING_UNITS - 1); }
    __device__ __forceinline__ void align_chunk() {
        pos = ((pos + CHUNK_UNITS - 1) & ~(uint32_t)(CHUNK_UNITS - 1)) & (RING_UNITS - 1);
    }
};

// ---- per-precision pieces --------------------------------------------------------
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
    static __device__ __forceinline__ void mma(f32x16& acc, lds_bytes_t unit, const Frag& b) {
        const bf16x8 a = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(unit);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
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
    static __device__ __forceinline__ Frag from_floats(const float (&v)[4]) {
        return Frag{v[0], v[1], v[2], v[3]};
    }
    // accumulator registers 4g..4g+3 of tile `a` -> B operands of chained k-unit 4T+g
    static __device__ __forceinline__ Frag from_acc(const f32x16& a, int g) {
        return Frag{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
    }
    static __device__ __forceinline__ void mma(f32x16& acc, lds_bytes_t unit, const Frag& b) {
        const f32x4 a = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(unit);
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

template <int PREC, int ACT, int NG>
__device__ __forceinline__ void apply(f32x16 (&acc)[NG]) {
#pragma unroll
    for (int t = 0; t < NG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = acc[t][r];
            if (ACT == NERFX_ACT_RELU) acc[t][r] = fmaxf(x, 0.0f);
            if (ACT == NERFX_ACT_LEAKY_RELU) acc[t][r] = x > 0.0f ? x : 0.01f * x;
            if (ACT == NERFX_ACT_SILU) acc[t][r] = x / (1.0f + Prec<PREC>::expneg(x));
        }
}

// ---- one Linear layer -----------------------------------------------------------------
// out = act(W . [encoder units | chained input] + b), computed NG output tiles at a time so
// that only NG accumulator tiles are live beside the input and output fragments.
//   hin  : B operands of the chained input (CT * UPT fragments used)
//   hout : receives the activated output as next-layer B operands (NT * UPT fragments)
//   SIGMA: also accumulate the density head's dot product on the fp32 activations
template <int PREC, int WAVES, int NT, int CT, bool SIGMA, int HN_IN, int HN_OUT>
__device__ __forceinline__ float linear(WeightRing<WAVES>& ring, lds_floats_t bias, int h, int activ,
                                        const typename Prec<PREC>::Frag (&hin)[HN_IN],
                                        typename Prec<PREC>::Frag (&hout)[HN_OUT],
                                        int eunits, int ekind, int en, lds_floats_t etable,
                                        float vx, float vy, float vz, lds_floats_t sigw) {
    using P = Prec<PREC>;
    constexpr int NG = group_tiles(PREC, NT);
    static_assert(CT * P::UPT <= HN_IN && NT * P::UPT <= HN_OUT, "fragment arrays too small");
    float dot = 0.0f;
#pragma unroll
    for (int grp = 0; grp < NT / NG; ++grp) {
        f32x16 acc[NG];
        // bias as the initial accumulator: register r of a tile is feature (r&3) + 8 (r>>2) + 4 h
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(
                    bias + 32 * (grp * NG + t) + 8 * g + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = b[i];
            }
        for (int u = 0; u < eunits; ++u) {
            ring.step();
            const typename P::Frag b = encode_unit<PREC>(ekind, en, etable, u, h, vx, vy, vz);
            const lds_bytes_t blk = ring.block_ptr();
#pragma unroll
            for (int t = 0; t < NG; ++t) P::mma(acc[t], blk + t * UNIT_BYTES, b);
            ring.advance(NG);
        }
#pragma unroll
        for (int ku = 0; ku < CT * P::UPT; ++ku) {
            ring.step();
            const lds_bytes_t blk = ring.block_ptr();
#pragma unroll
            for (int t = 0; t < NG; ++t) P::mma(acc[t], blk + t * UNIT_BYTES, hin[ku]);
            ring.advance(NG);
            __builtin_amdgcn_sched_barrier(0);
        }
        switch (activ) {                                           // wave-uniform
            case NERFX_ACT_RELU: apply<PREC, NERFX_ACT_RELU>(acc); break;
            case NERFX_ACT_LEAKY_RELU: apply<PREC, NERFX_ACT_LEAKY_RELU>(acc); break;
            case NERFX_ACT_SILU: apply<PREC, NERFX_ACT_SILU>(acc); break;
            default: break;                                        // plain Linear
        }
        if (SIGMA) {
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(
                        sigw + 32 * (grp * NG + t) + 8 * g + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dot = fmaf(acc[t][4 * g + i], w[i], dot);
                }
        }
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
            for (int s = 0; s < P::UPT; ++s) hout[(grp * NG + t) * P::UPT + s] = P::from_acc(acc[t], s);
    }
    return dot;
}

template <int PREC, int WT, int WAVES>
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
    ring.init(a.stream, (lds_bytes_t)smem + a.const_floats * (int)sizeof(float), a.n_chunks, wave, lane);
    for (int i = tid; i < a.const_floats; i += 64 * WAVES) cst[i] = a.consts[i];
    __syncthreads();

    const lds_floats_t tab_x = cst + a.tabx_off;
    const lds_floats_t tab_d = cst + a.tabd_off;
    const lds_floats_t sigw = cst + a.sigw_off;
    const bool colour = a.rgb != nullptr;

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

        Frag hin[HN], hout[HN];
        float dot = 0.0f;
        const int last = 2 * a.hd - 1;                             // last trunk layer feeds the density head
        // ---- fc_1 then fc_2; fc_2's first layer optionally sees [phi_x | h] (model.py:71-73)
        for (int li = 0; li <= last; ++li) {
            const lds_floats_t bias = cst + li * BIAS_STRIDE;
            if (li == 0) {
                if (last == 0) dot = linear<PREC, WAVES, WT, 0, true>(ring, bias, h, a.activ, hin, hout, a.ex_units, a.ex_kind, a.ex_n, tab_x, px, py, pz, sigw);
                else linear<PREC, WAVES, WT, 0, false>(ring, bias, h, a.activ, hin, hout, a.ex_units, a.ex_kind, a.ex_n, tab_x, px, py, pz, sigw);
            } else {
                const int eu = (li == a.hd && a.resid) ? a.ex_units : 0;
                if (li == last) dot = linear<PREC, WAVES, WT, WT, true>(ring, bias, h, a.activ, hin, hout, eu, a.ex_kind, a.ex_n, tab_x, px, py, pz, sigw);
                else linear<PREC, WAVES, WT, WT, false>(ring, bias, h, a.activ, hin, hout, eu, a.ex_kind, a.ex_n, tab_x, px, py, pz, sigw);
            }
#pragma unroll
            for (int i = 0; i < HN; ++i) hin[i] = hout[i];
        }
        // ---- density head (model.py:75, activations.py:33-35) ------------------------------
        {
            dot += __shfl_xor(dot, 32, 64);
            const float sx = (dot + cst[a.sigb_off]) - 1.0f;
            float sg = log1pf(expf(-fabsf(sx))) + (sx >= 0.0f ? sx : 0.0f);
            if (poisoned) sg = __builtin_nanf("");
            if (valid && h == 0) a.sigma[m] = sg;
        }
        if (colour) {
            ring.align_chunk();                                    // the colour layers start a chunk
            int li = last + 1;
            // ---- features: Linear without activation (model.py:51, :76) --------------------
            linear<PREC, WAVES, WT, WT, false>(ring, cst + li * BIAS_STRIDE, h, -1, hin, hout, 0, 0, 0, tab_x, px, py, pz, sigw);
            ++li;
            // ---- rgb.0 on [phi_d | features] (model.py:53, :76) -----------------------------
            linear<PREC, WAVES, RT, WT, false>(ring, cst + li * BIAS_STRIDE, h, a.activ, hout, hin, a.ed_units, a.ed_kind, a.ed_n, tab_d, dx, dy, dz, sigw);
            ++li;
            // ---- rgb.2 + widened sigmoid (model.py:54, activations.py:18-20) ----------------
            f32x16 o;
            {
                const lds_floats_t bias = cst + li * BIAS_STRIDE;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(bias + 8 * g + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[4 * g + i] = b[i];
                }
#pragma unroll
                for (int ku = 0; ku < RT * P::UPT; ++ku) {
                    ring.step();
                    P::mma(o, ring.block_ptr(), hin[ku]);
                    ring.advance(1);
                }
            }
            if (valid && h == 0) {                                   // rows 0..2 live in registers 0..2 of half 0
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = 0.5f * (1.0f + 1.002f * tanhf(0.5f * o[c]));
                    if (poisoned) v = __builtin_nanf("");
                    a.rgb[3 * (size_t)m + c] = v;
                }
            }
        }
        ring.align_chunk();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace nerfx

using namespace nerfx;

template <int PREC, int WT, int WAVES>
static int launch_cfg(const MlpArgs& a, size_t lds_bytes, hipStream_t stream) {
    auto kern = mlp_fwd_kernel<PREC, WT, WAVES>;
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

template <int PREC, int WAVES>
static int launch_prec(int wt, const MlpArgs& a, size_t lds, hipStream_t s) {
    switch (wt) {
        case 1: return launch_cfg<PREC, 1, WAVES>(a, lds, s);
        case 2: return launch_cfg<PREC, 2, WAVES>(a, lds, s);
        case 4: return launch_cfg<PREC, 4, WAVES>(a, lds, s);
        case 8: return launch_cfg<PREC, 8, WAVES>(a, lds, s);
    }
    return fail(NERFX_E_ARG, "unsupported tile count %d", wt);
}

static int mlp_launch(const nerfx_model_t* model, int precision, const float* p0, const float* p1, const float* t,
                      int64_t n_points, int n_samples, float* sigma, float* rgb, void* stream) {
    NERFX_REQUIRE(model != nullptr, "model is NULL");
    NERFX_REQUIRE(precision == NERFX_PREC_FP32 || precision == NERFX_PREC_BF16, "unknown precision %d", precision);
    NERFX_REQUIRE(n_points >= 0 && n_points < (int64_t)1 << 31, "n_points %lld out of range [0, 2^31)", (long long)n_points);
    if (n_points == 0) return 0;
    NERFX_REQUIRE(p0 && p1 && sigma, "NULL input or output");
    const ModelPlan& p = model->plan[precision];
    const int waves = precision == NERFX_PREC_BF16 ? 8 : 4;
    MlpArgs a{};
    a.stream = (const char*)model->stream_dev[precision];
    a.consts = model->const_dev;
    a.ro = p0; a.rd = p1; a.t = t; a.sigma = sigma; a.rgb = rgb;
    a.n_points = (uint32_t)n_points;
    a.n_samples = (uint32_t)n_samples;
    a.n_blocks = (uint32_t)((n_points + 32 * waves - 1) / (32 * waves));
    a.n_chunks = (uint32_t)((rgb ? p.total_units : p.trunk_units) / CHUNK_UNITS);
    a.hd = p.hd; a.resid = p.desc.resid; a.activ = p.desc.activ;
    a.ex_kind = p.desc.enc_x_kind; a.ex_n = p.desc.enc_x_n; a.ex_units = p.layer[0].enc_units;
    a.ed_kind = p.desc.enc_d_kind; a.ed_n = p.desc.enc_d_n; a.ed_units = p.layer[2 * p.hd + 1].enc_units;
    a.sigw_off = p.sigw_off; a.sigb_off = p.sigb_off; a.tabx_off = p.tabx_off; a.tabd_off = p.tabd_off;
    a.const_floats = p.const_floats;
    const size_t lds = (size_t)p.const_floats * sizeof(float) + (size_t)RING_SLOTS * CHUNK_UNITS * UNIT_BYTES;
    NERFX_REQUIRE(lds <= 160 * 1024, "model needs %zu bytes of LDS (> 160 KiB)", lds);
    if (precision == NERFX_PREC_BF16) return launch_prec<NERFX_PREC_BF16, 8>(p.wt, a, lds, (hipStream_t)stream);
    return launch_prec<NERFX_PREC_FP32, 4>(p.wt, a, lds, (hipStream_t)stream);
}

extern "C" int nerfx_mlp_fwd_points(const nerfx_model_t* model, int precision, const float* x, const float* d,
                                    int64_t n_points, float* sigma, float* rgb, void* stream) {
    return mlp_launch(model, precision, x, d, nullptr, n_points, 1, sigma, rgb, stream);
}

extern "C" int nerfx_mlp_fwd_rays(const nerfx_model_t* model, int precision, const float* ro, const float* rd,
                                  const float* t, int64_t n_rays, int n_samples, float* sigma, float* rgb,
                                  void* stream) {
    NERFX_REQUIRE(n_rays >= 0 && n_samples >= 1, "bad ray/sample count");
    NERFX_REQUIRE(n_rays == 0 || t != nullptr, "t is NULL");
    return mlp_launch(model, precision, ro, rd, t, n_rays * n_samples, n_samples, sigma, rgb, stream);
}
