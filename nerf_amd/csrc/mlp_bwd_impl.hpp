// K2b, part 1: the input-gradient chain of the fused MLP (autograd of model.py:68-78).
//
// Same orientation and weight ring as the forward kernel: a wavefront owns 32 samples, the
// gradient at a layer's pre-activation (Gz, 32 x 32 accumulator tiles with the sample on the
// lane) is converted in registers into the B operand of the next product
//     G_{l-1} = W_l[:, chained columns]^T . Gz_l ,      Gz_{l-1} = G_{l-1} * act'(z_{l-1}),
// and the weights come from the model's BACKWARD unit stream (transposed blocks, packed by
// model.hip).  z is read from the tape the training forward wrote; every Gz is written to the
// gradient tape (same tile map), from which the weight-gradient kernel forms dW = Gz^T . X.
// Gradients with respect to the encodings are never formed: positions and directions are
// inputs of the reference's training step, not parameters.
#pragma once
#include "mlp_fwd_impl.hpp"

namespace nerfx {

struct BwdArgs {
    const char* stream;          // packed backward units
    const float* consts;         // const block (global): the density head's weights are used
    const void* ztape;           // pre-activations (training forward)
    const void* stape;           // their sign words (bf16, ReLU-like activations), else NULL
    void* gtape;                 // receives the pre-activation gradients
    uint64_t m_pad;              // samples per tape tile
    const float* g_sigma;        // (M)
    const float* g_rgb;          // (M,3)
    const float* sigma;          // (M)   forward outputs: the output activations' derivatives
    const float* rgb;            // (M,3) are formed from them
    uint32_t n_points, n_blocks, n_pieces;
    int hd, sigw_off, const_floats;
    int z_tile[MAX_LAYERS], heads_tile, n_tiles;   // (bf16: z_tile of the folded-away `features` layer is unused)
};

template <int PREC, int ACT>
__device__ __forceinline__ float activation_slope(float z) {
    if (ACT == NERFX_ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
    if (ACT == NERFX_ACT_LEAKY_RELU) return z > 0.0f ? 1.0f : 0.01f;
    if (ACT == NERFX_ACT_SILU) {
        const float s = Prec<PREC>::sigmoid(z);
        return s * (1.0f + z * (1.0f - s));
    }
    return 1.0f;
}

// One step of the chain.  hin: Gz of the later layer (CT tiles); the product lands in NT
// tiles, gets the density head's contribution (SIGMA: + w_sigma[f] * gs), is multiplied by
// act'(z) of the rows it belongs to (ACT < 0: a plain Linear, no factor), is stored to the
// gradient tape at tiles `tile0`.. and becomes hout.
//
// Tape traffic and the hand-overs: a hand-over waits on the vector-memory counter, which retires
// in issue order.  The z tiles of a group (bf16) are therefore requested BEFORE the group's
// hand-over and used after its MFMA run, and the hand-over lets them, and the gradient tiles the
// previous group has just stored, stay in flight (PREV_OPS: what the caller's previous step left).
// Before the z tiles are used, the wait in turn leaves the pieces requested by this group's
// hand-overs in flight.
template <int PREC, class RG, int ACT, int NT, int CT, bool SIGMA, int PREV_OPS, int HN_IN, int HN_OUT>
__device__ __forceinline__ void linear_bwd(WeightRing<RG>& ring, int h, const typename Prec<PREC>::Frag (&hin)[HN_IN],
                                           typename Prec<PREC>::Frag (&hout)[HN_OUT], const Tape<PREC>& ztape,
                                           const Tape<PREC>& gtape, int tile0, lds_floats_t sigw, float gs) {
    using P = Prec<PREC>;
    constexpr int NG = group_tiles(PREC, NT);
    constexpr int SYNC_PIECES = RG::SYNC;
    constexpr int KB_PER_PIECE = PIECE_UNITS / NG;
    constexpr int CHAIN_KU = CT * P::UPT;
    constexpr int CHAIN_PIECES = pad_to_piece(CHAIN_KU * NG) / PIECE_UNITS;
    constexpr bool PREFETCH = ACT >= 0 && PREC == NERFX_PREC_BF16;
    constexpr bool SIGNS = ACT >= 0 && tape_has_signs(PREC, ACT);     // one 2-byte load per tile instead of the tile
    constexpr int Z_OPS = PREFETCH ? NG * (SIGNS ? 1 : Tape<PREC>::TILE_VMEM_OPS) : 0;
    static_assert(CHAIN_KU <= HN_IN && NT * P::UPT <= HN_OUT, "fragment arrays too small");
    auto group = [&](int grp, auto stored_before) {
        constexpr int YOUNGER = decltype(stored_before)::value + Z_OPS;
        typename Tape<NERFX_PREC_BF16>::Raw zraw[NG];
        uint32_t zsign[NG];
        if constexpr (SIGNS) {
#pragma unroll
            for (int t = 0; t < NG; ++t)                                // untracked by the compiler, like load_raw_async
                asm volatile("global_load_ushort %0, %1, off" : "=&v"(zsign[t]) : "v"(ztape.sign_at(tile0 + grp * NG + t, h)) : "memory");
        } else if constexpr (PREFETCH) {
#pragma unroll
            for (int t = 0; t < NG; ++t) zraw[t] = ztape.load_raw_async(tile0 + grp * NG + t, h);
        }
        f32x16 acc[NG];
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
        for (int p0 = 0; p0 < CHAIN_PIECES; p0 += SYNC_PIECES) {
            constexpr int TAIL = CHAIN_PIECES % SYNC_PIECES ? CHAIN_PIECES % SYNC_PIECES : SYNC_PIECES;
            const int ku0 = p0 * KB_PER_PIECE;
            if (CHAIN_PIECES - p0 >= SYNC_PIECES) {
                constexpr int KU = SYNC_PIECES * KB_PER_PIECE;
                if (p0 == 0) ring.template acquire<SYNC_PIECES, YOUNGER>();
                else ring.template acquire<SYNC_PIECES>();
                run_units<PREC, RG, NG, KU * NG>(ring, acc, [&](int k) { return hin[ku0 + k]; });
                ring.template release<SYNC_PIECES>();
            } else {
                constexpr int KU = CHAIN_KU - (CHAIN_PIECES - TAIL) * KB_PER_PIECE;
                if (p0 == 0) ring.template acquire<TAIL, YOUNGER>();
                else ring.template acquire<TAIL>();
                run_units<PREC, RG, NG, (KU > 0 ? KU : 1) * NG>(ring, acc, [&](int k) { return hin[ku0 + k]; });
                ring.template release<TAIL>();
            }
        }
        // the z tiles were requested before this group's hand-overs, which have issued CHAIN_PIECES
        // pieces since: everything older than those transfers has arrived after this wait
        if constexpr (PREFETCH)
            asm volatile("s_waitcnt vmcnt(%0) ; async tape loads landed" ::"n"(CHAIN_PIECES * WeightRing<RG>::LPW) : "memory");   // (the comment is what tools/check_async_loads.py looks for)
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int tile = grp * NG + t;
            if (SIGMA) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = NERFX_LDS_VEC(f32x4, sigw + 32 * tile + 8 * g + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = fmaf(w[i], gs, acc[t][4 * g + i]);
                }
            }
            if constexpr (SIGNS) {
                uint32_t w = zsign[t];
                asm volatile("" : "+v"(w));                            // nothing derived from it before the wait above
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int32_t neg = __builtin_amdgcn_sbfe((int32_t)w, 15 - r, 1);   // -1 where z < 0 (or -0)
                    if (ACT == NERFX_ACT_RELU) {
                        const float g = acc[t][r];                     // (a temporary: see Tape::sign_word)
                        acc[t][r] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, g) & ~(uint32_t)neg);
                    } else
                        acc[t][r] *= neg ? 0.01f : 1.0f;
                }
            } else if (ACT >= 0) {
                f32x16 z;
                if constexpr (PREFETCH) z = Tape<NERFX_PREC_BF16>::expand(zraw[t]);
                else z = ztape.load_tile(tile0 + tile, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] *= activation_slope<PREC, ACT>(z[r]);
            }
            gtape.store_tile(tile0 + tile, h, acc[t]);
#pragma unroll
            for (int s = 0; s < P::UPT; ++s) hout[tile * P::UPT + s] = P::from_acc(acc[t], s);
        }
    };
    // every group stores NG gradient tiles after its run: at least NG * TILE_VMEM_OPS instructions
    group(0, std::integral_constant<int, PREV_OPS>{});
#pragma unroll
    for (int grp = 1; grp < NT / NG; ++grp) group(grp, std::integral_constant<int, NG * Tape<PREC>::TILE_VMEM_OPS>{});
}

// Store instructions the last group of a linear_bwd with NT output tiles leaves behind (a lower bound).
template <int PREC>
__host__ __device__ constexpr int stores_left(int nt) { return group_tiles(PREC, nt) * Tape<PREC>::TILE_VMEM_OPS; }

template <int PREC, int WT, class RG, int ACT>
__global__ __launch_bounds__(64 * RG::WAVES, RG::BLOCKS_PER_CU * RG::WAVES / 4) void mlp_bwd_kernel(const BwdArgs a) {
    constexpr int WAVES = RG::WAVES;
    using P = Prec<PREC>;
    using Frag = typename P::Frag;
    constexpr int RT = WT > 1 ? WT / 2 : 1;
    constexpr int HN = WT * P::UPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) float* cst = (__attribute__((address_space(3))) float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    WeightRing<RG> ring;                                          // LDS: [const block | weight ring]
    ring.init(a.stream, (lds_bytes_t)smem + a.const_floats * (int)sizeof(float), a.n_pieces, wave, lane);
    for (int i = tid; i < a.const_floats; i += 64 * WAVES) cst[i] = a.consts[i];
    __syncthreads();
    const lds_floats_t sigw = cst + a.sigw_off;
    const int feat = 2 * a.hd;                                    // index of the `features` layer

    for (uint32_t blk = blockIdx.x; blk < a.n_blocks; blk += gridDim.x) {
        const uint32_t m = blk * (32 * WAVES) + wave * 32 + (lane & 31);
        const bool valid = m < a.n_points;
        // Output activations, differentiated through their outputs:
        //   rgb = 1.002 s(o) - 0.001  (activations.py:18-20)  => d rgb / d o = 1.002 s (1 - s)
        //   sigma = softplus(x - 1)   (activations.py:33-35)  => d sigma / d x = 1 - exp(-sigma)
        float go[3] = {0.0f, 0.0f, 0.0f}, gs = 0.0f;              // tail lanes contribute nothing
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float s = (a.rgb[3 * (size_t)m + c] + 0.001f) * (1.0f / 1.002f);
                go[c] = a.g_rgb[3 * (size_t)m + c] * (1.002f * s * (1.0f - s));
            }
            gs = a.g_sigma[m] * -expm1f(-a.sigma[m]);
        }
        const Tape<PREC> zt{(typename Tape<PREC>::T*)a.ztape, a.m_pad, m, (uint16_t*)a.stape, (uint32_t)a.n_tiles};
        const Tape<PREC> gt{(typename Tape<PREC>::T*)a.gtape, a.m_pad, m, nullptr, (uint32_t)a.n_tiles};
        if (h == 0) {                                               // heads tile: positions 0..2 and 3
            using TT = typename Tape<PREC>::T;
            TT* p = gt.at(a.heads_tile, 0);
            p[0] = (TT)go[0]; p[1] = (TT)go[1]; p[2] = (TT)go[2]; p[3] = (TT)gs;
        }
        // rgb.2's gradient as a 32-row tile: rows 0..2 are registers 0..2 of lane half 0
        f32x16 top;
#pragma unroll
        for (int r = 0; r < 16; ++r) top[r] = 0.0f;
        if (h == 0) { top[0] = go[0]; top[1] = go[1]; top[2] = go[2]; }
        Frag A[HN], B[HN];
#pragma unroll
        for (int s = 0; s < P::UPT; ++s) A[s] = P::from_acc(top, s);
        constexpr int S_RT = stores_left<PREC>(RT), S_WT = stores_left<PREC>(WT);
        // bf16: `features` is folded into rgb.0 in the streams (LayerPlan::folded), so rgb.0's transposed
        // block maps its gradient straight to the last trunk layer's output, where the density head's
        // contribution joins it; fp32 walks through `features` as the reference's autograd does.
        constexpr bool FOLD = folds_features(PREC);
        linear_bwd<PREC, RG, ACT, RT, 1, false, 0>(ring, h, A, B, zt, gt, a.z_tile[feat + 1], sigw, gs);          // -> rgb.0
        if constexpr (FOLD) {
            linear_bwd<PREC, RG, ACT, WT, RT, true, S_RT>(ring, h, B, A, zt, gt, a.z_tile[feat - 1], sigw, gs);  // -> last trunk layer
        } else {
            linear_bwd<PREC, RG, -1, WT, RT, false, S_RT>(ring, h, B, A, zt, gt, a.z_tile[feat], sigw, gs);      // -> features
            linear_bwd<PREC, RG, ACT, WT, WT, true, S_WT>(ring, h, A, B, zt, gt, a.z_tile[feat - 1], sigw, gs);  // -> last trunk layer
        }
        // trunk layers feat-1 .. 1, each producing the gradient of the layer below it
        Frag (&X)[HN] = FOLD ? A : B;                               // holds the last trunk layer's gradient
        Frag (&Y)[HN] = FOLD ? B : A;
        int li = feat - 1;
        if (li >= 1) {
            linear_bwd<PREC, RG, ACT, WT, WT, false, S_WT>(ring, h, X, Y, zt, gt, a.z_tile[li - 1], sigw, gs);
            --li;
        }
        for (; li >= 2; li -= 2) {
            linear_bwd<PREC, RG, ACT, WT, WT, false, S_WT>(ring, h, Y, X, zt, gt, a.z_tile[li - 1], sigw, gs);
            linear_bwd<PREC, RG, ACT, WT, WT, false, S_WT>(ring, h, X, Y, zt, gt, a.z_tile[li - 2], sigw, gs);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PREC, int WT, class RG, int ACT>
static int launch_bwd_cfg(const BwdArgs& a, hipStream_t stream) {
    auto kern = mlp_bwd_kernel<PREC, WT, RG, ACT>;
    const size_t lds_bytes = (size_t)a.const_floats * sizeof(float) + (size_t)RG::PIECES * PIECE_UNITS * UNIT_BYTES;
    NERFX_REQUIRE(lds_bytes <= 160 * 1024, "model needs %zu bytes of LDS per workgroup", lds_bytes);
    NERFX_OPT_IN_LDS(kern, lds_bytes);
    int dev = 0, cus = 256;
    NERFX_HIP(hipGetDevice(&dev));
    NERFX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const uint32_t grid = std::min<uint32_t>(a.n_blocks, (uint32_t)cus * RG::BLOCKS_PER_CU);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * RG::WAVES), lds_bytes, stream, a);
    return launch_status("mlp_bwd_kernel");
}

template <int PREC, class RG, int ACT>
static int launch_bwd_width(int wt, const BwdArgs& a, hipStream_t s) {
    switch (wt) {
        case 1: return launch_bwd_cfg<PREC, 1, RG, ACT>(a, s);
        case 2: return launch_bwd_cfg<PREC, 2, RG, ACT>(a, s);
        case 4: return launch_bwd_cfg<PREC, 4, RG, ACT>(a, s);
        case 8: return launch_bwd_cfg<PREC, 8, RG, ACT>(a, s);
    }
    return fail(NERFX_E_ARG, "unsupported tile count %d", wt);
}

template <int PREC, class RG>
static int launch_bwd_activation(int activ, int wt, const BwdArgs& a, hipStream_t s) {
    switch (activ) {
        case NERFX_ACT_RELU: return launch_bwd_width<PREC, RG, NERFX_ACT_RELU>(wt, a, s);
        case NERFX_ACT_LEAKY_RELU: return launch_bwd_width<PREC, RG, NERFX_ACT_LEAKY_RELU>(wt, a, s);
        case NERFX_ACT_SILU: return launch_bwd_width<PREC, RG, NERFX_ACT_SILU>(wt, a, s);
    }
    return fail(NERFX_E_ARG, "unknown activation %d", activ);
}

}  // namespace nerfx
