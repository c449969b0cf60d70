// K2b, part 2: weight and bias gradients of the MLP from the two training tapes.
//
//     dW_l[o][i] = sum over samples m of  Gz_l[m][o] * X_l[m][i]
//
// Both tapes are arrays of TILES [tile][sample][32 positions] (see Tape): a tile holds 32
// features of every sample, sample-major, exactly as the producing wavefronts had them in
// registers.  The contraction runs over samples, i.e. over the ROWS of both operands, so a
// workgroup stages a chunk of samples of up to 4 gradient tiles and 4 input tiles in LDS with
// plain 16-byte copies (the activation of z, which makes X, is applied on the way) and its four
// wavefronts read the MFMA operands transposed: ds_read_b64_tr_b16 for bf16, one dword per lane
// for the fp32 32x32x2 form.  X is the stored encoder slot, the activation of the previous
// layer's z; the bias gradient is the product with a constant operand (1 at position 0) that
// never touches memory, accumulated by the workgroup of a layer's first input block.  The sample range
// is split over workgroups; every (task, split) writes fp32 partials and a second kernel adds
// them in a fixed order and scatters into torch-layout gradients: deterministic, no atomics.
#include "mlp_fwd_impl.hpp"

#include <vector>

namespace nerfx {

enum { WB_RAW = 0, WB_ACT = 1 };                    // how an input tile becomes X
enum { WM_ENC = 0, WM_CHAIN = 1 };                  // where an input tile's positions go
enum { WO_LAYER = 0, WO_RGB = 1, WO_SIGMA = 2 };    // which weight rows a gradient tile's positions are
constexpr int WG_SIDE = 4;                          // tiles per side of a workgroup's block
constexpr int TASK_TILES = WG_SIDE * WG_SIDE + WG_SIDE;   // partial tiles per task: the block + one bias tile per gradient tile

// A block of one layer's gradient: up to 4 gradient tiles against up to 4 input tiles.
struct WTask {
    int n_a, n_b;
    int a_tile[WG_SIDE], a_kind[WG_SIDE], a_row0[WG_SIDE];                     // gradient tape
    int b_tile[WG_SIDE], b_kind[WG_SIDE], b_map[WG_SIDE], b_col0[WG_SIDE];     // pre-activation tape
    int w_index, b_index;                          // positions of weight and bias in the gradient list
    int bias;                                      // this block also sums its gradient tiles over the samples
    int out_dim, enc_kind, enc_n, enc_cols, ld, chain_dim;
};

struct GradList { float* p[2 + 2 * MAX_LAYERS + 8 + 2]; };   // the caller's tensors, then the folded rgb.0's weight and bias

static void build_tasks(const ModelPlan& p, std::vector<WTask>& out) {
    struct A { int tile, kind, row0; };
    struct B { int tile, kind, map, col0; };
    const int spu = slots_per_unit(p.prec);
    const int hd = p.hd, n_jobs = p.n_layers + 1;                   // every Linear + the density head
    const int n_tensors = 2 + 4 * hd + 8;
    for (int job = 0; job < n_jobs; ++job) {
        const bool sigma = job == p.n_layers;
        if (!sigma && p.layer[job].absent) continue;                 // `features`: its gradient comes out of rgb.0's (unfold_kernel)
        const LayerPlan& L = p.layer[sigma ? 2 * hd : job];          // density head: a width-wide input, like `features`' plan
        std::vector<A> outs;
        std::vector<B> ins;
        int w_index, out_dim;
        if (sigma) {
            outs.push_back({p.tape_heads_tile, WO_SIGMA, 0});
            w_index = 2 + 4 * hd; out_dim = 1;
        } else {
            w_index = job < 2 * hd ? 2 + 2 * job : 2 + 4 * hd + 2 + 2 * (job - 2 * hd);
            if (L.folded) w_index = n_tensors;                       // the folded layer's own gradient (GradList)
            out_dim = L.out_dim;
            if (job == p.n_layers - 1) outs.push_back({p.tape_heads_tile, WO_RGB, 0});
            else
                for (int t = 0; t < L.n_tiles; ++t)
                    if (L.out_dim > 32 * t) outs.push_back({p.tape_z_tile[job] + t, WO_LAYER, 32 * t});
        }
        if (!sigma && L.enc_units) {
            const int base = job == 2 * hd + 1 ? p.tape_ed_tile : p.tape_ex_tile;   // rgb.0 sees phi_d
            for (int t = 0; t < L.enc_units * spu / 32; ++t) ins.push_back({base + t, WB_RAW, WM_ENC, 32 * t});
        }
        if (L.chain_tiles) {
            const int src = sigma || L.folded ? 2 * hd - 1 : job - 1;   // the layer whose output feeds this one
            const int kind = src == 2 * hd ? WB_RAW : WB_ACT;        // `features` has no activation
            for (int t = 0; t < L.chain_tiles; ++t)
                if (L.chain_dim > 32 * t) ins.push_back({p.tape_z_tile[src] + t, kind, WM_CHAIN, 32 * t});
        }
        for (size_t o = 0; o < outs.size(); o += WG_SIDE)
            for (size_t i = 0; i < ins.size(); i += WG_SIDE) {
                WTask k{};
                k.n_a = (int)std::min<size_t>(WG_SIDE, outs.size() - o);
                k.n_b = (int)std::min<size_t>(WG_SIDE, ins.size() - i);
                for (int s = 0; s < k.n_a; ++s) { k.a_tile[s] = outs[o + s].tile; k.a_kind[s] = outs[o + s].kind; k.a_row0[s] = outs[o + s].row0; }
                for (int s = 0; s < k.n_b; ++s) {
                    k.b_tile[s] = ins[i + s].tile; k.b_kind[s] = ins[i + s].kind; k.b_map[s] = ins[i + s].map; k.b_col0[s] = ins[i + s].col0;
                }
                k.w_index = w_index; k.b_index = w_index + 1; k.out_dim = out_dim;
                k.bias = i == 0;
                k.enc_kind = L.enc_kind; k.enc_n = L.enc_n;
                k.enc_cols = sigma ? 0 : L.enc_cols; k.ld = sigma ? L.chain_dim : L.in_dim; k.chain_dim = L.chain_dim;
                out.push_back(k);
            }
    }
}

using s16x4 = __attribute__((ext_vector_type(4))) short;

template <int PREC> struct WOperand;
template <> struct WOperand<NERFX_PREC_BF16> {
    using T = __bf16;
    using V = bf16x8;                              // 16 bytes as copied
    static constexpr int EPV = 8;
    static constexpr int CHUNK = 64;               // samples staged per step
    static constexpr int KSTEP = 16;               // samples per MFMA
    static __device__ __forceinline__ V zero() { return __builtin_bit_cast(V, s16x8(0)); }
    template <int ACT>
    static __device__ __forceinline__ V act(const V& z) {
        if (ACT == NERFX_ACT_RELU) return Prec<NERFX_PREC_BF16>::relu(z);
        V o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)activate<NERFX_PREC_BF16, ACT>((float)z[e]);
        return o;
    }
    // Operand of one 32-position tile for MFMA step kk of the staged chunk: lane (i = lane & 31,
    // kh = lane >> 5) gets samples 16 kk + 8 kh .. + 7 of position i.  A 16-lane group reads
    // 4 samples x 16 positions per instruction, transposed by the LDS.
    using Frag = bf16x8;
    static __device__ __forceinline__ uint32_t lane_offset(int lane) {
        const int L = lane & 15, g = lane >> 4;
        return (uint32_t)((8 * (g >> 1) + (L >> 2)) * 64 + (16 * (g & 1) + 4 * (L & 3)) * 2);
    }
    static __device__ __forceinline__ Frag fetch(lds_bytes_t tile, uint32_t lane_off, int kk) {
        using P4 = __attribute__((address_space(3))) s16x4*;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P4)(tile + lane_off + kk * 1024));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P4)(tile + lane_off + kk * 1024 + 256));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(Frag, v);
    }
    static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    // the operand "1 at position 0 of every sample": lane (i, kh) holds 8 samples of position i
    static __device__ __forceinline__ Frag ones(int lane) {
        return __builtin_bit_cast(Frag, s16x8((lane & 31) == 0 ? (short)0x3F80 : (short)0));
    }
};
template <> struct WOperand<NERFX_PREC_FP32> {
    using T = float;
    using V = f32x4;
    static constexpr int EPV = 4;
    static constexpr int CHUNK = 32;
    static constexpr int KSTEP = 2;
    static __device__ __forceinline__ V zero() { return V{0.0f, 0.0f, 0.0f, 0.0f}; }
    template <int ACT>
    static __device__ __forceinline__ V act(const V& z) {
        V o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = activate<NERFX_PREC_FP32, ACT>(z[e]);
        return o;
    }
    using Frag = float;                            // lane (i, kh): sample 2 kk + kh of position i
    static __device__ __forceinline__ uint32_t lane_offset(int lane) { return (uint32_t)(((lane >> 5) * 32 + (lane & 31)) * 4); }
    static __device__ __forceinline__ Frag fetch(lds_bytes_t tile, uint32_t lane_off, int kk) {
        return NERFX_LDS_VEC(float, tile + lane_off + kk * 256);
    }
    static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ Frag ones(int lane) { return (lane & 31) == 0 ? 1.0f : 0.0f; }
};

// One workgroup (4 wavefronts) per (task, sample split).  Wavefront w accumulates gradient tiles
// 2 (w >> 1), +1 against input tiles 2 (w & 1), +1; in a block that carries the bias, the two
// wavefronts with (w & 1) == 0 also accumulate their gradient tiles against the constant operand.
template <int PREC, int ACT>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WTask* __restrict__ tasks, int n_tasks, int chunks_per_split,
                                                       int n_chunks, uint64_t m_pad,
                                                       const typename WOperand<PREC>::T* __restrict__ ztape,
                                                       const typename WOperand<PREC>::T* __restrict__ gtape,
                                                       float* __restrict__ partial) {
    using O = WOperand<PREC>;
    using T = typename O::T;
    using V = typename O::V;
    constexpr int TILE_BYTES = O::CHUNK * 32 * (int)sizeof(T);     // one tile's chunk in LDS: 4 KiB
    static_assert(TILE_BYTES == 256 * 16, "one 16-byte copy per thread per tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [2 buffers][8 tiles][TILE_BYTES]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups go to the 8 XCDs round-robin by index, each XCD with its own L2.  Four consecutive
    // (split, task) pairs are mostly blocks of ONE layer over the same samples and share their
    // tiles, so they are given to four consecutive slots of one XCD: the second reader hits L2.
    uint32_t vid = blockIdx.x;
    if (vid < (gridDim.x & ~31u)) {
        const uint32_t x = vid & 7, q = vid >> 3;
        vid = (q >> 2) * 32 + x * 4 + (q & 3);
    }
    const int ks = vid / n_tasks, task = vid - ks * n_tasks;
    const WTask& tk = tasks[task];
    const int c0 = ks * chunks_per_split, c1 = min(c0 + chunks_per_split, n_chunks);

    // this thread's 16 bytes of every staged tile: a wave-uniform 64-bit base per tile (at this
    // split's first chunk) plus one 32-bit offset, so that a load needs no address register of its own
    // Absent tiles of a partial block alias the block's first tile (same addresses: cache hits, no
    // extra HBM traffic) so that staging is straight-line code: eight loads in flight per thread.
    const char* base[2 * WG_SIDE];
    bool act_on[2 * WG_SIDE];
#pragma unroll
    for (int s = 0; s < WG_SIDE; ++s) {
        const int sa = s < tk.n_a ? s : 0, sb = s < tk.n_b ? s : 0;
        base[s] = (const char*)(gtape + (uint64_t)tk.a_tile[sa] * m_pad * 32 + (uint64_t)c0 * O::CHUNK * 32);
        act_on[s] = false;
        base[WG_SIDE + s] = (const char*)(ztape + (uint64_t)tk.b_tile[sb] * m_pad * 32 + (uint64_t)c0 * O::CHUNK * 32);
        act_on[WG_SIDE + s] = tk.b_kind[sb] == WB_ACT;
    }
    const uint32_t voff = (uint32_t)tid * 16u;
    V staged[2 * WG_SIDE];
    auto stage = [&](int c) {
        const uint32_t off = voff + (uint32_t)(c - c0) * (uint32_t)(O::CHUNK * 32 * sizeof(T));   // within this split
#pragma unroll
#if defined(WGRAD_NT_LOADS)
        for (int s = 0; s < 2 * WG_SIDE; ++s) staged[s] = __builtin_nontemporal_load(reinterpret_cast<const V*>(base[s] + off));
#elif defined(WGRAD_DIAG_HALF_LOADS)             // timing diagnostic (wrong results): only the gradient tiles are fetched
        for (int s = 0; s < 2 * WG_SIDE; ++s) staged[s] = *reinterpret_cast<const V*>(base[s % WG_SIDE] + off);
#else
        for (int s = 0; s < 2 * WG_SIDE; ++s) staged[s] = *reinterpret_cast<const V*>(base[s] + off);
#endif
#ifdef WGRAD_DIAG_NO_LOAD                        // timing diagnostic (wrong results): what does the rest cost without the tape?
#pragma unroll
        for (int s = 0; s < 2 * WG_SIDE; ++s) staged[s] = *reinterpret_cast<const V*>(base[s] + voff);
#endif
    };
    auto commit = [&](int buf) {
        lds_bytes_t dst = (lds_bytes_t)smem + buf * (2 * WG_SIDE * TILE_BYTES) + tid * 16;
#pragma unroll
        for (int s = 0; s < 2 * WG_SIDE; ++s) {
            V v = staged[s];
            if (s >= WG_SIDE) {
                const V a = O::template act<ACT>(v);
                v = act_on[s] ? a : v;
            }
#ifdef WGRAD_DIAG_NO_COMMIT                      // timing diagnostic (wrong results): no LDS stores (the values stay live)
            asm volatile("" :: "v"(v));
#else
            *reinterpret_cast<__attribute__((address_space(3))) V*>(dst + s * TILE_BYTES) = v;
#endif
        }
    };

    const int oa = 2 * (wave >> 1), ib = 2 * (wave & 1);
    const bool a_on[2] = {oa < tk.n_a, oa + 1 < tk.n_a}, b_on[2] = {ib < tk.n_b, ib + 1 < tk.n_b};
    const uint32_t lane_off = O::lane_offset(lane);
    const bool do_bias = tk.bias && (wave & 1) == 0 && a_on[0];
    const typename O::Frag ones = O::ones(lane);
    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[o][r] = 0.0f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][n][r] = 0.0f;
    }

    // The chunk loop exists twice, with and without the bias products, chosen once per wavefront
    // (every wavefront meets the same barriers either way): no branch inside the multiply loop, and
    // the operands of step kk + 1 are requested before the products of step kk are issued, so a
    // wavefront's LDS latency runs under its own MFMAs instead of in front of them.
    const bool has_work = a_on[0] && b_on[0];                       // wave-uniform
    auto chunks = [&](auto bias_tag) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        constexpr int KK = O::CHUNK / O::KSTEP;
        if (c0 < c1) { stage(c0); commit(0); }
        __syncthreads();
        for (int c = c0; c < c1; ++c) {
            const int buf = (c - c0) & 1;
            if (c + 1 < c1) stage(c + 1);
            if (has_work) {
                lds_bytes_t base = (lds_bytes_t)smem + buf * (2 * WG_SIDE * TILE_BYTES);
                lds_bytes_t ta = base + oa * TILE_BYTES, tb = base + (WG_SIDE + ib) * TILE_BYTES;
                typename O::Frag f[2][4];
                f[0][0] = O::fetch(ta, lane_off, 0); f[0][1] = O::fetch(ta + TILE_BYTES, lane_off, 0);
                f[0][2] = O::fetch(tb, lane_off, 0); f[0][3] = O::fetch(tb + TILE_BYTES, lane_off, 0);
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const int cur = kk & 1, nxt = cur ^ 1;
                    if (kk + 1 < KK) {
                        f[nxt][0] = O::fetch(ta, lane_off, kk + 1); f[nxt][1] = O::fetch(ta + TILE_BYTES, lane_off, kk + 1);
                        f[nxt][2] = O::fetch(tb, lane_off, kk + 1); f[nxt][3] = O::fetch(tb + TILE_BYTES, lane_off, kk + 1);
                    }
                    O::mma(acc[0][0], f[cur][0], f[cur][2]);
                    O::mma(acc[0][1], f[cur][0], f[cur][3]);
                    O::mma(acc[1][0], f[cur][1], f[cur][2]);
                    O::mma(acc[1][1], f[cur][1], f[cur][3]);
                    if constexpr (BIAS) {
                        O::mma(accb[0], f[cur][0], ones);
                        O::mma(accb[1], f[cur][1], ones);
                    }
                }
            }
            if (c + 1 < c1) commit(buf ^ 1);
            __syncthreads();
        }
    };
    if (do_bias) chunks(std::true_type{});
    else chunks(std::false_type{});
    // products with absent (aliased) tiles are never stored
    float* dst = partial + ((size_t)vid * TASK_TILES) * 1024 + lane * 16;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int n = 0; n < 2; ++n)
            if (a_on[o] && b_on[n]) *reinterpret_cast<f32x16*>(dst + ((oa + o) * WG_SIDE + ib + n) * 1024) = acc[o][n];
    if (do_bias) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
            if (a_on[o]) *reinterpret_cast<f32x16*>(dst + (WG_SIDE * WG_SIDE + oa + o) * 1024) = accb[o];
    }
}

// Position p of a tile written in accumulator-register order -> feature of the 32-feature tile.
__device__ __forceinline__ int tile_position_feature(int p) { return (p & 3) + 8 * ((p & 15) >> 2) + 4 * (p >> 4); }

// Sums the splits of one task element in split order and writes it where it belongs.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WTask* __restrict__ tasks, int n_tasks, int n_split,
                                                           const float* __restrict__ partial, GradList grads,
                                                           float* __restrict__ found_inf) {
    constexpr int PER_TASK = TASK_TILES * 1024;
    const int task = blockIdx.x / (PER_TASK / 256), e = (blockIdx.x % (PER_TASK / 256)) * 256 + threadIdx.x;
    const WTask& tk = tasks[task];
    const int tile = e >> 10, lane = (e >> 4) & 63, r = e & 15;
    const bool bias = tile >= WG_SIDE * WG_SIDE;                     // the block's bias tiles come last
    const int o = bias ? tile - WG_SIDE * WG_SIDE : tile / WG_SIDE, n = bias ? 0 : tile % WG_SIDE;
    if (o >= tk.n_a || (bias ? !tk.bias : n >= tk.n_b)) return;
    const int po = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), pi = lane & 31;          // positions in the two tiles
    int row;
    if (tk.a_kind[o] == WO_LAYER) row = tk.a_row0[o] + tile_position_feature(po);
    else if (tk.a_kind[o] == WO_RGB) row = po < 3 ? po : -1;
    else row = po == 3 ? 0 : -1;
    if (row < 0 || row >= tk.out_dim) return;
    int col = -1;
    float* dst = grads.p[tk.w_index];
    if (bias) {
        if (pi != 0) return;
        dst = grads.p[tk.b_index];
    } else if (tk.b_map[n] == WM_ENC) {
        col = enc_slot_col(tk.enc_kind, tk.enc_n, tk.b_col0[n] + pi);
        if (col < 0) return;
    } else {
        const int f = tk.b_col0[n] + tile_position_feature(pi);
        if (f >= tk.chain_dim) return;
        col = tk.enc_cols + f;
    }
    if (!dst) return;
    float sum = 0.0f;
    for (int ks = 0; ks < n_split; ++ks) sum += partial[((size_t)ks * n_tasks + task) * PER_TASK + e];
    // the GradScaler's check (train.py:81) where the gradient is written: every element passes here once
    if (found_inf && !(fabsf(sum) <= 3.402823466e38f)) *found_inf = 1.0f;
    if (col < 0) dst[row] = sum;
    else dst[(size_t)row * tk.ld + col] = sum;
}

// The gradients of `features` and rgb.0 from the gradient of the folded layer (LayerPlan::folded):
// with W' = [W_r's encoder columns | W_rc W_f], b' = b_r + W_rc b_f (W_rc = W_r's chained columns),
//     dW_rc = dW'_c W_f^T + db' b_f^T,   dW_r's encoder columns = dW'_e,   db_r = db',
//     dW_f  = W_rc^T dW'_c,              db_f = W_rc^T db',
// with the weights as they were when the streams of this step's forward were packed (fold_dev).
// Workgroup (row, column group, part): part 0 = row r of rgb.0's gradient, part 1 = row k of `features`'
// (column `width` = its bias); a row is a vector (in LDS) times a matrix whose rows are read coalesced;
// four quarters of the sum in double, in index order, added in a fixed order.
__global__ __launch_bounds__(256) void unfold_kernel(nerfx_model::FoldSpots f, int half, int width, int ed_cols,
                                                     float* __restrict__ g_wr, float* __restrict__ g_br,
                                                     float* __restrict__ g_wf, float* __restrict__ g_bf,
                                                     float* __restrict__ found_inf) {
    __shared__ float s_vec[256];
    __shared__ double s_part[4][64];
    const int ld = ed_cols + width, part = blockIdx.z, row = blockIdx.x;
    const int n = part == 0 ? width : half;                          // length of the sums
    if (row >= (part == 0 ? half : width)) return;
    if (part == 0) {
        for (int j = threadIdx.x; j < n; j += 256) s_vec[j] = f.gw[(int64_t)row * ld + ed_cols + j];
        if (blockIdx.y == 0) {
            if (g_wr) for (int c = threadIdx.x; c < ed_cols; c += 256) g_wr[(int64_t)row * ld + c] = f.gw[(int64_t)row * ld + c];
            if (g_br && threadIdx.x == 0) g_br[row] = f.gb[row];
        }
    } else {
        for (int r = threadIdx.x; r < n; r += 256) s_vec[r] = f.wr[(int64_t)r * width + row];
    }
    __syncthreads();
    const int j = threadIdx.x & 63, quarter = threadIdx.x >> 6, c = 64 * blockIdx.y + j;
    const int q0 = quarter * (n / 4), q1 = q0 + n / 4;               // (width and half are multiples of 8)
    const int cols = part == 0 ? width : width + 1;
    double s = 0.0;
    if (c < cols) {
        if (part == 0) {                                             // . column c of W_f^T
#pragma unroll 8
            for (int i = q0; i < q1; ++i) s = fma((double)s_vec[i], (double)f.wft[(int64_t)i * width + c], s);
        } else if (c < width) {                                      // . column c of dW'_c
#pragma unroll 8
            for (int i = q0; i < q1; ++i) s = fma((double)s_vec[i], (double)f.gw[(int64_t)i * ld + ed_cols + c], s);
        } else {                                                     // . db'
            for (int i = q0; i < q1; ++i) s = fma((double)s_vec[i], (double)f.gb[i], s);
        }
    }
    s_part[quarter][j] = s;
    __syncthreads();
    if (quarter != 0 || c >= cols) return;
    double total = ((s_part[0][j] + s_part[1][j]) + s_part[2][j]) + s_part[3][j];
    if (part == 0) total = fma((double)f.gb[row], (double)f.bf[c], total);
    const float v = (float)total;
    if (found_inf && !(fabsf(v) <= 3.402823466e38f)) *found_inf = 1.0f;
    if (part == 0) { if (g_wr) g_wr[(int64_t)row * ld + ed_cols + c] = v; }
    else if (c < width) { if (g_wf) g_wf[(int64_t)row * width + c] = v; }
    else if (g_bf) g_bf[row] = v;
}

// Sample splits for a launch: as many as fit in WGRAD_ROUNDS full rounds of resident workgroups
// (two per CU), never one more -- a few workgroups left over would run a round of their own.
#ifndef WGRAD_ROUNDS
#define WGRAD_ROUNDS 2
#endif
static int resident_workgroups() {
    static int slots[LdsOptIn::MAX_DEVICES] = {};                   // per device: two workgroups per CU
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LdsOptIn::MAX_DEVICES) return 2 * 256;
    if (!__atomic_load_n(&slots[dev], __ATOMIC_RELAXED)) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        __atomic_store_n(&slots[dev], 2 * cus, __ATOMIC_RELAXED);
    }
    return __atomic_load_n(&slots[dev], __ATOMIC_RELAXED);
}
static void split_plan(int n_tasks, int n_chunks, int& n_split, int& chunks_per_split) {
    if (n_chunks <= 0 || n_tasks <= 0) { n_split = 0; chunks_per_split = 1; return; }
    int want = WGRAD_ROUNDS * resident_workgroups() / n_tasks;
    want = std::max(1, std::min(want, n_chunks));
    chunks_per_split = (n_chunks + want - 1) / want;
    n_split = (n_chunks + chunks_per_split - 1) / chunks_per_split;
}
static int chunk_samples(int prec) { return prec == NERFX_PREC_BF16 ? WOperand<NERFX_PREC_BF16>::CHUNK : WOperand<NERFX_PREC_FP32>::CHUNK; }

int wgrad_tasks_create(nerfx_model* m) {
    for (int prec = 0; prec < 2; ++prec) {
        std::vector<WTask> tasks;
        build_tasks(m->plan[prec], tasks);
        m->n_wtasks[prec] = (int)tasks.size();
        NERFX_HIP(hipMalloc(&m->wtasks_dev[prec], tasks.size() * sizeof(WTask)));
        NERFX_HIP(hipMemcpy(m->wtasks_dev[prec], tasks.data(), tasks.size() * sizeof(WTask), hipMemcpyHostToDevice));
    }
    return 0;
}

int64_t wgrad_partial_bytes(const nerfx_model* m, int prec, uint64_t m_pad) {
    int n_split, cps;
    split_plan(m->n_wtasks[prec], (int)(m_pad / chunk_samples(prec)), n_split, cps);
    return (int64_t)n_split * m->n_wtasks[prec] * (TASK_TILES * 1024) * (int64_t)sizeof(float);
}

static void launch_unfold(const nerfx_model* m, float* g_wr, float* g_br, float* g_wf, float* g_bf, float* found_inf,
                          hipStream_t stream) {
    const ModelPlan& p = m->plan[0];
    const LayerPlan& R = p.layer[2 * p.hd + 1];
    const int half = R.out_dim, width = R.chain_dim;
    hipLaunchKernelGGL(unfold_kernel, dim3(std::max(half, width), (width + 1 + 63) / 64, 2), dim3(256), 0, stream, m->fold_spots(),
                       half, width, R.enc_cols, g_wr, g_br, g_wf, g_bf, found_inf);
}

template <int PREC>
static int launch_wgrad_prec(const nerfx_model* m, uint64_t m_pad, const void* ztape, const void* gtape, float* partial,
                             const GradList& grads, float* found_inf, hipStream_t stream) {
    using O = WOperand<PREC>;
    using T = typename O::T;
    const int n_tasks = m->n_wtasks[PREC], n_chunks = (int)(m_pad / O::CHUNK);
    int n_split, cps;
    split_plan(n_tasks, n_chunks, n_split, cps);
    const WTask* tasks = (const WTask*)m->wtasks_dev[PREC];
    const size_t lds = 2 * 2 * WG_SIDE * O::CHUNK * 32 * sizeof(T);          // 64 KiB: two buffers of 8 tiles
#define NERFX_WGRAD(ACT_)                                                                                         \
    do {                                                                                                          \
        auto kern = wgrad_kernel<PREC, ACT_>;                                                                     \
        NERFX_OPT_IN_LDS(kern, lds);                                                                              \
        hipLaunchKernelGGL(kern, dim3(n_tasks * n_split), dim3(256), lds, stream, tasks, n_tasks, cps, n_chunks,   \
                           m_pad, (const T*)ztape, (const T*)gtape, partial);                                      \
    } while (0)
    switch (m->plan[PREC].desc.activ) {
        case NERFX_ACT_RELU: NERFX_WGRAD(NERFX_ACT_RELU); break;
        case NERFX_ACT_LEAKY_RELU: NERFX_WGRAD(NERFX_ACT_LEAKY_RELU); break;
        default: NERFX_WGRAD(NERFX_ACT_SILU); break;
    }
#undef NERFX_WGRAD
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(n_tasks * (TASK_TILES * 1024 / 256)), dim3(256), 0, stream, tasks,
                       n_tasks, n_split, partial, grads, found_inf);
    if (folds_features(PREC)) {
        const int base = 2 + 4 * m->plan[PREC].hd + 2;               // `features` weight, bias, rgb.0 weight, bias
        launch_unfold(m, grads.p[base + 2], grads.p[base + 3], grads.p[base], grads.p[base + 1], found_inf, stream);
    }
    return launch_status("wgrad kernels");
}

int launch_wgrad(const nerfx_model* m, int prec, uint64_t m_pad, const void* ztape, const void* gtape, float* partial,
                 float* const* grads, int n_grads, float* found_inf, hipStream_t stream) {
    GradList g{};
    for (int i = 0; i < n_grads; ++i) g.p[i] = grads[i];
    const nerfx_model::FoldSpots f = m->fold_spots();
    g.p[n_grads] = f.gw;
    g.p[n_grads + 1] = f.gb;
    return prec == NERFX_PREC_BF16 ? launch_wgrad_prec<NERFX_PREC_BF16>(m, m_pad, ztape, gtape, partial, g, found_inf, stream)
                                   : launch_wgrad_prec<NERFX_PREC_FP32>(m, m_pad, ztape, gtape, partial, g, found_inf, stream);
}

}  // namespace nerfx

using namespace nerfx;

extern "C" int nerfx_debug_folded_head(const nerfx_model_t* model, float* weight, float* bias, void* stream) {
    NERFX_REQUIRE(model && weight && bias, "NULL argument");
    const LayerPlan& R = model->plan[0].layer[2 * model->plan[0].hd + 1];
    const nerfx_model::FoldSpots f = model->fold_spots();
    NERFX_HIP(hipMemcpyAsync(weight, f.w, (size_t)R.out_dim * R.in_dim * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    NERFX_HIP(hipMemcpyAsync(bias, f.b, (size_t)R.out_dim * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int nerfx_debug_unfold_grads(const nerfx_model_t* model, const float* g_w, const float* g_b, float* g_rgb0_weight,
                                        float* g_rgb0_bias, float* g_features_weight, float* g_features_bias, void* stream) {
    NERFX_REQUIRE(model && g_w && g_b, "NULL argument");
    const LayerPlan& R = model->plan[0].layer[2 * model->plan[0].hd + 1];
    const nerfx_model::FoldSpots f = model->fold_spots();
    NERFX_HIP(hipMemcpyAsync(f.gw, g_w, (size_t)R.out_dim * R.in_dim * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    NERFX_HIP(hipMemcpyAsync(f.gb, g_b, (size_t)R.out_dim * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    launch_unfold(model, g_rgb0_weight, g_rgb0_bias, g_features_weight, g_features_bias, nullptr, (hipStream_t)stream);
    return launch_status("unfold_kernel");
}
