// K2b, part 2: weight and bias gradients of the MLP from the two training tapes.
//
//     dW_l[o][i] = sum over samples m of  Gz_l[o][m] * X_l[i][m]
//
// Both tapes are feature-major (one row per feature, samples contiguous), so Gz rows are the
// MFMA A operand and X rows the B operand with the SAMPLE index as the contraction: a lane
// feeds 16 bytes of its row per step, straight from memory.  X is rebuilt on load from the
// pre-activation tape (activation of z, or the stored encoder slot, or the constant 1 whose
// "weight" gradient is the bias gradient).  The sample range is split over wavefronts; every
// (task, split) wavefront writes a 64 x 64 fp32 partial, and a second kernel adds the partials
// in a fixed order and scatters them into torch-layout gradients -- deterministic, no atomics.
#include "mlp_fwd_impl.hpp"

#include <vector>

namespace nerfx {

enum { WB_RAW = 0, WB_ACT = 1, WB_ONES = 2 };       // how an input tile becomes X
enum { WM_ENC = 0, WM_CHAIN = 1, WM_BIAS = 2 };     // where an input tile's columns go

// A 64 x 64 block of one layer's gradient: two 32-row tiles of Gz against two 32-row tiles of X.
struct WTask {
    int a_row[2], a_valid[2];                      // gradient-tape rows; valid = 0: tile absent
    int b_row[2], b_valid[2], b_kind[2];           // pre-activation-tape rows
    int out0[2];                                   // first weight row of each output tile
    int map_kind[2], map0[2];                      // WM_*, first slot / feature of each input tile
    int w_index, b_index;                          // positions of weight and bias in the gradient list
    int enc_kind, enc_n, enc_cols, ld, chain_dim;
};

struct GradList { float* p[2 + 2 * MAX_LAYERS + 8]; };

static void build_tasks(const ModelPlan& p, std::vector<WTask>& out) {
    struct Tile { int row, valid, kind, map_kind, map0; };
    const int spu = slots_per_unit(p.prec);
    const int hd = p.hd, n_jobs = p.n_layers + 1;                   // every Linear + the density head
    for (int job = 0; job < n_jobs; ++job) {
        const bool sigma = job == p.n_layers;
        const LayerPlan& L = p.layer[sigma ? 2 * hd : job];          // density head: same input as `features`
        std::vector<Tile> outs, ins;
        int w_index;
        if (sigma) {
            outs.push_back({p.tape_gs_row, 1, 0, 0, 0});
            w_index = 2 + 4 * hd;
        } else {
            w_index = job < 2 * hd ? 2 + 2 * job : 2 + 4 * hd + 2 + 2 * (job - 2 * hd);
            if (job == p.n_layers - 1) outs.push_back({p.tape_go_row, 3, 0, 0, 0});
            else
                for (int t = 0; t < L.n_tiles; ++t)
                    if (L.out_dim > 32 * t) outs.push_back({p.tape_z_row[job] + 32 * t, std::min(32, L.out_dim - 32 * t), 0, 0, 32 * t});
        }
        if (!sigma && L.enc_units) {
            const int base = job == 2 * hd + 1 ? p.tape_ed_row : p.tape_ex_row;   // rgb.0 sees phi_d
            for (int s = 0; s < L.enc_units * spu; s += 32) ins.push_back({base + s, 32, WB_RAW, WM_ENC, s});
        }
        if (L.chain_tiles) {
            const int src = sigma ? 2 * hd - 1 : job - 1;            // the layer whose output feeds this one
            const int kind = src == 2 * hd ? WB_RAW : WB_ACT;        // `features` has no activation
            for (int t = 0; t < L.chain_tiles; ++t)
                if (L.chain_dim > 32 * t)
                    ins.push_back({p.tape_z_row[src] + 32 * t, std::min(32, L.chain_dim - 32 * t), kind, WM_CHAIN, 32 * t});
        }
        ins.push_back({0, 1, WB_ONES, WM_BIAS, 0});
        for (size_t o = 0; o < outs.size(); o += 2)
            for (size_t i = 0; i < ins.size(); i += 2) {
                WTask k{};
                for (int s = 0; s < 2; ++s) {
                    if (o + s < outs.size()) { k.a_row[s] = outs[o + s].row; k.a_valid[s] = outs[o + s].valid; k.out0[s] = outs[o + s].map0; }
                    if (i + s < ins.size()) {
                        const Tile& T = ins[i + s];
                        k.b_row[s] = T.row; k.b_valid[s] = T.valid; k.b_kind[s] = T.kind; k.map_kind[s] = T.map_kind; k.map0[s] = T.map0;
                    }
                }
                k.w_index = w_index; k.b_index = w_index + 1;
                k.enc_kind = L.enc_kind; k.enc_n = L.enc_n;
                k.enc_cols = sigma ? 0 : L.enc_cols; k.ld = sigma ? L.chain_dim : L.in_dim; k.chain_dim = L.chain_dim;
                out.push_back(k);
            }
    }
}

template <int PREC> struct WOperand;
template <> struct WOperand<NERFX_PREC_BF16> {
    using T = __bf16;
    using V = bf16x8;
    static constexpr int EPL = 8;
    static __device__ __forceinline__ V zero() { return __builtin_bit_cast(V, s16x8(0)); }
    static __device__ __forceinline__ V ones() { return __builtin_bit_cast(V, s16x8(0x3F80)); }
    static __device__ __forceinline__ void mma(f32x16& acc, const V& a, const V& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    template <int ACT>
    static __device__ __forceinline__ V act(const V& z) {
        if (ACT == NERFX_ACT_RELU) return Prec<NERFX_PREC_BF16>::relu(z);
        V o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)activate<NERFX_PREC_BF16, ACT>((float)z[e]);
        return o;
    }
};
template <> struct WOperand<NERFX_PREC_FP32> {
    using T = float;
    using V = f32x4;
    static constexpr int EPL = 4;
    static __device__ __forceinline__ V zero() { return V{0.0f, 0.0f, 0.0f, 0.0f}; }
    static __device__ __forceinline__ V ones() { return V{1.0f, 1.0f, 1.0f, 1.0f}; }
    static __device__ __forceinline__ void mma(f32x16& acc, const V& a, const V& b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    }
    template <int ACT>
    static __device__ __forceinline__ V act(const V& z) {
        V o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = activate<NERFX_PREC_FP32, ACT>(z[e]);
        return o;
    }
};

constexpr int WG_UNROLL = 4;                        // k-steps loaded together

// One wavefront per (task, sample split).  Lane (i = lane & 31, kh = lane >> 5) supplies elements
// [k + kh * EPL, k + (kh + 1) * EPL) of row i to both operands.
template <int PREC, int ACT>
__global__ __launch_bounds__(256) void wgrad_kernel(const WTask* __restrict__ tasks, int n_tasks, int n_split,
                                                    uint32_t k_per_split, uint32_t m_pad,
                                                    const typename WOperand<PREC>::T* __restrict__ ztape,
                                                    const typename WOperand<PREC>::T* __restrict__ gtape,
                                                    float* __restrict__ partial) {
    using O = WOperand<PREC>;
    using V = typename O::V;
    constexpr int KSTEP = 2 * O::EPL;
    const int wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wave_id >= n_tasks * n_split) return;
    const int ks = wave_id / n_tasks, task = wave_id - ks * n_tasks;
    const WTask& tk = tasks[task];
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const typename O::T *pa[2], *pb[2];
    bool a_on[2], b_on[2];
    int b_kind[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        a_on[s] = i < tk.a_valid[s];
        b_on[s] = i < tk.b_valid[s];
        b_kind[s] = tk.b_kind[s];
        pa[s] = gtape + (uint64_t)(tk.a_row[s] + (a_on[s] ? i : 0)) * m_pad + kh * O::EPL;
        pb[s] = ztape + (uint64_t)(tk.b_row[s] + (b_on[s] ? i : 0)) * m_pad + kh * O::EPL;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][n][r] = 0.0f;
    const uint32_t k0 = ks * k_per_split, k1 = min(k0 + k_per_split, m_pad);
    for (uint32_t k = k0; k < k1; k += WG_UNROLL * KSTEP) {
        V a[WG_UNROLL][2], b[WG_UNROLL][2];
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                a[u][s] = *reinterpret_cast<const V*>(pa[s] + k + u * KSTEP);
                b[u][s] = *reinterpret_cast<const V*>(pb[s] + k + u * KSTEP);
            }
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (!a_on[s]) a[u][s] = O::zero();
                if (b_kind[s] == WB_ACT) b[u][s] = O::template act<ACT>(b[u][s]);
                if (b_kind[s] == WB_ONES) b[u][s] = O::ones();
                if (!b_on[s]) b[u][s] = O::zero();
            }
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int n = 0; n < 2; ++n) O::mma(acc[o][n], a[u][o], b[u][n]);
        }
    }
    float* dst = partial + ((size_t)wave_id * 4) * 1024 + lane * 16;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int n = 0; n < 2; ++n) *reinterpret_cast<f32x16*>(dst + (o * 2 + n) * 1024) = acc[o][n];
}

// Sums the splits of one task element in split order and writes it where it belongs.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WTask* __restrict__ tasks, int n_tasks, int n_split,
                                                           const float* __restrict__ partial, GradList grads) {
    const int task = blockIdx.x >> 4, e = ((blockIdx.x & 15) << 8) + threadIdx.x;       // 4096 elements per task
    const WTask& tk = tasks[task];
    const int tile = e >> 10, lane = (e >> 4) & 63, r = e & 15;
    const int o = tile >> 1, n = tile & 1;
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;              // out row, in row of the tile
    if (i >= tk.a_valid[o] || j >= tk.b_valid[n]) return;
    float sum = 0.0f;
    for (int ks = 0; ks < n_split; ++ks) sum += partial[((size_t)ks * n_tasks + task) * 4096 + e];
    const int row = tk.out0[o] + i;
    if (tk.map_kind[n] == WM_BIAS) {
        if (grads.p[tk.b_index]) grads.p[tk.b_index][row] = sum;
        return;
    }
    int col;
    if (tk.map_kind[n] == WM_ENC) {
        col = enc_slot_col(tk.enc_kind, tk.enc_n, tk.map0[n] + j);
    } else {
        const int f = tk.map0[n] + j;
        col = f < tk.chain_dim ? tk.enc_cols + f : -1;
    }
    if (col >= 0 && grads.p[tk.w_index]) grads.p[tk.w_index][(size_t)row * tk.ld + col] = sum;
}

// Sample splits for a launch: enough wavefronts to fill the device, at least 256 samples each.
static void split_plan(int n_tasks, uint64_t m_pad, int& n_split, uint32_t& k_per_split) {
    const uint64_t chunk = 256;
    const uint64_t max_split = (m_pad + chunk - 1) / chunk;
    uint64_t want = (4096 + n_tasks - 1) / n_tasks;
    want = std::max<uint64_t>(1, std::min(want, max_split));
    k_per_split = (uint32_t)(((m_pad + want - 1) / want + chunk - 1) / chunk * chunk);
    n_split = (int)((m_pad + k_per_split - 1) / k_per_split);
}

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
    int n_split; uint32_t kps;
    split_plan(m->n_wtasks[prec], m_pad, n_split, kps);
    return (int64_t)n_split * m->n_wtasks[prec] * 4096 * (int64_t)sizeof(float);
}

template <int PREC>
static int launch_wgrad_prec(const nerfx_model* m, uint64_t m_pad, const void* ztape, const void* gtape, float* partial,
                             const GradList& grads, hipStream_t stream) {
    using T = typename WOperand<PREC>::T;
    const int n_tasks = m->n_wtasks[PREC];
    int n_split; uint32_t kps;
    split_plan(n_tasks, m_pad, n_split, kps);
    const WTask* tasks = (const WTask*)m->wtasks_dev[PREC];
    const int waves = n_tasks * n_split, grid = (waves + 3) / 4;
#define NERFX_WGRAD(ACT_) \
    hipLaunchKernelGGL((wgrad_kernel<PREC, ACT_>), dim3(grid), dim3(256), 0, stream, tasks, n_tasks, n_split, kps, \
                       (uint32_t)m_pad, (const T*)ztape, (const T*)gtape, partial)
    switch (m->plan[PREC].desc.activ) {
        case NERFX_ACT_RELU: NERFX_WGRAD(NERFX_ACT_RELU); break;
        case NERFX_ACT_LEAKY_RELU: NERFX_WGRAD(NERFX_ACT_LEAKY_RELU); break;
        default: NERFX_WGRAD(NERFX_ACT_SILU); break;
    }
#undef NERFX_WGRAD
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(n_tasks * 16), dim3(256), 0, stream, tasks, n_tasks, n_split, partial, grads);
    return launch_status("wgrad kernels");
}

int launch_wgrad(const nerfx_model* m, int prec, uint64_t m_pad, const void* ztape, const void* gtape, float* partial,
                 float* const* grads, int n_grads, hipStream_t stream) {
    GradList g{};
    for (int i = 0; i < n_grads; ++i) g.p[i] = grads[i];
    return prec == NERFX_PREC_BF16 ? launch_wgrad_prec<NERFX_PREC_BF16>(m, m_pad, ztape, gtape, partial, g, stream)
                                   : launch_wgrad_prec<NERFX_PREC_FP32>(m, m_pad, ztape, gtape, partial, g, stream);
}

}  // namespace nerfx
