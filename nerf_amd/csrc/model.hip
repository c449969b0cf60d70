// Model handle: plans the unit stream of a radiance field and packs the caller's
// fp32 torch-layout weights into it on the device (no host round trip, so a
// refresh after every optimizer step is one short kernel sequence on the stream).
#include "mlp.hpp"

#include <new>

namespace nerfx {

int wgrad_tasks_create(nerfx_model* m);             // mlp_wgrad.hip

int make_plan(const nerfx_model_desc_t& d, int layout, ModelPlan& p) {
    const int prec = layout_prec(layout);
    NERFX_REQUIRE(d.width >= 16 && d.width <= 256 && d.width % 16 == 0,
                  "width %d unsupported (16..256, multiple of 16)", d.width);
    NERFX_REQUIRE(d.depth >= 2 && d.depth % 2 == 0 && d.depth <= 16, "depth %d unsupported (even, 2..16)", d.depth);
    NERFX_REQUIRE(d.activ >= NERFX_ACT_RELU && d.activ <= NERFX_ACT_SILU, "unknown activation %d", d.activ);
    for (int s = 0; s < 2; ++s) {
        const int kind = s ? d.enc_d_kind : d.enc_x_kind, n = s ? d.enc_d_n : d.enc_x_n;
        NERFX_REQUIRE(kind == NERFX_ENC_POSITIONAL || kind == NERFX_ENC_FOURIER, "unknown encoder kind %d", kind);
        NERFX_REQUIRE(n >= 1 && (kind == NERFX_ENC_POSITIONAL ? n : 3 * n) <= MAX_TABLE, "encoder size %d unsupported", n);
    }
    p.desc = d;
    p.prec = prec;
    auto pow2_tiles = [](int dim) { int t = 1; while (32 * t < dim) t *= 2; return t; };
    p.wt = pow2_tiles(d.width);          // tiles are padded to a power of two so that a
    p.rt = pow2_tiles(d.width / 2);      // k-block of n_tiles units never straddles a ring chunk
    const int tiles_per_32 = 32 / rows_per_tile(prec);              // output tiles per 32 rows
    p.hd = d.depth / 2;
    p.n_layers = 2 * p.hd + 3;
    const int ex_cols = enc_width(d.enc_x_kind, d.enc_x_n), ed_cols = enc_width(d.enc_d_kind, d.enc_d_n);
    const int ex_units = enc_units(d.enc_x_kind, d.enc_x_n, prec), ed_units = enc_units(d.enc_d_kind, d.enc_d_n, prec);
    int64_t unit = 0;
    for (int li = 0; li < p.n_layers; ++li) {
        LayerPlan& L = p.layer[li];
        L = LayerPlan{};
        L.n_tiles = p.wt * tiles_per_32; L.out_dim = d.width; L.chain_tiles = p.wt; L.chain_dim = d.width;
        const bool with_x = li == 0 || (li == p.hd && d.resid);
        if (with_x) { L.enc_units = ex_units; L.enc_kind = d.enc_x_kind; L.enc_n = d.enc_x_n; L.enc_cols = ex_cols; }
        if (li == 0) { L.chain_tiles = 0; L.chain_dim = 0; }
        if (li == 2 * p.hd + 1) {                                   // rgb.0
            L.n_tiles = p.rt * tiles_per_32; L.out_dim = d.width / 2;
            L.enc_units = ed_units; L.enc_kind = d.enc_d_kind; L.enc_n = d.enc_d_n; L.enc_cols = ed_cols;
        }
        if (li == 2 * p.hd + 2) {                                   // rgb.2
            L.n_tiles = 1; L.out_dim = 3; L.chain_tiles = p.rt; L.chain_dim = d.width / 2;
        }
        L.in_dim = L.enc_cols + L.chain_dim;
        L.group = group_tiles(prec, L.n_tiles);
        L.enc_major = layout == NERFX_STREAM_ENCMAJOR && L.enc_units > 0;
        L.enc_seg = L.enc_units / enc_block(prec) * pad_to_piece(enc_block(prec) * L.group);   // block-wise padding
        L.chain_seg = pad_to_piece(L.chain_tiles * units_per_tile(prec) * L.group);
        L.unit0 = unit;
        unit += units_of(L);
        if (li == 2 * p.hd - 1) p.trunk_units = unit;
    }
    p.total_units = unit;
    p.m16_static = 0;
    if (layout == NERFX_LAYOUT_M16 && m16_static_depth(p.hd) && p.trunk_units == m16_trunk_units(p.hd) &&
        p.total_units == m16_total_units(p.hd)) {
        p.m16_static = 1;
        p.total_units += M16_STATIC_PAD;                            // never read by an MFMA, only carried through the ring
    }
    // backward chain
    p.n_bwd = 2 * p.hd + 2;
    unit = 0;
    for (int j = 0; j < p.n_bwd; ++j) {
        const int src = 2 * p.hd + 2 - j;                           // rgb.2, rgb.0, features, trunk last .. 1
        const LayerPlan& F = p.layer[src];
        LayerPlan& L = p.bwd[j];
        L = LayerPlan{};
        L.n_tiles = F.chain_tiles; L.out_dim = F.chain_dim;
        L.chain_tiles = F.n_tiles; L.chain_dim = F.out_dim;
        L.transposed = 1; L.src_layer = src; L.src_ld = F.in_dim; L.src_col0 = F.enc_cols;
        L.group = group_tiles(prec, L.n_tiles);
        L.chain_seg = pad_to_piece(L.chain_tiles * units_per_tile(prec) * L.group);
        L.unit0 = unit;
        unit += units_of(L);
    }
    p.bwd_units = unit;
    // tape tiles
    int tile = 0;
    p.tape_ex_tile = tile; tile += ex_units * slots_per_unit(prec) / 32;
    p.tape_ed_tile = tile; tile += ed_units * slots_per_unit(prec) / 32;
    for (int li = 0; li < p.n_layers - 1; ++li) { p.tape_z_tile[li] = tile; tile += p.layer[li].n_tiles; }
    p.tape_heads_tile = p.tape_z_tile[p.n_layers - 1] = tile; tile += 1;
    p.tape_tiles = tile;
    p.sigw_off = p.n_layers * BIAS_STRIDE;
    p.sigb_off = p.sigw_off + BIAS_STRIDE;
    auto table_floats = [](int kind, int n) { return 4 * enc_pair_slots(kind, n); };
    p.tabx_off = p.sigb_off + 4;
    p.tabd_off = p.tabx_off + table_floats(d.enc_x_kind, d.enc_x_n);
    p.const_floats = p.tabd_off + table_floats(d.enc_d_kind, d.enc_d_n);
    return 0;
}

// One thread per packed 16-byte lane slice of one layer (padding units included).
template <int PREC>
__device__ __forceinline__ void pack_layer(const float* __restrict__ W, const LayerPlan& L, void* __restrict__ stream) {
    constexpr int EPL = PREC == NERFX_PREC_FP32 ? 4 : 8;          // elements per lane per unit
    constexpr int ROWS = rows_per_tile(PREC), EB = enc_block(PREC);
    const int group_units = L.enc_seg + L.chain_seg;
    const int64_t n_slices = (int64_t)units_of(L) * 64;
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < n_slices; s += (int64_t)gridDim.x * 256) {
        const int lane = (int)(s & 63);
        const int64_t unit = s >> 6;
        int grp = (int)(unit / group_units);
        int in_group = (int)(unit - (int64_t)grp * group_units);
        const int block_units = pad_to_piece(EB * L.group);          // ENC_BLOCK k-units of one group, padded
        if (L.enc_major) {                                          // [chunk][group][block] ... then [group][chained]
            const int n_groups = L.n_tiles / L.group, enc_total = n_groups * L.enc_seg;
            if (unit < enc_total) {
                const int chunk = (int)unit / (n_groups * block_units), rem = (int)unit - chunk * n_groups * block_units;
                grp = rem / block_units;
                in_group = chunk * block_units + rem % block_units;
            } else {
                grp = ((int)unit - enc_total) / L.chain_seg;
                in_group = L.enc_seg + ((int)unit - enc_total) % L.chain_seg;
            }
        }
        const bool enc = in_group < L.enc_seg;
        int ku, t;                                                  // [k-unit][tile in group]
        bool live;
        if (enc) {                                                  // ENC_BLOCK k-units per padded block
            const int eb = in_group / block_units, within = in_group - eb * block_units;
            ku = eb * EB + within / L.group;
            t = within % L.group;
            live = within < EB * L.group;
        } else {
            in_group -= L.enc_seg;
            ku = in_group / L.group;
            t = in_group - ku * L.group;
            live = ku < L.chain_tiles * units_per_tile(PREC);
        }
        const int row = ROWS * (grp * L.group + t) + lane % ROWS, h = lane / ROWS;
        float v[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int col;
            if (enc) {
                col = enc_slot_col(L.enc_kind, L.enc_n, enc_unit_slot(PREC, ku, h, e));
            } else {
                const int f = chain_unit_feature(PREC, ku, h, e);
                col = f < L.chain_dim ? L.enc_cols + f : -1;
            }
            if (L.transposed) {
                const int f = chain_unit_feature(PREC, ku, h, e);
                v[e] = (live && row < L.out_dim && f < L.chain_dim) ? W[(int64_t)f * L.src_ld + L.src_col0 + row] : 0.0f;
            } else {
                v[e] = (live && row < L.out_dim && col >= 0) ? W[(int64_t)row * L.in_dim + col] : 0.0f;
            }
        }
        char* dst = (char*)stream + (L.unit0 + unit) * UNIT_BYTES + lane * 16;
        if (PREC != NERFX_PREC_FP32) {
            __bf16 b[8];
#pragma unroll
            for (int e = 0; e < EPL; ++e) b[e] = (__bf16)v[e];
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(b);
        } else {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[EPL > 3 ? 3 : 0]);
        }
    }
}

struct PackSources { const float* w[2 * MAX_LAYERS]; };

// All layers of both streams of one precision in one launch: blockIdx.y picks the layer plan
// (forward layers first, then the backward chain's), blockIdx.x strides over its 16-byte slices.
template <int PREC>
__global__ __launch_bounds__(256) void pack_streams_kernel(const LayerPlan* __restrict__ plans, int n_fwd, PackSources src,
                                                           void* __restrict__ fwd, void* __restrict__ bwd) {
    const LayerPlan L = plans[blockIdx.y];
    pack_layer<PREC>(src.w[blockIdx.y], L, blockIdx.y < n_fwd ? fwd : bwd);
}

struct VectorJobs { const float* src[MAX_LAYERS + 2]; int n[MAX_LAYERS + 2], cap[MAX_LAYERS + 2], off[MAX_LAYERS + 2]; };

// Biases, density-head weights and bias into the const block: blockIdx.y picks the vector.
__global__ __launch_bounds__(256) void pack_vectors_kernel(VectorJobs jobs, float* __restrict__ consts) {
    const int j = blockIdx.y;
    for (int i = threadIdx.x; i < jobs.cap[j]; i += 256) consts[jobs.off[j] + i] = i < jobs.n[j] ? jobs.src[j][i] : 0.0f;
}

// Pair-coefficient table of one encoder (see enc_pair_slots): entry k = (cx, cy, cz, 0).
__global__ __launch_bounds__(256) void pack_pair_table_kernel(const float* __restrict__ src, int kind, int n, int entries,
                                                              float* __restrict__ dst) {
    for (int k = blockIdx.x * 256 + threadIdx.x; k < entries; k += gridDim.x * 256) {
        const int q = k - enc_raw_slots(kind) / 2;
        float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (q >= 0 && q < enc_pairs(kind, n)) {
            if (kind == NERFX_ENC_POSITIONAL) {
                const int f = q / 3, axis = q - 3 * f;
                const float w = src[f];
                c = make_float4(axis == 0 ? w : 0.0f, axis == 1 ? w : 0.0f, axis == 2 ? w : 0.0f, 0.0f);
            } else {
                c = make_float4(src[3 * q], src[3 * q + 1], src[3 * q + 2], 0.0f);
            }
        }
        reinterpret_cast<float4*>(dst)[k] = c;
    }
}

static int tensor_index(const ModelPlan& p, int li, bool bias) {
    const int hd = p.hd;
    int base;
    if (li < 2 * hd) base = 2 + 2 * li;                             // fc_1.*, fc_2.*
    else if (li == 2 * hd) base = 2 + 4 * hd + 2;                   // features (after sigma)
    else if (li == 2 * hd + 1) base = 2 + 4 * hd + 4;               // rgb.0
    else base = 2 + 4 * hd + 6;                                     // rgb.2
    return base + (bias ? 1 : 0);
}

// Packs the const block and the unit streams of the precisions in `mask` (bit NERFX_PREC_*);
// `tables`: also the encoders' coefficient tables (buffers, not parameters: an optimizer step
// never changes them).
static int pack_all(nerfx_model* m, const float* const* tensors, int mask, hipStream_t stream, bool tables = true) {
    const ModelPlan& p0 = m->plan[0];
    const nerfx_model_desc_t& d = p0.desc;
    for (int i = 0; i < m->n_tensors; ++i) NERFX_REQUIRE(tensors[i] != nullptr, "tensor %d is NULL", i);
    if ((mask >> NERFX_PREC_BF16 & 1) && m->has_m16) {              // the bf16 inference stream of the M16 kernel
        const ModelPlan& p = m->plan[NERFX_LAYOUT_M16];
        PackSources src{};
        for (int li = 0; li < p.n_layers; ++li) src.w[li] = tensors[tensor_index(p, li, false)];
        hipLaunchKernelGGL(pack_streams_kernel<NERFX_LAYOUT_M16>, dim3(64, p.n_layers), dim3(256), 0, stream,
                           m->plans_dev[NERFX_LAYOUT_M16], p.n_layers, src, m->stream_dev[NERFX_LAYOUT_M16], nullptr);
    }
    if ((mask >> NERFX_PREC_BF16 & 1) && m->has_encmajor) {         // the bf16 inference stream of the wide-encoder kernel
        const ModelPlan& p = m->plan[NERFX_STREAM_ENCMAJOR];
        PackSources src{};
        for (int li = 0; li < p.n_layers; ++li) src.w[li] = tensors[tensor_index(p, li, false)];
        hipLaunchKernelGGL(pack_streams_kernel<NERFX_PREC_BF16>, dim3(64, p.n_layers), dim3(256), 0, stream,
                           m->plans_dev[NERFX_STREAM_ENCMAJOR], p.n_layers, src, m->stream_dev[NERFX_STREAM_ENCMAJOR], nullptr);
    }
    for (int prec = 0; prec < 2; ++prec) {
        if (!(mask >> prec & 1)) continue;
        const ModelPlan& p = m->plan[prec];
        PackSources src{};
        for (int li = 0; li < p.n_layers + p.n_bwd; ++li)
            src.w[li] = tensors[tensor_index(p, li >= p.n_layers ? p.bwd[li - p.n_layers].src_layer : li, false)];
        const dim3 grid(64, p.n_layers + p.n_bwd);
        if (prec == NERFX_PREC_BF16)
            hipLaunchKernelGGL(pack_streams_kernel<NERFX_PREC_BF16>, grid, dim3(256), 0, stream, m->plans_dev[prec], p.n_layers, src,
                               m->stream_dev[prec], m->stream_bwd_dev[prec]);
        else
            hipLaunchKernelGGL(pack_streams_kernel<NERFX_PREC_FP32>, grid, dim3(256), 0, stream, m->plans_dev[prec], p.n_layers, src,
                               m->stream_dev[prec], m->stream_bwd_dev[prec]);
    }
    VectorJobs jobs{};
    int nj = 0;
    auto vec = [&](const float* src, int n, int cap, int off) { jobs.src[nj] = src; jobs.n[nj] = n; jobs.cap[nj] = cap; jobs.off[nj] = off; ++nj; };
    for (int li = 0; li < p0.n_layers; ++li)
        vec(tensors[tensor_index(p0, li, true)], p0.layer[li].out_dim, BIAS_STRIDE, li * BIAS_STRIDE);
    vec(tensors[2 + 4 * p0.hd], d.width, BIAS_STRIDE, p0.sigw_off);
    vec(tensors[2 + 4 * p0.hd + 1], 1, 4, p0.sigb_off);
    hipLaunchKernelGGL(pack_vectors_kernel, dim3(1, nj), dim3(256), 0, stream, jobs, m->const_dev);
    auto table = [&](const float* src, int kind, int n, int off) {
        const int entries = enc_pair_slots(kind, n);
        hipLaunchKernelGGL(pack_pair_table_kernel, dim3((entries + 255) / 256), dim3(256), 0, stream, src, kind, n, entries,
                           m->const_dev + off);
    };
    if (tables) {
        table(tensors[0], d.enc_x_kind, d.enc_x_n, p0.tabx_off);
        table(tensors[1], d.enc_d_kind, d.enc_d_n, p0.tabd_off);
    }
    return launch_status("pack kernels");
}

}  // namespace nerfx

using namespace nerfx;

extern "C" int nerfx_model_create(const nerfx_model_desc_t* desc, const float* const* tensors, int n_tensors,
                                  void* stream, nerfx_model_t** out) {
    NERFX_REQUIRE(desc && tensors && out, "NULL argument");
    *out = nullptr;
    nerfx_model* m = new (std::nothrow) nerfx_model{};
    if (!m) return fail(NERFX_E_NOMEM, "out of host memory");
    for (int prec = 0; prec < N_LAYOUTS; ++prec)
        if (int e = make_plan(*desc, prec, m->plan[prec])) { delete m; return e; }
    m->has_m16 = m16_supported(*desc);
    m->has_encmajor = encmajor_supported(*desc);
    m->n_tensors = 2 + 2 * desc->depth + 8;
    if (n_tensors != m->n_tensors) {
        const int want = m->n_tensors;
        delete m;
        return fail(NERFX_E_ARG, "expected %d tensors for depth %d, got %d", want, desc->depth, n_tensors);
    }
    hipError_t e = hipGetDevice(&m->device);
    for (int prec = 0; prec < N_LAYOUTS && e == hipSuccess; ++prec)
        if (m->has_layout(prec))
            e = hipMalloc(&m->stream_dev[prec], (size_t)m->plan[prec].total_units * UNIT_BYTES);
    if (e == hipSuccess && m->has_m16)                              // (its padding units are never packed)
        e = hipMemset(m->stream_dev[NERFX_LAYOUT_M16], 0, (size_t)m->plan[NERFX_LAYOUT_M16].total_units * UNIT_BYTES);
    for (int prec = 0; prec < 2 && e == hipSuccess; ++prec)
        e = hipMalloc(&m->stream_bwd_dev[prec], (size_t)m->plan[prec].bwd_units * UNIT_BYTES);
    if (e == hipSuccess) e = hipMalloc((void**)&m->const_dev, (size_t)m->plan[0].const_floats * sizeof(float));
    if (e != hipSuccess) {
        nerfx_model_destroy(m);
        return fail(e == hipErrorOutOfMemory ? NERFX_E_NOMEM : NERFX_E_HIP, "model allocation failed: %s", hipGetErrorString(e));
    }
    for (int prec = 0; prec < N_LAYOUTS && e == hipSuccess; ++prec) {   // the layer plans the pack kernel reads
        if (!m->has_layout(prec)) continue;
        const ModelPlan& p = m->plan[prec];
        LayerPlan all[2 * MAX_LAYERS];
        for (int li = 0; li < p.n_layers + p.n_bwd; ++li) all[li] = li < p.n_layers ? p.layer[li] : p.bwd[li - p.n_layers];
        const size_t bytes = (size_t)(p.n_layers + p.n_bwd) * sizeof(LayerPlan);
        e = hipMalloc((void**)&m->plans_dev[prec], bytes);
        if (e == hipSuccess) e = hipMemcpy(m->plans_dev[prec], all, bytes, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        nerfx_model_destroy(m);
        return fail(NERFX_E_HIP, "model set-up failed: %s", hipGetErrorString(e));
    }
    if (int rc = wgrad_tasks_create(m)) { nerfx_model_destroy(m); return rc; }
    if (int rc = pack_all(m, tensors, 3, (hipStream_t)stream)) { nerfx_model_destroy(m); return rc; }
    *out = m;
    return 0;
}

extern "C" int nerfx_model_refresh(nerfx_model_t* m, const float* const* tensors, int n_tensors, void* stream) {
    NERFX_REQUIRE(m && tensors, "NULL argument");
    NERFX_REQUIRE(n_tensors == m->n_tensors, "expected %d tensors, got %d", m->n_tensors, n_tensors);
    return pack_all(m, tensors, 3, (hipStream_t)stream);
}

extern "C" int nerfx_model_refresh_precision(nerfx_model_t* m, const float* const* tensors, int n_tensors, int precision,
                                             void* stream) {
    NERFX_REQUIRE(m && tensors, "NULL argument");
    NERFX_REQUIRE(n_tensors == m->n_tensors, "expected %d tensors, got %d", m->n_tensors, n_tensors);
    NERFX_REQUIRE(precision == NERFX_PREC_FP32 || precision == NERFX_PREC_BF16, "unknown precision %d", precision);
    return pack_all(m, tensors, 1 << precision, (hipStream_t)stream, false);
}

extern "C" void nerfx_model_destroy(nerfx_model_t* m) {
    if (!m) return;
    for (int prec = 0; prec < N_LAYOUTS; ++prec)
        if (m->stream_dev[prec]) (void)hipFree(m->stream_dev[prec]);
    for (int prec = 0; prec < 2; ++prec)
        if (m->stream_bwd_dev[prec]) (void)hipFree(m->stream_bwd_dev[prec]);
    for (int prec = 0; prec < 2; ++prec)
        if (m->wtasks_dev[prec]) (void)hipFree(m->wtasks_dev[prec]);
    for (int prec = 0; prec < N_LAYOUTS; ++prec)
        if (m->plans_dev[prec]) (void)hipFree(m->plans_dev[prec]);
    if (m->const_dev) (void)hipFree(m->const_dev);
    delete m;
}
