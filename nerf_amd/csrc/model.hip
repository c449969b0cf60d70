// Model handle: plans the unit stream of a radiance field and packs the caller's
// fp32 torch-layout weights into it on the device (no host round trip, so a
// refresh after every optimizer step is one short kernel sequence on the stream).
#include "mlp.hpp"

#include <new>

namespace nerfx {

int wgrad_tasks_create(nerfx_model* m);             // mlp_wgrad.hip
int narrow_pack(const NarrowPlan& p, const float* const* tensors, void* dst, hipStream_t stream);   // narrow.hip

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
        L.absent = folds_features(layout) && li == 2 * p.hd;        // `features`: folded into rgb.0 (see LayerPlan)
        L.folded = folds_features(layout) && li == 2 * p.hd + 1;
        L.group = group_tiles(prec, L.n_tiles);
        L.enc_major = layout == NERFX_STREAM_ENCMAJOR && L.enc_units > 0;
        L.enc_seg = L.enc_units / enc_block(prec) * pad_to_piece(enc_block(prec) * L.group);   // block-wise padding
        L.chain_seg = pad_to_piece(L.chain_tiles * units_per_tile(prec) * L.group);
        L.unit0 = unit;
        if (!L.absent) unit += units_of(L);
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
    const bool fold = folds_features(layout);
    p.n_bwd = 2 * p.hd + (fold ? 1 : 2);
    unit = 0;
    for (int j = 0; j < p.n_bwd; ++j) {
        const int src = !fold || j < 2 ? 2 * p.hd + 2 - j : 2 * p.hd + 1 - j;   // rgb.2, rgb.0, features unless folded, trunk last .. 1
        const LayerPlan& F = p.layer[src];
        LayerPlan& L = p.bwd[j];
        L = LayerPlan{};
        L.n_tiles = F.chain_tiles; L.out_dim = F.chain_dim;
        L.chain_tiles = F.n_tiles; L.chain_dim = F.out_dim;
        L.transposed = 1; L.src_layer = src; L.src_ld = F.in_dim; L.src_col0 = F.enc_cols;
        L.folded = F.folded;
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
    for (int li = 0; li < p.n_layers - 1; ++li) { p.tape_z_tile[li] = tile; tile += p.layer[li].absent ? 0 : p.layer[li].n_tiles; }
    p.tape_heads_tile = p.tape_z_tile[p.n_layers - 1] = tile; tile += 1;
    p.tape_tiles = tile;
    p.sigw_off = p.n_layers * BIAS_STRIDE;
    p.sigb_off = p.sigw_off + BIAS_STRIDE;
    auto table_floats = [](int kind, int n) { return 4 * enc_pair_slots(kind, n); };
    p.tabx_off = p.sigb_off + 4;
    p.tabd_off = p.tabx_off + table_floats(d.enc_x_kind, d.enc_x_n);
    p.foldb_off = p.tabd_off + table_floats(d.enc_d_kind, d.enc_d_n);
    p.const_floats = p.foldb_off + BIAS_STRIDE / 2;
    return 0;
}

// The kernels' ReLU is an integer maximum on the value's bits, which keeps a NaN only while its sign
// bit is clear.  NaNs made on the device have it clear; a parameter may hold one with the bit set
// (x86 writes such a NaN for inf - inf), so every parameter enters the packed copies through this.
__device__ __forceinline__ float sign_free_nan(float v) { return v != v ? __builtin_bit_cast(float, 0x7FC00000u) : v; }

// One thread per packed 16-byte lane slice of one layer (padding units included).
template <int PREC>
__device__ __forceinline__ void pack_layer(const float* __restrict__ W, const LayerPlan& L, void* __restrict__ stream,
                                           uint32_t nan_class, uint32_t* nan_flags) {
    bool saw_nan = false;
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
            saw_nan |= v[e] != v[e];
            v[e] = sign_free_nan(v[e]);
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
    if (saw_nan && nan_class) atomicOr(nan_flags, nan_class);
}

// Which outputs a NaN in a parameter reaches in the reference (model.py:68-78): the trunk feeds both
// heads, `features` / rgb.* only the colour, the density head only the density.
enum { NAN_SIGMA = 1, NAN_RGB = 2 };

// Everything one refresh packs, as ONE launch: blockIdx.y picks a job -- a layer of one stream
// (forward streams of the layouts in use, then the backward chains'), or one vector of the const
// block (biases, density-head weights and bias) -- and blockIdx.x strides over a layer's 16-byte
// slices.  A NaN parameter is recorded in `nan_set` (NAN_* bits): the MLP kernels' ReLU keeps only
// NaNs whose sign bit is clear and the matrix cores return others, so the kernels are told instead and
// make the outputs such a parameter reaches NaN, as they are in the reference.  `nan_clear` is the
// word the NEXT refresh will set bits in; this launch zeroes it (nothing reads it meanwhile).
struct PackJob { uint8_t layout, plan_index, tensor, nan_class; };
constexpr int MAX_PACK_JOBS = 6 * MAX_LAYERS;
constexpr int MAX_VECTORS = MAX_LAYERS + 3;
struct PackLaunch {
    const float* tensors[2 + 2 * 16 + 8 + 2];              // the caller's, then the folded rgb.0 weight and bias
    const LayerPlan* plans[N_LAYOUTS];
    void* fwd[N_LAYOUTS];
    void* bwd[N_LAYOUTS];
    int n_fwd[N_LAYOUTS];                                  // plans below this index go to fwd, the rest to bwd
    PackJob job[MAX_PACK_JOBS];
    int n_jobs, n_vec;
    uint8_t vec_tensor[MAX_VECTORS], vec_class[MAX_VECTORS];
    int vec_n[MAX_VECTORS], vec_cap[MAX_VECTORS], vec_off[MAX_VECTORS];
    float* consts;
    uint32_t *nan_set, *nan_clear;
};

__global__ __launch_bounds__(256) void pack_model_kernel(const PackLaunch a) {
    const int j = blockIdx.y;
    if (j < a.n_jobs) {
        const PackJob job = a.job[j];
        const LayerPlan L = a.plans[job.layout][job.plan_index];
        void* dst = job.plan_index < a.n_fwd[job.layout] ? a.fwd[job.layout] : a.bwd[job.layout];
        const float* W = a.tensors[job.tensor];
        switch (job.layout) {
            case NERFX_PREC_FP32: pack_layer<NERFX_PREC_FP32>(W, L, dst, job.nan_class, a.nan_set); break;
            case NERFX_LAYOUT_M16: pack_layer<NERFX_LAYOUT_M16>(W, L, dst, job.nan_class, a.nan_set); break;
            default: pack_layer<NERFX_PREC_BF16>(W, L, dst, job.nan_class, a.nan_set); break;   // also the encoder-major stream
        }
        return;
    }
    if (blockIdx.x) return;
    const int v = j - a.n_jobs;
    const float* src = a.tensors[a.vec_tensor[v]];
    bool saw_nan = false;
    for (int i = threadIdx.x; i < a.vec_cap[v]; i += 256) {
        const float x = i < a.vec_n[v] ? src[i] : 0.0f;
        saw_nan |= x != x;
        a.consts[a.vec_off[v] + i] = sign_free_nan(x);
    }
    if (saw_nan) atomicOr(a.nan_set, (uint32_t)a.vec_class[v]);
    if (v == 0 && threadIdx.x == 0) *a.nan_clear = 0u;
}

// rgb.0 with the `features` layer folded in (see LayerPlan::folded): F = [W_r's encoder columns |
// W_r's chained columns x W_f], fb = b_r + W_r's chained columns x b_f, in the layout of rgb.0.weight.
// Workgroup (r, g): output row r, columns 64 g .. 64 g + 63 of the product (column `width` is the
// bias); thread (j, quarter) sums a quarter of the k range in double, in index order, and the four
// quarters are added in a fixed order: the result does not depend on the launch.
__global__ __launch_bounds__(256) void fold_colour_head_kernel(const float* __restrict__ Wr, const float* __restrict__ br,
                                                               const float* __restrict__ Wf, const float* __restrict__ bf,
                                                               int width, int ed_cols, nerfx_model::FoldSpots out) {
    __shared__ float s_wr[256];
    __shared__ double s_part[4][64];
    const int r = blockIdx.x, ld = ed_cols + width;
    const float* wr = Wr + (int64_t)r * ld + ed_cols;
    for (int k = threadIdx.x; k < width; k += 256) s_wr[k] = wr[k];
    if (blockIdx.y == 0) {
        for (int c = threadIdx.x; c < ed_cols; c += 256) out.w[(int64_t)r * ld + c] = Wr[(int64_t)r * ld + c];
        // the copies the gradients are unfolded with: this row of W_r's chained columns, and rows r,
        // r + gridDim.x, .. of W_f (stored transposed) and b_f
        for (int k = threadIdx.x; k < width; k += 256) out.wr[(int64_t)r * width + k] = wr[k];
        for (int row = r; row < width; row += gridDim.x) {
            for (int k = threadIdx.x; k < width; k += 256) out.wft[(int64_t)k * width + row] = Wf[(int64_t)row * width + k];
            if (threadIdx.x == 0) out.bf[row] = bf[row];
        }
    }
    __syncthreads();
    const int j = threadIdx.x & 63, quarter = threadIdx.x >> 6, c = 64 * blockIdx.y + j;
    const int k0 = quarter * (width / 4), k1 = k0 + width / 4;      // (width is a multiple of 16)
    double s = 0.0;
    if (c < width) {
#pragma unroll 8
        for (int k = k0; k < k1; ++k) s = fma((double)s_wr[k], (double)Wf[(int64_t)k * width + c], s);
    } else if (c == width) {
        for (int k = k0; k < k1; ++k) s = fma((double)s_wr[k], (double)bf[k], s);
    }
    s_part[quarter][j] = s;
    __syncthreads();
    if (quarter == 0 && c <= width) {
        const double total = ((s_part[0][j] + s_part[1][j]) + s_part[2][j]) + s_part[3][j];
        if (c < width) out.w[(int64_t)r * ld + ed_cols + c] = (float)total;
        else out.b[r] = (float)((double)br[r] + total);
    }
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
    PackLaunch a{};
    for (int i = 0; i < m->n_tensors; ++i) a.tensors[i] = tensors[i];
    const int fold_w = m->n_tensors, fold_b = m->n_tensors + 1;
    {
        const LayerPlan& R = p0.layer[2 * p0.hd + 1];
        const nerfx_model::FoldSpots f = m->fold_spots();
        hipLaunchKernelGGL(fold_colour_head_kernel, dim3(R.out_dim, (d.width + 1 + 63) / 64), dim3(256), 0, stream,
                           tensors[tensor_index(p0, 2 * p0.hd + 1, false)], tensors[tensor_index(p0, 2 * p0.hd + 1, true)],
                           tensors[tensor_index(p0, 2 * p0.hd, false)], tensors[tensor_index(p0, 2 * p0.hd, true)], d.width,
                           R.enc_cols, f);
        a.tensors[fold_w] = f.w;
        a.tensors[fold_b] = f.b;
    }
    auto nan_class = [&](int li) { return (uint8_t)(li < 2 * p0.hd ? NAN_SIGMA | NAN_RGB : NAN_RGB); };
    auto add_stream = [&](int layout, bool with_bwd) {
        const ModelPlan& p = m->plan[layout];
        a.plans[layout] = m->plans_dev[layout];
        a.fwd[layout] = m->stream_dev[layout];
        a.bwd[layout] = with_bwd ? m->stream_bwd_dev[layout] : nullptr;
        a.n_fwd[layout] = p.n_layers;
        for (int li = 0; li < p.n_layers + (with_bwd ? p.n_bwd : 0); ++li) {
            const bool fwd = li < p.n_layers;
            const LayerPlan& L = fwd ? p.layer[li] : p.bwd[li - p.n_layers];
            if (L.absent) continue;
            const int tensor = L.folded ? fold_w : tensor_index(p, fwd ? li : L.src_layer, false);
            a.job[a.n_jobs++] = PackJob{(uint8_t)layout, (uint8_t)li, (uint8_t)tensor, fwd ? nan_class(li) : (uint8_t)0};
        }
    };
    const bool bf16 = mask >> NERFX_PREC_BF16 & 1;
    if (bf16 && m->has_m16) add_stream(NERFX_LAYOUT_M16, false);            // the bf16 inference stream of the M16 kernel
    if (bf16 && m->has_encmajor) add_stream(NERFX_STREAM_ENCMAJOR, false);  // ... and of the wide-encoder kernel
    for (int prec = 0; prec < 2; ++prec)
        if (mask >> prec & 1) add_stream(prec, true);
    auto vec = [&](int tensor, int n, int cap, int off, int cls) {
        const int v = a.n_vec++;
        a.vec_tensor[v] = (uint8_t)tensor; a.vec_class[v] = (uint8_t)cls; a.vec_n[v] = n; a.vec_cap[v] = cap; a.vec_off[v] = off;
    };
    for (int li = 0; li < p0.n_layers; ++li)                           // (the rows of `features` and rgb.0 are read by no kernel)
        vec(tensor_index(p0, li, true), p0.layer[li].out_dim, BIAS_STRIDE, li * BIAS_STRIDE, nan_class(li));
    vec(2 + 4 * p0.hd, d.width, BIAS_STRIDE, p0.sigw_off, NAN_SIGMA);
    vec(2 + 4 * p0.hd + 1, 1, 4, p0.sigb_off, NAN_SIGMA);
    vec(fold_b, d.width / 2, BIAS_STRIDE / 2, p0.foldb_off, NAN_RGB);
    a.consts = m->const_dev;
    m->pack_epoch += 1;                                                // see nerfx_model::nan_flags_dev
    a.nan_set = m->nan_flags_dev + (m->pack_epoch & 1);
    a.nan_clear = m->nan_flags_dev + ((m->pack_epoch + 1) & 1);
    hipLaunchKernelGGL(pack_model_kernel, dim3(64, a.n_jobs + a.n_vec), dim3(256), 0, stream, a);
    if (bf16 && m->narrow.ok)                                          // the whole-ray kernel's block (bf16 only)
        if (int rc = narrow_pack(m->narrow, tensors, m->narrow_dev, stream)) return rc;
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
    m->narrow = narrow_plan(*desc);
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
    if (e == hipSuccess && m->narrow.ok) e = hipMalloc(&m->narrow_dev, (size_t)m->narrow.bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&m->fold_dev, (size_t)m->fold_spots().floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->nan_flags_dev, 2 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(m->nan_flags_dev, 0, 2 * sizeof(uint32_t));
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
    if (m->nan_flags_dev) (void)hipFree(m->nan_flags_dev);
    if (m->narrow_dev) (void)hipFree(m->narrow_dev);
    if (m->fold_dev) (void)hipFree(m->fold_dev);
    delete m;
}
