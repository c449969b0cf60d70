// Layout shared by the weight packer (host + pack kernels) and the fused MLP
// kernels: how a radiance field's Linear layers become one linear "unit stream".
//
// A UNIT is 1 KiB = 64 lanes x 16 B: the A operands (weights) one wavefront
// reads with a single ds_read_b128 for one 32-row output tile and one k-unit
// (16 input features in bf16, 8 in fp32).  Units are stored in exactly the
// order the kernel consumes them -- layer by layer, k-unit by k-unit, output
// tile by output tile -- so the kernel's LDS ring is a sliding window over a
// flat array filled by global_load_lds, and lane l always reads unit + 16 l.
#pragma once
#include "common.hpp"

namespace nerfx {

constexpr int UNIT_BYTES = 1024;
constexpr int CHUNK_UNITS = 16;                 // ring granularity: 16 KiB
constexpr int RING_SLOTS = 8;                   // 128 KiB of LDS
constexpr int RING_AHEAD = 6;                   // chunks in flight ahead of the consumer
constexpr int MAX_LAYERS = 20;                  // 2*hd trunk + features + rgb.0 + rgb.2, depth <= 16
constexpr int BIAS_STRIDE = 256;                // floats of const block per layer
constexpr int MAX_TABLE = 1024;                 // floats per encoder table

// slots (input features) per k-unit and units per 32-feature activation tile
__host__ __device__ constexpr int slots_per_unit(int prec) { return prec == NERFX_PREC_BF16 ? 16 : 8; }
__host__ __device__ constexpr int units_per_tile(int prec) { return prec == NERFX_PREC_BF16 ? 2 : 4; }

// Encoder slot order used inside the kernels: slot 2p = sin(angle p), 2p+1 =
// cos(angle p); a positional encoder then appends the 3 raw coordinates.
// Returns the column of the REFERENCE layout (features.py:32-36, :52) held by
// `slot`, or -1 for padding.
__host__ __device__ inline int enc_slot_col(int kind, int n, int slot) {
    const int P = kind == NERFX_ENC_POSITIONAL ? 3 * n : n;
    const int q = slot >> 1, which = slot & 1;
    if (q < P) {
        if (kind == NERFX_ENC_POSITIONAL) { const int f = q / 3, c = q - 3 * f; return 3 + 6 * f + 3 * which + c; }
        return which * n + q;
    }
    const int c = slot - 2 * P;
    return (kind == NERFX_ENC_POSITIONAL && c < 3) ? c : -1;
}
__host__ __device__ inline int enc_units(int kind, int n, int prec) {
    const int slots = (kind == NERFX_ENC_POSITIONAL ? 6 * n + 3 : 2 * n);
    return (slots + slots_per_unit(prec) - 1) / slots_per_unit(prec);
}
// Encoder slot supplied by lane half h as element e of k-unit u.
__host__ __device__ inline int enc_unit_slot(int prec, int u, int h, int e) {
    const int spu = slots_per_unit(prec);
    return spu * u + (spu / 2) * h + e;
}
// Feature (row of the previous layer's 32x32 accumulator tiles) supplied by lane
// half h as element e of chained k-unit u: the accumulator registers of a
// v_mfma_f32_32x32x* result, reused in place as the next B operand.
__host__ __device__ inline int chain_unit_feature(int prec, int u, int h, int e) {
    if (prec == NERFX_PREC_BF16) return 16 * u + 8 * (e >> 2) + 4 * h + (e & 3);
    return 8 * u + 4 * h + e;
}

// Output tiles a layer computes at a time (its k-block is this many units): bounded by
// the accumulator registers that fit beside the input and output fragments.
__host__ __device__ constexpr int group_tiles(int prec, int n_tiles) {
    const int cap = prec == NERFX_PREC_BF16 ? 2 : 4;
    return n_tiles < cap ? n_tiles : cap;
}

// One Linear layer as the kernels see it.  Units are ordered
// [tile group][k-unit][tile in group].
struct LayerPlan {
    int n_tiles;        // output tiles of 32 rows
    int out_dim;        // true rows
    int enc_units;      // leading k-units fed by an encoder (0 = none)
    int enc_kind, enc_n, enc_cols;   // that encoder and its width in the reference layout
    int chain_tiles;    // 32-feature input tiles fed by the previous layer (0 for the first)
    int chain_dim;      // true chained inputs
    int in_dim;         // leading dimension of the reference weight (enc_cols + chain_dim)
    int64_t unit0;      // first unit of this layer in the stream
};

struct ModelPlan {
    nerfx_model_desc_t desc;
    int prec;
    int wt;             // hidden tiles: ceil(width / 32)
    int rt;             // rgb.0 output tiles: ceil(width / 2 / 32)
    int hd;             // layers per trunk half
    int n_layers;       // 2*hd + 3
    LayerPlan layer[MAX_LAYERS];
    int64_t trunk_units;   // stream units up to and including the last trunk layer (chunk padded)
    int64_t total_units;   // whole stream (chunk padded)
    // const block (floats): [layer biases | sigma w | sigma b | table x | table d]
    int sigw_off, sigb_off, tabx_off, tabd_off, const_floats;
};

inline int units_of(const LayerPlan& L, int prec) {
    return (L.enc_units + L.chain_tiles * units_per_tile(prec)) * L.n_tiles;
}

// Fills `p` from a model description; returns 0 or an error code.
int make_plan(const nerfx_model_desc_t& d, int prec, ModelPlan& p);

}  // namespace nerfx

// The opaque handle of include/nerfx.h.
struct nerfx_model {
    nerfx::ModelPlan plan[2];          // indexed by NERFX_PREC_*
    void* stream_dev[2];               // packed unit streams
    float* const_dev;                  // const block (same for both precisions)
    int n_tensors;
    int device;
};
