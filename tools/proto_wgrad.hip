// Timing prototype (GPU box; not part of the library): how fast can a weight-gradient block
// dW[256 x 256] += G[256 x samples] * X[samples x 256] stream its two tapes when every tile is
// fetched ONCE per layer and staged by LDS-DMA into a ring of chunks?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto_wgrad.hip -o nerf_amd/csrc/build/proto_wgrad
//   nerf_amd/csrc/build/proto_wgrad [layers=6] [samples=524288]      (-DPROTO_CHUNK=64 -DPROTO_DEPTH=2 -DPROTO_INTERLEAVED: variants)
//
// Layout as the training tape's: tiles [tile][sample][32 positions] of bf16.  One workgroup of
// 16 wavefronts per CU owns a range of samples of one layer (8 gradient tiles x 8 input tiles, 2 x 2
// tile products per wavefront); a chunk is 32 samples (16 tiles x 2 KiB), the ring holds DEPTH chunks;
// wavefront w brings tile w of every chunk (one LDS base, two 1 KiB loads).  Results are not checked
// (the partial sums are written so that nothing is optimised away); the number that matters is the
// time per layer against the library's kernel (0.68 ms for the 256 x 4 model's step, ~26 layer-blocks).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using lds_bytes_t = __attribute__((address_space(3))) char*;

#ifndef PROTO_CHUNK
#define PROTO_CHUNK 32
#endif
#ifndef PROTO_DEPTH
#define PROTO_DEPTH 4
#endif
constexpr int CHUNK = PROTO_CHUNK, DEPTH = PROTO_DEPTH, TILES = 16, TILE_BYTES = CHUNK * 64, CHUNK_BYTES = TILES * TILE_BYTES;
constexpr int WAVES = 16;

__device__ __forceinline__ bf16x8 fetch(lds_bytes_t tile, uint32_t lane_off, int kk) {
    using P4 = __attribute__((address_space(3))) s16x4*;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P4)(tile + lane_off + kk * 1024));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P4)(tile + lane_off + kk * 1024 + 256));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(64 * WAVES, 1) void proto_kernel(const __bf16* __restrict__ gtape, const __bf16* __restrict__ ztape,
                                                             uint64_t m_pad, int chunks_per_wg, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int layer = blockIdx.y;
    const int c0 = blockIdx.x * chunks_per_wg;
    // tile `wave` of this layer: gradient tiles 0..7 from gtape, input tiles 8..15 from ztape
    const __bf16* tape = wave < 8 ? gtape : ztape;
#ifdef PROTO_INTERLEAVED
    // alternative tape layout [layer][chunk][tile of the layer][sample][32 positions]: a workgroup's
    // chunk is two contiguous 8-tile regions instead of sixteen pieces 32 MiB apart
    const char* src = (const char*)(tape + ((uint64_t)layer * 8 * m_pad + ((uint64_t)c0 * 8 + (wave & 7)) * CHUNK) * 32) + lane * 16;
    constexpr size_t STEP = 8 * (size_t)TILE_BYTES;
#else
    const char* src = (const char*)(tape + ((uint64_t)(layer * 8 + (wave & 7)) * m_pad + (uint64_t)c0 * CHUNK) * 32) + lane * 16;
    constexpr size_t STEP = TILE_BYTES;
#endif
    auto request = [&](int c) {                                     // chunk c of this workgroup's range, into ring slot c % DEPTH
        const auto gp = (const void __attribute__((address_space(1)))*)(src + (size_t)c * STEP);
        const auto lp = (void __attribute__((address_space(3)))*)((lds_bytes_t)smem + (c % DEPTH) * CHUNK_BYTES + wave * TILE_BYTES);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
        if constexpr (CHUNK == 64) {
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
        }
    };
    constexpr int PER_CHUNK = CHUNK / 16;                           // loads per wavefront and chunk
    const int oa = 2 * (wave >> 2), ib = 2 * (wave & 3);            // this wavefront's 2 x 2 products
    const int L = lane & 15, g = lane >> 4;
    const uint32_t lane_off = (uint32_t)((8 * (g >> 1) + (L >> 2)) * 64 + (16 * (g & 1) + 4 * (L & 3)) * 2);
    f32x16 acc[2][2] = {};
#pragma unroll
    for (int c = 0; c < DEPTH - 1; ++c)
        if (c < chunks_per_wg) request(c);
    for (int c = 0; c < chunks_per_wg; ++c) {
        // this wavefront's loads of chunk c have landed (those of the DEPTH - 2 younger chunks may fly)
        if (c + DEPTH - 2 < chunks_per_wg) wait_vm<PER_CHUNK * (DEPTH - 2)>();
        else wait_vm<0>();
        __syncthreads();                                            // everybody's have; chunk c - 1's slot is free
        if (c + DEPTH - 1 < chunks_per_wg) request(c + DEPTH - 1);
        lds_bytes_t base = (lds_bytes_t)smem + (c % DEPTH) * CHUNK_BYTES;
        lds_bytes_t ta = base + oa * TILE_BYTES, tb = base + (8 + ib) * TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < CHUNK / 16; ++kk) {
            const bf16x8 a0 = fetch(ta, lane_off, kk), a1 = fetch(ta + TILE_BYTES, lane_off, kk);
            const bf16x8 b0 = fetch(tb, lane_off, kk), b1 = fetch(tb + TILE_BYTES, lane_off, kk);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float* dst = partial + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + (oa * 8 + ib)) * 1024 + lane * 16;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int n = 0; n < 2; ++n) *reinterpret_cast<f32x16*>(dst + (o * 8 + n) * 1024) = acc[o][n];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 6;
    const uint64_t m_pad = argc > 2 ? strtoull(argv[2], nullptr, 10) : 524288;
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int n_chunks = (int)(m_pad / CHUNK);
    const int wgs_per_layer = cus / layers > 0 ? (cus / layers) : 1; // one round of resident workgroups over all layers
    const int chunks_per_wg = n_chunks / wgs_per_layer;             // (the tail of the sample range is left out: timing only)
    const size_t tape_bytes = (size_t)layers * 8 * m_pad * 64;
    __bf16 *gt, *zt;
    float* partial;
    CK(hipMalloc(&gt, tape_bytes));
    CK(hipMalloc(&zt, tape_bytes));
    CK(hipMalloc(&partial, (size_t)layers * wgs_per_layer * 64 * 4096));
    CK(hipMemset(gt, 0, tape_bytes));
    CK(hipMemset(zt, 0, tape_bytes));
    const size_t lds = (size_t)DEPTH * CHUNK_BYTES;
    CK(hipFuncSetAttribute((const void*)proto_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(proto_kernel, dim3(wgs_per_layer, layers), dim3(64 * WAVES), lds, 0, gt, zt, m_pad, chunks_per_wg, partial);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int iters = 10;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(proto_kernel, dim3(wgs_per_layer, layers), dim3(64 * WAVES), lds, 0, gt, zt, m_pad, chunks_per_wg, partial);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= iters;
    const double bytes = 2.0 * (double)layers * 8 * (double)chunks_per_wg * wgs_per_layer * CHUNK * 64;
    const double flop = 2.0 * layers * 256.0 * 256.0 * (double)chunks_per_wg * wgs_per_layer * CHUNK;
    printf("{\"layers\": %d, \"samples\": %llu, \"workgroups\": %d, \"chunks_per_wg\": %d, \"ms\": %.4f, \"tape_GB\": %.3f, \"TBps\": %.3f, \"TFLOPs\": %.1f, \"ms_per_layer\": %.4f}\n",
           layers, (unsigned long long)m_pad, wgs_per_layer * layers, chunks_per_wg, ms, bytes / 1e9, bytes / ms / 1e9, flop / ms / 1e9, ms / layers);
    return 0;
}
