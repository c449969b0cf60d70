// Microbenchmark: how fast do the training kernels' tape stores reach HBM, by tape layout?  One
// workgroup of 8 wavefronts per CU walks over 256-sample blocks and stores, for each of T tiles, 2 KiB
// per wavefront (two non-temporal 1 KiB store instructions), nothing else.
//   layout 0: [tile][sample][32]            (the library's: a block's tiles lie M x 64 bytes apart)
//   layout 1: [block][tile][sample][32]     (a block's T tiles are one contiguous T x 16 KiB region)
//   layout 2: plain fill of the same bytes  (every wavefront writes consecutive 1 KiB pieces)
//   hipcc --offload-arch=gfx950 -O3 tape_write.hip -o tape_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using f4 = __attribute__((ext_vector_type(4))) float;

template <int LAYOUT>
__global__ __launch_bounds__(512) void writer(char* __restrict__ tape, uint64_t m_pad, int tiles, int n_blocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 v = {1.0f, 2.0f, 3.0f, (float)lane};
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        for (int t = 0; t < tiles; ++t) {
            char* p;
            if (LAYOUT == 0) p = tape + ((uint64_t)t * m_pad + (uint64_t)blk * 256 + wave * 32) * 64;
            else if (LAYOUT == 1) p = tape + ((uint64_t)blk * tiles + t) * 16384 + wave * 2048;
            else p = tape + (((uint64_t)blk * tiles + t) * 8 + wave) * 2048;
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p + lane * 16));
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p + 1024 + lane * 16));
        }
    }
}

template <int LAYOUT>
void run(const char* name, char* tape, uint64_t m_pad, int tiles) {
    const int n_blocks = (int)(m_pad / 256);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(writer<LAYOUT>, dim3(256), dim3(512), 0, 0, tape, m_pad, tiles, n_blocks);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    const int iters = 10;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(writer<LAYOUT>, dim3(256), dim3(512), 0, 0, tape, m_pad, tiles, n_blocks);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
    const double bytes = (double)m_pad * 64 * tiles;
    printf("%-40s %.3f ms  %.2f GB  %.2f TB/s\n", name, ms, bytes / 1e9, bytes / ms / 1e9);
}

int main() {
    const uint64_t m_pad = 524288; const int tiles = 41;
    char* tape; CHECK(hipMalloc(&tape, m_pad * 64 * tiles));
    run<0>("[tile][sample][32] (library)", tape, m_pad, tiles);
    run<1>("[block][tile][sample][32]", tape, m_pad, tiles);
    run<2>("plain fill", tape, m_pad, tiles);
    run<0>("[tile][sample][32] (library)", tape, m_pad, tiles);
    return 0;
}
