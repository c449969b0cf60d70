// Microbenchmark: how fast can the waves of one workgroup per CU stream an L2-resident buffer into
// LDS with LDS-DMA (global_load_lds, 16 or 4 bytes per lane), against plain 16-byte loads into
// registers?  Prints bytes per shader cycle and CU (measured: 59 for all three with 8 waves).   hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int UNIT = 1024;

template <int MODE, int WAVES>   // 0: DMA 16 B/lane, 1: DMA 4 B/lane, 2: registers 16 B/lane
__global__ __launch_bounds__(64 * WAVES) void stream_kernel(const char* __restrict__ src, int n_units, int rounds, long long* cycles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    using f4 = __attribute__((ext_vector_type(4))) float;
    f4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        for (int u = wave * 4; u + 3 < n_units; u += WAVES * 4) {
            const char* g = src + (size_t)u * UNIT + lane * 16;
            char* l = smem + (size_t)(wave * 4) * UNIT;
            if (MODE == 0) {
                const auto gp = (const void __attribute__((address_space(1)))*)g;
                const auto lp = (void __attribute__((address_space(3)))*)l;
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 0 * UNIT, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 1 * UNIT, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 2 * UNIT, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 3 * UNIT, 0);
            } else if (MODE == 1) {
                const char* g4 = src + (size_t)u * UNIT + lane * 4;
                const auto gp = (const void __attribute__((address_space(1)))*)g4;
                const auto lp = (void __attribute__((address_space(3)))*)l;
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 0, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 256, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 512, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 768, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 1024, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 1280, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 1536, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 1792, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 2048, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 2304, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 2560, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 2816, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 3072, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 3328, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 3584, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 4, 3840, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f4 v = *reinterpret_cast<const f4*>(g + k * UNIT);
                    acc += v;
                }
            }
            if (MODE != 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (MODE == 2) sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    else if (threadIdx.x == 0) sink[blockIdx.x] = ((float*)smem)[5];
}

template <int MODE, int WAVES>
void run(const char* name, const char* src, int n_units, long long* cyc, float* sink) {
    const int rounds = 64, grid = 256;
    const size_t lds = (size_t)WAVES * 4 * UNIT;
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL((stream_kernel<MODE, WAVES>), dim3(grid), dim3(64 * WAVES), lds, 0, src, n_units, rounds, cyc, sink);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<long long> h(grid);
    CHECK(hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto c : h) sum += (double)c;
    // (s_memtime counts shader cycles)
    const double bytes = (double)(n_units / (WAVES * 4)) * (WAVES * 4) * UNIT * rounds;
    printf("%-34s waves %d: %.1f bytes per cycle and CU\n", name, WAVES, bytes / (sum / grid));
}

int main() {
    const int n_units = 672;
    char* src; long long* cyc; float* sink;
    CHECK(hipMalloc(&src, (size_t)n_units * UNIT));
    CHECK(hipMemset(src, 1, (size_t)n_units * UNIT));
    CHECK(hipMalloc(&cyc, 256 * sizeof(long long)));
    CHECK(hipMalloc(&sink, 256 * 512 * sizeof(float)));
    run<0, 8>("LDS-DMA 16 B/lane", src, n_units, cyc, sink);
    run<0, 4>("LDS-DMA 16 B/lane", src, n_units, cyc, sink);
    run<1, 8>("LDS-DMA 4 B/lane", src, n_units, cyc, sink);
    run<2, 8>("global_load_dwordx4 -> registers", src, n_units, cyc, sink);
    run<2, 4>("global_load_dwordx4 -> registers", src, n_units, cyc, sink);
    return 0;
}
