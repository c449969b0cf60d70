// Counter-based uniforms for perturb=True (the reference draws rand_like(t) / rand(B, Nf) from
// torch's global generator: ray.py:99, :167).  Philox4x32-10 (Salmon et al., "Parallel random
// numbers: as easy as 1, 2, 3", SC'11): four 32-bit words per 128-bit counter, no state in memory,
// so the sampling kernels draw their own jitter instead of reading (B, N) uniforms from HBM.
//
// Element e of draw `which` (0 = coarse stratification, 1 = inverse-CDF positions) is word e % 4
// of the block with counter (e / 4, offset ^ which-tag) under key = seed; a float in [0, 1) is
// the top 24 bits of the word times 2^-24 (what torch.rand produces for float32).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace nerfx {

struct Rng {
    uint64_t seed, offset;
    uint32_t which;
    int on;                      // 0: no generator (the caller's uniforms, or none)
};

struct Philox4 { uint32_t w[4]; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// The four uniforms of elements 4 b .. 4 b + 3.
__host__ __device__ inline Philox4 rng_block(const Rng& g, uint64_t b) {
    const uint64_t hi = g.offset ^ ((uint64_t)g.which * 0x9E3779B97F4A7C15ull);
    return philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), (uint32_t)g.seed,
                         (uint32_t)(g.seed >> 32));
}
__host__ __device__ inline float rng_to_float(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }
__host__ __device__ inline float rng_uniform(const Rng& g, uint64_t element) {
    return rng_to_float(rng_block(g, element >> 2).w[element & 3]);
}

}  // namespace nerfx
