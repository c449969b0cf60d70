// The compositor's per-ray arithmetic (renderer.py:50-52 with ray.py:60-65 in front), shared by the
// compositing kernels (composite.hip) and by the kernel that composites the coarse pass and resamples
// in one go (pdf_sample.hip): one definition, compiled without contraction into fused multiply-adds
// (which the compiler would otherwise choose per call site), so both give the same bits.
#pragma once
#include "common.hpp"

namespace nerfx {

constexpr float K3_EPS = 1e-10f;                 // renderer.py:48

// Exclusive product scan across a group of LPR consecutive lanes (gl = lane index inside the group).
template <int LPR>
__device__ __forceinline__ float group_excl_prod(float v, int gl) {
#pragma clang fp contract(off)
    float inc = v;
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) {
        const float o = __shfl_up(inc, off, LPR);
        if (gl >= off) inc *= o;
    }
    const float prev = __shfl_up(inc, 1, LPR);
    return gl == 0 ? 1.0f : prev;
}

// A lane's EPL consecutive samples i0 .. of an n-sample ray: segment lengths (forward difference, 1e10
// behind the last sample, times |rd|: ray.py:62-65; `t_next_lane` = the next lane's first z value),
// alpha (renderer.py:50) and the factor each sample leaves of the transmittance (renderer.py:51; 1
// beyond the `valid` live samples of the slice).
template <int EPL>
__device__ __forceinline__ void ray_alpha(const float (&sigma)[EPL], const float (&t)[EPL], float t_next_lane, float len,
                                          int i0, int n, int valid, float (&delta)[EPL], float (&alpha)[EPL], float (&x)[EPL]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const float tn = e + 1 < EPL ? t[e + 1 < EPL ? e + 1 : e] : t_next_lane;
        const float d = i0 + e + 1 < n ? __fsub_rn(tn, t[e]) : 1e10f;
        delta[e] = __fmul_rn(d, len);
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        alpha[e] = 1.0f - expf(-(sigma[e] * delta[e]));
        x[e] = e < valid ? (1.0f - alpha[e]) + K3_EPS : 1.0f;
    }
}
// The same from given segment lengths.
template <int EPL>
__device__ __forceinline__ void ray_alpha_from_delta(const float (&sigma)[EPL], const float (&delta)[EPL], int valid,
                                                     float (&alpha)[EPL], float (&x)[EPL]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        alpha[e] = 1.0f - expf(-(sigma[e] * delta[e]));
        x[e] = e < valid ? (1.0f - alpha[e]) + K3_EPS : 1.0f;
    }
}
// Transmittance in front of each of the lane's samples: the product of the factors of all samples
// before it, in the lane and in the lanes before it in the ray's group.
template <int EPL, int LPR>
__device__ __forceinline__ void ray_transmittance(const float (&x)[EPL], int gl, float (&T)[EPL]) {
#pragma clang fp contract(off)
    float run = 1.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) run *= x[e];
    float before = group_excl_prod<LPR>(run, gl);
#pragma unroll
    for (int e = 0; e < EPL; ++e) { T[e] = before; before *= x[e]; }
}

// Weights (renderer.py:52).
template <int EPL>
__device__ __forceinline__ void ray_weights(const float (&alpha)[EPL], const float (&T)[EPL], float (&w)[EPL]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < EPL; ++e) w[e] = alpha[e] * T[e];
}

}  // namespace nerfx
