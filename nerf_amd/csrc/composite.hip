// K3 / K3b: alpha compositing along rays, forward and backward.
//
// One 64-lane wavefront owns one ray.  Lane l holds EPL consecutive samples
// (vector loads, fully coalesced per ray); transmittance is an in-lane running
// product followed by a 6-step cross-lane product scan, and the ray sums are
// wave reductions.  The kernels are HBM-bound: every input is read once and
// every output written once.
#include "common.hpp"

namespace nerfx {

constexpr int K3_WAVES = 4;                      // waves (= rays in flight) per workgroup
constexpr float K3_EPS = 1e-10f;                 // renderer.py:48

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Exclusive product scan across the 64 lanes.
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(inc, off, 64);
        if (lane >= off) inc *= o;
    }
    const float prev = __shfl_up(inc, 1, 64);
    return lane == 0 ? 1.0f : prev;
}
// Exclusive suffix sum across the 64 lanes (sum over lanes > this one).
__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_down(inc, off, 64);
        if (lane + off < 64) inc += o;
    }
    const float next = __shfl_down(inc, 1, 64);
    return lane == 63 ? 0.0f : next;
}

// Loads this lane's EPL samples of one ray and forms delta and alpha.
template <int EPL>
struct RaySlice {
    float sigma[EPL], t[EPL], delta[EPL], alpha[EPL], x[EPL];
    bool live[EPL];

    __device__ __forceinline__ void load(const float* __restrict__ sigma_g, const float* __restrict__ t_g,
                                         const float* __restrict__ rd_g, const float* __restrict__ delta_g,
                                         int64_t ray, int n, int lane) {
        const int64_t base = ray * n;
        const int i0 = lane * EPL;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            live[e] = i0 + e < n;
            sigma[e] = live[e] ? sigma_g[base + i0 + e] : 0.0f;
            t[e] = (live[e] && t_g) ? t_g[base + i0 + e] : 0.0f;
        }
        if (delta_g) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) delta[e] = live[e] ? delta_g[base + i0 + e] : 0.0f;
        } else {
            // ray.py:62-65: forward difference, 1e10 behind the last sample, times |rd|
            const float len = norm3(rd_g[3 * ray], rd_g[3 * ray + 1], rd_g[3 * ray + 2]);
            const float t_next_lane = __shfl_down(t[0], 1, 64);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const float tn = e + 1 < EPL ? t[e + 1 < EPL ? e + 1 : e] : t_next_lane;
                const float d = i0 + e + 1 < n ? __fsub_rn(tn, t[e]) : 1e10f;
                delta[e] = __fmul_rn(d, len);
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            alpha[e] = 1.0f - expf(-(sigma[e] * delta[e]));           // renderer.py:50
            x[e] = live[e] ? (1.0f - alpha[e]) + K3_EPS : 1.0f;       // renderer.py:51
        }
    }
    // Transmittance in front of each of this lane's samples.
    __device__ __forceinline__ void transmittance(float (&T)[EPL], int lane) const {
        float run = 1.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) run *= x[e];
        float before = wave_excl_prod(run, lane);
#pragma unroll
        for (int e = 0; e < EPL; ++e) { T[e] = before; before *= x[e]; }
    }
};

template <int EPL>
__global__ __launch_bounds__(64 * K3_WAVES) void composite_fwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ rgb, const float* __restrict__ t,
    const float* __restrict__ rd, const float* __restrict__ delta, int64_t n_rays, int n,
    float* __restrict__ weights, float* __restrict__ rgb_map, float* __restrict__ depth_map) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * K3_WAVES + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * K3_WAVES;
    for (int64_t ray = wave; ray < n_rays; ray += n_waves) {
        RaySlice<EPL> s;
        s.load(sigma, t, rd, delta, ray, n, lane);
        float T[EPL];
        s.transmittance(T, lane);
        float r = 0.f, g = 0.f, b = 0.f, d = 0.f;
        const int64_t base = ray * n + lane * EPL;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            if (!s.live[e]) continue;
            const float w = s.alpha[e] * T[e];                          // renderer.py:52
            if (weights) weights[base + e] = w;
            if (rgb) {
                r += w * rgb[(base + e) * 3 + 0];
                g += w * rgb[(base + e) * 3 + 1];
                b += w * rgb[(base + e) * 3 + 2];
            }
            d += w * s.t[e];
        }
        if (rgb_map) {
            r = wave_sum(r); g = wave_sum(g); b = wave_sum(b);
            if (lane == 0) { rgb_map[3 * ray] = r; rgb_map[3 * ray + 1] = g; rgb_map[3 * ray + 2] = b; }
        }
        if (depth_map) {
            d = wave_sum(d);
            if (lane == 0) depth_map[ray] = d;                          // renderer.py:203
        }
    }
}

// dL/dalpha_i = T_i G_i - (sum_{k>i} w_k G_k) / x_i,  G_i = dL/dw_i;
// dalpha_i/dsigma_i = delta_i (1 - alpha_i);  dL/drgb_ic = w_i g_c.
template <int EPL>
__global__ __launch_bounds__(64 * K3_WAVES) void composite_bwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ rgb, const float* __restrict__ t,
    const float* __restrict__ rd, const float* __restrict__ delta, int64_t n_rays, int n,
    const float* __restrict__ g_rgb_map, const float* __restrict__ g_depth,
    const float* __restrict__ g_weights, float* __restrict__ g_sigma, float* __restrict__ g_rgb) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * K3_WAVES + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * K3_WAVES;
    for (int64_t ray = wave; ray < n_rays; ray += n_waves) {
        RaySlice<EPL> s;
        s.load(sigma, t, rd, delta, ray, n, lane);
        float T[EPL];
        s.transmittance(T, lane);
        const float gr = g_rgb_map ? g_rgb_map[3 * ray] : 0.f, gg = g_rgb_map ? g_rgb_map[3 * ray + 1] : 0.f,
                    gb = g_rgb_map ? g_rgb_map[3 * ray + 2] : 0.f, gd = g_depth ? g_depth[ray] : 0.f;
        const int64_t base = ray * n + lane * EPL;
        float w[EPL], G[EPL], wG[EPL];
        float lane_total = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            w[e] = s.alpha[e] * T[e];
            G[e] = 0.f;
            if (s.live[e]) {
                if (rgb) G[e] = gr * rgb[(base + e) * 3] + gg * rgb[(base + e) * 3 + 1] + gb * rgb[(base + e) * 3 + 2];
                G[e] += gd * s.t[e];
                if (g_weights) G[e] += g_weights[base + e];
            }
            wG[e] = s.live[e] ? w[e] * G[e] : 0.f;
            lane_total += wG[e];
        }
        float after = wave_excl_suffix_sum(lane_total, lane);     // sum over later lanes
#pragma unroll
        for (int e = EPL - 1; e >= 0; --e) {
            if (s.live[e]) {
                const float g_alpha = T[e] * G[e] - after / s.x[e];
                // d alpha / d sigma = delta * exp(-sigma delta) = delta * (1 - alpha)
                g_sigma[base + e] = g_alpha * (s.delta[e] * (1.0f - s.alpha[e]));
                if (g_rgb) {
                    g_rgb[(base + e) * 3 + 0] = w[e] * gr;
                    g_rgb[(base + e) * 3 + 1] = w[e] * gg;
                    g_rgb[(base + e) * 3 + 2] = w[e] * gb;
                }
            }
            after += wG[e];
        }
    }
}

__global__ __launch_bounds__(256) void segment_lengths_kernel(const float* __restrict__ t, const float* __restrict__ rd,
                                                              int64_t n_rays, int n, float* __restrict__ delta) {
    const int64_t total = n_rays * n;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t ray = q / n;
        const int i = (int)(q - ray * n);
        const float len = norm3(rd[3 * ray], rd[3 * ray + 1], rd[3 * ray + 2]);
        delta[q] = __fmul_rn(i + 1 < n ? __fsub_rn(t[q + 1], t[q]) : 1e10f, len);    // ray.py:62-65
    }
}

static int composite_grid(int64_t n_rays) {
    return (int)std::min<int64_t>((n_rays + K3_WAVES - 1) / K3_WAVES, 256 * 8);
}

}  // namespace nerfx

using namespace nerfx;

#define K3_DISPATCH(EPL_EXPR, KERNEL, ...)                                                         \
    switch (EPL_EXPR) {                                                                            \
        case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                                 \
        case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                                 \
        case 3: hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); break;                                 \
        case 4: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;                                 \
        case 5: case 6: hipLaunchKernelGGL(KERNEL<6>, __VA_ARGS__); break;                         \
        default: hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__); break;                                \
    }

static int composite_check(const float* sigma, const float* t, const float* rd, const float* delta,
                           int64_t n_rays, int n) {
    NERFX_REQUIRE(n_rays >= 0, "n_rays < 0");
    NERFX_REQUIRE(n >= 1 && n <= 512, "n_samples %d out of range [1, 512]", n);
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(sigma != nullptr, "sigma is NULL");
    NERFX_REQUIRE(delta != nullptr || (t != nullptr && rd != nullptr), "need delta, or t and rd");
    return 0;
}

extern "C" int nerfx_composite_fwd(const float* sigma, const float* rgb, const float* t, const float* rd,
                                   const float* delta, int64_t n_rays, int n, float* weights,
                                   float* rgb_map, float* depth_map, void* stream) {
    if (int e = composite_check(sigma, t, rd, delta, n_rays, n)) return e;
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(!(rgb_map && !rgb), "rgb_map requested without rgb");
    NERFX_REQUIRE(!(depth_map && !t), "depth_map requested without t");
    const int epl = (n + 63) / 64;
    K3_DISPATCH(epl, composite_fwd_kernel, dim3(composite_grid(n_rays)), dim3(64 * K3_WAVES), 0,
                (hipStream_t)stream, sigma, rgb, t, rd, delta, n_rays, n, weights, rgb_map, depth_map);
    return launch_status("composite_fwd_kernel");
}

extern "C" int nerfx_composite_bwd(const float* sigma, const float* rgb, const float* t, const float* rd,
                                   const float* delta, int64_t n_rays, int n, const float* g_rgb_map,
                                   const float* g_depth, const float* g_weights, float* g_sigma,
                                   float* g_rgb, void* stream) {
    if (int e = composite_check(sigma, t, rd, delta, n_rays, n)) return e;
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(g_sigma != nullptr, "g_sigma is NULL");
    NERFX_REQUIRE(!(g_rgb_map && !rgb), "g_rgb_map given without rgb");
    NERFX_REQUIRE(!(g_depth && !t), "g_depth given without t");
    const int epl = (n + 63) / 64;
    K3_DISPATCH(epl, composite_bwd_kernel, dim3(composite_grid(n_rays)), dim3(64 * K3_WAVES), 0,
                (hipStream_t)stream, sigma, rgb, t, rd, delta, n_rays, n, g_rgb_map, g_depth, g_weights,
                g_sigma, g_rgb);
    return launch_status("composite_bwd_kernel");
}

extern "C" int nerfx_segment_lengths(const float* t, const float* rd, int64_t n_rays, int n, float* delta,
                                     void* stream) {
    NERFX_REQUIRE(n_rays >= 0 && n >= 1, "bad sizes");
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(t && rd && delta, "NULL argument");
    const int grid = (int)std::min<int64_t>((n_rays * n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(segment_lengths_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, t, rd, n_rays, n, delta);
    return launch_status("segment_lengths_kernel");
}
