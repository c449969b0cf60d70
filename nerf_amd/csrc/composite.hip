// K3 / K3b: alpha compositing along rays, forward and backward.
//
// A group of 32 lanes (64 beyond 256 samples) owns one ray.  A lane holds EPL consecutive
// samples (128-bit loads and stores when rows are whole 16-byte slices, as for 128 samples);
// transmittance is an in-lane running product followed by a cross-lane product scan inside the
// group, and the ray sums are group reductions.  The kernels are HBM-bound: every input is
// read once and every output written once.
#include "composite.hpp"

namespace nerfx {

constexpr int K3_WAVES = 4;                      // waves per workgroup

// A ray is owned by a GROUP of LPR consecutive lanes (LPR = 16: four rays per wavefront, 32: two, 64: one);
// all cross-lane steps stay inside the group.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, LPR);
    return v;
}
// Exclusive suffix sum across the group (sum over lanes after this one).
template <int LPR>
__device__ __forceinline__ float group_excl_suffix_sum(float v, int gl) {
    float inc = v;
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) {
        const float o = __shfl_down(inc, off, LPR);
        if (gl + off < LPR) inc += o;
    }
    const float next = __shfl_down(inc, 1, LPR);
    return gl == LPR - 1 ? 0.0f : next;
}

// N consecutive floats starting at p[0] (this lane's slice of a row).  `vec` (wave-uniform):
// every lane's slice is whole and aligned to its own size, so 64/128-bit accesses are legal.
// Every row is read once and written once per launch.  The per-sample rows (densities, z values,
// weights, density gradients), where a lane's slice is one whole 16-byte access, carry the
// non-temporal hint (NT): measured 0.58 -> 0.72 of the HBM peak for the weights-only pass at 128
// samples.  Colour rows (48 bytes per lane, cache lines shared between neighbouring lanes' three
// accesses) are better off without it (0.70 -> 0.67 with).
using k3_f4 = __attribute__((ext_vector_type(4))) float;
using k3_f2 = __attribute__((ext_vector_type(2))) float;
template <class V, bool NT> __device__ __forceinline__ V k3_load(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <class V, bool NT> __device__ __forceinline__ void k3_store(V v, V* p) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int N, bool NT = (N <= 8)>
__device__ __forceinline__ void load_slice(const float* __restrict__ p, int valid, bool vec, float (&v)[N]) {
    if (vec && N % 4 == 0) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k) {
            const k3_f4 q = k3_load<k3_f4, NT>(reinterpret_cast<const k3_f4*>(p) + k);
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        }
    } else if (vec && N % 2 == 0) {
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            const k3_f2 q = k3_load<k3_f2, NT>(reinterpret_cast<const k3_f2*>(p) + k);
            v[2 * k] = q.x; v[2 * k + 1] = q.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = k < valid ? p[k] : 0.0f;
    }
}
template <int N, bool NT = (N <= 8)>
__device__ __forceinline__ void store_slice(float* __restrict__ p, int valid, bool vec, const float (&v)[N]) {
    if (vec && N % 4 == 0) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k)
            k3_store<k3_f4, NT>(k3_f4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]}, reinterpret_cast<k3_f4*>(p) + k);
    } else if (vec && N % 2 == 0) {
#pragma unroll
        for (int k = 0; k < N / 2; ++k) k3_store<k3_f2, NT>(k3_f2{v[2 * k], v[2 * k + 1]}, reinterpret_cast<k3_f2*>(p) + k);
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (k < valid) p[k] = v[k];
    }
}

// Colour-gradient rows through LDS.  A lane's slice of a colour row is 3 EPL floats at a stride of
// 3 EPL floats: stored straight from the lanes, every 16-byte store of a wavefront touches three
// times the cache lines it fills.  For EPL = 4 the rows of a wavefront's rays are one contiguous
// 3 KiB region, so the lanes park their slices in a wave-private 3 KiB region of LDS (float4
// 3 l + k of lane l) and the wavefront stores the region as three fully coalesced 1 KiB accesses
// (float4 q from lane q % 64 of access q / 64).  LDS is in order per wavefront and the region is
// private, so no barrier is involved.  Backward at 262,144 rays x 128 samples: 0.65 -> 0.72 of the
// HBM peak.  Loading the colour rows the same way (LDS-DMA in, slices out) measured no gain in
// either direction (backward 0.70 with both, forward 0.74 against 0.75), so loads stay direct.
constexpr int K3_REGION_F4 = 192;                 // float4 per wavefront region (64 lanes x 12 floats)
using k3_lds_f4 = __attribute__((address_space(3))) k3_f4;
// The lanes' twelve floats each, written to `rows` from float4 `first` on as three coalesced accesses.
__device__ __forceinline__ void region_store(float* __restrict__ rows, int64_t first, int64_t total, int lane,
                                             k3_lds_f4* region, const float (&c)[12]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) region[3 * lane + k] = k3_f4{c[4 * k], c[4 * k + 1], c[4 * k + 2], c[4 * k + 3]};
    k3_f4* dst = reinterpret_cast<k3_f4*>(rows);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const k3_f4 v = region[lane + 64 * j];
        const int64_t q = first + lane + 64 * j;
        if (q < total) dst[q] = v;
    }
}

// This lane's EPL samples of one ray, with delta and alpha.
template <int EPL, int LPR>
struct RaySlice {
    float sigma[EPL], t[EPL], delta[EPL], alpha[EPL], x[EPL];
    int valid;                                    // live samples in this lane's slice
    bool vec;                                     // whole-row vector access is legal (wave-uniform)

    __device__ __forceinline__ void load(const float* __restrict__ sigma_g, const float* __restrict__ t_g,
                                         const float* __restrict__ rd_g, const float* __restrict__ delta_g,
                                         int64_t ray, int n, int gl) {
        const int64_t base = ray * n;
        const int i0 = gl * EPL;
        valid = min(max(n - i0, 0), EPL);
        vec = n == EPL * LPR;                     // full rows: slices are whole and naturally aligned
        load_slice<EPL>(sigma_g + base + i0, valid, vec, sigma);
        if (t_g) load_slice<EPL>(t_g + base + i0, valid, vec, t);
        else
#pragma unroll
            for (int e = 0; e < EPL; ++e) t[e] = 0.0f;
        if (delta_g) {
            load_slice<EPL>(delta_g + base + i0, valid, vec, delta);
            ray_alpha_from_delta<EPL>(sigma, delta, valid, alpha, x);
        } else {
            const float len = norm3(rd_g[3 * ray], rd_g[3 * ray + 1], rd_g[3 * ray + 2]);
            ray_alpha<EPL>(sigma, t, __shfl_down(t[0], 1, LPR), len, i0, n, valid, delta, alpha, x);
        }
    }
    // Transmittance in front of each of this lane's samples.
    __device__ __forceinline__ void transmittance(float (&T)[EPL], int gl) const { ray_transmittance<EPL, LPR>(x, gl, T); }
};

template <int EPL, int LPR>
__global__ __launch_bounds__(64 * K3_WAVES) void composite_fwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ rgb, const float* __restrict__ t,
    const float* __restrict__ rd, const float* __restrict__ delta, int64_t n_rays, int n,
    float* __restrict__ weights, float* __restrict__ rgb_map, float* __restrict__ depth_map) {
    constexpr int RPW = 64 / LPR;                                   // rays per wavefront
    const int lane = threadIdx.x & 63, gl = lane % LPR;
    const int64_t group = ((int64_t)blockIdx.x * K3_WAVES + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int64_t n_groups = (int64_t)gridDim.x * K3_WAVES * RPW;
    // whole wavefronts stay in the loop together (the shuffles need every lane of a group)
    for (int64_t ray0 = group - lane / LPR; ray0 < n_rays; ray0 += n_groups) {
        const int64_t ray = min(ray0 + lane / LPR, n_rays - 1);
        const bool mine = ray0 + lane / LPR < n_rays;
        RaySlice<EPL, LPR> s;
        const int64_t base = ray * n + gl * EPL;
        float w[EPL], c[3 * EPL];
        // colours are requested together with the densities: one round trip to HBM per ray, not two
        // (taking them through LDS like the backward does gains nothing here -- 0.74 against 0.75 of
        // the HBM peak -- and the LDS it reserves costs the weights-only passes a fifth of their rate)
        if (rgb) load_slice<3 * EPL>(rgb + base * 3, 3 * min(max(n - gl * EPL, 0), EPL), n == EPL * LPR, c);
        s.load(sigma, t, rd, delta, ray, n, gl);
        float T[EPL];
        s.transmittance(T, gl);
        ray_weights<EPL>(s.alpha, T, w);                                // renderer.py:52
        if (weights && mine) store_slice<EPL>(weights + base, s.valid, s.vec, w);
        float r = 0.f, g = 0.f, b = 0.f, d = 0.f;
        if (rgb) {
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                if (e < s.valid) { r += w[e] * c[3 * e]; g += w[e] * c[3 * e + 1]; b += w[e] * c[3 * e + 2]; }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (e < s.valid) d += w[e] * s.t[e];
        if (rgb_map) {
            r = group_sum<LPR>(r); g = group_sum<LPR>(g); b = group_sum<LPR>(b);
            if (gl == 0 && mine) { rgb_map[3 * ray] = r; rgb_map[3 * ray + 1] = g; rgb_map[3 * ray + 2] = b; }
        }
        if (depth_map) {
            d = group_sum<LPR>(d);
            if (gl == 0 && mine) depth_map[ray] = d;                    // renderer.py:203
        }
    }
}

// dL/dalpha_i = T_i G_i - (sum_{k>i} w_k G_k) / x_i,  G_i = dL/dw_i;
// dalpha_i/dsigma_i = delta_i (1 - alpha_i);  dL/drgb_ic = w_i g_c.
template <int EPL, int LPR>
__global__ __launch_bounds__(64 * K3_WAVES) void composite_bwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ rgb, const float* __restrict__ t,
    const float* __restrict__ rd, const float* __restrict__ delta, int64_t n_rays, int n,
    const float* __restrict__ g_rgb_map, const float* __restrict__ g_depth,
    const float* __restrict__ g_weights, float* __restrict__ g_sigma, float* __restrict__ g_rgb) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, gl = lane % LPR;
    const int64_t group = ((int64_t)blockIdx.x * K3_WAVES + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int64_t n_groups = (int64_t)gridDim.x * K3_WAVES * RPW;
    __shared__ k3_f4 regions[EPL == 4 ? K3_WAVES * K3_REGION_F4 : 1];
    k3_lds_f4* region = (k3_lds_f4*)regions + (EPL == 4 ? (threadIdx.x >> 6) * K3_REGION_F4 : 0);
    for (int64_t ray0 = group - lane / LPR; ray0 < n_rays; ray0 += n_groups) {
        const int64_t ray = min(ray0 + lane / LPR, n_rays - 1);
        const bool mine = ray0 + lane / LPR < n_rays;
        RaySlice<EPL, LPR> s;
        const int64_t base = ray * n + gl * EPL;
        float w[EPL], G[EPL], wG[EPL], c[3 * EPL], gw[EPL];
        // every input of the ray is requested up front: one round trip to HBM per ray
        const int live = min(max(n - gl * EPL, 0), EPL);
        const bool whole = n == EPL * LPR;
        const bool via_lds = EPL == 4 && whole && g_rgb;              // (wave-uniform)
        const int64_t first = ray0 * (3 * LPR), total = n_rays * (3 * LPR);   // in float4 (EPL == 4)
        if (rgb) load_slice<3 * EPL>(rgb + base * 3, 3 * live, whole, c);
        if (g_weights) load_slice<EPL>(g_weights + base, live, whole, gw);
        const float gr = g_rgb_map ? g_rgb_map[3 * ray] : 0.f, gg = g_rgb_map ? g_rgb_map[3 * ray + 1] : 0.f,
                    gb = g_rgb_map ? g_rgb_map[3 * ray + 2] : 0.f, gd = g_depth ? g_depth[ray] : 0.f;
        s.load(sigma, t, rd, delta, ray, n, gl);
        float T[EPL];
        s.transmittance(T, gl);
        float lane_total = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            w[e] = s.alpha[e] * T[e];
            G[e] = 0.f;
            if (e < s.valid) {
                if (rgb) G[e] = gr * c[3 * e] + gg * c[3 * e + 1] + gb * c[3 * e + 2];
                G[e] += gd * s.t[e];
                if (g_weights) G[e] += gw[e];
            }
            wG[e] = e < s.valid ? w[e] * G[e] : 0.f;
            lane_total += wG[e];
        }
        float after = group_excl_suffix_sum<LPR>(lane_total, gl);   // sum over later lanes
        float gs[EPL], gc[3 * EPL];
#pragma unroll
        for (int e = EPL - 1; e >= 0; --e) {
            const float g_alpha = T[e] * G[e] - after / s.x[e];
            // d alpha / d sigma = delta * exp(-sigma delta) = delta * (1 - alpha)
            gs[e] = g_alpha * (s.delta[e] * (1.0f - s.alpha[e]));
            gc[3 * e] = w[e] * gr; gc[3 * e + 1] = w[e] * gg; gc[3 * e + 2] = w[e] * gb;
            after += wG[e];
        }
        if (mine) store_slice<EPL>(g_sigma + base, s.valid, s.vec, gs);
        if constexpr (EPL == 4) {
            if (via_lds) {                                            // whole wavefront: lanes of absent rays store nothing
                region_store(g_rgb, first, total, lane, region, gc);
                continue;
            }
        }
        if (mine && g_rgb) store_slice<3 * EPL>(g_rgb + base * 3, 3 * s.valid, s.vec, gc);
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

// Workgroups to launch.  Kernels that also stream the colour rows (forward with rgb, backward) keep
// more bytes in flight per wavefront and run fastest with (nearly) one pass per wavefront; the light
// weights-only forward is fastest with the resident set (8 workgroups per CU) looping over rays.
static int composite_grid(int64_t n_rays, int n, bool colour_rows) {
    const int per_block = K3_WAVES * (n > 256 ? 1 : (n > 32 && n <= 64 ? 4 : 2));   // rays in flight per workgroup
    return (int)std::min<int64_t>((n_rays + per_block - 1) / per_block, colour_rows ? 256 * 128 : 256 * 8);
}

}  // namespace nerfx

using namespace nerfx;

// Lanes per ray and samples per lane for n samples: four rays per wavefront for 33..64 samples
// (the coarse pass), two up to 256 samples (16-byte slices when n is 64, 128 or 256), one beyond.
#define K3_CASE(EPL, LPR, KERNEL, ...) hipLaunchKernelGGL((KERNEL<EPL, LPR>), __VA_ARGS__)
#define K3_DISPATCH(N, KERNEL, ...)                                                                \
    do {                                                                                           \
        if ((N) <= 32) K3_CASE(1, 32, KERNEL, __VA_ARGS__);                                        \
        else if ((N) <= 64) K3_CASE(4, 16, KERNEL, __VA_ARGS__);                                   \
        else if ((N) <= 96) K3_CASE(3, 32, KERNEL, __VA_ARGS__);                                   \
        else if ((N) <= 128) K3_CASE(4, 32, KERNEL, __VA_ARGS__);                                  \
        else if ((N) <= 192) K3_CASE(6, 32, KERNEL, __VA_ARGS__);                                  \
        else if ((N) <= 256) K3_CASE(8, 32, KERNEL, __VA_ARGS__);                                  \
        else if ((N) <= 384) K3_CASE(6, 64, KERNEL, __VA_ARGS__);                                  \
        else K3_CASE(8, 64, KERNEL, __VA_ARGS__);                                                  \
    } while (0)

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
    K3_DISPATCH(n, composite_fwd_kernel, dim3(composite_grid(n_rays, n, rgb != nullptr)), dim3(64 * K3_WAVES), 0,
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
    K3_DISPATCH(n, composite_bwd_kernel, dim3(composite_grid(n_rays, n, true)), dim3(64 * K3_WAVES), 0,
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
