// K1 (standalone form): stratified sampling along rays + sin/cos encodings.
//
// One workgroup walks one ray at a time.  Encodings are produced pair-wise
// (one sincos per angle) into an LDS tile laid out exactly like the output rows,
// then streamed to HBM with full-width coalesced stores: the kernel is bound by
// the 4*(n*Ex) bytes per ray it writes, not by the transcendental math.
#include "common.hpp"
#include "philox.hpp"

namespace nerfx {

constexpr int K1_THREADS = 256;
constexpr int K1_TILE_FLOATS = 8192;   // 32 KiB LDS tile

// Encode `count` points whose coordinates sit in LDS (px/py/pz) into `out`
// (count x E, row-major, global).  All threads of the workgroup take part.
__device__ void encode_tile(int kind, int n, const float* __restrict__ table,
                            const float* px, const float* py, const float* pz, int count,
                            float* tile, float* __restrict__ out) {
    const int E = enc_width(kind, n), P = enc_pairs(kind, n);
    const int rows_per_pass = K1_TILE_FLOATS / E;
    for (int r0 = 0; r0 < count; r0 += rows_per_pass) {
        const int rows = min(rows_per_pass, count - r0);
        for (int q = threadIdx.x; q < rows * P; q += K1_THREADS) {
            const int r = q / P, p = q - r * P;
            float s, c;
            sincosf(enc_angle(kind, table, p, px[r0 + r], py[r0 + r], pz[r0 + r]), &s, &c);
            tile[r * E + enc_pair_col(kind, n, p, 0)] = s;
            tile[r * E + enc_pair_col(kind, n, p, 1)] = c;
        }
        if (kind == NERFX_ENC_POSITIONAL)
            for (int q = threadIdx.x; q < rows * 3; q += K1_THREADS) {
                const int r = q / 3, c = q - 3 * r;
                tile[r * E + c] = c == 0 ? px[r0 + r] : (c == 1 ? py[r0 + r] : pz[r0 + r]);
            }
        __syncthreads();
        float* dst = out + (size_t)r0 * E;
        for (int q = threadIdx.x; q < rows * E; q += K1_THREADS) dst[q] = tile[q];
        __syncthreads();
    }
}

constexpr int K1_MAX_SAMPLES = 1024;

__global__ __launch_bounds__(K1_THREADS) void sample_encode_kernel(
    const float* __restrict__ ro, const float* __restrict__ rd, int64_t n_rays, float tn, float tf,
    int n, const float* __restrict__ u, int kx, int nx, const float* __restrict__ tx, int kd, int nd,
    const float* __restrict__ td, float* __restrict__ t_out, float* __restrict__ delta_out,
    float* __restrict__ rx_out, float* __restrict__ phi_x, float* __restrict__ phi_d) {
    __shared__ float tile[K1_TILE_FLOATS];
    __shared__ float ts[K1_MAX_SAMPLES], px[K1_MAX_SAMPLES], py[K1_MAX_SAMPLES], pz[K1_MAX_SAMPLES];
    const int Ex = enc_width(kx, nx), Ed = enc_width(kd, nd);
    for (int64_t ray = blockIdx.x; ray < n_rays; ray += gridDim.x) {
        const float ox = ro[3 * ray], oy = ro[3 * ray + 1], oz = ro[3 * ray + 2];
        const float dx = rd[3 * ray], dy = rd[3 * ray + 1], dz = rd[3 * ray + 2];
        for (int i = threadIdx.x; i < n; i += K1_THREADS) {
            const float t = u ? stratified_at(tn, tf, n, i, u[ray * n + i]) : linspace_at(tn, tf, n, i);
            ts[i] = t;
            px[i] = __fadd_rn(ox, __fmul_rn(dx, t));           // ray.py:132, two roundings
            py[i] = __fadd_rn(oy, __fmul_rn(dy, t));
            pz[i] = __fadd_rn(oz, __fmul_rn(dz, t));
        }
        __syncthreads();
        const float len = norm3(dx, dy, dz);
        for (int i = threadIdx.x; i < n; i += K1_THREADS) {
            if (t_out) t_out[ray * n + i] = ts[i];
            if (delta_out)                                         // ray.py:62-65
                delta_out[ray * n + i] = __fmul_rn(i + 1 < n ? __fsub_rn(ts[i + 1], ts[i]) : 1e10f, len);
            if (rx_out) {
                rx_out[(ray * n + i) * 3 + 0] = px[i];
                rx_out[(ray * n + i) * 3 + 1] = py[i];
                rx_out[(ray * n + i) * 3 + 2] = pz[i];
            }
        }
        if (phi_x) encode_tile(kx, nx, tx, px, py, pz, n, tile, phi_x + (size_t)ray * n * Ex);
        if (phi_d) {
            __syncthreads();
            if (threadIdx.x == 0) { px[0] = dx; py[0] = dy; pz[0] = dz; }
            __syncthreads();
            encode_tile(kd, nd, td, px, py, pz, 1, tile, phi_d + (size_t)ray * Ed);
        }
        __syncthreads();
    }
}

// The form the render path launches: z values (and optionally segment lengths) only, one
// thread per sample, so a wavefront writes 256 contiguous bytes per output.  The neighbour's z
// value is recomputed rather than exchanged.
// `u` = the caller's uniforms, else `rng.on` = draw them here, else no jitter.
__global__ __launch_bounds__(256) void sample_t_kernel(const float* __restrict__ rd, int64_t n_rays, float tn, float tf,
                                                       int n, const float* __restrict__ u, Rng rng, float* __restrict__ t_out,
                                                       float* __restrict__ delta_out) {
    const int64_t total = n_rays * n;
    const bool jitter = u || rng.on;
    auto uni = [&](int64_t e) { return u ? u[e] : rng_uniform(rng, (uint64_t)e); };
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t ray = q / n;
        const int i = (int)(q - ray * n);
        const float t = jitter ? stratified_at(tn, tf, n, i, uni(q)) : linspace_at(tn, tf, n, i);
        if (t_out) t_out[q] = t;
        if (delta_out) {                                               // ray.py:62-65
            float d = 1e10f;
            if (i + 1 < n) d = __fsub_rn(jitter ? stratified_at(tn, tf, n, i + 1, uni(q + 1)) : linspace_at(tn, tf, n, i + 1), t);
            delta_out[q] = __fmul_rn(d, norm3(rd[3 * ray], rd[3 * ray + 1], rd[3 * ray + 2]));
        }
    }
}

// The same for sample counts that are multiples of four (up to 1024): a thread owns four
// consecutive samples of one ray and keeps its sample slot while it walks over rays, so the
// index arithmetic is one 32-bit division per thread and every store is 128 bits wide.
// With the generator, a thread's four samples are exactly one Philox block.
// MODE = where the jitter comes from: 0 none, 1 the caller's uniforms, 2 drawn in the kernel.
template <int MODE>
__global__ __launch_bounds__(256) void sample_t4_kernel(const float* __restrict__ rd, int64_t n_rays, float tn, float tf,
                                                        int n, const float* __restrict__ u, Rng rng, float* __restrict__ t_out,
                                                        float* __restrict__ delta_out) {
    const int tpr = n >> 2, rpb = 256 / tpr;                           // threads per ray, rays per workgroup pass
    const int slot = (int)threadIdx.x / tpr, i0 = ((int)threadIdx.x - slot * tpr) * 4;
    if (slot >= rpb) return;
    float lin[5];                                                      // without jitter the row is the same for every ray
#pragma unroll
    for (int e = 0; e < 5; ++e) lin[e] = linspace_at(tn, tf, n, min(i0 + e, n - 1));
    for (int64_t ray = (int64_t)blockIdx.x * rpb + slot; ray < n_rays; ray += (int64_t)gridDim.x * rpb) {
        const int64_t q = ray * n + i0;
        float t[5];
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 5; ++e) t[e] = lin[e];
        } else {
            float uv[4];
            if (MODE == 1) {
                const float4 v = *reinterpret_cast<const float4*>(u + q);
                uv[0] = v.x; uv[1] = v.y; uv[2] = v.z; uv[3] = v.w;
            } else {
                const Philox4 b = rng_block(rng, (uint64_t)q >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) uv[e] = rng_to_float(b.w[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = stratified_at(tn, tf, n, i0 + e, uv[e]);
            // The first sample of the next four belongs to the next lane, except across a wavefront's edge.
            t[4] = __shfl_down(t[0], 1, 64);
            if (delta_out && i0 + 4 < n && (threadIdx.x & 63) == 63)
                t[4] = stratified_at(tn, tf, n, i0 + 4, MODE == 1 ? u[q + 4] : rng_uniform(rng, (uint64_t)q + 4));
        }
        if (t_out) *reinterpret_cast<float4*>(t_out + q) = make_float4(t[0], t[1], t[2], t[3]);
        if (delta_out) {                                               // ray.py:62-65
            const float len = norm3(rd[3 * ray], rd[3 * ray + 1], rd[3 * ray + 2]);
            float d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = __fmul_rn(i0 + e + 1 < n ? __fsub_rn(t[e + 1], t[e]) : 1e10f, len);
            *reinterpret_cast<float4*>(delta_out + q) = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
}

__global__ __launch_bounds__(K1_THREADS) void encode_points_kernel(
    const float* __restrict__ v, int64_t n_points, int kind, int n, const float* __restrict__ table,
    float* __restrict__ phi) {
    __shared__ float tile[K1_TILE_FLOATS];
    __shared__ float px[K1_THREADS], py[K1_THREADS], pz[K1_THREADS];
    const int E = enc_width(kind, n);
    const int64_t n_tiles = (n_points + K1_THREADS - 1) / K1_THREADS;
    for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int64_t p0 = tl * K1_THREADS;
        const int count = (int)min((int64_t)K1_THREADS, n_points - p0);
        if ((int)threadIdx.x < count) {
            px[threadIdx.x] = v[(p0 + threadIdx.x) * 3 + 0];
            py[threadIdx.x] = v[(p0 + threadIdx.x) * 3 + 1];
            pz[threadIdx.x] = v[(p0 + threadIdx.x) * 3 + 2];
        }
        __syncthreads();
        encode_tile(kind, n, table, px, py, pz, count, tile, phi + (size_t)p0 * E);
        __syncthreads();
    }
}

static int check_encoder(int kind, int n, const float* table, const char* which) {
    NERFX_REQUIRE(kind == NERFX_ENC_POSITIONAL || kind == NERFX_ENC_FOURIER, "%s: unknown encoder kind %d", which, kind);
    NERFX_REQUIRE(n >= 1 && n <= 1024, "%s: encoder size %d out of range [1, 1024]", which, n);
    NERFX_REQUIRE(table != nullptr, "%s: encoder table is NULL", which);
    NERFX_REQUIRE(enc_width(kind, n) <= K1_TILE_FLOATS, "%s: encoding too wide", which);
    return 0;
}

__global__ __launch_bounds__(256) void rng_uniform_kernel(Rng rng, int64_t n, float* __restrict__ out) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) out[q] = rng_uniform(rng, (uint64_t)q);
}

// t (and optionally delta) for n_rays x n_samples: the form the render paths launch.
int sample_t_launch(const float* rd, int64_t n_rays, float tn, float tf, int n_samples, const float* u, const Rng& rng,
                    float* t, float* delta, hipStream_t stream) {
    const bool aligned = (((uintptr_t)t | (uintptr_t)delta | (uintptr_t)u) & 15) == 0;
    if (n_samples % 4 == 0 && aligned) {
        const int rpb = 256 / (n_samples / 4);
        const int grid4 = (int)std::min<int64_t>((n_rays + rpb - 1) / rpb, 256 * 16);
        auto kern = u ? sample_t4_kernel<1> : (rng.on ? sample_t4_kernel<2> : sample_t4_kernel<0>);
        hipLaunchKernelGGL(kern, dim3(grid4), dim3(256), 0, stream, rd, n_rays, tn, tf, n_samples, u, rng, t, delta);
        return launch_status("sample_t4_kernel");
    }
    const int grid = (int)std::min<int64_t>((n_rays * n_samples + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(sample_t_kernel, dim3(grid), dim3(256), 0, stream, rd, n_rays, tn, tf, n_samples, u, rng, t, delta);
    return launch_status("sample_t_kernel");
}

}  // namespace nerfx

using namespace nerfx;

extern "C" int nerfx_sample_encode_fwd(const float* ro, const float* rd, int64_t n_rays, float tn,
                                       float tf, int n_samples, const float* u, int kx, int nx,
                                       const float* tx, int kd, int nd, const float* td, float* t,
                                       float* delta, float* rx, float* phi_x, float* phi_d,
                                       void* stream) {
    NERFX_REQUIRE(n_rays >= 0, "n_rays < 0");
    NERFX_REQUIRE(n_samples >= 1 && n_samples <= K1_MAX_SAMPLES, "n_samples %d out of range [1, %d]",
                  n_samples, K1_MAX_SAMPLES);
    NERFX_REQUIRE(tf > tn, "tf must be > tn");                      // renderer.py:164
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(ro && rd, "ro / rd is NULL");
    if (phi_x) { int e = check_encoder(kx, nx, tx, "phi_x"); if (e) return e; }
    if (phi_d) { int e = check_encoder(kd, nd, td, "phi_d"); if (e) return e; }
    if (!rx && !phi_x && !phi_d) return sample_t_launch(rd, n_rays, tn, tf, n_samples, u, Rng{}, t, delta, (hipStream_t)stream);
    const int grid = (int)std::min<int64_t>(n_rays, 256 * 16);
    hipLaunchKernelGGL(sample_encode_kernel, dim3(grid), dim3(K1_THREADS), 0, (hipStream_t)stream, ro, rd,
                       n_rays, tn, tf, n_samples, u, kx, nx, tx, kd, nd, td, t, delta, rx, phi_x, phi_d);
    return launch_status("sample_encode_kernel");
}

extern "C" int nerfx_encode_fwd(const float* v, int64_t n_points, int kind, int n, const float* table,
                                float* phi, void* stream) {
    NERFX_REQUIRE(n_points >= 0, "n_points < 0");
    if (n_points == 0) return 0;
    NERFX_REQUIRE(v && phi, "v / phi is NULL");
    { int e = check_encoder(kind, n, table, "encode"); if (e) return e; }
    const int64_t tiles = (n_points + K1_THREADS - 1) / K1_THREADS;
    const int grid = (int)std::min<int64_t>(tiles, 256 * 16);
    hipLaunchKernelGGL(encode_points_kernel, dim3(grid), dim3(K1_THREADS), 0, (hipStream_t)stream, v,
                       n_points, kind, n, table, phi);
    return launch_status("encode_points_kernel");
}

extern "C" int nerfx_rng_uniform(uint64_t seed, uint64_t offset, int which, int64_t n, float* out, void* stream) {
    NERFX_REQUIRE(n >= 0 && which >= 0 && which < 256, "bad size or draw index");
    if (n == 0) return 0;
    NERFX_REQUIRE(out != nullptr, "out is NULL");
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(rng_uniform_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Rng{seed, offset, (uint32_t)which, 1}, n, out);
    return launch_status("rng_uniform_kernel");
}
