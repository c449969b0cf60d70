// What surrounds the render in one optimisation step (train.py:72-99): the batch gather, the
// MSE loss with its PSNR and gradient seed, and the Adam update with the loss scaler's inf check.
// Small, bandwidth-bound kernels whose point is that nothing in a step waits for the host.
#include "common.hpp"
#include "philox.hpp"

namespace nerfx {

// ---- batch gather (data/utils.py:26-32 + blender.py:224-235: one item = colour, origin, direction)
__global__ __launch_bounds__(256) void gather_rays_kernel(const float* __restrict__ C, const float* __restrict__ ro,
                                                          const float* __restrict__ rd, const int64_t* __restrict__ perm,
                                                          int64_t begin, int64_t n, int64_t n_total,
                                                          float* __restrict__ C_out, float* __restrict__ ro_out,
                                                          float* __restrict__ rd_out) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
        int64_t src = perm ? perm[begin + q] : begin + q;
        src = src < 0 ? 0 : (src >= n_total ? n_total - 1 : src);           // a bad index must not fault
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            C_out[3 * q + c] = C[3 * src + c];
            ro_out[3 * q + c] = ro[3 * src + c];
            rd_out[3 * q + c] = rd[3 * src + c];
        }
    }
}

// ---- MSE + PSNR + gradient seed (train.py:77, :92) ------------------------------------------
// One workgroup: a fixed summation order, so the loss is bit-reproducible from run to run.
constexpr int MSE_THREADS = 1024;
__global__ __launch_bounds__(MSE_THREADS) void mse_loss_kernel(const float* __restrict__ pred,
                                                               const float* __restrict__ target, int64_t n,
                                                               float* __restrict__ out2, float* __restrict__ seed,
                                                               const float* __restrict__ loss_scale,
                                                               float* __restrict__ found_inf) {
    __shared__ float part[MSE_THREADS / 64];
    // the seed of scaler.scale(loss).backward() (train.py:80): d (scale * loss) / d pred
    const float inv_n = 1.0f / (float)n, two_inv_n = (loss_scale ? *loss_scale : 1.0f) * (2.0f / (float)n);
    if (found_inf && threadIdx.x == 0) *found_inf = 0.f;             // the backward that follows raises it
    float acc = 0.f;
    for (int64_t q = threadIdx.x; q < n; q += MSE_THREADS) {
        const float d = pred[q] - target[q];
        acc = fmaf(d, d, acc);
        if (seed) seed[q] = d * two_inv_n;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < MSE_THREADS / 64; ++w) s += part[w];
        const float loss = s * inv_n;
        out2[0] = loss;
        out2[1] = -10.0f * log10f(loss);
    }
}

// ---- Adam (torch.optim.Adam, train.py:246) with the GradScaler's checks folded in -----------
// step_io[0] is the number of updates applied so far; the kernel applies update number
// step_io[0] + 1 and writes that count to step_io[1] (the caller swaps the two between calls),
// unless *found_inf is non-zero: then nothing changes, as when GradScaler.step skips optim.step.
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                        float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                        int64_t n, float lr, float beta1, float beta2, float eps,
                                                        float weight_decay, const float* __restrict__ inv_scale,
                                                        const float* __restrict__ grad_scale,
                                                        const float* __restrict__ found_inf,
                                                        const float* __restrict__ step_in, float* __restrict__ step_out) {
    const bool skip = found_inf && *found_inf != 0.f;
    const float step = *step_in + (skip ? 0.f : 1.f);
    if (blockIdx.x == 0 && threadIdx.x == 0) *step_out = step;
    if (skip) return;
    // bias corrections in double like the Python scalars of torch.optim.Adam
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    // GradScaler.unscale_ multiplies by 1 / scale, the reciprocal taken once (grad_scaler.py)
    const float unscale = inv_scale ? *inv_scale : (grad_scale ? 1.0f / *grad_scale : 1.0f);
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
        float g = grad[q] * unscale;
        const float p = param[q];
        if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
        const float m = fmaf(1.0f - beta1, g - exp_avg[q], exp_avg[q]);          // lerp_(grad, 1 - beta1)
        const float v = fmaf(1.0f - beta2, g * g, exp_avg_sq[q] * beta2);        // mul_(beta2).addcmul_(g, g, 1 - beta2)
        exp_avg[q] = m;
        exp_avg_sq[q] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        param[q] = fmaf(-step_size, m / denom, p);                               // addcdiv_(m, denom, -step_size)
    }
}

// Distillation batches (distill.py:62-82): `voxels` random cells of a regular voxels^3 grid over
// the domain, `samples` points per cell jittered around its centre with a normal of half a cell,
// random directions (uniform in [0,1)^3, normalised) and segment lengths uniform in [d0, d0 + |d1 - d0|).
// Point e draws: cell index = floor(u * voxels^3) of draw 0 element e / samples; three normals by
// Box-Muller from draw 1 elements 6 e .. 6 e + 5; direction from draw 2 elements 3 e .. 3 e + 2;
// length from draw 3 element e.
__global__ __launch_bounds__(256) void voxel_sample_kernel(float x0, float x1, float y0, float y1, float z0, float z1,
                                                           float d0, float d1, int voxels, int samples, Rng rng,
                                                           float* __restrict__ rx, float* __restrict__ rd,
                                                           float* __restrict__ delta) {
    const int64_t n = (int64_t)voxels * samples;
    const float lo[3] = {x0, y0, z0}, hi[3] = {x1, y1, z1};
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        Rng g = rng;
        g.which = 0;
        const uint32_t cells = (uint32_t)voxels * voxels * voxels;
        uint32_t cell = (uint32_t)(rng_uniform(g, (uint64_t)(e / samples)) * (float)cells);
        cell = cell < cells ? cell : cells - 1;
        const int idx[3] = {(int)(cell / ((uint32_t)voxels * voxels)), (int)(cell / voxels % voxels), (int)(cell % voxels)};
        g.which = 1;
        const Philox4 a = rng_block(g, (uint64_t)(6 * e) >> 2), b = rng_block(g, ((uint64_t)(6 * e) >> 2) + 1);
        const uint32_t w[8] = {a.w[0], a.w[1], a.w[2], a.w[3], b.w[0], b.w[1], b.w[2], b.w[3]};
        const int o = (int)((6 * e) & 3);                           // 6 e is even: the six words start at 0 or 2
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float u1 = rng_to_float(w[o + 2 * c]), u2 = rng_to_float(w[o + 2 * c + 1]);
            const float normal = sqrtf(-2.0f * logf(1.0f - u1)) * cosf(6.283185307179586f * u2);
            const float centre = linspace_at(lo[c], hi[c], voxels, idx[c]);
            const float half_cell = 0.5f * (fabsf(hi[c] - lo[c]) / (float)voxels);
            rx[3 * e + c] = fmaf(normal, half_cell, centre);
        }
        g.which = 2;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = rng_uniform(g, (uint64_t)(3 * e + c));
        const float len = norm3(v[0], v[1], v[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) rd[3 * e + c] = v[c] / len;
        g.which = 3;
        delta[e] = fmaf(fabsf(d1 - d0), rng_uniform(g, (uint64_t)e), d0);
    }
}

}  // namespace nerfx

using namespace nerfx;

extern "C" int nerfx_voxel_sample(float x0, float x1, float y0, float y1, float z0, float z1, float d0, float d1,
                                  int voxels, int samples, uint64_t seed, uint64_t offset, float* rx, float* rd,
                                  float* delta, void* stream) {
    NERFX_REQUIRE(voxels >= 1 && voxels <= 256 && samples >= 1, "voxels %d (1..256) / samples %d out of range", voxels, samples);
    NERFX_REQUIRE(rx && rd && delta, "an output is NULL");
    const int64_t n = (int64_t)voxels * samples;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(voxel_sample_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x0, x1, y0, y1, z0, z1, d0, d1,
                       voxels, samples, Rng{seed, offset, 0, 1}, rx, rd, delta);
    return launch_status("voxel_sample_kernel");
}

extern "C" int nerfx_gather_rays(const float* C, const float* ro, const float* rd, const int64_t* perm,
                                 int64_t begin, int64_t n, int64_t n_total, float* C_out, float* ro_out,
                                 float* rd_out, void* stream) {
    NERFX_REQUIRE(n >= 0 && begin >= 0 && n_total >= 0, "negative size");
    if (n == 0) return 0;
    NERFX_REQUIRE(n_total > 0 && begin + n <= n_total, "rays [%lld, +%lld) outside the %lld stored",
                  (long long)begin, (long long)n, (long long)n_total);
    NERFX_REQUIRE(C && ro && rd && C_out && ro_out && rd_out, "a ray array is NULL");
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(gather_rays_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, ro, rd, perm, begin, n,
                       n_total, C_out, ro_out, rd_out);
    return launch_status("gather_rays_kernel");
}

extern "C" int nerfx_mse_loss_scaled(const float* pred, const float* target, int64_t n_elements, float* loss_psnr,
                                     float* grad_seed, const float* loss_scale, float* found_inf, void* stream) {
    NERFX_REQUIRE(n_elements > 0, "n_elements <= 0");
    NERFX_REQUIRE(pred && target && loss_psnr, "pred / target / loss_psnr is NULL");
    hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(MSE_THREADS), 0, (hipStream_t)stream, pred, target, n_elements,
                       loss_psnr, grad_seed, loss_scale, found_inf);
    return launch_status("mse_loss_kernel");
}

extern "C" int nerfx_mse_loss(const float* pred, const float* target, int64_t n_elements, float* loss_psnr,
                              float* grad_seed, void* stream) {
    return nerfx_mse_loss_scaled(pred, target, n_elements, loss_psnr, grad_seed, nullptr, nullptr, stream);
}

static int adam_launch(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay,
                       const float* inv_scale, const float* grad_scale, const float* found_inf, const float* step_in,
                       float* step_out, void* stream) {
    NERFX_REQUIRE(n >= 0, "n < 0");
    NERFX_REQUIRE(step_in && step_out && step_in != step_out, "step_in / step_out must be two device floats");
    NERFX_REQUIRE(n == 0 || (param && grad && exp_avg && exp_avg_sq), "param / grad / exp_avg / exp_avg_sq is NULL");
    NERFX_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "betas outside [0, 1)");
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 8));
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, inv_scale, grad_scale, found_inf, step_in, step_out);
    return launch_status("adam_step_kernel");
}

extern "C" int nerfx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                               float lr, float beta1, float beta2, float eps, float weight_decay,
                               const float* inv_scale, const float* found_inf, const float* step_in,
                               float* step_out, void* stream) {
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, inv_scale, nullptr,
                       found_inf, step_in, step_out, stream);
}

extern "C" int nerfx_adam_step_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                      float lr, float beta1, float beta2, float eps, float weight_decay,
                                      const float* grad_scale, const float* found_inf, const float* step_in,
                                      float* step_out, void* stream) {
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, nullptr, grad_scale,
                       found_inf, step_in, step_out, stream);
}

// ---- distillation loss (distill.py:120-130) ---------------------------------------------------
// loss = mean_i [ mean_j (a(s_i, j) - a(t_i, j))^2 + mean_c (s_rgb_ic - t_rgb_ic)^2 ],
// a(x_i, j) = 1 - exp(-x_i * delta_j): the reference multiplies sigma.view(-1, 1) by the (B,) vector
// of segment lengths, so EVERY density meets EVERY length (a B x B table there; here row i is a
// loop over j with the lengths in LDS, and the gradient is formed in the same pass).
namespace nerfx {
constexpr int DISTILL_THREADS = 256;
__global__ __launch_bounds__(DISTILL_THREADS) void distill_rows_kernel(
    const float* __restrict__ t_sigma, const float* __restrict__ t_rgb, const float* __restrict__ s_sigma,
    const float* __restrict__ s_rgb, const float* __restrict__ delta, int n, float* __restrict__ row_loss,
    float* __restrict__ g_sigma, float* __restrict__ g_rgb) {
    __shared__ float d_tile[DISTILL_THREADS];
    const int i = blockIdx.x * DISTILL_THREADS + threadIdx.x;
    const bool on = i < n;
    const float ts = on ? t_sigma[i] : 0.f, ss = on ? s_sigma[i] : 0.f;
    const float inv_n = 1.0f / (float)n;
    float sq = 0.f, gs = 0.f;
    for (int j0 = 0; j0 < n; j0 += DISTILL_THREADS) {
        __syncthreads();
        d_tile[threadIdx.x] = j0 + threadIdx.x < n ? delta[j0 + threadIdx.x] : 0.f;
        __syncthreads();
        const int lim = min(DISTILL_THREADS, n - j0);
        for (int j = 0; j < lim; ++j) {
            const float d = d_tile[j];
            const float es = expf(-ss * d), et = expf(-ts * d);
            const float diff = (1.0f - es) - (1.0f - et);
            sq = fmaf(diff, diff, sq);
            gs = fmaf(diff * d, es, gs);                           // d a / d sigma = delta * exp(-sigma delta)
        }
    }
    if (!on) return;
    float rgb_sq = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float diff = s_rgb[3 * i + c] - t_rgb[3 * i + c];
        rgb_sq = fmaf(diff, diff, rgb_sq);
        if (g_rgb) g_rgb[3 * i + c] = 2.0f * diff * (inv_n / 3.0f);
    }
    row_loss[i] = sq * inv_n + rgb_sq * (1.0f / 3.0f);
    if (g_sigma) g_sigma[i] = 2.0f * gs * inv_n * inv_n;
}

__global__ __launch_bounds__(MSE_THREADS) void mean_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ out) {
    __shared__ float part[MSE_THREADS / 64];
    float acc = 0.f;
    for (int64_t q = threadIdx.x; q < n; q += MSE_THREADS) acc += v[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < MSE_THREADS / 64; ++w) s += part[w];
        *out = s / (float)n;
    }
}
}  // namespace nerfx

extern "C" int nerfx_distill_loss(const float* t_sigma, const float* t_rgb, const float* s_sigma, const float* s_rgb,
                                  const float* delta, int64_t n, float* workspace, float* loss, float* g_sigma,
                                  float* g_rgb, void* stream) {
    NERFX_REQUIRE(n > 0 && n < (1 << 30), "n out of range");
    NERFX_REQUIRE(t_sigma && t_rgb && s_sigma && s_rgb && delta && workspace && loss, "a required pointer is NULL");
    hipLaunchKernelGGL(distill_rows_kernel, dim3((unsigned)((n + DISTILL_THREADS - 1) / DISTILL_THREADS)),
                       dim3(DISTILL_THREADS), 0, (hipStream_t)stream, t_sigma, t_rgb, s_sigma, s_rgb, delta, (int)n,
                       workspace, g_sigma, g_rgb);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(MSE_THREADS), 0, (hipStream_t)stream, workspace, n, loss);
    return launch_status("distill loss kernels");
}
