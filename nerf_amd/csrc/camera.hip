// The two ends of a rendered frame: camera rays in (pinhole directions, camera-to-world
// projection, turn-around poses) and display frames out (depth normalisation, clipping, 8-bit
// packing).  All of it is per-pixel streaming work: one thread per pixel, 128-bit-friendly rows.
#include "common.hpp"

// One rounding per operation, as the host code these kernels replace: no fused multiply-adds
// formed behind the source's back (they would also make the fused and the two-step ray
// kernels disagree in the last bit).
#pragma clang fp contract(off)

namespace nerfx {

// ---- turn-around poses (data/path.py:78-128) -----------------------------------------
struct Mat4 {
    float m[4][4];
};
__device__ inline Mat4 matmul4(const Mat4& a, const Mat4& b) {
    Mat4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = a.m[i][0] * b.m[0][j];
#pragma unroll
            for (int k = 1; k < 4; ++k) s = s + a.m[i][k] * b.m[k][j];
            c.m[i][j] = s;
        }
    return c;
}
__device__ inline Mat4 identity4() {
    Mat4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.f : 0.f;
    return r;
}

// `linspace(lo, hi, samples + 1)[:-1]` per parameter, then
// diag(-1,1,1,1) @ R_theta @ R_phi @ R_psy @ T_z(radius), multiplied left to right.
__global__ __launch_bounds__(64) void orbit_poses_kernel(float th0, float th1, float ph0, float ph1,
                                                         float ps0, float ps1, float r0, float r1,
                                                         int samples, float* __restrict__ poses) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= samples) return;
    const float theta = linspace_at(th0, th1, samples + 1, i), phi = linspace_at(ph0, ph1, samples + 1, i);
    const float psy = linspace_at(ps0, ps1, samples + 1, i), radius = linspace_at(r0, r1, samples + 1, i);
    Mat4 flip = identity4(), t = identity4(), p = identity4(), y = identity4(), z = identity4();
    flip.m[0][0] = -1.f;
    float s, c;
    sincosf(theta, &s, &c);                       // zx plane (path.py:69-74)
    t.m[0][0] = c; t.m[0][2] = s; t.m[2][0] = -s; t.m[2][2] = c;
    sincosf(phi, &s, &c);                         // path.py:52-57
    p.m[1][1] = c; p.m[1][2] = -s; p.m[2][1] = s; p.m[2][2] = c;
    sincosf(psy, &s, &c);                         // path.py:35-40
    y.m[0][0] = c; y.m[0][1] = -s; y.m[1][0] = s; y.m[1][1] = c;
    z.m[2][3] = radius;                           // path.py:18-23
    const Mat4 out = matmul4(matmul4(matmul4(matmul4(flip, t), p), y), z);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) poses[(int64_t)i * 16 + r * 4 + q] = out.m[r][q];
}

// ---- camera rays (ray.py:10-46) --------------------------------------------------------
// Camera-space direction of pixel (row j, column i): ((i - W/2)/f, -(j - H/2)/f, -1).
__device__ __forceinline__ void pinhole_dir(int64_t pixel, int H, int W, float focal, float& x, float& y, float& z) {
    const int j = (int)(pixel / W), i = (int)(pixel - (int64_t)j * W);
    x = ((float)i - 0.5f * (float)W) / focal;
    y = -(((float)j - 0.5f * (float)H) / focal);
    z = -1.f;
}
// World-space ray of a camera-space direction: rotate, then normalise (ray.py:43-45).
__device__ __forceinline__ void project_dir(const float* __restrict__ c2w, float px, float py, float pz,
                                            float& ox, float& oy, float& oz, float& dx, float& dy, float& dz) {
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        r[k] = (px * c2w[4 * k] + py * c2w[4 * k + 1]) + pz * c2w[4 * k + 2];
    const float n = sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    dx = r[0] / n; dy = r[1] / n; dz = r[2] / n;
    ox = c2w[3]; oy = c2w[7]; oz = c2w[11];
}

__global__ __launch_bounds__(256) void pinhole_directions_kernel(int H, int W, float focal, float* __restrict__ prd) {
    const int64_t n = (int64_t)H * W;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
        float x, y, z;
        pinhole_dir(q, H, W, focal, x, y, z);
        prd[3 * q] = x; prd[3 * q + 1] = y; prd[3 * q + 2] = z;
    }
}

__global__ __launch_bounds__(256) void project_rays_kernel(const float* __restrict__ prd, int64_t n,
                                                           const float* __restrict__ c2w,
                                                           float* __restrict__ ro, float* __restrict__ rd) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
        float ox, oy, oz, dx, dy, dz;
        project_dir(c2w, prd[3 * q], prd[3 * q + 1], prd[3 * q + 2], ox, oy, oz, dx, dy, dz);
        ro[3 * q] = ox; ro[3 * q + 1] = oy; ro[3 * q + 2] = oz;
        rd[3 * q] = dx; rd[3 * q + 1] = dy; rd[3 * q + 2] = dz;
    }
}

// Both at once for pixels [begin, begin + n) of one pose: nothing but the 64-byte pose is read.
__global__ __launch_bounds__(256) void camera_rays_kernel(const float* __restrict__ c2w, int H, int W, float focal,
                                                          int64_t begin, int64_t n,
                                                          float* __restrict__ ro, float* __restrict__ rd) {
    __shared__ float pose[12];
    if (threadIdx.x < 12) pose[threadIdx.x] = c2w[threadIdx.x];
    __syncthreads();
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
        float px, py, pz, ox, oy, oz, dx, dy, dz;
        pinhole_dir(begin + q, H, W, focal, px, py, pz);
        project_dir(pose, px, py, pz, ox, oy, oz, dx, dy, dz);
        ro[3 * q] = ox; ro[3 * q + 1] = oy; ro[3 * q + 2] = oz;
        rd[3 * q] = dx; rd[3 * q + 1] = dy; rd[3 * q + 2] = dz;
    }
}

// ---- display frames (infer.py:58-66, :145-146) -----------------------------------------
constexpr int FRAME_BLOCKS = 1024;               // partial (min, max) pairs in the workspace

__device__ __forceinline__ float nan_to_num(float v) {      // torch.nan_to_num defaults
    if (v != v) return 0.f;
    if (v == INFINITY) return 3.4028234663852886e38f;
    if (v == -INFINITY) return -3.4028234663852886e38f;
    return v;
}
__device__ __forceinline__ float clip01(float v) {          // torch.clip keeps NaN
    return v != v ? v : fminf(fmaxf(v, 0.f), 1.f);
}
__device__ __forceinline__ void block_minmax(float& lo, float& hi, float* sh_lo, float* sh_hi) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off, 64));
        hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh_lo[wave] = lo; sh_hi[wave] = hi; }
    __syncthreads();
    lo = sh_lo[0]; hi = sh_hi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) { lo = fminf(lo, sh_lo[w]); hi = fmaxf(hi, sh_hi[w]); }
    __syncthreads();
}

__global__ __launch_bounds__(256) void frame_minmax_kernel(const float* __restrict__ depth, int64_t n,
                                                           float* __restrict__ partial) {
    __shared__ float sh_lo[4], sh_hi[4];
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
        const float v = nan_to_num(depth[q]);
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    block_minmax(lo, hi, sh_lo, sh_hi);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = lo; partial[2 * blockIdx.x + 1] = hi; }
}

// depth -> (depth - min) / (max - min + 1e-10), clipped, x 255, on three channels; rgb clipped x 255.
// Float frames (what `infer` returns) and / or truncated 8-bit frames (what the CLI stores).
__global__ __launch_bounds__(256) void frame_finish_kernel(const float* __restrict__ depth, const float* __restrict__ rgb,
                                                           int64_t n, const float* __restrict__ partial, int n_partial,
                                                           float* __restrict__ depth_f, float* __restrict__ rgb_f,
                                                           uint8_t* __restrict__ depth_u8, uint8_t* __restrict__ rgb_u8) {
    __shared__ float sh_lo[4], sh_hi[4];
    float lo = INFINITY, hi = -INFINITY;
    for (int k = threadIdx.x; k < n_partial; k += 256) {
        lo = fminf(lo, partial[2 * k]); hi = fmaxf(hi, partial[2 * k + 1]);
    }
    block_minmax(lo, hi, sh_lo, sh_hi);
    const float span = (hi - lo) + 1e-10f;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
        const float d = clip01((nan_to_num(depth[q]) - lo) / span) * 255.f;
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = clip01(rgb[3 * q + k]) * 255.f;
        if (depth_f) { depth_f[3 * q] = d; depth_f[3 * q + 1] = d; depth_f[3 * q + 2] = d; }
        if (rgb_f) { rgb_f[3 * q] = c[0]; rgb_f[3 * q + 1] = c[1]; rgb_f[3 * q + 2] = c[2]; }
        if (depth_u8) {
            const uint8_t b = (uint8_t)(int)d;
            depth_u8[3 * q] = b; depth_u8[3 * q + 1] = b; depth_u8[3 * q + 2] = b;
        }
        if (rgb_u8) {
#pragma unroll
            for (int k = 0; k < 3; ++k) rgb_u8[3 * q + k] = (uint8_t)(int)c[k];
        }
    }
}

inline int stream_grid(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 256 * 8); }

}  // namespace nerfx

using namespace nerfx;

extern "C" int nerfx_orbit_poses(float theta0, float theta1, float phi0, float phi1, float psy0, float psy1,
                                 float radius0, float radius1, int samples, float* poses, void* stream) {
    NERFX_REQUIRE(samples >= 0, "samples < 0");
    if (samples == 0) return 0;
    NERFX_REQUIRE(poses, "poses is NULL");
    hipLaunchKernelGGL(orbit_poses_kernel, dim3((samples + 63) / 64), dim3(64), 0, (hipStream_t)stream, theta0,
                       theta1, phi0, phi1, psy0, psy1, radius0, radius1, samples, poses);
    return launch_status("orbit_poses_kernel");
}

extern "C" int nerfx_pinhole_directions(int H, int W, float focal, float* prd, void* stream) {
    NERFX_REQUIRE(H >= 0 && W >= 0, "H or W < 0");
    NERFX_REQUIRE(focal != 0.f, "focal is 0");
    if ((int64_t)H * W == 0) return 0;
    NERFX_REQUIRE(prd, "prd is NULL");
    hipLaunchKernelGGL(pinhole_directions_kernel, dim3(stream_grid((int64_t)H * W)), dim3(256), 0,
                       (hipStream_t)stream, H, W, focal, prd);
    return launch_status("pinhole_directions_kernel");
}

extern "C" int nerfx_project_rays(const float* prd, int64_t n_rays, const float* c2w, float* ro, float* rd,
                                  void* stream) {
    NERFX_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(prd && c2w && ro && rd, "prd / c2w / ro / rd is NULL");
    hipLaunchKernelGGL(project_rays_kernel, dim3(stream_grid(n_rays)), dim3(256), 0, (hipStream_t)stream, prd,
                       n_rays, c2w, ro, rd);
    return launch_status("project_rays_kernel");
}

extern "C" int nerfx_camera_rays(const float* c2w, int H, int W, float focal, int64_t pixel_begin,
                                 int64_t n_pixels, float* ro, float* rd, void* stream) {
    NERFX_REQUIRE(H > 0 && W > 0, "H or W <= 0");
    NERFX_REQUIRE(focal != 0.f, "focal is 0");
    NERFX_REQUIRE(pixel_begin >= 0 && n_pixels >= 0 && pixel_begin + n_pixels <= (int64_t)H * W,
                  "pixel range [%lld, +%lld) outside the %d x %d frame", (long long)pixel_begin,
                  (long long)n_pixels, H, W);
    if (n_pixels == 0) return 0;
    NERFX_REQUIRE(c2w && ro && rd, "c2w / ro / rd is NULL");
    hipLaunchKernelGGL(camera_rays_kernel, dim3(stream_grid(n_pixels)), dim3(256), 0, (hipStream_t)stream, c2w, H,
                       W, focal, pixel_begin, n_pixels, ro, rd);
    return launch_status("camera_rays_kernel");
}

extern "C" int64_t nerfx_frame_workspace_bytes(void) { return (int64_t)FRAME_BLOCKS * 2 * sizeof(float); }

extern "C" int nerfx_frame_finish(const float* depth, const float* rgb, int64_t n_pixels, void* workspace,
                                  int64_t workspace_bytes, float* depth_f32, float* rgb_f32, uint8_t* depth_u8,
                                  uint8_t* rgb_u8, void* stream) {
    NERFX_REQUIRE(n_pixels >= 0, "n_pixels < 0");
    if (n_pixels == 0) return 0;
    NERFX_REQUIRE(depth && rgb, "depth / rgb is NULL");
    NERFX_REQUIRE(workspace && workspace_bytes >= nerfx_frame_workspace_bytes(), "workspace too small");
    const int blocks = std::min(stream_grid(n_pixels), FRAME_BLOCKS);
    hipLaunchKernelGGL(frame_minmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, depth, n_pixels,
                       (float*)workspace);
    hipLaunchKernelGGL(frame_finish_kernel, dim3(stream_grid(n_pixels)), dim3(256), 0, (hipStream_t)stream, depth,
                       rgb, n_pixels, (const float*)workspace, blocks, depth_f32, rgb_f32, depth_u8, rgb_u8);
    return launch_status("frame_finish_kernel");
}
