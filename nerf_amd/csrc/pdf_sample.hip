// K4: hierarchical (inverse-CDF) sampling + merge with the coarse samples.
//
// One 64-lane wavefront owns one ray: interior coarse weights -> pdf -> cdf by a
// wave scan, `searchsorted(right=True)` by binary search in LDS, the lerp of
// ray.py:182-186, then the nc + nf values are put in ascending order (rank merge seeded by the
// bin indices when the fine samples are already ascending, bitonic sort otherwise).
#include "common.hpp"
#include "philox.hpp"

namespace nerfx {

constexpr int K4_MAX = 256;                       // max nc and max nf

// cumsum in double, rounded to fp32 per element: what torch's CPU cumsum does.
__device__ __forceinline__ double wave_incl_sum(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// #{k in [0, n) : a[k] <= x}  ==  torch.searchsorted(a, x, right=True)
__device__ __forceinline__ int upper_bound(const float* a, int n, float x) {
    int lo = 0, len = n;
    while (len > 0) {
        const int half = len >> 1;
        const bool go = a[lo + half] <= x;
        lo = go ? lo + half + 1 : lo;
        len = go ? len - half - 1 : half;
    }
    return lo;
}
// #{k in [0, n) : a[k] < x}
__device__ __forceinline__ int lower_bound(const float* a, int n, float x) {
    int lo = 0, len = n;
    while (len > 0) {
        const int half = len >> 1;
        const bool go = a[lo + half] < x;
        lo = go ? lo + half + 1 : lo;
        len = go ? len - half - 1 : half;
    }
    return lo;
}

// The workgroup is ONE wavefront, whose LDS operations complete in issue order: steps that hand
// values between lanes through LDS only need the compiler to keep its memory operations in program
// order (a wavefront-scope fence, which emits no instruction).  A workgroup barrier here would also
// wait for every global access in flight, i.e. for the prefetch of the next ray and for the stores
// of the previous one, once per ray.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// R = values of each row a lane holds: the kernel is instantiated for rows of up to 64, 128 and
// 256 samples, so the usual 64 + 64 case carries no loops or predicates for longer rows.
// MODE = where the inverse-CDF positions come from: 0 linspace(0, 1, nf), 1 the caller's uniforms,
// 2 drawn in the kernel; a template parameter so that the unperturbed kernel carries neither the
// uniform rows nor the generator.
template <int R, int MODE>
__global__ __launch_bounds__(64) void pdf_sample_kernel(
    const float* __restrict__ t_c, const float* __restrict__ w_c, int64_t n_rays, int nc, int nf,
    const float* __restrict__ u_g, Rng rng, float* __restrict__ t_sorted, float* __restrict__ cdf_out,
    int64_t* __restrict__ idx_out, float* __restrict__ t_pdf_out) {
    constexpr int ROW = 64 * R;
    __shared__ float tc[ROW], cdf[ROW], bins[ROW], tp[ROW], srt[2 * ROW];
    __shared__ int guess[ROW], hist[ROW + 1];
    __shared__ int unsorted;
    const int lane = threadIdx.x;
    const int nb = nc - 1;                         // bins (mid points) and cdf entries
    const int nw = nc - 2;                         // interior weights
    // The per-ray work is a chain of dependent steps, so a wavefront hides HBM latency only by
    // loading the NEXT ray's inputs (registers) before it starts on the current one.
    float t_cur[R], w_cur[R], u_cur[R], t_nxt[R], w_nxt[R], u_nxt[R];
    auto fetch = [&](int64_t ray, float (&tr)[R], float (&wr)[R], float (&ur)[R]) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int i = lane + 64 * q;
            tr[q] = i < nc ? t_c[ray * nc + i] : 0.f;
            wr[q] = i < nc ? w_c[ray * nc + i] : 0.f;
            ur[q] = (MODE == 1 && i < nf) ? u_g[ray * nf + i] : 0.f;
        }
    };
    // The hand-over cur <- nxt sits in the MIDDLE of an iteration, before the output stores: the
    // wait for the prefetched values then also covers only stores issued half an iteration
    // earlier, which have long completed, instead of the stores issued just before the loop turns.
    if ((int64_t)blockIdx.x < n_rays) fetch(blockIdx.x, t_cur, w_cur, u_cur);
#pragma unroll
    for (int q = 0; q < R; ++q) {
        asm volatile("" : "+v"(t_cur[q]), "+v"(w_cur[q]), "+v"(u_cur[q]));       // landed before the loop starts
        t_nxt[q] = 0.f; w_nxt[q] = 0.f; u_nxt[q] = 0.f;
    }
    if ((int64_t)blockIdx.x + gridDim.x < n_rays) fetch((int64_t)blockIdx.x + gridDim.x, t_nxt, w_nxt, u_nxt);
    for (int64_t ray = blockIdx.x; ray < n_rays; ray += gridDim.x) {
        // -- pdf and cdf (ray.py:161-164, :220) --------------------------------
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int i = lane + 64 * q;
            if (i < nc) {
                tc[i] = t_cur[q];
                srt[i] = __fadd_rn(w_cur[q], 1e-5f);                     // weights + 1e-5, read back shifted by one
                if (i >= 1 && i <= nw) part += srt[i];
            }
        }
        if (lane == 0) unsorted = 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        const float total = part;
        wave_sync();
        double carry = 0.0;
#pragma unroll
        for (int q = 0; q < R; ++q) {              // cdf[k + 1] = sum_{j <= k} pdf_j
            const int k = 64 * q + lane;
            if (64 * q >= nw) break;
            const float pdf = k < nw ? srt[k + 1] / total : 0.f;
            const double inc = wave_incl_sum((double)pdf, lane) + carry;
            if (k < nw) cdf[k + 1] = (float)inc;
            carry = __shfl(inc, 63, 64);
        }
        if (lane == 0) cdf[0] = 0.f;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int i = lane + 64 * q;
            if (i < nb) bins[i] = __fmul_rn(0.5f, __fadd_rn(tc[i], tc[i + 1]));
        }
        wave_sync();
        if (cdf_out)
            for (int i = lane; i < nb; i += 64) cdf_out[ray * nb + i] = cdf[i];
        // -- inverse cdf (ray.py:166-186) ---------------------------------------
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int j = lane + 64 * q;
            if (j >= nf) continue;
            const float u = MODE == 1 ? u_cur[q] : (MODE == 2 ? rng_uniform(rng, (uint64_t)(ray * nf + j)) : linspace_at(0.f, 1.f, nf, j));
            const int idx = upper_bound(cdf, nb, u);
            const int below = max(idx - 1, 0), above = min(idx, nw);
            const float cb = cdf[below], ca = cdf[above], bb = bins[below], ba = bins[above];
            float den = __fsub_rn(ca, cb);
            den = den < 1e-5f ? 1.0f : den;
            const float frac = __fsub_rn(u, cb) / den;
            const float t = __fadd_rn(bb, __fmul_rn(frac, __fsub_rn(ba, bb)));
            tp[j] = t;
            guess[j] = idx;
            if (idx_out) idx_out[ray * nf + j] = idx;
            if (t_pdf_out) t_pdf_out[ray * nf + j] = t;
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < R; ++q) {
            t_cur[q] = t_nxt[q]; w_cur[q] = w_nxt[q]; u_cur[q] = u_nxt[q];
            asm volatile("" : "+v"(t_cur[q]), "+v"(w_cur[q]), "+v"(u_cur[q]));   // the wait for the prefetch stays here
        }
        if (ray + 2 * (int64_t)gridDim.x < n_rays) fetch(ray + 2 * (int64_t)gridDim.x, t_nxt, w_nxt, u_nxt);
        // -- merge (ray.py:223) ---------------------------------------------------
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int j = lane + 64 * q;
            if (j + 1 < nf && !(tp[j] <= tp[j + 1])) unsorted = 1;    // benign race: all write 1
            if (j + 1 < nc && !(tc[j] <= tc[j + 1])) unsorted = 1;
        }
        wave_sync();
        const int n = nc + nf;
        float* out = t_sorted + ray * n;
        if (!unsorted) {
            // Both runs ascending: final slot = own index + rank in the other run.
            // Fine sample j came out of bin idx_j, i.e. it lies between the mid points around
            // tc[idx_j], so its rank among the coarse values, r_j = #{i : tc[i] <= tp[j]}, is idx_j
            // or idx_j + 1: start there and let two short scans settle ties and roundings.
            // The coarse values then fill the slots left over: tc[i] goes to i + #{j : r_j <= i},
            // a histogram of r followed by a running sum.
#pragma unroll
            for (int q = 0; q < R; ++q) hist[lane + 64 * q] = 0;
            if (lane == 0) hist[ROW] = 0;
            wave_sync();
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int j = lane + 64 * q;
                if (j >= nf) continue;
                const float v = tp[j];
                int r = min(guess[j], nc);
                while (r < nc && tc[r] <= v) ++r;
                while (r > 0 && tc[r - 1] > v) --r;
                out[j + r] = v;
                atomicAdd(&hist[r], 1);
            }
            wave_sync();
            int carry = 0;                                           // #{j : r_j <= i}: inclusive running sum of hist
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int i = 64 * q + lane;
                if (64 * q >= nc) break;
                int inc = i < nc ? hist[i] : 0;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(inc, off, 64);
                    if (lane >= off) inc += o;
                }
                inc += carry;
                if (i < nc) out[i + inc] = tc[i];
                carry = __shfl(inc, 63, 64);
            }
        } else {
            int p2 = 1;
            while (p2 < n) p2 <<= 1;
            for (int i = lane; i < p2; i += 64) {
                float v = i < nc ? tc[i] : (i < n ? tp[i - nc] : INFINITY);
                srt[i] = v == v ? v : INFINITY;                          // NaNs sort last
            }
            wave_sync();
            for (int k = 2; k <= p2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = lane; i < p2; i += 64) {
                        const int partner = i ^ j;
                        if (partner > i) {
                            const float a = srt[i], b = srt[partner];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) { srt[i] = b; srt[partner] = a; }
                        }
                    }
                    wave_sync();
                }
            for (int i = lane; i < n; i += 64) out[i] = srt[i];
        }
        wave_sync();
    }
}

__global__ __launch_bounds__(256) void searchsorted_right_kernel(
    const float* __restrict__ cdf, const float* __restrict__ u, int64_t n_rows, int n_cdf, int n_u,
    int64_t* __restrict__ idx) {
    const int64_t total = n_rows * n_u;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t row = q / n_u;
        idx[q] = upper_bound(cdf + row * n_cdf, n_cdf, u[q]);
    }
}

// `u` = the caller's uniforms, else `rng.on` = draw them in the kernel, else linspace(0, 1, nf).
int pdf_sample_launch(const float* t_c, const float* w_c, int64_t n_rays, int nc, int nf, const float* u, const Rng& rng,
                      float* t_sorted, float* cdf, int64_t* idx, float* t_pdf, hipStream_t stream) {
    NERFX_REQUIRE(n_rays >= 0, "n_rays < 0");
    NERFX_REQUIRE(nc >= 3 && nc <= K4_MAX, "nc %d out of range [3, %d]", nc, K4_MAX);
    NERFX_REQUIRE(nf >= 1 && nf <= K4_MAX, "nf %d out of range [1, %d]", nf, K4_MAX);
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(t_c && w_c && t_sorted, "t_c / w_c / t_sorted is NULL");
    const int grid = (int)std::min<int64_t>(n_rays, 256 * 32);
    const int longest = std::max(nc, nf);
    using Kernel = void (*)(const float*, const float*, int64_t, int, int, const float*, Rng, float*, float*, int64_t*, float*);
    static const Kernel table[3][3] = {{pdf_sample_kernel<1, 0>, pdf_sample_kernel<1, 1>, pdf_sample_kernel<1, 2>},
                                       {pdf_sample_kernel<2, 0>, pdf_sample_kernel<2, 1>, pdf_sample_kernel<2, 2>},
                                       {pdf_sample_kernel<4, 0>, pdf_sample_kernel<4, 1>, pdf_sample_kernel<4, 2>}};
    const Kernel kern = table[longest <= 64 ? 0 : (longest <= 128 ? 1 : 2)][u ? 1 : (rng.on ? 2 : 0)];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, stream, t_c, w_c, n_rays, nc, nf, u, rng, t_sorted, cdf, idx, t_pdf);
    return launch_status("pdf_sample_kernel");
}

}  // namespace nerfx

using namespace nerfx;

extern "C" int nerfx_pdf_sample(const float* t_c, const float* w_c, int64_t n_rays, int nc, int nf,
                                const float* u, float* t_sorted, float* cdf, int64_t* idx, float* t_pdf,
                                void* stream) {
    return pdf_sample_launch(t_c, w_c, n_rays, nc, nf, u, Rng{}, t_sorted, cdf, idx, t_pdf, (hipStream_t)stream);
}

extern "C" int nerfx_searchsorted_right(const float* cdf, const float* u, int64_t n_rows, int n_cdf,
                                        int n_u, int64_t* idx, void* stream) {
    NERFX_REQUIRE(n_rows >= 0 && n_cdf >= 1 && n_u >= 1, "bad sizes");
    if (n_rows == 0) return 0;
    NERFX_REQUIRE(cdf && u && idx, "cdf / u / idx is NULL");
    const int grid = (int)std::min<int64_t>((n_rows * n_u + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(searchsorted_right_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, cdf, u,
                       n_rows, n_cdf, n_u, idx);
    return launch_status("searchsorted_right_kernel");
}
