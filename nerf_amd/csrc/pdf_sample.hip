// K4: hierarchical (inverse-CDF) sampling + merge with the coarse samples.
//
// One 64-lane wavefront owns one ray: interior coarse weights -> pdf -> cdf by a
// wave scan, `searchsorted(right=True)` by binary search in LDS, the lerp of
// ray.py:182-186, then the nc + nf values are put in ascending order (rank merge seeded by the
// bin indices when the fine samples are already ascending, bitonic sort otherwise).
#include "composite.hpp"
#include "philox.hpp"

#include <atomic>

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
        // neighbours first (1, 2, 4, ... 32): the four-rays-per-wavefront kernel below, whose lanes hold
        // four consecutive values each, forms the same sums in the same order without leaving its registers
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) part += __shfl_xor(part, off, 64);
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

// ---- the 64 + 64 case (the reference's configuration), four rays per wavefront ----------------
// Sixteen lanes own a ray and each lane four consecutive samples, so every load and store is 16
// bytes wide, the sums and scans run over 16 lanes plus a short serial part, and the scalar
// bookkeeping of an iteration serves four rays: about a fifth of the wave-instructions per ray of
// the one-ray-per-wavefront kernel above, with the same arithmetic in the same order (the cdf is
// a sum of doubles that never rounds, the interpolation and the linspace are the same expressions),
// so the outputs are the same bits.  Rays whose fine or coarse values are not ascending (NaNs,
// unsorted input) are sorted by the same bitonic network as above.
constexpr int K4F = 64;                            // nc = nf = 64

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
    return v;
}
// The value of the lane N places down its row of 16 lanes (0 for the first N lanes): a DPP operand
// modifier, no LDS round trip.
template <int N>
__device__ __forceinline__ int row_shr(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xF, 0xF, false); }
// The value of the next lane of the row (its own for the last lane).
__device__ __forceinline__ float row_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, false));
}
template <int N>
__device__ __forceinline__ double row_shr(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)row_shr<N>((int)(uint32_t)b), hi = (uint32_t)row_shr<N>((int)(uint32_t)(b >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// exclusive scan over the 16 lanes of a ray
template <class T>
__device__ __forceinline__ T group16_excl_sum(T v) {
    T inc = v;
    inc += row_shr<1>(inc);
    inc += row_shr<2>(inc);
    inc += row_shr<4>(inc);
    inc += row_shr<8>(inc);
    return inc - v;
}

// #{k in [0, 63) : a[k] <= x} for the 63-entry cdf: six fixed steps, no branches
__device__ __forceinline__ int upper_bound63(const float* a, float x) {
    int lo = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) lo += a[lo + step - 1] <= x ? step : 0;
    return lo;
}

// The value of lane l ^ M of the row, M = 1, 2, 4, 8.
template <int M>
__device__ __forceinline__ float row_xor(float v) {
    const int b = __builtin_bit_cast(int, v);
    if constexpr (M == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    else return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(b, (M << 10) | 0x1F));                               // bit mode: xor M
}
// 64 values, four per lane of a 16-lane row (value p = 4 l + e), into ascending order of p: the
// bitonic network, exchanges inside a lane by register, between lanes by DPP / swizzle.
__device__ __forceinline__ void group16_sort64(float (&k)[4], int l) {
    auto inside = [&](int j, int kk) {                          // partner p ^ j, j = 1 or 2
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = e ^ j;
            if (f > e) {
                const bool up = kk >= 64 ? true : (((4 * l + e) & kk) == 0);
                const float lo = fminf(k[e], k[f]), hi = fmaxf(k[e], k[f]);
                k[e] = up ? lo : hi;
                k[f] = up ? hi : lo;
            }
        }
    };
    auto across = [&](auto m_tag, int kk) {                     // partner lane l ^ M, same e
        constexpr int M = decltype(m_tag)::value;
        const bool lower = (l & M) == 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float o = row_xor<M>(k[e]);
            const bool up = kk >= 64 ? true : (((4 * l + e) & kk) == 0);
            k[e] = (lower == up) ? fminf(k[e], o) : fmaxf(k[e], o);
        }
    };
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>;
    inside(1, 2);
    inside(2, 4); inside(1, 4);
    across(I1{}, 8); inside(2, 8); inside(1, 8);
    across(I2{}, 16); across(I1{}, 16); inside(2, 16); inside(1, 16);
    across(I4{}, 32); across(I2{}, 32); across(I1{}, 32); inside(2, 32); inside(1, 32);
    across(I8{}, 64); across(I4{}, 64); across(I2{}, 64); across(I1{}, 64); inside(2, 64); inside(1, 64);
}

// FUSED: `w_c` holds the coarse pass's DENSITIES and the weights are formed here, with the compositor's
// own arithmetic (composite.hpp: same lane geometry as its four-rays-per-wavefront form, so the same
// bits), instead of being written by one kernel and read back by this one: renderer.py:95 and :135 in
// one launch, 1,036 B per ray (densities, z values, direction in; 128 z values out) instead of 1,804.
template <int MODE, bool FUSED>
__global__ __launch_bounds__(64) void pdf_sample_64x64_kernel(const float* __restrict__ t_c, const float* __restrict__ w_c,
                                                              const float* __restrict__ rd,
                                                              int64_t n_rays, const float* __restrict__ u_g, Rng rng,
                                                              float* __restrict__ t_sorted) {
    // MODE = where the inverse-CDF positions come from: 0 linspace(0, 1, 64), 1 the caller's uniforms,
    // 2 drawn in the kernel.  Uniforms are SORTED first: the result is the sorted union of coarse and
    // fine values whatever the order of the uniforms (ray.py:223), and ascending uniforms give
    // ascending fine values, which merge by rank instead of being sorted with the coarse ones.
    constexpr int N = K4F, NB = N - 1, NW = N - 2;
    // The four rays of a wavefront gather from their rows at about the same index at the same time (the
    // fine samples, the bins and the ranks all ascend along the lanes), so rows that start 64 floats
    // apart put all four on ONE of the 64 LDS banks.  Rows of 64 k + 1 floats shift ray g by g banks:
    // the four use different banks whenever their indices agree modulo 4 (a lane's own four values are
    // then only 4-byte aligned and move as pairs of dwords instead of 16 bytes).
    __shared__ float s_cdf[4][N + 1], s_bins[4][N + 1], s_tc[4][N + 1], s_out[4][2 * N + 1];
    __shared__ int s_hist[4][N + 1];
    auto store4 = [](auto* p, auto a, auto b, auto c, auto d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; };
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, i0 = 4 * l;
    float* cdf = s_cdf[g];
    float* bins = s_bins[g];
    float* tc = s_tc[g];
    float* out = s_out[g];
    int* hist = s_hist[g];
    using f4 = float4;
    struct Dir { float x, y, z; };
    auto fetch = [&](int64_t ray4, f4& tv, f4& wv, Dir& dv) {
        const int64_t ray = min(ray4 + g, n_rays - 1);
        tv = *reinterpret_cast<const f4*>(t_c + ray * N + i0);
        wv = *reinterpret_cast<const f4*>(w_c + ray * N + i0);
        if (FUSED) dv = Dir{rd[3 * ray], rd[3 * ray + 1], rd[3 * ray + 2]};
    };
    f4 t_cur, w_cur, t_nxt = {}, w_nxt = {};
    Dir d_cur = {}, d_nxt = {};
    (void)u_g; (void)rng;
    const int64_t first = (int64_t)blockIdx.x * 4, stride = (int64_t)gridDim.x * 4;
    if (first < n_rays) fetch(first, t_cur, w_cur, d_cur);
    asm volatile("" : "+v"(t_cur.x), "+v"(t_cur.y), "+v"(t_cur.z), "+v"(t_cur.w), "+v"(w_cur.x), "+v"(w_cur.y), "+v"(w_cur.z), "+v"(w_cur.w));
    if (first + stride < n_rays) fetch(first + stride, t_nxt, w_nxt, d_nxt);
    for (int64_t ray4 = first; ray4 < n_rays; ray4 += stride) {
        const int64_t ray = ray4 + g;
        const float tq[4] = {t_cur.x, t_cur.y, t_cur.z, t_cur.w};
        float wq[4] = {w_cur.x, w_cur.y, w_cur.z, w_cur.w};
        if (FUSED) {                                                // renderer.py:50-52 on the coarse densities
            float delta[4], alpha[4], x[4], T[4];
            ray_alpha<4>(wq, tq, __shfl_down(tq[0], 1, 16), norm3(d_cur.x, d_cur.y, d_cur.z), i0, N, 4, delta, alpha, x);
            ray_transmittance<4, 16>(x, l, T);
            ray_weights<4>(alpha, T, wq);
        }
        // -- pdf and cdf (ray.py:161-164, :220): interior weights + 1e-5, normalised, summed ----
        float wi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + e;
            wi[e] = (i >= 1 && i <= NW) ? __fadd_rn(wq[e], 1e-5f) : 0.f;
        }
        // the total in the one-ray kernel's order: a butterfly over 64 values, neighbours first, i.e.
        // a lane's four values, then the lanes of the ray at distances 1, 2, 4, 8 (the total only scales
        // the pdf, but the same order gives the same bits as that kernel)
        float total = (wi[0] + wi[1]) + (wi[2] + wi[3]);
        total += row_xor<1>(total); total += row_xor<2>(total); total += row_xor<4>(total); total += row_xor<8>(total);
        double d[4], run = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + e;
            const float pdf = (i >= 1 && i <= NW) ? wi[e] / total : 0.f;
            run += (double)pdf;
            d[e] = run;
        }
        const double before = group16_excl_sum(run);
        float cv[4], bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cv[e] = (float)(before + d[e]);                // cdf[i], i = i0 + e (cdf[0] = 0)
        const float t_next = row_next(tq[0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = __fmul_rn(0.5f, __fadd_rn(tq[e], e < 3 ? tq[e + 1] : t_next));
        store4(cdf + i0, cv[0], cv[1], cv[2], cv[3]);
        store4(bins + i0, bv[0], bv[1], bv[2], bv[3]);
        store4(tc + i0, tq[0], tq[1], tq[2], tq[3]);
        store4(hist + i0, 0, 0, 0, 0);
        if (l == 0) hist[N] = 0;
        wave_sync();
        // -- inverse cdf (ray.py:166-186) --------------------------------------------------
        float uq[4];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) uq[e] = linspace_at(0.f, 1.f, N, i0 + e);
        } else {
            if constexpr (MODE == 1) {
                const f4 uv = *reinterpret_cast<const f4*>(u_g + min(ray, n_rays - 1) * N + i0);
                uq[0] = uv.x; uq[1] = uv.y; uq[2] = uv.z; uq[3] = uv.w;
            } else {
                const Philox4 b = rng_block(rng, (uint64_t)(min(ray, n_rays - 1) * N + i0) >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) uq[e] = rng_to_float(b.w[e]);
            }
            group16_sort64(uq, l);
        }
        float tp[4], bin_lo[4], bin_hi[4];
        int guess[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = uq[e];
            const int idx = upper_bound63(cdf, u);
            static_assert(NB == 63, "the fixed-step search covers 63 entries");
            const int below = max(idx - 1, 0), above = min(idx, NW);
            const float cb = cdf[below], ca = cdf[above], bb = bins[below], ba = bins[above];
            float den = __fsub_rn(ca, cb);
            den = den < 1e-5f ? 1.0f : den;
            const float frac = __fsub_rn(u, cb) / den;
            tp[e] = __fadd_rn(bb, __fmul_rn(frac, __fsub_rn(ba, bb)));
            guess[e] = idx;
            bin_lo[e] = bb; bin_hi[e] = ba;
        }
        // -- the next rays' inputs take over; the ones after them are requested ---------------
        const f4 t_mine = t_cur;
        t_cur = t_nxt; w_cur = w_nxt; d_cur = d_nxt;
        asm volatile("" : "+v"(t_cur.x), "+v"(t_cur.y), "+v"(t_cur.z), "+v"(t_cur.w), "+v"(w_cur.x), "+v"(w_cur.y), "+v"(w_cur.z), "+v"(w_cur.w));
        if (ray4 + 2 * stride < n_rays) fetch(ray4 + 2 * stride, t_nxt, w_nxt, d_nxt);
        // -- merge (ray.py:223) ----------------------------------------------------------------
        const float tm[4] = {t_mine.x, t_mine.y, t_mine.z, t_mine.w};
        const float tp_next = row_next(tp[0]);
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool last = l == 15 && e == 3;
            bad |= !last && !(tp[e] <= (e < 3 ? tp[e + 1] : tp_next));
            bad |= !last && !(tm[e] <= (e < 3 ? tm[e + 1] : t_next));
        }
        const bool unsorted = (__ballot(bad) >> (16 * g)) & 0xFFFFull;
        if (!unsorted) {
            // fine sample j goes to j + #{i : tc[i] <= tp[j]} (a rank next to its bin index), coarse
            // sample i to i + #{j : rank_j <= i} (a histogram of the ranks and a running sum)
            // The sample was interpolated between the mid points around tc[idx]: tc[idx - 1] <= mid point
            // below <= sample and, unless the sum rounded up to it, sample < mid point above <= tc[idx + 1]
            // (a mid point never leaves the interval of its two ascending neighbours).  One coarse value
            // then settles the rank; ties, roundings and the bins at the ends go to the loops.
            int rank[4];
            bool settled = true;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = tp[e];
                const int r0 = guess[e];                                  // 0 .. 63
                const float mid = tc[r0];
                rank[e] = r0 + (mid <= v ? 1 : 0);
                settled &= r0 >= 1 && bin_lo[e] <= v && (mid > v || bin_hi[e] > v || r0 == N - 1);
            }
            if (__builtin_expect(!settled, 0)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = tp[e];
                    int r = rank[e];
                    while (r < N && tc[r] <= v) ++r;
                    while (r > 0 && tc[r - 1] > v) --r;
                    rank[e] = r;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                out[i0 + e + rank[e]] = tp[e];
                atomicAdd(&hist[rank[e]], 1);
            }
            wave_sync();
            const int h0 = hist[i0], h1 = hist[i0 + 1], h2 = hist[i0 + 2], h3 = hist[i0 + 3];
            int c[4] = {h0, h0 + h1, h0 + h1 + h2, h0 + h1 + h2 + h3};
            const int carry = group16_excl_sum(c[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[i0 + e + carry + c[e]] = tm[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = tm[e], b = tp[e];
                out[i0 + e] = a == a ? a : INFINITY;                        // NaNs sort last
                out[N + i0 + e] = b == b ? b : INFINITY;
            }
            wave_sync();
            for (int k = 2; k <= 2 * N; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = l; i < 2 * N; i += 16) {
                        const int partner = i ^ j;
                        if (partner > i) {
                            const float a = out[i], b = out[partner];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) { out[i] = b; out[partner] = a; }
                        }
                    }
                    wave_sync();
                }
        }
        wave_sync();
        if (ray < n_rays) {                                         // each store of a ray's 16 lanes is 256 contiguous bytes
            float* dst = t_sorted + ray * (2 * N) + 4 * l;
            *reinterpret_cast<f4*>(dst) = f4{out[4 * l], out[4 * l + 1], out[4 * l + 2], out[4 * l + 3]};
            *reinterpret_cast<f4*>(dst + N) = f4{out[N + 4 * l], out[N + 4 * l + 1], out[N + 4 * l + 2], out[N + 4 * l + 3]};
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

// Diagnostic switch: the 64 + 64 case can be kept on the general kernel (A/B timing, and the test that
// holds the two kernels against each other bit for bit).
static std::atomic<int> g_k4_fast{1};
static bool k4_fast_enabled() { return g_k4_fast.load(std::memory_order_relaxed) != 0; }

// `u` = the caller's uniforms, else `rng.on` = draw them in the kernel, else linspace(0, 1, nf).
int pdf_sample_launch(const float* t_c, const float* w_c, int64_t n_rays, int nc, int nf, const float* u, const Rng& rng,
                      float* t_sorted, float* cdf, int64_t* idx, float* t_pdf, hipStream_t stream) {
    NERFX_REQUIRE(n_rays >= 0, "n_rays < 0");
    NERFX_REQUIRE(nc >= 3 && nc <= K4_MAX, "nc %d out of range [3, %d]", nc, K4_MAX);
    NERFX_REQUIRE(nf >= 1 && nf <= K4_MAX, "nf %d out of range [1, %d]", nf, K4_MAX);
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(t_c && w_c && t_sorted, "t_c / w_c / t_sorted is NULL");
    const bool aligned = (((uintptr_t)t_c | (uintptr_t)w_c | (uintptr_t)t_sorted) & 15) == 0;
    if (nc == K4F && nf == K4F && !cdf && !idx && !t_pdf && aligned && (((uintptr_t)u & 15) == 0) && k4_fast_enabled()) {
        const int grid4 = (int)std::min<int64_t>((n_rays + 3) / 4, 256 * 64);       // (8,192 workgroups: 3 % slower)
        auto kern = u ? pdf_sample_64x64_kernel<1, false> : (rng.on ? pdf_sample_64x64_kernel<2, false> : pdf_sample_64x64_kernel<0, false>);
        hipLaunchKernelGGL(kern, dim3(grid4), dim3(64), 0, stream, t_c, w_c, nullptr, n_rays, u, rng, t_sorted);
        return launch_status("pdf_sample_64x64_kernel");
    }
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

// Coarse compositing (weights only) + hierarchical sampling.  64 + 64 samples on aligned arrays: one
// kernel; anything else: the compositor into `w_scratch` (n_rays x nc), then the sampler.
int coarse_pdf_launch(const float* sigma_c, const float* t_c, const float* rd, int64_t n_rays, int nc, int nf, const float* u,
                      const Rng& rng, float* w_scratch, float* t_sorted, hipStream_t stream) {
    NERFX_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(sigma_c && t_c && rd && t_sorted, "sigma_c / t_c / rd / t_sorted is NULL");
    const bool aligned = (((uintptr_t)t_c | (uintptr_t)sigma_c | (uintptr_t)t_sorted | (uintptr_t)u) & 15) == 0;
    if (nc == K4F && nf == K4F && aligned && k4_fast_enabled()) {
        const int grid4 = (int)std::min<int64_t>((n_rays + 3) / 4, 256 * 64);
        auto kern = u ? pdf_sample_64x64_kernel<1, true> : (rng.on ? pdf_sample_64x64_kernel<2, true> : pdf_sample_64x64_kernel<0, true>);
        hipLaunchKernelGGL(kern, dim3(grid4), dim3(64), 0, stream, t_c, sigma_c, rd, n_rays, u, rng, t_sorted);
        return launch_status("pdf_sample_64x64_kernel (with the coarse compositor)");
    }
    NERFX_REQUIRE(w_scratch != nullptr, "w_scratch is NULL (needed unless nc = nf = 64 on 16-byte aligned arrays)");
    if (int rc = nerfx_composite_fwd(sigma_c, nullptr, t_c, rd, nullptr, n_rays, nc, w_scratch, nullptr, nullptr, stream)) return rc;
    return pdf_sample_launch(t_c, w_scratch, n_rays, nc, nf, u, rng, t_sorted, nullptr, nullptr, nullptr, stream);
}

}  // namespace nerfx

using namespace nerfx;

extern "C" int nerfx_coarse_pdf_sample(const float* sigma_c, const float* t_c, const float* rd, int64_t n_rays, int nc, int nf,
                                       const float* u, float* w_scratch, float* t_sorted, void* stream) {
    return coarse_pdf_launch(sigma_c, t_c, rd, n_rays, nc, nf, u, Rng{}, w_scratch, t_sorted, (hipStream_t)stream);
}

extern "C" void nerfx_debug_set_k4_fast(int enable) { g_k4_fast.store(enable ? 1 : 0, std::memory_order_relaxed); }

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
