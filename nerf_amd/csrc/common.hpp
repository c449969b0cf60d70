// Shared host/device helpers for the nerfx library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/nerfx.h"

namespace nerfx {

// ---- thread-local error text --------------------------------------------
std::string& last_error_ref();
int fail(int code, const char* fmt, ...);

#define NERFX_HIP(expr)                                                              \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess)                                                        \
            return ::nerfx::fail(NERFX_E_HIP, "%s failed: %s (%s:%d)", #expr,       \
                                 hipGetErrorString(_e), __FILE__, __LINE__);         \
    } while (0)

#define NERFX_REQUIRE(cond, ...)                                                     \
    do {                                                                             \
        if (!(cond)) return ::nerfx::fail(NERFX_E_ARG, __VA_ARGS__);                 \
    } while (0)

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NERFX_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

// ---- dynamic LDS beyond the default limit ---------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a setting of one kernel ON ONE DEVICE: a host
// thread that drives several GPUs must make it on each.  `LdsOptIn` remembers, per device, the largest
// size a kernel has been opted into (one static instance per kernel instantiation); `ensure` makes the
// call for the current device when a launch needs more.  Two threads racing here at worst both make
// the same call.
struct LdsOptIn {
    static constexpr int MAX_DEVICES = 64;
    size_t bytes[MAX_DEVICES] = {};
    int ensure(const void* kern, size_t lds_bytes) {
        int dev = 0;
        NERFX_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= MAX_DEVICES || lds_bytes > __atomic_load_n(&bytes[dev], __ATOMIC_RELAXED)) {
            NERFX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            if (dev >= 0 && dev < MAX_DEVICES) __atomic_store_n(&bytes[dev], lds_bytes, __ATOMIC_RELAXED);
        }
        return 0;
    }
};
#define NERFX_OPT_IN_LDS(kern, lds_bytes)                                            \
    do {                                                                             \
        static ::nerfx::LdsOptIn _opt_in;                                            \
        if (int _rc = _opt_in.ensure((const void*)(kern), (lds_bytes))) return _rc;  \
    } while (0)

// ---- encoders ---------------------------------------------------------------
// Output width of an encoder: 3 + 6 L (features.py:97) or 2 F (features.py:133).
__host__ __device__ inline int enc_width(int kind, int n) {
    return kind == NERFX_ENC_POSITIONAL ? 3 + 6 * n : 2 * n;
}
// Number of sin/cos pairs an encoder evaluates per point.
__host__ __device__ inline int enc_pairs(int kind, int n) {
    return kind == NERFX_ENC_POSITIONAL ? 3 * n : n;
}

#ifdef __HIPCC__
// (__fmul_rn / __fadd_rn / __fsub_rn below and in the kernels are HIP's plain operators: they name the
// rounding of one operation, but hipcc still fuses a product with the sum it feeds into one FMA, so
// o + t * d and the interpolations round once -- at most one ulp from the twice-rounded value.)
// Angle of sin/cos pair p.  Positional: freq[p / 3] * v[p % 3], one rounding,
// as `freq * v` in features.py:34.  Fourier: (2 pi v) . b[p] (features.py:51).
template <class TablePtr>   // global or LDS pointer to floats
__device__ __forceinline__ float enc_angle(int kind, TablePtr table, int p, float vx, float vy, float vz) {
    if (kind == NERFX_ENC_POSITIONAL) {
        const int f = p / 3, c = p - 3 * f;
        const float v = c == 0 ? vx : (c == 1 ? vy : vz);
        return __fmul_rn(table[f], v);
    }
    const float two_pi = 6.283185307179586f;
    const float ax = __fmul_rn(two_pi, vx), ay = __fmul_rn(two_pi, vy), az = __fmul_rn(two_pi, vz);
    return fmaf(az, table[3 * p + 2], fmaf(ay, table[3 * p + 1], __fmul_rn(ax, table[3 * p])));
}
// Column of the reference layout that holds sin (which = 0) / cos (1) of pair p.
__host__ __device__ inline int enc_pair_col(int kind, int n, int p, int which) {
    if (kind == NERFX_ENC_POSITIONAL) {
        const int f = p / 3, c = p - 3 * f;
        return 3 + 6 * f + 3 * which + c;          // [v | sin f0 | cos f0 | sin f1 | ...]
    }
    return which * n + p;                          // [sin ... | cos ...]
}

// Coarse z value i of n (ray.py:91): torch.linspace's symmetric form.
__device__ __forceinline__ float linspace_at(float lo, float hi, int n, int i) {
    if (n == 1) return lo;
    const float step = (hi - lo) / (float)(n - 1);
    return i < n / 2 ? __fadd_rn(lo, __fmul_rn(step, (float)i))
                     : __fsub_rn(hi, __fmul_rn(step, (float)(n - 1 - i)));
}
// Stratified z value (ray.py:96-99) from the three neighbouring linspace values.
__device__ __forceinline__ float stratified_at(float lo, float hi, int n, int i, float u) {
    const float t0 = linspace_at(lo, hi, n, i);
    const float a = i == 0 ? t0 : __fmul_rn(0.5f, __fadd_rn(linspace_at(lo, hi, n, i - 1), t0));
    const float b = i == n - 1 ? t0 : __fmul_rn(0.5f, __fadd_rn(t0, linspace_at(lo, hi, n, i + 1)));
    return __fadd_rn(a, __fmul_rn(__fsub_rn(b, a), u));
}
__device__ __forceinline__ float norm3(float x, float y, float z) {
#pragma clang fp contract(off)                      // the same bits at every call site
    return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
}
#endif

}  // namespace nerfx
