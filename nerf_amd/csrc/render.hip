// Fused render: the kernel sequence behind BoundedVolumeRaymarcher.render_volume
// (train=False).  Everything is enqueued on the caller's stream with no host sync;
// intermediates live in the caller's workspace (16 B per sample at most).
#include "mlp.hpp"

#include <mutex>

using namespace nerfx;

// Workspace of the allocation-free entry point: one copy per GPU the library runs on, held until
// the process ends.  On the host this name only stands for the symbol; device code and HIP calls
// get the per-device address that hipGetSymbolAddress returns (static_scratch()).
static __device__ __align__(256) unsigned char scratch[NERFX_STATIC_WORKSPACE_BYTES];

namespace {
struct Workspace {
    float *t_c, *sigma_c, *w_c, *sigma_f, *rgb_f;
    int64_t bytes;
};
Workspace carve(void* base, int64_t n_rays, int nc, int nf) {
    auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
    Workspace w{};
    char* p = (char*)base;
    const int64_t n = nc + nf;
    auto take = [&](int64_t floats) { float* r = (float*)p; p += up(floats * 4); return r; };
    w.t_c = take(n_rays * nc);
    w.sigma_c = take(n_rays * nc);
    w.w_c = take(n_rays * nc);
    w.sigma_f = take(n_rays * n);
    w.rgb_f = take(n_rays * n * 3);
    w.bytes = p - (char*)base;
    return w;
}
}  // namespace

// Address of `scratch` on the current device, looked up once per device.
static int static_scratch(void** out) {
    constexpr int MAX_DEVICES = 64;
    static std::mutex lock;
    static void* addr[MAX_DEVICES] = {};
    int dev = 0;
    NERFX_HIP(hipGetDevice(&dev));
    NERFX_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device index %d beyond %d", dev, MAX_DEVICES);
    std::lock_guard<std::mutex> hold(lock);
    if (!addr[dev]) {
        void* p = nullptr;
        NERFX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        NERFX_REQUIRE(p != nullptr, "hipGetSymbolAddress returned NULL for the static workspace");
        addr[dev] = p;
    }
    *out = addr[dev];
    return 0;
}

extern "C" int64_t nerfx_render_workspace_bytes(int64_t n_rays, int nc, int nf) {
    if (n_rays < 0 || nc < 1 || nf < 0) return -1;
    return carve(nullptr, n_rays, nc, nf).bytes;
}

extern "C" int nerfx_render_fwd(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision,
                                const float* ro, const float* rd, int64_t n_rays, float tn, float tf, int nc,
                                int nf, const float* u_c, const float* u_f, void* workspace,
                                int64_t workspace_bytes, float* t, float* weights, float* depth_map,
                                float* rgb_map, void* stream) {
    NERFX_REQUIRE(n_rays >= 0, "n_rays < 0");
    NERFX_REQUIRE(nc >= 1 && nf >= 0, "bad sample counts %d + %d", nc, nf);     // renderer.py:165
    NERFX_REQUIRE(tf > tn, "tf must be > tn");                                  // renderer.py:164
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(coarse && (nf == 0 || fine), "model handle is NULL");
    NERFX_REQUIRE(ro && rd && t && weights && depth_map && rgb_map, "NULL ray input or output");
    NERFX_REQUIRE(workspace && workspace_bytes >= nerfx_render_workspace_bytes(n_rays, nc, nf),
                  "workspace too small: %lld < %lld bytes", (long long)workspace_bytes,
                  (long long)nerfx_render_workspace_bytes(n_rays, nc, nf));
    const Workspace w = carve(workspace, n_rays, nc, nf);
    int rc;
    if (nf == 0) {                                                              // renderer.py:202
        if ((rc = nerfx_sample_encode_fwd(ro, rd, n_rays, tn, tf, nc, u_c, 0, 0, nullptr, 0, 0, nullptr, t,
                                          nullptr, nullptr, nullptr, nullptr, stream))) return rc;
        if ((rc = nerfx_mlp_fwd_rays(coarse, precision, ro, rd, t, n_rays, nc, w.sigma_f, w.rgb_f, stream))) return rc;
        return nerfx_composite_fwd(w.sigma_f, w.rgb_f, t, rd, nullptr, n_rays, nc, weights, rgb_map, depth_map, stream);
    }
    // coarse pass, density only (renderer.py:134; the colour head's output is unused there)
    if ((rc = nerfx_sample_encode_fwd(ro, rd, n_rays, tn, tf, nc, u_c, 0, 0, nullptr, 0, 0, nullptr, w.t_c,
                                      nullptr, nullptr, nullptr, nullptr, stream))) return rc;
    if ((rc = nerfx_mlp_fwd_rays(coarse, precision, ro, rd, w.t_c, n_rays, nc, w.sigma_c, nullptr, stream))) return rc;
    if ((rc = nerfx_composite_fwd(w.sigma_c, nullptr, w.t_c, rd, nullptr, n_rays, nc, w.w_c, nullptr, nullptr, stream))) return rc;
    // importance sampling + merge (renderer.py:135)
    if ((rc = nerfx_pdf_sample(w.t_c, w.w_c, n_rays, nc, nf, u_f, t, nullptr, nullptr, nullptr, stream))) return rc;
    // fine pass on all nc + nf samples (renderer.py:138-142) and depth (renderer.py:203)
    const int n = nc + nf;
    if ((rc = nerfx_mlp_fwd_rays(fine, precision, ro, rd, t, n_rays, n, w.sigma_f, w.rgb_f, stream))) return rc;
    return nerfx_composite_fwd(w.sigma_f, w.rgb_f, t, rd, nullptr, n_rays, n, weights, rgb_map, depth_map, stream);
}

extern "C" int64_t nerfx_render_static_max_rays(int nc, int nf) {
    if (nc < 1 || nf < 0) return -1;
    // workspace bytes grow linearly with the ray count (plus at most 5 x 256 B of alignment)
    const int64_t per_ray = (int64_t)(3 * nc + 4 * (nc + nf)) * 4;
    int64_t n = (NERFX_STATIC_WORKSPACE_BYTES - 5 * 256) / per_ray;
    while (n > 0 && nerfx_render_workspace_bytes(n, nc, nf) > NERFX_STATIC_WORKSPACE_BYTES) --n;
    return n;
}

extern "C" int nerfx_render_fwd_static(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision,
                                       const float* ro, const float* rd, int64_t n_rays, float tn, float tf, int nc,
                                       int nf, const float* u_c, const float* u_f, float* t, float* weights,
                                       float* depth_map, float* rgb_map, void* stream) {
    NERFX_REQUIRE(n_rays >= 0, "n_rays < 0");
    NERFX_REQUIRE(nc >= 1 && nf >= 0, "bad sample counts %d + %d", nc, nf);
    const int64_t need = nerfx_render_workspace_bytes(n_rays, nc, nf);
    NERFX_REQUIRE(need <= NERFX_STATIC_WORKSPACE_BYTES,
                  "%lld rays of %d + %d samples need %lld workspace bytes; the static workspace holds %lld "
                  "(at most %lld rays per call: split the batch or use nerfx_render_fwd)",
                  (long long)n_rays, nc, nf, (long long)need, (long long)NERFX_STATIC_WORKSPACE_BYTES,
                  (long long)nerfx_render_static_max_rays(nc, nf));
    if (n_rays == 0) return 0;
    void* ws = nullptr;
    if (int rc = static_scratch(&ws)) return rc;
    // every workspace array is written in full by the kernel that produces it before anything reads
    // it, so nothing of the previous call's contents has to be cleared
    return nerfx_render_fwd(coarse, fine, precision, ro, rd, n_rays, tn, tf, nc, nf, u_c, u_f, ws,
                            NERFX_STATIC_WORKSPACE_BYTES, t, weights, depth_map, rgb_map, stream);
}
