// Fused render: the kernel sequence behind BoundedVolumeRaymarcher.render_volume
// (train=False).  Everything is enqueued on the caller's stream with no host sync;
// intermediates live in the caller's workspace (16 B per sample at most).
#include "mlp.hpp"
#include "philox.hpp"

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

namespace nerfx {
int sample_t_launch(const float* rd, int64_t n_rays, float tn, float tf, int n_samples, const float* u, const Rng& rng,
                    float* t, float* delta, hipStream_t stream);
int pdf_sample_launch(const float* t_c, const float* w_c, int64_t n_rays, int nc, int nf, const float* u, const Rng& rng,
                      float* t_sorted, float* cdf, int64_t* idx, float* t_pdf, hipStream_t stream);
int coarse_pdf_launch(const float* sigma_c, const float* t_c, const float* rd, int64_t n_rays, int nc, int nf, const float* u,
                      const Rng& rng, float* w_scratch, float* t_sorted, hipStream_t stream);
bool narrow_enabled();
int narrow_render_pass(const nerfx_model* model, const float* ro, const float* rd, const float* t, float tn, float tf,
                       float* t_out, int64_t n_rays, int n, float* weights, float* rgb_map, float* depth, hipStream_t stream);
}  // namespace nerfx

namespace {
// Where the last pass's MLP records its tape (training) -- none for inference.
struct TapeSpot { void* ws; int64_t bytes; };

// The kernel sequence of render_volume.  Uniforms: the caller's arrays where given, else drawn in
// the kernels when `rng` is given, else none (perturb off).
int render_sequence(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision, const float* ro,
                    const float* rd, int64_t n_rays, float tn, float tf, int nc, int nf, const float* u_c,
                    const float* u_f, const nerfx_rng_t* rng, const Workspace& w, const TapeSpot* tape, float* t,
                    float* weights, float* depth_map, float* rgb_map, void* stream) {
    auto draw = [&](uint32_t which, const float* given) {
        return (rng && !given) ? Rng{rng->seed, rng->offset, which, 1} : Rng{};
    };
    // Narrow models (narrow.hpp) render a pass in ONE kernel, bf16 inference only: densities and colours
    // stay in registers and the compositor is that kernel's epilogue.
    auto whole_ray = [&](const nerfx_model_t* m) { return !tape && precision == NERFX_PREC_BF16 && m->narrow.ok && narrow_enabled(); };
    auto last_pass = [&](const nerfx_model_t* m, int n) -> int {     // MLP with colour on the final samples + composite
        if (whole_ray(m)) return narrow_render_pass(m, ro, rd, t, tn, tf, nullptr, n_rays, n, weights, rgb_map, depth_map, (hipStream_t)stream);
        int rc = tape ? nerfx_mlp_fwd_train_rays(m, precision, ro, rd, t, n_rays, n, w.sigma_f, w.rgb_f, tape->ws, tape->bytes, stream)
                      : nerfx_mlp_fwd_rays(m, precision, ro, rd, t, n_rays, n, w.sigma_f, w.rgb_f, stream);
        if (rc) return rc;
        return nerfx_composite_fwd(w.sigma_f, w.rgb_f, t, rd, nullptr, n_rays, n, weights, rgb_map, depth_map, stream);
    };
    int rc;
    if (nf == 0) {                                                              // renderer.py:202
        if ((rc = sample_t_launch(rd, n_rays, tn, tf, nc, u_c, draw(0, u_c), t, nullptr, (hipStream_t)stream))) return rc;
        return last_pass(coarse, nc);
    }
    // coarse pass, density only (renderer.py:134; the colour head's output is unused there)
    const bool plain_t = !u_c && !rng;                      // perturb off: the whole-ray kernel makes its own z values
    if (!(whole_ray(coarse) && plain_t))
        if ((rc = sample_t_launch(rd, n_rays, tn, tf, nc, u_c, draw(0, u_c), w.t_c, nullptr, (hipStream_t)stream))) return rc;
    // then the coarse weights (renderer.py:95) and importance sampling + merge (renderer.py:135)
    if (whole_ray(coarse)) {
        if ((rc = narrow_render_pass(coarse, ro, rd, plain_t ? nullptr : w.t_c, tn, tf, w.t_c, n_rays, nc, w.w_c, nullptr, nullptr,
                                     (hipStream_t)stream))) return rc;
        if ((rc = pdf_sample_launch(w.t_c, w.w_c, n_rays, nc, nf, u_f, draw(1, u_f), t, nullptr, nullptr, nullptr, (hipStream_t)stream))) return rc;
    } else {
        if ((rc = nerfx_mlp_fwd_rays(coarse, precision, ro, rd, w.t_c, n_rays, nc, w.sigma_c, nullptr, stream))) return rc;
        if ((rc = coarse_pdf_launch(w.sigma_c, w.t_c, rd, n_rays, nc, nf, u_f, draw(1, u_f), w.w_c, t, (hipStream_t)stream))) return rc;
    }
    // fine pass on all nc + nf samples (renderer.py:138-142) and depth (renderer.py:203)
    return last_pass(fine, nc + nf);
}

int check_render_args(const nerfx_model_t* coarse, const nerfx_model_t* fine, const float* ro, const float* rd,
                      int64_t n_rays, float tn, float tf, int nc, int nf, const float* t, const float* weights,
                      const float* depth_map, const float* rgb_map) {
    NERFX_REQUIRE(n_rays >= 0, "n_rays < 0");
    NERFX_REQUIRE(nc >= 1 && nf >= 0, "bad sample counts %d + %d", nc, nf);     // renderer.py:165
    NERFX_REQUIRE(tf > tn, "tf must be > tn");                                  // renderer.py:164
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(coarse && (nf == 0 || fine), "model handle is NULL");
    NERFX_REQUIRE(ro && rd && t && weights && depth_map && rgb_map, "NULL ray input or output");
    return 0;
}

// Training workspace: [render workspace | d loss / d sigma | d loss / d rgb | tape of the last pass]
struct TrainSpots { Workspace w; float *g_sigma, *g_rgb; void* tape; int64_t tape_bytes, bytes; };
TrainSpots carve_train(const nerfx_model_t* last, int precision, void* base, int64_t n_rays, int nc, int nf) {
    auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
    TrainSpots s{};
    s.w = carve(base, n_rays, nc, nf);
    char* p = (char*)base + s.w.bytes;
    const int64_t m = n_rays * (nc + nf);
    s.g_sigma = (float*)p; p += up(m * 4);
    s.g_rgb = (float*)p; p += up(m * 12);
    s.tape = p;
    s.tape_bytes = last ? nerfx_mlp_train_workspace_bytes(last, precision, m) : 0;
    p += up(s.tape_bytes > 0 ? s.tape_bytes : 0);
    s.bytes = p - (char*)base;
    return s;
}
}  // namespace

extern "C" int nerfx_render_fwd_rng(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision,
                                    const float* ro, const float* rd, int64_t n_rays, float tn, float tf, int nc,
                                    int nf, const float* u_c, const float* u_f, const nerfx_rng_t* rng, void* workspace,
                                    int64_t workspace_bytes, float* t, float* weights, float* depth_map,
                                    float* rgb_map, void* stream) {
    if (int rc = check_render_args(coarse, fine, ro, rd, n_rays, tn, tf, nc, nf, t, weights, depth_map, rgb_map)) return rc;
    if (n_rays == 0) return 0;
    NERFX_REQUIRE(workspace && workspace_bytes >= nerfx_render_workspace_bytes(n_rays, nc, nf),
                  "workspace too small: %lld < %lld bytes", (long long)workspace_bytes,
                  (long long)nerfx_render_workspace_bytes(n_rays, nc, nf));
    return render_sequence(coarse, fine, precision, ro, rd, n_rays, tn, tf, nc, nf, u_c, u_f, rng, carve(workspace, n_rays, nc, nf),
                           nullptr, t, weights, depth_map, rgb_map, stream);
}

extern "C" int nerfx_render_fwd(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision,
                                const float* ro, const float* rd, int64_t n_rays, float tn, float tf, int nc,
                                int nf, const float* u_c, const float* u_f, void* workspace,
                                int64_t workspace_bytes, float* t, float* weights, float* depth_map,
                                float* rgb_map, void* stream) {
    return nerfx_render_fwd_rng(coarse, fine, precision, ro, rd, n_rays, tn, tf, nc, nf, u_c, u_f, nullptr, workspace,
                                workspace_bytes, t, weights, depth_map, rgb_map, stream);
}

extern "C" int64_t nerfx_render_train_workspace_bytes(const nerfx_model_t* last, int precision, int64_t n_rays, int nc, int nf) {
    if (!last || n_rays < 0 || nc < 1 || nf < 0 || (precision != NERFX_PREC_FP32 && precision != NERFX_PREC_BF16)) return -1;
    return carve_train(last, precision, nullptr, n_rays, nc, nf).bytes;
}

extern "C" int nerfx_render_train_fwd(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision,
                                      const float* ro, const float* rd, int64_t n_rays, float tn, float tf, int nc,
                                      int nf, const float* u_c, const float* u_f, const nerfx_rng_t* rng,
                                      void* workspace, int64_t workspace_bytes, float* t, float* weights,
                                      float* depth_map, float* rgb_map, void* stream) {
    if (int rc = check_render_args(coarse, fine, ro, rd, n_rays, tn, tf, nc, nf, t, weights, depth_map, rgb_map)) return rc;
    if (n_rays == 0) return 0;
    const nerfx_model_t* last = nf > 0 ? fine : coarse;
    const int64_t need = nerfx_render_train_workspace_bytes(last, precision, n_rays, nc, nf);
    NERFX_REQUIRE(need >= 0, "unknown precision %d", precision);
    NERFX_REQUIRE(workspace && workspace_bytes >= need, "training workspace too small: %lld < %lld bytes",
                  (long long)workspace_bytes, (long long)need);
    const TrainSpots s = carve_train(last, precision, workspace, n_rays, nc, nf);
    const TapeSpot tape{s.tape, s.tape_bytes};
    return render_sequence(coarse, fine, precision, ro, rd, n_rays, tn, tf, nc, nf, u_c, u_f, rng, s.w, &tape, t, weights,
                           depth_map, rgb_map, stream);
}

extern "C" int nerfx_render_bwd_checked(const nerfx_model_t* last, int precision, const float* rd, int64_t n_rays, int nc,
                                        int nf, const float* t, const float* g_rgb_map, const float* g_depth,
                                        const float* g_weights, void* workspace, int64_t workspace_bytes,
                                        float* const* grads, int n_grads, float* found_inf, void* stream) {
    NERFX_REQUIRE(n_rays > 0 && nc >= 1 && nf >= 0, "bad ray / sample counts");
    NERFX_REQUIRE(last && rd && t && grads, "NULL argument");
    const int64_t need = nerfx_render_train_workspace_bytes(last, precision, n_rays, nc, nf);
    NERFX_REQUIRE(need >= 0, "unknown precision %d", precision);
    NERFX_REQUIRE(workspace && workspace_bytes >= need, "training workspace too small: %lld < %lld bytes",
                  (long long)workspace_bytes, (long long)need);
    const TrainSpots s = carve_train(last, precision, workspace, n_rays, nc, nf);
    const int n = nc + nf;
    // autograd of renderer.py:50-57, :203, then of model.py:68-78
    if (int rc = nerfx_composite_bwd(s.w.sigma_f, s.w.rgb_f, t, rd, nullptr, n_rays, n, g_rgb_map, g_depth, g_weights,
                                     s.g_sigma, s.g_rgb, stream)) return rc;
    return nerfx_mlp_bwd_checked(last, precision, n_rays * n, s.w.sigma_f, s.w.rgb_f, s.g_sigma, s.g_rgb, s.tape,
                                 s.tape_bytes, grads, n_grads, found_inf, stream);
}

extern "C" int nerfx_render_bwd(const nerfx_model_t* last, int precision, const float* rd, int64_t n_rays, int nc, int nf,
                                const float* t, const float* g_rgb_map, const float* g_depth, const float* g_weights,
                                void* workspace, int64_t workspace_bytes, float* const* grads, int n_grads, void* stream) {
    return nerfx_render_bwd_checked(last, precision, rd, n_rays, nc, nf, t, g_rgb_map, g_depth, g_weights, workspace,
                                    workspace_bytes, grads, n_grads, nullptr, stream);
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
