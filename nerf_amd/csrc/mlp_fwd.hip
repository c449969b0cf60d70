// K2 entry points: validate, fill the launch arguments from the model plan, dispatch.
#include <atomic>
#include <cstdlib>

#include "mlp_bwd_impl.hpp"

namespace nerfx {
int launch_mlp_bf16(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_fp32(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_bf16_bigenc(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_fp32_bigenc(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_bf16_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s);
int launch_mlp_bf16_bigenc_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s);
int launch_mlp_fp32_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s);
int launch_mlp_m16(int activ, int hd, bool is_static, const MlpArgs& a, hipStream_t s);
int launch_mlp_bf16_train(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_fp32_train(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_bf16_train_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s);
int launch_mlp_fp32_train_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s);
int launch_mlp_bwd_bf16(int activ, int wt, const BwdArgs& a, hipStream_t s);
int launch_mlp_bwd_fp32(int activ, int wt, const BwdArgs& a, hipStream_t s);
int64_t wgrad_partial_bytes(const nerfx_model* m, int prec, uint64_t m_pad);
int launch_wgrad(const nerfx_model* m, int prec, uint64_t m_pad, const void* ztape, const void* gtape, float* partial,
                 float* const* grads, int n_grads, float* found_inf, hipStream_t stream);
}  // namespace nerfx

using namespace nerfx;

// Diagnostic hook (only meaningful in -DNERFX_STAMPS builds): device buffer of 6 cycle buckets.
static unsigned long long* g_stamps = nullptr;
extern "C" void nerfx_debug_set_stamps(unsigned long long* device_buffer) { g_stamps = device_buffer; }

// Diagnostic switch: bf16 inference of the models the 16x16x32 kernel takes can be kept on the
// 32x32x16 kernels (A/B timing, and tests that hold one kernel against the other).  Starts from
// NERFX_NO_M16=1 in the environment.
static std::atomic<int> g_m16{-1};
static bool m16_enabled() {
    int v = g_m16.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("NERFX_NO_M16");
        v = !(e && e[0] == '1');
        g_m16.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
extern "C" void nerfx_debug_set_m16(int enable) { g_m16.store(enable ? 1 : 0, std::memory_order_relaxed); }

static int mlp_launch(const nerfx_model_t* model, int precision, const float* p0, const float* p1, const float* t,
                      int64_t n_points, int n_samples, float* sigma, float* rgb, void* stream, void* tape = nullptr,
                      void* tape_signs = nullptr) {
    NERFX_REQUIRE(model != nullptr, "model is NULL");
    NERFX_REQUIRE(precision == NERFX_PREC_FP32 || precision == NERFX_PREC_BF16, "unknown precision %d", precision);
    NERFX_REQUIRE(n_points >= 0 && n_points < (int64_t)1 << 31, "n_points %lld out of range [0, 2^31)", (long long)n_points);
    if (n_points == 0) return 0;
    NERFX_REQUIRE(p0 && p1 && sigma, "NULL input or output");
    const ModelPlan& p = model->plan[precision];
    const int waves = precision == NERFX_PREC_BF16 ? RingBf16::WAVES : RingFp32::WAVES;
    MlpArgs a{};
    a.stream = (const char*)model->stream_dev[precision];
    a.consts = model->const_dev;
    a.nan_flag = model->nan_flag();
    a.ro = p0; a.rd = p1; a.t = t; a.sigma = sigma; a.rgb = rgb;
    a.n_points = (uint32_t)n_points;
    a.n_samples = (uint32_t)n_samples;
    a.n_blocks = (uint32_t)((n_points + 32 * waves - 1) / (32 * waves));
    a.n_pieces = (uint32_t)((rgb ? p.total_units : p.trunk_units) / PIECE_UNITS);
    a.hd = p.hd; a.resid = p.desc.resid;
    a.ex_kind = p.desc.enc_x_kind; a.ex_n = p.desc.enc_x_n; a.ex_units = p.layer[0].enc_units;
    a.ed_kind = p.desc.enc_d_kind; a.ed_n = p.desc.enc_d_n; a.ed_units = p.layer[2 * p.hd + 1].enc_units;
    a.sigw_off = p.sigw_off; a.sigb_off = p.sigb_off; a.tabx_off = p.tabx_off; a.tabd_off = p.tabd_off;
    a.const_floats = p.const_floats; a.foldb_off = p.foldb_off;
    a.stamps = g_stamps;
    if (tape) {                                                     // training forward: record the tape
        NERFX_REQUIRE(rgb != nullptr, "the training forward needs the colour output");
        a.tape = tape;
        a.tape_signs = tape_signs;
        a.tape_m_pad = tape_m_pad(n_points);
        a.n_blocks = (uint32_t)(a.tape_m_pad / (32 * waves));       // pad columns are written too
        a.tape_ex_tile = p.tape_ex_tile; a.tape_ed_tile = p.tape_ed_tile; a.tape_tiles = p.tape_tiles;
        for (int li = 0; li < p.n_layers; ++li) a.tape_z_tile[li] = p.tape_z_tile[li];
        const bool small_enc = a.ex_units <= ENC_BLOCK && a.ed_units <= ENC_BLOCK;
        if (small_enc && p.wt == 8 && p.desc.resid && (p.hd == 2 || p.hd == 4))
            return (precision == NERFX_PREC_BF16 ? launch_mlp_bf16_train_fixed : launch_mlp_fp32_train_fixed)(p.desc.activ, p.hd, a, (hipStream_t)stream);
        return (precision == NERFX_PREC_BF16 ? launch_mlp_bf16_train : launch_mlp_fp32_train)(p.desc.activ, p.wt, a, (hipStream_t)stream);
    }
    if (precision == NERFX_PREC_BF16 && model->has_m16 && m16_enabled()) {     // the 16x16x32 kernel and its stream
        const ModelPlan& p16 = model->plan[NERFX_LAYOUT_M16];
        a.stream = (const char*)model->stream_dev[NERFX_LAYOUT_M16];
        a.n_pieces = (uint32_t)((rgb ? p16.total_units : p16.trunk_units) / PIECE_UNITS);
        return launch_mlp_m16(p.desc.activ, p.hd, p16.m16_static != 0, a, (hipStream_t)stream);
    }
    // encoders that fit the kernels' register cache take the fast instantiations
    const bool big = a.ex_units > ENC_BLOCK || a.ed_units > ENC_BLOCK;
    if (precision == NERFX_PREC_BF16 && model->has_encmajor) {                 // wide encoders: the encoder-major stream
        const ModelPlan& pe = model->plan[NERFX_STREAM_ENCMAJOR];
        a.stream = (const char*)model->stream_dev[NERFX_STREAM_ENCMAJOR];
        a.n_pieces = (uint32_t)((rgb ? pe.total_units : pe.trunk_units) / PIECE_UNITS);
        return launch_mlp_bf16_bigenc_fixed(p.desc.activ, p.hd, a, (hipStream_t)stream);
    }
    if (!big && p.wt == 8 && p.desc.resid && (p.hd == 2 || p.hd == 4))          // compile-time depth
        return (precision == NERFX_PREC_BF16 ? launch_mlp_bf16_fixed : launch_mlp_fp32_fixed)(p.desc.activ, p.hd, a, (hipStream_t)stream);
    auto launch = precision == NERFX_PREC_BF16 ? (big ? launch_mlp_bf16_bigenc : launch_mlp_bf16)
                                               : (big ? launch_mlp_fp32_bigenc : launch_mlp_fp32);
    return launch(p.desc.activ, p.wt, a, (hipStream_t)stream);
}

extern "C" int nerfx_mlp_fwd_points(const nerfx_model_t* model, int precision, const float* x, const float* d,
                                    int64_t n_points, float* sigma, float* rgb, void* stream) {
    return mlp_launch(model, precision, x, d, nullptr, n_points, 1, sigma, rgb, stream);
}

extern "C" int nerfx_mlp_fwd_rays(const nerfx_model_t* model, int precision, const float* ro, const float* rd,
                                  const float* t, int64_t n_rays, int n_samples, float* sigma, float* rgb,
                                  void* stream) {
    NERFX_REQUIRE(n_rays >= 0 && n_samples >= 1, "bad ray/sample count");
    NERFX_REQUIRE(n_rays == 0 || t != nullptr, "t is NULL");
    return mlp_launch(model, precision, ro, rd, t, n_rays * n_samples, n_samples, sigma, rgb, stream);
}

// ---- training: forward with tape, then backward ------------------------------------------
namespace {
struct TrainWorkspace {
    char *ztape, *gtape;
    char* stape;                 // sign words of the z tape (bf16 with ReLU-like activations), else NULL
    float* partial;
    uint64_t m_pad;
    int64_t bytes;
};
TrainWorkspace carve_train(const nerfx_model_t* model, int precision, int64_t n_points, void* base) {
    const ModelPlan& p = model->plan[precision];
    TrainWorkspace w{};
    w.m_pad = tape_m_pad(n_points);
    const int64_t tape_bytes = (int64_t)p.tape_tiles * (int64_t)w.m_pad * 32 * (precision == NERFX_PREC_BF16 ? 2 : 4);
    char* c = (char*)base;
    w.ztape = c; c += tape_bytes;
    w.gtape = c; c += tape_bytes;
    if (tape_has_signs(precision, p.desc.activ)) { w.stape = c; c += (int64_t)p.tape_tiles * (int64_t)w.m_pad * 4; }
    w.partial = (float*)c; c += wgrad_partial_bytes(model, precision, w.m_pad);
    w.bytes = c - (char*)base;
    return w;
}
}  // namespace

extern "C" int64_t nerfx_mlp_train_workspace_bytes(const nerfx_model_t* model, int precision, int64_t n_points) {
    if (!model || (precision != NERFX_PREC_FP32 && precision != NERFX_PREC_BF16) || n_points < 0) return -1;
    return carve_train(model, precision, n_points, nullptr).bytes;
}

static int train_fwd(const nerfx_model_t* model, int precision, const float* p0, const float* p1, const float* t,
                     int64_t n_points, int n_samples, float* sigma, float* rgb, void* workspace,
                     int64_t workspace_bytes, void* stream) {
    NERFX_REQUIRE(model != nullptr, "model is NULL");
    NERFX_REQUIRE(precision == NERFX_PREC_FP32 || precision == NERFX_PREC_BF16, "unknown precision %d", precision);
    NERFX_REQUIRE(n_points >= 0 && n_points < (int64_t)1 << 31, "n_points %lld out of range [0, 2^31)", (long long)n_points);
    if (n_points == 0) return 0;
    NERFX_REQUIRE(workspace && workspace_bytes >= nerfx_mlp_train_workspace_bytes(model, precision, n_points),
                  "training workspace too small: %lld < %lld bytes", (long long)workspace_bytes,
                  (long long)nerfx_mlp_train_workspace_bytes(model, precision, n_points));
    const TrainWorkspace w = carve_train(model, precision, n_points, workspace);
    return mlp_launch(model, precision, p0, p1, t, n_points, n_samples, sigma, rgb, stream, w.ztape, w.stape);
}

extern "C" int nerfx_mlp_fwd_train_points(const nerfx_model_t* model, int precision, const float* x, const float* d,
                                          int64_t n_points, float* sigma, float* rgb, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
    return train_fwd(model, precision, x, d, nullptr, n_points, 1, sigma, rgb, workspace, workspace_bytes, stream);
}

extern "C" int nerfx_mlp_fwd_train_rays(const nerfx_model_t* model, int precision, const float* ro, const float* rd,
                                        const float* t, int64_t n_rays, int n_samples, float* sigma, float* rgb,
                                        void* workspace, int64_t workspace_bytes, void* stream) {
    NERFX_REQUIRE(n_rays >= 0 && n_samples >= 1, "bad ray/sample count");
    NERFX_REQUIRE(n_rays == 0 || t != nullptr, "t is NULL");
    return train_fwd(model, precision, ro, rd, t, n_rays * n_samples, n_samples, sigma, rgb, workspace, workspace_bytes, stream);
}

extern "C" int nerfx_mlp_bwd_checked(const nerfx_model_t* model, int precision, int64_t n_points, const float* sigma,
                                     const float* rgb, const float* g_sigma, const float* g_rgb, void* workspace,
                                     int64_t workspace_bytes, float* const* grads, int n_grads, float* found_inf,
                                     void* stream) {
    NERFX_REQUIRE(model != nullptr, "model is NULL");
    NERFX_REQUIRE(precision == NERFX_PREC_FP32 || precision == NERFX_PREC_BF16, "unknown precision %d", precision);
    NERFX_REQUIRE(n_points > 0 && n_points < (int64_t)1 << 31, "n_points %lld out of range (0, 2^31)", (long long)n_points);
    NERFX_REQUIRE(sigma && rgb && g_sigma && g_rgb && grads, "NULL input or output");
    NERFX_REQUIRE(n_grads == model->n_tensors, "expected %d gradient slots, got %d", model->n_tensors, n_grads);
    NERFX_REQUIRE(workspace && workspace_bytes >= nerfx_mlp_train_workspace_bytes(model, precision, n_points),
                  "training workspace too small");
    const ModelPlan& p = model->plan[precision];
    const TrainWorkspace w = carve_train(model, precision, n_points, workspace);
    const int waves = precision == NERFX_PREC_BF16 ? RingBf16::WAVES : RingFp32::WAVES;
    BwdArgs a{};
    a.stream = (const char*)model->stream_bwd_dev[precision];
    a.consts = model->const_dev;
    a.ztape = w.ztape; a.gtape = w.gtape; a.m_pad = w.m_pad;
    a.stape = w.stape;
    a.g_sigma = g_sigma; a.g_rgb = g_rgb; a.sigma = sigma; a.rgb = rgb;
    a.n_points = (uint32_t)n_points;
    a.n_blocks = (uint32_t)(w.m_pad / (32 * waves));
    a.n_pieces = (uint32_t)(p.bwd_units / PIECE_UNITS);
    a.hd = p.hd; a.sigw_off = p.sigw_off; a.const_floats = p.const_floats;
    for (int li = 0; li < p.n_layers; ++li) a.z_tile[li] = p.tape_z_tile[li];
    a.heads_tile = p.tape_heads_tile; a.n_tiles = p.tape_tiles;
    if (int rc = (precision == NERFX_PREC_BF16 ? launch_mlp_bwd_bf16 : launch_mlp_bwd_fp32)(p.desc.activ, p.wt, a, (hipStream_t)stream))
        return rc;
    return launch_wgrad(model, precision, w.m_pad, w.ztape, w.gtape, w.partial, grads, n_grads, found_inf, (hipStream_t)stream);
}

extern "C" int nerfx_mlp_bwd(const nerfx_model_t* model, int precision, int64_t n_points, const float* sigma,
                             const float* rgb, const float* g_sigma, const float* g_rgb, void* workspace,
                             int64_t workspace_bytes, float* const* grads, int n_grads, void* stream) {
    return nerfx_mlp_bwd_checked(model, precision, n_points, sigma, rgb, g_sigma, g_rgb, workspace, workspace_bytes,
                                 grads, n_grads, nullptr, stream);
}
