// K2 entry points: validate, fill the launch arguments from the model plan, dispatch.
#include "mlp_fwd_impl.hpp"

namespace nerfx {
int launch_mlp_bf16(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_fp32(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_bf16_bigenc(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_fp32_bigenc(int activ, int wt, const MlpArgs& a, hipStream_t s);
int launch_mlp_bf16_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s);
int launch_mlp_fp32_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s);
}  // namespace nerfx

using namespace nerfx;

// Diagnostic hook (only meaningful in -DNERFX_STAMPS builds): device buffer of 6 cycle buckets.
static unsigned long long* g_stamps = nullptr;
extern "C" void nerfx_debug_set_stamps(unsigned long long* device_buffer) { g_stamps = device_buffer; }

static int mlp_launch(const nerfx_model_t* model, int precision, const float* p0, const float* p1, const float* t,
                      int64_t n_points, int n_samples, float* sigma, float* rgb, void* stream) {
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
    a.ro = p0; a.rd = p1; a.t = t; a.sigma = sigma; a.rgb = rgb;
    a.n_points = (uint32_t)n_points;
    a.n_samples = (uint32_t)n_samples;
    a.n_blocks = (uint32_t)((n_points + 32 * waves - 1) / (32 * waves));
    a.n_pieces = (uint32_t)((rgb ? p.total_units : p.trunk_units) / PIECE_UNITS);
    a.hd = p.hd; a.resid = p.desc.resid;
    a.ex_kind = p.desc.enc_x_kind; a.ex_n = p.desc.enc_x_n; a.ex_units = p.layer[0].enc_units;
    a.ed_kind = p.desc.enc_d_kind; a.ed_n = p.desc.enc_d_n; a.ed_units = p.layer[2 * p.hd + 1].enc_units;
    a.sigw_off = p.sigw_off; a.sigb_off = p.sigb_off; a.tabx_off = p.tabx_off; a.tabd_off = p.tabd_off;
    a.const_floats = p.const_floats;
    a.stamps = g_stamps;
    // encoders that fit the kernels' register cache take the fast instantiations
    const bool big = a.ex_units > ENC_BLOCK || a.ed_units > ENC_BLOCK;
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
