/*
 * nerfx -- C ABI of the MI355X (gfx950) NeRF render/train hot path.
 *
 * The reference (yliess86/NeRF) has no FFI: its boundary is the Python API of
 * `nerf.core`.  Each entry point below names the reference function(s) whose
 * arithmetic it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - Every pointer is a BORROWED device pointer to contiguous row-major fp32
 *    (int64 where stated) unless marked "host".  The library never frees or
 *    keeps caller memory; outputs and workspaces are caller-allocated.
 *  - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and
 *    the call returns without synchronising.
 *  - Return value: 0 on success, a negative NERFX_E_* code on failure;
 *    `nerfx_last_error()` then returns a thread-local message.
 *  - Functions are re-entrant.  A `nerfx_model_t` caches packed weight copies;
 *    it may be used concurrently for rendering but `nerfx_model_refresh` must
 *    not overlap (on another stream) with kernels that read the same handle.
 */
#ifndef NERFX_H
#define NERFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERFX_VERSION 100            /* 0.1.0 */

#define NERFX_E_ARG      (-1)        /* bad argument / unsupported shape      */
#define NERFX_E_HIP      (-2)        /* a HIP runtime call failed             */
#define NERFX_E_NOMEM    (-3)        /* allocation failed (model create only) */

/* encoder kinds */
#define NERFX_ENC_POSITIONAL 0       /* features.py:22-36  */
#define NERFX_ENC_FOURIER    1       /* features.py:40-52  */
/* hidden activations (model.py:28; train.py:235) */
#define NERFX_ACT_RELU       0
#define NERFX_ACT_LEAKY_RELU 1       /* negative slope 0.01 (torch default)   */
#define NERFX_ACT_SILU       2
/* arithmetic of the MLP contractions */
#define NERFX_PREC_FP32      0       /* v_mfma_f32_32x32x2_f32, exact fp32    */
#define NERFX_PREC_BF16      1       /* v_mfma_f32_32x32x16_bf16, fp32 accum  */

typedef struct nerfx_model nerfx_model_t;

/* Shape of one radiance field -- the constructor arguments of
 * `NeRF(phi_x, phi_d, width, depth, activ, resid)` (model.py:22-55). */
typedef struct {
    int32_t enc_x_kind;  /* NERFX_ENC_*                                        */
    int32_t enc_x_n;     /* N_freqs (positional) or features F (fourier)        */
    int32_t enc_d_kind;
    int32_t enc_d_n;
    int32_t width;       /* 16..256, multiple of 16                             */
    int32_t depth;       /* even, >= 2: depth/2 layers in each trunk half       */
    int32_t activ;       /* NERFX_ACT_*                                         */
    int32_t resid;       /* 0/1: concatenate phi_x in front of fc_2's input     */
} nerfx_model_desc_t;

typedef struct {
    int32_t compute_units;
    int32_t wavefront;
    int32_t lds_bytes_per_cu;
    int32_t clock_khz;
    int64_t hbm_bytes;
    char    arch[32];
} nerfx_device_info_t;

/* ---- library ------------------------------------------------------------ */
int         nerfx_version(void);
const char* nerfx_last_error(void);
int         nerfx_device_info(int device, nerfx_device_info_t* out /* host */);

/* ---- model handle -------------------------------------------------------
 * `tensors` (host array of device pointers) lists, in this order:
 *   enc_x table, enc_d table        (freq_bands (N,) or b (F,3); features.py:102,138)
 *   fc_1.i.0.weight, fc_1.i.0.bias  for i in 0..depth/2-1          (model.py:46)
 *   fc_2.i.0.weight, fc_2.i.0.bias  for i in 0..depth/2-1          (model.py:47)
 *   sigma.0.weight, sigma.0.bias    (model.py:49)
 *   features.weight, features.bias  (model.py:51)
 *   rgb.0.weight, rgb.0.bias, rgb.2.weight, rgb.2.bias              (model.py:52-55)
 * i.e. 2 + 2*depth + 8 pointers; weights are torch Linear layout (out, in).
 * create() allocates the packed copies and packs once; refresh() re-packs from
 * the same pointers (call it after every optimizer step). */
int  nerfx_model_create(const nerfx_model_desc_t* desc /* host */,
                        const float* const* tensors /* host array */, int n_tensors,
                        void* stream, nerfx_model_t** out /* host */);
int  nerfx_model_refresh(nerfx_model_t* model, const float* const* tensors, int n_tensors,
                         void* stream);
/* The same for the weights and biases of ONE precision (NERFX_PREC_*): what a training loop
 * that stays in one precision needs per step.  The encoder tables (tensors[0], tensors[1]:
 * buffers, not parameters) are NOT re-read; after changing those call nerfx_model_refresh. */
int  nerfx_model_refresh_precision(nerfx_model_t* model, const float* const* tensors, int n_tensors,
                                   int precision, void* stream);
void nerfx_model_destroy(nerfx_model_t* model);

/* ---- K1: stratified sampling + encoding (standalone form) ----------------
 * Replaces uniform_bounded_z_values + segment_lengths + uniform_bounded_rays
 * (ray.py:91-99, :60-66, :129-135) and the encoders (features.py:32-36, :50-52).
 * `u` (B, n) = the uniforms `rand_like(t)` would draw, or NULL for perturb off.
 * Outputs (any may be NULL): t, delta (B, n); rx (B, n, 3);
 * phi_x (B, n, Ex); phi_d (B, Ed) -- the direction encoding is per ray. */
int nerfx_sample_encode_fwd(const float* ro, const float* rd, int64_t n_rays,
                            float tn, float tf, int n_samples, const float* u,
                            int enc_x_kind, int enc_x_n, const float* enc_x_table,
                            int enc_d_kind, int enc_d_n, const float* enc_d_table,
                            float* t, float* delta, float* rx, float* phi_x, float* phi_d,
                            void* stream);

/* Encode arbitrary points: phi (n_points, E) = encoder(v (n_points, 3)).
 * Replaces PositionalEncoding.forward / FourierFeatures.forward
 * (features.py:104-113, :140-149). */
int nerfx_encode_fwd(const float* v, int64_t n_points, int kind, int n, const float* table,
                     float* phi, void* stream);

/* ---- K2: the radiance-field MLP ------------------------------------------
 * Replaces NeRF.forward (model.py:68-78) including both encoders, both output
 * activations (activations.py:18-20, :33-35) and the two concatenations.
 * Point form: x (M, 3), d (M, 3) -> sigma (M), rgb (M, 3).
 * rgb == NULL computes the density only (what renderer.py:134 consumes). */
int nerfx_mlp_fwd_points(const nerfx_model_t* model, int precision,
                         const float* x, const float* d, int64_t n_points,
                         float* sigma, float* rgb, void* stream);

/* Ray form: sample m = ray * n_samples + i sits at ro + rd * t[m] and looks
 * along rd (ray.py:132-133, :226-227), so positions never touch memory. */
int nerfx_mlp_fwd_rays(const nerfx_model_t* model, int precision,
                       const float* ro, const float* rd, const float* t,
                       int64_t n_rays, int n_samples,
                       float* sigma, float* rgb, void* stream);

/* ---- K2b: training forward and backward of the MLP -----------------------------
 * Replace autograd through NeRF.forward (model.py:68-78) in the reference's
 * training step (train.py:75-86: render, mse_loss, backward).
 * The training forward computes what nerfx_mlp_fwd_* computes and records, in
 * `workspace`, the tape the backward reads: the encoder outputs and every
 * layer's pre-activation, feature-major, in the precision of the contraction.
 * nerfx_mlp_bwd then takes d loss / d sigma (M) and d loss / d rgb (M,3) plus
 * the forward's own sigma / rgb outputs and writes d loss / d parameter for
 * every Linear into `grads`: a host array of device pointers laid out like
 * `tensors` of nerfx_model_create (entries 0 and 1, the encoder tables, are
 * ignored; a NULL entry skips that tensor).  Gradients are written, not
 * accumulated, and are bit-reproducible from run to run.  The workspace must
 * not be touched between the two calls; its size depends on the model, the
 * precision and n_points. */
int64_t nerfx_mlp_train_workspace_bytes(const nerfx_model_t* model, int precision, int64_t n_points);
int nerfx_mlp_fwd_train_points(const nerfx_model_t* model, int precision,
                               const float* x, const float* d, int64_t n_points,
                               float* sigma, float* rgb,
                               void* workspace, int64_t workspace_bytes, void* stream);
int nerfx_mlp_fwd_train_rays(const nerfx_model_t* model, int precision,
                             const float* ro, const float* rd, const float* t,
                             int64_t n_rays, int n_samples, float* sigma, float* rgb,
                             void* workspace, int64_t workspace_bytes, void* stream);
int nerfx_mlp_bwd(const nerfx_model_t* model, int precision, int64_t n_points,
                  const float* sigma, const float* rgb,
                  const float* g_sigma, const float* g_rgb,
                  void* workspace, int64_t workspace_bytes,
                  float* const* grads /* host array */, int n_grads, void* stream);
/* The same, with the GradScaler's check of train.py:81 (scaler.unscale_ looks for inf / NaN in every
 * gradient) made where the gradients are written: *found_inf (a device float the caller has zeroed,
 * e.g. with nerfx_mse_loss_scaled; may be NULL) is set to 1 when any gradient element is not finite. */
int nerfx_mlp_bwd_checked(const nerfx_model_t* model, int precision, int64_t n_points,
                          const float* sigma, const float* rgb,
                          const float* g_sigma, const float* g_rgb,
                          void* workspace, int64_t workspace_bytes,
                          float* const* grads /* host array */, int n_grads, float* found_inf, void* stream);

/* ---- K3: alpha compositing ------------------------------------------------
 * Replaces segment_lengths (ray.py:60-66), raymarch_volume + exclusive_cumprod
 * (renderer.py:23-26, :50-57) and the depth reduction (renderer.py:203).
 * `delta` may be NULL: it is then formed from t and rd.  rgb / rgb_map /
 * depth_map may be NULL (weights only). */
int nerfx_composite_fwd(const float* sigma, const float* rgb, const float* t, const float* rd,
                        const float* delta, int64_t n_rays, int n_samples,
                        float* weights, float* rgb_map, float* depth_map, void* stream);

/* segment_lengths alone (ray.py:50-66): delta (B,n) from t (B,n) and rd (B,3). */
int nerfx_segment_lengths(const float* t, const float* rd, int64_t n_rays, int n_samples,
                          float* delta, void* stream);

/* Backward of the above w.r.t. sigma and rgb (autograd of renderer.py:50-57,
 * :203).  g_rgb_map (B,3), g_depth (B) and g_weights (B,n) may each be NULL
 * (= zero).  Outputs g_sigma (B,n), g_rgb (B,n,3). */
int nerfx_composite_bwd(const float* sigma, const float* rgb, const float* t, const float* rd,
                        const float* delta, int64_t n_rays, int n_samples,
                        const float* g_rgb_map, const float* g_depth, const float* g_weights,
                        float* g_sigma, float* g_rgb, void* stream);

/* ---- K4: hierarchical sampling ---------------------------------------------
 * Replaces pdf_z_values (ray.py:158-188) and the concatenate + sort of pdf_rays
 * (ray.py:220-223).  t_c, w_c (B, nc) from the coarse pass; `u` (B, nf) or NULL
 * for linspace(0, 1, nf).  Outputs: t_sorted (B, nc+nf) required; cdf (B, nc-1),
 * idx int64 (B, nf) and t_pdf (B, nf) optional (NULL to skip). */
int nerfx_pdf_sample(const float* t_c, const float* w_c, int64_t n_rays, int nc, int nf,
                     const float* u, float* t_sorted, float* cdf, int64_t* idx, float* t_pdf,
                     void* stream);

/* The coarse pass's weights and K4 in one call: raymarch_volume's weights (renderer.py:50-55, with
 * segment_lengths, ray.py:60-66) on the coarse densities sigma_c (B, nc), then pdf_z_values and the
 * merge of pdf_rays (ray.py:158-188, :220-223) as nerfx_pdf_sample.  Same results, bit for bit, as
 * nerfx_composite_fwd(sigma_c, NULL, t_c, rd, ...) followed by nerfx_pdf_sample.  With nc = nf = 64
 * and 16-byte aligned arrays it is one kernel and the weights never reach memory; otherwise the two
 * kernels run with the weights in `w_scratch` (B, nc), which may be NULL only in the first case. */
int nerfx_coarse_pdf_sample(const float* sigma_c, const float* t_c, const float* rd, int64_t n_rays,
                            int nc, int nf, const float* u, float* w_scratch, float* t_sorted,
                            void* stream);

/* torch.searchsorted(cdf, u, right=True) (ray.py:174) on caller-supplied rows:
 * cdf (B, n_cdf) ascending, u (B, n_u) -> idx int64 (B, n_u).  Bit-exact. */
int nerfx_searchsorted_right(const float* cdf, const float* u, int64_t n_rows, int n_cdf,
                             int n_u, int64_t* idx, void* stream);

/* ---- fused render -----------------------------------------------------------
 * Replaces BoundedVolumeRaymarcher.render_volume for train=False
 * (renderer.py:198-205): coarse pass (density only) -> K4 -> fine pass ->
 * composite.  nf == 0 renders the coarse pass with colour.  `u_c` (B,nc) /
 * `u_f` (B,nf) are optional uniforms (perturb=True).  `workspace` must hold
 * nerfx_render_workspace_bytes(n_rays, nc, nf) bytes.  Outputs: t, weights
 * (B, nc+nf), depth_map (B), rgb_map (B,3). */
int64_t nerfx_render_workspace_bytes(int64_t n_rays, int nc, int nf);
int nerfx_render_fwd(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision,
                     const float* ro, const float* rd, int64_t n_rays,
                     float tn, float tf, int nc, int nf, const float* u_c, const float* u_f,
                     void* workspace, int64_t workspace_bytes,
                     float* t, float* weights, float* depth_map, float* rgb_map, void* stream);

/* Uniforms for perturb=True drawn inside the sampling kernels (the reference draws them from
 * torch's global generator, ray.py:99, :167): Philox4x32-10 keyed by `seed`; element e of a
 * draw is word e % 4 of counter block e / 4 at `offset`.  Two calls with the same (seed, offset)
 * produce the same jitter; advance `offset` by at least one between calls.  Where a caller
 * passes its own u_c / u_f arrays, those win (test mode). */
typedef struct { uint64_t seed; uint64_t offset; } nerfx_rng_t;
/* nerfx_render_fwd with the generator: `rng` (host, may be NULL = perturb off). */
int nerfx_render_fwd_rng(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision,
                         const float* ro, const float* rd, int64_t n_rays,
                         float tn, float tf, int nc, int nf, const float* u_c, const float* u_f,
                         const nerfx_rng_t* rng, void* workspace, int64_t workspace_bytes,
                         float* t, float* weights, float* depth_map, float* rgb_map, void* stream);
/* The uniforms of draw `which` (0: coarse stratification, 1: inverse-CDF positions; the voxel
 * sampler uses 0..3) as the kernels see them: out[e], e < n.  For tests and for callers that want to replay a render. */
int nerfx_rng_uniform(uint64_t seed, uint64_t offset, int which, int64_t n, float* out, void* stream);

/* ---- training render: forward with tape, backward ----------------------------------------
 * Replace render_volume(train=True) and autograd through it in the reference's training step
 * (train.py:75-80; renderer.py:133-142): the coarse pass and the sample positions carry no
 * gradient, the last pass (`fine`, or `coarse` when nf == 0) records its tape in `workspace`,
 * and nerfx_render_bwd turns d loss / d rgb_map (B,3) [, d depth (B), d weights (B,n); NULL = 0]
 * into d loss / d parameter of that model (`grads` as for nerfx_mlp_bwd).  `t` is the forward's
 * output; the workspace must not be touched between the two calls. */
int64_t nerfx_render_train_workspace_bytes(const nerfx_model_t* last, int precision, int64_t n_rays, int nc, int nf);
int nerfx_render_train_fwd(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision,
                           const float* ro, const float* rd, int64_t n_rays,
                           float tn, float tf, int nc, int nf, const float* u_c, const float* u_f,
                           const nerfx_rng_t* rng, void* workspace, int64_t workspace_bytes,
                           float* t, float* weights, float* depth_map, float* rgb_map, void* stream);
int nerfx_render_bwd(const nerfx_model_t* last, int precision, const float* rd, int64_t n_rays, int nc, int nf,
                     const float* t, const float* g_rgb_map, const float* g_depth, const float* g_weights,
                     void* workspace, int64_t workspace_bytes, float* const* grads /* host array */, int n_grads,
                     void* stream);
/* nerfx_render_bwd with the gradient check of nerfx_mlp_bwd_checked. */
int nerfx_render_bwd_checked(const nerfx_model_t* last, int precision, const float* rd, int64_t n_rays, int nc, int nf,
                             const float* t, const float* g_rgb_map, const float* g_depth, const float* g_weights,
                             void* workspace, int64_t workspace_bytes, float* const* grads /* host array */, int n_grads,
                             float* found_inf, void* stream);

/* The same render for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * workspace is a static device array inside the library, NERFX_STATIC_WORKSPACE_BYTES per GPU,
 * held from the first call on a device until the process ends; the call allocates and frees
 * nothing, and rejects (NERFX_E_ARG) more rays than nerfx_render_static_max_rays(nc, nf)
 * (5,957 for 64 + 64 samples).  All calls on one GPU share that array: calls that overlap on the
 * same GPU (two streams, two threads) race; calls on one stream are ordered and safe.  Results
 * are bit-identical to nerfx_render_fwd's.  The FIRST call on a device (with a given model shape and
 * precision) resolves the array's address and opts its kernels into their LDS sizes, which may load
 * code objects: make one call per device outside any graph capture before capturing. */
#define NERFX_STATIC_WORKSPACE_BYTES (16 * 1024 * 1024)
int64_t nerfx_render_static_max_rays(int nc, int nf);
int nerfx_render_fwd_static(const nerfx_model_t* coarse, const nerfx_model_t* fine, int precision,
                            const float* ro, const float* rd, int64_t n_rays,
                            float tn, float tf, int nc, int nf, const float* u_c, const float* u_f,
                            float* t, float* weights, float* depth_map, float* rgb_map, void* stream);

/* ---- camera rays in, display frames out (the callers either side of the path) ----------
 * Turn-around camera poses: turnaround_poses (data/path.py:104-128) with turnaround
 * (path.py:78-100) for each of `samples` steps of linspace(x0, x1, samples + 1)[:-1];
 * poses (samples, 4, 4) camera-to-world. */
int nerfx_orbit_poses(float theta0, float theta1, float phi0, float phi1, float psy0, float psy1,
                      float radius0, float radius1, int samples, float* poses, void* stream);
/* pinhole_ray_directions (ray.py:10-26): prd (H, W, 3), camera space, not normalised. */
int nerfx_pinhole_directions(int H, int W, float focal, float* prd, void* stream);
/* phinhole_ray_projection (ray.py:30-46): prd (n, 3), c2w (4, 4) -> ro, rd (n, 3), rd unit. */
int nerfx_project_rays(const float* prd, int64_t n_rays, const float* c2w, float* ro, float* rd,
                       void* stream);
/* Both of the above for pixels [pixel_begin, pixel_begin + n_pixels) (row-major) of one pose,
 * without the directions or the per-frame ray arrays of infer.py:114-121 in memory. */
int nerfx_camera_rays(const float* c2w, int H, int W, float focal, int64_t pixel_begin,
                      int64_t n_pixels, float* ro, float* rd, void* stream);
/* Frame epilogue of infer (infer.py:58-66) and of its CLI (infer.py:145-146): depth (n) is
 * NaN-cleaned, shifted by its minimum, divided by (maximum + 1e-10), clipped and scaled to
 * [0, 255] on three channels; rgb (n, 3) is clipped and scaled.  Outputs (each may be NULL):
 * float frames (n, 3) and 8-bit frames (n, 3) truncated like numpy's astype(uint8).
 * `workspace`: nerfx_frame_workspace_bytes() bytes. */
int64_t nerfx_frame_workspace_bytes(void);
int nerfx_frame_finish(const float* depth, const float* rgb, int64_t n_pixels,
                       void* workspace, int64_t workspace_bytes,
                       float* depth_f32, float* rgb_f32, uint8_t* depth_u8, uint8_t* rgb_u8, void* stream);

/* ---- around the render in a training step (the caller `step`, train.py:72-99) -------------
 * A batch of rays out of device-resident arrays: item k = (C, ro, rd)[perm[begin + k]]
 * (perm == NULL: item begin + k), what the reference's DataLoader collates from
 * BlenderDataset.__getitem__ (blender.py:224-235; data/utils.py:26-32).  Arrays are (n_total, 3),
 * perm holds n_total entries; begin + n <= n_total is required with or without perm. */
int nerfx_gather_rays(const float* C, const float* ro, const float* rd, const int64_t* perm,
                      int64_t begin, int64_t n, int64_t n_total,
                      float* C_out, float* ro_out, float* rd_out, void* stream);
/* MSELoss(reduction="mean") (train.py:77, :278) and PSNR = -10 log10(loss) (train.py:92):
 * loss_psnr[0], loss_psnr[1]; grad_seed (n_elements, may be NULL) = d loss / d pred.
 * One fixed summation order: bit-reproducible. */
int nerfx_mse_loss(const float* pred, const float* target, int64_t n_elements, float* loss_psnr,
                   float* grad_seed, void* stream);
/* The same for a step behind a GradScaler (train.py:77-80): grad_seed = d (*loss_scale x loss) / d pred,
 * what scaler.scale(loss).backward() seeds the graph with (loss_psnr stay unscaled); *found_inf
 * (device float, may be NULL) is zeroed for the backward that follows (nerfx_render_bwd_checked). */
int nerfx_mse_loss_scaled(const float* pred, const float* target, int64_t n_elements, float* loss_psnr,
                          float* grad_seed, const float* loss_scale, float* found_inf, void* stream);
/* torch.optim.Adam's update (train.py:246) over flat fp32 arrays of n elements, with the
 * GradScaler's part of train.py:82-85 folded in: gradients are multiplied by *inv_scale when
 * given, and nothing is updated when *found_inf != 0 (device floats, may be NULL).  The update
 * count lives on the device: *step_in updates have been applied, *step_out receives the new
 * count (two distinct floats; swap them between calls). */
int nerfx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay,
                    const float* inv_scale, const float* found_inf,
                    const float* step_in, float* step_out, void* stream);
/* The same taking the loss scale itself (the device float a GradScaler keeps): gradients are
 * multiplied by 1 / *grad_scale, which is scaler.unscale_ (train.py:81) and scaler.step
 * (train.py:83) in one pass with the update. */
int nerfx_adam_step_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                           float lr, float beta1, float beta2, float eps, float weight_decay,
                           const float* grad_scale, const float* found_inf,
                           const float* step_in, float* step_out, void* stream);

/* Distillation loss of distill.py:120-130 and its gradient w.r.t. the student's outputs:
 * loss = mean_i [ mean_j (alpha(s_sigma_i, delta_j) - alpha(t_sigma_i, delta_j))^2
 *                 + mean_c (s_rgb_ic - t_rgb_ic)^2 ],  alpha(x, d) = 1 - exp(-x d)
 * -- the reference broadcasts sigma.view(-1, 1) against the (n,) lengths, so every density meets
 * every length.  sigma, delta (n); rgb (n, 3); workspace n floats; loss 1 float;
 * g_sigma (n), g_rgb (n, 3) may be NULL. */
int nerfx_distill_loss(const float* t_sigma, const float* t_rgb, const float* s_sigma, const float* s_rgb,
                       const float* delta, int64_t n, float* workspace, float* loss,
                       float* g_sigma, float* g_rgb, void* stream);

/* A distillation batch (VoxelSampler.sample, distill.py:62-82) drawn on the device: `voxels` random
 * cells of the voxels^3 grid whose centres are linspace(lo, hi, voxels) per axis, `samples` points
 * per cell with normal jitter of half a cell (|hi - lo| / voxels / 2), directions uniform in
 * [0,1)^3 then normalised, lengths uniform in [d0, d0 + |d1 - d0|).  Outputs rx, rd
 * (voxels * samples, 3) and delta (voxels * samples).  Same generator as nerfx_rng_t. */
int nerfx_voxel_sample(float x0, float x1, float y0, float y1, float z0, float z1, float d0, float d1,
                       int voxels, int samples, uint64_t seed, uint64_t offset,
                       float* rx, float* rd, float* delta, void* stream);

/* ---- diagnostics --------------------------------------------------------------------
 * No counterpart in the reference.  In a library built with -DNERFX_STAMPS the MLP kernels
 * add per-phase cycle counts (6 x uint64: prologue, ring hand-over, MFMA runs, epilogue,
 * heads, encode) into `device_buffer`; NULL switches it off.  A regular build ignores it. */
void nerfx_debug_set_stamps(unsigned long long* device_buffer);
/* No counterpart in the reference.  bf16 inference of 129..256-wide skip models of depth 4 or 8
 * runs on a kernel built on v_mfma_f32_16x16x32_bf16; enable = 0 keeps it on the general
 * v_mfma_f32_32x32x16_bf16 kernels instead (same arithmetic, other summation order), for A/B
 * timing and for tests that hold one kernel against the other.  Process-wide; default on. */
void nerfx_debug_set_m16(int enable);
/* No counterpart in the reference.  nerfx_pdf_sample with 64 coarse and 64 fine samples, no uniforms
 * and only t_sorted requested (what the render paths ask for) runs on a kernel that takes four rays
 * per wavefront; enable = 0 keeps it on the general one-ray-per-wavefront kernel (same bits), for A/B
 * timing and for the test that holds the two against each other.  Process-wide; default on. */
void nerfx_debug_set_k4_fast(int enable);
/* No counterpart in the reference.  bf16 render passes of narrow models (width 16 or 32, depth 2, no
 * skip connection: the distilled student of distill.py:193-195) run in one kernel per pass with the
 * compositor as its epilogue; enable = 0 keeps them on the general kernels (same results within the
 * bf16 bar), for A/B timing and for the test that holds the two against each other.  Process-wide;
 * default on. */
void nerfx_debug_set_narrow(int enable);

/* No counterpart in the reference.  The bf16 kernels evaluate rgb.0(phi_d, features(h)) (model.py:51-53,
 * :76) as one layer on [phi_d | h]: weight [W_r's encoder columns | W_rc W_f], bias b_r + W_rc b_f (W_rc =
 * the chained columns of rgb.0.weight), formed in fp32 at every refresh.  nerfx_debug_folded_head copies
 * that weight (width/2 x (E_d + width), row-major) and bias (width/2) out; nerfx_debug_unfold_grads runs
 * the step that turns the folded layer's gradient (g_w, g_b, same shapes) into the gradients of
 * rgb.0.weight / .bias and features.weight / .bias with the model's current copies of the weights
 * (any output may be NULL).  For tests that hold both against torch. */
int nerfx_debug_folded_head(const nerfx_model_t* model, float* weight, float* bias, void* stream);
int nerfx_debug_unfold_grads(const nerfx_model_t* model, const float* g_w, const float* g_b,
                             float* g_rgb0_weight, float* g_rgb0_bias, float* g_features_weight,
                             float* g_features_bias, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFX_H */
