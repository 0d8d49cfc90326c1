/* libipdm_hip.so -- C ABI of the MI355X-native IPDM partial-diffusion sampling hot path.
 *
 * The reference (LFY1998/IPDM-PyTorch) is pure Python on this path: its "FFI" is numba's
 * @jit/@cuda.jit for the FBP convertor and the guidance kernel, and torch.nn for the UNet.  Each
 * entry point below names the reference interface (file:line under /root/reference) it replaces;
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every `d_*` pointer is a DEVICE pointer owned by the caller
 *     (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void*.
 *   - handles (plans / schedules / nets) are owned by the library; distinct handles may be used
 *     from different threads, one handle may not.
 *   - every call is asynchronous on `stream`; no call allocates or synchronises except *_create /
 *     *_destroy (so a caller may capture calls into a hipGraph).
 *   - return value: 0 = IPDM_OK, negative = error (ipdm_last_error() gives the text); nothing
 *     throws across the ABI.
 */
#ifndef IPDM_HIP_H
#define IPDM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPDM_OK 0
#define IPDM_ERR_INVALID (-1)
#define IPDM_ERR_HIP (-2)
#define IPDM_ERR_WORKSPACE (-3)
#define IPDM_ERR_UNSUPPORTED (-4)

const char *ipdm_last_error(void);
/* ABI version of this header (bumped on any signature change or new entry point): 5. */
#define IPDM_ABI_VERSION 5
int ipdm_abi_version(void);

/* Process-wide switches of the library (A/B experiments, opt-in evaluation modes); no reference counterpart -- the
 * reference's only knobs are its option keys (Config/default_config.py), which stay in the Python layer.
 * `name` is lower case, e.g. "conv_no_up2" (Upsample layers in the reference's 3x3 form), "conv_no_wino", "gn_unfused",
 * "unet_transpose" (-1 | 0 | 1), "conv_nm" (0 | 1 | 2: the opt-in 16-cout MFMA form of the narrow layers) ...;
 * README.md lists them.  Every switch starts from the environment variable IPDM_<NAME> (read once, kept
 * as a debug alias) and changes only through this call afterwards.  Switches that shape packed weights or kernel choice
 * are recorded by ipdm_unet_create: a forward on a handle created under other values fails with IPDM_ERR_INVALID instead
 * of running on a mismatched layout; per-call switches (conv_no_up2, conv_no_wup2, conv_no_wino, conv_bf16x3, conv_no_pw, pw_item, pw_force, wino_v1, wino2_min_tiles, conv1x1_no_quarter, conv_nm, direct_no_skip_fuse, gn_unfused,
 * gn_two_stage, unet_transpose, attn_no_zseq, conv_dbg, art_per_view: every weight form they choose between is packed, the workspace
 * need is re-queried per forward) may change under a live handle.  Returns IPDM_ERR_INVALID for an unknown name. */
int ipdm_set_option(const char *name, int value);
int ipdm_get_option(const char *name, int *value);

/* ------------------------------------------------------------------ FBP domain convertor ---- */
/* Geometry of Recon/FBP_kernel.py:28-67 (FBP.__init__); the defaults of the reference are
 * n_views=2000 n_det=912 grid_n=512 da=0.0010125 det_offset=3.75 dtheta_deg=0.18
 * source_origin=59.5 fov_half=21. */
typedef struct ipdm_fbp_geom {
    int32_t n_views, n_det, grid_n;
    double da, det_offset, dtheta_deg, source_origin, fov_half;
} ipdm_fbp_geom;
typedef struct ipdm_fbp_plan ipdm_fbp_plan;

/* replaces FBP.__init__ + getrphi (Recon/FBP_kernel.py:28-84): builds theta/phi/r (float64),
 * nda/h_RL/cos weights (float32) on the host and uploads them. */
int ipdm_fbp_plan_create(const ipdm_fbp_geom *geom, ipdm_fbp_plan **out);
int ipdm_fbp_plan_destroy(ipdm_fbp_plan *plan);
/* bytes of scratch ipdm_fbp_forward needs for a batch of B sinograms (the filtered sinogram). */
size_t ipdm_fbp_workspace_bytes(const ipdm_fbp_plan *plan, int32_t B);
/* replaces FBP.convert (Recon/FBP_kernel.py:86-122) = flip + cos weight + dtheta + conv_pj /
 * conv_kernel (:125-143) + fbp_cpu / fbp_kernel (:146-184) + flip.  d_sino [B,n_views,n_det] f32,
 * d_img [B,grid_n,grid_n] f32 (overwritten).  `gain` multiplies the sinogram first (the G of
 * Utils/train_test_utils.py:455-458,476). */
int ipdm_fbp_forward(ipdm_fbp_plan *plan, const float *d_sino, float *d_img, int32_t B, int32_t flip,
                     float gain, void *d_ws, size_t ws_bytes, void *stream);
/* the two halves separately (parity tests): weighted+ramp-filtered sinogram, and back-projection of
 * an already filtered sinogram (no flips). */
int ipdm_fbp_filter(ipdm_fbp_plan *plan, const float *d_sino, float *d_filtered, int32_t B, int32_t flip,
                    float gain, void *stream);
int ipdm_fbp_backproject(ipdm_fbp_plan *plan, const float *d_filtered, float *d_img, int32_t B,
                         int32_t flip, void *stream);
/* detector coordinate u(t,p) = (alpha - nda[0])/da + 0.5 (float64) of the listed flat pixel indices,
 * d_u [n_views, npix]: the "FBP index map" (Recon/FBP_kernel.py:176-178). */
int ipdm_fbp_index_map(ipdm_fbp_plan *plan, const int32_t *d_pix, int32_t npix, double *d_u, void *stream);
/* host copies of the geometry tables, for parity tests against the reference's FBP.__init__:
 * which = 0 theta[n_views] f64, 1 phi[grid_n^2] f64, 2 r[grid_n^2] f64, 3 nda[n_det] f32,
 * 4 h_RL[2*n_det-1] f32, 5 weight[n_det] f32.  Returns the element count, or <0. */
int64_t ipdm_fbp_table(const ipdm_fbp_plan *plan, int32_t which, void *host_out, int64_t cap_elems);
/* replaces tensor_sharpen (Utils/train_test_utils.py:868-878), per slice, zero padding. */
int ipdm_sharpen3x3(const float *d_in, float *d_out, int32_t B, int32_t H, int32_t W, float n, void *stream);

/* ------------------------------------------------------------------ diffusion schedule ------ */
typedef struct ipdm_schedule ipdm_schedule;
/* replaces cosine_beta_schedule + GaussianDiffusion.__init__ (Model/model.py:366-421), float64. */
int ipdm_schedule_create(int32_t timesteps, double schedule_power, ipdm_schedule **out);
int ipdm_schedule_destroy(ipdm_schedule *s);
/* replaces _extract (Model/model.py:424-428) for the 8 tables the path uses; out[0..7] =
 * sqrt_ac, sqrt_1m_ac, sqrt_recip_ac, sqrt_recipm1_ac, post_mean_coef1, post_mean_coef2,
 * post_log_var_clipped, post_var, each gathered at t and cast to float32. */
int ipdm_schedule_coeffs(const ipdm_schedule *s, int32_t t, float out[8]);
/* alphas_cumprod[t] gathered and cast to float32 (the _extract of ddim_sample, Model/model.py:683-684). */
int ipdm_schedule_alpha_cumprod(const ipdm_schedule *s, int32_t t, float *out);
/* cosine_beta_schedule(ts, schedule_power=power)[i] (Model/model.py:546,552), float64. */
int ipdm_cosine_lambda(int32_t ts, double power, int32_t i, double *out);

/* ------------------------------------------------------------------ DDPM elementwise -------- */
/* counter-based N(0,1) generator (Philox4x32-10 + Box-Muller) replacing torch.randn_like
 * (Model/model.py:440,509).  Element e of slice b of draw `draw` depends only on
 * (seed, slice_id0+b, draw, e): results are invariant to how slices are sharded over GPUs. */
int ipdm_randn(float *d_out, int32_t B, int64_t n_per_slice, uint64_t seed, int64_t slice_id0,
               int64_t draw, void *stream);
/* replaces q_sample (Model/model.py:438-445): out = sa*x + s1m*noise. */
int ipdm_q_sample(const ipdm_schedule *s, int32_t t, const float *d_x, const float *d_noise, float *d_out,
                  int64_t n, void *stream);
size_t ipdm_ddpm_workspace_bytes(int32_t B);
/* replaces p_mean_variance_condition + p_sample_condition (Model/model.py:492-515) with per-slice
 * statistics.  All tensors [B, n_per_slice].  Guidance lambda: scalar `lambda_scalar` when
 * d_lambda_map == NULL, else the small map d_lambda_map [B, mh, mw] nearest-upsampled to [H, W]
 * (F.interpolate rule, Model/model.py:559-560); then n_per_slice must equal H*W. */
int ipdm_ddpm_step(const ipdm_schedule *s, int32_t t, const float *d_eps_pred, const float *d_x_t,
                   const float *d_x0, const float *d_noise, float *d_out, int32_t B, int32_t H, int32_t W,
                   double lambda_scalar, const float *d_lambda_map, int32_t mh, int32_t mw,
                   int32_t clip_denoised, void *d_ws, size_t ws_bytes, void *stream);
/* replaces one iteration of ddim_sample (Model/model.py:654-725; the sparse sampler of
 * sparse_guided_reverse_process :727-759): guided, whitened eps as in ipdm_ddpm_step (scalar lambda), then the DDIM
 * update from timestep t to t_prev.  d_cond is the guide image; d_noise may be NULL when ddim_eta == 0 (the reference
 * still draws it -- the host mirror advances its noise source). */
int ipdm_ddim_step(const ipdm_schedule *s, int32_t t, int32_t t_prev, const float *d_eps_pred, const float *d_x_t,
                   const float *d_cond, const float *d_noise, float *d_out, int32_t B, int64_t n_per_slice,
                   double lambda_scalar, double ddim_eta, int32_t clip_denoised, void *d_ws, size_t ws_bytes,
                   void *stream);
/* elementwise helpers of guided_reverse_process: out = clamp(x) (mode 0: [0,1], 1: min 0)
 * (Model/model.py:569-573); out = a*x + b*y + c*z (guide update :625-635; z may be NULL);
 * out = 0.5*(x+y) (:637-638). */
int ipdm_clamp(const float *d_x, float *d_out, int64_t n, int32_t mode, void *stream);
int ipdm_axpbypcz(const float *d_x, const float *d_y, const float *d_z, float *d_out, int64_t n,
                  double a, double b, double c, void *stream);
/* guidance map after pass 0 (Model/model.py:575-580 img / :596-600,614 proj) + weight_lambda curve
 * (Utils/train_test_utils.py:831-865).  mode 0 = img, 1 = proj.  d_x, d_img [B,H,W];
 * d_Lambda [B, H/k, W/k] f32 (the curve output the lambda kernel exponentiates with);
 * d_expmax [B] f32 = max of exp(amplitude*delta) per slice (adaptive branch, :602-613). */
size_t ipdm_guidance_workspace_bytes(int32_t B, int32_t H, int32_t W);
int ipdm_guidance_map(const float *d_x, const float *d_img, float *d_Lambda, float *d_expmax, int32_t B,
                      int32_t H, int32_t W, int32_t kernel, double amplitude, int32_t mode,
                      const double *p1, const double *p2, void *d_ws, size_t ws_bytes, void *stream);
/* replaces condition_lambda_ratio_cuda (Model/model.py:328-351) + np.clip(.,0.05,0.99) (:558):
 * d_out[B,mh,mw] f32 from d_Lambda for inner step i of a pass of ts steps. */
int ipdm_lambda_ratio(const float *d_Lambda, float *d_out, int64_t n, int32_t i, int32_t ts, void *stream);
/* torch.median over each slice (lower median), for tests of the selection kernel. */
int ipdm_slice_median(const float *d_x, float *d_med, int32_t B, int64_t n_per_slice, void *d_ws,
                      size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------ UNet denoiser ----------- */
/* UNetModel.__init__ arguments (Model/model.py:191-203). */
typedef struct ipdm_unet_cfg {
    int32_t in_channels, model_channels, out_channels, num_res_blocks, num_heads;
    int32_t n_mult, n_attn;
    double channel_mult[16];
    int32_t attention_resolutions[16];
} ipdm_unet_cfg;
typedef struct ipdm_unet ipdm_unet;

/* parameter inventory in the reference's state_dict key layout (Utils/loggerx.py:62-80 checkpoints):
 * name e.g. "down_blocks.1.0.conv1.2.weight"; shape padded with 1s to 4 dims. */
int ipdm_unet_param_count(const ipdm_unet_cfg *cfg);
int ipdm_unet_param_info(const ipdm_unet_cfg *cfg, int32_t idx, char *name, int32_t name_cap,
                         int32_t shape[4], int32_t *ndim);
/* replaces UNetModel.__init__ + load_state_dict: `weights[i]` is a HOST pointer to parameter i
 * (float32, contiguous, reference layout); the library repacks them into its own device layout. */
int ipdm_unet_create(const ipdm_unet_cfg *cfg, const float *const *weights, int32_t n_weights,
                     ipdm_unet **out);
int ipdm_unet_destroy(ipdm_unet *net);
size_t ipdm_unet_workspace_bytes(ipdm_unet *net, int32_t B, int32_t H, int32_t W);
/* replaces UNetModel.forward (Model/model.py:283-310) for one integer timestep shared by the batch
 * (the sampler's torch.full((1,), i), :564).  d_x [B,in_ch,H,W] -> d_eps [B,out_ch,H,W]. */
int ipdm_unet_forward(ipdm_unet *net, const float *d_x, int32_t t, float *d_eps, int32_t B, int32_t H,
                      int32_t W, void *d_ws, size_t ws_bytes, void *stream);

/* ipdm_unet_forward replayed from a captured hipGraph: one executable graph per (t, B, H, W, d_x, d_eps, d_ws), built on
 * the SECOND call with a key (the first runs eagerly).  Callers that want replays keep their input / output / workspace
 * buffers fixed (the host mirror copies into static buffers).  Same arithmetic, same results; `stream` must not be the
 * legacy default stream.  Reference call shape: model(x, t) once per reverse step, Utils/train_test_utils.py:290-294,
 * Model/model.py:496. */
int ipdm_unet_forward_graph(ipdm_unet *net, const float *d_x, int32_t t, float *d_eps, int32_t B, int32_t H,
                            int32_t W, void *d_ws, size_t ws_bytes, void *stream);

/* op-level entry points (parity tests of the individual kernels against torch-CPU ops) */
/* F.conv2d(cat(x1,x2) [upsampled to H,W by nearest], w, b, stride, padding=k/2) with optional fused
 * GroupNorm(+SiLU) prologue over the concatenated input and optional residual add.
 *   d_x1 [B,C1,Hs,Ws], d_x2 [B,C2,Hs,Ws] or NULL; source size (Hs,Ws) != (H,W) => nearest upsample
 *   (Model/model.py:168); w_host [Cout,C1+C2,k,k] HOST pointer in reference layout; act: 0 none,
 *   1 GN only, 2 GN+SiLU; d_res [B,Cout,Ho,Wo] or NULL. */
int ipdm_op_conv2d(const float *d_x1, int32_t C1, const float *d_x2, int32_t C2, int32_t B, int32_t Hs,
                   int32_t Ws, int32_t H, int32_t W, const float *w_host, const float *b_host, int32_t Cout,
                   int32_t ksize, int32_t stride, int32_t act, int32_t groups, const float *gamma_host,
                   const float *beta_host, const float *d_res, float *d_out, void *stream);
/* conv A (+bias, +residual) -> GroupNorm(+SiLU) -> conv B (3x3): the GN -> SiLU -> conv chain of ResidualBlock /
 * AttentionBlock.norm (Model/model.py:82-130,142-147) with the GroupNorm statistics taken from the per-tile partial
 * sums conv A's kernel leaves behind (no pass over the activations); *fused_rows receives the number of partial-sum
 * rows per sample that kernel wrote (0: that kernel family has no fused statistics and the activations were read).
 *   d_x [B,C,H,W]; wA_host [CA,C,ksA,ksA], wB_host [CB,CA,3,3] HOST, reference layout; d_resA [B,CA,Hm,Wm] or NULL;
 *   d_mid [B,CA,Hm,Wm] (conv A's output), d_out [B,CB,Hm,Wm]; act: 1 GN, 2 GN+SiLU. */
int ipdm_op_conv_gn_conv(const float *d_x, int32_t C, int32_t B, int32_t H, int32_t W, const float *wA_host,
                         const float *bA_host, int32_t CA, int32_t ksA, int32_t strideA, const float *d_resA,
                         int32_t groups, const float *gamma_host, const float *beta_host, int32_t act,
                         const float *wB_host, const float *bB_host, int32_t CB, float *d_mid, float *d_out,
                         int32_t *fused_rows, void *stream);
/* Test entry for the Upsample layer (Model/model.py Upsample: F.interpolate(scale 2, "nearest") + 3x3 conv) in the form the
 * executor runs it: conv A over the 2x up-sampled d_x [B,C,Hs,Ws] (on wide layers as four 2x2-tap parity convolutions over
 * the source grid with pre-added weights, output stored parity-planar: *used_up2 = 1, or 3 when those four convolutions run
 * in the Winograd F(2x2,2x2) domain (conv_wup2: whole 128-cout tiles, 16-channel chunks); on narrow layers the parity form
 * inside the direct kernel, NCHW output: 2; else the 3x3 form over nearest addressing: 0), then
 * GroupNorm(+SiLU) over cat(mid, d_skip) and conv B (ksB = 1 or 3) reading mid as stored.
 *   wA_host [CA,C,3,3], wB_host [CB,CA+C2,ksB,ksB], gamma/beta [CA+C2] HOST; d_skip [B,C2,2Hs,2Ws] or NULL (C2 = 0);
 *   d_mid [B,CA,2Hs,2Ws] (conv A's output as NCHW), d_out [B,CB,2Hs,2Ws]; act: 1 GN, 2 GN+SiLU. */
int ipdm_op_up_conv_chain(const float *d_x, int32_t C, int32_t B, int32_t Hs, int32_t Ws, const float *wA_host,
                          const float *bA_host, int32_t CA, const float *d_skip, int32_t C2, int32_t groups,
                          const float *gamma_host, const float *beta_host, int32_t act, const float *wB_host,
                          const float *bB_host, int32_t CB, int32_t ksB, float *d_mid, float *d_out,
                          int32_t *used_up2, void *stream);
/* AttentionBlock core (Model/model.py:148-153): d_qkv [B, heads*3*d, T] (per-head (q,k,v) chunks)
 * -> d_out [B, heads*d, T]. */
int ipdm_op_attention(const float *d_qkv, float *d_out, int32_t B, int32_t heads, int32_t d, int32_t T,
                      void *stream);

/* ------------------------------------------------------------------ ART convertor ------------ */
/* SART over a triangle-area lookup table + its forward projector: convertor="ART" and self.projection of the
 * reference (Utils/train_test_utils.py:225-233) = Recon/TASART2DNSL0 recons_torch / proj_torch
 * (TASART2DNSL0_PyAPI.cpp:33-80 -> TASART2DNSL0.cu DoReconstruction :721-975 / DoProjection :1335-1438).
 * SURVEY section 8(f) rank 3.  Geometry = the `Parameters` struct (TASART2DNSL0.h:23-42). */
typedef struct ipdm_art_geom {
    float dso, dsd;
    int32_t nx, ny;
    float dx, dy, offset_x, offset_y;
    int32_t nr;
    float dr, offset_r, angle_start;
    int32_t na, ta_dimx, ta_dimy;      /* na = number of view angles in `betas` */
    float ta_deltax, ta_deltay;
} ipdm_art_geom;
typedef struct ipdm_art_plan ipdm_art_plan;
/* lut_area_host: [ta_dimy][ta_dimx] f32 (Recon/Simens_alut.txt), betas_host: [na] view angles in degrees
 * (Recon/Simens_theta.txt) -- the two arrays recons_torch / proj_torch take.  Uploads them, precomputes the bin-edge
 * rays of every view (update_lines_kernel, .cu:270-302) and the per-view normalisation projection (_Fp_Ax(norm_proj,
 * footinfo, 1.0f), .cu:871).  Allocates and synchronises. */
int ipdm_art_plan_create(const ipdm_art_geom *geom, const float *lut_area_host, const float *betas_host,
                         ipdm_art_plan **out);
int ipdm_art_plan_destroy(ipdm_art_plan *plan);
size_t ipdm_art_workspace_bytes(const ipdm_art_plan *plan, int32_t B);
/* recons_torch(h_proj, lut_area, betas, nstart, ntv, sample_rate, permute): d_proj [B, na, nr] -> d_volume [B, ny, nx]
 * (NOT permuted: the caller applies permute(0,2,1) as a view, PyAPI.cpp:55-57).  sample_rate > 1 uses the first
 * na / sample_rate views and rows, as the reference does (PyAPI.cpp:37).  Every launch is on `stream` and all scalars
 * (dp, dg, alpha) stay on the device.  With one launch per view the call returns without synchronising; when a sweep
 * is ONE launch (the sweep's grid fits on the device, checked at plan creation) the call ends with one stream
 * synchronisation that reads whether a sweep's grid barrier expired (device shared with other work) -- such a
 * reconstruction is redone with one launch per view before the call returns, never handed back as success. */
int ipdm_art_reconstruct(ipdm_art_plan *plan, const float *d_proj, float *d_volume, int32_t B, int32_t nsart,
                         int32_t ntv, int32_t sample_rate, void *d_ws, size_t ws_bytes, void *stream);
/* proj_torch(h_volume, lut_area, betas): d_volume [B, ny, nx] -> d_proj [B, na, nr] */
int ipdm_art_project(ipdm_art_plan *plan, const float *d_volume, float *d_proj, int32_t B, void *d_ws,
                     size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------ measurement ------------- */
/* Per-launch HIP-event timing of the hot kernels on their launch stream (bench.py roofline leg; no
 * reference counterpart -- the reference has no profiling).  Classes: 0 = conv 3x3
 * stride-1 wide tile in its direct form, 1 = other conv variants, 2 = attention, 3 = the Winograd-domain form of
 * class 0's layers (recorded with its EXECUTED flops, 16/36 of the 3x3 count), 4 = the narrow direct convolutions
 * (bandwidth-bound: `out_flops[4]` holds their algorithmic HBM BYTES), 5 = the 128-cout-tile Winograd kernel
 * (conv_wino2, the dominant kernel; class 3 keeps the 64-cout-tile one), 6 = the narrow direct convolutions that read a
 * wide tensor (>= 64 input channels: bound by the f32 vector ALU, recorded with their flops; class 4 keeps the
 * bandwidth-bound ones), 7 = the wide Upsample layers in the Winograd F(2x2,2x2) domain of their parity form (conv_wup2;
 * executed flops: 9 products per source pixel and channel pair).  ipdm_profile_end needs the stream
 * synchronised; outputs are arrays of `n_classes` >= IPDM_PROF_CLASSES entries (a shorter array is an error, not an
 * overflow). */
#define IPDM_PROF_CLASSES 8
int ipdm_profile_begin(int32_t max_launches);
/* ... recording only the classes whose bit is set in class_mask (an event pair costs the stream about a microsecond per
 * launch: bench.py times its headline with the dominant kernel's classes only and the rest on an extra, untimed pass) */
int ipdm_profile_begin_classes(int32_t max_launches, uint32_t class_mask);
int ipdm_profile_end(double *out_flops, double *out_ms, int64_t *out_launches, int32_t n_classes);
/* Diagnostic: one wave that samples the shader clock (s_memtime) and the 100 MHz reference (s_memrealtime) every period_us,
 * `samples` times, into d_out[2 i], d_out[2 i + 1] (device memory).  On a stream of its own it co-resides with the kernels
 * that fill the chip: the quotient of the differences is the clock the chip holds under them, with no profiler attached
 * (tools/clock_probe.py). */
int ipdm_clock_probe(uint64_t *d_out, int32_t samples, int32_t period_us, void *stream);

/* kernel micro-benchmarks (tuning aid; allocate, fill with random data, time `iters` launches) */
int ipdm_bench_conv2d(int32_t B, int32_t C1, int32_t C2, int32_t H, int32_t W, int32_t Cout, int32_t ksize,
                      int32_t stride, int32_t act, int32_t with_res, int32_t iters, float *avg_ms);
int ipdm_bench_attention(int32_t B, int32_t heads, int32_t d, int32_t T, int32_t iters, float *avg_ms);
/* Which kernel family a convolution of this shape is packed for NOW (the environment switches are read when
 * weights are packed): 0 = plain layout (direct / legacy kernels), 2 | 4 = conv_ws cout-interleaved f32 MFMA.
 * Test aid: lets a parity test prove which path it ran. */
int32_t ipdm_conv_layout_code(int32_t Cout, int32_t ksize, int32_t stride);
/* Which KERNEL a plain convolution (one source, no resampling on the way in, K-split workspace available) of this shape
 * and batch takes NOW -- the dispatch of the executor's conv2d_launch, options included:
 *   1 = conv_wino (Winograd F(2x2,3x3), 64-cout tiles)      2 = conv_wino2 (Winograd, 128-cout tiles: the dominant kernel)
 *   3 = conv_ws (direct implicit GEMM, f32 MFMA)            4 = conv_ws with a K split + combine pass
 *   5 = conv_direct (narrow layers, packed-f32 VALU)        6 = conv_nm (opt-in 16-cout MFMA)
 *   7 = parity form of an Upsample (never for this plain shape)   8 = conv_igemm (the generic 4-wave kernel)
 *   9 = conv_wino2 with K slices + combine pass (the layers with too few tiles per sample)
 *   10 = conv_pw (wide 1x1 layers: the barrier-free pointwise kernel)
 *   11 = conv_wup2 (an Upsample's parity form in the Winograd F(2x2,2x2) domain; never for this plain shape)
 *   12 = conv_wino3 (under the opt-in option conv_bf16x3 the layers of conv_wino2's SHAPE, whatever the batch: products on the bf16 matrix pipe, 3-way split)
 *   -1 = bad argument.
 * Test aid (replaces nothing in the reference): a parity test asserts the kernel it believes it covers. */
int32_t ipdm_conv_kernel_code(int32_t B, int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, int32_t H, int32_t W);
/* ... the same for a layer whose output feeds a GroupNorm (the executor asks such a layer for fused statistics, and the
 * kernel rule of a statistics-producing layer looks at the layer alone, never at the batch): e.g. a wide 1x1 layer is 10
 * (conv_pw) here only if ONE sample brings >= 1024 items, whatever B. */
int32_t ipdm_conv_kernel_code_stats(int32_t B, int32_t Cout, int32_t Cin, int32_t ksize, int32_t stride, int32_t H, int32_t W);
/* Which attention kernel a launch with head dim d takes NOW: 0 = 4-wave kernel (d = 32, or IPDM_ATTN_LEGACY),
 * 1 = wave-specialised exact-f32 MFMA (the default for d = 64). */
int32_t ipdm_attention_kernel_code(int32_t d);

#ifdef __cplusplus
}
#endif
#endif /* IPDM_HIP_H */
