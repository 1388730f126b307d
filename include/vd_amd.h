/* vd_amd.h -- C ABI of the MI355X-native video-diffusion denoise engine (libvdamd.so).
 *
 * The reference (cliangyu/video-diffusion) is pure Python and has no FFI; its seam for this hot path
 * is a set of Python call signatures.  Each entry point below names the reference
 * interface it replaces (paths relative to /root/reference).  All `const float*`/`float*` tensor
 * arguments are DEVICE pointers owned by the caller unless the name says `host`; `stream` is a
 * hipStream_t passed as void*.  Every function returns 0 on success or a negative code
 * (-1 bad argument / unsupported configuration, -2 HIP runtime error); text via vd_last_error().
 * No entry point synchronises the stream except vd_load_weight (a blocking H2D copy) and vd_device_errors.
 * Contract: ONE device per process (the reference launches one process per GPU, command_launchers.py:32-62) and one
 * stream at a time per engine: kernel attributes are cached process-wide and an engine owns a single workspace.
 * vd_set_weight_storage binds the process to the current device and refuses a second one.
 */
#ifndef VD_AMD_H
#define VD_AMD_H
#ifdef __cplusplus
extern "C" {
#endif

/* Keys of video_model_and_diffusion_defaults() that shape the network
 * (improved_diffusion/script_util.py:15-57, create_video_model :229-300). */
typedef struct vd_config {
    int image_size;              /* 32 / 64 / 128 / 256 -> channel_mult table, script_util.py:255-264 */
    int num_channels;
    int num_res_blocks;
    int num_heads;
    int T;                       /* model's max_frames: only used by the frame-embedding period (unet.py:921) */
    int n_attention_ds;          /* attention_resolutions converted to downsample rates, script_util.py:266-268 */
    int attention_ds[8];
    int use_scale_shift_norm;
    int use_spatial_encoding;
    int use_frame_encoding;
    int enforce_position_invariance;
    int use_rpe_net;
    int allow_interactions_between_padding;
    float rp_alpha, rp_beta, rp_gamma;   /* bucket parameters of the table RPE (unet.py:330-340) */
    int time_embed_mult;         /* 4: time_embed_dim = 4*num_channels (unet.py:605) */
    int cond_emb_type;           /* CondMargVideoModel (unet.py:929-1020): 0 'channel' (5 input channels: frames + obs / kinda-marg
                                    indicators), 1 'duplicate' | 'all' (6: noisy frames | x0 * obs_mask), 2 't=0' (3: x itself,
                                    observed frames get timestep -1); the '-initzero' spellings differ at initialisation only */
    int learn_sigma;             /* 1: the network has 6 output channels, eps | variance values (script_util.py:129-131).  Boundary A
                                    only: the reference's own sampler cannot use them on video tensors -- p_mean_variance
                                    asserts model_output.shape == (B, 2*T, ...) (gaussian_diffusion.py:283, C = x.shape[1] = T) --
                                    so vd_p_sample & co. refuse such an engine as the reference does */
} vd_config;

typedef struct vd_engine vd_engine;

const char* vd_last_error(void);
const char* vd_version(void);
/* 16 hex digits: SHA-1 over the compiler flags and every source the library was built from (_lib.source_sha() recomputes it from
 * the sources on disk: a loaded library that does not match them is stale, whatever the file times say). */
const char* vd_source_sha(void);

/* CondMargVideoModel(...) constructor via create_video_model (script_util.py:229-300; unet.py:929-947).
 * Host-only: builds the topology and the parameter table; touches no GPU. */
int vd_create(const vd_config* cfg, vd_engine** out);
void vd_destroy(vd_engine* e);

/* model.state_dict() keys/shapes, in the reference's order (checkpoint format, train_util.py:570-574). */
int vd_param_count(vd_engine* e);
int vd_param_info(vd_engine* e, int index, char* name, int name_cap, int* ndim, long long shape[4]);

/* model.load_state_dict(sd) (scripts/video_sample.py:565).  Weights live in ONE packed device buffer in kernel-ready
 * layouts chosen at load time (split arithmetic: 3x3 stride-1 convs as the Winograd image U = G g G^T in three 16-bit
 * planes [I/16][16][O/32][3][64][8]; linear / 1x1 / stem / stride-2 convs as split MFMA fragments
 * [K/16][N/32][3][64][8], each image followed by its per-output scales; spatial_encoding -> [HW][C]) so that a
 * single RCCL broadcast replaces dist_util.sync_params' per-tensor broadcasts (dist_util.py:139-143).  The layout depends on
 * VD_MATH: vd_weights_layout_id() identifies it, and ranks compare it before accepting a broadcast buffer.
 * The buffer is caller-owned (e.g. a torch tensor) and must outlive the engine. */
long long vd_weights_bytes(vd_engine* e);
int vd_set_weight_storage(vd_engine* e, void* dev_buffer, long long bytes);
/* The same packed image assembled in HOST memory (no GPU needed): pack once, then ship it -- a broadcast, a file, one
 * H2D copy into a buffer later given to vd_set_weight_storage followed by vd_mark_weights_loaded.  The compute entry
 * points refuse an engine whose storage is still on the host. */
int vd_set_weight_storage_host(vd_engine* e, void* host_buffer, long long bytes);
int vd_load_weight(vd_engine* e, const char* name, const float* host_data, long long numel);
int vd_weights_missing(vd_engine* e);          /* number of parameters not loaded yet */
int vd_mark_weights_loaded(vd_engine* e);      /* after receiving the packed buffer by broadcast */
unsigned long long vd_weights_layout_id(vd_engine* e);   /* hash of (parameter table, kinds, offsets, arithmetic mode) */

/* Channels / resolution of the tensor the positional encodings are added to (unet.py:669-675,914-926). */
int vd_pos_channels(vd_engine* e);
int vd_pos_resolution(vd_engine* e);

/* Frequency tables of timestep_embedding / frame_embedding (nn.py:89-122), built by the host with
 * the reference's own float32 expression so the angles agree bit for bit. */
int vd_set_freqs(vd_engine* e, const float* host_time_freqs, int n_time, const float* host_frame_freqs, int n_frame);

/* SpacedDiffusion tables (respace.py:68-82, gaussian_diffusion.py:123-172,299-317).
 * host_tab: VD_NTAB rows of num_timesteps float32 (float64 tables cast like _extract_into_tensor,
 * gaussian_diffusion.py:1019-1031), row order VD_TAB_*.  timestep_map + rescale = _WrappedModel
 * (respace.py:111-119): t_model = map[t] * rescale (rescale = 1000/original_steps, or 1 with rescale off). */
enum { VD_TAB_SQRT_RECIP = 0, VD_TAB_SQRT_RECIPM1, VD_TAB_COEF1, VD_TAB_COEF2, VD_TAB_LOGVAR /* model log variance */,
       VD_TAB_ACP, VD_TAB_ACP_PREV, VD_TAB_SQRT_ACP, VD_TAB_SQRT_1M_ACP,
       VD_TAB_POST_LOGVAR /* posterior_log_variance_clipped */, VD_TAB_LOG_1M_ACP /* log_one_minus_alphas_cumprod */,
       VD_TAB_ALPHA /* alphas = 1 - betas (the guidance weight, gaussian_diffusion.py:363) */, VD_NTAB };
int vd_set_schedule(vd_engine* e, int num_timesteps, const float* host_tab, const int* host_timestep_map,
                    float rescale);
/* What the network's output is (ModelMeanType, gaussian_diffusion.py:29-36,326-341): 0 EPSILON (default); 1 START_X
 * (predict_xstart=True, script_util.py:429-431): pred_xstart = clamp(model_output), DDIM derives eps from it. */
int vd_set_model_mean_type(vd_engine* e, int type);


/* Timestep indices outside [0, num_timesteps) make the reference raise IndexError (_extract_into_tensor,
 * gaussian_diffusion.py:1019-1031).  The step entry points stay asynchronous: such a batch element is written as NaN
 * and a sticky device flag is set; this call waits for the whole DEVICE (steps issued on any stream, non-blocking side streams included, have finished),
 * copies the flags to the host, clears them, and the host
 * mirror raises IndexError.  bit 0: timestep index out of range.
 * bit 1: the network output a step consumed was not finite.  The reference would carry the NaN into its sample; here the clamp
 * of clip_denoised would turn it into a plausible -1, so the posterior kernels keep such an element NaN and set this bit
 * (vd_p_sample, vd_ddim_sample, vd_p_mean_variance, vd_posterior_update, vd_posterior_from_xstart, vd_vb_terms, vd_guided_step,
 * the window executor's captured step).  In the default f16x3 arithmetic (two fp16 pieces per fp32 operand) this is how an
 * operand beyond the split's range (|x| >= 2^15 = 32768: the scaled remainder (x - a0) 2^12 then leaves fp16; for a 3x3 conv the operand
 * is the Winograd-domain value, a signed sum of four inputs, so |input| < 2^13 is safe) anywhere in the network shows: VD_MATH=bf16x6
 * carries the full fp32 range.
 * The host mirror raises FloatingPointError. */
int vd_device_errors(vd_engine* e, int* flags);

/* Bytes of engine-owned workspace a (B, T) window needs; allocated lazily by the first call. */
int vd_workspace_bytes(vd_engine* e, int B, int T, long long* bytes);

/* observed_frames: 0 'x_0', 1 'x_t', 2 'x_t_minus_1' (unet.py:958-974,991-1013). */

/* Boundary A: model(x, timesteps, **model_kwargs) -> eps   (unet.py:949-1026 after respace.py:111-119).
 * x, obs_src, eps: [B][T][3][H][W] fp32;  obs/lat/km masks: [B*T] fp32;  frame_indices: [B*T] int64;
 * t_model: [B] fp32 value handed to the network. */
int vd_unet_forward(vd_engine* e, int B, int T, const float* x, const float* obs_src, const float* obs_mask,
                    const float* latent_mask, const float* kinda_marg_mask, const long long* frame_indices,
                    const float* t_model, int observed_frames, float* eps, void* stream);

/* diffusion.p_sample(model, x, t, clip_denoised, model_kwargs) -> {'sample','pred_xstart'}
 * (gaussian_diffusion.py:403-448 through SpacedDiffusion.p_mean_variance, respace.py:84-86).
 * t: [B] int64 respaced indices (device).  noise: explicit N(0,1) draws or NULL -> in-kernel
 * Philox4x32-10(seed, offset).  x is not modified; sample/pred_xstart/eps are outputs (the last two may be NULL). */
int vd_p_sample(vd_engine* e, int B, int T, const float* x, const float* obs_src, const float* obs_mask,
                const float* latent_mask, const float* kinda_marg_mask, const long long* frame_indices,
                const long long* t, int observed_frames, int clip_denoised, const float* noise,
                unsigned long long seed, unsigned long long offset, float* sample, float* pred_xstart, float* eps,
                void* stream);

/* diffusion.ddim_sample(..., eta) (gaussian_diffusion.py:597-634). */
int vd_ddim_sample(vd_engine* e, int B, int T, const float* x, const float* obs_src, const float* obs_mask,
                   const float* latent_mask, const float* kinda_marg_mask, const long long* frame_indices,
                   const long long* t, int observed_frames, int clip_denoised, float eta, const float* noise,
                   unsigned long long seed, unsigned long long offset, float* sample, float* pred_xstart, float* eps,
                   void* stream);

/* diffusion.p_mean_variance(model, x, t, clip_denoised, model_kwargs) (gaussian_diffusion.py:229-372): one UNet forward,
 * then 'pred_xstart' (clipped) and 'mean' = posterior mean of it; 'variance' / 'log_variance' are the schedule rows
 * VD_TAB_LOGVAR at t (the host mirror broadcasts them).  Any of mean / pred_xstart / eps may be NULL. */
int vd_p_mean_variance(vd_engine* e, int B, int T, const float* x, const float* obs_src, const float* obs_mask,
                       const float* latent_mask, const float* kinda_marg_mask, const long long* frame_indices,
                       const long long* t, int observed_frames, int clip_denoised, float* mean, float* pred_xstart,
                       float* eps, void* stream);

/* The NLL path of scripts/video_nll.py.  vd_vb_terms = GaussianDiffusion._vb_terms_bpd (gaussian_diffusion.py:750-790;
 * losses.py normal_kl / discretized_gaussian_log_likelihood) plus the two per-step MSEs of calc_bpd_loop_subsampled
 * (:975-990), given the eps of a forward pass at x_t: vb[B] = KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) in bits per dim,
 * or the discretised decoder NLL where t == 0; xstart_mse[B] = mean((pred_xstart - x_start)^2); mse[B] =
 * mean((eps_from_xstart - noise)^2) (needs `noise`).  latent_mask: [B*T] or NULL; as in the reference's
 * mean_flat(tensor, mask) the mask multiplies and the mean still runs over all elements.  vd_prior_bpd = _prior_bpd (:909-926).
 * With vd_set_model_mean_type(1) (predict_xstart=True) `eps` is the network output as vd_p_mean_variance returns it, i.e. the x_0
 * prediction itself: pred_xstart = clamp(eps) and the eps of the second MSE is derived from it (_predict_eps_from_xstart, :392-396). */
int vd_vb_terms(vd_engine* e, int B, int T, const float* x_start, const float* x_t, const float* eps, const float* noise,
                const long long* t, int clip_denoised, const float* latent_mask, float* vb, float* xstart_mse, float* mse,
                float* pred_xstart, void* stream);
int vd_prior_bpd(vd_engine* e, int B, int T, const float* x_start, const float* latent_mask, float* out, void* stream);

/* Window executor -- the loop of scripts/video_sample.py:149-168 (`for timestep in reversed(range(num_timesteps)):
 * local = diffusion.p_sample(model, local, t, ...)['sample']`) with the loop state on the device: the respaced index
 * t[B] and the Philox {seed, offset} live in engine-owned device memory, ONE hipGraph holds a whole step (t -> t_model,
 * UNet forward, posterior update of x IN PLACE, t -= 1, offset += B*T*3*H*W) and is captured once per window signature
 * (B, T, the six tensor addresses, observed_frames, sampler, clip, eta), so a window costs num_timesteps graph launches
 * instead of ~330 kernel launches per step from the host (BASELINE configs[4]: two signatures, Tw = 20 and Tw = 14).
 * vd_window_begin arms the counters (and captures if the signature is new: one eager forward, then the capture; `stream`
 * must not be the default stream); vd_window_run replays n_steps steps t_start, t_start-1, ...; x holds the result.
 * observed_frames: 0 x_0, 1 x_t, 2 x_t_minus_1 as p_sample_loop runs it (obs_src = the CLEAN frames, re-noised to t - 1 inside
 * every step from the second half of the step's Philox range, gaussian_diffusion.py:565-568), 3 x_t_minus_1 with obs_src read as it
 * is at every step (a direct p_sample caller: scripts/video_sample.py:149-166 hands x0).  Noise is always the in-kernel Philox stream (seed, offset + step*B*per + i):
 * identical to vd_p_sample(noise = NULL, seed, offset + step*B*per). */
int vd_window_begin(vd_engine* e, int B, int T, float* x, const float* obs_src, const float* obs_mask,
                    const float* latent_mask, const float* kinda_marg_mask, const long long* frame_indices,
                    int observed_frames, int sampler, int clip_denoised, float eta, unsigned long long seed,
                    unsigned long long offset, long long t_start, void* stream);
int vd_window_run(vd_engine* e, int n_steps, void* stream);
/* Bumped by every vd_window_begin.  The step counters are one set per engine: a host object that armed a window keeps the
 * value it saw and refuses to run when another begin has happened since (executor.py).  vd_window_run itself fails with
 * "window graphs invalidated" when the armed window's graph was dropped (workspace growth, vd_set_schedule). */
unsigned long long vd_window_generation(vd_engine* e);
int vd_window_graphs(vd_engine* e);            /* captured graphs held by the engine */

/* Window prefix cache (opt-in, off by default; a host-side choice like the executor itself).  The reference recomputes the
 * whole UNet for every frame at every step (unet.py:838-846).  Before the first attention layer the network treats frames as
 * independent batch entries (input_blocks 0 .. first attention block - 1: ResBlocks / Downsample, per-frame GroupNorm), and in
 * 'x_0' mode with the default cond_emb_type='channel' an OBSERVED frame's input there never changes during a window: its x0
 * pixels, the indicator channels and the timestep-0 embedding (unet.py:991-1013).  With the cache on, vd_window_begin runs
 * those blocks for the observed frames (obs_mask = 1, latent_mask = 0) once, into persistent full-size tensors, and the
 * captured step runs them on the remaining frames only (a compact batch), scatters each block output and its GroupNorm
 * partial sums beside the cached rows and continues with the full batch (attention, decoder skips).  Same arithmetic per
 * frame; the GroupNorm partial sums of a tensor are folded in a different (fp64) grouping.  The set of cached frames is part
 * of the graph signature.  vd_window_prefix_frames: how many frames of the armed window are served from the cache. */
int vd_set_window_prefix_cache(vd_engine* e, int on);
int vd_window_prefix_frames(vd_engine* e);

/* Window suffix skip (opt-in, off by default; the sibling of the prefix cache at the other end of the network).  Behind its
 * LAST attention layer the UNet treats frames as independent batch entries again -- the remaining decoder ResBlocks, the
 * Upsample convs and the output head (unet.py:820-839) -- and the caller of a window keeps only its latent frames
 * (scripts/video_sample.py:170-186), while a purely observed frame (obs_mask = 1, latent_mask = 0) re-enters the network as
 * the observation whatever the step wrote to it (unet.py:958-983: observed_frames 'x_0' and 'x_t_minus_1' with
 * cond_emb_type='channel'; NOT 'x_t', where its input is its own running sample).  With the skip on, the captured step
 * gathers every other frame out of the last attention layer's output and out of the skip tensors the remaining blocks read
 * (rows and GroupNorm partial sums), runs those blocks and the head on that compact batch with the kernel variants the full
 * batch would get, and writes eps = 0 for the skipped frames: every frame that is not a pure observation receives the eps
 * -- and so the sample -- of the full step, bit for bit; the skipped frames' entries of the window tensor are meaningless.
 * vd_window_suffix_frames: frames the suffix of the armed window runs on (0: all of them -- skip off or nothing to skip). */
int vd_set_window_suffix_skip(vd_engine* e, int on);
int vd_window_suffix_frames(vd_engine* e);

/* The posterior arithmetic alone, given eps (same formulas; mode 0 p_sample, 1 ddim). */
int vd_posterior_update(vd_engine* e, int mode, int B, long long per_sample, const float* x, const float* eps,
                        const long long* t, int clip_denoised, float eta, const float* noise,
                        unsigned long long seed, unsigned long long offset, float* sample, float* pred_xstart,
                        void* stream);

/* process_xstart with a caller-supplied `denoised_fn` (gaussian_diffusion.py:319-324): the host takes the unclipped
 * pred_xstart (vd_p_mean_variance, clip_denoised = 0), applies its function, and hands the result back here; the clamp,
 * q_posterior_mean_variance (:208-227) and the noise add (:438-443; ddim: eps re-derived from x_0, :597-634) run as in
 * vd_posterior_update.  Any of sample / pred_xstart / mean may be NULL. */
int vd_posterior_from_xstart(vd_engine* e, int mode, int B, long long per_sample, const float* x, const float* xstart_in,
                             const long long* t, int clip_denoised, float eta, const float* noise,
                             unsigned long long seed, unsigned long long offset, float* sample, float* pred_xstart,
                             float* mean, void* stream);

/* return_attn_weights (unet.py:457-466,799-836; gaussian_diffusion.py:277,496): per attention block the softmax weights
 * averaged over the heads, absolute value -- temporal (B*HW, T, T) and spatial (B*T, HW, HW) -- in execution order (input
 * blocks, middle, output blocks).  vd_attn_blocks / vd_attn_block_info size the buffers; vd_set_attn_capture arms the
 * capture for the following forwards (n = 0 clears it).  Two extra passes over q, k per block: the logging path. */
int vd_attn_blocks(vd_engine* e);
int vd_attn_block_info(vd_engine* e, int i, int* resolution, int* channels);
int vd_set_attn_capture(vd_engine* e, float* const* temporal, float* const* spatial, int n);

/* use_gradient_method (gaussian_diffusion.py:264-271,350-364; scripts/video_sample.py:429 `--use_gradient_method`).
 * The guidance needs d(loss)/d(x_t) through the whole UNet: backward-DATA only, no weight gradients.  Its matrix products
 * run on the forward kernels over a second packed image -- transposed linear weights, 180-degree-rotated transposed 3x3
 * kernels -- that exists only when asked for: vd_bwd_weights_bytes -> vd_set_bwd_weight_storage (device memory, or host
 * memory for a broadcast image) -> vd_load_weight_bwd per checkpoint tensor (a no-op for tensors the backward never reads).
 * vd_guided_step is p_mean_variance(..., use_gradient_method=True) (+ p_sample's noise add when `sample` is given):
 *   the network sees obs_mask := 0 and latent_mask := obs_mask + latent_mask; with the unguided mean / variance a sample
 *   x_{t-1} = mean + [t != 0] sigma_t * noise is drawn, loss = sum(((x_{t-1} - x_t_minus_1) * obs_mask)^2) is differentiated
 *   w.r.t. x, and mean' = mean - 10 * alpha_t * grad / 2.  Outputs (each may be NULL): mean', pred_xstart, grad,
 *   sample = mean' + [t != 0] sigma_t * noise2. */
long long vd_bwd_weights_bytes(vd_engine* e);
int vd_set_bwd_weight_storage(vd_engine* e, void* buf, long long bytes, int on_host);
int vd_load_weight_bwd(vd_engine* e, const char* name, const float* host_data, long long numel);
int vd_guided_step(vd_engine* e, int B, int T, const float* x, const float* obs_mask, const float* latent_mask,
                   const float* kinda_marg_mask, const long long* frame_indices, const long long* t, int clip_denoised,
                   const float* x_t_minus_1, const float* noise, const float* noise2, float* mean, float* pred_xstart,
                   float* grad, float* sample, void* stream);

/* diffusion.q_sample(x_start, t, noise) (gaussian_diffusion.py:190-206). */
int vd_q_sample(vd_engine* e, int B, long long per_sample, const float* x_start, const long long* t,
                const float* noise, float* out, void* stream);

/* th.randn(*shape) replacement on the engine's own counter-based generator. */
int vd_randn(float* out, long long n, unsigned long long seed, unsigned long long offset, void* stream);

/* Per-kernel-class timing with HIP events recorded on the launch stream (bench.py's roofline leg; no
 * reference counterpart).  Between begin and end every engine launch is bracketed by two events;
 * vd_profile_end synchronises and writes, per class i, out[4i..4i+3] = {launches, total ms,
 * algorithmic FLOPs, algorithmic bytes}. */
int vd_profile_begin(void);
int vd_profile_end(double* out, int cap);
int vd_profile_classes(void);
const char* vd_profile_class_name(int i);

/* ---- arithmetic of the matrix products (environment VD_MATH, read once per process) -------------
 *   0  f16x3  (default) an fp32 operand x is carried as two fp16 pieces, x ~ a0 + 2^-12 a1, a0 = f16(x), a1 = f16((x - a0) 2^12):
 *             22 significand bits (relative error <= 2^-22 for 2^-14 <= |x| < 2^15, absolute <= 2^-37 below; |x| >= 2^15 = 32768 -- not
 *             fp16's 65504: the scaled remainder (x - a0) 2^12 reaches 65536 there -- gives NaN, never a silently wrong number; inputs of
 *             a 3x3 conv: |x| < 2^13, its operand is a signed sum of four of them); a product is three piece products a0 b0 + a0 b1 + a1 (2^-12 b0) on
 *             v_mfma_f32_32x32x16_f16 with fp32 accumulation; weights carry a per-output power-of-two scale (image trailer).
 *   1  bf16x6 the exact split: three bf16 pieces per operand, six piece products (the default of earlier releases).
 *   2  fp32   every product on v_mfma_f32_32x32x2_f32.
 * The packed weight image depends on it (vd_weights_layout_id). */
int vd_math_mode(void);
/* uint16 count of a split weight image with n_out outputs and k_total inputs per output, trailer included
 * (= 3 n_out k_total + 4 n_out): [k/16][n_out/32][piece 3][lane 64][8 x 16 bit], then n_out float scales, n_out reciprocals. */
long long vd_split_image_u16(long long n_out, long long k_total);

/* ---- single-operator entry points (parity tests call the kernels through these) -------------- */
/* NHWC conv / linear on fp32 MFMA.  src1/C0: virtual channel concat; affA/affB: folded GroupNorm(+FiLM);
 * act: SiLU on the operand; res: residual in the epilogue; fbias: per-frame bias [nfr][fbias_ld]. */
/* w_packed: [tap][Cout][Cin] (generic kernel; may be NULL when w_frag / w_wino covers the shape);
 * w_frag: MFMA-fragment-major weights of a linear layer / 1x1 conv from vd_pack_linear_frag, or NULL;
 * w_wino: Winograd-transformed 3x3 weights from vd_pack_conv3_wino, or NULL (preferred when given and supported:
 *         one plain source tensor -- no concat, no affine/act prologue -- stride 1, square power-of-two >= 8x8,
 *         Cout % 64 == 0, Cin % 32 == 0). */
int vd_op_conv(const float* src0, const float* src1, int C0, int Cin, int nfr, int Hs, int Ws, int ups, int stride,
               int pad, int ksz, const float* w_packed, const float* w_frag, const float* w_wino, const float* bias,
               const float* affA, const float* affB, int act, const float* res, const float* fbias, int fbias_ld,
               float* out, int Cout, void* stream);
/* vd_op_conv whose epilogue also writes the GroupNorm partial sums of its OUTPUT (Winograd shapes only, -1 otherwise):
 * gn_part[nfr][split][Cout][2] doubles = per (frame, block of the frame, channel) [sum, sum of squares], with
 * split = vd_conv_stats_split(output height).  vd_op_gn_affine folds one or two such tables (the halves of a channel
 * concat; tables from vd_op_conv_stats or from a statistics pass have the same layout) into the (A, B) pair of
 * vd_op_gn_fold, so the consumer's GroupNorm (unet.py:185-198) never reads the tensor for its statistics. */
int vd_conv_stats_split(int Hout);
/* Output channels per block of the Winograd kernel a stride-1 3x3 conv of this shape runs on under the split arithmetic: 128
 * (conv_wino_z128.hip) or 64 (conv_wino_r64.hip).  Same weight image, same results up to summation order; tests use it to
 * know which kernel they exercised. */
int vd_conv_wino_block_couts(int nfr, int H, int Cin, int Cout);
int vd_op_conv_stats(const float* src0, int Cin, int nfr, int Hs, int Ws, int ups, const float* w_wino, const float* bias,
                     const float* res, const float* fbias, int fbias_ld, float* out, int Cout, double* gn_part,
                     void* stream);
int vd_op_gn_affine(const double* part0, int split0, int C0, const double* part1, int split1, int C, int nfr, int HW,
                    const float* gamma, const float* beta, const float* film, int film_ld, float* affA, float* affB,
                    void* stream);
/* Winograd F(2x2,3x3) image of a 3x3 weight for the fp32-MFMA kernel: U = G g G^T per (cout, cin) with row 2 negated (the
 * kernel negates row 2 of B^T as well), 16*O*I floats in [I/16][16][O/32][2][64][4]; pass as w_wino. */
int vd_pack_conv3_wino(const float* host_oihw, float* host_out, int O, int I);
/* nn.Linear / 1x1-conv weight [N][K] (N, K multiples of 32) -> [K/32][N/32][4][64][4] for the fp32-MFMA kernel; pass the
 * result as w_frag with ksz = 1. */
int vd_pack_linear_frag(const float* host_w, float* host_out, int N, int K);
/* Linear layer on the 16-bit matrix cores at fp32 accuracy (csrc/gemm_split.hip) in the process' arithmetic (f16x3 | bf16x6;
 * with VD_MATH=fp32 these entry points run bf16x6).  The weight is split on the host: [N][K] (N, K multiples of 32) ->
 * vd_split_image_u16(N, K) uint16.  out[m][n] = bias[n] + res[m][n] + sum_k f(a[m][k]) w[n][k], f = SiLU if act.  The engine's
 * kernel for every nn.Linear / 1x1 conv / the stem. */
int vd_pack_linear_split(const float* host_w, unsigned short* host_out, int N, int K);
/* 3x3 convolutions that Winograd does not cover (the stride-2 Downsample convs, unet.py:98) on the same arithmetic: the
 * split GEMM kernel walks an implicit im2col operand (k = tap*I + c; taps outside the image read 0).
 * Weights: OIHW -> the split image of the [O][9*I] matrix = vd_split_image_u16(O, 9*I) uint16.  stride 1 or 2, padding 1. */
int vd_pack_conv3_split(const float* host_oihw, unsigned short* host_out, int O, int I);
int vd_op_conv_split(const float* src0, int Cin, int nfr, int Hs, int Ws, int stride, const void* w_split, const float* bias,
                     const float* res, float* out, int Cout, void* stream);
/* The same arithmetic for the 3x3 stride-1 convs: Winograd F(2x2,3x3) whose element products run as piece products of the
 * split fp32 operands, 64 couts per block, input transform and split in the MFMA fragment layout, in registers
 * (csrc/conv_wino_r64.hip); maps >= 8x8.  Weights: OIHW -> U = G g G^T (fp64, row 3 negated) split into
 * [I/16][16][O/32][3][64][8] + trailer = vd_split_image_u16(O, 16*I) uint16.  One plain source tensor, stride 1, square
 * power-of-two maps, O % 64 == 0, I % 32 == 0; gn_part as vd_op_conv_stats (or NULL).  Big windows are cut along frames. */
int vd_pack_conv3_wino_split(const float* host_oihw, unsigned short* host_out, int O, int I);
int vd_op_conv_wino_split(const float* src0, int Cin, int nfr, int Hs, int Ws, int ups, const void* w_split, const float* bias,
                          const float* res, const float* fbias, int fbias_ld, float* out, int Cout, double* gn_part,
                          void* stream);
/* The ResBlock's `GroupNorm -> SiLU -> conv3x3` (unet.py:138-141,185-198) with the normalisation's folded affine and the SiLU applied
 * INSIDE the convolution kernel: out = conv3x3(silu(x * affA[frame][c] + affB[frame][c])) + bias (+ res), zero padding applied to the
 * ACTIVATED tensor as in the reference.  csrc/conv_wino_z128.hip stages the patch through registers and activates it there, so the
 * activation image (one write + one read of the tensor) does not exist.  The input may be the virtual channel concat of two tensors
 * (src1 != NULL: C0 channels from src0, Cin - C0 from src1: th.cat([h, hs.pop()], 1), unet.py:826-828).  Shapes: vd_conv_wino_act_ok(nfr, H, Cin, Cout) -- the f16x3
 * arithmetic, maps >= 16 x 16, Cout in {128, 256}, Cin <= 320, a grid that fills the chip. */
int vd_op_conv_wino_act(const float* src0, const float* src1, int C0, int Cin, int nfr, int Hs, int Ws, const void* w_split, const float* bias,
                        const float* affA, const float* affB, const float* res, float* out, int Cout, double* gn_part, void* stream);
int vd_conv_wino_act_ok(int nfr, int H, int Cin, int Cout);
/* Upsample (nearest x2, unet.py:70-77) + conv3x3 in its sub-pixel form on csrc/conv_wino_r64.hip: output pixel (2y + a, 2x + b) sees
 * only a 2 x 2 neighbourhood of the SOURCE map, i.e. four 3x3 "phase" kernels with one zero row and one zero column each
 * ((w0, w1 + w2, 0) for a = 0, (0, w0 + w1, w2) for a = 1; sums in fp64), convolved with the low-resolution map; in the Winograd
 * domain one of the four columns of every such kernel is zero and is skipped (a quarter of the matrix work of F(2x2,3x3) on
 * the upsampled map).  Weights: OIHW -> 4*O phase kernels -> the vd_pack_conv3_wino_split layout = vd_split_image_u16(4*O, 16*I) uint16.  src0
 * [nfr][Hs][Hs][Cin], out [nfr][2Hs][2Hs][Cout]; gn_part [nfr][vd_conv_ups_stats_split(Hs)][Cout][2] doubles or NULL. */
int vd_pack_conv3_wino_ups(const float* host_oihw, unsigned short* host_out, int O, int I);
int vd_conv_ups_stats_split(int Hs);
int vd_op_conv_wino_ups(const float* src0, int Cin, int nfr, int Hs, const void* w_ups, const float* bias, float* out, int Cout,
                        double* gn_part, void* stream);
int vd_op_linear_split(const float* a, int M, int K, const void* w_split, const float* bias, const float* res, int act,
                       float* out, int N, void* stream);
/* The same layer with the GroupNorm partial sums of its output from the epilogue (the rows are the HW pixels of M / HW
 * consecutive frames; unet.py:537-538 followed by the next block's normalization, nn.py:15-17): gn_part is
 * [M / HW][vd_linear_stats_split(M, N, HW)][N][2] doubles, per channel [sum, sum of squares]. */
int vd_op_linear_split_stats(const float* a, int M, int K, const void* w_split, const float* bias, const float* res, int act,
                             float* out, int N, int HW, double* gn_part, void* stream);
int vd_linear_stats_split(int M, int N, int HW);
/* GroupNorm32 statistics folded to y = x*A + B per (frame, channel); film ([nfr][2C] scale|shift) optional. */
int vd_op_gn_fold(const float* src0, const float* src1, int C0, int C, int nfr, int HW, const float* gamma,
                  const float* beta, const float* film, int film_ld, float* affA, float* affB, void* stream);
int vd_op_affine_apply(const float* x, const float* affA, const float* affB, int nfr, int HW, int C, float* y,
                       void* stream);
/* y[n][p][0..C) = SiLU?(concat(src0, src1)[n][p][c] * A[n][c] + B[n][c]): GroupNorm(+FiLM)+SiLU and the skip concat
 * materialised once as the input of a 3x3 conv (in_layers / out_layers of ResBlock, unet.py:150-199). */
int vd_op_affine_act(const float* src0, const float* src1, int C0, int C, const float* affA, const float* affB, int nfr,
                     int HW, int act, float* y, void* stream);
int vd_op_gn_temporal(const float* x, const float* gamma, const float* beta, int B, int T, int HW, int C, float* y,
                      void* stream);
int vd_op_attn_spatial(const float* qkv, int nfr, int L, int C, int heads, float* out, void* stream);
int vd_op_attn_temporal(const float* qkv, const float* Rk, const float* Rq, const float* Rv, const float* mask, int B,
                        int T, int HW, int C, int heads, int allow_pad, float* out, void* stream);
int vd_op_out_conv(const float* x, const float* affA, const float* affB, const float* w_packed, const float* bias,
                   int nfr, int H, int W, int C, int Cout, float* out_nchw, void* stream);

#ifdef __cplusplus
}
#endif
#endif
