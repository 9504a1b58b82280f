/*
 * gradtts_abi.h -- C ABI of libgradtts_gfx950.so, the MI355X-native Grad-TTS / DiffVC decoder sampling path.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own: its hot path is
 * Python calling torch ops.  Each entry point below therefore names the reference *Python* symbol it
 * replaces; the Python host (the modules under speech-backbones_amd/model/) keeps that symbol's signature and calls these
 * functions through ctypes.  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  All tensors are fp32, contiguous, row-major, in the
 *     reference's own layouts ([B,80,T] mels, [B,1,T] masks flattened to [B,T], NCHW inside).
 *   - the CALLER owns every device buffer (inputs, outputs, packed weights, workspace); the library never
 *     allocates device memory and keeps no device pointer after a call returns.
 *   - every call enqueues on the given hipStream_t and returns without synchronising (two documented exceptions, both off the
 *     sampling path: gtts_pack_weights of a GTTS_PREC_F16F8 plan and gtts_workspace_status).  The sampler may fan sub-batches
 *     out onto side streams, but only onto streams the CALLER registered with gtts_plan_set_streams (they are forked
 *     from and joined back into the call's stream inside the call, also on error paths).
 *   - a plan is host-side metadata.  Query functions take `const gtts_plan *` and touch nothing; the enqueueing
 *     functions take `gtts_plan *` and serialise on a mutex inside the plan for the duration of the host-side enqueue,
 *     so a plan may be shared by host threads as long as every call brings its own workspace.
 *   - no environment variable changes results; diagnostics (op skipping, phase traces, timing ablations) exist only in
 *     -DGTTS_DIAG builds, never in the product library.
 *   - every function returns 0 on success or a negative GTTS_E_* code; gtts_last_error() gives the text
 *     (thread-local).  No C++ exception crosses the boundary.
 *   - T (frames) must be a multiple of 4 (Grad-TTS/model/utils.py:13-17 fix_len_compatibility), n_feats a
 *     multiple of 4.
 */
#ifndef GRADTTS_ABI_H
#define GRADTTS_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTTS_ABI_VERSION 6

enum {
    GTTS_OK = 0,
    GTTS_E_NULL = -1,        /* required pointer is NULL                                  */
    GTTS_E_SHAPE = -2,       /* bad B/T/F (T % 4 != 0, non-positive sizes, ...)           */
    GTTS_E_CONFIG = -3,      /* unsupported model configuration                           */
    GTTS_E_HIP = -4,         /* HIP launch / runtime error (text in gtts_last_error)      */
    GTTS_E_PARAMS = -5,      /* parameter list does not match the plan's state_dict layout */
    GTTS_E_WORKSPACE = -6,   /* workspace too small                                       */
    GTTS_E_RANGE = -7        /* ABI 6: a value lies outside the range the plan's precision represents (GTTS_PREC_F16F8 weights) */
};

/* precision of the dense contractions (3x3 / 1x1 / transposed convs, attention products) */
enum {
    GTTS_PREC_BF16X3 = 0,    /* split-bf16 (hi/lo, 3 MFMAs, fp32 accumulate): fp32-grade accuracy                 */
    GTTS_PREC_BF16 = 1,      /* single bf16 MFMA, fp32 accumulate, fp32 activation storage                   */
    GTTS_PREC_BF16_STORE = 2,/* BASELINE.json config 3 as written: single bf16 MFMA, fp32 accumulate, and every
                                activation tensor of the U-Net stored as bf16 (weights are bf16 already); GroupNorm
                                statistics come from the fp32 accumulators before rounding.  Grad-TTS plans only. */
    GTTS_PREC_F16F8 = 3      /* ABI 5.  fp32-grade like BF16X3 (~2^-17 per product), two MFMA pass-equivalents instead of three
                                on the 3x3 Block convolutions and the Upsample transposed convolutions: x = fp16 hi + residual; hi*hi on the fp16 MFMA, both cross terms
                                (w * x_lo + w_lo * x, operands rounded to fp8 e4m3) in ONE fp8 MFMA per 32 channels.  Every other
                                contraction of the plan (1x1, Downsample, attention, the 2-channel first layer) stays BF16X3.
                                RANGE CONTRACT (ABI 6, enforced): the weights of those layers must satisfy |w| < 63.97 (w 2^10
                                in fp16) -- gtts_pack_weights checks every such weight on the device and returns GTTS_E_RANGE
                                naming a layer (nothing is ever packed as inf); an activation with |x| >= 1024 keeps an
                                fp16-grade cross term only (its fp8 operand saturates; beyond 65504 the fp16 half overflows) --
                                the staging kernels record every such event and the maximum |x| in the first 16 bytes of the
                                caller's workspace, read with gtts_workspace_status.
                                cfg.conv_ws selects the kernel (persistent / uniform waves) AND which layers take the split: the
                                64-channel layers only with conv_ws = 1.  A packed blob is valid for the plan it was packed for. */
};

typedef void *gtts_stream_t; /* hipStream_t */

/* Mirrors GradLogPEstimator2d.__init__ (Grad-TTS/model/diffusion.py:129-130) plus Diffusion's betas
 * (diffusion.py:228-230). */
typedef struct gtts_unet_cfg {
    int dim;             /* dec_dim, 64                                      */
    int n_feats;         /* 80                                               */
    int n_spks;          /* 1 -> 2 input channels; >1 -> 3 + spk_mlp         */
    int spk_emb_dim;     /* 64                                               */
    int groups;          /* GroupNorm groups, 8                              */
    float pe_scale;      /* 1000                                             */
    float beta_min;      /* 0.05                                             */
    float beta_max;      /* 20.0                                             */
    int precision;       /* GTTS_PREC_*                                      */
    int keep_intermediates; /* 1: every op output gets its own workspace slot (tests / debugging)       */
    /* ---- DiffVC decoder (DiffVC/model/diffusion.py:17-59): arch = 1, dim = dim_base (256), pe_scale 1000 */
    int arch;            /* 0: Grad-TTS GradLogPEstimator2d, 1: DiffVC GradLogPEstimator                 */
    int dim_cond;        /* 128: condition channels appended to [mean, x] (130 input channels)          */
    int use_ref_t;       /* 1: RefBlock on the diffused reference mel feeds the condition                */
    int c_dim;           /* 256: speaker-embedding width                                                 */
    double vc_beta_min;  /* DiffVC schedule scalars are Python doubles in the reference (diffusion.py:120-149) */
    double vc_beta_max;
    /* ---- ABI 3: which kernel runs the Block 3x3 convolutions of 128-channel-and-wider layers in GTTS_PREC_BF16X3 / F16F8 ---- */
    int conv_ws;         /* 0: uniform-wave kernel (conv_mfma.hip; overlaps with other streams' kernels: best with three
                            sub-batch streams in BF16X3 on the Grad-TTS dim-64 network).  1: persistent wave-specialised kernel
                            (conv_ws.hip; a higher MFMA rate per launch, but it fills every CU's registers: best where these
                            convolutions dominate -- the DiffVC dim-256 decoder -- and in GTTS_PREC_F16F8, where it also runs the
                            64-channel layers, unsplit).  Either way results do not depend on the batch split; the two modes
                            agree to fp32 rounding of the GroupNorm statistics. */
} gtts_unet_cfg;

typedef struct gtts_plan gtts_plan;   /* host-side metadata only */

int gtts_abi_version(void);
const char *gtts_last_error(void);

/* ---- plan ------------------------------------------------------------------------------------------ */
int gtts_plan_create(const gtts_unet_cfg *cfg, gtts_plan **out);
void gtts_plan_destroy(gtts_plan *plan);

/* state_dict layout the plan expects, in the reference's registration order:
 * name (relative to `estimator.`), rank and dims of parameter i; count via gtts_plan_num_params. */
int gtts_plan_num_params(const gtts_plan *plan);
int gtts_plan_param_info(const gtts_plan *plan, int i, const char **name, int *rank, int dims[4]);

size_t gtts_packed_weight_bytes(const gtts_plan *plan);
/* Workspace for gtts_estimator_forward / gtts_reverse_diffusion at (B,T); depends on the number of registered side
 * streams (each sub-batch works in its own slice), so query it AFTER gtts_plan_set_streams. */
size_t gtts_workspace_bytes(const gtts_plan *plan, int B, int T);

/* Register n in {0, 2, 3, 4} caller-owned side streams (hipStream_t) on which gtts_reverse_diffusion runs sub-batches
 * of the utterance batch side by side (bit-identical results; utterances are independent).  n = 0 (default): everything
 * runs on the stream passed to the call.  The streams must outlive the plan or be unregistered (n = 0) first. */
int gtts_plan_set_streams(gtts_plan *plan, const gtts_stream_t *streams, int n);

/* on != 0: gtts_reverse_diffusion captures the launches of a call into a hipGraph the first time it sees an argument
 * tuple (all pointers, shapes and the step range are part of the key) and replays it with one hipGraphLaunch on later
 * calls with the same tuple -- for the launch-bound small-batch regime (B = 1 inference, Grad-TTS/inference.py:62-76).
 * Keep the buffers alive and at the same addresses to hit the cache (at most 8 graphs are kept, LRU).  on == 0 drops
 * the cached graphs.  Results are identical to the eager path (same kernels, same order). */
int gtts_plan_set_graph(gtts_plan *plan, int on);

/* Re-layout the estimator parameters (device fp32 pointers, in gtts_plan_param_info order) into the packed
 * blob the kernels read (bf16 hi/lo MFMA fragment order for conv weights, fp32 for the rest).
 * `freq` = the 32 (dim/2) sinusoidal frequencies exp(-k ln(1e4)/(dim/2-1)) as fp32 device values computed by
 * the host exactly as SinusoidalPosEmb does (diffusion.py:121-122).
 * GTTS_PREC_F16F8 plans (ABI 6): the packer range-checks the 3x3 Block-convolution weights on the device and this call then
 * SYNCHRONISES `stream` to read the verdict: GTTS_E_RANGE (text names the count, max |w| and a layer) when a weight does not fit
 * the format; the blob must then not be used -- pack the model with a GTTS_PREC_BF16X3 plan instead. */
int gtts_pack_weights(const gtts_plan *plan, const void *const *param_ptrs, int n_params, const float *freq,
                      void *packed, gtts_stream_t stream);

/* ABI 6.  Activation range record of the last gtts_estimator_forward / gtts_reverse_diffusion (or gtts_vc_*) call that used
 * `workspace`: *n_events = number of staging lanes x launches that split an activation with |x| >= 1024 (GTTS_PREC_F16F8 plans
 * only), plus -- in every precision -- (samples x Block convolutions) whose GroupNorm statistics came out non-finite;
 * *max_abs = the largest |x| seen (inf for a non-finite one; 0 when n_events == 0).  Sticky over the step ranges of one
 * sampling run (reset where step_begin == 0, and by every estimator call).  This query -- and gtts_pack_weights of an F16F8 plan --
 * are the only calls of the ABI that SYNCHRONISE `stream`; do not call them inside a stream capture. */
int gtts_workspace_status(const void *workspace, unsigned *n_events, float *max_abs, gtts_stream_t stream);

/* ---- GradLogPEstimator2d.forward(x, mask, mu, t, spk)  diffusion.py:174-216 -------------------------- */
/* x, mu, out [B,F,T]; mask [B,T]; t [B]; spk [B,spk_emb_dim] (already embedded) or NULL. */
int gtts_estimator_forward(gtts_plan *plan, const void *packed, const float *x, const float *mask,
                           const float *mu, const float *t, const float *spk, float *out, void *workspace,
                           size_t workspace_bytes, int B, int T, gtts_stream_t stream);

/* ---- one update of Diffusion.reverse_diffusion  diffusion.py:264-274 --------------------------------- */
/* xt is updated IN PLACE.  noise == NULL: ODE branch; else SDE branch with the pre-drawn N(0,1) tensor. */
int gtts_euler_step(float *xt, const float *mu, const float *est, const float *mask, const float *noise,
                    float beta_t, float h, int B, int F, int T, gtts_stream_t stream);

/* ---- Diffusion.reverse_diffusion / forward  diffusion.py:254-279 (whole N-step loop) ------------------ */
/* z, mu, out [B,F,T]; mask [B,T]; spk nullable.  Runs steps [step_begin, step_end) of the n_timesteps-step schedule:
 * step_begin == 0 first sets out = z * mask, later ranges continue in place on `out`, so a host can draw the SDE noise
 * in bounded chunks.  noise nullable [step_end - step_begin, B,F,T] (stoc=True <=> noise != NULL). */
int gtts_reverse_diffusion(gtts_plan *plan, const void *packed, const float *z, const float *mask,
                           const float *mu, const float *spk, const float *noise, float *out, void *workspace,
                           size_t workspace_bytes, int B, int T, int n_timesteps, int step_begin, int step_end,
                           gtts_stream_t stream);

/* ---- DiffVC: GradLogPEstimator.forward(x, x_mask, mean, ref, ref_mask, c, t)  DiffVC/model/diffusion.py:61-106 -- */
/* x, mean, out [B,F,T]; x_mask [B,T]; xt_ref [B,1,F,T_ref] (the diffused reference, diffusion.py:173-176);
 * ref_mask [B,T_ref]; c [B,c_dim]; t [B].  T % 4 == 0; T_ref is free. */
size_t gtts_vc_workspace_bytes(const gtts_plan *plan, int B, int T, int T_ref);
int gtts_vc_estimator_forward(gtts_plan *plan, const void *packed, const float *x, const float *x_mask,
                              const float *mean, const float *xt_ref, const float *ref_mask, const float *c, const float *t,
                              float *out, void *workspace, size_t workspace_bytes, int B, int T, int T_ref,
                              gtts_stream_t stream);
/* ---- DiffVC: Diffusion.reverse_diffusion / forward  DiffVC/model/diffusion.py:164-205 ------------------------ */
/* mode 0 'pf', 1 'em', 2 'ml'; steps [step_begin, step_end) as in gtts_reverse_diffusion; noise
 * [step_end - step_begin, B,F,T] pre-drawn N(0,1) (required for em / ml, ignored for pf); ref, mean_ref [B,F,T_ref].
 * The schedule scalars (beta, gamma, mu, nu, sigma, kappa, omega) are host doubles. */
int gtts_vc_reverse_diffusion(gtts_plan *plan, const void *packed, const float *z, const float *mask,
                              const float *mean, const float *ref, const float *ref_mask, const float *mean_ref,
                              const float *c, const float *noise, float *out, void *workspace, size_t workspace_bytes, int B,
                              int T, int T_ref, int n_timesteps, int mode, int step_begin, int step_end,
                              gtts_stream_t stream);

/* ---- monotonic_align.maximum_path  monotonic_align/core.pyx:9-45 + __init__.py:8-23 ------------------- */
/* value [b,tx,ty] fp32 (NOT modified), mask [b,tx,ty] fp32 or NULL, t_x / t_y [b] int32 device arrays,
 * path [b,tx,ty] int32 (written: 0/1), scratch >= gtts_mas_scratch_bytes(b,tx,ty) device bytes.  t_x[i] > t_y[i] (an empty band in
 * core.pyx:18) follows the reference too: its backtrack over the untouched values. */
size_t gtts_mas_scratch_bytes(int b, int tx, int ty);
int gtts_mas_maximum_path(const float *value, const float *mask, const int *t_x, const int *t_y, int *path,
                          void *scratch, int b, int tx, int ty, gtts_stream_t stream);

/* CPU twin (host pointers, no stream): the reference's maximum_path accepts tensors on any device and runs its Cython
 * kernel on the host (__init__.py:8-23); bit-identical to the GPU kernel and to core.pyx. */
int gtts_mas_maximum_path_cpu(const float *value, const float *mask, const int *t_x, const int *t_y, int *path, int b,
                              int tx, int ty);

/* ---- multi-GPU: the single collective -- RCCL broadcast of the packed weight blob from `root` over xGMI.
 * comm = the caller's ncclComm_t.  RCCL is resolved from the running process at call time (no link-time dependency). */
int gtts_bcast_weights(void *packed, size_t bytes, int root, void *comm, gtts_stream_t stream);

/* ---- pre-decoder glue of GradTTS.forward: generate_path + mu_y = attn^T . mu_x + z  (tts.py:84-94, utils.py:26-39) -
 * duration [B,t_x] fp32 (= w_ceil incl. length_scale), x_mask [B,t_x] fp32, y_lengths [B] int32 (device),
 * mu_x [B,F,t_x]; noise [B,F,T] or NULL, temperature; outputs attn [B,t_x,T], mu_y [B,F,T], z [B,F,T] (z may be NULL;
 * noise NULL gives z = mu_y).  Bit-identical to the reference's CPU path (sequential fp32 cumsum, float compares). */
int gtts_expand_alignment(const float *duration, const float *x_mask, const int *y_lengths, const float *mu_x,
                          const float *noise, float temperature, float *attn, float *mu_y, float *z, int B, int F,
                          int t_x, int T, gtts_stream_t stream);

/* ---- MAS score matrix of GradTTS.compute_loss  (tts.py:130-139): log N(y_j; mu_x_i, I) for every (token i, frame j) --
 * mu_x [B,F,t_x], y [B,F,T] -> log_prior [B,t_x,T] (fp32, feeds gtts_mas_maximum_path without leaving the device).
 * Evaluated as -0.5 sum_f (y - mu)^2 - 0.5 F log(2 pi): equal to the reference's three-matmul form up to fp32 rounding. */
int gtts_log_prior(const float *mu_x, const float *y, float *log_prior, int B, int F, int t_x, int T, gtts_stream_t stream);

/* ---- HiFi-GAN generator: mel -> waveform, the step after the sampling path (Grad-TTS/inference.py:81;
 * Grad-TTS/hifi-gan/models.py:77-128, configuration Grad-TTS/checkpts/hifigan-config.json) ------------------------------ */
typedef struct gtts_voc_cfg {
    int n_mels;                       /* 80                                                            */
    int upsample_initial_channel;     /* 512                                                           */
    int n_ups;                        /* len(upsample_rates), <= 8                                     */
    int upsample_rates[8];            /* 8,8,2,2      (powers of two)                                  */
    int upsample_kernel_sizes[8];     /* 16,16,4,4    (= 2 * rate)                                     */
    int n_kernels;                    /* len(resblock_kernel_sizes), <= 8                              */
    int resblock_kernel_sizes[8];     /* 3,7,11       (odd, <= 11)                                     */
    int resblock_dilations[8][3];     /* 1,3,5 each   (ResBlock2 uses the first two)                   */
    int resblock_type;                /* 1: ResBlock1 (models.py:13-50), 2: ResBlock2 (:53-74)         */
} gtts_voc_cfg;
typedef struct gtts_voc gtts_voc;     /* host-side metadata only */
int gtts_voc_create(const gtts_voc_cfg *cfg, gtts_voc **out);
void gtts_voc_destroy(gtts_voc *voc);
/* parameter i of the layout the packer expects: `<module path>.weight` / `.bias` with weight normalisation already
 * folded (Generator.remove_weight_norm, models.py:122-128); Conv1d weights [cout][cin][k], ConvTranspose1d [cin][cout][k] */
int gtts_voc_num_params(const gtts_voc *voc);
int gtts_voc_param_info(const gtts_voc *voc, int i, const char **name, int *rank, int dims[4]);
size_t gtts_voc_packed_bytes(const gtts_voc *voc);
int gtts_voc_pack(const gtts_voc *voc, const void *const *param_ptrs, int n_params, void *packed, gtts_stream_t stream);
size_t gtts_voc_workspace_bytes(const gtts_voc *voc, int B, int T);
int gtts_voc_hop(const gtts_voc *voc);          /* output samples per mel frame (product of the rates: 256) */
/* Generator.forward: mel [B, n_mels, T] fp32 -> wav [B, 1, T * hop] fp32 in (-1, 1). */
int gtts_voc_forward(const gtts_voc *voc, const void *packed, const float *mel, float *wav, void *workspace,
                     size_t workspace_bytes, int B, int T, gtts_stream_t stream);

/* ---- encoders either side of the sampling path: Grad-TTS TextEncoder (Grad-TTS/model/text_encoder.py:281-326) and DiffVC
 * MelEncoder (DiffVC/model/encoder.py:257-284); inference only (dropout = identity) ------------------------------------- */
typedef struct gtts_enc_cfg {
    int mode;                 /* 0: TextEncoder (ids -> mu, logw), 1: MelEncoder (mel -> mel)                         */
    int n_vocab;              /* 149 (TextEncoder)                                                                    */
    int n_feats;              /* 80                                                                                   */
    int channels;             /* n_enc_channels 192                                                                   */
    int filter_channels;      /* 768                                                                                  */
    int filter_channels_dp;   /* 256 (TextEncoder's DurationPredictor)                                                */
    int n_heads;              /* 2                                                                                    */
    int n_layers;             /* 6                                                                                    */
    int kernel_size;          /* 3 (FFN / DurationPredictor convolutions; odd)                                        */
    int window_size;          /* 4 (relative-position window; 0: none)                                                */
} gtts_enc_cfg;
typedef struct gtts_enc gtts_enc;
int gtts_enc_create(const gtts_enc_cfg *cfg, gtts_enc **out);
void gtts_enc_destroy(gtts_enc *enc);
/* parameters in the reference module's registration order, names relative to the encoder module (e.g. `emb.weight`,
 * `prenet.conv_layers.0.weight`, `encoder.attn_layers.0.emb_rel_k`, `proj_w.norm_1.gamma`) */
int gtts_enc_num_params(const gtts_enc *enc);
int gtts_enc_param_info(const gtts_enc *enc, int i, const char **name, int *rank, int dims[4]);
size_t gtts_enc_packed_bytes(const gtts_enc *enc);
int gtts_enc_pack(const gtts_enc *enc, const void *const *param_ptrs, int n_params, void *packed, gtts_stream_t stream);
size_t gtts_enc_workspace_bytes(const gtts_enc *enc, int B, int L);
/* mode 0: ids [B,L] int64 (device), x_mask [B,L] fp32 -> mu [B,n_feats,L], logw [B,1,L];  mel is ignored.
 * mode 1: mel [B,n_feats,L], x_mask -> mu [B,n_feats,L] (the encoded mel); ids / logw are ignored. */
int gtts_enc_forward(const gtts_enc *enc, const void *packed, const long long *ids, const float *mel, const float *x_mask,
                     float *mu, float *logw, void *workspace, size_t workspace_bytes, int B, int L, gtts_stream_t stream);

/* ---- DiffVC PostNet (DiffVC/model/postnet.py:40-53; the last stage of the "average voice" encoder, vc.py:33-41):
 * x [B,n_feats,T], mask [B,T] -> out [B,n_feats,T].  dim % 64 == 0, 8 GroupNorm groups.  Parameter names are the module's
 * state_dict keys (`init_conv.weight`, `res_block.block1.block.0.weight` [dim,dim,7,7], ...). */
typedef struct gtts_postnet gtts_postnet;
int gtts_postnet_create(int dim, int n_feats, int groups, gtts_postnet **out);
void gtts_postnet_destroy(gtts_postnet *pn);
int gtts_postnet_num_params(const gtts_postnet *pn);
int gtts_postnet_param_info(const gtts_postnet *pn, int i, const char **name, int *rank, int dims[4]);
size_t gtts_postnet_packed_bytes(const gtts_postnet *pn);
int gtts_postnet_pack(const gtts_postnet *pn, const void *const *param_ptrs, int n_params, void *packed, gtts_stream_t stream);
size_t gtts_postnet_workspace_bytes(const gtts_postnet *pn, int B, int T);
int gtts_postnet_forward(const gtts_postnet *pn, const void *packed, const float *x, const float *mask, float *out,
                         void *workspace, size_t workspace_bytes, int B, int T, gtts_stream_t stream);

/* ---- training hot path, first kernels (Grad-TTS/train.py:105-119; Grad-TTS/model/diffusion.py:244-252,281-294) --------
 * The host (model/_train_ops.py) wraps these and the ABI-3 entry points further down in torch.autograd.Function; autograd only
 * sequences them (and adds gradient accumulations). */
/* Diffusion.forward_diffusion: xt = (x0 d + mu (1-d) + z sqrt(1 - e^{-cum})) mask, z_masked = z mask; d = e^{-cum/2},
 * cum = beta_min t + (beta_max - beta_min) t^2 / 2 per sample; x0, mu, z, xt, z_masked [B,F,T], mask [B,T], t [B]. */
int gtts_diffusion_noising(const float *x0, const float *mu, const float *z, const float *mask, const float *t, float beta_min,
                           float beta_max, float *xt, float *z_masked, int B, int F, int T, gtts_stream_t stream);
/* Diffusion.loss_t: r = eps sqrt(1 - e^{-cum}) + z_masked; partials[i] = sum of r^2 over 256-element blocks (fixed order;
 * loss = sum(partials) * inv_denom with inv_denom = 1 / (sum(mask) F)); grad_eps (nullable) = d loss / d eps. */
size_t gtts_score_loss_partials(int B, int F, int T);
int gtts_score_loss(const float *eps, const float *z_masked, const float *t, float beta_min, float beta_max, float inv_denom,
                    float *partials, float *grad_eps, int B, int F, int T, gtts_stream_t stream);
/* Block's convolution for training: y = Conv2d_3x3(x * mask) + bias with weights packed by gtts_conv3x3_pack (transposed = 0);
 * the data gradient is the same call on dy with weights packed transposed = 1 (cin / cout swapped), an all-ones mask and a
 * zero bias; the weight / bias gradient is gtts_conv3x3_wgrad (dw, db are overwritten).  x [B,cin,H,W], mask [B,W]. */
size_t gtts_conv3x3_packed_bytes(int cin, int cout);
int gtts_conv3x3_pack(const float *w, void *packed, int cin, int cout, int transposed, gtts_stream_t stream);
int gtts_conv3x3_masked(const float *x, const float *mask, const void *packed, const float *bias, float *y, int B, int cin,
                        int cout, int H, int W, gtts_stream_t stream);
int gtts_conv3x3_wgrad(const float *x, const float *mask, const float *dy, float *dw, float *db, int B, int cin, int cout, int H,
                       int W, gtts_stream_t stream);

/* The same weight / bias gradient as an LDS-tiled, deterministic reduction (no atomics): cin and cout multiples of 64;
 * workspace: gtts_conv3x3_wgrad_workspace_bytes(...) bytes of device memory (per-slice partial tiles). */
size_t gtts_conv3x3_wgrad_workspace_bytes(int B, int cin, int cout, int H, int W);
int gtts_conv3x3_wgrad_tiled(const float *x, const float *mask, const float *dy, float *dw, float *db, void *workspace,
                             size_t workspace_bytes, int B, int cin, int cout, int H, int W, gtts_stream_t stream);

/* Block's GroupNorm + Mish + mask for training (Grad-TTS/model/diffusion.py:53-58,13-15): out = Mish(GroupNorm(y)) * mask.
 * y, out [B,C,H,W]; gamma, beta [C]; mask [B,W] (columns).  stats: gtts_gn_mish_stats_floats(B, groups) floats, 8-byte aligned;
 * its first [B][groups][2] = (mean, 1/sqrt(var + eps)) are written by the forward call (the rest is its reduction scratch) and
 * read by the backward call, which overwrites dy [B,C,H,W], dgamma [C], dbeta [C];
 * scratch: gtts_gn_mish_scratch_bytes(B, C) bytes of device memory. */
size_t gtts_gn_mish_stats_floats(int B, int groups);
int gtts_gn_mish_forward(const float *y, const float *gamma, const float *beta, const float *mask, float *out, float *stats,
                         int B, int C, int H, int W, int groups, float eps, gtts_stream_t stream);
size_t gtts_gn_mish_scratch_bytes(int B, int C);
int gtts_gn_mish_backward(const float *dout, const float *y, const float *gamma, const float *beta, const float *mask,
                          const float *stats, float *dy, float *dgamma, float *dbeta, void *scratch, int B, int C, int H, int W,
                          int groups, gtts_stream_t stream);

/* ---- training hot path, the rest of the score network (ABI 3; Grad-TTS/model/diffusion.py:19-108,140-176) ----------------
 * Every tensor-sized op of Diffusion.compute_loss's forward and backward (Upsample's gradients run as stride-1 3x3 operations
 * over gtts_space_to_depth2 planes: no MIOpen / rocBLAS kernel is left in the step). */
/* Block's convolution on a channel concatenation read in place (torch.cat of the up path, diffusion.py:166): x [B,c0,H,W],
 * x1 [B,cin-c0,H,W] (nullptr: one source, c0 ignored); c0 a multiple of 16 (forward) / 64 (weight gradient). */
int gtts_conv3x3_masked2(const float *x, const float *x1, int c0, const float *mask, const void *packed, const float *bias, float *y,
                         int B, int cin, int cout, int H, int W, gtts_stream_t stream);
/* the same with a column mask on the OUTPUT (omask [B][W], nullable): the data gradient of a masked convolution in one pass */
int gtts_conv3x3_masked3(const float *x, const float *x1, int c0, const float *mask, const float *omask, const void *packed,
                         const float *bias, float *y, int B, int cin, int cout, int H, int W, gtts_stream_t stream);
int gtts_conv3x3_wgrad_tiled2(const float *x, const float *x1, int c0, const float *mask, const float *dy, float *dw, float *db,
                              void *workspace, size_t workspace_bytes, int B, int cin, int cout, int H, int W, gtts_stream_t stream);
/* 1x1 convolutions (res_conv, to_qkv, to_out: diffusion.py:70,87-88): y = Conv2d_1x1(x * mask) + bias; the data gradient is the
 * same call on dy with weights packed transposed = 1 (the mask then applies to dy: columns do not mix); weight gradient
 * dw [cout][cin], db [cout] (nullable); mask [B,W] (nullable in the weight gradient).  cin, cout multiples of 64. */
size_t gtts_conv1x1_packed_bytes(int cin, int cout);
int gtts_conv1x1_pack(const float *w, void *packed, int cin, int cout, int transposed, gtts_stream_t stream);
int gtts_conv1x1_masked(const float *x, const float *mask, const void *packed, const float *bias, float *y, int B, int cin,
                        int cout, int H, int W, gtts_stream_t stream);
size_t gtts_conv1x1_wgrad_workspace_bytes(int B, int cin, int cout, int H, int W);
int gtts_conv1x1_wgrad(const float *x, const float *mask, const float *dy, float *dw, float *db, void *workspace,
                       size_t workspace_bytes, int B, int cin, int cout, int H, int W, gtts_stream_t stream);
/* First-layer weight gradient (stacked (mu, x[, spk]) input: cin 2 or 3; ksize 3 or 1): dw [cout][cin][k][k], db [cout]. */
size_t gtts_conv_wgrad_small_scratch_floats(int B, int cin, int cout, int ksize);
int gtts_conv_wgrad_small(const float *x, const float *mask, const float *dy, float *dw, float *db, float *scratch, int B, int cin,
                          int cout, int H, int W, int ksize, gtts_stream_t stream);
/* Downsample (up = 0: Conv2d 3x3 stride 2 pad 1, w [cout][cin][3][3]) / Upsample (up = 1: ConvTranspose2d 4x4 stride 2 pad 1,
 * w [cin][cout][4][4]) of x * mask (diffusion.py:19-34); x [B,cin,H,W], mask [B,W], y [B,cout,H/2,W/2] or [B,cout,2H,2W].
 * Downsample's data gradient is the up = 1 call on dy with its forward weight zero-padded to 4x4. */
size_t gtts_conv_resample_packed_bytes(int cin, int cout, int up);
int gtts_conv_resample_pack(const float *w, void *packed, int cin, int cout, int up, gtts_stream_t stream);
int gtts_conv_resample(const float *x, const float *mask, const void *packed, const float *bias, float *y, int B, int cin, int cout,
                       int H, int W, int up, gtts_stream_t stream);
/* ABI 4: every weight pack of a training step in ONE launch.  item.kind: 0 Block 3x3 (gtts_conv3x3_pack), 1 1x1 (gtts_conv1x1_pack),
 * 2 Downsample, 3 Upsample (gtts_conv_resample_pack), 4 Downsample's data gradient (its forward weight [cout][cin][3][3] packed as the
 * zero-padded 4x4 transposed convolution; cin / cout those of the GRADIENT convolution); transposed as in the single-pack calls
 * (kinds 0 / 1).  gtts_pack_batch_describe turns n items into a descriptor table in HOST memory (gtts_pack_batch_desc_bytes(n)
 * bytes) and returns the launch width; the caller keeps a device copy of the table (addresses are stable across steps) and calls
 * gtts_pack_batch at the top of every step. */
typedef struct gtts_pack_item { const float *w; void *packed; int kind, cin, cout, transposed; } gtts_pack_item;
size_t gtts_pack_batch_desc_bytes(int n);
int gtts_pack_batch_describe(const gtts_pack_item *items, int n, void *desc_host, int *grid_x);
int gtts_pack_batch(const void *desc_dev, int n, int grid_x, gtts_stream_t stream);
int gtts_zero_insert2(const float *in, float *out, int B, int C, int h, int w, gtts_stream_t stream);
/* out [B,4C,h,w] = the four stride-2 phases of in [B,C,2h,2w], channel block (pr * 2 + pc) = rows 2y + 1 - pr, columns 2x + 1 - pc:
 * Upsample's data / weight gradient are the 3x3 stride-1 convolution / weight gradient over these planes. */
int gtts_space_to_depth2(const float *in, float *out, int B, int C, int h, int w, gtts_stream_t stream);
/* GroupNorm + Mish + mask with ResnetBlock's time term: out = Mish(GroupNorm(y)) * mask + tb[b,c] (tb, dtb [B][C], nullable). */
int gtts_gn_mish_forward_tb(const float *y, const float *gamma, const float *beta, const float *mask, const float *tb, float *out,
                            float *stats, int B, int C, int H, int W, int groups, float eps, gtts_stream_t stream);
int gtts_gn_mish_backward_tb(const float *dout, const float *y, const float *gamma, const float *beta, const float *mask,
                             const float *stats, float *dy, float *dgamma, float *dbeta, float *dtb, void *scratch, int B, int C,
                             int H, int W, int groups, gtts_stream_t stream);
/* LinearAttention between its two 1x1 convolutions (diffusion.py:90-100): qkv [B][384][N] -> out [B][128][N]; ctx [B][4][32][32]
 * and stat [B][4][32][2] (softmax row maximum, reciprocal sum) are kept for the backward call, which writes dqkv [B][384][N];
 * dctx [B][4][32][32], rdot [B][4][32]: outputs used as scratch; scratch: gtts_attn_train_scratch_floats(B, N) floats. */
size_t gtts_attn_train_scratch_floats(int B, int N);
int gtts_attn_train_forward(const float *qkv, float *out, float *ctx, float *stat, float *scratch, int B, int N, gtts_stream_t stream);
int gtts_attn_train_backward(const float *qkv, const float *dout, const float *ctx, const float *stat, float *dqkv, float *dctx,
                             float *rdot, float *scratch, int B, int N, gtts_stream_t stream);
/* Residual(Rezero(f)): y = f * g + x, g one device scalar (diffusion.py:40-46,103-108); backward df = dy * g, dg = sum(dy * f);
 * n floats, a multiple of 4; scratch: gtts_rezero_scratch_bytes(n) bytes. */
int gtts_rezero_forward(const float *f, const float *x, const float *g, float *y, size_t n, gtts_stream_t stream);
size_t gtts_rezero_scratch_bytes(size_t n);
int gtts_rezero_backward(const float *dy, const float *f, const float *g, float *df, float *dg, void *scratch, size_t n,
                         gtts_stream_t stream);
/* out = a + b * mask over [B,C,H,W], mask [B,W] (a nullable: b * mask; mask nullable: a + b); b_cstride > 0: b is a C-channel
 * slice of a contiguous tensor with b_cstride channels (the split of a concatenation's gradient). */
int gtts_add_masked(const float *a, const float *b, const float *mask, float *out, int B, int C, int H, int W, int b_cstride,
                    gtts_stream_t stream);
/* final_conv (C -> 1) with both masks (diffusion.py:175-176): out [B,1,H,W] = (sum_c w[c] x[b,c] m + bias) m and its gradients
 * dx [B,C,H,W], dw [C], db [1]; scratch: gtts_final_conv_scratch_floats(B, C, H, W) floats. */
int gtts_final_conv_forward(const float *x, const float *w, const float *bias, const float *mask, float *out, int B, int C, int H,
                            int W, gtts_stream_t stream);
size_t gtts_final_conv_scratch_floats(int B, int C, int H, int W);
int gtts_final_conv_backward(const float *x, const float *w, const float *mask, const float *dout, float *dx, float *dw, float *db,
                             float *scratch, int B, int C, int H, int W, gtts_stream_t stream);

/* ---- ABI 6: DiffVC decoder training, RefBlock's InstanceNorm2d(affine) + GLU(dim=1) pair (DiffVC/model/modules.py:128-157) ----------
 * y [B,2C,H,W] (the block's convolution output), gamma / beta [2C]; out [B,C,H,W] = IN(y[:, :C]) * sigmoid(IN(y[:, C:])), statistics
 * per (sample, channel) plane over H x W (biased variance, eps).  stats: gtts_in_glu_stats_floats(B, C) floats written by the forward
 * call ((mean, rstd) of both halves) and read by the backward call, which writes dy [B,2C,H,W], dgamma [2C], dbeta [2C];
 * scratch: gtts_in_glu_scratch_floats(B, C) floats. */
size_t gtts_in_glu_stats_floats(int B, int C);
size_t gtts_in_glu_scratch_floats(int B, int C);
int gtts_in_glu_forward(const float *y, const float *gamma, const float *beta, float *out, float *stats, int B, int C, int H, int W,
                        float eps, gtts_stream_t stream);
int gtts_in_glu_backward(const float *dout, const float *y, const float *gamma, const float *beta, const float *stats, float *dy,
                         float *dgamma, float *dbeta, float *scratch, int B, int C, int H, int W, gtts_stream_t stream);

/* ---- debugging / tests: named intermediates of the last estimator call (keep_intermediates plans) ----- */
int gtts_plan_num_tensors(const gtts_plan *plan);
/* offset is in bytes into the workspace for the given (B,T); dims = {B,C,H,W}. */
int gtts_plan_tensor_info(const gtts_plan *plan, int i, int B, int T, const char **name, size_t *offset,
                          int dims[4]);
/* same for DiffVC plans, whose RefBlock tensors live on the reference mel's frame axis (T_ref) */
int gtts_vc_tensor_info(const gtts_plan *plan, int i, int B, int T, int T_ref, const char **name, size_t *offset,
                        int dims[4]);

/* ---- measurement: per-op HIP-event timing of the op program (bench.py roofline) ------------------------- */
/* Ops are the kernel launches of one estimator call, in launch order; label = layer name, kernel = the HIP kernel
 * (template instance) it launches, flops / bytes = ALGORITHMIC work of that launch for the given (B,T)
 * (SURVEY.md section 8d model: each tensor read once, written once). */
int gtts_plan_num_ops(const gtts_plan *plan);
int gtts_plan_op_info(const gtts_plan *plan, int i, int B, int T, const char **label, const char **kernel,
                      double *flops, double *bytes);
/* on != 0: every later estimator / sampler call brackets each op launch with hipEventRecord on the call's stream;
 * while profiling, the sampler runs the batch unsplit on that one stream (no sub-batch streams). */
int gtts_profile_enable(gtts_plan *plan, int on);
/* Synchronises the recorded events, adds elapsed milliseconds and launch counts per op into the two arrays
 * (length gtts_plan_num_ops) and clears the record. */
int gtts_profile_collect(gtts_plan *plan, double *ms_per_op, long long *launches_per_op);
/* ABI 4: gtts_profile_enable(plan, 2) keeps the sub-batch streams ON and brackets every launch with events on ITS stream;
 * gtts_profile_timeline then returns, per launch (at most cap), the op index, the stream (0: the call's stream, 1 + h: side
 * stream h) and start / end in milliseconds relative to the first recorded launch (HIP event timestamps share one clock
 * across streams): an un-traced timeline of which kernels were in flight together.  Clears the record. */
int gtts_profile_timeline(gtts_plan *plan, int cap, int *op, int *stream, double *t0_ms, double *t1_ms, int *n);

/* ---- measurement: ceilings of THIS chip, measured (SURVEY.md section 8d "a measured hipMemcpy/stream-triad ceiling") ----
 * Both enqueue ONE kernel on `stream`; the caller times it (HIP events) and divides.  Nothing on the sampling path calls them.
 * gtts_ubench_mfma: `workgroups` x 4 waves each issue iters x 8 independent v_mfma_f32_32x32x16_bf16 whose operand fragments
 *   are read from src (>= 4096 bytes of bf16 data: pass random values -- the chip clocks to its power budget and all-zero
 *   operands overstate what live data reaches); out: gtts_ubench_mfma_out_floats(workgroups) floats; *flops = FLOPs enqueued;
 *   iters < 0: |iters| sweeps of 16 independent v_mfma_f32_16x16x32_bf16 instead (the other bf16 shape).
 * gtts_ubench_hbm: mode 0 c = a (8 bytes per element), 1 c = a + 1.5 b (12), 2 read-only sweep of a (4), + 4: the same with
 *   nontemporal loads / stores; n floats, a multiple of 4, buffers 16-byte aligned; *bytes = bytes the launch moves. */
size_t gtts_ubench_mfma_out_floats(int workgroups);
int gtts_ubench_mfma(const void *src, size_t src_bytes, float *out, int workgroups, int iters, double *flops, gtts_stream_t stream);
int gtts_ubench_hbm(const float *a, const float *b, float *c, size_t n, int mode, int workgroups, double *bytes, gtts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GRADTTS_ABI_H */
