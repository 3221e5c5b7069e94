/* dfot_hip.h -- C ABI of libdfot_hip.so: the MI355X (gfx950) DFoT denoising engine.
 *
 * Drop-in boundary for the reference's hot path (ktncktnc/diffusion-forcing-transformer):
 *   - dfot_uvit_*      replaces   UViT3DPose / BaseBackbone.forward
 *                                 algorithms/dfot/backbones/u_vit/u_vit3d_pose.py:63-131
 *                                 algorithms/dfot/backbones/base_backbone.py:78-86
 *                                 (called from diffusion/continuous_diffusion.py:119-121)
 *   - dfot_ray_encode  replaces   DFoTVideoPose._process_conditions (ray_encoding)
 *                                 algorithms/dfot/dfot_video_pose.py:64-110, utils/geometry_utils.py:49-81,244-295
 *   - dfot_hg_prepare  replaces   HistoryGuidance prepare (q_sample of history tokens per branch)
 *                                 algorithms/dfot/history_guidance.py:446-543,929-973 ; discrete_diffusion.py:242-250
 *   - dfot_ddim_compose replaces  v->x0/eps, ddim_sample_step, HG compose, context clamp
 *                                 diffusion/discrete_diffusion.py:213-223,454-538 ; history_guidance.py:545-568,978-982 ;
 *                                 dfot_video.py:750-752
 *   - dfot_dit_*       replaces   DiT3D / BaseBackbone.forward (Kinetics-600 backbone: DiTBase "full" variant, rope_3d)
 *                                 algorithms/dfot/backbones/dit/dit3d.py:146-192, dit/dit_base.py:150-196,391-419,
 *                                 dit/dit_blocks.py:49-128,378-542 (called from diffusion/discrete_diffusion.py:173-174)
 *   - dfot_op_*        unit-testable primitives the backbone is built from (GEMM / conv / attention / norms)
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every data pointer is a DEVICE
 * pointer unless it says "host".  `stream` is a hipStream_t passed as void*.  All functions return
 * 0 (DFOT_OK) or a DFOT_ERR_* code; dfot_last_error() returns the message of the last failure on
 * the calling thread.  Nothing here allocates or synchronises inside forward/step calls (graph-capturable)
 * except dfot_uvit_reserve / dfot_uvit_create / dfot_uvit_load_weight.
 */
#ifndef DFOT_HIP_H_
#define DFOT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  DFOT_OK = 0,
  DFOT_ERR_ARG = 1,    /* null pointer / bad enum                      (reference: TypeError/ValueError) */
  DFOT_ERR_SHAPE = 2,  /* shape or length violates the contract        (reference: ValueError / assert)   */
  DFOT_ERR_HIP = 3,    /* a HIP runtime call failed                                                      */
  DFOT_ERR_STATE = 4,  /* weights missing / workspace not reserved                                        */
  DFOT_ERR_NAME = 5    /* unknown state-dict key                       (reference: load_state_dict strict) */
};

typedef struct dfot_uvit_s* dfot_uvit_t;

/* Hyper-parameters of u_vit3d_pose (configurations/algorithm/backbone/u_vit3d_pose.yaml:1-14 overridden by
 * configurations/dataset_experiment/realestate10k_video_generation.yaml:39-44). block types are fixed to
 * [ResBlock, ResBlock, TransformerBlock, TransformerBlock] with RoPE-3D, as on every BASELINE config. */
typedef struct {
  int32_t channels[4];
  int32_t emb_channels;
  int32_t num_updown_blocks[3];
  int32_t num_mid_blocks;
  int32_t num_heads;
  int32_t in_channels;   /* 3 */
  int32_t resolution;    /* x_shape[-1], e.g. 256 */
  int32_t max_tokens;    /* temporal length T, 8 */
  int32_t cond_dim;      /* 180 (ray_encoding) */
  int32_t noise_dim;     /* 256 */
  float rope_theta;      /* 10000 */
  float eps;             /* 1e-6 */
} dfot_uvit_config;

const char* dfot_last_error(void);
int dfot_version(void);

/* ---- backbone handle ------------------------------------------------------------------------- */
int dfot_uvit_create(const dfot_uvit_config* cfg, dfot_uvit_t* out);
int dfot_uvit_destroy(dfot_uvit_t h);
/* state-dict inventory (reference key names) */
int dfot_uvit_num_params(dfot_uvit_t h);
const char* dfot_uvit_param_name(dfot_uvit_t h, int index);
int dfot_uvit_param_shape(dfot_uvit_t h, int index, int64_t shape[4], int* ndim);
/* copy one fp32 tensor (device pointer, contiguous, reference layout) into the engine's packed bf16/fp32 layout */
int dfot_uvit_load_weight(dfot_uvit_t h, const char* name, const float* data, const int64_t* shape, int ndim,
                          void* stream);
/* verify every key was loaded; build derived tables (RoPE cos/sin, fused out-projection bias) */
int dfot_uvit_finalize(dfot_uvit_t h, void* stream);
/* allocate activations for a model batch of up to `max_batch` videos (B*NFE) */
int dfot_uvit_reserve(dfot_uvit_t h, int max_batch);
size_t dfot_uvit_workspace_bytes(dfot_uvit_t h);
/* tuning/debug switches: "gemm_variant" (-1 auto, else a tile form as in dfot_op_gemm),
 * "attn_variant" (2 = tuned kernel (default), 0 = baseline with transposed LDS reads for V, 1 = baseline with scalar LDS reads,
 * 3 = tuned kernel with two K/V stages, 4 = two stages + per-tile Q reload (4 waves per SIMD; slower, kept for A/B)), "time_attn" (see below) */
int dfot_uvit_set_option(dfot_uvit_t h, const char* key, int value);
/* also: "attn_force_safe" (1 = level-2 attention always takes the running-max kernel, as weights with a large QK-norm bound would)
 * Read-outs (valid after finalize), by key: "score_bound_l2" = the largest bound of |q.k| log2(e)/sqrt(d) over the level-2 blocks, from
 * their q_norm / k_norm weights (u_vit_blocks.py:255-262); "attn_kernel_l2" = 14 (no running max, bound < 64) or 5 (running max):
 * the level-2 attention kernel forward runs */
int dfot_uvit_query(dfot_uvit_t h, const char* key, double* value);
/* "time_attn" = N > 0 records HIP events (on the launch stream) around the next N level-2 attention launches;
 * this call synchronises on them, returns the summed duration and the number of launches, and resets the count */
int dfot_uvit_attn_timing(dfot_uvit_t h, double* total_ms, int64_t* launches);

/* out[B,T,C,H,W] = model(x[B,T,C,H,W], noise_levels[B,T], external_cond[B,T,180,H,W], external_cond_mask[B])
 * all fp32 contiguous; noise_levels is the float level the reference passes (0.125*logsnr[k]);
 * external_cond_mask: NULL or B bytes, non-zero => that video's pose embedding is zeroed.
 * Inputs are not modified. */
int dfot_uvit_forward(dfot_uvit_t h, const float* x, const float* noise_levels, const float* external_cond,
                      const uint8_t* external_cond_mask, float* out, int batch, void* stream);
/* The camera-pose conditioning of a window does not change across its DDIM steps (and is zero for masked
 * videos), and every FiLM projection is linear in it.  dfot_uvit_set_conditions computes the pose patch-embedding,
 * its pyramid and every block's FiLM projection of it ONCE; dfot_uvit_forward_cached then runs the backbone for the
 * cached conditions (batch must match).  dfot_uvit_forward == set_conditions + forward_cached. */
int dfot_uvit_set_conditions(dfot_uvit_t h, const float* external_cond, const uint8_t* external_cond_mask, int batch,
                             void* stream);
int dfot_uvit_forward_cached(dfot_uvit_t h, const float* x, const float* noise_levels, float* out, int batch,
                             void* stream);
/* The same with a per-frame flag (device uint8 [batch * T], or NULL = every frame): the rows of `out` of frames whose flag is 0 are not
 * computed (zeros are written).  The sampler's composition step (history_guidance.py:545-568, dfot_video.py:750-752) never reads the
 * model output of context tokens; past the last transformer block the U-Net works frame by frame (ResBlocks, up-convolutions, output
 * projection: u_vit3d.py:30-185), so those frames are skipped there.  Attention levels always run on every frame (context frames are keys). */
int dfot_uvit_forward_cached_live(dfot_uvit_t h, const float* x, const float* noise_levels, float* out, int batch,
                                  const uint8_t* live_frames, void* stream);
/* ... and with a second flag per frame (device uint8 [batch * T], or NULL = every frame is fresh): fresh_frames == 0 states that this frame's
 * input, noise level and conditioning are those of the PREVIOUS forward of this handle (same batch) -- a clean context frame of the
 * conditional History-Guidance branch keeps its value and its level over the DDIM steps of a window (dfot_video.py:682-752) -- so its
 * activations in the frame-local down path (ResBlock levels, Downsample convolutions) are still in the workspace and are not recomputed.
 * A frozen frame must be a dead frame (live_frames == 0) in this forward AND in the previous one -- the up path overwrites the level
 * buffers of live frames in place -- so live_frames may not be NULL (DFOT_ERR_ARG).  DFOT_ERR_STATE when the previous forward ran another
 * batch.  The flags are ignored (everything is recomputed) when an image of the third level has fewer rows than the largest GEMM tile
 * (resolution < 256).  Results are bit-identical to the full forward. */
int dfot_uvit_forward_cached_masks(dfot_uvit_t h, const float* x, const float* noise_levels, float* out, int batch,
                                   const uint8_t* live_frames, const uint8_t* fresh_frames, void* stream);
/* debug/parity taps: copy an internal activation after the last forward (fp32). names: "pose_emb0","down0","down1",
 * "down2","mid","up2","up1","up0" in the oracle's NCHW layout. */
int dfot_uvit_read_tap(dfot_uvit_t h, const char* name, float* out, size_t capacity_floats, void* stream);

/* ---- DiT3D backbone (Kinetics-600 path) --------------------------------------------------------------------- */
typedef struct dfot_dit_s* dfot_dit_t;

/* Hyper-parameters of dit3d (configurations/algorithm/backbone/dit3d.yaml:1-9 overridden by shortcut/DiT/XL.yaml and
 * dataset_experiment/kinetics_600_video_generation.yaml:22-23): variant "full", pos_emb_type "rope_3d", no external
 * condition, no causal mask. */
typedef struct {
  int32_t hidden_size;   /* 1152 */
  int32_t depth;         /* 28 */
  int32_t num_heads;     /* 16 (head dim 72) */
  int32_t patch_size;    /* 1 */
  int32_t in_channels;   /* 16 latent channels */
  int32_t height;        /* x_shape[1], 16 */
  int32_t width;         /* x_shape[2], 16 */
  int32_t max_tokens;    /* temporal length of the RoPE grid, 5 */
  int32_t mlp_hidden;    /* int(hidden*spatial_mlp_ratio); 0 = attention-only blocks (dit3d.yaml leaves it unset) */
  int32_t noise_dim;     /* 256 (sinusoidal Timesteps channels) */
  int32_t timesteps;     /* number of discrete noise levels, 1000 */
  float rope_theta;      /* 10000 */
  float eps;             /* 1e-6 (LayerNorm) */
  /* variant 0: DiT3D "full" + rope_3d (fields below ignored).
   * variant 1: DifferenceDiT3D "factorized_matrix_attention" + sinusoidal_2d, merge_type "interleaved" (the bash/k600 model,
   *   configurations/algorithm/backbone/difference_dit3d_factorized_matrix.yaml + shortcut/FacMatDiT/group_XL/XL-64-1.yaml):
   *   per depth one per-frame spatial DiTBlock (num_heads heads, MLP mlp_hidden) and one MatrixDiTBlock whose attention
   *   treats every frame as ONE token (a P x hidden matrix projected by left/right factors, dit_blocks.py:211-350);
   *   max_tokens counts the merged (difference, frame) tokens = 2 x the algorithm's max_tokens; hidden_size = embed_row_dim */
  int32_t variant;
  int32_t embed_col_dim;        /* 64 */
  int32_t num_col_heads;        /* 1 */
  int32_t num_row_heads;        /* 16 */
  int32_t temporal_mlp_hidden;  /* int(hidden * mlp_ratio) of the matrix blocks, 4608 */
  int32_t use_bias;             /* qkv_bias / proj_bias of MatrixAttention present */
} dfot_dit_config;

int dfot_dit_create(const dfot_dit_config* cfg, dfot_dit_t* out);
int dfot_dit_destroy(dfot_dit_t h);
int dfot_dit_num_params(dfot_dit_t h);
const char* dfot_dit_param_name(dfot_dit_t h, int index);
int dfot_dit_param_shape(dfot_dit_t h, int index, int64_t shape[4], int* ndim);
int dfot_dit_load_weight(dfot_dit_t h, const char* name, const float* data, const int64_t* shape, int ndim, void* stream);
/* verify every key was loaded; build the RoPE table and the noise-level modulation table: the conditioning of this
 * backbone is a function of the integer noise level alone, so every AdaLN shift/scale/gate of every block is
 * evaluated for all `timesteps` levels once (one GEMM) and forward only indexes it */
int dfot_dit_finalize(dfot_dit_t h, void* stream);
int dfot_dit_reserve(dfot_dit_t h, int max_batch);
size_t dfot_dit_workspace_bytes(dfot_dit_t h);
/* "gemm_variant" (-1 auto), "time_attn" (as for dfot_uvit_set_option) */
int dfot_dit_set_option(dfot_dit_t h, const char* key, int value);
int dfot_dit_attn_timing(dfot_dit_t h, double* total_ms, int64_t* launches);
/* out[B,T,C,H,W] = model(x[B,T,C,H,W], noise_levels[B,T]) ; x/out fp32, noise_levels int32 in [0, timesteps)
 * (device pointer; out-of-range levels are clamped), T <= max_tokens with T*(H/p)*(W/p) % 128 == 0.
 * variant 1: x holds the interleaved (difference_0, frame_0, difference_1, ...) tokens, T even, (H/p)*(W/p) % 128 == 0. */
int dfot_dit_forward(dfot_dit_t h, const float* x, const int32_t* noise_levels, float* out, int batch, int tokens,
                     void* stream);
/* parity taps after the last forward: "emb" [timesteps][hidden] (noise-level embedding of every level),
 * "stream" [B*T*P][hidden] (residual stream after the last block) */
int dfot_dit_read_tap(dfot_dit_t h, const char* name, float* out, size_t capacity_floats, void* stream);

/* ---- DiT3D training path ("full" variant, attention-only blocks) ----------------------------------
 * Replaces torch autograd through DiT3D.forward (algorithms/dfot/backbones/dit/dit3d.py:153-192, dit_blocks.py:408-542) and the
 * optimizer step of DFoTVideo.training_step / configure_optimizers (dfot_video.py:41-75, base_pytorch_algo: AdamW).
 * Parameters, gradients and optimizer moments live in caller-owned flat fp32 buffers (reference state_dict order, every
 * tensor 16-byte aligned at dfot_dit_train_param_offset) so a data-parallel job all-reduces ONE buffer.
 *   create -> attach(params, grads) -> [copy weights into params] -> reserve(batch) -> sync_weights
 *   step: forward(x_t, levels) -> loss + dfot_vloss_grad -> backward(d_out) -> [all-reduce grads] -> dfot_sumsq + dfot_adamw_step
 *         -> sync_weights */
typedef struct dfot_dit_train_s* dfot_dit_train_t;
int dfot_dit_train_create(const dfot_dit_config* cfg, dfot_dit_train_t* out);
int dfot_dit_train_destroy(dfot_dit_train_t h);
int dfot_dit_train_num_params(dfot_dit_train_t h);
const char* dfot_dit_train_param_name(dfot_dit_train_t h, int i);
int dfot_dit_train_param_shape(dfot_dit_train_t h, int i, int64_t shape[4], int* ndim);
int64_t dfot_dit_train_param_offset(dfot_dit_train_t h, int i);
int64_t dfot_dit_train_total_numel(dfot_dit_train_t h);
size_t dfot_dit_train_workspace_bytes(dfot_dit_train_t h);
int dfot_dit_train_attach(dfot_dit_train_t h, float* params, float* grads);
int dfot_dit_train_reserve(dfot_dit_train_t h, int max_batch);
int dfot_dit_train_sync_weights(dfot_dit_train_t h, void* stream);
/* out = model(x, levels), saving what backward needs; x must stay valid until backward */
int dfot_dit_train_forward(dfot_dit_train_t h, const float* x, const int32_t* noise_levels, float* out, int batch, int tokens, void* stream);
/* grads <- d(sum(out * d_out))/d(params) for the last forward (overwrites the attached gradient buffer) */
int dfot_dit_train_backward(dfot_dit_train_t h, const float* d_out, void* stream);
/* dx[B,T,C,H,W] <- d(sum(out * d_out))/d(x) of the same forward / backward pair (call after dfot_dit_train_backward): what autograd gives
 * `x.grad` when x requires it -- reconstruction guidance differentiates the prediction w.r.t. x_t (discrete_diffusion.py:485-513) */
int dfot_dit_train_input_grad(dfot_dit_train_t h, float* dx, void* stream);
/* dv[B,T,F] = coef[b,t] * d/dv of the loss term of dfot_vspace_loss (vspace = 1) / dfot_vpred_loss (vspace = 0) */
int dfot_vloss_grad(const float* x, const float* noise, const float* v, const float* a, const float* sigma, const float* coef, float* dv,
                    int batch, int tokens, int64_t frame_elems, int vspace, void* stream);
/* out[0] = sum x^2 (device scalar) */
int dfot_sumsq(const float* x, int64_t n, float* out, void* stream);
/* torch.optim.AdamW on flat buffers; grad_sumsq (optional device scalar): clip the gradient to max_grad_norm first;
 * ema (optional): shadow = ema_decay * shadow + (1 - ema_decay) * new parameter, in the same pass (algorithms/common/ema.py:22-32) */
int dfot_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, const float* grad_sumsq, float max_grad_norm, float* ema, float ema_decay,
                    void* stream);

/* ---- camera-pose front end ------------------------------------------------------------------- */
/* raw poses [B,T,16] (fx,fy,px,py | 3x4 RT) -> ray encoding [B,T,180,res,res] fp32, normalised by frame 0 */
int dfot_ray_encode(const float* raw_poses, float* out, int batch, int tokens, int resolution, void* stream);
/* same encoding for poses the caller has already expressed in its world frame (no normalisation by frame 0): the options of
 * DFoTVideoPose._process_conditions off the default path -- normalize_by "mean", `bound`, and the interpolation of masked poses
 * under temporal History Guidance (dfot_video_pose.py:75-98, utils/geometry_utils.py:135-205) -- are a few 3x3 products per
 * frame and are done by the host (diffusion-forcing-transformer_amd/pose.py) before this call */
int dfot_ray_encode_normalized(const float* poses, float* out, int batch, int tokens, int resolution, void* stream);

/* ---- sampler step (per-step coefficient tables live in device memory) -------------------------- */
/* coef tables are [n_branch_batch = B*NFE][T] fp32, row-major, for ONE step:
 *   qa,qb : x_in = qa*x + qb*noise      (history token re-noising; (1,0) = keep)
 * x[B,T,F] fp32, noise[B*NFE,T,F] fp32 (may be NULL when every qb is 0), x_in[B*NFE,T,F] fp32. */
int dfot_hg_prepare(const float* x, const float* noise, const float* qa, const float* qb, float* x_in, int batch,
                    int nfe, int tokens, int64_t frame_elems, void* stream);
/* per (branch-batch,t): sa=sqrt(abar_k), s1=sqrt(1-abar_k), an=sqrt(abar_next), cn=sqrt(1-abar_next-sigma^2),
 * keep (curr==next) as 0/1 floats; weight[NFE]; gen[B,T] (1 = token being generated, 0 = context/padding).
 *   x0 = sa*x_in - s1*v ; eps = sa*v + s1*x_in ; x_pred = keep ? x_in : x0*an + eps*cn
 *   x_next[b,t] = gen ? sum_h weight[h]*x_pred[b*NFE+h,t] : x[b,t] */
int dfot_ddim_compose(const float* x, const float* x_in, const float* v, const float* sa, const float* s1,
                      const float* an, const float* cn, const float* keep, const float* weight, const uint8_t* gen,
                      float* x_next, int batch, int nfe, int tokens, int64_t frame_elems, void* stream);

/* same with weight[NFE][T]: one composition weight per (branch, token) -- History Guidance with several gen segments
 * (history_guidance.py:545-568: excluded gen tokens contribute 0, the sum is divided by the number of segments covering the token) */
int dfot_ddim_compose_tokw(const float* x, const float* x_in, const float* v, const float* sa, const float* s1,
                           const float* an, const float* cn, const float* keep, const float* weight, const uint8_t* gen,
                           float* x_next, int batch, int nfe, int tokens, int64_t frame_elems, void* stream);

/* stochastic part of a sampling step -- the `sigma * noise` term of ddim_sample_step for eta > 0
 * (algorithms/dfot/diffusion/discrete_diffusion.py:468-472,515-527) and the posterior-variance term of ddpm_sample_step
 * (:441-449; the DDPM mean is the dfot_ddim_compose form with an = coef1 + coef2*sa, cn = coef2*s1).  Composition is linear:
 *   x_next[b,t] += sum_h weight[h (,t)] * sigma[b*NFE+h,t] * noise[b*NFE+h,t]     where gen[b,t]
 * sigma[B*NFE][T] fp32 with 0 for kept tokens; noise[B*NFE,T,F] fp32, clamped by the caller; weight_per_token selects weight[NFE][T]. */
int dfot_ddim_noise(const float* noise, const float* sigma, const float* weight, const uint8_t* gen, float* x_next, int batch,
                    int nfe, int tokens, int64_t frame_elems, int weight_per_token, void* stream);

/* ---- denoising loss of one noised forward (training_step / validation denoising loss) -------------------------- */
/* replaces ContinuousDiffusion.forward's frame arithmetic (diffusion/continuous_diffusion.py:140-167):
 *   x_t = alpha*x + sigma*noise ; eps_hat = alpha*v + sigma*x_t ; loss[b,t] = mean_frame( weight*(eps_hat-noise)^2 ) ;
 *   x_pred = alpha*x_t - sigma*v (optional, may be NULL).  alpha/sigma/weight are [B*T] fp32 (host-computed from the
 *   cosine logSNR schedule); scratch holds dfot_vpred_loss_scratch_floats() floats.  v = backbone(x_t, 0.125*logsnr, cond). */
int dfot_vpred_loss(const float* x, const float* noise, const float* v, const float* alpha, const float* sigma,
                    const float* weight, float* x_pred, float* scratch, float* loss, int batch, int tokens,
                    int64_t frame_elems, void* stream);
int64_t dfot_vpred_loss_scratch_floats(int batch, int tokens, int64_t frame_elems);
/* DiscreteDiffusion.forward, objective pred_v (diffusion/discrete_diffusion.py:345-377): same inputs, but the error is taken
 * in v-space: loss[b,t] = mean_frame( weight * (v - (alpha*noise - sigma*x))^2 ); alpha/sigma = sqrt(abar_k), sqrt(1-abar_k) of the
 * token's integer level, weight = compute_loss_weights (min-SNR / fused min-SNR / sigmoid, host-computed, :274-343) */
int dfot_vspace_loss(const float* x, const float* noise, const float* v, const float* alpha, const float* sigma,
                     const float* weight, float* x_pred, float* scratch, float* loss, int batch, int tokens,
                     int64_t frame_elems, void* stream);

/* ---- unit-testable primitives ------------------------------------------------------------------ */
/* C[M,N] (fp32) = A[M,K] (bf16, row stride lda) * W[N,K]^T (bf16) + bias[N] (fp32 or NULL)
 * variant: -1 auto (pick by shape), 0 = 128x128 tile, register staging (independent reference), 1 = 128x128 LDS-DMA, 4 = 256x256
 * (16 waves), 6 = 512x128 (16 waves), 8 = 128x128 with K split inside the workgroup, 9 = 256x192, 10 = 128x192, 14 = 256x144 three-stage
 * ring; the 256-row forms need M % 256 == 0 (512 for 6).  Other numbers: DFOT_ERR_ARG */
int dfot_op_gemm(const void* a_bf16, int lda, const void* w_bf16, const float* bias, float* c, int m, int n, int k,
                 int variant, void* stream);
/* y[BT,H,W,Cout] (fp32) = conv3x3(pad 1)(a[BT,H,W,Cin] bf16, w[Cout][9*Cin] bf16 tap-major) + bias */
int dfot_op_conv3x3(const void* a_bf16, const void* w_bf16, const float* bias, float* y, int bt, int h, int w,
                    int cin, int cout, int variant, void* stream);
/* o[B,N,heads*d] (bf16, row stride ldo) = softmax(q k^T) v ; q,k,v [B,heads,N,d] bf16; q pre-scaled by
 * log2(e)/sqrt(d) (the kernel works in the exp2 domain). d in {64,128}; N % 128 == 0 (d=64) or % 64. */
int dfot_op_attention(const void* q, const void* k, const void* v, void* o, int ldo, int batch, int heads, int n,
                      int d, int variant, void* stream);
/* same with a logical head dim d <= 128 (d % 4 == 0) stored in rows of dstride = (d <= 64 ? 64 : 128) elements whose
 * pad columns are zero; o[B,N,heads*d] is compact (row stride ldo) */
int dfot_op_attention_padded(const void* q, const void* k, const void* v, void* o, int ldo, int batch, int heads, int n,
                             int d, void* stream);
/* training path, test entry: o = attention(q, k, v) as above and, for the upstream gradient d_o [B*N][ldo] (same compact layout as
 * o), dq / dk / dv in the layout of q / k / v; dq is the gradient of the UNSCALED q (q itself is passed pre-multiplied by
 * log2(e)/sqrt(d), as the forward wants it).  Replaces torch autograd through F.scaled_dot_product_attention
 * (algorithms/dfot/backbones/dit/dit_blocks.py:21-44,100-123). */
int dfot_op_attention_bwd(const void* q, const void* k, const void* v, const void* d_o, void* o, int ldo, void* dq, void* dk, void* dv,
                          int batch, int heads, int n, int d, void* stream);
/* training path of the UViT3DPose ResBlocks / resamplers, test entry: gradients of y = conv3x3(x, w, padding 1) + b for the upstream
 * gradient dy: x [BT,H,W,Cin] and dy [BT,H,W,Cout] bf16 channels-last, w fp32 [Cout][Cin][3][3]; dx fp32 [BT,H,W,Cin], dw fp32 like w,
 * db fp32 [Cout].  Channel counts multiples of 64.  Replaces autograd through F.conv2d (u_vit_blocks.py:16-93). */
int dfot_op_conv3x3_bwd(const void* x, const void* dy, const float* w, float* dx, float* dw, float* db, int bt, int h, int w_, int cin, int cout,
                        void* stream);
/* the same with the data gradient in fp32 (dx) OR in bf16 (dx_bf; exactly one of the two): a gradient that only feeds the backward of the
 * GroupNorm in front of the convolution is written once, at half the bytes -- what torch.autocast(bf16) leaves there in the reference
 * (F.conv2d runs in bf16 under the trainer's precision setting, configurations/experiment/base_pytorch_exp.yaml:12 `precision: bf16`). */
int dfot_op_conv3x3_bwd2(const void* x, const void* dy, const float* w, float* dx, void* dx_bf, float* dw, float* db, int bt, int h, int w_, int cin,
                         int cout, void* stream);
/* test entry: backward of y = SiLU(FiLM(GroupNorm32(x))) (ResBlock in_layers: film NULL; out_norm: film [BT*P][2C] bf16 = scale | shift,
 * u_vit_blocks.py:57-93): x, dy, dx fp32 [BT][P][C]; dfilm bf16 like film; dgamma / dbeta fp32 [C] */
int dfot_op_gn_silu_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const void* film, float eps, float* dx, void* dfilm,
                        float* dgamma, float* dbeta, int bt, int pixels, int channels, void* stream);
/* test entries of the UViT TransformerBlock backward pieces (u_vit_blocks.py:96-116,192-281):
 * NormalizeWithCond: xn = RMSNorm(x; w) (1 + scale) + shift with film [rows][2C] bf16 = (scale | shift): dx fp32, dfilm bf16, dw fp32 [C] */
int dfot_op_rms_film_bwd(const float* x, const float* dxn, const float* w, const void* film, float eps, float* dx, void* dfilm, float* dw,
                         int64_t rows, int channels, int accumulate_dx, void* stream);
/* the same out of place: dx = dres + the norm's input gradient (dres is the gradient of the residual path, left untouched);
 * dx_bf (optional, bf16 [rows][channels]): a bf16 copy of dx, the GEMM operand of the block below */
int dfot_op_rms_film_bwd_res(const float* x, const float* dxn, const float* w, const void* film, float eps, const float* dres, float* dx, void* dx_bf,
                             void* dfilm, float* dw, int64_t rows, int channels, void* stream);
/* per-head q / k RMSNorm + RoPE (rope_cs [ntok][d/2][2] = cos, sin): fused [rows][ld] bf16 holds (q | k | v) head-major in its first 3C
 * columns, dq / dk / dv [B][heads][ntok][d] bf16 are the attention backward's outputs; writes dfused [rows][ldo] columns [0, 3C), dqw / dkw [d] */
int dfot_op_qknorm_rope_bwd(const void* fused, int ld, const void* dq, const void* dk, const void* dv, const float* qw, const float* kw,
                            const float* rope_cs, float eps, void* dfused, int ldo, float* dqw, float* dkw, int64_t rows, int ntok, int heads, int d,
                            void* stream);
/* token-axis weight-gradient GEMM: out [M][N] fp32 = a^T b, a [rows][lda] and b [rows][ldb] bf16 in the activations' own
 * (feature-contiguous) layout (either may be a column block of a wider matrix); M, N multiples of 8, rows of 64.
 * slices == 0: tile form (128x128 / 256x256 / 256x192 / 192x256, LDS-DMA staged) and K slices chosen by shape (wgrad_plan);
 * slices >= 1: the 128x128 form with that many K slices.  Partial buffers are summed inside. */
int dfot_op_wgrad_nt(const void* a, int lda, const void* b, int ldb, float* out, int m, int n, int64_t rows, int slices, void* stream);
/* op-level entry points a training driver composes (all on device pointers, bf16 activations unless noted):
 * out = a w^T (+ bias) in bf16 / fp32 (+ resid); transposes; column sums; the training-form forwards of the UViT TransformerBlock pieces
 * (values the backward needs are kept: no fused norm / activation epilogues); attention with its log-sum-exp and the backward from it */
int dfot_op_gemm_bf16(const void* a, int lda, const void* w, const float* bias, void* out, int ldo, int m, int n, int k, void* stream);
int dfot_op_gemm_f32(const void* a, int lda, const void* w, const float* bias, const float* resid, float* out, int ldo, int m, int n, int k,
                     void* stream);
int dfot_op_transpose_bf16(const void* src, void* dst, int rows, int cols, void* stream);
int dfot_op_colsum_bf16(const void* src, int ld, float* out, int64_t rows, int n, void* stream);
int dfot_op_rms_film_fwd(const float* x, const float* w, const void* film, float eps, void* out, int64_t rows, int channels, void* stream);
/* fused_attn_mlp_proj of a TransformerBlock in its training form, one launch (u_vit_blocks.py:253-262): the raw projection (bias added) is kept
 * as bf16 `fused` [rows][7C] for the backward, and the same epilogue writes q, k (per-head RMSNorm + RoPE; q * qscale), v as
 * [B][heads][ntok][d] and SiLU(mlp_h) into cat[:, ccol0 : ccol0 + 4C] (row stride ldcat) */
int dfot_op_fused_proj_train(const void* a, int lda, const void* w, const float* bias, const float* qw, const float* kw, const float* rope_cs, float eps,
                             float qscale, void* fused, void* q, void* k, void* v, void* cat, int ldcat, int ccol0, int64_t rows, int ntok, int heads,
                             int d, void* stream);
int dfot_op_qknorm_rope_fwd(const void* fused, int ld, const float* qw, const float* kw, const float* rope_cs, float eps, float qscale, void* q,
                            void* k, void* v, int64_t rows, int ntok, int heads, int d, void* stream);
int dfot_op_silu_cols(const void* src, int lds_, int scol0, const void* grad, int ldg, int gcol0, void* dst, int ldd, int dcol0, int64_t rows,
                      int ncols, void* stream);
/* the same with a caller-supplied bound of the scores in the log2 domain (< 64: the d = 64 launch may run without a running max; anything
 * else, NaN included, takes the running-max kernel).  scratch: device buffer of at least dfot_op_attention_scratch_bytes(...) bytes for the
 * fp32 partial rows of the key-split tail, owned by the caller (one per trainer / stream); null = a process-wide grow-only block */
size_t dfot_op_attention_scratch_bytes(int batch, int heads, int n, int d);
int dfot_op_attention_fwd_lse_bounded(const void* q, const void* k, const void* v, void* o, int ldo, float* lse, int batch, int heads, int n, int d,
                                      float score_bound, void* scratch, size_t scratch_bytes, void* stream);
int dfot_op_attention_fwd_lse(const void* q, const void* k, const void* v, void* o, int ldo, float* lse, int batch, int heads, int n, int d,
                              void* stream);
int dfot_op_attention_bwd_lse(const void* q, const void* k, const void* v, const void* o, const void* d_o, int ldo, const float* lse, float* delta,
                              void* dq, void* dk, void* dv, int batch, int heads, int n, int d, void* stream);
/* ResBlock / resampler / embedding pieces of the UViT training driver (channels-last fp32 streams, bf16 GEMM operands).
 * Shape contract of the vectorised kernels: GroupNorm entries take 128, 256, 512 or 1024 channels; pool2_bwd / upsample_bwd channels
 * % 4 == 0; frame sums / split_bf16 lengths % 8 == 0 (anything else returns DFOT_ERR_SHAPE). */
int dfot_op_gn_silu_fwd(const float* x, const float* gamma, const float* beta, const void* film, float eps, void* out, float* stats, int bt,
                        int pixels, int channels, void* stream);


/* backward of GroupNorm (+ FiLM) + SiLU with saved statistics, the upstream gradient dy in bf16 [BT][P][C] (dx_bf of dfot_op_conv3x3_bwd2):
 * dx = (dres ? dres : 0) + input gradient, as fp32 (dx) and / or bf16 (dx_bf); dfilm optional with its row stride dfilm_ld */
int dfot_op_gn_silu_bwd5(const float* x, const void* dy_bf, const float* stats, const float* gamma, const float* beta, const void* film, const float* dres,
                         float* dx, void* dx_bf, void* dfilm, int64_t dfilm_ld, float* dgamma, float* dbeta, int bt, int pixels, int channels,
                         void* stream);
/* Folded FiLM (training).  The reference computes emb = PatchEmbed(pose rays) keep + noise embedding per pixel (u_vit3d_pose.py:63-131,
 * embeddings.py:390-428), pools it down the levels and every block projects it: film = emb_layer(emb) (u_vit_blocks.py:57-116).  All of
 * these maps are linear, so film = (W_emb_layer W_patch) patches + [W_emb_layer (b_patch keep + noise_emb[frame]) + b]: the per-row part is a
 * GEMM over the 768-wide pose patches instead of the 1024-wide embedding, the per-frame part a [frames][2C] fp32 table added in that GEMM's
 * epilogue, and the backward never forms a per-row embedding gradient (uvit_train.py, UViT3DPoseTrainer.sync).
 * out bf16 [M][N] = a w^T + frame_bias[row / rows_per_frame][:] */
int dfot_op_gemm_bf16_frame_bias(const void* a, int lda, const void* w, const float* frame_bias, int rows_per_frame, void* out, int ldo, int m, int n, int k,
                                 void* stream);
/* GroupNorm + FiLM + SiLU with the block's (scale | shift) columns given as a column block of a level-wide matrix (row pitch film_ld >= 2C) */
int dfot_op_gn_silu_fwd2(const float* x, const float* gamma, const float* beta, const void* film, int64_t film_ld, float eps, void* out, float* stats,
                         int bt, int pixels, int channels, void* stream);
int dfot_op_gn_silu_bwd6(const float* x, const void* dy_bf, const float* stats, const float* gamma, const float* beta, const void* film,
                         int64_t film_ld, const float* dres, float* dx, void* dx_bf, void* dfilm, int64_t dfilm_ld, float* dgamma, float* dbeta, int bt,
                         int pixels, int channels, void* stream);
/* out [bt][n] fp32 = per-frame column sums of src bf16 [bt * pixels][ld] (the gradient of the per-frame FiLM table from the FiLM gradients) */
int dfot_op_frame_sums_bf16(const void* src, int64_t ld, float* out, int bt, int pixels, int n, void* stream);
/* x fp32 = hi + lo, both bf16 (lo carries the next 8 mantissa bits): three bf16 products Ah Bh + Ah Bl + Al Bh with fp32 accumulation
 * reproduce an fp32 product to ~2^-16 -- how the weight-sized products of the folded FiLM run on the matrix cores */
int dfot_op_split_bf16(const float* x, void* hi, void* lo, int64_t n, void* stream);
/* C[i][j] (+)= sum_k A[i * sa_i + k * sa_k] * B[k * sb_k + j * sb_j] in fp32 with element strides: the weight-sized products of the
 * folded FiLM (W_emb_layer W_patch and the gradients back to the two factors) */
int dfot_op_sgemm(const float* a, int64_t sa_i, int64_t sa_k, const float* b, int64_t sb_k, int64_t sb_j, float* c, int64_t ldc, int m, int n, int k,
                  int accumulate, void* stream);
int dfot_op_pack_conv3(const float* w, void* out, int co, int ci, int dgrad, void* stream);
int dfot_op_conv3x3_f32(const void* a, const void* w, const float* bias, const float* resid, float* y, int bt, int h, int w_, int cin, int cout,
                        void* stream);
int dfot_op_pool2_bf16(const float* x, void* out, int bt, int h, int w, int c, void* stream);
int dfot_op_pool2_bwd(const float* dp, float* dx, int bt, int h, int w, int c, void* stream);
int dfot_op_sub_bf16(const float* a, const float* b, void* out, int64_t n, void* stream);
int dfot_op_upsample_add(const float* t, const float* skip, float* out, int bt, int h, int w, int c, void* stream);
int dfot_op_upsample_bwd(const float* dy, float* ds, int bt, int h, int w, int c, void* stream);
int dfot_op_axpy(float* a, const float* b, float alpha, int64_t n, void* stream);
int dfot_op_mul_cols(void* dst, int ldd, int dcol0, const void* mask, int64_t rows, int ncols, void* stream);
int dfot_op_emb_pyramid(const void* emb0, void* emb1, void* emb2, void* emb3, int bt, int r0, int e, void* stream);
int dfot_op_cond_repack(const float* cond, void* a, int bt, int res, int cdim, int kpad, void* stream);
int dfot_op_embed_input(const float* x, const float* w, const float* b, float* out, int bt, int res, int cin, int c0, void* stream);
int dfot_op_embed_input_wgrad(const float* dx0, const float* x, float* dw, float* db, int bt, int res, int cin, int c0, int ps, void* stream);
/* gradient w.r.t. the backbone input x [BT][Cin][R][R] of the patch embedding (dx0 fp32 [pix][C0], w [C0][Cin][2][2]): the last step of
 * d prediction / d x_t, which reconstruction guidance needs (discrete_diffusion.py:485-513) */
int dfot_op_embed_input_dgrad(const float* dx0, const float* w, float* dx, int bt, int res, int cin, int c0, int ps, void* stream);
int dfot_op_project_output(const float* x0, const float* w, const float* b, float* out, int bt, int res, int c0, int cout, void* stream);
int dfot_op_outgrad_gather(const float* dout, void* dpatch, int bt, int res, int cout, int ps, void* stream);
/* fp32 <-> bf16 helpers for tests */
int dfot_op_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int dfot_op_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);

/* ---- VideoVAE decoder pieces (algorithms/vae/video_vae/model.py:130-281, algorithms/vae/common/modules/{conv,resnet,attention,
 * updownsample,normalize}.py; called from BaseVideoAlgo._decode, algorithms/common/base_pytorch_video_algo.py:600-629).  Channels-last
 * activations [B][T][H][W][C]; the 3x3(x3) convolutions and 1x1x1 projections go through dfot_op_conv3x3_f32 / dfot_op_gemm_*. ------- */
/* GroupNorm(32 groups, eps) over the `pixels` (= T*H*W of one video) positions of each of the bt items, then optional SiLU: fp32 -> bf16.
 * scratch: dfot_op_groupnorm_scratch_floats(bt, pixels) floats.  channels in {128, 256, 512, 1024}. */
int64_t dfot_op_groupnorm_scratch_floats(int bt, int pixels);
int dfot_op_groupnorm(const float* x, const float* gamma, const float* beta, float eps, void* out, float* scratch, int bt, int pixels,
                      int channels, int silu, void* stream);
/* out[b][t] = x[b][max(t - shift, 0)] (bf16, frame_elems per frame): the first-frame-replicating causal pad of PaddedConv3D (conv.py:104-109) */
int dfot_op_frame_shift(const void* x, void* out, int batch, int frames, int64_t frame_elems, int shift, void* stream);
/* fp32 [B][T][H][W][C] -> [B][T'][2H][2W][C]; mode 0: nearest in (H, W), T' = T (SpatialUpsample2x, updownsample.py:77-83); mode 1: the
 * causal trilinear upsample of Spatial2xTime2x3DUpsample (:143-150), T' = 1 + 2 (T - 1) */
int dfot_op_upsample3d(const float* x, float* out, int batch, int frames, int h, int w, int channels, int mode, void* stream);
/* probs[r][:] = softmax(scale * scores[r][:]) (fp32 -> bf16): the frame-wise single-head attention of AttnBlock3D (attention.py:127-129) */
int dfot_op_softmax_rows(const float* scores, void* probs, int64_t rows, int n, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DFOT_HIP_H_ */
