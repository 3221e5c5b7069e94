// HBM-bound kernels of the DFoT backbone and sampler step (launch wrappers).
#pragma once
#include "common.h"

namespace dfot {

// ---- embeddings ----
int launch_noise_emb(const float* k, const float* freqs, const float* phases, const float* w1, const float* b1,
                     const float* w2, const float* b2, float* hidden, float* out, int bt, int ndim, int e, hipStream_t s);
int launch_embed_input(const float* x, const float* w, const float* b, float* out, int bt, int res, int cin, int c0,
                       hipStream_t s);
int launch_cond_repack(const float* cond, bf16* a, int bt, int res, int cdim, int kpad, hipStream_t s);
int launch_emb_pyramid(const bf16* emb0, bf16* emb1, bf16* emb2, bf16* emb3, int bt, int r0, int e, hipStream_t s);
// live (optional, device uint8 [bt]): frames whose flag is 0 are SKIPPED by the kernels that take it -- nothing read, nothing written
// (project_output writes zeros) -- for frames whose output the caller discards (the sampler's context tokens), uvit.hip
int launch_project_output(const float* x0, const float* w, const float* b, float* out, int bt, int res, int c0, int cout,
                          hipStream_t s, const uint8_t* live = nullptr);
// ---- norms ----
// GroupNorm(32): partial[bt][nblk][32][2] = (sum, sumsq) per block of pixels (standalone kernels below, or the fused
// GEMM epilogue with nblk = pixels/64); launch_gn_finalize reduces them deterministically to stats[bt][32][2] = (mean, rstd)
int launch_gn_partial_f32(const float* x, float* partial, int bt, int pixels, int c, hipStream_t s);
int launch_gn_partial_bf16(const bf16* x, float* partial, int bt, int pixels, int c, hipStream_t s);
int launch_gn_finalize(const float* partial, float* stats, int bt, int nblk, int pixels, int c, float eps, hipStream_t s);
int gn_partial_blocks(int pixels);
int launch_gn_apply_silu(const float* x, const float* stats, const float* gamma, const float* beta, bf16* out, int bt,
                         int pixels, int c, hipStream_t s, const uint8_t* live = nullptr);
// FiLM with the per-window pose cache (see kernels.hip). One FilmChunk per 64 rows of every FiLM projection:
struct FilmChunk {
  const bf16* w;   // &W_film[row0][0], row stride = emb dim
  const float* b;  // &bias[row0]
  long out_off;    // sv element (bt, r) of this chunk lives at out_off + bt*rows + r
  int rows;        // 2C of the owning block
};
int launch_film_vec(const FilmChunk* table, int chunks, const float* nemb, float* sv, int bt, int e, hipStream_t s);
int launch_gn_film_silu(const bf16* h, const float* stats, const float* gamma, const float* beta, const bf16* fcache,
                        const float* sv, const uint8_t* cond_mask, bf16* out, int bt, int pixels, int c, int tokens,
                        hipStream_t s, const uint8_t* live = nullptr);
// a residual stream whose last out-projection is still two or three K-slice partials: x += bias[c] + s0 + s1 (+ s2) (run_tr_block, uvit.hip)
struct RmsPending {
  void* x;  // where the completed sum goes: fp32 (launch_rms_film) or bf16 (launch_rms_film_bf16), the type of the stream
  const float* bias;
  const float* s0;
  const float* s1;
  const float* s2;  // optional third slice
};
int launch_rms_film(const float* x, const float* w, const bf16* fcache, const float* sv, const uint8_t* cond_mask, bf16* out,
                    long m, int c, int rows_per_bt, int tokens, float eps, hipStream_t s, const RmsPending* pend = nullptr);
int launch_rms_film_bf16(const bf16* x, const float* w, const bf16* fcache, const float* sv, const uint8_t* cond_mask, bf16* out,
                         long m, int c, int rows_per_bt, int tokens, float eps, hipStream_t s, const RmsPending* pend = nullptr);

// ---- resampling / skips ----
int launch_pool2_bf16(const float* x, bf16* out, int bt, int h, int w, int c, hipStream_t s);
// the same kernels on a bf16 residual stream (the inference engine's ResBlock levels)
int launch_pool2_bf16_bf16in(const bf16* x, bf16* out, int bt, int h, int w, int c, hipStream_t s);
int launch_sub_bf16_bf16in(const bf16* a, const float* b, bf16* out, long n, hipStream_t s, const uint8_t* live = nullptr, long frame_elems = 0);
int launch_upsample_add_bf16(const float* t, const bf16* skip, bf16* out, int bt, int h, int w, int c, hipStream_t s, const uint8_t* live = nullptr);
int launch_gn_apply_silu_bf16in(const bf16* x, const float* stats, const float* gamma, const float* beta, bf16* out, int bt, int pixels, int c,
                                hipStream_t s, const uint8_t* live = nullptr);
int launch_embed_input_bf16(const float* x, const float* w, const float* b, bf16* out, int bt, int res, int cin, int c0, hipStream_t s);
int launch_project_output_bf16(const bf16* x0, const float* w, const float* b, float* out, int bt, int res, int c0, int cout, hipStream_t s,
                               const uint8_t* live = nullptr);
int launch_sub_bf16(const float* a, const float* b, bf16* out, long n, hipStream_t s, const uint8_t* live = nullptr, long frame_elems = 0);
int launch_upsample_add(const float* t, const float* skip, float* out, int bt, int h, int w, int c, hipStream_t s, const uint8_t* live = nullptr);
// ---- pose ----
int launch_ray_encode(const float* poses, float* out, int b, int t, int res, int normalized, hipStream_t s);
// ---- sampler ----
int launch_hg_prepare(const float* x, const float* noise, const float* qa, const float* qb, float* x_in, int batch,
                      int nfe, int tokens, long f, hipStream_t s);
int launch_ddim_compose(const float* x, const float* x_in, const float* v, const float* sa, const float* s1,
                        const float* an, const float* cn, const float* keep, const float* weight, const uint8_t* gen,
                        float* x_next, int batch, int nfe, int tokens, long f, bool weight_per_token, hipStream_t s);
int launch_ddim_noise(const float* noise, const float* sigma, const float* weight, const uint8_t* gen, float* x_next, int batch, int nfe,
                      int tokens, long f, bool weight_per_token, hipStream_t s);
// v-prediction loss: partial = scratch [bt][vloss_chunks(f)], loss[bt] = mean over the frame of w*(eps_hat-eps)^2
int vloss_chunks(long f);
int launch_vloss(const float* x, const float* noise, const float* v, const float* a, const float* sg, const float* w,
                 float* x_pred, float* partial, float* loss, int bt, long f, bool vspace, hipStream_t s);
// ---- weight-gradient GEMM over the token axis (wgrad.hip): out[slices][M][N] = partial sums of A^T B, A [rows][lda], B [rows][ldb] ----
int launch_wgrad_nt(const bf16* a, long lda, const bf16* b, long ldb, float* out, int m, int n, long rows, int slices, hipStream_t s, int img_h = 0,
                    int img_w = 0, int sdy = 0, int sdx = 0, int all_taps = 0);
// tile form (0: 128 x 128 / 4 waves; 1: 256 x 256 / 16; 2: 256 x 192 / 12; 3: 192 x 256 / 12) and K slices of one weight gradient
struct WgradPlan {
  int form, slices;
};
WgradPlan wgrad_plan(int m, int n, long rows, long max_slices);
int launch_wgrad_conv_taps(const bf16* dy, const bf16* x, float* out, int co, int ci, long pix, int slices, int img_h, int img_w, hipStream_t s);
int wgrad_conv_tiles(int co, int ci, int* target);
int launch_wgrad_nt_plan(const bf16* a, long lda, const bf16* b, long ldb, float* out, int m, int n, long rows, WgradPlan plan, hipStream_t s);
// ---- training: loss gradient, gradient norm, AdamW (flat fp32 buffers) ----
int launch_vloss_grad(const float* x, const float* noise, const float* v, const float* a, const float* sg, const float* coef, float* dv,
                      int bt, long f, bool vspace, hipStream_t s);
int launch_sumsq(const float* x, long n, float* out, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                 const float* sumsq, float max_norm, float* ema, float ema_decay, hipStream_t s);
// ---- casts / weight packing ----
int launch_f32_to_bf16(const float* src, bf16* dst, long n, hipStream_t s);
int launch_bf16_to_f32(const bf16* src, float* dst, long n, hipStream_t s);
// dst[r][col0 + k] (row stride ldd) = src[map ? map[r] : r][k] for k < K ; zero for K <= k < kpad
int launch_pack_rows(const float* src, bf16* dst, const int* map, int rows, int k, int kpad, long ldd, int col0,
                     hipStream_t s);
// conv weight [Co][Ci][3][3] fp32 -> [Co][(ky*3+kx)*Ci + ci] bf16
int launch_pack_conv3(const float* src, bf16* dst, int co, int ci, hipStream_t s);
int launch_gather_f32(const float* src, float* dst, const int* map, int n, hipStream_t s);
// activation layout helpers for taps: [BT][P][C] fp32 -> [BT][C][P] fp32
int launch_nhwc_to_nchw(const float* src, float* dst, int bt, int p, int c, hipStream_t s);
int launch_bf16_nhwc_to_nchw(const bf16* src, float* dst, int bt, int p, int c, hipStream_t s);

// fp32 partial rows (O | m, l) of the key-split tail of the level-2 attention kernels.  A backbone handle owns one, sized by
// attention_scratch_bytes() in its reserve(); nullptr = the process-wide op-level scratch (grow-only, never freed: attention_v3.hip)
struct AttnScratch {
  float* p = nullptr;
  size_t bytes = 0;
};
size_t attention_scratch_bytes(int batch, int heads, int n, int d);
int launch_attention(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, int d,
                     int variant, hipStream_t stream, AttnScratch* scratch = nullptr);
// attention_v3.hip: d = 64, 64 query rows per wave, balanced tail (key-split left-over tiles + merge); nomax = caller bounds |score|
int launch_attention_v3(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, bool nomax,
                        hipStream_t stream, AttnScratch* scratch = nullptr);
// attention_v5.hip: attention_v3's NOMAX kernel with the key loop software-pipelined at half-tile granularity inside each wave
int launch_attention_v5(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, hipStream_t stream,
                        AttnScratch* scratch = nullptr, float* lse = nullptr);  // lse: [B][heads][N] log2-domain log-sum-exp (training)
// shared by the two: balanced tail (left-over query tiles split over the key axis) + merge of the fp32 partials
struct AttnSplit {
  int tiles, full, rem, nsplit;
  int slots;  // resident workgroups of one round: wgs_per_cu x the device's CU count
};
AttnSplit attn_plan_split(int batch, int heads, int n, int qrows, int wgs_per_cu);
int attn_partials(const AttnSplit& sp, int qrows, float** po, float** pml, AttnScratch* scratch, int dcols = 64);
// attention_ks.hip: 128-element rows, 8 waves per workgroup with the key axis split inside the workgroup (launches of few query tiles)
bool attention_ks_applies(int batch, int heads, int n, int d);
size_t attention_ks_scratch_bytes(int batch, int heads, int n, int d);
int launch_attention_ks(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, hipStream_t stream,
                        AttnScratch* scratch = nullptr);
int attn_launch_merge(const AttnSplit& sp, int qrows, const float* po, const float* pml, bf16* o, long ldo, int n, int heads,
                      hipStream_t stream, float* lse = nullptr);
int attention_dstride(int d);
// lse (optional, training): [B][heads][N] fp32, log2-domain log-sum-exp of every query row
int launch_attention_padded(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, int d,
                            hipStream_t stream, float* lse = nullptr);
// ---- attention backward (attention_bwd.hip) ----
// delta[b][head][n] = sum_c d_o * o over the head's columns; o / d_o compact [B*N][ldo] (head hd at column hd*d), d % 8 == 0
int launch_attention_bwd_delta(const bf16* o, const bf16* d_o, long ldo, float* delta, int batch, int heads, int n, int d, hipStream_t s);
// q (pre-scaled as in the forward) / k / v / dq / dk / dv: [B][heads][N][dstride]; d_o compact; dq is the gradient of the UNSCALED q
int launch_attention_bwd(const bf16* q, const bf16* k, const bf16* v, const bf16* d_o, long ldo, const float* l2, const float* delta,
                         bf16* dq, bf16* dk, bf16* dv, int batch, int heads, int n, int d, hipStream_t s);

}  // namespace dfot
