// UViT3DPose backbone on MI355X: weight packing, workspace and forward orchestration.
// Mirrors the module tree / state-dict names of the reference
// (algorithms/dfot/backbones/u_vit/u_vit3d_pose.py:63-131, u_vit3d.py:30-185, u_vit_blocks.py).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "dfot_hip.h"
#include "gemm.h"
#include "kernels.h"

namespace dfot {

// ------------------------------------------------------------------------------------------
// error text
// ------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

int tuning_flag(const char* name, int dflt) {
  std::string key = std::string("DFOT_") + name;
  const char* v = getenv(key.c_str());
  return v ? atoi(v) : dflt;
}

// ------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------
struct ResW {
  int c = 0;
  bf16* fcache = nullptr;  // per-window cache of W_film * pose_emb, [BT*P][2C]
  long sv_off = 0;
  bf16 *w_film = nullptr, *w1 = nullptr, *w2 = nullptr;
  float *b_film_raw = nullptr, *b_film = nullptr, *g1 = nullptr, *be1 = nullptr, *bias1 = nullptr, *g2 = nullptr,
        *be2 = nullptr, *bias2 = nullptr;
};
struct TrW {
  int c = 0;
  bf16* fcache = nullptr;
  long sv_off = 0;
  bf16 *w_film = nullptr, *w_fused = nullptr, *w_out = nullptr;
  float *b_film_raw = nullptr, *b_film = nullptr, *nw = nullptr, *b_fused = nullptr, *qw = nullptr, *kw = nullptr,
        *b_attn = nullptr, *b_mlp = nullptr, *b_out = nullptr;
  // bound of |q.k| * log2(e)/sqrt(d) after the per-head RMSNorm of q and k (u_vit_blocks.py:257-259): sqrt(d) * max|w_q| * max|w_k|
  // * log2(e); set at finalize.  Small enough => softmax needs no running max (attention_v3.hip, NOMAX)
  float score_bound = INFINITY;
};
struct ConvW {
  int cin = 0, cout = 0;
  bf16* w = nullptr;
  float* b = nullptr;
};

struct Param {
  std::string name;
  std::vector<int64_t> shape;
  std::function<int(const float*, hipStream_t)> load;
  bool loaded = false;
};

}  // namespace dfot

using namespace dfot;

struct dfot_uvit_s {
  dfot_uvit_config cfg{};
  int T = 0, E = 0, heads = 0, r[4] = {0, 0, 0, 0}, ch[4] = {0, 0, 0, 0};
  int kpose = 0;  // padded K of the pose patch-embed GEMM
  std::vector<Param> params;
  std::map<std::string, int> index;
  std::vector<void*> owned;  // every hipMalloc'ed block

  // weights
  float *ne_freqs = nullptr, *ne_phases = nullptr, *ne_w1 = nullptr, *ne_b1 = nullptr, *ne_w2 = nullptr, *ne_b2 = nullptr;
  bf16* pose_w = nullptr;
  float* pose_b = nullptr;
  float *ein_w = nullptr, *ein_b = nullptr, *pout_w = nullptr, *pout_b = nullptr;
  std::vector<ResW> down_res[2], up_res[2];
  std::vector<TrW> down_tr, mid_tr, up_tr;
  ConvW down_conv[3], up_conv[3];  // up_conv[l] : level l+1 -> l
  int* film_map[4] = {nullptr, nullptr, nullptr, nullptr};
  float* rope_cs[4] = {nullptr, nullptr, nullptr, nullptr};
  bf16* zeros = nullptr;
  bool finalized = false;

  // workspace
  int max_batch = 0;
  size_t ws_bytes = 0;
  std::vector<void*> ws_owned;
  float *nemb = nullptr, *nhid = nullptr, *X[4] = {nullptr, nullptr, nullptr, nullptr}, *HSA[3] = {nullptr, nullptr, nullptr}, *tmp = nullptr,
        *gn_partial = nullptr, *gn_partial2 = nullptr, *gn_stats = nullptr, *sv = nullptr;
  int gn1_nblk = 0;  // > 0: gn_partial holds that many partial blocks per image for the next GroupNorm input
  FilmChunk* film_table = nullptr;
  int film_chunks = 0;
  uint8_t* cond_mask = nullptr;  // device copy of the external_cond_mask of the cached conditions
  bool have_mask = false;
  int cond_batch = 0;            // batch the pose caches were built for (0 = none)
  bf16 *acond = nullptr, *emb[4] = {nullptr, nullptr, nullptr, nullptr}, *s1 = nullptr, *hbf = nullptr,
       *cat = nullptr, *q = nullptr, *k = nullptr, *v = nullptr;
  AttnScratch attn_scratch;   // key-split partials of the level-2 attention, owned by this handle (sized in reserve)
  const float* pend_part = nullptr;  // where the pending slices are
  float* out_part = nullptr;  // two fp32 partial slices of an out-projection (K split, see run_tr_block)
  size_t out_part_elems = 0;  // its capacity in floats: the split path is taken only when 2 * M * N fits
  // the slices of the last out-projection not yet added to X[pend_lvl] (pend_bias != nullptr): the next block's norm kernel adds
  // them while it reads the stream anyway; flush_pending() does it for every other reader
  const float* pend_bias = nullptr;
  int pend_lvl = 0, pend_c = 0, pend_slices = 2;
  long pend_m = 0;
  int last_batch = 0;
  // where the residual stream of a level currently lives: X[l], or HSA[l-1] right after the Downsample convolution (its output is both
  // the skip tensor and the next level's input: the first block of the level reads it there and writes X[l], no copy)
  const float* xin[4] = {nullptr, nullptr, nullptr, nullptr};
  // The residual stream between the blocks of a level is bf16 (XB[l]).  ResBlock levels (0, 1) -- what torch.autocast(bf16) keeps there in the reference: every
  // block's second convolution adds the residual in its bf16 epilogue.  A level is ENTERED from an fp32 tensor (level 1: the Downsample
  // output HSA[0], which stays fp32 as skip tensor) or from XB[l] itself (level 0: the patch embedding writes bf16; up path: upsample_add).
  // xin_bf[l]: the level's stream currently lives in XB[l] (xin[l] is then unused).
  // Transformer levels (2, 3): the same -- the out-projection adds the residual in its bf16 epilogue (level 2) or leaves K-slice
  // partials that the next block's norm kernel adds into the bf16 stream (level 3); a level entered from the fp32 Downsample output is
  // cast once.
  bf16* XB[4] = {nullptr, nullptr, nullptr, nullptr};
  bool xin_bf[4] = {false, false, false, false};
  int gemm_variant = GEMM_AUTO;
  int attn_variant = 2;
  bool attn_force_safe = false;  // level-2 attention: always the running-max kernel (what weights with a bound >= 64 get)
  // optional in-run timing of the level-2 attention launches (HIP events on the launch stream)
  bool time_attn = false;
  std::vector<hipEvent_t> ev_start, ev_stop;
  size_t ev_used = 0;
};

namespace dfot {

template <typename T>
static int dev_alloc(dfot_uvit_s* h, T** out, size_t count, bool workspace = false) {
  void* p = nullptr;
  const size_t bytes = count * sizeof(T);
  DFOT_CHECK_HIP(hipMalloc(&p, bytes ? bytes : 16));
  (workspace ? h->ws_owned : h->owned).push_back(p);
  if (workspace) h->ws_bytes += bytes;
  *out = reinterpret_cast<T*>(p);
  return DFOT_OK;
}

static int copy_f32(float* dst, const float* src, size_t n, hipStream_t s) {
  DFOT_CHECK_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return DFOT_OK;
}

static void add_param(dfot_uvit_s* h, const std::string& name, std::vector<int64_t> shape,
                      std::function<int(const float*, hipStream_t)> load) {
  h->index[name] = (int)h->params.size();
  h->params.push_back(Param{name, std::move(shape), std::move(load), false});
}

// plain fp32 tensor kept as-is
static int add_f32(dfot_uvit_s* h, const std::string& name, std::vector<int64_t> shape, float** dst) {
  size_t n = 1;
  for (auto d : shape) n *= (size_t)d;
  int rc = dev_alloc(h, dst, n);
  if (rc) return rc;
  float* d = *dst;
  add_param(h, name, shape, [d, n](const float* src, hipStream_t s) { return copy_f32(d, src, n, s); });
  return DFOT_OK;
}

static int add_res_block(dfot_uvit_s* h, const std::string& pre, int c, ResW* w) {
  const int e = h->E;
  w->c = c;
  int rc = 0;
  if ((rc = dev_alloc(h, &w->w_film, (size_t)2 * c * e))) return rc;
  if ((rc = dev_alloc(h, &w->b_film_raw, (size_t)2 * c))) return rc;
  if ((rc = dev_alloc(h, &w->b_film, (size_t)2 * c))) return rc;
  if ((rc = dev_alloc(h, &w->w1, (size_t)c * 9 * c))) return rc;
  if ((rc = dev_alloc(h, &w->w2, (size_t)c * 9 * c))) return rc;
  const int lvl = c == h->ch[0] ? 0 : 1;
  int* map = h->film_map[lvl];
  bf16* wf = w->w_film;
  add_param(h, pre + ".emb_layer.weight", {2 * c, e, 1, 1}, [=](const float* src, hipStream_t s) {
    return launch_pack_rows(src, wf, map, 2 * c, e, e, e, 0, s);
  });
  float *braw = w->b_film_raw, *bre = w->b_film;
  add_param(h, pre + ".emb_layer.bias", {2 * c}, [=](const float* src, hipStream_t s) {
    int r2 = copy_f32(braw, src, 2 * c, s);
    return r2 ? r2 : launch_gather_f32(braw, bre, map, 2 * c, s);
  });
  if ((rc = add_f32(h, pre + ".in_layers.0.weight", {c}, &w->g1))) return rc;
  if ((rc = add_f32(h, pre + ".in_layers.0.bias", {c}, &w->be1))) return rc;
  bf16* w1 = w->w1;
  add_param(h, pre + ".in_layers.2.weight", {c, c, 3, 3}, [=](const float* src, hipStream_t s) { return launch_pack_conv3(src, w1, c, c, s); });
  if ((rc = add_f32(h, pre + ".in_layers.2.bias", {c}, &w->bias1))) return rc;
  if ((rc = add_f32(h, pre + ".out_norm.weight", {c}, &w->g2))) return rc;
  if ((rc = add_f32(h, pre + ".out_norm.bias", {c}, &w->be2))) return rc;
  bf16* w2 = w->w2;
  add_param(h, pre + ".out_rest.1.weight", {c, c, 3, 3}, [=](const float* src, hipStream_t s) { return launch_pack_conv3(src, w2, c, c, s); });
  if ((rc = add_f32(h, pre + ".out_rest.1.bias", {c}, &w->bias2))) return rc;
  return DFOT_OK;
}

static int add_tr_block(dfot_uvit_s* h, const std::string& pre, int lvl, TrW* w) {
  const int e = h->E, c = h->ch[lvl], d = c / h->heads;
  w->c = c;
  int rc = 0;
  if ((rc = dev_alloc(h, &w->w_film, (size_t)2 * c * e))) return rc;
  if ((rc = dev_alloc(h, &w->b_film_raw, (size_t)2 * c))) return rc;
  if ((rc = dev_alloc(h, &w->b_film, (size_t)2 * c))) return rc;
  if ((rc = dev_alloc(h, &w->w_fused, (size_t)7 * c * c))) return rc;
  if ((rc = dev_alloc(h, &w->w_out, (size_t)c * 5 * c))) return rc;
  if ((rc = dev_alloc(h, &w->b_out, (size_t)c))) return rc;
  int* map = h->film_map[lvl];
  bf16* wf = w->w_film;
  add_param(h, pre + ".norm.emb_layer.weight", {2 * c, e}, [=](const float* src, hipStream_t s) {
    return launch_pack_rows(src, wf, map, 2 * c, e, e, e, 0, s);
  });
  float *braw = w->b_film_raw, *bre = w->b_film;
  add_param(h, pre + ".norm.emb_layer.bias", {2 * c}, [=](const float* src, hipStream_t s) {
    int r2 = copy_f32(braw, src, 2 * c, s);
    return r2 ? r2 : launch_gather_f32(braw, bre, map, 2 * c, s);
  });
  if ((rc = add_f32(h, pre + ".norm.norm.weight", {c}, &w->nw))) return rc;
  bf16* wfu = w->w_fused;
  add_param(h, pre + ".fused_attn_mlp_proj.weight", {7 * c, c}, [=](const float* src, hipStream_t s) {
    return launch_pack_rows(src, wfu, nullptr, 7 * c, c, c, c, 0, s);
  });
  if ((rc = add_f32(h, pre + ".fused_attn_mlp_proj.bias", {7 * c}, &w->b_fused))) return rc;
  if ((rc = add_f32(h, pre + ".q_norm.weight", {d}, &w->qw))) return rc;
  if ((rc = add_f32(h, pre + ".k_norm.weight", {d}, &w->kw))) return rc;
  bf16* wo = w->w_out;
  add_param(h, pre + ".attn_out.weight", {c, c}, [=](const float* src, hipStream_t s) {
    return launch_pack_rows(src, wo, nullptr, c, c, c, 5 * c, 0, s);
  });
  if ((rc = add_f32(h, pre + ".attn_out.bias", {c}, &w->b_attn))) return rc;
  add_param(h, pre + ".mlp_out.2.weight", {c, 4 * c}, [=](const float* src, hipStream_t s) {
    return launch_pack_rows(src, wo, nullptr, c, 4 * c, 4 * c, 5 * c, c, s);
  });
  if ((rc = add_f32(h, pre + ".mlp_out.2.bias", {c}, &w->b_mlp))) return rc;
  return DFOT_OK;
}

static int add_conv(dfot_uvit_s* h, const std::string& pre, int cin, int cout, ConvW* w) {
  w->cin = cin;
  w->cout = cout;
  int rc = 0;
  if ((rc = dev_alloc(h, &w->w, (size_t)cout * 9 * cin))) return rc;
  bf16* dst = w->w;
  add_param(h, pre + ".weight", {cout, cin, 3, 3}, [=](const float* src, hipStream_t s) { return launch_pack_conv3(src, dst, cout, cin, s); });
  return add_f32(h, pre + ".bias", {cout}, &w->b);
}

static int build(dfot_uvit_s* h) {
  const dfot_uvit_config& c = h->cfg;
  h->T = c.max_tokens;
  h->E = c.emb_channels;
  h->heads = c.num_heads;
  for (int l = 0; l < 4; ++l) {
    h->ch[l] = c.channels[l];
    h->r[l] = c.resolution / 2 / (1 << l);
  }
  h->kpose = ((c.cond_dim * 4 + 63) / 64) * 64;
  int rc = 0;
  // FiLM row maps: within every 64-column group, 32 scale rows then the 32 matching shift rows
  for (int l = 0; l < 4; ++l) {
    const int cc = h->ch[l];
    std::vector<int> m(2 * cc);
    for (int n = 0; n < 2 * cc; ++n) {
      const int g = n / 64, t = n % 64;
      m[n] = t < 32 ? 32 * g + t : cc + 32 * g + (t - 32);
    }
    if ((rc = dev_alloc(h, &h->film_map[l], (size_t)2 * cc))) return rc;
    DFOT_CHECK_HIP(hipMemcpy(h->film_map[l], m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  if ((rc = dev_alloc(h, &h->zeros, 256))) return rc;
  DFOT_CHECK_HIP(hipMemset(h->zeros, 0, 256 * sizeof(bf16)));

  const int e = h->E, nd = c.noise_dim;
  const std::string ne = "noise_level_pos_embedding.";
  if ((rc = add_f32(h, ne + "timesteps.freqs", {nd}, &h->ne_freqs))) return rc;
  if ((rc = add_f32(h, ne + "timesteps.phases", {nd}, &h->ne_phases))) return rc;
  if ((rc = add_f32(h, ne + "embedding.linear_1.weight", {e, nd}, &h->ne_w1))) return rc;
  if ((rc = add_f32(h, ne + "embedding.linear_1.bias", {e}, &h->ne_b1))) return rc;
  if ((rc = add_f32(h, ne + "embedding.linear_2.weight", {e, e}, &h->ne_w2))) return rc;
  if ((rc = add_f32(h, ne + "embedding.linear_2.bias", {e}, &h->ne_b2))) return rc;
  if ((rc = dev_alloc(h, &h->pose_w, (size_t)e * h->kpose))) return rc;
  {
    bf16* pw = h->pose_w;
    const int k = c.cond_dim * 4, kp = h->kpose;
    add_param(h, "external_cond_embedding.patch_embedder.proj.weight", {e, c.cond_dim, 2, 2},
              [=](const float* src, hipStream_t s) { return launch_pack_rows(src, pw, nullptr, e, k, kp, kp, 0, s); });
  }
  if ((rc = add_f32(h, "external_cond_embedding.patch_embedder.proj.bias", {e}, &h->pose_b))) return rc;
  if ((rc = add_f32(h, "embed_input.proj.weight", {h->ch[0], c.in_channels, 2, 2}, &h->ein_w))) return rc;
  if ((rc = add_f32(h, "embed_input.proj.bias", {h->ch[0]}, &h->ein_b))) return rc;
  if ((rc = add_f32(h, "project_output.proj.weight", {h->ch[0], c.in_channels, 2, 2}, &h->pout_w))) return rc;
  if ((rc = add_f32(h, "project_output.proj.bias", {c.in_channels}, &h->pout_b))) return rc;

  for (int l = 0; l < 3; ++l) {
    const int n = c.num_updown_blocks[l];
    const std::string pre = "down_blocks." + std::to_string(l) + ".";
    if (l < 2) {
      h->down_res[l].resize(n);
      for (int i = 0; i < n; ++i)
        if ((rc = add_res_block(h, pre + std::to_string(i), h->ch[l], &h->down_res[l][i]))) return rc;
    } else {
      h->down_tr.resize(n);
      for (int i = 0; i < n; ++i)
        if ((rc = add_tr_block(h, pre + std::to_string(i), l, &h->down_tr[i]))) return rc;
    }
    if ((rc = add_conv(h, pre + std::to_string(n) + ".conv", h->ch[l], h->ch[l + 1], &h->down_conv[l]))) return rc;
  }
  h->mid_tr.resize(c.num_mid_blocks);
  for (int i = 0; i < c.num_mid_blocks; ++i)
    if ((rc = add_tr_block(h, "mid_blocks." + std::to_string(i), 3, &h->mid_tr[i]))) return rc;
  for (int j = 0; j < 3; ++j) {
    const int l = 2 - j;
    const int n = c.num_updown_blocks[l];
    const std::string pre = "up_blocks." + std::to_string(j) + ".";
    if ((rc = add_conv(h, pre + "0.conv", h->ch[l + 1], h->ch[l], &h->up_conv[l]))) return rc;
    if (l < 2) {
      h->up_res[l].resize(n);
      for (int i = 0; i < n; ++i)
        if ((rc = add_res_block(h, pre + std::to_string(i + 1), h->ch[l], &h->up_res[l][i]))) return rc;
    } else {
      h->up_tr.resize(n);
      for (int i = 0; i < n; ++i)
        if ((rc = add_tr_block(h, pre + std::to_string(i + 1), l, &h->up_tr[i]))) return rc;
    }
  }
  return DFOT_OK;
}

// RoPE-3D angle table for one level: [N][d/2][2] = (cos, sin); axis split of the head dim follows
// RotaryEmbedding3D (embeddings.py:251-277), angle = position * theta^(-2j/dim_axis)
static int build_rope(dfot_uvit_s* h, int lvl) {
  const int d = h->ch[lvl] / h->heads, half = d / 2;
  const int q = half / 3, rem = half % 3;
  int parts[3] = {q, q, q};
  if (rem == 1) parts[0] = q + 1;
  if (rem == 2) parts[1] = parts[2] = q + 1;
  const int rr = h->r[lvl], n = h->T * rr * rr;
  std::vector<float> cs((size_t)n * half * 2);
  for (int tok = 0; tok < n; ++tok) {
    const int pos[3] = {tok / (rr * rr), (tok / rr) % rr, tok % rr};
    int pair = 0;
    for (int ax = 0; ax < 3; ++ax) {
      const int dim = 2 * parts[ax];
      for (int j = 0; j < parts[ax]; ++j, ++pair) {
        const float inv = 1.0f / powf(h->cfg.rope_theta, (float)(2 * j) / (float)dim);
        const float ang = (float)pos[ax] * inv;
        cs[((size_t)tok * half + pair) * 2 + 0] = cosf(ang);
        cs[((size_t)tok * half + pair) * 2 + 1] = sinf(ang);
      }
    }
  }
  int rc = dev_alloc(h, &h->rope_cs[lvl], cs.size());
  if (rc) return rc;
  DFOT_CHECK_HIP(hipMemcpy(h->rope_cs[lvl], cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
  return DFOT_OK;
}

__global__ void add_vec_kernel(const float* a, const float* b, float* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}

// ------------------------------------------------------------------------------------------
// forward pieces
// ------------------------------------------------------------------------------------------
// GroupNorm statistics are produced by whoever writes the tensor: the 3x3-conv GEMM epilogues emit per-64-row partial
// sums (conv1 -> statistics of h; conv2 / Downsample conv -> statistics of the next block's input); only the first block
// after embed_input / upsample_add needs the standalone partial kernel.  h->gn1_nblk > 0 means gn_partial already holds
// the partials of X[lvl].
static int run_res_block(dfot_uvit_s* h, const ResW& w, int lvl, int bt, hipStream_t s, const uint8_t* live = nullptr) {
  const int c = w.c, rr = h->r[lvl], pix = rr * rr;
  const int m = bt * pix;
  const int slots = pix / 64;
  bf16* xb = h->XB[lvl];
  int rc = 0;
  if (!h->xin_bf[lvl]) {
    // the level is entered from an fp32 tensor (level 1: the Downsample output, which stays fp32 as the skip tensor): one cast into the
    // bf16 stream (67 MB read at level 1 of config 2, once per forward).  The GroupNorm statistics the convolution's epilogue left
    // (gn1_nblk > 0) are those of the fp32 values: the same numbers up to the bf16 rounding of the elements.  Frozen frames are cast too
    // (their rows of this level's stream are never read again: frozen frames are dead frames, and the Downsample skips their tiles)
    if ((rc = launch_f32_to_bf16(h->xin[lvl], xb, (long)m * c, s))) return rc;
    h->xin_bf[lvl] = true;
  }
  if (h->gn1_nblk == 0) {
    if ((rc = launch_gn_partial_bf16(xb, h->gn_partial, bt, pix, c, s))) return rc;
    h->gn1_nblk = gn_partial_blocks(pix);
  }
  if ((rc = launch_gn_finalize(h->gn_partial, h->gn_stats, bt, h->gn1_nblk, pix, c, h->cfg.eps, s))) return rc;
  h->gn1_nblk = 0;
  if ((rc = launch_gn_apply_silu_bf16in(xb, h->gn_stats, w.g1, w.be1, h->s1, bt, pix, c, s, live))) return rc;
  GemmArgs g;
  g.A = h->s1; g.W = w.w1; g.M = m; g.N = c; g.K = 9 * c; g.H = rr; g.Wd = rr; g.Cin = c; g.zeros = h->zeros;
  g.bias = w.bias1; g.out_bf16 = h->hbf; g.ldo = c; g.live = live;
  g.gn_part = h->gn_partial2; g.gn_rows_per_bt = pix; g.gn_cpg = c / 32;
  if ((rc = launch_gemm(A_CONV3, E_BF16, h->gemm_variant, g, s))) return rc;
  if ((rc = launch_gn_finalize(h->gn_partial2, h->gn_stats, bt, slots, pix, c, h->cfg.eps, s))) return rc;
  if ((rc = launch_gn_film_silu(h->hbf, h->gn_stats, w.g2, w.be2, w.fcache, h->sv + w.sv_off,
                                h->have_mask ? h->cond_mask : nullptr, h->s1, bt, pix, c, h->T, s, live)))
    return rc;
  GemmArgs o;
  o.A = h->s1; o.W = w.w2; o.M = m; o.N = c; o.K = 9 * c; o.H = rr; o.Wd = rr; o.Cin = c; o.zeros = h->zeros;
  o.bias = w.bias2; o.out_bf16 = xb; o.ldo = c; o.live = live;
  o.resid_bf = xb;  // in place on the bf16 stream: a thread reads its elements before it writes them
  o.gn_part = h->gn_partial; o.gn_rows_per_bt = pix; o.gn_cpg = c / 32;
  if ((rc = launch_gemm(A_CONV3, E_BF16, h->gemm_variant, o, s))) return rc;
  h->xin_bf[lvl] = true;
  h->gn1_nblk = slots;
  return DFOT_OK;
}

// x += bias + slice0 + slice1 (+ slice2) on the bf16 stream (fp32 sums, 4 elements per thread): the reduce pass of the K-sliced
// out-projection where no norm kernel follows to do it (end of a level)
__global__ void out_reduce_kernel(bf16* x, const float* __restrict__ bias, const float* __restrict__ s0, const float* __restrict__ s1,
                                  const float* __restrict__ s2, long total4, int cq) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  f4 b = reinterpret_cast<const f4*>(bias)[i % cq] + reinterpret_cast<const f4*>(s0)[i] + reinterpret_cast<const f4*>(s1)[i];
  if (s2) b += reinterpret_cast<const f4*>(s2)[i];
  const bf16x4 xv = reinterpret_cast<const bf16x4*>(x)[i];
  reinterpret_cast<bf16x4*>(x)[i] = bf16x4{f2bf(bf2f(xv[0]) + b[0]), f2bf(bf2f(xv[1]) + b[1]), f2bf(bf2f(xv[2]) + b[2]), f2bf(bf2f(xv[3]) + b[3])};
}
// x (bf16 stream) += y (fp32): the fallback of the out-projection when a forced GEMM tile form has no bf16 residual epilogue
__global__ void add_f32_into_bf16_kernel(bf16* x, const float* __restrict__ y, long total4) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const f4 b = reinterpret_cast<const f4*>(y)[i];
  const bf16x4 xv = reinterpret_cast<const bf16x4*>(x)[i];
  reinterpret_cast<bf16x4*>(x)[i] = bf16x4{f2bf(bf2f(xv[0]) + b[0]), f2bf(bf2f(xv[1]) + b[1]), f2bf(bf2f(xv[2]) + b[2]), f2bf(bf2f(xv[3]) + b[3])};
}

static void swap_out_part(dfot_uvit_s* h) { h->pend_part = h->out_part; }

static int flush_pending(dfot_uvit_s* h, hipStream_t s) {
  if (!h->pend_bias) return DFOT_OK;
  const long total4 = h->pend_m * h->pend_c / 4;
  hipLaunchKernelGGL(out_reduce_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, s, h->XB[h->pend_lvl], h->pend_bias, h->pend_part,
                     h->pend_part + h->pend_m * h->pend_c, h->pend_slices == 3 ? h->pend_part + 2 * h->pend_m * h->pend_c : nullptr, total4,
                     h->pend_c / 4);
  h->pend_bias = nullptr;
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

static int run_tr_block(dfot_uvit_s* h, const TrW& w, int lvl, int batch, hipStream_t s) {
  const int c = w.c, rr = h->r[lvl], n = h->T * rr * rr, d = c / h->heads;
  const int m = batch * n;
  bf16* xb = h->XB[lvl];
  int rc = 0;
  if (h->pend_bias && (h->pend_lvl != lvl || h->pend_m != m || h->pend_c != c) && (rc = flush_pending(h, s))) return rc;
  if (!h->xin_bf[lvl]) {  // the level is entered from the fp32 Downsample output (which stays fp32 as the skip tensor): one cast
    if ((rc = launch_f32_to_bf16(h->xin[lvl], xb, (long)m * c, s))) return rc;
    h->xin_bf[lvl] = true;
  }
  RmsPending pend{xb, h->pend_bias, h->pend_part, h->pend_part + (long)m * c, h->pend_slices == 3 ? h->pend_part + 2L * m * c : nullptr};
  if ((rc = launch_rms_film_bf16(xb, w.nw, w.fcache, h->sv + w.sv_off, h->have_mask ? h->cond_mask : nullptr, h->s1, m, c, rr * rr, h->T,
                                 h->cfg.eps, s, h->pend_bias ? &pend : nullptr)))
    return rc;
  h->pend_bias = nullptr;
  GemmArgs p;
  p.A = h->s1; p.lda = c; p.W = w.w_fused; p.M = m; p.N = 7 * c; p.K = c; p.bias = w.b_fused;
  p.out2 = h->cat + c; p.ldo2 = 5 * c; p.split = 3 * c;
  p.q = h->q; p.k = h->k; p.v = h->v; p.qw = w.qw; p.kw = w.kw; p.rope_cs = h->rope_cs[lvl]; p.heads = h->heads; p.d = d;
  p.ntok = n; p.qscale = 1.4426950408889634f / sqrtf((float)d); p.eps = h->cfg.eps;
  // (d = 64, N = 21 x 192: persistent 256x192 tiles win in isolation, 109.5 vs 116.8 us with a plain epilogue, and lose inside the model,
  // 9.72 vs 9.79 frames/s: the picker's 256x256 tiles stay)
  if ((rc = launch_gemm(A_DENSE, E_QKV, h->gemm_variant, p, s))) return rc;
  const bool timed = h->time_attn && lvl == 2 && h->ev_used < h->ev_start.size();
  if (timed) DFOT_CHECK_HIP(hipEventRecord(h->ev_start[h->ev_used], s));
  // default (2): level 2 (d = 64) runs the 64-rows-per-wave kernel with the balanced tail; without a running max when the
  // QK-norm weights bound the scores far inside the fp32 / bf16 exponent range (2^64 * N keys << 2^127), else with it
  int av = h->attn_variant;
  if (av == 2 && d == 64 && n % 256 == 0) av = (w.score_bound < 64.0f && !h->attn_force_safe) ? 14 : 5;  // 14: attention_v5.hip (software-pipelined, no running max)
  if ((rc = launch_attention(h->q, h->k, h->v, h->cat, 5 * c, batch, h->heads, n, d, av, s, &h->attn_scratch))) return rc;
  if (timed) DFOT_CHECK_HIP(hipEventRecord(h->ev_stop[h->ev_used++], s));
  GemmArgs o;
  o.A = h->cat; o.lda = 5 * c; o.W = w.w_out; o.M = m; o.N = c; o.K = 5 * c; o.bias = w.b_out; o.ldo = c;
  // level 3 at small model batch: 256x144 tiles give M/256 x N/144 = 128 workgroups for 256 CUs; two K slices into partial buffers
  // make it 256 on the THREE-stage 256x144 ring (150 KB of LDS: the long-K loop no longer waits on the single k-tile a two-stage loop
  // has in flight), and the slices + bias are added into the fp32 residual stream by the NEXT block's norm kernel (deferred: pend_*).
  // Measured alternatives, dropped: one GEMM with the residual epilogue; fp32-atomic split-K (8.30 -> 6.61 frames/s); 256x256 tiles x
  // three K slices (240 workgroups: 9.99 vs 10.08 frames/s); the two-stage 256x144 kernel; the reduce pass right away.
  if (h->gemm_variant == GEMM_AUTO && h->out_part && (size_t)2 * m * c <= h->out_part_elems && m % 256 == 0 && c % 144 == 0 &&
      (long)(m / 256) * (c / 144) * 2 <= 256 && (5 * c / 64) >= 8) {
    GemmArgs p2 = o;
    p2.bias = nullptr; p2.out_f32 = h->out_part; p2.ksplit = 2; p2.slice_stride = (long)m * c;
    if ((rc = launch_gemm(A_DENSE, E_F32, GEMM_DMA3_256x144, p2, s))) return rc;
    h->pend_bias = w.b_out;
    h->pend_lvl = lvl; h->pend_m = m; h->pend_c = c; h->pend_slices = 2;
    swap_out_part(h);
    return DFOT_OK;
  }
  // level 2 (M = 16384, N = 576): 256x192 tiles are 192 workgroups -- a quarter of the chip idle for the whole kernel; 256x144 tiles
  // (N = 4 x 144) are exactly 256, one per CU, at 92 instead of 110 FLOP per operand byte, on the three-stage ring (10.10 vs 10.00
  // frames/s, two rounds; the two-stage form was equal to the 256x192 tiles)
  if (h->gemm_variant == GEMM_AUTO && c % 144 == 0 && m % 256 == 0 && (long)(m / 256) * ((c + 191) / 192) < 256 &&
      (long)(m / 256) * (c / 144) >= 200 && (long)(m / 256) * (c / 144) <= 256) {
    o.out_bf16 = xb;
    o.resid_bf = xb;  // in place on the bf16 stream: a thread reads its elements before it writes them
    return launch_gemm(A_DENSE, E_BF16, GEMM_DMA3_256x144, o, s);
  }
  // any other shape / a forced tile form: the plain fp32 epilogue into the level's fp32 scratch, then one pass adds it into the stream
  o.out_f32 = h->X[lvl];
  if ((rc = launch_gemm(A_DENSE, E_F32, h->gemm_variant, o, s))) return rc;
  hipLaunchKernelGGL(add_f32_into_bf16_kernel, dim3(cdiv((long)m * c / 4, 256)), dim3(256), 0, s, xb, h->X[lvl], (long)m * c / 4);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// (The block's two independent branches -- attention and MLP, u_vit_blocks.py:253-277 -- were run on two streams for a while; with
// this round's serial chain, cheaper epilogues and the two-slice ring out-projection, the fork measures SLOWER, 9.70 vs 9.78 frames/s,
// and was removed)

// out[row][c] = bias[c] + sum_s slab[s][row][c] for the rows of live images (fp32, 4 elements per thread)
__global__ void slab_reduce_kernel(float* __restrict__ out, const float* __restrict__ bias, const float* __restrict__ slabs, int nslab, long stride4,
                                   long total4, int cq, const uint8_t* __restrict__ live, unsigned img4) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  if (live && !live[(unsigned)i / img4]) return;
  f4 v = reinterpret_cast<const f4*>(bias)[i % cq];
  for (int sl = 0; sl < nslab; ++sl) v += reinterpret_cast<const f4*>(slabs)[sl * stride4 + i];
  reinterpret_cast<f4*>(out)[i] = v;
}

// The Downsample / Upsample convolutions between the transformer levels are few-tile, long-K implicit GEMMs (M = 4096 or 16384 pixels,
// K = 2304 ... 10368): on 128x128 tiles they ran at 400-690 TFLOP/s.  Where the shape allows it they take the three-stage 256x144 ring
// (256x256 tiles for N = 256) with K split over workgroups into fp32 slabs -- enough slices for ~256 workgroups -- and one reduce pass
// (bias + slabs; images skipped by the tile's flag are skipped there too, so their rows keep what they held).
static int conv_between_levels(dfot_uvit_s* h, GemmArgs g, hipStream_t s) {
  const long px = (long)g.H * g.Wd;
  if (h->gemm_variant == GEMM_AUTO && h->out_part && !g.gn_part && g.M % 256 == 0 && px % 256 == 0 && (g.N % 144 == 0 || g.N == 256) &&
      (long)g.M * g.N < (1L << 31)) {
    const int bn = g.N % 144 == 0 ? 144 : 256;
    const int variant = bn == 144 ? GEMM_DMA3_256x144 : GEMM_DMA_256x256;
    const int tiles = (g.M / 256) * (g.N / bn);
    int ks = 1;
    while (tiles * ks * 2 <= 256 && g.K / 64 / (ks * 2) >= 16) ks *= 2;
    if (tiles * ks >= 192 && tiles * ks <= 256) {
      if (ks == 1) return launch_gemm(A_CONV3, E_F32, variant, g, s);
      if ((size_t)ks * g.M * g.N <= h->out_part_elems) {
        GemmArgs p = g;
        float* out = g.out_f32;
        const float* bias = g.bias;
        p.bias = nullptr; p.resid = nullptr; p.out_f32 = h->out_part; p.ksplit = ks; p.slice_stride = (long)g.M * g.N;
        int rc = launch_gemm(A_CONV3, E_F32, variant, p, s);
        if (rc) return rc;
        const long total4 = (long)g.M * g.N / 4;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, s, out, bias, h->out_part, ks, total4, total4, g.N / 4, g.live,
                           (unsigned)(px * g.N / 4));
        DFOT_CHECK_HIP(hipGetLastError());
        return DFOT_OK;
      }
    }
  }
  return launch_gemm(A_CONV3, E_F32, h->gemm_variant, g, s);
}

static int run_down(dfot_uvit_s* h, int l, int bt, hipStream_t s, const uint8_t* live = nullptr) {
  const int rr = h->r[l], cin = h->ch[l], cout = h->ch[l + 1];
  int rc = 0;
  if ((rc = h->xin_bf[l] ? launch_pool2_bf16_bf16in(h->XB[l], h->s1, bt, rr, rr, cin, s) : launch_pool2_bf16(h->xin[l], h->s1, bt, rr, rr, cin, s)))
    return rc;
  GemmArgs g;
  g.A = h->s1; g.W = h->down_conv[l].w; g.M = bt * (rr / 2) * (rr / 2); g.N = cout; g.K = 9 * cin; g.H = rr / 2; g.Wd = rr / 2;
  g.Cin = cin; g.zeros = h->zeros; g.bias = h->down_conv[l].b; g.out_f32 = h->HSA[l]; g.ldo = cout;
  g.live = live;  // frozen frames (forward_cached_masks checked that whole tiles lie inside one image)
  const bool next_is_res = l + 1 < 2;
  if (next_is_res) {
    g.gn_part = h->gn_partial; g.gn_rows_per_bt = (rr / 2) * (rr / 2); g.gn_cpg = cout / 32;
  }
  if ((rc = conv_between_levels(h, g, s))) return rc;
  h->gn1_nblk = next_is_res ? g.gn_rows_per_bt / 64 : 0;
  h->xin[l + 1] = h->HSA[l];  // the skip tensor IS the next level's input: its first block reads it here and writes X[l + 1]
  h->xin_bf[l + 1] = false;
  return DFOT_OK;
}

static int run_up(dfot_uvit_s* h, int l, int bt, hipStream_t s, const uint8_t* live = nullptr) {  // level l+1 -> l
  const int rr = h->r[l + 1], cin = h->ch[l + 1], cout = h->ch[l];
  const long n_in = (long)bt * rr * rr * cin;
  int rc = 0;
  if ((rc = h->xin_bf[l + 1] ? launch_sub_bf16_bf16in(h->XB[l + 1], h->HSA[l], h->s1, n_in, s, live, (long)rr * rr * cin)
                             : launch_sub_bf16(h->xin[l + 1], h->HSA[l], h->s1, n_in, s, live, (long)rr * rr * cin)))
    return rc;
  GemmArgs g;
  g.A = h->s1; g.W = h->up_conv[l].w; g.M = bt * rr * rr; g.N = cout; g.K = 9 * cin; g.H = rr; g.Wd = rr; g.Cin = cin;
  g.zeros = h->zeros; g.bias = h->up_conv[l].b; g.out_f32 = h->tmp; g.ldo = cout;
  // (a 16x16 / 32x32 coarse map is smaller than the larger tiles: the flags apply only where whole tiles lie inside one image)
  if (live && (rr * rr) % 256 == 0 && h->gemm_variant == GEMM_AUTO) g.live = live;
  if ((rc = conv_between_levels(h, g, s))) return rc;
  h->gn1_nblk = 0;  // X[l] is rewritten by an elementwise kernel: its statistics come from the standalone kernel
  // the down path left the level's output in the bf16 stream; the sum goes back into it (elementwise, in place)
  if (!h->xin_bf[l]) {
    set_error("run_up: level %d stream is not in its bf16 buffer", l);
    return DFOT_ERR_STATE;
  }
  return launch_upsample_add_bf16(h->tmp, h->XB[l], h->XB[l], bt, rr, rr, cout, s, live);
}

}  // namespace dfot

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* dfot_last_error(void) { return get_error(); }
int dfot_version(void) { return 1; }

int dfot_uvit_create(const dfot_uvit_config* cfg, dfot_uvit_t* out) {
  DFOT_REQUIRE(cfg && out, DFOT_ERR_ARG, "dfot_uvit_create: null argument");
  DFOT_REQUIRE(cfg->resolution % 16 == 0 && cfg->resolution >= 32, DFOT_ERR_SHAPE, "resolution %d must be a multiple of 16", cfg->resolution);
  DFOT_REQUIRE(cfg->emb_channels % 64 == 0, DFOT_ERR_SHAPE, "emb_channels %d must be a multiple of 64", cfg->emb_channels);
  DFOT_REQUIRE(cfg->in_channels <= 3 && cfg->cond_dim % 20 == 0, DFOT_ERR_SHAPE, "in_channels %d / cond_dim %d unsupported", cfg->in_channels, cfg->cond_dim);
  for (int l = 0; l < 4; ++l)
    DFOT_REQUIRE(cfg->channels[l] % 64 == 0, DFOT_ERR_SHAPE, "channels[%d]=%d must be a multiple of 64", l, cfg->channels[l]);
  DFOT_REQUIRE(cfg->channels[0] % 128 == 0 && cfg->channels[1] % 128 == 0, DFOT_ERR_SHAPE, "ResBlock channels must be multiples of 128");
  for (int l = 2; l < 4; ++l) {
    const int d = cfg->channels[l] / cfg->num_heads;
    DFOT_REQUIRE(d * cfg->num_heads == cfg->channels[l] && (d == 64 || d == 128), DFOT_ERR_SHAPE,
                 "level %d head dim %d must be 64 or 128", l, d);
  }
  const int r3 = cfg->resolution / 16;
  DFOT_REQUIRE((cfg->max_tokens * r3 * r3) % 128 == 0, DFOT_ERR_SHAPE, "tokens at the coarsest level (%d) must be a multiple of 128", cfg->max_tokens * r3 * r3);
  auto* h = new dfot_uvit_s();
  h->cfg = *cfg;
  int rc = build(h);
  if (rc) {
    dfot_uvit_destroy(h);
    return rc;
  }
  *out = h;
  return DFOT_OK;
}

int dfot_uvit_destroy(dfot_uvit_t h) {
  if (!h) return DFOT_OK;
  for (void* p : h->owned) (void)hipFree(p);
  for (void* p : h->ws_owned) (void)hipFree(p);
  for (hipEvent_t e : h->ev_start) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_stop) (void)hipEventDestroy(e);
  delete h;
  return DFOT_OK;
}

int dfot_uvit_num_params(dfot_uvit_t h) { return h ? (int)h->params.size() : 0; }
const char* dfot_uvit_param_name(dfot_uvit_t h, int i) {
  return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : nullptr;
}
int dfot_uvit_param_shape(dfot_uvit_t h, int i, int64_t shape[4], int* ndim) {
  DFOT_REQUIRE(h && shape && ndim && i >= 0 && i < (int)h->params.size(), DFOT_ERR_ARG, "param_shape: bad argument");
  *ndim = (int)h->params[i].shape.size();
  for (int k = 0; k < *ndim; ++k) shape[k] = h->params[i].shape[k];
  return DFOT_OK;
}

int dfot_uvit_load_weight(dfot_uvit_t h, const char* name, const float* data, const int64_t* shape, int ndim, void* stream) {
  DFOT_REQUIRE(h && name && data && shape, DFOT_ERR_ARG, "load_weight: null argument");
  auto it = h->index.find(name);
  DFOT_REQUIRE(it != h->index.end(), DFOT_ERR_NAME, "load_weight: unexpected key '%s'", name);
  Param& p = h->params[it->second];
  bool same = (int)p.shape.size() == ndim;
  for (int k = 0; same && k < ndim; ++k) same = p.shape[k] == shape[k];
  DFOT_REQUIRE(same, DFOT_ERR_SHAPE, "load_weight: size mismatch for '%s'", name);
  int rc = p.load(data, (hipStream_t)stream);
  if (rc) return rc;
  p.loaded = true;
  h->finalized = false;
  return DFOT_OK;
}

int dfot_uvit_finalize(dfot_uvit_t h, void* stream) {
  DFOT_REQUIRE(h, DFOT_ERR_ARG, "finalize: null handle");
  for (const Param& p : h->params) DFOT_REQUIRE(p.loaded, DFOT_ERR_STATE, "finalize: missing key '%s'", p.name.c_str());
  hipStream_t s = (hipStream_t)stream;
  auto fuse = [&](std::vector<TrW>& v) -> int {
    for (TrW& w : v) {
      hipLaunchKernelGGL(add_vec_kernel, dim3(cdiv(w.c, 256)), dim3(256), 0, s, w.b_attn, w.b_mlp, w.b_out, w.c);
      DFOT_CHECK_HIP(hipGetLastError());
    }
    return DFOT_OK;
  };
  int rc = 0;
  if ((rc = fuse(h->down_tr)) || (rc = fuse(h->mid_tr)) || (rc = fuse(h->up_tr))) return rc;
  for (int l = 2; l < 4; ++l)
    if (!h->rope_cs[l] && (rc = build_rope(h, l))) return rc;
  DFOT_CHECK_HIP(hipStreamSynchronize(s));
  auto bound = [&](std::vector<TrW>& v) -> int {
    std::vector<float> qk;
    for (TrW& w : v) {
      const int d = w.c / h->heads;
      qk.resize(2 * d);
      DFOT_CHECK_HIP(hipMemcpy(qk.data(), w.qw, d * sizeof(float), hipMemcpyDeviceToHost));
      DFOT_CHECK_HIP(hipMemcpy(qk.data() + d, w.kw, d * sizeof(float), hipMemcpyDeviceToHost));
      // |q.k| after the per-head RMSNorm (|x^|^2 = d) and the rotation of the pairs (2j, 2j+1): sum_j |q_j||k_j| with
      // |q_j| <= a_j |x^q_j|, a_j = max(|w_q,2j|, |w_q,2j+1|) (b_j likewise) <= max_j(a_j b_j) |x^q||x^k| = max_j(a_j b_j) d:
      // the largest PRODUCT over a rotary pair, not the product of the two maxima taken over all channels
      float mqk = 0.f;
      for (int i = 0; i + 1 < d; i += 2)
        mqk = fmaxf(mqk, fmaxf(fabsf(qk[i]), fabsf(qk[i + 1])) * fmaxf(fabsf(qk[d + i]), fabsf(qk[d + i + 1])));
      w.score_bound = sqrtf((float)d) * mqk * 1.4426950408889634f;
      if (!(w.score_bound == w.score_bound)) w.score_bound = INFINITY;  // NaN weights: keep the general kernel
    }
    return DFOT_OK;
  };
  if ((rc = bound(h->down_tr)) || (rc = bound(h->mid_tr)) || (rc = bound(h->up_tr))) return rc;
  h->finalized = true;
  return DFOT_OK;
}

int dfot_uvit_reserve(dfot_uvit_t h, int max_batch) {
  DFOT_REQUIRE(h && max_batch > 0, DFOT_ERR_ARG, "reserve: bad argument");
  if (max_batch <= h->max_batch) return DFOT_OK;
  for (void* p : h->ws_owned) (void)hipFree(p);
  h->ws_owned.clear();
  h->ws_bytes = 0;
  h->max_batch = 0;
  h->out_part = nullptr;
  h->pend_part = nullptr;
  h->out_part_elems = 0;
  h->attn_scratch = AttnScratch{};
  const size_t bt = (size_t)max_batch * h->T;
  size_t pix[4];
  for (int l = 0; l < 4; ++l) pix[l] = (size_t)h->r[l] * h->r[l];
  int rc = 0;
  if ((rc = dev_alloc(h, &h->nemb, bt * h->E, true))) return rc;
  if ((rc = dev_alloc(h, &h->nhid, bt * h->E, true))) return rc;
  for (int l = 0; l < 4; ++l) {
    if ((rc = dev_alloc(h, &h->XB[l], bt * pix[l] * h->ch[l], true))) return rc;
    // fp32 scratch of a transformer level (the out-projection's fallback path, run_tr_block)
    if (l >= 2 && (rc = dev_alloc(h, &h->X[l], bt * pix[l] * h->ch[l], true))) return rc;
    if ((rc = dev_alloc(h, &h->emb[l], bt * pix[l] * h->E, true))) return rc;
  }
  for (int l = 0; l < 3; ++l)
    if ((rc = dev_alloc(h, &h->HSA[l], bt * pix[l + 1] * h->ch[l + 1], true))) return rc;
  if ((rc = dev_alloc(h, &h->acond, bt * pix[0] * h->kpose, true))) return rc;
  DFOT_CHECK_HIP(hipMemset(h->acond, 0, bt * pix[0] * h->kpose * sizeof(bf16)));
  size_t act = 0, tmpn = 0;
  for (int l = 0; l < 4; ++l) act = std::max(act, bt * pix[l] * h->ch[l]);
  for (int l = 0; l < 3; ++l) tmpn = std::max(tmpn, bt * pix[l + 1] * h->ch[l]);
  if ((rc = dev_alloc(h, &h->s1, act, true))) return rc;
  if ((rc = dev_alloc(h, &h->hbf, act, true))) return rc;
  if ((rc = dev_alloc(h, &h->tmp, tmpn, true))) return rc;
  if ((rc = dev_alloc(h, &h->gn_partial, bt * (pix[0] / 64) * 64, true))) return rc;
  if ((rc = dev_alloc(h, &h->gn_partial2, bt * (pix[0] / 64) * 64, true))) return rc;
  if ((rc = dev_alloc(h, &h->gn_stats, bt * 64, true))) return rc;
  size_t mtr = 0, mc = 0;
  for (int l = 2; l < 4; ++l) {
    mtr = std::max(mtr, bt * pix[l]);
    mc = std::max(mc, bt * pix[l] * h->ch[l]);
  }
  (void)mtr;
  // per-window FiLM caches + per-frame FiLM vectors + the chunk table of film_vec_kernel
  {
    std::vector<FilmChunk> table;
    long sv_off = 0;
    auto add = [&](bf16** fc, long* off, const bf16* wf, const float* bf_, int c, int lvl) -> int {
      int r2 = dev_alloc(h, fc, bt * pix[lvl] * 2 * c, true);
      if (r2) return r2;
      *off = sv_off;
      for (int r0 = 0; r0 < 2 * c; r0 += 64) table.push_back(FilmChunk{wf + (long)r0 * h->E, bf_ + r0, sv_off + r0, 2 * c});
      sv_off += (long)bt * 2 * c;
      return DFOT_OK;
    };
    for (int l = 0; l < 2; ++l) {
      for (ResW& w : h->down_res[l]) if ((rc = add(&w.fcache, &w.sv_off, w.w_film, w.b_film, w.c, l))) return rc;
      for (ResW& w : h->up_res[l]) if ((rc = add(&w.fcache, &w.sv_off, w.w_film, w.b_film, w.c, l))) return rc;
    }
    for (TrW& w : h->down_tr) if ((rc = add(&w.fcache, &w.sv_off, w.w_film, w.b_film, w.c, 2))) return rc;
    for (TrW& w : h->up_tr) if ((rc = add(&w.fcache, &w.sv_off, w.w_film, w.b_film, w.c, 2))) return rc;
    for (TrW& w : h->mid_tr) if ((rc = add(&w.fcache, &w.sv_off, w.w_film, w.b_film, w.c, 3))) return rc;
    if ((rc = dev_alloc(h, &h->sv, (size_t)sv_off, true))) return rc;
    if ((rc = dev_alloc(h, &h->film_table, table.size(), true))) return rc;
    DFOT_CHECK_HIP(hipMemcpy(h->film_table, table.data(), table.size() * sizeof(FilmChunk), hipMemcpyHostToDevice));
    h->film_chunks = (int)table.size();
    if ((rc = dev_alloc(h, &h->cond_mask, (size_t)max_batch, true))) return rc;
    h->cond_batch = 0;
  }
  if ((rc = dev_alloc(h, &h->cat, mc * 5, true))) return rc;
  if ((rc = dev_alloc(h, &h->q, mc, true))) return rc;
  if ((rc = dev_alloc(h, &h->k, mc, true))) return rc;
  if ((rc = dev_alloc(h, &h->v, mc, true))) return rc;
  {  // both transformer levels may take the two-slice out-projection at small sizes: room for the larger of their outputs, twice
    const size_t e2 = bt * pix[2] * h->ch[2], e3 = bt * pix[3] * h->ch[3];
    h->out_part_elems = 3 * (e2 > e3 ? e2 : e3);
    if ((rc = dev_alloc(h, &h->out_part, h->out_part_elems, true))) return rc;
  }
  // key-split partial buffers of the level-2 attention's balanced tail, large enough for every batch this workspace can serve:
  // owned by the handle (nothing is allocated, freed or shared with another handle inside forward / stream capture)
  {
    size_t need = 16;
    for (int b = 1; b <= max_batch; ++b)
      for (int l = 2; l < 4; ++l) need = std::max(need, attention_scratch_bytes(b, h->heads, h->T * h->r[l] * h->r[l], h->ch[l] / h->heads));
    if ((rc = dev_alloc(h, &h->attn_scratch.p, need / sizeof(float), true))) return rc;
    h->attn_scratch.bytes = need;
  }
  h->max_batch = max_batch;
  return DFOT_OK;
}

size_t dfot_uvit_workspace_bytes(dfot_uvit_t h) { return h ? h->ws_bytes : 0; }

int dfot_uvit_attn_timing(dfot_uvit_t h, double* total_ms, int64_t* launches) {
  DFOT_REQUIRE(h && total_ms && launches, DFOT_ERR_ARG, "attn_timing: null argument");
  double tot = 0.0;
  for (size_t i = 0; i < h->ev_used; ++i) {
    DFOT_CHECK_HIP(hipEventSynchronize(h->ev_stop[i]));
    float ms = 0.f;
    DFOT_CHECK_HIP(hipEventElapsedTime(&ms, h->ev_start[i], h->ev_stop[i]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int64_t)h->ev_used;
  h->ev_used = 0;
  return DFOT_OK;
}

int dfot_uvit_set_option(dfot_uvit_t h, const char* key, int value) {
  DFOT_REQUIRE(h && key, DFOT_ERR_ARG, "set_option: null argument");
  if (!strcmp(key, "gemm_variant")) h->gemm_variant = value;
  else if (!strcmp(key, "attn_variant")) h->attn_variant = value;
  else if (!strcmp(key, "attn_force_safe")) h->attn_force_safe = value != 0;
  else if (!strcmp(key, "time_attn")) {
    // value = number of launches to record (0 disables); events are created here, never inside forward
    h->time_attn = value > 0;
    h->ev_used = 0;
    while ((int)h->ev_start.size() < value) {
      hipEvent_t a, b;
      DFOT_CHECK_HIP(hipEventCreate(&a));
      DFOT_CHECK_HIP(hipEventCreate(&b));
      h->ev_start.push_back(a);
      h->ev_stop.push_back(b);
    }
  }
  else {
    set_error("set_option: unknown key '%s'", key);
    return DFOT_ERR_ARG;
  }
  return DFOT_OK;
}

int dfot_uvit_query(dfot_uvit_t h, const char* key, double* value) {
  DFOT_REQUIRE(h && key && value, DFOT_ERR_ARG, "query: null argument");
  DFOT_REQUIRE(h->finalized, DFOT_ERR_STATE, "query: weights not finalized");
  float bound = 0.f;
  for (const TrW& w : h->down_tr) bound = fmaxf(bound, w.score_bound);
  for (const TrW& w : h->up_tr) bound = fmaxf(bound, w.score_bound);
  if (!strcmp(key, "score_bound_l2")) *value = bound;
  else if (!strcmp(key, "attn_kernel_l2")) *value = (bound < 64.0f && !h->attn_force_safe) ? 14 : 5;
  else {
    set_error("query: unknown key '%s'", key);
    return DFOT_ERR_ARG;
  }
  return DFOT_OK;
}

int dfot_uvit_set_conditions(dfot_uvit_t h, const float* external_cond, const uint8_t* external_cond_mask, int batch,
                             void* stream) {
  DFOT_REQUIRE(h, DFOT_ERR_ARG, "set_conditions: null handle");
  DFOT_REQUIRE(external_cond != nullptr, DFOT_ERR_ARG, "External condition (camera pose) is required for U-ViT3DPose model.");
  DFOT_REQUIRE(h->finalized, DFOT_ERR_STATE, "set_conditions: weights not finalized");
  DFOT_REQUIRE(batch > 0 && batch <= h->max_batch, DFOT_ERR_STATE, "set_conditions: batch %d exceeds reserved %d", batch, h->max_batch);
  hipStream_t s = (hipStream_t)stream;
  const dfot_uvit_config& c = h->cfg;
  const int bt = batch * h->T, e = h->E;
  int rc = 0;
  h->cond_batch = 0;
  // pose patch-embed (+bias), then its average-pool pyramid: the pose half of `emb` at every level
  if ((rc = launch_cond_repack(external_cond, h->acond, bt, c.resolution, c.cond_dim, h->kpose, s))) return rc;
  {
    GemmArgs g;
    g.A = h->acond; g.lda = h->kpose; g.W = h->pose_w; g.M = bt * h->r[0] * h->r[0]; g.N = e; g.K = h->kpose;
    g.bias = h->pose_b; g.out_bf16 = h->emb[0]; g.ldo = e;
    if ((rc = launch_gemm(A_DENSE, E_BF16, h->gemm_variant, g, s))) return rc;
  }
  if ((rc = launch_emb_pyramid(h->emb[0], h->emb[1], h->emb[2], h->emb[3], bt, h->r[0], e, s))) return rc;
  // every block's FiLM projection of the pose term: F = W_film * pose_emb  (no bias; it lives in sv)
  auto fill = [&](bf16* fcache, const bf16* wf, int cc, int lvl) -> int {
    GemmArgs g;
    g.A = h->emb[lvl]; g.lda = e; g.W = wf; g.M = bt * h->r[lvl] * h->r[lvl]; g.N = 2 * cc; g.K = e;
    g.out_bf16 = fcache; g.ldo = 2 * cc;
    return launch_gemm(A_DENSE, E_BF16, h->gemm_variant, g, s);
  };
  for (int l = 0; l < 2; ++l) {
    for (ResW& w : h->down_res[l]) if ((rc = fill(w.fcache, w.w_film, w.c, l))) return rc;
    for (ResW& w : h->up_res[l]) if ((rc = fill(w.fcache, w.w_film, w.c, l))) return rc;
  }
  for (TrW& w : h->down_tr) if ((rc = fill(w.fcache, w.w_film, w.c, 2))) return rc;
  for (TrW& w : h->up_tr) if ((rc = fill(w.fcache, w.w_film, w.c, 2))) return rc;
  for (TrW& w : h->mid_tr) if ((rc = fill(w.fcache, w.w_film, w.c, 3))) return rc;
  h->have_mask = external_cond_mask != nullptr;
  if (h->have_mask) DFOT_CHECK_HIP(hipMemcpyAsync(h->cond_mask, external_cond_mask, batch, hipMemcpyDeviceToDevice, s));
  h->cond_batch = batch;
  return DFOT_OK;
}

int dfot_uvit_forward_cached(dfot_uvit_t h, const float* x, const float* noise_levels, float* out, int batch, void* stream) {
  return dfot_uvit_forward_cached_live(h, x, noise_levels, out, batch, nullptr, stream);
}

int dfot_uvit_forward_cached_live(dfot_uvit_t h, const float* x, const float* noise_levels, float* out, int batch, const uint8_t* live_frames,
                                  void* stream) {
  return dfot_uvit_forward_cached_masks(h, x, noise_levels, out, batch, live_frames, nullptr, stream);
}

int dfot_uvit_forward_cached_masks(dfot_uvit_t h, const float* x, const float* noise_levels, float* out, int batch, const uint8_t* live_frames,
                                   const uint8_t* fresh_frames, void* stream) {
  DFOT_REQUIRE(h && x && noise_levels && out, DFOT_ERR_ARG, "forward: null argument");
  DFOT_REQUIRE(!fresh_frames || h->last_batch == batch, DFOT_ERR_STATE,
               "forward: frozen frames need the previous forward of this handle to have run the same batch (%d, now %d)", h->last_batch, batch);
  DFOT_REQUIRE(!fresh_frames || live_frames, DFOT_ERR_ARG, "forward: frozen frames must also be dead frames (live_frames is null)");
  DFOT_REQUIRE(h->finalized, DFOT_ERR_STATE, "forward: weights not finalized");
  DFOT_REQUIRE(batch > 0 && batch == h->cond_batch, DFOT_ERR_STATE,
               "forward_cached: batch %d does not match the cached conditions (%d)", batch, h->cond_batch);
  hipStream_t s = (hipStream_t)stream;
  const dfot_uvit_config& c = h->cfg;
  const int bt = batch * h->T, e = h->E;
  int rc = 0;
  if ((rc = launch_noise_emb(noise_levels, h->ne_freqs, h->ne_phases, h->ne_w1, h->ne_b1, h->ne_w2, h->ne_b2, h->nhid,
                             h->nemb, bt, c.noise_dim, e, s)))
    return rc;
  if ((rc = launch_film_vec(h->film_table, h->film_chunks, h->nemb, h->sv, bt, e, s))) return rc;
  if ((rc = launch_embed_input_bf16(x, h->ein_w, h->ein_b, h->XB[0], bt, c.resolution, c.in_channels, h->ch[0], s))) return rc;
  h->gn1_nblk = 0;
  for (int l = 0; l < 4; ++l) h->xin[l] = h->X[l];
  h->xin_bf[0] = true;
  h->xin_bf[1] = h->xin_bf[2] = h->xin_bf[3] = false;

  // frames that are NOT fresh (fresh_frames[b * T + t] == 0): the caller states that this frame's input, noise level and conditioning
  // equal those of the previous forward of this handle (a clean context frame of the conditional branch across the DDIM steps of a
  // window).  The ResBlock levels work frame by frame, so the frame's rows of the skip tensors and of the level-2 input still hold
  // exactly what would be recomputed: its tiles and rows are skipped on the way down.  All or nothing: a frame skipped by one stage and
  // recomputed by the next would be recomputed from stale rows, so the flags are dropped unless every convolution of the down path
  // has whole tiles per image (the largest tile is 512 rows)
  if (fresh_frames && (h->gemm_variant != GEMM_AUTO || (h->r[2] * h->r[2]) % 512 != 0)) fresh_frames = nullptr;
  for (int l = 0; l < 2; ++l) {
    for (const ResW& w : h->down_res[l])
      if ((rc = run_res_block(h, w, l, bt, s, fresh_frames))) return rc;
    if ((rc = run_down(h, l, bt, s, fresh_frames))) return rc;
  }
  h->pend_bias = nullptr;
  for (const TrW& w : h->down_tr)
    if ((rc = run_tr_block(h, w, 2, batch, s))) return rc;
  if ((rc = flush_pending(h, s))) return rc;
  if ((rc = run_down(h, 2, bt, s))) return rc;
  for (const TrW& w : h->mid_tr)
    if ((rc = run_tr_block(h, w, 3, batch, s))) return rc;
  if ((rc = flush_pending(h, s))) return rc;
  if ((rc = run_up(h, 2, bt, s))) return rc;
  for (const TrW& w : h->up_tr)
    if ((rc = run_tr_block(h, w, 2, batch, s))) return rc;
  if ((rc = flush_pending(h, s))) return rc;
  // past the last transformer block every kernel works on one frame at a time (3x3 convolutions, per-frame GroupNorm, per-pixel
  // FiLM): frames whose output the caller discards (live_frames[b * T + t] == 0: the sampler's context tokens, whose v the composition
  // step never reads) are skipped there -- their rows of the output are zeros
  for (int l = 1; l >= 0; --l) {
    if ((rc = run_up(h, l, bt, s, live_frames))) return rc;
    for (const ResW& w : h->up_res[l])
      if ((rc = run_res_block(h, w, l, bt, s, live_frames))) return rc;
  }
  h->last_batch = batch;
  return launch_project_output_bf16(h->XB[0], h->pout_w, h->pout_b, out, bt, c.resolution, h->ch[0], c.in_channels, s, live_frames);
}

int dfot_uvit_forward(dfot_uvit_t h, const float* x, const float* noise_levels, const float* external_cond,
                      const uint8_t* external_cond_mask, float* out, int batch, void* stream) {
  DFOT_REQUIRE(h && x && noise_levels && out, DFOT_ERR_ARG, "forward: null argument");
  int rc = dfot_uvit_set_conditions(h, external_cond, external_cond_mask, batch, stream);
  if (rc) return rc;
  return dfot_uvit_forward_cached(h, x, noise_levels, out, batch, stream);
}

int dfot_uvit_read_tap(dfot_uvit_t h, const char* name, float* out, size_t capacity, void* stream) {
  DFOT_REQUIRE(h && name && out, DFOT_ERR_ARG, "read_tap: null argument");
  DFOT_REQUIRE(h->last_batch > 0, DFOT_ERR_STATE, "read_tap: no forward has run");
  hipStream_t s = (hipStream_t)stream;
  const int bt = h->last_batch * h->T;
  auto pixels = [&](int l) { return h->r[l] * h->r[l]; };
  struct Tap { const char* n; int lvl; int c; const float* f; const bf16* b; };
  const Tap taps[] = {
      {"pose_emb0", 0, h->E, nullptr, h->emb[0]}, {"down0", 1, h->ch[1], h->HSA[0], nullptr},
      {"down1", 2, h->ch[2], h->HSA[1], nullptr}, {"down2", 3, h->ch[3], h->HSA[2], nullptr},
      {"mid", 3, h->ch[3], nullptr, h->XB[3]}, {"up2", 2, h->ch[2], nullptr, h->XB[2]},
      {"up1", 1, h->ch[1], nullptr, h->XB[1]}, {"up0", 0, h->ch[0], nullptr, h->XB[0]},
  };
  for (const Tap& t : taps) {
    if (strcmp(t.n, name)) continue;
    const size_t need = (size_t)bt * pixels(t.lvl) * t.c;
    DFOT_REQUIRE(capacity >= need, DFOT_ERR_SHAPE, "read_tap: need %zu floats, got %zu", need, capacity);
    return t.f ? launch_nhwc_to_nchw(t.f, out, bt, pixels(t.lvl), t.c, s)
               : launch_bf16_nhwc_to_nchw(t.b, out, bt, pixels(t.lvl), t.c, s);
  }
  set_error("read_tap: unknown tap '%s'", name);
  return DFOT_ERR_NAME;
}

// ---- pose / sampler / primitives ------------------------------------------------------------
int dfot_ray_encode(const float* raw_poses, float* out, int batch, int tokens, int resolution, void* stream) {
  DFOT_REQUIRE(raw_poses && out && batch > 0 && tokens > 0 && resolution > 0, DFOT_ERR_ARG, "ray_encode: bad argument");
  return launch_ray_encode(raw_poses, out, batch, tokens, resolution, 0, (hipStream_t)stream);
}

int dfot_ray_encode_normalized(const float* poses, float* out, int batch, int tokens, int resolution, void* stream) {
  DFOT_REQUIRE(poses && out && batch > 0 && tokens > 0 && resolution > 0, DFOT_ERR_ARG, "ray_encode_normalized: bad argument");
  return launch_ray_encode(poses, out, batch, tokens, resolution, 1, (hipStream_t)stream);
}

int dfot_hg_prepare(const float* x, const float* noise, const float* qa, const float* qb, float* x_in, int batch, int nfe,
                    int tokens, int64_t frame_elems, void* stream) {
  DFOT_REQUIRE(x && qa && qb && x_in, DFOT_ERR_ARG, "hg_prepare: null argument");
  return launch_hg_prepare(x, noise, qa, qb, x_in, batch, nfe, tokens, (long)frame_elems, (hipStream_t)stream);
}

int dfot_ddim_compose(const float* x, const float* x_in, const float* v, const float* sa, const float* s1, const float* an,
                      const float* cn, const float* keep, const float* weight, const uint8_t* gen, float* x_next, int batch,
                      int nfe, int tokens, int64_t frame_elems, void* stream) {
  DFOT_REQUIRE(x && x_in && v && sa && s1 && an && cn && keep && weight && gen && x_next, DFOT_ERR_ARG, "ddim_compose: null argument");
  return launch_ddim_compose(x, x_in, v, sa, s1, an, cn, keep, weight, gen, x_next, batch, nfe, tokens, (long)frame_elems, false,
                             (hipStream_t)stream);
}
int dfot_ddim_compose_tokw(const float* x, const float* x_in, const float* v, const float* sa, const float* s1, const float* an,
                           const float* cn, const float* keep, const float* weight, const uint8_t* gen, float* x_next, int batch,
                           int nfe, int tokens, int64_t frame_elems, void* stream) {
  DFOT_REQUIRE(x && x_in && v && sa && s1 && an && cn && keep && weight && gen && x_next, DFOT_ERR_ARG, "ddim_compose_tokw: null argument");
  return launch_ddim_compose(x, x_in, v, sa, s1, an, cn, keep, weight, gen, x_next, batch, nfe, tokens, (long)frame_elems, true,
                             (hipStream_t)stream);
}

int dfot_ddim_noise(const float* noise, const float* sigma, const float* weight, const uint8_t* gen, float* x_next, int batch, int nfe,
                    int tokens, int64_t frame_elems, int weight_per_token, void* stream) {
  DFOT_REQUIRE(noise && sigma && weight && gen && x_next, DFOT_ERR_ARG, "ddim_noise: null argument");
  return launch_ddim_noise(noise, sigma, weight, gen, x_next, batch, nfe, tokens, (long)frame_elems, weight_per_token != 0, (hipStream_t)stream);
}

int dfot_vpred_loss(const float* x, const float* noise, const float* v, const float* alpha, const float* sigma,
                    const float* weight, float* x_pred, float* scratch, float* loss, int batch, int tokens, int64_t frame_elems,
                    void* stream) {
  DFOT_REQUIRE(x && noise && v && alpha && sigma && weight && scratch && loss, DFOT_ERR_ARG, "vpred_loss: null argument");
  return launch_vloss(x, noise, v, alpha, sigma, weight, x_pred, scratch, loss, batch * tokens, (long)frame_elems, false,
                      (hipStream_t)stream);
}
int dfot_vspace_loss(const float* x, const float* noise, const float* v, const float* alpha, const float* sigma,
                     const float* weight, float* x_pred, float* scratch, float* loss, int batch, int tokens, int64_t frame_elems,
                     void* stream) {
  DFOT_REQUIRE(x && noise && v && alpha && sigma && weight && scratch && loss, DFOT_ERR_ARG, "vspace_loss: null argument");
  return launch_vloss(x, noise, v, alpha, sigma, weight, x_pred, scratch, loss, batch * tokens, (long)frame_elems, true,
                      (hipStream_t)stream);
}
int64_t dfot_vpred_loss_scratch_floats(int batch, int tokens, int64_t frame_elems) {
  return (int64_t)batch * tokens * vloss_chunks((long)frame_elems);
}

static bf16* g_zero_page = nullptr;
static int zero_page(bf16** out) {
  if (!g_zero_page) {
    DFOT_CHECK_HIP(hipMalloc((void**)&g_zero_page, 512));
    DFOT_CHECK_HIP(hipMemset(g_zero_page, 0, 512));
  }
  *out = g_zero_page;
  return DFOT_OK;
}

int dfot_op_gemm(const void* a, int lda, const void* w, const float* bias, float* c, int m, int n, int k, int variant,
                 void* stream) {
  GemmArgs g;
  g.A = (const bf16*)a; g.lda = lda; g.W = (const bf16*)w; g.M = m; g.N = n; g.K = k; g.bias = bias; g.out_f32 = c; g.ldo = n;
  DFOT_REQUIRE(c, DFOT_ERR_ARG, "op_gemm: null output");
  return launch_gemm(A_DENSE, E_F32, variant, g, (hipStream_t)stream);
}

int dfot_op_conv3x3(const void* a, const void* w, const float* bias, float* y, int bt, int hh, int ww, int cin, int cout,
                    int variant, void* stream) {
  GemmArgs g;
  int rc = zero_page(const_cast<bf16**>(&g.zeros));
  if (rc) return rc;
  g.A = (const bf16*)a; g.W = (const bf16*)w; g.M = bt * hh * ww; g.N = cout; g.K = 9 * cin; g.H = hh; g.Wd = ww; g.Cin = cin;
  g.bias = bias; g.out_f32 = y; g.ldo = cout;
  DFOT_REQUIRE(y, DFOT_ERR_ARG, "op_conv3x3: null output");
  return launch_gemm(A_CONV3, E_F32, variant, g, (hipStream_t)stream);
}

int dfot_op_attention(const void* q, const void* k, const void* v, void* o, int ldo, int batch, int heads, int n, int d,
                      int variant, void* stream) {
  return launch_attention((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, ldo, batch, heads, n, d, variant,
                          (hipStream_t)stream);
}

int dfot_op_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  return launch_f32_to_bf16(src, (bf16*)dst, (long)n, (hipStream_t)stream);
}
int dfot_op_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  return launch_bf16_to_f32((const bf16*)src, dst, (long)n, (hipStream_t)stream);
}

}  // extern "C"
