// VideoVAE decoder pieces (algorithms/vae/video_vae/model.py:130-281, algorithms/vae/common/modules/*.py): what turns the Kinetics-600
// latents of the sampler into frames (`_decode`, algorithms/common/base_pytorch_video_algo.py:600-629).  The convolutions and the
// 1x1x1 projections run on the MFMA GEMM / implicit-GEMM kernels of gemm.hip through dfot_op_conv3x3_f32 / dfot_op_gemm_*; this file
// adds the HBM-bound pieces between them.  Activations are channels-last [B][T][H][W][C], fp32 streams and bf16 GEMM operands.
//   dfot_op_groupnorm          GroupNorm(32, eps) over (T, H, W) of one video (+ optional SiLU), fp32 -> bf16   (Normalize, normalize.py)
//   dfot_op_frame_shift        out[b][t] = x[b][max(t - shift, 0)]: the first-frame-replicating causal pad of PaddedConv3D (conv.py:104-109)
//                              -- a 3x3x3 causal convolution is three 3x3 convolutions over frames t-2, t-1, t accumulated in fp32
//   dfot_op_upsample3d         mode 0: nearest 2x in (H, W) (SpatialUpsample2x, updownsample.py:77-83); mode 1: the causal trilinear
//                              upsample of Spatial2xTime2x3DUpsample (:143-150): frame 0 bilinear 2x in (H, W), frames 1.. trilinear 2x in
//                              (T, H, W) (align_corners = False: out 2i -> 0.25 x[i-1] + 0.75 x[i], out 2i+1 -> 0.75 x[i] + 0.25 x[i+1], clamped)
//   dfot_op_softmax_rows       P = softmax(scale * S) per row, fp32 -> bf16 (AttnBlock3D, attention.py:127-129: one head of C channels per
//                              frame, so the scores go through the GEMM kernel, not the flash kernels whose head dim stops at 128)
#include "common.h"
#include "dfot_hip.h"
#include "kernels.h"

namespace dfot {
namespace {

typedef __attribute__((ext_vector_type(4))) float f4;

__global__ void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, bf16* __restrict__ out, long total4, int pixels, int c, int silu) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int cq = c / 4;
  const int c4 = (int)(idx % cq) * 4;
  const long pix = idx / cq;
  const int bt = (int)(pix / pixels);
  const int grp = c4 / (c / 32);  // channels per group >= 4: the four channels share a group
  const float mean = stats[((long)bt * 32 + grp) * 2], rstd = stats[((long)bt * 32 + grp) * 2 + 1];
  const f4 v = *reinterpret_cast<const f4*>(x + pix * c + c4);
  const f4 g = *reinterpret_cast<const f4*>(gamma + c4), b = *reinterpret_cast<const f4*>(beta + c4);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float z = (v[j] - mean) * rstd * g[j] + b[j];
    o[j] = f2bf(silu ? silu_f(z) : z);
  }
  *reinterpret_cast<bf16x4*>(out + pix * c + c4) = o;
}

__global__ void frame_shift_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, long total8, int t, long frame8, int shift) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total8) return;
  const long e = idx % frame8;
  const long bt = idx / frame8;
  const int tt = (int)(bt % t);
  const long b = bt / t;
  const int src_t = tt - shift < 0 ? 0 : tt - shift;
  reinterpret_cast<bf16x8*>(out)[idx] = reinterpret_cast<const bf16x8*>(x)[(b * t + src_t) * frame8 + e];
}

// linear interpolation source of output index o at scale 2, align_corners = False, n source samples: (i0, i1, w1)
__device__ __forceinline__ void lin2(int o, int n, int& i0, int& i1, float& w1) {
  float src = 0.5f * (float)o - 0.25f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  w1 = src - (float)i0;
}

__global__ void upsample3d_kernel(const float* __restrict__ x, float* __restrict__ out, long total4, int t_in, int h, int w, int c, int mode) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int cq = c / 4, ho = 2 * h, wo = 2 * w;
  const int t_out = mode == 1 ? 1 + 2 * (t_in - 1) : t_in;
  const int c4 = (int)(idx % cq) * 4;
  long r = idx / cq;
  const int xo = (int)(r % wo);
  r /= wo;
  const int yo = (int)(r % ho);
  r /= ho;
  const int to = (int)(r % t_out);
  const long b = r / t_out;
  const float* base = x + b * (long)t_in * h * w * c + c4;
  auto at = [&](int tt, int yy, int xx) { return *reinterpret_cast<const f4*>(base + (((long)tt * h + yy) * w + xx) * c); };
  f4 o;
  if (mode == 0) {
    o = at(to, yo >> 1, xo >> 1);
  } else {
    int y0, y1, x0, x1;
    float wy, wx;
    lin2(yo, h, y0, y1, wy);
    lin2(xo, w, x0, x1, wx);
    auto plane = [&](int tt) {
      const f4 a = at(tt, y0, x0) * (1.f - wx) + at(tt, y0, x1) * wx;
      const f4 bq = at(tt, y1, x0) * (1.f - wx) + at(tt, y1, x1) * wx;
      return a * (1.f - wy) + bq * wy;
    };
    if (to == 0) {
      o = plane(0);  // the first frame is upsampled in space only
    } else {
      int t0, t1;
      float wt;
      lin2(to - 1, t_in - 1, t0, t1, wt);  // frames 1.. form their own sequence of t_in - 1 samples
      o = plane(1 + t0) * (1.f - wt) + plane(1 + t1) * wt;
    }
  }
  *reinterpret_cast<f4*>(out + idx * 4) = o;
}

// one wave per row; n <= 64 * 64 columns
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16* __restrict__ p, long rows, int n, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* sr = s + row * n;
  float m = -INFINITY;
  for (int j = lane; j < n; j += 64) m = fmaxf(m, sr[j] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) sum += __expf(sr[j] * scale - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  for (int j = lane; j < n; j += 64) p[row * n + j] = f2bf(__expf(sr[j] * scale - m) * inv);
}

}  // namespace
}  // namespace dfot

extern "C" {
using namespace dfot;

int64_t dfot_op_groupnorm_scratch_floats(int bt, int pixels) { return (int64_t)bt * gn_partial_blocks(pixels) * 64 + (int64_t)bt * 64; }

int dfot_op_groupnorm(const float* x, const float* gamma, const float* beta, float eps, void* out, float* scratch, int bt, int pixels,
                      int channels, int silu, void* stream) {
  DFOT_REQUIRE(x && gamma && beta && out && scratch, DFOT_ERR_ARG, "op_groupnorm: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = gn_partial_blocks(pixels);
  float* partial = scratch;
  float* stats = scratch + (long)bt * nblk * 64;
  int rc = launch_gn_partial_f32(x, partial, bt, pixels, channels, s);
  if (rc) return rc;
  if ((rc = launch_gn_finalize(partial, stats, bt, nblk, pixels, channels, eps, s))) return rc;
  const long total4 = (long)bt * pixels * (channels / 4);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, s, x, stats, gamma, beta, (bf16*)out, total4, pixels, channels, silu);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

int dfot_op_frame_shift(const void* x, void* out, int batch, int frames, int64_t frame_elems, int shift, void* stream) {
  DFOT_REQUIRE(x && out && x != out, DFOT_ERR_ARG, "op_frame_shift: null or aliased argument");
  DFOT_REQUIRE(frame_elems % 8 == 0 && shift >= 0, DFOT_ERR_SHAPE, "op_frame_shift: frame elements %ld must be a multiple of 8", (long)frame_elems);
  const long total8 = (long)batch * frames * (frame_elems / 8);
  hipLaunchKernelGGL(frame_shift_kernel, dim3(cdiv(total8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, total8, frames,
                     (long)(frame_elems / 8), shift);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

int dfot_op_upsample3d(const float* x, float* out, int batch, int frames, int h, int w, int channels, int mode, void* stream) {
  DFOT_REQUIRE(x && out, DFOT_ERR_ARG, "op_upsample3d: null argument");
  DFOT_REQUIRE(channels % 4 == 0 && (mode == 0 || (mode == 1 && frames >= 1)), DFOT_ERR_SHAPE, "op_upsample3d: channels %% 4, mode in {0,1}");
  const int t_out = mode == 1 ? 1 + 2 * (frames - 1) : frames;
  const long total4 = (long)batch * t_out * (2 * h) * (2 * w) * (channels / 4);
  hipLaunchKernelGGL(upsample3d_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, out, total4, frames, h, w, channels, mode);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

int dfot_op_softmax_rows(const float* scores, void* probs, int64_t rows, int n, float scale, void* stream) {
  DFOT_REQUIRE(scores && probs && rows > 0 && n > 0, DFOT_ERR_ARG, "op_softmax_rows: bad argument");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, scores, (bf16*)probs, (long)rows, n, scale);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

}  // extern "C"
