// HBM-bound kernels of the DFoT backbone: embeddings, norms, resampling, weight packing (gfx950).
// All activations are channels-last ("[frame][pixel][channel]"), so the reference's
// "(b t) c h w -> b (t h w) c" rearranges at the transformer levels are no-ops here.
#include "kernels.h"
#include "dfot_hip.h"

namespace dfot {

typedef __attribute__((ext_vector_type(4))) float float4v;
typedef __attribute__((ext_vector_type(2))) float float2v;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// --------------------------------------------------------------------------------------------
// noise-level embedding: Fourier features -> Linear -> SiLU -> Linear  (embeddings.py:67-110)
// two launches; one wave per (token, output row) so the weight rows are read coalesced and the
// grid (tokens x E / 4 workgroups) fills the chip
// --------------------------------------------------------------------------------------------
template <bool FOURIER>
__global__ __launch_bounds__(256) void noise_mlp_kernel(const float* __restrict__ in, const float* __restrict__ freqs,
                                                        const float* __restrict__ phases, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ out, int kdim,
                                                        int e) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  const int bt = blockIdx.y;
  if (row >= e) return;
  float acc = 0.f;
  for (int i = lane; i < kdim; i += 64) {
    float f;
    if constexpr (FOURIER) {
      f = cosf(__fadd_rn(__fmul_rn(in[bt], freqs[i]), phases[i])) * 1.41421356237309515f;
    } else {
      f = in[(long)bt * kdim + i];
    }
    acc += w[(long)row * kdim + i] * f;
  }
  acc = wave_sum(acc);
  if (lane == 0) out[(long)bt * e + row] = FOURIER ? silu_f(acc + b[row]) : acc + b[row];
}

int launch_noise_emb(const float* k, const float* freqs, const float* phases, const float* w1, const float* b1,
                     const float* w2, const float* b2, float* hidden, float* out, int bt, int ndim, int e, hipStream_t s) {
  hipLaunchKernelGGL(noise_mlp_kernel<true>, dim3(cdiv(e, 4), bt), dim3(256), 0, s, k, freqs, phases, w1, b1, hidden, ndim, e);
  DFOT_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(noise_mlp_kernel<false>, dim3(cdiv(e, 4), bt), dim3(256), 0, s, hidden, nullptr, nullptr, w2, b2, out, e, e);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// The residual stream of the ResBlock levels is fp32 (training ops, tests of single ops) or bf16 (the inference engine: what
// torch.autocast(bf16) keeps there in the reference): stream accessors for either element type
template <typename T>
__device__ __forceinline__ void stream_ld8(const T* p, float (&v)[8]) {
  if constexpr (sizeof(T) == 4) {
    const float4v a = *reinterpret_cast<const float4v*>(p), b = *reinterpret_cast<const float4v*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
  } else {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf2f(a[j]);
  }
}
template <typename T>
__device__ __forceinline__ float4v stream_ld4(const T* p) {
  if constexpr (sizeof(T) == 4) {
    return *reinterpret_cast<const float4v*>(p);
  } else {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
    return float4v{bf2f(a[0]), bf2f(a[1]), bf2f(a[2]), bf2f(a[3])};
  }
}
template <typename T>
__device__ __forceinline__ void stream_st8(T* p, const float (&v)[8]) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4v*>(p) = float4v{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4v*>(p + 4) = float4v{v[4], v[5], v[6], v[7]};
  } else {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(v[j]);
    *reinterpret_cast<bf16x8*>(p) = o;
  }
}
template <typename T>
__device__ __forceinline__ void stream_st4(T* p, float4v v) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4v*>(p) = v;
  } else {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
  }
}

// --------------------------------------------------------------------------------------------
// EmbedInput: conv k2 s2, Cin(3) -> C0, NCHW fp32 in, channels-last fp32 out (u_vit_blocks.py:16-30)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_input_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ out, long total4,
                                                          int res, int cin, int c0) {
  extern __shared__ float wt[];  // [cin*4][c0] : tap-major so that consecutive channels are consecutive addresses
  const int taps = cin * 4;
  for (int i = threadIdx.x; i < taps * c0; i += blockDim.x) {
    const int c = i % c0, t = i / c0;
    wt[i] = w[(long)c * taps + t];
  }
  __syncthreads();
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const unsigned q = c0 / 4, r0 = res / 2, iu = (unsigned)idx;  // 32-bit index arithmetic (launcher: total < 2^31)
  const int c4 = (int)(iu % q) * 4;
  const unsigned pix = iu / q;
  const int px = (int)(pix % r0), py = (int)((pix / r0) % r0);
  const long bt = pix / (r0 * r0);
  float4v acc = *reinterpret_cast<const float4v*>(b + c4);
  for (int ci = 0; ci < cin; ++ci) {
    const float* xp = x + ((bt * cin + ci) * res + 2 * py) * (long)res + 2 * px;
    const float2v top = *reinterpret_cast<const float2v*>(xp);
    const float2v bot = *reinterpret_cast<const float2v*>(xp + res);
    const float* wp = wt + (ci * 4) * c0 + c4;
    acc += *reinterpret_cast<const float4v*>(wp) * top[0] + *reinterpret_cast<const float4v*>(wp + c0) * top[1] +
           *reinterpret_cast<const float4v*>(wp + 2 * c0) * bot[0] + *reinterpret_cast<const float4v*>(wp + 3 * c0) * bot[1];
  }
  *reinterpret_cast<float4v*>(out + (long)pix * c0 + c4) = acc;
}

// the same with the 4 x (CIN * 4) weights of the thread's channel quad in registers: the thread count is a multiple of c0 / 4, so a
// thread keeps its channels over the grid-stride loop and no workgroup stages the weights through LDS (32768 workgroups each
// re-reading 6 KiB of weights and synchronising was most of the 94 us this took for 134 MB of output)
template <int CIN, typename TOUT>
__global__ __launch_bounds__(256) void embed_input_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, TOUT* __restrict__ out, unsigned total4,
                                                              int res, int c0) {
  constexpr int TAPS = CIN * 4;
  const unsigned q = c0 / 4, r0 = res / 2;
  const unsigned nthreads = gridDim.x * 256u;
  unsigned iu = blockIdx.x * 256u + threadIdx.x;
  const int c4 = (int)(iu % q) * 4;
  float4v wt[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) wt[t][j] = w[(long)(c4 + j) * TAPS + t];
  const float4v bias = *reinterpret_cast<const float4v*>(b + c4);
  for (; iu < total4; iu += nthreads) {
    const unsigned pix = iu / q;
    const int px = (int)(pix % r0), py = (int)((pix / r0) % r0);
    const long bt = pix / (r0 * r0);
    float4v acc = bias;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float* xp = x + ((bt * CIN + ci) * res + 2 * py) * (long)res + 2 * px;
      const float2v top = *reinterpret_cast<const float2v*>(xp);
      const float2v bot = *reinterpret_cast<const float2v*>(xp + res);
      acc += wt[ci * 4] * top[0] + wt[ci * 4 + 1] * top[1] + wt[ci * 4 + 2] * bot[0] + wt[ci * 4 + 3] * bot[1];
    }
    stream_st4(out + (long)pix * c0 + c4, acc);
  }
}

int launch_embed_input(const float* x, const float* w, const float* b, float* out, int bt, int res, int cin, int c0,
                       hipStream_t s) {
  DFOT_REQUIRE(c0 % 4 == 0 && res % 2 == 0, DFOT_ERR_SHAPE, "embed_input: channels %d / resolution %d unsupported", c0, res);
  const long total4 = (long)bt * (res / 2) * (res / 2) * (c0 / 4);
  DFOT_REQUIRE(total4 < (1L << 31), DFOT_ERR_SHAPE, "embed_input: %ld work items exceed the 32-bit index range", total4);
  if (cin == 3 && 256 % (c0 / 4) == 0) {
    const long wgs = cdiv(total4, 256);
    hipLaunchKernelGGL((embed_input_reg_kernel<3, float>), dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, s, x, w, b, out, (unsigned)total4, res, c0);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  }
  hipLaunchKernelGGL(embed_input_kernel, dim3(cdiv(total4, 256)), dim3(256), (size_t)cin * 4 * c0 * sizeof(float), s, x, w, b,
                     out, total4, res, cin, c0);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// output in the bf16 residual stream (the inference engine; three input channels, the register-weight form)
int launch_embed_input_bf16(const float* x, const float* w, const float* b, bf16* out, int bt, int res, int cin, int c0, hipStream_t s) {
  DFOT_REQUIRE(c0 % 4 == 0 && res % 2 == 0 && cin == 3 && 256 % (c0 / 4) == 0, DFOT_ERR_SHAPE,
               "embed_input (bf16 stream): %d input channels / %d channels / resolution %d unsupported", cin, c0, res);
  const long total4 = (long)bt * (res / 2) * (res / 2) * (c0 / 4);
  DFOT_REQUIRE(total4 < (1L << 31), DFOT_ERR_SHAPE, "embed_input: %ld work items exceed the 32-bit index range", total4);
  const long wgs = cdiv(total4, 256);
  hipLaunchKernelGGL((embed_input_reg_kernel<3, bf16>), dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, s, x, w, b, out, (unsigned)total4, res, c0);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// --------------------------------------------------------------------------------------------
// pose conditioning repack: cond [BT][cdim][res][res] fp32 -> patch rows A[(bt,py,px)][k] bf16,
// k = c*4 + dy*2 + dx (the flatten order of the Conv2d weight), rows padded to kpad.
// one workgroup = one patch row py and a chunk of CC channels; transposed through LDS
// --------------------------------------------------------------------------------------------
constexpr int RP_CC = 20;
// workgroup = (20-channel chunk, patch row, frame): 16-byte loads along x (two patches' worth of one channel row), the patch rows
// assembled in LDS (row pitch 84 elements: 8-byte aligned), 8-byte stores of the 80 contiguous K entries of every patch
// (the scalar form of this kernel moved 0.43 TB/s: 1.8 ms per window at 16 frames)
__global__ __launch_bounds__(256) void cond_repack_kernel(const float* __restrict__ cond, bf16* __restrict__ a, int res,
                                                          int cdim, int kpad) {
  extern __shared__ __attribute__((aligned(16))) bf16 tile[];  // [r0][RP_CC*4 + 4]
  const int r0 = res / 2;
  constexpr int ld = RP_CC * 4 + 4;
  const int chunk = blockIdx.x, py = blockIdx.y, bt = blockIdx.z;
  const int c0 = chunk * RP_CC;
  const int xq = res / 4;  // float4 groups per image row
  for (int e = threadIdx.x; e < RP_CC * 2 * xq; e += blockDim.x) {
    const int x4 = e % xq;
    const int dy = (e / xq) & 1;
    const int cc = e / (2 * xq);
    const float4v v = *reinterpret_cast<const float4v*>(cond + (((long)bt * cdim + c0 + cc) * res + 2 * py + dy) * res + x4 * 4);
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 lo, hi;
    lo[0] = f2bf(v[0]); lo[1] = f2bf(v[1]); hi[0] = f2bf(v[2]); hi[1] = f2bf(v[3]);
    *reinterpret_cast<bf16x2*>(tile + (2 * x4) * ld + cc * 4 + dy * 2) = lo;
    *reinterpret_cast<bf16x2*>(tile + (2 * x4 + 1) * ld + cc * 4 + dy * 2) = hi;
  }
  __syncthreads();
  constexpr int kq = RP_CC;  // bf16x4 groups per patch
  for (int e = threadIdx.x; e < r0 * kq; e += blockDim.x) {
    const int px = e / kq, k4 = e % kq;
    *reinterpret_cast<bf16x4*>(a + ((long)(bt * r0 + py) * r0 + px) * kpad + c0 * 4 + k4 * 4) = *reinterpret_cast<const bf16x4*>(tile + px * ld + k4 * 4);
  }
}

int launch_cond_repack(const float* cond, bf16* a, int bt, int res, int cdim, int kpad, hipStream_t s) {
  DFOT_REQUIRE(cdim % RP_CC == 0 && res % 4 == 0 && kpad % 4 == 0, DFOT_ERR_SHAPE, "cond_repack: cond dim %d must be a multiple of %d (resolution %d, K %d of 4)",
               cdim, RP_CC, res, kpad);
  const int r0 = res / 2;
  const size_t lds = (size_t)r0 * (RP_CC * 4 + 4) * sizeof(bf16);
  hipLaunchKernelGGL(cond_repack_kernel, dim3(cdim / RP_CC, r0, bt), dim3(256), lds, s, cond, a, res, cdim, kpad);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// --------------------------------------------------------------------------------------------
// embedding pyramid: avg_pool2d(emb, 2^l) for l = 1,2,3 from the level-0 map (u_vit3d_pose.py:99-107)
// thread = (8x8 level-0 pixel block, 8 channels)
// --------------------------------------------------------------------------------------------
__global__ void emb_pyramid_kernel(const bf16* __restrict__ e0, bf16* __restrict__ e1, bf16* __restrict__ e2,
                                   bf16* __restrict__ e3, long total, int r0, int e) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ec = e / 8;
  const int c8 = (int)((unsigned)idx % (unsigned)ec);
  const long blk = (unsigned)idx / (unsigned)ec;
  const int r3 = r0 / 8;
  const int bx = (int)(blk % r3), by = (int)((blk / r3) % r3);
  const long bt = blk / ((long)r3 * r3);
  float s3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int qy = 0; qy < 2; ++qy)
    for (int qx = 0; qx < 2; ++qx) {
      float s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int hy = 0; hy < 2; ++hy)
        for (int hx = 0; hx < 2; ++hx) {
          float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
              const int y = by * 8 + qy * 4 + hy * 2 + py, x = bx * 8 + qx * 4 + hx * 2 + px;
              const bf16x8 v = *reinterpret_cast<const bf16x8*>(e0 + ((bt * r0 + y) * r0 + x) * e + c8 * 8);
#pragma unroll
              for (int j = 0; j < 8; ++j) s1[j] += bf2f(v[j]);
            }
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            o[j] = f2bf(s1[j] * 0.25f);
            s2[j] += s1[j];
          }
          const int y1 = by * 4 + qy * 2 + hy, x1 = bx * 4 + qx * 2 + hx;
          *reinterpret_cast<bf16x8*>(e1 + ((bt * (r0 / 2) + y1) * (r0 / 2) + x1) * e + c8 * 8) = o;
        }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = f2bf(s2[j] * (1.f / 16.f));
        s3[j] += s2[j];
      }
      const int y2 = by * 2 + qy, x2 = bx * 2 + qx;
      *reinterpret_cast<bf16x8*>(e2 + ((bt * (r0 / 4) + y2) * (r0 / 4) + x2) * e + c8 * 8) = o;
    }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(s3[j] * (1.f / 64.f));
  *reinterpret_cast<bf16x8*>(e3 + ((bt * r3 + by) * r3 + bx) * e + c8 * 8) = o;
}

int launch_emb_pyramid(const bf16* emb0, bf16* emb1, bf16* emb2, bf16* emb3, int bt, int r0, int e, hipStream_t s) {
  DFOT_REQUIRE(r0 % 8 == 0 && e % 8 == 0, DFOT_ERR_SHAPE, "emb_pyramid: level-0 size %d / emb %d must be multiples of 8", r0, e);
  const long total = (long)bt * (r0 / 8) * (r0 / 8) * (e / 8);
  DFOT_REQUIRE(total < (1L << 31), DFOT_ERR_SHAPE, "emb_pyramid: %ld work items exceed the 32-bit index range", total);
  hipLaunchKernelGGL(emb_pyramid_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, emb0, emb1, emb2, emb3, total, r0, e);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// --------------------------------------------------------------------------------------------
// ProjectOutput: conv_transpose k2 s2, C0 -> cout(3), channels-last in, NCHW out (u_vit_blocks.py:33-49)
// weight layout [C0][cout][2][2]; thread = one level-0 pixel, 4*cout outputs
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void project_output_kernel(const float* __restrict__ x0, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ out,
                                                             long npix, int res, int c0, int cout) {
  extern __shared__ float wl[];  // [c0][cout*4]
  const int nw = c0 * cout * 4;
  for (int i = threadIdx.x; i < nw; i += blockDim.x) wl[i] = w[i];
  __syncthreads();
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int r0 = res / 2;
  const unsigned pu = (unsigned)pix, ru = (unsigned)r0;
  const int px = (int)(pu % ru), py = (int)((pu / ru) % ru);
  const long bt = pu / (ru * ru);
  float acc[12];
  for (int o = 0; o < cout * 4; ++o) acc[o] = 0.f;
  const float* xp = x0 + pix * c0;
  for (int c = 0; c < c0; c += 4) {
    const float4v xv = *reinterpret_cast<const float4v*>(xp + c);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      for (int o = 0; o < cout * 4; ++o) acc[o] += xv[j] * wl[(c + j) * cout * 4 + o];
  }
  for (int co = 0; co < cout; ++co)
    for (int dy = 0; dy < 2; ++dy) {
      float2v v;
      v[0] = acc[co * 4 + dy * 2 + 0] + b[co];
      v[1] = acc[co * 4 + dy * 2 + 1] + b[co];
      *reinterpret_cast<float2v*>(out + ((bt * cout + co) * res + 2 * py + dy) * (long)res + 2 * px) = v;
    }
}

// one step of a transposed reduction: lanes whose bit BIT is set keep values W..2W-1, the others 0..W-1, and each adds its
// partner's copy of what it keeps -- W values per lane remain, each summed over one more lane bit
template <int W, int BIT>
__device__ __forceinline__ void halve_exchange(float* acc, int lane) {
  const bool hi = lane & BIT;
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const float keep = hi ? acc[j + W] : acc[j], send = hi ? acc[j] : acc[j + W];
    acc[j] = keep + __shfl_xor(send, BIT);
  }
}

// C0 = 128: sixteen lanes share a pixel, each with 8 of its channels (one coalesced 512-byte row per pixel instead of 64 lanes
// striding 512 bytes apart) and the 8 x (COUT * 4) weights of those channels in registers; the COUT * 4 <= 16 partial sums are
// reduced over the 16 lanes with a halving exchange (15 shuffles), which leaves output o = lane % 16 in its lane
template <int COUT, typename TIN>
__global__ __launch_bounds__(256) void project_output_c128_kernel(const TIN* __restrict__ x0, const float* __restrict__ w,
                                                                  const float* __restrict__ b, float* __restrict__ out, unsigned npix,
                                                                  int res, const uint8_t* __restrict__ live_frames) {
  constexpr int NO = COUT * 4;
  const int lane = threadIdx.x & 63, sub = lane & 15;
  float wt[8][NO];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int o = 0; o < NO; ++o) wt[c][o] = w[(sub * 8 + c) * NO + o];
  const float bias = sub < NO ? b[sub >> 2] : 0.f;
  const unsigned r0 = res / 2;
  const unsigned group = (blockIdx.x * 256u + threadIdx.x) >> 4, ngroups = gridDim.x * 16u;
  const unsigned iters = (npix + ngroups - 1) / ngroups;  // the same trip count for every lane: the shuffles need whole waves
  for (unsigned it = 0; it < iters; ++it) {
    const unsigned pix = group + it * ngroups;
    const bool live = pix < npix;
    // frames whose output the caller discards (context tokens of the sampler): zeros are written, nothing is read
    const bool dead = live && live_frames && !live_frames[pix / (r0 * r0)];
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.f;
    if (live && !dead) {
      float xv[8];
      stream_ld8(x0 + (long)pix * 128 + sub * 8, xv);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] += xv[j] * wt[j][o] + xv[4 + j] * wt[4 + j][o];
    }
    // halve the value count while exchanging across lane bits 3, 2, 1, 0
    halve_exchange<8, 8>(acc, lane);
    halve_exchange<4, 4>(acc, lane);
    halve_exchange<2, 2>(acc, lane);
    halve_exchange<1, 1>(acc, lane);
    if (live && sub < NO) {
      const int px = (int)(pix % r0), py = (int)((pix / r0) % r0);
      const long bt = pix / (r0 * r0);
      const int co = sub >> 2, dy = (sub >> 1) & 1, dx = sub & 1;
      out[((bt * COUT + co) * res + 2 * py + dy) * (long)res + 2 * px + dx] = dead ? 0.f : acc[0] + bias;
    }
  }
}

int launch_project_output(const float* x0, const float* w, const float* b, float* out, int bt, int res, int c0, int cout,
                          hipStream_t s, const uint8_t* live) {
  DFOT_REQUIRE(cout <= 3 && c0 % 4 == 0, DFOT_ERR_SHAPE, "project_output: cout=%d (<=3), c0=%d", cout, c0);
  const long npix = (long)bt * (res / 2) * (res / 2);
  DFOT_REQUIRE(npix < (1L << 31), DFOT_ERR_SHAPE, "project_output: %ld pixels exceed the 32-bit index range", npix);
  if (c0 == 128 && cout == 3) {
    const long wgs = cdiv(npix, 16);
    hipLaunchKernelGGL((project_output_c128_kernel<3, float>), dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, s, x0, w, b, out, (unsigned)npix, res, live);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  }
  hipLaunchKernelGGL(project_output_kernel, dim3(cdiv(npix, 256)), dim3(256), c0 * cout * 4 * sizeof(float), s, x0, w, b,
                     out, npix, res, c0, cout);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// input from the bf16 residual stream (the inference engine: 128 level-0 channels, 3 output channels)
int launch_project_output_bf16(const bf16* x0, const float* w, const float* b, float* out, int bt, int res, int c0, int cout, hipStream_t s,
                               const uint8_t* live) {
  DFOT_REQUIRE(c0 == 128 && cout == 3, DFOT_ERR_SHAPE, "project_output (bf16 stream): c0=%d cout=%d unsupported", c0, cout);
  const long npix = (long)bt * (res / 2) * (res / 2);
  DFOT_REQUIRE(npix < (1L << 31), DFOT_ERR_SHAPE, "project_output: %ld pixels exceed the 32-bit index range", npix);
  const long wgs = cdiv(npix, 16);
  hipLaunchKernelGGL((project_output_c128_kernel<3, bf16>), dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, s, x0, w, b, out, (unsigned)npix, res, live);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// --------------------------------------------------------------------------------------------
// GroupNorm(32 groups) statistics over channels-last tensors, deterministic two-stage reduction
// --------------------------------------------------------------------------------------------
constexpr int GN_PIX_PER_BLOCK = 128;  // (512 left the 64x64 level with 8 x 16 = 128 workgroups for 67 MB: 65 us; 128-pixel blocks: 512 and 2048 workgroups)
int gn_partial_blocks(int pixels) { return cdiv(pixels, GN_PIX_PER_BLOCK); }

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                         int pixels, int c) {
  __shared__ float red_s[256], red_q[256];
  const int bt = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int quads = c / 4;                 // 4-channel quads per pixel
  const int ppass = 256 / quads;           // pixels per pass
  const int q = threadIdx.x % quads, pp = threadIdx.x / quads;
  const int cpg = c / 32;                  // channels per group (>= 4)
  const int p0 = blk * GN_PIX_PER_BLOCK;
  const int p1 = min(p0 + GN_PIX_PER_BLOCK, pixels);
  float s = 0.f, ss = 0.f;
  {
    for (int p = p0 + pp; p < p1; p += ppass) {
      const T* src = x + ((long)bt * pixels + p) * c + q * 4;
      float v[4];
      if constexpr (sizeof(T) == 4) {
        const float4v t = *reinterpret_cast<const float4v*>(src);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
      } else {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(src);
        v[0] = bf2f(t[0]); v[1] = bf2f(t[1]); v[2] = bf2f(t[2]); v[3] = bf2f(t[3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s += v[j];
        ss += v[j] * v[j];
      }
    }
  }
  // deterministic in-block reduction: every group has exactly 8 contributing threads
  // ((cpg/4) quads x ppass pixel-lanes); one thread per (group, moment) sums them in fixed order
  red_s[threadIdx.x] = s;
  red_q[threadIdx.x] = ss;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int g = threadIdx.x & 31, which = threadIdx.x >> 5;
    const float* r = which ? red_q : red_s;
    const int qpg = cpg / 4;  // quads per group
    float t = 0.f;
    for (int pl = 0; pl < ppass; ++pl)
      for (int qq = 0; qq < qpg; ++qq) t += r[pl * quads + g * qpg + qq];
    partial[(((long)bt * nblk + blk) * 32 + g) * 2 + which] = t;
  }
}

// four workgroups per image, eight groups each: 32 threads per group sum nblk/32 partials each (8-byte loads) in a fixed order,
// then a fixed shuffle tree -- deterministic; 64 workgroups instead of 16 (one per image, 8 threads per group: 7.5 us per call, 24 calls
// per forward)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                          int nblk, float inv_count, float eps) {
  const int bt = blockIdx.x >> 2, g = (blockIdx.x & 3) * 8 + (threadIdx.x >> 5), part = threadIdx.x & 31;
  float s = 0.f, ss = 0.f;
  for (int b = part; b < nblk; b += 32) {
    const float2 v = *reinterpret_cast<const float2*>(partial + (((long)bt * nblk + b) * 32 + g) * 2);
    s += v.x;
    ss += v.y;
  }
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    s += __shfl_xor(s, o);
    ss += __shfl_xor(ss, o);
  }
  if (part == 0) {
    const float mean = s * inv_count;
    const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
    stats[(bt * 32 + g) * 2 + 0] = mean;
    stats[(bt * 32 + g) * 2 + 1] = rsqrtf(var + eps);
  }
}

int launch_gn_finalize(const float* partial, float* stats, int bt, int nblk, int pixels, int c, float eps, hipStream_t s) {
  const float inv = 1.f / ((float)pixels * (float)(c / 32));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(bt * 4), dim3(256), 0, s, partial, stats, nblk, inv, eps);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

template <typename T>
static int launch_gn_partial_t(const T* x, float* partial, int bt, int pixels, int c, hipStream_t s) {
  DFOT_REQUIRE(c == 128 || c == 256 || c == 512 || c == 1024, DFOT_ERR_SHAPE, "group_norm: channels %d not in {128,256,512,1024}", c);
  hipLaunchKernelGGL(gn_partial_kernel<T>, dim3(gn_partial_blocks(pixels), bt), dim3(256), 0, s, x, partial, pixels, c);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int launch_gn_partial_f32(const float* x, float* partial, int bt, int pixels, int c, hipStream_t s) {
  return launch_gn_partial_t<float>(x, partial, bt, pixels, c, s);
}
int launch_gn_partial_bf16(const bf16* x, float* partial, int bt, int pixels, int c, hipStream_t s) {
  return launch_gn_partial_t<bf16>(x, partial, bt, pixels, c, s);
}

// GroupNorm apply + SiLU, fp32 in -> bf16 out (feeds the 3x3 conv's A operand)
__global__ void gn_apply_silu_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     bf16* __restrict__ out, long total8, int pixels, int c, const uint8_t* __restrict__ live) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total8) return;
  const unsigned iu = (unsigned)idx, cq = (unsigned)(c / 8);  // 32-bit index arithmetic (launcher: total < 2^31)
  const int c8 = (int)(iu % cq);
  const long pix = iu / cq;
  const int bt = (int)((unsigned)pix / (unsigned)pixels);
  if (live && !live[bt]) return;  // a frame whose output is discarded: nothing read, nothing written
  const int cpg = c / 32;
  const float* src = x + pix * c + c8 * 8;
  const float4v a = *reinterpret_cast<const float4v*>(src);
  const float4v b = *reinterpret_cast<const float4v*>(src + 4);
  float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = c8 * 8 + j;
    const int g = ch / cpg;
    const float mean = stats[(bt * 32 + g) * 2], rs = stats[(bt * 32 + g) * 2 + 1];
    const float y = (v[j] - mean) * rs * gamma[ch] + beta[ch];
    o[j] = f2bf(silu_f(y));
  }
  *reinterpret_cast<bf16x8*>(out + pix * c + c8 * 8) = o;
}

// The same, laid out for streaming: a thread keeps ITS 8 channels for the whole workgroup -- GroupNorm statistics, gamma, beta (and the
// per-frame FiLM vector for gn_film_silu_rows_kernel below) are fetched ONCE with 16-byte loads into registers -- and walks down
// GN_ROWS_IT pixels, so that per pixel it issues only the 16-byte loads of the streams themselves (the one-pixel-per-thread form issued
// ~40 four-byte loads of cached constants next to its 3 stream loads and sat at 2.9 TB/s).  Workgroup = 256 / (C / 8) pixels per
// iteration x GN_ROWS_IT iterations of ONE frame; arithmetic and its order are those of the kernels above (bit-identical results).
constexpr int GN_ROWS_IT = 8;
template <int C>
struct GnRowConsts {
  float mean[8], rsg[8], gam[8], bet[8];
  __device__ __forceinline__ void load(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, int bt,
                                       int c0) {
    constexpr int CPG = C / 32;  // channels per group: 4 (C = 128) or 8 (C = 256): this thread's 8 channels lie in 2 or 1 groups
    const float4v g0 = *reinterpret_cast<const float4v*>(gamma + c0), g1 = *reinterpret_cast<const float4v*>(gamma + c0 + 4);
    const float4v b0 = *reinterpret_cast<const float4v*>(beta + c0), b1 = *reinterpret_cast<const float4v*>(beta + c0 + 4);
    const float* st = stats + ((long)bt * 32 + c0 / CPG) * 2;
    float m0, r0, m1, r1;
    if constexpr (CPG == 4) {
      const float4v t = *reinterpret_cast<const float4v*>(st);
      m0 = t[0], r0 = t[1], m1 = t[2], r1 = t[3];
    } else {
      m0 = m1 = st[0];
      r0 = r1 = st[1];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mean[j] = j < 4 ? m0 : m1;
      rsg[j] = j < 4 ? r0 : r1;
      gam[j] = j < 4 ? g0[j] : g1[j - 4];
      bet[j] = j < 4 ? b0[j] : b1[j - 4];
    }
  }
  __device__ __forceinline__ float norm(float v, int j) const { return (v - mean[j]) * rsg[j] * gam[j] + bet[j]; }
};

template <int C, typename TIN>
__global__ __launch_bounds__(256) void gn_apply_silu_rows_kernel(const TIN* __restrict__ x, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 bf16* __restrict__ out, int pixels, const uint8_t* __restrict__ live) {
  constexpr int TPP = C / 8, PPI = 256 / TPP;
  const int wg_per_bt = pixels / (PPI * GN_ROWS_IT);
  const int bt = blockIdx.x / wg_per_bt;
  if (live && !live[bt]) return;  // a frame whose output is discarded: nothing read, nothing written
  const int c0 = (threadIdx.x % TPP) * 8;
  GnRowConsts<C> k;
  k.load(stats, gamma, beta, bt, c0);
  long pix = (long)bt * pixels + (long)(blockIdx.x % wg_per_bt) * (PPI * GN_ROWS_IT) + threadIdx.x / TPP;
#pragma unroll 4
  for (int it = 0; it < GN_ROWS_IT; ++it, pix += PPI) {
    float v[8];
    stream_ld8(x + pix * C + c0, v);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(silu_f(k.norm(v[j], j)));
    *reinterpret_cast<bf16x8*>(out + pix * C + c0) = o;
  }
}

// bf16 input stream (the inference engine's ResBlock levels): the row-streaming form only (its shape conditions are the engine's)
int launch_gn_apply_silu_bf16in(const bf16* x, const float* stats, const float* gamma, const float* beta, bf16* out, int bt, int pixels, int c,
                                hipStream_t s, const uint8_t* live) {
  DFOT_REQUIRE((c == 128 || c == 256) && pixels % ((256 / (c / 8)) * GN_ROWS_IT) == 0, DFOT_ERR_SHAPE,
               "groupnorm apply (bf16 stream): %d channels / %d pixels per frame unsupported", c, pixels);
  const int grid = bt * (pixels / ((256 / (c / 8)) * GN_ROWS_IT));
  if (c == 128) hipLaunchKernelGGL((gn_apply_silu_rows_kernel<128, bf16>), dim3(grid), dim3(256), 0, s, x, stats, gamma, beta, out, pixels, live);
  else hipLaunchKernelGGL((gn_apply_silu_rows_kernel<256, bf16>), dim3(grid), dim3(256), 0, s, x, stats, gamma, beta, out, pixels, live);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

int launch_gn_apply_silu(const float* x, const float* stats, const float* gamma, const float* beta, bf16* out, int bt,
                         int pixels, int c, hipStream_t s, const uint8_t* live) {
  if ((c == 128 || c == 256) && pixels % ((256 / (c / 8)) * GN_ROWS_IT) == 0) {
    const int grid = bt * (pixels / ((256 / (c / 8)) * GN_ROWS_IT));
    if (c == 128) hipLaunchKernelGGL((gn_apply_silu_rows_kernel<128, float>), dim3(grid), dim3(256), 0, s, x, stats, gamma, beta, out, pixels, live);
    else hipLaunchKernelGGL((gn_apply_silu_rows_kernel<256, float>), dim3(grid), dim3(256), 0, s, x, stats, gamma, beta, out, pixels, live);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  }
  const long total8 = (long)bt * pixels * (c / 8);
  DFOT_REQUIRE(total8 < (1L << 31), DFOT_ERR_SHAPE, "groupnorm apply: %ld work items exceed the 32-bit index range", total8);
  hipLaunchKernelGGL(gn_apply_silu_kernel, dim3(cdiv(total8, 256)), dim3(256), 0, s, x, stats, gamma, beta, out, total8,
                     pixels, c, live);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// --------------------------------------------------------------------------------------------
// FiLM conditioning with the pose term cached per window.
//   emb = noise_emb[bt] (per frame) + pose_emb[pixel] (constant over the DDIM steps, zero when the video's
//   external_cond_mask is set), and emb_layer is linear, so
//   (scale|shift)[m] = F[m] + sv[bt],  F = W_film * pose_emb (cached bf16, [M][2C]),  sv = W_film * noise_emb + b.
// Column order of F / sv: within each 64-column group, 32 scale columns then the 32 matching shift columns
// (the row permutation applied to W_film at load time), so 8 consecutive channels read two 16-byte chunks.
// --------------------------------------------------------------------------------------------

// sv for every FiLM projection of the model in ONE launch (FilmChunk table: kernels.h)
__global__ __launch_bounds__(256) void film_vec_kernel(const FilmChunk* __restrict__ table, const float* __restrict__ nemb,
                                                       float* __restrict__ sv, int e, int nbt) {
  // workgroup = 4 rows of one 64-row chunk (blockIdx.x = chunk*16 + part); one wave per row keeps the weight row in
  // registers and loops over all frames, so every weight byte is read once
  const FilmChunk ck = table[blockIdx.x >> 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = (blockIdx.x & 15) * 4 + wave;
  const bf16* wr = ck.w + (long)r * e;
  float wv[2][8];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = lane * 8 + k * 512;
    if (i < e) {
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(wr + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[k][j] = bf2f(t[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[k][j] = 0.f;
    }
  }
  const float bias = ck.b[r];
  for (int bt = 0; bt < nbt; ++bt) {
    const float* ne = nemb + (long)bt * e;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = lane * 8 + k * 512;
      if (i < e) {
        const float4v a = *reinterpret_cast<const float4v*>(ne + i);
        const float4v b = *reinterpret_cast<const float4v*>(ne + i + 4);
        acc += wv[k][0] * a[0] + wv[k][1] * a[1] + wv[k][2] * a[2] + wv[k][3] * a[3] + wv[k][4] * b[0] + wv[k][5] * b[1] +
               wv[k][6] * b[2] + wv[k][7] * b[3];
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) sv[ck.out_off + (long)bt * ck.rows + r] = acc + bias;
  }
}
// E % 256 == 0: the frames' embeddings are staged ONCE per workgroup in LDS (the form above has every wave re-read all of them from
// L2: 64 K waves x 64 KiB = 4 GB for 132 MB of weights) and a wave takes two weight rows at a time, so one LDS read feeds
// two rows; the 2 x 16 dot products of a pass live one per lane pair after a halving exchange over lane bits 5..1 (32 shuffles
// per two rows instead of 96 per row).  Workgroup = 32 rows of a chunk (blockIdx.x = chunk * 2 + half), 16 frames per pass.
template <int EQ>  // E / 256
__global__ __launch_bounds__(256, 2) void film_vec_lds_kernel(const FilmChunk* __restrict__ table, const float* __restrict__ nemb,
                                                           float* __restrict__ sv, int nbt) {
  constexpr int E = EQ * 256;
  __shared__ float4v ne[16][E / 4];
  const FilmChunk ck = table[blockIdx.x >> 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int bt0 = 0; bt0 < nbt; bt0 += 16) {
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * (E / 4); i += 256) {
      const int f = i / (E / 4);
      float4v v = {0.f, 0.f, 0.f, 0.f};
      if (bt0 + f < nbt) v = reinterpret_cast<const float4v*>(nemb + (long)(bt0 + f) * E)[i % (E / 4)];
      ne[f][i % (E / 4)] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
      const int r0 = (blockIdx.x & 1) * 32 + pass * 8 + wave * 2;
      float wv[2][EQ][4];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < EQ; ++k) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(ck.w + (long)(r0 + r) * E + k * 256 + lane * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) wv[r][k][j] = bf2f(t[j]);
        }
      float acc[32];  // [row][frame]
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        float4v x[EQ];
#pragma unroll
        for (int k = 0; k < EQ; ++k) x[k] = ne[f][k * 64 + lane];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < EQ; ++k) a += wv[r][k][0] * x[k][0] + wv[r][k][1] * x[k][1] + wv[r][k][2] * x[k][2] + wv[r][k][3] * x[k][3];
          acc[r * 16 + f] = a;
        }
        if ((f & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most four frames of LDS reads hoisted: all sixteen cost 256 VGPRs
      }
      halve_exchange<16, 32>(acc, lane);
      halve_exchange<8, 16>(acc, lane);
      halve_exchange<4, 8>(acc, lane);
      halve_exchange<2, 4>(acc, lane);
      halve_exchange<1, 2>(acc, lane);
      const float v = acc[0] + __shfl_xor(acc[0], 1);
      // lane >> 1 = row * 16 + frame
      const int r = r0 + (lane >> 5), f = bt0 + ((lane >> 1) & 15);
      if (f < nbt && (lane & 1) == 0) sv[ck.out_off + (long)f * ck.rows + r] = v + ck.b[r];
    }
  }
}
int launch_film_vec(const FilmChunk* table, int chunks, const float* nemb, float* sv, int bt, int e, hipStream_t s) {
  DFOT_REQUIRE(e % 8 == 0 && e <= 1024, DFOT_ERR_SHAPE, "film_vec: emb dim %d must be a multiple of 8 and <= 1024", e);
  if (e == 1024 || e == 512) {
    if (e == 1024) hipLaunchKernelGGL(film_vec_lds_kernel<4>, dim3(chunks * 2), dim3(256), 0, s, table, nemb, sv, bt);
    else hipLaunchKernelGGL(film_vec_lds_kernel<2>, dim3(chunks * 2), dim3(256), 0, s, table, nemb, sv, bt);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  }
  hipLaunchKernelGGL(film_vec_kernel, dim3(chunks * 16), dim3(256), 0, s, table, nemb, sv, e, bt);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// ResBlock: out = SiLU( GroupNorm(h) * (1 + scale) + shift ), bf16 in/out; thread = 8 channels of one pixel
__global__ void gn_film_silu_kernel(const bf16* __restrict__ h, const float* __restrict__ stats,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const bf16* __restrict__ fcache, const float* __restrict__ sv,
                                    const uint8_t* __restrict__ cond_mask, bf16* __restrict__ out, long total8, int pixels,
                                    int c, int tokens, const uint8_t* __restrict__ live) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total8) return;
  const unsigned iu = (unsigned)idx, cq = (unsigned)(c / 8);  // 32-bit index arithmetic (launcher: total < 2^31)
  const int c8 = (int)(iu % cq);
  const long pix = iu / cq;
  const int bt = (int)((unsigned)pix / (unsigned)pixels);
  if (live && !live[bt]) return;
  const int cpg = c / 32;
  const int c0 = c8 * 8;
  const int col = (c0 >> 5) * 64 + (c0 & 31);  // scale columns col..col+7, shift columns col+32..col+39
  const bf16x8 hv = *reinterpret_cast<const bf16x8*>(h + pix * c + c0);
  const float* svp = sv + (long)bt * 2 * c + col;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = svp[j];
    sh[j] = svp[32 + j];
  }
  const bool use_pose = !(cond_mask && cond_mask[(unsigned)bt / (unsigned)tokens]);
  if (use_pose) {
    const bf16x8 fs = *reinterpret_cast<const bf16x8*>(fcache + pix * 2 * c + col);
    const bf16x8 fh = *reinterpret_cast<const bf16x8*>(fcache + pix * 2 * c + col + 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] += bf2f(fs[j]);
      sh[j] += bf2f(fh[j]);
    }
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = c0 + j;
    const int g = ch / cpg;
    const float mean = stats[(bt * 32 + g) * 2], rs = stats[(bt * 32 + g) * 2 + 1];
    const float y = ((bf2f(hv[j]) - mean) * rs * gamma[ch] + beta[ch]) * (1.f + sc[j]) + sh[j];
    o[j] = f2bf(silu_f(y));
  }
  *reinterpret_cast<bf16x8*>(out + pix * c + c0) = o;
}
// streaming form (see gn_apply_silu_rows_kernel): constants of the thread's 8 channels in registers, GN_ROWS_IT pixels per thread
template <int C>
__global__ __launch_bounds__(256) void gn_film_silu_rows_kernel(const bf16* __restrict__ h, const float* __restrict__ stats,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const bf16* __restrict__ fcache, const float* __restrict__ sv,
                                                                const uint8_t* __restrict__ cond_mask, bf16* __restrict__ out, int pixels,
                                                                int tokens, const uint8_t* __restrict__ live) {
  constexpr int TPP = C / 8, PPI = 256 / TPP;
  const int wg_per_bt = pixels / (PPI * GN_ROWS_IT);
  const int bt = blockIdx.x / wg_per_bt;
  if (live && !live[bt]) return;
  const int c0 = (threadIdx.x % TPP) * 8;
  const int col = (c0 >> 5) * 64 + (c0 & 31);  // scale columns col..col+7, shift columns col+32..col+39
  GnRowConsts<C> k;
  k.load(stats, gamma, beta, bt, c0);
  const float* svp = sv + (long)bt * 2 * C + col;
  float svs[8], svh[8];
  {
    const float4v s0 = *reinterpret_cast<const float4v*>(svp), s1 = *reinterpret_cast<const float4v*>(svp + 4);
    const float4v h0 = *reinterpret_cast<const float4v*>(svp + 32), h1 = *reinterpret_cast<const float4v*>(svp + 36);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      svs[j] = j < 4 ? s0[j] : s1[j - 4];
      svh[j] = j < 4 ? h0[j] : h1[j - 4];
    }
  }
  const bool use_pose = !(cond_mask && cond_mask[(unsigned)bt / (unsigned)tokens]);  // workgroup-uniform
  long pix = (long)bt * pixels + (long)(blockIdx.x % wg_per_bt) * (PPI * GN_ROWS_IT) + threadIdx.x / TPP;
  if (use_pose) {
#pragma unroll 4
    for (int it = 0; it < GN_ROWS_IT; ++it, pix += PPI) {
      const bf16x8 hv = *reinterpret_cast<const bf16x8*>(h + pix * C + c0);
      const bf16x8 fs = *reinterpret_cast<const bf16x8*>(fcache + pix * 2 * C + col);
      const bf16x8 fh = *reinterpret_cast<const bf16x8*>(fcache + pix * 2 * C + col + 32);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sc = svs[j] + bf2f(fs[j]), sh = svh[j] + bf2f(fh[j]);
        o[j] = f2bf(silu_f(k.norm(bf2f(hv[j]), j) * (1.f + sc) + sh));
      }
      *reinterpret_cast<bf16x8*>(out + pix * C + c0) = o;
    }
  } else {
#pragma unroll 4
    for (int it = 0; it < GN_ROWS_IT; ++it, pix += PPI) {
      const bf16x8 hv = *reinterpret_cast<const bf16x8*>(h + pix * C + c0);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(silu_f(k.norm(bf2f(hv[j]), j) * (1.f + svs[j]) + svh[j]));
      *reinterpret_cast<bf16x8*>(out + pix * C + c0) = o;
    }
  }
}

int launch_gn_film_silu(const bf16* h, const float* stats, const float* gamma, const float* beta, const bf16* fcache,
                        const float* sv, const uint8_t* cond_mask, bf16* out, int bt, int pixels, int c, int tokens,
                        hipStream_t s, const uint8_t* live) {
  if ((c == 128 || c == 256) && pixels % ((256 / (c / 8)) * GN_ROWS_IT) == 0) {
    const int grid = bt * (pixels / ((256 / (c / 8)) * GN_ROWS_IT));
    if (c == 128)
      hipLaunchKernelGGL(gn_film_silu_rows_kernel<128>, dim3(grid), dim3(256), 0, s, h, stats, gamma, beta, fcache, sv, cond_mask, out, pixels, tokens, live);
    else
      hipLaunchKernelGGL(gn_film_silu_rows_kernel<256>, dim3(grid), dim3(256), 0, s, h, stats, gamma, beta, fcache, sv, cond_mask, out, pixels, tokens, live);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  }
  const long total8 = (long)bt * pixels * (c / 8);
  DFOT_REQUIRE(total8 < (1L << 31), DFOT_ERR_SHAPE, "groupnorm apply: %ld work items exceed the 32-bit index range", total8);
  hipLaunchKernelGGL(gn_film_silu_kernel, dim3(cdiv(total8, 256)), dim3(256), 0, s, h, stats, gamma, beta, fcache, sv,
                     cond_mask, out, total8, pixels, c, tokens, live);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// TransformerBlock: xn = RMSNorm(x) * w * (1 + scale) + shift, fp32 stream in -> bf16 out; one wave per token
// PEND: the residual stream still lacks the previous block's out-projection, left as two K-slice partials: x += bias + s0 + s1 is
// applied here (and written back) instead of in a pass of its own
// TS: element type of the residual stream (fp32, or bf16: the value written back by PEND is then the bf16 rounding of the sum, and the
// norm is taken of that rounded value -- what the next reader of the stream sees)
template <int MAXCH, bool PEND, typename TS>
__global__ __launch_bounds__(256) void rms_film_kernel(const TS* x, const float* __restrict__ w,
                                                       const bf16* __restrict__ fcache, const float* __restrict__ sv,
                                                       const uint8_t* __restrict__ cond_mask, bf16* __restrict__ out,
                                                       long m, int c, int rows_per_bt, int tokens, float eps, TS* xw,
                                                       const float* __restrict__ pbias, const float* __restrict__ p0,
                                                       const float* __restrict__ p1, const float* __restrict__ p2) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const int lane = threadIdx.x & 63;
  const int nch = c / 8;
  const TS* src = x + row * c;
  float v[MAXCH][8];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      stream_ld8(src + ch * 8, v[k]);
      if constexpr (PEND) {
        const long o = row * c + ch * 8;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float4v add = *reinterpret_cast<const float4v*>(pbias + ch * 8 + 4 * hf) + *reinterpret_cast<const float4v*>(p0 + o + 4 * hf) +
                        *reinterpret_cast<const float4v*>(p1 + o + 4 * hf);
          if (p2) add += *reinterpret_cast<const float4v*>(p2 + o + 4 * hf);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[k][4 * hf + j] += add[j];
        }
        stream_st8(xw + o, v[k]);
        if constexpr (sizeof(TS) == 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[k][j] = bf2f(f2bf(v[k][j]));
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[k][j] * v[k][j];
    }
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)c + eps);
  const int bt = (int)(row / rows_per_bt);
  const bool use_pose = !(cond_mask && cond_mask[(unsigned)bt / (unsigned)tokens]);
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int ch = lane + 64 * k;
    if (ch < nch) {
      const int c0 = ch * 8;
      const int col = (c0 >> 5) * 64 + (c0 & 31);
      const float* svp = sv + (long)bt * 2 * c + col;
      float sc[8], sh[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sc[j] = svp[j];
        sh[j] = svp[32 + j];
      }
      if (use_pose) {
        const bf16x8 fs = *reinterpret_cast<const bf16x8*>(fcache + row * 2 * c + col);
        const bf16x8 fh = *reinterpret_cast<const bf16x8*>(fcache + row * 2 * c + col + 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          sc[j] += bf2f(fs[j]);
          sh[j] += bf2f(fh[j]);
        }
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(v[k][j] * rs * w[c0 + j] * (1.f + sc[j]) + sh[j]);
      *reinterpret_cast<bf16x8*>(out + row * c + c0) = o;
    }
  }
}
template <typename TS>
static int launch_rms_film_t(const TS* x, const float* w, const bf16* fcache, const float* sv, const uint8_t* cond_mask, bf16* out,
                             long m, int c, int rows_per_bt, int tokens, float eps, hipStream_t s, const RmsPending* pend) {
  DFOT_REQUIRE(c % 8 == 0 && c <= 8 * 64 * 3, DFOT_ERR_SHAPE, "rms_film: channels %d unsupported", c);
#define RMS_CALL(MC, P)                                                                                                             \
  hipLaunchKernelGGL((rms_film_kernel<MC, P, TS>), dim3(cdiv(m, 4)), dim3(256), 0, s, x, w, fcache, sv, cond_mask, out, m, c, rows_per_bt, \
                     tokens, eps, P ? (TS*)pend->x : nullptr, P ? pend->bias : nullptr, P ? pend->s0 : nullptr, P ? pend->s1 : nullptr, P ? pend->s2 : nullptr)
  if (pend) {
    // x is read, x + bias + slices is normalised AND written to pend->x (the same buffer, or X[l] when the stream still sits in the skip tensor)
    DFOT_REQUIRE(pend->x && pend->bias && pend->s0 && pend->s1, DFOT_ERR_ARG, "rms_film: pending sum needs a target, a bias and two slices");
    if (c <= 8 * 64 * 2) RMS_CALL(2, true); else RMS_CALL(3, true);
  } else {
    if (c <= 8 * 64 * 2) RMS_CALL(2, false); else RMS_CALL(3, false);
  }
#undef RMS_CALL
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int launch_rms_film(const float* x, const float* w, const bf16* fcache, const float* sv, const uint8_t* cond_mask, bf16* out,
                    long m, int c, int rows_per_bt, int tokens, float eps, hipStream_t s, const RmsPending* pend) {
  return launch_rms_film_t<float>(x, w, fcache, sv, cond_mask, out, m, c, rows_per_bt, tokens, eps, s, pend);
}
// the stream (x, and pend->x) in bf16
int launch_rms_film_bf16(const bf16* x, const float* w, const bf16* fcache, const float* sv, const uint8_t* cond_mask, bf16* out,
                         long m, int c, int rows_per_bt, int tokens, float eps, hipStream_t s, const RmsPending* pend) {
  return launch_rms_film_t<bf16>(x, w, fcache, sv, cond_mask, out, m, c, rows_per_bt, tokens, eps, s, pend);
}

// --------------------------------------------------------------------------------------------
// resampling and skip arithmetic (u_vit_blocks.py:284-314, u_vit3d_pose.py:124-127)
// --------------------------------------------------------------------------------------------
template <typename TIN>
__global__ void pool2_bf16_kernel(const TIN* __restrict__ x, bf16* __restrict__ out, long total4, int h, int w, int c) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const unsigned iu = (unsigned)idx, cq = (unsigned)(c / 4), w2 = (unsigned)(w / 2), h2 = (unsigned)(h / 2);
  const int c4 = (int)(iu % cq);
  const unsigned pix = iu / cq;
  const int ox = (int)(pix % w2), oy = (int)((pix / w2) % h2);
  const long bt = pix / (w2 * h2);
  const TIN* base = x + ((bt * h + 2 * oy) * w + 2 * ox) * (long)c + c4 * 4;
  const float4v a = stream_ld4(base);
  const float4v b = stream_ld4(base + c);
  const float4v cc = stream_ld4(base + (long)w * c);
  const float4v dd = stream_ld4(base + (long)w * c + c);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = f2bf((a[j] + b[j] + cc[j] + dd[j]) * 0.25f);
  *reinterpret_cast<bf16x4*>(out + pix * c + c4 * 4) = o;
}
int launch_pool2_bf16(const float* x, bf16* out, int bt, int h, int w, int c, hipStream_t s) {
  const long total4 = (long)bt * (h / 2) * (w / 2) * (c / 4);
  DFOT_REQUIRE(total4 < (1L << 31), DFOT_ERR_SHAPE, "pool2: %ld work items exceed the 32-bit index range", total4);
  hipLaunchKernelGGL(pool2_bf16_kernel<float>, dim3(cdiv(total4, 256)), dim3(256), 0, s, x, out, total4, h, w, c);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int launch_pool2_bf16_bf16in(const bf16* x, bf16* out, int bt, int h, int w, int c, hipStream_t s) {
  const long total4 = (long)bt * (h / 2) * (w / 2) * (c / 4);
  DFOT_REQUIRE(total4 < (1L << 31), DFOT_ERR_SHAPE, "pool2: %ld work items exceed the 32-bit index range", total4);
  hipLaunchKernelGGL(pool2_bf16_kernel<bf16>, dim3(cdiv(total4, 256)), dim3(256), 0, s, x, out, total4, h, w, c);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

template <typename TIN, typename TB = TIN>
__global__ void sub_bf16_kernel(const TIN* __restrict__ a, const TB* __restrict__ b, bf16* __restrict__ out, long n4,
                                const uint8_t* __restrict__ live, long frame4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n4) return;
  if (live && !live[idx / frame4]) return;
  const float4v x = stream_ld4(a + idx * 4);
  const float4v y = stream_ld4(b + idx * 4);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = f2bf(x[j] - y[j]);
  *reinterpret_cast<bf16x4*>(out + idx * 4) = o;
}
int launch_sub_bf16(const float* a, const float* b, bf16* out, long n, hipStream_t s, const uint8_t* live, long frame_elems) {
  DFOT_REQUIRE(n % 4 == 0 && (!live || (frame_elems > 0 && frame_elems % 4 == 0)), DFOT_ERR_SHAPE, "sub: length %ld must be a multiple of 4", n);
  hipLaunchKernelGGL((sub_bf16_kernel<float, float>), dim3(cdiv(n / 4, 256)), dim3(256), 0, s, a, b, out, n / 4, live, live ? frame_elems / 4 : 1L);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// a from the bf16 stream, b an fp32 skip tensor
int launch_sub_bf16_bf16in(const bf16* a, const float* b, bf16* out, long n, hipStream_t s, const uint8_t* live, long frame_elems) {
  DFOT_REQUIRE(n % 4 == 0 && (!live || (frame_elems > 0 && frame_elems % 4 == 0)), DFOT_ERR_SHAPE, "sub: length %ld must be a multiple of 4", n);
  hipLaunchKernelGGL((sub_bf16_kernel<bf16, float>), dim3(cdiv(n / 4, 256)), dim3(256), 0, s, a, b, out, n / 4, live, live ? frame_elems / 4 : 1L);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// out[bt][y][x][:] = t[bt][y/2][x/2][:] + skip[bt][y][x][:]   (h,w are the LOW-resolution sizes)
template <typename TS>
__global__ void upsample_add_kernel(const float* __restrict__ t, const TS* __restrict__ skip, TS* __restrict__ out,
                                    long total4, int h, int w, int c, const uint8_t* __restrict__ live) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const unsigned iu = (unsigned)idx, cq = (unsigned)(c / 4), w2 = (unsigned)(2 * w), h2 = (unsigned)(2 * h);
  const int c4 = (int)(iu % cq);
  const unsigned pix = iu / cq;
  const int x = (int)(pix % w2), y = (int)((pix / w2) % h2);
  const long bt = pix / (w2 * h2);
  if (live && !live[bt]) return;
  const float4v a = *reinterpret_cast<const float4v*>(t + ((bt * h + y / 2) * w + x / 2) * (long)c + c4 * 4);
  const float4v b = stream_ld4(skip + pix * c + c4 * 4);
  stream_st4(out + pix * c + c4 * 4, a + b);
}
int launch_upsample_add(const float* t, const float* skip, float* out, int bt, int h, int w, int c, hipStream_t s, const uint8_t* live) {
  const long total4 = (long)bt * 4 * h * w * (c / 4);
  DFOT_REQUIRE(total4 < (1L << 31), DFOT_ERR_SHAPE, "upsample_add: %ld work items exceed the 32-bit index range", total4);
  hipLaunchKernelGGL(upsample_add_kernel<float>, dim3(cdiv(total4, 256)), dim3(256), 0, s, t, skip, out, total4, h, w, c, live);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// skip and out in the bf16 stream (t stays the fp32 output of the Upsample convolution)
int launch_upsample_add_bf16(const float* t, const bf16* skip, bf16* out, int bt, int h, int w, int c, hipStream_t s, const uint8_t* live) {
  const long total4 = (long)bt * 4 * h * w * (c / 4);
  DFOT_REQUIRE(total4 < (1L << 31), DFOT_ERR_SHAPE, "upsample_add: %ld work items exceed the 32-bit index range", total4);
  hipLaunchKernelGGL(upsample_add_kernel<bf16>, dim3(cdiv(total4, 256)), dim3(256), 0, s, t, skip, out, total4, h, w, c, live);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// --------------------------------------------------------------------------------------------
// camera-ray encoding (geometry_utils.py:49-81,102-133,244-295): raw [B][T][16] -> [B][T][180][res][res]
// thread = one output element; the per-frame 3x3 algebra is recomputed per thread (27 FMAs)
// --------------------------------------------------------------------------------------------
// thread = four consecutive x of one (frame, channel) row: one 16-byte store; the origin channels (the first 90) do not depend on
// the pixel, so their sine is evaluated once per thread instead of once per element (accurate sinf with its range reduction is the
// cost of this kernel: 5 ms for the 64 frames of a training batch in the one-element-per-thread form)
__global__ void ray_encode_kernel(const float* __restrict__ poses, float* __restrict__ out, long total4, int t, int res,
                                  int normalized) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int xq = res / 4;
  const int x0 = (int)(idx % xq) * 4, y = (int)((idx / xq) % res);
  const int ch = (int)((idx / ((long)xq * res)) % 180);
  const long bt = idx / ((long)xq * res * 180);
  const long b = bt / t;
  const float* p = poses + bt * 16;
  const float* p0 = poses + b * t * 16;
  // R' = R R0^T ; T' = T - R' T0   (normalized: the caller already expressed the poses in its world frame)
  float r[3][3], tr[3];
  if (normalized) {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) r[i][j] = p[4 + i * 4 + j];
      tr[i] = p[4 + i * 4 + 3];
    }
  } else {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) {
        float acc = 0.f;
        for (int m = 0; m < 3; ++m) acc = fmaf(p[4 + i * 4 + m], p0[4 + j * 4 + m], acc);
        r[i][j] = acc;
      }
    }
    for (int i = 0; i < 3; ++i) {
      float acc = 0.f;
      for (int j = 0; j < 3; ++j) acc = fmaf(r[i][j], p0[4 + j * 4 + 3], acc);
      tr[i] = p[4 + i * 4 + 3] - acc;
    }
  }
  const int which = ch / 90;   // 0 origin, 1 direction
  const int rr = ch % 90;
  const int phase = rr / 45;   // 0: sin(arg), 1: sin(arg + pi/2)
  const int comp = (rr % 45) / 15, f = rr % 15;
  const float scale = 3.14159265358979323846f * (float)(1 << f);
  auto enc = [&](float val) {
    float arg = __fmul_rn(val, scale);
    if (phase) arg = __fadd_rn(arg, 0.5f * 3.14159265358979323846f);
    return sinf(arg);
  };
  float4v o;
  if (which == 0) {
    float acc = 0.f;
    for (int j = 0; j < 3; ++j) acc = fmaf(r[j][comp], tr[j], acc);  // (R'^T T')[comp]
    const float v = enc(-acc);
    o = float4v{v, v, v, v};
  } else {
    const float fres = (float)res;
    const float cy = __fdiv_rn(__fsub_rn((float)y + 0.5f, __fmul_rn(p[3], fres)), __fmul_rn(p[1], fres));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float cx = __fdiv_rn(__fsub_rn((float)(x0 + j) + 0.5f, __fmul_rn(p[2], fres)), __fmul_rn(p[0], fres));
      o[j] = enc(fmaf(r[2][comp], 1.f, fmaf(r[1][comp], cy, __fmul_rn(r[0][comp], cx))));
    }
  }
  *reinterpret_cast<float4v*>(out + idx * 4) = o;
}

int launch_ray_encode(const float* poses, float* out, int b, int t, int res, int normalized, hipStream_t s) {
  DFOT_REQUIRE(res % 4 == 0, DFOT_ERR_SHAPE, "ray_encode: resolution %d must be a multiple of 4", res);
  const long total4 = (long)b * t * 180 * res * (res / 4);
  hipLaunchKernelGGL(ray_encode_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, s, poses, out, total4, t, res, normalized);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// --------------------------------------------------------------------------------------------
// sampler step
// --------------------------------------------------------------------------------------------
__global__ void hg_prepare_kernel(const float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ qa,
                                  const float* __restrict__ qb, float* __restrict__ x_in, long total4, int nfe, int tokens,
                                  long f4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const unsigned iu = (unsigned)idx, fu = (unsigned)f4;  // 32-bit index arithmetic (launcher: total < 2^31)
  const long e = iu % fu;
  const long bt = iu / fu;  // over (b*nfe + h, t)
  const int tk = (int)((unsigned)bt % (unsigned)tokens);
  const long bh = (unsigned)bt / (unsigned)tokens;
  const long b = bh / nfe;
  const float a = qa[bt], c = qb[bt];
  const float4v xv = *reinterpret_cast<const float4v*>(x + ((b * tokens + tk) * f4 + e) * 4);
  float4v o = xv * a;
  if (c != 0.f) o += *reinterpret_cast<const float4v*>(noise + idx * 4) * c;
  *reinterpret_cast<float4v*>(x_in + idx * 4) = o;
}
int launch_hg_prepare(const float* x, const float* noise, const float* qa, const float* qb, float* x_in, int batch,
                      int nfe, int tokens, long f, hipStream_t s) {
  DFOT_REQUIRE(f % 4 == 0, DFOT_ERR_SHAPE, "hg_prepare: frame elements %ld must be a multiple of 4", f);
  const long total4 = (long)batch * nfe * tokens * (f / 4);
  DFOT_REQUIRE(total4 < (1L << 31), DFOT_ERR_SHAPE, "hg_prepare: %ld work items exceed the 32-bit index range", total4);
  hipLaunchKernelGGL(hg_prepare_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, s, x, noise, qa, qb, x_in, total4, nfe,
                     tokens, f / 4);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

__global__ void ddim_compose_kernel(const float* __restrict__ x, const float* __restrict__ x_in, const float* __restrict__ v,
                                    const float* __restrict__ sa, const float* __restrict__ s1, const float* __restrict__ an,
                                    const float* __restrict__ cn, const float* __restrict__ keep,
                                    const float* __restrict__ weight, const uint8_t* __restrict__ gen,
                                    float* __restrict__ x_next, long total4, int nfe, int tokens, long f4, int wstride) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const unsigned iu = (unsigned)idx, fu = (unsigned)f4;
  const long e = iu % fu;
  const long bt = iu / fu;  // over (b, t)
  const int tk = (int)((unsigned)bt % (unsigned)tokens);
  const long b = (unsigned)bt / (unsigned)tokens;
  float4v o;
  if (!gen[bt]) {
    o = *reinterpret_cast<const float4v*>(x + idx * 4);
  } else {
    o = float4v{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < nfe; ++h) {
      const long row = (b * nfe + h) * tokens + tk;
      const long off = (row * f4 + e) * 4;
      const float4v xi = *reinterpret_cast<const float4v*>(x_in + off);
      float4v xp;
      if (keep[row] != 0.f) {
        xp = xi;
      } else {
        const float4v vv = *reinterpret_cast<const float4v*>(v + off);
        const float4v x0 = xi * sa[row] - vv * s1[row];
        const float4v ep = vv * sa[row] + xi * s1[row];
        xp = x0 * an[row] + ep * cn[row];
      }
      o += xp * weight[wstride ? h * wstride + tk : h];  // wstride = tokens: per-(branch, token) weights (gen segments)
    }
  }
  *reinterpret_cast<float4v*>(x_next + idx * 4) = o;
}
int launch_ddim_compose(const float* x, const float* x_in, const float* v, const float* sa, const float* s1,
                        const float* an, const float* cn, const float* keep, const float* weight, const uint8_t* gen,
                        float* x_next, int batch, int nfe, int tokens, long f, bool weight_per_token, hipStream_t s) {
  DFOT_REQUIRE(f % 4 == 0, DFOT_ERR_SHAPE, "ddim_compose: frame elements %ld must be a multiple of 4", f);
  const long total4 = (long)batch * tokens * (f / 4);
  DFOT_REQUIRE(total4 < (1L << 31), DFOT_ERR_SHAPE, "ddim_compose: %ld work items exceed the 32-bit index range", total4);
  hipLaunchKernelGGL(ddim_compose_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, s, x, x_in, v, sa, s1, an, cn, keep,
                     weight, gen, x_next, total4, nfe, tokens, f / 4, weight_per_token ? tokens : 0);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// stochastic part of a sampling step (ddim_sample_step with eta > 0, discrete_diffusion.py:515-538; ddpm_sample_step, :441-449):
// every branch adds sigma * noise to its prediction BEFORE composition, and composition is linear, so
//   x_next[b,t] += sum_h w_h * sigma[b*nfe+h, t] * noise[b*nfe+h, t]   on generated tokens
// (sigma = 0 rows -- kept tokens, level 0, next level < 0 -- are set by the host).  noise is already clamped.
__global__ void ddim_noise_kernel(const float* __restrict__ noise, const float* __restrict__ sigma, const float* __restrict__ weight,
                                  const uint8_t* __restrict__ gen, float* __restrict__ x_next, long total4, int nfe, int tokens,
                                  long f4, int wstride) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const unsigned iu = (unsigned)idx, fu = (unsigned)f4;
  const long e = iu % fu;
  const long bt = iu / fu;
  if (!gen[bt]) return;
  const int tk = (int)((unsigned)bt % (unsigned)tokens);
  const long b = (unsigned)bt / (unsigned)tokens;
  float4v o = *reinterpret_cast<const float4v*>(x_next + idx * 4);
  for (int h = 0; h < nfe; ++h) {
    const long row = (b * nfe + h) * tokens + tk;
    const float c = sigma[row] * weight[wstride ? h * wstride + tk : h];
    if (c != 0.f) o += *reinterpret_cast<const float4v*>(noise + (row * f4 + e) * 4) * c;
  }
  *reinterpret_cast<float4v*>(x_next + idx * 4) = o;
}
int launch_ddim_noise(const float* noise, const float* sigma, const float* weight, const uint8_t* gen, float* x_next, int batch, int nfe,
                      int tokens, long f, bool weight_per_token, hipStream_t s) {
  DFOT_REQUIRE(f % 4 == 0, DFOT_ERR_SHAPE, "ddim_noise: frame elements %ld must be a multiple of 4", f);
  const long total4 = (long)batch * tokens * (f / 4);
  DFOT_REQUIRE(total4 < (1L << 31), DFOT_ERR_SHAPE, "ddim_noise: %ld work items exceed the 32-bit index range", total4);
  hipLaunchKernelGGL(ddim_noise_kernel, dim3(cdiv(total4, 256)), dim3(256), 0, s, noise, sigma, weight, gen, x_next, total4, nfe, tokens,
                     f / 4, weight_per_token ? tokens : 0);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// --------------------------------------------------------------------------------------------
// continuous-time v-prediction loss of one noised forward (ContinuousDiffusion.forward, continuous_diffusion.py:140-167;
// used by training_step and by the validation denoising loss): x_t = a x + s eps ; eps_hat = a v + s x_t ;
// loss = (eps_hat - eps)^2 * w ; x_pred = a x_t - s v.  Deterministic two-stage mean per (video, token).
// --------------------------------------------------------------------------------------------
// VSPACE (DiscreteDiffusion.forward with objective pred_v, discrete_diffusion.py:345-377): loss = (v - (a eps - s x))^2 * w.
constexpr int VL_CHUNK = 4096;
template <bool VSPACE>
__global__ __launch_bounds__(256) void vloss_partial_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                            const float* __restrict__ v, const float* __restrict__ a,
                                                            const float* __restrict__ sg, const float* __restrict__ w,
                                                            float* __restrict__ x_pred, float* __restrict__ partial, long f) {
  __shared__ float red[4];
  const int bt = blockIdx.y, chunk = blockIdx.x;
  const float av = a[bt], sv = sg[bt], wv = w[bt];
  const long base = (long)bt * f;
  const long e0 = (long)chunk * VL_CHUNK;
  float acc = 0.f;
  for (long e = e0 + threadIdx.x * 4; e < min(e0 + VL_CHUNK, f); e += 1024) {
    const float4v xv = *reinterpret_cast<const float4v*>(x + base + e);
    const float4v nv = *reinterpret_cast<const float4v*>(noise + base + e);
    const float4v vv = *reinterpret_cast<const float4v*>(v + base + e);
    const float4v xt = xv * av + nv * sv;
    const float4v d = VSPACE ? vv - (nv * av - xv * sv) : (vv * av + xt * sv) - nv;
    acc += (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * wv;
    if (x_pred) *reinterpret_cast<float4v*>(x_pred + base + e) = xt * av - vv * sv;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long)bt * gridDim.x + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void vloss_finalize_kernel(const float* __restrict__ partial, float* __restrict__ loss, int chunks, float inv_f) {
  const int bt = blockIdx.x * blockDim.x + threadIdx.x;
  if (bt >= (int)gridDim.x * (int)blockDim.x) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[(long)bt * chunks + c];
  loss[bt] = s * inv_f;
}
int vloss_chunks(long f) { return cdiv(f, VL_CHUNK); }
int launch_vloss(const float* x, const float* noise, const float* v, const float* a, const float* sg, const float* w,
                 float* x_pred, float* partial, float* loss, int bt, long f, bool vspace, hipStream_t s) {
  DFOT_REQUIRE(f % 4 == 0, DFOT_ERR_SHAPE, "vloss: frame elements %ld must be a multiple of 4", f);
  const int chunks = vloss_chunks(f);
  if (vspace)
    hipLaunchKernelGGL(vloss_partial_kernel<true>, dim3(chunks, bt), dim3(256), 0, s, x, noise, v, a, sg, w, x_pred, partial, f);
  else
    hipLaunchKernelGGL(vloss_partial_kernel<false>, dim3(chunks, bt), dim3(256), 0, s, x, noise, v, a, sg, w, x_pred, partial, f);
  DFOT_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(vloss_finalize_kernel, dim3(bt), dim3(1), 0, s, partial, loss, chunks, 1.0f / (float)f);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// --------------------------------------------------------------------------------------------
// casts, weight packing, layout taps
// --------------------------------------------------------------------------------------------
// casts: 8 elements per thread (two 16-byte loads, one 16-byte store) when the pointers are 16-byte aligned; scalar tail / fallback
__global__ void f32_to_bf16_kernel(const float* __restrict__ s, bf16* __restrict__ d, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = f2bf(s[i]);
}
__global__ void f32_to_bf16_kernel8(const float* __restrict__ s, bf16* __restrict__ d, long n8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4v a = reinterpret_cast<const float4v*>(s)[2 * i], b = reinterpret_cast<const float4v*>(s)[2 * i + 1];
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = f2bf(a[j]), o[4 + j] = f2bf(b[j]);
  reinterpret_cast<bf16x8*>(d)[i] = o;
}
__global__ void bf16_to_f32_kernel(const bf16* __restrict__ s, float* __restrict__ d, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = bf2f(s[i]);
}
__global__ void bf16_to_f32_kernel8(const bf16* __restrict__ s, float* __restrict__ d, long n8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const bf16x8 v = reinterpret_cast<const bf16x8*>(s)[i];
  float4v a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = bf2f(v[j]), b[j] = bf2f(v[4 + j]);
  reinterpret_cast<float4v*>(d)[2 * i] = a;
  reinterpret_cast<float4v*>(d)[2 * i + 1] = b;
}
static bool aligned16(const void* a, const void* b) { return (((uintptr_t)a | (uintptr_t)b) & 15) == 0; }
int launch_f32_to_bf16(const float* src, bf16* dst, long n, hipStream_t s) {
  const long n8 = aligned16(src, dst) ? n / 8 : 0;
  if (n8) hipLaunchKernelGGL(f32_to_bf16_kernel8, dim3(cdiv(n8, 256)), dim3(256), 0, s, src, dst, n8);
  if (n - 8 * n8) hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(cdiv(n - 8 * n8, 256)), dim3(256), 0, s, src + 8 * n8, dst + 8 * n8, n - 8 * n8);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int launch_bf16_to_f32(const bf16* src, float* dst, long n, hipStream_t s) {
  const long n8 = aligned16(src, dst) ? n / 8 : 0;
  if (n8) hipLaunchKernelGGL(bf16_to_f32_kernel8, dim3(cdiv(n8, 256)), dim3(256), 0, s, src, dst, n8);
  if (n - 8 * n8) hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(cdiv(n - 8 * n8, 256)), dim3(256), 0, s, src + 8 * n8, dst + 8 * n8, n - 8 * n8);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

__global__ void pack_rows_kernel(const float* __restrict__ src, bf16* __restrict__ dst, const int* __restrict__ map, long total,
                                 int k, int kpad, long ldd, int col0) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int kk = (int)(i % kpad);
  const long r = i / kpad;
  const long sr = map ? map[r] : r;
  dst[r * ldd + col0 + kk] = kk < k ? f2bf(src[sr * k + kk]) : f2bf(0.f);
}
int launch_pack_rows(const float* src, bf16* dst, const int* map, int rows, int k, int kpad, long ldd, int col0,
                     hipStream_t s) {
  const long total = (long)rows * kpad;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, src, dst, map, total, k, kpad, ldd, col0);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

__global__ void pack_conv3_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long total, int ci) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % ci);
  const int tap = (int)((i / ci) % 9);
  const long co = i / ((long)ci * 9);
  dst[i] = f2bf(src[(co * ci + c) * 9 + tap]);
}
int launch_pack_conv3(const float* src, bf16* dst, int co, int ci, hipStream_t s) {
  const long total = (long)co * ci * 9;
  hipLaunchKernelGGL(pack_conv3_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, src, dst, total, ci);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

__global__ void gather_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ map, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[map[i]];
}
int launch_gather_f32(const float* src, float* dst, const int* map, int n, hipStream_t s) {
  hipLaunchKernelGGL(gather_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, dst, map, n);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, long total, int p, int c) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pp = (int)(i % p);
  const int cc = (int)((i / p) % c);
  const long bt = i / ((long)p * c);
  dst[i] = (float)src[(bt * p + pp) * c + cc];
}
int launch_nhwc_to_nchw(const float* src, float* dst, int bt, int p, int c, hipStream_t s) {
  const long total = (long)bt * p * c;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, s, src, dst, total, p, c);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int launch_bf16_nhwc_to_nchw(const bf16* src, float* dst, int bt, int p, int c, hipStream_t s) {
  const long total = (long)bt * p * c;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(cdiv(total, 256)), dim3(256), 0, s, src, dst, total, p, c);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

}  // namespace dfot

// --------------------------------------------------------------------------------------------
// training: loss gradient, AdamW, gradient norm
// --------------------------------------------------------------------------------------------
namespace dfot {
// dv = coef[bt] * d(loss term)/dv of vloss_partial_kernel: VSPACE: v - (a eps - s x) ; else a * ((a v + s x_t) - eps)
template <bool VSPACE>
__global__ void vloss_grad_kernel(const float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ v,
                                  const float* __restrict__ a, const float* __restrict__ sg, const float* __restrict__ coef,
                                  float* __restrict__ dv, long f4, long total4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const long bt = i / f4;
  const float av = a[bt], sv = sg[bt], cv = coef[bt];
  const float4v xv = *reinterpret_cast<const float4v*>(x + i * 4);
  const float4v nv = *reinterpret_cast<const float4v*>(noise + i * 4);
  const float4v vv = *reinterpret_cast<const float4v*>(v + i * 4);
  float4v d;
  if (VSPACE) d = (vv - (nv * av - xv * sv)) * cv;
  else d = ((vv * av + (xv * av + nv * sv) * sv) - nv) * (cv * av);
  *reinterpret_cast<float4v*>(dv + i * 4) = d;
}
int launch_vloss_grad(const float* x, const float* noise, const float* v, const float* a, const float* sg, const float* coef, float* dv,
                      int bt, long f, bool vspace, hipStream_t s) {
  DFOT_REQUIRE(f % 4 == 0, DFOT_ERR_SHAPE, "vloss_grad: frame elements %ld must be a multiple of 4", f);
  const long total4 = (long)bt * (f / 4);
  if (vspace) hipLaunchKernelGGL(vloss_grad_kernel<true>, dim3(cdiv(total4, 256)), dim3(256), 0, s, x, noise, v, a, sg, coef, dv, f / 4, total4);
  else hipLaunchKernelGGL(vloss_grad_kernel<false>, dim3(cdiv(total4, 256)), dim3(256), 0, s, x, noise, v, a, sg, coef, dv, f / 4, total4);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// out[0] = sum x^2, bit-reproducible for a given input (data-parallel replicas must compute the SAME clip coefficient from the
// same all-reduced gradients, or they drift apart): fixed grid, per-workgroup partials, one workgroup adds them in a fixed order
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)SUMSQ_BLOCKS * 1024) {
    if (i + 3 < n) {
      const float4v v = *reinterpret_cast<const float4v*>(x + i);
      acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    } else {
      for (long j = i; j < n; ++j) acc += x[j] * x[j];
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < SUMSQ_BLOCKS; i += 256) acc += part[i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
int launch_sumsq(const float* x, long n, float* out, hipStream_t s) {
  DFOT_REQUIRE(((uintptr_t)x & 15) == 0, DFOT_ERR_ARG, "sumsq: buffer must be 16-byte aligned");
  static float* part = nullptr;  // one per process (one stream of optimizer steps)
  if (!part) DFOT_CHECK_HIP(hipMalloc(&part, SUMSQ_BLOCKS * sizeof(float)));
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, x, n, part);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, part, out);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// torch.optim.AdamW (decoupled weight decay) on flat fp32 buffers.  The gradient is first multiplied by
// min(1, max_norm / (sqrt(*sumsq) + 1e-6)) when sumsq is given (clip_grad_norm_ without a host round trip).
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             float lr, float b1, float b2, float eps, float wd, float c1, float c2, const float* __restrict__ sumsq,
                             float max_norm, float* __restrict__ ema, float ema_decay) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float clip = 1.f;
  if (sumsq) clip = fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));
  const float gv = g[i] * clip;
  const float mv = b1 * m[i] + (1.f - b1) * gv;
  const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
  m[i] = mv;
  v[i] = vv;
  const float pv = p[i] * (1.f - lr * wd);
  const float pn = pv - (lr / c1) * mv / (sqrtf(vv) / sqrtf(c2) + eps);
  p[i] = pn;
  if (ema) ema[i] = ema_decay * ema[i] + (1.f - ema_decay) * pn;  // EMAModel.step (algorithms/common/ema.py:22-32), same pass
}
int launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                 const float* sumsq, float max_norm, float* ema, float ema_decay, hipStream_t s) {
  DFOT_REQUIRE(step >= 1, DFOT_ERR_ARG, "adamw: step counts from 1");
  const float c1 = 1.f - powf(b1, (float)step), c2 = 1.f - powf(b2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, c1, c2, sumsq, max_norm, ema, ema_decay);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
}  // namespace dfot
