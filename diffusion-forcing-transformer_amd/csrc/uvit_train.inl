// Backward kernels and op entry points for the UViT3DPose backbone (training path of BASELINE config 5); the training step itself is
// sequenced by diffusion-forcing-transformer_amd/uvit_train.py over these entry points.  Included at the end of dit.hip (shares the training helpers of dit_train.inl).
//
// conv3x3 (padding 1, channels-last activations [BT][H][W][C] bf16), y = conv(x, W) + b with W [Co][Ci][3][3]:
//   dx = conv(dy, W')         W'[ci][2-ky][2-kx][co] = W[co][ci][ky][kx]      -> the forward's implicit-GEMM kernel on repacked weights
//   dW[co][ci][ky][kx] = sum_pix dy[pix][co] x[pix + (ky-1, kx-1)][ci]        -> 9 GEMMs over the pixel axis: dy^T [Co][pix] against
//                                                                                 the tap-shifted x^T [Ci][pix] (zero outside the image),
//                                                                                 K split over workgroups into partial buffers
//   db[co] = sum_pix dy[pix][co]
// Replaces torch autograd through F.conv2d in ResBlock / Downsample / Upsample (algorithms/dfot/backbones/u_vit/u_vit_blocks.py:16-93).
namespace dfot {
namespace {

// W [Co][Ci][3][3] fp32 -> W' [Ci][(2-ky)*3 + (2-kx)][Co] bf16: the data-gradient convolution's weights in the forward kernel's layout
__global__ void pack_conv3_dgrad_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int co, int ci) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)co * ci * 9) return;
  const int o = (int)(i % co);
  const int tap = (int)((i / co) % 9);
  const long c = i / ((long)co * 9);
  const int ky = 2 - tap / 3, kx = 2 - tap % 3;
  dst[i] = f2bf(src[((long)o * ci + c) * 9 + ky * 3 + kx]);
}

// dst[c][pix] = x[pix + (dy, dx)][c] inside the image, else 0   (x [BT][H][W][C] bf16; 64 pixels x 64 channels per workgroup)
__global__ __launch_bounds__(256) void transpose_shift_kernel(const bf16* __restrict__ x, bf16* __restrict__ dst, long pix, int H, int W, int C,
                                                              int dy, int dx) {
  __shared__ bf16 tile[64][66];
  const long p0 = (long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long p = p0 + ty * 16 + i;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const int sy = yy + dy, sx = xx + dx;
    const bool ok = sy >= 0 && sy < H && sx >= 0 && sx < W;
    tile[ty * 16 + i][tx] = ok ? x[(p + (long)dy * W + dx) * C + c0 + tx] : f2bf(0.f);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) dst[(long)(c0 + ty * 16 + i) * pix + p0 + tx] = tile[tx][ty * 16 + i];
}

// tmp [9][Co][Ci] fp32 -> dW [Co][Ci][3][3]
__global__ void conv_wgrad_repack_kernel(const float* __restrict__ tmp, float* __restrict__ dw, int co, int ci) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)co * ci * 9) return;
  const int tap = (int)(i % 9);
  const long oc = i / 9;  // co * ci + c
  dw[i] = tmp[(long)tap * co * ci + oc];
}

struct ConvBwdScratch {  // sized by the caller for the largest convolution it differentiates
  bf16* dyT = nullptr;   // [Co][pix]
  bf16* xT = nullptr;    // [Ci][pix]
  float* taps = nullptr; // [9][Co][Ci]
  float* ws = nullptr;   // split-K partial tiles
  size_t ws_floats = 0;
  const bf16* zeros = nullptr;
};

// dx (fp32 [pix][Ci], optional), dW (fp32 [Co][Ci][3][3]), db (fp32 [Co], optional; must be zeroed by the caller)
int conv3_backward(const bf16* x, const bf16* dy, const bf16* w_dgrad, float* dx, float* dw, float* db, int bt, int H, int W, int ci, int co,
                   const ConvBwdScratch& sc, hipStream_t s, bf16* dx_bf = nullptr) {
  const long pix = (long)bt * H * W;
  DFOT_REQUIRE(ci % 64 == 0 && co % 64 == 0 && pix % 64 == 0, DFOT_ERR_SHAPE, "conv3_backward: channels %d -> %d and %ld pixels must be multiples of 64", ci, co, pix);
  int rc = 0;
  if (dx || dx_bf) {  // dx_bf: the gradient only feeds a GroupNorm backward -- bf16 halves its bytes there and here
    GemmArgs g;
    g.zeros = sc.zeros;
    g.A = dy; g.W = w_dgrad; g.M = (int)pix; g.N = ci; g.K = 9 * co; g.H = H; g.Wd = W; g.Cin = co; g.ldo = ci;
    if (dx_bf) g.out_bf16 = dx_bf; else g.out_f32 = dx;
    if ((rc = launch_gemm(A_CONV3, dx_bf ? E_BF16 : E_F32, GEMM_AUTO, g, s))) return rc;
  }
  if (db) {
    if ((rc = launch_colsum_bf16(dy, db, pix, co, (long)co, s))) return rc;
  }
  // weight gradient: one token-axis GEMM per tap, both operands read in place (wgrad.hip conv mode: x rows shifted by the tap, zero outside
  // the image); few output tiles, K = pixels: split over workgroups into partial buffers
  // all nine taps in one launch (grid.y = tap): 9x the workgroups in flight, partial outputs [slice][tap][Co][Ci], one reduce pass
  int target = 0;
  const long tiles = 9L * wgrad_conv_tiles(co, ci, &target);
  int split = (int)(target / tiles);
  split = split < 1 ? 1 : (split > 128 ? 128 : split);
  while (split > 1 && (pix / 64 < 4L * split || (size_t)split * 9 * co * ci > sc.ws_floats)) --split;
  if (split == 1) {
    if ((rc = launch_wgrad_conv_taps(dy, x, sc.taps, co, ci, pix, 1, H, W, s))) return rc;
  } else {
    if ((rc = launch_wgrad_conv_taps(dy, x, sc.ws, co, ci, pix, split, H, W, s))) return rc;
    hipLaunchKernelGGL(slices_sum_kernel, dim3(cdiv(9L * co * ci / 4, 256)), dim3(256), 0, s, sc.ws, sc.taps, 9L * co * ci / 4, split, 9L * co * ci);
    DFOT_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(conv_wgrad_repack_kernel, dim3(cdiv((long)co * ci * 9, 256)), dim3(256), 0, s, sc.taps, dw, co, ci);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

}  // namespace
}  // namespace dfot

namespace dfot {
namespace {
// Grow-only device scratch of the op entry points (one buffer per slot, process lifetime): the ops run on one stream in program order, so a
// slot is free again when the next op that uses it is enqueued.  Replaces per-call hipMalloc / synchronize / hipFree.
int op_scratch(int slot, size_t bytes, void** out) {
  static void* buf[8] = {nullptr};
  static size_t cap[8] = {0};
  if (cap[slot] < bytes) {
    if (buf[slot]) {
      DFOT_CHECK_HIP(hipDeviceSynchronize());
      (void)hipFree(buf[slot]);
      buf[slot] = nullptr;
      cap[slot] = 0;
    }
    const size_t want = bytes + bytes / 4;
    DFOT_CHECK_HIP(hipMalloc(&buf[slot], want));
    DFOT_CHECK_HIP(hipMemset(buf[slot], 0, want));
    cap[slot] = want;
  }
  *out = buf[slot];
  return DFOT_OK;
}
}  // namespace
}  // namespace dfot

extern "C" {
using namespace dfot;

// test entry: x, dy bf16 channels-last; w fp32 [Co][Ci][3][3]; dx fp32 [pix][Ci]; dw fp32 [Co][Ci][3][3]; db fp32 [Co]
int dfot_op_conv3x3_bwd2(const void* x, const void* dy, const float* w, float* dx, void* dx_bf, float* dw, float* db, int bt, int hh, int ww, int cin,
                         int cout, void* stream) {
  DFOT_REQUIRE(x && dy && w && (dx != nullptr) != (dx_bf != nullptr) && dw && db, DFOT_ERR_ARG, "op_conv3x3_bwd2: null argument, or both / neither of dx and dx_bf");
  hipStream_t s = (hipStream_t)stream;
  ConvBwdScratch sc;
  void *taps = nullptr, *ws = nullptr, *wd = nullptr, *zeros = nullptr;
  // partial weight gradients of the K slices: up to 1024 / 9 slices of the nine taps for the 128-channel convolutions (9 output tiles)
  sc.ws_floats = (size_t)256 * cout * cin;
  if ((size_t)1024 * cout * cin <= ((size_t)32 << 20)) sc.ws_floats = (size_t)1024 * cout * cin;
  int rc = 0;
  if ((rc = op_scratch(0, (size_t)9 * cout * cin * sizeof(float), &taps)) || (rc = op_scratch(1, sc.ws_floats * sizeof(float), &ws)) ||
      (rc = op_scratch(2, (size_t)9 * cout * cin * sizeof(bf16), &wd)) || (rc = op_scratch(3, 256, &zeros)))
    return rc;
  sc.taps = (float*)taps; sc.ws = (float*)ws; sc.zeros = (const bf16*)zeros;  // slot 3 is only ever zero
  DFOT_CHECK_HIP(hipMemsetAsync(db, 0, (size_t)cout * sizeof(float), s));
  hipLaunchKernelGGL(pack_conv3_dgrad_kernel, dim3(cdiv((long)cout * cin * 9, 256)), dim3(256), 0, s, w, (bf16*)wd, cout, cin);
  DFOT_CHECK_HIP(hipGetLastError());
  return conv3_backward((const bf16*)x, (const bf16*)dy, (const bf16*)wd, dx_bf ? nullptr : dx, dw, db, bt, hh, ww, cin, cout, sc, s, (bf16*)dx_bf);
}
int dfot_op_conv3x3_bwd(const void* x, const void* dy, const float* w, float* dx, float* dw, float* db, int bt, int hh, int ww, int cin, int cout,
                        void* stream) {
  DFOT_REQUIRE(dx, DFOT_ERR_ARG, "op_conv3x3_bwd: null argument");
  return dfot_op_conv3x3_bwd2(x, dy, w, dx, nullptr, dw, db, bt, hh, ww, cin, cout, stream);
}

}  // extern "C"

// ---- GroupNorm(32) [+ FiLM] + SiLU backward (ResBlock in_layers / out_norm, u_vit_blocks.py:57-93) -------------------------------
// forward: xhat = (x - mean_g) rstd_g per (image, group) ; g = xhat gamma + beta ; z = FILM ? g (1 + scale) + shift : g ; y = SiLU(z)
// backward for dy: dz = dy SiLU'(z) ; FILM: dscale = dz g, dshift = dz, dg = dz (1 + scale) ; dgamma = sum dg xhat, dbeta = sum dg ;
//   dxhat = dg gamma ; dx = rstd (dxhat - mean_grp(dxhat) - xhat mean_grp(dxhat xhat))
// x, dy, dx fp32 channels-last [BT][P][C]; film / dfilm bf16 [BT*P][2C] (scale | shift); stats [BT][32][2] = (mean, rstd).
namespace dfot {
namespace {

__device__ __forceinline__ float silu_grad(float z) {
  const float sg = 1.0f / (1.0f + __expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}

// pass 1: per-workgroup partial sums of (sum dxhat, sum dxhat xhat) per group and of dgamma / dbeta per channel.  One workgroup =
// one (image, pixel chunk); C / 4 lanes cover a pixel row with 16-byte loads and the 256 / (C / 4) lane groups take alternate
// pixels; the groups are summed through LDS before the atomics.  C a multiple of 128 and at most 1024 (launcher)
// dy as fp32 or bf16 (the data gradient of a convolution that feeds nothing else)
template <typename TDY>
__device__ __forceinline__ __attribute__((ext_vector_type(4))) float gn_load_dy4(const TDY* p) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  if constexpr (sizeof(TDY) == 4) {
    return *reinterpret_cast<const f4*>(p);
  } else {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
  }
}

template <bool FILM, typename TDY>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ x, const TDY* __restrict__ dy, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const bf16* __restrict__ film, float* __restrict__ part, int P, int C, int chunk,
                                                            long ldfilm) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  __shared__ f4 red[4][256];  // [quantity][thread]
  const int bt = blockIdx.x, p0 = blockIdx.y * chunk;
  const int cq = C / 4, cpg = C / 32;
  const int passes = (cq + 255) / 256;            // > 1 only for C > 1024 (not used)
  const int lanes = cq < 256 ? cq : 256, groups = 256 / lanes;
  const int lane = threadIdx.x % lanes, rg = threadIdx.x / lanes;
  (void)passes;
  const int c = lane * 4, grp = c / cpg;
  const float mean = stats[((long)bt * 32 + grp) * 2], rstd = stats[((long)bt * 32 + grp) * 2 + 1];
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  f4 dgm = {0.f, 0.f, 0.f, 0.f}, dbt = dgm, s1 = dgm, s2 = dgm;
  const int p1 = p0 + chunk < P ? p0 + chunk : P;
  if (rg < groups) {
#pragma unroll 4
    for (int p = p0 + rg; p < p1; p += groups) {
      const long e = ((long)bt * P + p) * C + c;
      const f4 xv = *reinterpret_cast<const f4*>(x + e), dv = gn_load_dy4(dy + e);
      bf16x4 fs, fh;
      if (FILM) {
        const long f = ((long)bt * P + p) * ldfilm + c;
        fs = *reinterpret_cast<const bf16x4*>(film + f);
        fh = *reinterpret_cast<const bf16x4*>(film + f + C);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (xv[j] - mean) * rstd;
        const float gv = xh * ga[j] + be[j];
        float z = gv, mul = 1.f;
        if (FILM) {
          mul = 1.0f + bf2f(fs[j]);
          z = gv * mul + bf2f(fh[j]);
        }
        const float dg = dv[j] * silu_grad(z) * mul;
        dgm[j] += dg * xh;
        dbt[j] += dg;
        s1[j] += dg * ga[j];
        s2[j] += dg * ga[j] * xh;
      }
    }
  }
  red[0][threadIdx.x] = dgm; red[1][threadIdx.x] = dbt; red[2][threadIdx.x] = s1; red[3][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < lanes) {
    for (int g = 1; g < groups; ++g) {
      dgm += red[0][g * lanes + lane]; dbt += red[1][g * lanes + lane]; s1 += red[2][g * lanes + lane]; s2 += red[3][g * lanes + lane];
    }
    // the workgroup's partial row (no atomics: det_sum adds the rows in a fixed order): dgamma [C] | dbeta [C] | (sum dxhat, sum dxhat xhat) [32][2]
    float* prow = part + ((long)bt * gridDim.y + blockIdx.y) * (2L * C + 64);
    *reinterpret_cast<f4*>(prow + c) = dgm;
    *reinterpret_cast<f4*>(prow + C + c) = dbt;
    float a1 = (s1[0] + s1[1]) + (s1[2] + s1[3]), a2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    const int lpg = cpg / 4;  // lanes per group: 1, 2, 4 or 8 consecutive lanes (lanes == C / 4 <= 256, so they sit in this half of the if)
    for (int o = 1; o < lpg; o <<= 1) {
      a1 += __shfl_xor(a1, o);
      a2 += __shfl_xor(a2, o);
    }
    if (lane % lpg == 0) {
      prow[2L * C + grp * 2] = a1;
      prow[2L * C + grp * 2 + 1] = a2;
    }
  }
}

// pass 2: dx (+= when accumulate) and the FiLM gradients
template <bool FILM, typename TDY>
__global__ void gn_bwd_apply_kernel(const float* __restrict__ x, const TDY* __restrict__ dy, const float* __restrict__ stats,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, const bf16* __restrict__ film,
                                    const float* __restrict__ sums, float* dx, bf16* __restrict__ dfilm, long total4, int P, int C,
                                    int accumulate, long ldf, const float* dres, bf16* __restrict__ dx_bf, long ldfilm) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // 4 channels of one pixel (they share a group: C / 32 >= 4)
  if (i >= total4) return;
  const unsigned cq = (unsigned)(C / 4);
  const long row = i / cq;
  const int c = (int)(i % cq) * 4;
  const int bt = (int)(row / P);
  const int cpg = C / 32, grp = c / cpg;
  const float mean = stats[((long)bt * 32 + grp) * 2], rstd = stats[((long)bt * 32 + grp) * 2 + 1];
  const float inv_n = 1.0f / ((float)P * (float)cpg);
  const float s1 = sums[((long)bt * 32 + grp) * 2] * inv_n, s2 = sums[((long)bt * 32 + grp) * 2 + 1] * inv_n;
  const long e = row * C + c;
  const f4 xv = *reinterpret_cast<const f4*>(x + e), dv = gn_load_dy4(dy + e);
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  bf16x4 fs, fh, ds, dh;
  if (FILM) {
    fs = *reinterpret_cast<const bf16x4*>(film + row * ldfilm + c);
    fh = *reinterpret_cast<const bf16x4*>(film + row * ldfilm + C + c);
  }
  f4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xh = (xv[j] - mean) * rstd;
    const float gv = xh * ga[j] + be[j];
    float z = gv, mul = 1.f;
    if (FILM) {
      mul = 1.0f + bf2f(fs[j]);
      z = gv * mul + bf2f(fh[j]);
    }
    const float dz = dv[j] * silu_grad(z);
    if (FILM) {
      ds[j] = f2bf(dz * gv);
      dh[j] = f2bf(dz);
    }
    v[j] = rstd * (dz * mul * ga[j] - s1 - xh * s2);
  }
  if (FILM) {
    *reinterpret_cast<bf16x4*>(dfilm + row * ldf + c) = ds;  // ldf: dfilm may be a column block of a wider matrix
    *reinterpret_cast<bf16x4*>(dfilm + row * ldf + C + c) = dh;
  }
  // accumulate: dx = (dres, or the present dx) + v; dx / dx_bf: either output may be absent
  if (accumulate) v += *reinterpret_cast<const f4*>((dres ? dres : dx) + e);
  if (dx) *reinterpret_cast<f4*>(dx + e) = v;
  if (dx_bf) {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
    *reinterpret_cast<bf16x4*>(dx_bf + e) = o;
  }
}

// sums [BT][32][2] scratch; dgamma / dbeta are written (fixed-order sums of per-workgroup partial rows)
template <typename TDY>
int gn_silu_backward(const float* x, const TDY* dy, const float* stats, const float* gamma, const float* beta, const bf16* film, float* sums,
                     float* dx, bf16* dfilm, float* dgamma, float* dbeta, int bt, int P, int C, bool accumulate, hipStream_t s, long ldf = 0,
                     const float* dres = nullptr, bf16* dx_bf = nullptr, long ldfilm = 0) {
  if (ldf == 0) ldf = 2L * C;
  if (ldfilm == 0) ldfilm = 2L * C;
  DFOT_REQUIRE(dx || dx_bf, DFOT_ERR_ARG, "gn_silu_backward: no output");
  DFOT_REQUIRE(!accumulate || dres || dx, DFOT_ERR_ARG, "gn_silu_backward: nothing to accumulate onto");
  DFOT_REQUIRE(C % 128 == 0 && C <= 1024 && 256 % (C / 4 < 256 ? C / 4 : 256) == 0 && (film == nullptr) == (dfilm == nullptr), DFOT_ERR_ARG,
               "gn_silu_backward: channels %d must be 128, 256, 512 or 1024", C);
  const int chunk = P >= 4096 ? 512 : 64;  // pixels per workgroup
  const dim3 grid(bt, cdiv(P, chunk));
  const long total = (long)bt * P * (C / 4);
  // deterministic: one partial row per workgroup, then fixed-order sums (dit_train.inl, det_sum): per-image group sums and per-channel
  // dgamma / dbeta (written, not accumulated)
  const long rowlen = 2L * C + 64;
  float* part = nullptr;
  int rc = det_scratch(2, (size_t)grid.x * grid.y * rowlen, &part);
  if (rc) return rc;
  if (film)
    hipLaunchKernelGGL((gn_bwd_reduce_kernel<true, TDY>), grid, dim3(256), 0, s, x, dy, stats, gamma, beta, film, part, P, C, chunk, ldfilm);
  else
    hipLaunchKernelGGL((gn_bwd_reduce_kernel<false, TDY>), grid, dim3(256), 0, s, x, dy, stats, gamma, beta, film, part, P, C, chunk, ldfilm);
  DFOT_CHECK_HIP(hipGetLastError());
  if ((rc = det_sum(part + 2L * C, rowlen, (int)grid.y, 64, sums, false, s, bt, (long)grid.y * rowlen, 64))) return rc;
  if ((rc = det_sum(part, rowlen, (int)(grid.x * grid.y), C, dgamma, false, s))) return rc;
  if ((rc = det_sum(part + C, rowlen, (int)(grid.x * grid.y), C, dbeta, false, s))) return rc;
  if (film) {
    hipLaunchKernelGGL((gn_bwd_apply_kernel<true, TDY>), dim3(cdiv(total, 256)), dim3(256), 0, s, x, dy, stats, gamma, beta, film, sums, dx, dfilm, total, P, C,
                       accumulate ? 1 : 0, ldf, dres, dx_bf, ldfilm);
  } else {
    hipLaunchKernelGGL((gn_bwd_apply_kernel<false, TDY>), dim3(cdiv(total, 256)), dim3(256), 0, s, x, dy, stats, gamma, beta, film, sums, dx, dfilm, total, P, C,
                       accumulate ? 1 : 0, ldf, dres, dx_bf, ldfilm);
  }
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// GroupNorm statistics (mean, rstd) of x [BT][P][C] fp32 per (image, group): one workgroup per (image, group)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, int P, int C, float eps) {
  __shared__ float red[2][4];
  const int bt = blockIdx.x, grp = blockIdx.y, cpg = C / 32;
  const long n = (long)P * cpg;
  float s = 0.f, q = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float v = x[((long)bt * P + i / cpg) * C + grp * cpg + i % cpg];
    s += v;
    q += v * v;
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), sq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float mean = sum / (float)n;
    const float var = fmaxf(sq / (float)n - mean * mean, 0.f);
    stats[((long)bt * 32 + grp) * 2] = mean;
    stats[((long)bt * 32 + grp) * 2 + 1] = rsqrtf(var + eps);
  }
}

}  // namespace
}  // namespace dfot

extern "C" {
using namespace dfot;
// test entry: x, dy, dx fp32 [BT][P][C]; film (optional) bf16 [BT*P][2C] -> dfilm; dgamma / dbeta fp32 [C]
int dfot_op_gn_silu_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const void* film, float eps, float* dx, void* dfilm,
                        float* dgamma, float* dbeta, int bt, int pixels, int channels, void* stream) {
  DFOT_REQUIRE(x && dy && gamma && beta && dx && dgamma && dbeta, DFOT_ERR_ARG, "op_gn_silu_bwd: null argument");
  hipStream_t s = (hipStream_t)stream;
  float *stats = nullptr, *sums = nullptr;
  DFOT_CHECK_HIP(hipMalloc(&stats, (size_t)bt * 64 * sizeof(float)));
  DFOT_CHECK_HIP(hipMalloc(&sums, (size_t)bt * 64 * sizeof(float)));
  DFOT_CHECK_HIP(hipMemsetAsync(dgamma, 0, (size_t)channels * sizeof(float), s));
  DFOT_CHECK_HIP(hipMemsetAsync(dbeta, 0, (size_t)channels * sizeof(float), s));
  hipLaunchKernelGGL(gn_stats_kernel, dim3(bt, 32), dim3(256), 0, s, x, stats, pixels, channels, eps);
  int rc = gn_silu_backward(x, dy, stats, gamma, beta, (const bf16*)film, sums, dx, (bf16*)dfilm, dgamma, dbeta, bt, pixels, channels, false, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(stats); (void)hipFree(sums);
  return rc;
}
}  // extern "C"

// ---- TransformerBlock pieces (u_vit_blocks.py:96-116,192-281; normalization.py:5-53) ------------------------------------------------
// NormalizeWithCond: xn = RMSNorm(x; w) (1 + scale) + shift, (scale | shift) = film [rows][2C] bf16 (Linear of the token's embedding).
// backward for dxn: dshift = dxn, dscale = dxn y (y = x r w, r = rsqrt(mean x^2 + eps)), g = dxn (1 + scale) w,
//   dx = r (g - x r^2 mean(g x)), dw[c] += sum_rows dxn (1 + scale) x r.   One wave per token row; dw through per-wave partial sums.
namespace dfot {
namespace {

template <int VEC, int CNT>
__global__ __launch_bounds__(256) void rms_film_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dxn, const float* __restrict__ w,
                                                           const bf16* __restrict__ film, float* dx, bf16* __restrict__ dfilm,
                                                           float* __restrict__ dw, long rows, float eps, int accumulate, const float* dres,
                                                           bf16* __restrict__ dx_bf) {
  typedef typename VecT<VEC>::type V;
  constexpr int C = 64 * VEC * CNT;
  const int lane = threadIdx.x & 63;
  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  float dwacc[CNT * VEC];
#pragma unroll
  for (int i = 0; i < CNT * VEC; ++i) dwacc[i] = 0.f;
  for (long row = wave0; row < rows; row += nwaves) {
    const float* xr = x + row * C;
    const float* gr = dxn + row * C;
    const bf16* fr = film + row * 2 * C;
    V xv[CNT], gv[CNT];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      xv[i] = *reinterpret_cast<const V*>(xr + (i * 64 + lane) * VEC);
      ss += vdot<VEC>(xv[i], xv[i]);
    }
    const float r = rsqrtf(wave_sum(ss) / (float)C + eps);
    float gx = 0.f;
    bf16* dfr = dfilm + row * 2 * C;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      const int c0 = (i * 64 + lane) * VEC;
      const V d = *reinterpret_cast<const V*>(gr + c0);
      const V wv = *reinterpret_cast<const V*>(w + c0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float dj, wj, xj;
        if constexpr (VEC == 1) { dj = d; wj = wv; xj = xv[i]; } else { dj = d[j]; wj = wv[j]; xj = xv[i][j]; }
        const float sc = bf2f(fr[c0 + j]);
        const float y = xj * r * wj;
        dfr[c0 + j] = f2bf(dj * y);       // dscale
        dfr[C + c0 + j] = f2bf(dj);       // dshift
        const float dy = dj * (1.0f + sc);
        dwacc[i * VEC + j] += dy * xj * r;
        const float g = dy * wj;
        if constexpr (VEC == 1) gv[i] = g; else gv[i][j] = g;
        gx += g * xj;
      }
    }
    const float m = wave_sum(gx) / (float)C * r * r;
    float* orow = dx + row * C;
    const float* rrow = (dres ? dres : dx) + row * C;  // accumulate: dx = (dres or the present dx) + the norm's input gradient
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      const int c0 = (i * 64 + lane) * VEC;
      V o = (gv[i] - xv[i] * m) * r;
      if (accumulate) o += *reinterpret_cast<const V*>(rrow + c0);
      *reinterpret_cast<V*>(orow + c0) = o;
      if (dx_bf) {  // the next block's GEMM / weight-gradient operand, saving it a cast pass over dx
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float oj;
          if constexpr (VEC == 1) oj = o; else oj = o[j];
          dx_bf[row * C + c0 + j] = f2bf(oj);
        }
      }
    }
  }
  __shared__ float red[4][C];
#pragma unroll
  for (int i = 0; i < CNT; ++i)
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[threadIdx.x >> 6][(i * 64 + lane) * VEC + j] = dwacc[i * VEC + j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) dw[(long)blockIdx.x * C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);  // dw = partial rows
}

int rms_film_backward(const float* x, const float* dxn, const float* w, const bf16* film, float* dx, bf16* dfilm, float* dw, long rows, int hidden,
                      float eps, bool accumulate, hipStream_t s, const float* dres = nullptr, bf16* dx_bf = nullptr) {
  const int grid = (int)(rows / 4 < 512 ? (rows + 3) / 4 : 512);
  float* part = nullptr;  // one partial dw row per workgroup, added in a fixed order (deterministic; dw is written, not accumulated)
  int rc = det_scratch(2, (size_t)grid * hidden, &part);
  if (rc) return rc;
#define CALL(V, C) \
  hipLaunchKernelGGL((rms_film_bwd_kernel<V, C>), dim3(grid), dim3(256), 0, s, x, dxn, w, film, dx, dfilm, part, rows, eps, accumulate ? 1 : 0, dres, dx_bf)
  DIT_LN_DISPATCH(CALL)
#undef CALL
  DFOT_CHECK_HIP(hipGetLastError());
  return det_sum(part, hidden, grid, hidden, dw, false, s);
}

// q / k of the fused projection: per head RMSNorm (weights qw / kw [d]) then RoPE (then q *= qscale, folded into the attention backward's
// "gradient of the unscaled q").  Given the attention backward's dq / dk / dv [B][heads][ntok][d] (d = 64 or 128, no padding) and the
// saved projection output fused [rows][ld] (q | k | v head-major in the first 3C columns):
//   g = rope^T(dq) ; dq_pre = r (g qw - qh mean_d(g qw qh) ...) with qh = q r, r = rsqrt(mean q^2 + eps) ; dqw[e] += sum g qh
// writes d_fused [rows][ldo] columns [0, 3C) (v: plain copy).  One wave per (row, head); a lane owns d/64 consecutive elements.
template <int EPL>  // elements per lane: 1 (d = 64) or 2 (d = 128)
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(const bf16* __restrict__ fused, long ld, const bf16* __restrict__ dq,
                                                              const bf16* __restrict__ dk, const bf16* __restrict__ dv, const float* __restrict__ qw,
                                                              const float* __restrict__ kw, const float* __restrict__ rope_cs,
                                                              bf16* __restrict__ dfused, long ldo, float* __restrict__ dqw, float* __restrict__ dkw,
                                                              long rows, int ntok, int heads, float eps) {
  constexpr int D = 64 * EPL;
  const int lane = threadIdx.x & 63;
  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const int C = heads * D;
  float wacc[2][EPL];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < EPL; ++j) wacc[a][j] = 0.f;
  for (long item = wave0; item < rows * heads; item += nwaves) {
    const long row = item / heads;
    const int head = (int)(item % heads);
    const long b = row / ntok;
    const int tok = (int)(row % ntok);
    const long goff = ((b * heads + head) * ntok + tok) * (long)D + lane * EPL;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const bf16* src = fused + row * ld + (long)which * C + head * D + lane * EPL;
      const bf16* gsrc = (which == 0 ? dq : dk) + goff;
      const float* wt = (which == 0 ? qw : kw) + lane * EPL;
      float xv[EPL], gv[EPL];
      if constexpr (EPL == 2) {  // one 4-byte load per operand instead of two 2-byte ones
        const bf16x2 x2 = *reinterpret_cast<const bf16x2*>(src), g2 = *reinterpret_cast<const bf16x2*>(gsrc);
        xv[0] = bf2f(x2[0]); xv[1] = bf2f(x2[1]); gv[0] = bf2f(g2[0]); gv[1] = bf2f(g2[1]);
      } else {
        xv[0] = bf2f(src[0]); gv[0] = bf2f(gsrc[0]);
      }
      // transpose of the forward rotation (x0 c - x1 s, x1 c + x0 s) on the pair (2i, 2i+1)
      if constexpr (EPL == 2) {
        const float* cs = rope_cs + ((long)tok * (D / 2) + lane) * 2;
        const float g0 = gv[0], g1 = gv[1];
        gv[0] = g0 * cs[0] + g1 * cs[1];
        gv[1] = g1 * cs[0] - g0 * cs[1];
      } else {
        const float* cs = rope_cs + ((long)tok * (D / 2) + (lane >> 1)) * 2;
        const float other = __shfl_xor(gv[0], 1);
        gv[0] = (lane & 1) ? (gv[0] * cs[0] - other * cs[1]) : (gv[0] * cs[0] + other * cs[1]);
      }
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < EPL; ++j) ss += xv[j] * xv[j];
      const float r = rsqrtf(wave_sum(ss) / (float)D + eps);
      float gx = 0.f, gw[EPL];
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        wacc[which][j] += gv[j] * xv[j] * r;
        gw[j] = gv[j] * wt[j];
        gx += gw[j] * xv[j];
      }
      const float m = wave_sum(gx) / (float)D * r * r;
      bf16* dst = dfused + row * ldo + (long)which * C + head * D + lane * EPL;
      if constexpr (EPL == 2) {
        *reinterpret_cast<bf16x2*>(dst) = bf16x2{f2bf((gw[0] - xv[0] * m) * r), f2bf((gw[1] - xv[1] * m) * r)};
      } else {
        dst[0] = f2bf((gw[0] - xv[0] * m) * r);
      }
    }
    bf16* vdst = dfused + row * ldo + 2L * C + head * D + lane * EPL;
    if constexpr (EPL == 2) *reinterpret_cast<bf16x2*>(vdst) = *reinterpret_cast<const bf16x2*>(dv + goff);
    else vdst[0] = dv[goff];
  }
  // weight-gradient partials: the workgroup's four waves are summed in LDS, then one atomic per element and workgroup
  __shared__ float red[4][2][64 * EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    red[threadIdx.x >> 6][0][lane * EPL + j] = wacc[0][j];
    red[threadIdx.x >> 6][1][lane * EPL + j] = wacc[1][j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 64 * EPL; i += 256) {
    const int a = i / (64 * EPL), e = i % (64 * EPL);
    dqw[(long)blockIdx.x * (2 * 64 * EPL) + i] = (red[0][a][e] + red[1][a][e]) + (red[2][a][e] + red[3][a][e]);  // dqw = partial rows [2][D]
  }
}

// the same with 16-byte accesses (used at d = 64): LPH = D / 8 lanes share a (row, head) item, 8 consecutive elements each (a RoPE pair never leaves its
// lane), a wave takes 64 / LPH items at a time; sums over the head by xor shuffles inside the LPH lanes.  The one-wave-per-item form above
// moves 128 or 256 bytes per load instruction (2.1 - 2.7 TB/s on the config-5 shapes); kept for row pitches that are not multiples of 8.
template <int LPH>
__global__ __launch_bounds__(256) void qknorm_rope_bwd_vec_kernel(const bf16* __restrict__ fused, long ld, const bf16* __restrict__ dq,
                                                                  const bf16* __restrict__ dk, const bf16* __restrict__ dv,
                                                                  const float* __restrict__ qw, const float* __restrict__ kw,
                                                                  const float* __restrict__ rope_cs, bf16* __restrict__ dfused, long ldo,
                                                                  float* __restrict__ part, long rows, int ntok, int heads, float eps) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  constexpr int D = 8 * LPH, IPW = 64 / LPH;
  const int lane = threadIdx.x & 63, gl = lane % LPH, grp = lane / LPH, e0 = 8 * gl;
  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const long items = rows * heads;
  const int C = heads * D;
  float wt[2][8], wacc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float* wsrc = (a == 0 ? qw : kw) + e0;
    const f4 w0 = *reinterpret_cast<const f4*>(wsrc), w1 = *reinterpret_cast<const f4*>(wsrc + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { wt[a][j] = w0[j]; wt[a][4 + j] = w1[j]; wacc[a][j] = 0.f; wacc[a][4 + j] = 0.f; }
  }
  for (long blk = wave0; blk * IPW < items; blk += nwaves) {
    const long item = blk * IPW + grp;
    const bool live = item < items;
    const long it = live ? item : 0;
    const long row = it / heads;
    const int head = (int)(it % heads);
    const long b = row / ntok;
    const int tok = (int)(row % ntok);
    const long goff = ((b * heads + head) * ntok + tok) * (long)D + e0;
    const float* csp = rope_cs + ((long)tok * (D / 2) + e0 / 2) * 2;
    const f4 cs0 = *reinterpret_cast<const f4*>(csp), cs1 = *reinterpret_cast<const f4*>(csp + 4);
    const float cs[8] = {cs0[0], cs0[1], cs0[2], cs0[3], cs1[0], cs1[1], cs1[2], cs1[3]};  // (cos, sin) of the lane's four pairs
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const bf16x8 xb = *reinterpret_cast<const bf16x8*>(fused + row * ld + (long)which * C + head * D + e0);
      const bf16x8 gb = *reinterpret_cast<const bf16x8*>((which == 0 ? dq : dk) + goff);
      float xv[8], gv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { xv[j] = bf2f(xb[j]); gv[j] = bf2f(gb[j]); }
      // transpose of the forward rotation (x0 c - x1 s, x1 c + x0 s) on the pairs (2i, 2i+1)
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const float g0 = gv[2 * pr], g1 = gv[2 * pr + 1], c = cs[2 * pr], sn = cs[2 * pr + 1];
        gv[2 * pr] = g0 * c + g1 * sn;
        gv[2 * pr + 1] = g1 * c - g0 * sn;
      }
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += xv[j] * xv[j];
#pragma unroll
      for (int o = 1; o < LPH; o <<= 1) ss += __shfl_xor(ss, o);
      const float r = rsqrtf(ss / (float)D + eps);
      float gx = 0.f, gw[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (live) wacc[which][j] += gv[j] * xv[j] * r;
        gw[j] = gv[j] * wt[which][j];
        gx += gw[j] * xv[j];
      }
#pragma unroll
      for (int o = 1; o < LPH; o <<= 1) gx += __shfl_xor(gx, o);
      const float m = gx / (float)D * r * r;
      bf16x8 ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) ob[j] = f2bf((gw[j] - xv[j] * m) * r);
      if (live) *reinterpret_cast<bf16x8*>(dfused + row * ldo + (long)which * C + head * D + e0) = ob;
    }
    if (live) *reinterpret_cast<bf16x8*>(dfused + row * ldo + 2L * C + head * D + e0) = *reinterpret_cast<const bf16x8*>(dv + goff);
  }
  // the wave's items: lanes with the same position in the head (gl) are summed by xor shuffles over the item groups
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int o = LPH; o < 64; o <<= 1) wacc[a][j] += __shfl_xor(wacc[a][j], o);
  __shared__ float red[4][2][D];
  if (lane < LPH) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[threadIdx.x >> 6][a][e0 + j] = wacc[a][j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const int a = i / D, e = i % D;
    part[(long)blockIdx.x * (2 * D) + i] = (red[0][a][e] + red[1][a][e]) + (red[2][a][e] + red[3][a][e]);  // partial row [2][D] of this workgroup
  }
}

int qknorm_rope_backward(const bf16* fused, long ld, const bf16* dq, const bf16* dk, const bf16* dv, const float* qw, const float* kw,
                         const float* rope_cs, bf16* dfused, long ldo, float* dqw, float* dkw, long rows, int ntok, int heads, int d, float eps,
                         hipStream_t s) {
  DFOT_REQUIRE(d == 64 || d == 128, DFOT_ERR_SHAPE, "qknorm_rope_backward: head dim %d not in {64,128}", d);
  const long items = rows * heads;
  float* part = nullptr;  // per-workgroup partial rows (dqw [d] | dkw [d]), added in a fixed order: deterministic, written not accumulated
  int rc = 0;
  // d = 128: the one-wave-per-item form already moves 256 bytes per instruction and is the faster one there (108 vs 134 us at level 3)
  if (d == 64 && ld % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)fused & 15) == 0 && ((uintptr_t)dfused & 15) == 0) {
    const int ipw = 8;
    const long wblocks = cdiv(items, (long)ipw);
    const int vgrid = (int)(wblocks / 4 < 2048 ? (wblocks + 3) / 4 : 2048);
    if ((rc = det_scratch(2, (size_t)vgrid * 2 * d, &part))) return rc;
    hipLaunchKernelGGL(qknorm_rope_bwd_vec_kernel<8>, dim3(vgrid), dim3(256), 0, s, fused, ld, dq, dk, dv, qw, kw, rope_cs, dfused, ldo, part, rows, ntok, heads, eps);
    DFOT_CHECK_HIP(hipGetLastError());
    if ((rc = det_sum(part, 2L * d, vgrid, d, dqw, false, s))) return rc;
    return det_sum(part + d, 2L * d, vgrid, d, dkw, false, s);
  }
  const int grid = (int)(items / 4 < 1024 ? (items + 3) / 4 : 1024);
  if ((rc = det_scratch(2, (size_t)grid * 2 * d, &part))) return rc;
  if (d == 64)
    hipLaunchKernelGGL(qknorm_rope_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, fused, ld, dq, dk, dv, qw, kw, rope_cs, dfused, ldo, part, nullptr, rows, ntok, heads, eps);
  else
    hipLaunchKernelGGL(qknorm_rope_bwd_kernel<2>, dim3(grid), dim3(256), 0, s, fused, ld, dq, dk, dv, qw, kw, rope_cs, dfused, ldo, part, nullptr, rows, ntok, heads, eps);
  DFOT_CHECK_HIP(hipGetLastError());
  if ((rc = det_sum(part, 2L * d, grid, d, dqw, false, s))) return rc;
  return det_sum(part + d, 2L * d, grid, d, dkw, false, s);
}

}  // namespace
}  // namespace dfot

extern "C" {
using namespace dfot;
// test entries (dw / dqw / dkw are written, not accumulated)
int dfot_op_rms_film_bwd(const float* x, const float* dxn, const float* w, const void* film, float eps, float* dx, void* dfilm, float* dw,
                         int64_t rows, int channels, int accumulate_dx, void* stream) {
  DFOT_REQUIRE(x && dxn && w && film && dx && dfilm && dw, DFOT_ERR_ARG, "op_rms_film_bwd: null argument");
  return rms_film_backward(x, dxn, w, (const bf16*)film, dx, (bf16*)dfilm, dw, (long)rows, channels, eps, accumulate_dx != 0, (hipStream_t)stream);
}
// the same with the residual-path gradient read from `dres` (not modified): dx = dres + the norm's input gradient, out of place
int dfot_op_rms_film_bwd_res(const float* x, const float* dxn, const float* w, const void* film, float eps, const float* dres, float* dx, void* dx_bf,
                             void* dfilm, float* dw, int64_t rows, int channels, void* stream) {
  DFOT_REQUIRE(x && dxn && w && film && dres && dx && dfilm && dw && dres != dx, DFOT_ERR_ARG, "op_rms_film_bwd_res: null or aliased argument");
  return rms_film_backward(x, dxn, w, (const bf16*)film, dx, (bf16*)dfilm, dw, (long)rows, channels, eps, true, (hipStream_t)stream, dres, (bf16*)dx_bf);
}
int dfot_op_qknorm_rope_bwd(const void* fused, int ld, const void* dq, const void* dk, const void* dv, const float* qw, const float* kw,
                            const float* rope_cs, float eps, void* dfused, int ldo, float* dqw, float* dkw, int64_t rows, int ntok, int heads, int d,
                            void* stream) {
  DFOT_REQUIRE(fused && dq && dk && dv && qw && kw && rope_cs && dfused && dqw && dkw, DFOT_ERR_ARG, "op_qknorm_rope_bwd: null argument");
  return qknorm_rope_backward((const bf16*)fused, ld, (const bf16*)dq, (const bf16*)dk, (const bf16*)dv, qw, kw, rope_cs, (bf16*)dfused, ldo, dqw, dkw,
                              (long)rows, ntok, heads, d, eps, (hipStream_t)stream);
}
}  // extern "C"

extern "C" {
using namespace dfot;
// test entry: out [M][N] fp32 = a^T b with a [rows][lda], b [rows][ldb] bf16 (the weight gradient dY^T X in the activations' own layout)
int dfot_op_wgrad_nt(const void* a, int lda, const void* b, int ldb, float* out, int m, int n, int64_t rows, int slices, void* stream) {
  DFOT_REQUIRE(a && b && out && slices >= 0, DFOT_ERR_ARG, "op_wgrad_nt: bad argument");
  hipStream_t s = (hipStream_t)stream;
  // slices == 0: tile form and K slices chosen by shape (wgrad_plan); slices >= 1: the 128 x 128 form with that many slices
  const WgradPlan plan = slices == 0 ? wgrad_plan(m, n, (long)rows, 64) : WgradPlan{0, slices};
  if (plan.slices == 1) return launch_wgrad_nt_plan((const bf16*)a, lda, (const bf16*)b, ldb, out, m, n, (long)rows, plan, s);
  void* wsv = nullptr;
  int rc = op_scratch(5, (size_t)plan.slices * m * n * sizeof(float), &wsv);
  if (rc) return rc;
  float* ws = (float*)wsv;
  if ((rc = launch_wgrad_nt_plan((const bf16*)a, lda, (const bf16*)b, ldb, ws, m, n, (long)rows, plan, s))) return rc;
  hipLaunchKernelGGL(slices_sum_kernel, dim3(cdiv((long)m * n / 4, 256)), dim3(256), 0, s, ws, out, (long)m * n / 4, plan.slices, (long)m * n);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
}  // extern "C"

// ---- forward passes in training form (values the backward needs are kept: no fused norm / activation epilogues) and generic op
// entry points, so that a UViT training driver can be written over the C ABI op by op -------------------------------------------
namespace dfot {
namespace {

template <int VEC, int CNT>
__global__ __launch_bounds__(256) void rms_film_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const bf16* __restrict__ film,
                                                           bf16* __restrict__ out, long rows, float eps) {
  typedef typename VecT<VEC>::type V;
  constexpr int C = 64 * VEC * CNT;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  const bf16* fr = film + row * 2 * C;
  V xv[CNT];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    xv[i] = *reinterpret_cast<const V*>(xr + (i * 64 + lane) * VEC);
    ss += vdot<VEC>(xv[i], xv[i]);
  }
  const float r = rsqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const int c0 = (i * 64 + lane) * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float xj;
      if constexpr (VEC == 1) xj = xv[i]; else xj = xv[i][j];
      out[row * C + c0 + j] = f2bf(xj * r * w[c0 + j] * (1.0f + bf2f(fr[c0 + j])) + bf2f(fr[C + c0 + j]));
    }
  }
}

// fused [rows][ld] (q | k | v head-major) -> q (RMSNorm, RoPE, * qscale), k (RMSNorm, RoPE), v in the attention layout [B][heads][ntok][d]
template <int EPL>
__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(const bf16* __restrict__ fused, long ld, const float* __restrict__ qw,
                                                              const float* __restrict__ kw, const float* __restrict__ rope_cs, bf16* __restrict__ q,
                                                              bf16* __restrict__ k, bf16* __restrict__ v, long rows, int ntok, int heads, float eps,
                                                              float qscale) {
  constexpr int D = 64 * EPL;
  const int lane = threadIdx.x & 63;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= rows * heads) return;
  const long row = item / heads;
  const int head = (int)(item % heads);
  const long b = row / ntok;
  const int tok = (int)(row % ntok);
  const int C = heads * D;
  const long ooff = ((b * heads + head) * ntok + tok) * (long)D + lane * EPL;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const bf16* src = fused + row * ld + (long)which * C + head * D + lane * EPL;
    const float* wt = (which == 0 ? qw : kw) + lane * EPL;
    float xv[EPL];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) { xv[j] = bf2f(src[j]); ss += xv[j] * xv[j]; }
    const float r = rsqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < EPL; ++j) xv[j] = xv[j] * r * wt[j];
    const float mul = which == 0 ? qscale : 1.f;
    bf16* dst = (which == 0 ? q : k) + ooff;
    if constexpr (EPL == 2) {
      const float* cs = rope_cs + ((long)tok * (D / 2) + lane) * 2;
      dst[0] = f2bf((xv[0] * cs[0] - xv[1] * cs[1]) * mul);
      dst[1] = f2bf((xv[1] * cs[0] + xv[0] * cs[1]) * mul);
    } else {
      const float* cs = rope_cs + ((long)tok * (D / 2) + (lane >> 1)) * 2;
      const float other = __shfl_xor(xv[0], 1);
      dst[0] = f2bf(((lane & 1) ? (xv[0] * cs[0] + other * cs[1]) : (xv[0] * cs[0] - other * cs[1])) * mul);
    }
  }
  const bf16* vs = fused + row * ld + 2L * C + head * D + lane * EPL;
#pragma unroll
  for (int j = 0; j < EPL; ++j) v[ooff + j] = vs[j];
}

// dst[r][dcol0 + c] = SiLU(src[r][scol0 + c])            (grad == nullptr)
// dst[r][dcol0 + c] = grad[r][gcol0 + c] * SiLU'(src...)  (grad != nullptr);   8 columns per thread
__global__ void silu_cols_kernel(const bf16* __restrict__ src, long lds_, int scol0, const bf16* __restrict__ grad, long ldg, int gcol0,
                                 bf16* __restrict__ dst, long ldd, int dcol0, long rows, int ncols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = ncols / 8;
  if (i >= rows * c8) return;
  const long r = i / c8;
  const int c = (int)(i % c8) * 8;
  const bf16x8 xv = *reinterpret_cast<const bf16x8*>(src + r * lds_ + scol0 + c);
  bf16x8 o;
  if (grad) {
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(grad + r * ldg + gcol0 + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(gv[j]) * silu_grad(bf2f(xv[j])));
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(silu_f(bf2f(xv[j])));
  }
  *reinterpret_cast<bf16x8*>(dst + r * ldd + dcol0 + c) = o;
}

}  // namespace
}  // namespace dfot

extern "C" {
using namespace dfot;

int dfot_op_gemm_bf16(const void* a, int lda, const void* w, const float* bias, void* out, int ldo, int m, int n, int k, void* stream) {
  DFOT_REQUIRE(a && w && out, DFOT_ERR_ARG, "op_gemm_bf16: null argument");
  return tr_gemm_bf16((const bf16*)a, lda, (const bf16*)w, m, n, k, bias, (bf16*)out, ldo, (hipStream_t)stream);
}
// out bf16 [M][N] = a w^T + frame_bias[row / rows_per_frame][:]   (frame_bias fp32 [M / rows_per_frame][N]): the folded FiLM projection, whose
// per-frame part rides in the GEMM epilogue
int dfot_op_gemm_bf16_frame_bias(const void* a, int lda, const void* w, const float* frame_bias, int rows_per_frame, void* out, int ldo, int m, int n, int k,
                                 void* stream) {
  DFOT_REQUIRE(a && w && out && frame_bias && rows_per_frame > 0 && m % rows_per_frame == 0 && n % 8 == 0, DFOT_ERR_ARG,
               "op_gemm_bf16_frame_bias: null argument, or rows_per_frame = %d does not divide M = %d", rows_per_frame, m);
  return tr_gemm_bf16((const bf16*)a, lda, (const bf16*)w, m, n, k, frame_bias, (bf16*)out, ldo, (hipStream_t)stream, -rows_per_frame);
}
int dfot_op_gemm_f32(const void* a, int lda, const void* w, const float* bias, const float* resid, float* out, int ldo, int m, int n, int k,
                     void* stream) {
  DFOT_REQUIRE(a && w && out, DFOT_ERR_ARG, "op_gemm_f32: null argument");
  GemmArgs g;
  g.A = (const bf16*)a; g.lda = lda; g.W = (const bf16*)w; g.M = m; g.N = n; g.K = k; g.bias = bias; g.out_f32 = out; g.ldo = ldo; g.resid = resid;
  return launch_gemm(A_DENSE, E_F32, GEMM_AUTO, g, (hipStream_t)stream);
}
int dfot_op_transpose_bf16(const void* src, void* dst, int rows, int cols, void* stream) {
  DFOT_REQUIRE(src && dst, DFOT_ERR_ARG, "op_transpose_bf16: null argument");
  return tr_transpose((const bf16*)src, (bf16*)dst, rows, cols, (hipStream_t)stream);
}
int dfot_op_colsum_bf16(const void* src, int ld, float* out, int64_t rows, int n, void* stream) {
  DFOT_REQUIRE(src && out, DFOT_ERR_ARG, "op_colsum_bf16: null argument");
  hipStream_t s = (hipStream_t)stream;
  DFOT_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)n * sizeof(float), s));
  return launch_colsum_bf16((const bf16*)src, out, (long)rows, n, (long)ld, s);
}
int dfot_op_rms_film_fwd(const float* x, const float* w, const void* film, float eps, void* out, int64_t rows, int channels, void* stream) {
  DFOT_REQUIRE(x && w && film && out, DFOT_ERR_ARG, "op_rms_film_fwd: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int hidden = channels;
#define CALL(V, C) \
  hipLaunchKernelGGL((rms_film_fwd_kernel<V, C>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, w, (const bf16*)film, (bf16*)out, (long)rows, eps)
  DIT_LN_DISPATCH(CALL)
#undef CALL
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int dfot_op_qknorm_rope_fwd(const void* fused, int ld, const float* qw, const float* kw, const float* rope_cs, float eps, float qscale, void* q,
                            void* k, void* v, int64_t rows, int ntok, int heads, int d, void* stream) {
  DFOT_REQUIRE(fused && qw && kw && rope_cs && q && k && v && (d == 64 || d == 128), DFOT_ERR_ARG, "op_qknorm_rope_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(cdiv((long)rows * heads, 4));
  if (d == 64)
    hipLaunchKernelGGL(qknorm_rope_fwd_kernel<1>, grid, dim3(256), 0, s, (const bf16*)fused, (long)ld, qw, kw, rope_cs, (bf16*)q, (bf16*)k, (bf16*)v,
                       (long)rows, ntok, heads, eps, qscale);
  else
    hipLaunchKernelGGL(qknorm_rope_fwd_kernel<2>, grid, dim3(256), 0, s, (const bf16*)fused, (long)ld, qw, kw, rope_cs, (bf16*)q, (bf16*)k, (bf16*)v,
                       (long)rows, ntok, heads, eps, qscale);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// Training forward of fused_attn_mlp_proj in ONE launch (u_vit_blocks.py:253-262): fused = a W^T + bias is kept raw (bf16 [rows][7C], what the
// backward of the QK-norm and of SiLU needs) while the same epilogue writes q, k (per-head RMSNorm + RoPE, q scaled) and v as
// [B][heads][ntok][d] and SiLU(mlp_h) into cat[:, ccol0 : ccol0 + 4C] -- instead of a plain GEMM followed by a norm / RoPE pass and a
// SiLU pass that each re-read the 7C-wide projection.  a bf16 [rows][lda], w bf16 [7C][C]
int dfot_op_fused_proj_train(const void* a, int lda, const void* w, const float* bias, const float* qw, const float* kw, const float* rope_cs, float eps,
                             float qscale, void* fused, void* q, void* k, void* v, void* cat, int ldcat, int ccol0, int64_t rows, int ntok, int heads,
                             int d, void* stream) {
  DFOT_REQUIRE(a && w && bias && qw && kw && rope_cs && fused && q && k && v && cat && (d == 64 || d == 128) && ccol0 % 8 == 0 && rows % 128 == 0,
               DFOT_ERR_ARG, "op_fused_proj_train: bad argument");
  const int c = heads * d;
  GemmArgs g;
  g.A = (const bf16*)a; g.lda = lda; g.W = (const bf16*)w; g.M = (int)rows; g.N = 7 * c; g.K = c; g.bias = bias;
  g.out2 = (bf16*)cat + ccol0; g.ldo2 = ldcat; g.split = 3 * c;
  g.q = (bf16*)q; g.k = (bf16*)k; g.v = (bf16*)v; g.qw = qw; g.kw = kw; g.rope_cs = rope_cs; g.heads = heads; g.d = d; g.ntok = ntok;
  g.qscale = qscale; g.eps = eps;
  g.raw = (bf16*)fused; g.ldraw = 7L * c;
  return launch_gemm(A_DENSE, E_QKV, GEMM_AUTO, g, (hipStream_t)stream);
}
// grad == NULL: dst[:, dcol0:+ncols] = SiLU(src[:, scol0:+ncols]); else dst = grad[:, gcol0:+ncols] * SiLU'(src[:, scol0:+ncols])
int dfot_op_silu_cols(const void* src, int lds_, int scol0, const void* grad, int ldg, int gcol0, void* dst, int ldd, int dcol0, int64_t rows,
                      int ncols, void* stream) {
  DFOT_REQUIRE(src && dst && ncols % 8 == 0 && scol0 % 8 == 0 && dcol0 % 8 == 0 && gcol0 % 8 == 0, DFOT_ERR_ARG, "op_silu_cols: bad argument");
  hipLaunchKernelGGL(silu_cols_kernel, dim3(cdiv((long)rows * (ncols / 8), 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (long)lds_, scol0,
                     (const bf16*)grad, (long)ldg, gcol0, (bf16*)dst, (long)ldd, dcol0, (long)rows, ncols);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// attention with the log-sum-exp kept (lse [B][heads][N] fp32) and its backward from saved q / k / v / o / lse (delta: scratch like lse)
int dfot_op_attention_fwd_lse(const void* q, const void* k, const void* v, void* o, int ldo, float* lse, int batch, int heads, int n, int d,
                              void* stream) {
  return launch_attention_padded((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, ldo, batch, heads, n, d, (hipStream_t)stream, lse);
}
// ... for a caller that knows a bound of |q.k| log2(e)/sqrt(d) (UViT blocks: from the q_norm / k_norm weights, u_vit_blocks.py:255-262;
// the bound dfot_uvit_query reports): below 64 the d = 64 launch takes the pipelined kernel without a running max (attention_v5.hip;
// lse = log2 of the row sum), which the backward consumes unchanged (it recomputes P = exp2(S - lse))
size_t dfot_op_attention_scratch_bytes(int batch, int heads, int n, int d) { return attention_scratch_bytes(batch, heads, n, d); }
int dfot_op_attention_fwd_lse_bounded(const void* q, const void* k, const void* v, void* o, int ldo, float* lse, int batch, int heads, int n, int d,
                                      float score_bound, void* scratch, size_t scratch_bytes, void* stream) {
  // NaN-safe: only a bound that IS below 64 selects the kernel without a running max
  if (d == 64 && n % 256 == 0 && score_bound < 64.0f && lse) {
    // the key-split partial rows live in the CALLER's buffer (one per trainer / stream: nothing is allocated on this launch path and
    // two trainers never share partial rows); a null buffer falls back to the process-wide grow-only block
    AttnScratch own{reinterpret_cast<float*>(scratch), scratch_bytes};
    return launch_attention_v5((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, ldo, batch, heads, n, (hipStream_t)stream,
                               scratch ? &own : nullptr, lse);
  }
  return launch_attention_padded((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, ldo, batch, heads, n, d, (hipStream_t)stream, lse);
}
int dfot_op_attention_bwd_lse(const void* q, const void* k, const void* v, const void* o, const void* d_o, int ldo, const float* lse, float* delta,
                              void* dq, void* dk, void* dv, int batch, int heads, int n, int d, void* stream) {
  DFOT_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, DFOT_ERR_ARG, "op_attention_bwd_lse: null argument");
  hipStream_t s = (hipStream_t)stream;
  int rc = launch_attention_bwd_delta((const bf16*)o, (const bf16*)d_o, ldo, delta, batch, heads, n, d, s);
  if (rc) return rc;
  return launch_attention_bwd((const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)d_o, ldo, lse, delta, (bf16*)dq, (bf16*)dk, (bf16*)dv, batch,
                              heads, n, d, s);
}
}  // extern "C"

// ---- ResBlock / resampler / embedding pieces for the UViT training driver ------------------------------------------------------------
namespace dfot {
namespace {

// out bf16 = SiLU(GN(x) [* (1 + scale) + shift])   (x fp32 [BT][P][C], stats [BT][32][2], film bf16 [BT*P][2C] or null)
__global__ void gn_silu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const bf16* __restrict__ film, bf16* __restrict__ out, long total4, int P, int C,
                                   long ldfilm) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // 4 channels of one pixel (one group: C / 32 >= 4)
  if (i >= total4) return;
  const unsigned cq = (unsigned)(C / 4);
  const long row = i / cq;
  const int c = (int)(i % cq) * 4;
  const int bt = (int)(row / P), grp = c / (C / 32);
  const float mean = stats[((long)bt * 32 + grp) * 2], rstd = stats[((long)bt * 32 + grp) * 2 + 1];
  const f4 xv = *reinterpret_cast<const f4*>(x + row * C + c);
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  bf16x4 fs, fh, o;
  if (film) {  // ldfilm: the block's (scale | shift) columns may be a column block of the level's FiLM matrix
    fs = *reinterpret_cast<const bf16x4*>(film + row * ldfilm + c);
    fh = *reinterpret_cast<const bf16x4*>(film + row * ldfilm + C + c);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float z = (xv[j] - mean) * rstd * ga[j] + be[j];
    if (film) z = z * (1.0f + bf2f(fs[j])) + bf2f(fh[j]);
    o[j] = f2bf(silu_f(z));
  }
  *reinterpret_cast<bf16x4*>(out + row * C + c) = o;
}

// dx[bt][2y+a][2x+b][c] += dp[bt][y][x][c] / 4   (adjoint of the 2x2 average pool; dx fp32 [BT][H][W][C], dp fp32 [BT][H/2][W/2][C])
// 4 channels per thread (C % 4 == 0: launcher); total = elements / 4
__global__ void pool2_bwd_kernel(const float* __restrict__ dp, float* __restrict__ dx, long total, int H, int W, int C) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int cq = C / 4;
  const int c = (int)(e % cq) * 4;
  const long pix = e / cq;
  const int x = (int)(pix % W), y = (int)((pix / W) % H);
  const long bt = pix / ((long)W * H);
  const f4 v = *reinterpret_cast<const f4*>(dp + ((bt * (H / 2) + y / 2) * (W / 2) + x / 2) * C + c);
  *reinterpret_cast<f4*>(dx + pix * C + c) += v * 0.25f;
}
// ds[bt][y][x][c] = sum of the 2x2 block of dy (adjoint of the nearest-neighbour upsample; dy fp32 [BT][H][W][C])
// 4 channels per thread (C % 4 == 0: launcher)
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ ds, long total, int H, int W, int C) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;  // total = BT * (H/2) * (W/2) * C / 4
  const int cq = C / 4;
  const int c = (int)(e % cq) * 4;
  const long pix = e / cq;
  const int x = (int)(pix % (W / 2)), y = (int)((pix / (W / 2)) % (H / 2));
  const long bt = pix / ((long)(W / 2) * (H / 2));
  const float* b = dy + ((bt * H + 2 * y) * W + 2 * x) * C + c;
  auto ld = [&](long off) { return *reinterpret_cast<const f4*>(b + off); };
  *reinterpret_cast<f4*>(ds + pix * C + c) = (ld(0) + ld(C)) + (ld((long)W * C) + ld((long)W * C + C));
}
// y = a + alpha * b  (fp32, in place on a)
__global__ void axpy_kernel(float* __restrict__ a, const float* __restrict__ b, float alpha, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += alpha * b[i];
}
__global__ void axpy_kernel4(float* __restrict__ a, const float* __restrict__ b, float alpha, long n4) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) reinterpret_cast<f4*>(a)[i] += reinterpret_cast<const f4*>(b)[i] * alpha;
}
// dst[r][dcol0 + c] *= mask[r][c]   (dropout: mask holds 0 or 1 / (1 - p)); 8 columns per thread
__global__ void mul_cols_kernel(bf16* __restrict__ dst, long ldd, int dcol0, const bf16* __restrict__ mask, long rows, int ncols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = ncols / 8;
  if (i >= rows * c8) return;
  const long r = i / c8;
  const int c = (int)(i % c8) * 8;
  bf16x8 v = *reinterpret_cast<bf16x8*>(dst + r * ldd + dcol0 + c);
  const bf16x8 m = *reinterpret_cast<const bf16x8*>(mask + r * ncols + c);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * bf2f(m[j]));
  *reinterpret_cast<bf16x8*>(dst + r * ldd + dcol0 + c) = v;
}
// hi = bf16(x), lo = bf16(x - hi): 8 elements per thread
__global__ void split_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ hi, bf16* __restrict__ lo, long n8) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const f4 a = reinterpret_cast<const f4*>(x)[2 * i], b = reinterpret_cast<const f4*>(x)[2 * i + 1];
  bf16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = j < 4 ? a[j] : b[j - 4];
    h[j] = f2bf(v);
    l[j] = f2bf(v - bf2f(h[j]));
  }
  reinterpret_cast<bf16x8*>(hi)[i] = h;
  reinterpret_cast<bf16x8*>(lo)[i] = l;
}
// part[z][bt][c] = sum over the pixel chunk z of src[bt * P + p][c] (bf16 rows of pitch ld); 8 columns per thread
__global__ __launch_bounds__(256) void frame_sums_bf16_kernel(const bf16* __restrict__ src, long ld, float* __restrict__ part, int P, int n) {
  const int bt = blockIdx.y;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (c >= n) return;
  const int per = (P + gridDim.z - 1) / gridDim.z, p0 = blockIdx.z * per, p1 = p0 + per < P ? p0 + per : P;
  const bf16* b = src + (long)bt * P * ld + c;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 4
  for (int p = p0; p < p1; ++p) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(b + (long)p * ld);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
  }
  float* o = part + ((long)blockIdx.z * gridDim.y + bt) * n + c;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = acc[j];
}
// 64 x 64 output tile per workgroup, 16-deep K steps through LDS, 4 x 4 outputs per thread; strides in elements
__global__ __launch_bounds__(256) void sgemm_strided_kernel(const float* __restrict__ A, long sa_i, long sa_k, const float* __restrict__ B, long sb_k,
                                                            long sb_j, float* __restrict__ C, long ldc, int M, int N, int K, int accumulate) {
  __shared__ float As[16][64 + 4], Bs[16][64 + 4];
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int ti = (threadIdx.x / 16) * 4, tj = (threadIdx.x % 16) * 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int t = threadIdx.x; t < 16 * 64; t += 256) {
      // the fast index follows the operand's unit stride, so either orientation loads coalesced
      const int kk_a = sa_k == 1 ? t % 16 : t / 64, ii = sa_k == 1 ? t / 16 : t % 64;
      As[kk_a][ii] = (i0 + ii < M && k0 + kk_a < K) ? A[(long)(i0 + ii) * sa_i + (long)(k0 + kk_a) * sa_k] : 0.f;
      const int kk_b = sb_k == 1 ? t % 16 : t / 64, jj = sb_k == 1 ? t / 16 : t % 64;
      Bs[kk_b][jj] = (j0 + jj < N && k0 + kk_b < K) ? B[(long)(k0 + kk_b) * sb_k + (long)(j0 + jj) * sb_j] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) av[x] = As[kk][ti + x], bv[x] = Bs[kk][tj + x];
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] += av[x] * bv[y];
    }
    __syncthreads();
  }
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int i = i0 + ti + x, j = j0 + tj + y;
      if (i < M && j < N) C[(long)i * ldc + j] = (accumulate ? C[(long)i * ldc + j] : 0.f) + acc[x][y];
    }
}
// gradient of the ConvTranspose(k = s = p) output [BT][Co][R][R] gathered per input pixel: dpatch [pix][64] bf16, column (co, py, px)
__global__ void outgrad_gather_kernel(const float* __restrict__ dout, bf16* __restrict__ dpatch, long pix, int R, int co, int ps) {
  const int n = co * ps * ps;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pix * n) return;
  const long p = i / n;
  const int col = (int)(i % n);
  const int o = col / (ps * ps), py = (col / ps) % ps, px = col % ps;
  const int g = R / ps;
  const int x = (int)(p % g), y = (int)((p / g) % g);
  const long bt = p / ((long)g * g);
  dpatch[p * 64 + col] = f2bf(dout[((bt * co + o) * R + y * ps + py) * R + x * ps + px]);
}

}  // namespace
}  // namespace dfot

// one thread per 2x2 patch: dX[bt][ci][2py+dy][2px+dx] = sum_c dY[patch][c] * W[c][ci*4 + dy*2 + dx]; W staged in LDS (read as a broadcast)
__global__ __launch_bounds__(256) void embed_input_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                float* __restrict__ dx, long patches, int res, int cin, int c0) {
  extern __shared__ float wsh[];  // [c0][cin * 4]
  const int taps = cin * 4;
  for (int i = threadIdx.x; i < c0 * taps; i += 256) wsh[i] = w[i];
  __syncthreads();
  const long patch = (long)blockIdx.x * 256 + threadIdx.x;
  if (patch >= patches) return;
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  const float* row = dy + patch * c0;
  for (int c = 0; c < c0; c += 4) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        if (t < taps) acc[t] += d[j] * wsh[(c + j) * taps + t];
  }
  const unsigned r0 = (unsigned)res / 2;
  const unsigned pu = (unsigned)(patch % ((long)r0 * r0));
  const long b = patch / ((long)r0 * r0);
  const int py = (int)(pu / r0), px = (int)(pu % r0);
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (t < taps) {
      const int ci = t >> 2, ddy = (t >> 1) & 1, ddx = t & 1;
      dx[((b * cin + ci) * res + 2 * py + ddy) * (long)res + 2 * px + ddx] = acc[t];
    }
}

extern "C" {
using namespace dfot;

int dfot_op_gn_silu_fwd(const float* x, const float* gamma, const float* beta, const void* film, float eps, void* out, float* stats, int bt,
                        int pixels, int channels, void* stream) {
  return dfot_op_gn_silu_fwd2(x, gamma, beta, film, 2 * channels, eps, out, stats, bt, pixels, channels, stream);
}
int dfot_op_gn_silu_fwd2(const float* x, const float* gamma, const float* beta, const void* film, int64_t film_ld, float eps, void* out, float* stats,
                         int bt, int pixels, int channels, void* stream) {
  DFOT_REQUIRE(x && gamma && beta && out && stats && channels % 32 == 0 && (!film || (film_ld >= 2 * channels && film_ld % 4 == 0)), DFOT_ERR_ARG,
               "op_gn_silu_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  DFOT_REQUIRE(channels % 128 == 0, DFOT_ERR_SHAPE, "op_gn_silu_fwd: channels %d must be a multiple of 128", channels);
  // statistics: streaming partial sums over 64-pixel blocks of all channels + a deterministic finalize (the inference kernels); one
  // workgroup per (image, group) read 16-byte slivers of every 512-byte pixel row
  void* part = nullptr;
  const int nblk = gn_partial_blocks(pixels);
  int rc = op_scratch(6, (size_t)bt * nblk * 64 * sizeof(float), &part);
  if (rc) return rc;
  if ((rc = launch_gn_partial_f32(x, (float*)part, bt, pixels, channels, s))) return rc;
  if ((rc = launch_gn_finalize((const float*)part, stats, bt, nblk, pixels, channels, eps, s))) return rc;
  const long total = (long)bt * pixels * (channels / 4);
  hipLaunchKernelGGL(gn_silu_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, x, stats, gamma, beta, (const bf16*)film, (bf16*)out, total, pixels,
                     channels, (long)film_ld);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// backward with saved statistics and the upstream gradient in bf16 (what dfot_op_conv3x3_bwd2 leaves in dx_bf): dx = (dres ? dres : 0) + the
// norm's input gradient, written as fp32 (dx) and / or bf16 (dx_bf) -- the gradient that only feeds a convolution's data / weight gradient
// is needed in bf16 alone, the one that continues down the residual stream in both; dgamma / dbeta are written (deterministic two-stage sums)
int dfot_op_gn_silu_bwd5(const float* x, const void* dy_bf, const float* stats, const float* gamma, const float* beta, const void* film, const float* dres,
                         float* dx, void* dx_bf, void* dfilm, int64_t dfilm_ld, float* dgamma, float* dbeta, int bt, int pixels, int channels,
                         void* stream) {
  DFOT_REQUIRE(x && dy_bf && stats && gamma && beta && (dx || dx_bf) && dgamma && dbeta && (!dres || dres != dx), DFOT_ERR_ARG,
               "op_gn_silu_bwd5: null or aliased argument");
  DFOT_REQUIRE(!dfilm || (dfilm_ld >= 2 * channels && dfilm_ld % 4 == 0), DFOT_ERR_ARG, "op_gn_silu_bwd5: bad dfilm row stride");
  hipStream_t s = (hipStream_t)stream;
  void* sums = nullptr;
  int rc = op_scratch(4, (size_t)bt * 64 * sizeof(float), &sums);
  if (rc) return rc;
  return gn_silu_backward(x, (const bf16*)dy_bf, stats, gamma, beta, (const bf16*)film, (float*)sums, dx, (bf16*)dfilm, dgamma, dbeta, bt, pixels,
                          channels, dres != nullptr, s, (long)dfilm_ld, dres, (bf16*)dx_bf);
}
// dfot_op_gn_silu_bwd5 with the block's film columns given as a column block of a wider matrix (row pitch film_ld)
int dfot_op_gn_silu_bwd6(const float* x, const void* dy_bf, const float* stats, const float* gamma, const float* beta, const void* film,
                         int64_t film_ld, const float* dres, float* dx, void* dx_bf, void* dfilm, int64_t dfilm_ld, float* dgamma, float* dbeta, int bt,
                         int pixels, int channels, void* stream) {
  DFOT_REQUIRE(x && dy_bf && stats && gamma && beta && (dx || dx_bf) && dgamma && dbeta && (!dres || dres != dx) &&
                   (!film || (film_ld >= 2 * channels && film_ld % 4 == 0)),
               DFOT_ERR_ARG, "op_gn_silu_bwd6: null, aliased or misshaped argument");
  DFOT_REQUIRE(!dfilm || (dfilm_ld >= 2 * channels && dfilm_ld % 4 == 0), DFOT_ERR_ARG, "op_gn_silu_bwd6: bad dfilm row stride");
  hipStream_t s = (hipStream_t)stream;
  void* sums = nullptr;
  int rc = op_scratch(4, (size_t)bt * 64 * sizeof(float), &sums);
  if (rc) return rc;
  return gn_silu_backward(x, (const bf16*)dy_bf, stats, gamma, beta, (const bf16*)film, (float*)sums, dx, (bf16*)dfilm, dgamma, dbeta, bt, pixels,
                          channels, dres != nullptr, s, (long)dfilm_ld, dres, (bf16*)dx_bf, (long)film_ld);
}
// out [bt][n] fp32 = sum over the frame's `pixels` rows of src bf16 [bt * pixels][ld] (columns 0..n): per-frame column sums,
// deterministic (partial rows per pixel chunk + fixed-order sum); n % 8 == 0, ld % 8 == 0
int dfot_op_frame_sums_bf16(const void* src, int64_t ld, float* out, int bt, int pixels, int n, void* stream) {
  DFOT_REQUIRE(src && out && bt > 0 && pixels > 0, DFOT_ERR_ARG, "op_frame_sums_bf16: null argument");
  DFOT_REQUIRE(n % 8 == 0 && ld % 8 == 0 && ld >= n, DFOT_ERR_SHAPE, "op_frame_sums_bf16: n = %d and ld = %ld must be multiples of 8", n, (long)ld);
  hipStream_t s = (hipStream_t)stream;
  // pixel chunks per frame: ~2048 workgroups in all, at least 16 rows each
  const int xb = cdiv(n / 8, 256);
  int nz = 2048 / (xb * bt);
  if (nz > pixels / 16) nz = pixels / 16;
  if (nz < 1) nz = 1;
  float* part = nullptr;
  int rc = det_scratch(2, (size_t)nz * bt * n, &part);
  if (rc) return rc;
  hipLaunchKernelGGL(frame_sums_bf16_kernel, dim3(cdiv(n / 8, 256), bt, nz), dim3(256), 0, s, (const bf16*)src, (long)ld, part, pixels, n);
  DFOT_CHECK_HIP(hipGetLastError());
  return det_sum(part, (long)bt * n, nz, bt * n, out, false, s);
}
// x fp32 [n] = hi + lo with both parts in bf16 (n % 8 == 0): the operands of a three-product fp32-accurate GEMM on the bf16 matrix cores
int dfot_op_split_bf16(const float* x, void* hi, void* lo, int64_t n, void* stream) {
  DFOT_REQUIRE(x && hi && lo && n % 8 == 0, DFOT_ERR_ARG, "op_split_bf16: null argument or n not a multiple of 8");
  hipLaunchKernelGGL(split_bf16_kernel, dim3(cdiv((long)n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)hi, (bf16*)lo, (long)n / 8);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// C[i][j] (+)= sum_k A[i * sa_i + k * sa_k] * B[k * sb_k + j * sb_j], fp32, any strides (elements): the small dense products between
// weight-sized matrices (folded FiLM weights and their gradients) that must not round through bf16.  Not a throughput kernel.
int dfot_op_sgemm(const float* a, int64_t sa_i, int64_t sa_k, const float* b, int64_t sb_k, int64_t sb_j, float* c, int64_t ldc, int m, int n, int k,
                  int accumulate, void* stream) {
  DFOT_REQUIRE(a && b && c && m > 0 && n > 0 && k > 0 && ldc >= n, DFOT_ERR_ARG, "op_sgemm: bad argument");
  hipLaunchKernelGGL(sgemm_strided_kernel, dim3(cdiv(n, 64), cdiv(m, 64)), dim3(256), 0, (hipStream_t)stream, a, (long)sa_i, (long)sa_k, b, (long)sb_k,
                     (long)sb_j, c, (long)ldc, m, n, k, accumulate);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// w fp32 [Co][Ci][3][3] -> the forward kernel's layout [Co][tap][Ci] bf16 (dgrad = 0) or the data-gradient weights [Ci][tap'][Co] (dgrad = 1)
int dfot_op_pack_conv3(const float* w, void* out, int co, int ci, int dgrad, void* stream) {
  DFOT_REQUIRE(w && out, DFOT_ERR_ARG, "op_pack_conv3: null argument");
  if (!dgrad) return launch_pack_conv3(w, (bf16*)out, co, ci, (hipStream_t)stream);
  hipLaunchKernelGGL(pack_conv3_dgrad_kernel, dim3(cdiv((long)co * ci * 9, 256)), dim3(256), 0, (hipStream_t)stream, w, (bf16*)out, co, ci);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// y fp32 [BT,H,W,Cout] = conv3x3(a bf16 [BT,H,W,Cin], w packed [Cout][9*Cin]) + bias (+ resid)
int dfot_op_conv3x3_f32(const void* a, const void* w, const float* bias, const float* resid, float* y, int bt, int hh, int ww, int cin, int cout,
                        void* stream) {
  DFOT_REQUIRE(a && w && y, DFOT_ERR_ARG, "op_conv3x3_f32: null argument");
  static bf16* zeros = nullptr;
  if (!zeros) {
    DFOT_CHECK_HIP(hipMalloc(&zeros, 256));
    DFOT_CHECK_HIP(hipMemset(zeros, 0, 256));
  }
  GemmArgs g;
  g.zeros = zeros;
  g.A = (const bf16*)a; g.W = (const bf16*)w; g.M = bt * hh * ww; g.N = cout; g.K = 9 * cin; g.H = hh; g.Wd = ww; g.Cin = cin;
  g.bias = bias; g.resid = resid; g.out_f32 = y; g.ldo = cout;
  return launch_gemm(A_CONV3, E_F32, GEMM_AUTO, g, (hipStream_t)stream);
}
int dfot_op_pool2_bf16(const float* x, void* out, int bt, int h, int w, int c, void* stream) { return launch_pool2_bf16(x, (bf16*)out, bt, h, w, c, (hipStream_t)stream); }
int dfot_op_pool2_bwd(const float* dp, float* dx, int bt, int h, int w, int c, void* stream) {
  DFOT_REQUIRE(c % 4 == 0, DFOT_ERR_SHAPE, "op_pool2_bwd: channels %d must be a multiple of 4", c);
  const long total = (long)bt * h * w * (c / 4);
  hipLaunchKernelGGL(pool2_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dp, dx, total, h, w, c);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int dfot_op_sub_bf16(const float* a, const float* b, void* out, int64_t n, void* stream) { return launch_sub_bf16(a, b, (bf16*)out, (long)n, (hipStream_t)stream); }
int dfot_op_upsample_add(const float* t, const float* skip, float* out, int bt, int h, int w, int c, void* stream) {
  return launch_upsample_add(t, skip, out, bt, h, w, c, (hipStream_t)stream);
}
int dfot_op_upsample_bwd(const float* dy, float* ds, int bt, int h, int w, int c, void* stream) {
  DFOT_REQUIRE(c % 4 == 0, DFOT_ERR_SHAPE, "op_upsample_bwd: channels %d must be a multiple of 4", c);
  const long total = (long)bt * (h / 2) * (w / 2) * (c / 4);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, ds, total, h, w, c);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int dfot_op_axpy(float* a, const float* b, float alpha, int64_t n, void* stream) {
  const long n4 = ((((uintptr_t)a | (uintptr_t)b) & 15) == 0) ? (long)n / 4 : 0;
  if (n4) hipLaunchKernelGGL(axpy_kernel4, dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, n4);
  if (n - 4 * n4) hipLaunchKernelGGL(axpy_kernel, dim3(cdiv((long)n - 4 * n4, 256)), dim3(256), 0, (hipStream_t)stream, a + 4 * n4, b + 4 * n4, alpha, (long)n - 4 * n4);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int dfot_op_mul_cols(void* dst, int ldd, int dcol0, const void* mask, int64_t rows, int ncols, void* stream) {
  DFOT_REQUIRE(dst && mask && ncols % 8 == 0 && dcol0 % 8 == 0, DFOT_ERR_ARG, "op_mul_cols: bad argument");
  hipLaunchKernelGGL(mul_cols_kernel, dim3(cdiv((long)rows * (ncols / 8), 256)), dim3(256), 0, (hipStream_t)stream, (bf16*)dst, (long)ldd, dcol0,
                     (const bf16*)mask, (long)rows, ncols);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int dfot_op_emb_pyramid(const void* emb0, void* emb1, void* emb2, void* emb3, int bt, int r0, int e, void* stream) {
  return launch_emb_pyramid((const bf16*)emb0, (bf16*)emb1, (bf16*)emb2, (bf16*)emb3, bt, r0, e, (hipStream_t)stream);
}
int dfot_op_cond_repack(const float* cond, void* a, int bt, int res, int cdim, int kpad, void* stream) {
  return launch_cond_repack(cond, (bf16*)a, bt, res, cdim, kpad, (hipStream_t)stream);
}
int dfot_op_embed_input(const float* x, const float* w, const float* b, float* out, int bt, int res, int cin, int c0, void* stream) {
  return launch_embed_input(x, w, b, out, bt, res, cin, c0, (hipStream_t)stream);
}
// dW [C0][Cin][p][p] += , db [C0] += of the k = s = p patch embedding (dx0 fp32 [pix][C0], x fp32 [BT][Cin][R][R]); outputs zeroed here
int dfot_op_embed_input_wgrad(const float* dx0, const float* x, float* dw, float* db, int bt, int res, int cin, int c0, int ps, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)bt * (res / ps) * (res / ps);
  const int kdim = cin * ps * ps;
  DFOT_CHECK_HIP(hipMemsetAsync(dw, 0, (size_t)c0 * kdim * sizeof(float), s));
  DFOT_CHECK_HIP(hipMemsetAsync(db, 0, (size_t)c0 * sizeof(float), s));
  return launch_pe_wgrad(dx0, x, dw, db, cin, res, res, ps, c0, rows, s);
}
// dX [BT][Cin][R][R] = dx0 [pix][C0] . W [C0][Cin][p][p] of the k = s = p patch embedding: the gradient w.r.t. the backbone INPUT
// (reconstruction guidance differentiates the prediction w.r.t. x_t: discrete_diffusion.py:485-513).  One thread per patch.
int dfot_op_embed_input_dgrad(const float* dx0, const float* w, float* dx, int bt, int res, int cin, int c0, int ps, void* stream) {
  DFOT_REQUIRE(dx0 && w && dx, DFOT_ERR_ARG, "embed_input_dgrad: null pointer");
  DFOT_REQUIRE(ps == 2 && cin >= 1 && cin <= 4 && c0 % 4 == 0 && res % 2 == 0, DFOT_ERR_SHAPE,
               "embed_input_dgrad: patch %d, %d input channels, %d embedding channels unsupported", ps, cin, c0);
  const long patches = (long)bt * (res / 2) * (res / 2);
  hipLaunchKernelGGL(embed_input_dgrad_kernel, dim3(cdiv(patches, 256)), dim3(256), (size_t)c0 * cin * 4 * sizeof(float), (hipStream_t)stream,
                     dx0, w, dx, patches, res, cin, c0);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
int dfot_op_project_output(const float* x0, const float* w, const float* b, float* out, int bt, int res, int c0, int cout, void* stream) {
  return launch_project_output(x0, w, b, out, bt, res, c0, cout, (hipStream_t)stream);
}
int dfot_op_outgrad_gather(const float* dout, void* dpatch, int bt, int res, int cout, int ps, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long pix = (long)bt * (res / ps) * (res / ps);
  DFOT_REQUIRE(cout * ps * ps <= 64, DFOT_ERR_SHAPE, "outgrad_gather: more than 64 output values per pixel");
  DFOT_CHECK_HIP(hipMemsetAsync(dpatch, 0, (size_t)pix * 64 * sizeof(bf16), s));
  hipLaunchKernelGGL(outgrad_gather_kernel, dim3(cdiv(pix * cout * ps * ps, 256)), dim3(256), 0, s, dout, (bf16*)dpatch, pix, res, cout, ps);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
}  // extern "C"
