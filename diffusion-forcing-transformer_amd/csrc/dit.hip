// DiT3D backbone (Kinetics-600 path) on MI355X: weight packing, noise-level modulation table, forward orchestration.
// Mirrors the module tree / state-dict names of the reference
// (algorithms/dfot/backbones/dit/dit3d.py:146-192, dit/dit_base.py:150-196,391-419, dit/dit_blocks.py:49-128,378-542).
//
// Layout: tokens are channels-last rows [B*T*P][hidden]; the residual stream is fp32, GEMM operands bf16.
// Conditioning: c = MLP(sinusoidal(noise level)) depends on the integer level only, and every block consumes it through
// Linear(SiLU(c)).  finalize() therefore evaluates all modulations (shift|scale|gate per AdaLN-Zero, shift|scale for the
// final AdaLN) for every level with ONE GEMM into mod_table[level][...] (1000 x 99072 fp32 = 396 MB at DiT/XL; HBM is
// 288 GB), and forward never touches the embedding MLP or the 228 MB of modulation weights again.
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "dfot_hip.h"
#include "gemm.h"
#include "kernels.h"

namespace dfot {
namespace {

typedef __attribute__((ext_vector_type(4))) float float4v;

struct DitParam {
  std::string name;
  std::vector<int64_t> shape;
  std::function<int(const float*, hipStream_t)> load;
  bool loaded = false;
};

struct DitBlockW {
  bf16 *w_qkv = nullptr, *w_proj = nullptr, *w_fc1 = nullptr, *w_fc2 = nullptr;
  float *b_qkv = nullptr, *b_proj = nullptr, *b_fc1 = nullptr, *b_fc2 = nullptr;
  long mod1 = 0, mod2 = 0;  // column offsets of this block's (shift|scale|gate) triples inside a mod_table row
};

struct DitMatrixW {  // MatrixDiTBlock (temporal block of the factorized-matrix variant)
  bf16 *ut = nullptr, *vt = nullptr, *put = nullptr, *pvt = nullptr;  // qkv_u^T [E][P], qkv_v^T [3h][h], proj_u^T [P][E], proj_v^T [h][h]
  float *qkv_bias = nullptr, *proj_bias = nullptr;                      // [E][3h], [P][h]
  bf16 *w_fc1 = nullptr, *w_fc2 = nullptr;
  float *b_fc1 = nullptr, *b_fc2 = nullptr;
  long mod1 = 0, mod2 = 0;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- finalize-time kernels (run once per weight load) -------------------------------------------------------------
// get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): feat[level] = [cos(level*f_i) | sin(level*f_i)]
__global__ void tstep_features_kernel(const float* __restrict__ freqs, float* __restrict__ feat, int levels, int dim) {
  const int half = dim / 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)levels * dim) return;
  const int lv = (int)(i / dim), c = (int)(i % dim);
  const float a = __fmul_rn((float)lv, freqs[c < half ? c : c - half]);
  feat[i] = c < half ? cosf(a) : sinf(a);
}

// out[r][o] = act(b[o] + sum_k W[o][k] * in[r][k]); one wave per (o, r); ACT: 0 none, 1 SiLU
template <int ACT>
__global__ __launch_bounds__(256) void rows_linear_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ out,
                                                          bf16* __restrict__ out_silu_bf16, int kdim, int odim) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int o = blockIdx.x * 4 + wave;
  const long r = blockIdx.y;
  if (o >= odim) return;
  float acc = 0.f;
  for (int i = lane; i < kdim; i += 64) acc += w[(long)o * kdim + i] * in[r * kdim + i];
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = acc + b[o];
    if (ACT == 1) v = silu_f(v);
    out[r * odim + o] = v;
    if (out_silu_bf16) out_silu_bf16[r * odim + o] = f2bf(silu_f(v));
  }
}

// semb[(flag*lpad + level)][:] = bf16(SiLU(emb[level] + diff_table[flag]))  (DifferenceDiT3D: c = noise emb + diff emb)
__global__ void add_diff_silu_kernel(const float* __restrict__ emb, const float* __restrict__ diff_table, bf16* __restrict__ semb,
                                     int levels, int lpad, int hidden, int flags) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)flags * levels * hidden) return;
  const int c = (int)(i % hidden), lv = (int)((i / hidden) % levels), f = (int)(i / ((long)hidden * levels));
  semb[((long)f * lpad + lv) * hidden + c] = f2bf(silu_f(emb[(long)lv * hidden + c] + diff_table[(long)f * hidden + c]));
}

// dst[c][r] (bf16) = src[r][c] (fp32): weight packing of the matrix factors (stored (in, out) in the reference)
__global__ void pack_transpose_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int rows, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int r = (int)(i % rows), c = (int)(i / rows);
  dst[i] = f2bf(src[(long)r * cols + c]);
}

// table row of every (video, token): level + lpad * flag, flag = 1 for difference tokens (even positions of the interleaved
// merge, difference_dit3d.py:159-176 with diff_first=True)
__global__ void make_index_kernel(const int* __restrict__ levels, int* __restrict__ idx, int n, int tokens, int max_level, int lpad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int lv = levels[i];
  lv = lv < 0 ? 0 : (lv > max_level ? max_level : lv);
  idx[i] = lv + ((i % tokens) % 2 == 0 ? lpad : 0);
}

// per-frame 2-D transpose of a bf16 matrix: src [frames][R][C] -> dst [frames][C][R]; 64x64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int R, int C) {
  __shared__ bf16 tile[64][66];
  const long f = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const bf16* s = src + f * R * C;
  bf16* d = dst + f * R * C;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[ty * 16 + i][tx] = s[(long)(r0 + ty * 16 + i) * C + c0 + tx];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) d[(long)(c0 + ty * 16 + i) * R + r0 + tx] = tile[tx][ty * 16 + i];
}

// MatrixAttention core (dit_blocks.py:289-336, multi_token = False, no RoPE): every frame is one token whose q/k/v are
// (hn x hd) matrices; z [B*L*E][3h] holds (q|k|v) with columns (row head r, d) and rows (frame, col head c, n).
// One workgroup per (video, c, r): scores L x L = scale * <q_l, k_l'> over the hn*hd entries, softmax over l', o = P v.
// o [B*L*E][h] in the same row/column order.
// LT > 0: L == LT is a compile-time constant and every thread keeps the L x L partial scores of its slice of the hn*hd
// entries in registers (each q/k/v entry is read exactly once); LT == 0: generic L <= 32 (one pair of tokens per wave pass).
template <int LT>
__global__ __launch_bounds__(256) void matrix_attn_kernel(const bf16* __restrict__ z, bf16* __restrict__ o, int L, int E, int h,
                                                          int cc, int rr, float scale) {
  __shared__ float sc[32 * 32];
  __shared__ float part[4][LT > 0 ? LT * LT : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / (cc * rr), c = (blockIdx.x / rr) % cc, r = blockIdx.x % rr;
  const int hn = E / cc, hd = h / rr, ne = hn * hd / 4;
  const long ldz = 3L * h;
  auto zrow = [&](int l, int n) { return z + (((long)b * L + l) * E + c * hn + n) * ldz + r * hd; };
  if constexpr (LT > 0) {
    float acc[LT * LT];
#pragma unroll
    for (int i = 0; i < LT * LT; ++i) acc[i] = 0.f;
    for (int e = threadIdx.x; e < ne; e += 256) {
      const int n = (e * 4) / hd, d = (e * 4) % hd;
      float q[LT][4], k[LT][4];
#pragma unroll
      for (int l = 0; l < LT; ++l) {
        const bf16* p = zrow(l, n) + d;
        const bf16x4 q4 = *reinterpret_cast<const bf16x4*>(p);
        const bf16x4 k4 = *reinterpret_cast<const bf16x4*>(p + h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          q[l][j] = bf2f(q4[j]);
          k[l][j] = bf2f(k4[j]);
        }
      }
#pragma unroll
      for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int l2 = 0; l2 < LT; ++l2)
          acc[l * LT + l2] += (q[l][0] * k[l2][0] + q[l][1] * k[l2][1]) + (q[l][2] * k[l2][2] + q[l][3] * k[l2][3]);
    }
#pragma unroll
    for (int i = 0; i < LT * LT; ++i) {
      const float t = wave_sum(acc[i]);
      if (lane == 0) part[wave][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < LT * LT) sc[threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) * scale;
  } else {
    for (int pi = wave; pi < L * L; pi += 4) {
      const int l = pi / L, l2 = pi % L;
      float acc = 0.f;
      for (int e = lane; e < ne; e += 64) {
        const int n = (e * 4) / hd, d = (e * 4) % hd;
        const bf16x4 q4 = *reinterpret_cast<const bf16x4*>(zrow(l, n) + d);
        const bf16x4 k4 = *reinterpret_cast<const bf16x4*>(zrow(l2, n) + h + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += bf2f(q4[j]) * bf2f(k4[j]);
      }
      acc = wave_sum(acc);
      if (lane == 0) sc[pi] = acc * scale;
    }
  }
  __syncthreads();
  if (threadIdx.x < L) {
    float* row = sc + threadIdx.x * L;
    float mx = row[0];
    for (int j = 1; j < L; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) {
      row[j] = __expf(row[j] - mx);
      sum += row[j];
    }
    const float inv = 1.0f / sum;
    for (int j = 0; j < L; ++j) row[j] *= inv;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ne; e += 256) {
    const int n = (e * 4) / hd, d = (e * 4) % hd;
    if constexpr (LT > 0) {
      float v[LT][4];
#pragma unroll
      for (int l2 = 0; l2 < LT; ++l2) {
        const bf16x4 v4 = *reinterpret_cast<const bf16x4*>(zrow(l2, n) + 2 * h + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[l2][j] = bf2f(v4[j]);
      }
#pragma unroll
      for (int l = 0; l < LT; ++l) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l2 = 0; l2 < LT; ++l2) {
          const float pw = sc[l * LT + l2];
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] += pw * v[l2][j];
        }
        bf16x4 o4;
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = f2bf(a[j]);
        *reinterpret_cast<bf16x4*>(o + (((long)b * LT + l) * E + c * hn + n) * h + r * hd + d) = o4;
      }
    } else {
      for (int l = 0; l < L; ++l) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int l2 = 0; l2 < L; ++l2) {
          const bf16x4 v4 = *reinterpret_cast<const bf16x4*>(zrow(l2, n) + 2 * h + d);
          const float pw = sc[l * L + l2];
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] += pw * bf2f(v4[j]);
        }
        bf16x4 o4;
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = f2bf(a[j]);
        *reinterpret_cast<bf16x4*>(o + (((long)b * L + l) * E + c * hn + n) * h + r * hd + d) = o4;
      }
    }
  }
}

int launch_matrix_attn(const bf16* z, bf16* o, int batch, int L, int E, int h, int cc, int rr, float scale, hipStream_t s) {
  DFOT_REQUIRE(L > 0 && L <= 32, DFOT_ERR_SHAPE, "matrix attention: %d frame tokens (max 32)", L);
  const dim3 grid(batch * cc * rr), blk(256);
  switch (L) {
    case 2: hipLaunchKernelGGL(matrix_attn_kernel<2>, grid, blk, 0, s, z, o, L, E, h, cc, rr, scale); break;
    case 4: hipLaunchKernelGGL(matrix_attn_kernel<4>, grid, blk, 0, s, z, o, L, E, h, cc, rr, scale); break;
    case 6: hipLaunchKernelGGL(matrix_attn_kernel<6>, grid, blk, 0, s, z, o, L, E, h, cc, rr, scale); break;
    case 8: hipLaunchKernelGGL(matrix_attn_kernel<8>, grid, blk, 0, s, z, o, L, E, h, cc, rr, scale); break;
    case 10: hipLaunchKernelGGL(matrix_attn_kernel<10>, grid, blk, 0, s, z, o, L, E, h, cc, rr, scale); break;
    default: hipLaunchKernelGGL(matrix_attn_kernel<0>, grid, blk, 0, s, z, o, L, E, h, cc, rr, scale); break;
  }
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// ---- forward kernels --------------------------------------------------------------------------------------------
// PatchEmbed (Conv2d k = s = p): x [BT][C][H][W] fp32 -> tokens [BT*gh*gw][hidden] fp32.  8 tokens per workgroup so each
// weight row is fetched once per 8 tokens; thread = output channels t, t+256, ...
constexpr int PE_TOK = 8;
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, const float* __restrict__ pos,
                                                          float* __restrict__ out, int c, int hh, int ww, int ps, int hidden,
                                                          long rows) {
  extern __shared__ float patch[];  // [PE_TOK][kdim]
  const int gh = hh / ps, gw = ww / ps, kdim = c * ps * ps;
  const long row0 = (long)blockIdx.x * PE_TOK;
  for (int i = threadIdx.x; i < PE_TOK * kdim; i += 256) {
    const long row = row0 + i / kdim;
    const int kk = i % kdim;  // (ci, py, px) -- the Conv2d weight's own flattening
    float v = 0.f;
    if (row < rows) {
      const long bt = row / (gh * gw);
      const int g = (int)(row % (gh * gw)), gy = g / gw, gx = g % gw;
      const int ci = kk / (ps * ps), py = (kk / ps) % ps, px = kk % ps;
      v = x[((bt * c + ci) * hh + gy * ps + py) * ww + gx * ps + px];
    }
    patch[i] = v;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < hidden; o += 256) {
    float acc[PE_TOK];
    const float bias = b[o];
#pragma unroll
    for (int t = 0; t < PE_TOK; ++t) acc[t] = bias;
    for (int kk = 0; kk < kdim; ++kk) {
      const float wv = w[(long)o * kdim + kk];
#pragma unroll
      for (int t = 0; t < PE_TOK; ++t) acc[t] += wv * patch[t * kdim + kk];
    }
#pragma unroll
    for (int t = 0; t < PE_TOK; ++t)
      if (row0 + t < rows)  // pos: optional absolute positional embedding [P][hidden] of the token's patch (sinusoidal_2d)
        out[(row0 + t) * hidden + o] = acc[t] + (pos ? pos[((row0 + t) % (gh * gw)) * hidden + o] : 0.f);
  }
}

// AdaLN: m = LayerNorm(x) * (1 + scale) + shift, (shift|scale) = mod_table[level of the row's frame][off ...].
// One wave per token row; a lane owns CNT groups of VEC consecutive channels (hidden = 64 * VEC * CNT, compile-time so the
// row lives in registers without guards).  Writes m as fp32 (the block's residual base, in place) and bf16 (GEMM operand).
template <int VEC>
struct VecT;
template <>
struct VecT<1> { typedef float type; };
template <>
struct VecT<2> { typedef __attribute__((ext_vector_type(2))) float type; };
template <>
struct VecT<4> { typedef __attribute__((ext_vector_type(4))) float type; };
template <int VEC>
struct BVecT;
template <>
struct BVecT<1> { typedef bf16 type; };
template <>
struct BVecT<2> { typedef bf16x2 type; };
template <>
struct BVecT<4> { typedef bf16x4 type; };

template <int VEC>
__device__ __forceinline__ float vsum(const typename VecT<VEC>::type& v) {
  if constexpr (VEC == 1) return v;
  else if constexpr (VEC == 2) return v[0] + v[1];
  else return (v[0] + v[1]) + (v[2] + v[3]);
}
template <int VEC>
__device__ __forceinline__ float vdot(const typename VecT<VEC>::type& a, const typename VecT<VEC>::type& b) {
  if constexpr (VEC == 1) return a * b;
  else if constexpr (VEC == 2) return a[0] * b[0] + a[1] * b[1];
  else return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
}

// normalised + modulated row in registers: v[i] covers channels (i*64 + lane)*VEC ...
template <int VEC, int CNT>
__device__ __forceinline__ void ln_mod_row(const float* __restrict__ xr, const float* __restrict__ sh, int hidden, float eps,
                                           int lane, typename VecT<VEC>::type (&v)[CNT]) {
  typedef typename VecT<VEC>::type V;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    v[i] = *reinterpret_cast<const V*>(xr + (i * 64 + lane) * VEC);
    s += vsum<VEC>(v[i]);
  }
  const float mean = wave_sum(s) / (float)hidden;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    v[i] -= mean;
    q += vdot<VEC>(v[i], v[i]);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)hidden + eps);
  const float* sc = sh + hidden;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const V a = *reinterpret_cast<const V*>(sh + (i * 64 + lane) * VEC);
    const V g = *reinterpret_cast<const V*>(sc + (i * 64 + lane) * VEC);
    v[i] = v[i] * rstd * (1.0f + g) + a;
  }
}

template <int VEC, int CNT>
__global__ __launch_bounds__(256) void ln_mod_kernel(const float* xin, float* xout, bf16* __restrict__ obf,
                                                     const float* __restrict__ table, const int* __restrict__ levels,
                                                     long ldt, long off, int rows_per_frame, int rows, float eps,
                                                     int max_level) {
  typedef typename VecT<VEC>::type V;
  typedef typename BVecT<VEC>::type B;
  constexpr int hidden = 64 * VEC * CNT;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* xr = xout + (long)row * hidden;  // xout may alias xin (inference: in place); training keeps x for the backward
  int lv = levels[row / rows_per_frame];
  lv = lv < 0 ? 0 : (lv > max_level ? max_level : lv);
  V v[CNT];
  ln_mod_row<VEC, CNT>(xin + (long)row * hidden, table + (long)lv * ldt + off, hidden, eps, lane, v);
  bf16* orow = obf + (long)row * hidden;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    *reinterpret_cast<V*>(xr + (i * 64 + lane) * VEC) = v[i];
    B o;
    if constexpr (VEC == 1) {
      o = f2bf(v[i]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = f2bf(v[i][j]);
    }
    *reinterpret_cast<B*>(orow + (i * 64 + lane) * VEC) = o;
  }
}

// Final layer: AdaLN (shift|scale) -> Linear(hidden, p*p*C) -> unpatchify to [BT][C][H][W] (dit3d.py:129-144).
// One wave per token row; the weight (oc x hidden fp32, 72 KB at K600) is staged in LDS once per workgroup of FIN_ROWS rows.
constexpr int FIN_ROWS = 16;
template <int VEC, int CNT>
__global__ __launch_bounds__(256) void final_layer_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                          const int* __restrict__ levels, long ldt, long off,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ out, int rows_per_frame, int rows, float eps,
                                                          int max_level, int c, int hh, int ww, int ps) {
  typedef typename VecT<VEC>::type V;
  constexpr int hidden = 64 * VEC * CNT;
  extern __shared__ float wl[];  // [oc][hidden]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int oc = ps * ps * c;
  for (int i = threadIdx.x * 4; i < oc * hidden; i += 1024)
    *reinterpret_cast<float4v*>(wl + i) = *reinterpret_cast<const float4v*>(w + i);
  __syncthreads();
  const int gw = ww / ps;
  for (int rr = wave; rr < FIN_ROWS; rr += 4) {
    const int row = blockIdx.x * FIN_ROWS + rr;
    if (row >= rows) break;
    const int bt = row / rows_per_frame;
    int lv = levels[bt];
    lv = lv < 0 ? 0 : (lv > max_level ? max_level : lv);
    V v[CNT];
    ln_mod_row<VEC, CNT>(x + (long)row * hidden, table + (long)lv * ldt + off, hidden, eps, lane, v);
    const int g = row % rows_per_frame, gy = g / gw, gx = g % gw;
    for (int o = 0; o < oc; ++o) {  // o = (p, q, channel), channel fastest
      const float* wr = wl + o * hidden;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < CNT; ++i) acc += vdot<VEC>(v[i], *reinterpret_cast<const V*>(wr + (i * 64 + lane) * VEC));
      acc = wave_sum(acc);
      if (lane == 0) {
        const int ch = o % c, pq = o / c, py = pq / ps, px = pq % ps;
        out[(((long)bt * c + ch) * hh + gy * ps + py) * ww + gx * ps + px] = acc + b[o];
      }
    }
  }
}

// hidden = 64 * VEC * CNT with VEC = 2 when hidden % 128 == 0 (8-byte accesses), else 1
#define DIT_LN_DISPATCH(CALL)                                  \
  switch (hidden % 128 == 0 ? hidden / 128 : -(hidden / 64)) { \
    case 1: CALL(2, 1); break;                                 \
    case 2: CALL(2, 2); break;                                 \
    case 3: CALL(2, 3); break;                                 \
    case 4: CALL(2, 4); break;                                 \
    case 5: CALL(2, 5); break;                                 \
    case 6: CALL(2, 6); break;                                 \
    case 7: CALL(2, 7); break;                                 \
    case 8: CALL(2, 8); break;                                 \
    case 9: CALL(2, 9); break;                                 \
    case 10: CALL(2, 10); break;                               \
    case 12: CALL(2, 12); break;                               \
    case 16: CALL(2, 16); break;                               \
    case -1: CALL(1, 1); break;                                \
    case -3: CALL(1, 3); break;                                \
    case -5: CALL(1, 5); break;                                \
    case -7: CALL(1, 7); break;                                \
    case -9: CALL(1, 9); break;                                \
    default:                                                   \
      set_error("DiT: hidden size %d has no LayerNorm kernel instance", hidden); \
      return DFOT_ERR_SHAPE;                                   \
  }

int launch_ln_mod(const float* xin, float* x, bf16* obf, const float* table, const int* levels, long ldt, long off, int hidden, int rows_per_frame,
                  int rows, float eps, int max_level, hipStream_t s) {
#define CALL(V, C) \
  hipLaunchKernelGGL((ln_mod_kernel<V, C>), dim3(cdiv(rows, 4)), dim3(256), 0, s, xin, x, obf, table, levels, ldt, off, rows_per_frame, rows, eps, max_level)
  DIT_LN_DISPATCH(CALL)
#undef CALL
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

int launch_final_layer(const float* x, const float* table, const int* levels, long ldt, long off, const float* w, const float* b,
                       float* out, int hidden, int rows_per_frame, int rows, float eps, int max_level, int c, int hh, int ww, int ps,
                       hipStream_t s) {
  const int lds = ps * ps * c * hidden * (int)sizeof(float);
  DFOT_REQUIRE(lds <= 160 * 1024, DFOT_ERR_SHAPE, "final layer: weight (%d B) does not fit in LDS", lds);
#define CALL(V, C)                                                                                                          \
  {                                                                                                                         \
    auto kern = final_layer_kernel<V, C>;                                                                                   \
    static int lds_set = 0; /* once per instantiation and size (not inside a captured step) */                                \
    if (lds_set < lds) {                                                                                                    \
      DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
      lds_set = lds;                                                                                                        \
    }                                                                                                                       \
    hipLaunchKernelGGL(kern, dim3(cdiv(rows, FIN_ROWS)), dim3(256), lds, s, x, table, levels, ldt, off, w, b, out, rows_per_frame, \
                       rows, eps, max_level, c, hh, ww, ps);                                                                \
  }
  DIT_LN_DISPATCH(CALL)
#undef CALL
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

}  // namespace
}  // namespace dfot

using namespace dfot;

struct dfot_dit_s {
  dfot_dit_config cfg{};
  int gh = 0, gw = 0, P = 0, d = 0, dstride = 0, kpatch = 0, oc = 0;
  int lpad = 0;      // level count padded to the GEMM's row tile
  long ldt = 0;      // mod_table row stride (floats) = total modulation outputs
  std::vector<DitParam> params;
  std::map<std::string, int> index;
  std::vector<void*> owned, ws_owned;
  size_t ws_bytes = 0;
  // weights
  float *t_w1 = nullptr, *t_b1 = nullptr, *t_w2 = nullptr, *t_b2 = nullptr, *pe_w = nullptr, *pe_b = nullptr,
        *fin_w = nullptr, *fin_b = nullptr, *b_mod = nullptr;
  bf16* w_mod = nullptr;  // every modulation Linear stacked: [ldt][hidden]
  std::vector<DitBlockW> blocks;
  std::vector<DitMatrixW> tblocks;  // variant 1: one MatrixDiTBlock after every spatial block
  float *diff_table = nullptr, *pos2d = nullptr;
  long mod_final = 0;
  // derived at finalize
  float *freqs = nullptr, *feat = nullptr, *thid = nullptr, *emb = nullptr, *mod_table = nullptr, *rope_cs = nullptr;
  bf16* semb = nullptr;
  bool finalized = false;
  // workspace
  int max_batch = 0, last_rows = 0;
  float* X = nullptr;
  bf16 *A = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *hid = nullptr;
  bf16 *T1 = nullptr, *W1 = nullptr, *W2 = nullptr, *Z = nullptr;  // variant 1: transposes / left-factor products / qkv of frames
  int* idx = nullptr;                                              // variant 1: mod_table row per (video, token)
  int gemm_variant = GEMM_AUTO;
  bool time_attn = false;
  std::vector<hipEvent_t> ev_start, ev_stop;
  size_t ev_used = 0;
};

namespace dfot {
namespace {

template <typename T>
int dit_alloc(dfot_dit_s* h, T** out, size_t count, bool workspace = false) {
  void* p = nullptr;
  const size_t bytes = count * sizeof(T);
  DFOT_CHECK_HIP(hipMalloc(&p, bytes ? bytes : 16));
  (workspace ? h->ws_owned : h->owned).push_back(p);
  if (workspace) h->ws_bytes += bytes;
  *out = reinterpret_cast<T*>(p);
  return DFOT_OK;
}

void dit_add(dfot_dit_s* h, const std::string& name, std::vector<int64_t> shape, std::function<int(const float*, hipStream_t)> load) {
  h->index[name] = (int)h->params.size();
  h->params.push_back(DitParam{name, std::move(shape), std::move(load), false});
}

int dit_add_f32(dfot_dit_s* h, const std::string& name, std::vector<int64_t> shape, float** dst) {
  size_t n = 1;
  for (auto d : shape) n *= (size_t)d;
  int rc = dit_alloc(h, dst, n);
  if (rc) return rc;
  float* d = *dst;
  dit_add(h, name, shape, [d, n](const float* src, hipStream_t s) {
    DFOT_CHECK_HIP(hipMemcpyAsync(d, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return DFOT_OK;
  });
  return DFOT_OK;
}

// Linear weight [rows][k] fp32 -> bf16 at dst (row stride k)
int dit_add_bf16(dfot_dit_s* h, const std::string& name, int rows, int k, bf16* dst) {
  dit_add(h, name, {rows, k}, [=](const float* src, hipStream_t s) { return launch_pack_rows(src, dst, nullptr, rows, k, k, k, 0, s); });
  return DFOT_OK;
}

int dit_add_slice(dfot_dit_s* h, const std::string& name, int n, float* dst) {
  dit_add(h, name, {n}, [=](const float* src, hipStream_t s) {
    DFOT_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return DFOT_OK;
  });
  return DFOT_OK;
}

int dit_build(dfot_dit_s* h) {
  const dfot_dit_config& c = h->cfg;
  const int hd = c.hidden_size;
  h->gh = c.height / c.patch_size;
  h->gw = c.width / c.patch_size;
  h->P = h->gh * h->gw;
  h->d = hd / c.num_heads;
  h->dstride = attention_dstride(h->d);
  h->kpatch = c.in_channels * c.patch_size * c.patch_size;
  h->oc = h->kpatch;
  h->lpad = (c.timesteps + 255) / 256 * 256;
  const bool facmat = c.variant == 1;
  const int E = c.embed_col_dim, P = h->P;
  const int per_block = (c.mlp_hidden ? 6 * hd : 3 * hd) + (facmat ? (c.temporal_mlp_hidden ? 6 * hd : 3 * hd) : 0);
  h->ldt = (long)c.depth * per_block + 2 * hd;
  int rc = 0;
  // registration order == the reference module's state_dict order
  const std::string ne = "noise_level_pos_embedding.embedding";
  if ((rc = dit_add_f32(h, ne + ".linear_1.weight", {hd, c.noise_dim}, &h->t_w1))) return rc;
  if ((rc = dit_add_f32(h, ne + ".linear_1.bias", {hd}, &h->t_b1))) return rc;
  if ((rc = dit_add_f32(h, ne + ".linear_2.weight", {hd, hd}, &h->t_w2))) return rc;
  if ((rc = dit_add_f32(h, ne + ".linear_2.bias", {hd}, &h->t_b2))) return rc;
  if ((rc = dit_add_f32(h, "patch_embedder.proj.weight", {hd, c.in_channels, c.patch_size, c.patch_size}, &h->pe_w))) return rc;
  if ((rc = dit_add_f32(h, "patch_embedder.proj.bias", {hd}, &h->pe_b))) return rc;
  if (facmat && (rc = dit_add_f32(h, "diff_embedder.embedding_table.weight", {2, hd}, &h->diff_table))) return rc;
  if ((rc = dit_alloc(h, &h->w_mod, (size_t)h->ldt * hd))) return rc;
  if ((rc = dit_alloc(h, &h->b_mod, (size_t)h->ldt))) return rc;
  h->blocks.resize(c.depth);
  long off = 0;
  for (int i = 0; i < c.depth; ++i) {
    DitBlockW& w = h->blocks[i];
    const std::string pre = "dit_base.blocks." + std::to_string(i);
    w.mod1 = off;
    dit_add_bf16(h, pre + ".norm1.modulation.1.weight", 3 * hd, hd, h->w_mod + off * hd);
    dit_add_slice(h, pre + ".norm1.modulation.1.bias", 3 * hd, h->b_mod + off);
    off += 3 * hd;
    if ((rc = dit_alloc(h, &w.w_qkv, (size_t)3 * hd * hd))) return rc;
    dit_add_bf16(h, pre + ".attn.qkv.weight", 3 * hd, hd, w.w_qkv);
    if ((rc = dit_add_f32(h, pre + ".attn.qkv.bias", {3 * hd}, &w.b_qkv))) return rc;
    if ((rc = dit_alloc(h, &w.w_proj, (size_t)hd * hd))) return rc;
    dit_add_bf16(h, pre + ".attn.proj.weight", hd, hd, w.w_proj);
    if ((rc = dit_add_f32(h, pre + ".attn.proj.bias", {hd}, &w.b_proj))) return rc;
    if (c.mlp_hidden) {
      w.mod2 = off;
      dit_add_bf16(h, pre + ".norm2.modulation.1.weight", 3 * hd, hd, h->w_mod + off * hd);
      dit_add_slice(h, pre + ".norm2.modulation.1.bias", 3 * hd, h->b_mod + off);
      off += 3 * hd;
      if ((rc = dit_alloc(h, &w.w_fc1, (size_t)c.mlp_hidden * hd))) return rc;
      dit_add_bf16(h, pre + ".mlp.fc1.weight", c.mlp_hidden, hd, w.w_fc1);
      if ((rc = dit_add_f32(h, pre + ".mlp.fc1.bias", {c.mlp_hidden}, &w.b_fc1))) return rc;
      if ((rc = dit_alloc(h, &w.w_fc2, (size_t)hd * c.mlp_hidden))) return rc;
      dit_add_bf16(h, pre + ".mlp.fc2.weight", hd, c.mlp_hidden, w.w_fc2);
      if ((rc = dit_add_f32(h, pre + ".mlp.fc2.bias", {hd}, &w.b_fc2))) return rc;
    }
  }
  // dst[c][r] bf16 = src[r][c]: the matrix factors are stored (in, out); the GEMMs want [out][in]
  auto add_transposed = [&](const std::string& name, int rows, int cols, bf16* dst) {
    dit_add(h, name, {rows, cols}, [=](const float* src, hipStream_t s) {
      hipLaunchKernelGGL(pack_transpose_kernel, dim3(cdiv((long)rows * cols, 256)), dim3(256), 0, s, src, dst, rows, cols);
      DFOT_CHECK_HIP(hipGetLastError());
      return DFOT_OK;
    });
  };
  if (facmat) {
    h->tblocks.resize(c.depth);
    for (int i = 0; i < c.depth; ++i) {
      DitMatrixW& w = h->tblocks[i];
      const std::string pre = "dit_base.temporal_blocks." + std::to_string(i);
      w.mod1 = off;
      dit_add_bf16(h, pre + ".norm1.modulation.1.weight", 3 * hd, hd, h->w_mod + off * hd);
      dit_add_slice(h, pre + ".norm1.modulation.1.bias", 3 * hd, h->b_mod + off);
      off += 3 * hd;
      if ((rc = dit_alloc(h, &w.ut, (size_t)E * P)) || (rc = dit_alloc(h, &w.put, (size_t)P * E)) ||
          (rc = dit_alloc(h, &w.vt, (size_t)3 * hd * hd)) || (rc = dit_alloc(h, &w.pvt, (size_t)hd * hd)))
        return rc;
      add_transposed(pre + ".attn.qkv_u", P, E, w.ut);
      add_transposed(pre + ".attn.proj_u", E, P, w.put);
      add_transposed(pre + ".attn.qkv_v", hd, 3 * hd, w.vt);
      add_transposed(pre + ".attn.proj_v", hd, hd, w.pvt);
      if (c.use_bias) {
        if ((rc = dit_add_f32(h, pre + ".attn.qkv_bias", {E, 3 * hd}, &w.qkv_bias))) return rc;
        if ((rc = dit_add_f32(h, pre + ".attn.proj_bias", {P, hd}, &w.proj_bias))) return rc;
      }
      if (c.temporal_mlp_hidden) {
        const int th = c.temporal_mlp_hidden;
        w.mod2 = off;
        dit_add_bf16(h, pre + ".norm2.modulation.1.weight", 3 * hd, hd, h->w_mod + off * hd);
        dit_add_slice(h, pre + ".norm2.modulation.1.bias", 3 * hd, h->b_mod + off);
        off += 3 * hd;
        if ((rc = dit_alloc(h, &w.w_fc1, (size_t)th * hd))) return rc;
        dit_add_bf16(h, pre + ".mlp.fc1.weight", th, hd, w.w_fc1);
        if ((rc = dit_add_f32(h, pre + ".mlp.fc1.bias", {th}, &w.b_fc1))) return rc;
        if ((rc = dit_alloc(h, &w.w_fc2, (size_t)hd * th))) return rc;
        dit_add_bf16(h, pre + ".mlp.fc2.weight", hd, th, w.w_fc2);
        if ((rc = dit_add_f32(h, pre + ".mlp.fc2.bias", {hd}, &w.b_fc2))) return rc;
      }
    }
  }
  h->mod_final = off;
  dit_add_bf16(h, "dit_base.final_layer.norm_final.modulation.1.weight", 2 * hd, hd, h->w_mod + off * hd);
  dit_add_slice(h, "dit_base.final_layer.norm_final.modulation.1.bias", 2 * hd, h->b_mod + off);
  if ((rc = dit_add_f32(h, "dit_base.final_layer.linear.weight", {h->oc, hd}, &h->fin_w))) return rc;
  if ((rc = dit_add_f32(h, "dit_base.final_layer.linear.bias", {h->oc}, &h->fin_b))) return rc;

  // derived tables
  if ((rc = dit_alloc(h, &h->freqs, (size_t)c.noise_dim / 2))) return rc;
  if ((rc = dit_alloc(h, &h->feat, (size_t)h->lpad * c.noise_dim))) return rc;
  if ((rc = dit_alloc(h, &h->thid, (size_t)h->lpad * hd))) return rc;
  if ((rc = dit_alloc(h, &h->emb, (size_t)h->lpad * hd))) return rc;
  const int nflag = facmat ? 2 : 1;  // variant 1: the conditioning also depends on the token kind (difference / frame)
  if ((rc = dit_alloc(h, &h->semb, (size_t)nflag * h->lpad * hd))) return rc;
  if ((rc = dit_alloc(h, &h->mod_table, (size_t)nflag * h->lpad * h->ldt))) return rc;
  {
    const int half = c.noise_dim / 2;
    std::vector<float> f(half);
    for (int i = 0; i < half; ++i) f[i] = (float)std::exp(-std::log(10000.0) * (double)i / (double)half);
    DFOT_CHECK_HIP(hipMemcpy(h->freqs, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (facmat) {
    // sinusoidal_2d table [P][hidden] (get_nd_sincos_pos_embed, dit_base.py:527-572): np.meshgrid's default "xy" indexing
    // makes flattened entry m use position m % gh for the first half of the channels and m / gh for the second; each half
    // is [sin | cos] of pos * 10000^(-i/(half/2)), computed in float64 like numpy
    const int half = hd / 2, quarter = half / 2;
    std::vector<float> pe((size_t)P * hd);
    for (int m = 0; m < P; ++m) {
      const int pos[2] = {m % h->gh, m / h->gh};
      for (int a = 0; a < 2; ++a)
        for (int i = 0; i < quarter; ++i) {
          const double ang = (double)pos[a] / std::pow(10000.0, (double)i / (double)quarter);
          pe[(size_t)m * hd + a * half + i] = (float)std::sin(ang);
          pe[(size_t)m * hd + a * half + quarter + i] = (float)std::cos(ang);
        }
    }
    if ((rc = dit_alloc(h, &h->pos2d, pe.size()))) return rc;
    DFOT_CHECK_HIP(hipMemcpy(h->pos2d, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice));
  } else {  // RoPE-3D (cos, sin) table [Tmax*P][d/2][2]; axis split of the head dim as RotaryEmbedding3D (embeddings.py:251-277)
    const int half = h->d / 2, q = half / 3, rem = half % 3;
    int parts[3] = {q, q, q};
    if (rem == 1) parts[0] = q + 1;
    if (rem == 2) parts[1] = parts[2] = q + 1;
    const int n = c.max_tokens * h->P;
    std::vector<float> cs((size_t)n * half * 2);
    for (int tok = 0; tok < n; ++tok) {
      const int pos[3] = {tok / h->P, (tok / h->gw) % h->gh, tok % h->gw};
      int pair = 0;
      for (int ax = 0; ax < 3; ++ax) {
        const int dim = 2 * parts[ax];
        for (int j = 0; j < parts[ax]; ++j, ++pair) {
          const float inv = 1.0f / powf(c.rope_theta, (float)(2 * j) / (float)dim);
          const float ang = (float)pos[ax] * inv;
          cs[((size_t)tok * half + pair) * 2 + 0] = cosf(ang);
          cs[((size_t)tok * half + pair) * 2 + 1] = sinf(ang);
        }
      }
    }
    if ((rc = dit_alloc(h, &h->rope_cs, cs.size()))) return rc;
    DFOT_CHECK_HIP(hipMemcpy(h->rope_cs, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return DFOT_OK;
}

}  // namespace
}  // namespace dfot

extern "C" {

int dfot_dit_destroy(dfot_dit_t h) {
  if (!h) return DFOT_OK;
  for (void* p : h->owned) (void)hipFree(p);
  for (void* p : h->ws_owned) (void)hipFree(p);
  for (hipEvent_t e : h->ev_start) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_stop) (void)hipEventDestroy(e);
  delete h;
  return DFOT_OK;
}

int dfot_dit_create(const dfot_dit_config* cfg, dfot_dit_t* out) {
  DFOT_REQUIRE(cfg && out, DFOT_ERR_ARG, "dfot_dit_create: null argument");
  const dfot_dit_config& c = *cfg;
  DFOT_REQUIRE(c.hidden_size > 0 && c.hidden_size % 64 == 0 && c.hidden_size <= 2048, DFOT_ERR_SHAPE,
               "hidden_size %d must be a multiple of 64, <= 2048", c.hidden_size);
  DFOT_REQUIRE(c.num_heads > 0 && c.hidden_size % c.num_heads == 0, DFOT_ERR_SHAPE, "hidden_size %d not divisible by %d heads",
               c.hidden_size, c.num_heads);
  const int d = c.hidden_size / c.num_heads;
  DFOT_REQUIRE(d % 8 == 0 && d <= 128, DFOT_ERR_SHAPE, "head dim %d must be a multiple of 8, <= 128", d);
  DFOT_REQUIRE(c.patch_size > 0 && c.height % c.patch_size == 0 && c.width % c.patch_size == 0, DFOT_ERR_SHAPE,
               "x_shape %dx%d not divisible by patch %d", c.height, c.width, c.patch_size);
  DFOT_REQUIRE(c.in_channels > 0 && c.in_channels * c.patch_size * c.patch_size <= 256, DFOT_ERR_SHAPE, "patch vector too long");
  DFOT_REQUIRE(c.mlp_hidden >= 0 && c.mlp_hidden % 64 == 0, DFOT_ERR_SHAPE, "mlp_hidden %d must be a multiple of 64", c.mlp_hidden);
  DFOT_REQUIRE(c.noise_dim > 0 && c.noise_dim % 2 == 0 && c.timesteps > 0 && c.max_tokens > 0 && c.depth > 0, DFOT_ERR_SHAPE,
               "bad noise_dim / timesteps / max_tokens / depth");
  DFOT_REQUIRE(c.variant == 0 || c.variant == 1, DFOT_ERR_ARG, "variant %d unknown (0 = dit3d full/rope_3d, 1 = difference_dit3d factorized matrix)", c.variant);
  if (c.variant == 1) {
    const int P = (c.height / c.patch_size) * (c.width / c.patch_size);
    DFOT_REQUIRE(P % 128 == 0, DFOT_ERR_SHAPE, "factorized matrix variant: %d patches per frame must be a multiple of 128", P);
    DFOT_REQUIRE(c.embed_col_dim > 0 && c.embed_col_dim % 64 == 0, DFOT_ERR_SHAPE, "embed_col_dim %d must be a multiple of 64", c.embed_col_dim);
    DFOT_REQUIRE(c.num_col_heads > 0 && c.embed_col_dim % c.num_col_heads == 0 && c.num_row_heads > 0 &&
                     c.hidden_size % c.num_row_heads == 0 && (c.hidden_size / c.num_row_heads) % 4 == 0,
                 DFOT_ERR_SHAPE, "matrix attention heads (%d col, %d row) do not divide (%d, %d)", c.num_col_heads, c.num_row_heads,
                 c.embed_col_dim, c.hidden_size);
    DFOT_REQUIRE(c.max_tokens % 2 == 0 && c.max_tokens <= 32, DFOT_ERR_SHAPE, "max_tokens %d must be even (difference, frame pairs) and <= 32", c.max_tokens);
    DFOT_REQUIRE(c.temporal_mlp_hidden >= 0 && c.temporal_mlp_hidden % 64 == 0 && c.hidden_size % 4 == 0 && (c.hidden_size / 2) % 2 == 0,
                 DFOT_ERR_SHAPE, "temporal_mlp_hidden %d must be a multiple of 64", c.temporal_mlp_hidden);
  }
  auto* h = new dfot_dit_s();
  h->cfg = c;
  int rc = dit_build(h);
  if (rc) {
    dfot_dit_destroy(h);
    return rc;
  }
  *out = h;
  return DFOT_OK;
}

int dfot_dit_num_params(dfot_dit_t h) { return h ? (int)h->params.size() : 0; }
const char* dfot_dit_param_name(dfot_dit_t h, int i) {
  return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : nullptr;
}
int dfot_dit_param_shape(dfot_dit_t h, int i, int64_t shape[4], int* ndim) {
  DFOT_REQUIRE(h && shape && ndim && i >= 0 && i < (int)h->params.size(), DFOT_ERR_ARG, "param_shape: bad argument");
  *ndim = (int)h->params[i].shape.size();
  for (int k = 0; k < *ndim; ++k) shape[k] = h->params[i].shape[k];
  return DFOT_OK;
}

int dfot_dit_load_weight(dfot_dit_t h, const char* name, const float* data, const int64_t* shape, int ndim, void* stream) {
  DFOT_REQUIRE(h && name && data && shape, DFOT_ERR_ARG, "load_weight: null argument");
  auto it = h->index.find(name);
  DFOT_REQUIRE(it != h->index.end(), DFOT_ERR_NAME, "load_weight: unexpected key '%s'", name);
  DitParam& p = h->params[it->second];
  bool same = (int)p.shape.size() == ndim;
  for (int k = 0; same && k < ndim; ++k) same = p.shape[k] == shape[k];
  DFOT_REQUIRE(same, DFOT_ERR_SHAPE, "load_weight: size mismatch for '%s'", name);
  int rc = p.load(data, (hipStream_t)stream);
  if (rc) return rc;
  p.loaded = true;
  h->finalized = false;
  return DFOT_OK;
}

int dfot_dit_finalize(dfot_dit_t h, void* stream) {
  DFOT_REQUIRE(h, DFOT_ERR_ARG, "finalize: null handle");
  for (const DitParam& p : h->params) DFOT_REQUIRE(p.loaded, DFOT_ERR_STATE, "finalize: missing key '%s'", p.name.c_str());
  hipStream_t s = (hipStream_t)stream;
  const dfot_dit_config& c = h->cfg;
  const int hd = c.hidden_size, L = c.timesteps;
  // embedding of every level: features -> Linear -> SiLU -> Linear (emb) ; semb = bf16(SiLU(emb)) feeds every modulation
  hipLaunchKernelGGL(tstep_features_kernel, dim3(cdiv((long)L * c.noise_dim, 256)), dim3(256), 0, s, h->freqs, h->feat, L, c.noise_dim);
  DFOT_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(rows_linear_kernel<1>, dim3(cdiv(hd, 4), L), dim3(256), 0, s, h->feat, h->t_w1, h->t_b1, h->thid,
                     (bf16*)nullptr, c.noise_dim, hd);
  DFOT_CHECK_HIP(hipGetLastError());
  const int nflag = c.variant == 1 ? 2 : 1;
  DFOT_CHECK_HIP(hipMemsetAsync(h->semb, 0, (size_t)nflag * h->lpad * hd * sizeof(bf16), s));
  hipLaunchKernelGGL(rows_linear_kernel<0>, dim3(cdiv(hd, 4), L), dim3(256), 0, s, h->thid, h->t_w2, h->t_b2, h->emb,
                     c.variant == 1 ? (bf16*)nullptr : h->semb, hd, hd);
  DFOT_CHECK_HIP(hipGetLastError());
  if (c.variant == 1) {  // c = noise-level embedding + diff_embedder(token kind)  (difference_dit3d.py:199-204)
    hipLaunchKernelGGL(add_diff_silu_kernel, dim3(cdiv(2L * L * hd, 256)), dim3(256), 0, s, h->emb, h->diff_table, h->semb, L, h->lpad, hd, 2);
    DFOT_CHECK_HIP(hipGetLastError());
  }
  // mod_table[level][:] = W_mod * SiLU(emb[level]) + b_mod for every modulation of the model
  GemmArgs g;
  g.A = h->semb; g.lda = hd; g.W = h->w_mod; g.M = nflag * h->lpad; g.N = (int)h->ldt; g.K = hd;
  g.bias = h->b_mod; g.out_f32 = h->mod_table; g.ldo = h->ldt;
  int rc = launch_gemm(A_DENSE, E_F32, GEMM_AUTO, g, s);
  if (rc) return rc;
  DFOT_CHECK_HIP(hipStreamSynchronize(s));
  h->finalized = true;
  return DFOT_OK;
}

int dfot_dit_reserve(dfot_dit_t h, int max_batch) {
  DFOT_REQUIRE(h && max_batch > 0, DFOT_ERR_ARG, "reserve: bad argument");
  if (max_batch <= h->max_batch) return DFOT_OK;
  for (void* p : h->ws_owned) (void)hipFree(p);
  h->ws_owned.clear();
  h->ws_bytes = 0;
  h->max_batch = 0;
  const dfot_dit_config& c = h->cfg;
  const size_t rows = (size_t)max_batch * c.max_tokens * h->P;
  const size_t qkv = (size_t)max_batch * c.num_heads * c.max_tokens * h->P * h->dstride;
  int rc = 0;
  if ((rc = dit_alloc(h, &h->X, rows * c.hidden_size, true))) return rc;
  if ((rc = dit_alloc(h, &h->A, rows * c.hidden_size, true))) return rc;
  if ((rc = dit_alloc(h, &h->q, qkv, true)) || (rc = dit_alloc(h, &h->k, qkv, true)) || (rc = dit_alloc(h, &h->v, qkv, true))) return rc;
  // pad columns d..dstride of q/k/v are never written by the QKV epilogue: zero them once
  DFOT_CHECK_HIP(hipMemset(h->q, 0, qkv * sizeof(bf16)));
  DFOT_CHECK_HIP(hipMemset(h->k, 0, qkv * sizeof(bf16)));
  DFOT_CHECK_HIP(hipMemset(h->v, 0, qkv * sizeof(bf16)));
  const int hid_cols = c.variant == 1 && c.temporal_mlp_hidden > c.mlp_hidden ? c.temporal_mlp_hidden : c.mlp_hidden;
  if (hid_cols && (rc = dit_alloc(h, &h->hid, rows * hid_cols, true))) return rc;
  if (c.variant == 1) {
    const size_t frames = (size_t)max_batch * c.max_tokens;
    if ((rc = dit_alloc(h, &h->T1, rows * c.hidden_size, true))) return rc;
    if ((rc = dit_alloc(h, &h->W1, frames * c.embed_col_dim * c.hidden_size, true))) return rc;
    if ((rc = dit_alloc(h, &h->W2, frames * c.embed_col_dim * c.hidden_size, true))) return rc;
    if ((rc = dit_alloc(h, &h->Z, frames * c.embed_col_dim * 3 * c.hidden_size, true))) return rc;
    if ((rc = dit_alloc(h, &h->idx, frames, true))) return rc;
  }
  h->max_batch = max_batch;
  return DFOT_OK;
}

size_t dfot_dit_workspace_bytes(dfot_dit_t h) { return h ? h->ws_bytes : 0; }

int dfot_dit_set_option(dfot_dit_t h, const char* key, int value) {
  DFOT_REQUIRE(h && key, DFOT_ERR_ARG, "set_option: null argument");
  if (!strcmp(key, "gemm_variant")) {
    h->gemm_variant = value;
    return DFOT_OK;
  }
  if (!strcmp(key, "time_attn")) {
    h->time_attn = value > 0;
    h->ev_used = 0;
    while ((int)h->ev_start.size() < value) {
      hipEvent_t a, b;
      DFOT_CHECK_HIP(hipEventCreate(&a));
      DFOT_CHECK_HIP(hipEventCreate(&b));
      h->ev_start.push_back(a);
      h->ev_stop.push_back(b);
    }
    return DFOT_OK;
  }
  set_error("set_option: unknown key '%s'", key);
  return DFOT_ERR_NAME;
}

int dfot_dit_attn_timing(dfot_dit_t h, double* total_ms, int64_t* launches) {
  DFOT_REQUIRE(h && total_ms && launches, DFOT_ERR_ARG, "attn_timing: null argument");
  double tot = 0;
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

int dfot_dit_forward(dfot_dit_t h, const float* x, const int32_t* noise_levels, float* out, int batch, int tokens, void* stream) {
  DFOT_REQUIRE(h && x && noise_levels && out, DFOT_ERR_ARG, "forward: null argument");
  DFOT_REQUIRE(h->finalized, DFOT_ERR_STATE, "forward: weights not finalized");
  DFOT_REQUIRE(batch > 0 && batch <= h->max_batch, DFOT_ERR_STATE, "forward: batch %d exceeds the reserved %d", batch, h->max_batch);
  const dfot_dit_config& c = h->cfg;
  DFOT_REQUIRE(tokens > 0 && tokens <= c.max_tokens, DFOT_ERR_SHAPE, "forward: %d tokens, max_tokens is %d", tokens, c.max_tokens);
  const int n = tokens * h->P, hd = c.hidden_size;
  DFOT_REQUIRE(n % 128 == 0, DFOT_ERR_SHAPE, "forward: sequence length %d (tokens x patches) must be a multiple of 128", n);
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)batch * n;
  const bool facmat = c.variant == 1;
  const int frames = batch * tokens, P = h->P, E = c.embed_col_dim;
  DFOT_REQUIRE(!facmat || tokens % 2 == 0, DFOT_ERR_SHAPE, "forward: %d tokens; the difference model takes (difference, frame) pairs", tokens);
  int max_level = c.timesteps - 1;
  const int* lvl = noise_levels;
  int rc = 0;
  if (facmat) {  // table row = level + lpad * (token kind)
    hipLaunchKernelGGL(make_index_kernel, dim3(cdiv(frames, 256)), dim3(256), 0, s, noise_levels, h->idx, frames, tokens, max_level, h->lpad);
    DFOT_CHECK_HIP(hipGetLastError());
    lvl = h->idx;
    max_level = 2 * h->lpad - 1;
  }
  hipLaunchKernelGGL(patch_embed_kernel, dim3(cdiv(rows, PE_TOK)), dim3(256), PE_TOK * h->kpatch * sizeof(float), s, x, h->pe_w,
                     h->pe_b, h->pos2d, h->X, c.in_channels, c.height, c.width, c.patch_size, hd, rows);
  DFOT_CHECK_HIP(hipGetLastError());
  const float qscale = 1.4426950408889634f / sqrtf((float)h->d);  // attention works in the exp2 domain
  auto ln_mod = [&](long off) -> int {
    return launch_ln_mod(h->X, h->X, h->A, h->mod_table, lvl, h->ldt, off, hd, P, (int)rows, c.eps, max_level, s);
  };
  auto gated = [&](const bf16* a, int kdim, const bf16* w, const float* bias, int bias_rows, long gate_off) -> int {
    GemmArgs g;  // X <- X + gate * (a W^T + bias), in place
    g.A = a; g.lda = kdim; g.W = w; g.M = (int)rows; g.N = hd; g.K = kdim; g.bias = bias; g.bias_rows = bias_rows;
    g.out_f32 = h->X; g.ldo = hd; g.resid = h->X;
    g.gate = h->mod_table + gate_off; g.gate_index = lvl; g.ldg = h->ldt; g.gate_rows = P;
    return launch_gemm(A_DENSE, E_F32, h->gemm_variant, g, s);
  };
  auto mlp = [&](long mod2, int hidden_cols, const bf16* w1, const float* b1, const bf16* w2, const float* b2) -> int {
    int r2 = ln_mod(mod2);
    if (r2) return r2;
    GemmArgs g;
    g.A = h->A; g.lda = hd; g.W = w1; g.M = (int)rows; g.N = hidden_cols; g.K = hd; g.bias = b1;
    g.out_bf16 = h->hid; g.ldo = hidden_cols; g.act = 1;
    if ((r2 = launch_gemm(A_DENSE, E_BF16, h->gemm_variant, g, s))) return r2;
    return gated(h->hid, hidden_cols, w2, b2, 0, mod2 + 2 * hd);
  };
  auto transpose = [&](const bf16* src, bf16* dst, int R, int C) -> int {  // per frame [R][C] -> [C][R]
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(C / 64, R / 64, frames), dim3(256), 0, s, src, dst, R, C);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  };
  // attention sequences: the whole video (variant 0) or one frame (variant 1, per-frame spatial blocks without RoPE)
  const int seq = facmat ? P : n, nseq = facmat ? frames : batch;
  for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
    const DitBlockW& w = h->blocks[bi];
    if ((rc = ln_mod(w.mod1))) return rc;
    {
      GemmArgs g;
      g.A = h->A; g.lda = hd; g.W = w.w_qkv; g.M = (int)rows; g.N = 3 * hd; g.K = hd; g.bias = w.b_qkv;
      g.q = h->q; g.k = h->k; g.v = h->v; g.rope_cs = facmat ? nullptr : h->rope_cs; g.heads = c.num_heads; g.d = h->d;
      g.dstride = h->dstride; g.ntok = seq; g.qscale = qscale;
      if ((rc = launch_gemm(A_DENSE, E_QKV_DIT, h->gemm_variant, g, s))) return rc;
    }
    const bool timed = h->time_attn && h->ev_used < h->ev_start.size();
    if (timed) DFOT_CHECK_HIP(hipEventRecord(h->ev_start[h->ev_used], s));
    if ((rc = launch_attention_padded(h->q, h->k, h->v, h->A, hd, nseq, c.num_heads, seq, h->d, s))) return rc;
    if (timed) DFOT_CHECK_HIP(hipEventRecord(h->ev_stop[h->ev_used++], s));
    if ((rc = gated(h->A, hd, w.w_proj, w.b_proj, 0, w.mod1 + 2 * hd))) return rc;
    if (c.mlp_hidden && (rc = mlp(w.mod2, c.mlp_hidden, w.w_fc1, w.b_fc1, w.w_fc2, w.b_fc2))) return rc;
    if (!facmat) continue;

    // ---- MatrixDiTBlock: every frame is one token; qkv = U^T m V + bias, o = softmax(q k^T) v, out = U'^T o V' + bias' ----
    const DitMatrixW& t = h->tblocks[bi];
    if ((rc = ln_mod(t.mod1))) return rc;
    if ((rc = transpose(h->A, h->T1, P, hd))) return rc;  // m^T per frame: [hd][P]
    {
      GemmArgs g;  // left factor: w[frame][e][d] = sum_p U[p][e] m[frame][p][d]   (rows (frame, d), K = p, transposed store)
      g.A = h->T1; g.lda = P; g.W = t.ut; g.M = frames * hd; g.N = E; g.K = P; g.out_bf16 = h->W1; g.ldo = E; g.tr_rows = hd;
      if ((rc = launch_gemm(A_DENSE, E_BF16, h->gemm_variant, g, s))) return rc;
    }
    {
      GemmArgs g;  // right factor + bias[e][k]
      g.A = h->W1; g.lda = hd; g.W = t.vt; g.M = frames * E; g.N = 3 * hd; g.K = hd; g.bias = t.qkv_bias; g.bias_rows = t.qkv_bias ? E : 0;
      g.out_bf16 = h->Z; g.ldo = 3 * hd;
      if ((rc = launch_gemm(A_DENSE, E_BF16, h->gemm_variant, g, s))) return rc;
    }
    {
      const int hn = E / c.num_col_heads, hdr = hd / c.num_row_heads;
      if ((rc = launch_matrix_attn(h->Z, h->W1, batch, tokens, E, hd, c.num_col_heads, c.num_row_heads,
                                   1.0f / sqrtf((float)hn * (float)hdr), s)))
        return rc;
    }
    if ((rc = transpose(h->W1, h->W2, E, hd))) return rc;  // o^T per frame: [hd][E]
    {
      GemmArgs g;  // left factor of the projection: s[frame][p][d] = sum_e U'[e][p] o[frame][e][d]
      g.A = h->W2; g.lda = E; g.W = t.put; g.M = frames * hd; g.N = P; g.K = E; g.out_bf16 = h->T1; g.ldo = P; g.tr_rows = hd;
      if ((rc = launch_gemm(A_DENSE, E_BF16, h->gemm_variant, g, s))) return rc;
    }
    if ((rc = gated(h->T1, hd, t.pvt, t.proj_bias, t.proj_bias ? P : 0, t.mod1 + 2 * hd))) return rc;
    if (c.temporal_mlp_hidden && (rc = mlp(t.mod2, c.temporal_mlp_hidden, t.w_fc1, t.b_fc1, t.w_fc2, t.b_fc2))) return rc;
  }
  h->last_rows = (int)rows;
  return launch_final_layer(h->X, h->mod_table, lvl, h->ldt, h->mod_final, h->fin_w, h->fin_b, out, hd, P, (int)rows, c.eps,
                            max_level, c.in_channels, c.height, c.width, c.patch_size, s);
}

int dfot_dit_read_tap(dfot_dit_t h, const char* name, float* out, size_t capacity, void* stream) {
  DFOT_REQUIRE(h && name && out, DFOT_ERR_ARG, "read_tap: null argument");
  hipStream_t s = (hipStream_t)stream;
  const float* src = nullptr;
  size_t need = 0;
  if (!strcmp(name, "emb")) {
    DFOT_REQUIRE(h->finalized, DFOT_ERR_STATE, "read_tap: weights not finalized");
    src = h->emb;
    need = (size_t)h->cfg.timesteps * h->cfg.hidden_size;
  } else if (!strcmp(name, "stream")) {
    DFOT_REQUIRE(h->last_rows > 0, DFOT_ERR_STATE, "read_tap: no forward has run");
    src = h->X;
    need = (size_t)h->last_rows * h->cfg.hidden_size;
  } else {
    set_error("read_tap: unknown tap '%s'", name);
    return DFOT_ERR_NAME;
  }
  DFOT_REQUIRE(capacity >= need, DFOT_ERR_SHAPE, "read_tap: need %zu floats, got %zu", need, capacity);
  DFOT_CHECK_HIP(hipMemcpyAsync(out, src, need * sizeof(float), hipMemcpyDeviceToDevice, s));
  return DFOT_OK;
}

// test entry of the training path: forward (o, lse) + backward of one attention call; temporaries are allocated here
int dfot_op_attention_bwd(const void* q, const void* k, const void* v, const void* d_o, void* o, int ldo, void* dq, void* dk, void* dv,
                          int batch, int heads, int n, int d, void* stream) {
  DFOT_REQUIRE(q && k && v && d_o && o && dq && dk && dv, DFOT_ERR_ARG, "attention_bwd: null argument");
  hipStream_t s = (hipStream_t)stream;
  const size_t bhn = (size_t)batch * heads * n;
  float *lse = nullptr, *delta = nullptr;
  DFOT_CHECK_HIP(hipMalloc(&lse, bhn * sizeof(float)));
  DFOT_CHECK_HIP(hipMalloc(&delta, bhn * sizeof(float)));
  int rc = launch_attention_padded((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, ldo, batch, heads, n, d, s, lse);
  if (!rc) rc = launch_attention_bwd_delta((const bf16*)o, (const bf16*)d_o, ldo, delta, batch, heads, n, d, s);
  if (!rc) rc = launch_attention_bwd((const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)d_o, ldo, lse, delta, (bf16*)dq, (bf16*)dk,
                                     (bf16*)dv, batch, heads, n, d, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(lse); (void)hipFree(delta);
  return rc;
}

int dfot_op_attention_padded(const void* q, const void* k, const void* v, void* o, int ldo, int batch, int heads, int n, int d,
                             void* stream) {
  return launch_attention_padded((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, ldo, batch, heads, n, d, (hipStream_t)stream);
}

}  // extern "C"

#include "dit_train.inl"
#include "uvit_train.inl"
