// Flash-attention backward for the DFoT transformer blocks (gfx950), training path.
//
// Same conventions as the forward (attention.hip): q/k/v are [B][heads][N][D] bf16 rows (D = 64 or 128, logical head dim
// d <= D, pad columns zero), q is pre-multiplied by log2(e)/sqrt(d) so scores live in the exp2 domain, and the forward
// leaves L2[q] = m + log2(sum) per query.  With P = exp2(S' - L2) (= softmax), delta[q] = sum_c dO[q][c] O[q][c] and
// dS = P o (dP - delta):
//      dV = P^T dO            dP = dO V^T
//      dQ = sq * dS K         dK = sk * dS^T Q'        (sq = 1/sqrt(d) for the unscaled q, sk = ln 2 undoes q's log2 e)
// Two kernels, each a structural clone of the forward kernel, so every product keeps its reduction index on MFMA registers
// and its output index on the lanes -- no atomics and no transposes of P / dS through LDS, at the price of computing S
// and dP in both orientations (8 instead of 5 GEMMs):
//   * attn_bwd_dq : a wave owns 32 queries (q, dO fragments in registers), streams K/V tiles through LDS:
//        S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T   (K^T by ds_read_tr16_b64)
//   * attn_bwd_dkv: a wave owns 32 keys (k, v fragments in registers), streams Q/dO tiles (+ L2, delta) through LDS:
//        S = Q K^T, dP = dO V^T, P, dS, dV^T += dO^T P, dK^T += Q^T dS                 (dO^T, Q^T by ds_read_tr16_b64)
// LDS rows are padded by 16 bytes instead of swizzled (one layout serves row reads and transposed reads).
// v_mfma_f32_32x32x16_bf16 operand layout as in attention.hip: A lane (l32, h) = A[l32][8h..8h+7], B lane = B[8h..8h+7][l32],
// C lane = C[8g + 4h + j][l32] in register 4g + j.
#include "common.h"
#include "dfot_hip.h"
#include "kernels.h"

namespace dfot {
namespace {

template <int D>
struct BwdCfg {
  static constexpr int TR = 64;               // rows per streamed tile
  static constexpr int ROWB = D * 2 + 16;     // padded LDS row (bytes)
  static constexpr int TILE = TR * ROWB;
  static constexpr int CH = D / 8;            // 16-byte chunks per row
  static constexpr int PER_THREAD = TR * CH / 256;
};

// A fragment of X^T for rows c0..c0+31 (feature index) and the 16 permuted tile rows of step (kt2, s): the order matches
// the B fragment built from an accumulator (tile row = kt2*32 + 16s + 8(j>>2) + 4h + (j&3))
template <int D>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int c0, int kt2, int s, int lane) {
  using C = BwdCfg<D>;
  const int lh = lane >> 5;
  const int kb = kt2 * 32 + 16 * s + 4 * lh;
  const int q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int col = c0 + 16 * ((lane >> 4) & 1) + 4 * p4;
  const char* a0 = tile + (kb + q4) * C::ROWB + col * 2;
  const char* a1 = tile + (kb + 8 + q4) * C::ROWB + col * 2;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a1));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int D>
__device__ __forceinline__ void tile_load(const bf16* __restrict__ src, bf16x8 (&r)[BwdCfg<D>::PER_THREAD], int tid) {
  using C = BwdCfg<D>;
#pragma unroll
  for (int i = 0; i < C::PER_THREAD; ++i) {
    const int c = tid + i * 256;
    r[i] = *reinterpret_cast<const bf16x8*>(src + (long)(c / C::CH) * D + (c % C::CH) * 8);
  }
}
template <int D>
__device__ __forceinline__ void tile_store(char* dst, const bf16x8 (&r)[BwdCfg<D>::PER_THREAD], int tid) {
  using C = BwdCfg<D>;
#pragma unroll
  for (int i = 0; i < C::PER_THREAD; ++i) {
    const int c = tid + i * 256;
    *reinterpret_cast<bf16x8*>(dst + (c / C::CH) * C::ROWB + (c % C::CH) * 16) = r[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// DC / DW: head-dim columns actually multiplied when the logical head dim is smaller than the row stride D (pad columns are
// zero): DC (multiple of 16) where the head dim is contracted (S, dP), DW (multiple of 32) where it is the output (dQ, dK, dV)
template <int D, int DC = D, int DW = D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                          const bf16* __restrict__ V, const bf16* __restrict__ dO, long ldo,
                                                          const float* __restrict__ L2, const float* __restrict__ delta,
                                                          bf16* __restrict__ dQ, int N, int heads, int d, float sq) {
  using C = BwdCfg<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int qtiles = N / 128;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lin / qtiles;
  const long base = (long)bh * N * D;
  const int q0 = (lin % qtiles) * 128 + wave * 32;
  const bf16* Kb = K + base;
  const bf16* Vb = V + base;

  bf16x8 qf[DC / 16], dof[DC / 16];
  // dO is read from the compact activation layout [B*N][ldo] (head hd at column hd*d); columns >= d belong to the next head: zero
  const bf16* dorow = dO + ((long)(bh / heads) * N + q0 + lq) * ldo + (long)(bh % heads) * d;
#pragma unroll
  for (int ks = 0; ks < DC / 16; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(Q + base + (long)(q0 + lq) * D + ks * 16 + lh * 8);
    const int col = ks * 16 + lh * 8;
    const bf16 z = f2bf(0.f);
    dof[ks] = col < d ? *reinterpret_cast<const bf16x8*>(dorow + col) : bf16x8{z, z, z, z, z, z, z, z};
  }
  const float l2 = L2[(long)bh * N + q0 + lq], dl = delta[(long)bh * N + q0 + lq];

  f32x16 acc[DW / 32];
#pragma unroll
  for (int i = 0; i < DW / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  bf16x8 rk[C::PER_THREAD], rv[C::PER_THREAD];
  tile_load<D>(Kb, rk, tid);
  tile_load<D>(Vb, rv, tid);
  tile_store<D>(smem, rk, tid);
  tile_store<D>(smem + C::TILE, rv, tid);
  __syncthreads();
  const int nt = N / C::TR;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sk = smem + cur * 2 * C::TILE;
    const char* sv = sk + C::TILE;
    if (t + 1 < nt) {
      tile_load<D>(Kb + (long)(t + 1) * C::TR * D, rk, tid);
      tile_load<D>(Vb + (long)(t + 1) * C::TR * D, rv, tid);
    }
    bf16x8 dsf[2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = -l2; pacc[r] = -dl; }
      const int row = kt2 * 32 + lq;
#pragma unroll
      for (int ks = 0; ks < DC / 16; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + row * C::ROWB + (ks * 2 + lh) * 16);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc, 0, 0, 0);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sv + row * C::ROWB + (ks * 2 + lh) * 16);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], pacc, 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) dsf[kt2][s][j] = f2bf(__builtin_amdgcn_exp2f(sacc[8 * s + j]) * pacc[8 * s + j]);
    }
    // dQ^T[c][q] += K^T[c][key] dS^T[key][q]
#pragma unroll
    for (int dvt = 0; dvt < DW / 32; ++dvt)
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<D>(sk, dvt * 32, kt2, s, lane), dsf[kt2][s], acc[dvt], 0, 0, 0);
    if (t + 1 < nt) {
      char* nk = smem + (cur ^ 1) * 2 * C::TILE;
      tile_store<D>(nk, rk, tid);
      tile_store<D>(nk + C::TILE, rv, tid);
    }
    __syncthreads();
    cur ^= 1;
  }
  bf16* orow = dQ + base + (long)(q0 + lq) * D;
  if constexpr (DW < D) {  // pad columns beyond DW: zero (the caller's buffers are re-used across blocks)
#pragma unroll
    for (int c = DW + 4 * lh; c < D; c += 8) *reinterpret_cast<bf16x4*>(orow + c) = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
  }
#pragma unroll
  for (int dvt = 0; dvt < DW / 32; ++dvt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = f2bf(acc[dvt][4 * g4 + j] * sq);
      *reinterpret_cast<bf16x4*>(orow + dvt * 32 + 8 * g4 + 4 * lh) = o4;
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <int D, int DC = D, int DW = D>
__global__ __launch_bounds__(256, (DC <= 96 ? 2 : 1)) void attn_bwd_dkv_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                           const bf16* __restrict__ V, const bf16* __restrict__ dO, long ldo,
                                                           const float* __restrict__ L2, const float* __restrict__ delta,
                                                           bf16* __restrict__ dK, bf16* __restrict__ dV, int N, int heads, int d,
                                                           float sk_scale) {
  using C = BwdCfg<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int ktiles = N / 128;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lin / ktiles;
  const long base = (long)bh * N * D;
  const int k0 = (lin % ktiles) * 128 + wave * 32;
  const bf16* Qb = Q + base;
  const bf16* Ob = dO + (long)(bh / heads) * N * ldo + (long)(bh % heads) * d;  // compact [B*N][ldo], this head's columns
  const float* Lb = L2 + (long)bh * N;
  const float* Db = delta + (long)bh * N;

  bf16x8 kf[DC / 16], vf[DC / 16];
#pragma unroll
  for (int ks = 0; ks < DC / 16; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(K + base + (long)(k0 + lq) * D + ks * 16 + lh * 8);
    vf[ks] = *reinterpret_cast<const bf16x8*>(V + base + (long)(k0 + lq) * D + ks * 16 + lh * 8);
  }
  f32x16 dka[DW / 32], dva[DW / 32];
#pragma unroll
  for (int i = 0; i < DW / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dka[i][r] = 0.f; dva[i][r] = 0.f; }

  // LDS: [stage][Q tile | dO tile | L2[64] | delta[64]]
  constexpr int STAGE = 2 * C::TILE + 2 * C::TR * 4;
  bf16x8 rq[C::PER_THREAD], ro[C::PER_THREAD];
  float rs = 0.f;
  auto load = [&](int t) {
    tile_load<D>(Qb + (long)t * C::TR * D, rq, tid);
#pragma unroll
    for (int i = 0; i < C::PER_THREAD; ++i) {  // dO tile from the compact layout; columns >= d (next head / padding) read as zero
      const int c = tid + i * 256, row = c / C::CH, col = (c % C::CH) * 8;
      const bf16 z = f2bf(0.f);
      ro[i] = col < d ? *reinterpret_cast<const bf16x8*>(Ob + (long)(t * C::TR + row) * ldo + col) : bf16x8{z, z, z, z, z, z, z, z};
    }
    if (tid < 2 * C::TR) rs = -(tid < C::TR ? Lb[t * C::TR + tid] : Db[t * C::TR + tid - C::TR]);  // negated: read straight into the accumulators
  };
  auto store = [&](int stage) {
    char* b = smem + stage * STAGE;
    tile_store<D>(b, rq, tid);
    tile_store<D>(b + C::TILE, ro, tid);
    if (tid < 2 * C::TR) reinterpret_cast<float*>(b + 2 * C::TILE)[tid] = rs;
  };
  load(0);
  store(0);
  __syncthreads();
  const int nt = N / C::TR;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sq = smem + cur * STAGE;
    const char* so = sq + C::TILE;
    const float* sl = reinterpret_cast<const float*>(sq + 2 * C::TILE);
    const float* sd = sl + C::TR;
    bf16x8 pf[2][2], dsf[2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
      // accumulators start at -L2[q] / -delta[q] of their row (q = kt2*32 + 8g + 4h + j; the LDS copies are negated): the four 16-byte
      // reads ARE the accumulator tuple -- no move, no negation on the vector pipe
      f32x4 l4[4], d4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        l4[g] = *reinterpret_cast<const f32x4*>(sl + kt2 * 32 + 8 * g + 4 * lh);
        d4[g] = *reinterpret_cast<const f32x4*>(sd + kt2 * 32 + 8 * g + 4 * lh);
      }
      typedef __attribute__((ext_vector_type(8))) float f32x8;
      f32x16 sacc = __builtin_shufflevector(__builtin_shufflevector(l4[0], l4[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(l4[2], l4[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                            0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      f32x16 pacc = __builtin_shufflevector(__builtin_shufflevector(d4[0], d4[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(d4[2], d4[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                            0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      const int row = kt2 * 32 + lq;
#pragma unroll
      for (int ks = 0; ks < DC / 16; ++ks) {
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sq + row * C::ROWB + (ks * 2 + lh) * 16);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], sacc, 0, 0, 0);
        const bf16x8 oa = *reinterpret_cast<const bf16x8*>(so + row * C::ROWB + (ks * 2 + lh) * 16);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[ks], pacc, 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(sacc[8 * s + j]);
          pf[kt2][s][j] = f2bf(p);
          dsf[kt2][s][j] = f2bf(p * pacc[8 * s + j]);
        }
    }
#pragma unroll
    for (int dvt = 0; dvt < DW / 32; ++dvt)
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          dva[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<D>(so, dvt * 32, kt2, s, lane), pf[kt2][s], dva[dvt], 0, 0, 0);
          dka[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<D>(sq, dvt * 32, kt2, s, lane), dsf[kt2][s], dka[dvt], 0, 0, 0);
        }
    if (t + 1 < nt) {  // staged after the products: the tile registers are not live across them (register budget at D = 128)
      load(t + 1);
      store(cur ^ 1);
    }
    __syncthreads();
    cur ^= 1;
  }
  bf16* krow = dK + base + (long)(k0 + lq) * D;
  bf16* vrow = dV + base + (long)(k0 + lq) * D;
  if constexpr (DW < D) {
#pragma unroll
    for (int c = DW + 4 * lh; c < D; c += 8) {
      *reinterpret_cast<bf16x4*>(krow + c) = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
      *reinterpret_cast<bf16x4*>(vrow + c) = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
    }
  }
#pragma unroll
  for (int dvt = 0; dvt < DW / 32; ++dvt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 k4, v4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        k4[j] = f2bf(dka[dvt][4 * g4 + j] * sk_scale);
        v4[j] = f2bf(dva[dvt][4 * g4 + j]);
      }
      *reinterpret_cast<bf16x4*>(krow + dvt * 32 + 8 * g4 + 4 * lh) = k4;
      *reinterpret_cast<bf16x4*>(vrow + dvt * 32 + 8 * g4 + 4 * lh) = v4;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA forms of the two kernels (default for the 128-element-row instances; DFOT_ATTN_BWD_DMA=0 / 2: never / always).
// The streamed tiles are fetched by global_load_lds straight into an UNPADDED LDS image, two stages, the next tile in flight
// while the present one is multiplied: no staging registers (32 VGPRs: the d = 128 instances drop under 256 and run two waves
// per SIMD -- register-staged they were 266 / 280 VGPRs, one wave per SIMD, and the dk/dv kernel fetched its next tile AFTER the
// products with nothing to overlap the latency: MFMA utilisation 0.22-0.25 against 0.50 at d = 64).  One XOR swizzle of the
// 16-byte chunks, applied on the source side, keeps BOTH access patterns of a tile conflict-free:
//   row reads (ds_read_b128, 16 rows at one chunk position): the swizzle is a bijection of the row's low bits;
//   transposed reads (ds_read_b64_tr_b16, 4 rows x 64 contiguous bytes per half-wave): rows r..r+3 land in different 64-byte groups.
template <int D>
struct DmaCfg {
  static constexpr int TR = 64, ROWB = D * 2, TILE = TR * ROWB, CH = D / 8;
  static constexpr int RPI = 1024 / ROWB;        // rows per 1-KiB DMA instruction (8 or 4)
  static constexpr int IPW = (TILE / 1024) / 4;  // DMA instructions per wave and tile (2 or 4)
  __device__ static int swz(int row, int c) {
    return D == 64 ? (c ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) : (c ^ (((row & 3) << 2) | ((row >> 2) & 3)));
  }
};

typedef __attribute__((ext_vector_type(2))) unsigned bw_u32x2;
template <int OFF>
__device__ __forceinline__ bw_u32x2 bw_read_tr16(unsigned addr) {
  bw_u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void bw_wait(bw_u32x2& a, bw_u32x2& b, bw_u32x2& c, bw_u32x2& d, bw_u32x2& e, bw_u32x2& f, bw_u32x2& g, bw_u32x2& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ bf16x8 bw_bf16x8(bw_u32x2 lo, bw_u32x2 hi) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}
// byte offset inside a tile of this lane's transposed read for the feature block dvt (32 features) and row half h (rows 8h..): rows
// 4 lh + 8 h + q4 (+ 16 s + 32 kt2 as an immediate: multiples of 16 rows leave the swizzle unchanged)
template <int D>
__device__ __forceinline__ unsigned bw_tr_off(int dvt, int h, int lane) {
  using C = DmaCfg<D>;
  const int lh = lane >> 5, q4 = (lane & 15) >> 2, p4 = lane & 3, g = (lane >> 4) & 1;
  const int row = 4 * lh + 8 * h + q4;
  const int chunk = 4 * dvt + 2 * g + (p4 >> 1);
  return (unsigned)(row * C::ROWB + C::swz(row, chunk) * 16 + (p4 & 1) * 8);
}
// the four X^T fragments (kt2, s) of feature block dvt of one tile
template <int D>
__device__ __forceinline__ void bw_tr_frags(unsigned tile_addr, int dvt, int lane, bf16x8 (&f)[2][2]) {
  using C = DmaCfg<D>;
  const unsigned a0 = tile_addr + bw_tr_off<D>(dvt, 0, lane), a1 = tile_addr + bw_tr_off<D>(dvt, 1, lane);
  bw_u32x2 r000 = bw_read_tr16<0 * C::ROWB>(a0), r001 = bw_read_tr16<0 * C::ROWB>(a1);
  bw_u32x2 r010 = bw_read_tr16<16 * C::ROWB>(a0), r011 = bw_read_tr16<16 * C::ROWB>(a1);
  bw_u32x2 r100 = bw_read_tr16<32 * C::ROWB>(a0), r101 = bw_read_tr16<32 * C::ROWB>(a1);
  bw_u32x2 r110 = bw_read_tr16<48 * C::ROWB>(a0), r111 = bw_read_tr16<48 * C::ROWB>(a1);
  bw_wait(r000, r001, r010, r011, r100, r101, r110, r111);
  f[0][0] = bw_bf16x8(r000, r001);
  f[0][1] = bw_bf16x8(r010, r011);
  f[1][0] = bw_bf16x8(r100, r101);
  f[1][1] = bw_bf16x8(r110, r111);
}

template <int D, int DC = D, int DW = D>
__global__ __launch_bounds__(256, (D == 64 ? 3 : 2)) void attn_bwd_dq_dma_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                                 const bf16* __restrict__ V, const bf16* __restrict__ dO, long ldo,
                                                                 const float* __restrict__ L2, const float* __restrict__ delta,
                                                                 bf16* __restrict__ dQ, int N, int heads, int d, float sq) {
  using C = DmaCfg<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int qtiles = N / 128;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lin / qtiles;
  const long base = (long)bh * N * D;
  const int q0 = (lin % qtiles) * 128 + wave * 32;
  const bf16* Kb = K + base;
  const bf16* Vb = V + base;

  // per-lane DMA source offsets (elements) within a tile: LDS position (row, pos) receives source chunk swz(row, pos)
  int toff[C::IPW];
#pragma unroll
  for (int i = 0; i < C::IPW; ++i) {
    const int inst = wave * C::IPW + i;
    const int row = inst * C::RPI + lane / C::CH, pos = lane % C::CH;
    toff[i] = row * D + C::swz(row, pos) * 8;
  }
  auto issue = [&](int t, int stage) {
    char* sk = smem + stage * 2 * C::TILE;
    char* sv = sk + C::TILE;
    const bf16* kt = Kb + (long)t * C::TR * D;
    const bf16* vt = Vb + (long)t * C::TR * D;
#pragma unroll
    for (int i = 0; i < C::IPW; ++i) {
      const int inst = wave * C::IPW + i;
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(kt + toff[i]), DFOT_LDS_PTR(sk + inst * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(vt + toff[i]), DFOT_LDS_PTR(sv + inst * 1024), 16, 0, 0);
    }
  };
  issue(0, 0);

  bf16x8 qf[DC / 16], dof[DC / 16];
  // dO is read from the compact activation layout [B*N][ldo] (head hd at column hd*d); columns >= d belong to the next head: zero
  const bf16* dorow = dO + ((long)(bh / heads) * N + q0 + lq) * ldo + (long)(bh % heads) * d;
#pragma unroll
  for (int ks = 0; ks < DC / 16; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(Q + base + (long)(q0 + lq) * D + ks * 16 + lh * 8);
    const int col = ks * 16 + lh * 8;
    const bf16 z = f2bf(0.f);
    dof[ks] = col < d ? *reinterpret_cast<const bf16x8*>(dorow + col) : bf16x8{z, z, z, z, z, z, z, z};
  }
  const float l2 = L2[(long)bh * N + q0 + lq], dl = delta[(long)bh * N + q0 + lq];

  f32x16 acc[DW / 32];
#pragma unroll
  for (int i = 0; i < DW / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < DC / 16; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));  // see attn_bwd_dkv_dma_kernel
  float l2v = l2, dlv = dl;
  asm volatile("" : "+v"(l2v), "+v"(dlv));
  const int nt = N / C::TR;
  const unsigned lds0 = (unsigned)(size_t)DFOT_LDS_PTR(smem);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sk = smem + cur * 2 * C::TILE;
    const char* sv = sk + C::TILE;
    if (t + 1 < nt) issue(t + 1, cur ^ 1);
    // the lane id is made opaque once per tile: the swizzled LDS offsets below are a few VALU operations each, but loop-invariant,
    // and hoisted out of the tile loop they cost ~30 VGPRs (the d = 128 instances must stay under 256)
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int lqv = lv & 31, lhv = lv >> 5;
    bf16x8 dsf[2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = -l2v; pacc[r] = -dlv; }
      const int row = kt2 * 32 + lqv;
#pragma unroll
      for (int ks = 0; ks < DC / 16; ++ks) {
        const int off = row * C::ROWB + C::swz(row, ks * 2 + lhv) * 16;
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + off);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc, 0, 0, 0);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sv + off);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], pacc, 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) dsf[kt2][s][j] = f2bf(__builtin_amdgcn_exp2f(sacc[8 * s + j]) * pacc[8 * s + j]);
    }
    // dQ^T[c][q] += K^T[c][key] dS^T[key][q]
    const unsigned ka = lds0 + cur * 2 * C::TILE;
#pragma unroll
    for (int dvt = 0; dvt < DW / 32; ++dvt) {
      bf16x8 kt[2][2];
      bw_tr_frags<D>(ka, dvt, lv, kt);
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[kt2][s], dsf[kt2][s], acc[dvt], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    cur ^= 1;
  }
  bf16* orow = dQ + base + (long)(q0 + lq) * D;
  if constexpr (DW < D) {  // pad columns beyond DW: zero (the caller's buffers are re-used across blocks)
#pragma unroll
    for (int c = DW + 4 * lh; c < D; c += 8) *reinterpret_cast<bf16x4*>(orow + c) = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
  }
#pragma unroll
  for (int dvt = 0; dvt < DW / 32; ++dvt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = f2bf(acc[dvt][4 * g4 + j] * sq);
      *reinterpret_cast<bf16x4*>(orow + dvt * 32 + 8 * g4 + 4 * lh) = o4;
    }
}

// (the d = 128 instance holds k, v fragments (64 VGPRs) and the dK, dV accumulators (128): 256 cannot be met, it keeps one wave per
// SIMD -- what it gains is the prefetch of the next tile behind the products)
template <int D, int DC = D, int DW = D>
__global__ __launch_bounds__(256, (DC <= 96 ? 2 : 1)) void attn_bwd_dkv_dma_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                                  const bf16* __restrict__ V, const bf16* __restrict__ dO, long ldo,
                                                                  const float* __restrict__ L2, const float* __restrict__ delta,
                                                                  bf16* __restrict__ dK, bf16* __restrict__ dV, int N, int heads, int d,
                                                                  float sk_scale, const bf16* __restrict__ zeros) {
  using C = DmaCfg<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int ktiles = N / 128;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lin / ktiles;
  const long base = (long)bh * N * D;
  const int k0 = (lin % ktiles) * 128 + wave * 32;
  const bf16* Qb = Q + base;
  const bf16* Ob = dO + (long)(bh / heads) * N * ldo + (long)(bh % heads) * d;  // compact [B*N][ldo], this head's columns
  const float* Lb = L2 + (long)bh * N;
  const float* Db = delta + (long)bh * N;

  // LDS: [stage][Q tile | dO tile | L2[64] | delta[64]]; the two statistics rows are kept NEGATED (they are read straight into the S / dP
  // accumulators: no sign flips on the vector pipe) except in the <128, 96, 96> instance, which that costs 7 spilled registers
  constexpr int STAGE = 2 * C::TILE + 2 * C::TR * 4;
  constexpr bool NEGL = !(D == 128 && DC == 96);
  // per-lane DMA sources: Q rows have pitch D; dO rows pitch ldo, chunks at or beyond d (next head / padding) come from the zero buffer
  int qoff[C::IPW];
  long ooff[C::IPW];
  bool ook[C::IPW];
#pragma unroll
  for (int i = 0; i < C::IPW; ++i) {
    const int inst = wave * C::IPW + i;
    const int row = inst * C::RPI + lane / C::CH, ch = C::swz(row, lane % C::CH);
    qoff[i] = row * D + ch * 8;
    ooff[i] = (long)row * ldo + ch * 8;
    ook[i] = ch * 8 < d;
  }
  float rs = 0.f;
  auto issue = [&](int t, int stage) {
    char* sq = smem + stage * STAGE;
    char* so = sq + C::TILE;
    if (tid < 2 * C::TR) {  // before the DMAs: its wait leaves them in flight
      rs = tid < C::TR ? Lb[t * C::TR + tid] : Db[t * C::TR + tid - C::TR];
      if constexpr (NEGL) rs = -rs;
    }
    const bf16* qt = Qb + (long)t * C::TR * D;
    const bf16* ot = Ob + (long)t * C::TR * ldo;
#pragma unroll
    for (int i = 0; i < C::IPW; ++i) {
      const int inst = wave * C::IPW + i;
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(qt + qoff[i]), DFOT_LDS_PTR(sq + inst * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(ook[i] ? ot + ooff[i] : zeros), DFOT_LDS_PTR(so + inst * 1024), 16, 0, 0);
    }
  };
  auto store_stats = [&](int stage) {
    if (tid < 2 * C::TR) reinterpret_cast<float*>(smem + stage * STAGE + 2 * C::TILE)[tid] = rs;
  };
  issue(0, 0);

  bf16x8 kf[DC / 16], vf[DC / 16];
#pragma unroll
  for (int ks = 0; ks < DC / 16; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(K + base + (long)(k0 + lq) * D + ks * 16 + lh * 8);
    vf[ks] = *reinterpret_cast<const bf16x8*>(V + base + (long)(k0 + lq) * D + ks * 16 + lh * 8);
  }
  f32x16 dka[DW / 32], dva[DW / 32];
#pragma unroll
  for (int i = 0; i < DW / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dka[i][r] = 0.f; dva[i][r] = 0.f; }

  store_stats(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // the compiler does not see the wait above: without a use of the fragments HERE it puts its own `s_waitcnt vmcnt(0)` in front of
  // their first use inside the loop, every iteration, which drains the tile prefetch that was just issued
#pragma unroll
  for (int ks = 0; ks < DC / 16; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
  const int nt = N / C::TR;
  const unsigned lds0 = (unsigned)(size_t)DFOT_LDS_PTR(smem);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sq = smem + cur * STAGE;
    const char* so = sq + C::TILE;
    const float* sl = reinterpret_cast<const float*>(sq + 2 * C::TILE);
    const float* sd = sl + C::TR;
    if (t + 1 < nt) issue(t + 1, cur ^ 1);
    int lv = lane;  // opaque per tile: see attn_bwd_dq_dma_kernel
    asm volatile("" : "+v"(lv));
    const int lqv = lv & 31, lhv = lv >> 5;
    bf16x8 pf[2][2], dsf[2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
      f32x16 sacc, pacc;
      // accumulators start at -L2[q] / -delta[q] of their row: q = kt2*32 + 8g + 4h + j (the LDS copies are negated)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sl + kt2 * 32 + 8 * g + 4 * lhv);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sd + kt2 * 32 + 8 * g + 4 * lhv);
#pragma unroll
        for (int j = 0; j < 4; ++j) { sacc[4 * g + j] = NEGL ? l4[j] : -l4[j]; pacc[4 * g + j] = NEGL ? d4[j] : -d4[j]; }
      }
      const int row = kt2 * 32 + lqv;
#pragma unroll
      for (int ks = 0; ks < DC / 16; ++ks) {
        const int off = row * C::ROWB + C::swz(row, ks * 2 + lhv) * 16;
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sq + off);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], sacc, 0, 0, 0);
        const bf16x8 oa = *reinterpret_cast<const bf16x8*>(so + off);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[ks], pacc, 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(sacc[8 * s + j]);
          pf[kt2][s][j] = f2bf(p);
          dsf[kt2][s][j] = f2bf(p * pacc[8 * s + j]);
        }
    }
    const unsigned qa0 = lds0 + cur * STAGE, oa0 = qa0 + C::TILE;
#pragma unroll
    for (int dvt = 0; dvt < DW / 32; ++dvt) {
      bf16x8 ft[2][2];
      bw_tr_frags<D>(oa0, dvt, lv, ft);
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int s = 0; s < 2; ++s) dva[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[kt2][s], pf[kt2][s], dva[dvt], 0, 0, 0);
      bw_tr_frags<D>(qa0, dvt, lv, ft);
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int s = 0; s < 2; ++s) dka[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[kt2][s], dsf[kt2][s], dka[dvt], 0, 0, 0);
    }
    if (t + 1 < nt) store_stats(cur ^ 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cur ^= 1;
  }
  bf16* krow = dK + base + (long)(k0 + lq) * D;
  bf16* vrow = dV + base + (long)(k0 + lq) * D;
  if constexpr (DW < D) {
#pragma unroll
    for (int c = DW + 4 * lh; c < D; c += 8) {
      *reinterpret_cast<bf16x4*>(krow + c) = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
      *reinterpret_cast<bf16x4*>(vrow + c) = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
    }
  }
#pragma unroll
  for (int dvt = 0; dvt < DW / 32; ++dvt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 k4, v4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        k4[j] = f2bf(dka[dvt][4 * g4 + j] * sk_scale);
        v4[j] = f2bf(dva[dvt][4 * g4 + j]);
      }
      *reinterpret_cast<bf16x4*>(krow + dvt * 32 + 8 * g4 + 4 * lh) = k4;
      *reinterpret_cast<bf16x4*>(vrow + dvt * 32 + 8 * g4 + 4 * lh) = v4;
    }
}

// delta[b][hd][n] = sum_c dO[row][hd*d + c] * O[row][hd*d + c]  (compact [B*N][ldo] activations, d % 8 == 0); thread per (row, head)
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const bf16* __restrict__ O, const bf16* __restrict__ dO, long ldo,
                                                             float* __restrict__ delta, long rows, int N, int heads, int d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * heads) return;
  const long row = i / heads;
  const int hd = (int)(i % heads);
  const bf16* o = O + row * ldo + (long)hd * d;
  const bf16* g = dO + row * ldo + (long)hd * d;
  float acc = 0.f;
  for (int c = 0; c < d; c += 8) {
    const bf16x8 ov = *reinterpret_cast<const bf16x8*>(o + c);
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += bf2f(ov[j]) * bf2f(gv[j]);
  }
  delta[((row / N) * heads + hd) * N + row % N] = acc;
}

// the coalesced form for head dims of 64 or 128 (8 or 16 lanes per head, 16 bytes each): consecutive lanes read consecutive chunks of a row,
// the head's lanes are summed by xor shuffles (ldo % 8 == 0: launcher).
template <int SEG>
__global__ __launch_bounds__(256) void attn_bwd_delta_rows_kernel(const bf16* __restrict__ O, const bf16* __restrict__ dO, long ldo,
                                                                  float* __restrict__ delta, long rows, int N, int heads) {
  const int cpr = heads * SEG;  // 16-byte chunks of a row that hold attention output
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = g / cpr;
  const int chunk = (int)(g % cpr);
  float acc = 0.f;
  if (row < rows) {
    const bf16x8 ov = *reinterpret_cast<const bf16x8*>(O + row * ldo + chunk * 8);
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(dO + row * ldo + chunk * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += bf2f(ov[j]) * bf2f(gv[j]);
  }
#pragma unroll
  for (int o = 1; o < SEG; o <<= 1) acc += __shfl_xor(acc, o);
  if (row < rows && chunk % SEG == 0) delta[((row / N) * heads + chunk / SEG) * N + row % N] = acc;
}

// any head dim that is a multiple of 8 (DiT: d = 72): a thread takes one 16-byte chunk of a row (coalesced along the row), the d / 8
// partial products of a head are added through LDS.  One workgroup = RPW whole rows (RPW * heads * d / 8 <= 1024 threads).
__global__ __launch_bounds__(1024) void attn_bwd_delta_lds_kernel(const bf16* __restrict__ O, const bf16* __restrict__ dO, long ldo,
                                                                  float* __restrict__ delta, long rows, int N, int heads, int d, int rpw) {
  extern __shared__ float dpart[];
  const int cph = d / 8, cpr = heads * cph;
  const int lr = threadIdx.x / cpr, chunk = threadIdx.x % cpr;
  const long row = (long)blockIdx.x * rpw + lr;
  float acc = 0.f;
  if (lr < rpw && row < rows) {
    const bf16x8 ov = *reinterpret_cast<const bf16x8*>(O + row * ldo + chunk * 8);
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(dO + row * ldo + chunk * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += bf2f(ov[j]) * bf2f(gv[j]);
  }
  dpart[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < rpw * heads) {
    const int r = threadIdx.x / heads, hd = threadIdx.x % heads;
    const long orow = (long)blockIdx.x * rpw + r;
    if (orow < rows) {
      float t = 0.f;
      for (int c = 0; c < cph; ++c) t += dpart[r * cpr + hd * cph + c];
      delta[((orow / N) * heads + hd) * N + orow % N] = t;
    }
  }
}

}  // namespace

int launch_attention_bwd_delta(const bf16* o, const bf16* d_o, long ldo, float* delta, int batch, int heads, int n, int d, hipStream_t s) {
  DFOT_REQUIRE(d % 8 == 0 && ldo % 8 == 0, DFOT_ERR_SHAPE, "attention_bwd: head dim %d and row stride %ld must be multiples of 8", d, ldo);
  const long rows = (long)batch * n;
  if ((d == 64 || d == 128) && ((uintptr_t)o & 15) == 0 && ((uintptr_t)d_o & 15) == 0) {
    // a head's 8 / 16 lanes start at a multiple of 8 / 16 of the global thread index (the chunks per row are a multiple of it), so they
    // never straddle two waves
    const long threads = rows * (heads * d / 8);
    if (d == 64) hipLaunchKernelGGL(attn_bwd_delta_rows_kernel<8>, dim3(cdiv(threads, 256)), dim3(256), 0, s, o, d_o, ldo, delta, rows, n, heads);
    else hipLaunchKernelGGL(attn_bwd_delta_rows_kernel<16>, dim3(cdiv(threads, 256)), dim3(256), 0, s, o, d_o, ldo, delta, rows, n, heads);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  }
  const int cpr = heads * d / 8;
  if (cpr <= 1024 && ((uintptr_t)o & 15) == 0 && ((uintptr_t)d_o & 15) == 0) {
    const int rpw = 512 / cpr > 0 ? 512 / cpr : 1;  // ~512 threads per workgroup
    const int threads = rpw * cpr;
    hipLaunchKernelGGL(attn_bwd_delta_lds_kernel, dim3(cdiv(rows, (long)rpw)), dim3(threads), threads * sizeof(float), s, o, d_o, ldo, delta, rows, n, heads, d, rpw);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  }
  hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3(cdiv(rows * heads, 256)), dim3(256), 0, s, o, d_o, ldo, delta, rows, n, heads, d);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

const bf16* g_bwd_zeros = nullptr;

template <int D, int DC = D, int DW = D>
static int launch_bwd_t(const bf16* q, const bf16* k, const bf16* v, const bf16* d_o, long ldo, const float* l2, const float* delta, bf16* dq,
                        bf16* dk, bf16* dv, int batch, int heads, int n, int d, float sq, float sk, hipStream_t s) {
  // 1 (default): LDS-DMA forms for the 128-element-row instances only; 2: for all; 0: register-staged everywhere.  Measured
  // (RE10K step, same box): d = 128 dq 367 -> 253 us, dk/dv 567 -> 480 us; d = 64 dq 1782 -> 1782 us, dk/dv 2377 -> 2565 us (its
  // 226 VGPRs leave two waves per SIMD where the register-staged form's 167 leave three)
  static const int dma = tuning_flag("ATTN_BWD_DMA", 1);
  const int grid = (n / 128) * batch * heads;
  if (dma == 2 || (dma == 1 && D == 128)) {
    using C = DmaCfg<D>;
    const int lds1 = 4 * C::TILE, lds2 = 2 * (2 * C::TILE + 2 * C::TR * 4);
    static bool attr_set = false;
    if (!attr_set) {
      DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_dma_kernel<D, DC, DW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
      DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_dma_kernel<D, DC, DW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
      attr_set = true;
    }
    if (!g_bwd_zeros) {
      void* z = nullptr;
      DFOT_CHECK_HIP(hipMalloc(&z, 256));
      DFOT_CHECK_HIP(hipMemset(z, 0, 256));
      g_bwd_zeros = (const bf16*)z;
    }
    hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<D, DC, DW>), dim3(grid), dim3(256), lds1, s, q, k, v, d_o, ldo, l2, delta, dq, n, heads, d, sq);
    DFOT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL((attn_bwd_dkv_dma_kernel<D, DC, DW>), dim3(grid), dim3(256), lds2, s, q, k, v, d_o, ldo, l2, delta, dk, dv, n, heads, d, sk,
                       g_bwd_zeros);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  }
  using C = BwdCfg<D>;
  const int lds1 = 4 * C::TILE, lds2 = 2 * (2 * C::TILE + 2 * C::TR * 4);
  static bool attr_set = false;
  if (!attr_set) {
    DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<D, DC, DW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
    DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<D, DC, DW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_bwd_dq_kernel<D, DC, DW>), dim3(grid), dim3(256), lds1, s, q, k, v, d_o, ldo, l2, delta, dq, n, heads, d, sq);
  DFOT_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, DC, DW>), dim3(grid), dim3(256), lds2, s, q, k, v, d_o, ldo, l2, delta, dk, dv, n, heads, d, sk);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// q/k/v/dq/dk/dv: [B][heads][N][dstride(d)] bf16 ; d_o: compact [B*N][ldo] ; l2/delta: [B][heads][N] fp32
int launch_attention_bwd(const bf16* q, const bf16* k, const bf16* v, const bf16* d_o, long ldo, const float* l2, const float* delta,
                         bf16* dq, bf16* dk, bf16* dv, int batch, int heads, int n, int d, hipStream_t s) {
  DFOT_REQUIRE(q && k && v && d_o && l2 && delta && dq && dk && dv, DFOT_ERR_ARG, "attention_bwd: null pointer");
  DFOT_REQUIRE(d > 0 && d <= 128 && d % 8 == 0 && ldo % 8 == 0, DFOT_ERR_SHAPE, "attention_bwd: head dim %d must be a multiple of 8, <= 128", d);
  DFOT_REQUIRE(n > 0 && n % 128 == 0, DFOT_ERR_SHAPE, "attention_bwd: N=%d must be a multiple of 128", n);
  const float sq = 1.0f / sqrtf((float)d), sk = 0.6931471805599453f;
#define BWD_ARGS q, k, v, d_o, ldo, l2, delta, dq, dk, dv, batch, heads, n, d, sq, sk, s
  if (d <= 32) return launch_bwd_t<64, 32, 32>(BWD_ARGS);
  if (d <= 64) return launch_bwd_t<64>(BWD_ARGS);
  if (d <= 80) return launch_bwd_t<128, 80, 96>(BWD_ARGS);
  if (d <= 96) return launch_bwd_t<128, 96, 96>(BWD_ARGS);
  return launch_bwd_t<128>(BWD_ARGS);
#undef BWD_ARGS
}

}  // namespace dfot
