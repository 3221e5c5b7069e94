// Flash-style non-causal self-attention forward for the DFoT transformer levels (gfx950).
//   level 2: N = T*32*32 = 8192 tokens, d = 64 ; level 3: N = 2048, d = 128 ; 9 heads.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows.
// Both products are computed "transposed" so that the query index stays on the MFMA lane:
//   S^T (keys x q)  = K  (A operand, LDS rows)        x Q^T (B operand, registers)
//   O^T (dv   x q)  = V^T(A operand, LDS transposed read) x P^T (B operand = exp'd S^T registers)
// With v_mfma_f32_32x32x16_bf16 the accumulator of S^T holds, per lane, 16 keys of ONE query
// column, so the online-softmax statistics (running max m, sum l, rescale alpha) and the O^T
// accumulator columns are all lane-local; the only cross-lane traffic is one exchange with lane^32.
// The S^T accumulator registers are used directly as the B fragment of the second product; the
// k (=key) order inside a 16-key step is then permuted (key = 16s + 8(j>>2) + 4h + (j&3)), and the
// V^T fragment is fetched in that same order by two ds_read_b64_tr_b16 per step
// (cdna_hip_programming.md section 3 "accumulator tile as the next MFMA's operand", T10).
// Scores are in the exp2 domain: the caller folds log2(e)/sqrt(d) into q.
#include "common.h"
#include "dfot_hip.h"
#include "kernels.h"

namespace dfot {

template <int D>
struct AttnCfg {
  static constexpr int KV = 64;                  // keys per tile
  static constexpr int ROWB = D * 2;             // bytes per K/V row in LDS
  static constexpr int TILE = KV * ROWB;         // bytes of one K or V tile
  static constexpr int CH = D / 8;               // 16-byte chunks per row
  static constexpr int PER_THREAD = KV * CH / 256;
  __device__ static int swz_k(int row, int c) { return D == 64 ? (c ^ ((row >> 1) & 7)) : (c ^ (row & 15)); }
  __device__ static int swz_v(int row, int c) { return D == 64 ? (c ^ (((row >> 1) & 1) << 2)) : (c ^ ((row & 3) << 2)); }
};

template <int D, bool USE_TR>
__global__ __launch_bounds__(256) void attn_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                   const bf16* __restrict__ V, bf16* __restrict__ O, long ldo,
                                                   int N, int heads) {
  using C = AttnCfg<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int bh = blockIdx.y;
  const long base = (long)bh * N * D;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bf16* Qb = Q + base;
  const bf16* Kb = K + base;
  const bf16* Vb = V + base;

  // Q fragments: B operand, lane holds Q[q0+lq][16*ks + 8*lh + j]
  bf16x8 qf[D / 16];
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(Qb + (long)(q0 + lq) * D + ks * 16 + lh * 8);

  f32x16 oacc[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_i = -INFINITY, l_i = 0.f;

  bf16x8 rk[C::PER_THREAD], rv[C::PER_THREAD];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < C::PER_THREAD; ++i) {
      const int e = tid + 256 * i;
      const int row = e / C::CH, c = e % C::CH;
      const long off = (long)(t * C::KV + row) * D + c * 8;
      rk[i] = *reinterpret_cast<const bf16x8*>(Kb + off);
      rv[i] = *reinterpret_cast<const bf16x8*>(Vb + off);
    }
  };
  auto store_tile = [&](int stage) {
    char* sk = smem + stage * 2 * C::TILE;
    char* sv = sk + C::TILE;
#pragma unroll
    for (int i = 0; i < C::PER_THREAD; ++i) {
      const int e = tid + 256 * i;
      const int row = e / C::CH, c = e % C::CH;
      *reinterpret_cast<bf16x8*>(sk + row * C::ROWB + C::swz_k(row, c) * 16) = rk[i];
      *reinterpret_cast<bf16x8*>(sv + row * C::ROWB + C::swz_v(row, c) * 16) = rv[i];
    }
  };

  const int nt = N / C::KV;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    const char* sk = smem + cur * 2 * C::TILE;
    const char* sv = sk + C::TILE;
    if (t + 1 < nt) load_tile(t + 1);

    // ---- S^T = K Q^T : two 32-key sub-tiles ----
    f32x16 sacc[2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt2][r] = 0.f;
      const int row = kt2 * 32 + lq;
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + row * C::ROWB + C::swz_k(row, ks * 2 + lh) * 16);
        sacc[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kt2], 0, 0, 0);
      }
    }

    // ---- online softmax (exp2 domain), statistics lane-local per query column ----
    float mx = sacc[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_i, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_i - m_new);
    float rs = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(sacc[kt2][8 * s + j] - m_new);
          rs += p;
          pf[kt2][s][j] = f2bf(p);
        }
    l_i = l_i * alpha + rs;
    m_i = m_new;
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int dvt = 0; dvt < D / 32; ++dvt) {
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 vf;
          if constexpr (USE_TR) {
            // 16-lane group gi=lane>>4 reads the 4x16 block rows kb+{0..3}, cols dvt*32 + 16*(gi&1) + {0..15};
            // lane 4q+p of the group supplies row q, columns 4p..4p+3
            const int kb = kt2 * 32 + 16 * s + 4 * lh;
            const int q4 = (lane & 15) >> 2, p4 = lane & 3;
            const int col = dvt * 32 + 16 * ((lane >> 4) & 1) + 4 * p4;
            const int r0 = kb + q4, r1 = kb + 8 + q4;
            const char* a0 = sv + r0 * C::ROWB + C::swz_v(r0, col >> 3) * 16 + (col & 7) * 2;
            const char* a1 = sv + r1 * C::ROWB + C::swz_v(r1, col >> 3) * 16 + (col & 7) * 2;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(a0));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(a1));
            vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          } else {
            const int dv = dvt * 32 + lq;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int key = kt2 * 32 + 16 * s + 8 * (j >> 2) + 4 * lh + (j & 3);
              vf[j] = *reinterpret_cast<const bf16*>(sv + key * C::ROWB + C::swz_v(key, dv >> 3) * 16 + (dv & 7) * 2);
            }
          }
          oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt2][s], oacc[dvt], 0, 0, 0);
        }
      }
    }

    if (t + 1 < nt) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[q0+lq][dvt*32 + 8*g + 4*lh + {0..3}] in oacc[dvt][4g..4g+3] ----
  const float l_tot = l_i + __shfl_xor(l_i, 32);
  const float inv = 1.0f / l_tot;
  const int b = bh / heads, hd = bh % heads;
  bf16* orow = O + ((long)b * N + q0 + lq) * ldo + hd * D;
#pragma unroll
  for (int dvt = 0; dvt < D / 32; ++dvt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = f2bf(oacc[dvt][4 * g4 + j] * inv);
      *reinterpret_cast<bf16x4*>(orow + dvt * 32 + 8 * g4 + 4 * lh) = o4;
    }
}

typedef __attribute__((ext_vector_type(2))) unsigned v2_u32x2;
template <int OFF>
__device__ __forceinline__ v2_u32x2 v2_read_tr16(unsigned addr) {
  v2_u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void v2_lds_wait(v2_u32x2& a, v2_u32x2& b, v2_u32x2& c, v2_u32x2& d, v2_u32x2& e, v2_u32x2& f, v2_u32x2& g, v2_u32x2& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ bf16x8 v2_bf16x8(v2_u32x2 lo, v2_u32x2 hi) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---------------------------------------------------------------------------------------------
// Tuned variant (default).  Same products and layouts as attn_kernel above, plus:
//  * K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4); the bank swizzles are involutions applied to
//    the per-lane SOURCE chunk, so the reads above stay valid; no staging registers, no ds_write pass;
//  * the running max is folded into the MFMA: the S^T accumulator starts at -m (per query column, lane-local),
//    so the product leaves s - m ready and exp2 needs no subtract;
//  * deferred rescale (threshold 8 in the log2 domain): O, l are rescaled only when some query's max grows by
//    more than 2^8 -- the branch is wave-uniform; P is then bounded by 256, exact enough for bf16 P / fp32 sums;
//  * row max via v_max3.
// ---------------------------------------------------------------------------------------------
// DQK / DV: head dims actually multiplied (multiples of 16 / 32, <= D) when the logical head dim is smaller than the
// row stride D of the q/k/v layout (DiT: d = 72 in 128-element rows -> DQK 80, DV 96; the pad columns hold zeros).
// Output: head hd of query row r goes to O[r*ldo + hd*ohs + c] for c < dvalid.
template <int D, int NST, int DQK = D, int DV = D, bool QRELOAD = false>
__global__ __launch_bounds__(256, QRELOAD ? 4 : 1) void attn_kernel_v2(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                      const bf16* __restrict__ V, bf16* __restrict__ O, long ldo, int N,
                                                      int heads, int xcd, int ohs, int dvalid, float* __restrict__ lse) {
  static_assert(DQK % 16 == 0 && DV % 32 == 0 && DQK <= D && DV <= D, "head-dim sub-range");
  using C = AttnCfg<D>;
  constexpr float THR = 8.0f;
  constexpr int RPI = 1024 / C::ROWB;           // rows covered by one 1-KiB DMA instruction (8 or 4)
  constexpr int IPW = (C::TILE / 1024) / 4;     // DMA instructions per wave per tile (2 or 4)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  // 1-D grid, (batch*head)-major logical order handed out per XCD: the q-tiles of one head share K/V in one L2
  const int qtiles = N / 128;
  const int lin = xcd_remap(blockIdx.x, gridDim.x, xcd & 1);
  constexpr int kt0 = 0;
  const int kt1 = N / C::KV;
  const bool prio = (xcd & 2) != 0;  // A/B switch: raise the wave priority around the MFMA clusters
  const int bh = lin / qtiles;
  const long base = (long)bh * N * D;
  const int q0 = (lin % qtiles) * 128 + wave * 32;
  const bf16* Qb = Q + base;
  const bf16* Kb = K + base;
  const bf16* Vb = V + base;

  // QRELOAD: do not keep the Q fragments (DQK/16 x 4 VGPRs) live across the K/V loop; re-read them (L1/L2 hits) per tile so
  // that the kernel fits 128 VGPRs = 4 waves per SIMD
  bf16x8 qf[DQK / 16];
  const bf16* qrow = Qb + (long)(q0 + lq) * D + lh * 8;
  if constexpr (!QRELOAD) {
#pragma unroll
    for (int ks = 0; ks < DQK / 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 16);
  }

  // per-lane DMA source offsets (elements) within a tile: LDS position (row, pos) receives source chunk swz(row,pos)
  int koff[IPW], voff[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int inst = wave * IPW + i;
    const int row = inst * RPI + lane / C::CH;
    const int pos = lane % C::CH;
    koff[i] = row * D + C::swz_k(row, pos) * 8;
    voff[i] = row * D + C::swz_v(row, pos) * 8;
  }
  auto issue = [&](int t, int stage) {
    char* sk = smem + stage * 2 * C::TILE;
    char* sv = sk + C::TILE;
    const bf16* kt = Kb + (long)(kt0 + t) * C::KV * D;
    const bf16* vt = Vb + (long)(kt0 + t) * C::KV * D;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int inst = wave * IPW + i;
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(kt + koff[i]), DFOT_LDS_PTR(sk + inst * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(vt + voff[i]), DFOT_LDS_PTR(sv + inst * 1024), 16, 0, 0);
    }
  };

  f32x16 oacc[DV / 32];
#pragma unroll
  for (int i = 0; i < DV / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = 0.f, l_i = 0.f;

  const int nt = kt1 - kt0;
  // NST == 2: tile t+1 is loaded while t is computed (vmcnt(0) + barrier per tile).  NST == 3: tile t+2 stays in flight
  // across the barrier; the wait before the barrier is counted (2*IPW loads of the newest tile may be outstanding).
  issue(0, 0);
  if constexpr (NST == 3) {
    if (nt > 1) issue(1, 1);
    if (nt > 1) {
      if constexpr (IPW == 2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sk = smem + cur * 2 * C::TILE;
    const char* sv = sk + C::TILE;
    if constexpr (NST == 3) {
      if (t + 2 < nt) issue(t + 2, cur == 0 ? 2 : cur - 1);
    } else {
      if (t + 1 < nt) issue(t + 1, cur ^ 1);
    }

    // ---- S^T - m = K Q^T - m ----
    f32x16 sacc[2];
    if constexpr (QRELOAD) {
      const bf16* qp = qrow;
      asm volatile("" : "+v"(qp));  // opaque per iteration: keeps the loads inside the loop
#pragma unroll
      for (int ks = 0; ks < DQK / 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }
    if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt2][r] = -m_run;
      const int row = kt2 * 32 + lq;
#pragma unroll
      for (int ks = 0; ks < DQK / 16; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + row * C::ROWB + C::swz_k(row, ks * 2 + lh) * 16);
        sacc[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kt2], 0, 0, 0);
      }
    }
    if (prio) __builtin_amdgcn_s_setprio(0);

    // ---- row max of (s - m) over this lane's 32 keys and the partner half ----
    float mx = fmaxf(fmaxf(sacc[0][0], sacc[0][1]), sacc[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[0][r + 1]);
    mx = fmaxf(mx, sacc[0][15]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sacc[1][r]), sacc[1][r + 1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));

    // first tile: adopt the max outright; later: only when it grew by more than THR (wave-uniform branch)
    const bool grow = (t == 0) || (mx > THR);
    if (__any(grow)) {
      const float delta = (t == 0) ? mx : fmaxf(mx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-delta);  // t == 0: O and l are still zero
      m_run += delta;
      l_i *= alpha;
#pragma unroll
      for (int i = 0; i < DV / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kt2][r] -= delta;
    }

    float rs = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(sacc[kt2][8 * s + j]);
          rs += p;
          pf[kt2][s][j] = f2bf(p);
        }
    l_i += rs;

    // ---- O^T += V^T P^T ----
    // V^T fragments by ds_read_b64_tr_b16 through inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of the builtin form while an
    // LDS-DMA is in flight (it cannot see that the prefetched stage is another one), which drained the K/V ring at this point of every
    // tile.  The bank swizzle depends on q4 only, so one base address per lane and 32-column block; (kt2, s, +8) are immediate offsets.
    if (prio) __builtin_amdgcn_s_setprio(1);
    if constexpr (QRELOAD) {  // the 128-VGPR experiment keeps the builtin reads (the asm form's eight live results cost it spills)
#pragma unroll
      for (int dvt = 0; dvt < DV / 32; ++dvt)
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int kb = kt2 * 32 + 16 * s + 4 * lh;
            const int q4 = (lane & 15) >> 2, p4 = lane & 3;
            const int col = dvt * 32 + 16 * ((lane >> 4) & 1) + 4 * p4;
            const int r0 = kb + q4, r1 = kb + 8 + q4;
            const char* a0 = sv + r0 * C::ROWB + C::swz_v(r0, col >> 3) * 16 + (col & 7) * 2;
            const char* a1 = sv + r1 * C::ROWB + C::swz_v(r1, col >> 3) * 16 + (col & 7) * 2;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a1));
            const bf16x8 vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt2][s], oacc[dvt], 0, 0, 0);
          }
    } else {
  #pragma unroll
      for (int dvt = 0; dvt < DV / 32; ++dvt) {
        const int q4 = (lane & 15) >> 2, p4 = lane & 3;
        const int col = dvt * 32 + 16 * ((lane >> 4) & 1) + 4 * p4;
        const int r0 = 4 * lh + q4;
        const unsigned va = (unsigned)(size_t)DFOT_LDS_PTR(sv) + r0 * C::ROWB + C::swz_v(r0, col >> 3) * 16 + (col & 7) * 2;
        v2_u32x2 r000 = v2_read_tr16<0 * C::ROWB>(va), r001 = v2_read_tr16<8 * C::ROWB>(va);
        v2_u32x2 r010 = v2_read_tr16<16 * C::ROWB>(va), r011 = v2_read_tr16<24 * C::ROWB>(va);
        v2_u32x2 r100 = v2_read_tr16<32 * C::ROWB>(va), r101 = v2_read_tr16<40 * C::ROWB>(va);
        v2_u32x2 r110 = v2_read_tr16<48 * C::ROWB>(va), r111 = v2_read_tr16<56 * C::ROWB>(va);
        v2_lds_wait(r000, r001, r010, r011, r100, r101, r110, r111);
        oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2_bf16x8(r000, r001), pf[0][0], oacc[dvt], 0, 0, 0);
        oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2_bf16x8(r010, r011), pf[0][1], oacc[dvt], 0, 0, 0);
        oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2_bf16x8(r100, r101), pf[1][0], oacc[dvt], 0, 0, 0);
        oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2_bf16x8(r110, r111), pf[1][1], oacc[dvt], 0, 0, 0);
      }
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
    if constexpr (NST == 3) {
      if (t + 2 < nt) {
        if constexpr (IPW == 2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
      cur = cur == 2 ? 0 : cur + 1;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }

  const float l_tot = l_i + __shfl_xor(l_i, 32);
  const float inv = 1.0f / l_tot;
  const int b = bh / heads, hd = bh % heads;
  if (lse && lh == 0) lse[(long)bh * N + q0 + lq] = m_run + __log2f(l_tot);  // training: log2-domain log-sum-exp per query
  bf16* orow = O + ((long)b * N + q0 + lq) * ldo + hd * ohs;
#pragma unroll
  for (int dvt = 0; dvt < DV / 32; ++dvt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = f2bf(oacc[dvt][4 * g4 + j] * inv);
      const int c = dvt * 32 + 8 * g4 + 4 * lh;
      if (c < dvalid) *reinterpret_cast<bf16x4*>(orow + c) = o4;
    }
}

template <int D, int NST, int DQK = D, int DV = D, bool QRELOAD = false>
static int launch_attn_v2(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n,
                          hipStream_t stream, int ohs = D, int dvalid = D, float* lse = nullptr) {
  auto kern = attn_kernel_v2<D, NST, DQK, DV, QRELOAD>;
  constexpr int xcd_flag = 1 | 2;  // bit 0: XCD-aware tile order, bit 1: s_setprio around the MFMA clusters (+1-2 %): both always on
  const int lds = 2 * NST * AttnCfg<D>::TILE;
  static bool attr_set = false;
  if (!attr_set) {
    DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int tiles = (n / 128) * batch * heads;
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, stream, q, k, v, o, ldo, n, heads, xcd_flag, ohs, dvalid, lse);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

template <int D, bool TR>
static int launch_attn_t(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n,
                         hipStream_t stream) {
  auto kern = attn_kernel<D, TR>;
  const int lds = 4 * AttnCfg<D>::TILE;
  static bool attr_set = false;
  if (!attr_set) {
    DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(n / 128, batch * heads), dim3(256), lds, stream, q, k, v, o, ldo, n, heads);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

int launch_attention(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, int d,
                     int variant, hipStream_t stream, AttnScratch* scratch) {
  DFOT_REQUIRE(q && k && v && o, DFOT_ERR_ARG, "attention: null pointer");
  DFOT_REQUIRE(d == 64 || d == 128, DFOT_ERR_SHAPE, "attention: head dim %d not in {64,128}", d);
  DFOT_REQUIRE(n > 0 && n % 128 == 0, DFOT_ERR_SHAPE, "attention: N=%d must be a multiple of 128", n);
  DFOT_REQUIRE(ldo % 4 == 0, DFOT_ERR_SHAPE, "attention: output row stride %ld must be a multiple of 4", ldo);
  if ((variant == 5 || variant == 6) && d == 64 && n % 256 == 0)  // 64 query rows per wave, balanced tail; 6: no running max
    return launch_attention_v3(q, k, v, o, ldo, batch, heads, n, variant == 6, stream, scratch);
  if (variant == 14 && d == 64 && n % 256 == 0) return launch_attention_v5(q, k, v, o, ldo, batch, heads, n, stream, scratch);  // pipelined, no running max
  if (variant == 5 || variant == 6 || variant == 14) variant = 2;
  if (variant == 2 && attention_ks_applies(batch, heads, n, d)) return launch_attention_ks(q, k, v, o, ldo, batch, heads, n, stream, scratch);
  if (variant == 2) {  // tuned kernel; K/V ring depth chosen by measurement: 3 stages (48 KiB) at d = 64, 2 stages at d = 128
    return d == 64 ? launch_attn_v2<64, 3>(q, k, v, o, ldo, batch, heads, n, stream)
                   : launch_attn_v2<128, 2>(q, k, v, o, ldo, batch, heads, n, stream);
  }
  // A/B experiment kept for the record: two stages + Q fragments re-read per tile = 126 VGPRs, 4 waves per SIMD instead of 3.
  // Measured slower (580-780 vs 750-960 TF/s at N = 8192): occupancy is not what limits this kernel.
  if (variant == 4 && d == 64)
    return launch_attn_v2<64, 2, 64, 64, true>(q, k, v, o, ldo, batch, heads, n, stream);
  if (variant == 3) {  // tuned kernel, two stages for both head sizes (A/B reference)
    return d == 64 ? launch_attn_v2<64, 2>(q, k, v, o, ldo, batch, heads, n, stream)
                   : launch_attn_v2<128, 2>(q, k, v, o, ldo, batch, heads, n, stream);
  }
  if (d == 64) {
    return variant == 1 ? launch_attn_t<64, false>(q, k, v, o, ldo, batch, heads, n, stream)
                        : launch_attn_t<64, true>(q, k, v, o, ldo, batch, heads, n, stream);
  }
  return variant == 1 ? launch_attn_t<128, false>(q, k, v, o, ldo, batch, heads, n, stream)
                      : launch_attn_t<128, true>(q, k, v, o, ldo, batch, heads, n, stream);
}

// Attention over q/k/v stored [B][heads][N][dstride] with a logical head dim d <= dstride (dstride = 64 or 128, pad
// columns zero); the output is compact: O[row][head*d + c], c < d.  Used by the DiT blocks (d = 72).
int attention_dstride(int d) { return d <= 64 ? 64 : 128; }
int launch_attention_padded(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, int d,
                            hipStream_t stream, float* lse) {
  DFOT_REQUIRE(q && k && v && o, DFOT_ERR_ARG, "attention: null pointer");
  DFOT_REQUIRE(d > 0 && d <= 128 && d % 4 == 0, DFOT_ERR_SHAPE, "attention: head dim %d must be a multiple of 4, <= 128", d);
  DFOT_REQUIRE(n > 0 && n % 128 == 0, DFOT_ERR_SHAPE, "attention: N=%d must be a multiple of 128", n);
  DFOT_REQUIRE(ldo % 4 == 0, DFOT_ERR_SHAPE, "attention: output row stride %ld must be a multiple of 4", ldo);
  if (d <= 32) return launch_attn_v2<64, 3, 32, 32>(q, k, v, o, ldo, batch, heads, n, stream, d, d, lse);
  if (d <= 64) return launch_attn_v2<64, 3>(q, k, v, o, ldo, batch, heads, n, stream, d, d, lse);
  if (d <= 80) return launch_attn_v2<128, 2, 80, 96>(q, k, v, o, ldo, batch, heads, n, stream, d, d, lse);
  if (d <= 96) return launch_attn_v2<128, 2, 96, 96>(q, k, v, o, ldo, batch, heads, n, stream, d, d, lse);
  return launch_attn_v2<128, 2>(q, k, v, o, ldo, batch, heads, n, stream, d, d, lse);
}

}  // namespace dfot
