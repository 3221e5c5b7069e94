// Level-2 self-attention (d = 64), fourth structure: attention_v3.hip's kernel (64 query rows per wave, no running max, balanced
// tail) with the key loop SOFTWARE-PIPELINED at half-tile (32-key) granularity inside each wave.
//
// Why: PMC on attention_v3 (profiles/r02_f_*) shows MFMA and VALU work co-executing in only 26 % of the MFMA-busy cycles, and the
// 8-wave ping-pong experiment (attention_pp.hip) showed that an MFMA-only wave and a VALU-only wave on one SIMD do NOT overlap
// (5 % co-execution): the vector and matrix pipes overlap when ONE wave's stream interleaves them.  Within a tile the chain
// QK^T -> exp -> P.V is serial, so the independent work has to come from the neighbouring half-tiles:
//
//   iteration h (32 keys):   MFMA   S(h+1) = K(h+1) Q^T        and   O += V(h-1)^T P(h-1)      (16 MFMAs)
//                            VALU   P(h) = exp2(S(h)), row sums, bf16 pack                      (32 exp, 32 add, 16 pack per lane)
//
// with two S and two P register buffers.  The MFMAs of an iteration only need LDS fragments and values finished one iteration
// earlier, the VALU work only S(h) finished one iteration earlier: the scheduler is free to interleave them (sched_group_barrier
// asks for 1 MFMA : 4 VALU).  K/V tiles: 3-stage LDS-DMA ring as before; tile t-1's V is last read in iteration 2t and tile
// t+1's K first read in iteration 2t+1, so ONE barrier per tile between the two (behind vmcnt(0): only tile t+1 is in flight
// there) covers both hazards, and tile t+2 is issued right after it.
#include "common.h"
#include "dfot_hip.h"
#include "kernels.h"

#include <type_traits>

namespace dfot {

namespace {

constexpr int D = 64, KV = 64, ROWB = 128, TILE = KV * ROWB;
constexpr int QROWS = 256;

__device__ __forceinline__ int swz_k(int row, int c) { return c ^ ((row >> 1) & 7); }
__device__ __forceinline__ int swz_v(int row, int c) { return c ^ (((row >> 1) & 1) << 2); }

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// ds_read_b64_tr_b16 through inline asm: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the builtin form while an
// LDS-DMA is in flight (it cannot see that the prefetched stage is a different one), which drains the K/V ring every tile.
// The asm form is invisible to that pass; its completion is awaited by lds_wait() below, which passes the destination
// registers through the wait so that no consumer can be scheduled above it.
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr16(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void lds_wait(u32x2& a, u32x2& b, u32x2& c, u32x2& d, u32x2& e, u32x2& f, u32x2& g, u32x2& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x2 lo, u32x2 hi) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void lds_wait12(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x2& e, u32x2& f, u32x2& g, u32x2& h, u32x2& i,
                                           u32x2& j, u32x2& k, u32x2& l) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i), "+v"(j), "+v"(k), "+v"(l));
}
__device__ __forceinline__ void lds_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

template <int VPM, bool PIN>  // vector instructions requested per MFMA in the interleave; PIN: see iteration()
__global__ __launch_bounds__(256, 2) void attn64_kernel_v5(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                           const bf16* __restrict__ V, bf16* __restrict__ O, long ldo, int N,
                                                           int heads, int ohs, int full_tiles, int nsplit,
                                                           float* __restrict__ part_o, float* __restrict__ part_ml, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int qtiles = N / QROWS;
  const int ntk = N / KV;
  // work item: full tile (whole key range) or one key segment of a left-over tile
  int tile, t0, t1, seg = -1;
  if ((int)blockIdx.x < full_tiles) {
    tile = xcd_remap(blockIdx.x, full_tiles);
    t0 = 0;
    t1 = ntk;
  } else {
    const int nseg = gridDim.x - full_tiles;
    seg = xcd_remap(blockIdx.x - full_tiles, nseg);
    tile = full_tiles + seg / nsplit;
    const int c = seg % nsplit, per = ntk / nsplit;
    t0 = c * per;
    t1 = t0 + per;
    if (nsplit == 1) seg = -1;  // an unsplit left-over tile is a full tile
  }
  const int bh = tile / qtiles;
  const long base = (long)bh * N * D;
  const int q0 = (tile % qtiles) * QROWS + wave * 64;
  const bf16* Qb = Q + base;
  const bf16* Kb = K + base;
  const bf16* Vb = V + base;

  // Q fragments (B operand): lane holds Q[q0 + 32*qb + lq][16*ks + 8*lh + j]
  bf16x8 qf[2][D / 16];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks)
      qf[qb][ks] = *reinterpret_cast<const bf16x8*>(Qb + (long)(q0 + 32 * qb + lq) * D + ks * 16 + lh * 8);

  // per-lane DMA source offsets (elements) within a tile: LDS position (row, pos) receives source chunk swz(row, pos)
  int koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int inst = wave * 2 + i;
    const int row = inst * 8 + (lane >> 3), pos = lane & 7;
    koff[i] = row * D + swz_k(row, pos) * 8;
    voff[i] = row * D + swz_v(row, pos) * 8;
  }
  auto issue = [&](int t, int stage) {
    char* sk = smem + stage * 2 * TILE;
    char* sv = sk + TILE;
    const bf16* kt = Kb + (long)t * KV * D;
    const bf16* vt = Vb + (long)t * KV * D;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int inst = wave * 2 + i;
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(kt + koff[i]), DFOT_LDS_PTR(sk + inst * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(vt + voff[i]), DFOT_LDS_PTR(sv + inst * 1024), 16, 0, 0);
    }
  };

  f32x16 oacc[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
  float l_i[2] = {0.f, 0.f};

  const int nt = t1 - t0;
  issue(t0, 0);
  if (nt > 1) {
    issue(t0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // V^T fragments by transposed reads: the 16-lane group (lane>>4) reads rows kb + {0..3} (+8), columns dvt*32 + 16*(group&1)
  // + {0..15}; lane 4q+p of the group supplies row q, columns 4p..4p+3.  The bank swizzle depends on bit 1 of the row = bit 1
  // of q only, so one base address per lane and head-dim half; (kt2, s, +8) are immediate offsets.
  const int q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int vcol = 16 * ((lane >> 4) & 1) + 4 * p4;
  unsigned vaddr[2];
#pragma unroll
  for (int dvt = 0; dvt < 2; ++dvt) {
    const int col = dvt * 32 + vcol, r0 = 4 * lh + q4;
    vaddr[dvt] = (unsigned)(size_t)DFOT_LDS_PTR(smem) + TILE + r0 * ROWB + swz_v(r0, col >> 3) * 16 + (col & 7) * 2;
  }


  // K fragments (A operand of S^T): row kt2*32 + lq, chunk (2ks + lh) ^ ((lq>>1)&7): one base per lane, ks by XOR (the dynamic LDS
  // base is 0 here: no static LDS), kt2 = 1 is +32 rows
  const unsigned kbase = (unsigned)(size_t)DFOT_LDS_PTR(smem) + lq * ROWB + ((lh ^ ((lq >> 1) & 7)) * 16);

  f32x16 sacc[2][2];   // [query block][S buffer = half-tile parity]
  bf16x8 pf[2][2][2];  // [query block][P buffer][s]

  auto read_k = [&](auto kt2_c, int stage, u32x4 (&r)[4]) {
    constexpr int OFF = decltype(kt2_c)::value * 32 * ROWB;
    const unsigned kb = kbase + stage * (2 * TILE);
    r[0] = lds_read_b128<OFF>(kb), r[1] = lds_read_b128<OFF>(kb ^ 32), r[2] = lds_read_b128<OFF>(kb ^ 64), r[3] = lds_read_b128<OFF>(kb ^ 96);
  };
  // V^T fragments of the 32 keys kt2 of a tile, both head-dim halves: rows kt2*32 + 16*s (+8)
  auto read_v = [&](auto kt2_c, int stage, u32x2 (&r)[8]) {
    constexpr int OFF = decltype(kt2_c)::value * 32 * ROWB;
    const unsigned v0 = vaddr[0] + stage * (2 * TILE), v1 = vaddr[1] + stage * (2 * TILE);
    r[0] = lds_read_tr16<OFF>(v0), r[1] = lds_read_tr16<OFF + 8 * ROWB>(v0), r[2] = lds_read_tr16<OFF + 16 * ROWB>(v0);
    r[3] = lds_read_tr16<OFF + 24 * ROWB>(v0), r[4] = lds_read_tr16<OFF>(v1), r[5] = lds_read_tr16<OFF + 8 * ROWB>(v1);
    r[6] = lds_read_tr16<OFF + 16 * ROWB>(v1), r[7] = lds_read_tr16<OFF + 24 * ROWB>(v1);
  };
  auto qk_mfma = [&](auto buf_c, u32x4 (&r)[4]) {
    constexpr int buf = decltype(buf_c)::value;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int x = 0; x < 16; ++x) sacc[qb][buf][x] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = __builtin_bit_cast(bf16x8, r[ks]);
      sacc[0][buf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][ks], sacc[0][buf], 0, 0, 0);
      sacc[1][buf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][ks], sacc[1][buf], 0, 0, 0);
    }
  };
  auto pv_mfma = [&](auto buf_c, u32x2 (&r)[8]) {
    constexpr int buf = decltype(buf_c)::value;
#pragma unroll
    for (int dvt = 0; dvt < 2; ++dvt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 vf = as_bf16x8(r[4 * dvt + 2 * s], r[4 * dvt + 2 * s + 1]);
        oacc[0][dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[0][buf][s], oacc[0][dvt], 0, 0, 0);
        oacc[1][dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[1][buf][s], oacc[1][dvt], 0, 0, 0);
      }
  };
  // P(h) = exp2(S(h)) for the 32 keys of S buffer `buf`, part `part` (s = part): 16 scores per lane and query block
  auto softmax_part = [&](auto buf_c, auto s_c) {
    constexpr int buf = decltype(buf_c)::value, s = decltype(s_c)::value;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float p = __builtin_amdgcn_exp2f(sacc[qb][buf][8 * s + j]);
        acc += p;
        pf[qb][buf][s][j] = f2bf(p);
      }
      l_i[qb] += acc;
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  // one iteration: S buffer `cb` holds S(h); builds P(h) in P buffer cb, S(h+1) in S buffer 1-cb, adds P(h-1) V(h-1) from P buffer 1-cb
  auto iteration = [&](auto cb_c, auto next_c, int kstage, auto prev_c, int vstage) {
    constexpr int cb = decltype(cb_c)::value;
    constexpr bool has_next = decltype(next_c)::value, has_prev = decltype(prev_c)::value;
    using NB = std::integral_constant<int, 1 - cb>;
    using KT2N = std::integral_constant<int, 1 - cb>;  // half-tile h+1 has key parity 1 - (h & 1); h & 1 == cb by construction
    u32x4 kr[4];
    u32x2 vr[8];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (has_next) read_k(KT2N{}, kstage, kr);
    if constexpr (has_prev) read_v(KT2N{}, vstage, vr);  // half-tile h-1 has the same key parity as h+1
    softmax_part(cb_c, C0{});                            // vector work that needs no LDS data covers the read latency
    // ... provided it stays in front of the wait: without a use there LLVM sinks it behind the wait and the wave idles through the
    // latency of its twelve LDS reads at the top of every iteration (PIN = false restores that order: 335 vs 329 us, no loss once the
    // partner wave covers the latency, but 256 instead of 217 VGPRs)
    if constexpr (PIN) asm volatile("" : "+v"(pf[0][cb][0]), "+v"(pf[1][cb][0]), "+v"(l_i[0]), "+v"(l_i[1]));
    if constexpr (has_next && has_prev) lds_wait12(kr[0], kr[1], kr[2], kr[3], vr[0], vr[1], vr[2], vr[3], vr[4], vr[5], vr[6], vr[7]);
    else if constexpr (has_next) lds_wait4(kr[0], kr[1], kr[2], kr[3]);
    else if constexpr (has_prev) lds_wait(vr[0], vr[1], vr[2], vr[3], vr[4], vr[5], vr[6], vr[7]);
    if constexpr (has_next) qk_mfma(NB{}, kr);
    softmax_part(cb_c, C1{});
    if constexpr (has_prev) pv_mfma(NB{}, vr);
    // ask for a fine interleave of the two pipes in this region: 1 MFMA, then vector instructions (exp / add / pack)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, VPM, 0);
    }
    // pin this iteration's P(h) here: without a use in this block LLVM sinks the exp / pack work past the barrier into the
    // next iteration (its only consumer), which serialises it behind that iteration's MFMAs again
    asm volatile("" : "+v"(pf[0][cb][0]), "+v"(pf[0][cb][1]), "+v"(pf[1][cb][0]), "+v"(pf[1][cb][1]), "+v"(l_i[0]), "+v"(l_i[1]));
    __builtin_amdgcn_sched_barrier(0);
  };
  using Yes = std::true_type;
  using No = std::false_type;

  // prologue: S(0) = K(tile 0, keys 0..31) Q^T
  {
    u32x4 kr[4];
    read_k(C0{}, 0, kr);
    lds_wait4(kr[0], kr[1], kr[2], kr[3]);
    qk_mfma(C0{}, kr);
  }
  int cur = 0;  // ring stage of tile t
  // one key tile = two iterations; PREV1: the first has a previous half-tile (t > 0); NEXT2: the second has a next one (t + 1 < nt)
  auto kv_tile = [&](int t, auto prev1_c, auto next2_c) {
    const int prev = cur == 0 ? 2 : cur - 1, nxt = cur == 2 ? 0 : cur + 1;
    // h = 2t: S(h) in buffer 0; next half-tile = (t, keys 32..63); previous = (t-1, keys 32..63)
    iteration(C0{}, Yes{}, cur, prev1_c, prev);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // tile t+1 landed everywhere; tile t-1 no longer read
    if (t + 2 < nt) issue(t0 + t + 2, prev);
    // h = 2t+1: S(h) in buffer 1; next = (t+1, keys 0..31); previous = (t, keys 0..31)
    iteration(C1{}, next2_c, nxt, Yes{}, cur);
    cur = nxt;
  };
  if (nt == 1) {
    kv_tile(0, No{}, No{});
  } else {
    kv_tile(0, No{}, Yes{});
    for (int t = 1; t + 1 < nt; ++t) kv_tile(t, Yes{}, Yes{});
    kv_tile(nt - 1, Yes{}, No{});
  }
  // epilogue: O += V(last half-tile) P(last)
  {
    const int last = cur == 0 ? 2 : cur - 1;
    u32x2 vr[8];
    read_v(C1{}, last, vr);
    lds_wait(vr[0], vr[1], vr[2], vr[3], vr[4], vr[5], vr[6], vr[7]);
    pv_mfma(C1{}, vr);
  }

  // ---- epilogue: lane holds O[q0 + 32*qb + lq][dvt*32 + 8*g + 4*lh + {0..3}] in oacc[qb][dvt][4g..4g+3] ----
  const int b = bh / heads, hd = bh % heads;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = l_i[qb] + __shfl_xor(l_i[qb], 32);
    const int rloc = wave * 64 + 32 * qb + lq;
    if (seg < 0) {
      const float inv = 1.0f / l_tot;
      if (lse && lh == 0) lse[(long)bh * N + (tile % qtiles) * QROWS + rloc] = __log2f(l_tot);  // training: no running max, so lse = log2 l
      bf16* orow = O + ((long)b * N + (tile % qtiles) * QROWS + rloc) * ldo + hd * ohs;
#pragma unroll
      for (int dvt = 0; dvt < 2; ++dvt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 o4;
#pragma unroll
          for (int j = 0; j < 4; ++j) o4[j] = f2bf(oacc[qb][dvt][4 * g4 + j] * inv);
          *reinterpret_cast<bf16x4*>(orow + dvt * 32 + 8 * g4 + 4 * lh) = o4;
        }
    } else {
      float* prow = part_o + ((long)seg * QROWS + rloc) * D;
#pragma unroll
      for (int dvt = 0; dvt < 2; ++dvt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 o4;
#pragma unroll
          for (int j = 0; j < 4; ++j) o4[j] = oacc[qb][dvt][4 * g4 + j];
          *reinterpret_cast<f32x4*>(prow + dvt * 32 + 8 * g4 + 4 * lh) = o4;
        }
      if (lh == 0) {
        float2 ml = make_float2(0.f, l_tot);
        *reinterpret_cast<float2*>(part_ml + ((long)seg * QROWS + rloc) * 2) = ml;
      }
    }
  }
}

}  // namespace

int launch_attention_v5(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, hipStream_t stream,
                        AttnScratch* scratch, float* lse) {
  DFOT_REQUIRE(q && k && v && o, DFOT_ERR_ARG, "attention: null pointer");
  DFOT_REQUIRE(n > 0 && n % QROWS == 0, DFOT_ERR_SHAPE, "attention v5: N=%d must be a multiple of %d", n, QROWS);
  DFOT_REQUIRE(ldo % 4 == 0, DFOT_ERR_SHAPE, "attention: output row stride %ld must be a multiple of 4", ldo);
  const AttnSplit sp = attn_plan_split(batch, heads, n, QROWS, 2);
  float *po = nullptr, *pml = nullptr;
  int rc = attn_partials(sp, QROWS, &po, &pml, scratch);
  if (rc) return rc;
  const int lds = 2 * 3 * TILE;
  auto go = [&](auto kern) -> int {
    static bool attr_set = false;
    if (!attr_set) {
      DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(sp.full + sp.rem * sp.nsplit), dim3(256), lds, stream, q, k, v, o, ldo, n, heads, D, sp.full, sp.nsplit, po, pml, lse);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  };
  rc = go(attn64_kernel_v5<3, true>);
  if (rc) return rc;
  return attn_launch_merge(sp, QROWS, po, pml, o, ldo, n, heads, stream, lse);
}

}  // namespace dfot
