// Level-2 self-attention of the DFoT transformer blocks (u_vit_blocks.py:254-268), third structure: d = 64, N = T*32*32.
//
// Same math and LDS images as attn_kernel_v2 (attention.hip): S^T = K Q^T and O^T = V^T P^T with the query on the MFMA
// lane, K/V tiles of 64 keys streamed global -> LDS by LDS-DMA through a 3-stage ring, scores in the exp2 domain.
// What changes:
//  * a wave owns 64 query rows = TWO 32-row blocks: every K fragment (ds_read_b128) and every V^T fragment (two
//    ds_read_b64_tr_b16) feeds two MFMAs, so LDS reads, DMA issues and barriers per MFMA halve, and the two blocks'
//    independent softmax / MFMA chains give the scheduler VALU work to place beside MFMAs of the same wave;
//  * NOMAX: the caller guarantees |score| <= bound (the DFoT blocks RMS-normalise q and k per head, u_vit_blocks.py:257-259,
//    so |q.k|/sqrt(d) <= sqrt(d) max|w_q| max|w_k|): softmax is shift invariant, so exp2(s) is used as is -- no running
//    max, no accumulator preset, no rescale branch (the loop body is one basic block), and partial results over key
//    ranges add up without rescaling;
//  * balanced tail: with S = 2 workgroups per CU resident, T query tiles run as floor(T/S) full rounds; the T mod S
//    left-over tiles are split over the key axis into `nsplit` segments each (rem*nsplit <= S), so the last round is as
//    full as the others and 1/nsplit as long.  Segments write fp32 partial (O, m, l); attn64_merge_kernel combines them.
#include "common.h"
#include "dfot_hip.h"
#include "kernels.h"

namespace dfot {

namespace {

constexpr int D = 64, KV = 64, ROWB = 128, TILE = KV * ROWB;   // one K (or V) tile: 64 rows x 128 B = 8 KiB
constexpr int QROWS = 256;                                     // query rows per workgroup (4 waves x 64)
constexpr float THR = 8.0f;

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

template <int NST, bool NOMAX>
__global__ __launch_bounds__(256, 2) void attn64_kernel_v3(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                           const bf16* __restrict__ V, bf16* __restrict__ O, long ldo, int N,
                                                           int heads, int ohs, int full_tiles, int nsplit,
                                                           float* __restrict__ part_o, float* __restrict__ part_ml) {
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
  float m_run[2] = {0.f, 0.f}, l_i[2] = {0.f, 0.f};

  const int nt = t1 - t0;
  static_assert(NST == 3, "three-stage ring");
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

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sk = smem + cur * 2 * TILE;
    const char* sv = sk + TILE;
    if (t + 2 < nt) issue(t0 + t + 2, cur == 0 ? 2 : cur - 1);

    // ---- S^T (- m) = K Q^T (- m): two 32-key sub-tiles x two query blocks, every K fragment used twice ----
    f32x16 sacc[2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[qb][kt2][r] = NOMAX ? 0.f : -m_run[qb];
      const int row = kt2 * 32 + lq;
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + row * ROWB + swz_k(row, ks * 2 + lh) * 16);
        sacc[0][kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][ks], sacc[0][kt2], 0, 0, 0);
        sacc[1][kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][ks], sacc[1][kt2], 0, 0, 0);
      }
    }

    if constexpr (!NOMAX) {
      float mx[2];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float m = fmaxf(fmaxf(sacc[qb][0][0], sacc[qb][0][1]), sacc[qb][0][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, sacc[qb][0][r]), sacc[qb][0][r + 1]);
        m = fmaxf(m, sacc[qb][0][15]);
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = fmaxf(fmaxf(m, sacc[qb][1][r]), sacc[qb][1][r + 1]);
        mx[qb] = fmaxf(m, __shfl_xor(m, 32));
      }
      // first tile: adopt the max outright; later: only when it grew by more than THR (wave-uniform branch)
      const bool grow = (t == 0) || (mx[0] > THR) || (mx[1] > THR);
      if (__any(grow)) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const float delta = (t == 0) ? mx[qb] : fmaxf(mx[qb], 0.f);
          const float alpha = __builtin_amdgcn_exp2f(-delta);  // t == 0: O and l are still zero
          m_run[qb] += delta;
          l_i[qb] *= alpha;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
#pragma unroll
          for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[qb][kt2][r] -= delta;
        }
      }
    }

    // ---- P = exp2(S), row sums, bf16 P^T fragments (B operand of the second product) ----
    bf16x8 pf[2][2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float rs[2][2];
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float p = __builtin_amdgcn_exp2f(sacc[qb][kt2][8 * s + j]);
            acc += p;
            pf[qb][kt2][s][j] = f2bf(p);
          }
          rs[kt2][s] = acc;
        }
      l_i[qb] += (rs[0][0] + rs[0][1]) + (rs[1][0] + rs[1][1]);
    }

    // ---- O^T += V^T P^T: every V^T fragment used for both query blocks ----
#pragma unroll
    for (int dvt = 0; dvt < 2; ++dvt) {
      const unsigned va = vaddr[dvt] + cur * (2 * TILE);
      // rows kt2*32 + 16*s (+8): byte offsets (kt2*32 + 16*s + 8*h) * ROWB
      u32x2 r000 = lds_read_tr16<0 * ROWB>(va), r001 = lds_read_tr16<8 * ROWB>(va);
      u32x2 r010 = lds_read_tr16<16 * ROWB>(va), r011 = lds_read_tr16<24 * ROWB>(va);
      u32x2 r100 = lds_read_tr16<32 * ROWB>(va), r101 = lds_read_tr16<40 * ROWB>(va);
      u32x2 r110 = lds_read_tr16<48 * ROWB>(va), r111 = lds_read_tr16<56 * ROWB>(va);
      lds_wait(r000, r001, r010, r011, r100, r101, r110, r111);
      const bf16x8 vf[2][2] = {{as_bf16x8(r000, r001), as_bf16x8(r010, r011)}, {as_bf16x8(r100, r101), as_bf16x8(r110, r111)}};
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          oacc[0][dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt2][s], pf[0][kt2][s], oacc[0][dvt], 0, 0, 0);
          oacc[1][dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt2][s], pf[1][kt2][s], oacc[1][dvt], 0, 0, 0);
        }
    }

    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    cur = cur == 2 ? 0 : cur + 1;
  }

  // ---- epilogue: lane holds O[q0 + 32*qb + lq][dvt*32 + 8*g + 4*lh + {0..3}] in oacc[qb][dvt][4g..4g+3] ----
  const int b = bh / heads, hd = bh % heads;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = l_i[qb] + __shfl_xor(l_i[qb], 32);
    const int rloc = wave * 64 + 32 * qb + lq;
    if (seg < 0) {
      const float inv = 1.0f / l_tot;
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
        float2 ml = make_float2(m_run[qb], l_tot);
        *reinterpret_cast<float2*>(part_ml + ((long)seg * QROWS + rloc) * 2) = ml;
      }
    }
  }
}

// combine the key segments of the left-over tiles: O = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s.  One thread per
// (query row, 4 columns); 16 threads per row.
__global__ __launch_bounds__(256) void attn64_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                           bf16* __restrict__ O, long ldo, int N, int heads, int ohs,
                                                           int full_tiles, int nsplit, int rem_tiles, int qrows, float* __restrict__ lse) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = gid >> 4;
  const int c4 = (int)(gid & 15) * 4;
  if (row >= (long)rem_tiles * qrows) return;
  const int lt = (int)(row / qrows), rloc = (int)(row % qrows);
  const int qtiles = N / qrows;
  float mmax = -INFINITY;
  for (int s = 0; s < nsplit; ++s) mmax = fmaxf(mmax, part_ml[((long)(lt * nsplit + s) * qrows + rloc) * 2]);
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const long pr = (long)(lt * nsplit + s) * qrows + rloc;
    const float w = exp2f(part_ml[pr * 2] - mmax);
    l += w * part_ml[pr * 2 + 1];
    const f32x4 o = *reinterpret_cast<const f32x4*>(part_o + pr * D + c4);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += w * o[j];
  }
  const float inv = 1.0f / l;
  const int tile = full_tiles + lt;
  const int bh = tile / qtiles, b = bh / heads, hd = bh % heads;
  bf16x4 o4;
#pragma unroll
  for (int j = 0; j < 4; ++j) o4[j] = f2bf(acc[j] * inv);
  *reinterpret_cast<bf16x4*>(O + ((long)b * N + (tile % qtiles) * qrows + rloc) * ldo + hd * ohs + c4) = o4;
  if (lse && c4 == 0) lse[(long)bh * N + (tile % qtiles) * qrows + rloc] = mmax + __log2f(l);  // training: log2-domain log-sum-exp of the row
}

}  // namespace

// wgs_per_cu workgroups of qrows query rows are resident per CU (registers: 2 waves per SIMD)
AttnSplit attn_plan_split(int batch, int heads, int n, int qrows, int wgs_per_cu) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  }
  const int slots = wgs_per_cu * cus;
  AttnSplit sp;
  sp.slots = slots;
  sp.tiles = batch * heads * (n / qrows);
  sp.rem = sp.tiles % slots;
  sp.full = sp.tiles - sp.rem;
  sp.nsplit = 1;
  if (sp.rem) {
    const int ntk = n / KV;
    for (int f = 2; f <= 16 && sp.rem * f <= slots; f *= 2)
      if (ntk % f == 0 && ntk / f >= 4) sp.nsplit = f;
  }
  return sp;
}

// Process-wide scratch of the op-level entry points (dfot_op_attention, tools): it only ever GROWS by allocating a new block; the
// blocks it outgrows are kept until the process ends, so a kernel in flight or a captured graph that holds an old pointer stays
// valid (nothing is freed or synchronised on a launch path).  Backbone handles do not use it: they own an AttnScratch sized in
// their reserve() and pass it in, so two handles / streams never share partial rows and a reserve on one model cannot pull the
// buffer from under another model's captured graph.
AttnScratch* attention_default_scratch() {
  static AttnScratch g;
  return &g;
}

size_t attention_scratch_bytes(int batch, int heads, int n, int d) {
  if (d == 128) return attention_ks_scratch_bytes(batch, heads, n, d);  // the 8-wave key-split kernel of attention_ks.hip
  if (d != 64 || n % QROWS != 0) return 0;  // d = 64: the 64-rows-per-wave level-2 kernels (attention_v3 / v5) split their tail
  const AttnSplit sp = attn_plan_split(batch, heads, n, QROWS, 2);
  return sp.nsplit == 1 ? 0 : (size_t)sp.rem * sp.nsplit * QROWS * (D + 2) * sizeof(float);
}

int attn_partials(const AttnSplit& sp, int qrows, float** po, float** pml, AttnScratch* scratch, int dcols) {
  *po = *pml = nullptr;
  if (sp.nsplit == 1) return DFOT_OK;
  const size_t rows = (size_t)sp.rem * sp.nsplit * qrows;
  const size_t bytes = rows * (dcols + 2) * sizeof(float);
  if (!scratch) {
    scratch = attention_default_scratch();
    if (bytes > scratch->bytes) {  // grow: a NEW block; the old one is deliberately leaked (see above)
      void* p = nullptr;
      DFOT_CHECK_HIP(hipMalloc(&p, bytes));
      scratch->p = reinterpret_cast<float*>(p);
      scratch->bytes = bytes;
    }
  }
  DFOT_REQUIRE(scratch->p && bytes <= scratch->bytes, DFOT_ERR_STATE,
               "attention: key-split scratch of %zu bytes, launch needs %zu (reserve the handle for this batch first)", scratch->bytes, bytes);
  *po = scratch->p;
  *pml = scratch->p + rows * dcols;
  return DFOT_OK;
}

int attn_launch_merge(const AttnSplit& sp, int qrows, const float* po, const float* pml, bf16* o, long ldo, int n, int heads,
                      hipStream_t stream, float* lse) {
  if (sp.nsplit == 1) return DFOT_OK;
  const long threads = (long)sp.rem * qrows * 16;
  hipLaunchKernelGGL(attn64_merge_kernel, dim3(cdiv(threads, 256)), dim3(256), 0, stream, po, pml, o, ldo, n, heads, D, sp.full,
                     sp.nsplit, sp.rem, qrows, lse);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// q, k, v: [B][heads][N][64] bf16, q pre-scaled by log2(e)/sqrt(d); o: row r of batch b, head hd at o[(b*N + r)*ldo + hd*64].
// nomax: the caller guarantees bounded scores (see the header of this file).
int launch_attention_v3(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, bool nomax,
                        hipStream_t stream, AttnScratch* scratch) {
  DFOT_REQUIRE(q && k && v && o, DFOT_ERR_ARG, "attention: null pointer");
  DFOT_REQUIRE(n > 0 && n % QROWS == 0, DFOT_ERR_SHAPE, "attention v3: N=%d must be a multiple of %d", n, QROWS);
  DFOT_REQUIRE(ldo % 4 == 0, DFOT_ERR_SHAPE, "attention: output row stride %ld must be a multiple of 4", ldo);
  const AttnSplit sp = attn_plan_split(batch, heads, n, QROWS, 2);
  float *po = nullptr, *pml = nullptr;
  int rc0 = attn_partials(sp, QROWS, &po, &pml, scratch);
  if (rc0) return rc0;
  const int lds = 2 * 3 * TILE;
  const int grid = sp.full + sp.rem * sp.nsplit;
  auto go = [&](auto kern) -> int {
    static bool attr_set = false;
    if (!attr_set) {
      DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, q, k, v, o, ldo, n, heads, D, sp.full, sp.nsplit, po, pml);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  };
  int rc = nomax ? go(attn64_kernel_v3<3, true>) : go(attn64_kernel_v3<3, false>);
  if (rc) return rc;
  return attn_launch_merge(sp, QROWS, po, pml, o, ldo, n, heads, stream);
}

}  // namespace dfot
