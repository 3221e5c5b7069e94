// Flash attention forward for head dims 65..128 stored in 128-element rows: 64 query rows per wave (gfx950).
//
// The d = 128 instance of the restructuring attention_v3.hip applies at d = 64 to `attn_kernel_v2` (attention.hip; same math, same
// LDS images of the K / V tiles, same transposed S^T = K Q^T / O^T = V^T P^T products):
//  * a wave owns 64 query rows = two 32-row blocks, so every K fragment (ds_read_b128) and every V^T fragment (2 x
//    ds_read_b64_tr_b16) feeds two MFMAs: half the LDS bytes per FLOP of the 32-rows-per-wave kernel, which is LDS-read bound at
//    d = 128 (one workgroup per CU: 29 % of the CU's MFMA peak, two: 36 %);
//  * DQK / DV: head dims actually multiplied (DiT: d = 72 -> DQK 80, DV 96; the pad columns of the 128-element rows hold zeros);
//  * 256-row query tiles (4 waves), three-stage K/V ring (96 KiB of LDS: one workgroup per CU, one wave per SIMD, ~300 VGPRs);
//  * balanced tail as attention_v3.hip: T query tiles on S = #CU resident workgroups run floor(T/S) full rounds, the T mod S
//    left-over tiles are split over the key axis into `nsplit` segments (fp32 partials + attn_merge_kernel_d);
//  * running max with the deferred rescale of attn_kernel_v2 (the DiT blocks have no QK-norm: scores are unbounded);
//  * optional log-sum-exp output (training).
// MEASURED SLOWER than attn_kernel_v2<128> and therefore OFF by default (DFOT_ATTN_ROWS64_D128=1 selects it where the launch has at
// least one full round of 256-row tiles): 225 vs 207 us at B*H = 72, N = 2048, d = 128; Kinetics-600 DiT3D 60.3 vs 63.1 latent
// frames/s.  With one wave per SIMD nothing overlaps the MFMA phases with the softmax phases of the same wave; the two resident
// 128-row workgroups of attn_kernel_v2 do.  What this kernel needs is the software-pipelined key loop of attention_v5.hip (or
// <= 256 VGPRs: d = 72 is at 285).  Kept with its parity test as the starting point for that.
#include "common.h"
#include "dfot_hip.h"
#include "kernels.h"

namespace dfot {

namespace {

constexpr int D = 128, KV = 64, ROWB = 256, TILE = KV * ROWB;  // one K (or V) tile: 64 rows x 256 B = 16 KiB
constexpr int QROWS = 256;                                     // query rows per workgroup (4 waves x 64)
constexpr float THR = 8.0f;

__device__ __forceinline__ int swz_k(int row, int c) { return c ^ (row & 15); }
__device__ __forceinline__ int swz_v(int row, int c) { return c ^ ((row & 3) << 2); }

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// ds_read_b64_tr_b16 through inline asm (see attention_v3.hip: the builtin makes hipcc drain the LDS-DMA queue)
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

template <int DQK, int DV>
__global__ __launch_bounds__(256, 1) void attn128_kernel_v3(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                            bf16* __restrict__ O, long ldo, int N, int heads, int ohs, int dvalid, int full_tiles,
                                                            int nsplit, float* __restrict__ part_o, float* __restrict__ part_ml,
                                                            float* __restrict__ lse) {
  static_assert(DQK % 16 == 0 && DV % 32 == 0 && DQK <= D && DV <= D, "head-dim sub-range");
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
  bf16x8 qf[2][DQK / 16];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < DQK / 16; ++ks)
      qf[qb][ks] = *reinterpret_cast<const bf16x8*>(Qb + (long)(q0 + 32 * qb + lq) * D + ks * 16 + lh * 8);

  // per-lane DMA source offsets (elements) within a tile: a 1-KiB instruction covers 4 rows of 16 chunks; LDS position (row, pos)
  // receives source chunk swz(row, pos).  Four instructions per wave and operand.
  int koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int inst = wave * 4 + i;
    const int row = inst * 4 + (lane >> 4), pos = lane & 15;
    koff[i] = row * D + swz_k(row, pos) * 8;
    voff[i] = row * D + swz_v(row, pos) * 8;
  }
  auto issue = [&](int t, int stage) {
    char* sk = smem + stage * 2 * TILE;
    char* sv = sk + TILE;
    const bf16* kt = Kb + (long)t * KV * D;
    const bf16* vt = Vb + (long)t * KV * D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int inst = wave * 4 + i;
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(kt + koff[i]), DFOT_LDS_PTR(sk + inst * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(vt + voff[i]), DFOT_LDS_PTR(sv + inst * 1024), 16, 0, 0);
    }
  };

  f32x16 oacc[2][DV / 32];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < DV / 32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
  float m_run[2] = {0.f, 0.f}, l_i[2] = {0.f, 0.f};

  const int nt = t1 - t0;
  // three-stage ring: tile t + 2 stays in flight across the barrier (8 DMA instructions per wave and tile)
  issue(t0, 0);
  if (nt > 1) {
    issue(t0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // V^T fragments by transposed reads: the 16-lane group (lane>>4) reads rows kb + {0..3} (+8), columns dvt*32 + 16*(group&1) +
  // {0..15}; lane 4q+p of the group supplies row q, columns 4p..4p+3.  The bank swizzle depends on row & 3 = q only, so one base
  // address per lane and 32-column block; (kt2, s, +8) are immediate offsets.
  const int q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int vcol = 16 * ((lane >> 4) & 1) + 4 * p4;
  unsigned vaddr[DV / 32];
#pragma unroll
  for (int dvt = 0; dvt < DV / 32; ++dvt) {
    const int col = dvt * 32 + vcol, r0 = 4 * lh + q4;
    vaddr[dvt] = (unsigned)(size_t)DFOT_LDS_PTR(smem) + TILE + r0 * ROWB + swz_v(r0, col >> 3) * 16 + (col & 7) * 2;
  }

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const char* sk = smem + cur * 2 * TILE;
    if (t + 2 < nt) issue(t0 + t + 2, cur == 0 ? 2 : cur - 1);

    // ---- S^T - m = K Q^T - m: two 32-key sub-tiles x two query blocks, every K fragment used twice ----
    f32x16 sacc[2][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[qb][kt2][r] = -m_run[qb];
      const int row = kt2 * 32 + lq;
#pragma unroll
      for (int ks = 0; ks < DQK / 16; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + row * ROWB + swz_k(row, ks * 2 + lh) * 16);
        sacc[0][kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][ks], sacc[0][kt2], 0, 0, 0);
        sacc[1][kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][ks], sacc[1][kt2], 0, 0, 0);
      }
    }

    {
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
          for (int i = 0; i < DV / 32; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
#pragma unroll
          for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[qb][kt2][r] -= delta;
        }
      }
    }

    // ---- P = exp2(S - m), row sums, bf16 P^T fragments (B operand of the second product) ----
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
    for (int dvt = 0; dvt < DV / 32; ++dvt) {
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

    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    cur = cur == 2 ? 0 : cur + 1;
  }

  // ---- epilogue: lane holds O[q0 + 32*qb + lq][dvt*32 + 8*g + 4*lh + {0..3}] in oacc[qb][dvt][4g..4g+3] ----
  const int b = bh / heads, hd = bh % heads;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = l_i[qb] + __shfl_xor(l_i[qb], 32);
    const int rloc = wave * 64 + 32 * qb + lq;
    const int qrow = (tile % qtiles) * QROWS + rloc;
    if (seg < 0) {
      const float inv = 1.0f / l_tot;
      if (lse && lh == 0) lse[(long)bh * N + qrow] = m_run[qb] + __log2f(l_tot);  // training: log2-domain log-sum-exp per query
      bf16* orow = O + ((long)b * N + qrow) * ldo + hd * ohs;
#pragma unroll
      for (int dvt = 0; dvt < DV / 32; ++dvt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 o4;
#pragma unroll
          for (int j = 0; j < 4; ++j) o4[j] = f2bf(oacc[qb][dvt][4 * g4 + j] * inv);
          const int c = dvt * 32 + 8 * g4 + 4 * lh;
          if (c < dvalid) *reinterpret_cast<bf16x4*>(orow + c) = o4;
        }
    } else {
      float* prow = part_o + ((long)seg * QROWS + rloc) * D;
#pragma unroll
      for (int dvt = 0; dvt < DV / 32; ++dvt)
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

// combine the key segments of the left-over tiles: O = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s.  One thread per (query row,
// 4 columns); 32 threads per row (partial rows are 128 floats, columns >= dvalid are neither read nor written).
__global__ __launch_bounds__(256) void attn128_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                            bf16* __restrict__ O, long ldo, int N, int heads, int ohs, int dvalid,
                                                            int full_tiles, int nsplit, int rem_tiles, float* __restrict__ lse) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = gid >> 5;
  const int c4 = (int)(gid & 31) * 4;
  if (row >= (long)rem_tiles * QROWS) return;
  const int lt = (int)(row / QROWS), rloc = (int)(row % QROWS);
  const int qtiles = N / QROWS;
  float mmax = -INFINITY;
  for (int s = 0; s < nsplit; ++s) mmax = fmaxf(mmax, part_ml[((long)(lt * nsplit + s) * QROWS + rloc) * 2]);
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const long pr = (long)(lt * nsplit + s) * QROWS + rloc;
    const float w = exp2f(part_ml[pr * 2] - mmax);
    l += w * part_ml[pr * 2 + 1];
    if (c4 < dvalid) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(part_o + pr * D + c4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += w * o[j];
    }
  }
  const float inv = 1.0f / l;
  const int tile = full_tiles + lt;
  const int bh = tile / qtiles, b = bh / heads, hd = bh % heads;
  const int qrow = (tile % qtiles) * QROWS + rloc;
  if (lse && c4 == 0) lse[(long)bh * N + qrow] = mmax + __log2f(l);
  if (c4 >= dvalid) return;
  bf16x4 o4;
#pragma unroll
  for (int j = 0; j < 4; ++j) o4[j] = f2bf(acc[j] * inv);
  *reinterpret_cast<bf16x4*>(O + ((long)b * N + qrow) * ldo + hd * ohs + c4) = o4;
}

template <int DQK, int DV>
int launch_t(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, int ohs, int dvalid, float* lse,
             hipStream_t stream) {
  const AttnSplit sp = attn_plan_split(batch, heads, n, QROWS, 1);
  float *po = nullptr, *pml = nullptr;
  int rc = attn_partials(sp, QROWS, &po, &pml, D);
  if (rc) return rc;
  constexpr int lds = 2 * 3 * TILE;
  auto kern = attn128_kernel_v3<DQK, DV>;
  static bool attr_set = false;
  if (!attr_set) {
    DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int grid = sp.full + sp.rem * sp.nsplit;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, q, k, v, o, ldo, n, heads, ohs, dvalid, sp.full, sp.nsplit, po, pml, lse);
  DFOT_CHECK_HIP(hipGetLastError());
  if (sp.nsplit > 1) {
    const long threads = (long)sp.rem * QROWS * 32;
    hipLaunchKernelGGL(attn128_merge_kernel, dim3(cdiv(threads, 256)), dim3(256), 0, stream, po, pml, o, ldo, n, heads, ohs, dvalid, sp.full,
                       sp.nsplit, sp.rem, lse);
    DFOT_CHECK_HIP(hipGetLastError());
  }
  return DFOT_OK;
}

}  // namespace

// q, k, v: [B][heads][N][128] bf16 (columns >= d zero), q pre-scaled by log2(e)/sqrt(d); head hd of query row r goes to
// o[(b*N + r)*ldo + hd*d + c], c < d; lse optional [B][heads][N] (log2 domain).  N a multiple of 256, 64 < d <= 128, d % 4 == 0.
int launch_attention_rows64_d128(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, int d, float* lse,
                                 hipStream_t stream) {
  DFOT_REQUIRE(q && k && v && o, DFOT_ERR_ARG, "attention: null pointer");
  DFOT_REQUIRE(n > 0 && n % QROWS == 0 && d > 64 && d <= 128 && d % 4 == 0 && ldo % 4 == 0, DFOT_ERR_SHAPE,
               "attention (64 rows per wave, 128-element rows): N=%d must be a multiple of %d, 64 < d=%d <= 128", n, QROWS, d);
  if (d <= 80) return launch_t<80, 96>(q, k, v, o, ldo, batch, heads, n, d, d, lse, stream);
  if (d <= 96) return launch_t<96, 96>(q, k, v, o, ldo, batch, heads, n, d, d, lse, stream);
  return launch_t<128, 128>(q, k, v, o, ldo, batch, heads, n, d, d, lse, stream);
}

// pre-size the partial-output scratch for a launch shape (so that no launch of a captured / timed region reallocates it)
int attention_rows64_d128_reserve(int batch, int heads, int n) {
  if (n % QROWS != 0) return DFOT_OK;
  float *a, *b;
  return attn_partials(attn_plan_split(batch, heads, n, QROWS, 1), QROWS, &a, &b, D);
}

}  // namespace dfot
