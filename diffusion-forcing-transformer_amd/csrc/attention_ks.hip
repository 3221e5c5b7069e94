// Level-3 self-attention (128-element rows: N = 2048, d = 128 at the RE10K model) for launches with FEW query tiles: 8 waves per
// workgroup, the key axis split inside the workgroup.
//
// attn_kernel_v2<128> (attention.hip) gives a workgroup of 4 waves 128 query rows and the whole key range.  At model batch 2 that is
// 288 workgroups for 256 CUs: most CUs hold ONE workgroup = one wave per SIMD, and a wave alone on its SIMD runs the chain QK^T ->
// max / exp -> P.V serially (MFMA utilisation 0.23; two resident workgroups run 1.38 x faster each, profiles/r02_q_*).  Here a
// workgroup has TWO groups of 4 waves on the SAME 128 query rows: group g takes the key tiles kt0 + g, kt0 + g + 2, ... through its
// own two-stage LDS-DMA ring, so every SIMD holds two waves that overlap each other's MFMA and softmax phases, and the groups' partial
// results (O, m, l) are merged through LDS at the end (no HBM partials for this split).  One such workgroup fills a CU (128 KiB of
// LDS), so the launch now runs in lock-step rounds, and the left-over of the last round is balanced as in attention_v3.hip: the first
// `full` work items are whole tiles, the remaining tiles are split over the keys into `nsplit` segments each (fp32 partials in the
// caller's AttnScratch + attn128_merge_kernel).
// Same math as attn_kernel_v2: S^T = K Q^T with the running max folded into the accumulator init, deferred rescale (threshold 2^8),
// O^T += V^T P^T with V^T by transposed LDS reads, exp2 domain (q arrives scaled by log2(e) / sqrt(d)).
#include "common.h"
#include "dfot_hip.h"
#include "kernels.h"

namespace dfot {

namespace {

constexpr int D = 128, KV = 64, ROWB = D * 2, TILE = KV * ROWB, CH = D / 8;
constexpr int QR = 128;                          // query rows per workgroup
constexpr int RING = 2 * 2 * TILE;               // bytes of one group's ring: 2 stages x (K tile + V tile)
constexpr int IPW = (TILE / 1024) / 4;           // 1-KiB DMA instructions per wave per tile: 4
constexpr int RPI = 1024 / ROWB;                 // rows per DMA instruction: 4
constexpr float THR = 8.0f;

__device__ __forceinline__ int swz_k(int row, int c) { return c ^ (row & 15); }
__device__ __forceinline__ int swz_v(int row, int c) { return c ^ ((row & 3) << 2); }

typedef __attribute__((ext_vector_type(2))) unsigned ks_u32x2;
template <int OFF>
__device__ __forceinline__ ks_u32x2 ks_read_tr16(unsigned addr) {
  ks_u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void ks_lds_wait(ks_u32x2& a, ks_u32x2& b, ks_u32x2& c, ks_u32x2& d, ks_u32x2& e, ks_u32x2& f, ks_u32x2& g, ks_u32x2& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ bf16x8 ks_bf16x8(ks_u32x2 lo, ks_u32x2 hi) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512, 1) void attn128_ks2_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ V,
                                                             bf16* __restrict__ O, long ldo, int N, int heads, int full_tiles, int nsplit,
                                                             float* __restrict__ part_o, float* __restrict__ part_ml) {
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wl = wave & 3;  // key group, wave inside the group
  const int lq = lane & 31, lh = lane >> 5;
  char* smem = smem_all + grp * RING;
  const int qtiles = N / QR, ntk = N / KV;
  // work item: a whole tile, or one key segment of a left-over tile
  int tile, kt0 = 0, kt1 = ntk, seg = -1;
  if ((int)blockIdx.x < full_tiles) {
    tile = xcd_remap(blockIdx.x, full_tiles);
  } else {
    const int nseg = gridDim.x - full_tiles;
    seg = xcd_remap(blockIdx.x - full_tiles, nseg);
    tile = full_tiles + seg / nsplit;
    const int per = ntk / nsplit;
    kt0 = (seg % nsplit) * per;
    kt1 = kt0 + per;
    if (nsplit == 1) seg = -1;
  }
  const int bh = tile / qtiles;
  const long base = (long)bh * N * D;
  const int q0 = (tile % qtiles) * QR + wl * 32;
  const bf16* Qb = Q + base;
  const bf16* Kb = K + base;
  const bf16* Vb = V + base;

  bf16x8 qf[D / 16];
  const bf16* qrow = Qb + (long)(q0 + lq) * D + lh * 8;
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 16);

  int koff[IPW], voff[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int inst = wl * IPW + i;
    const int row = inst * RPI + lane / CH, pos = lane % CH;
    koff[i] = row * D + swz_k(row, pos) * 8;
    voff[i] = row * D + swz_v(row, pos) * 8;
  }
  auto issue = [&](int kt, int stage) {  // kt: global key-tile index
    char* sk = smem + stage * 2 * TILE;
    char* sv = sk + TILE;
    const bf16* ktp = Kb + (long)kt * KV * D;
    const bf16* vtp = Vb + (long)kt * KV * D;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int inst = wl * IPW + i;
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(ktp + koff[i]), DFOT_LDS_PTR(sk + inst * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(vtp + voff[i]), DFOT_LDS_PTR(sv + inst * 1024), 16, 0, 0);
    }
  };

  f32x16 oacc[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = 0.f, l_i = 0.f;

  // group g owns key tiles kt0 + g, kt0 + g + 2, ...; both groups run the same number of loop iterations (the barriers are
  // workgroup-wide), a group without a tile in an iteration skips its loads and products
  const int nt = kt1 - kt0;
  const int nit = (nt + 1) >> 1;
  const int mine = (nt - grp + 1) >> 1;  // tiles of this group
  if (mine > 0) issue(kt0 + grp, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int it = 0; it < nit; ++it) {
    const char* sk = smem + cur * 2 * TILE;
    const char* sv = sk + TILE;
    if (it + 1 < mine) issue(kt0 + grp + 2 * (it + 1), cur ^ 1);
    if (it < mine) {
      // ---- S^T - m = K Q^T - m ----
      f32x16 sacc[2];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kt2][r] = -m_run;
        const int row = kt2 * 32 + lq;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + row * ROWB + swz_k(row, ks * 2 + lh) * 16);
          sacc[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kt2], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);

      float mx = fmaxf(fmaxf(sacc[0][0], sacc[0][1]), sacc[0][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[0][r + 1]);
      mx = fmaxf(mx, sacc[0][15]);
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sacc[1][r]), sacc[1][r + 1]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));

      const bool grow = (it == 0) || (mx > THR);
      if (__any(grow)) {
        const float delta = (it == 0) ? mx : fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);  // it == 0: O and l are still zero
        m_run += delta;
        l_i *= alpha;
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
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
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int dvt = 0; dvt < D / 32; ++dvt) {
        const int q4 = (lane & 15) >> 2, p4 = lane & 3;
        const int col = dvt * 32 + 16 * ((lane >> 4) & 1) + 4 * p4;
        const int r0 = 4 * lh + q4;
        const unsigned va = (unsigned)(size_t)DFOT_LDS_PTR(sv) + r0 * ROWB + swz_v(r0, col >> 3) * 16 + (col & 7) * 2;
        ks_u32x2 r000 = ks_read_tr16<0 * ROWB>(va), r001 = ks_read_tr16<8 * ROWB>(va);
        ks_u32x2 r010 = ks_read_tr16<16 * ROWB>(va), r011 = ks_read_tr16<24 * ROWB>(va);
        ks_u32x2 r100 = ks_read_tr16<32 * ROWB>(va), r101 = ks_read_tr16<40 * ROWB>(va);
        ks_u32x2 r110 = ks_read_tr16<48 * ROWB>(va), r111 = ks_read_tr16<56 * ROWB>(va);
        ks_lds_wait(r000, r001, r010, r011, r100, r101, r110, r111);
        oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks_bf16x8(r000, r001), pf[0][0], oacc[dvt], 0, 0, 0);
        oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks_bf16x8(r010, r011), pf[0][1], oacc[dvt], 0, 0, 0);
        oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks_bf16x8(r100, r101), pf[1][0], oacc[dvt], 0, 0, 0);
        oacc[dvt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks_bf16x8(r110, r111), pf[1][1], oacc[dvt], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---- fold group 1 into group 0 through LDS: [wave][66 values][lane] fp32 (both rings are dead after the last barrier) ----
  float* red = reinterpret_cast<float*>(smem_all) + wl * (66 * 64);
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(i * 16 + r) * 64 + lane] = oacc[i][r];
    red[64 * 64 + lane] = m_run;
    red[65 * 64 + lane] = l_i;
  }
  __syncthreads();
  if (grp == 1) return;
  if (nt > 1) {  // group 1 had at least one tile
    const float m1 = red[64 * 64 + lane], l1 = red[65 * 64 + lane];
    const float mm = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
    m_run = mm;
    l_i = a0 * l_i + a1 * l1;
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] = a0 * oacc[i][r] + a1 * red[(i * 16 + r) * 64 + lane];
  }

  const float l_tot = l_i + __shfl_xor(l_i, 32);
  const int rloc = wl * 32 + lq;
  if (seg >= 0) {  // key segment: fp32 partial (O, m, l) for attn128_merge_kernel
    float* prow = part_o + ((long)seg * QR + rloc) * D;
#pragma unroll
    for (int dvt = 0; dvt < D / 32; ++dvt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 o4;
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = oacc[dvt][4 * g4 + j];
        *reinterpret_cast<f32x4*>(prow + dvt * 32 + 8 * g4 + 4 * lh) = o4;
      }
    if (lh == 0) *reinterpret_cast<float2*>(part_ml + ((long)seg * QR + rloc) * 2) = make_float2(m_run, l_tot);
    return;
  }
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

// combine the key segments of the left-over tiles: O = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s.  One thread per (query row, 4 columns)
__global__ __launch_bounds__(256) void attn128_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, bf16* __restrict__ O,
                                                            long ldo, int N, int heads, int full_tiles, int nsplit, int rem_tiles) {
  constexpr int TPR = D / 4;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = gid / TPR;
  const int c4 = (int)(gid % TPR) * 4;
  if (row >= (long)rem_tiles * QR) return;
  const int lt = (int)(row / QR), rloc = (int)(row % QR);
  const int qtiles = N / QR;
  float mmax = -INFINITY;
  for (int s = 0; s < nsplit; ++s) mmax = fmaxf(mmax, part_ml[((long)(lt * nsplit + s) * QR + rloc) * 2]);
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const long pr = (long)(lt * nsplit + s) * QR + rloc;
    const float w = exp2f(part_ml[pr * 2] - mmax);
    l += w * part_ml[pr * 2 + 1];
    const f32x4 o = *reinterpret_cast<const f32x4*>(part_o + pr * D + c4);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += w * o[j];
  }
  const float inv = 1.0f / l;
  const int tile = full_tiles + lt;
  const int bh = tile / qtiles, b = bh / heads, hd = bh % heads;
  const int qrow = (tile % qtiles) * QR + rloc;
  bf16x4 o4;
#pragma unroll
  for (int j = 0; j < 4; ++j) o4[j] = f2bf(acc[j] * inv);
  *reinterpret_cast<bf16x4*>(O + ((long)b * N + qrow) * ldo + hd * D + c4) = o4;
}

}  // namespace

static AttnSplit plan_ks(int batch, int heads, int n) { return attn_plan_split(batch, heads, n, QR, 1); }  // one 8-wave workgroup per CU

bool attention_ks_applies(int batch, int heads, int n, int d) {
  // few query tiles: at most one round of one workgroup per CU, or a second round small enough to be split over the keys.  Beyond that
  // the 4-wave kernel's two resident workgroups per CU already give every SIMD two waves (measured, B x H = 27, N = 2048: 73.6 us
  // against 82.5 us for this kernel without a balanced tail; B x H = 18: 67.5 -> 62.5 us; B x H = 9: 47.6 -> 38.2 us)
  if (d != 128 || n % QR != 0 || (n / KV) % 2 != 0) return false;
  const AttnSplit sp = plan_ks(batch, heads, n);
  return sp.full == 0 || (sp.full == sp.slots && sp.nsplit > 1);
}

size_t attention_ks_scratch_bytes(int batch, int heads, int n, int d) {
  if (!attention_ks_applies(batch, heads, n, d)) return 0;
  const AttnSplit sp = plan_ks(batch, heads, n);
  return sp.nsplit == 1 ? 0 : (size_t)sp.rem * sp.nsplit * QR * (D + 2) * sizeof(float);
}

// q, k, v: [B][heads][N][128] bf16, q pre-scaled by log2(e) / sqrt(d); o: row r of batch b, head hd at o[(b * N + r) * ldo + hd * 128]
int launch_attention_ks(const bf16* q, const bf16* k, const bf16* v, bf16* o, long ldo, int batch, int heads, int n, hipStream_t stream,
                        AttnScratch* scratch) {
  DFOT_REQUIRE(q && k && v && o, DFOT_ERR_ARG, "attention: null pointer");
  DFOT_REQUIRE(n > 0 && n % QR == 0 && (n / KV) % 2 == 0 && ldo % 4 == 0, DFOT_ERR_SHAPE, "attention ks: N=%d / ldo=%ld unsupported", n, ldo);
  const AttnSplit sp = plan_ks(batch, heads, n);
  float *po = nullptr, *pml = nullptr;
  int rc = attn_partials(sp, QR, &po, &pml, scratch, D);
  if (rc) return rc;
  const int lds = 2 * RING;
  static bool attr_set = false;
  if (!attr_set) {
    DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn128_ks2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(attn128_ks2_kernel, dim3(sp.full + sp.rem * sp.nsplit), dim3(512), lds, stream, q, k, v, o, ldo, n, heads,
                     sp.full + (sp.nsplit == 1 ? sp.rem : 0), sp.nsplit, po, pml);
  DFOT_CHECK_HIP(hipGetLastError());
  if (sp.nsplit > 1) {
    const long threads = (long)sp.rem * QR * (D / 4);
    hipLaunchKernelGGL(attn128_merge_kernel, dim3(cdiv(threads, 256)), dim3(256), 0, stream, po, pml, o, ldo, n, heads, sp.full, sp.nsplit, sp.rem);
    DFOT_CHECK_HIP(hipGetLastError());
  }
  return DFOT_OK;
}

}  // namespace dfot
