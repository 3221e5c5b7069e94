// bf16 MFMA GEMM (C = A * W^T) and implicit-GEMM 3x3 convolution for gfx950, with the DFoT
// backbone's fused epilogues.
//
// Tile: (128|256)x128x64 per workgroup of 4|8 waves, each wave 64x64 = 4x4 MFMA 16x16x32 bf16 tiles.  Both operands are K-major, so A and W fragments are read from LDS the
// same way: one ds_read_b128 per lane = 8 consecutive k of one row.  LDS rows are 128 B (64 bf16);
// the 16-byte chunk c of row r is stored at position c ^ ((r>>1)&7): any 16 consecutive rows
// read at one logical chunk hit 16 distinct slots of the 256-B bank row (conflict-free
// ds_read_b128, MI355X_MICROARCH.md LDS table).
// Staging: global_load_lds_dwordx4 (LDS-DMA).  The LDS image of one wave instruction is linear
// (8 rows x 128 B), so the swizzle is applied to the per-lane SOURCE address
// (cdna_hip_programming.md rule 21); a register-staged path (DMA=false) is kept for A/B testing.
// Two or three LDS stages; with three, a tile's DMA stays in flight across one barrier (counted vmcnt).
#include "gemm.h"
#include "dfot_hip.h"

namespace dfot {

constexpr int BK = 64;
// a wave owns WTM x 64 of the tile, or WTM x 48 when the tile width is a multiple of 48 but not of 64 (BN_T = 144: N = 576 and
// 1152 split into 4 / 8 column tiles, so M / 256 x N / 144 = 256 / 128 workgroups for the two out-projections of the model)
constexpr int wave_cols(int bn) { return bn % 64 == 0 ? 64 : 48; }
constexpr int tile_threads(int bm, int bn, int wtm, int ks) { return (bm / wtm) * (bn / wave_cols(bn)) * 64 * ks; }
constexpr int EP_LD = 68;  // fp32 row stride of the per-wave epilogue scratch (64 + 4: rows shift by 4 banks)

// BM_T = 128: 256 threads (waves 2x2), BM_T = 256: 512 threads (waves 4x2); every wave owns a 64x64 output block.
// NST = LDS stages.  NST == 2: load tile t+1 while computing t (vmcnt(0) + barrier per tile).
// NST == 3 (LDS-DMA only): tile t+2 is in flight across the barrier; the wait before the barrier is a COUNTED
// s_waitcnt vmcnt(loads of one tile), so a tile's DMA has two compute phases to land
// (cdna_hip_programming.md "Pipelining across barriers").
// wait until at most N of this wave's LDS-DMA loads are outstanding, then barrier (one asm statement: memory
// operations are not moved across it)
template <int N>
__device__ __forceinline__ void wait_all_but() {
  static_assert(N == 2 || N == 4 || N == 6 || N == 8 || N == 12, "unexpected load count");
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
}

// ... with a run-time (wave-uniform) count: tiles whose staging instructions do not divide evenly over the waves (256x144 over 12 waves:
// 5, 4 or 3 LDS-DMA instructions per k-tile depending on the wave)
__device__ __forceinline__ void wait_all_but_n(int n) {
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
}

// KS = 2: intra-workgroup split-K for grids too small to fill the chip (level-3 GEMMs at model batch 2): two groups of
// NW waves each own a private pair of LDS stages and alternate k-tiles (group g takes k-tiles g, g+2, ...), doubling the
// waves per CU and halving the serial k-loop; group 1's accumulators are folded into group 0's through LDS at the end.
// BKT = K-tile depth: 64 (LDS rows of 128 B) or 32 (rows of 64 B: half the stage size, so a 256x256 tile affords a 4-stage
// ring -- the two-stage loop is bound by the latency of the ONE prefetch it has in flight, see gemm_pick_variant)
template <int BM_T, int BN_T, int WTM, int NST, int AMODE, int EPI, bool DMA, int KS, int BKT>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int tile_index, const int tile_count) {
  static_assert(BKT == 64 || (BKT == 32 && AMODE == A_DENSE && DMA && KS == 1), "K-tile depth");
  constexpr int CPR = BKT / 8;                  // 16-byte chunks per LDS row
  constexpr int RPI = 64 / CPR;                 // rows staged by one wave instruction (1 KiB)
  constexpr int ROWB = BKT * 2;                 // bytes per LDS row
  constexpr int MI = WTM / 16;                  // 16-row MFMA tiles per wave along M (wave tile = WTM x WTN)
  constexpr int WTN = wave_cols(BN_T);          // columns per wave: 64, or 48 for BN_T = 144
  constexpr int NI = WTN / 16;                  // 16-column MFMA tiles per wave along N
  constexpr int WN = BN_T / WTN;                // waves along N
  static_assert(WTN == 64 || (EPI == E_F32 || EPI == E_BF16), "48-column wave tiles: plain epilogues only");
  static_assert(WTN == 64 || KS == 1, "48-column wave tiles: no intra-workgroup split-K");
  constexpr int NW = (BM_T / WTM) * WN;         // waves per k-group
  constexpr int NT = NW * 64;                   // threads per k-group
  // one wave instruction stages 8 rows (1 KiB) of a tile; AINS/WINS of them per k-tile are dealt round-robin to the NW
  // waves.  When NW does not divide them (256x192 tile: 32 + 24 over 12 waves) the last turn is guarded per wave.
  constexpr int AINS = BM_T / RPI, WINS = BN_T / RPI;
  constexpr int ACH = (AINS + NW - 1) / NW;     // A chunks (16 B) per thread per k-tile
  constexpr int WCH = (WINS + NW - 1) / NW;     // W chunks per thread per k-tile
  constexpr bool A_EVEN = AINS % NW == 0, W_EVEN = WINS % NW == 0;
  static_assert((A_EVEN && W_EVEN) || (DMA && KS == 1), "uneven staging: LDS-DMA kernels without intra-workgroup split-K only");
  constexpr int A_BYTES = BM_T * BKT * 2;
  constexpr int STAGE_BYTES = (BM_T + BN_T) * BKT * 2;
  constexpr int LOADS = ACH + WCH;              // LDS-DMA instructions per thread per k-tile
  static_assert(DMA || NST == 2, "register staging supports two stages only");
  static_assert(KS == 1 || (NST == 2 && DMA && EPI != E_QKV), "split-K: two-stage LDS-DMA kernels without in-epilogue barriers");
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int kgroup = KS == 1 ? 0 : (tid >> 6) / NW;     // which k-group this wave belongs to
  const int wave = KS == 1 ? (tid >> 6) : (tid >> 6) % NW;  // wave index inside the group
  char* smem = smem_all + kgroup * (NST * STAGE_BYTES);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (g.N + BN_T - 1) / BN_T;
  // split-K over workgroups (EPI == E_F32 only): consecutive workgroups share a tile and take consecutive K slices
  const int ksplit = (EPI == E_F32 && (AMODE == A_DENSE || g.slice_stride > 0) && g.ksplit > 1) ? g.ksplit : 1;
  const int kslice = ksplit > 1 ? tile_index % ksplit : 0;
  const int bid = ksplit > 1 ? xcd_remap(tile_index / ksplit, tile_count / ksplit, g.xcd) : xcd_remap(tile_index, tile_count, g.xcd);
  const int tn = bid % tiles_n, tm = bid / tiles_n;
  const int m0 = tm * BM_T, n0 = tn * BN_T;
  if constexpr (AMODE == A_CONV3) {
    if (g.live && !g.live[(unsigned)m0 / (unsigned)(g.H * g.Wd)]) return;  // workgroup-uniform, before any barrier
  }
  int nk = g.K / BKT;
  long kbeg = 0;  // first element of this workgroup's K range
  if (ksplit > 1) {
    const int per = nk / ksplit, rem = nk % ksplit;
    kbeg = (long)(kslice * per + (kslice < rem ? kslice : rem)) * BKT;
    nk = per + (kslice < rem ? 1 : 0);
  }
  // bank swizzle of the 16-byte chunk position inside a row (conflict-free ds_read_b128 fragment reads, MI355X_MICROARCH.md
  // "LDS": lane groups of 16): 128-B rows: chunk ^ ((row >> 1) & 7); 64-B rows (4 rows per 256-B bank line): chunk ^ ((row >> 2) & 2)
  auto swz = [](int r) { return BKT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 2); };

  // ---- per-thread staging geometry: ACH A chunks + WCH W chunks of 16 B per k-tile ----
  const int prow = lane / CPR;  // row within the group of RPI rows written by one wave instruction
  const int ppos = lane % CPR;  // 16-byte position within the LDS row
  const bf16* a_src[ACH];
  int a_y[ACH], a_x[ACH];
  const bf16* w_src[WCH];
  int a_chunk[ACH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int r = RPI * (NW * i + wave) + prow;
    const int c = ppos ^ swz(r);
    a_chunk[i] = c;
    const long m = (long)m0 + r;
    if constexpr (AMODE == A_DENSE) {
      a_src[i] = g.A + m * g.lda + c * 8 + kbeg;
      a_y[i] = a_x[i] = 0;
    } else {
      const unsigned mu = (unsigned)m;  // 32-bit divisions (M < 2^31)
      const int xw = (int)(mu % (unsigned)g.Wd);
      const int yh = (int)((mu / (unsigned)g.Wd) % (unsigned)g.H);
      a_x[i] = xw;
      a_y[i] = yh;
      a_src[i] = g.A + m * (long)g.Cin + c * 8;
    }
  }
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    const int r = RPI * (NW * i + wave) + prow;
    const int c = ppos ^ swz(r);
    int n = n0 + r;
    n = n < g.N ? n : g.N - 1;
    w_src[i] = g.W + (long)n * (AMODE == A_DENSE && g.ldw ? g.ldw : (long)g.K) + c * 8 + kbeg;
  }

  // conv3x3: k-tile kt covers channels [cv_c0, cv_c0 + BKT) of tap (cv_dy, cv_dx).  issue() is called with consecutive k-tiles
  // (stride KS), so the tap position is carried along instead of being re-derived with three integer divisions per k-tile
  [[maybe_unused]] int cv_c0 = 0, cv_dy = -1, cv_dx = -1;
  if constexpr (AMODE == A_CONV3) {
    const int kbase = kgroup * BKT + (int)kbeg;  // first k-tile of this k-group (of this workgroup's K slice: split-K into partial outputs)
    const int tap = kbase / g.Cin;
    cv_c0 = kbase - tap * g.Cin;
    cv_dy = tap / 3 - 1;
    cv_dx = tap % 3 - 1;
  }
  auto a_addr = [&](int i, int kt) -> const bf16* {
    if constexpr (AMODE == A_DENSE) {
      return a_src[i] + (long)kt * BKT;
    } else {
      const int yy = a_y[i] + cv_dy, xx = a_x[i] + cv_dx;
      const bool ok = (yy >= 0) && (yy < g.H) && (xx >= 0) && (xx < g.Wd);
      const bf16* p = a_src[i] + ((long)cv_dy * g.Wd + cv_dx) * g.Cin + cv_c0;
      return ok ? p : g.zeros + a_chunk[i] * 8;
    }
  };
  auto conv_advance = [&]() {  // to the k-tile of the next issue() of this k-group
    if constexpr (AMODE == A_CONV3) {
      cv_c0 += KS * BKT;
      while (cv_c0 >= g.Cin) {
        cv_c0 -= g.Cin;
        if (++cv_dx > 1) {
          cv_dx = -1;
          ++cv_dy;
        }
      }
    }
  };

  bf16x8 ra[ACH], rw[WCH];  // register staging (DMA=false)

  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE_BYTES;
    char* sw = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      if (!A_EVEN && NW * i + wave >= AINS) continue;  // wave-uniform
      const bf16* pa = a_addr(i, kt);
      if constexpr (DMA) {
        __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(pa), DFOT_LDS_PTR(sa + (NW * i + wave) * 1024), 16, 0, 0);
      } else {
        ra[i] = *reinterpret_cast<const bf16x8*>(pa);
      }
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      if (!W_EVEN && NW * i + wave >= WINS) continue;
      const bf16* pw = w_src[i] + (long)kt * BKT;
      if constexpr (DMA) {
        __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(pw), DFOT_LDS_PTR(sw + (NW * i + wave) * 1024), 16, 0, 0);
      } else {
        rw[i] = *reinterpret_cast<const bf16x8*>(pw);
      }
    }
    conv_advance();
  };
  auto commit = [&](int stage) {  // DMA=false only: registers -> LDS
    char* sa = smem + stage * STAGE_BYTES;
    char* sw = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < ACH; ++i) *reinterpret_cast<bf16x8*>(sa + (NW * i + wave) * 1024 + lane * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < WCH; ++i) *reinterpret_cast<bf16x8*>(sw + (NW * i + wave) * 1024 + lane * 16) = rw[i];
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  auto compute = [&](int stage) {
    const char* sa = smem + stage * STAGE_BYTES;
    const char* sw = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BKT / 32; ++ks) {
      bf16x8 af[MI], wf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int r = wm * WTM + mi * 16 + frow;
        const int pos = (ks * 4 + fk) ^ swz(r);
        af[mi] = *reinterpret_cast<const bf16x8*>(sa + r * ROWB + pos * 16);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int r = wn * WTN + ni * 16 + frow;
        const int pos = (ks * 4 + fk) ^ swz(r);
        wf[ni] = *reinterpret_cast<const bf16x8*>(sw + r * ROWB + pos * 16);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], wf[ni], acc[mi][ni], 0, 0, 0);
    }
  };

  // ---- main loop ----
  if constexpr (NST == 2 && KS == 1) {
    issue(0, 0);
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      commit(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
      compute(cur);
      if constexpr (DMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (kt + 1 < nk) commit(cur ^ 1);
      }
      __syncthreads();
    }
  } else if constexpr (NST == 2) {
    // split-K: both groups run the same number of iterations (barriers are workgroup-wide); a group without a k-tile
    // in an iteration simply skips its loads and MFMAs
    const int nit = (nk + KS - 1) / KS;
    if (kgroup < nk) issue(kgroup, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
      const int cur = it & 1;
      const int kt = it * KS + kgroup;
      if (kt + KS < nk) issue(kt + KS, cur ^ 1);
      if (kt < nk) compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // fold group 1 into group 0: [wave][acc register][lane] fp32, conflict-free and coalesced
    float* red = reinterpret_cast<float*>(smem_all) + wave * (MI * 16 * 64);
    if (kgroup == 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int j = 0; j < 4; ++j) red[((mi * 4 + ni) * 4 + j) * 64 + lane] = acc[mi][ni][j];
    }
    __syncthreads();
    if (kgroup == 1) return;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mi][ni][j] += red[((mi * 4 + ni) * 4 + j) * 64 + lane];
  } else {
    // NST-stage ring (NST >= 3).  Invariant at the top of iteration kt: tile kt has landed and is visible to every wave
    // (its waves waited for it, then passed a barrier); tiles kt+1 .. kt+NST-2 may still be in flight.
    static_assert(NST >= 3, "ring");
    // LDS-DMA instructions THIS wave issues per k-tile (wave-uniform; = LOADS when the staging divides evenly)
    [[maybe_unused]] const int my_loads =
        __builtin_amdgcn_readfirstlane((AINS - wave + NW - 1) / NW + (WINS - wave + NW - 1) / NW);
    auto ring_wait = [&]() {
      if constexpr (A_EVEN && W_EVEN) wait_all_but<(NST - 2) * LOADS>();
      else wait_all_but_n((NST - 2) * my_loads);
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
      if (t < nk) issue(t, t);
    if (nk >= NST - 1) {
      ring_wait();
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // stage (kt+NST-1) % NST == (kt-1) % NST was last read in iteration kt-1, which every wave left through a barrier
      const int nxt = st == 0 ? NST - 1 : st - 1;
      if (kt + NST - 1 < nk) issue(kt + NST - 1, nxt);
      compute(st);
      if (kt + NST - 1 < nk) {
        ring_wait();  // tile kt+1 has landed; the NST-2 newer ones may stay in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
      st = st == NST - 1 ? 0 : st + 1;
    }
  }

  // ---- epilogue: 16 output rows at a time are transposed through a per-wave LDS scratch so that global traffic is
  // 16 B per lane on whole 128/256-byte row segments (the MFMA C layout gives a lane one column and four rows).
  // All stage buffers are dead after the last barrier; same-wave LDS accesses complete in order.
  // (with split-K the scratch sits behind the reduction area, which other waves of group 0 may still be reading)
  //
  // LDS invariants of the epilogue (argued once, here; DESIGN.md section 7 "the round-3 q mismatch"):
  //  (1) ring -> scratch.  `ep` aliases stage 0 of the operand ring.  Every main-loop form ends with `s_waitcnt vmcnt(0)` (all of this
  //      wave's LDS-DMA writes have landed) followed by a workgroup barrier that every wave reaches after its last fragment read
  //      (the ds_reads feed MFMAs issued before the barrier), so no wave can still read operands, and no DMA can still write, where
  //      another wave starts writing its scratch.
  //  (2) scratch.  wave w touches only floats [w * 16 * EP_LD, (w + 1) * 16 * EP_LD): written in the MFMA C layout, read back as rows,
  //      by the same wave.  ds_write / ds_read of one wave execute in issue order, so the read-back of pass mi sees the writes of pass
  //      mi and is finished before the writes of pass mi + 1; no other wave and no DMA ever addresses the region: no barrier needed.
  //  (3) d = 128 exchange (`xch`, behind all scratch regions, two buffers by pass parity).  wave w writes slots [w * 16, w * 16 + 16) of
  //      buffer mi & 1, then the barrier, then reads the slots of wave w ^ 1.  The partner can overwrite buffer mi & 1 again only in
  //      pass mi + 2, i.e. after the barrier of pass mi + 1, which this wave reaches after its reads of pass mi: write-after-read safe
  //      with ONE barrier per pass.  The barrier is under a workgroup-uniform condition and no wave leaves the epilogue early.
  //  (4) row reductions are DPP (register-to-register inside a row of 8 lanes): nothing of them goes through LDS.
  // Nothing in (1)-(4) depends on which other workgroups (of this or another kernel) are resident on the CU: LDS allocations are
  // disjoint and every address above is relative to this workgroup's own.
  float* ep =reinterpret_cast<float*>(smem_all) + (KS == 2 ? NW * MI * 16 * 64 : 0) + wave * (16 * EP_LD);
  const int colq = lane & 15, rowq = (lane >> 4) * 4;
  const int nw = n0 + wn * WTN;
  [[maybe_unused]] const bool has_res = g.resid != nullptr;
  [[maybe_unused]] const bool has_gate = g.gate != nullptr;
  // fused GroupNorm partial sums of this wave's 64 rows: (gsum,gsq) = lane's first 4 columns, (gsum2,gsq2) = next 4
  [[maybe_unused]] float gsum = 0.f, gsq = 0.f, gsum2 = 0.f, gsq2 = 0.f;
  // lane geometry of the read-back: fp32 output = 4 columns x rows p*4 + lane/16; bf16 outputs = 8 columns x rows p*8 + lane/8
  const int cw = EPI == E_F32 ? (lane & 15) * 4 : (lane & 7) * 8;
  const int col = nw + cw;
  const bool live = col < g.N && cw < WTN;  // N is a multiple of the lane's column count, so a lane is entirely in or out
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
  // two-dimensional bias: bias_rows > 0: bias[row % bias_rows][col] (MatrixAttention qkv_bias / proj_bias); bias_rows < 0:
  // bias[row / -bias_rows][col] (one bias row per frame of -bias_rows consecutive rows: the per-frame part of the trainer's folded FiLM)
  [[maybe_unused]] const bool bias2d = g.bias_rows != 0;
  [[maybe_unused]] const unsigned bias_den = (unsigned)(g.bias_rows > 0 ? g.bias_rows : -g.bias_rows);
  [[maybe_unused]] const bool bias_mod = g.bias_rows > 0;
  [[maybe_unused]] float bc[NI];  // E_QKV: the bias in the MFMA C layout (one column per lane and 16-column tile), added on the way into LDS
  if constexpr (EPI == E_QKV) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int c = nw + ni * 16 + colq;
      bc[ni] = (g.bias && c < g.N) ? g.bias[c] : 0.f;
    }
  } else if (g.bias && live && !bias2d && kslice == 0) {  // split-K: slice 0 alone adds the bias
    b0 = *reinterpret_cast<const f32x4*>(g.bias + col);
    if constexpr (EPI != E_F32) b1 = *reinterpret_cast<const f32x4*>(g.bias + col + 4);
  }
  if constexpr (EPI == E_BF16) {
    if (g.tr_rows > 0) {
      // transposed store straight from the MFMA C layout (a lane owns one column and four consecutive rows):
      // out[frame][col][row % tr_rows], frame = row / tr_rows  -- no bias / activation / GroupNorm statistics
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const unsigned row = (unsigned)(m0 + wm * WTM + mi * 16 + rowq);  // 32-bit index arithmetic: M < 2^31 (launcher)
        const long frame = row / (unsigned)g.tr_rows;
        const int rin = (int)(row % (unsigned)g.tr_rows);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int c = nw + ni * 16 + colq;
          if (c < g.N) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f2bf(acc[mi][ni][j]);
            *reinterpret_cast<bf16x4*>(g.out_bf16 + (frame * g.N + c) * (long)g.tr_rows + rin) = o;
          }
        }
      }
      return;
    }
  }
  // E_QKV: everything that does not depend on the row is worked out once per lane, and the rotary table rows of a pass are fetched one
  // pass AHEAD, in front of the previous pass's stores: the vector-memory counter retires loads and stores in issue order, so a load
  // issued behind a pass's stores cannot be waited for without waiting for those stores too (four exposed store round trips per pass
  // in the first form of this epilogue: 20 of the 118 us of the level-2 fused projection)
  [[maybe_unused]] f32x4 res_cur[4], gate_cur;
  // adaLN gate: one gate vector per gate_rows output rows; when that is a multiple of 16 the 16 rows of a pass share it (4 registers)
  [[maybe_unused]] const bool pre_gate = EPI == E_F32 && has_gate && live && g.gate_rows % 16 == 0;
  [[maybe_unused]] auto load_gate = [&](int mi) {
    long gr = (unsigned)(m0 + wm * WTM + mi * 16) / (unsigned)g.gate_rows;  // 32-bit division (a 64-bit one costs ~100 instructions)
    if (g.gate_index) gr = g.gate_index[gr];
    gate_cur = *reinterpret_cast<const f32x4*>(g.gate + gr * g.ldg + col);
  };
  [[maybe_unused]] const bool pre_res = EPI == E_F32 && has_res && ksplit == 1 && live;
  [[maybe_unused]] auto load_res = [&](int mi) {  // residual rows of pass mi (16 rows of the wave's tile)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const long row = (long)m0 + wm * WTM + mi * 16 + p * 4 + (lane >> 4);
      res_cur[p] = *reinterpret_cast<const f32x4*>(g.resid + row * g.ldo + col);
    }
  };
  if constexpr (EPI == E_F32) {
    if (pre_res) load_res(0);
    if (pre_gate) load_gate(0);
  }
  [[maybe_unused]] int q_which = 0, q_head = 0, q_e0 = 0;
  [[maybe_unused]] bool q_rot = false;            // this lane's columns are q or k columns (normalised and rotated)
  [[maybe_unused]] float q_w[8], q_mul = 1.f;
  [[maybe_unused]] bf16* q_dst = nullptr;
  [[maybe_unused]] f32x4 cs_cur[2][2];
  [[maybe_unused]] unsigned t_b = 0, t_k = 0;     // batch element and token of the tile's first row
  [[maybe_unused]] auto row_token = [&](int ro, long& bidx, int& tok) {  // row m0 + ro -> (batch element, token), ro < BM_T
    unsigned t = t_k + (unsigned)ro, b = t_b;
    if (g.ntok >= BM_T) {  // uniform: a tile spans at most two batch elements
      if (t >= (unsigned)g.ntok) {
        t -= (unsigned)g.ntok;
        ++b;
      }
    } else {
      b += t / (unsigned)g.ntok;
      t = t % (unsigned)g.ntok;
    }
    bidx = b;
    tok = (int)t;
  };
  [[maybe_unused]] auto load_cs = [&](int mi, f32x4 (&dst)[2][2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      long bidx;
      int tok;
      row_token(wm * WTM + mi * 16 + p * 8 + (lane >> 3), bidx, tok);
      const float* cs = g.rope_cs + ((long)tok * (g.d / 2) + q_e0 / 2) * 2;
      dst[p][0] = *reinterpret_cast<const f32x4*>(cs);
      dst[p][1] = *reinterpret_cast<const f32x4*>(cs + 4);
    }
  };
  if constexpr (EPI == E_QKV) {
    // the wave's 64 columns lie inside one region and one head: region, head and destination are wave-uniform (scalar registers)
    t_b = (unsigned)m0 / (unsigned)g.ntok;
    t_k = (unsigned)m0 - t_b * (unsigned)g.ntok;
    const int colw = __builtin_amdgcn_readfirstlane(nw);
    if (live && colw < g.split) {
      const int cdim = g.split / 3;
      q_which = colw / cdim;
      const int ccw = colw - q_which * cdim;
      q_head = ccw / g.d;
      q_e0 = ccw % g.d + cw;
      q_dst = q_which == 0 ? g.q : (q_which == 1 ? g.k : g.v);
      q_rot = q_which < 2;
      if (q_rot) {
        const float* wgt = (q_which == 0 ? g.qw : g.kw) + q_e0;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wgt), w1 = *reinterpret_cast<const f32x4*>(wgt + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) q_w[j] = w0[j], q_w[4 + j] = w1[j];
        q_mul = q_which == 0 ? g.qscale : 1.f;
        load_cs(0, cs_cur);
      }
    }
  } else if constexpr (EPI == E_QKV_DIT) {
    // DiT blocks: q | k | v columns only, no RMS norm; the head dimension need not divide the wave's 64 columns (72 for DiT/XL), so
    // region / head / offset are per lane (a lane's 8 columns stay inside one head: d % 8 == 0)
    t_b = (unsigned)m0 / (unsigned)g.ntok;
    t_k = (unsigned)m0 - t_b * (unsigned)g.ntok;
    if (live) {
      const int cdim = g.heads * g.d;
      q_which = col / cdim;
      const int cc = col - q_which * cdim;
      q_head = cc / g.d;
      q_e0 = cc % g.d;
      q_dst = q_which == 0 ? g.q : (q_which == 1 ? g.k : g.v);
      q_mul = q_which == 0 ? g.qscale : 1.f;
      q_rot = q_which < 2 && g.rope_cs != nullptr;  // v, and q / k of the per-frame spatial blocks, are not rotated
      if (q_rot) load_cs(0, cs_cur);
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) ep[(rowq + j) * EP_LD + ni * 16 + colq] = EPI == E_QKV ? acc[mi][ni][j] + bc[ni] : acc[mi][ni][j];
    const long mw = (long)m0 + wm * WTM + mi * 16;
    if constexpr (EPI == E_F32) {
      if (live) {
        // the pass's four output rows are finished first, the NEXT pass's residual rows are fetched, and only then are this pass's
        // rows stored: a residual load issued behind a store could not be waited for without that store (in-order memory counter)
        f32x4 vout[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int r = p * 4 + (lane >> 4);
          f32x4 v = *reinterpret_cast<const f32x4*>(ep + r * EP_LD + cw) + b0;
          if (bias2d) v += *reinterpret_cast<const f32x4*>(g.bias + (long)(bias_mod ? (unsigned)(mw + r) % bias_den : (unsigned)(mw + r) / bias_den) * g.N + col);
          if (has_gate) {
            if (pre_gate) {
              v *= gate_cur;
            } else {
              long gr = (unsigned)(mw + r) / (unsigned)g.gate_rows;
              if (g.gate_index) gr = g.gate_index[gr];
              v *= *reinterpret_cast<const f32x4*>(g.gate + gr * g.ldg + col);
            }
          }
          if (pre_res) v += res_cur[p];
          vout[p] = v;
        }
        if (pre_res && mi + 1 < MI) load_res(mi + 1);
        if (pre_gate && mi + 1 < MI) load_gate(mi + 1);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int r = p * 4 + (lane >> 4);
          const f32x4 v = vout[p];
          const long off = (mw + r) * g.ldo + col + (ksplit > 1 ? (long)kslice * g.slice_stride : 0L);
          if (ksplit > 1 && g.slice_stride == 0) {  // out already holds the residual (resid == out, checked by the launcher) or zeros
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(g.out_f32 + off + j, v[j]);
          } else {
            *reinterpret_cast<f32x4*>(g.out_f32 + off) = v;
          }
          gsum += v[0] + v[1] + v[2] + v[3];
          gsq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
      }
    } else {
      [[maybe_unused]] float vals[2][8];
#pragma unroll
      for (int p = 0; p < (EPI == E_QKV ? 0 : 2); ++p) {
        const int r = p * 8 + (lane >> 3);
        f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + r * EP_LD + cw) + b0;
        f32x4 v1 = *reinterpret_cast<const f32x4*>(ep + r * EP_LD + cw + 4) + b1;
        if constexpr (EPI == E_BF16) {
          if (bias2d && live) {
            const float* bp = g.bias + (long)(bias_mod ? (unsigned)(mw + r) % bias_den : (unsigned)(mw + r) / bias_den) * g.N + col;
            v0 += *reinterpret_cast<const f32x4*>(bp);
            v1 += *reinterpret_cast<const f32x4*>(bp + 4);
          }

        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vals[p][j] = v0[j];
          vals[p][4 + j] = v1[j];
        }
      }
      if constexpr (EPI == E_BF16) {
        if (g.act == 1 && g.pre_act && live) {  // training: keep the pre-activation too (same shape and row stride as the output)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int r = p * 8 + (lane >> 3);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(vals[p][j]);
            *reinterpret_cast<bf16x8*>(g.pre_act + (mw + r) * g.ldo + col) = o;
          }
        }
        if (g.act == 1) {  // GELU, tanh approximation (timm Mlp act of the DiT blocks)
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float x = vals[p][j];
              const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
              vals[p][j] = x * (1.0f - 1.0f / (1.0f + __expf(2.0f * u)));  // 0.5 x (1 + tanh u)
            }
        }
        if (live) {
          const bool silu = g.act == 2;  // SiLU (the MLP half of fused_attn_mlp_proj when it runs as its own GEMM), applied at the conversion
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int r = p * 8 + (lane >> 3);
            bf16x8 o;
            if ((AMODE == A_CONV3 || BN_T == 144) && g.resid_bf) {  // (convolutions and the 256x144 ring: the persistent dense 256x192 tiles have no register to spare) bf16 residual stream (ResBlock levels of the inference engine): same rows / pitch as the output, added
              // last (after an activation, if any), loaded here so that it is live for one row only (the 128-register budget of the 16-wave tiles)
              const bf16x8 rb = *reinterpret_cast<const bf16x8*>(g.resid_bf + (mw + r) * g.ldo + col);
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = f2bf((silu ? silu_f(vals[p][j]) : vals[p][j]) + bf2f(rb[j]));
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = f2bf(silu ? silu_f(vals[p][j]) : vals[p][j]);
            }
            *reinterpret_cast<bf16x8*>(g.out_bf16 + (mw + r) * g.ldo + col) = o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // GroupNorm statistics of the values as stored (bf16-rounded)
              const float a = bf2f(o[j]), c = bf2f(o[4 + j]);
              gsum += a;
              gsq += a * a;
              gsum2 += c;
              gsq2 += c * c;
            }
          }
        }
      } else if constexpr (EPI == E_QKV_DIT) {
        if (live) {
          bf16x8 o2[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            bf16x8& o = o2[p];
            if (!q_rot) {
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = f2bf(vals[p][j] * q_mul);
            } else {
#pragma unroll
              for (int pr = 0; pr < 4; ++pr) {
                const float x0 = vals[p][2 * pr], x1 = vals[p][2 * pr + 1];
                const float co = cs_cur[p][pr >> 1][2 * (pr & 1)], si = cs_cur[p][pr >> 1][2 * (pr & 1) + 1];
                o[2 * pr] = f2bf((x0 * co - x1 * si) * q_mul);
                o[2 * pr + 1] = f2bf((x1 * co + x0 * si) * q_mul);
              }
            }
          }
          if (q_rot && mi + 1 < MI) load_cs(mi + 1, cs_cur);  // in front of this pass's stores (see the E_QKV epilogue)
          asm volatile("" ::: "memory");
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            long bidx;
            int tok;
            row_token(wm * WTM + mi * 16 + p * 8 + (lane >> 3), bidx, tok);
            *reinterpret_cast<bf16x8*>(q_dst + ((bidx * g.heads + q_head) * g.ntok + tok) * (long)g.dstride + q_e0) = o2[p];
          }
        }
      } else {  // E_QKV
        // The wave's 64 columns lie inside one region (q | k | v | MLP) and, for q/k, inside one head: the whole head
        // when d = 64, half of it when d = 128 -- then the other half belongs to wave^1 and the squared sums are
        // exchanged through LDS.  Everything up to the barrier is executed by every lane of the workgroup.
        // The 16 transposed rows stay in the wave's LDS scratch for the whole pass and are read again where they are used (8 values
        // at a time) instead of being held in registers across the row reduction: the 16-wave kernels sit at the 128-register cap.
        auto read_row = [&](int p, float (&v)[8]) {
          const int r = p * 8 + (lane >> 3);
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + r * EP_LD + cw);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(ep + r * EP_LD + cw + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v0[j], v[4 + j] = v1[j];
        };
        float ssq[2] = {0.f, 0.f};
        if (q_rot || g.d == 128) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            float v[8];
            read_row(p, v);
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += v[j] * v[j];
            // 8 lanes = this wave's 64 columns of one row: butterfly over lane bits 0, 1 (quad permutes) and 2 (mirror of the
            // 8-lane half row: after the first two steps the four lanes of a quad agree, so lane i <-> 7 - i pairs the quads)
            t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, false));
            t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, false));
            t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xF, 0xF, false));
            ssq[p] = t;
          }
        }
        if (g.d == 128) {  // workgroup-uniform
          float* xch = reinterpret_cast<float*>(smem) + NW * 16 * EP_LD + (mi & 1) * NW * 16;
          if ((lane & 7) == 0) {
            xch[wave * 16 + (lane >> 3)] = ssq[0];
            xch[wave * 16 + 8 + (lane >> 3)] = ssq[1];
          }
          __syncthreads();
          ssq[0] += xch[(wave ^ 1) * 16 + (lane >> 3)];
          ssq[1] += xch[(wave ^ 1) * 16 + 8 + (lane >> 3)];
        }
        if (live && g.raw) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int r = p * 8 + (lane >> 3);
            float v[8];
            read_row(p, v);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(v[j]);
            *reinterpret_cast<bf16x8*>(g.raw + (mw + r) * g.ldraw + col) = o;
          }
        }
        if (live) {
          if (col >= g.split) {  // MLP half: SiLU
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const int r = p * 8 + (lane >> 3);
              float v[8];
              read_row(p, v);
              bf16x8 o;
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = f2bf(silu_f(v[j]));
              *reinterpret_cast<bf16x8*>(g.out2 + (mw + r) * g.ldo2 + (col - g.split)) = o;
            }
          } else {
            bf16x8 o2[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              bf16x8& o = o2[p];
              float v[8];
              read_row(p, v);
              if (!q_rot) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(v[j]);
              } else {
                const float rs = rsqrtf(ssq[p] / (float)g.d + g.eps);
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                  const float x0 = v[2 * pr] * rs * q_w[2 * pr];
                  const float x1 = v[2 * pr + 1] * rs * q_w[2 * pr + 1];
                  const float co = cs_cur[p][pr >> 1][2 * (pr & 1)], si = cs_cur[p][pr >> 1][2 * (pr & 1) + 1];
                  o[2 * pr] = f2bf((x0 * co - x1 * si) * q_mul);
                  o[2 * pr + 1] = f2bf((x1 * co + x0 * si) * q_mul);
                }
              }
            }
            // the next pass's table rows go out behind the last use of this pass's and in FRONT of this pass's stores (the compiler
            // may not move either across the barrier: float loads and bf16 stores do not alias for it otherwise)
            if (q_rot && mi + 1 < MI) load_cs(mi + 1, cs_cur);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              long bidx;
              int tok;
              row_token(wm * WTM + mi * 16 + p * 8 + (lane >> 3), bidx, tok);
              *reinterpret_cast<bf16x8*>(q_dst + ((bidx * g.heads + q_head) * g.ntok + tok) * (long)g.d + q_e0) = o2[p];
            }
          }
        }
      }
    }
  }

  if constexpr (EPI == E_F32 || EPI == E_BF16) {
    if (g.gn_part) {  // wave-uniform (launcher guarantees a 64-row wave tile)
      const unsigned mrow = (unsigned)(m0 + wm * WTM);
      const int bt = (int)(mrow / (unsigned)g.gn_rows_per_bt);
      const int slots = g.gn_rows_per_bt / 64;
      const int slot = (int)((mrow % (unsigned)g.gn_rows_per_bt) / 64);
      float* dst = g.gn_part + ((long)bt * slots + slot) * 64;
      if constexpr (EPI == E_F32) {
        // lane = 4 columns (one group when cpg == 4, half a group when cpg == 8) x 16 rows; rows of the other
        // three 16-lane groups are folded in with xor 16 / 32
        if (g.gn_cpg == 8) {
          gsum += __shfl_xor(gsum, 1);
          gsq += __shfl_xor(gsq, 1);
        }
        gsum += __shfl_xor(gsum, 16);
        gsq += __shfl_xor(gsq, 16);
        gsum += __shfl_xor(gsum, 32);
        gsq += __shfl_xor(gsq, 32);
        if (lane < 16 && live && (g.gn_cpg == 4 || (lane & 1) == 0)) {
          const int grp = col / g.gn_cpg;
          dst[grp * 2] = gsum;
          dst[grp * 2 + 1] = gsq;
        }
      } else {
        // lane = 8 columns (two groups when cpg == 4, one when cpg == 8) x 8 rows; fold the other 7 row groups
        if (g.gn_cpg == 8) {
          gsum += gsum2;
          gsq += gsq2;
        }
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
          gsum += __shfl_xor(gsum, o);
          gsq += __shfl_xor(gsq, o);
          gsum2 += __shfl_xor(gsum2, o);
          gsq2 += __shfl_xor(gsq2, o);
        }
        if (lane < 8 && live) {
          const int grp = col / g.gn_cpg;
          dst[grp * 2] = gsum;
          dst[grp * 2 + 1] = gsq;
          if (g.gn_cpg == 4) {
            dst[grp * 2 + 2] = gsum2;
            dst[grp * 2 + 3] = gsq2;
          }
        }
      }
    }
  }
}

// One tile per workgroup.  (Wrapping this body in a tile loop costs the 16-wave kernels, which sit at the 128-VGPR cap, up to 35
// spilled VGPRs -- the fused QKV GEMM went from 96.7 to 117.6 us -- so the loop lives in a separate kernel below.)
template <int BM_T, int BN_T, int WTM, int NST, int AMODE, int EPI, bool DMA, int KS = 1, int BKT = BK>
__global__ __launch_bounds__(tile_threads(BM_T, BN_T, WTM, KS)) void gemm_kernel(GemmArgs g) {
  gemm_tile<BM_T, BN_T, WTM, NST, AMODE, EPI, DMA, KS, BKT>(g, blockIdx.x, gridDim.x);
}

// Persistent form: grid = resident workgroups, each walks tile, tile + grid, ... -- no workgroup relaunch between the tiles of
// a CU, and the stores of one tile drain while the next tile's first loads are in flight.  gridDim.x is a multiple of 8, so a
// tile keeps the XCD its index implies (xcd_remap).  Only instantiated for kernels of <= 12 waves (>= 168 VGPRs per wave).
template <int BM_T, int BN_T, int WTM, int NST, int AMODE, int EPI, bool DMA, int KS = 1, int BKT = BK>
__global__ __launch_bounds__(tile_threads(BM_T, BN_T, WTM, KS)) void gemm_kernel_persistent(GemmArgs g, int tile_count) {
  for (int tile = blockIdx.x; tile < tile_count; tile += gridDim.x) {
    gemm_tile<BM_T, BN_T, WTM, NST, AMODE, EPI, DMA, KS, BKT>(g, tile, tile_count);
    if (tile + (int)gridDim.x < tile_count) __syncthreads();  // the next tile's staging re-uses this tile's epilogue scratch
  }
}

template <int BM_T, int BN_T, int WTM, int NST, int AMODE, int EPI, bool DMA, int KS = 1, int BKT = BK>
static int launch_t(const GemmArgs& g, hipStream_t stream) {
  constexpr int nthreads = tile_threads(BM_T, BN_T, WTM, KS);
  constexpr int stage_lds = KS * NST * (BM_T + BN_T) * BKT * 2;
  constexpr int ep_lds = (nthreads / 64) * 16 * EP_LD * 4 + 2 * (nthreads / 64) * 16 * 4 +
                         (KS == 2 ? (nthreads / 128) * (WTM / 16) * 16 * 64 * 4 : 0);
  constexpr int lds = stage_lds > ep_lds ? stage_lds : ep_lds;
  const int ksplit = (EPI == E_F32 && (AMODE == A_DENSE || g.slice_stride > 0) && g.ksplit > 1) ? g.ksplit : 1;
  const int tiles = (g.M / BM_T) * ((g.N + BN_T - 1) / BN_T) * ksplit;
  // XCD-aware tile order and the persistent tile loop of the <= 12-wave kernels are always on (measured: persistent +0.6 % RE10K,
  // +2.2 % bash/k600 model)
  GemmArgs ga = g;
  ga.xcd = 1;
  ga.persist = 1;
  constexpr bool kPersistOK = KS == 1 && nthreads <= 768;
  if constexpr (kPersistOK) {
    if (ga.persist > 0 && ksplit == 1) {
      const int resident = 256 * (lds <= 80 * 1024 ? 2 : 1);  // workgroups the chip holds at once (LDS-limited)
      if (tiles > resident) {
        auto kp = gemm_kernel_persistent<BM_T, BN_T, WTM, NST, AMODE, EPI, DMA, KS, BKT>;
        static bool pattr_set = false;
        if (!pattr_set) {
          DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
          pattr_set = true;
        }
        hipLaunchKernelGGL(kp, dim3(resident), dim3(nthreads), lds, stream, ga, tiles);
        DFOT_CHECK_HIP(hipGetLastError());
        return DFOT_OK;
      }
    }
  }
  auto kern = gemm_kernel<BM_T, BN_T, WTM, NST, AMODE, EPI, DMA, KS, BKT>;
  static bool attr_set = false;
  if (!attr_set) {
    DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(nthreads), lds, stream, ga);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

template <int AMODE, int EPI>
static int launch_v(int variant, const GemmArgs& g, hipStream_t s) {
  switch (variant) {
    case GEMM_REGS_128: return launch_t<128, 128, 64, 2, AMODE, EPI, false>(g, s);
    case GEMM_DMA_128: return launch_t<128, 128, 64, 2, AMODE, EPI, true>(g, s);
    case GEMM_DMA_256x256: return launch_t<256, 256, 64, 2, AMODE, EPI, true>(g, s);
    case GEMM_DMA_512x128: return launch_t<512, 128, 64, 2, AMODE, EPI, true>(g, s);
    case GEMM_DMA_256x192:
      if constexpr (AMODE != A_DENSE) break;
      else return launch_t<256, 192, 64, 2, AMODE, EPI, true>(g, s);
    case GEMM_DMA3_256x144:  // (also the long-K Downsample / Upsample convolutions: run_down / run_up)
      if constexpr (EPI != E_F32 && !(EPI == E_BF16 && AMODE == A_DENSE)) break;  // E_BF16: the level-2 out-projection onto the bf16 stream
      else {
        if (g.gn_part) break;
        return launch_t<256, 144, 64, 3, AMODE, EPI, true>(g, s);
      }
    case GEMM_DMA_128x192:
      if constexpr (AMODE != A_DENSE || EPI == E_QKV) break;
      else return launch_t<128, 192, 64, 2, AMODE, EPI, true>(g, s);
    case GEMM_DMA_128_KS2:
      if constexpr (EPI == E_QKV) break;
      else return launch_t<128, 128, 64, 2, AMODE, EPI, true, 2>(g, s);
  }
  set_error("gemm: unknown variant %d", variant);
  return DFOT_ERR_ARG;
}

int gemm_pick_variant(int amode, int m, int n, int k, bool plain_f32) {
  // Measured on MI355X at the model's shapes (tools/bench_ops.py, profiles/): these GEMMs are bound by L2->LDS operand
  // traffic, so the 256x256 tile (128 FLOP per operand byte instead of 64) wins whenever it still fills the chip:
  // N wide enough that the padded columns are cheap, and enough tiles for the 256 CUs.
  const long tiles = (long)(m / 256) * ((n + 255) / 256);
  if (m % 256 == 0 && n >= 192 && amode == A_DENSE) {
    // one workgroup per CU: useful fraction of the issued tile-rounds (padding columns + the last, partly filled round).
    // 256x192 tiles win where N is a multiple of 192 but not of 256 (DiT/XL: 1152, 3456)
    auto util = [&](int bn) {
      const long t = (long)(m / 256) * ((n + bn - 1) / bn);
      return (double)m * n / ((double)((t + 255) / 256) * 256 * 256 * bn);
    };
    // long K and N a multiple of 144: the three-stage 256x144 ring where its tile count fills the rounds better than both
    // (M = 16384, N = 1152 at model batch 8: 512 tiles = 2 full rounds against 384 tiles of 256x192 = 1.5; K = 5760: 231 vs 261 us,
    // K = 8064: 319 vs 363 us; at equal utilisation the wider tiles win: M = 65536, N = 576, K = 2880: 262 vs 250 us)
    if (plain_f32 && n % 144 == 0 && k >= 2304 && (long)(m / 256) * (n / 144) >= 256 && util(144) > 1.1 * util(192) && util(144) > 1.1 * util(256))
      return GEMM_DMA3_256x144;
    if (util(192) > 1.05 * util(256) && (long)(m / 256) * ((n + 191) / 192) >= 160) return GEMM_DMA_256x192;
  }
  if (m % 256 == 0 && n >= 192 && tiles >= 160) return GEMM_DMA_256x256;
  // N = 128 (level-0 convolutions): one column of tiles, so grow the tile along M instead (16 waves, 102 FLOP/B)
  if (m % 512 == 0 && n <= 128 && m / 512 >= 256) return GEMM_DMA_512x128;
  const long tiles128 = (long)(m / 128) * ((n + 127) / 128);
  // (128x192 tiles for "a few more 128x128 tiles than CUs and a long K" gain 7 % in isolation and lose inside the model, 8.24 vs 8.33
  // frames/s: not picked here; the training weight gradients ask for that variant explicitly)
  // at most one 128x128 tile per CU and a long K (Upsample convolutions at 16x16 / 32x32): split K inside the workgroup
  if (tiles128 <= 256 && k >= 2048) return GEMM_DMA_128_KS2;
  return GEMM_DMA_128;
}

int launch_gemm(int amode, int epi, int variant, const GemmArgs& g, hipStream_t stream) {
  DFOT_REQUIRE(g.A && g.W, DFOT_ERR_ARG, "gemm: null operand");
  if (variant == GEMM_AUTO) variant = gemm_pick_variant(amode, g.M, g.N, g.K, epi == E_F32 && !g.gn_part);  // (the 256x144 ring: plain fp32 epilogue only)
  if (variant == GEMM_DMA_128_KS2 && epi == E_QKV) variant = GEMM_DMA_128;  // the QKV epilogue has workgroup barriers
  if (variant == GEMM_DMA_128x192 && epi == E_QKV) variant = GEMM_DMA_128;
  if (variant == GEMM_DMA_256x192 && epi == E_QKV && g.d != 64) variant = GEMM_DMA_256x256;  // the d = 128 head pairing needs 128-aligned tiles
  const int bm = variant == GEMM_DMA_512x128 ? 512 : (variant == GEMM_DMA_256x256 || variant == GEMM_DMA_256x192 || variant == GEMM_DMA3_256x144) ? 256 : 128;
  DFOT_REQUIRE(g.M > 0 && g.M % bm == 0, DFOT_ERR_SHAPE, "gemm: M=%d must be a positive multiple of %d", g.M, bm);
  DFOT_REQUIRE(g.K > 0 && g.K % BK == 0, DFOT_ERR_SHAPE, "gemm: K=%d must be a positive multiple of %d", g.K, BK);
  DFOT_REQUIRE(g.N > 0 && g.N % (epi == E_F32 ? 4 : 8) == 0, DFOT_ERR_SHAPE, "gemm: N=%d must be a multiple of %d", g.N, epi == E_F32 ? 4 : 8);
  DFOT_REQUIRE((epi == E_QKV || epi == E_QKV_DIT || g.ldo % (epi == E_F32 ? 4 : 8) == 0) && (epi != E_QKV || (g.ldo2 % 8 == 0 && g.split % 64 == 0)),
               DFOT_ERR_SHAPE, "gemm: output row strides must be multiples of %d", epi == E_F32 ? 4 : 8);
  DFOT_REQUIRE(g.ldw == 0 || (amode == A_DENSE && g.ldw >= g.K && g.ldw % 8 == 0), DFOT_ERR_SHAPE, "gemm: ldw=%ld must be >= K and a multiple of 8 (dense A only)", g.ldw);
  if (amode == A_CONV3) {
    DFOT_REQUIRE(g.Cin % BK == 0 && g.K == 9 * g.Cin, DFOT_ERR_SHAPE, "conv3x3: Cin=%d must be a multiple of %d", g.Cin, BK);
    DFOT_REQUIRE(g.zeros != nullptr, DFOT_ERR_ARG, "conv3x3: zero page missing");
    DFOT_REQUIRE(g.H > 0 && g.Wd > 0 && g.M % (g.H * g.Wd) == 0, DFOT_ERR_SHAPE, "conv3x3: M=%d not a whole number of %dx%d images", g.M, g.H, g.Wd);
    DFOT_REQUIRE(!g.live || (g.H * g.Wd) % bm == 0, DFOT_ERR_SHAPE, "conv3x3: live-image flags need whole %d-row tiles per %dx%d image", bm, g.H, g.Wd);
  }
  if (epi == E_QKV) {
    DFOT_REQUIRE(g.q && g.k && g.v && g.qw && g.kw && g.rope_cs && (g.out2 || g.N == g.split), DFOT_ERR_ARG, "qkv epilogue: null pointer");
    DFOT_REQUIRE(!g.raw || (g.ldraw >= g.N && g.ldraw % 8 == 0), DFOT_ERR_SHAPE, "qkv epilogue: raw copy needs ldraw >= N, a multiple of 8");
    DFOT_REQUIRE((g.d == 64 || g.d == 128) && g.heads > 0 && g.split == 3 * g.heads * g.d && g.ntok > 0 && g.M % g.ntok == 0,
                 DFOT_ERR_SHAPE, "qkv epilogue: heads=%d d=%d split=%d ntok=%d M=%d", g.heads, g.d, g.split, g.ntok, g.M);
  }
  if (epi == E_QKV_DIT) {
    DFOT_REQUIRE(g.q && g.k && g.v, DFOT_ERR_ARG, "dit qkv epilogue: null pointer");
    DFOT_REQUIRE(g.d > 0 && g.d % 8 == 0 && g.dstride >= g.d && g.dstride % 8 == 0 && g.heads > 0 && g.N == 3 * g.heads * g.d &&
                     g.ntok > 0 && g.M % g.ntok == 0,
                 DFOT_ERR_SHAPE, "dit qkv epilogue: heads=%d d=%d dstride=%d ntok=%d M=%d N=%d", g.heads, g.d, g.dstride, g.ntok, g.M, g.N);
  }
  if (g.gate) {
    DFOT_REQUIRE(epi == E_F32 && g.gate_rows > 0 && g.M % g.gate_rows == 0 && g.ldg % 4 == 0, DFOT_ERR_SHAPE,
                 "gemm: gate needs the fp32 epilogue, gate_rows dividing M and ldg %% 4 == 0");
  }
  if (g.ksplit > 1 && g.slice_stride > 0)
    DFOT_REQUIRE(!g.bias && !g.resid, DFOT_ERR_ARG, "gemm: split-K into partial outputs takes no bias / residual");
  if (g.ksplit > 1)
    DFOT_REQUIRE(epi == E_F32 && (amode == A_DENSE || g.slice_stride > 0) && !g.bias_rows && (!g.resid || g.resid == g.out_f32) && !g.gate &&
                     !g.gn_part && g.K / BK >= 2 * g.ksplit, DFOT_ERR_ARG,
                 "gemm: split-K over workgroups needs the fp32 epilogue with an in-place residual (or none), no gate, and K >= %d", 2 * g.ksplit * BK);
  if (g.bias_rows) DFOT_REQUIRE((epi == E_F32 || epi == E_BF16) && g.bias, DFOT_ERR_ARG, "gemm: 2-D bias needs a plain epilogue");
  if (g.resid_bf)
    DFOT_REQUIRE(epi == E_BF16 && (amode == A_CONV3 || variant == GEMM_DMA3_256x144) && !g.resid, DFOT_ERR_ARG,
                 "gemm: a bf16 residual needs the bf16 epilogue of a convolution or of the 256x144 ring");
  if (g.tr_rows) {
    DFOT_REQUIRE(epi == E_BF16 && !g.bias && !g.act && !g.gn_part && g.tr_rows % 4 == 0 && g.M % g.tr_rows == 0 && amode == A_DENSE,
                 DFOT_ERR_ARG, "gemm: transposed store needs the plain bf16 epilogue and tr_rows %% 4 == 0 dividing M");
  }
  if (g.gn_part) {
    DFOT_REQUIRE(epi != E_QKV && epi != E_QKV_DIT && (g.gn_cpg == 4 || g.gn_cpg == 8) && g.N == 32 * g.gn_cpg && g.gn_rows_per_bt % 64 == 0 &&
                     g.gn_rows_per_bt % bm == 0,
                 DFOT_ERR_SHAPE, "gemm: fused GroupNorm statistics need N = 32 groups of 4 or 8 channels and whole images per tile");
  }
  if (amode == A_DENSE) {
    switch (epi) {
      case E_F32: return launch_v<A_DENSE, E_F32>(variant, g, stream);
      case E_BF16: return launch_v<A_DENSE, E_BF16>(variant, g, stream);
      case E_QKV: return launch_v<A_DENSE, E_QKV>(variant, g, stream);
      case E_QKV_DIT: return launch_v<A_DENSE, E_QKV_DIT>(variant, g, stream);
    }
  } else if (amode == A_CONV3) {
    switch (epi) {
      case E_F32: return launch_v<A_CONV3, E_F32>(variant, g, stream);
      case E_BF16: return launch_v<A_CONV3, E_BF16>(variant, g, stream);
    }
  }
  set_error("gemm: unsupported mode/epilogue combination %d/%d", amode, epi);
  return DFOT_ERR_ARG;
}

}  // namespace dfot
