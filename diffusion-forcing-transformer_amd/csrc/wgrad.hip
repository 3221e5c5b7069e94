// Weight-gradient GEMM over the token axis, operands in their natural activation layout (gfx950).
//
//     C[M][N] (fp32) = sum_r A[r][m] * B[r][n]         A [rows][lda], B [rows][ldb] bf16, both feature-contiguous
//
// i.e. dW = dY^T X without materialising dY^T / X^T: the generic GEMM (gemm.hip) wants both operands K-contiguous, which for a weight
// gradient means two transposed copies per call (4-7 % of a training step).  Here a tile of 64 token rows x 128 features of each operand
// is staged in LDS as it lies in memory and the MFMA fragments are fetched TRANSPOSED with ds_read_tr16_b64 (the same access the
// attention kernels use for V^T / K^T): v_mfma_f32_32x32x16_bf16 with A = (A tile)^T [m][r], B = (B tile)^T [n][r]; both fragments use
// the same permutation of the 16 rows of a k-step, so the contraction is consistent.
// One workgroup = 4 waves = a 128 x 128 output tile (wave tile 64 x 64 = 2 x 2 MFMA tiles); grid = tiles x K slices, every slice
// stores its partial tile (plain stores) to out + slice * M * N and the caller sums the slices (few output tiles, very long K).
#include "common.h"
#include "dfot_hip.h"
#include "kernels.h"

namespace dfot {
namespace {

constexpr int WG_TR = 64;                 // token rows per staged tile
constexpr int WG_F = 128;                 // features per tile (both operands)
constexpr int WG_ROWB = WG_F * 2 + 16;    // padded LDS row (bytes): row reads never happen here, transposed reads stay conflict-light
constexpr int WG_TILE = WG_TR * WG_ROWB;
constexpr int WG_PER_THREAD = WG_TR * (WG_F / 8) / 256;  // 16-byte chunks per thread per tile

// A fragment of X^T for features c0..c0+31 and the 16 permuted tile rows of step (kt2, s) (as attention_bwd.hip::tr_frag)
__device__ __forceinline__ bf16x8 wg_frag(const char* tile, int c0, int kt2, int s, int lane) {
  const int lh = lane >> 5;
  const int kb = kt2 * 32 + 16 * s + 4 * lh;
  const int q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int col = c0 + 16 * ((lane >> 4) & 1) + 4 * p4;
  const char* a0 = tile + (kb + q4) * WG_ROWB + col * 2;
  const char* a1 = tile + (kb + 8 + q4) * WG_ROWB + col * 2;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a1));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// conv mode (img_h > 0): B row r is taken from pixel r shifted by (sdy, sdx) inside its img_h x img_w image (rows are pixels, row-major per
// image), zero outside -- the operand of tap (sdy, sdx) of a 3x3 convolution's weight gradient, read in place.
// M / N need only be multiples of 8: tiles are guarded (loads beyond M / N read as zero, stores are skipped).
__global__ __launch_bounds__(256) void wgrad_nt_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B, long ldb,
                                                       float* __restrict__ out, int M, int N, long rows, int slices, int img_h, int img_w, int sdy,
                                                       int sdx, int all_taps) {
  // all_taps: blockIdx.y = tap of a 3x3 convolution (shift (tap / 3 - 1, tap % 3 - 1)); partial outputs are laid out [slice][tap][M][N]
  if (all_taps) {
    sdy = (int)blockIdx.y / 3 - 1;
    sdx = (int)blockIdx.y % 3 - 1;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int tiles_n = (N + WG_F - 1) / WG_F;
  const int tile = blockIdx.x / slices, slice = blockIdx.x % slices;
  const int m0 = (tile / tiles_n) * WG_F, n0 = (tile % tiles_n) * WG_F;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  // this slice's range of 64-row tiles
  const long nt_all = rows / WG_TR;
  const long per = nt_all / slices, rem = nt_all % slices;
  const long t0 = slice * per + (slice < rem ? slice : rem), nt = per + (slice < rem ? 1 : 0);
  const bf16* Ab = A + m0;
  const bf16* Bb = B + n0;
  const bf16 zb = f2bf(0.f);
  const bf16x8 zero8 = bf16x8{zb, zb, zb, zb, zb, zb, zb, zb};

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 ra[WG_PER_THREAD], rb[WG_PER_THREAD];
  auto load = [&](long t) {
#pragma unroll
    for (int i = 0; i < WG_PER_THREAD; ++i) {
      const int c = tid + i * 256, row = c / (WG_F / 8), col = (c % (WG_F / 8)) * 8;
      const long r = (t0 + t) * WG_TR + row;
      ra[i] = (m0 + col < M) ? *reinterpret_cast<const bf16x8*>(Ab + r * lda + col) : zero8;
      long rs = r;
      bool ok = n0 + col < N;
      if (img_h > 0) {
        const int x = (int)(r % img_w) + sdx, y = (int)((r / img_w) % img_h) + sdy;
        ok = ok && x >= 0 && x < img_w && y >= 0 && y < img_h;
        rs = r + (long)sdy * img_w + sdx;
      }
      rb[i] = ok ? *reinterpret_cast<const bf16x8*>(Bb + rs * ldb + col) : zero8;
    }
  };
  auto store = [&](int stage) {
    char* sa = smem + stage * 2 * WG_TILE;
    char* sb = sa + WG_TILE;
#pragma unroll
    for (int i = 0; i < WG_PER_THREAD; ++i) {
      const int c = tid + i * 256, row = c / (WG_F / 8), col = (c % (WG_F / 8)) * 8;
      *reinterpret_cast<bf16x8*>(sa + row * WG_ROWB + col * 2) = ra[i];
      *reinterpret_cast<bf16x8*>(sb + row * WG_ROWB + col * 2) = rb[i];
    }
  };
  if (nt > 0) {
    load(0);
    store(0);
  }
  __syncthreads();
  int cur = 0;
  for (long t = 0; t < nt; ++t) {
    const char* sa = smem + cur * 2 * WG_TILE;
    const char* sb = sa + WG_TILE;
    if (t + 1 < nt) load(t + 1);
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 a0 = wg_frag(sa, wm, kt2, s, lane), a1 = wg_frag(sa, wm + 32, kt2, s, lane);
        const bf16x8 b0 = wg_frag(sb, wn, kt2, s, lane), b1 = wg_frag(sb, wn + 32, kt2, s, lane);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      }
    if (t + 1 < nt) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  // C lane layout: column n = lq, rows m = 8g + 4h + j in register 4g + j
  float* o = out + ((long)slice * (all_taps ? 9 : 1) + (all_taps ? (long)blockIdx.y : 0L)) * M * N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mm = m0 + wm + 32 * i + 8 * g + 4 * lh + r, nn = n0 + wn + 32 * j + lq;
          if (mm < M && nn < N) o[(long)mm * N + nn] = acc[i][j][4 * g + r];
        }
}


// ---- the large-tile form ------------------------------------------------------------------------------------------------------------
// (MW x 64) x (NW x 64) output tile, one 64 x 64 wave tile per wave (16 or 12 waves), K tiles of 64 token rows staged by LDS-DMA
// (global_load_lds, 16 bytes per lane) in two stages as they lie in memory: a row of the A tile is MW * 128 bytes of LDS, unpadded
// (the DMA writes wave-contiguous kilobytes), with the 16-byte chunks XOR-swizzled on the SOURCE side so that the four rows a
// transposed read touches fall into different bank groups.  Fragments: ds_read_b64_tr_b16 as above, through inline asm (the
// builtin makes hipcc drain the DMA queue, see attention_v3.hip), double-buffered over the four 16-row steps of a tile.
// Same LDS bytes per FLOP as the 256 x 256 kernel of gemm.hip; twice the wave count and half the staging traffic of the 128 x 128
// form above.  Dense operands only (the convolution taps stay on the form above); M, N multiples of 8 (out-of-range chunks are
// fetched from a zero buffer), rows a multiple of 64.
typedef __attribute__((ext_vector_type(2))) unsigned w2_u32x2;
template <int OFF>
__device__ __forceinline__ w2_u32x2 w2_read_tr16(unsigned addr) {
  w2_u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void w2_wait(w2_u32x2 (&a)[2][2], w2_u32x2 (&b)[2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
}
__device__ __forceinline__ bf16x8 w2_bf16x8(w2_u32x2 lo, w2_u32x2 hi) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}
// chunk swizzle of a row of F features: 512- and 256-byte rows start on the same bank, so rows r, r+1, r+2, r+3 move by 64 bytes each;
// 384-byte rows already alternate between the two bank halves, so only the row pairs move
template <int F>
__device__ __forceinline__ int w2_swz(int row, int chunk) {
  if constexpr (F == 256 || F == 128) return chunk ^ ((row & 3) << 2);
  else return chunk ^ (((row >> 1) & 1) << 2);
}

// CONV: blockIdx.y = tap of a 3x3 convolution; B row r is pixel r shifted by (tap / 3 - 1, tap % 3 - 1) inside its img_h x img_w image,
// zero outside (fetched from the zero buffer); partial outputs [slice][tap][M][N] -- as the all_taps mode of the form above
// KT: token rows per staged K tile (64; 32 for the 4-wave 128 x 128 form: half the LDS per workgroup, so four of them -- 16 waves --
// are resident per CU instead of two; at 8 waves per CU that form ran at 0.18 MFMA utilisation against 0.45 for the 16-wave forms)
template <int MW, int NW, bool CONV, int KT = 64>
__global__ __launch_bounds__(MW * NW * 64) void wgrad_nt_big_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B, long ldb,
                                                                    float* __restrict__ out, int M, int N, long rows, int slices,
                                                                    const bf16* __restrict__ zeros, int img_h, int img_w) {
  constexpr int NWV = MW * NW, FA = MW * 64, FB = NW * 64, CPA = FA / 8, CPB = FB / 8;
  constexpr int RA = FA * 2, RB = FB * 2;             // LDS row pitch (bytes)
  static_assert(KT == 64 || KT == 32, "K tile rows");
  constexpr int TA = KT * RA, TB = KT * RB, STG = TA + TB;
  constexpr int IA = KT * CPA / 64, IB = KT * CPB / 64;  // 1-KiB DMA instructions per tile (KT rows x CP chunks / 64 lanes)
  constexpr int PA = (IA + NWV - 1) / NWV, PB = (IB + NWV - 1) / NWV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int tiles_n = (N + FB - 1) / FB;
  // CONV: 1-D grid, tap fastest in the XCD-remapped order -- the nine taps of a (tile, slice) read the same dy rows and the same x rows
  // shifted by (+-1, +-W); consecutive remapped ids share an XCD and start together, so eight of the nine reads hit that XCD's L2
  // (with the tap in blockIdx.y the nine were `slices` dispatches apart on different XCDs: 7.3 TB/s of L2 misses at 128 channels)
  const int lin = CONV ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tap = CONV ? lin % 9 : 0, bx = CONV ? lin / 9 : lin;
  const int tile = bx / slices, slice = bx % slices;
  const int m0 = (tile / tiles_n) * FA, n0 = (tile % tiles_n) * FB;
  const int wm = (wave / NW) * 64, wn = (wave % NW) * 64;
  const long nt_all = rows / KT;
  const long per = nt_all / slices, rem = nt_all % slices;
  const long t0 = slice * per + (slice < rem ? slice : rem), nt = per + (slice < rem ? 1 : 0);

  // DMA sources of this lane: instruction j of an operand fills LDS bytes [j * 1024, +1024) of its tile = linear chunks j * 64 + lane
  const int sdy = CONV ? tap / 3 - 1 : 0, sdx = CONV ? tap % 3 - 1 : 0;
  const bf16* pa[PA];
  const bf16* pb[PB];
  long sa[PA], sb[PB];
  long rb[PB];    // CONV: the pixel index of this lane's B row in the next tile to fetch
  bool okb[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int q = (wave + i * NWV) * 64 + lane, row = q / CPA, col = w2_swz<FA>(row, q % CPA) * 8;
    const bool ok = m0 + col < M;
    pa[i] = ok ? A + (t0 * KT + row) * lda + m0 + col : zeros;
    sa[i] = ok ? KT * lda : 0;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int q = (wave + i * NWV) * 64 + lane, row = q / CPB, col = w2_swz<FB>(row, q % CPB) * 8;
    const bool ok = n0 + col < N;
    pb[i] = ok ? B + (t0 * KT + row) * ldb + n0 + col : zeros;
    sb[i] = ok ? KT * ldb : 0;
    rb[i] = t0 * KT + row;
    okb[i] = ok;
    if constexpr (CONV) pb[i] = B + n0 + col;  // the row offset is applied per tile
  }
  auto issue = [&](int stage) {
    char* la = smem + stage * STG;
    char* lb = la + TA;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int j = wave + i * NWV;
      if (j < IA) __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(pa[i]), DFOT_LDS_PTR(la + j * 1024), 16, 0, 0);
      pa[i] += sa[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int j = wave + i * NWV;
      if constexpr (CONV) {
        const long r = rb[i];
        // pixel (x, y) of row r: 32-bit arithmetic (launcher: pixels < 2^31), shifts / masks for power-of-two image sizes (64-bit
        // divisions here were two ~100-instruction sequences per DMA instruction and tile, next to 16 MFMAs)
        const unsigned ru = (unsigned)r, uw = (unsigned)img_w, uh = (unsigned)img_h;
        const bool p2w = (uw & (uw - 1)) == 0, p2h = (uh & (uh - 1)) == 0;
        const unsigned rowq = p2w ? ru >> (31 - __builtin_clz(uw)) : ru / uw;
        const int x = (int)(p2w ? (ru & (uw - 1)) : ru - rowq * uw) + sdx, y = (int)(p2h ? (rowq & (uh - 1)) : rowq % uh) + sdy;
        const bool ok = okb[i] && x >= 0 && x < img_w && y >= 0 && y < img_h;
        const bf16* src = ok ? pb[i] + (r + (long)sdy * img_w + sdx) * ldb : zeros;
        if (j < IB) __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(src), DFOT_LDS_PTR(lb + j * 1024), 16, 0, 0);
        rb[i] = r + KT;
      } else {
        if (j < IB) __builtin_amdgcn_global_load_lds(DFOT_GLOBAL_PTR(pb[i]), DFOT_LDS_PTR(lb + j * 1024), 16, 0, 0);
        pb[i] += sb[i];
      }
    }
  };

  // fragment addresses (stage 0): 16 lanes read rows kb + q4 (q4 = 0..3), 8 bytes at feature c0 + 16 g + 4 p4 each
  const int q4 = (lane & 15) >> 2, p4 = lane & 3, g = (lane >> 4) & 1;
  const int frow = 4 * lh + q4;
  unsigned aa[2], ab[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = (wm + 32 * i) / 8 + 2 * g + (p4 >> 1), cb = (wn + 32 * i) / 8 + 2 * g + (p4 >> 1);
    aa[i] = (unsigned)(size_t)(smem) + frow * RA + w2_swz<FA>(frow, ca) * 16 + (p4 & 1) * 8;
    ab[i] = (unsigned)(size_t)(smem) + TA + frow * RB + w2_swz<FB>(frow, cb) * 16 + (p4 & 1) * 8;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  w2_u32x2 fa[2][2][2], fb[2][2][2];  // [buffer][fragment][rows kb.. | kb + 8..]
#define W2_READ(KS, BUF, SO)                                                   \
  fa[BUF][0][0] = w2_read_tr16<(KS) * 16 * RA>(aa[0] + (SO));                  \
  fa[BUF][0][1] = w2_read_tr16<(KS) * 16 * RA + 8 * RA>(aa[0] + (SO));         \
  fa[BUF][1][0] = w2_read_tr16<(KS) * 16 * RA>(aa[1] + (SO));                  \
  fa[BUF][1][1] = w2_read_tr16<(KS) * 16 * RA + 8 * RA>(aa[1] + (SO));         \
  fb[BUF][0][0] = w2_read_tr16<(KS) * 16 * RB>(ab[0] + (SO));                  \
  fb[BUF][0][1] = w2_read_tr16<(KS) * 16 * RB + 8 * RB>(ab[0] + (SO));         \
  fb[BUF][1][0] = w2_read_tr16<(KS) * 16 * RB>(ab[1] + (SO));                  \
  fb[BUF][1][1] = w2_read_tr16<(KS) * 16 * RB + 8 * RB>(ab[1] + (SO));
#define W2_MMA(BUF)                                                                                        \
  {                                                                                                        \
    const bf16x8 a0 = w2_bf16x8(fa[BUF][0][0], fa[BUF][0][1]), a1 = w2_bf16x8(fa[BUF][1][0], fa[BUF][1][1]); \
    const bf16x8 b0 = w2_bf16x8(fb[BUF][0][0], fb[BUF][0][1]), b1 = w2_bf16x8(fb[BUF][1][0], fb[BUF][1][1]); \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);                       \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);                       \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);                       \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);                       \
  }

  if (nt > 0) issue(0);
  for (long t = 0; t < nt; ++t) {
    // tile t has landed for every wave, and every wave is done reading the other stage
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 1 < nt) issue((int)((t + 1) & 1));
    const unsigned so = (t & 1) ? (unsigned)STG : 0u;
    W2_READ(0, 0, so)
    w2_wait(fa[0], fb[0]);
    W2_READ(1, 1, so)
    W2_MMA(0)
    w2_wait(fa[1], fb[1]);
    if constexpr (KT == 64) {
      W2_READ(2, 0, so)
      W2_MMA(1)
      w2_wait(fa[0], fb[0]);
      W2_READ(3, 1, so)
      W2_MMA(0)
      w2_wait(fa[1], fb[1]);
    }
    W2_MMA(1)
  }
#undef W2_READ
#undef W2_MMA
  // C lane layout: column n = lq, rows m = 8g + 4h + j in register 4g + j
  float* o = out + (CONV ? (long)slice * 9 + tap : (long)slice) * M * N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mm = m0 + wm + 32 * i + 8 * gq + 4 * lh + r, nn = n0 + wn + 32 * j + lq;
          if (mm < M && nn < N) o[(long)mm * N + nn] = acc[i][j][4 * gq + r];
        }
}

const bf16* g_w2_zeros = nullptr;

template <int MW, int NW, bool CONV, int KT = 64>
int launch_big(const bf16* a, long lda, const bf16* b, long ldb, float* out, int m, int n, long rows, int slices, hipStream_t s, int img_h = 0,
               int img_w = 0) {
  constexpr int FA = MW * 64, FB = NW * 64, LDS = 2 * KT * (FA + FB) * 2;
  auto kern = wgrad_nt_big_kernel<MW, NW, CONV, KT>;
  static bool attr_set = false;
  if (!attr_set) {
    DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  if (!g_w2_zeros) {
    void* z = nullptr;
    DFOT_CHECK_HIP(hipMalloc(&z, 256));
    DFOT_CHECK_HIP(hipMemset(z, 0, 256));
    g_w2_zeros = (const bf16*)z;
  }
  const int tiles = ((m + FA - 1) / FA) * ((n + FB - 1) / FB);
  hipLaunchKernelGGL(kern, dim3(tiles * slices * (CONV ? 9 : 1)), dim3(MW * NW * 64), LDS, s, a, lda, b, ldb, out, m, n, rows, slices, g_w2_zeros, img_h,
                     img_w);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

}  // namespace

// out [slices][M][N] fp32 partial products (slices >= 1; the caller sums them); M, N multiples of 8, rows a multiple of 64.
// img_h > 0: conv mode, see the kernel.
int launch_wgrad_nt(const bf16* a, long lda, const bf16* b, long ldb, float* out, int m, int n, long rows, int slices, hipStream_t s, int img_h,
                    int img_w, int sdy, int sdx, int all_taps) {
  DFOT_REQUIRE(a && b && out, DFOT_ERR_ARG, "wgrad_nt: null pointer");
  DFOT_REQUIRE(m > 0 && n > 0 && m % 8 == 0 && n % 8 == 0 && rows > 0 && rows % WG_TR == 0 && lda % 8 == 0 && ldb % 8 == 0 && slices >= 1 &&
                   slices <= rows / WG_TR && (img_h == 0 || (img_w > 0 && rows % ((long)img_h * img_w) == 0)),
               DFOT_ERR_SHAPE, "wgrad_nt: M=%d N=%d must be multiples of 8, rows=%ld of 64 (and whole images in conv mode)", m, n, rows);
  const int lds = 4 * WG_TILE;
  static bool attr_set = false;
  if (!attr_set) {
    DFOT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_nt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  DFOT_REQUIRE(!all_taps || img_h > 0, DFOT_ERR_ARG, "wgrad_nt: all_taps needs conv mode");
  hipLaunchKernelGGL(wgrad_nt_kernel, dim3(((m + WG_F - 1) / WG_F) * ((n + WG_F - 1) / WG_F) * slices, all_taps ? 9 : 1), dim3(256), lds, s, a, lda, b,
                     ldb, out, m, n, rows, slices, img_h, img_w, sdy, sdx, all_taps);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// Tile form and K slices for one weight gradient: the candidates are priced as rounds of resident workgroups x K tiles per slice x
// the time of one K tile, plus the traffic of the partial outputs; `max_slices` bounds the workspace (slices * M * N floats).
// Per-workgroup K-tile times from the kernel trace of the RE10K step (round 2): 128 x 128 form 1.9 us (two resident per CU), large
// forms ~1.2 us per 64 x 64 wave tile of the 256 x 256 one.
WgradPlan wgrad_plan(int m, int n, long rows, long max_slices) {
  static const int force = tuning_flag("WGRAD_FORM", -1);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  }
  struct Form { int fa, fb, per_cu; double t_us; };
  const Form forms[4] = {{128, 128, 2, 1.9}, {256, 256, 1, 1.95}, {256, 192, 1, 1.5}, {192, 256, 1, 1.5}};
  const long kt = rows / 64;
  WgradPlan best{0, 1};
  double best_cost = 1e30;
  for (int f = 0; f < 4; ++f) {
    if (force >= 0 && f != force) continue;
    const long tiles = (long)((m + forms[f].fa - 1) / forms[f].fa) * ((n + forms[f].fb - 1) / forms[f].fb);
    const long slots = (long)cus * forms[f].per_cu;
    for (long sl = 1; sl <= 64 && sl <= max_slices && (sl == 1 || kt / sl >= 4); ++sl) {
      const long rounds = (tiles * sl + slots - 1) / slots;
      const double cost = (double)rounds * (double)((kt + sl - 1) / sl) * forms[f].t_us + (sl > 1 ? (double)sl * m * n * 8.0 / 4.0e6 : 0.0);
      if (cost < best_cost) {
        best_cost = cost;
        best = WgradPlan{f, (int)sl};
      }
    }
  }
  return best;
}

// all nine taps of a 3x3 convolution's weight gradient in one launch (a = dy [pixels][co], b = x [pixels][ci], out [slices][9][co][ci]):
// LDS-DMA form with a 128 x 128 (co, ci <= 128) or a 256 x 256 tile
int launch_wgrad_conv_taps(const bf16* dy, const bf16* x, float* out, int co, int ci, long pix, int slices, int img_h, int img_w, hipStream_t s) {
  DFOT_REQUIRE(dy && x && out && co % 8 == 0 && ci % 8 == 0 && pix % 64 == 0 && pix < (1L << 31) && slices >= 1 && slices <= pix / 64 && img_h > 0 && img_w > 0 &&
                   pix % ((long)img_h * img_w) == 0,
               DFOT_ERR_SHAPE, "wgrad_conv_taps: %d -> %d channels, %ld pixels unsupported", ci, co, pix);
  if (co <= 128 && ci <= 128) return launch_big<2, 2, true>(dy, co, x, ci, out, co, ci, pix, slices, s, img_h, img_w);
  return launch_big<4, 4, true>(dy, co, x, ci, out, co, ci, pix, slices, s, img_h, img_w);
}
// workgroups per tap and slice of launch_wgrad_conv_taps, and the workgroup count to aim for with K slices (two rounds of the
// 128 x 128 forms, two resident per CU; one round of the 256 x 256 form, whose partial outputs are four times as large)
int wgrad_conv_tiles(int co, int ci, int* target) {
  const int f = (co <= 128 && ci <= 128) ? 128 : 256;
  if (target) *target = f == 128 ? 1024 : 256;
  return ((co + f - 1) / f) * ((ci + f - 1) / f);
}

// out: [plan.slices][M][N] partial outputs (the caller sums them when slices > 1)
int launch_wgrad_nt_plan(const bf16* a, long lda, const bf16* b, long ldb, float* out, int m, int n, long rows, WgradPlan plan, hipStream_t s) {
  if (plan.form == 0) return launch_wgrad_nt(a, lda, b, ldb, out, m, n, rows, plan.slices, s);
  DFOT_REQUIRE(a && b && out, DFOT_ERR_ARG, "wgrad_nt: null pointer");
  DFOT_REQUIRE(m > 0 && n > 0 && m % 8 == 0 && n % 8 == 0 && rows > 0 && rows % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && plan.slices >= 1 &&
                   plan.slices <= rows / 64,
               DFOT_ERR_SHAPE, "wgrad_nt: M=%d N=%d must be multiples of 8, rows=%ld of 64", m, n, rows);
  switch (plan.form) {
    case 1: return launch_big<4, 4, false>(a, lda, b, ldb, out, m, n, rows, plan.slices, s);
    case 2: return launch_big<4, 3, false>(a, lda, b, ldb, out, m, n, rows, plan.slices, s);
    case 3: return launch_big<3, 4, false>(a, lda, b, ldb, out, m, n, rows, plan.slices, s);
  }
  DFOT_REQUIRE(false, DFOT_ERR_ARG, "wgrad_nt: unknown tile form %d", plan.form);
}

}  // namespace dfot
