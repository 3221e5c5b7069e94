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

}  // namespace dfot
