// bf16 MFMA GEMM / implicit-GEMM conv3x3 with fused epilogues (gfx950).
#pragma once
#include "common.h"

namespace dfot {

enum AMode { A_DENSE = 0, A_CONV3 = 1 };
enum Epi { E_F32 = 0, E_BF16 = 1, E_QKV = 2, E_QKV_DIT = 3 };

struct GemmArgs {
  // operands: A [M][K] bf16 (dense: row stride lda; conv: NHWC image [BT][H][Wd][Cin], K = 9*Cin tap-major)
  const bf16* A = nullptr;
  long lda = 0;
  const bf16* W = nullptr;  // [N][K] bf16 (torch Linear layout; conv: [Cout][tap][Cin])
  long ldw = 0;             // dense A: row stride of W in elements (0 = K); > K when the GEMM contracts a K sub-range of a wider matrix
  int M = 0, N = 0, K = 0;
  int H = 0, Wd = 0, Cin = 0;
  const bf16* zeros = nullptr;  // >= 128 B of zeros (conv padding source)
  // conv: optional per-image flags (device uint8 [M / (H * Wd)]); the tiles of an image whose flag is 0 exit at once (no loads, no stores,
  // no GroupNorm partials): images whose output the caller discards.  A tile must lie inside one image (launcher)
  const uint8_t* live = nullptr;
  // epilogue
  const float* bias = nullptr;
  float* out_f32 = nullptr;
  bf16* out_bf16 = nullptr;
  long ldo = 0;
  const float* resid = nullptr;  // E_F32: optional fp32 residual, same ld as out
  const bf16* resid_bf = nullptr;  // E_BF16: optional bf16 residual, same ld as out (may be out itself: each element is read before it is written)
  // E_F32: optional per-(frame, column) gate (DiT AdaLN-Zero): out = resid + gate[row / gate_rows][col] * (acc + bias)
  //   (gate_index: optional indirection, gate row = gate_index[row / gate_rows] -- the per-level modulation table)
  const float* gate = nullptr;
  const int* gate_index = nullptr;
  long ldg = 0;
  int gate_rows = 0;
  int act = 0;  // E_BF16: 0 = none, 1 = GELU(tanh approximation), 2 = SiLU applied before the bf16 store
  bf16* pre_act = nullptr;  // E_BF16 with act: optional second output, the value BEFORE the activation (training keeps both)
  // E_QKV: optional copy of the raw projection output (bias added, before QK-norm / RoPE / SiLU) as bf16 [M][ldraw]: the training forward
  // keeps it for the backward of the norm and the activation (uvit_train.py) while q, k, v and SiLU(mlp_h) come out of the same epilogue
  bf16* raw = nullptr;
  long ldraw = 0;
  int bias_rows = 0;  // E_F32 / E_BF16: > 0 = two-dimensional bias[(row % bias_rows)][N] (MatrixAttention), < 0 = bias[row / -bias_rows][N], else bias[N]
  // E_BF16: tr_rows = R > 0 stores the result transposed inside consecutive groups of R rows ("frames"):
  // out[(row / R)][col][row % R] -- the left factor of a matrix_mul contracts the ROW index of its operand
  int tr_rows = 0;
  // optional fused GroupNorm(32) partial statistics of the OUTPUT (E_F32 / E_BF16, N == channel count):
  // gn_part[((bt*slots + slot)*32 + group)*2 + {sum,sumsq}], slot = 64-row block index within the image
  float* gn_part = nullptr;
  int gn_rows_per_bt = 0;
  int gn_cpg = 0;                // channels per group: 4 or 8
  // E_QKV (fused_attn_mlp_proj): columns [0,C) q | [C,2C) k | [2C,3C) v | [3C,7C) MLP hidden, C = heads*d = split/3.
  //   q,k: per-head RMSNorm (weights qw/kw) + RoPE (rope_cs[tok][d/2][2] = cos,sin) (+ q *= qscale), written with v
  //   as [B][heads][ntok][d] bf16 for the attention kernel; MLP half: SiLU -> out2[m][col-split] (row stride ldo2)
  bf16* out2 = nullptr;
  long ldo2 = 0;
  int split = 0;
  bf16 *q = nullptr, *k = nullptr, *v = nullptr;
  const float *qw = nullptr, *kw = nullptr, *rope_cs = nullptr;
  int heads = 0, d = 0, ntok = 0;
  float qscale = 1.f;
  // E_QKV_DIT (DiT Attention.qkv): columns [0,C) q | [C,2C) k | [2C,3C) v, C = heads*d, d % 8 == 0 (no QK norm);
  //   q,k: RoPE (+ q *= qscale); all three written as [B][heads][ntok][dstride] bf16, dstride >= d (pad columns are
  //   never written: the caller zeroes them once)
  int dstride = 0;
  float eps = 1e-6f;
  int xcd = 1;  // XCD-aware tile order
  int persist = 0;  // > 0: persistent workgroups (grid = resident workgroups, each loops over tiles)
  // E_F32, dense A, no gate: ksplit = S > 1 launches S workgroups per tile, each over one slice of K, which ADD their partial
  // tile to out_f32 with atomics: out must already hold the value to add to -- zeros, or the residual when resid == out_f32
  // (in-place residual GEMMs); the bias is added by slice 0.  For few output tiles and a long K
  int ksplit = 1;
  // slice_stride > 0 (elements): slice s stores its partial tile with plain stores to out_f32 + s * slice_stride instead of
  // atomics (the caller reduces the S partial outputs afterwards); no bias / residual / gate then
  long slice_stride = 0;
};

enum GemmVariant {
  GEMM_REGS_128 = 0,  // 128x128 tile, register staging, 2 LDS stages (A/B reference)
  GEMM_DMA_128 = 1,   // 128x128 tile, LDS-DMA, 2 stages
  // (numbers 2, 3, 5, 7, 13 were tile forms measured slower on every model shape and removed in round 4: 128x128 / 256x128 three-stage
  // rings, 256x128 two-stage, 256x256 with 128x64 wave tiles, the two-stage 256x144 -- DESIGN.md section 7)
  GEMM_DMA_256x256 = 4,  // 256x256 tile (16 waves), LDS-DMA, 2 stages: half the L2->LDS operand traffic per FLOP
  GEMM_DMA_512x128 = 6,  // 512x128 tile (16 waves), 2 stages (N = 128 convolutions)
  GEMM_DMA_128_KS2 = 8,  // 128x128 tile, 8 waves: two k-groups alternate k-tiles (intra-workgroup split-K)
  GEMM_DMA_256x192 = 9,  // 256x192 tile (12 waves), 2 stages, dense A only: N = multiples of 192 (1152, 3456) without padding
  GEMM_DMA_128x192 = 10,  // 128x192 tile (6 waves), 2 stages, dense A: long-K GEMMs with slightly more 128x128 tiles than CUs
  // 256x144 tile (12 waves of 64x48), 3-stage ring (150 KB), plain fp32 epilogue: N = multiples of 144 (576, 1152) give M/256 x N/144
  // tiles = exactly one per CU for the level-2 out-projection (256) and, with two K slices, for the level-3 one (128 x 2); the long-K
  // loop no longer waits on the one k-tile a 2-stage loop has in flight
  GEMM_DMA3_256x144 = 14,
  GEMM_AUTO = -1      // pick by shape (gemm_pick_variant)
};
int gemm_pick_variant(int amode, int m, int n, int k, bool plain_f32 = false);  // plain_f32: the caller's epilogue is E_F32 without GroupNorm partials
// returns DFOT_OK / error; validates the divisibility contract before launching
int launch_gemm(int amode, int epi, int variant, const GemmArgs& g, hipStream_t stream);

}  // namespace dfot
