// bf16 MFMA GEMM / implicit-GEMM conv3x3 with fused epilogues (gfx950).
#pragma once
#include "common.h"

namespace dfot {

enum AMode { A_DENSE = 0, A_CONV3 = 1 };
enum Epi { E_F32 = 0, E_BF16 = 1, E_QKV = 2 };

struct GemmArgs {
  // operands: A [M][K] bf16 (dense: row stride lda; conv: NHWC image [BT][H][Wd][Cin], K = 9*Cin tap-major)
  const bf16* A = nullptr;
  long lda = 0;
  const bf16* W = nullptr;  // [N][K] bf16 (torch Linear layout; conv: [Cout][tap][Cin])
  int M = 0, N = 0, K = 0;
  int H = 0, Wd = 0, Cin = 0;
  const bf16* zeros = nullptr;  // >= 128 B of zeros (conv padding source)
  // epilogue
  const float* bias = nullptr;
  float* out_f32 = nullptr;
  bf16* out_bf16 = nullptr;
  long ldo = 0;
  const float* resid = nullptr;  // E_F32: optional fp32 residual, same ld as out
  bf16* out2 = nullptr;          // E_QKV: columns >= split -> silu -> out2[m][col-split]
  long ldo2 = 0;
  int split = 0;
};

// returns DFOT_OK / error; validates the divisibility contract before launching
int launch_gemm(int amode, int epi, bool lds_dma, const GemmArgs& g, hipStream_t stream);

}  // namespace dfot
