// Backward building blocks for the UViT3DPose backbone (training path of BASELINE config 5, built op by op; the orchestration of a
// whole UViT training step is not written yet).  Included at the end of dit.hip (shares the training helpers of dit_train.inl).
//
// conv3x3 (padding 1, channels-last activations [BT][H][W][C] bf16), y = conv(x, W) + b with W [Co][Ci][3][3]:
//   dx = conv(dy, W')         W'[ci][2-ky][2-kx][co] = W[co][ci][ky][kx]      -> the forward's implicit-GEMM kernel on repacked weights
//   dW[co][ci][ky][kx] = sum_pix dy[pix][co] x[pix + (ky-1, kx-1)][ci]        -> 9 GEMMs over the pixel axis: dy^T [Co][pix] against
//                                                                                 the tap-shifted x^T [Ci][pix] (zero outside the image),
//                                                                                 K split over workgroups into partial buffers
//   db[co] = sum_pix dy[pix][co]
// Replaces torch autograd through F.conv2d in ResBlock / Downsample / Upsample (algorithms/dfot/backbones/u_vit/u_vit_blocks.py:16-93).
namespace dfot {
namespace {

// W [Co][Ci][3][3] fp32 -> W' [Ci][(2-ky)*3 + (2-kx)][Co] bf16: the data-gradient convolution's weights in the forward kernel's layout
__global__ void pack_conv3_dgrad_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int co, int ci) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)co * ci * 9) return;
  const int o = (int)(i % co);
  const int tap = (int)((i / co) % 9);
  const long c = i / ((long)co * 9);
  const int ky = 2 - tap / 3, kx = 2 - tap % 3;
  dst[i] = f2bf(src[((long)o * ci + c) * 9 + ky * 3 + kx]);
}

// dst[c][pix] = x[pix + (dy, dx)][c] inside the image, else 0   (x [BT][H][W][C] bf16; 64 pixels x 64 channels per workgroup)
__global__ __launch_bounds__(256) void transpose_shift_kernel(const bf16* __restrict__ x, bf16* __restrict__ dst, long pix, int H, int W, int C,
                                                              int dy, int dx) {
  __shared__ bf16 tile[64][66];
  const long p0 = (long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long p = p0 + ty * 16 + i;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const int sy = yy + dy, sx = xx + dx;
    const bool ok = sy >= 0 && sy < H && sx >= 0 && sx < W;
    tile[ty * 16 + i][tx] = ok ? x[(p + (long)dy * W + dx) * C + c0 + tx] : f2bf(0.f);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) dst[(long)(c0 + ty * 16 + i) * pix + p0 + tx] = tile[tx][ty * 16 + i];
}

// tmp [9][Co][Ci] fp32 -> dW [Co][Ci][3][3]
__global__ void conv_wgrad_repack_kernel(const float* __restrict__ tmp, float* __restrict__ dw, int co, int ci) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)co * ci * 9) return;
  const int tap = (int)(i % 9);
  const long oc = i / 9;  // co * ci + c
  dw[i] = tmp[(long)tap * co * ci + oc];
}

struct ConvBwdScratch {  // sized by the caller for the largest convolution it differentiates
  bf16* dyT = nullptr;   // [Co][pix]
  bf16* xT = nullptr;    // [Ci][pix]
  float* taps = nullptr; // [9][Co][Ci]
  float* ws = nullptr;   // split-K partial tiles
  size_t ws_floats = 0;
  const bf16* zeros = nullptr;
};

// dx (fp32 [pix][Ci], optional), dW (fp32 [Co][Ci][3][3]), db (fp32 [Co], optional; must be zeroed by the caller)
int conv3_backward(const bf16* x, const bf16* dy, const bf16* w_dgrad, float* dx, float* dw, float* db, int bt, int H, int W, int ci, int co,
                   const ConvBwdScratch& sc, hipStream_t s) {
  const long pix = (long)bt * H * W;
  DFOT_REQUIRE(ci % 64 == 0 && co % 64 == 0 && pix % 64 == 0, DFOT_ERR_SHAPE, "conv3_backward: channels %d -> %d and %ld pixels must be multiples of 64", ci, co, pix);
  int rc = 0;
  if (dx) {
    GemmArgs g;
    g.zeros = sc.zeros;
    g.A = dy; g.W = w_dgrad; g.M = (int)pix; g.N = ci; g.K = 9 * co; g.H = H; g.Wd = W; g.Cin = co; g.out_f32 = dx; g.ldo = ci;
    if ((rc = launch_gemm(A_CONV3, E_F32, GEMM_AUTO, g, s))) return rc;
  }
  if ((rc = tr_transpose(dy, sc.dyT, (int)pix, co, s))) return rc;
  if (db) {
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cdiv(co, 256), cdiv(pix, 128)), dim3(256), 0, s, dy, db, pix, co, (long)co);
    DFOT_CHECK_HIP(hipGetLastError());
  }
  const int mpad = (co + 127) / 128 * 128;  // GEMM rows come in 128s: dyT is allocated (and zero) up to mpad rows
  for (int tap = 0; tap < 9; ++tap) {
    hipLaunchKernelGGL(transpose_shift_kernel, dim3(ci / 64, (int)(pix / 64)), dim3(256), 0, s, x, sc.xT, pix, H, W, ci, tap / 3 - 1, tap % 3 - 1);
    DFOT_CHECK_HIP(hipGetLastError());
    // few output tiles, K = pixels: split over workgroups into partial buffers
    const long tiles = (long)(mpad / 128) * ((ci + 127) / 128);
    int split = (int)(256 / tiles);
    split = split < 1 ? 1 : (split > 256 ? 256 : split);
    while (split > 1 && pix / 64 < 4L * split) --split;
    while (split > 1 && (size_t)split * mpad * ci > sc.ws_floats) --split;
    GemmArgs g;
    g.A = sc.dyT; g.lda = pix; g.W = sc.xT; g.M = mpad; g.N = ci; g.K = (int)pix; g.ldo = ci;
    float* out = sc.taps + (long)tap * co * ci;
    DFOT_REQUIRE((size_t)split * mpad * ci <= sc.ws_floats, DFOT_ERR_STATE, "conv3_backward: split-K workspace too small");
    g.out_f32 = sc.ws; g.ksplit = split; g.slice_stride = (long)mpad * ci;  // partial tiles (padded rows included) land in the workspace
    if ((rc = launch_gemm(A_DENSE, E_F32, GEMM_DMA_128, g, s))) return rc;
    hipLaunchKernelGGL(slices_sum_kernel, dim3(cdiv((long)co * ci / 4, 256)), dim3(256), 0, s, sc.ws, out, (long)co * ci / 4, split, (long)mpad * ci);
    DFOT_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(conv_wgrad_repack_kernel, dim3(cdiv((long)co * ci * 9, 256)), dim3(256), 0, s, sc.taps, dw, co, ci);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

}  // namespace
}  // namespace dfot

extern "C" {
using namespace dfot;

// test entry: x, dy bf16 channels-last; w fp32 [Co][Ci][3][3]; dx fp32 [pix][Ci]; dw fp32 [Co][Ci][3][3]; db fp32 [Co]
int dfot_op_conv3x3_bwd(const void* x, const void* dy, const float* w, float* dx, float* dw, float* db, int bt, int hh, int ww, int cin, int cout,
                        void* stream) {
  DFOT_REQUIRE(x && dy && w && dx && dw && db, DFOT_ERR_ARG, "op_conv3x3_bwd: null argument");
  hipStream_t s = (hipStream_t)stream;
  const long pix = (long)bt * hh * ww;
  const int mpad = (cout + 127) / 128 * 128;
  ConvBwdScratch sc;
  bf16* wd = nullptr;
  void* zeros = nullptr;
  sc.ws_floats = (size_t)64 * mpad * cin;
  DFOT_CHECK_HIP(hipMalloc(&sc.dyT, (size_t)mpad * pix * sizeof(bf16)));
  DFOT_CHECK_HIP(hipMalloc(&sc.xT, (size_t)cin * pix * sizeof(bf16)));
  DFOT_CHECK_HIP(hipMalloc(&sc.taps, (size_t)9 * cout * cin * sizeof(float)));
  DFOT_CHECK_HIP(hipMalloc(&sc.ws, sc.ws_floats * sizeof(float)));
  DFOT_CHECK_HIP(hipMalloc(&wd, (size_t)9 * cout * cin * sizeof(bf16)));
  DFOT_CHECK_HIP(hipMalloc(&zeros, 256));
  DFOT_CHECK_HIP(hipMemsetAsync(zeros, 0, 256, s));
  DFOT_CHECK_HIP(hipMemsetAsync(sc.dyT, 0, (size_t)mpad * pix * sizeof(bf16), s));
  DFOT_CHECK_HIP(hipMemsetAsync(db, 0, (size_t)cout * sizeof(float), s));
  sc.zeros = (const bf16*)zeros;
  hipLaunchKernelGGL(pack_conv3_dgrad_kernel, dim3(cdiv((long)cout * cin * 9, 256)), dim3(256), 0, s, w, wd, cout, cin);
  int rc = conv3_backward((const bf16*)x, (const bf16*)dy, wd, dx, dw, db, bt, hh, ww, cin, cout, sc, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(sc.dyT); (void)hipFree(sc.xT); (void)hipFree(sc.taps); (void)hipFree(sc.ws); (void)hipFree(wd); (void)hipFree(zeros);
  return rc;
}

}  // extern "C"
