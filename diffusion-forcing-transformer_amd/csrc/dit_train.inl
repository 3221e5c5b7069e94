// Training path of the DiT backbones: DiT3D ("full", rope_3d: the README @DiT/XL K600 model, with or without the MLP branch) and
// DifferenceDiT3D (factorized matrix attention, the bash/k600 model):
// forward with saved activations, hand-written backward, gradients in one flat fp32 buffer (reference parameter order).
// Included at the end of dit.hip (same translation unit: shares its kernels).
//
// Replaces, for this model, torch autograd through DiT3D.forward / DiTBlock.forward (dit3d.py:153-192, dit_blocks.py:408-437,
// 488-542) in DFoTVideo.training_step (dfot_video.py:41-75).  Layout of one block (fork semantics):
//     m = LN(x) (1 + scale) + shift ;  qkv = m Wqkv^T + b ; o = attention(rope(q), rope(k), v) ; a = o Wp^T + bp ; y = m + gate a
// Backward of a block for dY (fp32):   da = dY gate, dgate = sum_rows dY a ; dO = da Wp ; dWp = da^T o ; (dq,dk,dv) = attn_bwd ;
//     dqkv = pack(rope^-1(dq), rope^-1(dk), dv) ; dm = dY + dqkv Wqkv ; dWqkv = dqkv^T m ; dx = LN_bwd(dm (1+scale)),
//     dshift = sum_rows dm, dscale = sum_rows dm xhat.  The per-frame sums land in dmod[frame][ldt], whose GEMMs with
//     SiLU(c) give the gradients of every modulation Linear and of the noise-level embedding MLP.
// Weight gradients are GEMMs over the token axis (K = rows): both operands are transposed copies (HBM-bound passes).
namespace dfot {
namespace {

struct TrainBlock {
  bool matrix = false;  // false: DiTBlock (token attention); true: MatrixDiTBlock (every frame is one token, factorized projections)
  int mh = 0;           // width of the block's MLP branch (0: none)
  long o_mod_w = 0, o_mod_b = 0, o_qkv_w = 0, o_qkv_b = 0, o_proj_w = 0, o_proj_b = 0;  // offsets into the flat parameter / gradient buffers
  long mod = 0;                                                 // column of this block's (shift|scale|gate) in the table
  bf16 *w_qkv = nullptr, *w_qkvT = nullptr, *w_proj = nullptr, *w_projT = nullptr;  // bf16 compute copies: [out][in] and [in][out]
  // saved activations
  float* x_in = nullptr;   // [rows][hd] residual stream entering the block
  bf16 *m = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *o = nullptr, *a = nullptr;
  float* lse = nullptr;
  // MLP branch: m2 = LN(x_mid)(1+scale2)+shift2 ; u = m2 W1^T + b1 ; y = GELU(u) W2^T + b2 ; out = m2 + gate2 y
  long o_mod2_w = 0, o_mod2_b = 0, o_fc1_w = 0, o_fc1_b = 0, o_fc2_w = 0, o_fc2_b = 0, mod2 = 0;
  bf16 *w_fc1 = nullptr, *w_fc1T = nullptr, *w_fc2 = nullptr, *w_fc2T = nullptr;
  float* x_mid = nullptr;
  bf16 *m2 = nullptr, *u = nullptr, *hact = nullptr, *y = nullptr;
  // MatrixDiTBlock: qkv = U^T m V + bias[E][3h] per frame, o = attention over the frames, a = U'^T o V' + bias'[P][h]
  // parameters are stored (in, out): qkv_u [P][E], qkv_v [h][3h], proj_u [E][P], proj_v [h][h]
  long o_qkv_u = 0, o_qkv_v = 0, o_qkv_bias = -1, o_proj_u = 0, o_proj_v = 0, o_proj_bias = -1;
  bf16 *u_s = nullptr, *u_t = nullptr, *v_s = nullptr, *v_t = nullptr, *pu_s = nullptr, *pu_t = nullptr, *pv_s = nullptr, *pv_t = nullptr;  // as stored / transposed
  bf16 *w1 = nullptr, *z = nullptr, *o2 = nullptr, *sfac = nullptr;  // U^T m [frames*E][h], qkv [frames*E][3h], attention out, U'^T o [rows][h]
};

__global__ void tr_features_kernel(const float* __restrict__ freqs, const int* __restrict__ levels, float* __restrict__ feat, int frames,
                                   int dim, int max_level) {
  const int half = dim / 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)frames * dim) return;
  const int f = (int)(i / dim), c = (int)(i % dim);
  int lv = levels[f];
  lv = lv < 0 ? 0 : (lv > max_level ? max_level : lv);
  const float a = __fmul_rn((float)lv, freqs[c < half ? c : c - half]);
  feat[i] = c < half ? cosf(a) : sinf(a);
}
__global__ void iota_kernel(int* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}
__global__ void silu_fwd_kernel(const float* __restrict__ h, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = silu_f(h[i]);
}
// dh = dy * SiLU'(h)
__global__ void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ h, float* __restrict__ dh, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = h[i], sg = 1.0f / (1.0f + __expf(-x));
  dh[i] = dy[i] * sg * (1.0f + x * (1.0f - sg));
}
// GELU (tanh approximation, as the GEMM epilogue of the inference path): h = gelu(u) ; optionally du = dh * gelu'(u) in place of dh
__global__ void gelu_kernel(const bf16* __restrict__ u, bf16* __restrict__ h, bf16* __restrict__ dh_to_du, long n8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const bf16x8 uv = *reinterpret_cast<const bf16x8*>(u + i * 8);
  bf16x8 hv, gv;
  if (dh_to_du) gv = *reinterpret_cast<const bf16x8*>(dh_to_du + i * 8);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = bf2f(uv[j]);
    const float t = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float sg = 1.0f - 1.0f / (1.0f + __expf(2.0f * t));  // 0.5 (1 + tanh t)
    hv[j] = f2bf(x * sg);
    if (dh_to_du) {
      const float dt = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
      gv[j] = f2bf(bf2f(gv[j]) * (sg + x * 2.0f * sg * (1.0f - sg) * dt));  // d/dx [x s(x)], s = sigmoid(2t): s' = 2 s (1-s) t'
    }
  }
  if (h) *reinterpret_cast<bf16x8*>(h + i * 8) = hv;
  if (dh_to_du) *reinterpret_cast<bf16x8*>(dh_to_du + i * 8) = gv;
}

// dW[o][k] = sum_f dy[f][o] x[f][k] ; db[o] = sum_f dy[f][o]   (frames is small: one thread per (o, k))
__global__ void small_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dW, float* __restrict__ db,
                                   int frames, int odim, int kdim) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)odim * kdim) return;
  const int o = (int)(i / kdim), k = (int)(i % kdim);
  float acc = 0.f, bacc = 0.f;
  for (int f = 0; f < frames; ++f) {
    const float g = dy[(long)f * odim + o];
    acc += g * x[(long)f * kdim + k];
    bacc += g;
  }
  dW[i] = acc;
  if (k == 0) db[o] = bacc;
}
// dx[f][k] = sum_o dy[f][o] w[o][k]
__global__ void small_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int frames, int odim,
                                   int kdim) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)frames * kdim) return;
  const int f = (int)(i / kdim), k = (int)(i % kdim);
  float acc = 0.f;
  for (int o = 0; o < odim; ++o) acc += dy[(long)f * odim + o] * w[(long)o * kdim + k];
  dx[i] = acc;
}
// out[c] = sum over the first `frames` rows of src[f][c]
__global__ void frames_colsum_kernel(const float* __restrict__ src, float* __restrict__ out, int frames, long ld) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  float acc = 0.f;
  for (int f = 0; f < frames; ++f) acc += src[(long)f * ld + c];
  out[c] = acc;
}

// y = m + gate[frame] * a   (x holds m, fp32)
__global__ void gate_combine_kernel(const float* __restrict__ x, float* __restrict__ y, const bf16* __restrict__ a, const float* __restrict__ table,
                                    long ldt, long off, int hidden, int rows_per_frame, long total4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const long e = i * 4, row = e / hidden;
  const int c = (int)(e % hidden);
  const float4v g = *reinterpret_cast<const float4v*>(table + (row / rows_per_frame) * ldt + off + c);
  const bf16x4 av = *reinterpret_cast<const bf16x4*>(a + e);
  float4v xv = *reinterpret_cast<const float4v*>(x + e);
#pragma unroll
  for (int j = 0; j < 4; ++j) xv[j] += g[j] * bf2f(av[j]);
  *reinterpret_cast<float4v*>(y + e) = xv;
}

// Row chunks per frame in the per-(frame, channel) reductions.  One thread per channel with 4-byte loads measured FASTER than 16-byte
// loads per thread (gate_bwd 42 vs 60 us: fewer, fatter threads leave too few loads in flight) and than 16 one-wave chunks (4x the atomics).
constexpr int TR_CHUNKS = 4;
// da = dY * gate (bf16) ; dgate[frame][c] += sum_rows dY * a ; dbias[c] += sum_rows da
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dy, const bf16* __restrict__ a, const float* __restrict__ table,
                                                       long ldt, long off, bf16* __restrict__ da, float* __restrict__ dmod,
                                                       float* __restrict__ dbias, int hidden, int rows_per_frame) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= hidden) return;
  const long frame = blockIdx.x;
  const int per = rows_per_frame / TR_CHUNKS, r0 = blockIdx.z * per;
  const float g = table[frame * ldt + off + c];
  float sg = 0.f, sb = 0.f;
  for (int r = r0; r < r0 + per; ++r) {
    const long e = (frame * rows_per_frame + r) * hidden + c;
    const float d = dy[e];
    const bf16 o = f2bf(d * g);
    da[e] = o;
    sg += d * bf2f(a[e]);
    sb += bf2f(o);
  }
  atomicAdd(dmod + frame * ldt + off + c, sg);
  atomicAdd(dbias + c, sb);
}

// LayerNorm + modulation backward, row part: dx = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)), dxh = dm (1 + scale); stats = (mean, rstd)
template <int VEC, int CNT>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const float* __restrict__ dm, const float* __restrict__ x,
                                                          const float* __restrict__ table, long ldt, long off, float* __restrict__ dx,
                                                          float* __restrict__ stats, int rows_per_frame, int rows, float eps) {
  typedef typename VecT<VEC>::type V;
  constexpr int hidden = 64 * VEC * CNT;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * hidden;
  const float* dr = dm + (long)row * hidden;
  const float* sc = table + (long)(row / rows_per_frame) * ldt + off + hidden;
  V v[CNT], g[CNT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    v[i] = *reinterpret_cast<const V*>(xr + (i * 64 + lane) * VEC);
    s += vsum<VEC>(v[i]);
  }
  const float mean = wave_sum(s) / (float)hidden;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    v[i] -= mean;
    q += vdot<VEC>(v[i], v[i]);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)hidden + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    v[i] = v[i] * rstd;  // xhat
    const V d = *reinterpret_cast<const V*>(dr + (i * 64 + lane) * VEC);
    const V sv = *reinterpret_cast<const V*>(sc + (i * 64 + lane) * VEC);
    g[i] = d * (1.0f + sv);
    s1 += vsum<VEC>(g[i]);
    s2 += vdot<VEC>(g[i], v[i]);
  }
  s1 = wave_sum(s1) / (float)hidden;
  s2 = wave_sum(s2) / (float)hidden;
  float* orow = dx + (long)row * hidden;
#pragma unroll
  for (int i = 0; i < CNT; ++i) *reinterpret_cast<V*>(orow + (i * 64 + lane) * VEC) = (g[i] - s1 - v[i] * s2) * rstd;
  if (lane == 0) {
    stats[2 * (long)row] = mean;
    stats[2 * (long)row + 1] = rstd;
  }
}
// frame part: dshift[frame][c] += sum_rows dm ; dscale[frame][c] += sum_rows dm xhat
__global__ __launch_bounds__(256) void ln_bwd_frames_kernel(const float* __restrict__ dm, const float* __restrict__ x,
                                                            const float* __restrict__ stats, float* __restrict__ dmod, long ldt, long off,
                                                            int hidden, int rows_per_frame) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= hidden) return;
  const long frame = blockIdx.x;
  const int per = rows_per_frame / TR_CHUNKS, r0 = blockIdx.z * per;
  float ssh = 0.f, ssc = 0.f;
  for (int r = r0; r < r0 + per; ++r) {
    const long row = frame * rows_per_frame + r;
    const float d = dm[row * hidden + c];
    ssh += d;
    ssc += d * (x[row * hidden + c] - stats[2 * row]) * stats[2 * row + 1];
  }
  atomicAdd(dmod + frame * ldt + off + c, ssh);
  atomicAdd(dmod + frame * ldt + off + hidden + c, ssc);
}

// (dq, dk, dv) [B][heads][ntok][dstride] -> dqkv [rows][3*heads*d] bf16 in the Linear's column order (q | k | v, head-major);
// q and k are rotated back (the transpose of the forward's RoPE rotation).  One thread per 8 columns.
__global__ void qkv_grad_pack_kernel(const bf16* __restrict__ dq, const bf16* __restrict__ dk, const bf16* __restrict__ dv,
                                     const float* __restrict__ rope_cs, bf16* __restrict__ out, long rows, int ntok, int heads, int d,
                                     int dstride) {
  const int per_row = 3 * heads * d / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * per_row) return;
  const long row = i / per_row;
  const int col = (int)(i % per_row) * 8;
  const int cdim = heads * d, which = col / cdim, cc = col - which * cdim, head = cc / d, e0 = cc % d;
  const long b = row / ntok;
  const int tok = (int)(row % ntok);
  const bf16* src = (which == 0 ? dq : (which == 1 ? dk : dv)) + ((b * heads + head) * ntok + tok) * (long)dstride + e0;
  bf16x8 g = *reinterpret_cast<const bf16x8*>(src);
  if (which < 2 && rope_cs) {
    const float* cs = rope_cs + ((long)tok * (d / 2) + e0 / 2) * 2;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      const float g0 = bf2f(g[2 * pr]), g1 = bf2f(g[2 * pr + 1]);
      const float co = cs[2 * pr], si = cs[2 * pr + 1];
      g[2 * pr] = f2bf(g0 * co + g1 * si);
      g[2 * pr + 1] = f2bf(g1 * co - g0 * si);
    }
  }
  *reinterpret_cast<bf16x8*>(out + row * (long)(3 * cdim) + col) = g;
}

// ---- deterministic two-stage sums --------------------------------------------------------------------------------------------
// Kernels that used to end in float atomics (bias / norm-weight / embedding sums: run-to-run differences of up to 5e-3 on
// cancellation-heavy sums) store ONE partial row per workgroup with plain stores; det_sum adds the rows in a fixed order (16 row
// groups per column, then a fixed tree through LDS), so two runs of one step give bit-identical gradients.
// det_scratch: grow-only device buffers of the training ops (process lifetime).  A buffer that is outgrown is KEPT (never freed or
// synchronised on a launch path): kernels in flight that still read the old block stay valid.
static int det_scratch(int slot, size_t floats, float** out) {
  static float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  static size_t cap[4] = {0, 0, 0, 0};
  if (cap[slot] < floats) {
    const size_t want = floats + floats / 4 + 1024;
    void* p = nullptr;
    DFOT_CHECK_HIP(hipMalloc(&p, want * sizeof(float)));
    buf[slot] = reinterpret_cast<float*>(p);
    cap[slot] = want;
  }
  *out = buf[slot];
  return DFOT_OK;
}
// out[b * out_batch_stride + c] (+)= sum_{i < nparts} part[b * batch_stride + i * row_stride + c], c < len
__global__ __launch_bounds__(1024) void det_sum_kernel(const float* __restrict__ part, long row_stride, long batch_stride, int nparts, int len,
                                                       float* out, long out_batch_stride, int accumulate) {
  __shared__ float red[16][64];
  const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  const float* p = part + (long)blockIdx.y * batch_stride + c;
  float acc = 0.f;
  if (c < len)
    for (int i = q; i < nparts; i += 16) acc += p[(long)i * row_stride];
  red[q][l] = acc;
  __syncthreads();
  if (q == 0 && c < len) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = red[2 * j][l] + red[2 * j + 1][l];
    const float total = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    float* o = out + (long)blockIdx.y * out_batch_stride + c;
    *o = accumulate ? *o + total : total;
  }
}
static int det_sum(const float* part, long row_stride, int nparts, int len, float* out, bool accumulate, hipStream_t s, int batch = 1,
                   long batch_stride = 0, long out_batch_stride = 0) {
  hipLaunchKernelGGL(det_sum_kernel, dim3(cdiv(len, 64), batch), dim3(1024), 0, s, part, row_stride, batch_stride, nparts, len, out, out_batch_stride,
                     accumulate ? 1 : 0);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// out[c] += sum_rows src[row][c]   (bf16 source; 128 rows per workgroup)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16* __restrict__ src, float* __restrict__ part, long rows, int n, long ld) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const long r0 = (long)blockIdx.y * 128;
  float acc = 0.f;
  for (long r = r0; r < r0 + 128 && r < rows; ++r) acc += bf2f(src[r * ld + c]);
  part[(long)blockIdx.y * n + c] = acc;  // one partial row per row block: det_sum adds them in a fixed order
}
// the streaming form: a workgroup owns 128 rows x (LANES * 8) columns; LANES lanes cover one row with 16-byte loads, the 256 / LANES
// row groups take alternate rows (8 independent loads in flight per thread), the groups are summed through LDS and the workgroup adds
// its LANES * 8 column sums once.  n, ld multiples of 8, src 16-byte aligned (launcher)
template <int LANES>
__global__ __launch_bounds__(256) void colsum_bf16_kernel8(const bf16* __restrict__ src, float* __restrict__ part, long rows, int n, long ld,
                                                           int iters) {
  constexpr int GROUPS = 256 / LANES;
  __shared__ float sm[GROUPS][LANES * 8 + 4];
  const int lane = threadIdx.x % LANES, grp = threadIdx.x / LANES;
  const int c = (blockIdx.x * LANES + lane) * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c < n) {
    const bf16* p = src + c;
    for (int it = 0; it < iters; ++it) {
      const long r0 = ((long)blockIdx.y * iters + it) * 128;
      if (r0 + 128 <= rows) {
        bf16x8 v[128 / GROUPS];
#pragma unroll
        for (int i = 0; i < 128 / GROUPS; ++i) v[i] = *reinterpret_cast<const bf16x8*>(p + (r0 + grp + (long)i * GROUPS) * ld);
#pragma unroll
        for (int i = 0; i < 128 / GROUPS; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[i][j]);
      } else {
        for (long r = r0 + grp; r < rows; r += GROUPS) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + r * ld);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sm[grp][lane * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < LANES * 8) {
    const int cc = blockIdx.x * LANES * 8 + threadIdx.x;
    if (cc < n) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < GROUPS; ++g) t += sm[g][threadIdx.x];
      part[(long)blockIdx.y * n + cc] = t;
    }
  }
}
// out[c] += sum_rows src[row][c], deterministic (partial rows + det_sum)
static int launch_colsum_bf16(const bf16* src, float* out, long rows, int n, long ld, hipStream_t s) {
  float* part = nullptr;
  int rc = 0;
  if (n % 8 == 0 && ld % 8 == 0 && ((uintptr_t)src & 15) == 0) {
    // a workgroup takes several 128-row blocks on long inputs, so that at most ~2048 partial rows remain per column block
    const int xb = n <= 128 ? cdiv(n, 128) : cdiv(n, 256);
    const long yb = cdiv(rows, 128);
    int iters = 1;
    while (iters < 16 && yb / iters * xb > 2048) iters *= 2;
    const int ny = (int)cdiv(yb, iters);
    if ((rc = det_scratch(0, (size_t)ny * n, &part))) return rc;
    if (n <= 128)
      hipLaunchKernelGGL(colsum_bf16_kernel8<16>, dim3(xb, ny), dim3(256), 0, s, src, part, rows, n, ld, iters);
    else
      hipLaunchKernelGGL(colsum_bf16_kernel8<32>, dim3(xb, ny), dim3(256), 0, s, src, part, rows, n, ld, iters);
    DFOT_CHECK_HIP(hipGetLastError());
    return det_sum(part, n, ny, n, out, true, s);
  }
  const int ny = (int)cdiv(rows, 128);
  if ((rc = det_scratch(0, (size_t)ny * n, &part))) return rc;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cdiv(n, 256), ny), dim3(256), 0, s, src, part, rows, n, ld);
  DFOT_CHECK_HIP(hipGetLastError());
  return det_sum(part, n, ny, n, out, true, s);
}

// gradient of the unpatchified output [BT][C][H][W] gathered per token: dyp [rows][64] bf16 (columns >= oc stay zero) and its
// transpose dyt [.. >= oc rows][rows] (rows >= oc stay zero)
__global__ void final_gather_kernel(const float* __restrict__ dout, bf16* __restrict__ dyp, bf16* __restrict__ dyt, long rows, int c, int hh,
                                    int ww, int ps) {
  const int oc = ps * ps * c, gw = ww / ps, P = (hh / ps) * gw;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * oc) return;
  const long row = i / oc;
  const int o = (int)(i % oc);
  const long bt = row / P;
  const int g = (int)(row % P), gy = g / gw, gx = g % gw;
  const int ch = o % c, pq = o / c, py = pq / ps, px = pq % ps;
  const bf16 v = f2bf(dout[((bt * c + ch) * hh + gy * ps + py) * ww + gx * ps + px]);
  dyp[row * 64 + o] = v;
  dyt[(long)o * rows + row] = v;
}

// PatchEmbed weight gradient: dW[o][kk] += sum_rows dx0[row][o] patch[row][kk], db[o] += sum_rows dx0[row][o].  One workgroup owns
// 64 * chunks consecutive rows, staged 64 at a time through LDS; for kdim <= 16 (every model here: 3..4 channels x 2x2 patches) the
// products stay in registers over all chunks, so a workgroup issues hidden * (kdim + 1) atomics once (a 64-row workgroup per launch
// put 25 M atomics on the 1536 addresses of the 128-wide RE10K embedding: 4.9 ms).  hidden < 256 dividing 256: the 256 / hidden
// thread groups take alternate rows.
__global__ __launch_bounds__(256) void pe_wgrad_kernel(const float* __restrict__ dx0, const float* __restrict__ x, float* __restrict__ part,
                                                       int c, int hh, int ww, int ps, int hidden, long rows, int chunks) {
  extern __shared__ float patch[];  // [64][kdim]
  const int gh = hh / ps, gw = ww / ps, kdim = c * ps * ps;
  const int nsub = (hidden < 256 && 256 % hidden == 0) ? 256 / hidden : 1;
  const int sub = nsub > 1 ? threadIdx.x / hidden : 0;
  const int o = nsub > 1 ? threadIdx.x % hidden : blockIdx.x * 256 + threadIdx.x;
  const bool live = o < hidden;
  const bool keep = kdim <= 16;  // accumulate over the chunks in registers
  // the workgroup's partial row: dW [hidden][kdim] then db [hidden]; thread groups (nsub > 1) own one row each and are added by det_sum too
  float* prow = part + ((long)blockIdx.y * nsub + sub) * ((long)hidden * (kdim + 1));
  float acc[16], bsum = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int ch = 0; ch < chunks; ++ch) {
    const long row0 = ((long)blockIdx.y * chunks + ch) * 64;
    if (row0 >= rows) break;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * kdim; i += 256) {
      const long row = row0 + i / kdim;
      const int kk = i % kdim;
      float v = 0.f;
      if (row < rows) {
        const long bt = row / (gh * gw);
        const int g = (int)(row % (gh * gw)), gy = g / gw, gx = g % gw;
        const int ci = kk / (ps * ps), py = (kk / ps) % ps, px = kk % ps;
        v = x[((bt * c + ci) * hh + gy * ps + py) * ww + gx * ps + px];
      }
      patch[i] = v;
    }
    __syncthreads();
    if (!live) continue;
    for (int k0 = 0; k0 < kdim; k0 += 16) {
      if (!keep) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
      }
      if (row0 + 64 <= rows && (64 / nsub) % 8 == 0) {
        // whole chunk: eight gradient loads in flight per thread
        for (int r8 = 0; r8 < 64 / nsub; r8 += 8) {
          float g[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) g[u] = dx0[(row0 + sub + (r8 + u) * nsub) * hidden + o];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int r = sub + (r8 + u) * nsub;
            if (k0 == 0) bsum += g[u];
#pragma unroll
            for (int j = 0; j < 16; ++j)
              if (k0 + j < kdim) acc[j] += g[u] * patch[r * kdim + k0 + j];
          }
        }
      } else {
        for (int r = sub; r < 64 && row0 + r < rows; r += nsub) {
          const float g = dx0[(row0 + r) * hidden + o];
          if (k0 == 0) bsum += g;
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (k0 + j < kdim) acc[j] += g * patch[r * kdim + k0 + j];
        }
      }
      if (!keep) {  // (chunks == 1 for kdim > 16: see the launcher -- every element of the row is stored exactly once)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (k0 + j < kdim) prow[(long)o * kdim + k0 + j] = acc[j];
      }
    }
  }
  if (!live) return;
  if (keep) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < kdim) prow[(long)o * kdim + j] = acc[j];
  }
  prow[(long)hidden * kdim + o] = bsum;
}
// data gradient of the patch embedding (a Conv2d with stride = kernel = patch): dx[bt][ci][gy*ps+py][gx*ps+px] = sum_o dy[row][o] w[o][kk]
// with kk = (ci, py, px); one thread per (token row, kk), the row's dy broadcast over its kk threads, the weights coalesced over kk
__global__ __launch_bounds__(256) void pe_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int c, int hh,
                                                       int ww, int ps, int hidden, long rows) {
  const int kdim = c * ps * ps, gh = hh / ps, gw = ww / ps;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * kdim) return;
  const long row = i / kdim;
  const int kk = (int)(i % kdim);
  const float* d = dy + row * hidden;
  float acc = 0.f;
  for (int o = 0; o < hidden; ++o) acc += d[o] * w[(long)o * kdim + kk];
  const long bt = row / (gh * gw);
  const int g = (int)(row % (gh * gw)), gy = g / gw, gx = g % gw;
  const int ci = kk / (ps * ps), py = (kk / ps) % ps, px = kk % ps;
  dx[((bt * c + ci) * hh + gy * ps + py) * ww + gx * ps + px] = acc;
}
// dW [hidden][kdim] += , db [hidden] += , deterministic (one partial row per workgroup and thread group + det_sum)
static int launch_pe_wgrad(const float* dx0, const float* x, float* dW, float* db, int c, int hh, int ww, int ps, int hidden, long rows, hipStream_t s) {
  const int kdim = c * ps * ps;
  // keep >= 512 workgroups; rows of a workgroup are accumulated in registers only for kdim <= 16, else one 64-row chunk per workgroup
  const int chunks = kdim > 16 ? 1 : (rows >= 64L * 16 * 512 ? 16 : (rows >= 64L * 4 * 512 ? 4 : 1));
  const int nsub = (hidden < 256 && 256 % hidden == 0) ? 256 / hidden : 1;
  const int ny = (int)cdiv(rows, 64L * chunks);
  const long rowlen = (long)hidden * (kdim + 1);
  float* part = nullptr;
  int rc = det_scratch(1, (size_t)ny * nsub * rowlen, &part);
  if (rc) return rc;
  if (nsub > 1 && hidden * nsub < 256) DFOT_CHECK_HIP(hipMemsetAsync(part, 0, (size_t)ny * nsub * rowlen * sizeof(float), s));
  hipLaunchKernelGGL(pe_wgrad_kernel, dim3(cdiv(hidden, 256), ny), dim3(256), 64 * kdim * sizeof(float), s, dx0, x, part, c, hh, ww, ps, hidden, rows, chunks);
  DFOT_CHECK_HIP(hipGetLastError());
  if ((rc = det_sum(part, rowlen, ny * nsub, hidden * kdim, dW, true, s))) return rc;
  return det_sum(part + (long)hidden * kdim, rowlen, ny * nsub, hidden, db, true, s);
}

// ---- MatrixDiTBlock (factorized matrix attention, variant 1) ------------------------------------------------------------
// backward of matrix_attn_kernel: z [B*L*E][3h] (q|k|v), d_o [B*L*E][h] -> dz [B*L*E][3h].  The hn*hd entries of one (video, c, r)
// head are split over MA_CHUNKS workgroups:
// pass A (matrix_attn_bwd_scores): partial S = <q_l, k_l'> and dP = <do_l, v_l'> of the chunk, added into sc[head][2][L*L]
// pass B (matrix_attn_bwd_apply): softmax and dS = P (dP - sum P dP) scale from sc (L x L, recomputed per workgroup), then
//   dq_l = sum_l' dS[l][l'] k_l', dk_l' = sum_l dS[l][l'] q_l, dv_l' = sum_l P[l][l'] do_l for the chunk's entries (L2 hits)
constexpr int MA_CHUNKS = 8;
__global__ __launch_bounds__(256) void matrix_attn_bwd_scores_kernel(const bf16* __restrict__ z, const bf16* __restrict__ d_o, float* __restrict__ sc,
                                                                     int L, int E, int h, int cc, int rr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = blockIdx.x, chunk = blockIdx.y;
  const int b = head / (cc * rr), c = (head / rr) % cc, r = head % rr;
  const int hn = E / cc, hd = h / rr, ne = hn * hd / 4;
  const int per = (ne + MA_CHUNKS - 1) / MA_CHUNKS, e0 = chunk * per, e1 = e0 + per < ne ? e0 + per : ne;
  const long ldz = 3L * h;
  auto zoff = [&](int l, int n) { return (((long)b * L + l) * E + c * hn + n) * ldz + r * hd; };
  auto ooff = [&](int l, int n) { return (((long)b * L + l) * E + c * hn + n) * (long)h + r * hd; };
  for (int pi = wave; pi < L * L; pi += 4) {
    const int l = pi / L, l2 = pi % L;
    float as = 0.f, ap = 0.f;
    for (int e = e0 + lane; e < e1; e += 64) {
      const int n = (e * 4) / hd, d = (e * 4) % hd;
      const bf16x4 q4 = *reinterpret_cast<const bf16x4*>(z + zoff(l, n) + d);
      const bf16x4 k4 = *reinterpret_cast<const bf16x4*>(z + zoff(l2, n) + h + d);
      const bf16x4 v4 = *reinterpret_cast<const bf16x4*>(z + zoff(l2, n) + 2 * h + d);
      const bf16x4 g4 = *reinterpret_cast<const bf16x4*>(d_o + ooff(l, n) + d);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        as += bf2f(q4[j]) * bf2f(k4[j]);
        ap += bf2f(g4[j]) * bf2f(v4[j]);
      }
    }
    as = wave_sum(as);
    ap = wave_sum(ap);
    if (lane == 0) {
      atomicAdd(sc + ((long)head * 2) * L * L + pi, as);
      atomicAdd(sc + ((long)head * 2 + 1) * L * L + pi, ap);
    }
  }
}
__global__ __launch_bounds__(256) void matrix_attn_bwd_apply_kernel(const bf16* __restrict__ z, const bf16* __restrict__ d_o, const float* __restrict__ sc,
                                                                    bf16* __restrict__ dz, int L, int E, int h, int cc, int rr, float scale) {
  __shared__ float sS[32 * 32], sP[32 * 32];
  const int head = blockIdx.x, chunk = blockIdx.y;
  const int b = head / (cc * rr), c = (head / rr) % cc, r = head % rr;
  const int hn = E / cc, hd = h / rr, ne = hn * hd / 4;
  const int per = (ne + MA_CHUNKS - 1) / MA_CHUNKS, e0 = chunk * per, e1 = e0 + per < ne ? e0 + per : ne;
  const long ldz = 3L * h;
  auto zoff = [&](int l, int n) { return (((long)b * L + l) * E + c * hn + n) * ldz + r * hd; };
  auto ooff = [&](int l, int n) { return (((long)b * L + l) * E + c * hn + n) * (long)h + r * hd; };
  for (int i = threadIdx.x; i < L * L; i += 256) {
    sS[i] = sc[((long)head * 2) * L * L + i] * scale;
    sP[i] = sc[((long)head * 2 + 1) * L * L + i];
  }
  __syncthreads();
  if (threadIdx.x < L) {
    float* srow = sS + threadIdx.x * L;
    float* prow = sP + threadIdx.x * L;
    float mx = srow[0];
    for (int j = 1; j < L; ++j) mx = fmaxf(mx, srow[j]);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) {
      srow[j] = __expf(srow[j] - mx);
      sum += srow[j];
    }
    const float inv = 1.0f / sum;
    float dot = 0.f;
    for (int j = 0; j < L; ++j) {
      srow[j] *= inv;            // P
      dot += srow[j] * prow[j];  // sum_j P dP
    }
    for (int j = 0; j < L; ++j) {
      const float pj = srow[j];
      srow[j] = pj * (prow[j] - dot) * scale;  // sS <- dS (scaled)
      prow[j] = pj;                            // sP <- P
    }
  }
  __syncthreads();
  for (int e = e0 + threadIdx.x; e < e1; e += 256) {
    const int n = (e * 4) / hd, d = (e * 4) % hd;
    for (int l = 0; l < L; ++l) {  // dq_l
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      for (int l2 = 0; l2 < L; ++l2) {
        const bf16x4 k4 = *reinterpret_cast<const bf16x4*>(z + zoff(l2, n) + h + d);
        const float w = sS[l * L + l2];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += w * bf2f(k4[j]);
      }
      bf16x4 o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = f2bf(a[j]);
      *reinterpret_cast<bf16x4*>(dz + zoff(l, n) + d) = o4;
    }
    for (int l2 = 0; l2 < L; ++l2) {  // dk_l2, dv_l2
      float ak[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
      for (int l = 0; l < L; ++l) {
        const bf16x4 q4 = *reinterpret_cast<const bf16x4*>(z + zoff(l, n) + d);
        const bf16x4 g4 = *reinterpret_cast<const bf16x4*>(d_o + ooff(l, n) + d);
        const float ws = sS[l * L + l2], wp = sP[l * L + l2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ak[j] += ws * bf2f(q4[j]);
          av[j] += wp * bf2f(g4[j]);
        }
      }
      bf16x4 k4, v4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        k4[j] = f2bf(ak[j]);
        v4[j] = f2bf(av[j]);
      }
      *reinterpret_cast<bf16x4*>(dz + zoff(l2, n) + h + d) = k4;
      *reinterpret_cast<bf16x4*>(dz + zoff(l2, n) + 2 * h + d) = v4;
    }
  }
}

// out[i] = sum_f src[f][i]  (two-dimensional biases: the sum runs over the frames); n % 4 == 0
__global__ void frames_sum_bf16_kernel(const bf16* __restrict__ src, float* __restrict__ out, int frames, long n) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int f = 0; f < frames; ++f) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(src + (long)f * n + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] += bf2f(v[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) out[i + j] = a[j];
}
// src [frames][R][C] -> dst [R][frames][C]  (8 elements per thread, C % 8 == 0)
__global__ void permute_frames_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int frames, int R, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = C / 8;
  if (i >= (long)frames * R * c8) return;
  const int cc = (int)(i % c8);
  const int rrow = (int)((i / c8) % R);
  const long f = i / ((long)c8 * R);
  *reinterpret_cast<bf16x8*>(dst + ((long)rrow * frames + f) * C + cc * 8) = *reinterpret_cast<const bf16x8*>(src + i * 8);
}
// out = sum of S partial outputs (split-K GEMM slices, plain stores)
__global__ void slices_sum_kernel(const float* __restrict__ ws, float* __restrict__ out, long n4, int slices, long stride) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4v a = *reinterpret_cast<const float4v*>(ws + i * 4);
  for (int s = 1; s < slices; ++s) a += *reinterpret_cast<const float4v*>(ws + s * stride + i * 4);
  *reinterpret_cast<float4v*>(out + i * 4) = a;
}
// y (fp32) += x (bf16)
__global__ void add_bf16_kernel(float* __restrict__ y, const bf16* __restrict__ x, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(x + i * 4);
  float4v o = *reinterpret_cast<float4v*>(y + i * 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] += bf2f(v[j]);
  *reinterpret_cast<float4v*>(y + i * 4) = o;
}
// DifferenceDiT3D conditioning: c[f] += diff_table[kind(f)] (kind 1 = difference token = even position), semb = bf16(SiLU(c))
__global__ void add_diff_kernel(float* __restrict__ cemb, const float* __restrict__ diff_table, bf16* __restrict__ semb, int frames, int tokens,
                                int hidden) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)frames * hidden) return;
  const int f = (int)(i / hidden), c = (int)(i % hidden);
  const int kind = ((f % tokens) % 2 == 0) ? 1 : 0;
  const float v = cemb[i] + diff_table[(long)kind * hidden + c];
  cemb[i] = v;
  semb[i] = f2bf(silu_f(v));
}
__global__ void diff_grad_kernel(const float* __restrict__ dc, float* __restrict__ dtable, int frames, int tokens, int hidden) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * hidden) return;
  const int kind = i / hidden, c = i % hidden;
  float acc = 0.f;
  for (int f = 0; f < frames; ++f)
    if ((((f % tokens) % 2 == 0) ? 1 : 0) == kind) acc += dc[(long)f * hidden + c];
  dtable[i] = acc;
}

int launch_ln_bwd_rows(const float* dm, const float* x, const float* table, long ldt, long off, float* dx, float* stats, int hidden,
                       int rows_per_frame, int rows, float eps, hipStream_t s) {
#define CALL(V, C) \
  hipLaunchKernelGGL((ln_bwd_rows_kernel<V, C>), dim3(cdiv(rows, 4)), dim3(256), 0, s, dm, x, table, ldt, off, dx, stats, rows_per_frame, rows, eps)
  DIT_LN_DISPATCH(CALL)
#undef CALL
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

}  // namespace
}  // namespace dfot

struct dfot_dit_train_s {
  dfot_dit_config cfg{};
  int gh = 0, gw = 0, P = 0, d = 0, dstride = 0, kpatch = 0, oc = 0;
  long ldt = 0, total = 0;
  std::vector<dfot::DitParam> params;   // name / shape (load unused)
  std::vector<long> offsets;
  float *params_f32 = nullptr, *grads = nullptr;  // attached flat buffers (owned by the caller)
  long o_t_w1 = 0, o_t_b1 = 0, o_t_w2 = 0, o_t_b2 = 0, o_pe_w = 0, o_pe_b = 0, o_diff = -1, o_fmod_w = 0, o_fmod_b = 0, o_fin_w = 0, o_fin_b = 0;
  long mod_final = 0;
  std::vector<dfot::TrainBlock> blocks;  // execution order (variant 1: spatial 0, temporal 0, spatial 1, ...)
  std::vector<void*> owned, ws_owned;
  size_t ws_bytes = 0;
  // compute copies
  dfot::bf16 *w_mod = nullptr, *w_modT = nullptr, *wfT = nullptr;
  float *b_mod = nullptr, *freqs = nullptr, *rope_cs = nullptr, *pos2d = nullptr;
  bool synced = false;
  // workspace
  int max_batch = 0, fp = 0, batch = 0, tokens = 0;
  int* idx = nullptr;
  const float* x_saved = nullptr;  // the forward's input (caller keeps it alive until backward)
  const float* d_embed = nullptr;  // after a backward: gradient w.r.t. the patch-embedding output [rows][hidden] (for input_grad)
  float *feat = nullptr, *h1 = nullptr, *a1 = nullptr, *cemb = nullptr, *mod_table = nullptr, *X = nullptr, *x_fin = nullptr;
  dfot::bf16* semb = nullptr;
  float *dX = nullptr, *dX2 = nullptr, *stats = nullptr, *delta = nullptr, *dmod = nullptr, *dsemb = nullptr, *dwf = nullptr;
  float *dc = nullptr, *da1 = nullptr, *dh1 = nullptr, *dbmod = nullptr, *scratch_f = nullptr;
  dfot::bf16 *da = nullptr, *dO = nullptr, *dq = nullptr, *dk = nullptr, *dv = nullptr, *dqkv = nullptr, *dqkvT = nullptr, *T1 = nullptr,
             *T2 = nullptr, *dmod_bf = nullptr, *dmodT = nullptr, *sembT = nullptr, *dyp = nullptr, *dyt = nullptr, *mfin = nullptr, *dh = nullptr;
  // matrix-block workspace
  float* ma_sc = nullptr;  // matrix attention backward: per-head (S, dP) [L*L] partial sums
  float* wg_ws = nullptr;  // partial outputs of split-K weight-gradient GEMMs
  size_t wg_ws_floats = 0;
  dfot::bf16 *mt = nullptr, *do2 = nullptr, *dz = nullptr, *dw1 = nullptr, *perm_a = nullptr, *perm_b = nullptr;
};

namespace dfot {
namespace {

template <typename T>
int tr_alloc(dfot_dit_train_s* h, T** out, size_t count, bool workspace = false) {
  void* p = nullptr;
  DFOT_CHECK_HIP(hipMalloc(&p, count * sizeof(T)));
  DFOT_CHECK_HIP(hipMemset(p, 0, count * sizeof(T)));
  (workspace ? h->ws_owned : h->owned).push_back(p);
  if (workspace) h->ws_bytes += count * sizeof(T);
  *out = (T*)p;
  return DFOT_OK;
}

long tr_add(dfot_dit_train_s* h, const std::string& name, std::vector<int64_t> shape) {
  DitParam p;
  p.name = name;
  p.shape = shape;
  long n = 1;
  for (int64_t v : shape) n *= v;
  const long off = h->total;
  h->params.push_back(p);
  h->offsets.push_back(off);
  h->total += (n + 3) / 4 * 4;  // every tensor starts 16-byte aligned
  return off;
}

// [batches][R][C] -> [batches][C][R]
int tr_transpose(const bf16* src, bf16* dst, int R, int C, hipStream_t s, int batches = 1) {
  DFOT_REQUIRE(R % 64 == 0 && C % 64 == 0, DFOT_ERR_SHAPE, "transpose: %d x %d must be multiples of 64", R, C);
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(C / 64, R / 64, batches), dim3(256), 0, s, src, dst, R, C);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// out[M][N] fp32 = A[M][K] W[N][K]^T (+ resid)
int tr_gemm_f32(const bf16* A, long lda, const bf16* W, int M, int N, int K, float* out, long ldo, const float* resid, hipStream_t s,
                int variant = GEMM_AUTO, int ksplit = 1) {
  while (ksplit > 1 && K / 64 < 4 * ksplit) ksplit /= 2;
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = W; g.M = M; g.N = N; g.K = K; g.out_f32 = out; g.ldo = ldo; g.resid = resid; g.ksplit = ksplit;
  return launch_gemm(A_DENSE, E_F32, variant, g, s);
}
// Weight gradient out[M][N] = A[M][K] W[N][K]^T with M, N = feature counts (multiples of 128 only) and K = tokens (long).
// (Splitting K over workgroups with atomic accumulation was measured and dropped: 42.3 ms/step at 512 workgroups, 46.0 at 768,
// 48.8 at 1024 vs 40.9 unsplit on DiT/XL -- the 8 M fp32 atomics per GEMM cost more than the idle CUs they recruit.)
// What does pay for the long-K shapes whose 256x192 tiling leaves CUs idle (MLP weights: 4608 x 1152 -> 108 tiles): K split in
// two with each slice storing its partial tile to a workspace (plain stores) and one pass summing the slices (`ws`, `ws_floats`).
int tr_wgrad(const bf16* A, const bf16* W, int M, int N, int K, float* out, hipStream_t s, float* ws = nullptr, size_t ws_floats = 0) {
  if (ws) {
    const long t192 = (long)(M / 256) * (N / 192);
    if (M % 256 == 0 && N % 192 == 0 && t192 >= 64 && t192 < 160 && K >= 4096) {
      const int split = t192 <= 85 ? 3 : 2;
      if ((size_t)split * M * N <= ws_floats) {
        GemmArgs g;
        g.A = A; g.lda = K; g.W = W; g.M = M; g.N = N; g.K = K; g.out_f32 = ws; g.ldo = N; g.ksplit = split; g.slice_stride = (long)M * N;
        int rc = launch_gemm(A_DENSE, E_F32, GEMM_DMA_256x192, g, s);
        if (rc) return rc;
        hipLaunchKernelGGL(slices_sum_kernel, dim3(cdiv((long)M * N / 4, 256)), dim3(256), 0, s, ws, out, (long)M * N / 4, split, (long)M * N);
        DFOT_CHECK_HIP(hipGetLastError());
        return DFOT_OK;
      }
    }
    const long t128x192 = (long)(M / 128) * (N / 192);
    if (M % 256 != 0 && N % 192 == 0 && t128x192 >= 160 && t128x192 <= 256 && K >= 4096)  // one round of 128x192 tiles
      return tr_gemm_f32(A, K, W, M, N, K, out, N, nullptr, s, GEMM_DMA_128x192, 1);
    // few 128x128 tiles and a long K (out-projection weights: 81 tiles; matrix factors U, U': 2 tiles over K = frames x hidden)
    const long t128 = (long)(M / 128) * ((N + 127) / 128);
    if (t128 <= 128 && K >= 2048) {
      const bool use192 = N % 192 == 0;  // 128x192 tiles where N allows (PMC: 0.21 vs 0.15 MFMA utilisation)
      const long tiles = use192 ? (long)(M / 128) * (N / 192) : t128;
      int split = (int)(256 / tiles);
      split = split > 64 ? 64 : split;
      while (split > 1 && K / 64 < 4 * split) --split;
      if (split > 1 && (size_t)split * M * N <= ws_floats) {
        GemmArgs g;
        g.A = A; g.lda = K; g.W = W; g.M = M; g.N = N; g.K = K; g.out_f32 = ws; g.ldo = N; g.ksplit = split; g.slice_stride = (long)M * N;
        int rc = launch_gemm(A_DENSE, E_F32, use192 ? GEMM_DMA_128x192 : GEMM_DMA_128, g, s);
        if (rc) return rc;
        hipLaunchKernelGGL(slices_sum_kernel, dim3(cdiv((long)M * N / 4, 256)), dim3(256), 0, s, ws, out, (long)M * N / 4, split, (long)M * N);
        DFOT_CHECK_HIP(hipGetLastError());
        return DFOT_OK;
      }
    }
  }
  return tr_gemm_f32(A, K, W, M, N, K, out, N, nullptr, s, GEMM_AUTO, 1);
}
// dW[M][N] = dY^T X over `rows` tokens with both operands in their own layout (wgrad.hip): no transposed copies.  Returns
// DFOT_ERR_STATE (and does nothing) when the shape is not covered, so the caller falls back to transposes + tr_wgrad.
int tr_wgrad_nt(const bf16* dy, long ldy, const bf16* x, long ldx, int M, int N, long rows, float* out, hipStream_t s, float* ws, size_t ws_floats) {
  if (M % 8 != 0 || N % 8 != 0 || rows % 64 != 0) return DFOT_ERR_STATE;
  const WgradPlan plan = wgrad_plan(M, N, rows, ws ? (long)(ws_floats / ((size_t)M * N)) : 1);
  if (plan.slices == 1) return launch_wgrad_nt_plan(dy, ldy, x, ldx, out, M, N, rows, plan, s);
  int rc = launch_wgrad_nt_plan(dy, ldy, x, ldx, ws, M, N, rows, plan, s);
  if (rc) return rc;
  hipLaunchKernelGGL(slices_sum_kernel, dim3(cdiv((long)M * N / 4, 256)), dim3(256), 0, s, ws, out, (long)M * N / 4, plan.slices, (long)M * N);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}
// large zero fills: hipMemsetAsync of the 1 GB gradient buffer ran as ~170 fill launches of 6 MB at 0.8 TB/s (1.4 ms per DiT/XL step);
// one streaming kernel with 16-byte stores does it at HBM rate.  bytes and ptr multiples of 16 (hipMalloc'd buffers, sizes padded by 4 floats)
__global__ __launch_bounds__(256) void zero_fill_kernel(float4v* __restrict__ p, long n16) {
  const float4v z = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p[i] = z;
}
int zero_fill(void* ptr, size_t bytes, hipStream_t s) {
  if (bytes < ((size_t)1 << 20) || (bytes & 15) || ((uintptr_t)ptr & 15)) {
    DFOT_CHECK_HIP(hipMemsetAsync(ptr, 0, bytes, s));
    return DFOT_OK;
  }
  const long n16 = (long)(bytes / 16);
  const int grid = (int)(n16 / 256 < 4096 ? (n16 + 255) / 256 : 4096);
  hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, s, (float4v*)ptr, n16);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

int tr_gemm_bf16(const bf16* A, long lda, const bf16* W, int M, int N, int K, const float* bias, bf16* out, long ldo, hipStream_t s,
                 int bias_rows = 0, int tr_rows = 0) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = W; g.M = M; g.N = N; g.K = K; g.bias = bias; g.bias_rows = bias_rows; g.out_bf16 = out; g.ldo = ldo; g.tr_rows = tr_rows;
  return launch_gemm(A_DENSE, E_BF16, GEMM_AUTO, g, s);
}

}  // namespace
}  // namespace dfot

extern "C" {
using namespace dfot;

int dfot_dit_train_destroy(dfot_dit_train_t h) {
  if (!h) return DFOT_OK;
  for (void* p : h->owned) (void)hipFree(p);
  for (void* p : h->ws_owned) (void)hipFree(p);
  delete h;
  return DFOT_OK;
}

int dfot_dit_train_create(const dfot_dit_config* cfg, dfot_dit_train_t* out) {
  DFOT_REQUIRE(cfg && out, DFOT_ERR_ARG, "train_create: null argument");
  const dfot_dit_config& c = *cfg;
  DFOT_REQUIRE(c.variant == 0 || c.variant == 1, DFOT_ERR_ARG, "train_create: unknown variant %d", c.variant);
  const bool facmat = c.variant == 1;
  DFOT_REQUIRE(c.mlp_hidden >= 0 && c.mlp_hidden % 128 == 0 && c.temporal_mlp_hidden >= 0 && c.temporal_mlp_hidden % 128 == 0, DFOT_ERR_ARG,
               "train_create: MLP widths %d / %d must be multiples of 128", c.mlp_hidden, c.temporal_mlp_hidden);
  DFOT_REQUIRE(c.hidden_size % 128 == 0 && c.num_heads > 0 && c.hidden_size % c.num_heads == 0, DFOT_ERR_ARG,
               "train_create: hidden_size %d must be a multiple of 128 and of num_heads", c.hidden_size);
  DFOT_REQUIRE((c.hidden_size / c.num_heads) % 8 == 0 && c.hidden_size / c.num_heads <= 128, DFOT_ERR_ARG, "train_create: head dim must be a multiple of 8, <= 128");
  DFOT_REQUIRE(c.patch_size > 0 && c.height % c.patch_size == 0 && c.width % c.patch_size == 0, DFOT_ERR_ARG, "train_create: patch size");
  DFOT_REQUIRE(c.in_channels * c.patch_size * c.patch_size <= 64, DFOT_ERR_ARG, "train_create: patch_size^2 * channels must be <= 64");
  DFOT_REQUIRE(c.noise_dim > 0 && c.noise_dim % 2 == 0 && c.depth > 0 && c.timesteps > 0 && c.max_tokens > 0, DFOT_ERR_ARG, "train_create: bad config");
  if (facmat) {
    DFOT_REQUIRE(c.embed_col_dim > 0 && c.embed_col_dim % 64 == 0 && c.embed_col_dim <= 128, DFOT_ERR_ARG, "train_create: embed_col_dim %d must be 64 or 128", c.embed_col_dim);
    DFOT_REQUIRE(c.num_col_heads > 0 && c.num_row_heads > 0 && c.embed_col_dim % c.num_col_heads == 0 && c.hidden_size % c.num_row_heads == 0 &&
                     (c.hidden_size / c.num_row_heads) % 4 == 0 && c.max_tokens <= 32,
                 DFOT_ERR_ARG, "train_create: matrix attention heads");
  }
  auto* h = new dfot_dit_train_s();
  h->cfg = c;
  const int hd = c.hidden_size;
  h->gh = c.height / c.patch_size;
  h->gw = c.width / c.patch_size;
  h->P = h->gh * h->gw;
  h->d = hd / c.num_heads;
  h->dstride = attention_dstride(h->d);
  h->kpatch = c.in_channels * c.patch_size * c.patch_size;
  h->oc = h->kpatch;
  const int mh = c.mlp_hidden, th = facmat ? c.temporal_mlp_hidden : 0, E = c.embed_col_dim, P = h->P;
  h->ldt = (long)c.depth * ((mh ? 6 : 3) + (facmat ? (th ? 6 : 3) : 0)) * hd + 2 * hd;
  if (h->P % TR_CHUNKS != 0 || (facmat && h->P % 128 != 0)) {
    set_error("train_create: %d patches per frame must be a multiple of %d", h->P, facmat ? 128 : TR_CHUNKS);
    delete h;
    return DFOT_ERR_ARG;
  }
  // registration order == the reference module's state_dict order (as dit_build): all spatial blocks, then all temporal blocks
  const std::string ne = "noise_level_pos_embedding.embedding";
  h->o_t_w1 = tr_add(h, ne + ".linear_1.weight", {hd, c.noise_dim});
  h->o_t_b1 = tr_add(h, ne + ".linear_1.bias", {hd});
  h->o_t_w2 = tr_add(h, ne + ".linear_2.weight", {hd, hd});
  h->o_t_b2 = tr_add(h, ne + ".linear_2.bias", {hd});
  h->o_pe_w = tr_add(h, "patch_embedder.proj.weight", {hd, c.in_channels, c.patch_size, c.patch_size});
  h->o_pe_b = tr_add(h, "patch_embedder.proj.bias", {hd});
  if (facmat) h->o_diff = tr_add(h, "diff_embedder.embedding_table.weight", {2, hd});
  std::vector<TrainBlock> spatial(c.depth), temporal(facmat ? c.depth : 0);
  long off = 0;
  auto add_mlp = [&](TrainBlock& b, const std::string& pre, int width) {
    b.mh = width;
    if (!width) return;
    b.mod2 = off;
    off += 3 * hd;
    b.o_mod2_w = tr_add(h, pre + ".norm2.modulation.1.weight", {3 * hd, hd});
    b.o_mod2_b = tr_add(h, pre + ".norm2.modulation.1.bias", {3 * hd});
    b.o_fc1_w = tr_add(h, pre + ".mlp.fc1.weight", {width, hd});
    b.o_fc1_b = tr_add(h, pre + ".mlp.fc1.bias", {width});
    b.o_fc2_w = tr_add(h, pre + ".mlp.fc2.weight", {hd, width});
    b.o_fc2_b = tr_add(h, pre + ".mlp.fc2.bias", {hd});
  };
  for (int i = 0; i < c.depth; ++i) {
    TrainBlock& b = spatial[i];
    const std::string pre = "dit_base.blocks." + std::to_string(i);
    b.mod = off;
    off += 3 * hd;
    b.o_mod_w = tr_add(h, pre + ".norm1.modulation.1.weight", {3 * hd, hd});
    b.o_mod_b = tr_add(h, pre + ".norm1.modulation.1.bias", {3 * hd});
    b.o_qkv_w = tr_add(h, pre + ".attn.qkv.weight", {3 * hd, hd});
    b.o_qkv_b = tr_add(h, pre + ".attn.qkv.bias", {3 * hd});
    b.o_proj_w = tr_add(h, pre + ".attn.proj.weight", {hd, hd});
    b.o_proj_b = tr_add(h, pre + ".attn.proj.bias", {hd});
    add_mlp(b, pre, mh);
  }
  for (int i = 0; i < (int)temporal.size(); ++i) {
    TrainBlock& b = temporal[i];
    b.matrix = true;
    const std::string pre = "dit_base.temporal_blocks." + std::to_string(i);
    b.mod = off;
    off += 3 * hd;
    b.o_mod_w = tr_add(h, pre + ".norm1.modulation.1.weight", {3 * hd, hd});
    b.o_mod_b = tr_add(h, pre + ".norm1.modulation.1.bias", {3 * hd});
    b.o_qkv_u = tr_add(h, pre + ".attn.qkv_u", {P, E});
    b.o_proj_u = tr_add(h, pre + ".attn.proj_u", {E, P});
    b.o_qkv_v = tr_add(h, pre + ".attn.qkv_v", {hd, 3 * hd});
    b.o_proj_v = tr_add(h, pre + ".attn.proj_v", {hd, hd});
    if (c.use_bias) {
      b.o_qkv_bias = tr_add(h, pre + ".attn.qkv_bias", {E, 3 * hd});
      b.o_proj_bias = tr_add(h, pre + ".attn.proj_bias", {P, hd});
    }
    add_mlp(b, pre, th);
  }
  for (int i = 0; i < c.depth; ++i) {
    h->blocks.push_back(spatial[i]);
    if (facmat) h->blocks.push_back(temporal[i]);
  }
  h->mod_final = off;
  h->o_fmod_w = tr_add(h, "dit_base.final_layer.norm_final.modulation.1.weight", {2 * hd, hd});
  h->o_fmod_b = tr_add(h, "dit_base.final_layer.norm_final.modulation.1.bias", {2 * hd});
  h->o_fin_w = tr_add(h, "dit_base.final_layer.linear.weight", {h->oc, hd});
  h->o_fin_b = tr_add(h, "dit_base.final_layer.linear.bias", {h->oc});
  int rc = 0;
  auto fail = [&](int code) { dfot_dit_train_destroy(h); return code; };
  if ((rc = tr_alloc(h, &h->w_mod, (size_t)h->ldt * hd)) || (rc = tr_alloc(h, &h->w_modT, (size_t)h->ldt * hd)) ||
      (rc = tr_alloc(h, &h->b_mod, (size_t)h->ldt)) || (rc = tr_alloc(h, &h->wfT, (size_t)hd * 64)) ||
      (rc = tr_alloc(h, &h->freqs, (size_t)c.noise_dim / 2)))
    return fail(rc);
  for (TrainBlock& b : h->blocks) {
    if (!b.matrix) {
      if ((rc = tr_alloc(h, &b.w_qkv, (size_t)3 * hd * hd)) || (rc = tr_alloc(h, &b.w_qkvT, (size_t)3 * hd * hd)) ||
          (rc = tr_alloc(h, &b.w_proj, (size_t)hd * hd)) || (rc = tr_alloc(h, &b.w_projT, (size_t)hd * hd)))
        return fail(rc);
    } else {
      if ((rc = tr_alloc(h, &b.u_s, (size_t)P * E)) || (rc = tr_alloc(h, &b.u_t, (size_t)P * E)) || (rc = tr_alloc(h, &b.pu_s, (size_t)P * E)) ||
          (rc = tr_alloc(h, &b.pu_t, (size_t)P * E)) || (rc = tr_alloc(h, &b.v_s, (size_t)3 * hd * hd)) || (rc = tr_alloc(h, &b.v_t, (size_t)3 * hd * hd)) ||
          (rc = tr_alloc(h, &b.pv_s, (size_t)hd * hd)) || (rc = tr_alloc(h, &b.pv_t, (size_t)hd * hd)))
        return fail(rc);
    }
    if (b.mh && ((rc = tr_alloc(h, &b.w_fc1, (size_t)b.mh * hd)) || (rc = tr_alloc(h, &b.w_fc1T, (size_t)b.mh * hd)) ||
                 (rc = tr_alloc(h, &b.w_fc2, (size_t)b.mh * hd)) || (rc = tr_alloc(h, &b.w_fc2T, (size_t)b.mh * hd))))
      return fail(rc);
  }
  {
    const int half = c.noise_dim / 2;
    std::vector<float> f(half);
    for (int i = 0; i < half; ++i) f[i] = (float)std::exp(-std::log(10000.0) * (double)i / (double)half);
    if (hipMemcpy(h->freqs, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(DFOT_ERR_HIP);
  }
  if (facmat) {  // sinusoidal_2d table, as dit_build
    const int half = hd / 2, quarter = half / 2;
    std::vector<float> pe((size_t)P * hd);
    for (int m = 0; m < P; ++m) {
      const int pos[2] = {m % h->gh, m / h->gh};
      for (int a = 0; a < 2; ++a)
        for (int i = 0; i < quarter; ++i) {
          const double ang = (double)pos[a] / std::pow(10000.0, (double)i / (double)quarter);
          pe[(size_t)m * hd + a * half + i] = (float)std::sin(ang);
          pe[(size_t)m * hd + a * half + quarter + i] = (float)std::cos(ang);
        }
    }
    if ((rc = tr_alloc(h, &h->pos2d, pe.size()))) return fail(rc);
    if (hipMemcpy(h->pos2d, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(DFOT_ERR_HIP);
  } else {  // RoPE-3D table, as dit_build
    const int half = h->d / 2, q = half / 3, rem = half % 3;
    int parts[3] = {q, q, q};
    if (rem == 1) parts[0] = q + 1;
    if (rem == 2) parts[1] = parts[2] = q + 1;
    const int n = c.max_tokens * h->P;
    std::vector<float> cs((size_t)n * half * 2);
    for (int tok = 0; tok < n; ++tok) {
      const int pos[3] = {tok / h->P, (tok / h->gw) % h->gh, tok % h->gw};
      int pair = 0;
      for (int ax = 0; ax < 3; ++ax) {
        const int dim = 2 * parts[ax];
        for (int j = 0; j < parts[ax]; ++j, ++pair) {
          const float inv = 1.0f / powf(c.rope_theta, (float)(2 * j) / (float)dim);
          const float ang = (float)pos[ax] * inv;
          cs[((size_t)tok * half + pair) * 2 + 0] = cosf(ang);
          cs[((size_t)tok * half + pair) * 2 + 1] = sinf(ang);
        }
      }
    }
    if ((rc = tr_alloc(h, &h->rope_cs, cs.size()))) return fail(rc);
    if (hipMemcpy(h->rope_cs, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(DFOT_ERR_HIP);
  }
  *out = h;
  return DFOT_OK;
}

int dfot_dit_train_num_params(dfot_dit_train_t h) { return h ? (int)h->params.size() : 0; }
const char* dfot_dit_train_param_name(dfot_dit_train_t h, int i) {
  return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : nullptr;
}
int dfot_dit_train_param_shape(dfot_dit_train_t h, int i, int64_t shape[4], int* ndim) {
  DFOT_REQUIRE(h && i >= 0 && i < (int)h->params.size() && shape && ndim, DFOT_ERR_ARG, "train_param_shape: bad argument");
  *ndim = (int)h->params[i].shape.size();
  for (int j = 0; j < *ndim; ++j) shape[j] = h->params[i].shape[j];
  return DFOT_OK;
}
int64_t dfot_dit_train_param_offset(dfot_dit_train_t h, int i) { return (h && i >= 0 && i < (int)h->offsets.size()) ? h->offsets[i] : -1; }
int64_t dfot_dit_train_total_numel(dfot_dit_train_t h) { return h ? h->total : 0; }
size_t dfot_dit_train_workspace_bytes(dfot_dit_train_t h) { return h ? h->ws_bytes : 0; }

int dfot_dit_train_attach(dfot_dit_train_t h, float* params, float* grads) {
  DFOT_REQUIRE(h && params && grads, DFOT_ERR_ARG, "train_attach: null argument");
  DFOT_REQUIRE(((uintptr_t)params & 15) == 0 && ((uintptr_t)grads & 15) == 0, DFOT_ERR_ARG, "train_attach: buffers must be 16-byte aligned");
  h->params_f32 = params;
  h->grads = grads;
  h->synced = false;
  return DFOT_OK;
}

// fp32 master weights -> bf16 compute copies ([out][in] and transposed), stacked modulation Linear; call after every optimizer step
int dfot_dit_train_sync_weights(dfot_dit_train_t h, void* stream) {
  DFOT_REQUIRE(h && h->params_f32, DFOT_ERR_STATE, "train_sync_weights: no parameter buffer attached");
  hipStream_t s = (hipStream_t)stream;
  const int hd = h->cfg.hidden_size, P = h->P, E = h->cfg.embed_col_dim;
  const float* p = h->params_f32;
  int rc = 0;
  auto mod = [&](long o_w, long o_b, long col, int n) -> int {
    int r = launch_f32_to_bf16(p + o_w, h->w_mod + col * hd, (long)n * hd, s);
    if (r) return r;
    DFOT_CHECK_HIP(hipMemcpyAsync(h->b_mod + col, p + o_b, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return DFOT_OK;
  };
  auto pair = [&](long o_w, bf16* as_stored, bf16* transposed, int R, int C) -> int {  // [R][C] fp32 -> bf16 and its transpose [C][R]
    int r = launch_f32_to_bf16(p + o_w, as_stored, (long)R * C, s);
    return r ? r : tr_transpose(as_stored, transposed, R, C, s);
  };
  for (TrainBlock& b : h->blocks) {
    if ((rc = mod(b.o_mod_w, b.o_mod_b, b.mod, 3 * hd))) return rc;
    if (!b.matrix) {
      if ((rc = pair(b.o_qkv_w, b.w_qkv, b.w_qkvT, 3 * hd, hd)) || (rc = pair(b.o_proj_w, b.w_proj, b.w_projT, hd, hd))) return rc;
    } else {
      if ((rc = pair(b.o_qkv_u, b.u_s, b.u_t, P, E)) || (rc = pair(b.o_proj_u, b.pu_s, b.pu_t, E, P)) ||
          (rc = pair(b.o_qkv_v, b.v_s, b.v_t, hd, 3 * hd)) || (rc = pair(b.o_proj_v, b.pv_s, b.pv_t, hd, hd)))
        return rc;
    }
    if (b.mh) {
      if ((rc = mod(b.o_mod2_w, b.o_mod2_b, b.mod2, 3 * hd))) return rc;
      if ((rc = pair(b.o_fc1_w, b.w_fc1, b.w_fc1T, b.mh, hd)) || (rc = pair(b.o_fc2_w, b.w_fc2, b.w_fc2T, hd, b.mh))) return rc;
    }
  }
  if ((rc = mod(h->o_fmod_w, h->o_fmod_b, h->mod_final, 2 * hd))) return rc;
  if ((rc = tr_transpose(h->w_mod, h->w_modT, (int)h->ldt, hd, s))) return rc;
  // wfT[c][o] = fin_w[o][c] (bf16, rows of 64, zero padded)
  DFOT_CHECK_HIP(hipMemsetAsync(h->wfT, 0, (size_t)hd * 64 * sizeof(bf16), s));
  for (int o = 0; o < h->oc; ++o)
    if ((rc = launch_pack_rows(p + h->o_fin_w + (long)o * hd, h->wfT, nullptr, hd, 1, 1, 64, o, s))) return rc;
  h->synced = true;
  return DFOT_OK;
}

int dfot_dit_train_reserve(dfot_dit_train_t h, int max_batch) {
  DFOT_REQUIRE(h && max_batch > 0, DFOT_ERR_ARG, "train_reserve: bad argument");
  if (max_batch <= h->max_batch) return DFOT_OK;
  DFOT_CHECK_HIP(hipDeviceSynchronize());
  for (void* p : h->ws_owned) (void)hipFree(p);
  h->ws_owned.clear();
  h->ws_bytes = 0;
  h->max_batch = 0;
  const dfot_dit_config& c = h->cfg;
  const bool facmat = c.variant == 1;
  const int hd = c.hidden_size, nd = c.noise_dim, E = c.embed_col_dim;
  const size_t rows = (size_t)max_batch * c.max_tokens * h->P;
  const int frames = max_batch * c.max_tokens;
  const int fp = (frames + 255) / 256 * 256;
  const size_t bhn = (size_t)max_batch * c.num_heads * c.max_tokens * h->P;
  const size_t qsz = bhn * h->dstride;
  const size_t fe = (size_t)frames * E;
  int widest = 3 * hd;
  for (const TrainBlock& b : h->blocks) widest = b.mh > widest ? b.mh : widest;
  int rc = 0;
#define WS(ptr, count) if ((rc = tr_alloc(h, &(ptr), (count), true))) return rc
  WS(h->idx, frames);
  WS(h->feat, (size_t)frames * nd); WS(h->h1, (size_t)frames * hd); WS(h->a1, (size_t)frames * hd); WS(h->cemb, (size_t)frames * hd);
  WS(h->semb, (size_t)fp * hd); WS(h->sembT, (size_t)fp * hd); WS(h->mod_table, (size_t)fp * h->ldt);
  WS(h->X, rows * hd); WS(h->x_fin, rows * hd);
  bool any_mlp = false;
  for (TrainBlock& b : h->blocks) {
    WS(b.x_in, rows * hd); WS(b.m, rows * hd); WS(b.a, rows * hd);
    if (!b.matrix) {
      WS(b.q, qsz); WS(b.k, qsz); WS(b.v, qsz); WS(b.o, rows * hd); WS(b.lse, bhn);
    } else {
      WS(b.w1, fe * hd); WS(b.z, fe * 3 * hd); WS(b.o2, fe * hd); WS(b.sfac, rows * hd);
    }
    if (b.mh) { WS(b.x_mid, rows * hd); WS(b.m2, rows * hd); WS(b.u, rows * b.mh); WS(b.hact, rows * b.mh); WS(b.y, rows * hd); any_mlp = true; }
  }
  if (any_mlp) { WS(h->dh, rows * widest); }
  h->wg_ws_floats = (size_t)3 * widest * hd;
  WS(h->wg_ws, h->wg_ws_floats);
  WS(h->dX, rows * hd); WS(h->dX2, rows * hd); WS(h->stats, rows * 2); WS(h->delta, bhn);
  WS(h->dmod, (size_t)fp * h->ldt); WS(h->dmod_bf, (size_t)fp * h->ldt); WS(h->dmodT, (size_t)fp * h->ldt); WS(h->dbmod, (size_t)h->ldt);
  WS(h->dsemb, (size_t)fp * hd); WS(h->dwf, (size_t)256 * hd > (size_t)128 * h->P ? (size_t)256 * hd : (size_t)128 * h->P);
  WS(h->dc, (size_t)frames * hd); WS(h->da1, (size_t)frames * hd); WS(h->dh1, (size_t)frames * hd); WS(h->scratch_f, (size_t)hd);
  WS(h->da, rows * hd); WS(h->dO, rows * hd); WS(h->dq, qsz); WS(h->dk, qsz); WS(h->dv, qsz);
  WS(h->dqkv, rows * 3 * hd); WS(h->dqkvT, rows * widest); WS(h->T1, rows * widest); WS(h->T2, rows * hd); WS(h->dyp, rows * 64); WS(h->dyt, (size_t)256 * rows);
  WS(h->mfin, rows * hd);
  if (facmat) {
    WS(h->ma_sc, (size_t)max_batch * c.num_col_heads * c.num_row_heads * 2 * c.max_tokens * c.max_tokens);
    WS(h->mt, rows * hd); WS(h->do2, fe * hd); WS(h->dz, fe * 3 * hd); WS(h->dw1, fe * hd);
    WS(h->perm_a, (size_t)(h->P > 128 ? h->P : 128) * frames * hd); WS(h->perm_b, (size_t)(h->P > 128 ? h->P : 128) * frames * hd);
  }
#undef WS
  h->max_batch = max_batch;
  h->fp = fp;
  return DFOT_OK;
}

// out[B,T,C,H,W] = model(x, levels) with every activation the backward needs kept in the workspace; x must stay alive until backward
int dfot_dit_train_forward(dfot_dit_train_t h, const float* x, const int32_t* noise_levels, float* out, int batch, int tokens, void* stream) {
  DFOT_REQUIRE(h && x && noise_levels && out, DFOT_ERR_ARG, "train_forward: null argument");
  DFOT_REQUIRE(h->synced, DFOT_ERR_STATE, "train_forward: call dfot_dit_train_sync_weights after attaching / updating the parameters");
  DFOT_REQUIRE(batch > 0 && batch <= h->max_batch, DFOT_ERR_STATE, "train_forward: batch %d exceeds the reserved %d", batch, h->max_batch);
  const dfot_dit_config& c = h->cfg;
  const bool facmat = c.variant == 1;
  DFOT_REQUIRE(tokens > 0 && tokens <= c.max_tokens, DFOT_ERR_SHAPE, "train_forward: %d tokens, max_tokens is %d", tokens, c.max_tokens);
  DFOT_REQUIRE(!facmat || tokens % 2 == 0, DFOT_ERR_SHAPE, "train_forward: %d tokens; the difference model takes (difference, frame) pairs", tokens);
  const int n = tokens * h->P, hd = c.hidden_size, P = h->P, frames = batch * tokens, nd = c.noise_dim, E = c.embed_col_dim;
  DFOT_REQUIRE(n % 128 == 0, DFOT_ERR_SHAPE, "train_forward: sequence length %d (tokens x patches) must be a multiple of 128", n);
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)batch * n;
  const float* p = h->params_f32;
  int rc = 0;
  h->batch = batch; h->tokens = tokens; h->x_saved = x;
  h->d_embed = nullptr;
  // ---- conditioning: c = Linear2(SiLU(Linear1(features(level)))) [+ diff embedding] per frame; table = Linear_mod(SiLU(c)) ----
  hipLaunchKernelGGL(iota_kernel, dim3(cdiv(frames, 256)), dim3(256), 0, s, h->idx, frames);
  hipLaunchKernelGGL(tr_features_kernel, dim3(cdiv((long)frames * nd, 256)), dim3(256), 0, s, h->freqs, noise_levels, h->feat, frames, nd, c.timesteps - 1);
  hipLaunchKernelGGL(rows_linear_kernel<0>, dim3(cdiv(hd, 4), frames), dim3(256), 0, s, h->feat, p + h->o_t_w1, p + h->o_t_b1, h->h1, (bf16*)nullptr, nd, hd);
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(cdiv((long)frames * hd, 256)), dim3(256), 0, s, h->h1, h->a1, (long)frames * hd);
  DFOT_CHECK_HIP(hipMemsetAsync(h->semb, 0, (size_t)h->fp * hd * sizeof(bf16), s));
  hipLaunchKernelGGL(rows_linear_kernel<0>, dim3(cdiv(hd, 4), frames), dim3(256), 0, s, h->a1, p + h->o_t_w2, p + h->o_t_b2, h->cemb, h->semb, hd, hd);
  if (facmat)
    hipLaunchKernelGGL(add_diff_kernel, dim3(cdiv((long)frames * hd, 256)), dim3(256), 0, s, h->cemb, p + h->o_diff, h->semb, frames, tokens, hd);
  DFOT_CHECK_HIP(hipGetLastError());
  {
    GemmArgs g;
    g.A = h->semb; g.lda = hd; g.W = h->w_mod; g.M = h->fp; g.N = (int)h->ldt; g.K = hd; g.bias = h->b_mod; g.out_f32 = h->mod_table; g.ldo = h->ldt;
    if ((rc = launch_gemm(A_DENSE, E_F32, GEMM_AUTO, g, s))) return rc;
  }
  // the residual stream lives in the blocks' own x_in buffers (each is what the backward of that block needs): no copies
  hipLaunchKernelGGL(patch_embed_kernel, dim3(cdiv(rows, PE_TOK)), dim3(256), PE_TOK * h->kpatch * sizeof(float), s, x, p + h->o_pe_w,
                     p + h->o_pe_b, (const float*)h->pos2d, h->blocks[0].x_in, c.in_channels, c.height, c.width, c.patch_size, hd, rows);
  DFOT_CHECK_HIP(hipGetLastError());
  const float qscale = 1.4426950408889634f / sqrtf((float)h->d);
  // attention sequences: the whole video with RoPE-3D (variant 0) or one frame without RoPE (variant 1 spatial blocks)
  const int seq = facmat ? P : n, nseq = facmat ? frames : batch;
  auto combine = [&](const bf16* a, long gate_off, float* dst) -> int {
    hipLaunchKernelGGL(gate_combine_kernel, dim3(cdiv(rows * hd / 4, 256)), dim3(256), 0, s, h->X, dst, a, h->mod_table, h->ldt, gate_off, hd, P,
                       rows * hd / 4);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  };
  for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
    TrainBlock& b = h->blocks[bi];
    float* next = bi + 1 < h->blocks.size() ? h->blocks[bi + 1].x_in : h->x_fin;
    float* after_attn = b.mh ? b.x_mid : next;
    if ((rc = launch_ln_mod(b.x_in, h->X, b.m, h->mod_table, h->idx, h->ldt, b.mod, hd, P, (int)rows, c.eps, frames - 1, s))) return rc;
    if (!b.matrix) {
      GemmArgs g;
      g.A = b.m; g.lda = hd; g.W = b.w_qkv; g.M = (int)rows; g.N = 3 * hd; g.K = hd; g.bias = p + b.o_qkv_b;
      g.q = b.q; g.k = b.k; g.v = b.v; g.rope_cs = facmat ? nullptr : h->rope_cs; g.heads = c.num_heads; g.d = h->d; g.dstride = h->dstride; g.ntok = seq;
      g.qscale = qscale;
      if ((rc = launch_gemm(A_DENSE, E_QKV_DIT, GEMM_AUTO, g, s))) return rc;
      if ((rc = launch_attention_padded(b.q, b.k, b.v, b.o, hd, nseq, c.num_heads, seq, h->d, s, b.lse))) return rc;
      if ((rc = tr_gemm_bf16(b.o, hd, b.w_proj, (int)rows, hd, hd, p + b.o_proj_b, b.a, hd, s))) return rc;
    } else {
      const bool bias = b.o_qkv_bias >= 0;
      if ((rc = tr_transpose(b.m, h->mt, P, hd, s, frames))) return rc;                                                    // m^T per frame [hd][P]
      if ((rc = tr_gemm_bf16(h->mt, P, b.u_t, frames * hd, E, P, nullptr, b.w1, E, s, 0, hd))) return rc;                   // w1[f][e][d] = sum_p U[p][e] m[f][p][d]
      if ((rc = tr_gemm_bf16(b.w1, hd, b.v_t, frames * E, 3 * hd, hd, bias ? p + b.o_qkv_bias : nullptr, b.z, 3 * hd, s, bias ? E : 0))) return rc;
      {
        const int hn = E / c.num_col_heads, hdr = hd / c.num_row_heads;
        if ((rc = launch_matrix_attn(b.z, b.o2, batch, tokens, E, hd, c.num_col_heads, c.num_row_heads, 1.0f / sqrtf((float)hn * (float)hdr), s))) return rc;
      }
      if ((rc = tr_transpose(b.o2, h->mt, E, hd, s, frames))) return rc;                                                   // o^T per frame [hd][E]
      if ((rc = tr_gemm_bf16(h->mt, E, b.pu_t, frames * hd, P, E, nullptr, b.sfac, P, s, 0, hd))) return rc;                // s[f][p][d] = sum_e U'[e][p] o[f][e][d]
      if ((rc = tr_gemm_bf16(b.sfac, hd, b.pv_t, (int)rows, hd, hd, bias ? p + b.o_proj_bias : nullptr, b.a, hd, s, bias ? P : 0))) return rc;
    }
    if ((rc = combine(b.a, b.mod + 2 * hd, after_attn))) return rc;
    if (b.mh) {
      if ((rc = launch_ln_mod(b.x_mid, h->X, b.m2, h->mod_table, h->idx, h->ldt, b.mod2, hd, P, (int)rows, c.eps, frames - 1, s))) return rc;
      {
        GemmArgs g;  // h = GELU(u), u = m2 W1^T + b1: both kept (u for GELU', h for the fc2 weight gradient)
        g.A = b.m2; g.lda = hd; g.W = b.w_fc1; g.M = (int)rows; g.N = b.mh; g.K = hd; g.bias = p + b.o_fc1_b; g.out_bf16 = b.hact; g.ldo = b.mh;
        g.act = 1; g.pre_act = b.u;
        if ((rc = launch_gemm(A_DENSE, E_BF16, GEMM_AUTO, g, s))) return rc;
      }
      if ((rc = tr_gemm_bf16(b.hact, b.mh, b.w_fc2, (int)rows, hd, b.mh, p + b.o_fc2_b, b.y, hd, s))) return rc;
      if ((rc = combine(b.y, b.mod2 + 2 * hd, next))) return rc;
    }
  }
  return launch_final_layer(h->x_fin, h->mod_table, h->idx, h->ldt, h->mod_final, p + h->o_fin_w, p + h->o_fin_b, out, hd, P, (int)rows, c.eps,
                            frames - 1, c.in_channels, c.height, c.width, c.patch_size, s);
}

// gradients of every parameter for the upstream gradient d_out [B,T,C,H,W] of the last forward's output; OVERWRITES the grads buffer
int dfot_dit_train_backward(dfot_dit_train_t h, const float* d_out, void* stream) {
  DFOT_REQUIRE(h && d_out, DFOT_ERR_ARG, "train_backward: null argument");
  DFOT_REQUIRE(h->batch > 0 && h->x_saved, DFOT_ERR_STATE, "train_backward: no forward to differentiate");
  const dfot_dit_config& c = h->cfg;
  const bool facmat = c.variant == 1;
  const int batch = h->batch, tokens = h->tokens, n = tokens * h->P, hd = c.hidden_size, P = h->P, frames = batch * tokens, nd = c.noise_dim;
  const int fp = h->fp, E = c.embed_col_dim;
  const int seq = facmat ? P : n, nseq = facmat ? frames : batch;
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)batch * n;
  const float* p = h->params_f32;
  float* G = h->grads;
  int rc = 0;
  if ((rc = zero_fill(G, (size_t)h->total * sizeof(float), s))) return rc;
  if ((rc = zero_fill(h->dmod, (size_t)fp * h->ldt * sizeof(float), s))) return rc;
  const dim3 fgrid(frames, cdiv(hd, 256), TR_CHUNKS);

  // ---- final layer: out = Linear(mfin), mfin = LN(x_fin)(1 + scale) + shift ----
  if ((rc = zero_fill(h->dyp, (size_t)rows * 64 * sizeof(bf16), s)) || (rc = zero_fill(h->dyt, (size_t)256 * rows * sizeof(bf16), s))) return rc;
  hipLaunchKernelGGL(final_gather_kernel, dim3(cdiv(rows * h->oc, 256)), dim3(256), 0, s, d_out, h->dyp, h->dyt, rows, c.in_channels, c.height,
                     c.width, c.patch_size);
  launch_colsum_bf16(h->dyp, G + h->o_fin_b, rows, h->oc, 64L, s);
  DFOT_CHECK_HIP(hipGetLastError());
  float *dY = h->dX, *dN = h->dX2;  // gradient of the current block's output / scratch for the next one
  if ((rc = launch_ln_mod(h->x_fin, dN, h->mfin, h->mod_table, h->idx, h->ldt, h->mod_final, hd, P, (int)rows, c.eps, frames - 1, s))) return rc;
  if ((rc = tr_transpose(h->mfin, h->T2, (int)rows, hd, s))) return rc;                                   // mfin^T [hd][rows]
  if ((rc = tr_wgrad(h->dyt, h->T2, 256, hd, (int)rows, h->dwf, s, h->wg_ws, h->wg_ws_floats))) return rc;   // dWf in the first oc rows (18 tiles: K split)
  DFOT_CHECK_HIP(hipMemcpyAsync(G + h->o_fin_w, h->dwf, (size_t)h->oc * hd * sizeof(float), hipMemcpyDeviceToDevice, s));
  if ((rc = tr_gemm_f32(h->dyp, 64, h->wfT, (int)rows, hd, 64, dY, hd, nullptr, s))) return rc;             // d mfin
  auto ln_bwd = [&](const float* x_in, long off) -> int {  // dY holds dm; leaves dx in dY
    int r = launch_ln_bwd_rows(dY, x_in, h->mod_table, h->ldt, off, dN, h->stats, hd, P, (int)rows, c.eps, s);
    if (r) return r;
    hipLaunchKernelGGL(ln_bwd_frames_kernel, fgrid, dim3(256), 0, s, dY, x_in, h->stats, h->dmod, h->ldt, off, hd, P);
    DFOT_CHECK_HIP(hipGetLastError());
    std::swap(dY, dN);
    return DFOT_OK;
  };
  auto gate_bwd = [&](const bf16* a, long gate_off, float* dbias) -> int {  // h->da = dY * gate, dgate into dmod, row sums of da into dbias
    hipLaunchKernelGGL(gate_bwd_kernel, fgrid, dim3(256), 0, s, dY, a, h->mod_table, h->ldt, gate_off, h->da, h->dmod, dbias, hd, P);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  };
  auto frames_sum = [&](const bf16* src, float* dst, long per_frame) -> int {
    hipLaunchKernelGGL(frames_sum_bf16_kernel, dim3(cdiv(per_frame / 4, 256)), dim3(256), 0, s, src, dst, frames, per_frame);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  };
  auto permute = [&](const bf16* src, bf16* dst, int R) -> int {  // [frames][R][hd] -> [R][frames][hd]
    hipLaunchKernelGGL(permute_frames_kernel, dim3(cdiv((long)frames * R * (hd / 8), 256)), dim3(256), 0, s, src, dst, frames, R, hd);
    DFOT_CHECK_HIP(hipGetLastError());
    return DFOT_OK;
  };
  // dW[M][N] = dy^T x over `nrows` tokens: operands in place when the shape allows, else through transposed copies
  auto wgrad = [&](const bf16* dyp, int M, const bf16* xp, int N, long nrows, float* outp) -> int {
    int r = tr_wgrad_nt(dyp, M, xp, N, M, N, nrows, outp, s, h->wg_ws, h->wg_ws_floats);
    if (r != DFOT_ERR_STATE) return r;
    if ((r = tr_transpose(dyp, h->T1, (int)nrows, M, s)) || (r = tr_transpose(xp, h->dqkvT, (int)nrows, N, s))) return r;
    return tr_wgrad(h->T1, h->dqkvT, M, N, (int)nrows, outp, s, h->wg_ws, h->wg_ws_floats);
  };
  if ((rc = ln_bwd(h->x_fin, h->mod_final))) return rc;

  // ---- blocks, last to first ----
  for (int bi = (int)h->blocks.size() - 1; bi >= 0; --bi) {
    TrainBlock& b = h->blocks[bi];
    if (const int mh = b.mh) {  // out = m2 + gate2 * y, y = GELU(m2 W1^T + b1) W2^T + b2
      if ((rc = gate_bwd(b.y, b.mod2 + 2 * hd, G + b.o_fc2_b))) return rc;
      if ((rc = tr_gemm_bf16(h->da, hd, b.w_fc2T, (int)rows, mh, hd, nullptr, h->dh, mh, s))) return rc;        // dh = dy W2
      if ((rc = wgrad(h->da, hd, b.hact, mh, rows, G + b.o_fc2_w))) return rc;                                   // dW2 = dy^T h
      hipLaunchKernelGGL(gelu_kernel, dim3(cdiv(rows * mh / 8, 256)), dim3(256), 0, s, b.u, (bf16*)nullptr, h->dh, rows * mh / 8);  // du
      launch_colsum_bf16(h->dh, G + b.o_fc1_b, rows, mh, (long)mh, s);
      DFOT_CHECK_HIP(hipGetLastError());
      if ((rc = tr_gemm_f32(h->dh, mh, b.w_fc1T, (int)rows, hd, mh, dY, hd, dY, s))) return rc;                    // dm2 = dY + du W1
      if ((rc = wgrad(h->dh, mh, b.m2, hd, rows, G + b.o_fc1_w))) return rc;                                    // dW1 = du^T m2
      if ((rc = ln_bwd(b.x_mid, b.mod2))) return rc;
    }
    if (!b.matrix) {
      if ((rc = gate_bwd(b.a, b.mod + 2 * hd, G + b.o_proj_b))) return rc;
      if ((rc = tr_gemm_bf16(h->da, hd, b.w_projT, (int)rows, hd, hd, nullptr, h->dO, hd, s))) return rc;      // dO = da Wp
      if ((rc = wgrad(h->da, hd, b.o, hd, rows, G + b.o_proj_w))) return rc;  // dWp = da^T o
      if ((rc = launch_attention_bwd_delta(b.o, h->dO, hd, h->delta, nseq, c.num_heads, seq, h->d, s))) return rc;
      if ((rc = launch_attention_bwd(b.q, b.k, b.v, h->dO, hd, b.lse, h->delta, h->dq, h->dk, h->dv, nseq, c.num_heads, seq, h->d, s))) return rc;
      hipLaunchKernelGGL(qkv_grad_pack_kernel, dim3(cdiv(rows * (3 * hd / 8), 256)), dim3(256), 0, s, h->dq, h->dk, h->dv,
                         facmat ? (const float*)nullptr : h->rope_cs, h->dqkv, rows, seq, c.num_heads, h->d, h->dstride);
      launch_colsum_bf16(h->dqkv, G + b.o_qkv_b, rows, 3 * hd, (long)3 * hd, s);
      DFOT_CHECK_HIP(hipGetLastError());
      if ((rc = tr_gemm_f32(h->dqkv, 3 * hd, b.w_qkvT, (int)rows, hd, 3 * hd, dY, hd, dY, s))) return rc;       // dm = dY + dqkv Wqkv (in place)
      if ((rc = wgrad(h->dqkv, 3 * hd, b.m, hd, rows, G + b.o_qkv_w))) return rc;  // dWqkv = dqkv^T m
    } else {
      const bool bias = b.o_qkv_bias >= 0;
      const int fe = frames * E;
      const long fk = (long)frames * hd;  // contraction length of the left-factor gradients
      // a = s V' + bias'[p] ; s[f][p][d] = sum_e U'[e][p] o[f][e][d]
      if ((rc = gate_bwd(b.a, b.mod + 2 * hd, h->scratch_f))) return rc;
      if (bias && (rc = frames_sum(h->da, G + b.o_proj_bias, (long)P * hd))) return rc;
      if ((rc = tr_gemm_bf16(h->da, hd, b.pv_s, (int)rows, hd, hd, nullptr, h->dO, hd, s))) return rc;            // ds = da V'^T  (V' stored (in, out))
      if ((rc = wgrad(b.sfac, hd, h->da, hd, rows, G + b.o_proj_v))) return rc;                                  // dV'[in][out] = s^T da
      if ((rc = tr_transpose(h->dO, h->mt, P, hd, s, frames))) return rc;                                        // ds^T per frame [hd][P]
      if ((rc = tr_gemm_bf16(h->mt, P, b.pu_s, frames * hd, E, P, nullptr, h->do2, E, s, 0, hd))) return rc;       // do[f][e][d] = sum_p U'[e][p] ds[f][p][d]
      // dU'[e][p] = sum_{f,d} o[f][e][d] ds[f][p][d]: operands regrouped to [e][(f,d)] / [p][(f,d)]; E rows padded to 128
      DFOT_CHECK_HIP(hipMemsetAsync(h->perm_a, 0, (size_t)128 * fk * sizeof(bf16), s));
      if ((rc = permute(b.o2, h->perm_a, E)) || (rc = permute(h->dO, h->perm_b, P))) return rc;
      if ((rc = tr_wgrad(h->perm_a, h->perm_b, 128, P, (int)fk, h->dwf, s, h->wg_ws, h->wg_ws_floats))) return rc;
      DFOT_CHECK_HIP(hipMemcpyAsync(G + b.o_proj_u, h->dwf, (size_t)E * P * sizeof(float), hipMemcpyDeviceToDevice, s));
      // attention over the frames
      {
        const int hn = E / c.num_col_heads, hdr = hd / c.num_row_heads;
        const int nheads = batch * c.num_col_heads * c.num_row_heads;
        DFOT_CHECK_HIP(hipMemsetAsync(h->ma_sc, 0, (size_t)nheads * 2 * tokens * tokens * sizeof(float), s));
        hipLaunchKernelGGL(matrix_attn_bwd_scores_kernel, dim3(nheads, MA_CHUNKS), dim3(256), 0, s, b.z, h->do2, h->ma_sc, tokens, E, hd,
                           c.num_col_heads, c.num_row_heads);
        hipLaunchKernelGGL(matrix_attn_bwd_apply_kernel, dim3(nheads, MA_CHUNKS), dim3(256), 0, s, b.z, h->do2, h->ma_sc, h->dz, tokens, E, hd,
                           c.num_col_heads, c.num_row_heads, 1.0f / sqrtf((float)hn * (float)hdr));
        DFOT_CHECK_HIP(hipGetLastError());
      }
      // z = w1 V + bias[e] ; w1[f][e][d] = sum_p U[p][e] m[f][p][d]
      if (bias && (rc = frames_sum(h->dz, G + b.o_qkv_bias, (long)E * 3 * hd))) return rc;
      if ((rc = tr_gemm_bf16(h->dz, 3 * hd, b.v_s, fe, hd, 3 * hd, nullptr, h->dw1, hd, s))) return rc;           // dw1 = dz V^T  (V stored (in, out))
      if ((rc = wgrad(b.w1, hd, h->dz, 3 * hd, fe, G + b.o_qkv_v))) return rc;                                   // dV[in][out] = w1^T dz
      if ((rc = tr_transpose(h->dw1, h->mt, E, hd, s, frames))) return rc;                                       // dw1^T per frame [hd][E]
      if ((rc = tr_gemm_bf16(h->mt, E, b.u_s, frames * hd, P, E, nullptr, h->dO, P, s, 0, hd))) return rc;         // dm[f][p][d] = sum_e U[p][e] dw1[f][e][d]
      hipLaunchKernelGGL(add_bf16_kernel, dim3(cdiv(rows * hd / 4, 256)), dim3(256), 0, s, dY, h->dO, rows * hd / 4);
      DFOT_CHECK_HIP(hipGetLastError());
      // dU[p][e] = sum_{f,d} m[f][p][d] dw1[f][e][d]
      if ((rc = permute(b.m, h->perm_a, P)) || (rc = permute(h->dw1, h->perm_b, E))) return rc;
      if ((rc = tr_wgrad(h->perm_a, h->perm_b, P, E, (int)fk, G + b.o_qkv_u, s, h->wg_ws, h->wg_ws_floats))) return rc;
    }
    if ((rc = ln_bwd(b.x_in, b.mod))) return rc;
  }

  // ---- patch embedding ----
  launch_pe_wgrad(dY, h->x_saved, G + h->o_pe_w, G + h->o_pe_b, c.in_channels, c.height, c.width, c.patch_size, hd, rows, s);
  DFOT_CHECK_HIP(hipGetLastError());
  h->d_embed = dY;  // gradient w.r.t. the patch-embedding output: dfot_dit_train_input_grad turns it into d / d x

  // ---- modulation Linears: table = SiLU(c) W_mod^T + b_mod over the frames ----
  hipLaunchKernelGGL(frames_colsum_kernel, dim3(cdiv(h->ldt, 256)), dim3(256), 0, s, h->dmod, h->dbmod, frames, h->ldt);
  DFOT_CHECK_HIP(hipGetLastError());
  if ((rc = launch_f32_to_bf16(h->dmod, h->dmod_bf, (long)fp * h->ldt, s))) return rc;
  hipLaunchKernelGGL(pack_transpose_kernel, dim3(cdiv((long)fp * h->ldt, 256)), dim3(256), 0, s, h->dmod, h->dmodT, fp, (int)h->ldt);
  DFOT_CHECK_HIP(hipGetLastError());
  if ((rc = tr_transpose(h->semb, h->sembT, fp, hd, s))) return rc;
  // dW_mod = dmod^T SiLU(c), one GEMM per modulation Linear written straight into its gradient tensor (K = frames is short: the
  // fp32 output dominates, so it is written once, where it belongs)
  auto scatter = [&](long col, int nrow, long o_w, long o_b) -> int {
    int r = tr_gemm_f32(h->dmodT + col * fp, fp, h->sembT, nrow, hd, fp, G + o_w, hd, nullptr, s);
    if (r) return r;
    DFOT_CHECK_HIP(hipMemcpyAsync(G + o_b, h->dbmod + col, (size_t)nrow * sizeof(float), hipMemcpyDeviceToDevice, s));
    return DFOT_OK;
  };
  for (TrainBlock& b : h->blocks) {
    if ((rc = scatter(b.mod, 3 * hd, b.o_mod_w, b.o_mod_b))) return rc;
    if (b.mh && (rc = scatter(b.mod2, 3 * hd, b.o_mod2_w, b.o_mod2_b))) return rc;
  }
  if ((rc = scatter(h->mod_final, 2 * hd, h->o_fmod_w, h->o_fmod_b))) return rc;
  // d SiLU(c) = dmod W_mod: 256 x hidden output over K = every modulation column (99072 for DiT/XL): K split into partial buffers
  if ((rc = tr_wgrad(h->dmod_bf, h->w_modT, fp, hd, (int)h->ldt, h->dsemb, s, h->wg_ws, h->wg_ws_floats))) return rc;

  // ---- noise-level embedding MLP (frames x hidden, fp32) ----
  const long fh = (long)frames * hd;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(cdiv(fh, 256)), dim3(256), 0, s, h->dsemb, h->cemb, h->dc, fh);
  if (facmat) hipLaunchKernelGGL(diff_grad_kernel, dim3(cdiv(2 * hd, 256)), dim3(256), 0, s, h->dc, G + h->o_diff, frames, tokens, hd);
  hipLaunchKernelGGL(small_wgrad_kernel, dim3(cdiv((long)hd * hd, 256)), dim3(256), 0, s, h->dc, h->a1, G + h->o_t_w2, G + h->o_t_b2, frames, hd, hd);
  hipLaunchKernelGGL(small_dgrad_kernel, dim3(cdiv(fh, 256)), dim3(256), 0, s, h->dc, p + h->o_t_w2, h->da1, frames, hd, hd);
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(cdiv(fh, 256)), dim3(256), 0, s, h->da1, h->h1, h->dh1, fh);
  hipLaunchKernelGGL(small_wgrad_kernel, dim3(cdiv((long)hd * nd, 256)), dim3(256), 0, s, h->dh1, h->feat, G + h->o_t_w1, G + h->o_t_b1, frames, hd, nd);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

// d(sum(out * d_out)) / d x for the last forward / backward pair: the data gradient of the patch embedding.  What reconstruction
// guidance differentiates (discrete_diffusion.py:485-513: the prediction w.r.t. x_t); the positional embedding and the noise-level
// path do not depend on x.
int dfot_dit_train_input_grad(dfot_dit_train_t h, float* dx, void* stream) {
  DFOT_REQUIRE(h && dx, DFOT_ERR_ARG, "train_input_grad: null argument");
  DFOT_REQUIRE(h->batch > 0 && h->d_embed, DFOT_ERR_STATE, "train_input_grad: run dfot_dit_train_backward first");
  const dfot_dit_config& c = h->cfg;
  const long rows = (long)h->batch * h->tokens * h->P;
  const int kdim = c.in_channels * c.patch_size * c.patch_size;
  hipLaunchKernelGGL(pe_dgrad_kernel, dim3(cdiv(rows * kdim, 256)), dim3(256), 0, (hipStream_t)stream, h->d_embed, h->params_f32 + h->o_pe_w, dx,
                     c.in_channels, c.height, c.width, c.patch_size, c.hidden_size, rows);
  DFOT_CHECK_HIP(hipGetLastError());
  return DFOT_OK;
}

/* ---- flat-buffer optimizer pieces (generic) ---- */
int dfot_vloss_grad(const float* x, const float* noise, const float* v, const float* a, const float* sigma, const float* coef, float* dv,
                    int batch, int tokens, int64_t frame_elems, int vspace, void* stream) {
  DFOT_REQUIRE(x && noise && v && a && sigma && coef && dv, DFOT_ERR_ARG, "vloss_grad: null argument");
  return launch_vloss_grad(x, noise, v, a, sigma, coef, dv, batch * tokens, frame_elems, vspace != 0, (hipStream_t)stream);
}
int dfot_sumsq(const float* x, int64_t n, float* out, void* stream) {
  DFOT_REQUIRE(x && out && n > 0, DFOT_ERR_ARG, "sumsq: bad argument");
  DFOT_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream));
  return launch_sumsq(x, n, out, (hipStream_t)stream);
}
int dfot_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, const float* grad_sumsq, float max_grad_norm, float* ema, float ema_decay,
                    void* stream) {
  DFOT_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0, DFOT_ERR_ARG, "adamw_step: bad argument");
  return launch_adamw(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_sumsq, max_grad_norm, ema,
                      ema_decay, (hipStream_t)stream);
}

}  // extern "C"
