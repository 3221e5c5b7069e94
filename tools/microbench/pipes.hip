// Instruction-rate microbenchmark for one gfx950 SIMD: cycles (s_memtime) per wave-level instruction for the vector, transcendental and
// matrix pipes, alone and interleaved, at 1 / 2 / 4 waves per SIMD.  Diagnostic tool, not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -o pipes pipes.hip && ./pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int ITER = 2000;

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * i); b[i] = (__bf16)(0.02f * i); }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
    if constexpr (MODE == 0) {  // 8 independent v_exp_f32
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (MODE == 1) {  // 8 independent v_mul_f32
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i]));
    } else if constexpr (MODE == 2) {  // 4 independent v_pk_mul_f32 (8 elements)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 p = {v[2 * i], v[2 * i + 1]};
        asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
        v[2 * i] = p[0], v[2 * i + 1] = p[1];
      }
    } else if constexpr (MODE == 3) {  // 4 independent MFMA 32x32x16 bf16
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    } else if constexpr (MODE == 4) {  // 4 MFMA, each followed by 2 v_exp_f32 (32 cycles of quarter-rate work per MFMA if exp = 16)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[2 * i]), "+v"(v[2 * i + 1]));
      }
    } else if constexpr (MODE == 5) {  // 4 MFMA, each followed by 8 v_mul_f32
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[j]));
      }
    } else if constexpr (MODE == 6) {  // 8 v_cvt_pk_bf16_f32
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
    } else if constexpr (MODE == 7) {  // 4 MFMA 16x16x32 bf16 (f32x4 accumulators)
      typedef __attribute__((ext_vector_type(4))) float f32x4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 c = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        acc[i][0] = c[0], acc[i][1] = c[1], acc[i][2] = c[2], acc[i][3] = c[3];
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, float* out, long long* cyc, int blocks) {
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (long long c : h) avg += c;
    avg /= blocks;
    // s_memtime ticks at a fixed 100 MHz on this part: report wall time per wave-instruction too
    const double insts = (double)ITER * per_iter * waves_per_simd;  // per SIMD
    printf("%-34s waves/SIMD %d: %8.2f ticks/inst/SIMD  %8.3f ns/inst/SIMD  (kernel %.3f ms, %d blocks)\n", name, waves_per_simd, avg / insts,
           ms * 1e6 / insts, ms, blocks);
  }
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;  // one workgroup per CU
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * 1024);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  run<0>("v_exp_f32 x8", 8, out, cyc, blocks);
  run<1>("v_mul_f32 x8", 8, out, cyc, blocks);
  run<2>("v_pk_mul_f32 x4", 4, out, cyc, blocks);
  run<6>("v_cvt_pk_bf16_f32 x8", 8, out, cyc, blocks);
  run<3>("mfma_32x32x16_bf16 x4", 4, out, cyc, blocks);
  run<7>("mfma_16x16x32_bf16 x4", 4, out, cyc, blocks);
  run<4>("[mfma32 + 2 exp] x4 (per group)", 4, out, cyc, blocks);
  run<5>("[mfma32 + 8 mul] x4 (per group)", 4, out, cyc, blocks);
  return 0;
}
