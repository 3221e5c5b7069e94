// Shared device/host helpers for the DFoT HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace dfot {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define DFOT_LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define DFOT_GLOBAL_PTR(p) ((const void __attribute__((address_space(1)))*)(p))

// XCD-aware workgroup order (cdna_hip_programming.md T1): blocks b and b+8 share an XCD (and its L2), so hand each
// XCD a CONTIGUOUS range of the logical tile order; neighbours in that order (same A rows / same K,V head) then hit
// the same L2.  Bijective for any grid size.  Placement affects speed only.
__device__ __forceinline__ int xcd_remap(int orig, int nwg, int enable = 1) {
  if (!enable) return orig;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division: the division is ten instructions per element, and
// the GEMM epilogues that apply SiLU to a 256x256 tile run with the matrix pipe idle (14 of the 118 us of the level-2 fused
// projection were this division; profiles/r03 notes in DESIGN.md)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

int tuning_flag(const char* name, int dflt);  // DFOT_<NAME> environment override, read once (A/B experiments)

// last error text for the C ABI (thread-local: one sampler thread per process in practice)
void set_error(const char* fmt, ...);
const char* get_error();

#define DFOT_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::dfot::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return DFOT_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

#define DFOT_REQUIRE(cond, code, ...)      \
  do {                                     \
    if (!(cond)) {                         \
      ::dfot::set_error(__VA_ARGS__);      \
      return code;                         \
    }                                      \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace dfot
