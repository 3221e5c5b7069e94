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

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

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
