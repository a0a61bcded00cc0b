// Shared device/host helpers for libdevis_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/devis_hip.h"

typedef __bf16 bf16;

namespace devis {

extern thread_local int g_last_hip_error;

inline int check_hip(hipError_t e) {
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return DEVIS_ERR_HIP;
  }
  return DEVIS_OK;
}
inline int launch_status() { return check_hip(hipGetLastError()); }

template <typename T> __device__ __forceinline__ float ld(const T* p) { return (float)(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v) { *p = (T)v; }

__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// two floats -> two bf16 (round to nearest even) in one dword, low half = a.  The vector conversion compiles to ONE
// v_cvt_pk_bf16_f32; converting the halves one by one costs four instructions (two converts, a shift, an or).
typedef __attribute__((ext_vector_type(2))) float devis_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 devis_bf16x2;
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  union { devis_bf16x2 v; unsigned u; } r;
  r.v = __builtin_convertvector((devis_f32x2){a, b}, devis_bf16x2);
  return r.u;
}

// wave64 reductions (DPP/shuffle; no LDS)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// dispatch a templated launcher on the storage dtype
#define DEVIS_DISPATCH_DTYPE(dtype, ...)                      \
  do {                                                        \
    if ((dtype) == DEVIS_F32) {                               \
      typedef float T;                                        \
      __VA_ARGS__;                                            \
    } else if ((dtype) == DEVIS_BF16) {                       \
      typedef bf16 T;                                         \
      __VA_ARGS__;                                            \
    } else {                                                  \
      return DEVIS_ERR_BAD_ARG;                               \
    }                                                         \
  } while (0)

}  // namespace devis
