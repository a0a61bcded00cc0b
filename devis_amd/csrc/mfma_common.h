// Small helpers shared by the MFMA kernels (gfx950).
#pragma once
#include "common.h"

namespace devis {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// 16 bytes in flight between a global load and an LDS store.  Use this, not HIP's uint4, for register
// arrays that live across loop iterations: arrays of the uint4 struct are left in scratch memory there.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
  union { bf16 h[2]; unsigned u; } r;
  r.h[0] = (bf16)a;
  r.h[1] = (bf16)b;
  return r.u;
}

// relu(scale * x + shift) on 8 packed bf16 (the fused norm prologue)
__device__ __forceinline__ uint4 affine_relu8(uint4 v, const float* sc, const float* sh) {
  unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = fmaxf(fmaf(bf_lo(w[j]), sc[2 * j], sh[2 * j]), 0.f);
    const float hi = fmaxf(fmaf(bf_hi(w[j]), sc[2 * j + 1], sh[2 * j + 1]), 0.f);
    w[j] = pack_bf2(lo, hi);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ u32x4 affine_relu8(u32x4 v, const float* sc, const float* sh) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = fmaxf(fmaf(bf_lo(v[j]), sc[2 * j], sh[2 * j]), 0.f);
    const float hi = fmaxf(fmaf(bf_hi(v[j]), sc[2 * j + 1], sh[2 * j + 1]), 0.f);
    v[j] = pack_bf2(lo, hi);
  }
  return v;
}

}  // namespace devis
