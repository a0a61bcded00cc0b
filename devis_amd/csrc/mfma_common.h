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
__device__ __forceinline__ unsigned pack_bf2(float a, float b) { return pack_bf16x2(a, b); }

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

// ---- normalisation statistics gathered in a conv epilogue (no separate pass over the tensor) ----
// mode 1: per-channel {sum y, sum y^2} of the stored (bf16-rounded) output: GroupNorm/BatchNorm forward moments.
// mode 2: the kernel runs as a dgrad and y is the gradient w.r.t. relu(scale*xref+shift):
//         g = y * [scale*xref+shift > 0] -> {sum g, sum g*xref}: the two moments norm backward needs.
// A block leaves one partial per channel it owns: partial[n][chunk][C][2] (fp64), chunk = its tile index
// inside sample n -- the layout k_moments writes, so the *_finalize_partials kernels take either.
struct StatsArgs {
  int mode;             // 0 = off
  int chunks;           // chunks per sample (host: must equal the kernel's tiles per sample)
  int C;                // channels of y
  double* partial;
  const bf16* xref;     // mode 2: the tensor the norm was applied to (same shape as y)
  const float* sc;      // mode 2: [N][C]
  const float* sh;
};

// one lane's 4 consecutive channels of one voxel; f0/f1 = running fp32 sums
template <int STATS>
__device__ __forceinline__ void stats_accum(uint2 o, uint2 xr, const float* scv, const float* shv, float* f0, float* f1) {
  const float v[4] = {bf_lo(o.x), bf_hi(o.x), bf_lo(o.y), bf_hi(o.y)};
  if (STATS == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { f0[j] += v[j]; f1[j] = fmaf(v[j], v[j], f1[j]); }
  } else if (STATS == 2) {
    const float x[4] = {bf_lo(xr.x), bf_hi(xr.x), bf_lo(xr.y), bf_hi(xr.y)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = fmaf(x[j], scv[j], shv[j]) > 0.f ? v[j] : 0.f;
      f0[j] += g;
      f1[j] = fmaf(g, x[j], f1[j]);
    }
  }
}

// Block-level finish for the MFMA conv kernels' output layout (lane = 4 consecutive cout lq*4.. of m-tile m,
// voxel l15).  f0/f1[m][j]: the lane's fp32 sums.  Lanes sharing channels combine in fp64 with a fixed xor
// tree, the four waves through LDS in a fixed order (deterministic).  `red` = >= 4*MT*32 doubles of LDS that
// no wave still reads (callers barrier before).  pout = &partial[n][chunk][co_base][0].
template <int MT>
__device__ __forceinline__ void stats_finish(const float (&f0)[MT][4], const float (&f1)[MT][4], double* red, double* pout) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = (double)f0[m][j], b = (double)f1[m][j];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (l15 == 0) {
        red[((wave * MT + m) * 16 + lq * 4 + j) * 2 + 0] = a;
        red[((wave * MT + m) * 16 + lq * 4 + j) * 2 + 1] = b;
      }
    }
  __syncthreads();
  if (tid < MT * 32) pout[tid] = ((red[tid] + red[MT * 32 + tid]) + red[2 * MT * 32 + tid]) + red[3 * MT * 32 + tid];
}

// Same contract as stats_finish, cheaper for small tiles: the 16 lanes that share channels (one DPP row) add
// their fp32 sums with four DPP adds per value (quad swaps, then the half-row and row mirrors: every lane ends
// with the row total) instead of eight 64-bit cross-lane permutes; the four waves then add up in fp64 through
// LDS in a fixed order.  At most 16 * NT addends meet in fp32.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);     // row_half_mirror
  v = dpp_add<0x140>(v);     // row_mirror
  return v;
}
template <int MT>
__device__ __forceinline__ void stats_finish_dpp(const float (&f0)[MT][4], const float (&f1)[MT][4], double* red, double* pout) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = row16_sum(f0[m][j]), b = row16_sum(f1[m][j]);
      if (l15 == 0) {
        red[((wave * MT + m) * 16 + lq * 4 + j) * 2 + 0] = (double)a;
        red[((wave * MT + m) * 16 + lq * 4 + j) * 2 + 1] = (double)b;
      }
    }
  __syncthreads();
  if (tid < MT * 32) pout[tid] = ((red[tid] + red[MT * 32 + tid]) + red[2 * MT * 32 + tid]) + red[3 * MT * 32 + tid];
}

}  // namespace devis
