// Second stage of the deterministic weight-gradient reductions: sums the per-block fp32 partials the
// wgrad kernels leave in the workspace and scatters the result into the PyTorch weight layout.
// A block is LANES float4 columns x KG partial groups: lanes run along the partial (coalesced reads),
// group g adds partials g, g+KG, ... (4 independent loads in flight), the KG groups combine through LDS
// in a fixed-shape tree.  No atomics: the same inputs always give the same bits.
#pragma once
#include "common.h"

namespace devis {

enum { RED_CONV3 = 0, RED_PW = 1, RED_CONV3F = 2 };

struct RedArgs {
  const float* part;    // [nparts][pstride]
  const float* bpart;   // [nbparts][Cb] or null
  float* dw;
  float* dbias;         // null: no bias
  int nparts, nbparts, Cb;
  int64_t pstride;      // floats between partials
  int64_t total;        // floats of a partial that are summed (multiple of 4)
  int mode;             // RED_CONV3: o = (t*A + co)*B + ci -> dw[(co*cin_real + ci)*T + t], ci < cin_real, T = total/(A*B)
  int A, B, cin_real;   // RED_PW:    o = p*B + mn          -> dw[mn*A + p]   (A = taps T, B = M*N)
                        // RED_CONV3F (27 taps, accumulator order of k_wgrad3_slab_march): o = ((((slab*27 + t)*2 + mo)*2 + mi)*64
                        //   + lane)*4 + j, slab = (co/32)*(B/32) + ci/32 -> co = co0 + mo*16 + (lane>>4)*4 + j, ci = ci0 + mi*16 + lane%16
  // RED_PW with channel slabs (nms > 0): mn = (z*msub + m)*nsub + n, z = ns*nms + ms -> row ms*msub + m, column ns*nsub + n
  int nms = 0, msub = 0, nsub = 0, ntot = 0;
};

// LANES float4 columns per block x KG = 1024/LANES partial groups.  Small outputs (a 16x16x27 weight is
// 1728 columns, a 1x1 weight 128) take narrow blocks so that the launch still covers the chip.
template <int MODE, int LANES>
__global__ void __launch_bounds__(1024)
k_reduce_partials(RedArgs a) {
  constexpr int KG = 1024 / LANES;
  __shared__ float4 red[1024];
  const int lane = threadIdx.x % LANES, kg = threadIdx.x / LANES;
  const int64_t n4 = a.total / 4;
  const int64_t o4 = (int64_t)blockIdx.x * LANES + lane;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (o4 < n4) {
    const float* p = a.part + o4 * 4;
    int k = kg;
    for (; k + 3 * KG < a.nparts; k += 4 * KG) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (int64_t)k * a.pstride);
      const float4 v1 = *reinterpret_cast<const float4*>(p + (int64_t)(k + KG) * a.pstride);
      const float4 v2 = *reinterpret_cast<const float4*>(p + (int64_t)(k + 2 * KG) * a.pstride);
      const float4 v3 = *reinterpret_cast<const float4*>(p + (int64_t)(k + 3 * KG) * a.pstride);
      s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
      s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
      s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
      s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
    }
    for (; k < a.nparts; k += KG) {
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)k * a.pstride);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  } else if (a.dbias) {
    const int64_t c0 = (o4 - n4) * 4;
    float* sv = reinterpret_cast<float*>(&s);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < a.Cb)
        for (int k = kg; k < a.nbparts; k += KG) sv[j] += a.bpart[(int64_t)k * a.Cb + c0 + j];
  }
  red[kg * LANES + lane] = s;
  __syncthreads();
  // fixed-shape tree over the partial groups: same inputs, same bits
#pragma unroll
  for (int h = KG / 2; h >= 1; h >>= 1) {
    if (kg < h) {
      float4 u = red[kg * LANES + lane];
      const float4 v = red[(kg + h) * LANES + lane];
      u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
      red[kg * LANES + lane] = u;
    }
    __syncthreads();
  }
  if (kg != 0) return;
  s = red[lane];
  const float* sv = reinterpret_cast<const float*>(&s);
  if (o4 < n4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t o = o4 * 4 + j;
      if (MODE == RED_CONV3) {
        const int ci = (int)(o % a.B), co = (int)((o / a.B) % a.A), t = (int)(o / ((int64_t)a.A * a.B));
        const int T = (int)(a.total / ((int64_t)a.A * a.B));          // 27 (3x3x3) or 9 (3x3)
        if (ci < a.cin_real) a.dw[((int64_t)co * a.cin_real + ci) * T + t] = sv[j];
      } else if (MODE == RED_CONV3F) {
        const int ln = (int)((o >> 2) & 63), mi = (int)((o >> 8) & 1), mo = (int)((o >> 9) & 1);
        const int ts = (int)(o >> 10), t = ts % 27, slab = ts / 27, nci = a.B >> 5;
        const int co = (slab / nci) * 32 + mo * 16 + (ln >> 4) * 4 + j, ci = (slab % nci) * 32 + mi * 16 + (ln & 15);
        a.dw[((int64_t)co * a.B + ci) * 27 + t] = sv[j];
      } else {
        const int64_t mn = o % a.B;
        const int p = (int)(o / a.B);
        if (a.nms > 0) {
          const int n = (int)(mn % a.nsub), m = (int)((mn / a.nsub) % a.msub), z = (int)(mn / ((int64_t)a.nsub * a.msub));
          const int row = (z % a.nms) * a.msub + m, col = (z / a.nms) * a.nsub + n;
          a.dw[((int64_t)row * a.ntot + col) * a.A + p] = sv[j];
        } else {
          a.dw[mn * a.A + p] = sv[j];
        }
      }
    }
  } else if (a.dbias) {
    const int64_t c0 = (o4 - n4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < a.Cb) a.dbias[c0 + j] = sv[j];
  }
}

template <int LANES>
static inline void launch_reduce_lanes(const RedArgs& a, int64_t n4, hipStream_t s) {
  const dim3 grid((unsigned)cdiv(n4, LANES));
  if (a.mode == RED_CONV3) hipLaunchKernelGGL((k_reduce_partials<RED_CONV3, LANES>), grid, dim3(1024), 0, s, a);
  else if (a.mode == RED_CONV3F) hipLaunchKernelGGL((k_reduce_partials<RED_CONV3F, LANES>), grid, dim3(1024), 0, s, a);
  else hipLaunchKernelGGL((k_reduce_partials<RED_PW, LANES>), grid, dim3(1024), 0, s, a);
}

static inline int launch_reduce_partials(const RedArgs& a, hipStream_t s) {
  const int64_t n4 = a.total / 4 + (a.dbias ? (a.Cb + 3) / 4 : 0);
  if (n4 >= 64 * 128) launch_reduce_lanes<64>(a, n4, s);
  else if (n4 >= 16 * 64) launch_reduce_lanes<16>(a, n4, s);
  else launch_reduce_lanes<4>(a, n4, s);
  return launch_status();
}

}  // namespace devis
