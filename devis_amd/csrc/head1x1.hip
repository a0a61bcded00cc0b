// The network's last layer, endconv: a 1x1 conv from 16 feature channels to K = 4 class logits
// (UNet3DZoo.py:274,303) at full resolution.  On the MFMA gather-GEMM kernels it wastes 12 of 16 tile rows
// (forward) or 28 of 32 K lanes (dgrad) and ran at ~1.7 TB/s; it is a 64-FMA-per-voxel elementwise problem,
// so here one thread owns a voxel: 32 B in / 8 B out forward, 8 B in / 32 B out backward, weights in registers,
// fully coalesced.  The 2-D U-Nets' conv10 (32 features -> K = 2, UNet2DZoo.py:128,157) is the <32, 2> instance.
// dgrad can gather the norm-backward moments of the DeBlock it feeds (mode 2 of
// mfma_common.h's StatsArgs: g = dx*[sc*xref+sh > 0] -> {sum g, sum g*xref} per channel).
#include "common.h"
#include "mfma_common.h"

namespace devis {

#define H1_VPB 2048      // voxels per block (8 per thread)

__device__ __forceinline__ void unpack8(const uint4 v, float* o) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = bf_lo(w[i]); o[2 * i + 1] = bf_hi(w[i]); }
}

// y[v][co] = b[co] + sum_ci w[co][ci] x[v][ci];  x [NV][CIN], y [NV][COUT], w packed [COUT][CIN]
template <int CIN, int COUT>
__global__ void __launch_bounds__(256)
k_head_fwd(const bf16* __restrict__ x, const bf16* __restrict__ w, const float* __restrict__ bias,
           bf16* __restrict__ y, int64_t NV) {
  float wf[COUT][CIN], bf[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
#pragma unroll
    for (int q = 0; q < CIN / 8; ++q) unpack8(*reinterpret_cast<const uint4*>(w + co * CIN + q * 8), wf[co] + q * 8);
    bf[co] = bias ? bias[co] : 0.f;
  }
  const int64_t v0 = (int64_t)blockIdx.x * H1_VPB;
  const int64_t v1 = v0 + H1_VPB < NV ? v0 + H1_VPB : NV;
  for (int64_t v = v0 + threadIdx.x; v < v1; v += 256) {
    float xf[CIN];
#pragma unroll
    for (int q = 0; q < CIN / 8; ++q) unpack8(*reinterpret_cast<const uint4*>(x + v * CIN + q * 8), xf + q * 8);
    float o[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float a = bf[co];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) a = fmaf(wf[co][ci], xf[ci], a);
      o[co] = a;
    }
    if constexpr (COUT == 4) {
      uint2 r;
      r.x = pack_bf2(o[0], o[1]);
      r.y = pack_bf2(o[2], o[3]);
      *reinterpret_cast<uint2*>(y + v * 4) = r;
    } else {
      *reinterpret_cast<unsigned*>(y + v * 2) = pack_bf2(o[0], o[1]);
    }
  }
}

// dx[v][ci] = sum_co w[co][ci] dy[v][co] (+ statistics).  grid = (chunks, N): blocks never straddle samples.
// (statistics: the <16, 4> instance only)
template <int CIN, int COUT, int STATS>
__global__ void __launch_bounds__(256)
k_head_dgrad(const bf16* __restrict__ dy, const bf16* __restrict__ w, bf16* __restrict__ dx, int64_t V, StatsArgs st) {
  static_assert(!STATS || CIN == 16, "statistics are gathered for 16 feature channels only");
  __shared__ double red[4][16][2];
  float wf[COUT][CIN];
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
#pragma unroll
    for (int q = 0; q < CIN / 8; ++q) unpack8(*reinterpret_cast<const uint4*>(w + co * CIN + q * 8), wf[co] + q * 8);
  }
  const int64_t n = blockIdx.y;
  float f0[16], f1[16], sc[16], sh[16];
  if (STATS) {
#pragma unroll
    for (int c = 0; c < 16; ++c) { f0[c] = 0.f; f1[c] = 0.f; sc[c] = st.sc[n * 16 + c]; sh[c] = st.sh[n * 16 + c]; }
  }
  const int64_t v0 = (int64_t)blockIdx.x * H1_VPB;
  const int64_t v1 = v0 + H1_VPB < V ? v0 + H1_VPB : V;
  for (int64_t vv = v0 + threadIdx.x; vv < v1; vv += 256) {
    const int64_t v = n * V + vv;
    uint2 g = make_uint2(0u, 0u);
    if (COUT == 4) g = *reinterpret_cast<const uint2*>(dy + v * 4);
    else g.x = *reinterpret_cast<const unsigned*>(dy + v * 2);
    uint4 xa = make_uint4(0u, 0u, 0u, 0u), xb = xa;
    if (STATS) {                                  // issued with dy: the stores below would hold a later load back
      xa = *reinterpret_cast<const uint4*>(st.xref + v * 16);
      xb = *reinterpret_cast<const uint4*>(st.xref + v * 16 + 8);
    }
    const float gf[4] = {bf_lo(g.x), bf_hi(g.x), bf_lo(g.y), bf_hi(g.y)};
    float o[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      float a = 0.f;
#pragma unroll
      for (int co = 0; co < COUT; ++co) a = fmaf(wf[co][ci], gf[co], a);
      o[ci] = a;
    }
    uint4 rr[CIN / 8];
#pragma unroll
    for (int q = 0; q < CIN / 8; ++q) {
      rr[q].x = pack_bf2(o[q * 8 + 0], o[q * 8 + 1]); rr[q].y = pack_bf2(o[q * 8 + 2], o[q * 8 + 3]);
      rr[q].z = pack_bf2(o[q * 8 + 4], o[q * 8 + 5]); rr[q].w = pack_bf2(o[q * 8 + 6], o[q * 8 + 7]);
      *reinterpret_cast<uint4*>(dx + v * CIN + q * 8) = rr[q];
    }
    const uint4 ra = rr[0], rb = rr[CIN / 8 - 1];
    if (STATS) {
      float xf[16], of[16];                       // statistics of the stored (rounded) gradient
      unpack8(xa, xf); unpack8(xb, xf + 8);
      unpack8(ra, of); unpack8(rb, of + 8);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float gm = fmaf(xf[c], sc[c], sh[c]) > 0.f ? of[c] : 0.f;
        f0[c] += gm;
        f1[c] = fmaf(gm, xf[c], f1[c]);
      }
    }
  }
  if (STATS) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double a = wave_sum((double)f0[c]), b = wave_sum((double)f1[c]);
      if (lane == 0) { red[wave][c][0] = a; red[wave][c][1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
      st.partial[((n * st.chunks + blockIdx.x) * 16 + c) * 2 + k] = ((red[0][c][k] + red[1][c][k]) + red[2][c][k]) + red[3][c][k];
    }
  }
}

// 1: the 3-D nets' endconv (16 -> 4); 2: the 2-D nets' conv10 (32 -> 2); 0: not a head
static int head_shape(const devis_conv_desc* d) {
  if (!(d->dtype == DEVIS_BF16 && d->KD == 1 && d->KH == 1 && d->KW == 1 && d->SD == 1 && d->SH == 1 && d->SW == 1 &&
        d->PD == 0 && d->PH == 0 && d->PW == 0 && d->C2 == 0)) return 0;
  if (d->C1 == 16 && d->Cout == 4) return 1;
  if (d->C1 == 32 && d->Cout == 2) return 2;
  return 0;
}

int head_fwd(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const void* res, void* y, hipStream_t s) {
  const int hs = head_shape(d);
  if (!hs || res) return DEVIS_ERR_UNSUPPORTED;
  const int64_t NV = d->N * d->ID * d->IH * d->IW;
  if (cdiv(NV, H1_VPB) > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)cdiv(NV, H1_VPB));
  if (hs == 1) hipLaunchKernelGGL((k_head_fwd<16, 4>), grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)w, bias, (bf16*)y, NV);
  else hipLaunchKernelGGL((k_head_fwd<32, 2>), grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)w, bias, (bf16*)y, NV);
  return launch_status();
}

// plan_chunks != null: only report the statistics chunks per sample
int head_dgrad(const devis_conv_desc* d, const void* dy, const void* w, const void* res, void* dx, const StatsArgs* st,
               int64_t* plan_chunks, hipStream_t s) {
  if (plan_chunks) *plan_chunks = 0;
  const int hs = head_shape(d);
  if (!hs || res || d->N > 65535) return DEVIS_ERR_UNSUPPORTED;
  const int64_t V = d->ID * d->IH * d->IW;
  const int64_t chunks = cdiv(V, H1_VPB);
  if (chunks > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const int mode = st ? st->mode : 0;
  if (mode == 1 || (mode == 2 && hs != 1)) return DEVIS_ERR_UNSUPPORTED;
  if (plan_chunks) { *plan_chunks = chunks; return DEVIS_OK; }
  const dim3 grid((unsigned)chunks, (unsigned)d->N);
  StatsArgs a = StatsArgs{};
  if (mode == 2) {
    a = *st;
    if (a.chunks != chunks || a.C != 16) return DEVIS_ERR_BAD_ARG;
    hipLaunchKernelGGL((k_head_dgrad<16, 4, 2>), grid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)w, (bf16*)dx, V, a);
  } else if (hs == 1) {
    hipLaunchKernelGGL((k_head_dgrad<16, 4, 0>), grid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)w, (bf16*)dx, V, a);
  } else {
    hipLaunchKernelGGL((k_head_dgrad<32, 2, 0>), grid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)w, (bf16*)dx, V, a);
  }
  return launch_status();
}

}  // namespace devis
