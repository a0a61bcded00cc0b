// The 2-D U-Nets' first layer (UNet2DZoo.py:111,188: DoubleConv(in_ch, 32), in_ch = 1 or 3): a 3x3 conv
// whose input has too few channels for a matrix-core K step.  It is 18*in_ch*32 FLOP against 64 output bytes
// per pixel, i.e. a streaming problem, so it runs on the vector ALUs with coalesced 64-byte rows:
//   forward: one thread per pixel, the [9*CIN][32] fp32 weights broadcast from LDS, 32 accumulators;
//   wgrad:   a block is 8 pixel lanes x 32 cout; a thread keeps its cout's 9*CIN sums (+ the bias sum) over the
//            pixels of its lane, lanes combine through LDS in a fixed order, one partial per block and the
//            fixed-order reduction of wgrad_reduce.h (deterministic, no atomics).
// bf16 channels-last, stride 1, padding 1, no prologue (the network input is not normalised).
#include "common.h"
#include "mfma_common.h"
#include "wgrad_reduce.h"

namespace devis {

#define THIN_CO 32

template <int CIN>
__global__ void __launch_bounds__(256)
k_first2d_fwd(const bf16* __restrict__ x, const bf16* __restrict__ w, const float* __restrict__ bias,
              bf16* __restrict__ y, int N, int H, int W) {
  __shared__ __attribute__((aligned(16))) float wl[9 * CIN][THIN_CO];
  __shared__ __attribute__((aligned(16))) float bl[THIN_CO];
  for (int i = threadIdx.x; i < 9 * CIN * THIN_CO; i += 256) {
    const int co = i % THIN_CO, k = i / THIN_CO;            // k = tap * CIN + ci; w is [co][tap][ci]
    wl[k][co] = (float)w[co * 9 * CIN + k];
  }
  if (threadIdx.x < THIN_CO) bl[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int64_t NV = (int64_t)N * H * W;
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= NV) return;
  const int pw = (int)(v % W), ph = (int)((v / W) % H);
  float xin[9 * CIN];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ih = ph + t / 3 - 1, iw = pw + t % 3 - 1;
    const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
    const int64_t iv = v + (int64_t)(t / 3 - 1) * W + (t % 3 - 1);
#pragma unroll
    for (int c = 0; c < CIN; ++c) xin[t * CIN + c] = ok ? (float)x[iv * CIN + c] : 0.f;
  }
  float acc[THIN_CO];
#pragma unroll
  for (int co = 0; co < THIN_CO; ++co) acc[co] = bl[co];
#pragma unroll
  for (int k = 0; k < 9 * CIN; ++k) {
#pragma unroll
    for (int c4 = 0; c4 < THIN_CO / 4; ++c4) {
      const float4 wv = *reinterpret_cast<const float4*>(&wl[k][c4 * 4]);     // same address in every lane
      acc[c4 * 4 + 0] = fmaf(wv.x, xin[k], acc[c4 * 4 + 0]);
      acc[c4 * 4 + 1] = fmaf(wv.y, xin[k], acc[c4 * 4 + 1]);
      acc[c4 * 4 + 2] = fmaf(wv.z, xin[k], acc[c4 * 4 + 2]);
      acc[c4 * 4 + 3] = fmaf(wv.w, xin[k], acc[c4 * 4 + 3]);
    }
  }
  bf16* yo = y + v * THIN_CO;
#pragma unroll
  for (int q = 0; q < THIN_CO / 8; ++q) {
    uint4 o;
    o.x = pack_bf2(acc[q * 8 + 0], acc[q * 8 + 1]); o.y = pack_bf2(acc[q * 8 + 2], acc[q * 8 + 3]);
    o.z = pack_bf2(acc[q * 8 + 4], acc[q * 8 + 5]); o.w = pack_bf2(acc[q * 8 + 6], acc[q * 8 + 7]);
    *reinterpret_cast<uint4*>(yo + q * 8) = o;
  }
}

// part[block][tap][co][ci], bpart[block][co]
template <int CIN>
__global__ void __launch_bounds__(256)
k_first2d_wgrad(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ part,
                float* __restrict__ bpart, int N, int H, int W) {
  constexpr int NA = 9 * CIN;
  __shared__ float red[8][THIN_CO][NA + 1];
  const int co = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int64_t NV = (int64_t)N * H * W;
  float acc[NA], bs = 0.f;
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * 8 + pl; v < NV; v += (int64_t)gridDim.x * 8) {
    const float g = (float)dy[v * THIN_CO + co];
    const int pw = (int)(v % W), ph = (int)((v / W) % H);
    bs += g;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ih = ph + t / 3 - 1, iw = pw + t % 3 - 1;
      const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
      const int64_t iv = v + (int64_t)(t / 3 - 1) * W + (t % 3 - 1);
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const float a = ok ? (float)x[iv * CIN + c] : 0.f;
        acc[t * CIN + c] = fmaf(g, a, acc[t * CIN + c]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NA; ++k) red[pl][co][k] = acc[k];
  red[pl][co][NA] = bs;
  __syncthreads();
  // fixed order over the 8 pixel lanes
  for (int i = threadIdx.x; i < THIN_CO * (NA + 1); i += 256) {
    const int c = i / (NA + 1), k = i % (NA + 1);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][c][k];
    if (k == NA) { if (bpart) bpart[(size_t)blockIdx.x * THIN_CO + c] = s; }
    else part[(size_t)blockIdx.x * (NA * THIN_CO) + ((k / CIN) * THIN_CO + c) * CIN + k % CIN] = s;
  }
}

static bool first2d_shape(const devis_conv_desc* d) {
  return d->dtype == DEVIS_BF16 && d->KD == 1 && d->KH == 3 && d->KW == 3 && d->SD == 1 && d->SH == 1 && d->SW == 1 &&
         d->PD == 0 && d->PH == 1 && d->PW == 1 && d->ID == 1 && d->C2 == 0 && (d->C1 == 1 || d->C1 == 3) &&
         d->Cout == THIN_CO;
}

int first2d_fwd(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const void* res, void* y,
                hipStream_t s) {
  if (!first2d_shape(d) || res) return DEVIS_ERR_UNSUPPORTED;
  const int64_t NV = d->N * d->IH * d->IW;
  if (NV >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)cdiv(NV, 256));
  if (d->C1 == 1) hipLaunchKernelGGL(k_first2d_fwd<1>, grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)w, bias, (bf16*)y, (int)d->N, (int)d->IH, (int)d->IW);
  else hipLaunchKernelGGL(k_first2d_fwd<3>, grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)w, bias, (bf16*)y, (int)d->N, (int)d->IH, (int)d->IW);
  return launch_status();
}

static int64_t first2d_blocks(const devis_conv_desc* d) {
  const int64_t NV = d->N * d->IH * d->IW;
  const int64_t g = cdiv(NV, 8 * 16);          // >= 16 pixels per thread
  return g < 1 ? 1 : (g > 4096 ? 4096 : g);
}

size_t first2d_wgrad_workspace(const devis_conv_desc* d) {
  if (!first2d_shape(d)) return 0;
  return (size_t)first2d_blocks(d) * (9 * d->C1 * THIN_CO + THIN_CO) * sizeof(float);
}

int first2d_wgrad(const devis_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                  size_t ws_bytes, hipStream_t s) {
  if (!first2d_shape(d)) return DEVIS_ERR_UNSUPPORTED;
  const int64_t NV = d->N * d->IH * d->IW;
  if (NV >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  const int64_t g = first2d_blocks(d);
  const int cin = (int)d->C1;
  if (!ws || ws_bytes < first2d_wgrad_workspace(d)) return DEVIS_ERR_WORKSPACE;
  float* part = (float*)ws;
  float* bpart = dbias ? part + (size_t)g * 9 * cin * THIN_CO : nullptr;
  if (cin == 1) hipLaunchKernelGGL(k_first2d_wgrad<1>, dim3((unsigned)g), dim3(256), 0, s, (const bf16*)x, (const bf16*)dy, part, bpart, (int)d->N, (int)d->IH, (int)d->IW);
  else hipLaunchKernelGGL(k_first2d_wgrad<3>, dim3((unsigned)g), dim3(256), 0, s, (const bf16*)x, (const bf16*)dy, part, bpart, (int)d->N, (int)d->IH, (int)d->IW);
  int rc = launch_status();
  if (rc) return rc;
  RedArgs r;
  r.part = part; r.bpart = bpart; r.dw = dw; r.dbias = dbias;
  r.nparts = (int)g; r.nbparts = (int)g; r.Cb = THIN_CO;
  r.pstride = (int64_t)9 * THIN_CO * cin; r.total = r.pstride;
  r.mode = RED_CONV3; r.A = THIN_CO; r.B = cin; r.cin_real = cin;
  return launch_reduce_partials(r, s);
}

}  // namespace devis
