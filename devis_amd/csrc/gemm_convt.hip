// The wide ConvTranspose2d(k=2, s=2) layers of the 2-D U-Nets (up6..up8: 512->256, 256->128, 128->64,
// UNet2DZoo.py:116-125) as plain MFMA GEMMs, bf16 channels-last, gfx950.  With hundreds of channels these layers
// are matrix-core work (4.3 GFLOP each at batch 16), which the register-only gather-GEMM of pointwise_mfma.hip
// (built for the HBM-bound 16..64-channel layers of the 3-D nets) ran at ~45 TFLOP/s.
//
//   forward:  y[fine(q, p)][co] = b[co] + sum_ci x[q][ci] * Wp[p*Cout + co][ci]        M = T*Cout, K = Cin
//   dgrad:    dx[q][ci]         = sum_{p,co} dy[fine(q, p)][co] * Wt[ci][p*Cout + co]  M = Cin,    K = T*Cout
//
// q = coarse pixel, p = tap, fine(q, p) = the output pixel the tap lands on.  A block computes 64 M rows of 256
// consecutive coarse pixels; per stage it puts 128 K channels of both operands in LDS (4 swizzled 32-channel
// images each, bulk coalesced loads, next stage prefetched into registers) and every wave owns 4 M tiles x 4
// pixel tiles: 8 LDS reads per 16 MFMAs.  The forward scatters its rows to the fine grid in the epilogue, the
// dgrad gathers its K chunks from the four fine pixels while staging.
#include "common.h"
#include "mfma_common.h"

namespace devis {

struct GtArgs {
  const bf16* b;      // forward: x [NV][K];  dgrad: dy [NV*T][Cout]
  const bf16* a;      // [M][K]
  const float* bias;  // forward: [Cout] or null
  bf16* y;
  int NV, H, W, K, M, Cout, KH, KW;
};

template <int MODE>     // 0 forward, 1 dgrad
__global__ void __launch_bounds__(256)
k_gemm_convt(GtArgs a) {
  constexpr int KC = 4, MT = 4, NT = 4;
  constexpr int XB = KC * 256 * 64;          // 64 KB: KC images of [256 px][64 B]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* ws = smem + XB;                      // KC images of [64 rows][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int piece = tid & 3;
  const int px0 = blockIdx.x * 256, m0 = blockIdx.y * 64;
  const int OW = a.W * a.KW;

  // this thread's 4 pixels (slots k & 3) and, for the dgrad, their fine base pixel
  unsigned pb[4];
  bool pok[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int gp = px0 + (tid >> 2) + 64 * u;
    pok[u] = gp < a.NV;
    const int g = pok[u] ? gp : 0;
    if (MODE == 0) {
      pb[u] = (unsigned)g;
    } else {
      const int w = g % a.W, h = (g / a.W) % a.H, n = g / (a.W * a.H);
      pb[u] = (unsigned)((n * a.H * a.KH + h * a.KH) * OW + w * a.KW);
    }
  }
  auto fetch = [&](int st, u32x4 (&xr)[16], u32x4 (&wr)[KC]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int kc = k >> 2, u = k & 3;
      const int kk = st * 128 + kc * 32;
      if (MODE == 0) {
        xr[k] = *reinterpret_cast<const u32x4*>(a.b + pb[u] * (unsigned)a.K + kk + piece * 8);
      } else {
        const int p = kk / a.Cout, c0 = kk % a.Cout;
        const unsigned fine = pb[u] + (unsigned)((p / a.KW) * OW + p % a.KW);
        xr[k] = *reinterpret_cast<const u32x4*>(a.b + fine * (unsigned)a.Cout + c0 + piece * 8);
      }
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      wr[kc] = *reinterpret_cast<const u32x4*>(a.a + (unsigned)(m0 + (tid >> 2)) * (unsigned)a.K + st * 128 + kc * 32 + piece * 8);
  };
  auto stage = [&](u32x4 (&xr)[16], u32x4 (&wr)[KC]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int kc = k >> 2, u = k & 3;
      const int pxl = (tid >> 2) + 64 * u;
      u32x4 v = xr[k];
      if (!pok[u]) v = (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(xs + kc * 16384 + pxl * 64 + ((piece ^ ((pxl >> 1) & 3)) << 4)) = v;
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int row = tid >> 2;
      *reinterpret_cast<u32x4*>(ws + kc * 4096 + row * 64 + ((piece ^ ((row >> 1) & 3)) << 4)) = wr[kc];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nst = a.K >> 7;
  u32x4 xr[16], wr[KC];
  fetch(0, xr, wr);
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    __syncthreads();
    stage(xr, wr);
    __syncthreads();
    if (st + 1 < nst) fetch(st + 1, xr, wr);
    __builtin_amdgcn_sched_barrier(0);
    auto read_kc = [&](int kc, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + l15;
        af[m] = *reinterpret_cast<const bf16x8*>(ws + kc * 4096 + row * 64 + ((lq ^ ((row >> 1) & 3)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int u = (wave * NT + j) * 16 + l15;
        bf[j] = *reinterpret_cast<const bf16x8*>(xs + kc * 16384 + u * 64 + ((lq ^ ((u >> 1) & 3)) << 4));
      }
    };
    bf16x8 afA[MT], bfA[NT], afB[MT], bfB[NT];
    read_kc(0, afA, bfA);
#pragma unroll
    for (int kc = 0; kc < KC; kc += 2) {
      read_kc(kc + 1, afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afA[m], bfA[j], acc[m][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kc + 2 < KC) read_kc(kc + 2, afA, bfA);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afB[m], bfB[j], acc[m][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: lane = 4 consecutive M rows (lq*4..) of pixel l15 of tile j
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int gp = px0 + (wave * NT + j) * 16 + l15;
    if (gp >= a.NV) continue;
    unsigned fb = 0;
    if (MODE == 0) {
      const int w = gp % a.W, h = (gp / a.W) % a.H, n = gp / (a.W * a.H);
      fb = (unsigned)((n * a.H * a.KH + h * a.KH) * OW + w * a.KW);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mrow = m0 + m * 16 + lq * 4;
      float v0 = acc[m][j][0], v1 = acc[m][j][1], v2 = acc[m][j][2], v3 = acc[m][j][3];
      bf16* dst;
      if (MODE == 0) {
        const int p = mrow / a.Cout, co = mrow % a.Cout;
        if (a.bias) {
          const float4 bb = *reinterpret_cast<const float4*>(a.bias + co);
          v0 += bb.x; v1 += bb.y; v2 += bb.z; v3 += bb.w;
        }
        dst = a.y + (size_t)(fb + (unsigned)((p / a.KW) * OW + p % a.KW)) * a.Cout + co;
      } else {
        dst = a.y + (size_t)gp * a.M + mrow;
      }
      uint2 o;
      o.x = pack_bf2(v0, v1);
      o.y = pack_bf2(v2, v3);
      *reinterpret_cast<uint2*>(dst) = o;
    }
  }
}

template <int MODE>
static int launch_gemm_convt(const GtArgs& a, hipStream_t s) {
  constexpr int LDS = 4 * 256 * 64 + 4 * 64 * 64;      // 80 KB
  auto k = k_gemm_convt<MODE>;
  static bool attr_done = false;
  if (!attr_done) { if (check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS))) return DEVIS_ERR_HIP; attr_done = true; }
  const dim3 grid((unsigned)cdiv(a.NV, 256), (unsigned)(a.M / 64));
  hipLaunchKernelGGL(k, grid, dim3(256), LDS, s, a);
  return launch_status();
}

static bool gemm_convt_shape(const devis_convT_desc* d) {
  if (d->dtype != DEVIS_BF16 || d->KD != 1 || d->ID != 1) return false;
  const int64_t T = (int64_t)d->KH * d->KW, NV = d->N * d->IH * d->IW;
  return NV * T * (d->Cout > d->Cin ? d->Cout : d->Cin) < ((int64_t)1 << 31) && NV >= 1024;
}

// forward: Cin % 128 == 0 (K stages), T*Cout % 64 == 0, Cout % 4 == 0;  w = packed [T][Cout][Cin]
int convT_fwd_gemm(const devis_convT_desc* d, const void* x, const void* w, const float* bias, void* y, hipStream_t s) {
  if (!gemm_convt_shape(d) || d->Cin % 128 || d->Cout % 4 || (d->KH * d->KW * d->Cout) % 64) return DEVIS_ERR_UNSUPPORTED;
  GtArgs a;
  a.b = (const bf16*)x; a.a = (const bf16*)w; a.bias = bias; a.y = (bf16*)y;
  a.NV = (int)(d->N * d->IH * d->IW); a.H = (int)d->IH; a.W = (int)d->IW;
  a.K = (int)d->Cin; a.Cout = (int)d->Cout; a.KH = d->KH; a.KW = d->KW; a.M = a.KH * a.KW * a.Cout;
  return launch_gemm_convt<0>(a, s);
}

// dgrad: T*Cout % 128 == 0, Cout % 32 == 0 (a 32-channel K chunk stays inside one tap), Cin % 64 == 0;  w_t = [Cin][T*Cout]
int convT_dgrad_gemm(const devis_convT_desc* d, const void* dy, const void* w_t, void* dx, hipStream_t s) {
  if (!w_t || !gemm_convt_shape(d) || d->Cout % 32 || (d->KH * d->KW * d->Cout) % 128 || d->Cin % 64) return DEVIS_ERR_UNSUPPORTED;
  GtArgs a;
  a.b = (const bf16*)dy; a.a = (const bf16*)w_t; a.bias = nullptr; a.y = (bf16*)dx;
  a.NV = (int)(d->N * d->IH * d->IW); a.H = (int)d->IH; a.W = (int)d->IW;
  a.Cout = (int)d->Cout; a.KH = d->KH; a.KW = d->KW; a.K = a.KH * a.KW * a.Cout; a.M = (int)d->Cin;
  return launch_gemm_convt<1>(a, s);
}

}  // namespace devis
