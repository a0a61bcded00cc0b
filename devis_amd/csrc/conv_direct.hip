// Direct (VALU) convolution kernels, any shape, fp32 or bf16 storage, fp32 accumulate.
// These are the exact-arithmetic path (fp32 parity runs) and the fallback for shapes the
// MFMA implicit-GEMM kernels (conv_mfma.hip) do not take.  Channels-last everywhere.
#include "common.h"

namespace devis {

struct ConvP {
  int64_t N, ID, IH, IW, OD, OH, OW, C1, C2, Cout;
  int KD, KH, KW, SD, SH, SW, PD, PH, PW;
};

static inline ConvP to_p(const devis_conv_desc* d) {
  ConvP p;
  p.N = d->N; p.ID = d->ID; p.IH = d->IH; p.IW = d->IW; p.OD = d->OD; p.OH = d->OH; p.OW = d->OW;
  p.C1 = d->C1; p.C2 = d->C2; p.Cout = d->Cout;
  p.KD = d->KD; p.KH = d->KH; p.KW = d->KW; p.SD = d->SD; p.SH = d->SH; p.SW = d->SW;
  p.PD = d->PD; p.PH = d->PH; p.PW = d->PW;
  return p;
}

static inline bool desc_ok(const devis_conv_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->ID <= 0 || d->IH <= 0 || d->IW <= 0 || d->C1 <= 0 || d->C2 < 0 || d->Cout <= 0) return false;
  if (d->KD <= 0 || d->KH <= 0 || d->KW <= 0 || d->SD <= 0 || d->SH <= 0 || d->SW <= 0) return false;
  if (d->PD < 0 || d->PH < 0 || d->PW < 0) return false;
  // output size must be what the reference's conv produces
  if (d->OD != (d->ID + 2 * d->PD - d->KD) / d->SD + 1) return false;
  if (d->OH != (d->IH + 2 * d->PH - d->KH) / d->SH + 1) return false;
  if (d->OW != (d->IW + 2 * d->PW - d->KW) / d->SW + 1) return false;
  if (d->OD <= 0 || d->OH <= 0 || d->OW <= 0) return false;
  return true;
}

// ------------------------------------------------------------------ forward
template <typename T, bool PRO>
__global__ void __launch_bounds__(256)
k_conv_fwd(ConvP p, const T* __restrict__ x1, const T* __restrict__ x2, const T* __restrict__ w,
           const float* __restrict__ bias, const float* __restrict__ sc, const float* __restrict__ sh,
           const T* __restrict__ res, T* __restrict__ y) {
  const int64_t total = p.N * p.OD * p.OH * p.OW * p.Cout;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t co = idx % p.Cout;
  int64_t ov = idx / p.Cout;
  const int64_t ow = ov % p.OW; ov /= p.OW;
  const int64_t oh = ov % p.OH; ov /= p.OH;
  const int64_t od = ov % p.OD;
  const int64_t n = ov / p.OD;
  const int64_t Cin = p.C1 + p.C2;
  const int Tn = p.KD * p.KH * p.KW;
  const float* scn = PRO ? sc + n * Cin : nullptr;
  const float* shn = PRO ? sh + n * Cin : nullptr;
  float acc = bias ? bias[co] : 0.f;
  for (int kd = 0; kd < p.KD; ++kd) {
    const int64_t id = od * p.SD + kd - p.PD;
    if (id < 0 || id >= p.ID) continue;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int64_t ih = oh * p.SH + kh - p.PH;
      if (ih < 0 || ih >= p.IH) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int64_t iw = ow * p.SW + kw - p.PW;
        if (iw < 0 || iw >= p.IW) continue;
        const int t = (kd * p.KH + kh) * p.KW + kw;
        const int64_t iv = ((n * p.ID + id) * p.IH + ih) * p.IW + iw;
        const T* wp = w + (co * Tn + t) * Cin;
        const T* xp = x1 + iv * p.C1;
        for (int64_t ci = 0; ci < p.C1; ++ci) {
          float v = ld(xp + ci);
          if (PRO) v = fmaxf(fmaf(v, scn[ci], shn[ci]), 0.f);
          acc = fmaf(v, ld(wp + ci), acc);
        }
        if (p.C2 > 0) {
          const T* xq = x2 + iv * p.C2;
          for (int64_t ci = 0; ci < p.C2; ++ci) {
            float v = ld(xq + ci);
            if (PRO) v = fmaxf(fmaf(v, scn[p.C1 + ci], shn[p.C1 + ci]), 0.f);
            acc = fmaf(v, ld(wp + p.C1 + ci), acc);
          }
        }
      }
    }
  }
  if (res) acc += ld(res + idx);
  st(y + idx, acc);
}

// -------------------------------------------------------------------- dgrad
// dA[n, iv, ci] = sum_t sum_co dy[n, ov(iv,t), co] * w[co][t][ci]
template <typename T>
__global__ void __launch_bounds__(256)
k_conv_dgrad(ConvP p, const T* __restrict__ dy, const T* __restrict__ w, const T* __restrict__ res, T* __restrict__ dx1, T* __restrict__ dx2) {
  const int64_t Cin = p.C1 + p.C2;
  const int64_t total = p.N * p.ID * p.IH * p.IW * Cin;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t ci = idx % Cin;
  int64_t iv = idx / Cin;
  const int64_t ivf = iv;
  const int64_t iw = iv % p.IW; iv /= p.IW;
  const int64_t ih = iv % p.IH; iv /= p.IH;
  const int64_t id = iv % p.ID;
  const int64_t n = iv / p.ID;
  const int Tn = p.KD * p.KH * p.KW;
  float acc = 0.f;
  for (int kd = 0; kd < p.KD; ++kd) {
    const int64_t nd = id + p.PD - kd;
    if (nd < 0 || nd % p.SD) continue;
    const int64_t od = nd / p.SD;
    if (od >= p.OD) continue;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int64_t nh = ih + p.PH - kh;
      if (nh < 0 || nh % p.SH) continue;
      const int64_t oh = nh / p.SH;
      if (oh >= p.OH) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int64_t nw = iw + p.PW - kw;
        if (nw < 0 || nw % p.SW) continue;
        const int64_t ow = nw / p.SW;
        if (ow >= p.OW) continue;
        const int t = (kd * p.KH + kh) * p.KW + kw;
        const T* g = dy + (((n * p.OD + od) * p.OH + oh) * p.OW + ow) * p.Cout;
        const T* wp = w + (int64_t)t * Cin + ci;
        for (int64_t co = 0; co < p.Cout; ++co) acc = fmaf(ld(g + co), ld(wp + co * Tn * Cin), acc);
      }
    }
  }
  if (res) acc += ld(res + ivf * p.C1 + ci);   // host: residual only when C2 == 0
  if (ci < p.C1) st(dx1 + ivf * p.C1 + ci, acc);
  else st(dx2 + ivf * p.C2 + (ci - p.C1), acc);
}

// -------------------------------------------------------------------- wgrad
// block = 16x16 (co, ci) pairs; grid.x = voxel chunk, grid.y = tap, grid.z = (co tile, ci tile)
#define WG_CHUNK 2048
template <typename T, bool PRO>
__global__ void __launch_bounds__(256)
k_conv_wgrad(ConvP p, const T* __restrict__ x1, const T* __restrict__ x2, const float* __restrict__ sc,
             const float* __restrict__ sh, const T* __restrict__ dy, float* __restrict__ dw) {
  const int64_t Cin = p.C1 + p.C2;
  const int Tn = p.KD * p.KH * p.KW;
  const int t = blockIdx.y;
  const int kw = t % p.KW, kh = (t / p.KW) % p.KH, kd = t / (p.KW * p.KH);
  const int64_t ci_tiles = cdiv(Cin, 16);
  const int64_t co = (blockIdx.z / ci_tiles) * 16 + (threadIdx.x >> 4);
  const int64_t ci = (blockIdx.z % ci_tiles) * 16 + (threadIdx.x & 15);
  const bool live = (co < p.Cout) && (ci < Cin);
  const int64_t OV = p.N * p.OD * p.OH * p.OW;
  const int64_t v0 = (int64_t)blockIdx.x * WG_CHUNK;
  const int64_t v1 = v0 + WG_CHUNK < OV ? v0 + WG_CHUNK : OV;
  const bool second = ci >= p.C1;
  float acc = 0.f;
  if (live) {
    for (int64_t v = v0; v < v1; ++v) {
      int64_t r = v;
      const int64_t ow = r % p.OW; r /= p.OW;
      const int64_t oh = r % p.OH; r /= p.OH;
      const int64_t od = r % p.OD;
      const int64_t n = r / p.OD;
      const int64_t id = od * p.SD + kd - p.PD, ih = oh * p.SH + kh - p.PH, iw = ow * p.SW + kw - p.PW;
      if (id < 0 || id >= p.ID || ih < 0 || ih >= p.IH || iw < 0 || iw >= p.IW) continue;
      const int64_t iv = ((n * p.ID + id) * p.IH + ih) * p.IW + iw;
      float a = second ? ld(x2 + iv * p.C2 + (ci - p.C1)) : ld(x1 + iv * p.C1 + ci);
      if (PRO) a = fmaxf(fmaf(a, sc[n * Cin + ci], sh[n * Cin + ci]), 0.f);
      acc = fmaf(ld(dy + v * p.Cout + co), a, acc);
    }
    atomicAdd(dw + (co * Cin + ci) * Tn + t, acc);
  }
}

// out[c] += sum over rows of X[row][c]; always 256 threads
template <typename T>
__global__ void __launch_bounds__(256)
k_colsum(const T* __restrict__ x, int64_t rows, int64_t C, float* __restrict__ out) {
  const int64_t r0 = (int64_t)blockIdx.x * WG_CHUNK;
  const int64_t r1 = r0 + WG_CHUNK < rows ? r0 + WG_CHUNK : rows;
  if (C <= 256) {
    const int64_t lanes_per_c = 256 / C;
    const int64_t c = threadIdx.x % C, sub = threadIdx.x / C;
    if (sub >= lanes_per_c) return;
    float acc = 0.f;
    for (int64_t r = r0 + sub; r < r1; r += lanes_per_c) acc += ld(x + r * C + c);
    atomicAdd(out + c, acc);
  } else {
    for (int64_t c = threadIdx.x; c < C; c += 256) {
      float acc = 0.f;
      for (int64_t r = r0; r < r1; ++r) acc += ld(x + r * C + c);
      atomicAdd(out + c, acc);
    }
  }
}

// ------------------------------------------------------------ ConvTranspose
struct ConvTP {
  int64_t N, ID, IH, IW, Cin, Cout;
  int KD, KH, KW;
};

template <typename T>
__global__ void __launch_bounds__(256)
k_convT_fwd(ConvTP p, const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y) {
  const int64_t OD = p.ID * p.KD, OH = p.IH * p.KH, OW = p.IW * p.KW;
  const int64_t total = p.N * OD * OH * OW * p.Cout;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t co = idx % p.Cout;
  int64_t ov = idx / p.Cout;
  const int64_t ow = ov % OW; ov /= OW;
  const int64_t oh = ov % OH; ov /= OH;
  const int64_t od = ov % OD;
  const int64_t n = ov / OD;
  const int t = (int)(((od % p.KD) * p.KH + (oh % p.KH)) * p.KW + (ow % p.KW));
  const int64_t iv = ((n * p.ID + od / p.KD) * p.IH + oh / p.KH) * p.IW + ow / p.KW;
  const T* xp = x + iv * p.Cin;
  const T* wp = w + ((int64_t)t * p.Cout + co) * p.Cin;
  float acc = bias ? bias[co] : 0.f;
  for (int64_t ci = 0; ci < p.Cin; ++ci) acc = fmaf(ld(xp + ci), ld(wp + ci), acc);
  st(y + idx, acc);
}

template <typename T>
__global__ void __launch_bounds__(256)
k_convT_dgrad(ConvTP p, const T* __restrict__ dy, const T* __restrict__ w, T* __restrict__ dx) {
  const int64_t OH = p.IH * p.KH, OW = p.IW * p.KW;
  const int64_t OD = p.ID * p.KD;
  const int64_t total = p.N * p.ID * p.IH * p.IW * p.Cin;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t ci = idx % p.Cin;
  int64_t iv = idx / p.Cin;
  const int64_t iw = iv % p.IW; iv /= p.IW;
  const int64_t ih = iv % p.IH; iv /= p.IH;
  const int64_t id = iv % p.ID;
  const int64_t n = iv / p.ID;
  float acc = 0.f;
  for (int kd = 0; kd < p.KD; ++kd)
    for (int kh = 0; kh < p.KH; ++kh)
      for (int kw = 0; kw < p.KW; ++kw) {
        const int t = (kd * p.KH + kh) * p.KW + kw;
        const int64_t ov = ((n * OD + id * p.KD + kd) * OH + ih * p.KH + kh) * OW + iw * p.KW + kw;
        const T* g = dy + ov * p.Cout;
        const T* wp = w + (int64_t)t * p.Cout * p.Cin + ci;
        for (int64_t co = 0; co < p.Cout; ++co) acc = fmaf(ld(g + co), ld(wp + co * p.Cin), acc);
      }
  st(dx + idx, acc);
}

// grid.x = input-voxel chunk, grid.y = tap, grid.z = (ci tile, co tile); dw[ci][co][t]
template <typename T>
__global__ void __launch_bounds__(256)
k_convT_wgrad(ConvTP p, const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw) {
  const int Tn = p.KD * p.KH * p.KW;
  const int t = blockIdx.y;
  const int kw = t % p.KW, kh = (t / p.KW) % p.KH, kd = t / (p.KW * p.KH);
  const int64_t co_tiles = cdiv(p.Cout, 16);
  const int64_t ci = (blockIdx.z / co_tiles) * 16 + (threadIdx.x >> 4);
  const int64_t co = (blockIdx.z % co_tiles) * 16 + (threadIdx.x & 15);
  if (ci >= p.Cin || co >= p.Cout) return;
  const int64_t OD = p.ID * p.KD, OH = p.IH * p.KH, OW = p.IW * p.KW;
  const int64_t IV = p.N * p.ID * p.IH * p.IW;
  const int64_t v0 = (int64_t)blockIdx.x * WG_CHUNK;
  const int64_t v1 = v0 + WG_CHUNK < IV ? v0 + WG_CHUNK : IV;
  float acc = 0.f;
  for (int64_t v = v0; v < v1; ++v) {
    int64_t r = v;
    const int64_t iw = r % p.IW; r /= p.IW;
    const int64_t ih = r % p.IH; r /= p.IH;
    const int64_t id = r % p.ID;
    const int64_t n = r / p.ID;
    const int64_t ov = ((n * OD + id * p.KD + kd) * OH + ih * p.KH + kh) * OW + iw * p.KW + kw;
    acc = fmaf(ld(x + v * p.Cin + ci), ld(dy + ov * p.Cout + co), acc);
  }
  atomicAdd(dw + (ci * p.Cout + co) * Tn + t, acc);
}

int convT_fwd_mfma(const devis_convT_desc*, const void*, const void*, const float*, void*, hipStream_t);
int convT_dgrad_mfma(const devis_convT_desc*, const void*, const void*, void*, hipStream_t);

size_t convT_wgrad_mfma_workspace(const devis_convT_desc*);
int convT_wgrad_mfma(const devis_convT_desc*, const void*, const void*, float*, float*, void*, size_t, hipStream_t);

static inline bool descT_ok(const devis_convT_desc* d) {
  if (!d || d->N <= 0 || d->ID <= 0 || d->IH <= 0 || d->IW <= 0 || d->Cin <= 0 || d->Cout <= 0) return false;
  if (d->KD < 1 || d->KD > 2 || d->KH < 1 || d->KH > 2 || d->KW < 1 || d->KW > 2) return false;
  return true;
}
static inline ConvTP to_tp(const devis_convT_desc* d) {
  ConvTP p;
  p.N = d->N; p.ID = d->ID; p.IH = d->IH; p.IW = d->IW; p.Cin = d->Cin; p.Cout = d->Cout;
  p.KD = d->KD; p.KH = d->KH; p.KW = d->KW;
  return p;
}

template <typename T>
static int colsum_launch(const T* x, int64_t rows, int64_t C, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_colsum<T>, dim3((unsigned)cdiv(rows, WG_CHUNK)), dim3(256), 0, s, x, rows, C, out);
  return launch_status();
}

// Entry points of the direct path (called from conv_dispatch in capi.hip)
int conv_fwd_direct(const devis_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                    const float* sc, const float* sh, const void* res, void* y, hipStream_t s) {
  const ConvP p = to_p(d);
  const int64_t total = p.N * p.OD * p.OH * p.OW * p.Cout;
  const dim3 grid((unsigned)cdiv(total, 256));
  DEVIS_DISPATCH_DTYPE(d->dtype, {
    if (sc) hipLaunchKernelGGL((k_conv_fwd<T, true>), grid, dim3(256), 0, s, p, (const T*)x1, (const T*)x2, (const T*)w, bias, sc, sh, (const T*)res, (T*)y);
    else hipLaunchKernelGGL((k_conv_fwd<T, false>), grid, dim3(256), 0, s, p, (const T*)x1, (const T*)x2, (const T*)w, bias, sc, sh, (const T*)res, (T*)y);
  });
  return launch_status();
}

int conv_dgrad_direct(const devis_conv_desc* d, const void* dy, const void* w, const void* res, void* dx1, void* dx2, hipStream_t s) {
  const ConvP p = to_p(d);
  const int64_t total = p.N * p.ID * p.IH * p.IW * (p.C1 + p.C2);
  DEVIS_DISPATCH_DTYPE(d->dtype, hipLaunchKernelGGL(k_conv_dgrad<T>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, p, (const T*)dy, (const T*)w, (const T*)res, (T*)dx1, (T*)dx2));
  return launch_status();
}

int conv_wgrad_direct(const devis_conv_desc* d, const void* x1, const void* x2, const float* sc, const float* sh,
                      const void* dy, float* dw, float* dbias, hipStream_t s) {
  const ConvP p = to_p(d);
  const int64_t Cin = p.C1 + p.C2;
  const int Tn = p.KD * p.KH * p.KW;
  const int64_t OV = p.N * p.OD * p.OH * p.OW;
  int rc = check_hip(hipMemsetAsync(dw, 0, sizeof(float) * p.Cout * Cin * Tn, s));
  if (rc) return rc;
  const dim3 grid((unsigned)cdiv(OV, WG_CHUNK), (unsigned)Tn, (unsigned)(cdiv(p.Cout, 16) * cdiv(Cin, 16)));
  DEVIS_DISPATCH_DTYPE(d->dtype, {
    if (sc) hipLaunchKernelGGL((k_conv_wgrad<T, true>), grid, dim3(256), 0, s, p, (const T*)x1, (const T*)x2, sc, sh, (const T*)dy, dw);
    else hipLaunchKernelGGL((k_conv_wgrad<T, false>), grid, dim3(256), 0, s, p, (const T*)x1, (const T*)x2, sc, sh, (const T*)dy, dw);
  });
  rc = launch_status();
  if (rc || !dbias) return rc;
  rc = check_hip(hipMemsetAsync(dbias, 0, sizeof(float) * p.Cout, s));
  if (rc) return rc;
  DEVIS_DISPATCH_DTYPE(d->dtype, rc = colsum_launch<T>((const T*)dy, OV, p.Cout, dbias, s));
  return rc;
}

}  // namespace devis

namespace devis {
// conv_f32_mfma.hip: the fp32 parity mode on the matrix cores
int convT_fwd_f32_mfma(const devis_convT_desc*, const void*, const void*, const float*, void*, hipStream_t);
int convT_dgrad_f32_mfma(const devis_convT_desc*, const void*, const void*, void*, hipStream_t);
size_t convT_wgrad_f32_workspace(const devis_convT_desc*);
int convT_wgrad_f32_mfma(const devis_convT_desc*, const void*, const void*, float*, float*, bool*, void*, size_t, hipStream_t);
}  // namespace devis

using namespace devis;

extern "C" {

int devis_convT_fwd(const devis_convT_desc* d, const void* x, const void* w, const float* bias, void* y, void* stream) {
  if (!descT_ok(d) || !x || !w || !y) return DEVIS_ERR_BAD_ARG;
  {
    int rc = convT_fwd_mfma(d, x, w, bias, y, (hipStream_t)stream);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
    rc = convT_fwd_f32_mfma(d, x, w, bias, y, (hipStream_t)stream);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  }
  const ConvTP p = to_tp(d);
  const int64_t total = p.N * p.ID * p.KD * p.IH * p.KH * p.IW * p.KW * p.Cout;
  DEVIS_DISPATCH_DTYPE(d->dtype, hipLaunchKernelGGL(k_convT_fwd<T>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p, (const T*)x, (const T*)w, bias, (T*)y));
  return launch_status();
}

int devis_convT_dgrad(const devis_convT_desc* d, const void* dy, const void* w, const void* w_t, void* dx, void* stream) {
  if (!descT_ok(d) || !dy || !w || !dx) return DEVIS_ERR_BAD_ARG;
  {
    int rc = convT_dgrad_mfma(d, dy, w_t, dx, (hipStream_t)stream);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
    rc = convT_dgrad_f32_mfma(d, dy, w_t, dx, (hipStream_t)stream);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  }
  const ConvTP p = to_tp(d);
  const int64_t total = p.N * p.ID * p.IH * p.IW * p.Cin;
  DEVIS_DISPATCH_DTYPE(d->dtype, hipLaunchKernelGGL(k_convT_dgrad<T>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p, (const T*)dy, (const T*)w, (T*)dx));
  return launch_status();
}

size_t devis_convT_wgrad_workspace_bytes(const devis_convT_desc* d) {
  if (!descT_ok(d)) return 0;
  const size_t a = convT_wgrad_mfma_workspace(d), b = convT_wgrad_f32_workspace(d);
  return a > b ? a : b;
}

int devis_convT_wgrad(const devis_convT_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!descT_ok(d) || !x || !dy || !dw) return DEVIS_ERR_BAD_ARG;
  {
    const int rc = convT_wgrad_mfma(d, x, dy, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  }
  const ConvTP p = to_tp(d);
  hipStream_t s = (hipStream_t)stream;
  const int Tn = p.KD * p.KH * p.KW;
  const int64_t IV = p.N * p.ID * p.IH * p.IW;
  bool bias_done = false;
  int rc = convT_wgrad_f32_mfma(d, x, dy, dw, dbias, &bias_done, workspace, workspace_bytes, s);   // fp32 on the matrix cores
  if (bias_done) return rc;
  if (rc == DEVIS_ERR_UNSUPPORTED) {
    rc = check_hip(hipMemsetAsync(dw, 0, sizeof(float) * p.Cin * p.Cout * Tn, s));
    if (rc) return rc;
    const dim3 grid((unsigned)cdiv(IV, WG_CHUNK), (unsigned)Tn, (unsigned)(cdiv(p.Cin, 16) * cdiv(p.Cout, 16)));
    DEVIS_DISPATCH_DTYPE(d->dtype, hipLaunchKernelGGL(k_convT_wgrad<T>, grid, dim3(256), 0, s, p, (const T*)x, (const T*)dy, dw));
    rc = launch_status();
  }
  if (rc || !dbias) return rc;
  rc = check_hip(hipMemsetAsync(dbias, 0, sizeof(float) * p.Cout, s));
  if (rc) return rc;
  DEVIS_DISPATCH_DTYPE(d->dtype, rc = colsum_launch<T>((const T*)dy, IV * Tn, p.Cout, dbias, s));
  return rc;
}

}  // extern "C"
