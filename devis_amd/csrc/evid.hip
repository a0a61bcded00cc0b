// Fused subjective-logic evidential head + loss.
//   pass 1 (fwd):  z,y -> sum_v(L_ace + lambda*KL), I_k, P_k, R_k   (+ optional evidence / per-voxel terms)
//   pass 2 (bwd):  z,y -> dloss/dz
//   inference:     z   -> evidence, argmax mask, u = K/S, entropy map, u-threshold bit masks
// digamma / trigamma / lgamma are evaluated in registers (argument >= 1 always: alpha = softplus+1).
#include "common.h"

namespace devis {

#define EV_MAXK DEVIS_EVID_MAX_K
#define EV_BLOCKS 2048
#define EV_NACC (1 + 3 * EV_MAXK)

__device__ __forceinline__ float softplus_f(float z) { return z > 20.f ? z : log1pf(expf(z)); }
// d softplus / dz exactly as torch's softplus_backward: exp(z)/(exp(z)+1), 1 above the threshold
__device__ __forceinline__ float dsoftplus_f(float z) {
  if (z > 20.f) return 1.f;
  const float t = expf(z);
  return t / (t + 1.f);
}

// psi(x), x >= 1: recurrence up to x >= 6, then the asymptotic series (error < 2e-10 at x = 6)
__device__ __forceinline__ float digamma_f(float x) {
  float r = 0.f;
  while (x < 6.f) { r -= 1.f / x; x += 1.f; }
  const float xi = 1.f / x, x2 = xi * xi;
  const float ser = x2 * (8.3333333333e-2f - x2 * (8.3333333333e-3f - x2 * (3.9682539683e-3f - x2 * (4.1666666667e-3f - x2 * 7.5757575758e-3f))));
  return r + (logf(x) - 0.5f * xi - ser);
}
// psi'(x), x >= 1
__device__ __forceinline__ float trigamma_f(float x) {
  float r = 0.f;
  while (x < 6.f) { const float xi = 1.f / x; r = fmaf(xi, xi, r); x += 1.f; }
  const float xi = 1.f / x, x2 = xi * xi;
  // 1/x + 1/(2x^2) + 1/(6x^3) - 1/(30x^5) + 1/(42x^7) - 1/(30x^9) + 5/(66 x^11)
  const float ser = xi * (1.f + xi * 0.5f + x2 * (1.6666666667e-1f - x2 * (3.3333333333e-2f - x2 * (2.3809523810e-2f - x2 * (3.3333333333e-2f - x2 * 7.5757575758e-2f)))));
  return r + ser;
}
// ln Gamma(x), x >= 1: shift to x >= 7 with one log of the product, then Stirling
__device__ __forceinline__ float lgamma_f(float x) {
  float prod = 1.f;
  while (x < 7.f) { prod *= x; x += 1.f; }
  const float xi = 1.f / x, x2 = xi * xi;
  const float ser = xi * (8.3333333333e-2f - x2 * (2.7777777778e-3f - x2 * (7.9365079365e-4f - x2 * 5.9523809524e-4f)));
  return (x - 0.5f) * logf(x) - x + 0.91893853320467f + ser - logf(prod);
}

__device__ __forceinline__ int load_label(const void* labels, int ldt, int64_t i) {
  if (ldt == DEVIS_LABEL_I64) return (int)((const int64_t*)labels)[i];
  if (ldt == DEVIS_LABEL_U8) return (int)((const uint8_t*)labels)[i];
  return ((const int32_t*)labels)[i];
}

struct EvidShape {
  int64_t sN, sV, sK, N, V;
  int K;
};

template <typename T>
__device__ __forceinline__ void load_logits(const T* z, const EvidShape& s, int64_t n, int64_t v, float* out) {
  const T* p = z + n * s.sN + v * s.sV;
#pragma unroll
  for (int k = 0; k < EV_MAXK; ++k) out[k] = k < s.K ? ld(p + k * s.sK) : 0.f;
}

// ----------------------------------------------------------------- forward
template <typename T>
__global__ void __launch_bounds__(256)
k_evid_fwd(const T* __restrict__ z, EvidShape s, const void* __restrict__ labels, int ldt, float lambda,
           float* __restrict__ evidence_out, float* __restrict__ per_voxel_out, double* __restrict__ partial) {
  const int K = s.K;
  const int64_t NV = s.N * s.V;
  float acc[EV_NACC];
#pragma unroll
  for (int i = 0; i < EV_NACC; ++i) acc[i] = 0.f;
  const float lgK = lgamma_f((float)K);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < NV; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / s.V, v = i % s.V;
    float zz[EV_MAXK];
    load_logits<T>(z, s, n, v, zz);
    const int y = load_label(labels, ldt, i);
    float S = 0.f, e_y = 0.f, sum_lg = 0.f;
    float e[EV_MAXK];
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) {
      if (k < K) {
        e[k] = softplus_f(zz[k]);
        S += e[k] + 1.f;
        if (k == y) e_y = e[k];
        if (evidence_out) evidence_out[(n * K + k) * s.V + v] = e[k];
      } else {
        e[k] = 0.f;
      }
    }
    const float St = S - e_y;            // strength of alpha~ (label class reset to 1)
    const float psiSt = digamma_f(St);
    float kl_sum = 0.f;
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) {
      if (k < K && k != y) {
        const float a = e[k] + 1.f;
        sum_lg += lgamma_f(a);
        kl_sum = fmaf(e[k], digamma_f(a) - psiSt, kl_sum);
      }
    }
    const float l_ace = digamma_f(S) - digamma_f(e_y + 1.f);
    const float kl = lgamma_f(St) - sum_lg - lgK + kl_sum;
    const float pv = fmaf(lambda, kl, l_ace);
    if (per_voxel_out) per_voxel_out[i] = pv;
    acc[0] += pv;
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) {
      if (k < K) {
        const bool hit = (k == y);
        acc[1 + k] += hit ? e[k] : 0.f;            // I_k
        acc[1 + EV_MAXK + k] += e[k];              // P_k
        acc[1 + 2 * EV_MAXK + k] += hit ? 1.f : 0.f;  // R_k
      }
    }
  }
  __shared__ double red[4][EV_NACC];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < EV_NACC; ++i) {
    const double w = wave_sum((double)acc[i]);
    if (lane == 0) red[wave][i] = w;
  }
  __syncthreads();
  if (threadIdx.x < EV_NACC)
    partial[(int64_t)blockIdx.x * EV_NACC + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void k_evid_finalize(const double* __restrict__ partial, int blocks, int K, double NV, float* __restrict__ scal) {
  // 256 threads: accumulator i = tid % 32, block slice = tid / 32; slices combined in a fixed order
  __shared__ double tot[EV_NACC];
  __shared__ double red[8][32];
  {
    const int i = threadIdx.x & 31, sl = threadIdx.x >> 5;
    double a = 0.0;
    if (i < EV_NACC)
      for (int b = sl; b < blocks; b += 8) a += partial[(int64_t)b * EV_NACC + i];
    red[sl][i] = a;
  }
  __syncthreads();
  if (threadIdx.x < EV_NACC) {
    double a = 0.0;
    for (int sl = 0; sl < 8; ++sl) a += red[sl][threadIdx.x];
    tot[threadIdx.x] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double dice = 0.0;
    for (int k = 0; k < K; ++k) {
      const double I = tot[1 + k], P = tot[1 + EV_MAXK + k], R = tot[1 + 2 * EV_MAXK + k];
      dice += -log((2.0 * I + 1e-5) / (R + P + 1.0 + 1e-5));
      scal[4 + k] = (float)I;
      scal[4 + K + k] = (float)P;
      scal[4 + 2 * K + k] = (float)R;
    }
    dice /= (double)K;
    const double mean_pv = tot[0] / NV;
    scal[0] = (float)(mean_pv + dice);
    scal[1] = (float)dice;
    scal[2] = (float)mean_pv;
    scal[3] = (float)NV;
  }
}

// ---------------------------------------------------------------- backward
template <typename T>
__global__ void __launch_bounds__(256)
k_evid_bwd(const T* __restrict__ z, EvidShape s, const void* __restrict__ labels, int ldt, float lambda,
           const float* __restrict__ scal, const float* __restrict__ grad_loss, const float* __restrict__ grad_pv,
           const float* __restrict__ grad_dice, const float* __restrict__ grad_ev, T* __restrict__ dz) {
  const int K = s.K;
  const int64_t NV = s.N * s.V;
  // dice coefficients: dL_dice/de_k(v) = cD_k - [y=k]*cI_k
  float cI[EV_MAXK], cD[EV_MAXK];
  const float gd = grad_pv ? (grad_dice ? grad_dice[0] : 0.f) : grad_loss[0];
  const float gmean = grad_pv ? 0.f : grad_loss[0] / (float)NV;
#pragma unroll
  for (int k = 0; k < EV_MAXK; ++k) {
    if (k < K) {
      const float I = scal[4 + k], P = scal[4 + K + k], R = scal[4 + 2 * K + k];
      cI[k] = gd * 2.f / ((2.f * I + 1e-5f) * (float)K);
      cD[k] = gd / ((R + P + 1.f + 1e-5f) * (float)K);
    } else { cI[k] = 0.f; cD[k] = 0.f; }
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < NV; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / s.V, v = i % s.V;
    float zz[EV_MAXK];
    load_logits<T>(z, s, n, v, zz);
    const int y = load_label(labels, ldt, i);
    float e[EV_MAXK];
    float S = 0.f, e_y = 0.f;
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) {
      e[k] = k < K ? softplus_f(zz[k]) : 0.f;
      if (k < K) S += e[k] + 1.f;
      if (k == y) e_y = e[k];
    }
    const float St = S - e_y;
    const float tS = trigamma_f(S);
    const float tSt = trigamma_f(St) * (St - (float)K);
    const float g = grad_pv ? grad_pv[i] : gmean;
    T* out = dz + n * s.sN + v * s.sV;
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) {
      if (k < K) {
        const float a = e[k] + 1.f;
        const float ta = trigamma_f(a);
        float d = (k == y) ? (tS - ta) : fmaf(lambda, fmaf(e[k], ta, -tSt), tS);
        d = fmaf(g, d, cD[k] - ((k == y) ? cI[k] : 0.f));
        if (grad_ev) d += grad_ev[(n * K + k) * s.V + v];
        st(out + k * s.sK, d * dsoftplus_f(zz[k]));
      }
    }
  }
}

// --------------------------------------------------------------- inference
struct Thr { float t[16]; int n; };

template <typename T>
__global__ void __launch_bounds__(256)
k_evid_infer(const T* __restrict__ z, EvidShape s, float* __restrict__ evidence_out, uint8_t* __restrict__ pred,
             float* __restrict__ u, float* __restrict__ mne, uint16_t* __restrict__ umask, Thr thr) {
  const int K = s.K;
  const int64_t NV = s.N * s.V;
  const float inv_logK = 1.f / logf((float)K);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < NV; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / s.V, v = i % s.V;
    float zz[EV_MAXK];
    load_logits<T>(z, s, n, v, zz);
    float a[EV_MAXK];
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) {
      if (k < K) {
        const float e = softplus_f(zz[k]);
        if (evidence_out) evidence_out[(n * K + k) * s.V + v] = e;
        a[k] = e + 1.f;
        S += a[k];
      } else a[k] = 0.f;
    }
    // prob = alpha / S; first maximum wins (torch.max), trainUMIS.py:289-290
    float best = -1.f, ent = 0.f;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) {
      if (k < K) {
        const float p = a[k] / S;
        if (p > best) { best = p; bi = k; }
        if (k >= 1) ent -= p * logf(p) * inv_logK;   // Uentropy_our, predict.py:92-105
      }
    }
    const float uu = (float)K / S;
    if (pred) pred[i] = (uint8_t)bi;
    if (u) u[i] = uu;
    if (mne) mne[i] = ent;
    if (umask) {
      unsigned m = 0;
      for (int j = 0; j < thr.n; ++j) m |= (uu > thr.t[j]) ? (1u << j) : 0u;
      umask[i] = (uint16_t)m;
    }
  }
}

__global__ void __launch_bounds__(256)
k_softplus_bwd(const float* __restrict__ z, const float* __restrict__ g, float* __restrict__ dz, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dz[i] = g[i] * dsoftplus_f(z[i]);
}

static inline unsigned ev_grid(int64_t NV) {
  const int64_t b = cdiv(NV, 256);
  return (unsigned)(b < 1 ? 1 : (b > EV_BLOCKS ? EV_BLOCKS : b));
}

}  // namespace devis

using namespace devis;

extern "C" {

size_t devis_evid_workspace_bytes(int64_t N, int64_t V, int64_t K) {
  (void)N; (void)V; (void)K;
  return (size_t)EV_BLOCKS * EV_NACC * sizeof(double);
}

int devis_evid_loss_fwd(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype, const void* labels,
                        int label_dtype, int64_t N, int64_t V, int64_t K, float lambda, float* evidence_out,
                        float* per_voxel_out, float* scal, void* ws, size_t ws_bytes, void* stream) {
  if (!logits || !labels || !scal || !ws || N <= 0 || V <= 0 || K < 2 || K > EV_MAXK) return DEVIS_ERR_BAD_ARG;
  if (label_dtype < 0 || label_dtype > 2) return DEVIS_ERR_BAD_ARG;
  if (ws_bytes < devis_evid_workspace_bytes(N, V, K)) return DEVIS_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const EvidShape sh = {sN, sV, sK, N, V, (int)K};
  const unsigned blocks = ev_grid(N * V);
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_evid_fwd<T>, dim3(blocks), dim3(256), 0, s, (const T*)logits, sh, labels, label_dtype, lambda, evidence_out, per_voxel_out, (double*)ws));
  int rc = launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(k_evid_finalize, dim3(1), dim3(256), 0, s, (const double*)ws, (int)blocks, (int)K, (double)(N * V), scal);
  return launch_status();
}

int devis_evid_loss_bwd(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype, const void* labels,
                        int label_dtype, int64_t N, int64_t V, int64_t K, float lambda, const float* scal,
                        const float* grad_loss, const float* grad_pv, const float* grad_dice, const float* grad_ev,
                        void* dlogits, void* stream) {
  if (!logits || !labels || !scal || !dlogits || N <= 0 || V <= 0 || K < 2 || K > EV_MAXK) return DEVIS_ERR_BAD_ARG;
  if (!grad_loss && !grad_pv) return DEVIS_ERR_BAD_ARG;
  if (label_dtype < 0 || label_dtype > 2) return DEVIS_ERR_BAD_ARG;
  const EvidShape sh = {sN, sV, sK, N, V, (int)K};
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_evid_bwd<T>, dim3(ev_grid(N * V)), dim3(256), 0, (hipStream_t)stream, (const T*)logits, sh, labels, label_dtype, lambda, scal, grad_loss, grad_pv, grad_dice, grad_ev, (T*)dlogits));
  return launch_status();
}

int devis_evid_infer(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype, int64_t N, int64_t V, int64_t K,
                     float* evidence_out, uint8_t* pred, float* u, float* mne, uint16_t* umask,
                     const float* thresholds_host, int n_thr, void* stream) {
  if (!logits || N <= 0 || V <= 0 || K < 2 || K > EV_MAXK || n_thr < 0 || n_thr > 16) return DEVIS_ERR_BAD_ARG;
  if (umask && n_thr > 0 && !thresholds_host) return DEVIS_ERR_BAD_ARG;
  Thr thr;
  thr.n = umask ? n_thr : 0;
  for (int j = 0; j < 16; ++j) thr.t[j] = (j < thr.n) ? thresholds_host[j] : 0.f;
  const EvidShape sh = {sN, sV, sK, N, V, (int)K};
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_evid_infer<T>, dim3(ev_grid(N * V)), dim3(256), 0, (hipStream_t)stream, (const T*)logits, sh, evidence_out, pred, u, mne, umask, thr));
  return launch_status();
}

int devis_softplus_bwd(const float* z, const float* g, float* dz, int64_t n, void* stream) {
  if (!z || !g || !dz || n <= 0) return DEVIS_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_softplus_bwd, dim3(ev_grid(n)), dim3(256), 0, (hipStream_t)stream, z, g, dz, n);
  return launch_status();
}

}  // extern "C"
