// Fused subjective-logic evidential head + loss.
//   pass 1 (fwd):  z,y -> sum_v(L_ace + lambda*KL), I_k, P_k, R_k   (+ optional evidence / per-voxel terms)
//   pass 2 (bwd):  z,y -> dloss/dz
//   inference:     z   -> evidence, argmax mask, u = K/S, entropy map, u-threshold bit masks
// digamma / trigamma / lgamma are evaluated in registers (argument >= 1 always: alpha = softplus+1).
#include "common.h"
#include <mutex>

#define EV_MAXK DEVIS_EVID_MAX_K
#define EV_BLOCKS 1024
#define EV_NACC (1 + 3 * EV_MAXK)

// Scratch of the one entry point that allocates nothing (devis_evid_loss_fwd_noalloc, bottom of this file): the
// per-block fp64 partial sums of k_crit_fwd, which the other entry points take from the caller's workspace.
// Every GPU the library runs on holds its own copy of this array until the process ends (200 KB).  In host code
// the NAME is only a handle for hipGetSymbolAddress; kernels are given the resolved per-device pointer.
#define EV_SCRATCH_BYTES (EV_BLOCKS * EV_NACC * sizeof(double))
static __device__ __align__(256) unsigned char scratch[EV_SCRATCH_BYTES];

namespace devis {

__device__ __forceinline__ float softplus_f(float z) { return z > 20.f ? z : log1pf(expf(z)); }
// d softplus / dz exactly as torch's softplus_backward: exp(z)/(exp(z)+1), 1 above the threshold
__device__ __forceinline__ float dsoftplus_f(float z) {
  if (z > 20.f) return 1.f;
  const float t = expf(z);
  return t / (t + 1.f);
}

// Training-path variants on the hardware exp/log/rcp (1 ulp): |error| < 1e-7 + 1e-6 * value, far inside the
// parity bars; the inference head keeps the libm forms above (its argmax / threshold masks are bit-exact).
__device__ __forceinline__ float softplus_fast(float z) {
  const float t = __builtin_amdgcn_exp2f(fminf(z, 20.f) * 1.4426950408889634f);
  const float small = t * (1.f - t * (0.5f - t * 0.33333333f));      // log1p(t), t < 0.01
  const float big = __builtin_amdgcn_logf(1.f + t) * 0.69314718055994531f;   // 1 + t >= 1: no denormal rescaling needed
  return z > 20.f ? z : (t < 0.01f ? small : big);
}
__device__ __forceinline__ float dsoftplus_fast(float z) {
  const float t = __builtin_amdgcn_exp2f(fminf(z, 20.f) * 1.4426950408889634f);
  return z > 20.f ? 1.f : t * __builtin_amdgcn_rcpf(t + 1.f);
}

// ---- special functions on x >= 1 (alpha = softplus + 1 and sums of alphas): no argument shift, no branch.
// With t = 1/x in (0, 1]:
//   psi(x)      = ln x - t/2 + t^2 * P(t)                           (P -> -1/12 + t^2/120 - ...)
//   ln Gamma(x) = (x - 1/2) ln x - x + ln(2 pi)/2 + t * Q(t)         (Q ->  1/12 - t^2/360 + ...)
//   psi'(x)     = t * (1 + t * (1/2 + t * R(t)))                     (R ->  1/6  - t^2/30  + ...)
// P, Q (degree 5) and R (degree 6) are weighted minimax fits on the whole of (0, 1] to the remainders of the
// Stirling series (fitted against mpmath at 40 digits, tools/fit_special.py): |psi error| < 2.9e-8,
// |lnGamma error| < 1.8e-8, relative psi' error < 3.6e-8 before fp32 rounding (half an fp32 ulp of a value of
// order one, which is what fp32 arithmetic on these functions leaves anyway) - one reciprocal and one logarithm per
// argument (hardware rcp / log2, 1 ulp), where the shift-by-6 recurrence of round 1 needed two of each plus twice
// the multiplies.  The kernels here are bound by the vector ALU, not by HBM, so this is where their time goes.
__device__ __forceinline__ float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }
// raw v_log_f32 (log2) * ln 2: every argument here is >= 1, so __logf's denormal-range rescaling (five extra
// instructions per call) is dead weight
__device__ __forceinline__ float log_f(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
__device__ __forceinline__ float exp_f(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ float psi_rem(float t) {      // P(t), degree 5
  float r = 3.2940332945e-04f;
  r = fmaf(r, t, -2.8618493519e-04f);
  r = fmaf(r, t, -3.3689002729e-03f);
  r = fmaf(r, t, 9.6790001597e-03f);
  r = fmaf(r, t, -2.5322229087e-04f);
  return fmaf(r, t, -8.3315732563e-02f);
}
__device__ __forceinline__ float lgamma_rem(float t) {   // Q(t), degree 5
  float r = -1.5870583415e-04f;
  r = fmaf(r, t, 4.1252054178e-04f);
  r = fmaf(r, t, 3.6647560987e-04f);
  r = fmaf(r, t, -2.9122530426e-03f);
  r = fmaf(r, t, 2.1162378472e-05f);
  return fmaf(r, t, 8.3332249886e-02f);
}
__device__ __forceinline__ float trigamma_rem(float t) { // R(t), degree 6
  float r = 3.1959869627e-03f;
  r = fmaf(r, t, -1.2391032702e-02f);
  r = fmaf(r, t, 1.3949583783e-02f);
  r = fmaf(r, t, 9.7518234909e-03f);
  r = fmaf(r, t, -3.6750482461e-02f);
  r = fmaf(r, t, 5.4306552782e-04f);
  return fmaf(r, t, 1.6663515729e-01f);
}
// from t = 1/x and lx = ln x
__device__ __forceinline__ float psi_from(float t, float lx) { return fmaf(t * t, psi_rem(t), fmaf(-0.5f, t, lx)); }
__device__ __forceinline__ float lgamma_from(float x, float t, float lx) {
  return fmaf(t, lgamma_rem(t), fmaf(x - 0.5f, lx, 0.91893853320467f - x));
}
// psi(x) and ln Gamma(x) together (they share 1/x and ln x)
__device__ __forceinline__ void psi_lgamma_f(float x, float& psi, float& lg) {
  const float t = rcp_f(x), lx = log_f(x);
  psi = psi_from(t, lx);
  lg = lgamma_from(x, t, lx);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void psi_lgamma_f2(f32x2 x, f32x2& psi, f32x2& lg) {
  float p0, l0, p1, l1;
  psi_lgamma_f(x.x, p0, l0);
  psi_lgamma_f(x.y, p1, l1);
  psi = (f32x2){p0, p1};
  lg = (f32x2){l0, l1};
}
__device__ __forceinline__ float digamma_f(float x) { return psi_from(rcp_f(x), log_f(x)); }
__device__ __forceinline__ float lgamma_f(float x) { return lgamma_from(x, rcp_f(x), log_f(x)); }
// psi'(x), x >= 1
__device__ __forceinline__ float trigamma_f(float x) {
  const float t = rcp_f(x);
  return t * fmaf(t, fmaf(t, trigamma_rem(t), 0.5f), 1.f);
}

__device__ __forceinline__ int load_label(const void* labels, int ldt, int64_t i) {
  if (ldt == DEVIS_LABEL_I64) return (int)((const int64_t*)labels)[i];
  if (ldt == DEVIS_LABEL_U8) return (int)((const uint8_t*)labels)[i];
  return ((const int32_t*)labels)[i];
}

struct EvidShape {
  int64_t sN, sV, sK, N, V;
  int K;
  int vec;   // channels-last rows of exactly 4 contiguous, aligned elements: one vector access per voxel
};

// criterion selector (devis_evid_params.kind) and the per-call constants
enum { CRIT_DCE = 0, CRIT_DCEU = 1, CRIT_CE = 2, CRIT_MSE = 3, CRIT_KL = 4 };
struct EvidParams {
  int raw;         // 0: input = logits (softplus inside); 1: input = alpha (evidence = alpha - 1)
  int disentangle;
  float lambda, a_au, eps;
};

__device__ __forceinline__ void load4(const float* p, float* out) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16* p, float* out) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  out[0] = __uint_as_float(t.x << 16); out[1] = __uint_as_float(t.x & 0xffff0000u);
  out[2] = __uint_as_float(t.y << 16); out[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16* p, const float* v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

// KT: compile-time class count (4) or 0 for the run-time K <= EV_MAXK path
template <typename T, int KT>
__device__ __forceinline__ void load_logits(const T* z, const EvidShape& s, int64_t n, int64_t v, float* out) {
  const int K = KT ? KT : s.K;
  const T* p = z + n * s.sN + v * s.sV;
  if (KT == 4 && s.vec) {
    load4(p, out);
#pragma unroll
    for (int k = 4; k < EV_MAXK; ++k) out[k] = 0.f;
    return;
  }
#pragma unroll
  for (int k = 0; k < EV_MAXK; ++k) out[k] = k < K ? ld(p + k * s.sK) : 0.f;
}

// e, alpha of one voxel from the input row; ed = the tensor the dice term is taken on
template <typename T, int KT>
__device__ __forceinline__ void load_voxel(const T* in, const float* ev_in, const EvidShape& s, const EvidParams& p,
                                           int64_t n, int64_t v, float* zz, float* e, float* al, float* ed) {
  const int K = KT ? KT : s.K;
  load_logits<T, KT>(in, s, n, v, zz);
#pragma unroll
  for (int k = 0; k < EV_MAXK; ++k) {
    if (k < K) {
      if (p.raw) { al[k] = zz[k]; e[k] = zz[k] - 1.f; }
      else { e[k] = softplus_fast(zz[k]); al[k] = e[k] + 1.f; }
      ed[k] = ev_in ? ev_in[(n * K + k) * s.V + v] : e[k];
    } else { e[k] = 0.f; al[k] = 0.f; ed[k] = 0.f; }
  }
}

// softmax over the K dice inputs (DiceLoss: exp(log_softmax), criterions.py:176)
__device__ __forceinline__ void softmax_k(const float* ed, int K, float* prob) {
  float m = ed[0];
#pragma unroll
  for (int k = 1; k < EV_MAXK; ++k) if (k < K) m = fmaxf(m, ed[k]);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < EV_MAXK; ++k) { prob[k] = k < K ? expf(ed[k] - m) : 0.f; sum += prob[k]; }
  const float inv = 1.f / sum;
#pragma unroll
  for (int k = 0; k < EV_MAXK; ++k) prob[k] *= inv;
}

// ----------------------------------------------------------------- forward
// acc[0] = sum of per-voxel terms; acc[1+k], acc[1+MAXK+k], acc[1+2MAXK+k] = the three per-class sums of
// the dice term: (I, P, R) for the soft dice of CRIT_DCE, (TP, FP, FN) for the Tversky dice of CRIT_DCEU.
template <typename T, int KIND, int KT>
__global__ void __launch_bounds__(256)
k_crit_fwd(const T* __restrict__ in, EvidShape s, const float* __restrict__ ev_in, const void* __restrict__ labels, int ldt,
           EvidParams p, float* __restrict__ evidence_out, float* __restrict__ per_voxel_out, double* __restrict__ partial) {
  const int K = KT ? KT : s.K;
  float acc[EV_NACC];
#pragma unroll
  for (int i = 0; i < EV_NACC; ++i) acc[i] = 0.f;
  const float lgK = lgamma_f((float)K);
  // grid = (blocks per sample, N): no 64-bit division per voxel
  const int64_t n = blockIdx.y;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < s.V; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = n * s.V + v;
    float zz[EV_MAXK], e[EV_MAXK], al[EV_MAXK], ed[EV_MAXK];
    load_voxel<T, KT>(in, ev_in, s, p, n, v, zz, e, al, ed);
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) {
      if (k < K) {
        S += al[k];
        if (evidence_out) evidence_out[(n * K + k) * s.V + v] = e[k];
      }
    }
    float pv;
    if (KIND == CRIT_KL) {
      // KL(Dir(alpha) || Dir(1)), criterions.py:117-127
      const float psiS = digamma_f(S);
      float sum_lg = 0.f, kl_sum = 0.f;
#pragma unroll
      for (int k = 0; k < EV_MAXK; ++k) {
        if (k < K) {
          float ps, lg;
          psi_lgamma_f(al[k], ps, lg);
          sum_lg += lg;
          kl_sum = fmaf(al[k] - 1.f, ps - psiS, kl_sum);
        }
      }
      pv = lgamma_f(S) - sum_lg - lgK + kl_sum;
    } else {
      const int y = load_label(labels, ldt, i);
      float e_y = 0.f, a_y = 1.f;
#pragma unroll
      for (int k = 0; k < EV_MAXK; ++k) if (k == y) { e_y = e[k]; a_y = al[k]; }
      const float St = S - e_y;            // strength of alpha~ (label class reset to 1)
      float psiSt, lgSt, psiS4 = 0.f, psiAy4 = 0.f;
      float sum_lg = 0.f, kl_sum = 0.f;
      if (KT == 4 && (KIND == CRIT_DCE || KIND == CRIT_CE || KIND == CRIT_DCEU)) {
        // K = 4: the four classes as two packed pairs, (St, S) as a third; psi(alpha_y) is one of the class values
        f32x2 ps01, lg01, ps23, lg23, psS, lgS;
        psi_lgamma_f2((f32x2){al[0], al[1]}, ps01, lg01);
        psi_lgamma_f2((f32x2){al[2], al[3]}, ps23, lg23);
        psi_lgamma_f2((f32x2){St, S}, psS, lgS);
        psiSt = psS.x; lgSt = lgS.x; psiS4 = psS.y;
        const float ps[4] = {ps01.x, ps01.y, ps23.x, ps23.y}, lgv[4] = {lg01.x, lg01.y, lg23.x, lg23.y};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool lab = k == y;
          sum_lg += lab ? 0.f : lgv[k];
          kl_sum = lab ? kl_sum : fmaf(e[k], ps[k] - psiSt, kl_sum);
          psiAy4 = lab ? ps[k] : psiAy4;
        }
      } else {
        psi_lgamma_f(St, psiSt, lgSt);
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) {
          if (k < K && k != y) {
            float ps, lg;
            psi_lgamma_f(al[k], ps, lg);
            sum_lg += lg;
            kl_sum = fmaf(e[k], ps - psiSt, kl_sum);
          }
        }
      }
      const float kl = lgSt - sum_lg - lgK + kl_sum;
      if (KIND == CRIT_MSE) {
        // criterions.py:144-154: sum (label - alpha/S)^2 + sum alpha(S-alpha)/(S^2 (S+1))
        float A = 0.f, B = 0.f;
        const float den = S * S * (S + 1.f);
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) {
          if (k < K) {
            const float d = (k == y ? 1.f : 0.f) - al[k] / S;
            A = fmaf(d, d, A);
            B += al[k] * (S - al[k]) / den;
          }
        }
        pv = (A + B) + p.lambda * kl;
      } else {
        if (KT == 4) pv = fmaf(p.lambda, kl, psiS4 - psiAy4);
        else pv = fmaf(p.lambda, kl, digamma_f(S) - digamma_f(a_y));
      }
      if (KIND == CRIT_DCE) {
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) {
          if (k < K) {
            const bool hit = (k == y);
            acc[1 + k] += hit ? ed[k] : 0.f;              // I_k
            acc[1 + EV_MAXK + k] += ed[k];                // P_k
            acc[1 + 2 * EV_MAXK + k] += hit ? 1.f : 0.f;  // R_k
          }
        }
      }
      if (KIND == CRIT_DCEU) {
        // accuracy-vs-uncertainty term, criterions.py:397-411: score = max_k alpha_k/S (first max), u = K/S
        float score = -1.f;
        int cls = 0;
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) {
          if (k < K) { const float pk = al[k] / S; if (pk > score) { score = pk; cls = k; } }
        }
        const float u = (float)K / S;
        float l_au;
        if (cls == y) {
          const float t = p.disentangle ? -logf(score * (1.f - u) + p.eps) : -score * logf(1.f - u + p.eps);
          l_au = p.a_au * t;
        } else {
          const float t = p.disentangle ? -logf((1.f - score) * u + p.eps) : -(1.f - score) * logf(u + p.eps);
          l_au = (1.f - p.a_au) * t;
        }
        pv += l_au;
        // Tversky dice sums on softmax(evidence), criterions.py:166-215; class_mask = 1e-8 with 1 at the label
        float prob[EV_MAXK];
        softmax_k(ed, K, prob);
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) {
          if (k < K) {
            const float m = (k == y) ? 1.f : 1e-8f;
            acc[1 + k] = fmaf(prob[k], m, acc[1 + k]);                               // TP_k
            acc[1 + EV_MAXK + k] = fmaf(prob[k], 1.f - m, acc[1 + EV_MAXK + k]);     // FP_k
            acc[1 + 2 * EV_MAXK + k] = fmaf(1.f - prob[k], m, acc[1 + 2 * EV_MAXK + k]);  // FN_k
          }
        }
      }
    }
    if (per_voxel_out) per_voxel_out[i] = pv;
    acc[0] += pv;
  }
  if (!partial) return;
  __shared__ double red[4][EV_NACC];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < EV_NACC; ++i) {
    const bool used = i == 0 || (i - 1) % EV_MAXK < K;       // accumulators of classes >= K stay zero
    const double w = used ? wave_sum((double)acc[i]) : 0.0;
    if (lane == 0) red[wave][i] = w;
  }
  __syncthreads();
  if (threadIdx.x < EV_NACC)
    partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * EV_NACC + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Tversky dice of DiceLoss (criterions.py:195-215) and its derivatives w.r.t. the three sums
struct Tversky { double dice, gTP, gFP, gFN; };
__host__ __device__ inline Tversky tversky(double TP, double FP, double FN) {
  const double sm = 1e-5;
  const double q = FP + FN + sm;
  double A = FP / q;
  double dA_dFP = (FN + sm) / (q * q), dA_dFN = -FP / (q * q);
  if (A < 0.2) { A = 0.2; dA_dFP = 0.0; dA_dFN = 0.0; }
  if (A > 0.8) { A = 0.8; dA_dFP = 0.0; dA_dFN = 0.0; }
  const double den = TP + A * FP + (1.0 - A) * FN;
  const double dd = den + sm;
  Tversky t;
  t.dice = TP / dd;
  // L_k = 1 - TP/dd
  t.gTP = -(dd - TP) / (dd * dd);
  t.gFP = TP * (A + (FP - FN) * dA_dFP) / (dd * dd);
  t.gFN = TP * ((1.0 - A) + (FP - FN) * dA_dFN) / (dd * dd);
  return t;
}

// 1024 threads.  The partial table [blocks][EV_NACC] is read as one contiguous array: thread t < 40 * EV_NACC owns
// column t % EV_NACC of the rows t / EV_NACC, + 40, + 80, ... (a wave load is 512 contiguous bytes, eight loads in
// flight per thread); the 40 row groups of a column then add up in row-group order through LDS - a fixed order, so
// the result does not depend on the launch.  The K dice terms (fp64 log) are evaluated by K threads side by side.
__global__ void __launch_bounds__(1024)
k_evid_finalize(const double* __restrict__ partial, int blocks, int K, double NV, int kind, float a_au,
                float* __restrict__ scal) {
  constexpr int RG = 40;
  __shared__ double red[RG][EV_NACC];
  __shared__ double tot[EV_NACC];
  __shared__ double dterm[EV_MAXK];
  const int t = threadIdx.x;
  if (t < RG * EV_NACC) {
    const int col = t % EV_NACC, rg = t / EV_NACC;
    const double* p = partial + col;
    double a = 0.0;
    int b = rg;
    for (; b + 7 * RG < blocks; b += 8 * RG) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(b + u * RG) * EV_NACC];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; b < blocks; b += RG) a += p[(int64_t)b * EV_NACC];
    red[rg][col] = a;
  }
  __syncthreads();
  if (t < EV_NACC) {
    double s = 0.0;
#pragma unroll 8
    for (int w = 0; w < RG; ++w) s += red[w][t];
    tot[t] = s;
  }
  __syncthreads();
  if (t < K) {
    const double s0 = tot[1 + t], s1 = tot[1 + EV_MAXK + t], s2 = tot[1 + 2 * EV_MAXK + t];
    double d = 0.0;
    if (kind == CRIT_DCE) d = -log((2.0 * s0 + 1e-5) / (s2 + s1 + 1.0 + 1e-5));
    if (kind == CRIT_DCEU) d = 1.0 - tversky(s0, s1, s2).dice;
    dterm[t] = d;
    scal[4 + t] = (float)s0;
    scal[4 + K + t] = (float)s1;
    scal[4 + 2 * K + t] = (float)s2;
  }
  __syncthreads();
  if (t == 0) {
    double dice = 0.0;
    for (int k = 0; k < K; ++k) dice += dterm[k];
    dice /= (double)K;
    const double mean_pv = tot[0] / NV;
    const double wd = kind == CRIT_DCEU ? 1.0 - (double)a_au : 1.0;     // criterions.py:417
    scal[0] = (float)(mean_pv + wd * dice);
    scal[1] = (float)dice;
    scal[2] = (float)mean_pv;
    scal[3] = (float)NV;
  }
}

// ---------------------------------------------------------------- backward
// d_in = d loss / d input (logits: chain through softplus; alpha in raw mode).  In raw mode with a
// separate dice tensor, the dice term's gradient goes to d_ev_in instead.
template <typename T, int KIND, int KT>
__global__ void __launch_bounds__(256)
k_crit_bwd(const T* __restrict__ in, EvidShape s, const float* __restrict__ ev_in, const void* __restrict__ labels, int ldt,
           EvidParams p, const float* __restrict__ scal, const float* __restrict__ grad_loss,
           const float* __restrict__ grad_pv, const float* __restrict__ grad_dice, const float* __restrict__ grad_ev,
           T* __restrict__ d_in, float* __restrict__ d_ev_in) {
  const int K = KT ? KT : s.K;
  const int64_t NV = s.N * s.V;
  // dice coefficients.  CRIT_DCE: dL/de_k(v) = c1_k - [y=k]*c0_k.   CRIT_DCEU: dL/dprob_k(v) = [y=k] ? c0_k : c1_k.
  float c0[EV_MAXK], c1[EV_MAXK];
  float gd = 0.f;
  if (KIND == CRIT_DCE || KIND == CRIT_DCEU) {
    if (grad_pv) gd = grad_dice ? grad_dice[0] : 0.f;
    else gd = grad_loss[0] * (KIND == CRIT_DCEU ? 1.f - p.a_au : 1.f);
  }
  const float gmean = grad_pv ? 0.f : grad_loss[0] / (float)NV;
#pragma unroll
  for (int k = 0; k < EV_MAXK; ++k) {
    c0[k] = 0.f; c1[k] = 0.f;
    if (k < K && KIND == CRIT_DCE) {
      const float I = scal[4 + k], P = scal[4 + K + k], R = scal[4 + 2 * K + k];
      c0[k] = gd * 2.f / ((2.f * I + 1e-5f) * (float)K);
      c1[k] = gd / ((R + P + 1.f + 1e-5f) * (float)K);
    }
    if (k < K && KIND == CRIT_DCEU) {
      const Tversky t = tversky((double)scal[4 + k], (double)scal[4 + K + k], (double)scal[4 + 2 * K + k]);
      const double w = (double)gd / (double)K;
      c0[k] = (float)(w * (t.gTP - t.gFN));                            // mask = 1: tp and fn rows
      c1[k] = (float)(w * (t.gTP * 1e-8 + t.gFP * (double)(1.f - 1e-8f) - t.gFN * 1e-8));
    }
  }
  // grid = (blocks per sample, N): no 64-bit division per voxel
  const int64_t n = blockIdx.y;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < s.V; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = n * s.V + v;
    float zz[EV_MAXK], e[EV_MAXK], al[EV_MAXK], ed[EV_MAXK];
    load_voxel<T, KT>(in, ev_in, s, p, n, v, zz, e, al, ed);
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) if (k < K) S += al[k];
    const float g = grad_pv ? grad_pv[i] : gmean;
    float d[EV_MAXK], dd[EV_MAXK];
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) { d[k] = 0.f; dd[k] = 0.f; }
    const float tS = trigamma_f(S);
    if (KIND == CRIT_KL) {
      const float tail = tS * (S - (float)K);
#pragma unroll
      for (int k = 0; k < EV_MAXK; ++k)
        if (k < K) d[k] = g * fmaf(al[k] - 1.f, trigamma_f(al[k]), -tail);
    } else {
      const int y = load_label(labels, ldt, i);
      float e_y = 0.f;
#pragma unroll
      for (int k = 0; k < EV_MAXK; ++k) if (k == y) e_y = e[k];
      const float St = S - e_y;
      const float tSt = trigamma_f(St) * (St - (float)K);
      if (KIND == CRIT_MSE) {
        // A = sum (l - m)^2, m = alpha/S;  B = (S^2 - Q)/(S^3 + S^2), Q = sum alpha^2
        float Q = 0.f, cross = 0.f;
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) {
          if (k < K) {
            const float m = al[k] / S;
            Q = fmaf(al[k], al[k], Q);
            cross = fmaf((k == y ? 1.f : 0.f) - m, m, cross);
          }
        }
        const float D = S * S * (S + 1.f);
        const float dD = 3.f * S * S + 2.f * S;
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) {
          if (k < K) {
            const float m = al[k] / S;
            const float dA = -2.f / S * (((k == y ? 1.f : 0.f) - m) - cross);
            const float dB = ((2.f * S - 2.f * al[k]) * D - (S * S - Q) * dD) / (D * D);
            const float dk = (k == y) ? 0.f : p.lambda * fmaf(e[k], trigamma_f(al[k]), -tSt);
            d[k] = g * (dA + dB + dk);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) {
          if (k < K) {
            const float ta = trigamma_f(al[k]);
            d[k] = g * ((k == y) ? (tS - ta) : fmaf(p.lambda, fmaf(e[k], ta, -tSt), tS));
          }
        }
      }
      if (KIND == CRIT_DCE) {
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) if (k < K) dd[k] = c1[k] - ((k == y) ? c0[k] : 0.f);
      }
      if (KIND == CRIT_DCEU) {
        float score = -1.f;
        int cls = 0;
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) {
          if (k < K) { const float pk = al[k] / S; if (pk > score) { score = pk; cls = k; } }
        }
        const float u = (float)K / S;
        const float du = -u / S;                       // d u / d alpha_k (every k)
        // f = w * t(score, u): df/dalpha_k = w * (t_s * dscore_k + t_u * du), dscore_k = ([k=cls] - score)/S
        float w, t_s, t_u;
        if (cls == y) {
          w = p.a_au;
          if (p.disentangle) { const float q = score * (1.f - u) + p.eps; t_s = -(1.f - u) / q; t_u = score / q; }
          else { const float q = 1.f - u + p.eps; t_s = -logf(q); t_u = score / q; }
        } else {
          w = 1.f - p.a_au;
          if (p.disentangle) { const float q = (1.f - score) * u + p.eps; t_s = u / q; t_u = -(1.f - score) / q; }
          else { const float q = u + p.eps; t_s = logf(q); t_u = -(1.f - score) / q; }
        }
        const float gw = g * w;
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k)
          if (k < K) d[k] += gw * fmaf(t_s, ((k == cls ? 1.f : 0.f) - score) / S, t_u * du);
        float prob[EV_MAXK];
        softmax_k(ed, K, prob);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) if (k < K) dot = fmaf(prob[k], (k == y) ? c0[k] : c1[k], dot);
#pragma unroll
        for (int k = 0; k < EV_MAXK; ++k) if (k < K) dd[k] = prob[k] * (((k == y) ? c0[k] : c1[k]) - dot);
      }
    }
    T* out = d_in + n * s.sN + v * s.sV;
    float rr[EV_MAXK];
#pragma unroll
    for (int k = 0; k < EV_MAXK; ++k) {
      rr[k] = 0.f;
      if (k < K) {
        float r = d[k];
        if (d_ev_in) d_ev_in[(n * K + k) * s.V + v] = dd[k];
        else r += dd[k];
        if (!p.raw) {
          if (grad_ev) r += grad_ev[(n * K + k) * s.V + v];
          r *= dsoftplus_fast(zz[k]);
        }
        rr[k] = r;
      }
    }
    if (KT == 4 && s.vec) {
      store4(out, rr);
    } else {
#pragma unroll
      for (int k = 0; k < EV_MAXK; ++k) if (k < K) st(out + k * s.sK, rr[k]);
    }
  }
}

// ------------------------------------------- the training hot path: K = 4, logits in, channels-last rows
// devis_evid_loss_fwd / _bwd on the 3-D U-Net's endconv output (TrustworthySeg.py:144-155 + criterions.py:421-456).
// Same arithmetic as k_crit_fwd / k_crit_bwd<T, CRIT_DCE, 4>, written straight-line: one vector load per voxel,
// no per-class branches, no scratch, two voxels in flight per thread, the label counts R_k taken with wave-wide
// ballots on the scalar unit.  Vector-ALU budget per voxel: 20 transcendentals (4 exp2 + 4 log2 for softplus,
// one rcp + one log2 for each of alpha_0..3, S~, S) + about 230 fused multiply-adds.

// softplus(z) = max(z, 0) + log1p(exp(-|z|)); t = exp(-|z|) in (0, 1], u = fl(1 + t).  Equal to torch's
// thresholded form (z for z > 20) to within 2e-9 relative.  log1p(t) = ln(u) + d / u with d = t - (u - 1) the
// rounding error of u (exact in fp32); d / u is replaced by d: the difference d t / u is below 6e-8 t, so tiny
// evidences keep their relative accuracy (u = 1 gives exactly t) without a branch or a select.
__device__ __forceinline__ float softplus_tu(float z, float& t, float& u) {
  t = __builtin_amdgcn_exp2f(fabsf(z) * -1.4426950408889634f);
  u = 1.f + t;
  const float l1p = fmaf(__builtin_amdgcn_logf(u), 0.69314718055994531f, t - (u - 1.f));
  return fmaxf(z, 0.f) + l1p;
}

template <int LDT> __device__ __forceinline__ int label_at(const void* labels, int64_t i) {
  if (LDT == DEVIS_LABEL_I64) return ((const int*)labels)[2 * i];      // low half: class indices are tiny
  if (LDT == DEVIS_LABEL_U8) return (int)((const uint8_t*)labels)[i];
  return ((const int32_t*)labels)[i];
}

// per-voxel term L_ace + lambda * KL and the evidence of one voxel
__device__ __forceinline__ float dce4_voxel(const float* z, int y, float lambda, float lgK, float* e) {
  float al[4], tk[4], lk[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t, u;
    e[k] = softplus_tu(z[k], t, u);
    al[k] = e[k] + 1.f;
  }
  const float S = (al[0] + al[1]) + (al[2] + al[3]);
  float e_y = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) e_y = (y == k) ? e[k] : e_y;
  const float St = S - e_y;                    // strength of alpha~ (label class reset to 1)
#pragma unroll
  for (int k = 0; k < 4; ++k) { tk[k] = rcp_f(al[k]); lk[k] = log_f(al[k]); }
  const float tS = rcp_f(S), lS = log_f(S), tT = rcp_f(St), lT = log_f(St);
  const float psiS = psi_from(tS, lS), psiT = psi_from(tT, lT), lgT = lgamma_from(St, tT, lT);
  float psi_y = 0.f, sum_lg = 0.f, kl = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float ps = psi_from(tk[k], lk[k]), lg = lgamma_from(al[k], tk[k], lk[k]);
    const bool lab = (y == k);
    psi_y = lab ? ps : psi_y;
    sum_lg += lab ? 0.f : lg;
    kl = lab ? kl : fmaf(e[k], ps - psiT, kl);
  }
  if (y < 0 || y > 3) psi_y = psi_from(1.f, 0.f);          // no class matched: alpha_y = 1, as k_crit_fwd has it
  return fmaf(lambda, (lgT - sum_lg - lgK) + kl, psiS - psi_y);
}

template <typename T, int LDT, bool EVOUT, bool PVOUT>
__global__ void __launch_bounds__(256)
k_dce4_fwd(const T* __restrict__ z, int64_t sN, const void* __restrict__ labels, int64_t V, float lambda,
           float* __restrict__ evidence_out, float* __restrict__ per_voxel_out, double* __restrict__ partial) {
  const float lgK = 1.791759469228055f;                    // ln Gamma(4) = ln 6
  const int64_t n = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const T* zn = z + n * sN;                                 // the sample stride is the caller's (>= 4 V, a multiple of 4)
  float apv = 0.f, aI[4] = {0.f, 0.f, 0.f, 0.f}, aP[4] = {0.f, 0.f, 0.f, 0.f};
  int rc[4] = {0, 0, 0, 0};                                 // wave-uniform label counts
  for (int64_t v0 = (int64_t)blockIdx.x * 256 + threadIdx.x; v0 < V; v0 += 2 * stride) {
    const int64_t v1 = v0 + stride;
    const bool two = v1 < V;
    float za[4], zb[4] = {0.f, 0.f, 0.f, 0.f};
    load4(zn + v0 * 4, za);
    const int ya = label_at<LDT>(labels, n * V + v0);
    int yb = -1;
    if (two) { load4(zn + v1 * 4, zb); yb = label_at<LDT>(labels, n * V + v1); }
    float ea[4], eb[4];
    const float pa = dce4_voxel(za, ya, lambda, lgK, ea);
    float pb = dce4_voxel(zb, yb, lambda, lgK, eb);
    if (!two) { pb = 0.f; eb[0] = eb[1] = eb[2] = eb[3] = 0.f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (EVOUT) {
        evidence_out[(n * 4 + k) * V + v0] = ea[k];
        if (two) evidence_out[(n * 4 + k) * V + v1] = eb[k];
      }
      aI[k] += ((ya == k) ? ea[k] : 0.f) + ((yb == k) ? eb[k] : 0.f);
      aP[k] += ea[k] + eb[k];
      rc[k] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ya == k)) +
               __builtin_popcountll(__builtin_amdgcn_ballot_w64(yb == k));
    }
    if (PVOUT) {
      per_voxel_out[n * V + v0] = pa;
      if (two) per_voxel_out[n * V + v1] = pb;
    }
    apv += pa + pb;
  }
  if (!partial) return;
  __shared__ double red[4][13];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    const double w = wave_sum((double)apv);
    if (lane == 0) red[wave][0] = w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double wi = wave_sum((double)aI[k]), wp = wave_sum((double)aP[k]);
    if (lane == 0) { red[wave][1 + k] = wi; red[wave][5 + k] = wp; red[wave][9 + k] = (double)rc[k]; }
  }
  __syncthreads();
  if (threadIdx.x < EV_NACC) {
    // slot i of this block's row: 0 = per-voxel sum, 1 + k = I_k, 1 + MAXK + k = P_k, 1 + 2 MAXK + k = R_k
    const int i = threadIdx.x, grp = i == 0 ? -1 : (i - 1) / EV_MAXK, k = i == 0 ? 0 : (i - 1) % EV_MAXK;
    double s = 0.0;
    if (i == 0 || k < 4) {
      const int j = i == 0 ? 0 : 1 + 4 * grp + k;
      s = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    }
    partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * EV_NACC + i] = s;
  }
}

// d loss / d z of one voxel: sigma(z_k) * { g [psi'(S) - [k=y] psi'(a_y) + lambda [k!=y] (e_k psi'(a_k) - psi'(S~)(S~-K))]
//                                            + c1_k - [k=y] c0_k }        
__device__ __forceinline__ void dce4_voxel_bwd(const float* z, int y, float lambda, float g, const float* c0,
                                               const float* c1, float* r) {
  float e[4], al[4], sg[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t, u;
    e[k] = softplus_tu(z[k], t, u);
    al[k] = e[k] + 1.f;
    sg[k] = (z[k] >= 0.f ? 1.f : t) * rcp_f(u);              // sigmoid(z) = d softplus / dz
  }
  const float S = (al[0] + al[1]) + (al[2] + al[3]);
  float e_y = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) e_y = (y == k) ? e[k] : e_y;
  const float St = S - e_y;
  const float trS = trigamma_f(S), tail = trigamma_f(St) * (St - 4.f);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float ta = trigamma_f(al[k]);
    const bool lab = (y == k);
    const float d = lab ? (trS - ta) : fmaf(lambda, fmaf(e[k], ta, -tail), trS);
    r[k] = fmaf(g, d, c1[k] - (lab ? c0[k] : 0.f)) * sg[k];
  }
}

template <typename T, int LDT, bool GEV>
__global__ void __launch_bounds__(256)
k_dce4_bwd(const T* __restrict__ z, int64_t sN, const void* __restrict__ labels, int64_t N, int64_t V, float lambda,
           const float* __restrict__ scal, const float* __restrict__ grad_loss, const float* __restrict__ grad_ev,
           T* __restrict__ dz) {
  const float gl = grad_loss[0];
  const float g = gl / (float)(N * V);
  float c0[4], c1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float I = scal[4 + k], P = scal[8 + k], R = scal[12 + k];
    c0[k] = gl * 2.f / ((2.f * I + 1e-5f) * 4.f);
    c1[k] = gl / ((R + P + 1.f + 1e-5f) * 4.f);
  }
  const int64_t n = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const T* zn = z + n * sN;
  T* dn = dz + n * sN;
  for (int64_t v0 = (int64_t)blockIdx.x * 256 + threadIdx.x; v0 < V; v0 += 2 * stride) {
    const int64_t v1 = v0 + stride;
    const bool two = v1 < V;
    float za[4], zb[4] = {0.f, 0.f, 0.f, 0.f};
    load4(zn + v0 * 4, za);
    const int ya = label_at<LDT>(labels, n * V + v0);
    int yb = 0;
    if (two) { load4(zn + v1 * 4, zb); yb = label_at<LDT>(labels, n * V + v1); }
    float ra[4], rb[4];
    dce4_voxel_bwd(za, ya, lambda, g, c0, c1, ra);
    dce4_voxel_bwd(zb, yb, lambda, g, c0, c1, rb);
    if (GEV) {
      // + grad_evidence * sigmoid(z): recompute sigma cheaply from the row (rare path: the train loop drops evidence)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ra[k] += grad_ev[(n * 4 + k) * V + v0] * dsoftplus_fast(za[k]);
        if (two) rb[k] += grad_ev[(n * 4 + k) * V + v1] * dsoftplus_fast(zb[k]);
      }
    }
    store4(dn + v0 * 4, ra);
    if (two) store4(dn + v1 * 4, rb);
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
    load_logits<T, 0>(z, s, n, v, zz);
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

// (blocks per sample, N) with at most EV_BLOCKS blocks in all (N <= EV_BLOCKS), else one block per sample
static inline dim3 ev_grid2(int64_t N, int64_t V) {
  int64_t bx = cdiv(V, 256 * 4);                       // >= 4 voxels per thread: amortises the final reduction
  const int64_t cap = N < EV_BLOCKS ? EV_BLOCKS / N : 1;
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  return dim3((unsigned)bx, (unsigned)N);
}

static inline unsigned ev_grid(int64_t NV) {
  const int64_t b = cdiv(NV, 256);
  return (unsigned)(b < 1 ? 1 : (b > EV_BLOCKS ? EV_BLOCKS : b));
}

}  // namespace devis

using namespace devis;

extern "C" {

size_t devis_evid_workspace_bytes(int64_t N, int64_t V, int64_t K) {
  (void)V; (void)K;
  return (size_t)(N > EV_BLOCKS ? N : EV_BLOCKS) * EV_NACC * sizeof(double);
}

static bool crit_args_ok(const devis_evid_params* p, const void* in, const void* labels, int label_dtype, int64_t N,
                         int64_t V, int64_t K) {
  if (!p || !in || N <= 0 || N > 65535 || V <= 0 || K < 2 || K > EV_MAXK) return false;
  if (p->kind < 0 || p->kind > CRIT_KL || (p->input_kind != 0 && p->input_kind != 1)) return false;
  if (p->kind != CRIT_KL && (!labels || label_dtype < 0 || label_dtype > 2)) return false;
  return true;
}

static int evid_vec(const void* p, int64_t sN, int64_t sV, int64_t sK, int64_t K, int dtype) {
  const size_t es = dtype == DEVIS_F32 ? 4 : 2;
  return K == 4 && sK == 1 && sV == 4 && sN % 4 == 0 && ((size_t)p % (4 * es)) == 0;
}

#define DEVIS_DISPATCH_KT(K, ...)                         \
  if ((K) == 4) { constexpr int KT = 4; __VA_ARGS__; }    \
  else { constexpr int KT = 0; __VA_ARGS__; }

#define DEVIS_DISPATCH_KIND(kind, ...)                                              \
  switch (kind) {                                                                   \
    case CRIT_DCE: { constexpr int KIND = CRIT_DCE; __VA_ARGS__; } break;           \
    case CRIT_DCEU: { constexpr int KIND = CRIT_DCEU; __VA_ARGS__; } break;         \
    case CRIT_CE: { constexpr int KIND = CRIT_CE; __VA_ARGS__; } break;             \
    case CRIT_MSE: { constexpr int KIND = CRIT_MSE; __VA_ARGS__; } break;           \
    default: { constexpr int KIND = CRIT_KL; __VA_ARGS__; } break;                  \
  }

int devis_evid_crit_fwd(const devis_evid_params* p, const void* in, int64_t sN, int64_t sV, int64_t sK, int dtype,
                        const float* evidence_in, const void* labels, int label_dtype, int64_t N, int64_t V, int64_t K,
                        float* evidence_out, float* per_voxel_out, float* scal, void* ws, size_t ws_bytes, void* stream) {
  if (!crit_args_ok(p, in, labels, label_dtype, N, V, K)) return DEVIS_ERR_BAD_ARG;
  if (!scal && !per_voxel_out) return DEVIS_ERR_BAD_ARG;
  if (scal && (!ws || ws_bytes < devis_evid_workspace_bytes(N, V, K))) return ws ? DEVIS_ERR_WORKSPACE : DEVIS_ERR_BAD_ARG;
  if (evidence_in && !p->input_kind) return DEVIS_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const EvidShape sh = {sN, sV, sK, N, V, (int)K, evid_vec(in, sN, sV, sK, K, dtype)};
  const EvidParams ep = {p->input_kind, p->disentangle, p->lambda, p->a_au, p->eps};
  if (N > 65535) return DEVIS_ERR_UNSUPPORTED;
  const dim3 grid2 = ev_grid2(N, V);
  const unsigned blocks = grid2.x * grid2.y;
  double* part = scal ? (double*)ws : nullptr;
  if (p->kind == CRIT_DCE && !p->input_kind && !evidence_in && sh.vec) {
    // the training hot path (K = 4 is implied by vec): the straight-line kernel
#define LAUNCH_DCE4(LDT, EV, PV) hipLaunchKernelGGL((k_dce4_fwd<T, LDT, EV, PV>), grid2, dim3(256), 0, s, (const T*)in, sN, labels, V, p->lambda, evidence_out, per_voxel_out, part)
#define LAUNCH_DCE4_L(LDT) do { if (evidence_out && per_voxel_out) LAUNCH_DCE4(LDT, true, true); else if (evidence_out) LAUNCH_DCE4(LDT, true, false); else if (per_voxel_out) LAUNCH_DCE4(LDT, false, true); else LAUNCH_DCE4(LDT, false, false); } while (0)
    DEVIS_DISPATCH_DTYPE(dtype, {
      if (label_dtype == DEVIS_LABEL_I64) LAUNCH_DCE4_L(DEVIS_LABEL_I64);
      else if (label_dtype == DEVIS_LABEL_U8) LAUNCH_DCE4_L(DEVIS_LABEL_U8);
      else LAUNCH_DCE4_L(DEVIS_LABEL_I32);
    });
#undef LAUNCH_DCE4_L
#undef LAUNCH_DCE4
  } else
  DEVIS_DISPATCH_DTYPE(dtype, DEVIS_DISPATCH_KIND(p->kind, DEVIS_DISPATCH_KT(K, hipLaunchKernelGGL((k_crit_fwd<T, KIND, KT>), grid2, dim3(256), 0, s, (const T*)in, sh, evidence_in, labels, label_dtype, ep, evidence_out, per_voxel_out, part))));
  int rc = launch_status();
  if (rc || !scal) return rc;
  hipLaunchKernelGGL(k_evid_finalize, dim3(1), dim3(1024), 0, s, (const double*)ws, (int)blocks, (int)K, (double)(N * V), p->kind, p->a_au, scal);
  return launch_status();
}

int devis_evid_crit_bwd(const devis_evid_params* p, const void* in, int64_t sN, int64_t sV, int64_t sK, int dtype,
                        const float* evidence_in, const void* labels, int label_dtype, int64_t N, int64_t V, int64_t K,
                        const float* scal, const float* grad_loss, const float* grad_pv, const float* grad_dice,
                        const float* grad_ev, void* d_in, float* d_evidence_in, void* stream) {
  if (!crit_args_ok(p, in, labels, label_dtype, N, V, K) || !d_in) return DEVIS_ERR_BAD_ARG;
  if (!grad_loss && !grad_pv) return DEVIS_ERR_BAD_ARG;
  const bool has_dice = p->kind == CRIT_DCE || p->kind == CRIT_DCEU;
  if (has_dice && !scal) return DEVIS_ERR_BAD_ARG;
  if ((evidence_in != nullptr) != (d_evidence_in != nullptr)) return DEVIS_ERR_BAD_ARG;
  if (evidence_in && (!p->input_kind || !has_dice)) return DEVIS_ERR_BAD_ARG;
  const EvidShape sh = {sN, sV, sK, N, V, (int)K, evid_vec(in, sN, sV, sK, K, dtype) && evid_vec(d_in, sN, sV, sK, K, dtype)};
  const EvidParams ep = {p->input_kind, p->disentangle, p->lambda, p->a_au, p->eps};
  if (p->kind == CRIT_DCE && !p->input_kind && !evidence_in && sh.vec && grad_loss && !grad_pv) {
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid2 = ev_grid2(N, V);
#define LAUNCH_DCE4B(LDT, GEV) hipLaunchKernelGGL((k_dce4_bwd<T, LDT, GEV>), grid2, dim3(256), 0, s, (const T*)in, sN, labels, N, V, p->lambda, scal, grad_loss, grad_ev, (T*)d_in)
#define LAUNCH_DCE4B_L(LDT) do { if (grad_ev) LAUNCH_DCE4B(LDT, true); else LAUNCH_DCE4B(LDT, false); } while (0)
    DEVIS_DISPATCH_DTYPE(dtype, {
      if (label_dtype == DEVIS_LABEL_I64) LAUNCH_DCE4B_L(DEVIS_LABEL_I64);
      else if (label_dtype == DEVIS_LABEL_U8) LAUNCH_DCE4B_L(DEVIS_LABEL_U8);
      else LAUNCH_DCE4B_L(DEVIS_LABEL_I32);
    });
#undef LAUNCH_DCE4B_L
#undef LAUNCH_DCE4B
    return launch_status();
  }
  DEVIS_DISPATCH_DTYPE(dtype, DEVIS_DISPATCH_KIND(p->kind, DEVIS_DISPATCH_KT(K, hipLaunchKernelGGL((k_crit_bwd<T, KIND, KT>), ev_grid2(N, V), dim3(256), 0, (hipStream_t)stream, (const T*)in, sh, evidence_in, labels, label_dtype, ep, scal, grad_loss, grad_pv, grad_dice, grad_ev, (T*)d_in, d_evidence_in))));
  return launch_status();
}

int devis_evid_loss_fwd(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype, const void* labels,
                        int label_dtype, int64_t N, int64_t V, int64_t K, float lambda, float* evidence_out,
                        float* per_voxel_out, float* scal, void* ws, size_t ws_bytes, void* stream) {
  if (!scal) return DEVIS_ERR_BAD_ARG;
  const devis_evid_params p = {CRIT_DCE, 0, 0, lambda, 0.f, 0.f};
  return devis_evid_crit_fwd(&p, logits, sN, sV, sK, dtype, nullptr, labels, label_dtype, N, V, K, evidence_out,
                             per_voxel_out, scal, ws, ws_bytes, stream);
}

int devis_evid_loss_bwd(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype, const void* labels,
                        int label_dtype, int64_t N, int64_t V, int64_t K, float lambda, const float* scal,
                        const float* grad_loss, const float* grad_pv, const float* grad_dice, const float* grad_ev,
                        void* dlogits, void* stream) {
  const devis_evid_params p = {CRIT_DCE, 0, 0, lambda, 0.f, 0.f};
  return devis_evid_crit_bwd(&p, logits, sN, sV, sK, dtype, nullptr, labels, label_dtype, N, V, K, scal, grad_loss,
                             grad_pv, grad_dice, grad_ev, dlogits, nullptr, stream);
}

// ---- the entry point for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid
// hipMalloc): devis_evid_loss_fwd with the partial sums in the file-scope `scratch` array instead of a workspace.
#define EV_MAX_DEVICES 64
static void* g_scratch_ptr[EV_MAX_DEVICES];     // device address of `scratch` on device i, resolved on first use
static std::mutex g_scratch_mu;

// -> the address of the CURRENT device's copy of `scratch` (each GPU has its own copy at its own address)
static int scratch_of_current_device(void** out) {
  int dev = -1;
  int rc = check_hip(hipGetDevice(&dev));
  if (rc) return rc;
  if (dev < 0 || dev >= EV_MAX_DEVICES) return DEVIS_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    rc = check_hip(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (rc) return rc;
    if (!ptr) { g_last_hip_error = (int)hipErrorInvalidSymbol; return DEVIS_ERR_HIP; }
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return DEVIS_OK;
}

int devis_evid_loss_fwd_noalloc(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype, const void* labels,
                                int label_dtype, int64_t N, int64_t V, int64_t K, float lambda, float* evidence_out,
                                float* per_voxel_out, float* scal, void* stream) {
  if (!scal) return DEVIS_ERR_BAD_ARG;
  // the fixed scratch holds EV_BLOCKS block partials: enough for any V, and for N <= EV_BLOCKS samples
  static_assert(DEVIS_EVID_NOALLOC_MAX_N == EV_BLOCKS, "header limit and scratch size must agree");
  if (N > DEVIS_EVID_NOALLOC_MAX_N || devis_evid_workspace_bytes(N, V, K) > EV_SCRATCH_BYTES) return DEVIS_ERR_SCRATCH_LIMIT;
  void* ptr = nullptr;
  const int rc = scratch_of_current_device(&ptr);
  if (rc) return rc;
  // Nothing in the scratch has to start at zero: every block of k_crit_fwd overwrites its own slot and
  // k_evid_finalize reads exactly the slots of this launch, so the previous call's contents are never seen.
  return devis_evid_loss_fwd(logits, sN, sV, sK, dtype, labels, label_dtype, N, V, K, lambda, evidence_out,
                             per_voxel_out, scal, ptr, EV_SCRATCH_BYTES, stream);
}

int devis_evid_infer(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype, int64_t N, int64_t V, int64_t K,
                     float* evidence_out, uint8_t* pred, float* u, float* mne, uint16_t* umask,
                     const float* thresholds_host, int n_thr, void* stream) {
  if (!logits || N <= 0 || V <= 0 || K < 2 || K > EV_MAXK || n_thr < 0 || n_thr > 16) return DEVIS_ERR_BAD_ARG;
  if (umask && n_thr > 0 && !thresholds_host) return DEVIS_ERR_BAD_ARG;
  Thr thr;
  thr.n = umask ? n_thr : 0;
  for (int j = 0; j < 16; ++j) thr.t[j] = (j < thr.n) ? thresholds_host[j] : 0.f;
  const EvidShape sh = {sN, sV, sK, N, V, (int)K, 0};
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_evid_infer<T>, dim3(ev_grid(N * V)), dim3(256), 0, (hipStream_t)stream, (const T*)logits, sh, evidence_out, pred, u, mne, umask, thr));
  return launch_status();
}

int devis_softplus_bwd(const float* z, const float* g, float* dz, int64_t n, void* stream) {
  if (!z || !g || !dz || n <= 0) return DEVIS_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_softplus_bwd, dim3(ev_grid(n)), dim3(256), 0, (hipStream_t)stream, z, g, dz, n);
  return launch_status();
}

}  // extern "C"
