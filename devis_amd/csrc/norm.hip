// GroupNorm / BatchNorm as "per-(n,c) moments -> per-(n,c) affine": one HBM-bound
// reduction pass (fp64 accumulation, deterministic two-stage), a tiny finalize, and
// an elementwise apply that is normally fused into the consuming conv's prologue.
// Also: residual add, MaxPool2d.
#include "common.h"

namespace devis {

// rows (voxels) per block of the moments kernels: 16 rows per thread
static inline int64_t mom_rows(int64_t C, int vn) {
  int64_t lpr = C / vn;
  if (lpr > 256) lpr = 256;
  if (lpr < 1) lpr = 1;
  return 16 * (256 / lpr);
}

template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int N = 4; typedef float4 raw; };
template <> struct Vec<bf16> { static constexpr int N = 8; typedef uint4 raw; };

template <typename T> __device__ __forceinline__ void load_vec(const T* p, float* out);
template <> __device__ __forceinline__ void load_vec<float>(const float* p, float* out) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <> __device__ __forceinline__ void load_vec<bf16>(const bf16* p, float* out) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(w[i] << 16);
    out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float* in);
template <> __device__ __forceinline__ void store_vec<float>(float* p, const float* in) {
  *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
}
template <> __device__ __forceinline__ void store_vec<bf16>(bf16* p, const float* in) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(in[0], in[1]), pack_bf16x2(in[2], in[3]), pack_bf16x2(in[4], in[5]),
                                             pack_bf16x2(in[6], in[7]));
}

// MODE 0: {sum x, sum x^2};  MODE 1: dy' = dy*[sc*x+sh>0] -> {sum dy', sum dy'*x}
// grid = (chunks, N); partial[n][chunk][C][2].  A thread sums MOM_RPT rows in fp32 (all loads issued up
// front), the wave combines lanes that share channels with an fp64 xor tree, the four waves combine through
// LDS in a fixed order: deterministic, fp64 from the 16th addend on.
#define MOM_RPT 16
template <typename T, int MODE>
__global__ void __launch_bounds__(256)
k_moments(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ sc,
          const float* __restrict__ sh, int64_t V, int64_t C, int rows_per_block, double* __restrict__ partial) {
  constexpr int VN = Vec<T>::N;
  __shared__ double red[4 * 64 * 2 * VN];
  const int64_t n = blockIdx.y;
  const int lanes_per_row = (int)(C / VN);          // host: power of two
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < V ? r0 + rows_per_block : V;
  double* pout = partial + ((n * gridDim.x + blockIdx.x) * C) * 2;
  const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
  for (int cbase = 0; cbase < lanes_per_row; cbase += 256) {
    const int lpr = lanes_per_row < 256 ? lanes_per_row : 256;
    const int lane = threadIdx.x % lpr;
    const int sub = threadIdx.x / lpr;
    const int nsub = 256 / lpr;
    const int64_t c0 = (int64_t)(cbase + lane) * VN;
    float f0[VN], f1[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) { f0[j] = 0.f; f1[j] = 0.f; }
    float scv[VN], shv[VN];
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < VN; ++j) { scv[j] = sc[n * C + c0 + j]; shv[j] = sh[n * C + c0 + j]; }
    }
    double s0[VN], s1[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) { s0[j] = 0.0; s1[j] = 0.0; }
    for (int64_t rb = r0 + sub; rb < r1; rb += (int64_t)nsub * MOM_RPT) {
#pragma unroll
      for (int k = 0; k < MOM_RPT; ++k) {
        const int64_t r = rb + (int64_t)k * nsub;
        if (r < r1) {
          float xv[VN];
          load_vec<T>(x + (n * V + r) * C + c0, xv);
          if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < VN; ++j) { f0[j] += xv[j]; f1[j] = fmaf(xv[j], xv[j], f1[j]); }
          } else {
            float gv[VN];
            load_vec<T>(dy + (n * V + r) * C + c0, gv);
#pragma unroll
            for (int j = 0; j < VN; ++j) {
              const float g = fmaf(xv[j], scv[j], shv[j]) > 0.f ? gv[j] : 0.f;
              f0[j] += g;
              f1[j] = fmaf(g, xv[j], f1[j]);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < VN; ++j) { s0[j] += (double)f0[j]; s1[j] += (double)f1[j]; f0[j] = 0.f; f1[j] = 0.f; }
    }
    for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < VN; ++j) { s0[j] += __shfl_xor(s0[j], o, 64); s1[j] += __shfl_xor(s1[j], o, 64); }
    }
    __syncthreads();
    const int hl = lpr < 64 ? lpr : 64;               // lanes of a wave that hold distinct channel groups
    if (wl < hl) {
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        red[((wave * 64 + wl) * VN + j) * 2 + 0] = s0[j];
        red[((wave * 64 + wl) * VN + j) * 2 + 1] = s1[j];
      }
    }
    __syncthreads();
    if (sub == 0) {
      // waves holding this thread's channels: all 4 (lpr <= 64), waves w and w+2 (lpr = 128), own wave only (256)
      const int wstep = lpr <= 64 ? 1 : lpr / 64;
      const int w0 = lpr <= 64 ? 0 : wave;
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        double a = 0.0, b = 0.0;
        for (int w = w0; w < 4; w += wstep) {
          a += red[((w * 64 + (lane & 63)) * VN + j) * 2 + 0];
          b += red[((w * 64 + (lane & 63)) * VN + j) * 2 + 1];
        }
        pout[(c0 + j) * 2 + 0] = a;
        pout[(c0 + j) * 2 + 1] = b;
      }
    }
  }
}

// out[n][c][j] = sum_chunk partial[n][chunk][c][j]
__global__ void k_moments_final(const double* __restrict__ partial, int64_t chunks, int64_t C, int64_t total, double* __restrict__ out) {
  // 32 lanes per output (chunk = sub, sub+32, ...), combined with a fixed xor tree: deterministic
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;   // over N*C*2
  const int sub = threadIdx.x & 31;
  double a = 0.0;
  if (i < total) {
    const int64_t n = i / (C * 2), rem = i % (C * 2);
    const double* p = partial + n * chunks * C * 2 + rem;
    for (int64_t k = sub; k < chunks; k += 32) a += p[k * C * 2];
  }
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) a += __shfl_xor(a, o, 64);
  if (i < total && sub == 0) out[i] = a;
}

__global__ void k_gn_finalize(const double* __restrict__ mom, int64_t N, int64_t V, int64_t C, int64_t G,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                              float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean,
                              float* __restrict__ rstd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over N*G
  if (i >= N * G) return;
  const int64_t n = i / G, g = i % G, cpg = C / G;
  double s = 0.0, ss = 0.0;
  for (int64_t c = g * cpg; c < (g + 1) * cpg; ++c) { s += mom[(n * C + c) * 2]; ss += mom[(n * C + c) * 2 + 1]; }
  const double M = (double)cpg * (double)V;
  const double mu = s / M;
  double var = ss / M - mu * mu;
  if (var < 0.0) var = 0.0;
  const float r = 1.0f / sqrtf((float)var + eps);
  mean[i] = (float)mu;
  rstd[i] = r;
  for (int64_t c = g * cpg; c < (g + 1) * cpg; ++c) {
    const float a = gamma[c] * r;
    scale[n * C + c] = a;
    shift[n * C + c] = beta[c] - (float)mu * a;
  }
}

__global__ void k_bn_finalize(const double* __restrict__ mom, int64_t N, int64_t V, int64_t C,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                              float momentum, int training, float* __restrict__ rmean, float* __restrict__ rvar,
                              float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean,
                              float* __restrict__ rstd) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mu, r;
  if (training) {
    double s = 0.0, ss = 0.0;
    for (int64_t n = 0; n < N; ++n) { s += mom[(n * C + c) * 2]; ss += mom[(n * C + c) * 2 + 1]; }
    const double M = (double)N * (double)V;
    const double m = s / M;
    double var = ss / M - m * m;
    if (var < 0.0) var = 0.0;
    mu = (float)m;
    r = 1.0f / sqrtf((float)var + eps);
    if (rmean) {
      const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
  } else {
    mu = rmean[c];
    r = 1.0f / sqrtf(rvar[c] + eps);
  }
  mean[c] = mu;
  rstd[c] = r;
  const float a = gamma[c] * r, b = beta[c] - mu * a;
  for (int64_t n = 0; n < N; ++n) { scale[n * C + c] = a; shift[n * C + c] = b; }
}

// one thread per channel c; loops over n (and over the group's channels for GN)
__global__ void k_norm_bwd_finalize(const double* __restrict__ mom, int64_t N, int64_t V, int64_t C, int64_t G, int bn,
                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, float* __restrict__ cA, float* __restrict__ cB,
                                    float* __restrict__ cC, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double dg = 0.0, db = 0.0;
  if (bn) {
    double m0 = 0.0, m1 = 0.0;
    for (int64_t n = 0; n < N; ++n) { m0 += mom[(n * C + c) * 2]; m1 += mom[(n * C + c) * 2 + 1]; }
    const double mu = mean[c], r = rstd[c], gm = gamma[c];
    const double M = (double)N * (double)V;
    const double s1 = gm * m0, sxy = gm * m1;
    const double s2 = r * (sxy - mu * s1);
    // bn == 2: BatchNorm normalising with its RUNNING statistics (eval mode): mean and rstd are constants of x
    const double B = bn == 2 ? 0.0 : -r * r * s2 / M;
    const double Cc = bn == 2 ? 0.0 : -r * s1 / M - mu * B;
    for (int64_t n = 0; n < N; ++n) { cA[n * C + c] = (float)(gm * r); cB[n * C + c] = (float)B; cC[n * C + c] = (float)Cc; }
    dg = r * (m1 - mu * m0);
    db = m0;
  } else {
    const int64_t cpg = C / G, g = c / cpg;
    const double M = (double)cpg * (double)V;
    for (int64_t n = 0; n < N; ++n) {
      double s1 = 0.0, sxy = 0.0;
      for (int64_t k = g * cpg; k < (g + 1) * cpg; ++k) {
        s1 += (double)gamma[k] * mom[(n * C + k) * 2];
        sxy += (double)gamma[k] * mom[(n * C + k) * 2 + 1];
      }
      const double mu = mean[n * G + g], r = rstd[n * G + g];
      const double s2 = r * (sxy - mu * s1);
      const double B = -r * r * s2 / M;
      cA[n * C + c] = (float)((double)gamma[c] * r);
      cB[n * C + c] = (float)B;
      cC[n * C + c] = (float)(-r * s1 / M - mu * B);
      const double m0 = mom[(n * C + c) * 2], m1 = mom[(n * C + c) * 2 + 1];
      dg += r * (m1 - mu * m0);
      db += m0;
    }
  }
  dgamma[c] = (float)dg;
  dbeta[c] = (float)db;
}

// ------------------------------------------------- finalize straight from the per-chunk partials
// The kernels below fold k_moments_final into the finalize step (one launch instead of two on the serial
// path of every norm layer).  reduce_chunks() adds the chunk partials of `nout` consecutive doubles starting at
// partial[n][.][c0][0], for nn samples from n0, into sm[(n - n0) * nout + o].
__device__ __forceinline__ void reduce_chunks(const double* __restrict__ partial, int64_t chunks, int64_t C, int64_t n0,
                                              int nn, int64_t c0, int nout, double* __restrict__ sm, int fast) {
  const int64_t st = C * 2;
  if (fast) {
    // nout is a power of two <= 64 and the block has 1024 threads: lanes run along the nout contiguous doubles
    // of a chunk row first (whole 32-byte sectors per wave load), then along chunks; eight independent loads in
    // flight per lane; equal-j lanes combine in a fixed xor tree, the 16 waves through LDS in wave order.
    const int t = threadIdx.x, nw = (int)(blockDim.x >> 6), lane = t & 63, wave = t >> 6;
    const int j = t & (nout - 1), r = t / nout, RT = (int)blockDim.x / nout;
    double* wsum = sm + nn * nout;                     // [nn][nw][nout]
    if (nn == 2) {
      // the batch of the BraTS step: both samples' loads are issued before either sum waits (one memory round trip
      // instead of two; same addends, same order per sample)
      const double* p0 = partial + (n0 * chunks * C + c0) * 2 + j;
      const double* p1 = p0 + chunks * C * 2;
      double a[2][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[0][u] = 0.0; a[1][u] = 0.0; }
      int64_t k = r;
      for (; k + 7 * RT < chunks; k += 8 * RT) {
        double v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { v0[u] = p0[(k + u * RT) * st]; v1[u] = p1[(k + u * RT) * st]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[0][u] += v0[u]; a[1][u] += v1[u]; }
      }
      for (; k < chunks; k += RT) { const double v0 = p0[k * st], v1 = p1[k * st]; a[0][0] += v0; a[1][0] += v1; }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        double s = ((a[n][0] + a[n][1]) + (a[n][2] + a[n][3])) + ((a[n][4] + a[n][5]) + (a[n][6] + a[n][7]));
        for (int w = nout; w < 64; w <<= 1) s += __shfl_xor(s, w, 64);
        if (lane < nout) wsum[(n * nw + wave) * nout + lane] = s;
      }
    } else
    for (int n = 0; n < nn; ++n) {
      const double* p = partial + ((n0 + n) * chunks * C + c0) * 2 + j;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
      int64_t k = r;
      for (; k + 7 * RT < chunks; k += 8 * RT) {
        a0 += p[k * st]; a1 += p[(k + RT) * st]; a2 += p[(k + 2 * RT) * st]; a3 += p[(k + 3 * RT) * st];
        a4 += p[(k + 4 * RT) * st]; a5 += p[(k + 5 * RT) * st]; a6 += p[(k + 6 * RT) * st]; a7 += p[(k + 7 * RT) * st];
      }
      for (; k < chunks; k += RT) a0 += p[k * st];
      double a = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
      for (int w = nout; w < 64; w <<= 1) a += __shfl_xor(a, w, 64);
      if (lane < nout) wsum[(n * nw + wave) * nout + lane] = a;
    }
    __syncthreads();
    for (int o = t; o < nn * nout; o += (int)blockDim.x) {
      const double* q = wsum + (o / nout) * nw * nout + o % nout;
      double s = 0.0;
      for (int w = 0; w < nw; ++w) s += q[w * nout];
      sm[o] = s;
    }
    return;
  }
  // one wave per output, four independent loads in flight per lane, fixed xor tree: deterministic
  const int sub = threadIdx.x & 63;
  for (int o = threadIdx.x >> 6; o < nn * nout; o += (int)(blockDim.x >> 6)) {
    const int64_t n = n0 + o / nout;
    const double* p = partial + (n * chunks * C + c0) * 2 + o % nout;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t k = sub;
    for (; k + 192 < chunks; k += 256) {
      a0 += p[k * st]; a1 += p[(k + 64) * st]; a2 += p[(k + 128) * st]; a3 += p[(k + 192) * st];
    }
    for (; k < chunks; k += 64) a0 += p[k * st];
    double a = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int w = 1; w < 64; w <<= 1) a += __shfl_xor(a, w, 64);
    if (sub == 0) sm[o] = a;
  }
}

// block = one (n, group)
__global__ void __launch_bounds__(1024)
k_gn_finalize_p(const double* __restrict__ partial, int64_t chunks, int64_t N, int64_t V, int64_t C, int64_t G,
                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean,
                float* __restrict__ rstd, int fast) {
  extern __shared__ double sm[];
  const int64_t i = blockIdx.x, n = i / G, g = i % G, cpg = C / G;
  reduce_chunks(partial, chunks, C, n, 1, g * cpg, (int)(cpg * 2), sm, fast);
  __syncthreads();
  double s = 0.0, ss = 0.0;
  for (int64_t c = 0; c < cpg; ++c) { s += sm[c * 2]; ss += sm[c * 2 + 1]; }
  const double M = (double)cpg * (double)V;
  const double mu = s / M;
  double var = ss / M - mu * mu;
  if (var < 0.0) var = 0.0;
  const float r = 1.0f / sqrtf((float)var + eps);
  if (threadIdx.x == 0) { mean[i] = (float)mu; rstd[i] = r; }
  for (int64_t c = g * cpg + threadIdx.x; c < (g + 1) * cpg; c += blockDim.x) {
    const float a = gamma[c] * r;
    scale[n * C + c] = a;
    shift[n * C + c] = beta[c] - (float)mu * a;
  }
}

// block = one channel (training-mode batch statistics); also bumps num_batches_tracked
__global__ void __launch_bounds__(1024)
k_bn_finalize_p(const double* __restrict__ partial, int64_t chunks, int64_t N, int64_t V, int64_t C,
                const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ tracked,
                float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean,
                float* __restrict__ rstd, int fast) {
  extern __shared__ double sm[];
  const int64_t c = blockIdx.x;
  reduce_chunks(partial, chunks, C, 0, (int)N, c, 2, sm, fast);
  __syncthreads();
  double s = 0.0, ss = 0.0;
  for (int64_t n = 0; n < N; ++n) { s += sm[n * 2]; ss += sm[n * 2 + 1]; }
  const double M = (double)N * (double)V;
  const double m = s / M;
  double var = ss / M - m * m;
  if (var < 0.0) var = 0.0;
  const float mu = (float)m;
  const float r = 1.0f / sqrtf((float)var + eps);
  if (threadIdx.x == 0) {
    if (rmean) {
      const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
    if (tracked && c == 0) tracked[0] += 1;
    mean[c] = mu;
    rstd[c] = r;
  }
  const float a = gamma[c] * r, b = beta[c] - mu * a;
  for (int64_t n = threadIdx.x; n < N; n += blockDim.x) { scale[n * C + c] = a; shift[n * C + c] = b; }
}

// block = one channel (BatchNorm) or one group (GroupNorm; all n)
__global__ void __launch_bounds__(1024)
k_norm_bwd_finalize_p(const double* __restrict__ partial, int64_t chunks, int64_t N, int64_t V, int64_t C, int64_t G,
                      int bn, const float* __restrict__ gamma, const float* __restrict__ mean,
                      const float* __restrict__ rstd, float* __restrict__ cA, float* __restrict__ cB,
                      float* __restrict__ cC, float* __restrict__ dgamma, float* __restrict__ dbeta, int fast) {
  extern __shared__ double sm[];
  if (bn) {
    const int64_t c = blockIdx.x;
    reduce_chunks(partial, chunks, C, 0, (int)N, c, 2, sm, fast);
    __syncthreads();
    double m0 = 0.0, m1 = 0.0;
    for (int64_t n = 0; n < N; ++n) { m0 += sm[n * 2]; m1 += sm[n * 2 + 1]; }
    const double mu = mean[c], r = rstd[c], gm = gamma[c];
    const double M = (double)N * (double)V;
    const double s1 = gm * m0, sxy = gm * m1;
    const double s2 = r * (sxy - mu * s1);
    // bn == 2: BatchNorm normalising with its RUNNING statistics (eval mode): mean and rstd are constants of x
    const double B = bn == 2 ? 0.0 : -r * r * s2 / M;
    const double Cc = bn == 2 ? 0.0 : -r * s1 / M - mu * B;
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) {
      cA[n * C + c] = (float)(gm * r); cB[n * C + c] = (float)B; cC[n * C + c] = (float)Cc;
    }
    if (threadIdx.x == 0) { dgamma[c] = (float)(r * (m1 - mu * m0)); dbeta[c] = (float)m0; }
  } else {
    const int64_t g = blockIdx.x, cpg = C / G;
    reduce_chunks(partial, chunks, C, 0, (int)N, g * cpg, (int)(cpg * 2), sm, fast);
    __syncthreads();
    const double M = (double)cpg * (double)V;
    for (int64_t cl = threadIdx.x; cl < cpg; cl += blockDim.x) {
      const int64_t c = g * cpg + cl;
      double dg = 0.0, db = 0.0;
      for (int64_t n = 0; n < N; ++n) {
        const double* mm = sm + n * cpg * 2;
        double s1 = 0.0, sxy = 0.0;
        for (int64_t k = 0; k < cpg; ++k) {
          s1 += (double)gamma[g * cpg + k] * mm[k * 2];
          sxy += (double)gamma[g * cpg + k] * mm[k * 2 + 1];
        }
        const double mu = mean[n * G + g], r = rstd[n * G + g];
        const double s2 = r * (sxy - mu * s1);
        const double B = -r * r * s2 / M;
        cA[n * C + c] = (float)((double)gamma[c] * r);
        cB[n * C + c] = (float)B;
        cC[n * C + c] = (float)(-r * s1 / M - mu * B);
        const double m0 = mm[cl * 2], m1 = mm[cl * 2 + 1];
        dg += r * (m1 - mu * m0);
        db += m0;
      }
      dgamma[c] = (float)dg;
      dbeta[c] = (float)db;
    }
  }
}

// ---------------------------------------------------------------- elementwise
// Elementwise passes over a channels-last tensor with per-(n,c) coefficients.  grid = (row chunks, N); a thread
// owns ONE vector of VN channels for its whole walk, so the coefficients sit in registers (loaded once) and the
// loop body is loads, a few FMAs and a store - no per-element coefficient loads, no 64-bit divisions.
// Consecutive threads cover consecutive 16-byte pieces (rows x channel vectors): fully coalesced.
#define EW_UNROLL 4
static inline int64_t ew_rows(int64_t N, int64_t V, int64_t C, int vn) {
  int64_t lpr = C / vn;
  if (lpr > 256) lpr = 256;
  const int64_t nsub = 256 / lpr;                      // rows per block per load round
  int64_t rows = nsub * EW_UNROLL;
  while (N * cdiv(V, rows) > 4096) rows *= 2;          // ~2-4 k blocks on the big tensors
  return rows;
}

template <typename T, bool RELU, bool RES>
__global__ void __launch_bounds__(256)
k_affine_act(const T* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
             const T* __restrict__ res, T* __restrict__ y, int64_t V, int64_t C, int rows_per_block) {
  constexpr int VN = Vec<T>::N;
  const int64_t n = blockIdx.y;
  const int lanes_per_row = (int)(C / VN);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < V ? r0 + rows_per_block : V;
  for (int cbase = 0; cbase < lanes_per_row; cbase += 256) {
    const int lpr = lanes_per_row < 256 ? lanes_per_row : 256;
    const int lane = threadIdx.x % lpr, sub = threadIdx.x / lpr, nsub = 256 / lpr;
    const int64_t c0 = (int64_t)(cbase + lane) * VN;
    float scv[VN], shv[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) { scv[j] = sc[n * C + c0 + j]; shv[j] = sh[n * C + c0 + j]; }
    const T* xb = x + n * V * C + c0;
    const T* rb = RES ? res + n * V * C + c0 : nullptr;
    T* yb = y + n * V * C + c0;
    for (int64_t rb0 = r0 + sub; rb0 < r1; rb0 += (int64_t)nsub * EW_UNROLL) {
      float v[EW_UNROLL][VN], r[EW_UNROLL][VN];
#pragma unroll
      for (int k = 0; k < EW_UNROLL; ++k) {
        const int64_t row = rb0 + (int64_t)k * nsub;
        if (row < r1) {
          load_vec<T>(xb + row * C, v[k]);
          if (RES) load_vec<T>(rb + row * C, r[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < EW_UNROLL; ++k) {
        const int64_t row = rb0 + (int64_t)k * nsub;
        if (row < r1) {
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            float t = fmaf(v[k][j], scv[j], shv[j]);
            if (RELU) t = fmaxf(t, 0.f);
            if (RES) t += r[k][j];
            v[k][j] = t;
          }
          store_vec<T>(yb + row * C, v[k]);
        }
      }
    }
  }
}

template <typename T, bool EXTRA>
__global__ void __launch_bounds__(256)
k_norm_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ sc,
                 const float* __restrict__ sh, const float* __restrict__ cA, const float* __restrict__ cB,
                 const float* __restrict__ cC, const T* __restrict__ extra, T* __restrict__ dx, int64_t V,
                 int64_t C, int rows_per_block) {
  constexpr int VN = Vec<T>::N;
  const int64_t n = blockIdx.y;
  const int lanes_per_row = (int)(C / VN);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < V ? r0 + rows_per_block : V;
  for (int cbase = 0; cbase < lanes_per_row; cbase += 256) {
    const int lpr = lanes_per_row < 256 ? lanes_per_row : 256;
    const int lane = threadIdx.x % lpr, sub = threadIdx.x / lpr, nsub = 256 / lpr;
    const int64_t c0 = (int64_t)(cbase + lane) * VN;
    float scv[VN], shv[VN], av[VN], bv[VN], cv[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const int64_t k = n * C + c0 + j;
      scv[j] = sc[k]; shv[j] = sh[k]; av[j] = cA[k]; bv[j] = cB[k]; cv[j] = cC[k];
    }
    const T* gb = dy + n * V * C + c0;
    const T* xb = x + n * V * C + c0;
    const T* eb = EXTRA ? extra + n * V * C + c0 : nullptr;
    T* ob = dx + n * V * C + c0;
    for (int64_t rb0 = r0 + sub; rb0 < r1; rb0 += (int64_t)nsub * EW_UNROLL) {
      float g[EW_UNROLL][VN], xv[EW_UNROLL][VN], ex[EW_UNROLL][VN];
#pragma unroll
      for (int k = 0; k < EW_UNROLL; ++k) {
        const int64_t row = rb0 + (int64_t)k * nsub;
        if (row < r1) {
          load_vec<T>(gb + row * C, g[k]);
          load_vec<T>(xb + row * C, xv[k]);
          if (EXTRA) load_vec<T>(eb + row * C, ex[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < EW_UNROLL; ++k) {
        const int64_t row = rb0 + (int64_t)k * nsub;
        if (row < r1) {
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            float gm = fmaf(xv[k][j], scv[j], shv[j]) > 0.f ? g[k][j] : 0.f;
#ifdef DEVIS_PK_NOP      // diagnostic build only (tools/race_probe.py packed): two wait states between the
            asm volatile("s_nop 1" : "+v"(gm));      // select that writes gm and the packed FMA that reads it
#endif
            float t = fmaf(av[j], gm, fmaf(bv[j], xv[k][j], cv[j]));
            if (EXTRA) t += ex[k][j];
            g[k][j] = t;
          }
          store_vec<T>(ob + row * C, g[k]);
        }
      }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_add(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ z, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    st(z + i, ld(a + i) + ld(b + i));
}

// z = a * b (elementwise): nn.Dropout's `input * noise` with the noise tensor given (Unetdrop, UNet3DZoo.py:352-366)
template <typename T>
__global__ void __launch_bounds__(256)
k_mul(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ z, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    st(z + i, ld(a + i) * ld(b + i));
}

// -------------------------------------------------------------------- maxpool
template <typename T>
__global__ void __launch_bounds__(256)
k_maxpool_fwd(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int64_t N, int64_t H, int64_t W, int64_t C) {
  const int64_t OH = H / 2, OW = W / 2;
  const int64_t total = N * OH * OW * C;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t c = i % C;
  int64_t r = i / C;
  const int64_t ow = r % OW; r /= OW;
  const int64_t oh = r % OH;
  const int64_t n = r / OH;
  float best = 0.f;
  int bi = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // scan order (kh, kw); strictly-greater keeps the first maximum
    const float v = ld(x + ((n * H + oh * 2 + (k >> 1)) * W + ow * 2 + (k & 1)) * C + c);
    if (k == 0 || v > best || v != v) { best = v; bi = k; }
  }
  st(y + i, best);
  idx[i] = (uint8_t)bi;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_maxpool_bwd(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int64_t N, int64_t H, int64_t W, int64_t C) {
  const int64_t OH = H / 2, OW = W / 2;
  const int64_t total = N * H * W * C;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t c = i % C;
  int64_t r = i / C;
  const int64_t w = r % W; r /= W;
  const int64_t h = r % H;
  const int64_t n = r / H;
  float g = 0.f;
  const int64_t oh = h / 2, ow = w / 2;
  if (oh < OH && ow < OW) {
    const int64_t o = ((n * OH + oh) * OW + ow) * C + c;
    if (idx[o] == (uint8_t)((h & 1) * 2 + (w & 1))) g = ld(dy + o);
  }
  st(dx + i, g);
}

// bf16, C % 8 == 0: one thread per (2x2 cell, 8-channel piece), 16-byte accesses
__global__ void __launch_bounds__(256)
k_maxpool_fwd_v8(const bf16* __restrict__ x, bf16* __restrict__ y, uint8_t* __restrict__ idx, int N, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2, PC = C / 8;
  const int64_t total = (int64_t)N * OH * OW * PC;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pc = (int)(i % PC);
  int64_t r = i / PC;
  const int ow = (int)(r % OW); r /= OW;
  const int oh = (int)(r % OH);
  const int n = (int)(r / OH);
  float best[8];
  unsigned bi[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // scan order (kh, kw); strictly-greater keeps the first maximum, NaN propagates
    const uint4 t = *reinterpret_cast<const uint4*>(x + ((size_t)(n * H + oh * 2 + (k >> 1)) * W + ow * 2 + (k & 1)) * C + pc * 8);
    const unsigned w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (j & 1) ? __uint_as_float(w4[j >> 1] & 0xffff0000u) : __uint_as_float(w4[j >> 1] << 16);
      if (k == 0 || v > best[j] || v != v) { best[j] = v; bi[j] = (unsigned)k; }
    }
  }
  union { bf16 h[8]; uint4 u; } o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.h[j] = (bf16)best[j];
  const size_t ob = ((size_t)(n * OH + oh) * OW + ow) * C + pc * 8;
  *reinterpret_cast<uint4*>(y + ob) = o.u;
  uint2 ib;
  ib.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
  ib.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
  *reinterpret_cast<uint2*>(idx + ob) = ib;
}

// dx = scatter(dy by idx) (+ skip): the gradient of the pooled branch and the skip branch's gradient in one pass
__global__ void __launch_bounds__(256)
k_maxpool_bwd_v8(const bf16* __restrict__ dy, const uint8_t* __restrict__ idx, const bf16* __restrict__ skip,
                 bf16* __restrict__ dx, int N, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2, CH = (H + 1) / 2, CW = (W + 1) / 2, PC = C / 8;
  const int64_t total = (int64_t)N * CH * CW * PC;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pc = (int)(i % PC);
  int64_t r = i / PC;
  const int cw = (int)(r % CW); r /= CW;
  const int ch = (int)(r % CH);
  const int n = (int)(r / CH);
  float g[8];
  unsigned bi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { g[j] = 0.f; bi[j] = 4u; }
  if (ch < OH && cw < OW) {
    const size_t ob = ((size_t)(n * OH + ch) * OW + cw) * C + pc * 8;
    const uint4 t = *reinterpret_cast<const uint4*>(dy + ob);
    const uint2 ib = *reinterpret_cast<const uint2*>(idx + ob);
    const unsigned w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      g[j] = (j & 1) ? __uint_as_float(w4[j >> 1] & 0xffff0000u) : __uint_as_float(w4[j >> 1] << 16);
      bi[j] = ((j < 4 ? ib.x : ib.y) >> (8 * (j & 3))) & 0xffu;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int h = ch * 2 + (k >> 1), w = cw * 2 + (k & 1);
    if (h >= H || w >= W) continue;
    const size_t xb = ((size_t)(n * H + h) * W + w) * C + pc * 8;
    float sk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sk[j] = 0.f;
    if (skip) {
      const uint4 t = *reinterpret_cast<const uint4*>(skip + xb);
      const unsigned w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) sk[j] = (j & 1) ? __uint_as_float(w4[j >> 1] & 0xffff0000u) : __uint_as_float(w4[j >> 1] << 16);
    }
    union { bf16 h8[8]; uint4 u; } o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.h8[j] = (bf16)((bi[j] == (unsigned)k ? g[j] : 0.f) + sk[j]);
    *reinterpret_cast<uint4*>(dx + xb) = o.u;
  }
}

static inline bool vec_ok(int64_t C, int dtype) {
  const int vn = dtype == DEVIS_F32 ? 4 : 8;
  if (C % vn) return false;
  const int64_t lpr = C / vn;
  return (256 % lpr == 0) || (lpr % 256 == 0);
}

static inline unsigned ew_grid(int64_t nvec) {
  const int64_t b = cdiv(nvec, 256);
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace devis

using namespace devis;

extern "C" {

size_t devis_moments_workspace_bytes(int64_t N, int64_t V, int64_t C) {
  if (N <= 0 || V <= 0 || C <= 0) return 0;
  return (size_t)(N * cdiv(V, mom_rows(C, 4)) * C * 2) * sizeof(double);   // fp32 vector width: upper bound for bf16
}

// first stage only: partial[n][chunk][C][2] in `ws`; *chunks_out = chunks per sample
static int moments_partials_impl(int mode, const void* x, const void* dy, const float* sc, const float* sh, int64_t N,
                                 int64_t V, int64_t C, int dtype, void* ws, size_t ws_bytes, int64_t* chunks_out,
                                 void* stream) {
  if (!x || !ws || N <= 0 || V <= 0 || C <= 0 || N > 65535) return DEVIS_ERR_BAD_ARG;
  if (!vec_ok(C, dtype)) return DEVIS_ERR_UNSUPPORTED;
  if (ws_bytes < devis_moments_workspace_bytes(N, V, C)) return DEVIS_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  // rows per block: at least 16 per thread, more on large tensors so that about 512 blocks cover the tensor
  // (each block then leaves one partial; the second stage stays a few microseconds)
  int64_t rows = mom_rows(C, dtype == DEVIS_F32 ? 4 : 8);
  const int64_t mult = (N * V) / (rows * 512);
  if (mult > 1) rows *= mult;
  const int64_t chunks = cdiv(V, rows);
  if (chunks > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)chunks, (unsigned)N);
  DEVIS_DISPATCH_DTYPE(dtype, {
    if (mode == 0) hipLaunchKernelGGL((k_moments<T, 0>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, sc, sh, V, C, (int)rows, (double*)ws);
    else hipLaunchKernelGGL((k_moments<T, 1>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, sc, sh, V, C, (int)rows, (double*)ws);
  });
  *chunks_out = chunks;
  return launch_status();
}

static int moments_impl(int mode, const void* x, const void* dy, const float* sc, const float* sh, int64_t N,
                        int64_t V, int64_t C, int dtype, double* out, void* ws, size_t ws_bytes, void* stream) {
  if (!out) return DEVIS_ERR_BAD_ARG;
  int64_t chunks = 0;
  int rc = moments_partials_impl(mode, x, dy, sc, sh, N, V, C, dtype, ws, ws_bytes, &chunks, stream);
  if (rc) return rc;
  const int64_t total = N * C * 2;
  hipLaunchKernelGGL(k_moments_final, dim3((unsigned)cdiv(total * 32, 256)), dim3(256), 0, (hipStream_t)stream, (const double*)ws, chunks, C, total, out);
  return launch_status();
}

int devis_channel_moments_partials(const void* x, int64_t N, int64_t V, int64_t C, int dtype, void* ws,
                                   size_t ws_bytes, int64_t* chunks_host, void* stream) {
  if (!chunks_host) return DEVIS_ERR_BAD_ARG;
  return moments_partials_impl(0, x, nullptr, nullptr, nullptr, N, V, C, dtype, ws, ws_bytes, chunks_host, stream);
}

int devis_norm_bwd_moments_partials(const void* dy, const void* x, const float* sc, const float* sh, int64_t N,
                                    int64_t V, int64_t C, int dtype, void* ws, size_t ws_bytes, int64_t* chunks_host,
                                    void* stream) {
  if (!dy || !sc || !sh || !chunks_host) return DEVIS_ERR_BAD_ARG;
  return moments_partials_impl(1, x, dy, sc, sh, N, V, C, dtype, ws, ws_bytes, chunks_host, stream);
}

#define DEVIS_FINALIZE_SMEM_MAX (48 * 1024)
// one wave per output, at most 16 waves
static inline unsigned fin_threads(int64_t nout) { return (unsigned)(64 * (nout < 1 ? 1 : (nout > 16 ? 16 : nout))); }
// The sector-coalesced reduction (reduce_chunks, fast): nout a power of two <= 64, 17x the LDS, 1024 threads.
struct FinPlan { int fast; unsigned threads; size_t smem; };
static inline FinPlan fin_plan(int64_t nn, int64_t nout) {
  // block size of the sector-coalesced path: these 8..16-block kernels sit on backward's serial path while the
  // weight-gradient kernels of the side stream hold most of every CU, and a smaller block finds room sooner
  // measured in the step, GPU-bound: 1024 threads 4.565 ms, 512: 4.483, 256: 4.508, one wave per output 4.552
  const unsigned t = 512u;
  const size_t slow = (size_t)(nn * nout) * sizeof(double), quick = slow * (t / 64 + 1);
  if (nout >= 1 && nout <= 64 && (nout & (nout - 1)) == 0 && quick <= DEVIS_FINALIZE_SMEM_MAX) return {1, t, quick};
  return {0, fin_threads(nn * nout), slow};
}

int devis_gn_finalize_partials(const double* partials, int64_t chunks, int64_t N, int64_t V, int64_t C, int64_t G,
                               const float* gamma, const float* beta, float eps, float* scale, float* shift,
                               float* mean, float* rstd, void* stream) {
  if (!partials || chunks <= 0 || !gamma || !beta || !scale || !shift || !mean || !rstd || N <= 0 || V <= 0 || G <= 0 || C % G) return DEVIS_ERR_BAD_ARG;
  const FinPlan fp = fin_plan(1, (C / G) * 2);
  if (fp.smem > DEVIS_FINALIZE_SMEM_MAX || N * G > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_gn_finalize_p, dim3((unsigned)(N * G)), dim3(fp.threads), fp.smem, (hipStream_t)stream, partials, chunks, N, V, C, G, gamma, beta, eps, scale, shift, mean, rstd, fp.fast);
  return launch_status();
}

int devis_bn_finalize_partials(const double* partials, int64_t chunks, int64_t N, int64_t V, int64_t C,
                               const float* gamma, const float* beta, float eps, float momentum, float* rmean,
                               float* rvar, int64_t* num_batches_tracked, float* scale, float* shift, float* mean,
                               float* rstd, void* stream) {
  if (!partials || chunks <= 0 || !gamma || !beta || !scale || !shift || !mean || !rstd || N <= 0 || V <= 0 || C <= 0) return DEVIS_ERR_BAD_ARG;
  if ((rmean == nullptr) != (rvar == nullptr)) return DEVIS_ERR_BAD_ARG;
  const FinPlan fp = fin_plan(N, 2);
  if (fp.smem > DEVIS_FINALIZE_SMEM_MAX || C > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_bn_finalize_p, dim3((unsigned)C), dim3(fp.threads), fp.smem, (hipStream_t)stream, partials, chunks, N, V, C, gamma, beta, eps, momentum, rmean, rvar, num_batches_tracked, scale, shift, mean, rstd, fp.fast);
  return launch_status();
}

int devis_norm_bwd_finalize_partials(const double* partials, int64_t chunks, int64_t N, int64_t V, int64_t C,
                                     int64_t G, int bn, const float* gamma, const float* mean, const float* rstd,
                                     float* cA, float* cB, float* cC, float* dgamma, float* dbeta, void* stream) {
  if (!partials || chunks <= 0 || !gamma || !mean || !rstd || !cA || !cB || !cC || !dgamma || !dbeta || N <= 0 || V <= 0 || C <= 0) return DEVIS_ERR_BAD_ARG;
  if (!bn && (G <= 0 || C % G)) return DEVIS_ERR_BAD_ARG;
  const FinPlan fp = fin_plan(N, (bn ? 1 : C / G) * 2);
  if (fp.smem > DEVIS_FINALIZE_SMEM_MAX) return DEVIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_norm_bwd_finalize_p, dim3((unsigned)(bn ? C : G)), dim3(fp.threads), fp.smem, (hipStream_t)stream, partials, chunks, N, V, C, G, bn, gamma, mean, rstd, cA, cB, cC, dgamma, dbeta, fp.fast);
  return launch_status();
}

int devis_channel_moments(const void* x, int64_t N, int64_t V, int64_t C, int dtype, double* out, void* ws,
                          size_t ws_bytes, void* stream) {
  return moments_impl(0, x, nullptr, nullptr, nullptr, N, V, C, dtype, out, ws, ws_bytes, stream);
}

int devis_norm_bwd_moments(const void* dy, const void* x, const float* sc, const float* sh, int64_t N, int64_t V,
                           int64_t C, int dtype, double* out, void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !sc || !sh) return DEVIS_ERR_BAD_ARG;
  return moments_impl(1, x, dy, sc, sh, N, V, C, dtype, out, ws, ws_bytes, stream);
}

int devis_gn_finalize(const double* mom, int64_t N, int64_t V, int64_t C, int64_t G, const float* gamma,
                      const float* beta, float eps, float* scale, float* shift, float* mean, float* rstd, void* stream) {
  if (!mom || !gamma || !beta || !scale || !shift || !mean || !rstd || N <= 0 || V <= 0 || G <= 0 || C % G) return DEVIS_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_gn_finalize, dim3((unsigned)cdiv(N * G, 64)), dim3(64), 0, (hipStream_t)stream, mom, N, V, C, G, gamma, beta, eps, scale, shift, mean, rstd);
  return launch_status();
}

int devis_bn_finalize(const double* mom, int64_t N, int64_t V, int64_t C, const float* gamma, const float* beta,
                      float eps, float momentum, int training, float* rmean, float* rvar, float* scale, float* shift,
                      float* mean, float* rstd, void* stream) {
  if (!gamma || !beta || !scale || !shift || !mean || !rstd || N <= 0 || V <= 0 || C <= 0) return DEVIS_ERR_BAD_ARG;
  if (training && !mom) return DEVIS_ERR_BAD_ARG;
  if (!training && (!rmean || !rvar)) return DEVIS_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, mom, N, V, C, gamma, beta, eps, momentum, training, rmean, rvar, scale, shift, mean, rstd);
  return launch_status();
}

int devis_norm_bwd_finalize(const double* mom, int64_t N, int64_t V, int64_t C, int64_t G, int bn, const float* gamma,
                            const float* mean, const float* rstd, float* cA, float* cB, float* cC, float* dgamma,
                            float* dbeta, void* stream) {
  if (!mom || !gamma || !mean || !rstd || !cA || !cB || !cC || !dgamma || !dbeta || N <= 0 || V <= 0 || C <= 0) return DEVIS_ERR_BAD_ARG;
  if (!bn && (G <= 0 || C % G)) return DEVIS_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_norm_bwd_finalize, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, mom, N, V, C, G, bn, gamma, mean, rstd, cA, cB, cC, dgamma, dbeta);
  return launch_status();
}

int devis_affine_act(const void* x, const float* sc, const float* sh, const void* res, void* y, int64_t N, int64_t V,
                     int64_t C, int dtype, int relu, void* stream) {
  if (!x || !sc || !sh || !y || N <= 0 || V <= 0 || C <= 0 || N > 65535) return DEVIS_ERR_BAD_ARG;
  if (!vec_ok(C, dtype)) return DEVIS_ERR_UNSUPPORTED;
  const int64_t rows = ew_rows(N, V, C, dtype == DEVIS_F32 ? 4 : 8);
  if (cdiv(V, rows) > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(V, rows), (unsigned)N);
#define LAUNCH_AA(R, S) hipLaunchKernelGGL((k_affine_act<T, R, S>), grid, dim3(256), 0, s, (const T*)x, sc, sh, (const T*)res, (T*)y, V, C, (int)rows)
  DEVIS_DISPATCH_DTYPE(dtype, {
    if (relu && res) LAUNCH_AA(true, true);
    else if (relu) LAUNCH_AA(true, false);
    else if (res) LAUNCH_AA(false, true);
    else LAUNCH_AA(false, false);
  });
#undef LAUNCH_AA
  return launch_status();
}

int devis_norm_bwd_apply(const void* dy, const void* x, const float* sc, const float* sh, const float* cA,
                         const float* cB, const float* cC, const void* extra, void* dx, int64_t N, int64_t V,
                         int64_t C, int dtype, void* stream) {
  if (!dy || !x || !sc || !sh || !cA || !cB || !cC || !dx || N <= 0 || V <= 0 || C <= 0 || N > 65535) return DEVIS_ERR_BAD_ARG;
  if (!vec_ok(C, dtype)) return DEVIS_ERR_UNSUPPORTED;
  const int64_t rows = ew_rows(N, V, C, dtype == DEVIS_F32 ? 4 : 8);
  if (cdiv(V, rows) > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(V, rows), (unsigned)N);
  DEVIS_DISPATCH_DTYPE(dtype, {
    if (extra) hipLaunchKernelGGL((k_norm_bwd_apply<T, true>), grid, dim3(256), 0, s, (const T*)dy, (const T*)x, sc, sh, cA, cB, cC, (const T*)extra, (T*)dx, V, C, (int)rows);
    else hipLaunchKernelGGL((k_norm_bwd_apply<T, false>), grid, dim3(256), 0, s, (const T*)dy, (const T*)x, sc, sh, cA, cB, cC, (const T*)extra, (T*)dx, V, C, (int)rows);
  });
  return launch_status();
}

int devis_add(const void* a, const void* b, void* z, int64_t n, int dtype, void* stream) {
  if (!a || !b || !z || n <= 0) return DEVIS_ERR_BAD_ARG;
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_add<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, (T*)z, n));
  return launch_status();
}

int devis_mul(const void* a, const void* b, void* z, int64_t n, int dtype, void* stream) {
  if (!a || !b || !z || n <= 0) return DEVIS_ERR_BAD_ARG;
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_mul<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, (T*)z, n));
  return launch_status();
}

int devis_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int64_t N, int64_t H, int64_t W, int64_t C, int dtype, void* stream) {
  if (!x || !y || !idx || N <= 0 || H < 2 || W < 2 || C <= 0) return DEVIS_ERR_BAD_ARG;
  const int64_t total = N * (H / 2) * (W / 2) * C;
  if (dtype == DEVIS_BF16 && C % 8 == 0 && N * H * W < ((int64_t)1 << 31) && C < ((int64_t)1 << 31)) {
    hipLaunchKernelGGL(k_maxpool_fwd_v8, dim3((unsigned)cdiv(total / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, (bf16*)y, idx, (int)N, (int)H, (int)W, (int)C);
    return launch_status();
  }
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_maxpool_fwd<T>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, idx, N, H, W, C));
  return launch_status();
}

int devis_maxpool2d_bwd_add(const void* dy, const uint8_t* idx, const void* skip, void* dx, int64_t N, int64_t H, int64_t W,
                            int64_t C, int dtype, void* stream) {
  if (!dy || !dx || !idx || N <= 0 || H < 2 || W < 2 || C <= 0) return DEVIS_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = N * H * W * C;
  if (dtype == DEVIS_BF16 && C % 8 == 0 && N * H * W < ((int64_t)1 << 31) && C < ((int64_t)1 << 31)) {
    const int64_t cells = N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    hipLaunchKernelGGL(k_maxpool_bwd_v8, dim3((unsigned)cdiv(cells, 256)), dim3(256), 0, s, (const bf16*)dy, idx,
                       (const bf16*)skip, (bf16*)dx, (int)N, (int)H, (int)W, (int)C);
    return launch_status();
  }
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_maxpool_bwd<T>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, (const T*)dy, idx, (T*)dx, N, H, W, C));
  int rc = launch_status();
  if (rc || !skip) return rc;
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_add<T>, dim3(ew_grid(total)), dim3(256), 0, s, (const T*)dx, (const T*)skip, (T*)dx, total));
  return launch_status();
}

int devis_maxpool2d_bwd(const void* dy, const uint8_t* idx, void* dx, int64_t N, int64_t H, int64_t W, int64_t C, int dtype, void* stream) {
  return devis_maxpool2d_bwd_add(dy, idx, nullptr, dx, N, H, W, C, dtype, stream);
}

}  // extern "C"
