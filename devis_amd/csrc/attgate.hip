// Attention gate of the 3-D attention U-Net (Attention_block, UNet3DZoo.py:69-100), the part behind its two
// 1x1 convolutions W_g / W_x (those run on the gather-GEMM kernels):
//
//   a[v,c] = relu(BN_g(g1)[v,c] + BN_x(x1)[v,c])          C = F_int channels
//   s[v]   = b + sum_c w[c] a[v,c]                         psi conv, ONE output channel -> kept fp32
//   p[v]   = sigmoid(BN_psi(s)[v])
//   out    = skip * p
//
// The reference materialises BN_g(g1), BN_x(x1), their sum, the ReLU, the 1-channel conv output, its BatchNorm,
// the sigmoid and the product as eight tensors (+ autograd copies).  Here: two forward kernels (s with the
// moments BN_psi needs; the gated skip) and two backward kernels, with the three BatchNorms reduced to
// per-(n,c) affine coefficients and their statistics gathered inside these kernels (partial[n][chunk][C][2]
// fp64, the layout of norm.hip, so its *_finalize_partials kernels finish them).  All HBM-bound elementwise /
// row-reduction work; activations channels-last in the compute dtype, s / dz fp32 [N][V].
#include "common.h"
#include "wgrad_reduce.h"

namespace devis {

template <typename T> struct AVec;
template <> struct AVec<float> { static constexpr int N = 4; };
template <> struct AVec<bf16> { static constexpr int N = 8; };

template <typename T> __device__ __forceinline__ void a_load(const T* p, float* out);
template <> __device__ __forceinline__ void a_load<float>(const float* p, float* out) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <> __device__ __forceinline__ void a_load<bf16>(const bf16* p, float* out) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(w[i] << 16);
    out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <typename T> __device__ __forceinline__ void a_store(T* p, const float* in);
template <> __device__ __forceinline__ void a_store<float>(float* p, const float* in) {
  *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
}
template <> __device__ __forceinline__ void a_store<bf16>(bf16* p, const float* in) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(in[0], in[1]), pack_bf16x2(in[2], in[3]), pack_bf16x2(in[4], in[5]),
                                             pack_bf16x2(in[6], in[7]));
}

#define ATT_VPB 2048          // voxels per block of the thread-per-voxel kernels (8 per thread)
#define ATT_MAXC 64           // F_int limit of the LDS coefficient tables

// block-level {sum a, sum b} -> partial[2] (fp64; fixed xor tree, four waves in fixed order)
__device__ __forceinline__ void att_block_sum2(double a, double b, double* red /* [8] */, double* pout) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[wave * 2] = a; red[wave * 2 + 1] = b; }
  __syncthreads();
  if (threadIdx.x < 2) pout[threadIdx.x] = ((red[threadIdx.x] + red[2 + threadIdx.x]) + red[4 + threadIdx.x]) + red[6 + threadIdx.x];
}

// ---- s = psi(relu(BN(g1) + BN(x1))) and the moments of s.  grid = (chunks, N), one thread per voxel.
template <typename T>
__global__ void __launch_bounds__(256)
k_att_psi_fwd(const T* __restrict__ g1, const T* __restrict__ x1, const float* __restrict__ scG,
              const float* __restrict__ shG, const float* __restrict__ scX, const float* __restrict__ shX,
              const float* __restrict__ w, const float* __restrict__ b, int64_t V, int C, float* __restrict__ s,
              double* __restrict__ partial) {
  constexpr int VN = AVec<T>::N;
  __shared__ float co[5][ATT_MAXC];
  __shared__ double red[8];
  const int64_t n = blockIdx.y;
  for (int i = threadIdx.x; i < C; i += 256) {
    co[0][i] = scG[n * C + i]; co[1][i] = shG[n * C + i]; co[2][i] = scX[n * C + i]; co[3][i] = shX[n * C + i];
    co[4][i] = w[i];
  }
  __syncthreads();
  const float bias = b[0];
  const int64_t v0 = (int64_t)blockIdx.x * ATT_VPB;
  const int64_t v1 = v0 + ATT_VPB < V ? v0 + ATT_VPB : V;
  float f0 = 0.f, f1 = 0.f;
  for (int64_t v = v0 + threadIdx.x; v < v1; v += 256) {
    const T* gp = g1 + (n * V + v) * C;
    const T* xp = x1 + (n * V + v) * C;
    float acc = bias;
    for (int c = 0; c < C; c += VN) {
      float gv[VN], xv[VN];
      a_load<T>(gp + c, gv);
      a_load<T>(xp + c, xv);
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        const float a = fmaxf(fmaf(gv[j], co[0][c + j], co[1][c + j]) + fmaf(xv[j], co[2][c + j], co[3][c + j]), 0.f);
        acc = fmaf(co[4][c + j], a, acc);
      }
    }
    s[n * V + v] = acc;
    f0 += acc;
    f1 = fmaf(acc, acc, f1);
  }
  att_block_sum2((double)f0, (double)f1, red, partial + (n * gridDim.x + blockIdx.x) * 2);
}

// ---- out = skip * sigmoid(sc*s + sh)
template <typename T>
__global__ void __launch_bounds__(256)
k_att_gate_fwd(const T* __restrict__ skip, const float* __restrict__ s, const float* __restrict__ scP,
               const float* __restrict__ shP, int64_t V, int C, T* __restrict__ out) {
  constexpr int VN = AVec<T>::N;
  const int64_t n = blockIdx.y;
  const float sc = scP[n], sh = shP[n];
  const int64_t v0 = (int64_t)blockIdx.x * ATT_VPB;
  const int64_t v1 = v0 + ATT_VPB < V ? v0 + ATT_VPB : V;
  for (int64_t v = v0 + threadIdx.x; v < v1; v += 256) {
    const float p = 1.f / (1.f + expf(-fmaf(s[n * V + v], sc, sh)));
    const T* kp = skip + (n * V + v) * C;
    T* op = out + (n * V + v) * C;
    for (int c = 0; c < C; c += VN) {
      float kv[VN];
      a_load<T>(kp + c, kv);
#pragma unroll
      for (int j = 0; j < VN; ++j) kv[j] *= p;
      a_store<T>(op + c, kv);
    }
  }
}

// ---- backward of the gate: dskip = dout*p, dz = p(1-p) * sum_c dout*skip, partial {sum dz, sum dz*s}
template <typename T>
__global__ void __launch_bounds__(256)
k_att_gate_bwd(const T* __restrict__ dout, const T* __restrict__ skip, const float* __restrict__ s,
               const float* __restrict__ scP, const float* __restrict__ shP, int64_t V, int C,
               T* __restrict__ dskip, float* __restrict__ dz, double* __restrict__ partial) {
  constexpr int VN = AVec<T>::N;
  __shared__ double red[8];
  const int64_t n = blockIdx.y;
  const float sc = scP[n], sh = shP[n];
  const int64_t v0 = (int64_t)blockIdx.x * ATT_VPB;
  const int64_t v1 = v0 + ATT_VPB < V ? v0 + ATT_VPB : V;
  float f0 = 0.f, f1 = 0.f;
  for (int64_t v = v0 + threadIdx.x; v < v1; v += 256) {
    const float sv = s[n * V + v];
    const float p = 1.f / (1.f + expf(-fmaf(sv, sc, sh)));
    const T* dp = dout + (n * V + v) * C;
    const T* kp = skip + (n * V + v) * C;
    T* op = dskip + (n * V + v) * C;
    float dot = 0.f;
    for (int c = 0; c < C; c += VN) {
      float dv[VN], kv[VN];
      a_load<T>(dp + c, dv);
      a_load<T>(kp + c, kv);
#pragma unroll
      for (int j = 0; j < VN; ++j) { dot = fmaf(dv[j], kv[j], dot); dv[j] *= p; }
      a_store<T>(op + c, dv);
    }
    const float g = dot * p * (1.f - p);
    dz[n * V + v] = g;
    f0 += g;
    f1 = fmaf(g, sv, f1);
  }
  att_block_sum2((double)f0, (double)f1, red, partial + (n * gridDim.x + blockIdx.x) * 2);
}

// ---- backward of psi conv + ReLU + the sum of the two BatchNorm outputs.  Moments-style mapping (a thread owns
// VN channels and walks rows): ds = cA*dz + cB*s + cC (BatchNorm_psi backward), dpre = w*ds*[a > 0] (written),
// per-channel sums {dpre, dpre*g1, dpre*x1} (the two BatchNorm backward moments) and ds*a (psi weight gradient),
// sum ds (psi bias gradient).  grid = (chunks, N); rows_per_block rows each.
#define ATT_RPT 8
template <typename T>
__global__ void __launch_bounds__(256)
k_att_psi_bwd(const float* __restrict__ dz, const float* __restrict__ s, const float* __restrict__ cA,
              const float* __restrict__ cB, const float* __restrict__ cC, const T* __restrict__ g1,
              const T* __restrict__ x1, const float* __restrict__ scG, const float* __restrict__ shG,
              const float* __restrict__ scX, const float* __restrict__ shX, const float* __restrict__ w, int64_t V,
              int C, int rows_per_block, T* __restrict__ dpre, double* __restrict__ momG, double* __restrict__ momX,
              float* __restrict__ wpart, float* __restrict__ bpart) {
  constexpr int VN = AVec<T>::N;
  __shared__ double red[4][16][VN][4];
  const int64_t n = blockIdx.y;
  const int lpr = C / VN;                       // host: power of two, <= 16
  const int lane = threadIdx.x % lpr, sub = threadIdx.x / lpr, nsub = 256 / lpr;
  const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
  const int c0 = lane * VN;
  const float a_ = cA[n], b_ = cB[n], c_ = cC[n];
  float sg[VN], hg[VN], sx[VN], hx[VN], wv[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) {
    sg[j] = scG[n * C + c0 + j]; hg[j] = shG[n * C + c0 + j]; sx[j] = scX[n * C + c0 + j]; hx[j] = shX[n * C + c0 + j];
    wv[j] = w[c0 + j];
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < V ? r0 + rows_per_block : V;
  double q[VN][4];
#pragma unroll
  for (int j = 0; j < VN; ++j) { q[j][0] = q[j][1] = q[j][2] = q[j][3] = 0.0; }
  double dsum = 0.0;
  for (int64_t rb = r0 + sub; rb < r1; rb += (int64_t)nsub * ATT_RPT) {
    float f[VN][4];
#pragma unroll
    for (int j = 0; j < VN; ++j) { f[j][0] = f[j][1] = f[j][2] = f[j][3] = 0.f; }
    float fs = 0.f;
#pragma unroll
    for (int k = 0; k < ATT_RPT; ++k) {
      const int64_t r = rb + (int64_t)k * nsub;
      if (r < r1) {
        const float ds = fmaf(a_, dz[n * V + r], fmaf(b_, s[n * V + r], c_));
        float gv[VN], xv[VN], dv[VN];
        a_load<T>(g1 + (n * V + r) * C + c0, gv);
        a_load<T>(x1 + (n * V + r) * C + c0, xv);
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          const float a = fmaf(gv[j], sg[j], hg[j]) + fmaf(xv[j], sx[j], hx[j]);
          const float d = a > 0.f ? wv[j] * ds : 0.f;
          dv[j] = d;
          f[j][0] += d;
          f[j][1] = fmaf(d, gv[j], f[j][1]);
          f[j][2] = fmaf(d, xv[j], f[j][2]);
          f[j][3] = fmaf(ds, fmaxf(a, 0.f), f[j][3]);
        }
        a_store<T>(dpre + (n * V + r) * C + c0, dv);
        fs += ds;
      }
    }
#pragma unroll
    for (int j = 0; j < VN; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) q[j][i] += (double)f[j][i];
    dsum += (double)fs;
  }
  for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < VN; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) q[j][i] += __shfl_xor(q[j][i], o, 64);
    dsum += __shfl_xor(dsum, o, 64);
  }
  if (wl < lpr) {
#pragma unroll
    for (int j = 0; j < VN; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][wl][j][i] = q[j][i];
  }
  __shared__ double reds[4];
  if (wl == 0) reds[wave] = dsum;          // lane 0 owns channel vector 0: every row counted once
  __syncthreads();
  if (threadIdx.x < lpr) {
    const int64_t blk = n * gridDim.x + blockIdx.x;
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      double t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = ((red[0][lane][j][i] + red[1][lane][j][i]) + red[2][lane][j][i]) + red[3][lane][j][i];
      momG[(blk * C + c0 + j) * 2 + 0] = t[0];
      momG[(blk * C + c0 + j) * 2 + 1] = t[1];
      momX[(blk * C + c0 + j) * 2 + 0] = t[0];
      momX[(blk * C + c0 + j) * 2 + 1] = t[2];
      wpart[blk * C + c0 + j] = (float)t[3];
    }
    if (threadIdx.x == 0) bpart[blk] = (float)(((reds[0] + reds[1]) + reds[2]) + reds[3]);
  }
}

static inline bool att_c_ok(int64_t C, int dtype) {
  const int vn = dtype == DEVIS_F32 ? 4 : 8;
  if (C <= 0 || C % vn || C > ATT_MAXC) return false;
  const int64_t lpr = C / vn;
  return (lpr & (lpr - 1)) == 0 && lpr <= 16;
}
static inline int64_t att_psi_bwd_rows(int64_t C, int dtype) {
  const int vn = dtype == DEVIS_F32 ? 4 : 8;
  return (256 / (C / vn)) * ATT_RPT * 2;
}

}  // namespace devis

using namespace devis;

extern "C" {

int64_t devis_att_chunks(int64_t V) { return V > 0 ? cdiv(V, ATT_VPB) : 0; }

int64_t devis_att_psi_bwd_chunks(int64_t V, int64_t C, int dtype) {
  if (V <= 0 || !att_c_ok(C, dtype)) return 0;
  return cdiv(V, att_psi_bwd_rows(C, dtype));
}

int devis_att_psi_fwd(const void* g1, const void* x1, const float* scG, const float* shG, const float* scX,
                      const float* shX, const float* w_psi, const float* b_psi, int64_t N, int64_t V, int64_t C,
                      int dtype, float* s, double* partials, size_t partials_bytes, void* stream) {
  if (!g1 || !x1 || !scG || !shG || !scX || !shX || !w_psi || !b_psi || !s || !partials || N <= 0 || N > 65535 || V <= 0) return DEVIS_ERR_BAD_ARG;
  if (!att_c_ok(C, dtype)) return DEVIS_ERR_UNSUPPORTED;
  const int64_t chunks = devis_att_chunks(V);
  if (chunks > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  if (partials_bytes < (size_t)(N * chunks * 2) * sizeof(double)) return DEVIS_ERR_WORKSPACE;
  const dim3 grid((unsigned)chunks, (unsigned)N);
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_att_psi_fwd<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)g1, (const T*)x1, scG, shG, scX, shX, w_psi, b_psi, V, (int)C, s, partials));
  return launch_status();
}

int devis_att_gate_fwd(const void* skip, const float* s, const float* scP, const float* shP, void* out, int64_t N,
                       int64_t V, int64_t C, int dtype, void* stream) {
  if (!skip || !s || !scP || !shP || !out || N <= 0 || N > 65535 || V <= 0 || C <= 0) return DEVIS_ERR_BAD_ARG;
  if (C % (dtype == DEVIS_F32 ? 4 : 8)) return DEVIS_ERR_UNSUPPORTED;
  const int64_t chunks = devis_att_chunks(V);
  if (chunks > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)chunks, (unsigned)N);
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_att_gate_fwd<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)skip, s, scP, shP, V, (int)C, (T*)out));
  return launch_status();
}

int devis_att_gate_bwd(const void* dout, const void* skip, const float* s, const float* scP, const float* shP,
                       void* dskip, float* dz, double* partials, size_t partials_bytes, int64_t N, int64_t V,
                       int64_t C, int dtype, void* stream) {
  if (!dout || !skip || !s || !scP || !shP || !dskip || !dz || !partials || N <= 0 || N > 65535 || V <= 0 || C <= 0) return DEVIS_ERR_BAD_ARG;
  if (C % (dtype == DEVIS_F32 ? 4 : 8)) return DEVIS_ERR_UNSUPPORTED;
  const int64_t chunks = devis_att_chunks(V);
  if (chunks > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  if (partials_bytes < (size_t)(N * chunks * 2) * sizeof(double)) return DEVIS_ERR_WORKSPACE;
  const dim3 grid((unsigned)chunks, (unsigned)N);
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_att_gate_bwd<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)dout, (const T*)skip, s, scP, shP, V, (int)C, (T*)dskip, dz, partials));
  return launch_status();
}

size_t devis_att_psi_bwd_workspace_bytes(int64_t N, int64_t V, int64_t C, int dtype) {
  const int64_t chunks = devis_att_psi_bwd_chunks(V, C, dtype);
  if (chunks <= 0 || N <= 0) return 0;
  return (size_t)(N * chunks * (C + 1)) * sizeof(float) + 64;
}

int devis_att_psi_bwd(const float* dz, const float* s, const float* cA, const float* cB, const float* cC,
                      const void* g1, const void* x1, const float* scG, const float* shG, const float* scX,
                      const float* shX, const float* w_psi, int64_t N, int64_t V, int64_t C, int dtype, void* dpre,
                      double* momG, double* momX, size_t mom_bytes, float* dw_psi, float* db_psi, void* ws,
                      size_t ws_bytes, void* stream) {
  if (!dz || !s || !cA || !cB || !cC || !g1 || !x1 || !scG || !shG || !scX || !shX || !w_psi || !dpre || !momG || !momX || !dw_psi || !db_psi || !ws) return DEVIS_ERR_BAD_ARG;
  if (N <= 0 || N > 65535 || V <= 0) return DEVIS_ERR_BAD_ARG;
  if (!att_c_ok(C, dtype)) return DEVIS_ERR_UNSUPPORTED;
  const int64_t chunks = devis_att_psi_bwd_chunks(V, C, dtype);
  if (chunks > 0x7fffffff || N * chunks > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  if (mom_bytes < (size_t)(N * chunks * C * 2) * sizeof(double)) return DEVIS_ERR_WORKSPACE;
  if (ws_bytes < devis_att_psi_bwd_workspace_bytes(N, V, C, dtype)) return DEVIS_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* wpart = (float*)ws;
  float* bpart = wpart + N * chunks * C;
  const dim3 grid((unsigned)chunks, (unsigned)N);
  const int rows = (int)att_psi_bwd_rows(C, dtype);
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_att_psi_bwd<T>, grid, dim3(256), 0, st, dz, s, cA, cB, cC, (const T*)g1, (const T*)x1, scG, shG, scX, shX, w_psi, V, (int)C, rows, (T*)dpre, momG, momX, wpart, bpart));
  int rc = launch_status();
  if (rc) return rc;
  // dw_psi[c] = sum_blocks wpart[blk][c]; db_psi = sum_blocks bpart[blk]   (fixed-order reduction, wgrad_reduce.h)
  RedArgs r;
  r.part = wpart; r.bpart = bpart; r.dw = dw_psi; r.dbias = db_psi;
  r.nparts = (int)(N * chunks); r.nbparts = r.nparts; r.Cb = 1;
  r.pstride = C; r.total = C;
  r.mode = RED_PW; r.A = 1; r.B = (int)C; r.cin_real = 0;
  return launch_reduce_partials(r, st);
}

}  // extern "C"
