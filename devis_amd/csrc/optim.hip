// One-launch Adam over flat fp32 buffers: the optimizer step either side of the hot path
// (trainUMIS.py:1394 `optim.Adam(model.parameters(), lr, weight_decay=1e-5)`).
// HBM-bound: 16 B read + 12 B written per parameter, float4 per lane, grid-stride.
// Arithmetic order is torch's single-tensor Adam (coupled L2 decay, lerp for exp_avg, addcmul for
// exp_avg_sq, sqrt / bias_correction2_sqrt + eps, addcdiv), so the result agrees with torch.optim.Adam
// to the last bit or two of fp32.
#include "common.h"

namespace devis {

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float wd, one_minus_b1, b2, one_minus_b2, bc2_sqrt, eps, step_size;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a) {
  g = fmaf(a.wd, p, g);                           // grad.add(param, alpha=weight_decay)
  m = fmaf(a.one_minus_b1, g - m, m);             // exp_avg.lerp_(grad, 1 - beta1)
  v = fmaf(a.one_minus_b2 * g, g, v * a.b2);      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = fmaf(-a.step_size, m / denom, p);           // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ void __launch_bounds__(256)
k_adam(AdamArgs a) {
  const int64_t n4 = a.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 m = reinterpret_cast<float4*>(a.m)[i];
    float4 v = reinterpret_cast<float4*>(a.v)[i];
    adam1(p.x, g.x, m.x, v.x, a);
    adam1(p.y, g.y, m.y, v.y, a);
    adam1(p.z, g.z, m.z, v.z, a);
    adam1(p.w, g.w, m.w, v.w, a);
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
  }
  // tail (n % 4 elements): first lanes of block 0
  const int64_t t = (n4 << 2) + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 4 && t < a.n) {
    float p = a.p[t], m = a.m[t], v = a.v[t];
    adam1(p, a.g[t], m, v, a);
    a.p[t] = p; a.m[t] = m; a.v[t] = v;
  }
}

}  // namespace devis

using namespace devis;

extern "C" int devis_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                               double beta2, double eps, double weight_decay, int64_t step, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return DEVIS_ERR_BAD_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return DEVIS_ERR_BAD_ARG;
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
  // scalars as torch derives them in Python doubles (torch/optim/adam.py: _single_tensor_adam)
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  a.wd = (float)weight_decay;
  a.one_minus_b1 = (float)(1.0 - beta1);
  a.b2 = (float)beta2;
  a.one_minus_b2 = (float)(1.0 - beta2);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  a.step_size = (float)(lr / bc1);
  int64_t blocks = cdiv(n >> 2, 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}
