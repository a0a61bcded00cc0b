// On-GPU evaluation reductions for val()/test().  The reference copies the whole
// prediction, target and uncertainty volumes to the host and counts with numpy (trainUMIS.py:286-331,446-548;
// predict.py:296-300,386-420; numpyfunctions.py:86-107).  Here ONE pass over the device-resident maps the
// inference head (k_evid_infer) left produces every count those metrics are made of:
//   conf[t][p]        voxels with target class t and predicted class p (K x K, int64)
//                     -> dice_score / mIOU / softmax_output_dice / softmax_output_litsdice are ratios of its sums
//   ucount[i][0..7]   tp, tn, fp, fn, tpu, tnu, fpu, fnu of numpyfunctions.uncertainty() for threshold i, with
//                     prediction = pred > 0, target = target > 0, thresholded_uncertainty = bit i of umask
//   sums[0..2]        sum u, sum (1-u), sum exp(-u) (fp64): mean uncertainty / conf / confe of trainUMIS.py:541-548
// HBM-bound: 1 + s_t + 2 + 4 bytes per voxel.  Integer results are exact and order-independent (integer atomics);
// the fp64 sums go through per-block partials and a fixed-order final tree (deterministic).
#include "common.h"

namespace devis {

#define EM_MAXK 8
#define EM_MAXT 16
#define EM_BLOCKS 1024

__device__ __forceinline__ int em_label(const void* t, int ldt, int64_t i) {
  if (ldt == DEVIS_LABEL_I64) return (int)((const int64_t*)t)[i];
  if (ldt == DEVIS_LABEL_U8) return (int)((const uint8_t*)t)[i];
  return ((const int32_t*)t)[i];
}

__global__ void __launch_bounds__(256)
k_eval_counts(const uint8_t* __restrict__ pred, const void* __restrict__ target, int ldt,
              const uint16_t* __restrict__ umask, int n_thr, const float* __restrict__ u, int64_t NV, int K,
              unsigned long long* __restrict__ conf, unsigned long long* __restrict__ ucount,
              double* __restrict__ psum, int* __restrict__ bad) {
  __shared__ unsigned s_conf[4][EM_MAXK * EM_MAXK];
  __shared__ unsigned s_cat[4][4];
  __shared__ unsigned s_catu[4][4 * EM_MAXT];
  __shared__ double s_sum[4][3];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 4 * EM_MAXK * EM_MAXK; i += 256) (&s_conf[0][0])[i] = 0u;
  for (int i = tid; i < 16; i += 256) (&s_cat[0][0])[i] = 0u;
  for (int i = tid; i < 4 * 4 * EM_MAXT; i += 256) (&s_catu[0][0])[i] = 0u;
  __syncthreads();
  double su = 0.0, sc = 0.0, se = 0.0;
  int oob = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < NV; i += (int64_t)gridDim.x * 256) {
    const int p = pred[i];
    const int t = em_label(target, ldt, i);
    if ((unsigned)p >= (unsigned)K || (unsigned)t >= (unsigned)K) { oob = 1; continue; }
    atomicAdd(&s_conf[wave][t * K + p], 1u);
    // numpyfunctions.py:92-95: tps = t & p, tns = ~t & ~p, fps = ~t & p, fns = t & ~p
    const int tb = t > 0, pb = p > 0;
    const int cat = tb ? (pb ? 0 : 3) : (pb ? 2 : 1);
    // one ballot per category / (category, threshold): wave-uniform popcounts, one LDS add per wave
    const unsigned m = umask ? umask[i] : 0u;
    const int leader = __ffsll((long long)__ballot(1)) - 1;        // lane 0 may have left the loop already
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned long long b = __ballot(cat == c);
      if (b == 0ull) continue;
      if (lane == leader) s_cat[wave][c] += (unsigned)__popcll(b);
      for (int j = 0; j < n_thr; ++j) {
        const unsigned long long bj = __ballot(cat == c && ((m >> j) & 1u));
        if (lane == leader && bj) s_catu[wave][c * EM_MAXT + j] += (unsigned)__popcll(bj);
      }
    }
    if (u) {
      const float uu = u[i];
      su += (double)uu;
      sc += (double)(1.f - uu);
      se += (double)expf(-uu);
    }
  }
  su = wave_sum(su); sc = wave_sum(sc); se = wave_sum(se);
  if (lane == 0) { s_sum[wave][0] = su; s_sum[wave][1] = sc; s_sum[wave][2] = se; }
  if (oob) *bad = 1;
  __syncthreads();
  for (int i = tid; i < K * K; i += 256) {
    const unsigned v = s_conf[0][i] + s_conf[1][i] + s_conf[2][i] + s_conf[3][i];
    if (v) atomicAdd(&conf[i], (unsigned long long)v);
  }
  for (int i = tid; i < 4 * n_thr; i += 256) {
    const int c = i / n_thr, j = i % n_thr;
    const unsigned all = s_cat[0][c] + s_cat[1][c] + s_cat[2][c] + s_cat[3][c];
    const unsigned un = s_catu[0][c * EM_MAXT + j] + s_catu[1][c * EM_MAXT + j] + s_catu[2][c * EM_MAXT + j] + s_catu[3][c * EM_MAXT + j];
    if (all) atomicAdd(&ucount[j * 8 + c], (unsigned long long)all);
    if (un) atomicAdd(&ucount[j * 8 + 4 + c], (unsigned long long)un);
  }
  if (tid < 3) psum[blockIdx.x * 3 + tid] = ((s_sum[0][tid] + s_sum[1][tid]) + s_sum[2][tid]) + s_sum[3][tid];
}

// sums[j] = sum_b psum[b][j], fixed order
__global__ void __launch_bounds__(64)
k_eval_final(const double* __restrict__ psum, int nblocks, double* __restrict__ sums) {
  const int lane = threadIdx.x;
  for (int j = 0; j < 3; ++j) {
    double a = 0.0;
    for (int b = lane; b < nblocks; b += 64) a += psum[b * 3 + j];
    a = wave_sum(a);
    if (lane == 0) sums[j] = a;
  }
}

}  // namespace devis

using namespace devis;

extern "C" {

size_t devis_eval_counts_workspace_bytes(int64_t NV) {
  if (NV <= 0) return 0;
  return (size_t)EM_BLOCKS * 3 * sizeof(double) + 16;
}

int devis_eval_counts(const uint8_t* pred, const void* target, int target_dtype, const uint16_t* umask, int n_thr,
                      const float* u, int64_t NV, int K, int64_t* conf, int64_t* ucount, double* sums, int* bad_label,
                      void* ws, size_t ws_bytes, void* stream) {
  if (!pred || !target || !conf || !bad_label || NV <= 0 || K < 1 || K > EM_MAXK || n_thr < 0 || n_thr > EM_MAXT) return DEVIS_ERR_BAD_ARG;
  if (target_dtype != DEVIS_LABEL_I64 && target_dtype != DEVIS_LABEL_U8 && target_dtype != DEVIS_LABEL_I32) return DEVIS_ERR_BAD_ARG;
  if (n_thr > 0 && (!umask || !ucount)) return DEVIS_ERR_BAD_ARG;
  if ((u != nullptr) != (sums != nullptr)) return DEVIS_ERR_BAD_ARG;
  if (!ws || ws_bytes < devis_eval_counts_workspace_bytes(NV)) return DEVIS_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  // the counters are accumulated with integer atomics: start from zero (memset, no allocation)
  int rc = check_hip(hipMemsetAsync(conf, 0, (size_t)K * K * sizeof(int64_t), s));
  if (rc) return rc;
  if (n_thr > 0) { rc = check_hip(hipMemsetAsync(ucount, 0, (size_t)n_thr * 8 * sizeof(int64_t), s)); if (rc) return rc; }
  rc = check_hip(hipMemsetAsync(bad_label, 0, sizeof(int), s));
  if (rc) return rc;
  int64_t blocks = cdiv(NV, 256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > EM_BLOCKS) blocks = EM_BLOCKS;
  hipLaunchKernelGGL(k_eval_counts, dim3((unsigned)blocks), dim3(256), 0, s, pred, target, target_dtype, umask, n_thr, u, NV, K,
                     (unsigned long long*)conf, (unsigned long long*)ucount, (double*)ws, bad_label);
  rc = launch_status();
  if (rc) return rc;
  if (sums) {
    hipLaunchKernelGGL(k_eval_final, dim3(1), dim3(64), 0, s, (const double*)ws, (int)blocks, sums);
    rc = launch_status();
  }
  return rc;
}

}  // extern "C"
