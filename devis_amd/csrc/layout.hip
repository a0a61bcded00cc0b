// Layout conversion between the reference's NC(D)HW fp32 tensors and the
// channels-last tensors the kernels work on; weight packing.
#include "common.h"

namespace devis {

thread_local int g_last_hip_error = 0;

template <typename T>
__global__ void k_ncx_to_nxc(const float* __restrict__ src, T* __restrict__ dst, int64_t C, int64_t V) {
  const int64_t n = blockIdx.y;
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const float* s = src + n * C * V + v;
  T* d = dst + (n * V + v) * C;
  for (int64_t c = 0; c < C; ++c) st(d + c, s[c * V]);
}

// the 4-channel input of the 3-D nets in bf16: one 8-byte store per voxel instead of four 2-byte ones
__global__ void k_ncx_to_nxc_c4_bf16(const float* __restrict__ src, bf16* __restrict__ dst, int64_t V) {
  const int64_t n = blockIdx.y;
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const float* s = src + n * 4 * V + v;
  uint2 o;
  o.x = pack_bf16x2(s[0], s[V]);
  o.y = pack_bf16x2(s[2 * V], s[3 * V]);
  *reinterpret_cast<uint2*>(dst + (n * V + v) * 4) = o;
}

template <typename T>
__global__ void k_nxc_to_ncx(const T* __restrict__ src, float* __restrict__ dst, int64_t C, int64_t V) {
  const int64_t n = blockIdx.y;
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const T* s = src + (n * V + v) * C;
  float* d = dst + n * C * V + v;
  for (int64_t c = 0; c < C; ++c) d[c * V] = ld(s + c);
}

// ---- fragment-major layout of the deep 3x3x3 layers (bf16, 27 taps, Cin == Cout in {64, 128}) ----
// conv_deepk.hip keeps a wave's weights in registers as MFMA A fragments; read from the canonical [row][tap][col]
// layout each fragment load touches 16 rows x 64 bytes (measured: 67 cycles per wave-instruction, 10 us per launch
// before the first MFMA).  For these layers every packing call therefore writes fragment order instead: element
// (row, tap, col) of the canonical array sits at
//   (((((row / 32) * (C / 32) + col / 32) * 27 + tap) * 2 + (row / 16) % 2) * 64 + ((col / 8) % 4) * 16 + row % 16) * 8 + col % 8
// i.e. [cout group of 32][32-channel K chunk][tap][cout tile of 16][lane][8 channels]: a wave's 55 KB are contiguous and
// every fragment load is one 1 KB run.  Only conv3_deepk consumes such weights (devis_conv_fwd / _dgrad refuse any other
// route for them); include/devis_hip.h states the rule for callers that pack weights themselves.
__host__ __device__ inline bool deepk_layout(int64_t rows, int64_t cols, int64_t taps, int elem_bytes) {
  return elem_bytes == 2 && taps == 27 && rows == cols && (rows == 64 || rows == 128);
}
__device__ __forceinline__ int64_t deepk_index(int64_t row, int64_t t, int64_t col, int64_t C) {
  return (((((row >> 5) * (C >> 5) + (col >> 5)) * 27 + t) * 2 + ((row >> 4) & 1)) * 64 + ((col >> 3) & 3) * 16 + (row & 15)) * 8 + (col & 7);
}

// dst[a][t][b] = src[a][b][t]
template <typename T>
__global__ void k_pack_abt(const float* __restrict__ src, T* __restrict__ dst, int64_t A, int64_t B, int64_t Tn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A * B * Tn) return;
  const int64_t b = i % B, t = (i / B) % Tn, a = i / (B * Tn);
  const int64_t o = deepk_layout(A, B, Tn, sizeof(T)) ? deepk_index(a, t, b, A) : i;
  st(dst + o, src[(a * B + b) * Tn + t]);
}
// dst[t][co][ci] = src[ci][co][t]
template <typename T>
__global__ void k_pack_convT(const float* __restrict__ src, T* __restrict__ dst, int64_t Cin, int64_t Cout, int64_t Tn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cin * Cout * Tn) return;
  const int64_t ci = i % Cin, co = (i / Cin) % Cout, t = i / (Cin * Cout);
  st(dst + i, src[(ci * Cout + co) * Tn + t]);
}

// dst[ci][T-1-t][co] = src[co][t][ci]   (forward-packed -> dgrad-packed, taps flipped)
template <typename T>
__global__ void k_repack_dgrad(const T* __restrict__ src, T* __restrict__ dst, int64_t Cout, int64_t Cin, int64_t Tn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * Cin * Tn) return;
  const int64_t co = i % Cout, tf = (i / Cout) % Tn, ci = i / (Cout * Tn);
  if (deepk_layout(Cout, Cin, Tn, sizeof(T))) dst[deepk_index(ci, tf, co, Cin)] = src[deepk_index(co, Tn - 1 - tf, ci, Cin)];
  else dst[i] = src[(co * Tn + (Tn - 1 - tf)) * Cin + ci];
}

// All weights of a model in one launch.  A chunk is (item, start):
//  * item without DEVIS_PACK_ROW_FLAG: the block handles DEVIS_PACK_CHUNK output elements of the item from element
//    `start` on, each gathered from the fp32 PyTorch tensor on its own (the transposed-conv modes, whose tap runs are
//    8 floats, and rows longer than DEVIS_PACK_ROW_MAX);
//  * item | DEVIS_PACK_ROW_FLAG (the conv modes): the block handles the whole output ROW `start` - one cout of the
//    forward layout, one cin of the dgrad layout.  Read as gathers, consecutive output elements are 4 * taps bytes apart
//    in the source (6.8x read amplification by the counters, 38-46 us per step for the 3-D U-Net's 4.8 M weights); a
//    row's sources, however, are one contiguous run (forward: [cin][tap] of that cout) or Cout runs of `taps` floats
//    (dgrad), so the block loads them coalesced into LDS and writes the row out in output order.  The LDS image keeps
//    the source's [.][tap] order: the transposing reads walk it with stride taps (27, 9: odd, no bank conflicts).
template <typename T>
__global__ void __launch_bounds__(256)
k_pack_batch(const devis_pack_item* __restrict__ items, const int2* __restrict__ chunks) {
  __shared__ float row[DEVIS_PACK_ROW_MAX];
  const int2 ch = chunks[blockIdx.x];
  const devis_pack_item it = items[ch.x & ~DEVIS_PACK_ROW_FLAG];
  T* dst = reinterpret_cast<T*>(it.dst);
  if (ch.x & DEVIS_PACK_ROW_FLAG) {
    const int R = it.rows, Cc = it.cols, Tn = it.taps, r = ch.y;
    const bool frag = deepk_layout(R, Cc, Tn, sizeof(T));       // the deep 3x3x3 layers: fragment order (above)
    // a table that breaks the rule of include/devis_hip.h (mode, row length, row index) packs nothing
    if (it.mode > DEVIS_PACK_CONV_DGRAD || r < 0 || Tn <= 0 || R <= 0 || Cc <= 0) return;
    if ((int64_t)(it.mode == DEVIS_PACK_CONV ? Cc : R) * Tn > DEVIS_PACK_ROW_MAX || r >= (it.mode == DEVIS_PACK_CONV ? R : Cc)) return;
    if (it.mode == DEVIS_PACK_CONV) {               // row r = cout: src [r][ci][t] (one run) -> dst [r][t][ci]
      const int L = Cc * Tn;
      const float* src = it.src + (int64_t)r * L;
      for (int j = threadIdx.x; j < L; j += 256) row[j] = src[j];
      __syncthreads();
      for (int j = threadIdx.x; j < L; j += 256) {
        const int t = j / Cc, ci = j - t * Cc;
        st(dst + (frag ? deepk_index(r, t, ci, R) : (int64_t)r * L + j), row[ci * Tn + t]);
      }
    } else {                                        // row r = cin: src [co][r][t] (Cout runs) -> dst [r][T-1-t][co]
      const int L = R * Tn;
      for (int j = threadIdx.x; j < L; j += 256) {
        const int co = j / Tn, t = j - co * Tn;
        row[j] = it.src[((int64_t)co * Cc + r) * Tn + t];
      }
      __syncthreads();
      for (int j = threadIdx.x; j < L; j += 256) {
        const int tf = j / R, co = j - tf * R;
        st(dst + (frag ? deepk_index(r, tf, co, R) : (int64_t)r * L + j), row[co * Tn + (Tn - 1 - tf)]);
      }
    }
    return;
  }
  const int64_t R = it.rows, Cc = it.cols, Tn = it.taps;
  const int64_t total = R * Cc * Tn;
#pragma unroll
  for (int k = 0; k < DEVIS_PACK_CHUNK / 256; ++k) {
    const int64_t i = (int64_t)ch.y + k * 256 + threadIdx.x;
    if (i >= total) return;
    int64_t si, di = i;
    const bool frag = deepk_layout(R, Cc, Tn, sizeof(T));
    if (it.mode == DEVIS_PACK_CONV) {               // src [co][ci][t] -> dst [co][t][ci]
      const int64_t ci = i % Cc, t = (i / Cc) % Tn, co = i / (Cc * Tn);
      si = (co * Cc + ci) * Tn + t;
      if (frag) di = deepk_index(co, t, ci, R);
    } else if (it.mode == DEVIS_PACK_CONV_DGRAD) {  // src [co][ci][t] -> dst [ci][T-1-t][co]
      const int64_t co = i % R, tf = (i / R) % Tn, ci = i / (R * Tn);
      si = (co * Cc + ci) * Tn + (Tn - 1 - tf);
      if (frag) di = deepk_index(ci, tf, co, R);
    } else if (it.mode == DEVIS_PACK_CONVT) {       // src [ci][co][t] -> dst [t][co][ci]
      const int64_t ci = i % R, co = (i / R) % Cc, t = i / (R * Cc);
      si = (ci * Cc + co) * Tn + t;
    } else {                                        // src [ci][co][t] -> dst [ci][t][co]
      const int64_t co = i % Cc, t = (i / Cc) % Tn, ci = i / (Cc * Tn);
      si = (ci * Cc + co) * Tn + t;
    }
    st(dst + di, it.src[si]);
  }
}

}  // namespace devis

using namespace devis;

extern "C" {

int devis_version(void) { return 101; }   // 101: row chunks in devis_pack_weights_batch
int devis_last_hip_error(void) { return g_last_hip_error; }
const char* devis_last_hip_error_string(void) { return hipGetErrorString((hipError_t)g_last_hip_error); }

int devis_ncx_to_nxc(const float* src, void* dst, int64_t N, int64_t C, int64_t V, int dtype, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || V <= 0 || N > 65535) return DEVIS_ERR_BAD_ARG;
  dim3 grid((unsigned)cdiv(V, 256), (unsigned)N);
  if (C == 4 && dtype == DEVIS_BF16) {
    hipLaunchKernelGGL(k_ncx_to_nxc_c4_bf16, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, V);
    return launch_status();
  }
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_ncx_to_nxc<T>, grid, dim3(256), 0, (hipStream_t)stream, src, (T*)dst, C, V));
  return launch_status();
}
int devis_nxc_to_ncx(const void* src, float* dst, int64_t N, int64_t C, int64_t V, int dtype, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || V <= 0 || N > 65535) return DEVIS_ERR_BAD_ARG;
  dim3 grid((unsigned)cdiv(V, 256), (unsigned)N);
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_nxc_to_ncx<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)src, dst, C, V));
  return launch_status();
}
int devis_pack_conv_weight(const float* w, void* dst, int64_t Cout, int64_t Cin, int64_t Tn, int dtype, void* stream) {
  if (!w || !dst || Cout <= 0 || Cin <= 0 || Tn <= 0) return DEVIS_ERR_BAD_ARG;
  const int64_t n = Cout * Cin * Tn;
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_pack_abt<T>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (T*)dst, Cout, Cin, Tn));
  return launch_status();
}
int devis_pack_convT_weight(const float* w, void* dst, int64_t Cin, int64_t Cout, int64_t Tn, int dtype, void* stream) {
  if (!w || !dst || Cout <= 0 || Cin <= 0 || Tn <= 0) return DEVIS_ERR_BAD_ARG;
  const int64_t n = Cout * Cin * Tn;
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_pack_convT<T>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (T*)dst, Cin, Cout, Tn));
  return launch_status();
}

int devis_pack_weights_batch(const devis_pack_item* items_dev, const int32_t* chunks_dev, int64_t n_chunks, int dtype,
                             void* stream) {
  if (!items_dev || !chunks_dev || n_chunks <= 0 || n_chunks > 0x7fffffff) return DEVIS_ERR_BAD_ARG;
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_pack_batch<T>, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, items_dev, reinterpret_cast<const int2*>(chunks_dev)));
  return launch_status();
}

int devis_repack_conv_weight_dgrad(const void* w_packed, void* dst, int64_t Cout, int64_t Cin, int64_t Tn, int dtype, void* stream) {
  if (!w_packed || !dst || Cout <= 0 || Cin <= 0 || Tn <= 0) return DEVIS_ERR_BAD_ARG;
  const int64_t n = Cout * Cin * Tn;
  DEVIS_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_repack_dgrad<T>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)w_packed, (T*)dst, Cout, Cin, Tn));
  return launch_status();
}

}  // extern "C"
