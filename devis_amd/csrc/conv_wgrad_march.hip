// Plane-marching MFMA weight gradient of the 16-channel, full-resolution 3x3x3 convolutions
// (bf16, channels-last, stride 1, gfx950); the companion of conv_march.hip.
//
//   dW[co][tap][ci] = sum_v dy[v][co] * A[v + off(tap)][ci],   A = relu(scale*x + shift)
//
// GEMM: M = cout, N = cin, K = voxels.  A block owns an (8 x 32) column and walks it along D with a
// ring of 4 activation planes (halo included, prologue applied) and a double-buffered dy plane in
// LDS; plane d+2 of x and plane d+1 of dy are in flight from HBM while plane d is multiplied.
// Both operands need K (voxels) per lane while memory is channels-last, so both are fetched with
// the transposing LDS read (ds_read_b64_tr_b16).  Wave w keeps the accumulators of taps w, w+4, ...
// (7 x 16x16 fp32) for the whole walk; one partial [27][16][16] per block goes to the workspace and
// k_wgrad_reduce adds the partials in a fixed order (deterministic, no atomics).
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {

int launch_wgrad_reduce_frag(const float* part, const float* bpart, int gx, int Cout, int Cin, float* dw, float* dbias,
                             hipStream_t s);
int launch_wgrad_reduce(const float* part, const float* bpart, int gx, int Cout, int Cin, int cin_real, float* dw,
                        float* dbias, hipStream_t s);

struct WgMarchArgs {
  const bf16* x;
  const bf16* dy;
  const float* sc;   // [N][16] or null
  const float* sh;
  float* part;       // [blocks][27][16][16]
  float* bpart;      // [blocks][16] or null
  int N, D, H, W;
  int nth, ntw, ndc, DL;
};

__device__ __forceinline__ bf16x8 tr_pair_m(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = a;
  u.h[1] = b;
  return u.v;
}

// ---- The waves split the VOXELS.  (The round-1 kernel, deleted in r03, had the four waves share the voxels and split the
// taps, so every wave read the whole dy tile and 7 shifted x fragments per K step: 16 transposing LDS reads per 7
// MFMAs, 32 KB of LDS traffic per 32 voxels and block - the LDS port bounded it, 55 us of LDS time at 2 x 128^3.)
// Here the waves split the VOXELS (wave w takes K steps w and w+4 of a plane's 8) and each keeps
// all 27 tap accumulators: one dy fragment feeds 27 MFMAs (28 reads per 27 MFMAs, 2.2x less LDS traffic).  The four
// waves' accumulators are added in wave order through LDS at the end of the walk (once per block).
template <bool PRO, bool CIN4>
__global__ void __launch_bounds__(256, 2)
k_wgrad3_c16_ksplit(WgMarchArgs a) {
  constexpr int TH = 8, TW = 32;
  constexpr int HH = TH + 2, HW = TW + 2, HP = HH * HW;
  constexpr int NPT = (HP * 2 + 255) / 256;
  constexpr int PLB = NPT * 256 * 16;
  constexpr int R = 4;
  constexpr int YPL = TH * TW * 32;
  constexpr int CINR = CIN4 ? 4 : 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ys = smem + R * PLB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;

  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int qq = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (bid >> 3);
  }
  const int twi = tile % a.ntw;
  const int thi = (tile / a.ntw) % a.nth;
  const int dci = (tile / (a.ntw * a.nth)) % a.ndc;
  const int n = tile / (a.ntw * a.nth * a.ndc);
  const int oh0 = thi * TH, ow0 = twi * TW;
  const int d0 = dci * a.DL;
  const int d1 = min(d0 + a.DL, a.D);
  const size_t plane = (size_t)a.H * a.W;

  unsigned goff[NPT];
  bool ok[NPT];
  const int piece = tid & 1;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int i = tid + k * 256;
    const int u = i >> 1;
    const int hh = u / HW, hw = u - hh * HW;
    const int ih = oh0 - 1 + hh, iw = ow0 - 1 + hw;
    ok[k] = i < HP * 2 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W && !(CIN4 && piece);
    goff[k] = ok[k] ? (unsigned)((ih * a.W + iw) * CINR + (CIN4 ? 0 : piece * 8)) : 0u;
  }
  unsigned yoff[2];
  bool yok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = (tid + k * 256) >> 1;
    const int oh = oh0 + v / TW, ow = ow0 + v % TW;
    yok[k] = oh < a.H && ow < a.W;
    yoff[k] = yok[k] ? (unsigned)((oh * a.W + ow) * 16 + piece * 8) : 0u;
  }
  float scv[8], shv[8];
  if (PRO) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      scv[j] = a.sc[n * 16 + piece * 8 + j];
      shv[j] = a.sh[n * 16 + piece * 8 + j];
    }
  }
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

  auto load_x = [&](int id, uint4 (&rg)[NPT]) {
    const int idc = min(max(id, 0), a.D - 1);
    const bf16* xp = a.x + ((size_t)n * a.D + idc) * plane * CINR;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      if (CIN4) {
        const uint2 t = *reinterpret_cast<const uint2*>(xp + goff[k]);
        rg[k] = make_uint4(t.x, t.y, 0u, 0u);
      } else {
        rg[k] = *reinterpret_cast<const uint4*>(xp + goff[k]);
      }
    }
  };
  auto store_x = [&](int id, int slot, uint4 (&rg)[NPT]) {
    const bool pin = id >= 0 && id < a.D;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      uint4 v = rg[k];
      if (PRO) v = affine_relu8(v, scv, shv);
      const bool keep = pin && ok[k];
      v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
      *reinterpret_cast<uint4*>(smem + slot * PLB + (tid + k * 256) * 16) = v;
    }
  };
  auto load_y = [&](int od, uint4 (&rg)[2]) {
    const int odc = min(od, a.D - 1);
    const bf16* yp = a.dy + ((size_t)n * a.D + odc) * plane * 16;
#pragma unroll
    for (int k = 0; k < 2; ++k) rg[k] = *reinterpret_cast<const uint4*>(yp + yoff[k]);
  };
  auto store_y = [&](int od, int buf, uint4 (&rg)[2]) {
    const bool pin = od < d1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      uint4 v = rg[k];
      const bool keep = pin && yok[k];
      v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
      bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
      bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
      *reinterpret_cast<uint4*>(ys + buf * YPL + (tid + k * 256) * 16) = v;
    }
  };

  {
    uint4 r0[NPT], r1[NPT], r2[NPT], y0[2];
    load_x(d0 - 1, r0);
    load_x(d0, r1);
    load_x(d0 + 1, r2);
    load_y(d0, y0);
    store_x(d0 - 1, 0, r0);
    store_x(d0, 1, r1);
    store_x(d0 + 1, 2, r2);
    store_y(d0, 0, y0);
  }
  __syncthreads();

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wv = 4 * g + q;
  const int lanep = wv * 32 + p * 8;
  // this wave's two K steps of a plane: s = wave, wave + 4 -> (row pair rp = s >> 1, column half h = s & 1)
  int yo[2], xo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = wave + 4 * j, rp = s >> 1, h = s & 1;
    yo[j] = ((2 * rp) * TW + 16 * h) * 32 + lanep;
    xo[j] = ((2 * rp) * HW + 16 * h) * 32 + lanep;
  }

  // The planes stored at the end of iteration `it` (x plane od+2, dy plane od+1) were requested one iteration
  // EARLIER (prefetch distance 2, two register stages): one plane of MFMAs (about 1.5 us) does not cover an HBM
  // round trip under load, two do.
  const int np = d1 - d0;
  uint4 pxA[NPT], pyA[2], pxB[NPT], pyB[2];
  load_x(d0 + 2, pxA);
  load_y(d0 + 1, pyA);
  auto plane_step = [&](int it, uint4 (&prex)[NPT], uint4 (&prey)[2], uint4 (&nx)[NPT], uint4 (&ny)[2]) __attribute__((always_inline)) {
    const int od = d0 + it;
    load_x(od + 3, nx);
    load_y(od + 2, ny);
    __builtin_amdgcn_sched_barrier(0);

    const char* yb = ys + (it & 1) * YPL;
    // group gi = (step j, kd): 9 taps (kh, kw) of x plane it + kd
    auto read_group = [&](int gi, bf16x8 (&bf)[9]) {
      const int j = gi / 3, kd = gi % 3;
      const char* xb = smem + ((it + kd) & (R - 1)) * PLB + xo[j];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int o = ((t / 3) * HW + t % 3) * 32;
        bf[t] = tr_pair_m(xb + o, xb + o + HW * 32);
      }
    };
    bf16x8 af[2], bA[9], bB[9];
#pragma unroll
    for (int j = 0; j < 2; ++j) af[j] = tr_pair_m(yb + yo[j], yb + yo[j] + TW * 32);
    read_group(0, bA);
#define KS_INTERLEAVE                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 9; ++i_) {        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      \
  }
#pragma unroll
    for (int gi = 0; gi < 6; gi += 2) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t)
        acc[(gi % 3) * 9 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[gi / 3], bA[t], acc[(gi % 3) * 9 + t], 0, 0, 0);
      read_group(gi + 1, bB);
      KS_INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t)
        acc[((gi + 1) % 3) * 9 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[(gi + 1) / 3], bB[t], acc[((gi + 1) % 3) * 9 + t], 0, 0, 0);
      if (gi + 2 < 6) read_group(gi + 2, bA);
      KS_INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);
    }
#undef KS_INTERLEAVE

    store_x(od + 2, (it + 3) & (R - 1), prex);
    store_y(od + 1, (it + 1) & 1, prey);
    __syncthreads();
  };
  for (int it = 0; it < np; it += 2) {
    plane_step(it, pxA, pyA, pxB, pyB);
    if (it + 1 < np) plane_step(it + 1, pxB, pyB, pxA, pyA);
  }

  // ---- add the four waves' accumulators in wave order ((w0 + w1) + w2) + w3 through LDS (27 KB per wave)
  float* red = reinterpret_cast<float*>(smem);            // two regions of 27 * 256 floats
  if (wave == 1 || wave == 2) {
    float* r = red + (wave - 1) * 27 * 256;
#pragma unroll
    for (int t = 0; t < 27; ++t) *reinterpret_cast<f32x4*>(r + (t * 64 + lane) * 4) = acc[t];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(red + w * 27 * 256 + (t * 64 + lane) * 4);
        acc[t][0] += v[0]; acc[t][1] += v[1]; acc[t][2] += v[2]; acc[t][3] += v[3];
      }
  }
  __syncthreads();
  if (wave == 3) {
#pragma unroll
    for (int t = 0; t < 27; ++t) *reinterpret_cast<f32x4*>(red + (t * 64 + lane) * 4) = acc[t];
  }
  __syncthreads();
  if (wave == 0) {
    float* part = a.part + (size_t)blockIdx.x * 27 * 256;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(red + (t * 64 + lane) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) part[(t * 16 + g * 4 + j) * 16 + (lane & 15)] = acc[t][j] + v[j];
    }
  }
  if (a.bpart) {
    __syncthreads();
    float* bred = reinterpret_cast<float*>(smem);       // [256][8]
#pragma unroll
    for (int j = 0; j < 8; ++j) bred[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < 16) {
      const int pc = tid / 8, j = tid % 8;
      float s = 0.f;
      for (int k = pc; k < 256; k += 2) s += bred[k * 8 + j];
      a.bpart[(size_t)blockIdx.x * 16 + tid] = s;
    }
  }
}

static int march_dl(int64_t cols, int D) {
  int dl = 32;
  while (dl > 8 && cols * cdiv(D, dl) < 512) dl >>= 1;
  return dl;
}

// stride-1 3x3x3, Cout = 16, Cin = 16 (or 4), bf16.  query: only report the workspace size.
int wgrad3_c16_march(const devis_conv_desc* d, const void* x, const float* sc, const float* sh, const void* dy,
                     float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s, bool query, size_t* need) {
  if (d->dtype != DEVIS_BF16 || d->Cout != 16 || (d->C1 != 16 && d->C1 != 4) || d->C2 != 0 || d->SD != 1) return DEVIS_ERR_UNSUPPORTED;
  if (d->OD != d->ID || d->OH != d->IH || d->OW != d->IW) return DEVIS_ERR_UNSUPPORTED;
  if (d->IH * d->IW * 16 >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  if (d->C1 == 4 && sc) return DEVIS_ERR_UNSUPPORTED;
  WgMarchArgs a;
  a.N = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW;
  a.nth = (int)cdiv(a.H, 8); a.ntw = (int)cdiv(a.W, 32);
  a.DL = march_dl((int64_t)a.N * a.nth * a.ntw, a.D);
  a.ndc = (int)cdiv(a.D, a.DL);
  const int64_t blocks = (int64_t)a.N * a.ndc * a.nth * a.ntw;
  if (blocks > 65535) return DEVIS_ERR_UNSUPPORTED;
  const size_t part_bytes = (size_t)blocks * 27 * 256 * sizeof(float);
  const size_t total = part_bytes + (size_t)blocks * 16 * sizeof(float);
  if (query) { *need = total; return DEVIS_OK; }
  if (!ws || ws_bytes < total) return DEVIS_ERR_WORKSPACE;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.sc = sc; a.sh = sh;
  a.part = (float*)ws; a.bpart = dbias ? (float*)((char*)ws + part_bytes) : nullptr;
  constexpr int LDS = 4 * 3 * 256 * 16 + 2 * 8 * 32 * 32;
  const dim3 grid((unsigned)blocks);
  if (d->C1 == 4) hipLaunchKernelGGL((k_wgrad3_c16_ksplit<false, true>), grid, dim3(256), LDS, s, a);
  else if (sc) hipLaunchKernelGGL((k_wgrad3_c16_ksplit<true, false>), grid, dim3(256), LDS, s, a);
  else hipLaunchKernelGGL((k_wgrad3_c16_ksplit<false, false>), grid, dim3(256), LDS, s, a);
  int rc = launch_status();
  if (rc) return rc;
  return launch_wgrad_reduce(a.part, a.bpart, (int)blocks, 16, 16, (int)d->C1, dw, dbias, s);
}

// ------------------------------------------------------------------------------------------------
// 32 x 32-channel slabs of the wider layers (Cin, Cout multiples of 32; stride 1): the same walk with
// an (8 x 16) column, 64-byte voxel rows in LDS (both 32-byte halves XOR-swizzled for the transposing
// reads), 7 taps x 2 x 2 accumulator tiles per wave.  grid = (columns x D chunks, cin slabs x cout slabs);
// one partial of the whole [27][Cout][Cin] gradient per grid.x, each slab block writing its own part.
struct WgSlabArgs {
  const bf16* x;
  const bf16* dy;
  const float* sc;   // [N][Cin] or null
  const float* sh;
  float* part;       // [gx][27][Cout][Cin]
  float* bpart;      // [gx][Cout] or null
  int N, D, H, W, Cin, Cout;
  int nth, ntw, ndc, DL;
};

__device__ __forceinline__ int img64(int u, int half) { return u * 64 + ((half ^ ((u >> 2) & 1)) << 5); }

#ifdef DEVIS_PHASE_TIMING
#define PHASE_MARK(i) { const long long now = clock64(); ph[i] += now - tmark; tmark = now; }
#else
#define PHASE_MARK(i)
#endif

// scheduling pattern of one unit: 8 x { 1 MFMA, n LDS reads }
#define SLAB_INTERLEAVE(n)                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, n, 0);      \
  }

// r03: PRODUCER / CONSUMER halves.  The r02 kernel ran everything in its four waves, one after the other per plane:
// issue the next planes' loads (650 cycles), fragments + MFMAs (2.5k), affine + LDS stores (1.1k), barrier - the matrix
// pipe idle for 40 % of every plane on top of fill and drain (tools/bench_slab.hip).  Now a block has EIGHT waves, two
// per SIMD: waves 0-3 only read fragments and issue MFMAs (the r02 loop unchanged), waves 4-7 only fetch: each plane's
// loads are issued two iterations ahead (two register sets), prologue + zero padding + LDS stores + the bias sums run
// beside the other half's MFMAs (vector and matrix pipes of a SIMD work side by side for two different waves).  The
// planes' hand-over is still one block barrier per plane.  The partial sums leave in accumulator order (one 16-byte
// store per lane and tile, 1 KB per wave-instruction instead of four 64-byte rows); the reduction kernel undoes it.
template <bool PRO>
__global__ void __launch_bounds__(512)
k_wgrad3_slab_march(WgSlabArgs a) {
#ifdef DEVIS_PHASE_TIMING
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tmark = clock64();
#endif
  constexpr int TH = 8, TW = 16;
  constexpr int HH = TH + 2, HW = TW + 2, HP = HH * HW;          // 180 voxels
  constexpr int HWL = 24;                                        // LDS row pitch in voxels: a multiple of 8, so the
                                                                 // swizzle bit of a read depends on (lane, kw) only
  constexpr int NPT = (HP * 4 + 255) / 256;                      // 3
  constexpr int PLB = HH * HWL * 64;                             // 15 KB per x plane
  constexpr int R = 4;
  constexpr int YPL = TH * TW * 64;                              // 8 KB per dy plane tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ys = smem + R * PLB;
  const int lane = threadIdx.x & 63, wave8 = threadIdx.x >> 6;
  const bool producer = wave8 >= 4;
  const int tid = threadIdx.x & 255, wave = wave8 & 3;           // index inside the half
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int nci = a.Cin / 32;
  const int slab = blockIdx.y;
  const int ci0 = (slab % nci) * 32, co0 = (slab / nci) * 32;

  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int qq = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (bid >> 3);
  }
  const int twi = tile % a.ntw;
  const int thi = (tile / a.ntw) % a.nth;
  const int dci = (tile / (a.ntw * a.nth)) % a.ndc;
  const int n = tile / (a.ntw * a.nth * a.ndc);
  const int oh0 = thi * TH, ow0 = twi * TW;
  const int d0 = dci * a.DL;
  const int d1 = min(d0 + a.DL, a.D);
  const int np = d1 - d0;
  const size_t plane = (size_t)a.H * a.W;

  if (producer) {
    // ------------------------------------------------------------------ waves 4-7: HBM -> LDS
    const int piece = tid & 3;                                      // 8 channels of the slab's 32
    unsigned goff[NPT], loff[NPT];
    bool ok[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int i = tid + k * 256;
      const int u = i >> 2;
      const int hh = u / HW, hw = u - hh * HW;
      const int ih = oh0 - 1 + hh, iw = ow0 - 1 + hw;
      ok[k] = i < HP * 4 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
      goff[k] = ok[k] ? (unsigned)((ih * a.W + iw) * a.Cin + ci0 + piece * 8) : (unsigned)ci0;
      loff[k] = (unsigned)(img64(hh * HWL + hw, piece >> 1) + (piece & 1) * 16);
    }
    unsigned yoff[2], yl[2];
    bool yok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = (tid + k * 256) >> 2;                           // voxel of the 8x16 tile, row-major
      const int oh = oh0 + v / TW, ow = ow0 + v % TW;
      yok[k] = oh < a.H && ow < a.W;
      yoff[k] = yok[k] ? (unsigned)((oh * a.W + ow) * a.Cout + co0 + piece * 8) : (unsigned)co0;
      yl[k] = (unsigned)(img64(v, piece >> 1) + (piece & 1) * 16);
    }
    float scv[8], shv[8];
    if (PRO) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        scv[j] = a.sc[n * a.Cin + ci0 + piece * 8 + j];
        shv[j] = a.sh[n * a.Cin + ci0 + piece * 8 + j];
      }
    }
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    const bool do_bias = a.bpart != nullptr && ci0 == 0;

    auto load_x = [&](int id, u32x4 (&rg)[NPT]) __attribute__((always_inline)) {
      const int idc = min(max(id, 0), a.D - 1);
      const bf16* xp = a.x + ((size_t)n * a.D + idc) * plane * a.Cin;
#pragma unroll
      for (int k = 0; k < NPT; ++k) rg[k] = *reinterpret_cast<const u32x4*>(xp + goff[k]);
    };
    auto store_x = [&](int id, int slot, u32x4 (&rg)[NPT]) __attribute__((always_inline)) {
      const bool pin = id >= 0 && id < a.D;
#pragma unroll
      for (int k = 0; k < NPT; ++k) {
        u32x4 v = rg[k];
        if (PRO) v = affine_relu8(v, scv, shv);
        if (!(pin && ok[k])) v = (u32x4){0u, 0u, 0u, 0u};
        if (k < NPT - 1 || tid + k * 256 < HP * 4) *reinterpret_cast<u32x4*>(smem + slot * PLB + loff[k]) = v;
      }
    };
    auto load_y = [&](int od, u32x4 (&rg)[2]) __attribute__((always_inline)) {
      const int odc = min(od, a.D - 1);
      const bf16* yp = a.dy + ((size_t)n * a.D + odc) * plane * a.Cout;
#pragma unroll
      for (int k = 0; k < 2; ++k) rg[k] = *reinterpret_cast<const u32x4*>(yp + yoff[k]);
    };
    auto store_y = [&](int od, int buf, u32x4 (&rg)[2]) __attribute__((always_inline)) {
      const bool pin = od < d1;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        u32x4 v = rg[k];
        if (!(pin && yok[k])) v = (u32x4){0u, 0u, 0u, 0u};
        if (do_bias) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { bsum[2 * j] += bf_lo(v[j]); bsum[2 * j + 1] += bf_hi(v[j]); }
        }
        *reinterpret_cast<u32x4*>(ys + buf * YPL + yl[k]) = v;
      }
    };

    // two register sets: planes (od + 2, od + 1) of iteration `it` were requested during iteration it - 1
    u32x4 xa[NPT], ya[2], xb[NPT], yb[2];
    {
      u32x4 r0[NPT], r1[NPT], r2[NPT], y0[2];
      load_x(d0 - 1, r0);
      load_x(d0, r1);
      load_x(d0 + 1, r2);
      load_y(d0, y0);
      load_x(d0 + 2, xa);                       // iteration 0's planes
      load_y(d0 + 1, ya);
      store_x(d0 - 1, 0, r0);
      store_x(d0, 1, r1);
      store_x(d0 + 1, 2, r2);
      store_y(d0, 0, y0);
    }
    PHASE_MARK(4)
    __syncthreads();
    for (int it = 0; it < np; it += 2) {
      {
        const int od = d0 + it;
        load_x(od + 3, xb);                     // for iteration it + 1
        load_y(od + 2, yb);
        PHASE_MARK(0)
        store_x(od + 2, (it + 3) & (R - 1), xa);
        store_y(od + 1, (it + 1) & 1, ya);
        PHASE_MARK(2)
        __syncthreads();
        PHASE_MARK(3)
      }
      if (it + 1 < np) {
        const int od = d0 + it + 1;
        load_x(od + 3, xa);
        load_y(od + 2, ya);
        PHASE_MARK(0)
        store_x(od + 2, (it + 4) & (R - 1), xb);
        store_y(od + 1, (it + 2) & 1, yb);
        PHASE_MARK(2)
        __syncthreads();
        PHASE_MARK(3)
      }
    }
    if (do_bias) {
      // the consumers are done with the LDS planes behind the walk's last barrier: the bias sums go through them
      float* red = reinterpret_cast<float*>(smem);       // [256][8]
#pragma unroll
      for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bsum[j];
    }
    __syncthreads();                                     // (A) matched by the consumers
    float* red = reinterpret_cast<float*>(smem);
    float* red2 = red + 2048;                            // [8][32]
    if (do_bias) {
      // channel c = tid % 32 (piece c/8, element c%8) is held by threads k = c/8 (mod 4); 8 groups of 8 threads
      // each, then the 8 group sums in order
      const int c = tid & 31, grp = tid >> 5, pc = c >> 3, j = c & 7;
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) sacc += red[(pc + 4 * (grp * 8 + i)) * 8 + j];
      red2[grp * 32 + c] = sacc;
    }
    __syncthreads();                                     // (B)
    if (do_bias && tid < 32) {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) sacc += red2[i * 32 + tid];
      a.bpart[(size_t)blockIdx.x * a.Cout + co0 + tid] = sacc;
    }
    PHASE_MARK(6)
#ifdef DEVIS_PHASE_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
      long long* dbg = reinterpret_cast<long long*>(a.bpart + (size_t)gridDim.x * a.Cout) + (4 + wave) * 8;
      for (int i = 0; i < 8; ++i) dbg[i] = ph[i];
    }
#endif
    return;
  }

  // -------------------------------------------------------------------- waves 0-3: fragments + MFMAs
  f32x4 acc[7][2][2];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[t][mo][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wv = 4 * g + q;                       // voxel column this lane addresses
  // Lane part of every fragment address, computed once: with 8-voxel-aligned rows the XOR bit of voxel
  // (row, wv + kw) is ((wv + kw) >> 2) & 1, so the K step and the row only add compile-time offsets.
  int tkd[7], xl[7][2], ya[2];
#pragma unroll
  for (int tt = 0; tt < 7; ++tt) {
    int t = wave + 4 * tt;
    if (t > 26) t = 26;                           // phantom tap of wave 3: computed, never written
    tkd[tt] = t / 9;
    const int kh = (t / 3) % 3, col = wv + t % 3;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) xl[tt][mi] = (kh * HWL + col) * 64 + ((mi ^ ((col >> 2) & 1)) << 5) + p * 8;
  }
#pragma unroll
  for (int mo = 0; mo < 2; ++mo) ya[mo] = wv * 64 + ((mo ^ ((wv >> 2) & 1)) << 5) + p * 8;
  const bool last_real = wave + 24 < 27;

  PHASE_MARK(4)
  __syncthreads();
  for (int it = 0; it < np; ++it) {
    const char* yb[2];
#pragma unroll
    for (int mo = 0; mo < 2; ++mo) yb[mo] = ys + (it & 1) * YPL + ya[mo];
    const char* xb[7][2];
#pragma unroll
    for (int tt = 0; tt < 7; ++tt)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) xb[tt][mi] = smem + ((it + tkd[tt]) & (R - 1)) * PLB + xl[tt][mi];

    // K step s: rows 2s and 2s+1 of the tile, voxel column wv
    auto read_a = [&](int s, bf16x8 (&af)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int mo = 0; mo < 2; ++mo) af[mo] = tr_pair_m(yb[mo] + (2 * s) * TW * 64, yb[mo] + (2 * s + 1) * TW * 64);
    };
    // unit (s, tg): taps tt = 2tg, 2tg+1 (tg = 3: tap 6 only)
    auto read_b = [&](int s, int tg, bf16x8 (&bf)[2][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tt = 2 * tg + h;
        if (tt < 7) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            bf[h][mi] = tr_pair_m(xb[tt][mi] + (2 * s) * HWL * 64, xb[tt][mi] + (2 * s + 1) * HWL * 64);
        }
      }
    };
    auto mma = [&](int tg, bf16x8 (&af)[2], bf16x8 (&bf)[2][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tt = 2 * tg + h;
        if (tt < 7) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int mo = 0; mo < 2; ++mo)
              acc[tt][mo][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mo], bf[h][mi], acc[tt][mo][mi], 0, 0, 0);
        }
      }
    };
    bf16x8 afA[2], afB[2], bA[2][2], bB[2][2];
    read_a(0, afA);
    read_b(0, 0, bA);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // four units of this K step; fragments of the next unit are in flight under the MFMAs of the current one
      bf16x8 (&af)[2] = (s & 1) ? afB : afA;
      bf16x8 (&afn)[2] = (s & 1) ? afA : afB;
      // each unit's MFMAs are issued with the next unit's fragment reads slotted between them (one MFMA,
      // then the reads that fit under its 16 cycles), so neither the LDS queue nor the matrix pipe waits
      __builtin_amdgcn_sched_barrier(0);
      mma(0, af, bA);
      read_b(s, 1, bB);
      SLAB_INTERLEAVE(1)
      __builtin_amdgcn_sched_barrier(0);
      mma(1, af, bB);
      read_b(s, 2, bA);
      SLAB_INTERLEAVE(1)
      __builtin_amdgcn_sched_barrier(0);
      mma(2, af, bA);
      read_b(s, 3, bB);
      if (s + 1 < 4) read_a(s + 1, afn);
      SLAB_INTERLEAVE(2)
      __builtin_amdgcn_sched_barrier(0);
      mma(3, af, bB);
      if (s + 1 < 4) read_b(s + 1, 0, bA);
      SLAB_INTERLEAVE(2)
      __builtin_amdgcn_sched_barrier(0);
    }
    PHASE_MARK(1)
    __syncthreads();
    PHASE_MARK(3)
  }

  // ---- partial results in accumulator order: part[gx][slab][tap][mo][mi][lane][4] (lane: ci = lane & 15, the four
  // values co = (lane >> 4) * 4 + j): one 16-byte store per lane and tile; RED_CONV3F in wgrad_reduce.h reads it back
  float* part = a.part + ((size_t)blockIdx.x * gridDim.y + slab) * (27 * 4 * 64 * 4);
#pragma unroll
  for (int tt = 0; tt < 7; ++tt) {
    const int t = wave + 4 * tt;
    if (tt < 6 || last_real) {
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          *reinterpret_cast<f32x4*>(part + (((t * 2 + mo) * 2 + mi) * 64 + lane) * 4) = acc[tt][mo][mi];
    }
  }
  PHASE_MARK(5)
  __syncthreads();                                       // (A), (B): the producers' bias reduction
  __syncthreads();
#ifdef DEVIS_PHASE_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
    long long* dbg = reinterpret_cast<long long*>(a.bpart + (size_t)gridDim.x * a.Cout) + wave * 8;
    for (int i = 0; i < 8; ++i) dbg[i] = ph[i];
  }
#endif
}

// stride-1 3x3x3 with Cin, Cout multiples of 32 (<= 128), bf16.  query: only report the workspace size.
int wgrad3_slab_march(const devis_conv_desc* d, const void* x, const float* sc, const float* sh, const void* dy,
                      float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s, bool query, size_t* need) {
  if (d->dtype != DEVIS_BF16 || d->C2 != 0 || d->SD != 1 || d->C1 % 32 || d->Cout % 32 || d->C1 > 128 || d->Cout > 128) return DEVIS_ERR_UNSUPPORTED;
  if (d->OD != d->ID || d->OH != d->IH || d->OW != d->IW) return DEVIS_ERR_UNSUPPORTED;
  if (d->IH * d->IW * (d->C1 > d->Cout ? d->C1 : d->Cout) >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  WgSlabArgs a;
  a.N = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW; a.Cin = (int)d->C1; a.Cout = (int)d->Cout;
  a.nth = (int)cdiv(a.H, 8); a.ntw = (int)cdiv(a.W, 16);
  const int nslab = (a.Cin / 32) * (a.Cout / 32);
  // D chunk: about 256 blocks in all (one per CU), never shorter than 4 planes (each chunk re-reads 2 halo
  // planes and leaves one more partial for the reduction)
  const int64_t cols = (int64_t)a.N * a.nth * a.ntw * nslab;
  int dl = 32;
  while (dl > 4 && cols * cdiv(a.D, dl) < 256) dl >>= 1;
  a.DL = dl;
  a.ndc = (int)cdiv(a.D, dl);
  const int64_t gx = (int64_t)a.N * a.ndc * a.nth * a.ntw;
  if (gx > 0x7fffffff || nslab > 65535) return DEVIS_ERR_UNSUPPORTED;
  const size_t part_bytes = (size_t)gx * 27 * a.Cout * a.Cin * sizeof(float);
  const size_t total = part_bytes + (size_t)gx * a.Cout * sizeof(float);
  if (query) { *need = total; return DEVIS_OK; }
  if (!ws || ws_bytes < total) return DEVIS_ERR_WORKSPACE;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.sc = sc; a.sh = sh;
  a.part = (float*)ws; a.bpart = dbias ? (float*)((char*)ws + part_bytes) : nullptr;
  constexpr int LDS = 4 * 10 * 24 * 64 + 2 * 8 * 16 * 64;
  const dim3 grid((unsigned)gx, (unsigned)nslab);
  if (sc) hipLaunchKernelGGL((k_wgrad3_slab_march<true>), grid, dim3(512), LDS, s, a);
  else hipLaunchKernelGGL((k_wgrad3_slab_march<false>), grid, dim3(512), LDS, s, a);
  int rc = launch_status();
  if (rc) return rc;
  return launch_wgrad_reduce_frag(a.part, a.bpart, (int)gx, a.Cout, a.Cin, dw, dbias, s);
}

}  // namespace devis
