// MFMA 3x3x3 convolution for the deep, small-volume layers (64 and 128 channels at 32^3 / 16^3; bf16,
// channels-last, stride 1, gfx950).  Here the weights (442 KB - 1.7 MB per layer) outweigh the
// activations, and a block has few voxels to amortise them over, so both operands go through LDS:
//
//   * a block computes 32 cout of a (TD x TH x 16)-voxel tile.  Per 32-channel K chunk it stages the
//     activation halo tile (norm affine + ReLU applied) AND the chunk's [27][32 cout][32 ci] weight
//     slab (54 KB) in LDS with bulk coalesced loads - no per-tap weight fetch sits in the MFMA loop.
//   * the next chunk's activations and weights are fetched into registers while the 27 taps of the
//     current chunk run on the matrix cores; A/B fragments are double-buffered across taps.
//   * both LDS images use the XOR swizzle of conv_mfma.hip (64-byte rows, conflict-free b128 reads).
//   * GEMM orientation as everywhere: M = cout, N = 16 voxels along W, K = (tap, ci) -> 8-byte
//     channels-last stores; bias and residual in the epilogue.
// dgrad of these layers is the same kernel on the tap-flipped [Cin][27][Cout] weights.
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {

struct DeepArgs {
  const bf16* x;
  const bf16* w;      // [Cout][27][CIN]
  const float* bias;  // [Cout] or null
  const float* sc;    // [N][CIN] or null
  const float* sh;
  const bf16* res;    // like y, or null
  bf16* y;
  int N, D, H, W, Cout;
  int ntd, nth, ntw;
  StatsArgs st;       // norm statistics gathered in the epilogue (mode 0: off)
};

template <int CIN, int TD, int TH, bool PRO, int STATS>
__global__ void __launch_bounds__(256)
k_conv3_deep(DeepArgs a) {
  constexpr int NCH = CIN / 32, MT = 2;
  constexpr int NTB = TD * TH, NT = NTB / 4;
  constexpr int HD = TD + 2, HH = TH + 2, HW = 18, HV = HD * HH * HW;
  constexpr int XPT = (HV * 4 + 255) / 256;           // 16-byte pieces per thread: activations
  constexpr int WROWS = 27 * 32;
  constexpr int WPT = (WROWS * 4 + 255) / 256;        // ... weights (13.5 -> 14)
  constexpr int XB = XPT * 256 * 16;
  static_assert(NTB % 4 == 0, "N tiles must split over 4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* ws = smem + XB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;

  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int twi = tile % a.ntw;
  const int thi = (tile / a.ntw) % a.nth;
  const int tdi = (tile / (a.ntw * a.nth)) % a.ntd;
  const int n = tile / (a.ntw * a.nth * a.ntd);
  const int od0 = tdi * TD, oh0 = thi * TH, ow0 = twi * 16;
  const int co_base = blockIdx.y * 32;
  const int piece = tid & 3;

  // ---- per-thread slots (the same for every chunk, apart from the channel offset)
  unsigned xg[XPT], xl[XPT];
  bool xok[XPT];
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int i = tid + k * 256;
    const int u = i >> 2;
    const int hw = u % HW, hh = (u / HW) % HH, hd = u / (HW * HH);
    const int id = od0 - 1 + hd, ih = oh0 - 1 + hh, iw = ow0 - 1 + hw;
    xok[k] = i < HV * 4 && id >= 0 && id < a.D && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
    xg[k] = xok[k] ? (unsigned)((((n * a.D + id) * a.H + ih) * a.W + iw) * CIN + piece * 8) : 0u;
    xl[k] = (unsigned)(u * 64 + ((piece ^ ((u >> 1) & 3)) << 4));
  }
  unsigned wg[WPT], wl[WPT];
  bool wok[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int i = tid + k * 256;
    const int row = i >> 2;                     // tap * 32 + co
    const int t = row >> 5, co = row & 31;
    wok[k] = i < WROWS * 4;
    wg[k] = wok[k] ? (unsigned)(((co_base + co) * 27 + t) * CIN + piece * 8) : 0u;
    wl[k] = (unsigned)(row * 64 + ((piece ^ ((row >> 1) & 3)) << 4));
  }

  auto fetch = [&](int cc, u32x4 (&xr)[XPT], u32x4 (&wr)[WPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < XPT; ++k) xr[k] = *reinterpret_cast<const u32x4*>(a.x + xg[k] + cc * 32);
#pragma unroll
    for (int k = 0; k < WPT; ++k) wr[k] = *reinterpret_cast<const u32x4*>(a.w + wg[k] + cc * 32);
  };
  auto stage = [&](int cc, u32x4 (&xr)[XPT], u32x4 (&wr)[WPT]) __attribute__((always_inline)) {
    float scv[8], shv[8];
    if (PRO) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        scv[j] = a.sc[n * CIN + cc * 32 + piece * 8 + j];
        shv[j] = a.sh[n * CIN + cc * 32 + piece * 8 + j];
      }
    }
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      u32x4 v = xr[k];
      if (PRO) v = affine_relu8(v, scv, shv);
      if (!xok[k]) v = (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(xs + xl[k]) = v;
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k)
      if (k < WPT - 1 || wok[k]) *reinterpret_cast<u32x4*>(ws + wl[k]) = wr[k];
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // LDS voxel index of each of this wave's N tiles at tap (0,0,0), lane's voxel included
  int ubase[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int g = wave * NT + j;
    ubase[j] = ((g / TH) * HH + g % TH) * HW + l15;
  }

  u32x4 xr[XPT], wr[WPT];
  fetch(0, xr, wr);
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc) {
    __syncthreads();                       // the previous chunk's fragments have all been read
    stage(cc, xr, wr);
    __syncthreads();
    if (cc + 1 < NCH) fetch(cc + 1, xr, wr);
    __builtin_amdgcn_sched_barrier(0);

    auto read_tap = [&](int t, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = t * 32 + m * 16 + l15;
        af[m] = *reinterpret_cast<const bf16x8*>(ws + row * 64 + ((lq ^ ((row >> 1) & 3)) << 4));
      }
      const int toff = ((t / 9) * HH + (t / 3) % 3) * HW + t % 3;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int u = ubase[j] + toff;
        bf[j] = *reinterpret_cast<const bf16x8*>(xs + u * 64 + ((lq ^ ((u >> 1) & 3)) << 4));
      }
    };
    bf16x8 afA[MT], bfA[NT], afB[MT], bfB[NT];
    read_tap(0, afA, bfA);
#pragma unroll
    for (int t = 0; t < 27; t += 2) {
      if (t + 1 < 27) read_tap(t + 1, afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afA[m], bfA[j], acc[m][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < 27) read_tap(t + 2, afA, bfA);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < 27) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afB[m], bfB[j], acc[m][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: + bias (+ residual), 4 consecutive cout of one voxel per lane -> 8-byte stores
  float f0[MT][4], f1[MT][4], scs[MT][4], shs[MT][4];
  if (STATS) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f0[m][k] = 0.f; f1[m][k] = 0.f;
        if (STATS == 2) {
          scs[m][k] = a.st.sc[n * a.Cout + co_base + m * 16 + lq * 4 + k];
          shs[m][k] = a.st.sh[n * a.Cout + co_base + m * 16 + lq * 4 + k];
        }
      }
  }
  // mode 2 reads the normalised tensor at this lane's outputs: all loads first, then the stores (a load issued
  // behind a store would wait for it: vmcnt counts both)
  uint2 xref[MT][NT];
  if (STATS == 2) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int g = wave * NT + j;
      const int od = od0 + g / TH, oh = oh0 + g % TH, ow = ow0 + l15;
      const bool in = od < a.D && oh < a.H && ow < a.W;
      const size_t vox = in ? (size_t)((n * a.D + od) * a.H + oh) * a.W + ow : 0;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        xref[m][j] = *reinterpret_cast<const uint2*>(a.st.xref + vox * a.Cout + co_base + m * 16 + lq * 4);
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int g = wave * NT + j;
    const int od = od0 + g / TH, oh = oh0 + g % TH, ow = ow0 + l15;
    if (od >= a.D || oh >= a.H || ow >= a.W) continue;
    const size_t vox = (size_t)((n * a.D + od) * a.H + oh) * a.W + ow;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co = co_base + m * 16 + lq * 4;
      float v0 = acc[m][j][0], v1 = acc[m][j][1], v2 = acc[m][j][2], v3 = acc[m][j][3];
      if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
      }
      if (a.res) {
        const uint2 r = *reinterpret_cast<const uint2*>(a.res + vox * a.Cout + co);
        v0 += bf_lo(r.x); v1 += bf_hi(r.x); v2 += bf_lo(r.y); v3 += bf_hi(r.y);
      }
      uint2 o;
      o.x = pack_bf2(v0, v1);
      o.y = pack_bf2(v2, v3);
      *reinterpret_cast<uint2*>(a.y + vox * a.Cout + co) = o;
      if (STATS) {
        stats_accum<STATS>(o, STATS == 2 ? xref[m][j] : o, scs[m], shs[m], f0[m], f1[m]);
      }
    }
  }
  if (STATS) {
    __syncthreads();               // every wave is done reading its LDS fragments
    const int chunk = tile % (a.ntw * a.nth * a.ntd);
    stats_finish_dpp<MT>(f0, f1, reinterpret_cast<double*>(smem),
                     a.st.partial + (((size_t)n * a.st.chunks + chunk) * a.Cout + co_base) * 2);
  }
}

template <int CIN, int TD, int TH, bool PRO, int STATS>
static int launch_deep_k(const DeepArgs& a, dim3 grid, int lds, hipStream_t s) {
  auto k = k_conv3_deep<CIN, TD, TH, PRO, STATS>;
  static bool attr_done = false;
  if (!attr_done) { if (check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds))) return DEVIS_ERR_HIP; attr_done = true; }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  return launch_status();
}

// plan_chunks != null: only report the statistics chunks per sample this launch would leave
template <int CIN, int TD, int TH>
static int launch_deep(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
                       const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks,
                       hipStream_t s) {
  constexpr int HV = (TD + 2) * (TH + 2) * 18;
  constexpr int LDS = ((HV * 4 + 255) / 256) * 256 * 16 + 27 * 32 * 64;
  DeepArgs a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.sc = sc; a.sh = sh; a.res = (const bf16*)res; a.y = (bf16*)y;
  a.N = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW; a.Cout = (int)d->Cout;
  a.ntd = (int)cdiv(a.D, TD); a.nth = (int)cdiv(a.H, TH); a.ntw = (int)cdiv(a.W, 16);
  const int64_t tiles = (int64_t)a.N * a.ntd * a.nth * a.ntw;
  if (tiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  if (plan_chunks) {
    *plan_chunks = (st && st->mode == 2 && sc) ? 0 : (int64_t)a.ntd * a.nth * a.ntw;
    return DEVIS_OK;
  }
  const dim3 grid((unsigned)tiles, (unsigned)(d->Cout / 32));
  a.st = StatsArgs{};
  const int mode = st ? st->mode : 0;
  if (mode) {
    a.st = *st;
    if (a.st.chunks != a.ntd * a.nth * a.ntw || a.st.C != a.Cout) return DEVIS_ERR_BAD_ARG;
    if (mode == 2 && sc) return DEVIS_ERR_UNSUPPORTED;
  }
  if (mode == 1) return sc ? launch_deep_k<CIN, TD, TH, true, 1>(a, grid, LDS, s) : launch_deep_k<CIN, TD, TH, false, 1>(a, grid, LDS, s);
  if (mode == 2) return launch_deep_k<CIN, TD, TH, false, 2>(a, grid, LDS, s);
  return sc ? launch_deep_k<CIN, TD, TH, true, 0>(a, grid, LDS, s) : launch_deep_k<CIN, TD, TH, false, 0>(a, grid, LDS, s);
}

// stride-1 3x3x3 with Cin == Cout in {64, 128}, bf16; anything else is unsupported
int conv3_deep(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
               const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  if (d->dtype != DEVIS_BF16 || d->C2 != 0 || d->SD != 1 || d->C1 != d->Cout) return DEVIS_ERR_UNSUPPORTED;
  if (d->OD != d->ID || d->OH != d->IH || d->OW != d->IW) return DEVIS_ERR_UNSUPPORTED;
  if (d->N * d->ID * d->IH * d->IW * d->C1 >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  if (d->C1 == 128) {
    // (2, 2, 16) tiles (75 KB of LDS: two blocks per CU) were measured against (2, 4, 16) (one block per CU):
    // 21.5 vs 20.0 us forward, 20.9 vs 18.0 us dgrad at 2 x 16^3 - the weight slabs are then staged twice as often
    static const bool small = getenv("DEVIS_DEEP128_SMALL_TILE") != nullptr;
    if (small) return launch_deep<128, 2, 2>(d, x, w, bias, sc, sh, res, y, st, plan_chunks, s);
    return launch_deep<128, 2, 4>(d, x, w, bias, sc, sh, res, y, st, plan_chunks, s);
  }
  if (d->C1 == 64) return launch_deep<64, 4, 4>(d, x, w, bias, sc, sh, res, y, st, plan_chunks, s);
  return DEVIS_ERR_UNSUPPORTED;
}

}  // namespace devis
