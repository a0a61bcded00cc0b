// MFMA implicit-GEMM 3x3x3 convolution, bf16 channels-last, gfx950.
//
//   out[co][v] = sum_{tap, ci} W[co][tap][ci] * A[v + off(tap)][ci]
//
// GEMM orientation: M = cout (A operand = packed weights, k-contiguous rows), N = 16 output
// voxels along W (B operand = activations), K = (tap, ci).  With v_mfma_f32_16x16x32_bf16 a lane
// then owns 4 consecutive cout of ONE voxel -> 8-byte coalesced channels-last stores, and the
// B fragment of a lane is 16 contiguous bytes of one input voxel -> one ds_read_b128.
// The input halo tile is staged once per ci-chunk through registers into LDS, where the preceding
// GroupNorm/BatchNorm affine + ReLU is applied on the fly (the normalised tensor never exists in
// HBM); bias and the residual add run in the epilogue.  Shapes these kernels do not take return
// DEVIS_ERR_UNSUPPORTED and the caller routes to the direct kernels.
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {

struct Conv3Args {
  const bf16* x;
  const bf16* w;      // [Cout][27][Cin]
  const float* bias;  // [Cout] or null
  const float* sc;    // [N][Cin] or null
  const float* sh;
  const bf16* res;    // like y, or null
  bf16* y;
  int N, ID, IH, IW, OD, OH, OW, Cout;
  int ntd, nth, ntw;
  int cin_real;       // 4 for the first layer (zero-extended to 16 in LDS), else CIN
  StatsArgs st;       // norm statistics gathered in the epilogue (mode 0: off)
};

template <int CIN, int COUT_T, int TD, int TH, int TW, int S>
struct Cfg {
  static constexpr int CK = CIN >= 32 ? 32 : 16;   // channels per K chunk held in LDS
  static constexpr int NCH = CIN / CK;
  static constexpr int MT = COUT_T / 16;
  static constexpr int NTW = TW / 16;
  static constexpr int NTB = TD * TH * NTW;         // 16-voxel N tiles per block
  static constexpr int NT = NTB / 4;                // per wave
  static constexpr int HD = (TD - 1) * S + 3, HH = (TH - 1) * S + 3, HW = (TW - 1) * S + 3;
  static constexpr int HV = HD * HH * HW;
  static constexpr int PITCH = CK * 2;
  static constexpr int LDS_BYTES = HV * PITCH;
  static_assert(NTB % 4 == 0, "N tiles must split over 4 waves");
  static_assert(CIN % CK == 0 && COUT_T % 16 == 0 && TW % 16 == 0, "tile shape");
};

template <int HH, int HW>
__device__ __forceinline__ constexpr int tap_off(int t) {
  return ((t / 9) * HH + (t / 3) % 3) * HW + t % 3;
}

template <int CIN, int COUT_T, int TD, int TH, int TW, int S, bool PRO, int STATS>
__global__ void __launch_bounds__(256)
k_conv3_mfma(Conv3Args a) {
  using C = Cfg<CIN, COUT_T, TD, TH, TW, S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;

  // XCD-aware (bijective) block -> tile map: blocks b and b+8 share an XCD, give them adjacent tiles
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
  const int od0 = tdi * TD, oh0 = thi * TH, ow0 = twi * TW;
  const int co_base = blockIdx.y * COUT_T;

  f32x4 acc[C::MT][C::NT];
#pragma unroll
  for (int m = 0; m < C::MT; ++m)
#pragma unroll
    for (int j = 0; j < C::NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane LDS voxel index of each of this wave's N tiles at tap (0,0,0)
  int ubase[C::NT];
#pragma unroll
  for (int j = 0; j < C::NT; ++j) {
    const int g = wave * C::NT + j;
    const int ltw = g % C::NTW, lth = (g / C::NTW) % TH, ltd = g / (C::NTW * TH);
    ubase[j] = ((ltd * S) * C::HH + lth * S) * C::HW + (ltw * 16 + l15) * S;
  }

  const bf16* wrow = a.w + (size_t)(co_base + l15) * 27 * a.cin_real;

  // ---- halo-tile staging, split in two so that a thread's XPT loads are all in flight together (one load
  // per loop trip would leave the kernel waiting on HBM latency XPT times per chunk): fetch() issues them into
  // registers - for chunk cc+1 while the MFMAs of chunk cc run - and stage() applies the norm prologue and
  // writes LDS.  Slots are per thread and the same for every chunk; out-of-volume slots load a clamped
  // address and are zeroed by select, so no memory instruction sits under a divergent branch.
  constexpr int PPV = C::CK / 8;                          // 16-byte pieces per voxel
  constexpr int XPT = (C::HV * PPV + 255) / 256;
  const int piece = tid % PPV;
  const bool cin4 = CIN == 16 && a.cin_real == 4;
  unsigned xg[XPT], xl[XPT];
  bool xok[XPT];
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int i = tid + k * 256;
    const int u = i / PPV;
    const int hw = u % C::HW, hh = (u / C::HW) % C::HH, hd = u / (C::HW * C::HH);
    const int id = od0 * S - 1 + hd, ih = oh0 * S - 1 + hh, iw = ow0 * S - 1 + hw;
    xok[k] = i < C::HV * PPV && id >= 0 && id < a.ID && ih >= 0 && ih < a.IH && iw >= 0 && iw < a.IW && !(cin4 && piece);
    const unsigned vox = (unsigned)(((n * a.ID + id) * a.IH + ih) * a.IW + iw);
    xg[k] = xok[k] ? (cin4 ? vox * 4u : vox * (unsigned)CIN + (unsigned)(piece * 8)) : 0u;
    const int slot = C::CK == 32 ? (piece ^ ((u >> 1) & 3)) : piece;   // XOR swizzle: conflict-free b128 reads
    xl[k] = (unsigned)(u * C::PITCH + slot * 16);
  }
  auto fetch = [&](int cc, u32x4 (&xr)[XPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      if (CIN == 16 && cin4) {
        const uint2 t = *reinterpret_cast<const uint2*>(a.x + xg[k]);
        xr[k] = (u32x4){t.x, t.y, 0u, 0u};
      } else {
        xr[k] = *reinterpret_cast<const u32x4*>(a.x + xg[k] + cc * C::CK);
      }
    }
  };
  auto stage = [&](int cc, u32x4 (&xr)[XPT]) __attribute__((always_inline)) {
    float scv[8], shv[8];
    if (PRO) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        scv[j] = a.sc[n * CIN + cc * C::CK + piece * 8 + j];
        shv[j] = a.sh[n * CIN + cc * C::CK + piece * 8 + j];
      }
    }
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      u32x4 v = xr[k];
      if (PRO) v = affine_relu8(v, scv, shv);
      if (!xok[k]) v = (u32x4){0u, 0u, 0u, 0u};
      if (k < XPT - 1 || tid + k * 256 < C::HV * PPV) *reinterpret_cast<u32x4*>(smem + xl[k]) = v;
    }
  };

  u32x4 xr[XPT];
  fetch(0, xr);
  for (int cc = 0; cc < C::NCH; ++cc) {
    if (cc > 0) __syncthreads();
    stage(cc, xr);
    __syncthreads();
    if (cc + 1 < C::NCH) fetch(cc + 1, xr);

    // ---- MFMA over the 27 taps of this channel chunk
    if (C::CK == 32) {
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        bf16x8 af[C::MT];
#pragma unroll
        for (int m = 0; m < C::MT; ++m)
          af[m] = *reinterpret_cast<const bf16x8*>(wrow + (size_t)m * 16 * 27 * CIN + t * CIN + cc * 32 + lq * 8);
        const int toff = tap_off<C::HH, C::HW>(t);
#pragma unroll
        for (int j = 0; j < C::NT; ++j) {
          const int u = ubase[j] + toff;
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(smem + u * 64 + ((lq ^ ((u >> 1) & 3)) << 4));
#pragma unroll
          for (int m = 0; m < C::MT; ++m)
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr, acc[m][j], 0, 0, 0);
        }
      }
    } else {
      // CK == 16: one K chunk = two taps x 16 channels; lanes 0-31 take tap 2c, lanes 32-63 tap 2c+1
      const int half = lq & 1, hi = lq >> 1;
#pragma unroll
      for (int c = 0; c < 14; ++c) {
        const int t0 = 2 * c, t1 = 2 * c + 1 < 27 ? 2 * c + 1 : 26;      // phantom tap 27: real data x zero weights
        bf16x8 af[C::MT];
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
          union { uint4 u; bf16x8 v; } f;
          f.u = make_uint4(0u, 0u, 0u, 0u);
          if (2 * c + hi < 27) {
            if (a.cin_real == 4) {
              if (half == 0) {
                const uint2 t = *reinterpret_cast<const uint2*>(wrow + (size_t)m * 16 * 27 * 4 + (t0 + hi) * 4);
                f.u.x = t.x; f.u.y = t.y;
              }
            } else {
              f.u = *reinterpret_cast<const uint4*>(wrow + (size_t)m * 16 * 27 * CIN + (t0 + hi) * 16 + half * 8);
            }
          }
          af[m] = f.v;
        }
        const int toff = hi ? tap_off<C::HH, C::HW>(t1) : tap_off<C::HH, C::HW>(t0);
#pragma unroll
        for (int j = 0; j < C::NT; ++j) {
          const int u = ubase[j] + toff;
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(smem + u * 32 + half * 16);
#pragma unroll
          for (int m = 0; m < C::MT; ++m)
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr, acc[m][j], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: + bias (+ residual), 4 consecutive cout of one voxel per lane -> 8-byte stores
  float f0[C::MT][4], f1[C::MT][4], scs[C::MT][4], shs[C::MT][4];
  if (STATS) {
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
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
  uint2 xref[C::MT][C::NT];
  if (STATS == 2) {
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const int g = wave * C::NT + j;
      const int ltw = g % C::NTW, lth = (g / C::NTW) % TH, ltd = g / (C::NTW * TH);
      const int od = od0 + ltd, oh = oh0 + lth, ow = ow0 + ltw * 16 + l15;
      const bool in = od < a.OD && oh < a.OH && ow < a.OW;
      const size_t vox = in ? (size_t)((n * a.OD + od) * a.OH + oh) * a.OW + ow : 0;
#pragma unroll
      for (int m = 0; m < C::MT; ++m)
        xref[m][j] = *reinterpret_cast<const uint2*>(a.st.xref + vox * a.Cout + co_base + m * 16 + lq * 4);
    }
  }
#pragma unroll
  for (int j = 0; j < C::NT; ++j) {
    const int g = wave * C::NT + j;
    const int ltw = g % C::NTW, lth = (g / C::NTW) % TH, ltd = g / (C::NTW * TH);
    const int od = od0 + ltd, oh = oh0 + lth, ow = ow0 + ltw * 16 + l15;
    if (od >= a.OD || oh >= a.OH || ow >= a.OW) continue;
    const size_t vox = (size_t)((n * a.OD + od) * a.OH + oh) * a.OW + ow;
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
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
    stats_finish_dpp<C::MT>(f0, f1, reinterpret_cast<double*>(smem),
                        a.st.partial + (((size_t)n * a.st.chunks + chunk) * a.Cout + co_base) * 2);
  }
}

template <int CIN, int COUT_T, int TD, int TH, int TW, int S, bool PRO, int STATS>
static int launch_conv3_k(const Conv3Args& a, dim3 grid, hipStream_t s) {
  using C = Cfg<CIN, COUT_T, TD, TH, TW, S>;
  auto k = k_conv3_mfma<CIN, COUT_T, TD, TH, TW, S, PRO, STATS>;
  if (C::LDS_BYTES > 65536) {
    static bool once = false;
    if (!once) { if (check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES))) return DEVIS_ERR_HIP; once = true; }
  }
  hipLaunchKernelGGL(k, grid, dim3(256), C::LDS_BYTES, s, a);
  return launch_status();
}

// ST: 0 = this shape never gathers statistics, 1 = forward moments (mode 1), 2 = mode 1 and, without a
// prologue, the norm-backward moments (mode 2).  plan_chunks != null: only report the chunks per sample.
template <int CIN, int COUT_T, int TD, int TH, int TW, int S, int ST = 0>
static int launch_conv3(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
                        const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks,
                        hipStream_t s) {
  using C = Cfg<CIN, COUT_T, TD, TH, TW, S>;
  Conv3Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.sc = sc; a.sh = sh; a.res = (const bf16*)res; a.y = (bf16*)y;
  a.N = (int)d->N; a.ID = (int)d->ID; a.IH = (int)d->IH; a.IW = (int)d->IW;
  a.OD = (int)d->OD; a.OH = (int)d->OH; a.OW = (int)d->OW; a.Cout = (int)d->Cout;
  a.cin_real = (int)d->C1;
  a.ntd = (int)cdiv(d->OD, TD); a.nth = (int)cdiv(d->OH, TH); a.ntw = (int)cdiv(d->OW, TW);
  const int64_t tiles = (int64_t)a.N * a.ntd * a.nth * a.ntw;
  if (tiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  if (plan_chunks) {
    const bool can = ST >= 1 && !(st && st->mode == 2 && (ST < 2 || sc));
    *plan_chunks = can ? (int64_t)a.ntd * a.nth * a.ntw : 0;
    return DEVIS_OK;
  }
  const dim3 grid((unsigned)tiles, (unsigned)(d->Cout / COUT_T));
  static_assert(C::LDS_BYTES >= 4 * C::MT * 32 * 8, "LDS too small for the statistics reduction");
  a.st = StatsArgs{};
  const int mode = st ? st->mode : 0;
  if (mode) {
    if (ST == 0 || (mode == 2 && (ST < 2 || sc))) return DEVIS_ERR_UNSUPPORTED;
    a.st = *st;
    if (a.st.chunks != a.ntd * a.nth * a.ntw || a.st.C != a.Cout) return DEVIS_ERR_BAD_ARG;
  }
  if (mode == 1 && ST >= 1) {
    if (sc) return launch_conv3_k<CIN, COUT_T, TD, TH, TW, S, true, ST >= 1 ? 1 : 0>(a, grid, s);
    return launch_conv3_k<CIN, COUT_T, TD, TH, TW, S, false, ST >= 1 ? 1 : 0>(a, grid, s);
  }
  if (mode == 2 && ST >= 2) return launch_conv3_k<CIN, COUT_T, TD, TH, TW, S, false, ST >= 2 ? 2 : 0>(a, grid, s);
  if (sc) return launch_conv3_k<CIN, COUT_T, TD, TH, TW, S, true, 0>(a, grid, s);
  return launch_conv3_k<CIN, COUT_T, TD, TH, TW, S, false, 0>(a, grid, s);
}

int conv3_c16_march(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
                    const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks, hipStream_t s);

// conv_deepk.hip: the 64- / 128-channel layers, weights stationary in registers and K split over the waves
int conv3_deepk(const devis_conv_desc*, const void*, const void*, const float*, const float*, const float*, const void*, void*,
                const StatsArgs*, int64_t*, hipStream_t);

int conv3_c32_march(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
                    const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks, hipStream_t s);

int conv3_s2_c16_march(const devis_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                       const StatsArgs* st, int64_t* plan_chunks, hipStream_t s);

static bool is_3x3x3(const devis_conv_desc* d) {
  return d->KD == 3 && d->KH == 3 && d->KW == 3 && d->PD == 1 && d->PH == 1 && d->PW == 1 && d->SD == d->SH && d->SH == d->SW;
}
static bool fits_int(const devis_conv_desc* d) {
  const int64_t big = d->N * d->ID * d->IH * d->IW * (d->C1 + d->C2);
  const int64_t bigo = d->N * d->OD * d->OH * d->OW * d->Cout;
  return big < (int64_t)1 << 32 && bigo < (int64_t)1 << 40 && d->N * d->ID * d->IH * d->IW < 0x7fffffff;
}

// st: statistics to gather in the epilogue (null / mode 0: none).  plan_chunks != null: launch nothing, only
// report how many statistics chunks per sample the kernel for this shape leaves (0: it cannot gather them).
int conv_fwd_mfma(const devis_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                  const float* sc, const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks,
                  hipStream_t s) {
  if (plan_chunks) *plan_chunks = 0;
  if (d->dtype != DEVIS_BF16 || d->C2 != 0 || x2 || !is_3x3x3(d) || !fits_int(d)) return DEVIS_ERR_UNSUPPORTED;
  const int ci = (int)d->C1, co = (int)d->Cout, st_ = d->SD;
#define CONV3(...) launch_conv3<__VA_ARGS__>(d, x1, w, bias, sc, sh, res, y, st, plan_chunks, s)
  if (st_ == 1) {
    if (co == 16 && (ci == 16 || (ci == 4 && !sc))) return conv3_c16_march(d, x1, w, bias, sc, sh, res, y, st, plan_chunks, s);
    // 64 / 128 channels: their packed weights are in fragment order (csrc/layout.hip), which only this kernel reads
    if (ci == co && (ci == 64 || ci == 128)) return conv3_deepk(d, x1, w, bias, sc, sh, res, y, st, plan_chunks, s);
    if (ci == 32 && co == 32) return conv3_c32_march(d, x1, w, bias, sc, sh, res, y, st, plan_chunks, s);
  } else if (st_ == 2) {
    // 32 -> 64: the (2,4,16) tile re-reads less halo than (2,2,16): 36 vs 41 us at 64^3 (tools/time_s2_fwd.py);
    // for 16 -> 32 the larger (4,4,16) tile was measured at 145 vs 86 us (one block per CU) and is not built
    // 16 -> 32 (EnDown1), 32 -> 64 (EnDown2): plane march (conv_s2_march.hip); with a prologue, a residual or
    // norm-backward statistics the tile kernel below
    if (((ci == 16 && co == 32) || (ci == 32 && co == 64)) && !sc && !res && !(st && st->mode == 2))
      return conv3_s2_c16_march(d, x1, w, bias, y, st, plan_chunks, s);
    if (ci == 16 && co == 32) return CONV3(16, 32, 2, 4, 16, 2, 1);
    if (ci == 32 && co == 64) return CONV3(32, 64, 2, 4, 16, 2, 1);
    if (ci == 64 && co == 128) return CONV3(64, 64, 2, 2, 16, 2, 1);
  }
#undef CONV3
  return DEVIS_ERR_UNSUPPORTED;
}

// dgrad of a stride-1 3x3x3 conv = the same conv with flipped taps and swapped channel roles:
// w_dgrad is [Cin][27][Cout] with tap t' = 26 - t (devis_pack_conv_weight_dgrad).
int conv_dgrad_mfma(const devis_conv_desc* d, const void* dy, const void* w_dgrad, const void* res, void* dx1, void* dx2,
                    const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  if (plan_chunks) *plan_chunks = 0;
  if (!w_dgrad || d->dtype != DEVIS_BF16 || d->C2 != 0 || dx2 || !is_3x3x3(d) || d->SD != 1 || !fits_int(d)) return DEVIS_ERR_UNSUPPORTED;
  devis_conv_desc t = *d;
  t.C1 = d->Cout; t.Cout = d->C1;
  t.ID = d->OD; t.IH = d->OH; t.IW = d->OW; t.OD = d->ID; t.OH = d->IH; t.OW = d->IW;
  return conv_fwd_mfma(&t, dy, nullptr, w_dgrad, nullptr, nullptr, nullptr, res, dx1, st, plan_chunks, s);
}

}  // namespace devis
