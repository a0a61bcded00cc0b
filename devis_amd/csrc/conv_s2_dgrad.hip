// dgrad of the stride-2 3x3x3 'EnDown' convs (UNet3DZoo.py:59-67), bf16 channels-last, gfx950.
//
//   dx[i][ci] = sum over taps t and coarse voxels o with 2o + t - 1 = i of  W[co][t][ci] * dy[o][co]
//
// Per dimension an even fine index i = 2j takes tap 1 from coarse j; an odd one, i = 2j + 1, takes tap 0 from
// coarse j + 1 and tap 2 from coarse j.  So every one of the 27 taps belongs to exactly one of the 8 output
// parity classes and reads the coarse tile at one of 8 shifts (0/1 per dimension).  A block owns a 2x4x16
// coarse tile (= 4x8x32 fine voxels): per 32-channel K chunk of dy it stages the (3x5x17)-voxel coarse halo tile
// and the chunk's [27][MT*16][32] weight slab in LDS with bulk coalesced loads (dy is read from HBM once, not
// once per tap), the next chunk is prefetched into registers, and a wave (two coarse rows) keeps its 8 shifted
// B fragments in registers: 27*MT + 16 LDS reads feed 54*MT MFMAs per chunk.  In the epilogue lane pairs swap
// the two W-parity classes so that every lane stores 16 contiguous bytes and a store instruction covers 32
// consecutive fine voxels.  Optional residual (the skip gradient) is added there.
#include "common.h"
#include "mfma_common.h"

namespace devis {

struct S2Args {
  const bf16* dy;     // [N][OD][OH][OW][CO]
  const bf16* w;      // tap-flipped dgrad packing [Cin][27][CO]
  const bf16* res;    // like dx, or null
  bf16* dx;           // [N][ID][IH][IW][Cin]
  int N, ID, IH, IW, OD, OH, OW, Cin, CO;
  int ntd, nth, ntw;
};

template <int MT>
__global__ void __launch_bounds__(256)
k_s2_dgrad(S2Args a) {
  constexpr int TD = 2, TH = 4, HD = TD + 1, HH = TH + 1, HW = 17, HV = HD * HH * HW;      // 255 coarse voxels
  constexpr int XPT = (HV * 4 + 255) / 256;            // 4
  constexpr int WROWS = 27 * MT * 16;
  constexpr int WPT = (WROWS * 4 + 255) / 256;
  constexpr int XB = XPT * 256 * 16;
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
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * 16;          // coarse origin
  const int ci_base = blockIdx.y * (MT * 16);
  const int piece = tid & 3;

  unsigned xg[XPT], xl[XPT];
  bool xok[XPT];
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int i = tid + k * 256;
    const int u = i >> 2;
    const int hw = u % HW, hh = (u / HW) % HH, hd = u / (HW * HH);
    const int od = d0 + hd, oh = h0 + hh, ow = w0 + hw;
    xok[k] = i < HV * 4 && od < a.OD && oh < a.OH && ow < a.OW;
    xg[k] = xok[k] ? (unsigned)((((n * a.OD + od) * a.OH + oh) * a.OW + ow) * a.CO + piece * 8) : 0u;
    xl[k] = (unsigned)(u * 64 + ((piece ^ ((u >> 1) & 3)) << 4));
  }
  unsigned wg[WPT], wl[WPT];
  bool wok[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int i = tid + k * 256;
    const int row = i >> 2;                     // tap * (MT*16) + ci
    const int t = row / (MT * 16), ci = row % (MT * 16);
    wok[k] = i < WROWS * 4;
    wg[k] = wok[k] ? (unsigned)(((ci_base + ci) * 27 + (26 - t)) * a.CO + piece * 8) : 0u;
    wl[k] = (unsigned)(row * 64 + ((piece ^ ((row >> 1) & 3)) << 4));
  }
  auto fetch = [&](int cc, u32x4 (&xr)[XPT], u32x4 (&wr)[WPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < XPT; ++k) xr[k] = *reinterpret_cast<const u32x4*>(a.dy + xg[k] + cc * 32);
#pragma unroll
    for (int k = 0; k < WPT; ++k) wr[k] = *reinterpret_cast<const u32x4*>(a.w + wg[k] + cc * 32);
  };
  auto stage = [&](u32x4 (&xr)[XPT], u32x4 (&wr)[WPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      u32x4 v = xr[k];
      if (!xok[k]) v = (u32x4){0u, 0u, 0u, 0u};
      if (k < XPT - 1 || tid + k * 256 < HV * 4) *reinterpret_cast<u32x4*>(xs + xl[k]) = v;
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k)
      if (k < WPT - 1 || wok[k]) *reinterpret_cast<u32x4*>(ws + wl[k]) = wr[k];
  };

  f32x4 acc[8][MT][2];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[c][m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this wave's two coarse rows: r = 2*wave + j -> (d, h) = (r / TH, r % TH); LDS index of the lane's voxel at shift 0
  int ubase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wave * 2 + j;
    ubase[j] = ((r / TH) * HH + r % TH) * HW + l15;
  }

  const int nch = a.CO >> 5;
  u32x4 xr[XPT], wr[WPT];
  fetch(0, xr, wr);
#pragma unroll 1
  for (int cc = 0; cc < nch; ++cc) {
    __syncthreads();
    stage(xr, wr);
    __syncthreads();
    if (cc + 1 < nch) fetch(cc + 1, xr, wr);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 bf[2][8];                      // [row][shift = sd*4 + sh*2 + sw]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int sft = 0; sft < 8; ++sft) {
        const int u = ubase[j] + (((sft >> 2) & 1) * HH + ((sft >> 1) & 1)) * HW + (sft & 1);
        bf[j][sft] = *reinterpret_cast<const bf16x8*>(xs + u * 64 + ((lq ^ ((u >> 1) & 3)) << 4));
      }
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
      const int cls = (kd != 1) * 4 + (kh != 1) * 2 + (kw != 1);           // output parity (pd, ph, pw)
      const int sft = (kd == 0) * 4 + (kh == 0) * 2 + (kw == 0);           // coarse shift
      bf16x8 af[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = t * (MT * 16) + m * 16 + l15;
        af[m] = *reinterpret_cast<const bf16x8*>(ws + row * 64 + ((lq ^ ((row >> 1) & 3)) << 4));
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[cls][m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[j][sft], acc[cls][m][j], 0, 0, 0);
    }
  }

  // ---- epilogue.  A lane's accumulators are 4 consecutive cin (lq*4..) of coarse voxel l15, i.e. of the fine voxels
  // 2w and 2w+1 for the two W-parity classes.  Lane pairs (lq, lq^1) swap one class each, so that every lane ends
  // with 8 consecutive cin of ONE fine voxel (even lq: 2w, odd lq: 2w+1): 16-byte residual loads and stores, and a
  // store instruction covers 32 consecutive fine voxels.
  const int odd = lq & 1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wave * 2 + j;
    const int cd = d0 + r / TH, chh = h0 + r % TH, fw = 2 * (w0 + l15) + odd;
    uint4 rv[4][MT];
    bool ok[4];
    size_t off[4];
#pragma unroll
    for (int pdh = 0; pdh < 4; ++pdh) {              // all residual loads of the row first, then the stores
      const int fd = 2 * cd + (pdh >> 1), fh = 2 * chh + (pdh & 1);
      ok[pdh] = fd < a.ID && fh < a.IH && fw < a.IW;
      off[pdh] = ok[pdh] ? ((size_t)((n * a.ID + fd) * a.IH + fh) * a.IW + fw) * a.Cin + ci_base + (lq & 2) * 4 : 0;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        rv[pdh][m] = make_uint4(0u, 0u, 0u, 0u);
        if (a.res && ok[pdh]) rv[pdh][m] = *reinterpret_cast<const uint4*>(a.res + off[pdh] + m * 16);
      }
    }
#pragma unroll
    for (int pdh = 0; pdh < 4; ++pdh) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 v0 = acc[pdh * 2][m][j], v1 = acc[pdh * 2 + 1][m][j];
        float lo[4], hi[4];                         // channels +0..3 and +4..7 of this lane's voxel
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float keep = odd ? v1[k] : v0[k];
          const float recv = __shfl_xor(odd ? v0[k] : v1[k], 16, 64);
          lo[k] = odd ? recv : keep;
          hi[k] = odd ? keep : recv;
        }
        if (!ok[pdh]) continue;
        const uint4 rr = rv[pdh][m];
        uint4 o;
        o.x = pack_bf2(lo[0] + bf_lo(rr.x), lo[1] + bf_hi(rr.x));
        o.y = pack_bf2(lo[2] + bf_lo(rr.y), lo[3] + bf_hi(rr.y));
        o.z = pack_bf2(hi[0] + bf_lo(rr.z), hi[1] + bf_hi(rr.z));
        o.w = pack_bf2(hi[2] + bf_lo(rr.w), hi[3] + bf_hi(rr.w));
        *reinterpret_cast<uint4*>(a.dx + off[pdh] + m * 16) = o;
      }
    }
  }
}

template <int MT>
static int launch_s2(const S2Args& a, hipStream_t s) {
  constexpr int LDS = 4 * 256 * 16 + 27 * MT * 16 * 64;
  auto k = k_s2_dgrad<MT>;
  static bool attr_done = false;
  if (!attr_done) { if (check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS))) return DEVIS_ERR_HIP; attr_done = true; }
  const int64_t tiles = (int64_t)a.N * a.ntd * a.nth * a.ntw;
  const dim3 grid((unsigned)tiles, (unsigned)(a.Cin / (MT * 16)));
  hipLaunchKernelGGL(k, grid, dim3(256), LDS, s, a);
  return launch_status();
}

// dgrad of a k3-s2-p1 conv; w_dgrad = tap-flipped [Cin][27][Cout]; Cout % 32 == 0, Cin % 16 == 0
int conv3s2_dgrad_lds(const devis_conv_desc* d, const void* dy, const void* w_dgrad, const void* res, void* dx, hipStream_t s) {
  if (!w_dgrad || d->dtype != DEVIS_BF16 || d->C2 != 0) return DEVIS_ERR_UNSUPPORTED;
  if (!(d->KD == 3 && d->KH == 3 && d->KW == 3 && d->PD == 1 && d->PH == 1 && d->PW == 1 && d->SD == 2 && d->SH == 2 && d->SW == 2)) return DEVIS_ERR_UNSUPPORTED;
  if (d->Cout % 32 || d->C1 % 16) return DEVIS_ERR_UNSUPPORTED;
  if (d->N * d->ID * d->IH * d->IW * d->C1 >= ((int64_t)1 << 31) || d->N * d->OD * d->OH * d->OW * d->Cout >= ((int64_t)1 << 31)) return DEVIS_ERR_UNSUPPORTED;
  if (d->C1 * 27 * d->Cout >= ((int64_t)1 << 31)) return DEVIS_ERR_UNSUPPORTED;
  S2Args a;
  a.dy = (const bf16*)dy; a.w = (const bf16*)w_dgrad; a.res = (const bf16*)res; a.dx = (bf16*)dx;
  a.N = (int)d->N; a.ID = (int)d->ID; a.IH = (int)d->IH; a.IW = (int)d->IW;
  a.OD = (int)d->OD; a.OH = (int)d->OH; a.OW = (int)d->OW; a.Cin = (int)d->C1; a.CO = (int)d->Cout;
  // every fine voxel belongs to the tile of coarse floor(i / 2): tiles cover ceil(I / 2) coarse positions per dim
  a.ntd = (int)cdiv(cdiv(d->ID, 2), 2); a.nth = (int)cdiv(cdiv(d->IH, 2), 4); a.ntw = (int)cdiv(cdiv(d->IW, 2), 16);
  const int64_t tiles = (int64_t)a.N * a.ntd * a.nth * a.ntw;
  if (tiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  // two cin tiles per block when that still fills the chip
  if (a.Cin % 32 == 0 && tiles * (a.Cin / 32) >= 512) return launch_s2<2>(a, s);
  return launch_s2<1>(a, s);
}

}  // namespace devis
