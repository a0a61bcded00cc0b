// Plane-marching MFMA 3x3x3 convolution for the 32-channel layers (64^3 level of the U-Net; bf16,
// channels-last, stride 1, gfx950): conv_march.hip's design with K = 32 input channels per tap.
//
//   * a block owns 16 of the 32 output channels of an (8 x 16) column and walks it along D; a ring of
//     R = 4 input planes (halo included) lives in LDS, plane d+2 is in flight from HBM in registers
//     while the MFMAs of output plane d run.  The two blocks that share a column (the two cout halves)
//     sit on the same XCD, so the second one finds the planes in L2.
//   * the 27 x 16 x 32 weights of the block are 27 A fragments = 108 VGPRs, held for the whole walk
//     (the generic kernel re-fetched them from L1/L2 for every tile).
//   * one MFMA 16x16x32 = one tap: M = 16 cout, N = 16 voxels along W, K = 32 ci.
//   * LDS plane layout is PIECE-PLANAR: the four 16-byte pieces (8 channels each) of a voxel live in four
//     sub-planes PQ bytes apart (PQ a multiple of 256), voxel pitch 16 bytes inside each.  A lane's B
//     fragment is then one ds_read_b128 at  lane base + compile-time (tap, tile) offset:  conflict-free
//     and with NO address arithmetic in the MFMA loop.  (The XOR-swizzled 64-byte voxel rows of the
//     generic kernel cost ~6 VALU instructions per read, more issue time than the MFMA they feed.)
//   * GroupNorm/BatchNorm affine + ReLU on the way into LDS, bias / residual / norm statistics
//     (mfma_common.h) on the way out; 8-byte channels-last stores.
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {

struct March32Args {
  const bf16* x;
  const bf16* w;      // [32][27][32]
  const float* bias;  // [32] or null
  const float* sc;    // [N][32] or null
  const float* sh;
  const bf16* res;    // like y, or null
  bf16* y;
  int N, D, H, W;
  int nth, ntw, ndc, DL;
  int paired;         // block -> (tile, cout half) map: 1 = halves 8 block ids apart (same XCD), 0 = adjacent ids
  StatsArgs st;
};

template <bool PRO, bool RES, bool FULL, int STATS>
__global__ void __launch_bounds__(256)
k_conv3_c32_march(March32Args a) {
  constexpr int TH = 8, TW = 16, C = 32;
  constexpr int HH = TH + 2, HW = TW + 2, HP = HH * HW;   // 180 halo voxels per plane
  constexpr int NPT = (HP * 4 + 255) / 256;               // 16-byte pieces per thread per plane (3)
  constexpr int PQ = ((HP * 16 + 255) / 256) * 256;       // bytes per piece sub-plane (3072)
  constexpr int PLB = 4 * PQ;                             // bytes per LDS plane (12 KB)
  constexpr int R = 4;
  constexpr int NT = TH / 4;                              // N tiles (rows) per wave per plane: 2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;

  int tile, half;
  {
    const int bid = blockIdx.x;
    if (a.paired) {
      // ids b and b+8 share an XCD: give them the two cout halves of one tile; tiles themselves are laid out
      // XCD-contiguous (each XCD's L2 sees a compact part of the volume)
      const int ntile = gridDim.x >> 1;                   // host: ntile % 8 == 0
      const int xcd = bid & 7, local = bid >> 4;
      half = (bid >> 3) & 1;
      tile = xcd * (ntile >> 3) + local;
    } else {
      tile = bid >> 1;
      half = bid & 1;
    }
  }
  const int co0 = half * 16;
  const int twi = tile % a.ntw;
  const int thi = (tile / a.ntw) % a.nth;
  const int dci = (tile / (a.ntw * a.nth)) % a.ndc;
  const int n = tile / (a.ntw * a.nth * a.ndc);
  const int oh0 = thi * TH, ow0 = twi * TW;
  const int d0 = dci * a.DL;
  const int d1 = min(d0 + a.DL, a.D);
  const size_t plane = (size_t)a.H * a.W;

  // ---- A fragments: tap t, row = cout co0 + l15, k = ci lq*8 .. lq*8+7
  bf16x8 af[27];
  {
    const bf16* wrow = a.w + (size_t)(co0 + l15) * 27 * C + lq * 8;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      union { uint4 u; bf16x8 v; } f;
      f.u = *reinterpret_cast<const uint4*>(wrow + t * C);
      af[t] = f.v;
    }
  }

  // ---- per-thread load slots of one halo plane (the same for every plane): piece pc of voxel u
  unsigned goff[NPT], loff[NPT];
  bool ok[NPT];
  const int pc = tid & 3;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int i = tid + k * 256;
    const int u = i >> 2;
    const int hh = u / HW, hw = u - hh * HW;
    const int ih = oh0 - 1 + hh, iw = ow0 - 1 + hw;
    ok[k] = i < HP * 4 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
    goff[k] = ok[k] ? (unsigned)((ih * a.W + iw) * C + pc * 8) : 0u;
    loff[k] = (unsigned)(pc * PQ + (i < HP * 4 ? u : 0) * 16);
  }
  float scv[8], shv[8];
  if (PRO) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      scv[j] = a.sc[n * C + pc * 8 + j];
      shv[j] = a.sh[n * C + pc * 8 + j];
    }
  }

  auto load_plane = [&](int id, u32x4 (&rg)[NPT]) __attribute__((always_inline)) {
    const int idc = min(max(id, 0), a.D - 1);
    const bf16* xp = a.x + ((size_t)n * a.D + idc) * plane * C;
#pragma unroll
    for (int k = 0; k < NPT; ++k) rg[k] = *reinterpret_cast<const u32x4*>(xp + goff[k]);
  };
  auto store_plane = [&](int id, int slot, u32x4 (&rg)[NPT]) __attribute__((always_inline)) {
    const bool pin = id >= 0 && id < a.D;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      u32x4 v = rg[k];
      if (PRO) v = affine_relu8(v, scv, shv);
      if (!(pin && ok[k])) v = (u32x4){0u, 0u, 0u, 0u};
      if (k < NPT - 1 || tid + k * 256 < HP * 4) *reinterpret_cast<u32x4*>(smem + slot * PLB + loff[k]) = v;
    }
  };

  // ---- fill the ring with planes d0-1, d0, d0+1
  {
    u32x4 r0[NPT], r1[NPT], r2[NPT];
    load_plane(d0 - 1, r0);
    load_plane(d0, r1);
    load_plane(d0 + 1, r2);
    store_plane(d0 - 1, 0, r0);
    store_plane(d0, 1, r1);
    store_plane(d0 + 1, 2, r2);
  }
  __syncthreads();

  // lane part of the B-fragment address: piece lq, voxel (wave's first row, column l15)
  const int lanebase = lq * PQ + ((wave * NT) * HW + l15) * 16;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + co0 + lq * 4);

  // per-lane output slots inside a plane (element offsets)
  unsigned ooff[NT];
  bool oin[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int oh = oh0 + wave * NT + j, ow = ow0 + l15;
    oin[j] = FULL || (oh < a.H && ow < a.W);
    ooff[j] = oin[j] ? (unsigned)((oh * a.W + ow) * C + co0 + lq * 4) : (unsigned)(co0 + lq * 4);
  }

  float f0[1][4] = {{0.f, 0.f, 0.f, 0.f}}, f1[1][4] = {{0.f, 0.f, 0.f, 0.f}};
  float scs[4] = {0.f, 0.f, 0.f, 0.f}, shs[4] = {0.f, 0.f, 0.f, 0.f};
  if (STATS == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { scs[j] = a.st.sc[n * C + co0 + lq * 4 + j]; shs[j] = a.st.sh[n * C + co0 + lq * 4 + j]; }
  }

  const int np = d1 - d0;
  for (int it = 0; it < np; ++it) {
    const int od = d0 + it;
    u32x4 pre[NPT];
    load_plane(od + 2, pre);                    // consumed by iteration it+1 (harmless on the last one)
    const size_t pbase = ((size_t)n * a.D + od) * plane * C;
    uint2 rv[NT], xv[NT];
    if (RES) {
      const bf16* rp = a.res + pbase;
#pragma unroll
      for (int j = 0; j < NT; ++j) rv[j] = *reinterpret_cast<const uint2*>(rp + ooff[j]);
    }
    if (STATS == 2) {
      const bf16* xp = a.st.xref + pbase;
#pragma unroll
      for (int j = 0; j < NT; ++j) xv[j] = *reinterpret_cast<const uint2*>(xp + ooff[j]);
    }
    __builtin_amdgcn_sched_barrier(0);

    const char* sb[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) sb[kd] = smem + ((it + kd) & (R - 1)) * PLB + lanebase;

    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // B fragments double-buffered over taps: the reads of tap t+1 are in flight under the MFMAs of tap t
    auto read_b = [&](int t, bf16x8 (&b)[NT]) __attribute__((always_inline)) {
      const char* p = sb[t / 9] + (((t / 3) % 3) * HW + t % 3) * 16;
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(p + j * HW * 16);
    };
    bf16x8 bA[NT], bB[NT];
    read_b(0, bA);
#define M32_INTERLEAVE                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < NT; ++i_) {       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      \
  }
#pragma unroll
    for (int t = 0; t < 27; t += 2) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bA[j], acc[j], 0, 0, 0);
      if (t + 1 < 27) read_b(t + 1, bB);
      M32_INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < 27) {
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t + 1], bB[j], acc[j], 0, 0, 0);
        if (t + 2 < 27) read_b(t + 2, bA);
        M32_INTERLEAVE
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef M32_INTERLEAVE

    // ---- epilogue of plane od
    bf16* yp = a.y + pbase;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float v0 = acc[j][0] + bv.x, v1 = acc[j][1] + bv.y, v2 = acc[j][2] + bv.z, v3 = acc[j][3] + bv.w;
      if (RES) { v0 += bf_lo(rv[j].x); v1 += bf_hi(rv[j].x); v2 += bf_lo(rv[j].y); v3 += bf_hi(rv[j].y); }
      uint2 o;
      o.x = pack_bf2(v0, v1);
      o.y = pack_bf2(v2, v3);
      if (FULL || oin[j]) *reinterpret_cast<uint2*>(yp + ooff[j]) = o;
      if (STATS && (FULL || oin[j])) stats_accum<STATS>(o, STATS == 2 ? xv[j] : o, scs, shs, f0[0], f1[0]);
    }
    __builtin_amdgcn_sched_barrier(0);

    store_plane(od + 2, (it + 3) & (R - 1), pre);
    __syncthreads();
  }
  if (STATS) {
    const int chunk = tile % (a.ntw * a.nth * a.ndc);
    stats_finish<1>(f0, f1, reinterpret_cast<double*>(smem),
                    a.st.partial + (((size_t)n * a.st.chunks + chunk) * C + co0) * 2);
  }
}

template <bool PRO, bool RES, int STATS>
static void launch_m32_v(bool full, dim3 grid, hipStream_t s, const March32Args& a) {
  constexpr int LDS = 4 * 4 * ((((8 + 2) * (16 + 2)) * 16 + 255) / 256) * 256;
  if (full) hipLaunchKernelGGL((k_conv3_c32_march<PRO, RES, true, STATS>), grid, dim3(256), LDS, s, a);
  else hipLaunchKernelGGL((k_conv3_c32_march<PRO, RES, false, STATS>), grid, dim3(256), LDS, s, a);
}

// stride-1 3x3x3, Cin = Cout = 32, bf16; anything else is unsupported.
// plan_chunks != null: only report the statistics chunks per sample this launch would leave.
int conv3_c32_march(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
                    const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  if (d->dtype != DEVIS_BF16 || d->Cout != 32 || d->C1 != 32 || d->C2 != 0 || d->SD != 1) return DEVIS_ERR_UNSUPPORTED;
  if (d->OD != d->ID || d->OH != d->IH || d->OW != d->IW) return DEVIS_ERR_UNSUPPORTED;
  if (d->IH * d->IW * 32 >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  March32Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.sc = sc; a.sh = sh; a.res = (const bf16*)res; a.y = (bf16*)y;
  a.N = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW;
  a.nth = (int)cdiv(a.H, 8); a.ntw = (int)cdiv(a.W, 16);
  // D chunk: enough blocks (two per tile) to fill 256 CUs about twice; each chunk re-reads 2 halo planes
  int dl = 32;
  const int64_t cols = (int64_t)a.N * a.nth * a.ntw;
  while (dl > 4 && cols * 2 * cdiv(a.D, dl) < 512) dl >>= 1;
  a.DL = dl;
  a.ndc = (int)cdiv(a.D, dl);
  const int64_t tiles = cols * a.ndc;
  if (tiles * 2 > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const int mode = st ? st->mode : 0;
  if (plan_chunks) {
    const bool can = !(mode == 2 && (sc || res));
    *plan_chunks = can ? (int64_t)a.ndc * a.nth * a.ntw : 0;
    return DEVIS_OK;
  }
  a.paired = tiles % 8 == 0 ? 1 : 0;
  const dim3 grid((unsigned)(tiles * 2));
  const bool full = a.H % 8 == 0 && a.W % 16 == 0;
  a.st = StatsArgs{};
  if (mode) {
    a.st = *st;
    if (a.st.chunks != a.ndc * a.nth * a.ntw || a.st.C != 32) return DEVIS_ERR_BAD_ARG;
    if (mode == 2 && (sc || res)) return DEVIS_ERR_UNSUPPORTED;
  }
#define M32(P, R_) \
  do { if (mode == 1) launch_m32_v<P, R_, 1>(full, grid, s, a); else launch_m32_v<P, R_, 0>(full, grid, s, a); } while (0)
  if (mode == 2) launch_m32_v<false, false, 2>(full, grid, s, a);
  else if (sc) { if (res) M32(true, true); else M32(true, false); }
  else { if (res) M32(false, true); else M32(false, false); }
#undef M32
  return launch_status();
}

}  // namespace devis
