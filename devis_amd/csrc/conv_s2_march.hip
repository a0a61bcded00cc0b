// Plane-marching MFMA 3x3x3 convolution with stride 2, 16 -> 32 channels (EnDown1 of the 3-D U-Net,
// UNet3DZoo.py:59-67; bf16 channels-last, gfx950).  The layer reads a full-resolution 16-channel tensor once and
// writes an eighth of the voxels: HBM-bound on the read (13.5 FLOP/B of input), so it is organised like
// conv_march.hip, not like the tile kernels of conv_mfma.hip (which re-read 1.45x of the input as halo and
// re-load the weights per 128 outputs: 85 us for 134 MB):
//   * a block owns an (8 x 16) column of OUTPUT voxels and walks it along D; output plane od needs input planes
//     2od-1, 2od, 2od+1, so a ring of three input planes (17 x 33 voxels each, halo included) lives in LDS and
//     two new planes per step are in flight from HBM in registers while the MFMAs run;
//   * an LDS plane is stored with its even and odd columns apart ([row][parity][col/2]): the stride-2 reads of
//     a tap then walk consecutive voxels (32-byte pitch, conflict-free ds_read_b128) like the stride-1 kernel's;
//   * all 27*16*32 weights are A fragments in 112 VGPRs for the whole walk; K = (2 taps x 16 ci) per MFMA.
// Epilogue: + bias, 8-byte channels-last stores, and the following GroupNorm's moments (statistics mode 1).
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {

struct S2MarchArgs {
  const bf16* x;      // [N][ID][IH][IW][16]
  const bf16* w;      // [Cout][27][Cin]
  const float* bias;  // [Cout] or null
  bf16* y;            // [N][OD][OH][OW][Cout]
  int N, ID, IH, IW, OD, OH, OW;
  int nth, ntw, ndc, DL;
  StatsArgs st;
};

template <bool FULL, int STATS>
__global__ void __launch_bounds__(256, 2)
k_conv3_s2_c16_march(S2MarchArgs a) {
  constexpr int TH = 8, TW = 16;
  constexpr int HH = 2 * TH + 1, HW = 2 * TW + 1, HP = HH * HW;     // 17 x 33 = 561 voxels
  constexpr int NPT = (HP * 2 + 255) / 256;                        // 5 sixteen-byte pieces per thread per plane
  constexpr int PLB = NPT * 256 * 16;                              // 20 KB per LDS plane (tail padding: no store guard)
  constexpr int R = 3;
  constexpr int NT = 2, MT = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4, half = lq & 1, hi = lq >> 1;

  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int twi = tile % a.ntw;
  const int thi = (tile / a.ntw) % a.nth;
  const int dci = (tile / (a.ntw * a.nth)) % a.ndc;
  const int n = tile / (a.ntw * a.nth * a.ndc);
  const int oh0 = thi * TH, ow0 = twi * TW;
  const int d0 = dci * a.DL;
  const int d1 = min(d0 + a.DL, a.OD);
  const size_t iplane = (size_t)a.IH * a.IW, oplane = (size_t)a.OH * a.OW;

  // ---- A fragments: chunk c = taps (2c, 2c+1) x 16 ci; lanes 0-31 tap 2c, lanes 32-63 tap 2c+1; two cout tiles
  bf16x8 af[14][MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const bf16* wrow = a.w + (size_t)(m * 16 + l15) * 27 * 16;
#pragma unroll
    for (int c = 0; c < 14; ++c) {
      union { uint4 u; bf16x8 v; } f;
      f.u = make_uint4(0u, 0u, 0u, 0u);
      if (2 * c + hi < 27) f.u = *reinterpret_cast<const uint4*>(wrow + (2 * c + hi) * 16 + half * 8);
      af[c][m] = f.v;
    }
  }

  // ---- per-thread load slots of one halo plane (the same for every plane); LDS image: [row][col parity][col/2]
  unsigned goff[NPT], loff[NPT];
  bool ok[NPT];
  const int piece = tid & 1;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int i = tid + k * 256;
    const int u = i >> 1;
    const int hh = u / HW, hw = u - hh * HW;
    const int ih = 2 * oh0 - 1 + hh, iw = 2 * ow0 - 1 + hw;
    ok[k] = i < HP * 2 && ih >= 0 && ih < a.IH && iw >= 0 && iw < a.IW;
    goff[k] = ok[k] ? (unsigned)((ih * a.IW + iw) * 16 + piece * 8) : 0u;
    // slots past the plane (i >= HP*2) land in the plane's padding
    loff[k] = i < HP * 2 ? (unsigned)((hh * HW + (hw & 1) * (TW + 1) + (hw >> 1)) * 32 + piece * 16) : (unsigned)(HP * 32 + (i - HP * 2) * 16);
  }
  static_assert(HP * 32 + (NPT * 256 - HP * 2) * 16 <= PLB, "padding slots must stay inside the plane");

  auto load_plane = [&](int id, uint4 (&rg)[NPT]) {
    const int idc = min(max(id, 0), a.ID - 1);
    const bf16* xp = a.x + ((size_t)n * a.ID + idc) * iplane * 16;
#pragma unroll
    for (int k = 0; k < NPT; ++k) rg[k] = *reinterpret_cast<const uint4*>(xp + goff[k]);
  };
  auto store_plane = [&](int id, int slot, uint4 (&rg)[NPT]) {
    const bool pin = id >= 0 && id < a.ID;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      uint4 v = rg[k];
      const bool keep = pin && ok[k];
      v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
      *reinterpret_cast<uint4*>(smem + slot * PLB + loff[k]) = v;
    }
  };

  // ---- fill the ring with input planes 2 d0 - 1, 2 d0, 2 d0 + 1
  {
    uint4 r0[NPT], r1[NPT], r2[NPT];
    load_plane(2 * d0 - 1, r0);
    load_plane(2 * d0, r1);
    load_plane(2 * d0 + 1, r2);
    store_plane(2 * d0 - 1, 0, r0);
    store_plane(2 * d0, 1, r1);
    store_plane(2 * d0 + 1, 2, r2);
  }
  __syncthreads();

  // lane part of the B-fragment address: output row 2*wave + j reads input rows 2*(2*wave + j) + kh
  const int lanebase = ((4 * wave) * HW + l15) * 32 + half * 16;
  float4 bv[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    bv[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bv[m] = *reinterpret_cast<const float4*>(a.bias + m * 16 + lq * 4);
  }
  unsigned ooff[NT];
  bool oin[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int oh = oh0 + 2 * wave + j, ow = ow0 + l15;
    oin[j] = FULL || (oh < a.OH && ow < a.OW);
    ooff[j] = oin[j] ? (unsigned)((oh * a.OW + ow) * 32 + lq * 4) : 0u;
  }
  float f0[MT][4], f1[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int k = 0; k < 4; ++k) { f0[m][k] = 0.f; f1[m][k] = 0.f; }
  const float* nosc = nullptr;

  const int np = d1 - d0;
  int s0 = 0;                                    // ring slot of input plane 2 od - 1
  for (int it = 0; it < np; ++it) {
    const int od = d0 + it;
    uint4 preA[NPT], preB[NPT];
    load_plane(2 * od + 2, preA);                // consumed by iteration it+1 (harmless on the last one)
    load_plane(2 * od + 3, preB);
    __builtin_amdgcn_sched_barrier(0);

    int sb[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) { int s = s0 + kd; s = s >= R ? s - R : s; sb[kd] = s * PLB; }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto tapoff = [](int t) { const int kh = (t / 3) % 3, kw = t % 3; return (kh * HW + (kw == 1 ? TW + 1 : (kw == 2 ? 1 : 0))) * 32; };
    auto read_b = [&](int c, bf16x8 (&b)[NT]) {
      const int t0 = 2 * c, t1 = 2 * c + 1 < 27 ? 2 * c + 1 : 26;   // phantom tap 27: real data x zero weights
      const int b0 = sb[t0 / 9] + tapoff(t0);
      const int b1 = sb[t1 / 9] + tapoff(t1);
      const char* p = smem + (hi ? b1 : b0) + lanebase;
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(p + j * (2 * HW * 32));
    };
    bf16x8 bA[NT], bB[NT];
    read_b(0, bA);
#define S2_INTERLEAVE                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < NT; ++i_) {       \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      \
  }
#pragma unroll
    for (int c = 0; c < 14; c += 2) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c][m], bA[j], acc[m][j], 0, 0, 0);
      read_b(c + 1, bB);
      S2_INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c + 1][m], bB[j], acc[m][j], 0, 0, 0);
      if (c + 2 < 14) read_b(c + 2, bA);
      S2_INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);
    }
#undef S2_INTERLEAVE

    // ---- epilogue of output plane od
    bf16* yp = a.y + ((size_t)n * a.OD + od) * oplane * 32;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        uint2 o;
        o.x = pack_bf2(acc[m][j][0] + bv[m].x, acc[m][j][1] + bv[m].y);
        o.y = pack_bf2(acc[m][j][2] + bv[m].z, acc[m][j][3] + bv[m].w);
        if (FULL || oin[j]) *reinterpret_cast<uint2*>(yp + ooff[j] + m * 16) = o;
        if (STATS && (FULL || oin[j])) stats_accum<STATS>(o, o, nosc, nosc, f0[m], f1[m]);
      }
    __builtin_amdgcn_sched_barrier(0);

    // planes 2od-1 and 2od are done with: their slots take 2od+2 and 2od+3
    int s1 = s0 + 1; s1 = s1 >= R ? s1 - R : s1;
    __syncthreads();                             // every wave has read this step's fragments
    store_plane(2 * od + 2, s0, preA);
    store_plane(2 * od + 3, s1, preB);
    __syncthreads();
    s0 = s0 + 2 >= R ? s0 + 2 - R : s0 + 2;
  }
  if (STATS) {
    const int chunk = tile % (a.ntw * a.nth * a.ndc);
    stats_finish<MT>(f0, f1, reinterpret_cast<double*>(smem), a.st.partial + ((size_t)n * a.st.chunks + chunk) * 64);
  }
}

// ---- 32 -> 64 channels (EnDown2).  K = one tap x 32 ci per MFMA; the four waves split the 64 couts (16 each, 27 A
// fragments = 108 VGPRs) and every wave reads all four N tiles of the (4 x 16) output tile: one ds_read_b128 per MFMA.
// Voxels are 64 bytes in LDS, their four 16-byte pieces XOR-swizzled by the voxel index (conflict-free b128 reads).
template <bool FULL, int STATS>
__global__ void __launch_bounds__(256)       // ~300 VGPRs: one block per CU, which is all a 2 x 32^3 output gives anyway
k_conv3_s2_c32_march(S2MarchArgs a) {
  constexpr int TH = 4, TW = 16;
  constexpr int HH = 2 * TH + 1, HW = 2 * TW + 1, HP = HH * HW;     // 9 x 33 = 297 voxels
  constexpr int NPT = (HP * 4 + 255) / 256;                        // 5 sixteen-byte pieces per thread per plane
  constexpr int PLB = NPT * 256 * 16;                              // 20 KB per LDS plane
  constexpr int R = 3;
  constexpr int NT = TH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  const int dci = (tile / (a.ntw * a.nth)) % a.ndc;
  const int n = tile / (a.ntw * a.nth * a.ndc);
  const int oh0 = thi * TH, ow0 = twi * TW;
  const int d0 = dci * a.DL;
  const int d1 = min(d0 + a.DL, a.OD);
  const size_t iplane = (size_t)a.IH * a.IW, oplane = (size_t)a.OH * a.OW;

  bf16x8 af[27];
  {
    const bf16* wrow = a.w + (size_t)(wave * 16 + l15) * 27 * 32 + lq * 8;
#pragma unroll
    for (int t = 0; t < 27; ++t) af[t] = *reinterpret_cast<const bf16x8*>(wrow + t * 32);
  }

  unsigned goff[NPT], loff[NPT];
  bool ok[NPT];
  const int piece = tid & 3;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int i = tid + k * 256;
    const int u = i >> 2;
    const int hh = u / HW, hw = u - hh * HW;
    const int ih = 2 * oh0 - 1 + hh, iw = 2 * ow0 - 1 + hw;
    ok[k] = i < HP * 4 && ih >= 0 && ih < a.IH && iw >= 0 && iw < a.IW;
    goff[k] = ok[k] ? (unsigned)((ih * a.IW + iw) * 32 + piece * 8) : 0u;
    const int ul = hh * HW + (hw & 1) * (TW + 1) + (hw >> 1);      // [row][col parity][col/2]
    loff[k] = i < HP * 4 ? (unsigned)(ul * 64 + ((piece ^ ((ul >> 1) & 3)) << 4)) : (unsigned)(i * 16);
  }

  auto load_plane = [&](int id, uint4 (&rg)[NPT]) {
    const int idc = min(max(id, 0), a.ID - 1);
    const bf16* xp = a.x + ((size_t)n * a.ID + idc) * iplane * 32;
#pragma unroll
    for (int k = 0; k < NPT; ++k) rg[k] = *reinterpret_cast<const uint4*>(xp + goff[k]);
  };
  auto store_plane = [&](int id, int slot, uint4 (&rg)[NPT]) {
    const bool pin = id >= 0 && id < a.ID;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      uint4 v = rg[k];
      const bool keep = pin && ok[k];
      v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
      *reinterpret_cast<uint4*>(smem + slot * PLB + loff[k]) = v;
    }
  };
  {
    uint4 r0[NPT], r1[NPT], r2[NPT];
    load_plane(2 * d0 - 1, r0);
    load_plane(2 * d0, r1);
    load_plane(2 * d0 + 1, r2);
    store_plane(2 * d0 - 1, 0, r0);
    store_plane(2 * d0, 1, r1);
    store_plane(2 * d0 + 1, 2, r2);
  }
  __syncthreads();

  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + wave * 16 + lq * 4);
  unsigned ooff[NT];
  bool oin[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int oh = oh0 + j, ow = ow0 + l15;
    oin[j] = FULL || (oh < a.OH && ow < a.OW);
    ooff[j] = oin[j] ? (unsigned)((oh * a.OW + ow) * 64 + wave * 16 + lq * 4) : 0u;
  }
  float f0[4] = {0.f, 0.f, 0.f, 0.f}, f1[4] = {0.f, 0.f, 0.f, 0.f};
  const float* nosc = nullptr;

  const int np = d1 - d0;
  int s0 = 0;
  for (int it = 0; it < np; ++it) {
    const int od = d0 + it;
    uint4 preA[NPT], preB[NPT];
    load_plane(2 * od + 2, preA);
    load_plane(2 * od + 3, preB);
    __builtin_amdgcn_sched_barrier(0);

    int sb[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) { int s = s0 + kd; s = s >= R ? s - R : s; sb[kd] = s * PLB; }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // voxel of (output row j, output col l15) at tap (kh, kw): row 2j + kh, parity-split column
    auto read_b = [&](int t, bf16x8 (&b)[NT]) {
      const int kh = (t / 3) % 3, kw = t % 3;
      const char* p = smem + sb[t / 9];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int ul = (2 * j + kh) * HW + (kw == 1 ? TW + 1 : (kw == 2 ? 1 : 0)) + l15;
        b[j] = *reinterpret_cast<const bf16x8*>(p + ul * 64 + ((lq ^ ((ul >> 1) & 3)) << 4));
      }
    };
    bf16x8 bA[NT], bB[NT];
    read_b(0, bA);
#define S2_INTERLEAVE                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < NT; ++i_) {       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      \
  }
#pragma unroll
    for (int t = 0; t < 27; t += 2) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bA[j], acc[j], 0, 0, 0);
      if (t + 1 < 27) {
        read_b(t + 1, bB);
        S2_INTERLEAVE
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t + 1], bB[j], acc[j], 0, 0, 0);
        read_b(t + 2, bA);
        S2_INTERLEAVE
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef S2_INTERLEAVE

    bf16* yp = a.y + ((size_t)n * a.OD + od) * oplane * 64;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      uint2 o;
      o.x = pack_bf2(acc[j][0] + bv.x, acc[j][1] + bv.y);
      o.y = pack_bf2(acc[j][2] + bv.z, acc[j][3] + bv.w);
      if (FULL || oin[j]) *reinterpret_cast<uint2*>(yp + ooff[j]) = o;
      if (STATS && (FULL || oin[j])) stats_accum<STATS>(o, o, nosc, nosc, f0, f1);
    }
    __builtin_amdgcn_sched_barrier(0);

    int s1 = s0 + 1; s1 = s1 >= R ? s1 - R : s1;
    __syncthreads();
    store_plane(2 * od + 2, s0, preA);
    store_plane(2 * od + 3, s1, preB);
    __syncthreads();
    s0 = s0 + 2 >= R ? s0 + 2 - R : s0 + 2;
  }
  if (STATS) {
    // each wave owns its 16 channels: the 16 lanes that share them add up in fp64 (fixed xor tree), no LDS pass
    const int chunk = tile % (a.ntw * a.nth * a.ndc);
    double* pout = a.st.partial + ((size_t)n * a.st.chunks + chunk) * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double sa = (double)f0[j], sbq = (double)f1[j];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { sa += __shfl_xor(sa, o, 64); sbq += __shfl_xor(sbq, o, 64); }
      if (l15 == 0) {
        pout[(wave * 16 + lq * 4 + j) * 2 + 0] = sa;
        pout[(wave * 16 + lq * 4 + j) * 2 + 1] = sbq;
      }
    }
  }
}

// 3x3x3, stride 2, pad 1, 16 -> 32 or 32 -> 64 channels, bf16, no prologue, no residual; statistics mode 0 or 1.
// plan_chunks != null: only report the statistics chunks per sample this launch would leave.
int conv3_s2_c16_march(const devis_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                       const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  const bool c16 = d->C1 == 16 && d->Cout == 32, c32 = d->C1 == 32 && d->Cout == 64;
  if (d->dtype != DEVIS_BF16 || !(c16 || c32) || d->C2 != 0 || d->SD != 2) return DEVIS_ERR_UNSUPPORTED;
  if (d->IH * d->IW * d->C1 >= (int64_t)1 << 31 || d->OH * d->OW * d->Cout >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  if (st && st->mode == 2) return DEVIS_ERR_UNSUPPORTED;
  const int TH = c16 ? 8 : 4;
  S2MarchArgs a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.y = (bf16*)y;
  a.N = (int)d->N; a.ID = (int)d->ID; a.IH = (int)d->IH; a.IW = (int)d->IW;
  a.OD = (int)d->OD; a.OH = (int)d->OH; a.OW = (int)d->OW;
  a.nth = (int)cdiv(a.OH, TH); a.ntw = (int)cdiv(a.OW, 16);
  // D chunk: about two blocks per CU; each chunk re-reads one input halo plane
  int dl = 32;
  const int64_t cols = (int64_t)a.N * a.nth * a.ntw;
  while (dl > 4 && cols * cdiv(a.OD, dl) < 512) dl >>= 1;
  a.DL = dl;
  a.ndc = (int)cdiv(a.OD, dl);
  const int64_t tiles = cols * a.ndc;
  if (tiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  if (plan_chunks) { *plan_chunks = (int64_t)a.ndc * a.nth * a.ntw; return DEVIS_OK; }
  a.st = StatsArgs{};
  const int mode = st ? st->mode : 0;
  if (mode) {
    a.st = *st;
    if (a.st.chunks != a.ndc * a.nth * a.ntw || a.st.C != d->Cout) return DEVIS_ERR_BAD_ARG;
  }
  const dim3 grid((unsigned)tiles);
  const bool full = a.OH % TH == 0 && a.OW % 16 == 0;
  if (c16) {
    constexpr int LDS = 3 * (((17 * 33 * 2 + 255) / 256) * 256 * 16);
    if (mode) {
      if (full) hipLaunchKernelGGL((k_conv3_s2_c16_march<true, 1>), grid, dim3(256), LDS, s, a);
      else hipLaunchKernelGGL((k_conv3_s2_c16_march<false, 1>), grid, dim3(256), LDS, s, a);
    } else {
      if (full) hipLaunchKernelGGL((k_conv3_s2_c16_march<true, 0>), grid, dim3(256), LDS, s, a);
      else hipLaunchKernelGGL((k_conv3_s2_c16_march<false, 0>), grid, dim3(256), LDS, s, a);
    }
  } else {
    constexpr int LDS = 3 * (((9 * 33 * 4 + 255) / 256) * 256 * 16);
    if (mode) {
      if (full) hipLaunchKernelGGL((k_conv3_s2_c32_march<true, 1>), grid, dim3(256), LDS, s, a);
      else hipLaunchKernelGGL((k_conv3_s2_c32_march<false, 1>), grid, dim3(256), LDS, s, a);
    } else {
      if (full) hipLaunchKernelGGL((k_conv3_s2_c32_march<true, 0>), grid, dim3(256), LDS, s, a);
      else hipLaunchKernelGGL((k_conv3_s2_c32_march<false, 0>), grid, dim3(256), LDS, s, a);
    }
  }
  return launch_status();
}

}  // namespace devis
