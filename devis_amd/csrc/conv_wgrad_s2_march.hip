// Weight gradient of the stride-2 3x3x3 conv 16 -> 32 (EnDown1, UNet3DZoo.py:59-67) as a plane march (bf16 channels-last,
// gfx950):  dW[co][tap][ci] = sum_v dy[v][co] * x[2v + tap - 1][ci].  The layer reads the full-resolution 16-channel
// tensor once (134 MB at 2 x 128^3) plus dy (33.5 MB): HBM-bound, 30 us at the wall; the tile kernel
// (conv_wgrad_mfma.hip) took 102 us.  Built from conv_s2_march.hip (ring of three input planes per output plane, LDS
// planes with even and odd columns apart so that a tap's stride-2 voxels are consecutive) and the K-split weight
// gradient of conv_wgrad_march.hip: a block owns an (8 x 16) column of dy voxels and walks it along D; a K step is
// 2 rows x 16 columns = 32 voxels; waves 0-1 take cout tile 0, waves 2-3 cout tile 1, and the two waves of a pair
// split the plane's four K steps; every wave keeps all 27 tap accumulators (16 co x 16 ci each), so one dy fragment
// feeds 27 MFMAs.  Operands come out of LDS with the transposing read (K = voxels).  The pair's accumulators are
// added in wave order through LDS at the end of the walk; per-block partials go to the workspace and
// k_reduce_partials adds them in a fixed order (deterministic).
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {

int launch_wgrad_reduce(const float* part, const float* bpart, int gx, int Cout, int Cin, int cin_real, float* dw,
                        float* dbias, hipStream_t s);

struct WgS2Args {
  const bf16* x;     // [N][ID][IH][IW][16]
  const bf16* dy;    // [N][OD][OH][OW][32]
  float* part;       // [blocks][27][32][16]
  float* bpart;      // [blocks][32] or null
  int N, ID, IH, IW, OD, OH, OW;
  int nth, ntw, ndc, DL;
};

__device__ __forceinline__ bf16x8 tr_pair_s2(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = a;
  u.h[1] = b;
  return u.v;
}

__global__ void __launch_bounds__(256)
k_wgrad3_s2_c16(WgS2Args a) {
  constexpr int TH = 8, TW = 16;
  constexpr int HH = 2 * TH + 1, HW = 2 * TW + 1, HP = HH * HW;     // 17 x 33 input voxels per plane
  constexpr int NPT = (HP * 2 + 255) / 256;                        // 5
  constexpr int PLB = NPT * 256 * 16;                              // 20 KB per x plane
  constexpr int R = 3;
  constexpr int YSUB = TH * TW * 32;                               // one cout tile's dy image of a plane: 4 KB
  constexpr int YPL = 2 * YSUB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ys = smem + R * PLB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int mt = wave >> 1, kh2 = wave & 1;          // cout tile, half of the plane's K steps

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
  const int d1 = min(d0 + a.DL, a.OD);
  const size_t iplane = (size_t)a.IH * a.IW, oplane = (size_t)a.OH * a.OW;

  // ---- load slots: x halo plane (5 pieces per thread, parity-split LDS image) and the dy plane tile (2 pieces)
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
    loff[k] = i < HP * 2 ? (unsigned)((hh * HW + (hw & 1) * (TW + 1) + (hw >> 1)) * 32 + piece * 16) : (unsigned)(i * 16);
  }
  unsigned yoff[2], yl[2];
  bool yok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + k * 256;                      // piece pc (8 couts) of voxel v of the 8 x 16 tile
    const int v = i >> 2, pc = i & 3;
    const int oh = oh0 + v / TW, ow = ow0 + v % TW;
    yok[k] = oh < a.OH && ow < a.OW;
    yoff[k] = yok[k] ? (unsigned)((oh * a.OW + ow) * 32 + pc * 8) : 0u;
    yl[k] = (unsigned)((pc >> 1) * YSUB + v * 32 + (pc & 1) * 16);     // cout tile pc >> 1, 32-byte voxel rows
  }
  float bsum[2][8];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[k][j] = 0.f;

  auto load_x = [&](int id, uint4 (&rg)[NPT]) {
    const int idc = min(max(id, 0), a.ID - 1);
    const bf16* xp = a.x + ((size_t)n * a.ID + idc) * iplane * 16;
#pragma unroll
    for (int k = 0; k < NPT; ++k) rg[k] = *reinterpret_cast<const uint4*>(xp + goff[k]);
  };
  auto store_x = [&](int id, int slot, uint4 (&rg)[NPT]) {
    const bool pin = id >= 0 && id < a.ID;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      uint4 v = rg[k];
      const bool keep = pin && ok[k];
      v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
      *reinterpret_cast<uint4*>(smem + slot * PLB + loff[k]) = v;
    }
  };
  auto load_y = [&](int od, uint4 (&rg)[2]) {
    const int odc = min(od, a.OD - 1);
    const bf16* yp = a.dy + ((size_t)n * a.OD + odc) * oplane * 32;
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
      bsum[k][0] += bf_lo(v.x); bsum[k][1] += bf_hi(v.x); bsum[k][2] += bf_lo(v.y); bsum[k][3] += bf_hi(v.y);
      bsum[k][4] += bf_lo(v.z); bsum[k][5] += bf_hi(v.z); bsum[k][6] += bf_lo(v.w); bsum[k][7] += bf_hi(v.w);
      *reinterpret_cast<uint4*>(ys + buf * YPL + yl[k]) = v;
    }
  };

  {
    uint4 r0[NPT], r1[NPT], r2[NPT], y0[2];
    load_x(2 * d0 - 1, r0);
    load_x(2 * d0, r1);
    load_x(2 * d0 + 1, r2);
    load_y(d0, y0);
    store_x(2 * d0 - 1, 0, r0);
    store_x(2 * d0, 1, r1);
    store_x(2 * d0 + 1, 2, r2);
    store_y(d0, 0, y0);
  }
  __syncthreads();

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wv = 4 * g + q;                          // voxel column this lane addresses
  const int lanep = wv * 32 + p * 8;
  // this wave's two K steps of a plane: s = 2 * kh2 + j -> dy rows 2s, 2s+1 -> x rows 4s + kh, 4s + 2 + kh
  int yo[2], xo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = 2 * kh2 + j;
    yo[j] = mt * YSUB + (2 * s) * TW * 32 + lanep;
    xo[j] = (4 * s) * HW * 32 + lanep;
  }

  const int np = d1 - d0;
  int s0 = 0;                                        // ring slot of input plane 2 od - 1
  for (int it = 0; it < np; ++it) {
    const int od = d0 + it;
    uint4 preA[NPT], preB[NPT], prey[2];
    load_x(2 * od + 2, preA);
    load_x(2 * od + 3, preB);
    load_y(od + 1, prey);
    __builtin_amdgcn_sched_barrier(0);

    int sb[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) { int s = s0 + kd; s = s >= R ? s - R : s; sb[kd] = s * PLB; }
    const char* yb = ys + (it & 1) * YPL;
    // group gi = (step j, kd): the 9 taps (kh, kw) of x plane 2 od - 1 + kd
    auto read_group = [&](int gi, bf16x8 (&bf)[9]) {
      const int j = gi / 3, kd = gi % 3;
      const char* xb = smem + sb[kd] + xo[j];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int kh = t / 3, kw = t % 3;
        const int o = (kh * HW + (kw == 1 ? TW + 1 : (kw == 2 ? 1 : 0))) * 32;
        bf[t] = tr_pair_s2(xb + o, xb + o + 2 * HW * 32);
      }
    };
    bf16x8 af[2], bA[9], bB[9];
#pragma unroll
    for (int j = 0; j < 2; ++j) af[j] = tr_pair_s2(yb + yo[j], yb + yo[j] + TW * 32);
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

    int s1 = s0 + 1; s1 = s1 >= R ? s1 - R : s1;
    store_y(od + 1, (it + 1) & 1, prey);             // the other dy buffer: nobody reads it in this iteration
    __syncthreads();                                 // every wave has read this step's x planes
    store_x(2 * od + 2, s0, preA);
    store_x(2 * od + 3, s1, preB);
    __syncthreads();
    s0 = s0 + 2 >= R ? s0 + 2 - R : s0 + 2;
  }

  // ---- the pair's accumulators in wave order (even wave + odd wave) through LDS, then the block's partial
  float* red = reinterpret_cast<float*>(smem);        // [2 cout tiles][27][256] floats = 54 KB (the x ring is free now)
  if (kh2 == 1) {
    float* r = red + mt * 27 * 256;
#pragma unroll
    for (int t = 0; t < 27; ++t) *reinterpret_cast<f32x4*>(r + (t * 64 + lane) * 4) = acc[t];
  }
  __syncthreads();
  if (kh2 == 0) {
    const float* r = red + mt * 27 * 256;
    float* part = a.part + (size_t)blockIdx.x * 27 * 512;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(r + (t * 64 + lane) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) part[(t * 32 + mt * 16 + g * 4 + j) * 16 + (lane & 15)] = acc[t][j] + v[j];
    }
  }
  if (a.bpart) {
    __syncthreads();
    float* bred = reinterpret_cast<float*>(smem);     // [2][256][8]
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) bred[(k * 256 + tid) * 8 + j] = bsum[k][j];
    __syncthreads();
    if (tid < 32) {
      // channel c = pc * 8 + j is held by the slots i = tid' + 256 k with i % 4 == pc, i.e. threads tid' % 4 == pc
      const int pc = tid >> 3, j = tid & 7;
      float s = 0.f;
      for (int k = 0; k < 2; ++k)
        for (int t = pc; t < 256; t += 4) s += bred[(k * 256 + t) * 8 + j];
      a.bpart[(size_t)blockIdx.x * 32 + tid] = s;
    }
  }
}

// 3x3x3, stride 2, pad 1, Cin = 16, Cout = 32, bf16, no prologue.  query: only report the workspace size.
int wgrad3_s2_c16_march(const devis_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                        size_t ws_bytes, hipStream_t s, bool query, size_t* need) {
  if (d->dtype != DEVIS_BF16 || d->C1 != 16 || d->C2 != 0 || d->Cout != 32 || d->SD != 2) return DEVIS_ERR_UNSUPPORTED;
  if (d->IH * d->IW * 16 >= (int64_t)1 << 31 || d->OH * d->OW * 32 >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  WgS2Args a;
  a.N = (int)d->N; a.ID = (int)d->ID; a.IH = (int)d->IH; a.IW = (int)d->IW;
  a.OD = (int)d->OD; a.OH = (int)d->OH; a.OW = (int)d->OW;
  a.nth = (int)cdiv(a.OH, 8); a.ntw = (int)cdiv(a.OW, 16);
  int dl = 32;
  const int64_t cols = (int64_t)a.N * a.nth * a.ntw;
  while (dl > 4 && cols * cdiv(a.OD, dl) < 256) dl >>= 1;
  a.DL = dl;
  a.ndc = (int)cdiv(a.OD, dl);
  const int64_t blocks = cols * a.ndc;
  if (blocks > 65535) return DEVIS_ERR_UNSUPPORTED;
  const size_t part_bytes = (size_t)blocks * 27 * 512 * sizeof(float);
  const size_t total = part_bytes + (size_t)blocks * 32 * sizeof(float);
  if (query) { *need = total; return DEVIS_OK; }
  if (!ws || ws_bytes < total) return DEVIS_ERR_WORKSPACE;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy;
  a.part = (float*)ws; a.bpart = dbias ? (float*)((char*)ws + part_bytes) : nullptr;
  constexpr int LDS = 3 * (((17 * 33 * 2 + 255) / 256) * 256 * 16) + 2 * 2 * (8 * 16 * 32);
  hipLaunchKernelGGL(k_wgrad3_s2_c16, dim3((unsigned)blocks), dim3(256), LDS, s, a);
  int rc = launch_status();
  if (rc) return rc;
  return launch_wgrad_reduce(a.part, a.bpart, (int)blocks, 32, 16, 16, dw, dbias, s);
}

}  // namespace devis
