// Plane-marching MFMA 3x3x3 convolution for the 16-channel, full-resolution layers (bf16,
// channels-last, stride 1, gfx950).  These layers carry 216 FLOP per byte of activation traffic,
// below the MI355X ridge, so the kernel is organised around HBM, not around the matrix cores:
//
//   * a block owns an (TH x TW) column of the volume and walks it along D.  A ring of R = 4 input
//     planes (halo included) lives in LDS; while the MFMAs of output plane d run, plane d+2 is in
//     flight from HBM in registers, and is written to the ring slot plane d-2 has just vacated.
//     Every input voxel is fetched once per block (halo only in H and W: 1.33x for 8x32).
//   * all 27*16*16 weights are A fragments held in 56 VGPRs for the whole walk.
//   * GEMM orientation as conv_mfma.hip: M = cout, N = 16 voxels along W, K = (2 taps x 16 ci); a
//     lane's B fragment is one ds_read_b128 (32-byte voxel pitch: conflict-free), its result is 4
//     consecutive cout of one voxel -> 8-byte channels-last stores.
//   * the preceding GroupNorm/BatchNorm affine + ReLU is applied on the way into LDS, bias and
//     the residual add on the way out.
//   * the walk loop has no divergent or data-dependent branch around a memory instruction (loads
//     go to clamped addresses and are zeroed by select), so the compiler can count its waits:
//     the prefetch stays in flight across the MFMAs and the output stores across the barrier.
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {

struct MarchArgs {
  const bf16* x;
  const bf16* w;      // [16][27][cin_real]
  const float* bias;  // [16] or null
  const float* sc;    // [N][16] or null
  const float* sh;
  const bf16* res;    // like y, or null
  bf16* y;
  int N, D, H, W;
  int nth, ntw, ndc, DL;
  StatsArgs st;       // norm statistics gathered in the epilogue (mode 0: off)
  // NB (norm-backward prologue): the conv input is dh = cA * g * [sc * xn + sh > 0] + cB * xn + cC, formed from
  // x (= g) and xn while the tile is staged (devis_norm_bwd_apply fused); sc / sh above are the gate's coefficients
  const bf16* xn;
  const float* cA;    // [N][16]
  const float* cB;
  const float* cC;
  bf16* dh;           // dh is also written here, once (the layer's weight gradient reads it), or null
};

// FULL: H % TH == 0 and W % TW == 0 (no partial tiles -> unconditional stores)
template <int TH, int TW, bool PRO, bool RES, bool CIN4, bool FULL, int STATS, bool NB = false>
__global__ void __launch_bounds__(256)
k_conv3_c16_march(MarchArgs a) {
  static_assert(!(NB && (PRO || CIN4)), "the norm-backward prologue replaces the affine one; 16 channels only");
  constexpr int HH = TH + 2, HW = TW + 2, HP = HH * HW;
  constexpr int NPT = (HP * 2 + 255) / 256;    // 16-byte pieces per thread per plane
  constexpr int PLB = NPT * 256 * 16;          // bytes per LDS plane (tail padding: no store guard)
  constexpr int R = 4;
  constexpr int NTW = TW / 16;
  constexpr int NT = TH * NTW / 4;             // N tiles per wave per plane
  constexpr int CINR = CIN4 ? 4 : 16;
  static_assert((TH * NTW) % 4 == 0 && NT % NTW == 0, "a wave's N tiles must cover whole rows");
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
  const int d1 = min(d0 + a.DL, a.D);
  const size_t plane = (size_t)a.H * a.W;

  // ---- A fragments: chunk c = taps (2c, 2c+1) x 16 ci; lanes 0-31 tap 2c, lanes 32-63 tap 2c+1
  bf16x8 af[14];
  {
    const bf16* wrow = a.w + (size_t)l15 * 27 * CINR;
#pragma unroll
    for (int c = 0; c < 14; ++c) {
      union { uint4 u; bf16x8 v; } f;
      f.u = make_uint4(0u, 0u, 0u, 0u);
      if (2 * c + hi < 27) {
        if (CIN4) {
          if (half == 0) {
            const uint2 t = *reinterpret_cast<const uint2*>(wrow + (2 * c + hi) * 4);
            f.u.x = t.x; f.u.y = t.y;
          }
        } else {
          f.u = *reinterpret_cast<const uint4*>(wrow + (2 * c + hi) * 16 + half * 8);
        }
      }
      af[c] = f.v;
    }
  }

  // ---- per-thread load slots of one halo plane (the same for every plane)
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
  float scv[8], shv[8];
  if (PRO || NB) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      scv[j] = a.sc[n * 16 + piece * 8 + j];
      shv[j] = a.sh[n * 16 + piece * 8 + j];
    }
  }
  float cav[8], cbv[8], ccv[8];
  bool ctr[NPT];                                // NB: this piece lies in the tile itself (not in its halo ring)
  if (NB) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cav[j] = a.cA[n * 16 + piece * 8 + j];
      cbv[j] = a.cB[n * 16 + piece * 8 + j];
      ccv[j] = a.cC[n * 16 + piece * 8 + j];
    }
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int u = (tid + k * 256) >> 1;
      const int hh = u / HW, hw = u - hh * HW;
      ctr[k] = ok[k] && hh >= 1 && hh <= TH && hw >= 1 && hw <= TW;
    }
  }

  // plane id is clamped for the address (always a real plane) and zeroed by store_plane when outside
  auto load_plane = [&](int id, uint4 (&rg)[NPT]) {
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
  auto store_plane = [&](int id, int slot, uint4 (&rg)[NPT]) {
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
  // NB: the second tensor of a plane, and the staging that forms dh (k_norm_bwd_apply's arithmetic, same rounding)
  auto load_plane_n = [&](int id, uint4 (&rx)[NPT]) {
    const int idc = min(max(id, 0), a.D - 1);
    const bf16* xp = a.xn + ((size_t)n * a.D + idc) * plane * 16;
#pragma unroll
    for (int k = 0; k < NPT; ++k) rx[k] = *reinterpret_cast<const uint4*>(xp + goff[k]);
  };
  auto store_plane_nb = [&](int id, int slot, uint4 (&rg)[NPT], uint4 (&rx)[NPT]) {
    const bool pin = id >= 0 && id < a.D;
    const bool own = a.dh != nullptr && id >= d0 && id < d1;      // this block's planes: it writes their dh
    const int idc = min(max(id, 0), a.D - 1);
    bf16* hp = a.dh + ((size_t)n * a.D + idc) * plane * 16;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const unsigned gw[4] = {rg[k].x, rg[k].y, rg[k].z, rg[k].w}, xw[4] = {rx[k].x, rx[k].y, rx[k].z, rx[k].w};
      unsigned ow[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = bf_lo(xw[j]), x1 = bf_hi(xw[j]);
        const float g0 = fmaf(x0, scv[2 * j], shv[2 * j]) > 0.f ? bf_lo(gw[j]) : 0.f;
        const float g1 = fmaf(x1, scv[2 * j + 1], shv[2 * j + 1]) > 0.f ? bf_hi(gw[j]) : 0.f;
        const float t0 = fmaf(cav[2 * j], g0, fmaf(cbv[2 * j], x0, ccv[2 * j]));
        const float t1 = fmaf(cav[2 * j + 1], g1, fmaf(cbv[2 * j + 1], x1, ccv[2 * j + 1]));
        ow[j] = pack_bf2(t0, t1);
      }
      uint4 v = make_uint4(ow[0], ow[1], ow[2], ow[3]);
      const bool keep = pin && ok[k];
      v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
      *reinterpret_cast<uint4*>(smem + slot * PLB + (tid + k * 256) * 16) = v;
      if (own && ctr[k]) *reinterpret_cast<uint4*>(hp + goff[k]) = v;
    }
  };

  // ---- fill the ring with planes d0-1, d0, d0+1
  {
    uint4 r0[NPT], r1[NPT], r2[NPT];
    load_plane(d0 - 1, r0);
    load_plane(d0, r1);
    load_plane(d0 + 1, r2);
    if (NB) {
      uint4 q0[NPT], q1[NPT], q2[NPT];
      load_plane_n(d0 - 1, q0);
      load_plane_n(d0, q1);
      load_plane_n(d0 + 1, q2);
      store_plane_nb(d0 - 1, 0, r0, q0);
      store_plane_nb(d0, 1, r1, q1);
      store_plane_nb(d0 + 1, 2, r2, q2);
    } else {
      store_plane(d0 - 1, 0, r0);
      store_plane(d0, 1, r1);
      store_plane(d0 + 1, 2, r2);
    }
  }
  __syncthreads();

  // lane part of the B-fragment address; N tile j of the wave adds a compile-time offset
  const int lanebase = (wave * (NT / NTW) * HW + l15) * 32 + half * 16;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + lq * 4);

  // per-lane output slots inside a plane (element offsets), clamped when the tile is partial
  unsigned ooff[NT];
  bool oin[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int oh = oh0 + wave * (NT / NTW) + j / NTW, ow = ow0 + (j % NTW) * 16 + l15;
    oin[j] = FULL || (oh < a.H && ow < a.W);
    ooff[j] = oin[j] ? (unsigned)((oh * a.W + ow) * 16 + lq * 4) : 0u;
  }

  // epilogue statistics: fp32 running sums of this lane's 4 channels (<= DL * NT addends), see mfma_common.h
  float f0[1][4] = {{0.f, 0.f, 0.f, 0.f}}, f1[1][4] = {{0.f, 0.f, 0.f, 0.f}};
  float scs[4] = {0.f, 0.f, 0.f, 0.f}, shs[4] = {0.f, 0.f, 0.f, 0.f};
  if (STATS == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { scs[j] = a.st.sc[n * 16 + lq * 4 + j]; shs[j] = a.st.sh[n * 16 + lq * 4 + j]; }
  }

  const int np = d1 - d0;
  for (int it = 0; it < np; ++it) {
    const int od = d0 + it;
    uint4 pre[NPT], pren[NPT];
    load_plane(od + 2, pre);                    // consumed by iteration it+1 (harmless on the last one)
    if (NB) load_plane_n(od + 2, pren);
    const size_t pbase = ((size_t)n * a.D + od) * plane * 16;
    uint2 rv[NT];
    if (RES) {
      const bf16* rp = a.res + pbase;
#pragma unroll
      for (int j = 0; j < NT; ++j) rv[j] = *reinterpret_cast<const uint2*>(rp + ooff[j]);
    }
    uint2 xv[NT];
    if (STATS == 2) {
      const bf16* xp = a.st.xref + pbase;
#pragma unroll
      for (int j = 0; j < NT; ++j) xv[j] = *reinterpret_cast<const uint2*>(xp + ooff[j]);
    }
    __builtin_amdgcn_sched_barrier(0);

    int sb[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) sb[kd] = ((it + kd) & (R - 1)) * PLB;

    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // B fragments are double-buffered: the reads of chunk c+1 are in flight under the MFMAs of
    // chunk c (the scheduling barriers keep the compiler from folding them back into one register)
    auto read_b = [&](int c, bf16x8 (&b)[NT]) {
      const int t0 = 2 * c, t1 = 2 * c + 1 < 27 ? 2 * c + 1 : 26;   // phantom tap 27: real data x zero weights
      const int b0 = sb[t0 / 9] + (((t0 / 3) % 3) * HW + t0 % 3) * 32;
      const int b1 = sb[t1 / 9] + (((t1 / 3) % 3) * HW + t1 % 3) * 32;
      const char* p = smem + (hi ? b1 : b0) + lanebase;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(p + ((j / NTW) * HW + (j % NTW) * 16) * 32);
    };
    bf16x8 bA[NT], bB[NT];
    read_b(0, bA);
    // chunk c's MFMAs are issued with chunk c+1's fragment reads slotted between them (1 MFMA, 1 read, ...)
#define MARCH_INTERLEAVE                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < NT; ++i_) {       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      \
  }
#pragma unroll
    for (int c = 0; c < 14; c += 2) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bA[j], acc[j], 0, 0, 0);
      read_b(c + 1, bB);
      MARCH_INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c + 1], bB[j], acc[j], 0, 0, 0);
      if (c + 2 < 14) read_b(c + 2, bA);
      MARCH_INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);
    }
#undef MARCH_INTERLEAVE

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

    if (NB) store_plane_nb(od + 2, (it + 3) & (R - 1), pre, pren);
    else store_plane(od + 2, (it + 3) & (R - 1), pre);
    __syncthreads();
  }
  if (STATS) {
    const int chunk = tile % (a.ntw * a.nth * a.ndc);
    stats_finish<1>(f0, f1, reinterpret_cast<double*>(smem), a.st.partial + ((size_t)n * a.st.chunks + chunk) * 32);
  }
}

template <int TH, int TW, bool PRO, bool RES, bool CIN4, int STATS>
static void launch_march_v(bool full, dim3 grid, hipStream_t s, const MarchArgs& a) {
  constexpr int LDS = 4 * (((TH + 2) * (TW + 2) * 2 + 255) / 256) * 256 * 16;
  if (full) hipLaunchKernelGGL((k_conv3_c16_march<TH, TW, PRO, RES, CIN4, true, STATS>), grid, dim3(256), LDS, s, a);
  else hipLaunchKernelGGL((k_conv3_c16_march<TH, TW, PRO, RES, CIN4, false, STATS>), grid, dim3(256), LDS, s, a);
}

template <int TH, int TW, int STATS>
static int launch_march_s(const devis_conv_desc* d, bool full, dim3 grid, hipStream_t s, const MarchArgs& a) {
  const bool sc = a.sc != nullptr, res = a.res != nullptr;
  if (d->C1 == 4) {
    if (sc || res || STATS == 2) return DEVIS_ERR_UNSUPPORTED;
    launch_march_v<TH, TW, false, false, true, STATS == 2 ? 0 : STATS>(full, grid, s, a);
  } else if (STATS == 2) {            // dgrad in front of a norm backward: no prologue, no residual
    if (sc || res) return DEVIS_ERR_UNSUPPORTED;
    launch_march_v<TH, TW, false, false, false, STATS>(full, grid, s, a);
  } else if (sc) {
    if (res) launch_march_v<TH, TW, true, true, false, STATS == 2 ? 0 : STATS>(full, grid, s, a);
    else launch_march_v<TH, TW, true, false, false, STATS == 2 ? 0 : STATS>(full, grid, s, a);
  } else {
    if (res) launch_march_v<TH, TW, false, true, false, STATS == 2 ? 0 : STATS>(full, grid, s, a);
    else launch_march_v<TH, TW, false, false, false, STATS == 2 ? 0 : STATS>(full, grid, s, a);
  }
  return launch_status();
}

// D chunk: enough blocks to fill 256 CUs about twice; each chunk re-reads 2 halo planes
static int march_dl(int N, int nth, int ntw, int D) {
  int dl = 32;
  const int64_t cols = (int64_t)N * nth * ntw;
  while (dl > 8 && cols * cdiv(D, dl) < 512) dl >>= 1;
  return dl;
}

// plan_chunks != null: only report the statistics chunks per sample this launch would leave
template <int TH, int TW>
static int launch_march(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
                        const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks,
                        hipStream_t s) {
  MarchArgs a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.sc = sc; a.sh = sh; a.res = (const bf16*)res; a.y = (bf16*)y;
  a.xn = nullptr; a.cA = a.cB = a.cC = nullptr; a.dh = nullptr;
  a.N = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW;
  a.nth = (int)cdiv(a.H, TH); a.ntw = (int)cdiv(a.W, TW);
  a.DL = march_dl(a.N, a.nth, a.ntw, a.D);
  a.ndc = (int)cdiv(a.D, a.DL);
  const int64_t tiles = (int64_t)a.N * a.ndc * a.nth * a.ntw;
  if (tiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  if (plan_chunks) {
    const bool can = !(st && st->mode == 2 && (sc || res || d->C1 == 4));
    *plan_chunks = can ? (int64_t)a.ndc * a.nth * a.ntw : 0;
    return DEVIS_OK;
  }
  const dim3 grid((unsigned)tiles);
  const bool full = a.H % TH == 0 && a.W % TW == 0;
  a.st = StatsArgs{};
  if (st && st->mode) {
    a.st = *st;
    if (a.st.chunks != a.ndc * a.nth * a.ntw || a.st.C != 16) return DEVIS_ERR_BAD_ARG;
    if (a.st.mode == 1) return launch_march_s<TH, TW, 1>(d, full, grid, s, a);
    return launch_march_s<TH, TW, 2>(d, full, grid, s, a);
  }
  return launch_march_s<TH, TW, 0>(d, full, grid, s, a);
}

// The same conv with the norm-backward prologue (the layer's input dh is formed from g and xn; see MarchArgs): the two
// variants the 3-D backward uses - statistics mode 2 without a residual, a residual without statistics (or neither).
int conv3_c16_march_nb(const devis_conv_desc* d, const void* g, const void* xn, const float* sc, const float* sh,
                       const float* cA, const float* cB, const float* cC, const void* w, const void* res, void* y, void* dh,
                       const StatsArgs* st, hipStream_t s) {
  if (d->dtype != DEVIS_BF16 || d->Cout != 16 || d->C1 != 16 || d->C2 != 0) return DEVIS_ERR_UNSUPPORTED;
  // this entry is reached straight from the C ABI (devis_conv_dgrad_normbwd), not through conv_fwd_mfma's checks: the
  // kernel walks 27 taps of a pad-1 stride-1 layer and indexes with 32-bit offsets, so anything else is refused here
  if (d->KD != 3 || d->KH != 3 || d->KW != 3 || d->PD != 1 || d->PH != 1 || d->PW != 1 || d->SD != 1 || d->SH != 1 || d->SW != 1)
    return DEVIS_ERR_UNSUPPORTED;
  if (d->OD != d->ID || d->OH != d->IH || d->OW != d->IW || d->OW <= 16) return DEVIS_ERR_UNSUPPORTED;
  if (d->IH * d->IW * 16 >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  if (d->N * d->ID * d->IH * d->IW * 16 >= (int64_t)1 << 32 || d->N * d->ID * d->IH * d->IW >= 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const int mode = st ? st->mode : 0;
  if (mode == 1 || (mode == 2 && res)) return DEVIS_ERR_UNSUPPORTED;
  constexpr int TH = 8, TW = 32;
  MarchArgs a;
  a.x = (const bf16*)g; a.w = (const bf16*)w; a.bias = nullptr; a.sc = sc; a.sh = sh; a.res = (const bf16*)res; a.y = (bf16*)y;
  a.xn = (const bf16*)xn; a.cA = cA; a.cB = cB; a.cC = cC; a.dh = (bf16*)dh;
  a.N = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW;
  a.nth = (int)cdiv(a.H, TH); a.ntw = (int)cdiv(a.W, TW);
  a.DL = march_dl(a.N, a.nth, a.ntw, a.D);     // the chunking of launch_march: devis_conv_stats_chunks counts on it
  a.ndc = (int)cdiv(a.D, a.DL);
  const int64_t tiles = (int64_t)a.N * a.ndc * a.nth * a.ntw;
  if (tiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  a.st = StatsArgs{};
  if (mode == 2) {
    a.st = *st;
    if (a.st.chunks != a.ndc * a.nth * a.ntw || a.st.C != 16) return DEVIS_ERR_BAD_ARG;
  }
  const dim3 grid((unsigned)tiles);
  const bool full = a.H % TH == 0 && a.W % TW == 0;
  constexpr int LDS = 4 * (((TH + 2) * (TW + 2) * 2 + 255) / 256) * 256 * 16;
#define NBL(RES_, FULL_, ST_) hipLaunchKernelGGL((k_conv3_c16_march<TH, TW, false, RES_, false, FULL_, ST_, true>), grid, dim3(256), LDS, s, a)
  if (mode == 2) { if (full) NBL(false, true, 2); else NBL(false, false, 2); }
  else if (res) { if (full) NBL(true, true, 0); else NBL(true, false, 0); }
  else { if (full) NBL(false, true, 0); else NBL(false, false, 0); }
#undef NBL
  return launch_status();
}

// stride-1 3x3x3, Cout = 16, Cin = 16 (or 4: the first layer), bf16; anything else is unsupported
int conv3_c16_march(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
                    const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  if (d->dtype != DEVIS_BF16 || d->Cout != 16 || (d->C1 != 16 && d->C1 != 4) || d->C2 != 0 || d->SD != 1) return DEVIS_ERR_UNSUPPORTED;
  if (d->OD != d->ID || d->OH != d->IH || d->OW != d->IW) return DEVIS_ERR_UNSUPPORTED;
  if (d->IH * d->IW * 16 >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  if (d->OW <= 16) return launch_march<16, 16>(d, x, w, bias, sc, sh, res, y, st, plan_chunks, s);
  return launch_march<8, 32>(d, x, w, bias, sc, sh, res, y, st, plan_chunks, s);
}

}  // namespace devis
