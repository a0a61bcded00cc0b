// MFMA "gather-GEMM" for the layers without a spatial halo (bf16 channels-last, gfx950):
//   1x1 conv forward (virtual concat of two inputs) and its dgrad (two outputs),
//   ConvTranspose k2s2 forward (scatter store) and dgrad (gather over the 2x2x2 taps),
//   dgrad of the stride-2 3x3x3 convs (one GEMM per output-parity class, 1/2/4/8 taps),
//   the k2-s2 'down' conv of the V-Net: forward = the ConvT-dgrad gather, dgrad = the ConvT-forward scatter.
// out[m][v] = sum_k A[m][k] * B[k][v]: M = output channels (x taps for ConvT fwd), N = 16 voxels along W,
// K = source channels (x taps).  B fragments are 16-byte global loads (every source byte is used once, so
// there is nothing to stage in LDS); A fragments (weights) come from L1/L2.  These layers are HBM-bound.
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {

enum { PW_CONV1 = 0, PW_CONVT_FWD = 1, PW_CONVT_DGRAD = 2, PW_S2_DGRAD = 3, PW_DOWN_DGRAD = 4 };

struct PwArgs {
  const bf16* x1; const bf16* x2; const bf16* w; const float* bias; const bf16* res;
  bf16* y1; bf16* y2;
  int C1, C2, K, M, M1;
  int N, D, H, W;        // voxel grid the N dimension runs over
  int Cs;                // channels per tap of the gathered tensor (ConvT dgrad / s2 dgrad: Cout; ConvT fwd: Cout)
  int KD, KH, KW;        // ConvT kernel (1 or 2 per dim)
  int OD, OH, OW;        // s2 dgrad: coarse (dy) grid;  ConvT: fine (output) grid
  int ntw;               // 16-voxel tiles per row
  int64_t ntiles;
  StatsArgs st;          // PW_CONV1 only: norm-backward moments gathered in the epilogue (mode 2), see mfma_common.h
  int blocks_per_n;      // with st: blocks per sample (host guarantees blocks do not straddle samples)
  int iters;             // with st: groups of 4*NT tiles a block walks
};

// 8 channels [c, c+8) of a row holding C channels (C % 4 == 0), zero-extended past the end
__device__ __forceinline__ uint4 pw_piece(const bf16* row, int c, int C) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (c + 8 <= C) v = *reinterpret_cast<const uint4*>(row + c);
  else if (c + 4 <= C) { const uint2 t = *reinterpret_cast<const uint2*>(row + c); v.x = t.x; v.y = t.y; }
  return v;
}

template <int MODE>
__device__ __forceinline__ uint4 pw_load_b(const PwArgs& a, int k0, int n, int d, int h, int w, bool vox_ok, int cls) {
  uint4 z = make_uint4(0u, 0u, 0u, 0u);
  if (!vox_ok || k0 >= a.K) return z;
  if (MODE == PW_CONV1 || MODE == PW_CONVT_FWD || MODE == PW_DOWN_DGRAD) {
    const size_t v = ((size_t)(n * a.D + d) * a.H + h) * a.W + w;
    if (k0 < a.C1) return pw_piece(a.x1 + v * a.C1, k0, a.C1);
    return pw_piece(a.x2 + v * a.C2, k0 - a.C1, a.C2);
  } else if (MODE == PW_CONVT_DGRAD) {
    const int p = k0 / a.Cs, c = k0 % a.Cs;
    const int pw = p % a.KW, ph = (p / a.KW) % a.KH, pd = p / (a.KW * a.KH);
    const size_t ov = ((size_t)(n * a.OD + d * a.KD + pd) * a.OH + h * a.KH + ph) * a.OW + w * a.KW + pw;
    return *reinterpret_cast<const uint4*>(a.x1 + ov * a.Cs + c);
  } else {   // PW_S2_DGRAD: (d,h,w) are class coordinates, fine voxel = 2*j + parity
    const int tj = k0 / a.Cs, c = k0 % a.Cs;
    const int bw = cls & 1, bh = (cls >> 1) & 1, bd = (cls >> 2) & 1;
    // per odd dim one bit of tj picks kd in {0,2} (0 -> kd=0 -> coarse j+1 ; 1 -> kd=2 -> coarse j); even dim: kd=1 -> j
    int bit = 0;
    const int sw = bw ? ((tj >> bit++) & 1) : -1;
    const int sh = bh ? ((tj >> bit++) & 1) : -1;
    const int sd = bd ? ((tj >> bit++) & 1) : -1;
    const int ow = (sw == 0) ? w + 1 : w, oh = (sh == 0) ? h + 1 : h, od = (sd == 0) ? d + 1 : d;
    if (ow >= a.OW || oh >= a.OH || od >= a.OD) return z;
    const size_t ov = ((size_t)(n * a.OD + od) * a.OH + oh) * a.OW + ow;
    return *reinterpret_cast<const uint4*>(a.x1 + ov * a.Cs + c);
  }
}

template <int MODE>
__device__ __forceinline__ uint4 pw_load_a(const PwArgs& a, int m, int k0, int cls) {
  uint4 z = make_uint4(0u, 0u, 0u, 0u);
  if (m >= a.M || k0 >= a.K) return z;
  if (MODE == PW_DOWN_DGRAD) {   // row m = (tap p, ci c) of the conv-dgrad packing [Cin][T reversed][Cout]
    const int T = a.KD * a.KH * a.KW, p = m / a.Cs, c = m % a.Cs;
    return pw_piece(a.w + ((size_t)c * T + (T - 1 - p)) * a.K, k0, a.K);
  }
  if (MODE != PW_S2_DGRAD) return pw_piece(a.w + (size_t)m * a.K, k0, a.K);
  const int tj = k0 / a.Cs, c = k0 % a.Cs;
  const int bw = cls & 1, bh = (cls >> 1) & 1, bd = (cls >> 2) & 1;
  int bit = 0;
  const int kw = bw ? 2 * ((tj >> bit++) & 1) : 1;
  const int kh = bh ? 2 * ((tj >> bit++) & 1) : 1;
  const int kd = bd ? 2 * ((tj >> bit++) & 1) : 1;
  const int t = (kd * 3 + kh) * 3 + kw;
  return *reinterpret_cast<const uint4*>(a.w + ((size_t)m * 27 + (26 - t)) * a.Cs + c);   // tap-flipped dgrad packing
}

template <int MODE, int MTP, int NT, int STATS = 0>
__global__ void __launch_bounds__(256)
k_pw_mfma(PwArgs a) {
  __shared__ double stat_red[STATS ? 4 * MTP * 32 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  // with statistics a block walks `iters` consecutive groups of 4*NT tiles (same sample), so that the fp64
  // reduction at the end is paid once per iters*4*NT*16 voxels; otherwise iters = 1
  const int iters = STATS ? a.iters : 1;
  float f0[MTP][4], f1[MTP][4], scs[MTP][4], shs[MTP][4];
  if (STATS) {
    const int n_blk = (int)(blockIdx.x / a.blocks_per_n);      // a block never straddles samples (host)
#pragma unroll
    for (int m = 0; m < MTP; ++m)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f0[m][k] = 0.f; f1[m][k] = 0.f;
        scs[m][k] = a.st.sc[n_blk * a.M + m * 16 + lq * 4 + k];
        shs[m][k] = a.st.sh[n_blk * a.M + m * 16 + lq * 4 + k];
      }
  }
  for (int it = 0; it < iters; ++it) {
  int nn[NT], dd[NT], hh[NT], ww[NT];
  bool ok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    // 32-bit index arithmetic (hosts check voxels < 2^31): 64-bit divisions cost ~100 VALU instructions each
    const unsigned tile = (((unsigned)blockIdx.x * (unsigned)iters + (unsigned)it) * 4u + (unsigned)wave) * NT + j;
    const bool tv = tile < (unsigned)a.ntiles;
    const unsigned tt = tv ? tile : 0u;
    const int twi = (int)(tt % (unsigned)a.ntw);
    unsigned r = tt / (unsigned)a.ntw;
    hh[j] = (int)(r % (unsigned)a.H); r /= (unsigned)a.H;
    dd[j] = (int)(r % (unsigned)a.D);
    nn[j] = (int)(r / (unsigned)a.D);
    ww[j] = twi * 16 + l15;
    ok[j] = tv && ww[j] < a.W;
  }

  // stride-2 dgrad: the block runs all 8 output-parity classes of its coarse tiles back to back, so the
  // interleaved 8-byte pieces they write (and the residual they read) meet in L2 while the line is resident;
  // one class per block (grid.y) left every 64-byte line half written until another block came by
  // (small layers keep one class per block, grid.y = 8: they need the blocks more than the locality)
  const int cls0 = (MODE == PW_S2_DGRAD && gridDim.y == 8) ? (int)blockIdx.y : 0;
  const int cls1 = MODE == PW_S2_DGRAD ? (gridDim.y == 8 ? cls0 + 1 : 8) : 1;
  for (int cls = cls0; cls < cls1; ++cls) {
  PwArgs b = a;
  if (MODE == PW_S2_DGRAD) b.K = a.Cs << (((cls >> 0) & 1) + ((cls >> 1) & 1) + ((cls >> 2) & 1));
  const int kchunks = (b.K + 31) >> 5;
  for (int m0 = 0; m0 < a.M; m0 += MTP * 16) {
    f32x4 acc[MTP][NT];
#pragma unroll
    for (int m = 0; m < MTP; ++m)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // statistics: the normalised tensor at this lane's outputs, fetched with the operands (a load issued
    // after the epilogue's stores would wait for them: vmcnt counts both)
    uint2 xref[MTP][NT];
    if (STATS == 2 && MODE == PW_CONV1) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const size_t v = ((size_t)(nn[j] * a.D + dd[j]) * a.H + hh[j]) * a.W + ww[j];
#pragma unroll
        for (int m = 0; m < MTP; ++m) {
          const int mrow = m0 + m * 16 + lq * 4;
          xref[m][j] = (ok[j] && mrow < a.M) ? *reinterpret_cast<const uint2*>(a.st.xref + v * a.M + mrow) : make_uint2(0u, 0u);
        }
      }
    }
    for (int kc = 0; kc < kchunks; ++kc) {
      const int k0 = kc * 32 + lq * 8;
      union { uint4 u; bf16x8 v; } af[MTP], bf[NT];
#pragma unroll
      for (int m = 0; m < MTP; ++m) af[m].u = pw_load_a<MODE>(b, m0 + m * 16 + l15, k0, cls);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j].u = pw_load_b<MODE>(b, k0, nn[j], dd[j], hh[j], ww[j], ok[j], cls);
#pragma unroll
      for (int m = 0; m < MTP; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m].v, bf[j].v, acc[m][j], 0, 0, 0);
    }
    // epilogue: lane owns rows mrow..mrow+3 of voxel l15
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (!ok[j]) continue;
#pragma unroll
      for (int m = 0; m < MTP; ++m) {
        const int mrow = m0 + m * 16 + lq * 4;
        if (mrow >= a.M) continue;
        float v0 = acc[m][j][0], v1 = acc[m][j][1], v2 = acc[m][j][2], v3 = acc[m][j][3];
        bf16* dst;
        int brow = mrow;
        if (MODE == PW_CONV1) {
          const size_t v = ((size_t)(nn[j] * a.D + dd[j]) * a.H + hh[j]) * a.W + ww[j];
          dst = mrow < a.M1 ? a.y1 + v * a.M1 + mrow : a.y2 + v * (a.M - a.M1) + (mrow - a.M1);
          if (a.res) {
            const uint2 r = *reinterpret_cast<const uint2*>(a.res + v * a.M1 + mrow);
            v0 += bf_lo(r.x); v1 += bf_hi(r.x); v2 += bf_lo(r.y); v3 += bf_hi(r.y);
          }
        } else if (MODE == PW_CONVT_FWD || MODE == PW_DOWN_DGRAD) {
          const int p = mrow / a.Cs, c = mrow % a.Cs;
          const int pw = p % a.KW, ph = (p / a.KW) % a.KH, pd = p / (a.KW * a.KH);
          const size_t ov = ((size_t)(nn[j] * a.OD + dd[j] * a.KD + pd) * a.OH + hh[j] * a.KH + ph) * a.OW + ww[j] * a.KW + pw;
          dst = a.y1 + ov * a.Cs + c;
          brow = c;
          if (MODE == PW_DOWN_DGRAD && a.res) {
            const uint2 r = *reinterpret_cast<const uint2*>(a.res + ov * a.Cs + c);
            v0 += bf_lo(r.x); v1 += bf_hi(r.x); v2 += bf_lo(r.y); v3 += bf_hi(r.y);
          }
        } else if (MODE == PW_CONVT_DGRAD) {
          const size_t v = ((size_t)(nn[j] * a.D + dd[j]) * a.H + hh[j]) * a.W + ww[j];
          dst = a.y1 + v * a.M + mrow;
        } else {
          const int fw = 2 * ww[j] + (cls & 1), fh = 2 * hh[j] + ((cls >> 1) & 1), fd = 2 * dd[j] + ((cls >> 2) & 1);
          const size_t v = ((size_t)(nn[j] * (2 * a.D) + fd) * (2 * a.H) + fh) * (2 * a.W) + fw;
          dst = a.y1 + v * a.M + mrow;
          if (a.res) {
            const uint2 r = *reinterpret_cast<const uint2*>(a.res + v * a.M + mrow);
            v0 += bf_lo(r.x); v1 += bf_hi(r.x); v2 += bf_lo(r.y); v3 += bf_hi(r.y);
          }
        }
        if (a.bias) {
          const float4 bb = *reinterpret_cast<const float4*>(a.bias + brow);
          v0 += bb.x; v1 += bb.y; v2 += bb.z; v3 += bb.w;
        }
        uint2 o;
        o.x = pack_bf2(v0, v1);
        o.y = pack_bf2(v2, v3);
        *reinterpret_cast<uint2*>(dst) = o;
        if (STATS == 2 && MODE == PW_CONV1) stats_accum<2>(o, xref[m][j], scs[m], shs[m], f0[m], f1[m]);
      }
    }
  }
  }
  }
  if (STATS) {       // host: M <= MTP*16, one m0 pass
    const int n_blk = (int)(blockIdx.x / a.blocks_per_n), chunk = (int)(blockIdx.x % a.blocks_per_n);
    stats_finish<MTP>(f0, f1, stat_red, a.st.partial + ((size_t)n_blk * a.st.chunks + chunk) * a.M * 2);
  }
}

// tiles per block of the launch pw_launch picks for M output rows
// (8 tiles per wave for the one-K-chunk layers was measured and is slower: 1x1 64 -> 89 us, stride-2 dgrad
// 154 -> 184 us at 128^3; kept as a switch)
static inline bool pw_wide(int M, int64_t ntiles) { (void)M; (void)ntiles; return false; }
static inline int pw_tiles_per_block(int M, int64_t ntiles) { return cdiv(M, 16) >= 4 ? 8 : (pw_wide(M, ntiles) ? 32 : 16); }

// PW_CONV1 with norm-backward statistics: plan_chunks != null only reports the chunks per sample (0: cannot)
static int pw_launch_stats(PwArgs& a, const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  a.ntw = (int)cdiv(a.W, 16);
  a.ntiles = (int64_t)a.N * a.D * a.H * a.ntw;
  const int64_t tiles_per_n = (int64_t)a.D * a.H * a.ntw;
  const int tpb = pw_tiles_per_block(a.M, a.ntiles);
  bool can = st->mode == 2 && a.M1 == a.M && a.M % 16 == 0 && a.M <= 64 && tiles_per_n % tpb == 0 && a.W % 16 == 0;
  // groups per block: as many as keep >= 1024 blocks in the launch and divide a sample's groups
  int iters = 2;
  while (iters > 1 && ((tiles_per_n / tpb) % iters != 0 || a.ntiles / tpb / iters < 1024)) iters >>= 1;
  const int64_t bpn = tiles_per_n / tpb / iters;
  can = can && bpn >= 1 && bpn <= 0x7fffffff;
  if (plan_chunks) { *plan_chunks = can ? bpn : 0; return DEVIS_OK; }
  if (!can) return DEVIS_ERR_UNSUPPORTED;
  a.st = *st;
  a.blocks_per_n = (int)bpn;
  a.iters = iters;
  if (a.st.chunks != a.blocks_per_n || a.st.C != a.M) return DEVIS_ERR_BAD_ARG;
  const dim3 grid((unsigned)(bpn * a.N));
  const int mt = (int)cdiv(a.M, 16);
  if (mt >= 4) hipLaunchKernelGGL((k_pw_mfma<PW_CONV1, 4, 2, 2>), grid, dim3(256), 0, s, a);
  else if (mt >= 2) hipLaunchKernelGGL((k_pw_mfma<PW_CONV1, 2, 4, 2>), grid, dim3(256), 0, s, a);
  else if (pw_wide(a.M, a.ntiles)) hipLaunchKernelGGL((k_pw_mfma<PW_CONV1, 1, 8, 2>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_pw_mfma<PW_CONV1, 1, 4, 2>), grid, dim3(256), 0, s, a);
  return launch_status();
}

template <int MODE>
static int pw_launch(PwArgs& a, int ngrid_y, hipStream_t s) {
  a.ntw = (int)cdiv(a.W, 16);
  a.ntiles = (int64_t)a.N * a.D * a.H * a.ntw;
  const int mt = (int)cdiv(a.M, 16);
  if (mt >= 4) {
    const dim3 grid((unsigned)cdiv(a.ntiles, 4 * 2), (unsigned)ngrid_y);
    hipLaunchKernelGGL((k_pw_mfma<MODE, 4, 2>), grid, dim3(256), 0, s, a);
  } else if (mt >= 2) {
    const dim3 grid((unsigned)cdiv(a.ntiles, 4 * 4), (unsigned)ngrid_y);
    hipLaunchKernelGGL((k_pw_mfma<MODE, 2, 4>), grid, dim3(256), 0, s, a);
  } else if (pw_wide(a.M, a.ntiles)) {
    const dim3 grid((unsigned)cdiv(a.ntiles, 4 * 8), (unsigned)ngrid_y);
    hipLaunchKernelGGL((k_pw_mfma<MODE, 1, 8>), grid, dim3(256), 0, s, a);
  } else {
    const dim3 grid((unsigned)cdiv(a.ntiles, 4 * 4), (unsigned)ngrid_y);
    hipLaunchKernelGGL((k_pw_mfma<MODE, 1, 4>), grid, dim3(256), 0, s, a);
  }
  return launch_status();
}

static bool small_enough(int64_t v) { return v > 0 && v < 0x7fffffff; }

// ---- 1x1 conv forward: y = W x (+bias, +residual), x = virtual cat(x1, x2)
int conv1_fwd_mfma(const devis_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                   const void* res, void* y, hipStream_t s) {
  if (d->dtype != DEVIS_BF16 || d->KD != 1 || d->KH != 1 || d->KW != 1 || d->SD != 1 || d->SH != 1 || d->SW != 1) return DEVIS_ERR_UNSUPPORTED;
  if (d->PD || d->PH || d->PW) return DEVIS_ERR_UNSUPPORTED;
  if (d->C1 % 4 || d->C2 % 8 || d->Cout % 4 || (d->C2 && d->C1 % 8)) return DEVIS_ERR_UNSUPPORTED;
  if (!small_enough(d->N * d->ID * d->IH * d->IW)) return DEVIS_ERR_UNSUPPORTED;
  PwArgs a = {};
  a.x1 = (const bf16*)x1; a.x2 = (const bf16*)x2; a.w = (const bf16*)w; a.bias = bias; a.res = (const bf16*)res;
  a.y1 = (bf16*)y; a.y2 = nullptr;
  a.C1 = (int)d->C1; a.C2 = (int)d->C2; a.K = a.C1 + a.C2; a.M = (int)d->Cout; a.M1 = a.M;
  a.N = (int)d->N; a.D = (int)d->ID; a.H = (int)d->IH; a.W = (int)d->IW;
  return pw_launch<PW_CONV1>(a, 1, s);
}

// ---- 1x1 conv dgrad: (dx1, dx2) = W^T dy; w_t is [Cin][Cout] (devis_repack_conv_weight_dgrad with T = 1)
int conv1_dgrad_mfma(const devis_conv_desc* d, const void* dy, const void* w_t, const void* res, void* dx1, void* dx2,
                     const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  if (plan_chunks) *plan_chunks = 0;
  if (!w_t || d->dtype != DEVIS_BF16 || d->KD != 1 || d->KH != 1 || d->KW != 1 || d->SD != 1 || d->SH != 1 || d->SW != 1) return DEVIS_ERR_UNSUPPORTED;
  if (d->PD || d->PH || d->PW) return DEVIS_ERR_UNSUPPORTED;
  if (d->Cout % 4 || d->C1 % 16 || d->C2 % 4) return DEVIS_ERR_UNSUPPORTED;    // M tiles must not straddle dx1/dx2
  if (!small_enough(d->N * d->ID * d->IH * d->IW)) return DEVIS_ERR_UNSUPPORTED;
  PwArgs a = {};
  a.x1 = (const bf16*)dy; a.w = (const bf16*)w_t; a.res = (const bf16*)res;
  a.y1 = (bf16*)dx1; a.y2 = (bf16*)dx2;
  a.C1 = (int)d->Cout; a.C2 = 0; a.K = a.C1; a.M = (int)(d->C1 + d->C2); a.M1 = (int)d->C1;
  a.N = (int)d->N; a.D = (int)d->ID; a.H = (int)d->IH; a.W = (int)d->IW;
  if ((st && st->mode) || plan_chunks) {
    if (!st || res) { if (plan_chunks) return DEVIS_OK; return DEVIS_ERR_UNSUPPORTED; }
    return pw_launch_stats(a, st, plan_chunks, s);
  }
  return pw_launch<PW_CONV1>(a, 1, s);
}

// ---- dgrad of a stride-2 3x3x3 conv (even input sizes); w_dgrad = tap-flipped [Cin][27][Cout]
int conv3s2_dgrad_mfma(const devis_conv_desc* d, const void* dy, const void* w_dgrad, const void* res, void* dx,
                       hipStream_t s) {
  if (!w_dgrad || d->dtype != DEVIS_BF16 || d->C2 != 0) return DEVIS_ERR_UNSUPPORTED;
  if (!(d->KD == 3 && d->KH == 3 && d->KW == 3 && d->PD == 1 && d->PH == 1 && d->PW == 1 && d->SD == 2 && d->SH == 2 && d->SW == 2)) return DEVIS_ERR_UNSUPPORTED;
  if ((d->ID | d->IH | d->IW) & 1) return DEVIS_ERR_UNSUPPORTED;
  if (d->Cout % 8 || d->C1 % 4) return DEVIS_ERR_UNSUPPORTED;
  if (!small_enough(d->N * d->ID * d->IH * d->IW)) return DEVIS_ERR_UNSUPPORTED;
  PwArgs a = {};
  a.x1 = (const bf16*)dy; a.w = (const bf16*)w_dgrad; a.res = (const bf16*)res; a.y1 = (bf16*)dx;
  a.Cs = (int)d->Cout; a.K = a.Cs * 8; a.M = (int)d->C1; a.M1 = a.M;
  a.N = (int)d->N; a.D = (int)d->ID / 2; a.H = (int)d->IH / 2; a.W = (int)d->IW / 2;
  a.OD = (int)d->OD; a.OH = (int)d->OH; a.OW = (int)d->OW;
    // tiles per block: 16 (M <= 32) or 8; loop the classes inside the block when that still leaves >= 1024 blocks
  const int64_t ntiles = (int64_t)a.N * a.D * a.H * cdiv(a.W, 16);
  const int64_t blocks = cdiv(ntiles, a.M >= 64 ? 8 : 16);
  (void)blocks;                      // measured on EnDown1 (128^3): looping the classes in the block is no faster (164 vs 154 us)
  return pw_launch<PW_S2_DGRAD>(a, 8, s);
}

// ---- k2-s2-p0 'down' conv (V-Net): forward gathers the 8 fine taps (the ConvT-dgrad GEMM on the packed conv
// weight [Cout][8][Cin], plus bias); dgrad scatters to the 8 fine positions (the ConvT-forward GEMM on the
// conv-dgrad packing), plus an optional residual (the skip gradient)
static bool down_shape(const devis_conv_desc* d) {
  return d->dtype == DEVIS_BF16 && d->KD == 2 && d->KH == 2 && d->KW == 2 && d->SD == 2 && d->SH == 2 && d->SW == 2 &&
         d->PD == 0 && d->PH == 0 && d->PW == 0 && d->C2 == 0 && d->ID == 2 * d->OD && d->IH == 2 * d->OH &&
         d->IW == 2 * d->OW && d->C1 % 8 == 0 && d->Cout % 8 == 0 && small_enough(d->N * d->ID * d->IH * d->IW);
}

int conv2s2_fwd_mfma(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const void* res, void* y,
                     hipStream_t s) {
  if (!down_shape(d) || res) return DEVIS_ERR_UNSUPPORTED;
  PwArgs a = {};
  a.x1 = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.y1 = (bf16*)y;
  a.Cs = (int)d->C1; a.KD = a.KH = a.KW = 2;
  a.K = a.Cs * 8; a.M = (int)d->Cout; a.M1 = a.M;
  a.N = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW;
  a.OD = (int)d->ID; a.OH = (int)d->IH; a.OW = (int)d->IW;
  return pw_launch<PW_CONVT_DGRAD>(a, 1, s);
}

int conv2s2_dgrad_mfma(const devis_conv_desc* d, const void* dy, const void* w_dgrad, const void* res, void* dx,
                       hipStream_t s) {
  if (!w_dgrad || !down_shape(d)) return DEVIS_ERR_UNSUPPORTED;
  PwArgs a = {};
  a.x1 = (const bf16*)dy; a.w = (const bf16*)w_dgrad; a.res = (const bf16*)res; a.y1 = (bf16*)dx;
  a.C1 = (int)d->Cout; a.C2 = 0; a.K = a.C1; a.Cs = (int)d->C1;
  a.KD = a.KH = a.KW = 2;
  a.M = a.Cs * 8; a.M1 = a.M;
  a.N = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW;
  a.OD = (int)d->ID; a.OH = (int)d->IH; a.OW = (int)d->IW;
  return pw_launch<PW_DOWN_DGRAD>(a, 1, s);
}

// ---- ConvTranspose (kernel == stride) forward and dgrad
// gemm_convt.hip: the wide 2-D layers
int convT_fwd_gemm(const devis_convT_desc*, const void*, const void*, const float*, void*, hipStream_t);
int convT_dgrad_gemm(const devis_convT_desc*, const void*, const void*, void*, hipStream_t);

int convT_fwd_mfma(const devis_convT_desc* d, const void* x, const void* w, const float* bias, void* y, hipStream_t s) {
  {
    const int rc = convT_fwd_gemm(d, x, w, bias, y, s);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  }
  if (d->dtype != DEVIS_BF16 || d->Cin % 8 || d->Cout % 4) return DEVIS_ERR_UNSUPPORTED;
  if (!small_enough(d->N * d->ID * d->IH * d->IW * d->KD * d->KH * d->KW)) return DEVIS_ERR_UNSUPPORTED;
  PwArgs a = {};
  a.x1 = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.y1 = (bf16*)y;
  a.C1 = (int)d->Cin; a.C2 = 0; a.K = a.C1; a.Cs = (int)d->Cout;
  a.KD = d->KD; a.KH = d->KH; a.KW = d->KW;
  a.M = a.Cs * a.KD * a.KH * a.KW; a.M1 = a.M;
  a.N = (int)d->N; a.D = (int)d->ID; a.H = (int)d->IH; a.W = (int)d->IW;
  a.OD = a.D * a.KD; a.OH = a.H * a.KH; a.OW = a.W * a.KW;
  return pw_launch<PW_CONVT_FWD>(a, 1, s);
}

// w_t: [Cin][T*Cout] = devis_repack_conv_weight_dgrad(packed convT weight viewed as [T*Cout][1][Cin])
int convT_dgrad_mfma(const devis_convT_desc* d, const void* dy, const void* w_t, void* dx, hipStream_t s) {
  {
    const int rc = convT_dgrad_gemm(d, dy, w_t, dx, s);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  }
  if (!w_t || d->dtype != DEVIS_BF16 || d->Cout % 8 || d->Cin % 4) return DEVIS_ERR_UNSUPPORTED;
  if (!small_enough(d->N * d->ID * d->IH * d->IW * d->KD * d->KH * d->KW)) return DEVIS_ERR_UNSUPPORTED;
  PwArgs a = {};
  a.x1 = (const bf16*)dy; a.w = (const bf16*)w_t; a.y1 = (bf16*)dx;
  a.Cs = (int)d->Cout; a.KD = d->KD; a.KH = d->KH; a.KW = d->KW;
  a.K = a.Cs * a.KD * a.KH * a.KW; a.M = (int)d->Cin; a.M1 = a.M;
  a.N = (int)d->N; a.D = (int)d->ID; a.H = (int)d->IH; a.W = (int)d->IW;
  a.OD = a.D * a.KD; a.OH = a.H * a.KH; a.OW = a.W * a.KW;
  return pw_launch<PW_CONVT_DGRAD>(a, 1, s);
}

}  // namespace devis
