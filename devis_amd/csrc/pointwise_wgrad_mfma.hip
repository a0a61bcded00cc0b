// MFMA weight-gradients of the halo-free layers (bf16 channels-last, gfx950):
//   1x1 conv (virtual concat):  dW[co][ci]    = sum_v dy[v][co] * x[v][ci]
//   ConvTranspose k == stride:  dW[ci][co][p] = sum_i x[i][ci]  * dy[2i+p][co]      (one grid.y slice per tap p)
// K = voxels.  Every wave takes its own 32-voxel K steps: it copies the two [voxel][channel] slabs into a
// private LDS image ([16-channel tile][32 voxels][32 B], conflict-free for the transposing reads), reads
// them back with ds_read_b64_tr_b16 and accumulates all (M tile, N tile) pairs in registers; the next
// step's global loads are in flight meanwhile.  The four waves' sums are combined in LDS, one partial per
// block goes to the workspace and k_reduce_partials (wgrad_reduce.h) adds the partials in a fixed order
// (deterministic) and writes the PyTorch fp32 layouts plus dbias.
#include "common.h"
#include <stdlib.h>
#include "mfma_common.h"
#include "wgrad_reduce.h"

namespace devis {

enum { PWG_CONV1 = 0, PWG_CONVT = 1 };

struct PwWgArgs {
  const bf16* msrc;              // M side: CONV1 dy [V][M];  CONVT x [V][M]
  const bf16* n1; const bf16* n2;  // N side: CONV1 x1/x2;   CONVT dy [OV][N]
  int M, N, C1, C2;
  int NB, D, H, W, KD, KH, KW, OD, OH, OW;
  int64_t NV;
  int iters;
  float* part;                   // [nparts][T][M][N]
  float* bpart;                  // [nparts][T][Cb] or null
  int bias_side;                 // 1: bias channels are the M side (CONV1), 2: the N side (CONVT)
  // wide ConvT layers: grid.z slabs of M x N channels out of rows ldm / ldn wide (slab z = ns * nms + ms)
  int ldm, ldn, nms;
};

__device__ __forceinline__ bf16x8 tr_pair2(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  return u.v;
}

// load up to 8 channels [c, c+8) of a row with C channels (C % 4 == 0, or C == 2), zero-extended
__device__ __forceinline__ uint4 load_piece(const bf16* row, int c, int C) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (c + 8 <= C) v = *reinterpret_cast<const uint4*>(row + c);
  else if (c + 4 <= C) { const uint2 t = *reinterpret_cast<const uint2*>(row + c); v.x = t.x; v.y = t.y; }
  else if (c + 2 <= C) v.x = *reinterpret_cast<const unsigned*>(row + c);      // the 2-class head of the 2-D nets
  return v;
}

template <int MODE, int MTO, int MTI>
__global__ void __launch_bounds__(256)
k_pw_wgrad(PwWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  char* mimg = smem + wave * (MTO + MTI) * 1024;
  char* nimg = mimg + MTO * 1024;
  const int tap = blockIdx.y;
  const int zs = blockIdx.z, ms = zs % a.nms, ns = zs / a.nms;
  const bf16* msrc = a.msrc + ms * a.M;
  const bf16* nsrc = a.n1 + ns * a.N;          // ConvT only (a 1x1 layer is one slab)
  const int tw = tap % a.KW, th = (tap / a.KW) % a.KH, td = tap / (a.KW * a.KH);

  f32x4 acc[MTO][MTI];
#pragma unroll
  for (int mo = 0; mo < MTO; ++mo)
#pragma unroll
    for (int mi = 0; mi < MTI; ++mi) acc[mo][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

  constexpr int MPPV = MTO * 2, NPPV = MTI * 2;       // 16-byte pieces per voxel in the (padded) images
  // A wave's 32-voxel step is MTO + MTI pieces per lane.  The pieces of step it+1 are fetched into
  // registers before the MFMAs of step it run, so HBM latency hides behind the LDS/MFMA phase.
  uint4 mr[MTO], nr[MTI];
  auto fetch = [&](int it) {
    const int64_t seg = ((int64_t)blockIdx.x * a.iters + it) * 4 + wave;
    const int64_t v0 = seg * 32;
#pragma unroll
    for (int k = 0; k < MTO; ++k) {
      const int i = lane + 64 * k;
      const int v = i / MPPV, pc = i % MPPV;
      const int64_t gv = v0 + v;
      mr[k] = make_uint4(0u, 0u, 0u, 0u);
      if (gv < a.NV) mr[k] = load_piece(msrc + gv * a.ldm, pc * 8, a.M);
    }
#pragma unroll
    for (int k = 0; k < MTI; ++k) {
      const int i = lane + 64 * k;
      const int v = i / NPPV, pc = i % NPPV;
      const int64_t gv = v0 + v;
      uint4 val = make_uint4(0u, 0u, 0u, 0u);
      if (gv < a.NV) {
        const int c = pc * 8;
        if (MODE == PWG_CONV1) {
          if (c < a.C1) val = load_piece(a.n1 + gv * a.C1, c, a.C1);
          else val = load_piece(a.n2 + gv * a.C2, c - a.C1, a.C2);
        } else {
          unsigned r = (unsigned)gv;          // voxel counts are < 2^31 (host check): no 64-bit divisions
          const int w = (int)(r % (unsigned)a.W); r /= (unsigned)a.W;
          const int h = (int)(r % (unsigned)a.H); r /= (unsigned)a.H;
          const int d = (int)(r % (unsigned)a.D);
          const int n = (int)(r / (unsigned)a.D);
          const int64_t ov = ((int64_t)(n * a.OD + d * a.KD + td) * a.OH + h * a.KH + th) * a.OW + w * a.KW + tw;
          val = load_piece(nsrc + ov * a.ldn, c, a.N);
        }
      }
      nr[k] = val;
    }
  };
  fetch(0);
  for (int it = 0; it < a.iters; ++it) {
    // The images are private to the wave and a wave's LDS instructions execute in order: no workgroup barrier between
    // the reads of step it-1, the stores of step it and the reads of step it (it cost two s_barrier per MFMA in the
    // 16 x 16-channel layers and kept the four waves in lockstep); wave_barrier only pins the compiler's order.
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < MTO; ++k) {
      const int i = lane + 64 * k;
      const int v = i / MPPV, pc = i % MPPV;
      const uint4 val = mr[k];
      if (a.bias_side == 1 && blockIdx.y == 0) {      // M-side bias: once, not once per tap slice
        bsum[0] += bf_lo(val.x); bsum[1] += bf_hi(val.x); bsum[2] += bf_lo(val.y); bsum[3] += bf_hi(val.y);
        bsum[4] += bf_lo(val.z); bsum[5] += bf_hi(val.z); bsum[6] += bf_lo(val.w); bsum[7] += bf_hi(val.w);
      }
      *reinterpret_cast<uint4*>(mimg + ((pc >> 1) * 32 + v) * 32 + (pc & 1) * 16) = val;
    }
#pragma unroll
    for (int k = 0; k < MTI; ++k) {
      const int i = lane + 64 * k;
      const int v = i / NPPV, pc = i % NPPV;
      const uint4 val = nr[k];
      if (a.bias_side == 2) {
        bsum[0] += bf_lo(val.x); bsum[1] += bf_hi(val.x); bsum[2] += bf_lo(val.y); bsum[3] += bf_hi(val.y);
        bsum[4] += bf_lo(val.z); bsum[5] += bf_hi(val.z); bsum[6] += bf_lo(val.w); bsum[7] += bf_hi(val.w);
      }
      *reinterpret_cast<uint4*>(nimg + ((pc >> 1) * 32 + v) * 32 + (pc & 1) * 16) = val;
    }
    __builtin_amdgcn_wave_barrier();
    if (it + 1 < a.iters) fetch(it + 1);
    const int vq = 4 * g + q;       // this lane addresses voxel rows vq and 16+vq of the 32-voxel step
    bf16x8 af[MTO];
#pragma unroll
    for (int mo = 0; mo < MTO; ++mo)
      af[mo] = tr_pair2(mimg + (mo * 32 + vq) * 32 + p * 8, mimg + (mo * 32 + 16 + vq) * 32 + p * 8);
#pragma unroll
    for (int mi = 0; mi < MTI; ++mi) {
      const bf16x8 bfr = tr_pair2(nimg + (mi * 32 + vq) * 32 + p * 8, nimg + (mi * 32 + 16 + vq) * 32 + p * 8);
#pragma unroll
      for (int mo = 0; mo < MTO; ++mo)
        acc[mo][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mo], bfr, acc[mo][mi], 0, 0, 0);
    }
  }

  // ---- the four waves' accumulators are added in wave order through LDS: one partial per block
  float* red = reinterpret_cast<float*>(smem);          // MTO*MTI KB <= 4*(MTO+MTI) KB
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int mo = 0; mo < MTO; ++mo)
#pragma unroll
        for (int mi = 0; mi < MTI; ++mi)
          *reinterpret_cast<f32x4*>(red + ((mo * MTI + mi) * 64 + lane) * 4) = acc[mo][mi];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int mo = 0; mo < MTO; ++mo)
#pragma unroll
        for (int mi = 0; mi < MTI; ++mi) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((mo * MTI + mi) * 64 + lane) * 4);
          acc[mo][mi][0] += v[0]; acc[mo][mi][1] += v[1]; acc[mo][mi][2] += v[2]; acc[mo][mi][3] += v[3];
        }
    }
  }
  const int T = gridDim.y;
  const int64_t pidx = blockIdx.x;
  if (wave == 0) {
    float* part = a.part + ((pidx * T + tap) * gridDim.z + zs) * a.M * a.N;
#pragma unroll
    for (int mo = 0; mo < MTO; ++mo)
#pragma unroll
      for (int mi = 0; mi < MTI; ++mi) {
        const int n = mi * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mo * 16 + g * 4 + j;
          if (m < a.M && n < a.N) part[(int64_t)m * a.N + n] = acc[mo][mi][j];
        }
      }
  }
  if (a.bpart) {
    // lanes with equal (lane % PPV) hold the same 8 channels: combine them through LDS (fixed order),
    // then the four waves in wave order
    const int PPV = a.bias_side == 1 ? MPPV : NPPV;
    const int Cb = a.bias_side == 1 ? a.M : a.N * (int)(gridDim.z / a.nms);     // all bias channels
    const int cboff = a.bias_side == 1 ? 0 : ns * a.N;
    const bool bwrite = a.bias_side == 1 || ms == 0;
    __syncthreads();
    float* wred = reinterpret_cast<float*>(mimg);     // [64][8] floats = 2 KB <= the wave's own region
#pragma unroll
    for (int j = 0; j < 8; ++j) wred[lane * 8 + j] = bsum[j];
    __syncthreads();
    // channel cb = 8 * piece + j; up to 128 bias channels: two rounds of 64 lanes
    // (`red` aliases wave 0's `wred`, so both rounds are read out before `red` is written)
    float s2[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int cb = r * 64 + lane;
      if (cb < PPV * 8) {
        const int pc = cb / 8, j = cb % 8;
        for (int k = pc; k < 64; k += PPV) s2[r] += wred[k * 8 + j];
      }
    }
    __syncthreads();
    red[wave * 64 + lane] = s2[0];
    red[256 + wave * 64 + lane] = s2[1];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int cb = r * 64 + lane;
        const float* rr = red + r * 256;
        if (bwrite && cb < PPV * 8 && cboff + cb < Cb)
          a.bpart[(pidx * T + tap) * Cb + cboff + cb] = ((rr[lane] + rr[64 + lane]) + rr[128 + lane]) + rr[192 + lane];
      }
    }
  }
}

struct PwWgPlan { int gx, iters, nparts, T; size_t part_bytes, total_bytes; };

static PwWgPlan pw_plan(int64_t NV, int T, int M, int N, int Cb, int nz = 1) {
  PwWgPlan p;
  const int64_t segs = cdiv(NV, 32);
  // every block leaves one fp32 partial [T-slice][M][N]: enough blocks to keep several waves per SIMD in
  // flight on the big layers (their partials are tiny), but at most ~16 MB of partials in total
  int64_t gx = cdiv(segs, 4);
  int64_t cap = 1024 / (T * nz);
  // (a partial of 64 KB or more - the 64-channel ConvT of DeUpCat4 - is worth more than the parallelism it buys: its
  // blocks spend their time adding the four waves' tiles and writing the partial; 64 -> 16 partials: 34.6 -> 21.1 us)
  const int64_t pbytes = (int64_t)T * M * N * (int64_t)sizeof(float);
  const int64_t by_bytes = ((int64_t)(pbytes >= (64 << 10) ? 2 : 16) << 20) / pbytes;
  if (by_bytes < cap) cap = by_bytes;
  if (cap < 16) cap = 16;
  if (gx > cap) gx = cap;
  p.gx = (int)gx;
  p.iters = (int)cdiv(segs, gx * 4);
  p.nparts = p.gx;
  p.T = T;
  p.part_bytes = (size_t)p.nparts * T * M * N * sizeof(float);
  p.total_bytes = p.part_bytes + (size_t)p.nparts * T * Cb * sizeof(float);
  return p;
}

template <int MODE>
static int pw_wgrad_launch(PwWgArgs& a, const PwWgPlan& pl, int Cb, float* dw, float* dbias, void* ws, hipStream_t s) {
  a.iters = pl.iters;
  a.part = (float*)ws;
  a.bpart = dbias ? (float*)((char*)ws + pl.part_bytes) : nullptr;
  int nz = 1, ntot = a.N;
  if (a.nms == 0) { a.nms = 1; a.ldm = a.M; a.ldn = a.N; }
  else { nz = a.nms * (a.ldn / a.N); ntot = a.ldn; }
  const int mto = (int)cdiv(a.M, 16), mti = (int)cdiv(a.N, 16);
  const dim3 grid((unsigned)pl.gx, (unsigned)pl.T, (unsigned)nz);
#define PWG(MO, MI) hipLaunchKernelGGL((k_pw_wgrad<MODE, MO, MI>), grid, dim3(256), 4 * (MO + MI) * 1024, s, a)
  if (mto == 4 && mti == 8) PWG(4, 8);
  else if (mto == 2 && mti == 4) PWG(2, 4);
  else if (mto == 1 && mti == 2) PWG(1, 2);
  else if (mto == 1 && mti == 1) PWG(1, 1);
  else if (mto == 4 && mti == 4) PWG(4, 4);
  else if (mto == 2 && mti == 2) PWG(2, 2);
  else if (mto == 2 && mti == 1) PWG(2, 1);          // the k2-s2 down convs (Cout = 2 Cin)
  else if (mto == 4 && mti == 2) PWG(4, 2);
  else if (mto == 8 && mti == 4) PWG(8, 4);
  else return DEVIS_ERR_UNSUPPORTED;
#undef PWG
  int rc = launch_status();
  if (rc) return rc;
  // out[(m*N + n)*T + p] = sum_k part[((k*T + p)*M + m)*N + n];  dbias[c] = sum_{k,p} bpart[(k*T + p)*Cb + c]
  RedArgs r;
  r.part = a.part; r.bpart = a.bpart; r.dw = dw; r.dbias = dbias;
  r.nparts = pl.nparts; r.nbparts = pl.nparts * pl.T; r.Cb = Cb;
  r.pstride = (int64_t)pl.T * nz * a.M * a.N; r.total = r.pstride;
  r.mode = RED_PW; r.A = pl.T; r.B = nz * a.M * a.N; r.cin_real = 0;
  if (nz > 1) { r.nms = a.nms; r.msub = a.M; r.nsub = a.N; r.ntot = ntot; }
  return launch_reduce_partials(r, s);
}

static bool pw_tiles_ok(int M, int N) {
  const int mto = (M + 15) / 16, mti = (N + 15) / 16;
  return (mto == 4 && mti == 8) || (mto == 2 && mti == 4) || (mto == 1 && mti == 2) || (mto == 1 && mti == 1) ||
         (mto == 4 && mti == 4) || (mto == 2 && mti == 2) || (mto == 2 && mti == 1) || (mto == 4 && mti == 2) ||
         (mto == 8 && mti == 4);
}

static bool conv1_wgrad_ok(const devis_conv_desc* d) {
  if (d->dtype != DEVIS_BF16 || d->KD != 1 || d->KH != 1 || d->KW != 1 || d->SD != 1 || d->SH != 1 || d->SW != 1) return false;
  if (d->PD || d->PH || d->PW) return false;
  if ((d->Cout % 4 && d->Cout != 2) || d->C1 % 8 || d->C2 % 8) return false;
  return pw_tiles_ok((int)d->Cout, (int)(d->C1 + d->C2));
}

size_t conv1_wgrad_mfma_workspace(const devis_conv_desc* d) {
  if (!conv1_wgrad_ok(d)) return 0;
  return pw_plan(d->N * d->ID * d->IH * d->IW, 1, (int)d->Cout, (int)(d->C1 + d->C2), (int)d->Cout).total_bytes;
}

int conv1_wgrad_mfma(const devis_conv_desc* d, const void* x1, const void* x2, const void* dy, float* dw, float* dbias,
                     void* ws, size_t ws_bytes, hipStream_t s) {
  if (!conv1_wgrad_ok(d)) return DEVIS_ERR_UNSUPPORTED;
  PwWgArgs a = {};
  a.msrc = (const bf16*)dy; a.n1 = (const bf16*)x1; a.n2 = (const bf16*)x2;
  a.M = (int)d->Cout; a.N = (int)(d->C1 + d->C2); a.C1 = (int)d->C1; a.C2 = (int)d->C2;
  a.NV = d->N * d->ID * d->IH * d->IW;
  a.KD = a.KH = a.KW = 1;
  a.bias_side = dbias ? 1 : 0;
  const PwWgPlan pl = pw_plan(a.NV, 1, a.M, a.N, a.M);
  if (!ws || ws_bytes < pl.total_bytes) return DEVIS_ERR_WORKSPACE;
  return pw_wgrad_launch<PWG_CONV1>(a, pl, a.M, dw, dbias, ws, s);
}

// wide layers (the 2-D U-Net's 256 -> 128 and 512 -> 256) run as grid.z slabs of 128 x 64 channels
static bool convT_wide(const devis_convT_desc* d) { return d->Cin > 128 && d->Cin % 128 == 0 && d->Cout % 64 == 0; }

static bool convT_wgrad_ok(const devis_convT_desc* d) {
  if (d->dtype != DEVIS_BF16 || d->Cin % 8 || d->Cout % 8) return false;
  if (d->N * d->ID * d->IH * d->IW * d->KD * d->KH * d->KW >= ((int64_t)1 << 31)) return false;   // 32-bit voxel indices
  return convT_wide(d) || pw_tiles_ok((int)d->Cin, (int)d->Cout);
}

size_t convT_wgrad_mfma_workspace(const devis_convT_desc* d) {
  if (!convT_wgrad_ok(d)) return 0;
  const int nz = convT_wide(d) ? (int)((d->Cin / 128) * (d->Cout / 64)) : 1;
  return pw_plan(d->N * d->ID * d->IH * d->IW, d->KD * d->KH * d->KW, (int)d->Cin, (int)d->Cout, (int)d->Cout, nz).total_bytes;
}

int convT_wgrad_mfma(const devis_convT_desc* d, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                     size_t ws_bytes, hipStream_t s) {
  if (!convT_wgrad_ok(d)) return DEVIS_ERR_UNSUPPORTED;
  PwWgArgs a = {};
  a.msrc = (const bf16*)x; a.n1 = (const bf16*)dy;
  a.M = (int)d->Cin; a.N = (int)d->Cout;
  a.NB = (int)d->N; a.D = (int)d->ID; a.H = (int)d->IH; a.W = (int)d->IW;
  a.KD = d->KD; a.KH = d->KH; a.KW = d->KW;
  a.OD = a.D * a.KD; a.OH = a.H * a.KH; a.OW = a.W * a.KW;
  a.NV = d->N * d->ID * d->IH * d->IW;
  a.bias_side = dbias ? 2 : 0;
  const int T = d->KD * d->KH * d->KW;
  const int nz = convT_wide(d) ? (int)((d->Cin / 128) * (d->Cout / 64)) : 1;
  const PwWgPlan pl = pw_plan(a.NV, T, a.M, a.N, a.N, nz);      // partial bytes: the whole [Cin][Cout] weight
  if (!ws || ws_bytes < pl.total_bytes) return DEVIS_ERR_WORKSPACE;
  if (nz > 1) { a.ldm = a.M; a.ldn = a.N; a.nms = a.M / 128; a.M = 128; a.N = 64; }
  return pw_wgrad_launch<PWG_CONVT>(a, pl, (int)d->Cout, dw, dbias, ws, s);
}

// ---- k2-s2-p0 'down' conv wgrad: dw[co][ci][t] = sum dy_coarse[co] * x_fine[tap t][ci] - the ConvT wgrad GEMM with
// the coarse tensor on the M side - and dbias[co] = sum dy_coarse[co] (M-side bias)
static bool down_wgrad_ok(const devis_conv_desc* d) {
  return d->dtype == DEVIS_BF16 && d->KD == 2 && d->KH == 2 && d->KW == 2 && d->SD == 2 && d->SH == 2 && d->SW == 2 &&
         d->PD == 0 && d->PH == 0 && d->PW == 0 && d->C2 == 0 && d->ID == 2 * d->OD && d->IH == 2 * d->OH &&
         d->IW == 2 * d->OW && d->C1 % 8 == 0 && d->Cout % 8 == 0 && pw_tiles_ok((int)d->Cout, (int)d->C1) &&
         d->N * d->ID * d->IH * d->IW < ((int64_t)1 << 31);
}

size_t conv2s2_wgrad_mfma_workspace(const devis_conv_desc* d) {
  if (!down_wgrad_ok(d)) return 0;
  return pw_plan(d->N * d->OD * d->OH * d->OW, 8, (int)d->Cout, (int)d->C1, (int)d->Cout).total_bytes;
}

int conv2s2_wgrad_mfma(const devis_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                       size_t ws_bytes, hipStream_t s) {
  if (!down_wgrad_ok(d)) return DEVIS_ERR_UNSUPPORTED;
  PwWgArgs a = {};
  a.msrc = (const bf16*)dy; a.n1 = (const bf16*)x;
  a.M = (int)d->Cout; a.N = (int)d->C1;
  a.NB = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW;
  a.KD = a.KH = a.KW = 2;
  a.OD = (int)d->ID; a.OH = (int)d->IH; a.OW = (int)d->IW;
  a.NV = d->N * d->OD * d->OH * d->OW;
  a.bias_side = dbias ? 1 : 0;
  const PwWgPlan pl = pw_plan(a.NV, 8, a.M, a.N, a.M);
  if (!ws || ws_bytes < pl.total_bytes) return DEVIS_ERR_WORKSPACE;
  return pw_wgrad_launch<PWG_CONVT>(a, pl, a.M, dw, dbias, ws, s);
}

}  // namespace devis
