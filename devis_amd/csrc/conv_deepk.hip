// MFMA 3x3x3 convolution for the deep, small-volume layers (64 channels at 32^3, 128 channels at 16^3; bf16,
// channels-last, stride 1, gfx950), third design: WEIGHTS STATIONARY IN REGISTERS, K SPLIT OVER THE WAVES.
//
// The r02 kernels it replaces (conv_deep.hip / conv_deep64.hip, deleted) streamed both operands through LDS per 32-channel K chunk: every MFMA needed about one
// 1 KB LDS fragment read (twice what the LDS can deliver at the matrix rate), and every chunk pays two barriers and a
// staging pass with one wave per SIMD (16-18 us / 26-30 us per launch for 2.9 / 5.8 us of matrix time).  Here
//   * the block's activation halo tile (ALL input channels) is staged into LDS once - one barrier for the whole K loop;
//   * wave w owns ONE 32-channel K chunk for 32 output channels and keeps its 27 x [32 cout][32 cin] weight
//     fragments in registers (216 VGPRs, loaded straight from global memory in fragment shape while the tile is
//     staged; the first taps start as soon as their fragments land): the MFMA loop reads only activation fragments,
//     one ds_read_b128 per two MFMAs with the address in the instruction's offset field (no vector address work);
//   * the waves' partial sums meet through LDS in a fixed order (deterministic), each wave finishing a share of the
//     tile's rows in the usual epilogue (bias, residual, 8-byte stores, norm statistics).
// 128 channels: 4 chunks = the 4 waves, 32 cout per block (grid.y = 4), (2,4,16) tile.  64 channels: 2 chunks x 2
// cout groups = the 4 waves, all 64 cout per block, (4,4,16) tile.  One round of 256 blocks either way at BraTS B = 2.
// dgrad of these layers is the same kernel on the tap-flipped [Cin][27][Cout] weights.
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {

struct DeepKArgs {
  const bf16* x;
  const bf16* w;      // [Cout][27][CIN] in the fragment-major order of csrc/layout.hip (deepk_index)
  const float* bias;  // [Cout] or null
  const float* sc;    // [N][CIN] or null
  const float* sh;
  const bf16* res;    // like y, or null
  bf16* y;
  int N, D, H, W, Cout;
  int ntd, nth, ntw;
  StatsArgs st;       // norm statistics gathered in the epilogue (mode 0: off)
#ifdef DEVIS_PHASE_TIMING
  long long* dbg;     // diagnostic build (tools/bench_deepk.hip): [block][8] shader-clock stamps + realtime
#endif
};

#ifdef DEVIS_PHASE_TIMING
#define DK_STAMP(i) { if (lane == 0 && wave == 0) a.dbg[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); }
#else
#define DK_STAMP(i)
#endif

template <int CIN, int TD, int TH>
struct DeepKCfg {
  static constexpr int NCH = CIN / 32;              // K chunks = waves that share a cout group
  static constexpr int MG = 4 / NCH;                // cout groups (of 32) per block
  static constexpr int NTB = TD * TH;               // rows of 16 voxels in the tile
  static constexpr int HV = (TD + 2) * (TH + 2) * 18;
  static constexpr int XBYTES = NCH * HV * 64;      // the halo tile, one [HV][64 B] image per chunk
  static constexpr int RBYTES = 4 * 2 * NTB * 1024; // the waves' partial sums
  static constexpr int LDS = XBYTES > RBYTES ? XBYTES : RBYTES;
};

template <int CIN, int TD, int TH, bool PRO, int STATS>
__global__ void __launch_bounds__(256)
k_conv3_deepk(DeepKArgs a) {
  using Cfg = DeepKCfg<CIN, TD, TH>;
  constexpr int NCH = Cfg::NCH, MG = Cfg::MG, MT = 2, NTB = Cfg::NTB, RPW = NTB / NCH;
  constexpr int HH = TH + 2, HW = 18, HV = Cfg::HV;
  constexpr int PPV = CIN / 8;                       // 16-byte pieces per voxel
  constexpr int XCH = HV * 64;                       // bytes of one chunk image
  static_assert(256 % PPV == 0 && NTB % NCH == 0 && 4 % NCH == 0, "tile / channel split");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int kc = wave % NCH, mg = wave / NCH;        // this wave's K chunk and cout group

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
  const int co_base = blockIdx.y * (32 * MG);

  DK_STAMP(0)
#ifdef DEVIS_PHASE_TIMING
  if (lane == 0 && wave == 0) a.dbg[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  // ---- 1. the halo tile: every load in flight at once.  Wave w fetches halo rows w, w + 4, ... (a row = 18 voxels of
  // one (hd, hh)), so everything that depends on the row - bounds, base address - is wave-uniform (scalar unit) and a
  // lane only adds its fixed (voxel-in-row, 8-channel piece) offset.  (The first version split the tile's 16-byte
  // pieces evenly over the threads: ~40 vector instructions of index arithmetic per piece, 7k cycles = 3 us before the
  // first load - more than the loads themselves, and the reason the r02 kernels of these layers were no faster.)
  constexpr int HD = TD + 2, NR = HD * HH, RW = NR / 4;  // rows, rows per wave
  constexpr int S = (HW * PPV + 63) / 64;                 // 64-lane slots per row (the last one partly idle)
  constexpr int VPS = 64 / PPV;                           // voxels per slot
  static_assert(NR % 4 == 0, "halo rows must split over 4 waves");
  const int pv = lane % PPV;                              // this lane's 8-channel piece of every voxel it stages
  const int wu = __builtin_amdgcn_readfirstlane(wave);
  int goff[S];
  unsigned loff[S];
  bool lok[S], lst[S];
#pragma unroll
  for (int q = 0; q < S; ++q) {
    const int hw = lane / PPV + q * VPS, iw = ow0 - 1 + hw;
    lst[q] = hw < HW;
    lok[q] = lst[q] && iw >= 0 && iw < a.W;
    goff[q] = hw * CIN + pv * 8;
    // voxel u = row * 18 + hw: (u >> 1) & 3 = (row + (hw >> 1)) & 3 and row & 3 = w for every row of this wave
    loff[q] = (unsigned)((pv >> 2) * XCH + hw * 64 + (((pv & 3) ^ ((wu + (hw >> 1)) & 3)) << 4));
  }
  u32x4 xr[RW][S];
  bool rok[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int row = wu + 4 * i, hd = row / HH, hh = row - hd * HH;
    const int id = od0 - 1 + hd, ih = oh0 - 1 + hh;
    rok[i] = id >= 0 && id < a.D && ih >= 0 && ih < a.H;
    const bf16* rp = a.x + ((ptrdiff_t)((n * a.D + id) * a.H + ih) * a.W + (ow0 - 1)) * CIN;
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const bf16* p = (rok[i] && lok[q]) ? rp + goff[q] : a.x;
#ifdef DK_SKIP_X
      xr[i][q] = (u32x4){(unsigned)(size_t)p, 0u, 0u, 0u};
#else
      xr[i][q] = *reinterpret_cast<const u32x4*>(p);
#endif
    }
  }
  // ---- 2. this wave's weights, in MFMA A-fragment shape (lane = cout row l15, K piece lq; fragment-major layout,
  // csrc/layout.hip: the wave's 27 x 2 fragments are 54 consecutive 1 KB runs).  Only the kd = 0 plane (9 taps) is
  // requested here, behind the tile's loads: ISSUING vector loads is what this phase waits on (measured: 63 loads per
  // wave took 9-10k cycles to issue with every CU pulling its 330 KB through L2 at once), so the kd = 1 and kd = 2
  // planes are requested from inside the MFMA loop, one halo plane before their first use, and stream in under it.
  // The order is pinned with scheduling barriers: left to itself the compiler moved a tile load behind the weights,
  // and the block barrier then waited for all of them before the first MFMA.
  bf16x8 wf[27][MT];
  const bf16* wp = a.w + ((size_t)((co_base / 32 + mg) * NCH + kc) * (27 * MT * 64) + lane) * 8;
#ifdef DK_SKIP_W
#define DK_WLOAD(p) (*reinterpret_cast<const bf16x8*>(a.w + (lane + (((p) - wp) >> 9)) * 8))     /* one hot 1.5 KB */
#else
#define DK_WLOAD(p) (*reinterpret_cast<const bf16x8*>(p))
#endif
#define DK_LOAD_PLANE(KD)                                                                         \
  _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_)                                                \
    _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                             \
      wf[(KD) * 9 + t_][m_] = DK_WLOAD(wp + (((KD) * 9 + t_) * MT + m_) * 512);
  __builtin_amdgcn_sched_barrier(0);
  DK_LOAD_PLANE(0)
  __builtin_amdgcn_sched_barrier(0);
  DK_STAMP(6)
  // ---- 3. stage the tile (norm affine + ReLU, zero padding)
  {
    float scv[8], shv[8];
    if (PRO) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { scv[j] = a.sc[n * CIN + pv * 8 + j]; shv[j] = a.sh[n * CIN + pv * 8 + j]; }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int q = 0; q < S; ++q) {
        u32x4 v = xr[i][q];
        if (PRO) v = affine_relu8(v, scv, shv);
        if (!(rok[i] && lok[q])) v = (u32x4){0u, 0u, 0u, 0u};
        if (lst[q]) *reinterpret_cast<u32x4*>(smem + loff[q] + (wu + 4 * i) * (HW * 64)) = v;
      }
  }
  __builtin_amdgcn_sched_barrier(0);
  // B-fragment addresses: voxel u = (compile-time tap / row offset) + l15; the swizzle term depends on (u >> 1) & 3,
  // i.e. on (offset mod 8) and the lane: eight per-lane bases, the rest goes into the instruction's offset field
  unsigned boff[8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
    boff[r] = (unsigned)(kc * XCH + l15 * 64 + ((lq ^ (((r + l15) >> 1) & 3)) << 4));

  f32x4 acc[MT][NTB];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NTB; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  DK_STAMP(1)
  __syncthreads();
  DK_STAMP(2)

  // ---- 4. every activation fragment (halo row x kw) is read ONCE and feeds all the (kd, kh) taps / output rows that
  // use it: up to 9 x 2 MFMAs per ds_read_b128, a quarter of the LDS reads of a tap-major loop.  Fragments are read
  // PF steps ahead of their MFMAs (a rolling window of registers).  Weight plane kd is first needed at halo plane
  // hd = kd and is requested one plane earlier.
  constexpr int NF = HD * HH * 3, PF = 6;
  bf16x8 bw[PF];
#define DK_FRAG(F) (*reinterpret_cast<const bf16x8*>(smem + boff[(((F) / 3) * HW + (F) % 3) & 7] + (((F) / 3) * HW + (F) % 3) * 64))
#pragma unroll
  for (int f = 0; f < PF; ++f) bw[f] = DK_FRAG(f);               // f = (hd * HH + hh) * 3 + kw
#pragma unroll
  for (int hd = 0; hd < HD; ++hd) {
    if (hd < 2) {
      __builtin_amdgcn_sched_barrier(0);
      if (hd == 0) { DK_LOAD_PLANE(1) } else { DK_LOAD_PLANE(2) }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int hh = 0; hh < HH; ++hh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int f = (hd * HH + hh) * 3 + kw;
        const bf16x8 b = bw[f % PF];
        // pinned: left alone the compiler sinks every read to just before its first MFMA (one register set, the whole
        // LDS latency exposed 108 times: 20k cycles for 13.8k of matrix work)
        __builtin_amdgcn_sched_barrier(0);
        if (f + PF < NF) bw[f % PF] = DK_FRAG(f + PF);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int od = hd - kd, oh = hh - kh;
            if (od >= 0 && od < TD && oh >= 0 && oh < TH) {
              const int j = od * TH + oh, t = (kd * 3 + kh) * 3 + kw;
#pragma unroll
              for (int m = 0; m < MT; ++m)
                acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][m], b, acc[m][j], 0, 0, 0);
            }
          }
      }
  }

  DK_STAMP(3)
  // ---- 5. the chunks' partial sums meet in LDS; wave (kc, mg) finishes rows kc*RPW .. of cout group mg
  __syncthreads();                               // every wave has read its last fragment: the tile may be overwritten
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NTB; ++j)
      *reinterpret_cast<f32x4*>(smem + (((wave * MT + m) * NTB + j) * 64 + lane) * 16) = acc[m][j];
  __syncthreads();
  f32x4 sum[MT][RPW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int j = kc * RPW + r;
      f32x4 s = *reinterpret_cast<const f32x4*>(smem + ((((mg * NCH + 0) * MT + m) * NTB + j) * 64 + lane) * 16);
#pragma unroll
      for (int cc = 1; cc < NCH; ++cc) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(smem + ((((mg * NCH + cc) * MT + m) * NTB + j) * 64 + lane) * 16);
        s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
      }
      sum[m][r] = s;
    }

  DK_STAMP(4)
  // ---- 6. epilogue: + bias (+ residual), 4 consecutive cout of one voxel per lane -> 8-byte stores
  float f0[MT][4], f1[MT][4], scs[MT][4], shs[MT][4];
  if (STATS) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f0[m][k] = 0.f; f1[m][k] = 0.f;
        if (STATS == 2) {
          scs[m][k] = a.st.sc[n * a.Cout + co_base + mg * 32 + m * 16 + lq * 4 + k];
          shs[m][k] = a.st.sh[n * a.Cout + co_base + mg * 32 + m * 16 + lq * 4 + k];
        }
      }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int j = kc * RPW + r;
    const int od = od0 + j / TH, oh = oh0 + j % TH, ow = ow0 + l15;
    const bool in = od < a.D && oh < a.H && ow < a.W;
    const size_t vox = in ? (size_t)((n * a.D + od) * a.H + oh) * a.W + ow : 0;
    // loads (residual, the normalised tensor of mode 2) before the stores of this row
    uint2 xref[MT], rres[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xref[m] = make_uint2(0u, 0u); rres[m] = make_uint2(0u, 0u);
      const size_t e = vox * a.Cout + co_base + mg * 32 + m * 16 + lq * 4;
      if (STATS == 2) xref[m] = *reinterpret_cast<const uint2*>(a.st.xref + e);
      if (a.res) rres[m] = *reinterpret_cast<const uint2*>(a.res + e);
    }
    if (!in) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co = co_base + mg * 32 + m * 16 + lq * 4;
      float v0 = sum[m][r][0], v1 = sum[m][r][1], v2 = sum[m][r][2], v3 = sum[m][r][3];
      if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
      }
      v0 += bf_lo(rres[m].x); v1 += bf_hi(rres[m].x); v2 += bf_lo(rres[m].y); v3 += bf_hi(rres[m].y);
      uint2 o;
      o.x = pack_bf2(v0, v1);
      o.y = pack_bf2(v2, v3);
      *reinterpret_cast<uint2*>(a.y + vox * a.Cout + co) = o;
      if (STATS) stats_accum<STATS>(o, STATS == 2 ? xref[m] : o, scs[m], shs[m], f0[m], f1[m]);
    }
  }
  DK_STAMP(5)
  if (STATS) {
    // the 16 lanes that share channels add up by DPP, the waves of a cout group in fp64 through LDS, fixed order
    __syncthreads();               // every wave is done reading the partial sums
    double* red = reinterpret_cast<double*>(smem);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s0 = row16_sum(f0[m][j]), s1 = row16_sum(f1[m][j]);
        if (l15 == 0) {
          red[wave * (MT * 32) + (m * 16 + lq * 4 + j) * 2 + 0] = (double)s0;
          red[wave * (MT * 32) + (m * 16 + lq * 4 + j) * 2 + 1] = (double)s1;
        }
      }
    __syncthreads();
    if (tid < MG * MT * 32) {
      const int g = tid / (MT * 32), idx = tid % (MT * 32);
      double s = red[(g * NCH + 0) * (MT * 32) + idx];
#pragma unroll
      for (int cc = 1; cc < NCH; ++cc) s += red[(g * NCH + cc) * (MT * 32) + idx];
      const int chunk = tile % (a.ntw * a.nth * a.ntd);
      a.st.partial[(((size_t)n * a.st.chunks + chunk) * a.Cout + co_base) * 2 + tid] = s;
    }
  }
}

template <int CIN, int TD, int TH, bool PRO, int STATS>
static int launch_deepk_k(const DeepKArgs& a, dim3 grid, hipStream_t s) {
  auto k = k_conv3_deepk<CIN, TD, TH, PRO, STATS>;
  constexpr int lds = DeepKCfg<CIN, TD, TH>::LDS;
  static bool attr_done = false;
  if (!attr_done) { if (check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds))) return DEVIS_ERR_HIP; attr_done = true; }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  return launch_status();
}

// plan_chunks != null: only report the statistics chunks per sample this launch would leave
template <int CIN, int TD, int TH>
static int launch_deepk(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
                        const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks,
                        hipStream_t s) {
  DeepKArgs a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.sc = sc; a.sh = sh; a.res = (const bf16*)res; a.y = (bf16*)y;
  a.N = (int)d->N; a.D = (int)d->OD; a.H = (int)d->OH; a.W = (int)d->OW; a.Cout = (int)d->Cout;
  a.ntd = (int)cdiv(a.D, TD); a.nth = (int)cdiv(a.H, TH); a.ntw = (int)cdiv(a.W, 16);
  const int64_t tiles = (int64_t)a.N * a.ntd * a.nth * a.ntw;
  if (tiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  if (plan_chunks) {
    *plan_chunks = (st && st->mode == 2 && sc) ? 0 : (int64_t)a.ntd * a.nth * a.ntw;
    return DEVIS_OK;
  }
  const dim3 grid((unsigned)tiles, (unsigned)(d->Cout / (32 * DeepKCfg<CIN, TD, TH>::MG)));
  a.st = StatsArgs{};
  const int mode = st ? st->mode : 0;
  if (mode) {
    a.st = *st;
    if (a.st.chunks != a.ntd * a.nth * a.ntw || a.st.C != a.Cout) return DEVIS_ERR_BAD_ARG;
    if (mode == 2 && sc) return DEVIS_ERR_UNSUPPORTED;
  }
  if (mode == 1) return sc ? launch_deepk_k<CIN, TD, TH, true, 1>(a, grid, s) : launch_deepk_k<CIN, TD, TH, false, 1>(a, grid, s);
  if (mode == 2) return launch_deepk_k<CIN, TD, TH, false, 2>(a, grid, s);
  return sc ? launch_deepk_k<CIN, TD, TH, true, 0>(a, grid, s) : launch_deepk_k<CIN, TD, TH, false, 0>(a, grid, s);
}

// stride-1 3x3x3 with Cin == Cout in {64, 128}, bf16; anything else is unsupported
int conv3_deepk(const devis_conv_desc* d, const void* x, const void* w, const float* bias, const float* sc,
                const float* sh, const void* res, void* y, const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  if (d->dtype != DEVIS_BF16 || d->C2 != 0 || d->SD != 1 || d->C1 != d->Cout) return DEVIS_ERR_UNSUPPORTED;
  if (d->OD != d->ID || d->OH != d->IH || d->OW != d->IW) return DEVIS_ERR_UNSUPPORTED;
  if (d->N * d->ID * d->IH * d->IW * d->C1 >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  if (d->C1 == 128) return launch_deepk<128, 2, 4>(d, x, w, bias, sc, sh, res, y, st, plan_chunks, s);
  if (d->C1 == 64) return launch_deepk<64, 4, 4>(d, x, w, bias, sc, sh, res, y, st, plan_chunks, s);
  return DEVIS_ERR_UNSUPPORTED;
}

}  // namespace devis
