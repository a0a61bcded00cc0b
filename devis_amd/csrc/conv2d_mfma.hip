// MFMA 3x3 convolutions of the 2-D U-Nets (models/lib/UNet2DZoo.py:8-21: DoubleConv at 32..512 channels;
// bf16, channels-last [N][H][W][C], stride 1, padding 1, gfx950).  These layers are matrix-core bound
// (cin*cout*18 FLOP per pixel against (cin+cout)*2 bytes), so the kernels are built around the MFMA : LDS
// read ratio:
//
//   forward / dgrad  k_conv2_mfma<MT,NT,PRO,STATS>: a block computes MT*16 cout of a (4*NT x 16)-pixel tile,
//     looping over 32-channel K chunks of the (virtually concatenated) input.  Per chunk the halo tile (norm
//     affine + ReLU applied) and the chunk's [9][MT*16][32] weight slab are staged in LDS with bulk coalesced
//     loads, the next chunk is fetched into registers while the 9 taps run, and every wave owns NT rows x
//     MT cout tiles: MT + NT LDS reads feed MT*NT MFMAs per tap (16 MFMAs per 8 reads at MT = NT = 4).
//     dgrad is the same kernel on the tap-flipped [Cin][9][Cout] weights; for a layer with two inputs it
//     writes the two gradients to their own tensors.
//   wgrad  k_wgrad2_mfma<PRO>: dW[co][t][ci] = sum_p dy[p][co] * act(x[p+t][ci]), K = pixels.  A block owns a
//     32x32 (cout x cin) slab for all 9 taps and walks 16x16-pixel tiles; the four waves split the tile's
//     K steps (32 pixels each, operands read with the transposing LDS reads), their accumulators are added
//     in wave order at the end; per-block fp32 partials + the fixed-order reduction of wgrad_reduce.h.
#include "common.h"
#include "mfma_common.h"
#include "wgrad_reduce.h"

namespace devis {

struct Conv2Args {
  const bf16* x1;     // [N][H][W][C1]
  const bf16* x2;     // [N][H][W][C2] or null
  const bf16* w;      // [Cout][9][Cin]
  const float* bias;  // [Cout] or null
  const float* sc;    // [N][Cin] or null
  const float* sh;
  const bf16* res;    // like y1 (Co2 == 0), or null
  bf16* y1;           // channels [0, Co1)
  bf16* y2;           // channels [Co1, Cout) or null
  int N, H, W, C1, C2, Cin, Cout, Co1, Co2;
  int nth, ntw;
  StatsArgs st;
};

template <int MT, int NT, bool PRO, int STATS>
__global__ void __launch_bounds__(256)
k_conv2_mfma(Conv2Args a) {
  constexpr int TH = 4 * NT, HH = TH + 2, HW = 18, HV = HH * HW;
  constexpr int XPT = (HV * 4 + 255) / 256;           // 16-byte pieces per thread: activations
  constexpr int WROWS = 9 * MT * 16;
  constexpr int WPT = (WROWS * 4 + 255) / 256;        // ... weights
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
  const int n = tile / (a.ntw * a.nth);
  const int oh0 = thi * TH, ow0 = twi * 16;
  const int co_base = blockIdx.y * (MT * 16);
  const int piece = tid & 3;

  // ---- per-thread slots (the same for every chunk, apart from the channel offset)
  unsigned xpix[XPT], xl[XPT];
  bool xok[XPT];
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int i = tid + k * 256;
    const int u = i >> 2;
    const int hw = u % HW, hh = u / HW;
    const int ih = oh0 - 1 + hh, iw = ow0 - 1 + hw;
    xok[k] = i < HV * 4 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
    xpix[k] = xok[k] ? (unsigned)((n * a.H + ih) * a.W + iw) : 0u;
    xl[k] = (unsigned)(u * 64 + ((piece ^ ((u >> 1) & 3)) << 4));
  }
  unsigned wg[WPT], wl[WPT];
  bool wok[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int i = tid + k * 256;
    const int row = i >> 2;                     // tap * (MT*16) + co
    const int t = row / (MT * 16), co = row % (MT * 16);
    wok[k] = i < WROWS * 4;
    wg[k] = wok[k] ? (unsigned)(((co_base + co) * 9 + t) * a.Cin + piece * 8) : 0u;
    wl[k] = (unsigned)(row * 64 + ((piece ^ ((row >> 1) & 3)) << 4));
  }

  auto fetch = [&](int cc, u32x4 (&xr)[XPT], u32x4 (&wr)[WPT]) __attribute__((always_inline)) {
    const int c0 = cc * 32;
    const bool second = c0 >= a.C1;
    const bf16* xb = second ? a.x2 + (c0 - a.C1) : a.x1 + c0;
    const unsigned cs = second ? a.C2 : a.C1;
#pragma unroll
    for (int k = 0; k < XPT; ++k) xr[k] = *reinterpret_cast<const u32x4*>(xb + xpix[k] * cs + piece * 8);
#pragma unroll
    for (int k = 0; k < WPT; ++k) wr[k] = *reinterpret_cast<const u32x4*>(a.w + wg[k] + c0);
  };
  auto stage = [&](int cc, u32x4 (&xr)[XPT], u32x4 (&wr)[WPT]) __attribute__((always_inline)) {
    float scv[8], shv[8];
    if (PRO) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        scv[j] = a.sc[n * a.Cin + cc * 32 + piece * 8 + j];
        shv[j] = a.sh[n * a.Cin + cc * 32 + piece * 8 + j];
      }
    }
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      u32x4 v = xr[k];
      if (PRO) v = affine_relu8(v, scv, shv);
      if (!xok[k]) v = (u32x4){0u, 0u, 0u, 0u};
      if (k < XPT - 1 || tid + k * 256 < HV * 4) *reinterpret_cast<u32x4*>(xs + xl[k]) = v;
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

  // LDS pixel index of each of this wave's rows at tap (0,0), lane's pixel included
  int ubase[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) ubase[j] = (wave * NT + j) * HW + l15;

  const int nch = a.Cin >> 5;
  u32x4 xr[XPT], wr[WPT];
  fetch(0, xr, wr);
#pragma unroll 1
  for (int cc = 0; cc < nch; ++cc) {
    __syncthreads();                       // the previous chunk's fragments have all been read
    stage(cc, xr, wr);
    __syncthreads();
    if (cc + 1 < nch) fetch(cc + 1, xr, wr);
    __builtin_amdgcn_sched_barrier(0);

    auto read_tap = [&](int t, bf16x8 (&af)[MT], bf16x8 (&bf)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = t * (MT * 16) + m * 16 + l15;
        af[m] = *reinterpret_cast<const bf16x8*>(ws + row * 64 + ((lq ^ ((row >> 1) & 3)) << 4));
      }
      const int toff = (t / 3) * HW + t % 3;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int u = ubase[j] + toff;
        bf[j] = *reinterpret_cast<const bf16x8*>(xs + u * 64 + ((lq ^ ((u >> 1) & 3)) << 4));
      }
    };
    bf16x8 afA[MT], bfA[NT], afB[MT], bfB[NT];
    read_tap(0, afA, bfA);
#pragma unroll
    for (int t = 0; t < 9; t += 2) {
      if (t + 1 < 9) read_tap(t + 1, afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afA[m], bfA[j], acc[m][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < 9) read_tap(t + 2, afA, bfA);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < 9) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afB[m], bfB[j], acc[m][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: + bias (+ residual), 4 consecutive cout of one pixel per lane -> 8-byte stores
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
  // mode 2 reads the normalised tensor at this lane's outputs: all loads first, then the stores
  uint2 xref[MT][NT];
  if (STATS == 2) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int oh = oh0 + wave * NT + j, ow = ow0 + l15;
      const bool in = oh < a.H && ow < a.W;
      const size_t pix = in ? (size_t)(n * a.H + oh) * a.W + ow : 0;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        xref[m][j] = *reinterpret_cast<const uint2*>(a.st.xref + pix * a.Cout + co_base + m * 16 + lq * 4);
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int oh = oh0 + wave * NT + j, ow = ow0 + l15;
    if (oh >= a.H || ow >= a.W) continue;
    const size_t pix = (size_t)(n * a.H + oh) * a.W + ow;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co = co_base + m * 16 + lq * 4;
      float v0 = acc[m][j][0], v1 = acc[m][j][1], v2 = acc[m][j][2], v3 = acc[m][j][3];
      if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
      }
      if (a.res) {
        const uint2 r = *reinterpret_cast<const uint2*>(a.res + pix * a.Cout + co);
        v0 += bf_lo(r.x); v1 += bf_hi(r.x); v2 += bf_lo(r.y); v3 += bf_hi(r.y);
      }
      uint2 o;
      o.x = pack_bf2(v0, v1);
      o.y = pack_bf2(v2, v3);
      bf16* dst = co < a.Co1 ? a.y1 + pix * a.Co1 + co : a.y2 + pix * a.Co2 + (co - a.Co1);
      *reinterpret_cast<uint2*>(dst) = o;
      if (STATS) {
        stats_accum<STATS>(o, STATS == 2 ? xref[m][j] : o, scs[m], shs[m], f0[m], f1[m]);
      }
    }
  }
  if (STATS) {
    __syncthreads();               // every wave is done reading its LDS fragments
    const int chunk = tile % (a.ntw * a.nth);
    stats_finish_dpp<MT>(f0, f1, reinterpret_cast<double*>(smem),
                     a.st.partial + (((size_t)n * a.st.chunks + chunk) * a.Cout + co_base) * 2);
  }
}

template <int MT, int NT, bool PRO, int STATS>
static int launch_conv2_k(const Conv2Args& a, dim3 grid, int lds, hipStream_t s) {
  auto k = k_conv2_mfma<MT, NT, PRO, STATS>;
  static bool attr_done = false;
  if (!attr_done) { if (check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds))) return DEVIS_ERR_HIP; attr_done = true; }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  return launch_status();
}

template <int MT, int NT>
static int launch_conv2(Conv2Args& a, const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  constexpr int TH = 4 * NT;
  constexpr int HV = (TH + 2) * 18;
  constexpr int LDS = ((HV * 4 + 255) / 256) * 256 * 16 + 9 * MT * 16 * 64;
  a.nth = (int)cdiv(a.H, TH); a.ntw = (int)cdiv(a.W, 16);
  const int64_t tiles = (int64_t)a.N * a.nth * a.ntw;
  if (tiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const int mode = st ? st->mode : 0;
  if (plan_chunks) {
    *plan_chunks = (mode == 2 && (a.sc || a.Co2)) ? 0 : (int64_t)a.nth * a.ntw;
    return DEVIS_OK;
  }
  const dim3 grid((unsigned)tiles, (unsigned)(a.Cout / (MT * 16)));
  a.st = StatsArgs{};
  if (mode) {
    a.st = *st;
    if (a.st.chunks != a.nth * a.ntw || a.st.C != a.Cout) return DEVIS_ERR_BAD_ARG;
    if (mode == 2 && (a.sc || a.Co2)) return DEVIS_ERR_UNSUPPORTED;
  }
  if (mode == 1) return a.sc ? launch_conv2_k<MT, NT, true, 1>(a, grid, LDS, s) : launch_conv2_k<MT, NT, false, 1>(a, grid, LDS, s);
  if (mode == 2) return launch_conv2_k<MT, NT, false, 2>(a, grid, LDS, s);
  return a.sc ? launch_conv2_k<MT, NT, true, 0>(a, grid, LDS, s) : launch_conv2_k<MT, NT, false, 0>(a, grid, LDS, s);
}

static bool is_2d_3x3(const devis_conv_desc* d) {
  return d->KD == 1 && d->KH == 3 && d->KW == 3 && d->SD == 1 && d->SH == 1 && d->SW == 1 && d->PD == 0 && d->PH == 1 &&
         d->PW == 1 && d->ID == 1;
}

static int conv2_route(Conv2Args& a, const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  const int64_t big = (int64_t)a.N * a.H * a.W * (a.C1 > a.Cout ? a.C1 : a.Cout);
  if (big >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  if (a.Cin % 32 || a.C1 % 32 || a.Cout % 32 || a.Co1 % 16 || (int64_t)a.Cout * 9 * a.Cin >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  // tall tiles while they still fill the chip; 64 cout per block where the layer has them
  const int64_t tiles16 = (int64_t)a.N * cdiv(a.H, 16) * cdiv(a.W, 16);
  if (a.Cout % 64 == 0) {
    if (tiles16 * (a.Cout / 64) >= 512) return launch_conv2<4, 4>(a, st, plan_chunks, s);
    return launch_conv2<4, 2>(a, st, plan_chunks, s);
  }
  if (tiles16 * (a.Cout / 32) >= 512) return launch_conv2<2, 4>(a, st, plan_chunks, s);
  return launch_conv2<2, 2>(a, st, plan_chunks, s);
}

// stride-1 3x3 (D = 1), bf16, every channel count a multiple of 32; x2 = second half of a virtual concat
int conv2d_fwd_mfma(const devis_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                    const float* sc, const float* sh, const void* res, void* y, const StatsArgs* st,
                    int64_t* plan_chunks, hipStream_t s) {
  if (plan_chunks) *plan_chunks = 0;
  if (d->dtype != DEVIS_BF16 || !is_2d_3x3(d)) return DEVIS_ERR_UNSUPPORTED;
  Conv2Args a;
  a.x1 = (const bf16*)x1; a.x2 = (const bf16*)x2; a.w = (const bf16*)w; a.bias = bias; a.sc = sc; a.sh = sh;
  a.res = (const bf16*)res; a.y1 = (bf16*)y; a.y2 = nullptr;
  a.N = (int)d->N; a.H = (int)d->IH; a.W = (int)d->IW; a.C1 = (int)d->C1; a.C2 = (int)d->C2; a.Cin = a.C1 + a.C2;
  a.Cout = (int)d->Cout; a.Co1 = a.Cout; a.Co2 = 0;
  if (a.C2 % 32) return DEVIS_ERR_UNSUPPORTED;
  return conv2_route(a, st, plan_chunks, s);
}

// dgrad = the same conv on dy with the tap-flipped [C1+C2][9][Cout] weights; the gradient of a two-input layer
// lands in dx1 (C1 channels) and dx2 (C2 channels)
int conv2d_dgrad_mfma(const devis_conv_desc* d, const void* dy, const void* w_dgrad, const void* res, void* dx1, void* dx2,
                      const StatsArgs* st, int64_t* plan_chunks, hipStream_t s) {
  if (plan_chunks) *plan_chunks = 0;
  if (!w_dgrad || d->dtype != DEVIS_BF16 || !is_2d_3x3(d)) return DEVIS_ERR_UNSUPPORTED;
  if (res && d->C2) return DEVIS_ERR_UNSUPPORTED;
  Conv2Args a;
  a.x1 = (const bf16*)dy; a.x2 = nullptr; a.w = (const bf16*)w_dgrad; a.bias = nullptr; a.sc = nullptr; a.sh = nullptr;
  a.res = (const bf16*)res; a.y1 = (bf16*)dx1; a.y2 = (bf16*)dx2;
  a.N = (int)d->N; a.H = (int)d->IH; a.W = (int)d->IW; a.C1 = (int)d->Cout; a.C2 = 0; a.Cin = a.C1;
  a.Cout = (int)(d->C1 + d->C2); a.Co1 = (int)d->C1; a.Co2 = (int)d->C2;
  return conv2_route(a, st, plan_chunks, s);
}

// ------------------------------------------------------------------------------------------------ wgrad
struct Wg2Args {
  const bf16* x1;    // [N][H][W][C1]
  const bf16* x2;    // [N][H][W][C2] or null
  const bf16* dy;    // [N][H][W][Cout]
  const float* sc;   // [N][Cin] or null
  const float* sh;
  float* part;       // [gx][9][Cout][Cin]
  float* bpart;      // [gx][Cout] or null
  int N, H, W, C1, C2, Cin, Cout;
  int nth, ntw, ntiles;
};

// 32-byte half h of pixel u lives at ((h ^ ((u >> 2) & 1)) * 32) of its 64-byte row (conv_wgrad_mfma.hip)
__device__ __forceinline__ int img2_off(int u, int half) { return u * 64 + ((half ^ ((u >> 2) & 1)) << 5); }

__device__ __forceinline__ bf16x8 tr_pair3(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = a;
  u.h[1] = b;
  return u.v;
}

template <bool PRO>
__global__ void __launch_bounds__(256)
k_wgrad2_mfma(Wg2Args a) {
  constexpr int TH = 16, HH = TH + 2, HW = 18, HV = HH * HW, TV = TH * 16;
  constexpr int XPT = (HV * 4 + 255) / 256;      // 6
  constexpr int YPT = TV * 4 / 256;              // 4
  constexpr int X_BYTES = XPT * 256 * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* ys = smem + X_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int nci = a.Cin >> 5;
  const int ci0 = (blockIdx.y % nci) * 32, co0 = (blockIdx.y / nci) * 32;
  const bool second = ci0 >= a.C1;
  const bf16* xb = second ? a.x2 + (ci0 - a.C1) : a.x1 + ci0;
  const unsigned cs = second ? a.C2 : a.C1;
  const int piece = tid & 3;

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[t][mo][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  const bool do_bias = a.bpart != nullptr && ci0 == 0;

  // per-thread LDS slots (tile independent)
  unsigned xl[XPT], yl[YPT];
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int u = (tid + k * 256) >> 2;
    xl[k] = (unsigned)(img2_off(u, piece >> 1) + (piece & 1) * 16);
  }
#pragma unroll
  for (int k = 0; k < YPT; ++k) {
    const int v = (tid + k * 256) >> 2;
    yl[k] = (unsigned)(img2_off(v, piece >> 1) + (piece & 1) * 16);
  }

  u32x4 xr[XPT], yr[YPT];
  unsigned xokm = 0u, yokm = 0u;
  int ncur = 0;
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int twi = tile % a.ntw;
    const int thi = (tile / a.ntw) % a.nth;
    const int n = tile / (a.ntw * a.nth);
    const int oh0 = thi * TH, ow0 = twi * 16;
    ncur = n;
    xokm = 0u; yokm = 0u;
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int i = tid + k * 256;
      const int u = i >> 2;
      const int hw = u % HW, hh = u / HW;
      const int ih = oh0 - 1 + hh, iw = ow0 - 1 + hw;
      const bool ok = i < HV * 4 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
      const unsigned pix = ok ? (unsigned)((n * a.H + ih) * a.W + iw) : 0u;
      xr[k] = *reinterpret_cast<const u32x4*>(xb + pix * cs + piece * 8);
      xokm |= (ok ? 1u : 0u) << k;
    }
#pragma unroll
    for (int k = 0; k < YPT; ++k) {
      const int v = (tid + k * 256) >> 2;
      const int oh = oh0 + (v >> 4), ow = ow0 + (v & 15);
      const bool ok = oh < a.H && ow < a.W;
      const unsigned pix = ok ? (unsigned)((n * a.H + oh) * a.W + ow) : 0u;
      yr[k] = *reinterpret_cast<const u32x4*>(a.dy + pix * (unsigned)a.Cout + co0 + piece * 8);
      yokm |= (ok ? 1u : 0u) << k;
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
    float scv[8], shv[8];
    if (PRO) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        scv[j] = a.sc[ncur * a.Cin + ci0 + piece * 8 + j];
        shv[j] = a.sh[ncur * a.Cin + ci0 + piece * 8 + j];
      }
    }
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      u32x4 v = xr[k];
      if (PRO) v = affine_relu8(v, scv, shv);
      if (!((xokm >> k) & 1u)) v = (u32x4){0u, 0u, 0u, 0u};
      if (k < XPT - 1 || tid + k * 256 < HV * 4) *reinterpret_cast<u32x4*>(xs + xl[k]) = v;
    }
#pragma unroll
    for (int k = 0; k < YPT; ++k) {
      u32x4 v = yr[k];
      if (!((yokm >> k) & 1u)) v = (u32x4){0u, 0u, 0u, 0u};
      if (do_bias) {
        bsum[0] += bf_lo(v[0]); bsum[1] += bf_hi(v[0]); bsum[2] += bf_lo(v[1]); bsum[3] += bf_hi(v[1]);
        bsum[4] += bf_lo(v[2]); bsum[5] += bf_hi(v[2]); bsum[6] += bf_lo(v[3]); bsum[7] += bf_hi(v[3]);
      }
      *reinterpret_cast<u32x4*>(ys + yl[k]) = v;
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) fetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();          // previous tile fully consumed
    stage();
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    // ---- this wave's K steps: row pairs wave and wave + 4
#pragma unroll 1
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = (wave + 4 * ks) * 2;
      const int wv = 4 * g + q;               // this lane addresses pixel column wv of rows r0, r0 + 1
      bf16x8 af[2];
      {
        const int v0 = r0 * 16 + wv, v1 = v0 + 16;
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
          af[mo] = tr_pair3(ys + img2_off(v0, mo) + p * 8, ys + img2_off(v1, mo) + p * 8);
      }
      const int ub = r0 * HW + wv;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int u0 = ub + (t / 3) * HW + t % 3, u1 = u0 + HW;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const bf16x8 bfr = tr_pair3(xs + img2_off(u0, mi) + p * 8, xs + img2_off(u1, mi) + p * 8);
#pragma unroll
          for (int mo = 0; mo < 2; ++mo)
            acc[t][mo][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mo], bfr, acc[t][mo][mi], 0, 0, 0);
        }
      }
    }
  }

  // ---- the four waves' accumulators are added in wave order through LDS (36 KB per wave)
  float* red = reinterpret_cast<float*>(smem);
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            *reinterpret_cast<f32x4*>(red + (((t * 2 + mo) * 2 + mi) * 64 + lane) * 4) = acc[t][mo][mi];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(red + (((t * 2 + mo) * 2 + mi) * 64 + lane) * 4);
            acc[t][mo][mi][0] += v[0]; acc[t][mo][mi][1] += v[1]; acc[t][mo][mi][2] += v[2]; acc[t][mo][mi][3] += v[3];
          }
    }
  }
  // D layout: col = ci = lane & 15, rows co = (lane >> 4) * 4 + j
  if (wave == 0) {
    float* part = a.part + (size_t)blockIdx.x * a.Cout * a.Cin * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int ci = ci0 + mi * 16 + (lane & 15);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = co0 + mo * 16 + g * 4 + j;
            part[((size_t)t * a.Cout + co) * a.Cin + ci] = acc[t][mo][mi][j];
          }
        }
  }
  if (do_bias) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < 32) {              // channel = 8 * piece + j; threads with equal (tid & 3) hold the same 8 channels
      const int pc = tid / 8, j = tid % 8;
      float s = 0.f;
      for (int k = pc; k < 256; k += 4) s += red[k * 8 + j];
      a.bpart[(size_t)blockIdx.x * a.Cout + co0 + tid] = s;
    }
  }
}

static int wgrad2_dispatch(const devis_conv_desc* d, const void* x1, const void* x2, const float* sc, const float* sh,
                           const void* dy, float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s, bool query,
                           size_t* need) {
  if (d->dtype != DEVIS_BF16 || !is_2d_3x3(d)) return DEVIS_ERR_UNSUPPORTED;
  const int C1 = (int)d->C1, C2 = (int)d->C2, Cin = C1 + C2, Cout = (int)d->Cout;
  if (C1 % 32 || C2 % 32 || Cout % 32) return DEVIS_ERR_UNSUPPORTED;
  if (d->N * d->IH * d->IW * (int64_t)(C1 > Cout ? C1 : Cout) >= (int64_t)1 << 31) return DEVIS_ERR_UNSUPPORTED;
  Wg2Args a;
  a.nth = (int)cdiv(d->IH, 16); a.ntw = (int)cdiv(d->IW, 16);
  const int64_t ntiles = d->N * a.nth * a.ntw;
  if (ntiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const int nslab = (Cin / 32) * (Cout / 32);
  // blocks = gx * nslab: about two per CU; every block leaves one fp32 partial of its slab
  int64_t gx = 512 / nslab;
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  const size_t part_bytes = (size_t)gx * Cout * Cin * 9 * sizeof(float);
  const size_t total = part_bytes + (size_t)gx * Cout * sizeof(float);
  if (query) { *need = total; return DEVIS_OK; }
  if (!ws || ws_bytes < total) return DEVIS_ERR_WORKSPACE;
  a.x1 = (const bf16*)x1; a.x2 = (const bf16*)x2; a.dy = (const bf16*)dy; a.sc = sc; a.sh = sh;
  a.part = (float*)ws; a.bpart = dbias ? (float*)((char*)ws + part_bytes) : nullptr;
  a.N = (int)d->N; a.H = (int)d->IH; a.W = (int)d->IW; a.C1 = C1; a.C2 = C2; a.Cin = Cin; a.Cout = Cout;
  a.ntiles = (int)ntiles;
  constexpr int LDS = 6 * 256 * 16 + 256 * 64;
  const dim3 grid((unsigned)gx, (unsigned)nslab);
  static bool attr_done = false;
  if (!attr_done) {
    if (check_hip(hipFuncSetAttribute((const void*)k_wgrad2_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS))) return DEVIS_ERR_HIP;
    if (check_hip(hipFuncSetAttribute((const void*)k_wgrad2_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS))) return DEVIS_ERR_HIP;
    attr_done = true;
  }
  if (sc) hipLaunchKernelGGL(k_wgrad2_mfma<true>, grid, dim3(256), LDS, s, a);
  else hipLaunchKernelGGL(k_wgrad2_mfma<false>, grid, dim3(256), LDS, s, a);
  int rc = launch_status();
  if (rc) return rc;
  RedArgs r;
  r.part = a.part; r.bpart = a.bpart; r.dw = dw; r.dbias = dbias;
  r.nparts = (int)gx; r.nbparts = (int)gx; r.Cb = Cout;
  r.pstride = (int64_t)9 * Cout * Cin; r.total = r.pstride;
  r.mode = RED_CONV3; r.A = Cout; r.B = Cin; r.cin_real = Cin;
  return launch_reduce_partials(r, s);
}

size_t conv2d_wgrad_mfma_workspace(const devis_conv_desc* d) {
  size_t need = 0;
  if (wgrad2_dispatch(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true, &need) != DEVIS_OK) return 0;
  return need;
}

int conv2d_wgrad_mfma(const devis_conv_desc* d, const void* x1, const void* x2, const float* sc, const float* sh,
                      const void* dy, float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s) {
  size_t need = 0;
  return wgrad2_dispatch(d, x1, x2, sc, sh, dy, dw, dbias, ws, ws_bytes, s, false, &need);
}

}  // namespace devis
