// fp32 convolution on the matrix cores (v_mfma_f32_16x16x4_f32), channels-last, gfx950: the parity mode
// (compute_dtype = float32, the reference's own precision) at a usable speed.  The generic direct kernels of
// conv_direct.hip - one thread per output element, no reuse - ran the BraTS step at 1.9 s; they remain the route for
// the shapes refused here.
//
//   y[v][co] = sum_{tap, ci} W[co][tap][ci] * A[v * S + tap - P][ci] (+ bias, + residual),   A = relu(sc * x + sh)
//
// One kernel for every kernel size (taps are a runtime loop), stride 1 or 2 per dimension, one or two input tensors
// (virtual concat, channel counts multiples of 16; or ONE tensor of 4 channels, zero-extended - the first layer).
// GEMM orientation as in conv_mfma.hip: M = 16 cout, N = 16 output voxels along W, K = 4 channels per MFMA; the
// K index of an MFMA is spread over the lane quads so that a lane's four K steps of a 16-channel chunk are 16
// contiguous bytes: lane (l15, lq) holds channels 4 lq .. 4 lq + 3 of voxel l15 (B, one ds_read_b128) and of weight
// row l15 (A, one 16-byte global load).  A block stages the halo tile of one 16-channel chunk in LDS (prologue applied,
// zero padding after it), runs all taps over it, and moves to the next chunk.
//
// The same kernel is the data gradient: it runs on the tap-flipped weights [Cin][T'][Cout] with pad K - 1 - P.  For a
// strided layer the input voxels fall into S^3 parity classes, each of which sees only the taps of its own parity: one
// launch per class, a dense stride-1 convolution of the gradient in memory with the class's taps (`T0` / `TS`: first
// tap and tap step per dimension inside the full kernel) whose outputs are scattered to the class's voxels (`YS*` /
// `YO*`) - exactly the useful FLOPs, no second kernel.
#include "common.h"
#include "mfma_common.h"
#include "wgrad_reduce.h"

namespace devis {

struct F32ConvArgs {
  const float* x1;
  const float* x2;     // second input of a virtual concat, or null
  const float* w;      // [rows >= Cout][T][Cin] fp32
  const float* bias;   // [Cout] or null
  const float* sc;     // [N][Cin] or null
  const float* sh;
  const float* res;    // like y, or null
  float* y;
  int N, MD, MH, MW;   // spatial size of x1 / x2
  int OD, OH, OW;      // output
  int C1, C2, Cout;    // C1 == 4: one 4-channel input, zero-extended to a 16-channel chunk
  int KD, KH, KW, SD, SH, SW, PD, PH, PW;   // K*: taps VISITED per dimension (may be 0: the output is bias + residual)
  int WT, WKH, WKW;    // the weight rows hold WT taps of a (. , WKH, WKW) kernel, of which tap (T0D + a TSD, ...) is visited
  int T0D, T0H, T0W, TSD, TSH, TSW;
  int HD, HH, HW;      // halo tile
  int ntd, nth, ntw;
  // where output voxel (od, oh, ow) lives in y (and res): voxel (od * YSD + YOD, ...) of a (YD, YH, YW) grid.
  // Identity for a convolution; a transposed convolution with kernel == stride is one 1x1 conv per tap, scattered.
  int YD, YH, YW, YSD, YSH, YSW, YOD, YOH, YOW;
};

template <int MT, int TD, int TH, bool PRO>
__global__ void __launch_bounds__(256)
k_conv_f32_mfma(F32ConvArgs a) {
  constexpr int TW = 16;
  constexpr int NT = TD * TH / 4;                 // 16-voxel N tiles per wave
  static_assert(TD * TH % 4 == 0, "tiles split over four waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;

  const int tile = blockIdx.x;
  const int twi = tile % a.ntw;
  const int thi = (tile / a.ntw) % a.nth;
  const int tdi = (tile / (a.ntw * a.nth)) % a.ntd;
  const int n = tile / (a.ntw * a.nth * a.ntd);
  const int od0 = tdi * TD, oh0 = thi * TH, ow0 = twi * TW;
  const int co_base = blockIdx.y * (MT * 16);
  const bool cin4 = a.C1 == 4;
  const int Cin = cin4 ? 4 : a.C1 + a.C2;         // row length of the weights, channels of sc / sh
  const int T = a.WT;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int ubase[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int g = wave * NT + j;
    const int lth = g % TH, ltd = g / TH;
    ubase[j] = ((ltd * a.SD) * a.HH + lth * a.SH) * a.HW + l15 * a.SW;
  }
  // weight rows of this lane (rows >= Cout do not exist: their fragment is zero)
  const float* wrow[MT];
  bool wok[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int co = co_base + m * 16 + l15;
    wok[m] = co < a.Cout;
    wrow[m] = a.w + (size_t)(wok[m] ? co : 0) * T * Cin;
  }

  const int HV = a.HD * a.HH * a.HW;
  const int nchunks = cin4 ? 1 : (a.C1 + a.C2) / 16;
  const int jd0 = od0 * a.SD - a.PD, jh0 = oh0 * a.SH - a.PH, jw0 = ow0 * a.SW - a.PW;
  for (int cc = 0; cc < nchunks; ++cc) {
    if (cc > 0) __syncthreads();                  // every wave is done with the previous chunk's tile
    // ---- stage the halo tile of channels [16 cc, 16 cc + 16): 4 pieces of 16 bytes per voxel
    const int c0 = cc * 16;
    const bool second = !cin4 && c0 >= a.C1;
    const float* xb = second ? a.x2 : a.x1;
    const int cx = second ? a.C2 : a.C1, cofs = second ? c0 - a.C1 : c0;
    // four slots per trip, their loads issued together (a slot outside the tensor loads element 0 and is zeroed by a
    // select: no memory instruction under a divergent branch, and one HBM round trip per trip instead of four)
    const int piece = tid & 3;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), h4 = s4;
    if (PRO) {
      s4 = *reinterpret_cast<const float4*>(a.sc + (size_t)n * Cin + c0 + piece * 4);
      h4 = *reinterpret_cast<const float4*>(a.sh + (size_t)n * Cin + c0 + piece * 4);
    }
    for (int i0 = tid; i0 < HV * 4; i0 += 1024) {
      float4 v[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int u = (i0 + k * 256) >> 2;
        const int hw = u % a.HW, r = u / a.HW, hh = r % a.HH, hd = r / a.HH;
        const int id = jd0 + hd, ih = jh0 + hh, iw = jw0 + hw;
        ok[k] = u < HV && id >= 0 && ih >= 0 && iw >= 0 && id < a.MD && ih < a.MH && iw < a.MW && !(cin4 && piece);
        const size_t vox = ok[k] ? (size_t)((n * a.MD + id) * a.MH + ih) * a.MW + iw : 0;
        v[k] = *reinterpret_cast<const float4*>(xb + (ok[k] ? vox * cx + cofs + piece * 4 : 0));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int u = (i0 + k * 256) >> 2;
        float4 o = v[k];
        if (PRO) {
          o.x = fmaxf(fmaf(o.x, s4.x, h4.x), 0.f); o.y = fmaxf(fmaf(o.y, s4.y, h4.y), 0.f);
          o.z = fmaxf(fmaf(o.z, s4.z, h4.z), 0.f); o.w = fmaxf(fmaf(o.w, s4.w, h4.w), 0.f);
        }
        if (!ok[k]) o = make_float4(0.f, 0.f, 0.f, 0.f);
        // XOR swizzle of the piece slot: 16 consecutive voxels x one piece land in 16 different 16-byte bank groups
        if (u < HV) *reinterpret_cast<float4*>(smem + u * 64 + ((piece ^ ((u >> 2) & 3)) << 4)) = o;
      }
    }
    __syncthreads();

    // ---- all taps over this chunk
    for (int kd = 0; kd < a.KD; ++kd)
      for (int kh = 0; kh < a.KH; ++kh)
        for (int kw = 0; kw < a.KW; ++kw) {
          const int toff = (kd * a.HH + kh) * a.HW + kw;
          const int t = ((a.T0D + kd * a.TSD) * a.WKH + a.T0H + kh * a.TSH) * a.WKW + a.T0W + kw * a.TSW;
          f32x4 af[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            af[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (wok[m] && !(cin4 && lq)) af[m] = *reinterpret_cast<const f32x4*>(wrow[m] + (size_t)t * Cin + c0 + lq * 4);
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int u = ubase[j] + toff;
            const f32x4 bf = *reinterpret_cast<const f32x4*>(smem + u * 64 + ((lq ^ ((u >> 2) & 3)) << 4));
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int k = 0; k < 4; ++k)
                acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][k], bf[k], acc[m][j], 0, 0, 0);
          }
        }
  }

  // ---- epilogue: a lane holds 4 consecutive cout (4 lq ..) of voxel l15 of each of its tiles
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int g = wave * NT + j;
    const int lth = g % TH, ltd = g / TH;
    const int od = od0 + ltd, oh = oh0 + lth, ow = ow0 + l15;
    if (od >= a.OD || oh >= a.OH || ow >= a.OW) continue;
    const size_t vox = (size_t)((n * a.YD + od * a.YSD + a.YOD) * a.YH + oh * a.YSH + a.YOH) * a.YW + ow * a.YSW + a.YOW;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co = co_base + m * 16 + lq * 4;
      if (co >= a.Cout) continue;                 // Cout is a multiple of 4: a group of four exists or does not
      float4 o = make_float4(acc[m][j][0], acc[m][j][1], acc[m][j][2], acc[m][j][3]);
      if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
      }
      if (a.res) {
        const float4 r = *reinterpret_cast<const float4*>(a.res + vox * a.Cout + co);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      *reinterpret_cast<float4*>(a.y + vox * a.Cout + co) = o;
    }
  }
}

template <int MT, int TD, int TH>
static int launch_f32(F32ConvArgs& a, hipStream_t s) {
  a.HD = (TD - 1) * a.SD + a.KD; a.HH = (TH - 1) * a.SH + a.KH; a.HW = 15 * a.SW + a.KW;
  a.ntd = (int)cdiv(a.OD, TD); a.nth = (int)cdiv(a.OH, TH); a.ntw = (int)cdiv(a.OW, 16);
  const int64_t tiles = (int64_t)a.N * a.ntd * a.nth * a.ntw;
  const size_t lds = (size_t)a.HD * a.HH * a.HW * 64;
  if (tiles > 0x7fffffff || lds > 160 * 1024) return DEVIS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)tiles, (unsigned)cdiv(a.Cout, MT * 16));
  if (grid.y > 65535) return DEVIS_ERR_UNSUPPORTED;
  auto launch = [&](auto k) -> int {
    if (lds > 65536) {
      // raised once per kernel and process; the largest tile any caller can ask for is bounded by the check above
      static thread_local const void* raised[8] = {};
      bool have = false;
      for (const void* p : raised) have = have || p == (const void*)k;
      if (!have) {
        if (check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))) return DEVIS_ERR_HIP;
        for (const void*& p : raised) if (!p) { p = (const void*)k; break; }
      }
    }
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
    return launch_status();
  };
  if (a.sc) return launch(k_conv_f32_mfma<MT, TD, TH, true>);
  return launch(k_conv_f32_mfma<MT, TD, TH, false>);
}

// every tap of the kernel, outputs where a convolution puts them
static void dense_output(F32ConvArgs& a) {
  a.YD = a.OD; a.YH = a.OH; a.YW = a.OW; a.YSD = a.YSH = a.YSW = 1; a.YOD = a.YOH = a.YOW = 0;
}
static void all_taps(F32ConvArgs& a) {
  a.WT = a.KD * a.KH * a.KW; a.WKH = a.KH; a.WKW = a.KW;
  a.T0D = a.T0H = a.T0W = 0; a.TSD = a.TSH = a.TSW = 1;
}

static int run_f32(F32ConvArgs& a, hipStream_t s) {
  // voxel indices are 32-bit in the kernel (element offsets are not)
  if ((int64_t)a.N * a.MD * a.MH * a.MW >= ((int64_t)1 << 31) || (int64_t)a.N * a.OD * a.OH * a.OW >= ((int64_t)1 << 31) ||
      (int64_t)a.N * a.YD * a.YH * a.YW >= ((int64_t)1 << 31))
    return DEVIS_ERR_UNSUPPORTED;
  const bool strided = a.SD > 1 || a.SH > 1 || a.SW > 1;
  if (a.OD == 1) {                                     // the 2-D nets: a second plane in the tile would be idle
    if (a.Cout >= 32) return launch_f32<2, 1, 8>(a, s);
    if (strided) return launch_f32<1, 1, 8>(a, s);
    return launch_f32<1, 1, 16>(a, s);
  }
  if (a.Cout >= 32) return launch_f32<2, 2, 4>(a, s);
  if (strided) return launch_f32<1, 2, 4>(a, s);       // an 8-row tile of a stride-2 layer does not fit the LDS
  return launch_f32<1, 2, 8>(a, s);
}

static bool f32_shape_ok(const devis_conv_desc* d) {
  if (d->dtype != DEVIS_F32) return false;
  if (d->SD < 1 || d->SD > 2 || d->SH < 1 || d->SH > 2 || d->SW < 1 || d->SW > 2) return false;
  if (d->KD > 7 || d->KH > 7 || d->KW > 7) return false;
  return true;
}

// forward.  x2 / C2 > 0: virtual concat.
int conv_fwd_f32_mfma(const devis_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                      const float* sc, const float* sh, const void* res, void* y, hipStream_t s) {
  if (!f32_shape_ok(d) || d->Cout % 4) return DEVIS_ERR_UNSUPPORTED;
  const bool first = d->C1 == 4 && d->C2 == 0;
  if (!first && (d->C1 % 16 || d->C2 % 16)) return DEVIS_ERR_UNSUPPORTED;
  F32ConvArgs a;
  a.x1 = (const float*)x1; a.x2 = (const float*)x2; a.w = (const float*)w; a.bias = bias; a.sc = sc; a.sh = sh;
  a.res = (const float*)res; a.y = (float*)y;
  a.N = (int)d->N; a.MD = (int)d->ID; a.MH = (int)d->IH; a.MW = (int)d->IW;
  a.OD = (int)d->OD; a.OH = (int)d->OH; a.OW = (int)d->OW;
  a.C1 = (int)d->C1; a.C2 = (int)d->C2; a.Cout = (int)d->Cout;
  a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.SD = d->SD; a.SH = d->SH; a.SW = d->SW; a.PD = d->PD; a.PH = d->PH; a.PW = d->PW;
  dense_output(a);
  all_taps(a);
  return run_f32(a, s);
}

// data gradient of the layer `d` describes: w_flip = [C1 + C2][T flipped][Cout] (devis_repack_conv_weight_dgrad).
// dx1 (and dx2 for a two-input layer) are written by one launch each.
int conv_dgrad_f32_mfma(const devis_conv_desc* d, const void* dy, const void* w_flip, const void* res, void* dx1, void* dx2,
                        hipStream_t s) {
  if (!f32_shape_ok(d) || !w_flip) return DEVIS_ERR_UNSUPPORTED;
  if ((d->Cout % 16 && d->Cout != 4) || d->C1 % 4 || d->C2 % 4 || (res && d->C2)) return DEVIS_ERR_UNSUPPORTED;
  if (d->PD > d->KD - 1 || d->PH > d->KH - 1 || d->PW > d->KW - 1) return DEVIS_ERR_UNSUPPORTED;
  const int T = d->KD * d->KH * d->KW;
  const int K[3] = {d->KD, d->KH, d->KW}, S[3] = {d->SD, d->SH, d->SW}, I[3] = {(int)d->ID, (int)d->IH, (int)d->IW};
  const int Pf[3] = {d->KD - 1 - d->PD, d->KH - 1 - d->PH, d->KW - 1 - d->PW};       // pad of the flipped convolution
  for (int part = 0; part < (d->C2 ? 2 : 1); ++part)
    for (int cls = 0; cls < S[0] * S[1] * S[2]; ++cls) {
      // input voxels i = S k + p of this class, per dimension: dx[i] = sum_{t' = t0 + S a} dy[k + off + a] Wf[t'],
      // t0 = (p + P') mod S, off = (p - P' + t0) / S (exact), a = 0 .. ceil((K - t0) / S) - 1
      const int p[3] = {cls / (S[1] * S[2]), (cls / S[2]) % S[1], cls % S[2]};
      int t0[3], na[3], off[3], cnt[3];
      bool empty = false;
      for (int k = 0; k < 3; ++k) {
        t0[k] = (p[k] + Pf[k]) % S[k];
        na[k] = t0[k] < K[k] ? (K[k] - t0[k] + S[k] - 1) / S[k] : 0;
        off[k] = (p[k] - Pf[k] + t0[k]) / S[k];            // the numerator is a multiple of S, of either sign
        cnt[k] = I[k] > p[k] ? (I[k] - p[k] + S[k] - 1) / S[k] : 0;
        empty = empty || cnt[k] == 0;
      }
      if (empty) continue;
      F32ConvArgs a;
      a.x1 = (const float*)dy; a.x2 = nullptr; a.bias = nullptr; a.sc = nullptr; a.sh = nullptr;
      a.w = (const float*)w_flip + (part ? (size_t)d->C1 * T * d->Cout : 0);
      a.res = part ? nullptr : (const float*)res; a.y = (float*)(part ? dx2 : dx1);
      a.N = (int)d->N; a.MD = (int)d->OD; a.MH = (int)d->OH; a.MW = (int)d->OW;
      a.OD = cnt[0]; a.OH = cnt[1]; a.OW = cnt[2];
      a.C1 = (int)d->Cout; a.C2 = 0; a.Cout = (int)(part ? d->C2 : d->C1);
      a.KD = na[0]; a.KH = na[1]; a.KW = na[2]; a.SD = a.SH = a.SW = 1;
      a.PD = -off[0]; a.PH = -off[1]; a.PW = -off[2];
      a.WT = T; a.WKH = d->KH; a.WKW = d->KW;
      a.T0D = t0[0]; a.T0H = t0[1]; a.T0W = t0[2]; a.TSD = S[0]; a.TSH = S[1]; a.TSW = S[2];
      a.YD = I[0]; a.YH = I[1]; a.YW = I[2]; a.YSD = S[0]; a.YSH = S[1]; a.YSW = S[2]; a.YOD = p[0]; a.YOH = p[1]; a.YOW = p[2];
      const int rc = run_f32(a, s);
      if (rc) return rc;
    }
  return DEVIS_OK;
}

// ---- transposed convolution with kernel == stride (each K 1 or 2), fp32
// forward: y[K i + t][co] = sum_ci W[t][co][ci] x[i][ci] + b: one 1x1 conv per tap on the packed weight's slice
// [Cout][Cin] of tap t, its output scattered to the fine voxels of that tap.
int convT_fwd_f32_mfma(const devis_convT_desc* d, const void* x, const void* w, const float* bias, void* y, hipStream_t s) {
  if (d->dtype != DEVIS_F32 || d->Cin % 16 || d->Cout % 4) return DEVIS_ERR_UNSUPPORTED;
  for (int kd = 0; kd < d->KD; ++kd)
    for (int kh = 0; kh < d->KH; ++kh)
      for (int kw = 0; kw < d->KW; ++kw) {
        const int t = (kd * d->KH + kh) * d->KW + kw;
        F32ConvArgs a;
        a.x1 = (const float*)x; a.x2 = nullptr; a.w = (const float*)w + (size_t)t * d->Cout * d->Cin; a.bias = bias;
        a.sc = nullptr; a.sh = nullptr; a.res = nullptr; a.y = (float*)y;
        a.N = (int)d->N; a.MD = (int)d->ID; a.MH = (int)d->IH; a.MW = (int)d->IW;
        a.OD = (int)d->ID; a.OH = (int)d->IH; a.OW = (int)d->IW;
        a.C1 = (int)d->Cin; a.C2 = 0; a.Cout = (int)d->Cout;
        a.KD = a.KH = a.KW = 1; a.SD = a.SH = a.SW = 1; a.PD = a.PH = a.PW = 0;
        all_taps(a);
        a.YD = (int)d->ID * d->KD; a.YH = (int)d->IH * d->KH; a.YW = (int)d->IW * d->KW;
        a.YSD = d->KD; a.YSH = d->KH; a.YSW = d->KW; a.YOD = kd; a.YOH = kh; a.YOW = kw;
        const int rc = run_f32(a, s);
        if (rc) return rc;
      }
  return DEVIS_OK;
}

// data gradient: dx[i][ci] = sum_{t, co} dy[K i + t][co] W[ci][co][t] - a convolution of dy with kernel = stride = K,
// no padding, on w_t = [Cin][T][Cout] (devis_repack_conv_weight_dgrad of the packed weight viewed as [T Cout][1][Cin])
int convT_dgrad_f32_mfma(const devis_convT_desc* d, const void* dy, const void* w_t, void* dx, hipStream_t s) {
  if (d->dtype != DEVIS_F32 || !w_t || d->Cout % 16 || d->Cin % 4) return DEVIS_ERR_UNSUPPORTED;
  F32ConvArgs a;
  a.x1 = (const float*)dy; a.x2 = nullptr; a.w = (const float*)w_t; a.bias = nullptr; a.sc = nullptr; a.sh = nullptr;
  a.res = nullptr; a.y = (float*)dx;
  a.N = (int)d->N; a.MD = (int)d->ID * d->KD; a.MH = (int)d->IH * d->KH; a.MW = (int)d->IW * d->KW;
  a.OD = (int)d->ID; a.OH = (int)d->IH; a.OW = (int)d->IW;
  a.C1 = (int)d->Cout; a.C2 = 0; a.Cout = (int)d->Cin;
  a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.SD = d->KD; a.SH = d->KH; a.SW = d->KW; a.PD = a.PH = a.PW = 0;
  dense_output(a);
  all_taps(a);
  return run_f32(a, s);
}

// ------------------------------------------------------------------------------------------------
// fp32 weight gradient on the matrix cores, deterministic (no atomics):
//
//   dW[co][tap][ci] = sum_v dy[v][co] * A[v * S + tap - P][ci],   A = relu(sc * x + sh)
//
// GEMM: M = 16 cout, N = 16 cin, K = voxels, 4 per MFMA (lane quad lq <-> voxel 4 q + lq of a row of 16).  A block owns
// NO x NI slabs of 16 x 16 (cout x cin) and marches a (TD x TH x 16)-voxel tile along D; per step it stages the x halo
// tile (the slab's cin, prologue applied) and the dy tile (the slab's cout) in LDS.  A wave owns ONE slab and keeps the
// accumulators of all taps (27 x 4 VGPRs); with fewer than four slabs per block the waves of a slab split the K steps
// and leave a partial each.  One dy fragment feeds every tap: T + 1 LDS reads (ds_read_b32) per T MFMAs.  Partials
// [part][T][Cout][Cin16] are summed in a fixed order by k_reduce_partials (wgrad_reduce.h), which also scatters into
// the PyTorch layout; the bias gradient is summed per block from the staged dy tile and reduced by the same launch.

// four consecutive floats of which only the first `nvalid` (0 .. 4) exist; `vec`: nvalid is 0 or 4 and p is 16-byte aligned
__device__ __forceinline__ float4 ld4_masked(const float* p, int nvalid, bool vec) {
  if (vec) return *reinterpret_cast<const float4*>(p);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid > 0) r.x = p[0];
  if (nvalid > 1) r.y = p[1];
  if (nvalid > 2) r.z = p[2];
  if (nvalid > 3) r.w = p[3];
  return r;
}

struct F32WgArgs {
  const float* x1;
  const float* x2;
  const float* dy;
  const float* sc;
  const float* sh;
  float* part;
  float* bpart;        // [spatial blocks][Cout] or null
  int N, ID, IH, IW, OD, OH, OW;
  int C1, C2, Cout, cin16;   // cin16: Cin rounded up to 16 (row length of a partial)
  int SD, SH, SW, PD, PH, PW;
  int HD, HH, HW;
  int ntd, nth, ntw, ndc, dsteps;
  int64_t pstride;
};

template <int KD, int KH, int KW, int NO, int NI, int TD, int TH, bool PRO>
__global__ void __launch_bounds__(256)
k_wgrad_f32_mfma(F32WgArgs a) {
  constexpr int T = KD * KH * KW;
  constexpr int NSLAB = NO * NI, NK = 4 / NSLAB;            // waves that share a slab split the K steps
  constexpr int CI = NI * 16, CO = NO * 16;
  constexpr int PX = CI == 32 ? 48 : 16, PY = CO == 32 ? 48 : 16;   // pitches (floats): lane quads on distinct banks
  constexpr int NV = TD * TH * 16, NSTEPS = NV / 4;
  constexpr int SB = 4;                                       // x-tile slots a thread loads per trip (1 measured the same)
  static_assert(NSLAB == 1 || NSLAB == 2 || NSLAB == 4, "slabs per block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int slab = wave / NK, kpart = wave % NK;
  const int mo = slab / NI, mi = slab % NI;

  int sb = blockIdx.x;                                        // spatial block: (n, D chunk, h tile, w tile)
  const int twi = sb % a.ntw; sb /= a.ntw;
  const int thi = sb % a.nth; sb /= a.nth;
  const int dci = sb % a.ndc;
  const int n = sb / a.ndc;
  const int co0 = blockIdx.y * CO, ci0 = blockIdx.z * CI;
  const bool csmall = a.C1 < 16;                              // ONE input of fewer than 16 channels, zero-extended
  const int Cin = csmall ? a.C1 : a.C1 + a.C2;
  const bool second = !csmall && ci0 >= a.C1;                 // a slab lies in ONE input (C1 is a multiple of CI: host)
  const float* xb = second ? a.x2 : a.x1;
  const int cx = second ? a.C2 : a.C1, cofs = second ? ci0 - a.C1 : ci0;
  const bool xvec = (cx & 3) == 0, yvec = (a.Cout & 3) == 0;  // else: scalar loads of the channels that exist

  const int HV = a.HD * a.HH * a.HW;
  float* xl = reinterpret_cast<float*>(smem);
  float* yl = xl + (size_t)HV * PX;
  float* bred = yl + NV * PY;                                 // 256 floats

  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int bc = tid % CO, bp = tid / CO;                     // bias: channel and voxel residue of this thread
  constexpr int BP = 256 / CO;

  const int oh0 = thi * TH, ow0 = twi * 16;
  const int jh0 = oh0 * a.SH - a.PH, jw0 = ow0 * a.SW - a.PW;
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), h4 = s4;    // prologue coefficients of this thread's piece
  if (PRO) {
    s4 = *reinterpret_cast<const float4*>(a.sc + (size_t)n * Cin + ci0 + (tid % (CI / 4)) * 4);
    h4 = *reinterpret_cast<const float4*>(a.sh + (size_t)n * Cin + ci0 + (tid % (CI / 4)) * 4);
  }
  const int td_begin = dci * a.dsteps, td_end = min(td_begin + a.dsteps, a.ntd);
  for (int tdi = td_begin; tdi < td_end; ++tdi) {
    const int od0 = tdi * TD;
    const int jd0 = od0 * a.SD - a.PD;
    if (tdi > td_begin) __syncthreads();                      // the previous tile's fragments are read
    // ---- x halo tile, CI channels: CI / 4 pieces of 16 bytes per voxel
    // (SB slots per trip with their loads issued together, as in k_conv_f32_mfma; a thread's piece is fixed)
    {
      constexpr int PV = CI / 4;
      const int piece = tid % PV;
      const int xn = min(max(cx - cofs - piece * 4, 0), 4);  // channels of this piece that exist
      for (int i0 = tid; i0 < HV * PV; i0 += SB * 256) {
        float4 v[SB];
        bool ok[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int u = (i0 + k * 256) / PV;
          const int hw = u % a.HW, r = u / a.HW, hh = r % a.HH, hd = r / a.HH;
          const int id = jd0 + hd, ih = jh0 + hh, iw = jw0 + hw;
          ok[k] = u < HV && id >= 0 && id < a.ID && ih >= 0 && ih < a.IH && iw >= 0 && iw < a.IW && xn > 0;
          const size_t vox = ok[k] ? (size_t)((n * a.ID + id) * a.IH + ih) * a.IW + iw : 0;
          v[k] = ld4_masked(xb + (ok[k] ? vox * cx + cofs + piece * 4 : 0), ok[k] ? xn : (xvec ? 4 : 0), xvec);
        }
#pragma unroll
        for (int k = 0; k < SB; ++k) {
          const int u = (i0 + k * 256) / PV;
          float4 o = v[k];
          if (PRO) {
            o.x = fmaxf(fmaf(o.x, s4.x, h4.x), 0.f); o.y = fmaxf(fmaf(o.y, s4.y, h4.y), 0.f);
            o.z = fmaxf(fmaf(o.z, s4.z, h4.z), 0.f); o.w = fmaxf(fmaf(o.w, s4.w, h4.w), 0.f);
          }
          if (!ok[k]) o = make_float4(0.f, 0.f, 0.f, 0.f);
          if (u < HV) *reinterpret_cast<float4*>(xl + (size_t)u * PX + piece * 4) = o;
        }
      }
    }
    // ---- dy tile, CO channels (channels >= Cout and voxels outside the tensor are zero)
    {
      constexpr int PV = CO / 4;
      const int piece = tid % PV, co = co0 + piece * 4;
      const int yn = min(max(a.Cout - co, 0), 4);
      for (int i0 = tid; i0 < NV * PV; i0 += 1024) {
        float4 v[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v_ = (i0 + k * 256) / PV;
          const int l = v_ & 15, row = v_ >> 4, lth = row % TH, ltd = row / TH;
          const int od = od0 + ltd, oh = oh0 + lth, ow = ow0 + l;
          ok[k] = v_ < NV && od < a.OD && oh < a.OH && ow < a.OW && yn > 0;
          const size_t vox = ok[k] ? (size_t)((n * a.OD + od) * a.OH + oh) * a.OW + ow : 0;
          v[k] = ld4_masked(a.dy + (ok[k] ? vox * a.Cout + co : 0), ok[k] ? yn : (yvec ? 4 : 0), yvec);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v_ = (i0 + k * 256) / PV;
          if (v_ < NV) *reinterpret_cast<float4*>(yl + v_ * PY + piece * 4) = ok[k] ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    __syncthreads();
    if (a.bpart && blockIdx.z == 0) {
      for (int v_ = bp; v_ < NV; v_ += BP) bsum += yl[v_ * PY + bc];
    }
    // ---- K steps of this wave: 4 voxels each
    for (int s = kpart; s < NSTEPS; s += NK) {
      const int row = s >> 2, quad = s & 3, lth = row % TH, ltd = row / TH;
      const float af = yl[(row * 16 + quad * 4 + lq) * PY + mo * 16 + l15];
      const float* xp = xl + (size_t)(((ltd * a.SD) * a.HH + lth * a.SH) * a.HW + (quad * 4 + lq) * a.SW) * PX + mi * 16 + l15;
#pragma unroll
      for (int kd = 0; kd < KD; ++kd)
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
          for (int kw = 0; kw < KW; ++kw) {
            const int t = (kd * KH + kh) * KW + kw;
            const float bf = xp[((kd * a.HH + kh) * a.HW + kw) * PX];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[t], 0, 0, 0);
          }
    }
  }

  // ---- partial of this wave: rows co0 + 16 mo + 4 lq + j, column ci0 + 16 mi + l15 of every tap
  float* part = a.part + (size_t)((size_t)blockIdx.x * NK + kpart) * a.pstride;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + mo * 16 + lq * 4 + j;
      if (co < a.Cout) part[((size_t)t * a.Cout + co) * a.cin16 + ci0 + mi * 16 + l15] = acc[t][j];
    }
  if (a.bpart && blockIdx.z == 0) {
    __syncthreads();
    bred[tid] = bsum;                                         // [bp][bc]
    __syncthreads();
    if (tid < CO && co0 + tid < a.Cout) {
      float s = 0.f;
      for (int k = 0; k < BP; ++k) s += bred[k * CO + tid];
      a.bpart[(size_t)blockIdx.x * a.Cout + co0 + tid] = s;
    }
  }
}

struct F32WgPlan {
  int no, ni, td, th;
  int ndc, dsteps, nk;
  int64_t nspatial;
  size_t lds, part_bytes, total_bytes;
  F32WgArgs a;
};

static bool f32_wgrad_plan(const devis_conv_desc* d, F32WgPlan* p) {
  if (d->dtype != DEVIS_F32) return false;
  const bool k333 = d->KD == 3 && d->KH == 3 && d->KW == 3, k133 = d->KD == 1 && d->KH == 3 && d->KW == 3;
  const bool k111 = d->KD == 1 && d->KH == 1 && d->KW == 1, k222 = d->KD == 2 && d->KH == 2 && d->KW == 2;
  const bool k122 = d->KD == 1 && d->KH == 2 && d->KW == 2;
  if (!(k333 || k133 || k111 || k222 || k122)) return false;
  if (d->SD < 1 || d->SD > 2 || d->SH < 1 || d->SH > 2 || d->SW < 1 || d->SW > 2) return false;
  const bool first = d->C1 < 16 && d->C2 == 0;                // one input of fewer than 16 channels: zero-extended
  if (!first && (d->C1 % 16 || d->C2 % 16)) return false;
  if (d->N * d->ID * d->IH * d->IW >= ((int64_t)1 << 31) || d->N * d->OD * d->OH * d->OW >= ((int64_t)1 << 31)) return false;
  const bool strided = d->SD > 1 || d->SH > 1 || d->SW > 1;
  const int64_t cin = d->C1 + d->C2;
  // slabs per block: 2 x 2 where both sides have 32 channels (and a slab pair lies in one input), else fewer
  if (!strided && d->Cout >= 32 && d->Cout % 32 == 0 && cin >= 32 && d->C1 % 32 == 0 && d->C2 % 32 == 0) { p->no = 2; p->ni = 2; p->td = 2; p->th = 4; }
  else if (strided && d->Cout >= 32 && d->Cout % 32 == 0) { p->no = 2; p->ni = 1; p->td = 2; p->th = 4; }
  else if (strided) { p->no = 1; p->ni = 1; p->td = 2; p->th = 4; }
  else { p->no = 1; p->ni = 1; p->td = 2; p->th = 8; }
  if (d->OD == 1 && d->KD == 1) { p->td = 1; p->th *= 2; }     // the 2-D nets: one plane, twice the rows
  p->nk = 4 / (p->no * p->ni);
  F32WgArgs& a = p->a;
  a.N = (int)d->N; a.ID = (int)d->ID; a.IH = (int)d->IH; a.IW = (int)d->IW; a.OD = (int)d->OD; a.OH = (int)d->OH; a.OW = (int)d->OW;
  a.C1 = (int)d->C1; a.C2 = (int)d->C2; a.Cout = (int)d->Cout; a.cin16 = first ? 16 : (int)cin;
  a.SD = d->SD; a.SH = d->SH; a.SW = d->SW; a.PD = d->PD; a.PH = d->PH; a.PW = d->PW;
  a.HD = (p->td - 1) * d->SD + d->KD; a.HH = (p->th - 1) * d->SH + d->KH; a.HW = 15 * d->SW + d->KW;
  a.ntd = (int)cdiv(d->OD, p->td); a.nth = (int)cdiv(d->OH, p->th); a.ntw = (int)cdiv(d->OW, 16);
  const int px = p->ni == 2 ? 48 : 16, py = p->no == 2 ? 48 : 16;
  p->lds = ((size_t)a.HD * a.HH * a.HW * px + (size_t)p->td * p->th * 16 * py + 256) * sizeof(float);
  if (p->lds > 160 * 1024) return false;
  // D tiles per block: as many blocks as it takes to cover the chip about twice, no more (every block leaves partials)
  const int64_t cols = (int64_t)a.N * a.nth * a.ntw * cdiv(d->Cout, p->no * 16) * cdiv(a.cin16, p->ni * 16);
  int ndc = (int)cdiv(512, cols);
  if (ndc < 1) ndc = 1;
  if (ndc > a.ntd) ndc = a.ntd;
  a.dsteps = (int)cdiv(a.ntd, ndc);
  a.ndc = (int)cdiv(a.ntd, a.dsteps);
  p->ndc = a.ndc; p->dsteps = a.dsteps;
  p->nspatial = (int64_t)a.N * a.ndc * a.nth * a.ntw;
  const int T = d->KD * d->KH * d->KW;
  a.pstride = (int64_t)T * d->Cout * a.cin16;
  p->part_bytes = (size_t)p->nspatial * p->nk * a.pstride * sizeof(float);
  p->total_bytes = p->part_bytes + (size_t)p->nspatial * d->Cout * sizeof(float);
  if (p->nspatial > 0x7fffffff || p->total_bytes > ((size_t)1 << 31)) return false;
  if (cdiv(d->Cout, p->no * 16) > 65535 || cdiv(a.cin16, p->ni * 16) > 65535) return false;
  return true;
}

size_t conv_wgrad_f32_workspace(const devis_conv_desc* d) {
  F32WgPlan p;
  return f32_wgrad_plan(d, &p) ? p.total_bytes : 0;
}

template <int KD, int KH, int KW>
static int launch_wg_f32(const F32WgPlan& p, hipStream_t s) {
  const F32WgArgs& a = p.a;
  const dim3 grid((unsigned)p.nspatial, (unsigned)cdiv(a.Cout, p.no * 16), (unsigned)cdiv(a.cin16, p.ni * 16));
  const size_t lds = p.lds;
  auto go = [&](auto k) -> int {
    if (lds > 65536) {
      static thread_local const void* raised[32] = {};
      bool have = false;
      for (const void* q : raised) have = have || q == (const void*)k;
      if (!have) {
        if (check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))) return DEVIS_ERR_HIP;
        for (const void*& q : raised) if (!q) { q = (const void*)k; break; }
      }
    }
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
    return launch_status();
  };
#define WGF(NO, NI, TD, TH) \
  if (p.no == NO && p.ni == NI && p.td == TD && p.th == TH) \
    return a.sc ? go(k_wgrad_f32_mfma<KD, KH, KW, NO, NI, TD, TH, true>) : go(k_wgrad_f32_mfma<KD, KH, KW, NO, NI, TD, TH, false>)
  if constexpr (KD == 1) {                             // one-plane tiles of the 2-D nets
    WGF(2, 2, 1, 8);
    WGF(2, 1, 1, 8);
    WGF(1, 1, 1, 8);
    WGF(1, 1, 1, 16);
  }
  if constexpr (KD > 1 || (KH == 1 && KW == 1)) {
    WGF(2, 2, 2, 4);
    WGF(2, 1, 2, 4);
    WGF(1, 1, 2, 4);
    WGF(1, 1, 2, 8);
  }
#undef WGF
  return DEVIS_ERR_UNSUPPORTED;
}

int conv_wgrad_f32_mfma(const devis_conv_desc* d, const void* x1, const void* x2, const float* sc, const float* sh,
                        const void* dy, float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s) {
  F32WgPlan p;
  if (!f32_wgrad_plan(d, &p)) return DEVIS_ERR_UNSUPPORTED;
  if (d->C1 < 16 && sc) return DEVIS_ERR_UNSUPPORTED;         // the prologue reads its coefficients four at a time
  if (!ws || ws_bytes < p.total_bytes) return DEVIS_ERR_WORKSPACE;
  F32WgArgs& a = p.a;
  a.x1 = (const float*)x1; a.x2 = (const float*)x2; a.dy = (const float*)dy; a.sc = sc; a.sh = sh;
  a.part = (float*)ws; a.bpart = dbias ? (float*)((char*)ws + p.part_bytes) : nullptr;
  int rc;
  if (d->KD == 3) rc = launch_wg_f32<3, 3, 3>(p, s);
  else if (d->KH == 3) rc = launch_wg_f32<1, 3, 3>(p, s);
  else if (d->KD == 2) rc = launch_wg_f32<2, 2, 2>(p, s);
  else if (d->KH == 2) rc = launch_wg_f32<1, 2, 2>(p, s);
  else rc = launch_wg_f32<1, 1, 1>(p, s);
  if (rc) return rc;
  RedArgs r;
  r.part = a.part; r.bpart = a.bpart; r.dw = dw; r.dbias = dbias;
  r.nparts = (int)(p.nspatial * p.nk); r.nbparts = (int)p.nspatial; r.Cb = a.Cout;
  r.pstride = a.pstride; r.total = a.pstride;
  r.mode = RED_CONV3; r.A = a.Cout; r.B = a.cin16; r.cin_real = a.C1 < 16 ? a.C1 : a.cin16;
  return launch_reduce_partials(r, s);
}

// Column sums of a [rows][C] fp32 matrix in a fixed order (the bias gradient of a transposed convolution: its dy is not
// an operand the weight-gradient kernel stages as "dy").  C a power of two <= 256.  Block b sums its share of the rows
// per channel (256 / C threads per channel, combined through LDS in thread order) into bpart[b][C]; k_reduce_partials
// adds the blocks.
__global__ void __launch_bounds__(256)
k_colsum_det(const float* __restrict__ x, int64_t rows, int C, int64_t rows_per_block, float* __restrict__ bpart) {
  __shared__ float red[256];
  const int c = threadIdx.x % C, sub = threadIdx.x / C, nsub = 256 / C;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc = 0.f;
  for (int64_t r = r0 + sub; r < r1; r += nsub) acc += x[r * C + c];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < nsub; ++k) t += red[k * C + threadIdx.x];
    bpart[(size_t)blockIdx.x * C + threadIdx.x] = t;
  }
}
static bool colsum_det_ok(int64_t C) { return C >= 1 && C <= 256 && (C & (C - 1)) == 0; }
static const int COLSUM_BLOCKS = 1024;

// Weight gradient of the transposed convolution (kernel == stride): dW[ci][co][t] = sum_i x[i][ci] dy[K i + t][co] is
// the weight gradient of the convolution dy -> x with kernel = stride = K (its "input" is dy, its "output gradient" x),
// and the reduction's [cout'][cin'][T] order is PyTorch's [Cin][Cout][T].  The bias gradient (a column sum of dy) is
// left to the caller.
static bool convT_as_conv(const devis_convT_desc* d, devis_conv_desc* c) {
  if (d->dtype != DEVIS_F32 || d->Cout % 16 || d->Cin % 4) return false;
  c->N = d->N; c->ID = d->ID * d->KD; c->IH = d->IH * d->KH; c->IW = d->IW * d->KW;
  c->OD = d->ID; c->OH = d->IH; c->OW = d->IW;
  c->C1 = d->Cout; c->C2 = 0; c->Cout = d->Cin;
  c->KD = d->KD; c->KH = d->KH; c->KW = d->KW; c->SD = d->KD; c->SH = d->KH; c->SW = d->KW; c->PD = c->PH = c->PW = 0;
  c->dtype = d->dtype;
  return true;
}
size_t convT_wgrad_f32_workspace(const devis_convT_desc* d) {
  devis_conv_desc c;
  if (!convT_as_conv(d, &c)) return 0;
  const size_t w = conv_wgrad_f32_workspace(&c);
  return w ? w + (size_t)COLSUM_BLOCKS * d->Cout * sizeof(float) : 0;
}
// dbias: written here (fixed-order column sums of dy) when the channel count allows, else *bias_done stays false and
// the caller sums it its own way.
int convT_wgrad_f32_mfma(const devis_convT_desc* d, const void* x, const void* dy, float* dw, float* dbias, bool* bias_done,
                         void* ws, size_t ws_bytes, hipStream_t s) {
  devis_conv_desc c;
  *bias_done = false;
  if (!convT_as_conv(d, &c)) return DEVIS_ERR_UNSUPPORTED;
  const size_t wbytes = conv_wgrad_f32_workspace(&c);
  if (!wbytes) return DEVIS_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < wbytes + (size_t)COLSUM_BLOCKS * d->Cout * sizeof(float)) return DEVIS_ERR_WORKSPACE;
  int rc = conv_wgrad_f32_mfma(&c, dy, nullptr, nullptr, nullptr, x, dw, nullptr, ws, wbytes, s);
  if (rc || !dbias || !colsum_det_ok(d->Cout)) return rc;
  const int64_t rows = d->N * d->ID * d->KD * d->IH * d->KH * d->IW * d->KW;
  const int64_t rpb = cdiv(rows, COLSUM_BLOCKS);
  const int blocks = (int)cdiv(rows, rpb);
  float* bpart = reinterpret_cast<float*>((char*)ws + wbytes);
  hipLaunchKernelGGL(k_colsum_det, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dy, rows, (int)d->Cout, rpb, bpart);
  rc = launch_status();
  if (rc) return rc;
  RedArgs r;
  r.part = bpart; r.bpart = bpart; r.dw = dbias; r.dbias = dbias;
  r.nparts = 0; r.nbparts = blocks; r.Cb = (int)d->Cout;
  r.pstride = 0; r.total = 0; r.mode = RED_CONV3; r.A = 1; r.B = 1; r.cin_real = 1;
  rc = launch_reduce_partials(r, s);
  *bias_done = rc == DEVIS_OK;
  return rc;
}

}  // namespace devis
