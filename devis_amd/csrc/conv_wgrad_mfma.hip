// MFMA weight-gradient of the 3x3x3 convolutions, bf16 channels-last, gfx950.
//
//   dW[co][tap][ci] = sum_v dy[v][co] * A[v*S + off(tap)][ci],   A = relu(scale*x + shift) (fused prologue)
//
// GEMM: M = cout, N = cin, K = voxels.  Both operands need K (voxels) per lane while memory is
// channels-last, so both are staged as [voxel][channel] LDS images and read with the gfx950
// transposing LDS read (ds_read_b64_tr_b16): two reads give a lane 8 voxels of one channel.
// One K step = 32 voxels = 2 rows of 16 along W.  A block owns a (cout-slab x cin-slab) pair of up
// to 32x32 channels and walks many spatial tiles keeping the 27 tap accumulators in registers
// (4 waves x 7 taps); partial sums go to a workspace and a second kernel adds them in a fixed order
// (deterministic, no atomics) and writes the PyTorch [Cout][Cin][27] fp32 layout plus dbias.
#include "common.h"
#include "mfma_common.h"
#include "wgrad_reduce.h"

#include <stdlib.h>

namespace devis {

struct WgArgs {
  const bf16* x;     // [N][ID][IH][IW][cin_real]
  const bf16* dy;    // [N][OD][OH][OW][Cout]
  const float* sc;   // [N][cin_real] or null
  const float* sh;
  float* part;       // [gx][27][Cout][Cin]
  float* bpart;      // [gx][Cout] or null
  int N, ID, IH, IW, OD, OH, OW, Cin, Cout, cin_real;
  int ntd, nth, ntw, ntiles;
};

template <int CIS, int COS, int TD, int TH, int S>
struct WCfg {
  static constexpr int MTI = CIS / 16, MTO = COS / 16;
  static constexpr int HD = (TD - 1) * S + 3, HH = (TH - 1) * S + 3, HW = 15 * S + 3;
  static constexpr int HV = HD * HH * HW;
  static constexpr int XP = CIS * 2, YP = COS * 2;          // LDS pitches (bytes per voxel)
  static constexpr int TV = TD * TH * 16;
  static constexpr int NSEG = TV / 32;
  static constexpr int X_BYTES = HV * XP;
  static constexpr int LDS_BYTES = X_BYTES + TV * YP;
  static_assert(TH % 2 == 0, "a K step is two rows");
};

// 32-byte half h of voxel u lives at ((h ^ ((u >> 2) & 1)) * 32): 8 consecutive voxels then hit 8 distinct
// 32-byte bank groups for the transposed reads (conflict-free for any starting voxel).
template <int PITCH>
__device__ __forceinline__ int img_off(int u, int half) {
  return PITCH == 64 ? u * 64 + ((half ^ ((u >> 2) & 1)) << 5) : u * 32;
}

__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = a;
  u.h[1] = b;
  return u.v;
}

template <int CIS, int COS, int TD, int TH, int S, bool PRO, bool SMALLC>
__global__ void __launch_bounds__(256)
k_wgrad3_mfma(WgArgs a) {
  using C = WCfg<CIS, COS, TD, TH, S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* ys = smem + C::X_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int nci = a.Cin / CIS;
  const int ci0 = (blockIdx.y % nci) * CIS, co0 = (blockIdx.y / nci) * COS;

  f32x4 acc[7][C::MTO][C::MTI];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int mo = 0; mo < C::MTO; ++mo)
#pragma unroll
      for (int mi = 0; mi < C::MTI; ++mi) acc[t][mo][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int XPPV = CIS / 8, YPPV = COS / 8;
  const int xpiece = tid % XPPV, ypiece = tid % YPPV;
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  const bool do_bias = a.bpart != nullptr && ci0 == 0;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int twi = tile % a.ntw;
    const int thi = (tile / a.ntw) % a.nth;
    const int tdi = (tile / (a.ntw * a.nth)) % a.ntd;
    const int n = tile / (a.ntw * a.nth * a.ntd);
    const int od0 = tdi * TD, oh0 = thi * TH, ow0 = twi * 16;
    __syncthreads();          // previous tile fully consumed
    // ---- stage x halo tile (fused prologue, zero padding)
    {
      float scv[8], shv[8];
      if (PRO) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = ci0 + xpiece * 8 + j;
          scv[j] = c < a.cin_real ? a.sc[n * a.cin_real + c] : 0.f;
          shv[j] = c < a.cin_real ? a.sh[n * a.cin_real + c] : 0.f;
        }
      }
      for (int i = tid; i < C::HV * XPPV; i += 256) {
        const int u = i / XPPV;
        const int hw = u % C::HW, hh = (u / C::HW) % C::HH, hd = u / (C::HW * C::HH);
        const int id = od0 * S - 1 + hd, ih = oh0 * S - 1 + hh, iw = ow0 * S - 1 + hw;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (id >= 0 && id < a.ID && ih >= 0 && ih < a.IH && iw >= 0 && iw < a.IW) {
          const size_t vox = (size_t)((n * a.ID + id) * a.IH + ih) * a.IW + iw;
          if (SMALLC) {       // 4 real channels, zero-extended to 16 in LDS
            if (xpiece == 0) {
              const uint2 t = *reinterpret_cast<const uint2*>(a.x + vox * 4);
              v.x = t.x; v.y = t.y;
            }
          } else {
            v = *reinterpret_cast<const uint4*>(a.x + vox * a.Cin + ci0 + xpiece * 8);
          }
          if (PRO) {
            v = affine_relu8(v, scv, shv);
            if (SMALLC) { v.z = 0u; v.w = 0u; if (xpiece) { v.x = 0u; v.y = 0u; } }
          }
        }
        *reinterpret_cast<uint4*>(xs + img_off<C::XP>(u, xpiece >> 1) + (xpiece & 1) * 16) = v;
      }
    }
    // ---- stage dy tile
    for (int i = tid; i < C::TV * YPPV; i += 256) {
      const int vd = i / YPPV;
      const int w = vd & 15, h = (vd >> 4) % TH, d = vd / (16 * TH);
      const int od = od0 + d, oh = oh0 + h, ow = ow0 + w;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (od < a.OD && oh < a.OH && ow < a.OW) {
        v = *reinterpret_cast<const uint4*>(a.dy + ((size_t)((n * a.OD + od) * a.OH + oh) * a.OW + ow) * a.Cout + co0 + ypiece * 8);
        if (do_bias) {
          bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
          bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
        }
      }
      *reinterpret_cast<uint4*>(ys + img_off<C::YP>(vd, ypiece >> 1) + (ypiece & 1) * 16) = v;
    }
    __syncthreads();
    // ---- MFMA: every wave walks all K steps of the tile for its own taps (wave, wave+4, ...)
#pragma unroll 1
    for (int sgm = 0; sgm < C::NSEG; ++sgm) {
      const int ltd = sgm / (TH / 2), lth0 = (sgm % (TH / 2)) * 2;
      const int wv = 4 * g + q;               // this lane addresses voxel column wv of rows lth0, lth0+1
      bf16x8 af[C::MTO];
      {
        const int v0 = (ltd * TH + lth0) * 16 + wv, v1 = v0 + 16;
#pragma unroll
        for (int mo = 0; mo < C::MTO; ++mo)
          af[mo] = tr_pair(ys + img_off<C::YP>(v0, mo) + p * 8, ys + img_off<C::YP>(v1, mo) + p * 8);
      }
      const int ub = ((ltd * S) * C::HH + lth0 * S) * C::HW + wv * S;
#pragma unroll
      for (int tt = 0; tt < 7; ++tt) {
        const int t = wave + 4 * tt;          // wave-uniform
        if (t < 27) {
          const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
          const int u0 = ub + (kd * C::HH + kh) * C::HW + kw, u1 = u0 + S * C::HW;
#pragma unroll
          for (int mi = 0; mi < C::MTI; ++mi) {
            const bf16x8 bfr = tr_pair(xs + img_off<C::XP>(u0, mi) + p * 8, xs + img_off<C::XP>(u1, mi) + p * 8);
#pragma unroll
            for (int mo = 0; mo < C::MTO; ++mo)
              acc[tt][mo][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mo], bfr, acc[tt][mo][mi], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- partial results: D layout col = ci = lane&15, rows co = (lane>>4)*4 + j
  float* part = a.part + (size_t)blockIdx.x * a.Cout * a.Cin * 27;
#pragma unroll
  for (int tt = 0; tt < 7; ++tt) {
    const int t = wave + 4 * tt;
    if (t < 27) {
#pragma unroll
      for (int mo = 0; mo < C::MTO; ++mo)
#pragma unroll
        for (int mi = 0; mi < C::MTI; ++mi) {
          const int ci = ci0 + mi * 16 + (lane & 15);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = co0 + mo * 16 + g * 4 + j;
            part[((size_t)t * a.Cout + co) * a.Cin + ci] = acc[tt][mo][mi][j];   // ci contiguous over lanes
          }
        }
    }
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);       // [256][8]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < COS) {
      const int pc = tid / 8, j = tid % 8;
      float s = 0.f;
      for (int k = pc; k < 256; k += YPPV) s += red[k * 8 + j];
      a.bpart[(size_t)blockIdx.x * a.Cout + co0 + tid] = s;
    }
  }
}

// dw[co][ci<cin_real][t] = sum_gx part[gx][t][co][ci];  dbias[co] = sum_gx bpart[gx][co]   (wgrad_reduce.h)
thread_local bool g_wgrad_stage1_only = false;     // devis_conv_wgrad_stage1 (capi.hip): profiling the first stage alone

int launch_wgrad_reduce(const float* part, const float* bpart, int gx, int Cout, int Cin, int cin_real, float* dw,
                        float* dbias, hipStream_t s) {
  if (g_wgrad_stage1_only) return DEVIS_OK;
  RedArgs r;
  r.part = part; r.bpart = bpart; r.dw = dw; r.dbias = dbias;
  r.nparts = gx; r.nbparts = gx; r.Cb = Cout;
  r.pstride = (int64_t)27 * Cout * Cin; r.total = r.pstride;
  r.mode = RED_CONV3; r.A = Cout; r.B = Cin; r.cin_real = cin_real;
  return launch_reduce_partials(r, s);
}

// the same for partials in the accumulator order of k_wgrad3_slab_march (RED_CONV3F in wgrad_reduce.h)
int launch_wgrad_reduce_frag(const float* part, const float* bpart, int gx, int Cout, int Cin, float* dw, float* dbias,
                             hipStream_t s) {
  if (g_wgrad_stage1_only) return DEVIS_OK;
  RedArgs r;
  r.part = part; r.bpart = bpart; r.dw = dw; r.dbias = dbias;
  r.nparts = gx; r.nbparts = gx; r.Cb = Cout;
  r.pstride = (int64_t)27 * Cout * Cin; r.total = r.pstride;
  r.mode = RED_CONV3F; r.A = Cout; r.B = Cin; r.cin_real = Cin;
  return launch_reduce_partials(r, s);
}

int wgrad3_c16_march(const devis_conv_desc* d, const void* x, const float* sc, const float* sh, const void* dy,
                     float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s, bool query, size_t* need);
int wgrad3_slab_march(const devis_conv_desc* d, const void* x, const float* sc, const float* sh, const void* dy,
                      float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s, bool query, size_t* need);

static int wg_gx(int64_t ntiles, int nslab) {
  int64_t cap = 512 / nslab;
  if (cap < 8) cap = 8;
  return (int)(ntiles < cap ? ntiles : cap);
}

template <int CIS, int COS, int TD, int TH, int S, bool SMALLC>
static int launch_wgrad(const devis_conv_desc* d, const void* x, const float* sc, const float* sh, const void* dy,
                        float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s, bool query, size_t* need) {
  using C = WCfg<CIS, COS, TD, TH, S>;
  const int cin_real = (int)d->C1;
  const int Cin = SMALLC ? 16 : cin_real, Cout = (int)d->Cout;
  WgArgs a;
  a.ntd = (int)cdiv(d->OD, TD); a.nth = (int)cdiv(d->OH, TH); a.ntw = (int)cdiv(d->OW, 16);
  const int64_t ntiles = d->N * a.ntd * a.nth * a.ntw;
  if (ntiles > 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const int nslab = (Cin / CIS) * (Cout / COS);
  const int gx = wg_gx(ntiles, nslab);
  const size_t part_bytes = (size_t)gx * Cout * Cin * 27 * sizeof(float);
  const size_t total = part_bytes + (size_t)gx * Cout * sizeof(float);
  if (query) { *need = total; return DEVIS_OK; }
  if (!ws || ws_bytes < total) return DEVIS_ERR_WORKSPACE;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.sc = sc; a.sh = sh;
  a.part = (float*)ws; a.bpart = dbias ? (float*)((char*)ws + part_bytes) : nullptr;
  a.N = (int)d->N; a.ID = (int)d->ID; a.IH = (int)d->IH; a.IW = (int)d->IW; a.OD = (int)d->OD; a.OH = (int)d->OH; a.OW = (int)d->OW;
  a.Cin = Cin; a.Cout = Cout; a.cin_real = cin_real; a.ntiles = (int)ntiles;
  const dim3 grid((unsigned)gx, (unsigned)nslab);
  if (sc) hipLaunchKernelGGL((k_wgrad3_mfma<CIS, COS, TD, TH, S, true, SMALLC>), grid, dim3(256), C::LDS_BYTES, s, a);
  else hipLaunchKernelGGL((k_wgrad3_mfma<CIS, COS, TD, TH, S, false, SMALLC>), grid, dim3(256), C::LDS_BYTES, s, a);
  int rc = launch_status();
  if (rc) return rc;
  return launch_wgrad_reduce(a.part, a.bpart, gx, Cout, Cin, cin_real, dw, dbias, s);
}

int wgrad3_s2_c16_march(const devis_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                        size_t ws_bytes, hipStream_t s, bool query, size_t* need);

static int wgrad_dispatch(const devis_conv_desc* d, const void* x, const float* sc, const float* sh, const void* dy,
                          float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s, bool query, size_t* need) {
  if (d->dtype != DEVIS_BF16 || d->C2 != 0) return DEVIS_ERR_UNSUPPORTED;
  if (!(d->KD == 3 && d->KH == 3 && d->KW == 3 && d->PD == 1 && d->PH == 1 && d->PW == 1 && d->SD == d->SH && d->SH == d->SW))
    return DEVIS_ERR_UNSUPPORTED;
  if (d->N * d->ID * d->IH * d->IW >= 0x7fffffff) return DEVIS_ERR_UNSUPPORTED;
  const int ci = (int)d->C1, co = (int)d->Cout, st = d->SD;
#define WG(CIS, COS, TD, TH, S, SM) return launch_wgrad<CIS, COS, TD, TH, S, SM>(d, x, sc, sh, dy, dw, dbias, ws, ws_bytes, s, query, need)
  if (st == 1) {
    if (co == 16 && (ci == 16 || (ci == 4 && !sc))) {
      const int rc = wgrad3_c16_march(d, x, sc, sh, dy, dw, dbias, ws, ws_bytes, s, query, need);
      if (rc != DEVIS_ERR_UNSUPPORTED && !(query && ci == 4)) return rc;
      if (query && ci == 4) {       // the prologue variant of the 4-channel layer runs on the tile kernel: size for both
        size_t other = 0;
        const int rc2 = launch_wgrad<16, 16, 4, 4, 1, true>(d, x, sc, sh, dy, dw, dbias, ws, ws_bytes, s, true, &other);
        if (rc2 == DEVIS_OK && other > *need) *need = other;
        return rc == DEVIS_OK ? rc : rc2;
      }
    }
    if (ci % 32 == 0 && co % 32 == 0 && ci <= 128 && co <= 128) {
      const int rc = wgrad3_slab_march(d, x, sc, sh, dy, dw, dbias, ws, ws_bytes, s, query, need);
      if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
    }
    if (ci == 4 && co == 16) WG(16, 16, 4, 4, 1, true);
    if (ci == 16 && co == 16) WG(16, 16, 4, 4, 1, false);
    if (ci % 32 == 0 && co % 32 == 0 && ci <= 128 && co <= 128) WG(32, 32, 4, 4, 1, false);
  } else if (st == 2) {
    // 16 -> 32 without a prologue (EnDown1): plane march (conv_wgrad_s2_march.hip).
    // A query sizes the workspace for both (the prologue variant stays on the tile kernel).
    if (ci == 16 && co == 32 && (query || !sc)) {
      const int rc = wgrad3_s2_c16_march(d, x, dy, dw, dbias, ws, ws_bytes, s, query, need);
      if (!query && rc != DEVIS_ERR_UNSUPPORTED) return rc;
      if (query) {
        size_t other = 0;
        const int rc2 = launch_wgrad<16, 32, 2, 2, 2, false>(d, x, sc, sh, dy, dw, dbias, ws, ws_bytes, s, true, &other);
        if (rc != DEVIS_OK) { *need = other; return rc2; }
        if (rc2 == DEVIS_OK && other > *need) *need = other;
        return DEVIS_OK;
      }
    }
    if (ci == 16 && co % 32 == 0 && co <= 128) WG(16, 32, 2, 2, 2, false);
    if (ci % 32 == 0 && co % 32 == 0 && ci <= 128 && co <= 128) WG(32, 32, 2, 2, 2, false);
  }
#undef WG
  return DEVIS_ERR_UNSUPPORTED;
}

size_t conv_wgrad_mfma_workspace(const devis_conv_desc* d) {
  size_t need = 0;
  if (wgrad_dispatch(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true, &need) != DEVIS_OK) return 0;
  return need;
}

int conv_wgrad_mfma(const devis_conv_desc* d, const void* x1, const void* x2, const float* sc, const float* sh,
                    const void* dy, float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t s) {
  if (x2) return DEVIS_ERR_UNSUPPORTED;
  size_t need = 0;
  return wgrad_dispatch(d, x1, sc, sh, dy, dw, dbias, ws, ws_bytes, s, false, &need);
}

}  // namespace devis
