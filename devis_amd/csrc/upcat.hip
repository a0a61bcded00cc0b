// DeUp_Cat tail fused (bf16 channels-last, gfx950): the reference's
//     up = ConvTranspose3d(C, C, k=2, s=2)(t);  y = Conv3d(2C, C, k=1)(cat((prev, up), 1))      (UNet3DZoo.py:121-129)
// is ONE pass  y[v] = W3a prev[v] + M[parity(v)] t[coarse(v)] + b'  with the composite matrices
//     M[p] = W3b Wt[p]  (C x C, one per 2x2x2 parity p),  b' = b3 + W3b bt,   W3 = [W3a | W3b]:
// `up` (a full-resolution C-channel tensor: written once, read once) never exists, and neither does its gradient.
// HBM bytes per fine voxel: read C*s (prev) + C*s/8 (t), write C*s  -  against 4 C*s + C*s/8 for the two layers.
// Backward uses the same matrices: d prev = W3a^T dy (1x1 dgrad), d t = sum_p M[p]^T dy[fine(p)] (the k2-s2 gather
// GEMM on `wq`), and the two weight gradients come from  dW3a = dy (x) prev  and  G[p] = dy[fine(p)] (x) t  (the 1x1 and
// the ConvT weight-gradient kernels) through k_upcat_combine:
//     dW3b = sum_p G[p] Wt[p]^T + s bt^T,  dWt[p] = W3b^T G[p],  db3 = s = sum_v dy[v],  dbt = W3b^T s.
#include "common.h"
#include "mfma_common.h"

namespace devis {

struct UpcatArgs {
  const bf16* prev; const bf16* t; const bf16* wc; const float* bias; bf16* y;
  int N, D, H, W;          // COARSE grid (t); prev / y are (2D, 2H, 2W)
  int ntw;                 // 16-voxel coarse tiles per row
  unsigned ntiles;
};

// ---- C = 16.  A wave takes 16 consecutive coarse voxels of one row = 2 x 2 x 32 fine voxels.  K = 32 = 16 prev
// channels + 16 t channels: one MFMA per parity class and 16 fine voxels.  One 64-lane load fetches a whole 1 KB
// fine row segment: lanes lq = 0,1 hold the two 16-byte halves of fine voxel 2j, lanes lq = 2,3 those of voxel
// 2j+1; t is loaded in all four lane groups.  The even-w MFMA takes [prev | t] as its K order, the odd-w one
// [t | prev] (its weight rows are stored with the halves swapped), so no operand crosses lanes.  After the MFMAs a
// lane owns 4 couts of voxel 2j and 4 of voxel 2j+1; lane pairs (lq, lq^1) swap one of them so that every lane
// stores 16 contiguous bytes and a wave store covers the 1 KB row segment exactly.
template <int TPW>
__global__ void __launch_bounds__(256)
k_upcat16_fwd(UpcatArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const bool lo = lq < 2, odd = lq & 1;
  union U { uint4 u; bf16x8 v; u32x4 w; };
  U af[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) af[p].u = *reinterpret_cast<const uint4*>(a.wc + ((p * 16 + l15) * 32 + lq * 8));
  const float4 bb = *reinterpret_cast<const float4*>(a.bias + lq * 4);
  const int D2 = 2 * a.D, H2 = 2 * a.H, W2 = 2 * a.W;
  (void)D2;
  uint4 pl[TPW][4], tl[TPW];
  size_t rowb[TPW][4];
  bool ok[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const unsigned tile = ((unsigned)blockIdx.x * 4u + (unsigned)wave) * TPW + i;
    ok[i] = tile < a.ntiles;
    const unsigned tt = ok[i] ? tile : 0u;
    const int twi = (int)(tt % (unsigned)a.ntw);
    unsigned r = tt / (unsigned)a.ntw;
    const int h = (int)(r % (unsigned)a.H); r /= (unsigned)a.H;
    const int d = (int)(r % (unsigned)a.D);
    const int n = (int)(r / (unsigned)a.D);
    const int w0 = twi * 16;
    tl[i] = make_uint4(0u, 0u, 0u, 0u);
    if (ok[i]) tl[i] = *reinterpret_cast<const uint4*>(a.t + ((((size_t)n * a.D + d) * a.H + h) * a.W + w0 + l15) * 16 + odd * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pd = q >> 1, ph = q & 1;
      // element offset of this lane's 16-byte piece of fine row (2d+pd, 2h+ph), fine voxels 2*w0 .. 2*w0+31
      rowb[i][q] = ((((size_t)n * D2 + 2 * d + pd) * H2 + 2 * h + ph) * W2 + 2 * w0 + 2 * l15 + (lq >> 1)) * 16 + odd * 8;
      pl[i][q] = make_uint4(0u, 0u, 0u, 0u);
      if (ok[i]) pl[i][q] = *reinterpret_cast<const uint4*>(a.prev + rowb[i][q]);
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    if (!ok[i]) continue;                      // wave-uniform
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      U b0, b1;
      b0.u = lo ? pl[i][q] : tl[i];            // even w: K = [prev(2j) | t]
      b1.u = lo ? tl[i] : pl[i][q];            // odd w:  K = [t | prev(2j+1)]
      const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q * 2].v, b0.v, z, 0, 0, 0);
      f32x4 c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q * 2 + 1].v, b1.v, z, 0, 0, 0);
      // lane: couts lq*4..+3 of fine voxels 2j (c0) and 2j+1 (c1)
      const unsigned o00 = pack_bf2(c0[0] + bb.x, c0[1] + bb.y), o01 = pack_bf2(c0[2] + bb.z, c0[3] + bb.w);
      const unsigned o10 = pack_bf2(c1[0] + bb.x, c1[1] + bb.y), o11 = pack_bf2(c1[2] + bb.z, c1[3] + bb.w);
      // even lq keeps voxel 2j and gets couts (lq+1)*4.. of it from lane+16; odd lq keeps voxel 2j+1
      const unsigned s0 = odd ? o00 : o10, s1 = odd ? o01 : o11;
      const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
      const uint4 out = odd ? make_uint4(r0, r1, o10, o11) : make_uint4(o00, o01, r0, r1);
      // voxel 2j + odd, couts (lq>>1)*8 .. +7
      const size_t off = rowb[i][q] - (size_t)((lq >> 1) * 16 + odd * 8) + (size_t)(odd * 16 + (lq >> 1) * 8);
      *reinterpret_cast<uint4*>(a.y + off) = out;
    }
  }
}

// ---- C = 32.  K = 64 = two MFMA K steps (32 prev channels, 32 t channels), 32 couts = two M tiles.  A wave takes
// 16 coarse voxels; per fine row (pd, ph) and w parity it loads fine voxels 2j+pw (64 B each: four lanes), so the
// two parities' loads and stores interleave in the same 2 KB row segment.  The W3a fragments are shared by all
// parities (2 fragments), the M[p] fragments (16) stay in registers as well.  After the MFMAs lane pairs (lq, lq^1)
// swap one M tile, so a lane stores 8 consecutive couts (16 B) and four lanes cover a voxel's 64-byte row.
__global__ void __launch_bounds__(256)
k_upcat32_fwd(UpcatArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const bool odd = lq & 1;
  union U { uint4 u; bf16x8 v; };
  U wa[2], wm[8][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    wa[mt].u = *reinterpret_cast<const uint4*>(a.wc + ((size_t)(mt * 16 + l15) * 64 + lq * 8));          // parity 0: [W3a | M]
#pragma unroll
    for (int p = 0; p < 8; ++p)
      wm[p][mt].u = *reinterpret_cast<const uint4*>(a.wc + ((size_t)((p * 32 + mt * 16 + l15) * 64) + ((p & 1) ? 0 : 32) + lq * 8));
  }
  float4 bb[2];
  bb[0] = *reinterpret_cast<const float4*>(a.bias + lq * 4);
  bb[1] = *reinterpret_cast<const float4*>(a.bias + 16 + lq * 4);
  const unsigned tile = (unsigned)blockIdx.x * 4u + (unsigned)wave;
  if (tile >= a.ntiles) return;                 // wave-uniform; no barriers in this kernel
  const int twi = (int)(tile % (unsigned)a.ntw);
  unsigned r = tile / (unsigned)a.ntw;
  const int h = (int)(r % (unsigned)a.H); r /= (unsigned)a.H;
  const int d = (int)(r % (unsigned)a.D);
  const int n = (int)(r / (unsigned)a.D);
  const int w0 = twi * 16;
  const int D2 = 2 * a.D, H2 = 2 * a.H, W2 = 2 * a.W;
  (void)D2;
  U tl, pl[8];
  size_t vox[8];
  tl.u = *reinterpret_cast<const uint4*>(a.t + ((((size_t)n * a.D + d) * a.H + h) * a.W + w0 + l15) * 32 + lq * 8);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int pd = p >> 2, ph = (p >> 1) & 1, pw = p & 1;
    vox[p] = (((size_t)n * D2 + 2 * d + pd) * H2 + 2 * h + ph) * W2 + 2 * (w0 + l15) + pw;
    pl[p].u = *reinterpret_cast<const uint4*>(a.prev + vox[p] * 32 + lq * 8);
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    unsigned o[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[mt].v, pl[p].v, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[p][mt].v, tl.v, c, 0, 0, 0);
      o[mt][0] = pack_bf2(c[0] + bb[mt].x, c[1] + bb[mt].y);
      o[mt][1] = pack_bf2(c[2] + bb[mt].z, c[3] + bb[mt].w);
    }
    // even lq keeps M tile 0 (couts lq*4 .. +7 with lane+16's half), odd lq keeps M tile 1 (couts 16 + (lq-1)*4 .. +7)
    const unsigned s0 = odd ? o[0][0] : o[1][0], s1 = odd ? o[0][1] : o[1][1];
    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
    const uint4 out = odd ? make_uint4(r0, r1, o[1][0], o[1][1]) : make_uint4(o[0][0], o[0][1], r0, r1);
    const int choff = odd ? 16 + (lq - 1) * 4 : lq * 4;
    *reinterpret_cast<uint4*>(a.y + vox[p] * 32 + choff) = out;
  }
}

// ---- data gradients, one pass over dy (fine resolution):  d prev[v] = W3a^T dy[v],  d t[c] = sum_p M[p]^T dy[fine(c, p)].
// Same walk and lane layout as the forward kernels; the loads of dy take the place of prev's and the stores of
// d prev the place of y's.  C = 16: the MFMA B operand is K = 32 = [dy(2j) | dy(2j+1)], so d t takes one MFMA per
// fine row pair (A = [M[q,0]^T | M[q,1]^T] from `wq`), and d prev one per w parity with the other half of K zero
// (A = [W3a^T | 0], [0 | W3a^T]).  A wave's two tiles share the 16-byte stores of d t: even lane groups keep tile 0,
// odd ones tile 1.
struct UpcatBwdArgs {
  const bf16* dy; const bf16* w3a_t; const bf16* wq; bf16* dprev; bf16* dt;
  int N, D, H, W;          // COARSE grid (dt); dy / dprev are (2D, 2H, 2W)
  int ntw;
  unsigned ntiles;
};

__global__ void __launch_bounds__(256)
k_upcat16_bwd(UpcatBwdArgs a) {
  constexpr int TPW = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const bool lo = lq < 2, odd = lq & 1;
  union U { uint4 u; bf16x8 v; };
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
  U ae, ao, at[4];
  {
    const uint4 w = *reinterpret_cast<const uint4*>(a.w3a_t + l15 * 16 + odd * 8);
    ae.u = lo ? w : z4;
    ao.u = lo ? z4 : w;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) at[q].u = *reinterpret_cast<const uint4*>(a.wq + (l15 * 8 + q * 2 + (lq >> 1)) * 16 + odd * 8);
  const int D2 = 2 * a.D, H2 = 2 * a.H, W2 = 2 * a.W;
  (void)D2;
  uint4 dl[TPW][4];
  size_t rowb[TPW][4], tb[TPW];
  bool ok[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const unsigned tile = ((unsigned)blockIdx.x * 4u + (unsigned)wave) * TPW + i;
    ok[i] = tile < a.ntiles;
    const unsigned tt = ok[i] ? tile : 0u;
    const int twi = (int)(tt % (unsigned)a.ntw);
    unsigned r = tt / (unsigned)a.ntw;
    const int h = (int)(r % (unsigned)a.H); r /= (unsigned)a.H;
    const int d = (int)(r % (unsigned)a.D);
    const int n = (int)(r / (unsigned)a.D);
    const int w0 = twi * 16;
    tb[i] = ((((size_t)n * a.D + d) * a.H + h) * a.W + w0 + l15) * 16 + (lq >> 1) * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pd = q >> 1, ph = q & 1;
      rowb[i][q] = ((((size_t)n * D2 + 2 * d + pd) * H2 + 2 * h + ph) * W2 + 2 * w0 + 2 * l15 + (lq >> 1)) * 16 + odd * 8;
      dl[i][q] = z4;
      if (ok[i]) dl[i][q] = *reinterpret_cast<const uint4*>(a.dy + rowb[i][q]);
    }
  }
  unsigned tp[TPW][2];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    f32x4 ct = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      U b;
      b.u = dl[i][q];
      const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ae.v, b.v, z, 0, 0, 0);
      const f32x4 c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao.v, b.v, z, 0, 0, 0);
      ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[q].v, b.v, ct, 0, 0, 0);
      // lane: channels lq*4..+3 of d prev at fine voxels 2j (c0) and 2j+1 (c1)
      const unsigned o00 = pack_bf2(c0[0], c0[1]), o01 = pack_bf2(c0[2], c0[3]);
      const unsigned o10 = pack_bf2(c1[0], c1[1]), o11 = pack_bf2(c1[2], c1[3]);
      const unsigned s0 = odd ? o00 : o10, s1 = odd ? o01 : o11;
      const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
      const uint4 out = odd ? make_uint4(r0, r1, o10, o11) : make_uint4(o00, o01, r0, r1);
      const size_t off = rowb[i][q] - (size_t)((lq >> 1) * 16 + odd * 8) + (size_t)(odd * 16 + (lq >> 1) * 8);
      if (ok[i]) *reinterpret_cast<uint4*>(a.dprev + off) = out;
    }
    tp[i][0] = pack_bf2(ct[0], ct[1]);
    tp[i][1] = pack_bf2(ct[2], ct[3]);
  }
  {
    // lane: channels lq*4..+3 of d t at coarse voxel j of both tiles; even lq keeps tile 0, odd lq tile 1
    const unsigned s0 = odd ? tp[0][0] : tp[1][0], s1 = odd ? tp[0][1] : tp[1][1];
    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
    const uint4 out = odd ? make_uint4(r0, r1, tp[1][0], tp[1][1]) : make_uint4(tp[0][0], tp[0][1], r0, r1);
    if (odd ? ok[1] : ok[0]) *reinterpret_cast<uint4*>(a.dt + (odd ? tb[1] : tb[0])) = out;
  }
}

// C = 32: K = 32 channels of one fine voxel per MFMA; two M tiles for d prev (W3a^T, shared by all parities) and two
// accumulating ones per parity for d t (`wq`).  Lane pairs swap one M tile as in the forward kernel.
__global__ void __launch_bounds__(256)
k_upcat32_bwd(UpcatBwdArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const bool odd = lq & 1;
  union U { uint4 u; bf16x8 v; };
  U wa[2], wm[8][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    wa[mt].u = *reinterpret_cast<const uint4*>(a.w3a_t + (size_t)(mt * 16 + l15) * 32 + lq * 8);
#pragma unroll
    for (int p = 0; p < 8; ++p)
      wm[p][mt].u = *reinterpret_cast<const uint4*>(a.wq + (size_t)((mt * 16 + l15) * 8 + p) * 32 + lq * 8);
  }
  const unsigned tile = (unsigned)blockIdx.x * 4u + (unsigned)wave;
  if (tile >= a.ntiles) return;                 // wave-uniform; no barriers in this kernel
  const int twi = (int)(tile % (unsigned)a.ntw);
  unsigned r = tile / (unsigned)a.ntw;
  const int h = (int)(r % (unsigned)a.H); r /= (unsigned)a.H;
  const int d = (int)(r % (unsigned)a.D);
  const int n = (int)(r / (unsigned)a.D);
  const int w0 = twi * 16;
  const int D2 = 2 * a.D, H2 = 2 * a.H, W2 = 2 * a.W;
  (void)D2;
  U pl[8];
  size_t vox[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int pd = p >> 2, ph = (p >> 1) & 1, pw = p & 1;
    vox[p] = (((size_t)n * D2 + 2 * d + pd) * H2 + 2 * h + ph) * W2 + 2 * (w0 + l15) + pw;
    pl[p].u = *reinterpret_cast<const uint4*>(a.dy + vox[p] * 32 + lq * 8);
  }
  const int choff = odd ? 16 + (lq - 1) * 4 : lq * 4;
  f32x4 ct[2];
  ct[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  ct[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    unsigned o[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[mt].v, pl[p].v, z, 0, 0, 0);
      ct[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[p][mt].v, pl[p].v, ct[mt], 0, 0, 0);
      o[mt][0] = pack_bf2(c[0], c[1]);
      o[mt][1] = pack_bf2(c[2], c[3]);
    }
    const unsigned s0 = odd ? o[0][0] : o[1][0], s1 = odd ? o[0][1] : o[1][1];
    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
    const uint4 out = odd ? make_uint4(r0, r1, o[1][0], o[1][1]) : make_uint4(o[0][0], o[0][1], r0, r1);
    *reinterpret_cast<uint4*>(a.dprev + vox[p] * 32 + choff) = out;
  }
  {
    unsigned o[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { o[mt][0] = pack_bf2(ct[mt][0], ct[mt][1]); o[mt][1] = pack_bf2(ct[mt][2], ct[mt][3]); }
    const unsigned s0 = odd ? o[0][0] : o[1][0], s1 = odd ? o[0][1] : o[1][1];
    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
    const uint4 out = odd ? make_uint4(r0, r1, o[1][0], o[1][1]) : make_uint4(o[0][0], o[0][1], r0, r1);
    *reinterpret_cast<uint4*>(a.dt + ((((size_t)n * a.D + d) * a.H + h) * a.W + w0 + l15) * 32 + choff) = out;
  }
}

// ---- composite weights of one DeUp_Cat (grid = 8 blocks, one per parity p; C <= 64).  fp32 masters in, kernel
// layouts out:
//   wc   bf16 [8][C][2C]: parity p = (pd*2+ph)*2+pw; even pw rows = [W3a | M[p]], odd pw rows = [M[p] | W3a]
//   bias fp32 [C]       : b3 + W3b bt
//   w3a  bf16 [C][C]    : W3a as a packed 1x1 conv weight [Cout][1][Cin];  w3a_t: its dgrad packing [Cin][1][Cout]
//   wq   bf16 [C][8][C] : wq[c][p][o] = M[p][o][c] - the k2-s2 'down conv' packing [Cout][T][Cin] of d t = sum_p M[p]^T dy
// The two C x C factors of the block's parity are staged in LDS (rows padded by one float: conflict-free).
__global__ void __launch_bounds__(256)
k_upcat_prepare(const float* __restrict__ w3, const float* __restrict__ b3, const float* __restrict__ wt,
                const float* __restrict__ bt, int C, bf16* __restrict__ wc, float* __restrict__ bias,
                bf16* __restrict__ w3a, bf16* __restrict__ w3a_t, bf16* __restrict__ wq) {
  __shared__ float sW[64 * 65], sT[64 * 65];       // W3b[o][m], Wt[.][.][p] as [c][m]
  const int K2 = 2 * C, p = blockIdx.x, LD = C + 1;
  for (int i0 = threadIdx.x; i0 < C * C; i0 += 256 * 4) {        // all loads of a round before the LDS stores
    float vw[4], vt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u, m = i % C, r = i / C;
      if (i < C * C) { vw[u] = w3[r * K2 + C + m]; vt[u] = wt[(r * C + m) * 8 + p]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u, m = i % C, r = i / C;
      if (i < C * C) { sW[r * LD + m] = vw[u]; sT[r * LD + m] = vt[u]; }
    }
  }
  __syncthreads();
  const bool oddw = p & 1;
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
    const int c = i % C, o = i / C;
    float m = 0.f;
    for (int k = 0; k < C; ++k) m = fmaf(sW[o * LD + k], sT[c * LD + k], m);
    const float a = w3[o * K2 + c];
    wc[(p * C + o) * K2 + (oddw ? 0 : C) + c] = (bf16)m;
    wc[(p * C + o) * K2 + (oddw ? C : 0) + c] = (bf16)a;
    wq[(c * 8 + p) * C + o] = (bf16)m;
    if (p == 0) { w3a[o * C + c] = (bf16)a; w3a_t[c * C + o] = (bf16)a; }
  }
  if (p == 0)
    for (int o = threadIdx.x; o < C; o += blockDim.x) {
      float b = b3[o];
      for (int k = 0; k < C; ++k) b = fmaf(sW[o * LD + k], bt[k], b);
      bias[o] = b;
    }
}

// ---- weight gradients of conv3 / conv2 from the fused backward's sums (fixed summation order: deterministic):
//   g [C(c)][C(o)][8] = sum_i t[i][c] dy[fine(i,p)][o]   (ConvT weight-gradient layout),  dw3a [C(o)][C(k)],  s [C]
// C = 16 or 32; grid = 8 + C*C/256 blocks.  Block p < 8: dWt[.][.][p] = W3b^T G[p] (both factors staged in LDS).
// Block 8 + b: R = 256/C rows of dW3 = [dW3a | sum_p G[p] Wt[p]^T + s bt^T], one thread per entry, Wt and the block's
// rows of G in LDS (rows padded by one float).  Block 8 also leaves db3 = s and dbt = W3b^T s.
__global__ void __launch_bounds__(256)
k_upcat_combine(const float* __restrict__ g, const float* __restrict__ dw3a, const float* __restrict__ s,
                const float* __restrict__ w3, const float* __restrict__ wt, const float* __restrict__ bt, int C,
                float* __restrict__ dw3, float* __restrict__ db3, float* __restrict__ dwt, float* __restrict__ dbt) {
  extern __shared__ float dyn[];
  const int K2 = 2 * C, LD = C + 1;
  if (blockIdx.x < 8) {
    float* sW = dyn;                 // W3b[o][m]
    float* sG = dyn + C * LD;        // G[.][.][p] as [c][o]
    const int p = blockIdx.x;
    {
      float vw[4], vg[4];                        // C * C / 256 = 1 or 4 elements per thread, all loads first
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + 256 * u, m = i % C, r = i / C;
        if (i < C * C) { vw[u] = w3[r * K2 + C + m]; vg[u] = g[(r * C + m) * 8 + p]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + 256 * u, m = i % C, r = i / C;
        if (i < C * C) { sW[r * LD + m] = vw[u]; sG[r * LD + m] = vg[u]; }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
      const int m = i % C, c = i / C;
      float v = 0.f;
      for (int o = 0; o < C; ++o) v = fmaf(sW[o * LD + m], sG[c * LD + o], v);
      dwt[(c * C + m) * 8 + p] = v;
    }
    return;
  }
  // row block: R = 256 / C rows of dW3; thread = (row ol, column m); both operands of the 8C-long dot products in LDS
  const int R = 256 / C, I = 8 * C, LDI = I + 1;
  const int o0 = ((int)blockIdx.x - 8) * R;
  float* sWt = dyn;                   // [m][i = c*8 + p] = Wt[c][m][p]
  float* sGr = dyn + C * LDI;         // [ol][i]          = G[c][o0 + ol][p]
  // eight loads in flight per thread (a load -> LDS-store loop left rolled pays one L2 latency per element)
  for (int L0 = threadIdx.x; L0 < C * I; L0 += 256 * 8) {       // C * I = 8 C^2 is a multiple of 2048
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = wt[L0 + 256 * u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int L = L0 + 256 * u, p = L & 7, m = (L >> 3) % C, c = (L >> 3) / C;
      sWt[m * LDI + c * 8 + p] = v[u];
    }
  }
  for (int L0 = threadIdx.x; L0 < R * I; L0 += 256 * 8) {       // R * I = 2048
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int L = L0 + 256 * u, p = L & 7, ol = (L >> 3) % R, c = (L >> 3) / R;
      v[u] = g[(c * C + o0 + ol) * 8 + p];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int L = L0 + 256 * u, p = L & 7, ol = (L >> 3) % R, c = (L >> 3) / R;
      sGr[ol * LDI + c * 8 + p] = v[u];
    }
  }
  __syncthreads();
  const int ol = threadIdx.x / C, m = threadIdx.x % C, o = o0 + ol;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const float* gr = sGr + ol * LDI;
  const float* wr = sWt + m * LDI;
  for (int i = 0; i < I; i += 4) {
    a0 = fmaf(gr[i], wr[i], a0); a1 = fmaf(gr[i + 1], wr[i + 1], a1);
    a2 = fmaf(gr[i + 2], wr[i + 2], a2); a3 = fmaf(gr[i + 3], wr[i + 3], a3);
  }
  dw3[o * K2 + C + m] = fmaf(s[o], bt[m], (a0 + a1) + (a2 + a3));
  dw3[o * K2 + m] = dw3a[o * C + m];
  if (blockIdx.x == 8 && threadIdx.x < C) {
    const int mm = threadIdx.x;
    db3[mm] = s[mm];
    float v = 0.f;
    for (int oo = 0; oo < C; ++oo) v = fmaf(w3[oo * K2 + C + mm], s[oo], v);
    dbt[mm] = v;
  }
}

}  // namespace devis

using namespace devis;

extern "C" {

int devis_upcat_supported(int64_t N, int64_t D, int64_t H, int64_t W, int64_t C, int dtype) {
  if (dtype != DEVIS_BF16 || (C != 16 && C != 32) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || W % 16) return 0;
  return N * D * H * W * 8 < ((int64_t)1 << 31) ? 1 : 0;
}

int devis_upcat_prepare(const float* w3, const float* b3, const float* wt, const float* bt, int64_t C, void* wc,
                        float* bias, void* w3a, void* w3a_t, void* wq, void* stream) {
  if (!w3 || !b3 || !wt || !bt || !wc || !bias || !w3a || !w3a_t || !wq || C <= 0 || C > 64) return DEVIS_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_upcat_prepare, dim3(8), dim3(256), 0, (hipStream_t)stream, w3, b3, wt, bt, (int)C, (bf16*)wc, bias,
                     (bf16*)w3a, (bf16*)w3a_t, (bf16*)wq);
  return launch_status();
}

int devis_upcat_fwd(const void* prev, const void* t, const void* wc, const float* bias, void* y, int64_t N, int64_t D,
                    int64_t H, int64_t W, int64_t C, int dtype, void* stream) {
  if (!prev || !t || !wc || !bias || !y) return DEVIS_ERR_BAD_ARG;
  if (!devis_upcat_supported(N, D, H, W, C, dtype)) return DEVIS_ERR_UNSUPPORTED;
  UpcatArgs a;
  a.prev = (const bf16*)prev; a.t = (const bf16*)t; a.wc = (const bf16*)wc; a.bias = bias; a.y = (bf16*)y;
  a.N = (int)N; a.D = (int)D; a.H = (int)H; a.W = (int)W;
  a.ntw = (int)(W / 16);
  a.ntiles = (unsigned)(N * D * H * a.ntw);
  constexpr int TPW = 2;
  if (C == 16) hipLaunchKernelGGL((k_upcat16_fwd<TPW>), dim3((unsigned)cdiv(a.ntiles, 4 * TPW)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_upcat32_fwd, dim3((unsigned)cdiv(a.ntiles, 4)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int devis_upcat_bwd_data(const void* dy, const void* w3a_t, const void* wq, void* dprev, void* dt, int64_t N, int64_t D,
                         int64_t H, int64_t W, int64_t C, int dtype, void* stream) {
  if (!dy || !w3a_t || !wq || !dprev || !dt) return DEVIS_ERR_BAD_ARG;
  if (!devis_upcat_supported(N, D, H, W, C, dtype)) return DEVIS_ERR_UNSUPPORTED;
  UpcatBwdArgs a;
  a.dy = (const bf16*)dy; a.w3a_t = (const bf16*)w3a_t; a.wq = (const bf16*)wq; a.dprev = (bf16*)dprev; a.dt = (bf16*)dt;
  a.N = (int)N; a.D = (int)D; a.H = (int)H; a.W = (int)W;
  a.ntw = (int)(W / 16);
  a.ntiles = (unsigned)(N * D * H * a.ntw);
  if (C == 16) hipLaunchKernelGGL(k_upcat16_bwd, dim3((unsigned)cdiv(a.ntiles, 8)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_upcat32_bwd, dim3((unsigned)cdiv(a.ntiles, 4)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int devis_upcat_grad_combine(const float* g, const float* dw3a, const float* s, const float* w3, const float* wt,
                             const float* bt, int64_t C, float* dw3, float* db3, float* dwt, float* dbt, void* stream) {
  if (!g || !dw3a || !s || !w3 || !wt || !bt || !dw3 || !db3 || !dwt || !dbt) return DEVIS_ERR_BAD_ARG;
  if (C != 16 && C != 32) return DEVIS_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(C + 256 / C) * (8 * C + 1) * sizeof(float);     // >= the parity blocks' 2 C (C+1) floats
  hipLaunchKernelGGL(k_upcat_combine, dim3((unsigned)(8 + C * C / 256)), dim3(256), lds, (hipStream_t)stream, g, dw3a, s, w3, wt,
                     bt, (int)C, dw3, db3, dwt, dbt);
  return launch_status();
}

}  // extern "C"
