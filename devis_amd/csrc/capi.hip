// C-ABI entry points for convolution: validate, then route to the MFMA implicit-GEMM
// kernels (conv_mfma.hip) when they take the shape, else to the direct kernels.
#include "common.h"
#include "mfma_common.h"

#include <stdlib.h>

namespace devis {
// conv_direct.hip
int conv_fwd_direct(const devis_conv_desc*, const void*, const void*, const void*, const float*, const float*,
                    const float*, const void*, void*, hipStream_t);
int conv_dgrad_direct(const devis_conv_desc*, const void*, const void*, const void*, void*, void*, hipStream_t);
// conv_f32_mfma.hip: fp32 on the matrix cores (any kernel size, stride 1 / 2, channel counts multiples of 16)
int conv_fwd_f32_mfma(const devis_conv_desc*, const void*, const void*, const void*, const float*, const float*, const float*,
                      const void*, void*, hipStream_t);
int conv_dgrad_f32_mfma(const devis_conv_desc*, const void*, const void*, const void*, void*, void*, hipStream_t);
size_t conv_wgrad_f32_workspace(const devis_conv_desc*);
int conv_wgrad_f32_mfma(const devis_conv_desc*, const void*, const void*, const float*, const float*, const void*, float*, float*,
                        void*, size_t, hipStream_t);
int conv_wgrad_direct(const devis_conv_desc*, const void*, const void*, const float*, const float*, const void*,
                      float*, float*, hipStream_t);
// conv_mfma.hip: return DEVIS_ERR_UNSUPPORTED when the shape is not theirs
int conv_fwd_mfma(const devis_conv_desc*, const void*, const void*, const void*, const float*, const float*,
                  const float*, const void*, void*, const StatsArgs*, int64_t*, hipStream_t);
int conv_dgrad_mfma(const devis_conv_desc*, const void*, const void*, const void*, void*, void*, const StatsArgs*,
                    int64_t*, hipStream_t);
// conv2d_mfma.hip: the 2-D U-Nets' 3x3 layers
int conv2d_fwd_mfma(const devis_conv_desc*, const void*, const void*, const void*, const float*, const float*,
                    const float*, const void*, void*, const StatsArgs*, int64_t*, hipStream_t);
int conv2d_dgrad_mfma(const devis_conv_desc*, const void*, const void*, const void*, void*, void*, const StatsArgs*,
                      int64_t*, hipStream_t);
size_t conv2d_wgrad_mfma_workspace(const devis_conv_desc*);
int conv2d_wgrad_mfma(const devis_conv_desc*, const void*, const void*, const float*, const float*, const void*,
                      float*, float*, void*, size_t, hipStream_t);
// pointwise_mfma.hip
int conv1_fwd_mfma(const devis_conv_desc*, const void*, const void*, const void*, const float*, const void*, void*, hipStream_t);
int conv1_dgrad_mfma(const devis_conv_desc*, const void*, const void*, const void*, void*, void*, const StatsArgs*, int64_t*,
                     hipStream_t);
// the V-Net's k2-s2 down conv on the gather-GEMM kernels
int conv2s2_fwd_mfma(const devis_conv_desc*, const void*, const void*, const float*, const void*, void*, hipStream_t);
int conv2s2_dgrad_mfma(const devis_conv_desc*, const void*, const void*, const void*, void*, hipStream_t);
size_t conv2s2_wgrad_mfma_workspace(const devis_conv_desc*);
int conv2s2_wgrad_mfma(const devis_conv_desc*, const void*, const void*, float*, float*, void*, size_t, hipStream_t);
// conv_thin.hip: the 2-D nets' first layer (1 or 3 input channels)
int first2d_fwd(const devis_conv_desc*, const void*, const void*, const float*, const void*, void*, hipStream_t);
size_t first2d_wgrad_workspace(const devis_conv_desc*);
int first2d_wgrad(const devis_conv_desc*, const void*, const void*, float*, float*, void*, size_t, hipStream_t);
// head1x1.hip: the 16 -> 4 endconv (and the 2-D nets' 32 -> 2 conv10)
int head_fwd(const devis_conv_desc*, const void*, const void*, const float*, const void*, void*, hipStream_t);
int head_dgrad(const devis_conv_desc*, const void*, const void*, const void*, void*, const StatsArgs*, int64_t*, hipStream_t);
int conv3_c16_march_nb(const devis_conv_desc*, const void*, const void*, const float*, const float*, const float*, const float*,
                       const float*, const void*, const void*, void*, void*, const StatsArgs*, hipStream_t);
int conv3s2_dgrad_mfma(const devis_conv_desc*, const void*, const void*, const void*, void*, hipStream_t);
// conv_s2_dgrad.hip: the same layers with the coarse tile and the weights staged in LDS
int conv3s2_dgrad_lds(const devis_conv_desc*, const void*, const void*, const void*, void*, hipStream_t);
size_t conv1_wgrad_mfma_workspace(const devis_conv_desc*);
int conv1_wgrad_mfma(const devis_conv_desc*, const void*, const void*, const void*, float*, float*, void*, size_t, hipStream_t);
size_t conv_wgrad_mfma_workspace(const devis_conv_desc*);
int conv_wgrad_mfma(const devis_conv_desc*, const void*, const void*, const float*, const float*, const void*,
                    float*, float*, void*, size_t, hipStream_t);

// bf16 weights of a 27-tap layer with Cin == Cout in {64, 128} are packed in fragment order (csrc/layout.hip), which
// only conv3_deepk reads: such a layer must never reach a kernel that expects the canonical [Cout][T][Cin] array
static bool deepk_weights(const devis_conv_desc* d) {
  return d->dtype == DEVIS_BF16 && d->KD * d->KH * d->KW == 27 && d->C2 == 0 && d->C1 == d->Cout && (d->C1 == 64 || d->C1 == 128);
}

extern thread_local bool g_wgrad_stage1_only;   // conv_wgrad_mfma.hip

static bool conv_desc_ok(const devis_conv_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->ID <= 0 || d->IH <= 0 || d->IW <= 0 || d->C1 <= 0 || d->C2 < 0 || d->Cout <= 0) return false;
  if (d->KD <= 0 || d->KH <= 0 || d->KW <= 0 || d->SD <= 0 || d->SH <= 0 || d->SW <= 0) return false;
  if (d->PD < 0 || d->PH < 0 || d->PW < 0) return false;
  if (d->dtype != DEVIS_F32 && d->dtype != DEVIS_BF16) return false;
  // the output size must be what the reference's conv produces for this input
  if (d->OD != (d->ID + 2 * d->PD - d->KD) / d->SD + 1) return false;
  if (d->OH != (d->IH + 2 * d->PH - d->KH) / d->SH + 1) return false;
  if (d->OW != (d->IW + 2 * d->PW - d->KW) / d->SW + 1) return false;
  return d->OD > 0 && d->OH > 0 && d->OW > 0;
}
}  // namespace devis

using namespace devis;

extern "C" {

// validated copy of the caller's statistics request (mode 0 when there is none)
static int stats_args(const devis_conv_stats* st, int64_t C, int64_t N, StatsArgs* out) {
  *out = StatsArgs{};
  if (!st || st->mode == 0) return DEVIS_OK;
  if (st->mode != 1 && st->mode != 2) return DEVIS_ERR_BAD_ARG;
  if (!st->partials || st->chunks <= 0 || st->chunks > 0x7fffffff) return DEVIS_ERR_BAD_ARG;
  if (st->mode == 2 && (!st->x_ref || !st->scale || !st->shift)) return DEVIS_ERR_BAD_ARG;
  if (st->partials_bytes < (size_t)(N * st->chunks * C * 2) * sizeof(double)) return DEVIS_ERR_WORKSPACE;
  out->mode = st->mode; out->chunks = (int)st->chunks; out->C = (int)C; out->partial = st->partials;
  out->xref = (const bf16*)st->x_ref; out->sc = st->scale; out->sh = st->shift;
  return DEVIS_OK;
}

int64_t devis_conv_stats_chunks(const devis_conv_desc* d, int dgrad, int mode, int has_prologue, int has_residual) {
  if (!conv_desc_ok(d) || (mode != 1 && mode != 2)) return 0;
  // placeholders: plan-only calls return before anything is dereferenced or launched
  const void* ph = reinterpret_cast<const void*>((uintptr_t)256);
  const float* phf = reinterpret_cast<const float*>((uintptr_t)256);
  StatsArgs st = StatsArgs{};
  st.mode = mode;
  int64_t chunks = 0;
  int rc;
  if (dgrad) {
    if (has_prologue) return 0;
    rc = conv_dgrad_mfma(d, ph, ph, has_residual ? ph : nullptr, const_cast<void*>(ph), nullptr, &st, &chunks, nullptr);
    if (rc == DEVIS_ERR_UNSUPPORTED)
      rc = conv2d_dgrad_mfma(d, ph, ph, has_residual ? ph : nullptr, const_cast<void*>(ph),
                             d->C2 ? const_cast<void*>(ph) : nullptr, &st, &chunks, nullptr);
    if (rc == DEVIS_ERR_UNSUPPORTED)
      rc = head_dgrad(d, ph, ph, has_residual ? ph : nullptr, const_cast<void*>(ph), &st, &chunks, nullptr);
    if (rc == DEVIS_ERR_UNSUPPORTED && d->C2 == 0)      // 1x1 layers: the gather-GEMM kernel
      rc = conv1_dgrad_mfma(d, ph, ph, has_residual ? ph : nullptr, const_cast<void*>(ph), nullptr, &st, &chunks, nullptr);
  } else {
    rc = conv_fwd_mfma(d, ph, nullptr, ph, phf, has_prologue ? phf : nullptr, has_prologue ? phf : nullptr,
                       has_residual ? ph : nullptr, const_cast<void*>(ph), &st, &chunks, nullptr);
    if (rc == DEVIS_ERR_UNSUPPORTED)
      rc = conv2d_fwd_mfma(d, ph, d->C2 ? ph : nullptr, ph, phf, has_prologue ? phf : nullptr, has_prologue ? phf : nullptr,
                           has_residual ? ph : nullptr, const_cast<void*>(ph), &st, &chunks, nullptr);
  }
  return rc == DEVIS_OK ? chunks : 0;
}

int devis_conv_fwd_stats(const devis_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                         const float* in_scale, const float* in_shift, const void* residual, void* y,
                         const devis_conv_stats* stats, void* stream) {
  if (!conv_desc_ok(d) || !x1 || !w || !y) return DEVIS_ERR_BAD_ARG;
  if ((d->C2 > 0) != (x2 != nullptr)) return DEVIS_ERR_BAD_ARG;
  if ((in_scale != nullptr) != (in_shift != nullptr)) return DEVIS_ERR_BAD_ARG;
  StatsArgs st;
  int rc = stats_args(stats, d->Cout, d->N, &st);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  rc = conv_fwd_mfma(d, x1, x2, w, bias, in_scale, in_shift, residual, y, &st, nullptr, s);
  if (rc == DEVIS_ERR_UNSUPPORTED && deepk_weights(d)) return rc;
  if (rc == DEVIS_ERR_UNSUPPORTED) rc = conv2d_fwd_mfma(d, x1, x2, w, bias, in_scale, in_shift, residual, y, &st, nullptr, s);
  if (rc != DEVIS_ERR_UNSUPPORTED || st.mode) return rc;      // statistics only come from the MFMA kernels
  if (!in_scale) {
    rc = head_fwd(d, x1, w, bias, residual, y, s);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
    rc = conv1_fwd_mfma(d, x1, x2, w, bias, residual, y, s);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
    rc = conv2s2_fwd_mfma(d, x1, w, bias, residual, y, s);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
    rc = first2d_fwd(d, x1, w, bias, residual, y, s);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  }
  rc = conv_fwd_f32_mfma(d, x1, x2, w, bias, in_scale, in_shift, residual, y, s);     // the fp32 parity mode
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  return conv_fwd_direct(d, x1, x2, w, bias, in_scale, in_shift, residual, y, s);
}

int devis_conv_fwd(const devis_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                   const float* in_scale, const float* in_shift, const void* residual, void* y, void* stream) {
  return devis_conv_fwd_stats(d, x1, x2, w, bias, in_scale, in_shift, residual, y, nullptr, stream);
}

int devis_conv_dgrad_stats(const devis_conv_desc* d, const void* dy, const void* w, const void* w_dgrad,
                           const void* residual, void* dx1, void* dx2, const devis_conv_stats* stats, void* stream) {
  if (!conv_desc_ok(d) || !dy || !w || !dx1) return DEVIS_ERR_BAD_ARG;
  if ((d->C2 > 0) != (dx2 != nullptr)) return DEVIS_ERR_BAD_ARG;
  if (residual && d->C2 > 0) return DEVIS_ERR_BAD_ARG;
  StatsArgs st;
  int rc = stats_args(stats, d->C1, d->N, &st);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  rc = conv_dgrad_mfma(d, dy, w_dgrad, residual, dx1, dx2, &st, nullptr, s);
  if (rc != DEVIS_ERR_UNSUPPORTED || deepk_weights(d)) return rc;
  rc = conv2d_dgrad_mfma(d, dy, w_dgrad, residual, dx1, dx2, &st, nullptr, s);
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  rc = head_dgrad(d, dy, w, residual, dx1, &st, nullptr, s);
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  rc = conv1_dgrad_mfma(d, dy, w_dgrad, residual, dx1, dx2, &st, nullptr, s);
  if (rc != DEVIS_ERR_UNSUPPORTED || st.mode) return rc;     // statistics only come from the MFMA kernels
  rc = conv3s2_dgrad_lds(d, dy, w_dgrad, residual, dx1, s);
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  rc = conv3s2_dgrad_mfma(d, dy, w_dgrad, residual, dx1, s);
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  rc = conv2s2_dgrad_mfma(d, dy, w_dgrad, residual, dx1, s);
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  rc = conv_dgrad_f32_mfma(d, dy, w_dgrad, residual, dx1, dx2, s);                     // the fp32 parity mode
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  return conv_dgrad_direct(d, dy, w, residual, dx1, dx2, s);
}

int devis_conv_dgrad_normbwd(const devis_conv_desc* d, const void* g, const void* xn, const float* scale, const float* shift,
                             const float* cA, const float* cB, const float* cC, const void* w_dgrad, const void* residual,
                             void* dx, void* dh_out, const devis_conv_stats* stats, void* stream) {
  if (!conv_desc_ok(d) || !g || !xn || !scale || !shift || !cA || !cB || !cC || !w_dgrad || !dx) return DEVIS_ERR_BAD_ARG;
  if (d->C2 != 0 || d->dtype != DEVIS_BF16 || d->C1 != d->Cout) return DEVIS_ERR_UNSUPPORTED;
  StatsArgs st;
  int rc = stats_args(stats, d->C1, d->N, &st);
  if (rc) return rc;
  // the dgrad is the forward kernel on the tap-flipped weights with the channel roles swapped (conv_dgrad_mfma)
  devis_conv_desc t = *d;
  t.C1 = d->Cout; t.Cout = d->C1;
  t.ID = d->OD; t.IH = d->OH; t.IW = d->OW; t.OD = d->ID; t.OH = d->IH; t.OW = d->IW;
  return conv3_c16_march_nb(&t, g, xn, scale, shift, cA, cB, cC, w_dgrad, residual, dx, dh_out, &st, (hipStream_t)stream);
}

int devis_conv_dgrad(const devis_conv_desc* d, const void* dy, const void* w, const void* w_dgrad,
                     const void* residual, void* dx1, void* dx2, void* stream) {
  return devis_conv_dgrad_stats(d, dy, w, w_dgrad, residual, dx1, dx2, nullptr, stream);
}

size_t devis_conv_wgrad_workspace_bytes(const devis_conv_desc* d) {
  if (!conv_desc_ok(d)) return 0;
  const size_t v[6] = {conv_wgrad_mfma_workspace(d), conv1_wgrad_mfma_workspace(d), conv2s2_wgrad_mfma_workspace(d),
                       conv2d_wgrad_mfma_workspace(d), first2d_wgrad_workspace(d), conv_wgrad_f32_workspace(d)};
  size_t m = 0;
  for (size_t b : v) m = b > m ? b : m;
  return m;
}

int devis_conv_wgrad(const devis_conv_desc* d, const void* x1, const void* x2, const float* in_scale,
                     const float* in_shift, const void* dy, float* dw, float* dbias, void* ws, size_t ws_bytes,
                     void* stream) {
  if (!conv_desc_ok(d) || !x1 || !dy || !dw) return DEVIS_ERR_BAD_ARG;
  if ((d->C2 > 0) != (x2 != nullptr)) return DEVIS_ERR_BAD_ARG;
  if ((in_scale != nullptr) != (in_shift != nullptr)) return DEVIS_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  int rc = conv_wgrad_mfma(d, x1, x2, in_scale, in_shift, dy, dw, dbias, ws, ws_bytes, s);
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  rc = conv2d_wgrad_mfma(d, x1, x2, in_scale, in_shift, dy, dw, dbias, ws, ws_bytes, s);
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  if (!in_scale) {
    rc = conv1_wgrad_mfma(d, x1, x2, dy, dw, dbias, ws, ws_bytes, s);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
    rc = conv2s2_wgrad_mfma(d, x1, dy, dw, dbias, ws, ws_bytes, s);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
    rc = first2d_wgrad(d, x1, dy, dw, dbias, ws, ws_bytes, s);
    if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  }
  rc = conv_wgrad_f32_mfma(d, x1, x2, in_scale, in_shift, dy, dw, dbias, ws, ws_bytes, s);   // the fp32 parity mode
  if (rc != DEVIS_ERR_UNSUPPORTED) return rc;
  return conv_wgrad_direct(d, x1, x2, in_scale, in_shift, dy, dw, dbias, s);
}

int devis_conv_wgrad_stage1(const devis_conv_desc* d, const void* x1, const float* in_scale, const float* in_shift,
                            const void* dy, int want_bias, void* ws, size_t ws_bytes, void* stream) {
  if (!conv_desc_ok(d) || !x1 || !dy || !ws || d->C2 != 0) return DEVIS_ERR_BAD_ARG;
  if ((in_scale != nullptr) != (in_shift != nullptr)) return DEVIS_ERR_BAD_ARG;
  // the partial-sum kernels never touch dw / dbias (only their second stage does): the workspace stands in for them
  g_wgrad_stage1_only = true;
  const int rc = conv_wgrad_mfma(d, x1, nullptr, in_scale, in_shift, dy, (float*)ws, want_bias ? (float*)ws : nullptr, ws, ws_bytes,
                                 (hipStream_t)stream);
  g_wgrad_stage1_only = false;
  return rc;
}

}  // extern "C"
