/* devis_hip.h — C ABI of libdevis_hip.so, the MI355X (gfx950) kernels behind the
 * DEviS evidential U-Net hot path.
 *
 * The reference (Cocofeat/DEviS) has no FFI: its hot path is a chain of implicit
 * ATen/cuDNN launches issued from Python.  Each entry point
 * below replaces the ATen op(s) named in its comment, cited as
 * `file:line` relative to the reference root.  The Python host side
 * (devis_amd/) binds these with ctypes inside torch.autograd.Function.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *  - activations are channels-last: [N][D][H][W][C] (2-D: D = 1), element type
 *    selected by `dtype` (DEVIS_F32 / DEVIS_BF16); statistics, biases, norm
 *    coefficients and parameter gradients are always fp32 (moments fp64);
 *  - packed conv weights are [Cout][KD*KH*KW][Cin] in `dtype`
 *    (devis_pack_conv_weight); parameter GRADIENTS are written in the
 *    reference/PyTorch layout, fp32, so they can be handed to torch.optim.Adam;
 *  - the caller owns every buffer including workspaces (size from the matching
 *    *_workspace_bytes); the library never allocates, frees or synchronises;
 *  - `stream` is a hipStream_t (0 = default stream);
 *  - return value: DEVIS_OK or a negative devis_status; nothing throws.
 */
#ifndef DEVIS_HIP_H
#define DEVIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum devis_status {
  DEVIS_OK = 0,
  DEVIS_ERR_BAD_ARG = -1,
  DEVIS_ERR_UNSUPPORTED = -2,
  DEVIS_ERR_HIP = -3,
  DEVIS_ERR_WORKSPACE = -4,
  DEVIS_ERR_SCRATCH_LIMIT = -5 /* input exceeds the fixed scratch of a no-allocation entry point */
};
enum devis_dtype { DEVIS_F32 = 0, DEVIS_BF16 = 1 };
enum devis_label_dtype { DEVIS_LABEL_I64 = 0, DEVIS_LABEL_U8 = 1, DEVIS_LABEL_I32 = 2 };

int devis_version(void);
/* last hipError_t seen by this thread (0 = hipSuccess) and its string */
int devis_last_hip_error(void);
const char* devis_last_hip_error_string(void);

/* ---------------------------------------------------------------- layout -- */
/* NC(D)HW fp32 -> channels-last `dtype`, and back.  Replaces the implicit
 * layout of every reference tensor ("fp32 contiguous NCDHW"). */
int devis_ncx_to_nxc(const float* src, void* dst, int64_t N, int64_t C, int64_t V, int dtype, void* stream);
int devis_nxc_to_ncx(const void* src, float* dst, int64_t N, int64_t C, int64_t V, int dtype, void* stream);
/* nn.Conv{2,3}d.weight [Cout][Cin][T] fp32 -> [Cout][T][Cin] dtype.
 * One exception, the same in every packing call of this header (devis_pack_conv_weight, devis_pack_weights_batch,
 * devis_repack_conv_weight_dgrad): a bf16 weight with T = 27 and Cin == Cout in {64, 128} - the deep 3x3x3 layers
 * EnBlock3_x / EnBlock4_x / DeBlock4 of UNet3DZoo.py:35-56,145-165 - is written in FRAGMENT ORDER, the order in which
 * the kernel of those layers (csrc/conv_deepk.hip) keeps a wave's weights in registers: element (row, tap, col) of
 * the [rows][T][cols] array sits at
 *   (((((row/32)*(C/32) + col/32)*27 + tap)*2 + (row/16)%2)*64 + ((col/8)%4)*16 + row%16)*8 + col%8 .
 * Such a buffer is only meaningful to devis_conv_fwd* / devis_conv_dgrad* of that same layer; a caller that builds
 * packed weights itself must go through these packing calls.  fp32 weights are never permuted. */
int devis_pack_conv_weight(const float* w, void* dst, int64_t Cout, int64_t Cin, int64_t T, int dtype, void* stream);
/* nn.ConvTranspose{2,3}d.weight [Cin][Cout][T] fp32 -> [T][Cout][Cin] dtype */
int devis_pack_convT_weight(const float* w, void* dst, int64_t Cin, int64_t Cout, int64_t T, int dtype, void* stream);

/* Every weight of a model packed by ONE launch (the per-layer calls above cost a launch each, ~90 per
 * step for the 3-D U-Net).  `items_dev` (device memory) lists the tensors; `chunks_dev` (device memory,
 * n_chunks pairs of int32) assigns each block one item and the first of its DEVIS_PACK_CHUNK output
 * elements.  Both tables stay valid while the parameters keep their addresses, so the caller builds them once.
 *   DEVIS_PACK_CONV        src [rows=Cout][cols=Cin][taps]  -> dst [Cout][taps][Cin]   (devis_pack_conv_weight)
 *   DEVIS_PACK_CONV_DGRAD  same src                         -> dst [Cin][taps][Cout], taps reversed
 *                                                              (= devis_repack_conv_weight_dgrad of the above)
 *   DEVIS_PACK_CONVT       src [rows=Cin][cols=Cout][taps]  -> dst [taps][Cout][Cin]   (devis_pack_convT_weight)
 *   DEVIS_PACK_CONVT_DGRAD same src                         -> dst [Cin][taps][Cout]   (w_t of devis_convT_dgrad) */
enum { DEVIS_PACK_CONV = 0, DEVIS_PACK_CONV_DGRAD = 1, DEVIS_PACK_CONVT = 2, DEVIS_PACK_CONVT_DGRAD = 3 };
#define DEVIS_PACK_CHUNK 1024
/* Row chunks of the two conv modes: chunk = (item | DEVIS_PACK_ROW_FLAG, row): the block packs the whole output row
 * `row` (a cout of DEVIS_PACK_CONV, a cin of DEVIS_PACK_CONV_DGRAD) through LDS with coalesced reads; allowed when a row
 * (cols * taps, resp. rows * taps elements) has at most DEVIS_PACK_ROW_MAX elements.  Element chunks stay valid for
 * every mode. */
#define DEVIS_PACK_ROW_FLAG 0x40000000
#define DEVIS_PACK_ROW_MAX 8192
typedef struct devis_pack_item {
  const float* src;
  void* dst;
  int32_t rows, cols, taps, mode;
} devis_pack_item;
int devis_pack_weights_batch(const devis_pack_item* items_dev, const int32_t* chunks_dev, int64_t n_chunks,
                             int dtype, void* stream);

/* ------------------------------------------------------------------ conv -- */
typedef struct devis_conv_desc {
  int64_t N;
  int64_t ID, IH, IW;  /* input spatial size */
  int64_t OD, OH, OW;  /* output spatial size */
  int64_t C1, C2;      /* input channels taken from x1 and x2 (virtual concat; C2 = 0: none) */
  int64_t Cout;
  int32_t KD, KH, KW;
  int32_t SD, SH, SW;
  int32_t PD, PH, PW;
  int32_t dtype;
} devis_conv_desc;

/* y = conv(relu(in_scale*x + in_shift)) + bias (+ residual).
 * in_scale/in_shift: [N][C1+C2] fp32 or NULL (no prologue).  Zero padding is
 * applied AFTER the prologue, as in the reference where the norm+ReLU output
 * is what gets padded.
 * Replaces nn.Conv3d / nn.Conv2d forward (UNet3DZoo.py:25,41,45,62,120,122,151,152,274;
 * UNet2DZoo.py:12,15,128) fused with the preceding GroupNorm/BatchNorm apply +
 * ReLU (UNet3DZoo.py:48-53,158-162), torch.cat (UNet3DZoo.py:128; UNet2DZoo.py:153,156,181)
 * and the residual add (UNet3DZoo.py:54). */
int devis_conv_fwd(const devis_conv_desc* d, const void* x1, const void* x2, const void* w_packed,
                   const float* bias, const float* in_scale, const float* in_shift,
                   const void* residual, void* y, void* stream);
/* dA = conv_transpose(dy, w): gradient w.r.t. the conv's (post-prologue) input,
 * split over the two virtual-concat inputs (dx2 may be NULL when C2 = 0).
 * Replaces cuDNN dgrad of the same layers. */
/* w_dgrad: optional [Cin][T][Cout] tap-flipped packing (devis_repack_conv_weight_dgrad) that lets the
 * stride-1 3x3x3 layers run dgrad on the MFMA forward kernel; residual: optional tensor added to dx1
 * (skip-connection gradient, only when C2 == 0). */
int devis_repack_conv_weight_dgrad(const void* w_packed, void* dst, int64_t Cout, int64_t Cin, int64_t T,
                                   int dtype, void* stream);
int devis_conv_dgrad(const devis_conv_desc* d, const void* dy, const void* w_packed, const void* w_dgrad,
                     const void* residual, void* dx1, void* dx2, void* stream);
/* Normalisation statistics gathered in the conv epilogue: the conv that PRODUCES a tensor also leaves the
 * per-channel moments the following nn.GroupNorm / nn.BatchNorm3d needs (UNet3DZoo.py:39-52,149-162), so the
 * tensor is not read a second time.  partials = [N][chunks][C][2] fp64, the layout of
 * devis_channel_moments_partials; feed it to devis_{gn,bn,norm_bwd}_finalize_partials.
 *   mode 1 (forward):  {sum y, sum y^2} of the stored output y.
 *   mode 2 (dgrad):    dx1 is the gradient w.r.t. relu(scale*x_ref + shift);
 *                      g = dx1*[scale*x_ref+shift > 0] -> {sum g, sum g*x_ref}  (autograd of norm+ReLU).
 * devis_conv_stats_chunks: chunks per sample the kernel for this shape leaves, or 0 when that kernel cannot
 * gather statistics (fp32, shapes served by the direct kernels): then run the conv without statistics and
 * use devis_channel_moments_partials / devis_norm_bwd_moments_partials.  chunks must be passed back unchanged. */
typedef struct devis_conv_stats {
  int32_t mode;            /* 0 none, 1, 2 */
  int32_t reserved;
  double* partials;
  size_t partials_bytes;   /* >= N*chunks*C*2*8 */
  int64_t chunks;
  const void* x_ref;       /* mode 2: the normalised tensor, shape of dx1 */
  const float* scale;      /* mode 2: [N][C] */
  const float* shift;
} devis_conv_stats;
int64_t devis_conv_stats_chunks(const devis_conv_desc* d, int dgrad, int mode, int has_prologue, int has_residual);
int devis_conv_fwd_stats(const devis_conv_desc* d, const void* x1, const void* x2, const void* w_packed,
                         const float* bias, const float* in_scale, const float* in_shift, const void* residual,
                         void* y, const devis_conv_stats* stats, void* stream);
int devis_conv_dgrad_stats(const devis_conv_desc* d, const void* dy, const void* w_packed, const void* w_dgrad,
                           const void* residual, void* dx1, void* dx2, const devis_conv_stats* stats, void* stream);
/* devis_norm_bwd_apply (below) fused into the dgrad that consumes its result: the gradient that enters this conv's
 * dgrad is dh = cA * g * [scale * xn + shift > 0] + cB * xn + cC - autograd's backward of the norm + ReLU in front of the
 * conv (UNet3DZoo.py:12-15,44-56,149-165) - formed from g and xn while the tile is staged, with devis_norm_bwd_apply's
 * arithmetic and rounding (same bits as the two calls).  dh_out (may be NULL) receives dh once: the layer's weight
 * gradient reads it.  g, xn, dh_out: the conv OUTPUT's shape; scale .. cC: [N][Cout] (devis_norm_bwd_finalize*);
 * residual / dx / stats as devis_conv_dgrad_stats (a residual and statistics exclude each other).  bf16, 3x3x3 stride 1,
 * 16 -> 16 channels, W > 16 only: DEVIS_ERR_UNSUPPORTED otherwise (run the two calls). */
int devis_conv_dgrad_normbwd(const devis_conv_desc* d, const void* g, const void* xn, const float* scale,
                             const float* shift, const float* cA, const float* cB, const float* cC, const void* w_dgrad,
                             const void* residual, void* dx, void* dh_out, const devis_conv_stats* stats, void* stream);
/* dw [Cout][C1+C2][T] fp32 (PyTorch layout) and dbias [Cout] fp32 (may be NULL).
 * The conv input is recomputed from x1/x2 with the same prologue. */
size_t devis_conv_wgrad_workspace_bytes(const devis_conv_desc* d);
int devis_conv_wgrad(const devis_conv_desc* d, const void* x1, const void* x2,
                     const float* in_scale, const float* in_shift, const void* dy,
                     float* dw, float* dbias, void* workspace, size_t workspace_bytes, void* stream);
/* The first stage of devis_conv_wgrad alone, for profiling (no counterpart in the reference): the bf16 MFMA
 * weight-gradient kernels leave per-block fp32 partial sums in `workspace` (internal layout, sized by
 * devis_conv_wgrad_workspace_bytes) and the fixed-order reduction that would turn them into dw / dbias is NOT run.
 * bench.py times the partial-sum kernel by itself with it (roofline line).  One input tensor (C2 = 0);
 * DEVIS_ERR_UNSUPPORTED for layers the MFMA weight-gradient kernels do not take. */
int devis_conv_wgrad_stage1(const devis_conv_desc* d, const void* x1, const float* in_scale, const float* in_shift,
                            const void* dy, int want_bias, void* workspace, size_t workspace_bytes, void* stream);

/* ConvTranspose with kernel == stride in every dim (k=2,s=2; 2-D: KD=1).
 * Replaces nn.ConvTranspose3d (UNet3DZoo.py:121) / nn.ConvTranspose2d (UNet2DZoo.py:120-126).
 * w_packed: [T][Cout][Cin] (devis_pack_convT_weight); dw: [Cin][Cout][T] fp32. */
typedef struct devis_convT_desc {
  int64_t N;
  int64_t ID, IH, IW;
  int64_t Cin, Cout;
  int32_t KD, KH, KW; /* each 1 or 2; output size = input size * K */
  int32_t dtype;
} devis_convT_desc;
int devis_convT_fwd(const devis_convT_desc* d, const void* x, const void* w_packed, const float* bias,
                    void* y, void* stream);
/* w_t: optional [Cin][T*Cout] transpose of the packed weight (devis_repack_conv_weight_dgrad on the packed
 * weight viewed as [T*Cout][1][Cin]) that lets dgrad run on the MFMA gather-GEMM. */
int devis_convT_dgrad(const devis_convT_desc* d, const void* dy, const void* w_packed, const void* w_t, void* dx,
                      void* stream);
size_t devis_convT_wgrad_workspace_bytes(const devis_convT_desc* d);
int devis_convT_wgrad(const devis_convT_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                      void* workspace, size_t workspace_bytes, void* stream);

/* DeUp_Cat tail fused (3-D U-Net decoder, UNet3DZoo.py:121-129: ConvTranspose3d(C,C,2,2) -> cat((prev, up), 1) ->
 * Conv3d(2C, C, 1)).  One pass y[v] = W3a prev[v] + M[parity(v)] t[coarse(v)] + b' with the composite matrices
 * M[p] = W3b Wt[p], b' = b3 + W3b bt: the up-sampled tensor and its gradient are never written.
 * (N, D, H, W) is the COARSE grid of t; prev / y are channels-last (N, 2D, 2H, 2W, C).  bf16 only.
 * devis_upcat_supported: 1 when devis_upcat_fwd takes the shape (else run the two layers separately).
 * devis_upcat_prepare (fp32 masters: w3 = conv3.weight [C][2C], b3, wt = conv2.weight [C][C][8], bt) fills
 *   wc [8][C][2C] bf16 (forward), bias [C] fp32, w3a [C][1][C] + w3a_t (packed 1x1 weight of the prev half and its
 *   dgrad packing: d prev = devis_conv_dgrad on them), wq [C][8][C] bf16 (k2-s2 conv packing: d t = devis_conv_fwd
 *   of dy with kernel = stride = 2 on it).
 * devis_upcat_bwd_data: both data gradients (autograd of the lines above) in one pass over dy: dprev (N, 2D, 2H, 2W, C)
 *   = W3a^T dy and dt (N, D, H, W, C) = sum_p M[p]^T dy[fine(p)], from w3a_t and wq of devis_upcat_prepare.
 * devis_upcat_grad_combine: g = devis_convT_wgrad(t, dy) [C][C][8], dw3a = devis_conv_wgrad(prev, dy) [C][C],
 *   s = its dbias -> dw3 [C][2C], db3, dwt [C][C][8], dbt (the reference's conv3 / conv2 parameter gradients). */
int devis_upcat_supported(int64_t N, int64_t D, int64_t H, int64_t W, int64_t C, int dtype);
int devis_upcat_prepare(const float* w3, const float* b3, const float* wt, const float* bt, int64_t C, void* wc,
                        float* bias, void* w3a, void* w3a_t, void* wq, void* stream);
int devis_upcat_fwd(const void* prev, const void* t, const void* wc, const float* bias, void* y, int64_t N, int64_t D,
                    int64_t H, int64_t W, int64_t C, int dtype, void* stream);
int devis_upcat_bwd_data(const void* dy, const void* w3a_t, const void* wq, void* dprev, void* dt, int64_t N, int64_t D,
                         int64_t H, int64_t W, int64_t C, int dtype, void* stream);
int devis_upcat_grad_combine(const float* g, const float* dw3a, const float* s, const float* w3, const float* wt,
                             const float* bt, int64_t C, float* dw3, float* db3, float* dwt, float* dbt, void* stream);

/* ------------------------------------------------------- normalisation -- */
/* per-(n,c) moments of a channels-last tensor: out[n][c] = {sum x, sum x^2} (fp64).
 * First half of nn.GroupNorm / nn.BatchNorm forward (UNet3DZoo.py:12,149,153; UNet2DZoo.py:13,16). */
size_t devis_moments_workspace_bytes(int64_t N, int64_t V, int64_t C);
int devis_channel_moments(const void* x, int64_t N, int64_t V, int64_t C, int dtype, double* out,
                          void* workspace, size_t workspace_bytes, void* stream);
/* GroupNorm(G, C): moments -> per-(n,c) affine  y = scale*x + shift, plus mean/rstd [N][G]. */
int devis_gn_finalize(const double* moments, int64_t N, int64_t V, int64_t C, int64_t G,
                      const float* gamma, const float* beta, float eps, float* scale, float* shift,
                      float* mean, float* rstd, void* stream);
/* BatchNorm: training != 0 -> batch statistics (+ running-stat update, momentum, unbiased var);
 * training == 0 -> running statistics (moments may be NULL).  scale/shift are [N][C] (rows equal),
 * mean/rstd are [C]. */
int devis_bn_finalize(const double* moments, int64_t N, int64_t V, int64_t C, const float* gamma,
                      const float* beta, float eps, float momentum, int training, float* running_mean,
                      float* running_var, float* scale, float* shift, float* mean, float* rstd,
                      void* stream);
/* y = [relu](scale*x + shift) (+ residual): materialised norm+ReLU (+ residual add, UNet3DZoo.py:163). */
int devis_affine_act(const void* x, const float* scale, const float* shift, const void* residual,
                     void* y, int64_t N, int64_t V, int64_t C, int dtype, int relu, void* stream);
/* backward: dy' = dy * [scale*x+shift > 0];  out[n][c] = {sum dy', sum dy'*x} (fp64) */
int devis_norm_bwd_moments(const void* dy, const void* x, const float* scale, const float* shift,
                           int64_t N, int64_t V, int64_t C, int dtype, double* out, void* workspace,
                           size_t workspace_bytes, void* stream);
/* moments -> dx = cA*dy' + cB*x + cC  coefficients [N][C], and dgamma/dbeta [C].
 * `bn` = 0: GroupNorm with G groups (mean/rstd [N][G]); 1: BatchNorm on batch statistics (mean/rstd [C]);
 * 2: BatchNorm normalising with its running statistics (eval mode: cB = cC = 0, as autograd gives for
 * F.batch_norm(training=False)). */
int devis_norm_bwd_finalize(const double* moments, int64_t N, int64_t V, int64_t C, int64_t G, int bn,
                            const float* gamma, const float* mean, const float* rstd, float* cA,
                            float* cB, float* cC, float* dgamma, float* dbeta, void* stream);
/* Two-launch statistics (the training path): the *_partials calls run only the first reduction stage and leave
 * partial[n][chunk][C][2] (fp64) in `workspace` (*chunks_host = chunks per sample, written on the host);
 * the *_finalize_partials calls add the chunks in a fixed order and finalize in the same kernel.  Same results
 * as moments + finalize above (nn.GroupNorm / nn.BatchNorm3d forward and their autograd, UNet3DZoo.py:12,39-52,
 * 149-162).  devis_bn_finalize_partials is training mode only and also adds 1 to *num_batches_tracked
 * (int64 device scalar, may be NULL). */
int devis_channel_moments_partials(const void* x, int64_t N, int64_t V, int64_t C, int dtype, void* workspace,
                                   size_t workspace_bytes, int64_t* chunks_host, void* stream);
int devis_norm_bwd_moments_partials(const void* dy, const void* x, const float* scale, const float* shift,
                                    int64_t N, int64_t V, int64_t C, int dtype, void* workspace,
                                    size_t workspace_bytes, int64_t* chunks_host, void* stream);
int devis_gn_finalize_partials(const double* partials, int64_t chunks, int64_t N, int64_t V, int64_t C, int64_t G,
                               const float* gamma, const float* beta, float eps, float* scale, float* shift,
                               float* mean, float* rstd, void* stream);
int devis_bn_finalize_partials(const double* partials, int64_t chunks, int64_t N, int64_t V, int64_t C,
                               const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float* scale, float* shift, float* mean, float* rstd, void* stream);
int devis_norm_bwd_finalize_partials(const double* partials, int64_t chunks, int64_t N, int64_t V, int64_t C,
                                     int64_t G, int bn, const float* gamma, const float* mean, const float* rstd,
                                     float* cA, float* cB, float* cC, float* dgamma, float* dbeta, void* stream);
/* dx = cA*dy' + cB*x + cC (+ extra): autograd of group_norm/batch_norm + relu (+ skip gradient) */
int devis_norm_bwd_apply(const void* dy, const void* x, const float* scale, const float* shift,
                         const float* cA, const float* cB, const float* cC, const void* extra, void* dx,
                         int64_t N, int64_t V, int64_t C, int dtype, void* stream);
/* z = a + b: elementwise add of two channels-last tensors (skip-gradient accumulation) */
int devis_add(const void* a, const void* b, void* z, int64_t n_elems, int dtype, void* stream);
/* z = a * b: nn.Dropout / F.dropout3d as the reference evaluates them, `input * noise` (Unetdrop,
 * UNet3DZoo.py:30,352,358,362,366), with the noise tensor supplied by the caller; also its backward. */
int devis_mul(const void* a, const void* b, void* z, int64_t n_elems, int dtype, void* stream);

/* nn.MaxPool2d(2) (UNet2DZoo.py:112-118), floor mode; idx = argmax position 0..3 inside the window */
int devis_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int64_t N, int64_t H, int64_t W, int64_t C,
                        int dtype, void* stream);
int devis_maxpool2d_bwd(const void* dy, const uint8_t* idx, void* dx, int64_t N, int64_t H, int64_t W,
                        int64_t C, int dtype, void* stream);
/* devis_maxpool2d_bwd plus the skip branch's gradient in the same pass: dx = scatter(dy by idx) + skip (skip may
 * be NULL).  Replaces the autograd accumulation of the two gradients that reach c1..c4 in Unet2D.forward
 * (models/lib/UNet2DZoo.py:131-141: each c_k feeds both the pooling and a merge). */
int devis_maxpool2d_bwd_add(const void* dy, const uint8_t* idx, const void* skip, void* dx, int64_t N, int64_t H,
                            int64_t W, int64_t C, int dtype, void* stream);

/* -------------------------------------------------- evidential head/loss -- */
/* Element (n, v, k) of `logits` lives at logits[n*sN + v*sV + k*sK] (element strides), so both
 * channels-last (sV=K, sK=1) and the reference's NC(D)HW (sV=1, sK=V) are accepted.
 *
 * devis_evid_loss_fwd replaces TMSU.forward's head + dce_evidence_loss
 * (TrustworthySeg.py:144-155; criterions.py:421-456,117-127,226-257,353-370):
 *   e = softplus(z), alpha = e+1, loss = mean_v(L_ace + lambda*KL) + L_dice.
 * scal (fp32, DEVIS_EVID_SCAL_FLOATS entries) receives
 *   [0] loss  [1] L_dice  [2] mean_v(L_ace + lambda*KL)  [3] NV
 *   [4+k] I_k  [4+K+k] P_k  [4+2K+k] R_k            (k < K <= DEVIS_EVID_MAX_K)
 * evidence_out: optional NC(D)HW fp32 copy of the evidence (returned by TMSU.forward in train mode).
 * per_voxel_out: optional [N*V] fp32 = L_ace + lambda*KL per voxel (the (NV,1) tensor of
 *   criterions.py:456 minus the broadcast L_dice). */
#define DEVIS_EVID_MAX_K 8
#define DEVIS_EVID_SCAL_FLOATS (4 + 3 * DEVIS_EVID_MAX_K)
size_t devis_evid_workspace_bytes(int64_t N, int64_t V, int64_t K);
int devis_evid_loss_fwd(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype,
                        const void* labels, int label_dtype, int64_t N, int64_t V, int64_t K,
                        float lambda, float* evidence_out, float* per_voxel_out, float* scal,
                        void* workspace, size_t workspace_bytes, void* stream);
/* devis_evid_loss_fwd for callers that CANNOT ALLOCATE.  It has no counterpart in the reference; it exists for
 * callers that cannot call hipMalloc or size a workspace: code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  Same arguments, kernels and results as devis_evid_loss_fwd, minus the
 * workspace: the per-block partial sums live in a fixed 200 KB `__device__` array inside the library (one copy per
 * GPU, held until the process ends).  Limit: N <= DEVIS_EVID_NOALLOC_MAX_N samples (any V, K <= DEVIS_EVID_MAX_K);
 * beyond it the call returns DEVIS_ERR_SCRATCH_LIMIT and launches nothing.  It calls no hipMalloc / hipFree.  The
 * device that owns `stream` must be current.  All calls on one GPU share that GPU's array: calls that overlap on
 * one GPU (two streams, two threads) race on it - serialise them, or use devis_evid_loss_fwd with a workspace per
 * stream.  The first call on a device resolves the array's address (hipGetSymbolAddress): make it once outside
 * any graph capture. */
#define DEVIS_EVID_NOALLOC_MAX_N 1024
int devis_evid_loss_fwd_noalloc(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype,
                                const void* labels, int label_dtype, int64_t N, int64_t V, int64_t K,
                                float lambda, float* evidence_out, float* per_voxel_out, float* scal,
                                void* stream);
/* dlogits (same strides/dtype as logits) = grad_loss * dloss/dz  [+ grad_evidence * sigmoid(z)]
 * grad_loss: device pointer to one fp32;  grad_evidence: optional NC(D)HW fp32.
 * grad_per_voxel: optional [N*V] fp32 upstream gradient of the (NV,1) tensor (then grad_loss is
 *   ignored for the per-voxel terms and `grad_dice` (device fp32 scalar) scales the dice term). */
int devis_evid_loss_bwd(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype,
                        const void* labels, int label_dtype, int64_t N, int64_t V, int64_t K,
                        float lambda, const float* scal, const float* grad_loss,
                        const float* grad_per_voxel, const float* grad_dice,
                        const float* grad_evidence, void* dlogits, void* stream);
/* General form of the two calls above, covering every evidential criterion of models/criterions.py.
 *   kind  DEVIS_CRIT_DCE   dce_evidence_loss     criterions.py:421-456 (soft dice on the evidence)
 *         DEVIS_CRIT_DCEU  dce_evidence_u_loss   criterions.py:373-419 + DiceLoss :166-215 (Tversky dice on
 *                          softmax(evidence), accuracy-vs-uncertainty term with a_au, eps, disentangle)
 *         DEVIS_CRIT_CE    ce_loss               criterions.py:130-142   (per voxel only)
 *         DEVIS_CRIT_MSE   mse_loss              criterions.py:144-154   (per voxel only)
 *         DEVIS_CRIT_KL    KL(alpha, c)          criterions.py:117-127   (per voxel only, no labels)
 *   input_kind 0: `in` holds logits (softplus inside, as TMSU.forward :144-146); 1: `in` holds alpha, the
 *   evidence is alpha - 1 (criterions.py:441) and `evidence_in` (optional, NC(D)HW fp32) is the separate
 *   tensor the dice term is taken on (the `evidence` argument of the reference functions).
 * per_voxel_out [N*V] = the criterion's (NV,1) result without the broadcast dice term; scal as above with
 *   [0] = mean(per voxel) + w * L_dice (w = 1, or 1 - a_au for DCEU), and for DCEU [4..] = TP_k, FP_k, FN_k.
 *   scal (with workspace) may be NULL when only per_voxel_out is wanted.
 * bwd: d_in (strides/dtype of `in`) = gradient w.r.t. logits or alpha; d_evidence_in (NC(D)HW fp32, iff
 *   evidence_in was given) = the dice term's gradient w.r.t. that tensor.  grad_* as devis_evid_loss_bwd. */
enum { DEVIS_CRIT_DCE = 0, DEVIS_CRIT_DCEU = 1, DEVIS_CRIT_CE = 2, DEVIS_CRIT_MSE = 3, DEVIS_CRIT_KL = 4 };
typedef struct devis_evid_params {
  int kind;
  int input_kind;
  int disentangle;
  float lambda; /* KL annealing coefficient min(1, step / lambda_step) */
  float a_au;   /* DCEU: 0.01 * exp(-ln(0.01) * step / total_step), criterions.py:392-393 */
  float eps;    /* DCEU */
} devis_evid_params;
int devis_evid_crit_fwd(const devis_evid_params* p, const void* in, int64_t sN, int64_t sV, int64_t sK,
                        int dtype, const float* evidence_in, const void* labels, int label_dtype,
                        int64_t N, int64_t V, int64_t K, float* evidence_out, float* per_voxel_out,
                        float* scal, void* workspace, size_t workspace_bytes, void* stream);
int devis_evid_crit_bwd(const devis_evid_params* p, const void* in, int64_t sN, int64_t sV, int64_t sK,
                        int dtype, const float* evidence_in, const void* labels, int label_dtype,
                        int64_t N, int64_t V, int64_t K, const float* scal, const float* grad_loss,
                        const float* grad_per_voxel, const float* grad_dice, const float* grad_evidence,
                        void* d_in, float* d_evidence_in, void* stream);

/* val/test head (trainUMIS.py:286-290,446-451,521; predict.py:92-105):
 *   evidence (optional, NC(D)HW fp32), pred = argmax_k alpha_k/S (first max; uint8),
 *   u = K/S (fp32), mne = normalised entropy over classes >= 1 (optional fp32),
 *   umask bit i = [u > thresholds[i]] (optional uint16, n_thr <= 16). */
int devis_evid_infer(const void* logits, int64_t sN, int64_t sV, int64_t sK, int dtype, int64_t N,
                     int64_t V, int64_t K, float* evidence_out, uint8_t* pred, float* u, float* mne,
                     uint16_t* umask, const float* thresholds_host, int n_thr, void* stream);

/* dz = g * d softplus(z)/dz, fp32 flat arrays: autograd of TMSU.infer (TrustworthySeg.py:160-166)
 * when the evidence is consumed outside the fused loss. */
int devis_softplus_bwd(const float* z, const float* g, float* dz, int64_t n, void* stream);

/* ---------------------------------------------------------- attention gate -- */
/* Attention_block of the 3-D attention U-Net (UNet3DZoo.py:69-100) behind its 1x1 convs W_g / W_x:
 *   a = relu(BN_g(g1) + BN_x(x1));  s = psi_conv(a) (one channel, fp32 [N][V]);  out = skip * sigmoid(BN_psi(s)).
 * The three BatchNorms enter as per-(n,c) affine coefficients (devis_bn_finalize*), their statistics leave as
 * partial[n][chunk][C][2] fp64 for devis_bn_finalize_partials / devis_norm_bwd_finalize_partials:
 *   devis_att_psi_fwd   g1, x1 [N][V][C] -> s and the moments {sum s, sum s^2}          (chunks = devis_att_chunks(V), C = 1)
 *   devis_att_gate_fwd  skip [N][V][Cs], s, scale/shift of BN_psi [N] -> out
 *   devis_att_gate_bwd  dout -> dskip = dout*p, dz = p(1-p) sum_c dout*skip (fp32), moments {sum dz, sum dz*s}
 *   devis_att_psi_bwd   dz, s, BN_psi backward coefficients cA/cB/cC [N] -> dpre = d(BN_g out) = d(BN_x out)
 *                       [N][V][C], the BatchNorm-backward moments of both branches (chunks =
 *                       devis_att_psi_bwd_chunks), and the psi conv's weight / bias gradients (fp32 [C], [1]).
 * C = F_int <= 64 with C/8 (bf16) or C/4 (fp32) a power of two <= 16. */
int64_t devis_att_chunks(int64_t V);
int64_t devis_att_psi_bwd_chunks(int64_t V, int64_t C, int dtype);
int devis_att_psi_fwd(const void* g1, const void* x1, const float* scale_g, const float* shift_g,
                      const float* scale_x, const float* shift_x, const float* w_psi, const float* b_psi,
                      int64_t N, int64_t V, int64_t C, int dtype, float* s, double* partials,
                      size_t partials_bytes, void* stream);
int devis_att_gate_fwd(const void* skip, const float* s, const float* scale_psi, const float* shift_psi, void* out,
                       int64_t N, int64_t V, int64_t C, int dtype, void* stream);
int devis_att_gate_bwd(const void* dout, const void* skip, const float* s, const float* scale_psi,
                       const float* shift_psi, void* dskip, float* dz, double* partials, size_t partials_bytes,
                       int64_t N, int64_t V, int64_t C, int dtype, void* stream);
size_t devis_att_psi_bwd_workspace_bytes(int64_t N, int64_t V, int64_t C, int dtype);
int devis_att_psi_bwd(const float* dz, const float* s, const float* cA, const float* cB, const float* cC,
                      const void* g1, const void* x1, const float* scale_g, const float* shift_g,
                      const float* scale_x, const float* shift_x, const float* w_psi, int64_t N, int64_t V,
                      int64_t C, int dtype, void* dpre, double* moments_g, double* moments_x, size_t moments_bytes,
                      float* dw_psi, float* db_psi, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------- evaluation counts -- */
/* One pass over device-resident prediction / target / uncertainty maps that yields every count the reference's
 * val()/test() metrics are made of (it copies the volumes to the host and counts with numpy):
 *   conf[t*K + p]   int64: voxels with target t, prediction p -> dice_score, mIOU, softmax_output_dice,
 *                   softmax_output_litsdice (predict.py:296-300,386-420; trainUMIS.py:286-331) are ratios of its sums;
 *   ucount[i*8 + c] int64: tp, tn, fp, fn, tpu, tnu, fpu, fnu of numpyfunctions.uncertainty() (:86-107) with
 *                   prediction = pred > 0, target = target > 0, thresholded_uncertainty = bit i of umask
 *                   (devis_evid_infer: u > thresholds[i], trainUMIS.py:521), i < n_thr <= 16;
 *   sums[0..2]      fp64: sum u, sum (1-u), sum exp(-u) -> mean uncertainty / conf / confe (trainUMIS.py:541-548).
 * pred uint8, target per target_dtype (devis_label_dtype), umask/ucount may be NULL with n_thr = 0, u/sums may
 * both be NULL.  *bad_label (device int) is set to 1 if a label or prediction is outside [0, K). */
size_t devis_eval_counts_workspace_bytes(int64_t NV);
int devis_eval_counts(const uint8_t* pred, const void* target, int target_dtype, const uint16_t* umask, int n_thr,
                      const float* u, int64_t NV, int K, int64_t* conf, int64_t* ucount, double* sums,
                      int* bad_label, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- optimizer -- */
/* One Adam step with coupled L2 weight decay over flat fp32 arrays of n elements (16-byte aligned), in place:
 * torch.optim.Adam(model.parameters(), lr, weight_decay=1e-5) as trainUMIS.py:1394 builds it and :261 steps
 * it, with lr set per iteration by adjust_learning_rate (:217-219).  `step` = 1 for the first update (bias
 * correction).  Hyper-parameters are doubles because torch derives 1-beta, lr/bias_correction1 and
 * sqrt(bias_correction2) in double before rounding to fp32. */
int devis_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                    double beta1, double beta2, double eps, double weight_decay, int64_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEVIS_HIP_H */
