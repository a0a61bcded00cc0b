"""Evidential criteria on MI355X - the call surface of the reference's models/criterions.py.

Two levels:

* ``evidential_head_loss`` is the fused path TMSU.forward uses: logits -> softplus evidence,
  alpha = e + 1, digamma evidential CE + annealed KL(Dir(alpha~)||Dir(1)) + dice term (+ the
  accuracy-vs-uncertainty term of the Uncertainty_Loss variant), in two HBM passes
  (forward-reduce, backward), SURVEY.md Appendix A.
* the reference's own functions (``KL``, ``ce_loss``, ``mse_loss``, ``dce_evidence_loss``,
  ``dce_evidence_u_loss``) with their signatures and return shapes, taking ``alpha`` / ``evidence``
  tensors, differentiable, each one launch of the same kernels in "alpha given" mode.

``get_soft_label``, ``SDiceLoss`` and ``DiceLoss`` are kept for API compatibility as a few torch
tensor ops on the device: the fused kernels never build a one-hot tensor, so nothing on the
training path calls them.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib, ops

Tensor = torch.Tensor


def annealing_au(current_step, total_step) -> float:
    """annealing_AU of criterions.py:392-393, evaluated in fp32 like the reference's tensor ops."""
    a0 = torch.tensor(0.01, dtype=torch.float32)
    return float(a0 * torch.exp(-torch.log(a0) / total_step * current_step))


# --------------------------------------------------------------------------- fused head (logits in)
class _EvidLossFn(torch.autograd.Function):
    """(logits, labels) -> (loss, evidence).  criterions.py:373-456 + TrustworthySeg.py:144-155."""

    @staticmethod
    def forward(ctx, logits, labels, kind, lam, a_au, eps, disentangle, channels_last, want_evidence):
        labels = labels.contiguous()
        logits = logits.contiguous()
        params = ops.crit_params(kind, False, lam, a_au, eps, disentangle)
        strides = ops.logits_strides(logits, channels_last)
        n, v, k = strides[:3]
        sp = tuple(logits.shape[1:-1]) if channels_last else tuple(logits.shape[2:])
        scal, evidence, _ = ops.crit_fwd(params, logits, strides, labels, want_evidence=want_evidence,
                                         want_per_voxel=False, ev_shape=(n, k) + sp)
        ctx.save_for_backward(logits, labels, scal)
        ctx.params, ctx.strides = params, strides
        # the train loop drops `evidence` (trainUMIS.py:255): without this autograd would hand backward a
        # zero-filled gradient of its size (134 MB at BraTS B=2) for the kernel to read
        ctx.set_materialize_grads(False)
        loss = scal[0].clone()
        if evidence is None:
            evidence = logits.new_zeros(())
            ctx.mark_non_differentiable(evidence)
        return loss, evidence

    @staticmethod
    def backward(ctx, g_loss, g_evidence):
        logits, labels, scal = ctx.saved_tensors
        if g_loss is None:
            g_loss = torch.zeros((), dtype=torch.float32, device=logits.device)
        g_loss = g_loss.contiguous().float()
        g_ev = None
        if g_evidence is not None and g_evidence.dim() > 0:
            g_ev = g_evidence.contiguous().float()
        dz, _ = ops.crit_bwd(ctx.params, logits, ctx.strides, labels, scal, grad_loss=g_loss, grad_evidence=g_ev)
        return (dz,) + (None,) * 8


def evidential_head_loss(logits: Tensor, labels: Tensor, current_step, lamda_step, channels_last: bool,
                         want_evidence: bool = True, u_loss: bool = False, total_step=None, eps: float = 1e-10,
                         disentangle: bool = False):
    """Fused head + loss: returns (evidence NC(D)HW fp32 | None, mean loss scalar)."""
    lam = min(1, current_step / lamda_step)           # annealing_coef, criterions.py:390,450
    if u_loss:
        kind, a_au = _lib.CRIT_DCEU, annealing_au(current_step, total_step)
    else:
        kind, a_au = _lib.CRIT_DCE, 0.0
    loss, evidence = _EvidLossFn.apply(logits, labels, kind, float(lam), a_au, float(eps), bool(disentangle),
                                       channels_last, want_evidence)
    return (evidence if want_evidence else None), loss


# --------------------------------------------------------------------------- reference API (alpha in)
def _as_labels(p: Tensor) -> Tensor:
    if p.dtype not in (torch.int64, torch.int32, torch.uint8):
        p = p.to(torch.int64)
    return p.contiguous()


class _CritFn(torch.autograd.Function):
    """(alpha, evidence|None, labels|None) -> (per-voxel (NV,1), dice scalar) for one criterion kind."""

    @staticmethod
    def forward(ctx, alpha, evidence, labels, params, strides):
        alpha = alpha.contiguous().float()
        ev = evidence.contiguous().float() if evidence is not None else None
        has_dice = params.kind in (_lib.CRIT_DCE, _lib.CRIT_DCEU)
        scal, _, pv = ops.crit_fwd(params, alpha, strides, labels, evidence_in=ev if has_dice else None,
                                   want_per_voxel=True, want_scal=has_dice)
        ctx.params, ctx.strides, ctx.has_ev = params, strides, (ev is not None and has_dice)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(alpha, ev if ctx.has_ev else None, labels, scal)
        dice = scal[1].clone() if has_dice else alpha.new_zeros(())
        return pv, dice

    @staticmethod
    def backward(ctx, g_pv, g_dice):
        alpha, ev, labels, scal = ctx.saved_tensors
        n, v = ctx.strides[0], ctx.strides[1]
        f = dict(dtype=torch.float32, device=alpha.device)
        g_pv = torch.zeros((n * v, 1), **f) if g_pv is None else g_pv.contiguous().float()
        g_dice = torch.zeros((), **f) if g_dice is None else g_dice.contiguous().float()
        d_alpha, d_ev = ops.crit_bwd(ctx.params, alpha, ctx.strides, labels, scal, evidence_in=ev,
                                     grad_per_voxel=g_pv, grad_dice=g_dice)
        return d_alpha, d_ev, None, None, None


def _row_strides(alpha: Tensor):
    """(NV, K) row-major alpha, as KL / mse_loss take it."""
    nv, k = alpha.shape
    return 1, nv, k, nv * k, k, 1


def KL(alpha: Tensor, c: int) -> Tensor:
    """KL(Dir(alpha) || Dir(1)) per row -> (NV, 1).  criterions.py:117-127."""
    a2 = alpha.reshape(-1, alpha.shape[-1]) if alpha.dim() != 2 else alpha
    if a2.shape[1] != c:
        raise ValueError("alpha must have c columns")
    params = ops.crit_params(_lib.CRIT_KL, True)
    pv, _ = _CritFn.apply(a2, None, None, params, _row_strides(a2))
    return pv


def ce_loss(p: Tensor, alpha: Tensor, c: int, global_step, annealing_step) -> Tensor:
    """criterions.py:130-142: alpha (N,C,D,H,W), p (N,D,H,W) -> (N,1,D,H,W)."""
    params = ops.crit_params(_lib.CRIT_CE, True, min(1, global_step / annealing_step))
    pv, _ = _CritFn.apply(alpha, None, _as_labels(p), params, ops.logits_strides(alpha, False))
    return pv.view((alpha.shape[0], 1) + tuple(alpha.shape[2:]))


def mse_loss(p: Tensor, alpha: Tensor, c: int, global_step, annealing_step=1) -> Tensor:
    """criterions.py:144-154: alpha (NV,C) rows, p (NV,) -> (NV,1)."""
    params = ops.crit_params(_lib.CRIT_MSE, True, min(1, global_step / annealing_step))
    pv, _ = _CritFn.apply(alpha, None, _as_labels(p), params, _row_strides(alpha))
    return pv


def dce_evidence_loss(p, alpha, c, current_step, lamda_step, total_step, eps, disentangle, evidence, backbone_pred):
    """criterions.py:421-456 -> (NV,1) = L_ace + L_dice + lambda*KL (soft dice on `evidence`)."""
    params = ops.crit_params(_lib.CRIT_DCE, True, min(1, current_step / lamda_step))
    pv, dice = _CritFn.apply(alpha, evidence, _as_labels(p), params, ops.logits_strides(alpha, False))
    return pv + dice


def dce_evidence_u_loss(p, alpha, c, current_step, lamda_step, total_step, eps, disentangle, evidence, backbone_pred):
    """criterions.py:373-419 -> (NV,1) = L_ace + lambda*KL + (1 - a)*L_dice + L_AU (Tversky dice on softmax(evidence))."""
    a_au = annealing_au(current_step, total_step)
    params = ops.crit_params(_lib.CRIT_DCEU, True, min(1, current_step / lamda_step), a_au, eps, disentangle)
    pv, dice = _CritFn.apply(alpha, evidence, _as_labels(p), params, ops.logits_strides(alpha, False))
    return pv + (1 - a_au) * dice


# --------------------------------------------------------------------------- API-compat shims (torch ops)
def get_soft_label(input_tensor: Tensor, num_class: int) -> Tensor:
    """criterions.py:353-370: (N,1,...) integer labels -> (N,...,K) float one-hot."""
    t = input_tensor.permute(0, 2, 3, 4, 1) if input_tensor.dim() == 5 else input_tensor.permute(0, 2, 3, 1)
    return torch.cat([torch.eq(t, i * torch.ones_like(t)) for i in range(num_class)], dim=-1).float()


class SDiceLoss(nn.Module):
    """criterions.py:216-257 (weight_map=None path)."""

    def forward(self, preds, targets, weight_map=None):
        if weight_map is not None:
            raise NotImplementedError("SDiceLoss with a weight_map is off the DEviS training path")
        k = preds.size(1)
        pr = preds.permute(0, 2, 3, 4, 1) if preds.dim() == 5 else preds.permute(0, 2, 3, 1)
        pred = pr.contiguous().view(-1, k)
        ground = targets.view(-1, k)
        ref_vol, intersect, seg_vol = torch.sum(ground, 0), torch.sum(ground * pred, 0), torch.sum(pred, 0)
        dice = (2.0 * intersect + 1e-5) / (ref_vol + seg_vol + 1.0 + 1e-5)
        return torch.mean(-torch.log(dice))


class DiceLoss(nn.Module):
    """criterions.py:156-215 (weight=False, reduce, size_average): Tversky dice on softmax(preds)."""

    def forward(self, preds, targets, weight=False):
        if weight:
            raise NotImplementedError("DiceLoss(weight=True) is off the DEviS training path")
        c = preds.size(1)
        pr = preds.permute(0, 2, 3, 4, 1) if preds.dim() == 5 else preds.permute(0, 2, 3, 1)
        pr = pr.contiguous().view(-1, c)
        tg = targets.reshape(-1, 1)
        prob = torch.exp(torch.log_softmax(pr, dim=1))
        mask = torch.full_like(pr, 1e-8).scatter_(1, tg, 1.0)
        tp, fp, fn = (prob * mask).sum(0), (prob * (1 - mask)).sum(0), ((1 - prob) * mask).sum(0)
        a = torch.clamp(fp / (fp + fn + 1e-5), min=0.2, max=0.8)
        dice = tp / (tp + a * fp + (1 - a) * fn + 1e-5)
        return (1 - dice).sum() / c
