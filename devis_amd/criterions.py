"""Evidential criteria on MI355X - the call surface of the reference's models/criterions.py.

``evidential_head_loss`` is the fused path TMSU.forward uses: logits -> softplus evidence,
alpha = e + 1, digamma evidential CE + annealed KL(Dir(alpha~)||Dir(1)) + soft-dice on the
evidence, in two HBM passes (forward-reduce, backward), SURVEY.md Appendix A.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

Tensor = torch.Tensor


class _EvidLossFn(torch.autograd.Function):
    """(logits, labels) -> (loss, evidence).  criterions.py:421-456 + TrustworthySeg.py:144-155."""

    @staticmethod
    def forward(ctx, logits, labels, lam, channels_last, want_evidence):
        labels = labels.contiguous()
        logits = logits.contiguous()
        scal, evidence, _ = ops.evid_loss_fwd(logits, channels_last, labels, lam, want_evidence)
        ctx.save_for_backward(logits, labels, scal)
        ctx.lam, ctx.channels_last = lam, channels_last
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
        dz = ops.evid_loss_bwd(logits, ctx.channels_last, labels, ctx.lam, scal, g_loss, g_ev)
        return dz, None, None, None, None


def evidential_head_loss(logits: Tensor, labels: Tensor, current_step, lamda_step, channels_last: bool,
                         want_evidence: bool = True):
    """Fused head + default loss: returns (evidence NC(D)HW fp32 | None, loss scalar)."""
    lam = min(1, current_step / lamda_step)           # annealing_coef, criterions.py:450
    loss, evidence = _EvidLossFn.apply(logits, labels, float(lam), channels_last, want_evidence)
    return (evidence if want_evidence else None), loss


def _as_labels(p: Tensor) -> Tensor:
    if p.dtype not in (torch.int64, torch.int32, torch.uint8):
        p = p.to(torch.int64)
    return p.contiguous()


def dce_evidence_u_loss(p, alpha, c, current_step, lamda_step, total_step, eps, disentangle, evidence, backbone_pred):
    raise NotImplementedError("dce_evidence_u_loss kernels not built yet")
