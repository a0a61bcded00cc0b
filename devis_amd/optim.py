"""Fused Adam for the evidential train step: ONE HIP launch over flat fp32 buffers.

Same update rule as the optimizer the reference builds (trainUMIS.py:1394
``torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=1e-5)``: coupled L2 decay,
betas (0.9, 0.999), eps 1e-8, no amsgrad) and the same ``param_groups[i]['lr']`` handle its
``adjust_learning_rate`` (:217-219) writes every iteration.

Parameters are moved into one flat buffer (``p.data`` becomes a view of it), laid out in the order the
backbone writes its flat gradient buffer (devis_amd.unet3d.Unet._grad_layout), so the step reads the
gradients where backward (and the data-parallel all-reduce) left them: no gather, no per-tensor launches.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import _lib
from ._lib import check


class FusedAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, module: Optional[torch.nn.Module] = None):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FusedAdam got an empty parameter list")
        dev = params[0].device
        for p in params:
            if not p.is_cuda or p.device != dev or p.dtype != torch.float32:
                raise _lib.DevisError("FusedAdam needs fp32 parameters on one GPU (there is no CPU fallback)")
        self.param_groups = [dict(params=params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        # layout: the backbone's gradient layout when every parameter belongs to it, else registration order
        backbone = getattr(module, "backbone", module)
        layout = None
        if backbone is not None and hasattr(backbone, "_grad_layout"):
            own = {id(p): n for n, p in backbone.named_parameters()}
            if len(own) == len(params) and all(id(p) in own for p in params):
                layout = [(p, backbone._grad_layout[own[id(p)]][0]) for p in params]
                self._backbone = backbone
        if layout is None:
            self._backbone = None
            off, layout = 0, []
            for p in params:
                layout.append((p, off))
                off += p.numel()
        self._layout = layout
        self.total = sum(p.numel() for p in params)
        self.flat_param = torch.empty((self.total,), dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in layout:
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self._own_grad = None
        self._own_views = None
        self.step_count = 0

    # ---- torch.optim.Optimizer surface used by the reference's loop
    def zero_grad(self, set_to_none: bool = True) -> None:
        for p, _ in self._layout:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def _flat_grad(self) -> torch.Tensor:
        """The flat gradient buffer: the backbone's own when every .grad is the expected view of it,
        else the gradients gathered into a buffer of this optimizer.  A parameter WITHOUT a gradient is an error:
        torch.optim.Adam skips such a parameter entirely (no decay, no moment update), which one fused launch over
        the flat buffers cannot express; every parameter of the reference's models gets a gradient each step."""
        bb = self._backbone
        flat = getattr(bb, "last_flat_grad", None) if bb is not None else None
        if flat is not None and flat.numel() == self.total:
            base = flat.data_ptr()
            if all(p.grad is not None and p.grad.data_ptr() == base + 4 * o and p.grad.is_contiguous()
                   for p, o in self._layout):
                return flat
        if self._own_grad is None:
            self._own_grad = torch.empty_like(self.flat_param)
        g = self._own_grad
        if self._own_views is None:
            self._own_views = [g[o:o + p.numel()].view(p.shape) for p, o in self._layout]
        if any(p.grad is None for p, _ in self._layout):
            raise RuntimeError("FusedAdam.step(): a parameter has no gradient (torch.optim.Adam would skip it; "
                               "FusedAdam updates every parameter in one launch) - use torch.optim.Adam for "
                               "models with unused parameters")
        # one multi-tensor launch instead of one copy per parameter
        torch._foreach_copy_(self._own_views, [p.grad for p, _ in self._layout])
        return g

    @torch.no_grad()
    def step(self) -> None:
        grp = self.param_groups[0]
        g = self._flat_grad()
        self.step_count += 1
        check(_lib.load().devis_adam_step(self.flat_param.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                          self.exp_avg_sq.data_ptr(), self.total, float(grp['lr']),
                                          float(grp['betas'][0]), float(grp['betas'][1]), float(grp['eps']),
                                          float(grp['weight_decay']), self.step_count,
                                          torch.cuda.current_stream().cuda_stream), "adam_step")

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != 'params'}])

    def load_state_dict(self, sd) -> None:
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.param_groups[0].update(sd['param_groups'][0])
