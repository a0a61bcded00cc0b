"""TMSU: the evidential segmentation wrapper (reference models/TrustworthySeg.py:21-166).

Same constructor argument (`args` namespace), attributes and ``forward(X, y, global_step,
mode, dataset)`` / ``infer(input)`` behaviour; the backbone and the head+loss are the HIP
kernels of this package.  Deviations from the reference, all documented in DESIGN.md:
  * ``self.backbone.cuda()`` (TrustworthySeg.py:116) becomes ``.to(current cuda device)``;
  * the ISIC branch, which raises NameError in the reference (TrustworthySeg.py:51 uses the
    undefined names ``model``/``classes``), selects ``Unet2D(3, num_classes)`` for model 'U';
  * backbones outside SURVEY.md §8 (ResU and the 2-D AU) raise NameError like an unknown model does.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .criterions import evidential_head_loss
from .predict import tailor_and_concat
from .unet2d import Unet2D, Unet2DDRIVE
from .unet3d import Unet
from .attunet3d import AttUnet
from .vnet3d import VNet
from .vnet2d import VNet2D


class TMSU(nn.Module):
    def __init__(self, args):
        super().__init__()
        num_classes = args.num_classes
        dataset = args.dataset
        model_name = args.model_name
        input_modality = args.input_modality
        dtype = getattr(args, "compute_dtype", torch.bfloat16)
        backbone = None
        if dataset == 'BraTS':
            if model_name == 'U' and input_modality == 'four':
                backbone = Unet(in_channels=4, base_channels=16, num_classes=num_classes, compute_dtype=dtype)
            elif model_name == 'AU' and input_modality == 'four':            # TrustworthySeg.py:40-41
                backbone = AttUnet(in_channels=4, base_channels=16, num_classes=num_classes, compute_dtype=dtype)
            elif model_name == 'V' and input_modality == 'four':             # TrustworthySeg.py:42-44
                backbone = VNet(n_channels=4, n_classes=num_classes, n_filters=16, normalization='gn',
                                has_dropout=False, compute_dtype=dtype)
        elif dataset == 'LiTS':
            if model_name == 'U':
                backbone = Unet(in_channels=1, base_channels=16, num_classes=num_classes, compute_dtype=dtype)
            elif model_name == 'AU':                                         # TrustworthySeg.py:65
                backbone = AttUnet(in_channels=1, base_channels=16, num_classes=num_classes, compute_dtype=dtype)
            elif model_name == 'V':                                          # TrustworthySeg.py:67-69
                backbone = VNet(n_channels=1, n_classes=num_classes, n_filters=16, normalization='gn',
                                has_dropout=False, compute_dtype=dtype)
        elif dataset == 'ISIC':
            if model_name == 'U':
                backbone = Unet2D(3, num_classes, compute_dtype=dtype)
            elif model_name == 'V':                                          # TrustworthySeg.py:53-55
                backbone = _vnet2d(num_classes, dtype)
        elif dataset == 'DRIVE':
            if model_name == 'U':
                backbone = Unet2DDRIVE(3, num_classes, compute_dtype=dtype)
            elif model_name == 'V':                                          # TrustworthySeg.py:77-79
                backbone = _vnet2d(num_classes, dtype)
        elif dataset in ('TOAR', 'HC'):
            if model_name == 'U':
                backbone = Unet2D(1, num_classes, compute_dtype=dtype)
            elif model_name == 'V':                                          # TrustworthySeg.py:90-92,103-105
                backbone = _vnet2d(num_classes, dtype)
        else:
            print('There is no this dataset')
            raise NameError
        if backbone is None:
            print('There is no this model')
            raise NameError
        self.backbone = backbone
        if torch.cuda.is_available():
            self.backbone.to(torch.device('cuda', torch.cuda.current_device()))
        self.classes = num_classes
        self.disentangle = False
        self.eps = 1e-10
        self.lambda_epochs = args.lambda_epochs
        self.total_epochs = args.end_epochs + 1
        self.u_loss = args.Uncertainty_Loss

    def forward(self, X, y, global_step, mode, dataset):
        if mode == 'train':
            logits = self.backbone.forward_channels_last(X)
            evidence, loss = evidential_head_loss(logits, _labels(y), global_step, self.lambda_epochs,
                                                  channels_last=True, want_evidence=True, u_loss=bool(self.u_loss),
                                                  total_step=self.total_epochs, eps=self.eps,
                                                  disentangle=self.disentangle)
            return evidence, loss
        if dataset == 'BraTS':
            out = tailor_and_concat(X, self.backbone)
        else:
            out = self.backbone(X)
        return self.infer(out)

    def infer(self, input):
        """softplus evidence of NC(D)HW logits (TrustworthySeg.py:160-166)."""
        return _SoftplusFn.apply(input)


def _vnet2d(num_classes, dtype):
    return VNet2D(n_channels=3, n_classes=num_classes, n_filters=32, normalization='gn', has_dropout=False,
                  compute_dtype=dtype)


def _labels(y: torch.Tensor) -> torch.Tensor:
    if y.dtype not in (torch.int64, torch.int32, torch.uint8):
        y = y.to(torch.int64)          # the reference casts with y.to(torch.int64), TrustworthySeg.py:154
    return y.contiguous()


def ops_from_cl(logits_cl):
    from .unet3d import _FromChannelsLast
    return _FromChannelsLast.apply(logits_cl)


class _SoftplusFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        z = z.contiguous().float()
        ctx.save_for_backward(z)
        return ops.evid_infer(z, False, (), want_evidence=True)["evidence"]

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        return ops.softplus_bwd(z, g.contiguous().float())
