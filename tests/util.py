"""Helpers shared by the parity tests."""
import hashlib

import numpy as np
import torch


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def rel_err(a, b):
    """max |a-b| / max |b| (b = reference)."""
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def brats_args(**kw):
    import argparse
    d = dict(num_classes=4, dataset='BraTS', model_name='U', input_modality='four', end_epochs=99,
             lambda_epochs=50, Uncertainty_Loss=False)
    d.update(kw)
    return argparse.Namespace(**d)
