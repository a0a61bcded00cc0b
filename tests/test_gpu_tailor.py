"""GPU: sliding-window stitch (predict.py:159-191) index arithmetic vs golden, and TMSU val mode."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_tailor_index_arithmetic(golden):
    from devis_amd import tailor_and_concat
    G = golden("tailor")
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 4, 240, 240, 160), generator=g).cuda()
    a = tailor_and_concat(x, lambda t: t * 2 + 1).cpu()
    assert a.shape == (1, 4, 240, 240, 155)
    assert hashlib.sha256(a.contiguous().numpy().tobytes()).hexdigest() == str(G["fake_sha256"])

    def pos(t):
        d, h, w = t.shape[2:]
        base = (torch.arange(d).view(d, 1, 1) * 1e4 + torch.arange(h).view(1, h, 1) * 1e2
                + torch.arange(w).view(1, 1, w)).float().to(t.device)
        return t * 0 + base
    b = tailor_and_concat(x, pos).cpu()
    assert hashlib.sha256(b.contiguous().numpy().tobytes()).hexdigest() == str(G["pos_sha256"])


def test_tmsu_val_mode_matches_oracle_on_one_patch():
    """val-mode TMSU on a full volume; patch 0's region must equal the oracle's patch-0 forward."""
    from oracle import devis_oracle as O
    from devis_amd import TMSU
    from tests.util import brats_args, rel_err
    torch.manual_seed(0)
    m = TMSU(brats_args(compute_dtype=torch.float32))
    sd = {k: v.detach().cpu().clone() for k, v in m.backbone.state_dict().items()}
    m.eval()
    g = torch.Generator().manual_seed(9)
    x = torch.randn((1, 4, 240, 240, 160), generator=g)
    with torch.no_grad():
        ev = m(x.cuda(), None, 1, 'val', 'BraTS')
    assert ev.shape == (1, 4, 240, 240, 155)
    with torch.no_grad():
        ref = torch.nn.functional.softplus(O.unet3d_forward(sd, x[..., :128, :128, :128][..., :64].contiguous()
                                                            if False else x[..., :128, :128, :128], False))
    assert rel_err(ev[..., :128, :128, :128], ref) < 5e-5
