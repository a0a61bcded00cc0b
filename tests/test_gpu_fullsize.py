"""BASELINE's full BraTS size (B=2, 4x128^3, K=4, epoch 25) for the other two 3-D backbones TMSU selects
(SURVEY.md §8(f)-1 AttUnet, (f)-4 VNet): one train step against the CPU oracle run live on the same seeded tensors.
Bars as in tests/test_gpu_unet3d.py::test_full_size_config3_train_step_vs_oracle_live."""
import numpy as np
import pytest
import torch

from tests.util import GRAD_L2_TOL, brats_args, l2_rel_err, rel_err
from tests.util import BF16_BAR, BF16_MAX_EXCEPTION, FP32_BAR, LOSS_BF16_BAR, check

# bf16 evidence, MAX-norm relative error over the 1 M sampled values at 2x4x128^3: north_star's 1e-2 where it is
# met; exceptions (with the measured value) are listed per backbone
BF16_EVIDENCE_MAX = dict(BF16_MAX_EXCEPTION)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model_name", ["AU", "V"])
def test_full_size_train_step_vs_oracle_live(model_name):
    from oracle import devis_oracle as O
    from devis_amd import TMSU
    fwd = {"AU": lambda sd, t: O.attunet3d_forward(sd, t, True), "V": lambda sd, t: O.vnet3d_forward(sd, t)}[model_name]
    g = torch.Generator().manual_seed(4100)
    x = torch.randn((2, 4, 128, 128, 128), generator=g)
    y = torch.randint(0, 4, (2, 128, 128, 128), generator=g)

    def make(dtype):
        torch.manual_seed(12)
        return TMSU(brats_args(model_name=model_name, compute_dtype=dtype)).train()

    m = make(torch.float32)
    sd = {k: v.detach().cpu().clone() for k, v in m.backbone.state_dict().items()}
    names = [k for k, _ in m.backbone.named_parameters()]
    for k in names:
        sd[k].requires_grad_(True)
    ev_ref, loss_ref = O.tmsu_forward(lambda t: fwd(sd, t), x, y, 25, 'train', 'BraTS', 4, 50, 100)
    loss_ref.backward()
    ref = np.array([sd[k].grad.double().norm().item() for k in sorted(names)])
    ev_ref = ev_ref.detach()[..., ::4, ::4, ::4].clone()
    loss_ref = float(loss_ref.detach())
    del m
    for dtype, ltol, etol, gtol, gatol in ((torch.float32, FP32_BAR * abs(loss_ref), FP32_BAR, GRAD_L2_TOL, 1e-5),
                                           (torch.bfloat16, LOSS_BF16_BAR, BF16_EVIDENCE_MAX[model_name], 0.08, 2e-3)):
        m = make(dtype)
        ev, loss = m(x.cuda(), y.cuda(), 25, 'train', 'BraTS')
        loss.backward()
        check(abs(loss.item() - loss_ref), ltol, "%s loss %s" % (model_name, dtype))
        check(rel_err(ev.detach()[..., ::4, ::4, ::4], ev_ref), etol, "%s evidence %s" % (model_name, dtype))
        check(l2_rel_err(ev.detach()[..., ::4, ::4, ::4], ev_ref), BF16_BAR if dtype == torch.bfloat16 else FP32_BAR,
              "%s evidence L2 %s" % (model_name, dtype))
        params = dict(m.backbone.named_parameters())
        norms = np.array([params[k].grad.double().norm().item() for k in sorted(params)])
        # biases in front of a norm layer have a zero true gradient (rounding noise on both sides): the absolute
        # term, relative to the largest gradient norm of the network, covers them
        bad = np.abs(norms - ref) > gtol * ref + gatol * ref.max()
        assert not bad.any(), (dtype, [(k, n, r) for k, n, r, b in zip(sorted(params), norms, ref, bad) if b][:5])
        del m, ev, loss, params
        torch.cuda.empty_cache()
