"""GPU parity: Unetdrop (UNet3DZoo.py:308-372) against the goldens the real reference module
produced (tests/golden/unetdrop.npz).  The dropout noise is re-drawn from the same torch RNG state with the oracle's
helper and handed to the HIP path through ``Unetdrop.forward(x, masks=...)``."""
import numpy as np
import pytest
import torch

from tests.util import BF16_BAR, FP32_BAR, GRAD_L2_TOL, GRAD_MAX_TOL, check, l2_rel_err, rel_err, sd_hash

pytestmark = pytest.mark.gpu


def _case(golden, seed, dtype):
    from oracle import devis_oracle as O
    from devis_amd import Unetdrop
    G = golden("unetdrop")
    pre = "s%d_" % seed
    shape = tuple(int(v) for v in G[pre + "shape"])
    train = bool(int(G[pre + "train"]))
    torch.manual_seed(seed)
    net = Unetdrop(4, 16, 4, compute_dtype=dtype)
    assert sd_hash({k: v for k, v in net.state_dict().items()}) == str(G[pre + "sd_hash"])
    net.cuda().train(train)
    x = torch.randn(shape, generator=torch.Generator().manual_seed(5000 + seed))
    torch.manual_seed(1000 + seed)
    masks = O.unetdrop_masks(x.shape, train)
    return G, pre, net, x, masks, train


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_unetdrop_fp32_vs_golden(golden, seed):
    G, pre, net, x, masks, train = _case(golden, seed, torch.float32)
    y = net(x.cuda(), masks=masks)
    y.square().mean().backward()
    torch.cuda.synchronize()
    check(rel_err(y.detach(), G[pre + "logits"]), FP32_BAR, "logits")
    params = dict(net.named_parameters())
    for key in G.files:
        if not key.startswith(pre + "grad_"):
            continue
        k = key[len(pre + "grad_"):]
        g = params[k].grad
        g = g.reshape(-1)[::37] if g.numel() > 20000 else g
        ref = G[key]
        if l2_rel_err(g, ref) < GRAD_L2_TOL and rel_err(g, ref) < GRAD_MAX_TOL:
            continue
        # a ReLU gate within rounding of zero flipped (tests/util.py): be as close to the fp64 gradient as the reference is
        g64 = G[pre + "grad64_" + k]
        assert l2_rel_err(g, g64) <= 1.5 * l2_rel_err(ref, g64) + 1e-6, (k, l2_rel_err(g, ref), l2_rel_err(g, g64), l2_rel_err(ref, g64))
    norms = np.array([params[k].grad.double().norm().item() for k in sorted(params)])
    ref = G[pre + "gradnorm_all"]
    check(np.max((np.abs(norms - ref) - 1e-6 * ref.max()) / (ref + 1e-30)), GRAD_L2_TOL, "grad norms")
    if train:
        check(rel_err(net.state_dict()["DeBlock2.bn2.running_var"], G[pre + "bn_running_var"]), FP32_BAR, "running_var")


@pytest.mark.parametrize("seed", [0, 1])
def test_unetdrop_bf16_vs_golden(golden, seed):
    G, pre, net, x, masks, train = _case(golden, seed, torch.bfloat16)
    y = net(x.cuda(), masks=masks)
    y.square().mean().backward()
    check(rel_err(y.detach(), G[pre + "logits"]), 3 * BF16_BAR, "logits max-norm")     # 16^3 volumes: as the small V-Net goldens
    check(l2_rel_err(y.detach(), G[pre + "logits"]), 2 * BF16_BAR, "logits L2")
    params = dict(net.named_parameters())
    norms = np.array([params[k].grad.double().norm().item() for k in sorted(params)])
    ref = G[pre + "gradnorm_all"]
    assert np.all(np.abs(norms - ref) <= 0.1 * ref + 2e-3 * ref.max()), np.max(np.abs(norms - ref) / (ref + 1e-12))


def test_unetdrop_draws_its_own_noise():
    """forward(x) without masks: channel dropout is active in eval mode too (F.dropout3d's default training=True,
    UNet3DZoo.py:30), element dropout only in training mode; two calls differ."""
    from devis_amd import Unetdrop
    torch.manual_seed(3)
    net = Unetdrop(4, 16, 4).cuda().eval()
    x = torch.randn((1, 4, 16, 16, 16), generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        outs = [net(x) for _ in range(4)]
    assert all(torch.isfinite(o).all() for o in outs)
    assert any(not torch.equal(outs[0], o) for o in outs[1:])
