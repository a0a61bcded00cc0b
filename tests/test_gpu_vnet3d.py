"""GPU parity: 3-D V-Net + fused evidential head/loss against the goldens the real reference
produced through TMSU(model_name='V') (tests/golden/vnet.npz), plus the checkpoint round trip of
trainUMIS.py:326-330,402-405.  Bars as for the Unet (tests/util.py); where the reference's own fp32 gradient is
off its fp64 value by more than the bar, being at least as close to fp64 is accepted."""
import io

import numpy as np
import pytest
import torch

from tests.util import GRAD_L2_TOL, GRAD_MAX_TOL, brats_args, l2_rel_err, rel_err, sd_hash
from tests.util import BF16_BAR, FP32_BAR, LOSS_BF16_BAR, check

pytestmark = pytest.mark.gpu


def _make(seed, dtype):
    from devis_amd import TMSU
    torch.manual_seed(seed)
    m = TMSU(brats_args(model_name='V', compute_dtype=dtype))
    m.train()
    return m


def _inputs(seed, shape):
    g = torch.Generator().manual_seed(3000 + seed)
    x = torch.randn(shape, generator=g)
    y = torch.randint(0, 4, (shape[0],) + tuple(shape[2:]), generator=g)
    return x, y


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_vnet_train_step_fp32_vs_golden(golden, seed):
    G = golden("vnet")
    pre = "s%d_" % seed
    shape = tuple(int(v) for v in G[pre + "shape"])
    m = _make(seed, torch.float32)
    sd0 = {k: v.cpu() for k, v in m.backbone.state_dict().items()}
    assert list(sd0.keys()) == [str(k) for k in G[pre + "keys"]], "state_dict keys / order differ from the reference's"
    assert sd_hash(sd0) == str(G[pre + "sd_hash"]), "seeded init differs from the reference's"
    x, y = _inputs(seed, shape)
    ev, loss = m(x.cuda(), y.cuda(), 25, 'train', 'BraTS')
    loss.requires_grad_(True).backward()
    torch.cuda.synchronize()
    check(abs(loss.item() - float(G[pre + "loss"])) / abs(float(G[pre + "loss"])), FP32_BAR, "loss")
    check(rel_err(ev.detach()[..., ::2], G[pre + "evidence"]), FP32_BAR, "evidence")
    params = dict(m.backbone.named_parameters())
    for key in G.files:
        if not key.startswith(pre + "grad_"):
            continue
        k = key[len(pre + "grad_"):]
        g = params[k].grad
        g = g.reshape(-1)[::37] if g.numel() > 20000 else g
        ref = G[key]
        if k.endswith("conv.0.bias") and ("_dw" in k or "_up" in k) and ref.size <= 16:
            continue        # 16 channels in 16 groups: the bias in front of that GroupNorm has a zero true gradient
        if l2_rel_err(g, ref) < GRAD_L2_TOL and rel_err(g, ref) < GRAD_MAX_TOL:
            continue
        g64 = G[pre + "grad64_" + k]
        assert l2_rel_err(g, g64) <= max(2 * l2_rel_err(ref, g64), 1e-4), (k, l2_rel_err(g, ref), l2_rel_err(g, g64))
    with torch.no_grad():
        lg = m.backbone(x.cuda())
    check(rel_err(lg[..., ::2], G[pre + "logits"]), FP32_BAR, "logits")


def test_vnet_train_step_bf16_vs_golden(golden):
    G = golden("vnet")
    pre = "s0_"
    shape = tuple(int(v) for v in G[pre + "shape"])
    m = _make(0, torch.bfloat16)
    x, y = _inputs(0, shape)
    ev, loss = m(x.cuda(), y.cuda(), 25, 'train', 'BraTS')
    loss.backward()
    check(abs(loss.item() - float(G[pre + "loss"])), LOSS_BF16_BAR, "loss")
    from tests.util import BF16_MAX_EXCEPTION, l2_rel_err as l2
    check(rel_err(ev.detach()[..., ::2], G[pre + "evidence"]), BF16_MAX_EXCEPTION["V"], "evidence max-norm")
    check(l2(ev.detach()[..., ::2], G[pre + "evidence"]), BF16_BAR, "evidence L2")
    params = dict(m.backbone.named_parameters())
    norms = np.array([params[k].grad.double().norm().item() for k in sorted(params)])
    ref = G[pre + "gradnorm_all"]
    # atol 5e-3 of the largest norm: conv biases in front of a GroupNorm keep only the small part of their gradient
    # that the norm does not cancel (block_one_dw.conv.0.bias: 7.5e-3 vs 6.5e-3 in bf16)
    assert np.all(np.abs(norms - ref) <= 0.08 * ref + 5e-3 * ref.max()), np.max(np.abs(norms - ref) / (ref + 1e-12))


@pytest.mark.parametrize("name", ["U", "AU", "V"])
def test_checkpoint_round_trip(name):
    """{'epoch', 'state_dict'} as trainUMIS.py:326-330 saves it and :402-405 loads it: same keys ('backbone.*'),
    values restored bit-exactly, and the restored model computes the same logits."""
    from devis_amd import TMSU
    torch.manual_seed(3)
    a = TMSU(brats_args(model_name=name, compute_dtype=torch.float32)).eval()
    buf = io.BytesIO()
    torch.save({'epoch': 7, 'state_dict': a.state_dict()}, buf)
    buf.seek(0)
    ck = torch.load(buf, weights_only=True)
    torch.manual_seed(4)
    b = TMSU(brats_args(model_name=name, compute_dtype=torch.float32)).eval()
    b.load_state_dict(ck['state_dict'])
    assert ck['epoch'] == 7 and all(k.startswith('backbone.') for k in ck['state_dict'])
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    x = torch.randn((1, 4, 16, 16, 16), generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        assert torch.equal(a.backbone(x), b.backbone(x))
