"""GPU parity: the 2-D V-Net TMSU selects for ISIC / DRIVE / TOAR / HC with model_name='V' (SURVEY.md §8(f)-4;
models/lib/VNet2D.py:145-265, TrustworthySeg.py:54,78,91,104) against goldens written from the REAL
TMSU(dataset='DRIVE', model_name='V') (tests/golden/make_goldens.py --only vnet2d).  Case s1 has a width that is
not divisible by 16: offsetAdd pads with torch.rand from the global CPU generator, seeded alike on both sides."""
import numpy as np
import pytest
import torch

from tests.util import (BF16_BAR, FP32_BAR, GRAD_L2_TOL, GRAD_MAX_TOL, LOSS_BF16_BAR, brats_args, check, l2_rel_err,
                        rel_err, sd_hash)

pytestmark = pytest.mark.gpu


def _make(seed, dtype):
    from devis_amd import TMSU
    torch.manual_seed(seed)
    return TMSU(brats_args(dataset='DRIVE', model_name='V', num_classes=2, compute_dtype=dtype)).train()


def _inputs(seed, shape):
    g = torch.Generator().manual_seed(5000 + seed)
    x = torch.randn(shape, generator=g)
    y = (torch.rand((shape[0], 1) + tuple(shape[2:]), generator=g) > 0.5).long()
    return x, y


@pytest.mark.parametrize("seed", [0, 1])
def test_vnet2d_train_step_fp32_vs_golden(golden, seed):
    G = golden("vnet2d")
    pre = "s%d_" % seed
    shape = tuple(int(v) for v in G[pre + "shape"])
    m = _make(seed, torch.float32)
    sd0 = {k: v.cpu() for k, v in m.backbone.state_dict().items()}
    assert list(sd0.keys()) == [str(k) for k in G[pre + "keys"]], "state_dict keys / order differ from the reference's"
    assert sd_hash(sd0) == str(G[pre + "sd_hash"]), "seeded init differs from the reference's"
    x, y = _inputs(seed, shape)
    torch.manual_seed(99)
    ev, loss = m(x.cuda(), y.cuda(), 25, 'train', 'DRIVE')
    loss.requires_grad_(True).backward()
    torch.cuda.synchronize()
    check(abs(loss.item() - float(G[pre + "loss"])) / abs(float(G[pre + "loss"])), FP32_BAR, "loss")
    check(rel_err(ev.detach(), G[pre + "evidence"]), FP32_BAR, "evidence")
    params = dict(m.backbone.named_parameters())
    for key in G.files:
        if not key.startswith(pre + "grad_"):
            continue
        k = key[len(pre + "grad_"):]
        g = params[k].grad
        g = g.reshape(-1)[::37] if g.numel() > 20000 else g
        ref = G[key]
        # tiny 2-D golden images (the deep stages normalise over 16-64 values): as in tests/test_gpu_unet2d.py one
        # flipped ReLU gate moves a weight gradient by up to ~1e-2, hence 4x the 3-D bars
        if l2_rel_err(g, ref) < 4 * GRAD_L2_TOL and rel_err(g, ref) < 4 * GRAD_MAX_TOL:
            continue
        # a ReLU gate within rounding of zero flipped (tests/util.py): then we must be as close to the fp64 gradient
        # as the reference's own fp32 run is
        g64 = G[pre + "grad64_" + k]
        assert l2_rel_err(g, g64) <= max(2 * l2_rel_err(ref, g64), 1e-4), (k, l2_rel_err(g, ref), l2_rel_err(g, g64))
    m.eval()
    with torch.no_grad():
        torch.manual_seed(99)
        lg = m.backbone(x.cuda())
    check(rel_err(lg, G[pre + "logits"]), FP32_BAR, "logits")


@pytest.mark.parametrize("seed", [0, 1])
def test_vnet2d_train_step_bf16_vs_golden(golden, seed):
    G = golden("vnet2d")
    pre = "s%d_" % seed
    shape = tuple(int(v) for v in G[pre + "shape"])
    m = _make(seed, torch.bfloat16)
    x, y = _inputs(seed, shape)
    torch.manual_seed(99)
    ev, loss = m(x.cuda(), y.cuda(), 25, 'train', 'DRIVE')
    loss.backward()
    check(abs(loss.item() - float(G[pre + "loss"])), LOSS_BF16_BAR, "loss")
    check(rel_err(ev.detach(), G[pre + "evidence"]), BF16_BAR, "evidence")
    params = dict(m.backbone.named_parameters())
    norms = np.array([params[k].grad.double().norm().item() for k in sorted(params)])
    ref = G[pre + "gradnorm_all"]
    check(np.max((np.abs(norms - ref) - 5e-3 * ref.max()) / (ref + 1e-12)), 0.08, "grad norms")


def test_tmsu_selects_vnet2d_for_the_2d_datasets():
    from devis_amd import TMSU, VNet2D
    for ds, k in (("ISIC", 2), ("DRIVE", 2), ("TOAR", 7), ("HC", 10)):
        m = TMSU(brats_args(dataset=ds, model_name='V', num_classes=k))
        assert isinstance(m.backbone, VNet2D) and m.backbone.out_conv.out_channels == k
        assert m.backbone.block_one.conv[0].in_channels == 3            # TrustworthySeg.py:54,78,91,104: always 3
    with pytest.raises(NameError):
        TMSU(brats_args(dataset='DRIVE', model_name='ResU', num_classes=2))


def test_vnet2d_mismatched_up_first_widths_raise():
    """offsetAdd(up, skip) pads the SKIP when widths differ, which cannot add up in the reference either
    (VNet2D.py:220-224): W = 40 -> x2 is 20 wide... fine; W = 44 gives x3 = 11 -> x7_up = 22 = x2: fine; W = 36:
    x2 = 18, x3 = 9, x4 = 4 -> x6_up 8 vs x3 9 padded, x7_up = 18: fine.  W = 34: x2 = 17, x7_up = 16 -> raises."""
    from devis_amd import VNet2D
    net = VNet2D(3, 2, 32, compute_dtype=torch.float32).cuda()
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 32, 34).cuda())
