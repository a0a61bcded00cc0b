"""GPU parity: 2-D U-Nets (Unet2D / Unet2DDRIVE) + evidential loss vs golden vectors from the reference,
including the torch.rand skip padding drawn from the global CPU generator (UNet2DZoo.py:162-182)."""
import pytest
import torch

from tests.util import GRAD_L2_TOL, GRAD_MAX_TOL, l2_rel_err, rel_err, sd_hash

pytestmark = pytest.mark.gpu

CASES = [("u2d_w76", False, 3), ("u2d_sq", False, 1), ("drive_h", True, 3), ("drive_w", True, 3)]
GRADS = ("conv10.weight", "conv1.conv.0.weight", "up7.weight", "conv5.conv.4.weight", "conv9.conv.0.weight")


@pytest.mark.parametrize("tag,drive,cin", CASES)
def test_unet2d_fp32_vs_golden(golden, tag, drive, cin):
    from devis_amd import Unet2D, Unet2DDRIVE
    from devis_amd.criterions import evidential_head_loss
    G = golden("unet2d")
    shape = tuple(int(v) for v in G[tag + "_shape"])
    torch.manual_seed(3)
    net = (Unet2DDRIVE if drive else Unet2D)(cin, 2, compute_dtype=torch.float32)
    assert sd_hash(net.state_dict()) == str(G[tag + "_sd_hash"])
    net.cuda().train()
    g = torch.Generator().manual_seed(2000)
    x = torch.randn(shape, generator=g)
    y = (torch.rand((shape[0], 1) + shape[2:], generator=g) > 0.5).long()
    torch.manual_seed(99)                       # same global-RNG state as the reference run
    logits_cl = net.forward_channels_last(x.cuda())
    _, loss = evidential_head_loss(logits_cl, y.cuda(), 25, 50, channels_last=True, want_evidence=False)
    loss.backward()
    logits = logits_cl.detach().permute(0, 3, 1, 2)
    assert rel_err(logits, G[tag + "_logits"]) < 5e-5
    assert abs(loss.item() - float(G[tag + "_loss"])) < 1e-5
    params = dict(net.named_parameters())
    for k in GRADS:
        # 2-D golden images are tiny (conv5 normalises over 16-32 values) and add MaxPool argmax gates to the ReLU
        # gates discussed in tests/util.py: one flipped gate moves a weight gradient by up to ~1e-2 here.
        assert l2_rel_err(params[k].grad, G[tag + "_grad_" + k]) < 4 * GRAD_L2_TOL and rel_err(params[k].grad, G[tag + "_grad_" + k]) < 4 * GRAD_MAX_TOL, k


def test_unet2d_bf16(golden):
    from devis_amd import Unet2D
    G = golden("unet2d")
    tag = "u2d_sq"
    shape = tuple(int(v) for v in G[tag + "_shape"])
    torch.manual_seed(3)
    net = Unet2D(1, 2, compute_dtype=torch.bfloat16).cuda().train()
    g = torch.Generator().manual_seed(2000)
    x = torch.randn(shape, generator=g)
    logits = net(x.cuda())
    assert rel_err(logits, G[tag + "_logits"]) < 3e-2


def test_isic_plumbing_shape():
    """configs[0]: ISIC-shaped 3x224x300, K=2 - the rand padding fires at merge6 and merge7."""
    from devis_amd import TMSU
    from tests.util import brats_args
    torch.manual_seed(0)
    m = TMSU(brats_args(dataset='ISIC', num_classes=2, compute_dtype=torch.float32)).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn((1, 3, 224, 300), generator=g).cuda()
    y = (torch.rand((1, 1, 224, 300), generator=g) > 0.5).long().cuda()
    ev, loss = m(x, y, 25, 'train', 'ISIC')
    loss.backward()
    assert ev.shape == (1, 2, 224, 300) and torch.isfinite(loss)
