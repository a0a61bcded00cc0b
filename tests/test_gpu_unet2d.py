"""GPU parity: 2-D U-Nets (Unet2D / Unet2DDRIVE) + evidential loss vs golden vectors from the reference,
including the torch.rand skip padding drawn from the global CPU generator (UNet2DZoo.py:162-182)."""
import pytest
import torch

from tests.util import BF16_BAR, BF16_MAX_EXCEPTION, BF16_OVER_ROUNDING_POINTS, FP32_BAR, GRAD_L2_TOL, GRAD_MAX_TOL, LOSS_BF16_BAR, check, l2_rel_err, rel_err, sd_hash

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
    check(rel_err(logits, G[tag + "_logits"]), FP32_BAR, "logits")
    check(abs(loss.item() - float(G[tag + "_loss"])) / abs(float(G[tag + "_loss"])), FP32_BAR, "loss")
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
    check(rel_err(logits, G[tag + "_logits"]), BF16_MAX_EXCEPTION["2D"], "logits max-norm")
    check(l2_rel_err(logits, G[tag + "_logits"]), BF16_MAX_EXCEPTION["2D_L2"], "logits L2")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_isic_config0_vs_oracle_live(dtype):
    """BASELINE configs[0] at its full size: ISIC-shaped 1x3x224x300, K=2, TMSU forward + evidential loss (and the
    backward), against the CPU oracle run live on the same tensors.  W = 300 is not divisible by 16, so the
    torch.rand skip padding fires at merge6 and merge7 (UNet2DZoo.py:162-182): both sides draw it from the global
    CPU generator seeded alike."""
    import numpy as np
    from oracle import devis_oracle as O
    from devis_amd import TMSU
    from tests.util import brats_args
    torch.manual_seed(0)
    m = TMSU(brats_args(dataset='ISIC', num_classes=2, compute_dtype=dtype)).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.backbone.state_dict().items()}
    names = [k for k, _ in m.backbone.named_parameters()]
    g = torch.Generator().manual_seed(1)
    x = torch.randn((1, 3, 224, 300), generator=g)
    y = (torch.rand((1, 1, 224, 300), generator=g) > 0.5).long()
    torch.manual_seed(99)
    ev, loss = m(x.cuda(), y.cuda(), 25, 'train', 'ISIC')
    loss.backward()
    assert ev.shape == (1, 2, 224, 300) and torch.isfinite(loss)
    for k in names:
        sd[k].requires_grad_(True)
    torch.manual_seed(99)
    ev_ref, loss_ref = O.tmsu_forward(lambda t: O.unet2d_forward(sd, t, True), x, y, 25, 'train', 'ISIC', 2, 50, 100)
    loss_ref.backward()
    fp32 = dtype == torch.float32
    check(abs(loss.item() - loss_ref.item()) / (abs(loss_ref.item()) if fp32 else 1.0), FP32_BAR if fp32 else LOSS_BF16_BAR,
          "loss")
    check(rel_err(ev.detach(), ev_ref.detach()), FP32_BAR if fp32 else BF16_MAX_EXCEPTION["2D"], "evidence max-norm")
    check(l2_rel_err(ev.detach(), ev_ref.detach()), FP32_BAR if fp32 else BF16_BAR, "evidence L2")
    params = dict(m.backbone.named_parameters())
    norms = np.array([params[k].grad.double().norm().item() for k in sorted(names)])
    ref = np.array([sd[k].grad.double().norm().item() for k in sorted(names)])
    gtol, gatol = (GRAD_L2_TOL, 1e-6) if fp32 else (0.08, 1e-3)
    check(np.max((np.abs(norms - ref) - gatol * ref.max()) / (ref + 1e-30)), gtol, "grad norms")


def test_full_size_config2_train_step_vs_oracle_live():
    """BASELINE configs[1] at its full size (Unet2D(1,2), batch 16 of 1x256x256, epoch 25): loss, logits and every
    parameter's gradient norm of one train step against the CPU oracle run live on the same seeded tensors.
    fp32: loss 1e-5 rel, logits 1e-5 rel-max, gradient norms GRAD_L2_TOL; bf16: loss 1e-3 abs, logits 1e-2,
    gradient norms 8 % (north_star's bars; measured values are recorded)."""
    import numpy as np
    from oracle import devis_oracle as O
    from devis_amd import Unet2D
    from devis_amd.criterions import evidential_head_loss
    g = torch.Generator().manual_seed(2100)
    x = torch.randn((16, 1, 256, 256), generator=g)
    y = (torch.rand((16, 1, 256, 256), generator=g) > 0.5).long()
    torch.manual_seed(4)
    sd = {k: v.detach().clone() for k, v in Unet2D(1, 2, compute_dtype=torch.float32).state_dict().items()}
    names = [k for k in sd if sd[k].is_floating_point() and k.split('.')[-1] in ("weight", "bias")]
    for k in names:
        sd[k].requires_grad_(True)
    logits_ref = []
    def backbone(t):
        logits_ref.append(O.unet2d_forward(sd, t, True))
        return logits_ref[0]
    _, loss_ref = O.tmsu_forward(backbone, x, y, 25, 'train', 'HC', 2, 50, 100)
    loss_ref.backward()
    ref = np.array([sd[k].grad.double().norm().item() for k in sorted(names)])
    lref = logits_ref[0].detach()
    loss_ref = float(loss_ref.detach())
    for dtype, ltol, ztol, gtol, gatol in ((torch.float32, FP32_BAR * abs(loss_ref), FP32_BAR, GRAD_L2_TOL, 1e-6),
                                           (torch.bfloat16, LOSS_BF16_BAR, BF16_MAX_EXCEPTION["2D"], 0.08, 1e-3)):
        torch.manual_seed(4)
        net = Unet2D(1, 2, compute_dtype=dtype).cuda().train()
        logits_cl = net.forward_channels_last(x.cuda())
        _, loss = evidential_head_loss(logits_cl, y.cuda(), 25, 50, channels_last=True, want_evidence=False)
        loss.backward()
        check(abs(loss.item() - loss_ref), ltol, "loss %s" % dtype)
        check(rel_err(logits_cl.detach().permute(0, 3, 1, 2).float(), lref), ztol, "logits max-norm %s" % dtype)
        check(l2_rel_err(logits_cl.detach().permute(0, 3, 1, 2).float(), lref),
              FP32_BAR if dtype == torch.float32 else BF16_MAX_EXCEPTION["2D_L2"], "logits L2 %s" % dtype)
        params = dict(net.named_parameters())
        assert sorted(params) == sorted(names)
        norms = np.array([params[k].grad.double().norm().item() for k in sorted(params)])
        check(np.max((np.abs(norms - ref) - gatol * ref.max()) / (ref + 1e-30)), gtol, "grad norms %s" % dtype)


def test_bf16_error_is_what_the_rounding_points_cost_layer_by_layer():
    """configs[1] in bf16 misses north_star's 1e-2 on the logits (2.0e-2).  This test pins down WHY, so that the
    relaxed caps of tests/util.py are a measured property of bf16 on this network and not slack in the kernels: the
    fp32 oracle with bf16 roundings put in at the points where any bf16 pipeline has them (weights, stored tensors, conv
    operands: oracle.unet2d_forward_bf16_points) is as far from the fp32 oracle as devis_amd's bf16 path is - after
    every DoubleConv and ConvTranspose, and at the logits."""
    import os
    from tools.bf16_layer_audit import audit, fmt
    rows = audit(16, 256)
    text = fmt(rows, 16, 256)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "bf16_layer_audit.txt"), "w") as f:
        f.write(text + "\n")
    for name, ours_l2, pts_l2, diff_l2, ours_mx, pts_mx, diff_mx in rows:
        check(ours_l2 / pts_l2, BF16_OVER_ROUNDING_POINTS, "L2 ours / rounding points, " + name)
        check(ours_mx / pts_mx, 1.25, "max-norm ours / rounding points, " + name)       # one worst element: noisier
        check(diff_l2, max(0.5 * pts_l2, 1e-3), "ours vs rounding-point oracle, " + name)
    logits = rows[-1]
    check(logits[2], BF16_MAX_EXCEPTION["2D_L2"], "rounding points alone, logits L2 (recorded)")
