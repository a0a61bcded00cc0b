"""Fused DeUp_Cat tail (csrc/upcat.hip) against the two reference layers it replaces (UNet3DZoo.py:121-129):
ConvTranspose3d(C, C, 2, 2) -> cat((prev, up), 1) -> Conv3d(2C, C, 1), forward and every gradient."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import BF16_BAR, bf16_ulps, check, l2_rel_err, rel_err

pytestmark = pytest.mark.gpu


def cl(t):
    return t.permute(0, 2, 3, 4, 1).contiguous().bfloat16().cuda()


def uncl(t):
    return t.float().permute(0, 4, 1, 2, 3).contiguous().cpu()


@pytest.mark.parametrize("shape", [(2, 16, 4, 6, 16), (1, 16, 3, 5, 32), (1, 16, 3, 5, 16), (2, 16, 8, 8, 16), (2, 32, 3, 4, 16), (1, 32, 5, 2, 48)])
def test_upcat_forward_and_gradients(shape):
    from devis_amd import ops
    from devis_amd.unet3d import G1, G2S2
    n, c, d, h, w = shape
    g = torch.Generator().manual_seed(7)
    bf = lambda t: t.bfloat16().float()
    prev = bf(torch.randn((n, c, 2 * d, 2 * h, 2 * w), generator=g))
    t = bf(torch.randn((n, c, d, h, w), generator=g))
    w3 = torch.randn((c, 2 * c, 1, 1, 1), generator=g) * 0.2
    b3 = torch.randn((c,), generator=g)
    wt = torch.randn((c, c, 2, 2, 2), generator=g) * 0.3
    bt = torch.randn((c,), generator=g)
    dy = bf(torch.randn((n, c, 2 * d, 2 * h, 2 * w), generator=g))
    assert ops.upcat_supported(cl(prev), cl(t))
    m = ops.upcat_prepare(w3.cuda(), b3.cuda(), wt.cuda(), bt.cuda())
    y = uncl(ops.upcat_fwd(cl(prev), cl(t), m))

    # (1) exactly what the kernel evaluates: composite matrices rounded to bf16 once, fp32 accumulation, one rounding
    M = torch.einsum('om,cmp->poc', w3[:, c:, 0, 0, 0].double(), wt.reshape(c, c, 8).double()).float()      # [p][o][c]
    Mb, w3a = bf(M), bf(w3[:, :c, 0, 0, 0])
    bias = (b3.double() + w3[:, c:, 0, 0, 0].double() @ bt.double()).float()
    ref = F.conv3d(prev, w3a.view(c, c, 1, 1, 1))
    wt_eff = Mb.permute(2, 1, 0).reshape(c, c, 2, 2, 2).contiguous()     # ConvT weight [cin][cout][taps] = M[p][o][c]
    ref = ref + F.conv_transpose3d(t, wt_eff, stride=2) + bias.view(1, c, 1, 1, 1)
    check(bf16_ulps(y, ref), 1.0, "y vs composite reference [bf16 ulps]")
    # (2) the two reference layers in fp32 on the same bf16 inputs (the composite weights are rounded once more)
    P = dict(prev=prev.clone().requires_grad_(True), t=t.clone().requires_grad_(True), w3=w3.clone().requires_grad_(True),
             b3=b3.clone().requires_grad_(True), wt=wt.clone().requires_grad_(True), bt=bt.clone().requires_grad_(True))
    up = F.conv_transpose3d(P["t"], P["wt"], P["bt"], stride=2)
    yr = F.conv3d(torch.cat((P["prev"], up), 1), P["w3"], P["b3"])
    check(rel_err(y, yr.detach()), BF16_BAR, "y vs the two layers")
    yr.backward(dy)

    # backward pieces, as unet3d._Run._up_bwd strings them together
    dyc = cl(dy)
    dprev = uncl(ops.conv_dgrad(dyc, m["w3a"], G1, cl(prev).shape, wd=m["w3a_t"])[0])
    check(rel_err(dprev, P["prev"].grad), BF16_BAR, "d prev")
    dt = uncl(ops.conv_fwd(dyc, None, m["wq"], None, G2S2))
    check(rel_err(dt, P["t"].grad), BF16_BAR, "d t")
    # both in one pass over dy (same bf16 weights, fp32 accumulation: only the summation order differs)
    dprev_f, dt_f = ops.upcat_bwd_data(dyc, m)
    check(bf16_ulps(uncl(dprev_f), dprev), 1.0, "fused d prev vs the 1x1 dgrad kernel [bf16 ulps]")
    check(bf16_ulps(uncl(dt_f), dt), 1.0, "fused d t vs the k2-s2 gather kernel [bf16 ulps]")
    check(rel_err(uncl(dprev_f), P["prev"].grad), BF16_BAR, "fused d prev")
    check(rel_err(uncl(dt_f), P["t"].grad), BF16_BAR, "fused d t")
    f = dict(dtype=torch.float32, device="cuda")
    gA, s1 = ops.conv_wgrad(cl(prev), None, dyc, G1, (c, c, 1, 1, 1))
    gG, s2 = ops.convT_wgrad(cl(t), dyc, (2, 2, 2), (c, c, 2, 2, 2))
    out = (torch.empty((c, 2 * c, 1, 1, 1), **f), torch.empty((c,), **f), torch.empty((c, c, 2, 2, 2), **f), torch.empty((c,), **f))
    ops.upcat_grad_combine(gG, gA, s1, w3.cuda(), wt.cuda(), bt.cuda(), out)
    # fp32 sums of bf16 operands, fp32 algebra: 1e-4 of the largest entry
    for name, ours, ref_ in (("dw3", out[0], P["w3"].grad), ("db3", out[1], P["b3"].grad), ("dwt", out[2], P["wt"].grad),
                             ("dbt", out[3], P["bt"].grad)):
        check(rel_err(ours, ref_), 1e-4, name)
        check(l2_rel_err(ours, ref_), 1e-4, name + " L2")
    check(rel_err(s2, s1), 1e-5, "sum dy from the two weight-gradient kernels")


def test_upcat_unsupported_shapes_fall_back():
    from devis_amd import ops
    z = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device="cuda")
    assert not ops.upcat_supported(z(1, 4, 4, 24, 16), z(1, 2, 2, 12, 16))        # coarse W not a multiple of 16
    assert not ops.upcat_supported(z(1, 4, 4, 32, 8), z(1, 2, 2, 16, 8))          # channels
    assert not ops.upcat_supported(z(1, 4, 4, 32, 64), z(1, 2, 2, 16, 64))
    assert not ops.upcat_supported(z(1, 4, 4, 32, 16).float(), z(1, 2, 2, 16, 16).float())
