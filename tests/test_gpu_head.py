"""GPU parity of the fused evidential head/loss kernels vs golden vectors from the real reference."""
import numpy as np
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _ncx(t):     # channels-last -> NC...
    return t.permute(0, t.dim() - 1, *range(1, t.dim() - 1)).contiguous()


def _cl(t):      # NC... -> channels-last contiguous
    perm = (0,) + tuple(range(2, t.dim())) + (1,)
    return t.permute(*perm).contiguous()


@pytest.mark.parametrize("tag,K", [("head3d", 4), ("head2d", 2)])
@pytest.mark.parametrize("epoch", [1, 25, 50, 80])
@pytest.mark.parametrize("layout", ["ncx", "cl"])
def test_fused_loss_fp32(golden, tag, K, epoch, layout):
    from devis_amd.criterions import evidential_head_loss
    G = golden("head")
    z = torch.from_numpy(G[tag + "_logits"]).cuda()
    lab = torch.from_numpy(G[tag + "_labels"]).cuda()
    zin = (_cl(z) if layout == "cl" else z.clone()).requires_grad_(True)
    ev, loss = evidential_head_loss(zin, lab, epoch, 50, channels_last=(layout == "cl"))
    loss.backward()
    pre = "%s_dce_e%d_" % (tag, epoch)
    assert abs(loss.item() - float(G[pre + "loss"])) < 2e-6 * max(1.0, abs(float(G[pre + "loss"])))
    assert rel_err(ev, G[tag + "_evidence"]) < 1e-6
    grad = zin.grad
    if layout == "cl":
        perm = (0, grad.dim() - 1) + tuple(range(1, grad.dim() - 1))
        grad = grad.permute(*perm)
    assert rel_err(grad, G[pre + "grad"]) < 2e-5


@pytest.mark.parametrize("label_dtype", [torch.int64, torch.int32, torch.uint8])
def test_label_dtypes_and_bf16(golden, label_dtype):
    from devis_amd.criterions import evidential_head_loss
    G = golden("head")
    z = torch.from_numpy(G["head3d_logits"]).cuda()
    lab = torch.from_numpy(G["head3d_labels"]).cuda().to(label_dtype)
    zb = _cl(z).to(torch.bfloat16).requires_grad_(True)
    ev, loss = evidential_head_loss(zb, lab, 25, 50, channels_last=True)
    loss.backward()
    assert zb.grad.dtype == torch.bfloat16 and torch.isfinite(zb.grad.float()).all()
    # The kernel reads bf16 logits and computes in fp32, so against the oracle evaluated on the SAME rounded logits the
    # loss meets the bf16 loss bar (1e-3 relative) and the gradient is the oracle's rounded to bf16 once (2^-8 of the
    # largest element); against the golden loss of the unrounded logits the difference is the input rounding.
    from oracle.devis_oracle import tmsu_forward
    from tests.util import LOSS_BF16_BAR, check
    zq = z.to(torch.bfloat16).float().cpu().requires_grad_(True)
    _, want = tmsu_forward(lambda t: t, zq, lab.cpu().to(torch.int64), 25, 'train', 'BraTS', 4, 50, 81)
    want.backward()
    check(abs(loss.item() - want.item()) / abs(want.item()), LOSS_BF16_BAR, "head bf16 loss vs oracle on the rounded logits (%s)" % label_dtype)
    check(rel_err(_ncx(zb.grad.float()), zq.grad), 2.0 ** -8, "head bf16 dlogits vs oracle on the rounded logits")
    assert abs(loss.item() - float(G["head3d_dce_e25_loss"])) < 2e-2


def test_infer_head_bit_exact_masks(golden):
    """argmax mask and uncertainty-threshold masks: bit-exact in fp32."""
    from devis_amd import ops
    from oracle.devis_oracle import UNCERTAINTY_THRESHOLDS
    G = golden("head")
    for tag, K in (("head3d", 4), ("head2d", 2)):
        z = torch.from_numpy(G[tag + "_logits"]).cuda()
        out = ops.evid_infer(z, False, UNCERTAINTY_THRESHOLDS, want_evidence=True, want_mne=True)
        assert np.array_equal(out["pred"].cpu().numpy(), G[tag + "_pred"])
        u_ref = G[tag + "_u"]
        assert rel_err(out["u"], u_ref) < 1e-6
        assert rel_err(out["mne"], G[tag + "_mne"]) < 1e-5
        m = out["umask"].cpu().numpy().astype(np.uint16)
        for i, t in enumerate(UNCERTAINTY_THRESHOLDS):
            ref = (u_ref > np.float32(t))
            near = np.abs(u_ref - np.float32(t)) < 1e-6          # guard band: none expected in the fixture
            got = ((m >> i) & 1).astype(bool).reshape(ref.shape)
            assert not near.any()
            assert np.array_equal(got, ref), "threshold %g" % t


def test_full_size_properties():
    """BASELINE full size (2 x 4 x 128^3): size-independent properties of the fused loss."""
    from devis_amd.criterions import evidential_head_loss
    g = torch.Generator().manual_seed(3)
    z = torch.randn((2, 128, 128, 128, 4), generator=g).cuda()
    lab = torch.randint(0, 4, (2, 128, 128, 128), generator=g).cuda()
    _, l_all = evidential_head_loss(z, lab, 25, 50, channels_last=True, want_evidence=False)
    # permutation invariance over voxels (loss is a mean + global sums)
    perm = torch.randperm(2 * 128 ** 3, generator=g).cuda()
    zp = z.view(-1, 4)[perm].view_as(z).contiguous()
    lp = lab.view(-1)[perm].view_as(lab).contiguous()
    _, l_perm = evidential_head_loss(zp, lp, 25, 50, channels_last=True, want_evidence=False)
    assert abs(l_all.item() - l_perm.item()) < 1e-5
    # lambda = 0 and lambda = 1 bracket lambda = 0.5 linearly
    _, l0 = evidential_head_loss(z, lab, 0, 50, channels_last=True, want_evidence=False)
    _, l1 = evidential_head_loss(z, lab, 50, 50, channels_last=True, want_evidence=False)
    assert abs(0.5 * (l0.item() + l1.item()) - l_all.item()) < 1e-5
    # determinism: two runs agree bit for bit (two-stage reduction, no atomics)
    _, l_again = evidential_head_loss(z, lab, 25, 50, channels_last=True, want_evidence=False)
    assert l_again.item() == l_all.item()


# --------------------------------------------------------------------------- Uncertainty_Loss variant
@pytest.mark.parametrize("tag,K", [("head3d", 4), ("head2d", 2)])
@pytest.mark.parametrize("epoch", [1, 25, 50, 80])
@pytest.mark.parametrize("layout", ["ncx", "cl"])
def test_fused_uloss_fp32(golden, tag, K, epoch, layout):
    """dce_evidence_u_loss (criterions.py:373-419) fused with the head, vs the reference's own numbers."""
    from devis_amd.criterions import evidential_head_loss
    G = golden("head")
    z = torch.from_numpy(G[tag + "_logits"]).cuda()
    lab = torch.from_numpy(G[tag + "_labels"]).cuda()
    zin = (_cl(z) if layout == "cl" else z.clone()).requires_grad_(True)
    ev, loss = evidential_head_loss(zin, lab, epoch, 50, channels_last=(layout == "cl"), u_loss=True,
                                    total_step=100, eps=1e-10, disentangle=False)
    loss.backward()
    pre = "%s_dceu_e%d_" % (tag, epoch)
    assert abs(loss.item() - float(G[pre + "loss"])) < 2e-6 * max(1.0, abs(float(G[pre + "loss"])))
    assert rel_err(ev, G[tag + "_evidence"]) < 1e-6
    grad = zin.grad
    if layout == "cl":
        perm = (0, grad.dim() - 1) + tuple(range(1, grad.dim() - 1))
        grad = grad.permute(*perm)
    assert rel_err(grad, G[pre + "grad"]) < 2e-5


@pytest.mark.parametrize("variant", ["dce", "dceu"])
@pytest.mark.parametrize("tag,K", [("head3d", 4), ("head2d", 2)])
def test_reference_signature_losses(golden, variant, tag, K):
    """dce_evidence_loss / dce_evidence_u_loss called exactly as TMSU.forward calls them in the reference
    (separate alpha and evidence tensors): the (NV,1) result and the gradient reaching the logits."""
    from devis_amd import criterions as crit
    G = golden("head")
    fn = crit.dce_evidence_loss if variant == "dce" else crit.dce_evidence_u_loss
    lab = torch.from_numpy(G[tag + "_labels"]).cuda()
    for epoch in (1, 25, 80):
        z = torch.from_numpy(G[tag + "_logits"]).cuda().requires_grad_(True)
        ev = torch.nn.functional.softplus(z)
        al = ev + 1
        pv = fn(lab, al, K, epoch, 50, 100, 1e-10, False, ev, None)
        pre = "%s_%s_e%d_" % (tag, variant, epoch)
        assert pv.shape == G[pre + "per_voxel"].shape
        assert rel_err(pv.detach(), G[pre + "per_voxel"]) < 1e-5
        pv.mean().backward()
        assert rel_err(z.grad, G[pre + "grad"]) < 2e-5


def test_kl_ce_mse_vs_golden_and_oracle_grads(golden):
    from devis_amd import criterions as crit
    from oracle import devis_oracle as O
    G = golden("head")
    # KL(alpha, c), criterions.py:117-127
    for tag, K in (("head3d", 4), ("head2d", 2)):
        al = torch.from_numpy(G[tag + "_evidence"]) + 1
        rows = O._rows(al)
        a_gpu = rows.cuda().requires_grad_(True)
        kl = crit.KL(a_gpu, K)
        # lgamma(S) ~ 450 here and KL ~ 2 is a difference of such terms: one fp32 ulp of them is 3e-5
        assert kl.shape == G[tag + "_KL"].shape and rel_err(kl.detach(), G[tag + "_KL"]) < 5e-5
        w = torch.linspace(0.5, 1.5, kl.numel()).view_as(kl)
        (kl * w.cuda()).sum().backward()
        a_cpu = rows.clone().requires_grad_(True)
        (O.kl_dirichlet(a_cpu, K) * w).sum().backward()
        assert rel_err(a_gpu.grad, a_cpu.grad) < 2e-5
    # ce_loss (N,C,D,H,W) and mse_loss (rows)
    al = torch.from_numpy(G["ce_alpha"])
    lab = torch.from_numpy(G["ce_labels"])
    a_gpu = al.cuda().requires_grad_(True)
    ce = crit.ce_loss(lab.cuda(), a_gpu, 4, 10, 50)
    assert ce.shape == G["ce_loss"].shape and rel_err(ce.detach(), G["ce_loss"]) < 1e-5
    ce.sum().backward()
    a_cpu = al.clone().requires_grad_(True)
    O.ce_loss(lab, a_cpu, 4, 10, 50).sum().backward()
    assert rel_err(a_gpu.grad, a_cpu.grad) < 2e-5
    rows, labr = O._rows(al), lab.view(-1)
    r_gpu = rows.cuda().requires_grad_(True)
    mse = crit.mse_loss(labr.cuda(), r_gpu, 4, 10, 50)
    assert mse.shape == G["mse_loss"].shape and rel_err(mse.detach(), G["mse_loss"]) < 1e-5
    mse.sum().backward()
    r_cpu = rows.clone().requires_grad_(True)
    O.mse_loss(labr, r_cpu, 4, 10, 50).sum().backward()
    assert rel_err(r_gpu.grad, r_cpu.grad) < 2e-5


def test_uloss_disentangle_and_separate_evidence_vs_oracle(golden):
    """disentangle=True branch (criterions.py:404-407) and an `evidence` tensor that is NOT alpha-1:
    both gradients (to alpha and to evidence) against the oracle's autograd, run live."""
    from devis_amd import criterions as crit
    from oracle import devis_oracle as O
    G = golden("head")
    z = torch.from_numpy(G["head3d_logits"])
    lab = torch.from_numpy(G["head3d_labels"])
    g = torch.Generator().manual_seed(21)
    ev0 = torch.rand(z.shape, generator=g) * 3
    for fn_g, fn_o in ((crit.dce_evidence_u_loss, O.dce_evidence_u_loss), (crit.dce_evidence_loss, O.dce_evidence_loss)):
        for dis in (True, False):
            al_c = (torch.nn.functional.softplus(z) + 1).requires_grad_(True)
            ev_c = ev0.clone().requires_grad_(True)
            ref = fn_o(lab, al_c, 4, 30, 50, 100, 1e-10, dis, ev_c, None)
            ref.mean().backward()
            al_g = al_c.detach().cuda().requires_grad_(True)
            ev_g = ev0.cuda().requires_grad_(True)
            out = fn_g(lab.cuda(), al_g, 4, 30, 50, 100, 1e-10, dis, ev_g, None)
            out.mean().backward()
            assert rel_err(out.detach(), ref.detach()) < 1e-5
            assert rel_err(al_g.grad, al_c.grad) < 2e-5
            assert rel_err(ev_g.grad, ev_c.grad) < 2e-5


def test_tmsu_uncertainty_loss_train_step():
    """TMSU(Uncertainty_Loss=True) end to end vs the oracle (TrustworthySeg.py:124-158 with u_loss)."""
    from oracle import devis_oracle as O
    from devis_amd import TMSU
    from tests.util import brats_args, GRAD_L2_TOL, l2_rel_err
    args = brats_args(compute_dtype=torch.float32)
    args.Uncertainty_Loss = True
    torch.manual_seed(4)
    m = TMSU(args)
    m.train()
    sd = {k: v.detach().cpu().clone() for k, v in m.backbone.state_dict().items()}
    for k in sd:
        if sd[k].is_floating_point() and k.split('.')[-1] in ("weight", "bias"):
            sd[k].requires_grad_(True)
    g = torch.Generator().manual_seed(40)
    x = torch.randn((1, 4, 16, 16, 16), generator=g)
    y = torch.randint(0, 4, (1, 16, 16, 16), generator=g)
    ev, loss = m(x.cuda(), y.cuda(), 30, 'train', 'BraTS')
    loss.backward()
    ev_o, loss_o = O.tmsu_forward(lambda t: O.unet3d_forward(sd, t, True), x, y, 30, 'train', 'BraTS', 4, 50, 100,
                                  u_loss=True)
    loss_o.backward()
    assert abs(loss.item() - loss_o.item()) < 1e-5 * abs(loss_o.item()) + 1e-6
    assert rel_err(ev.detach(), ev_o.detach()) < 2e-5
    params = dict(m.backbone.named_parameters())
    for k in ("endconv.weight", "DeBlock2.conv2.weight", "EnBlock1.conv1.weight"):
        assert l2_rel_err(params[k].grad, sd[k].grad) < GRAD_L2_TOL, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_k4_fast_path_honours_the_sample_stride(dtype):
    """The C ABI takes explicit strides: K=4 channels-last logits whose sample stride is larger than 4 V (a padded
    batch buffer) must give the contiguous call's loss sums and gradient, through the straight-line K=4 kernels
    (k_dce4_*), and leave the padding alone (they indexed samples as n*V*4)."""
    import ctypes as C
    from devis_amd import _lib, ops
    lib = _lib.load()
    n, v, pad = 3, 1000, 24
    g = torch.Generator().manual_seed(11)
    z = (torch.randn((n, v, 4), generator=g) * 3).to(dtype).cuda()
    lab = torch.randint(0, 4, (n, v), generator=g).cuda()
    one = torch.ones((), device="cuda")
    scal_c, ev_c, _ = ops.evid_loss_fwd(z, True, lab, 0.5, True)
    dz_c = ops.evid_loss_bwd(z, True, lab, 0.5, scal_c, one)
    zp = torch.full((n, v + pad, 4), 7.0, dtype=dtype, device="cuda")
    zp[:, :v] = z
    dzp = torch.full_like(zp, -5.0)
    sn = (v + pad) * 4
    scal = torch.empty_like(scal_c)
    ev = torch.empty_like(ev_c)
    nb = lib.devis_evid_workspace_bytes(n, v, 4)
    ws = ops.workspace(nb, z.device)
    st = torch.cuda.current_stream().cuda_stream
    dt = _lib.DEVIS_F32 if dtype == torch.float32 else _lib.DEVIS_BF16
    _lib.check(lib.devis_evid_loss_fwd(zp.data_ptr(), sn, 4, 1, dt, lab.data_ptr(), _lib.LABEL_I64, n, v, 4, C.c_float(0.5),
                                       ev.data_ptr(), None, scal.data_ptr(), ws.data_ptr(), nb, st), "fwd")
    _lib.check(lib.devis_evid_loss_bwd(zp.data_ptr(), sn, 4, 1, dt, lab.data_ptr(), _lib.LABEL_I64, n, v, 4, C.c_float(0.5),
                                       scal.data_ptr(), one.data_ptr(), None, None, None, dzp.data_ptr(), st), "bwd")
    torch.cuda.synchronize()
    assert torch.equal(scal[:16], scal_c[:16])          # 4 scalars + I_k, P_k, R_k of K = 4; the rest is never written
    assert torch.equal(ev, ev_c)
    assert torch.equal(dzp[:, :v], dz_c)
    assert (dzp[:, v:] == -5.0).all()
