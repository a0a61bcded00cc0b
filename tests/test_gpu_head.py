"""GPU parity of the fused evidential head/loss kernels vs golden vectors from the real reference."""
import numpy as np
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


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
    # reference evaluated on the bf16-rounded logits would differ by the rounding only
    assert abs(loss.item() - float(G["head3d_dce_e25_loss"])) < 2e-2
    assert zb.grad.dtype == torch.bfloat16 and torch.isfinite(zb.grad.float()).all()


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
