"""FusedAdam (devis_adam_step) against torch.optim.Adam: the optimizer the reference builds at
trainUMIS.py:1394 (coupled L2 weight decay 1e-5) and re-tunes every iteration (:217-219).
Bar: fp32, 2e-6 relative on parameters and both moments after several steps (the kernel follows torch's
operation order; the residue is fused-multiply-add rounding)."""
import pytest
import torch

from tests.util import rel_err, brats_args

pytestmark = pytest.mark.gpu

TOL = 2e-6


def _params(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) for s in shapes]


@pytest.mark.parametrize("shapes", [[(5,), (3, 7), (2, 3, 3, 3, 3)], [(1,)], [(1024, 33), (17,)]])
def test_fused_adam_matches_torch(shapes):
    from devis_amd.optim import FusedAdam
    from devis_amd.train import adjust_learning_rate
    init = _params(shapes, 0)
    ref = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    mine = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    o_ref = torch.optim.Adam(ref, lr=0.002, weight_decay=1e-5)
    o_mine = FusedAdam(mine, lr=0.002, weight_decay=1e-5)
    for it in range(6):
        adjust_learning_rate(o_ref, 10 + it, 99, 0.002)
        adjust_learning_rate(o_mine, 10 + it, 99, 0.002)
        grads = _params(shapes, 100 + it)
        for p, q, g in zip(ref, mine, grads):
            p.grad = g.clone().cuda()
            q.grad = g.clone().cuda()
        o_ref.step()
        o_mine.step()
    for p, q in zip(ref, mine):
        assert rel_err(q.data, p.data) < TOL
        st = o_ref.state[p]
        o = [o for r, o in o_mine._layout if r is q][0]
        assert rel_err(o_mine.exp_avg[o:o + q.numel()].view(q.shape), st['exp_avg']) < TOL
        assert rel_err(o_mine.exp_avg_sq[o:o + q.numel()].view(q.shape), st['exp_avg_sq']) < TOL


def test_fused_adam_rejects_a_missing_gradient():
    """torch.optim.Adam skips a parameter whose .grad is None; the one-launch step cannot, so it must say so."""
    from devis_amd.optim import FusedAdam
    ps = [torch.nn.Parameter(t.cuda()) for t in _params([(5,), (3, 7)], 0)]
    opt = FusedAdam(ps, lr=0.002)
    ps[0].grad = torch.ones_like(ps[0])
    with pytest.raises(RuntimeError, match="no gradient"):
        opt.step()


def test_fused_adam_on_the_backbone_reads_the_flat_gradient():
    """One train step of TMSU with FusedAdam == the same step with torch.optim.Adam (fp32 kernels; the gradients
    are bit-identical, so only the update arithmetic differs).  A second step is not compared: conv biases in
    front of a BatchNorm have a zero true gradient, Adam normalises their rounding noise to +-lr, and a 1-ulp
    difference after step one then shows up as a 2*lr difference in those biases."""
    from devis_amd import TMSU
    from devis_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 4, 16, 16, 16), generator=g).cuda()
    y = torch.randint(0, 4, (2, 16, 16, 16), generator=g).cuda()
    out = []
    for fused in (False, True):
        torch.manual_seed(0)
        m = TMSU(brats_args(compute_dtype=torch.float32)).train()
        opt = (FusedAdam(m.parameters(), lr=0.002, weight_decay=1e-5, module=m) if fused
               else torch.optim.Adam(m.parameters(), lr=0.002, weight_decay=1e-5))
        for _ in range(1):
            _, loss = m(x, y, 25, 'train', 'BraTS')
            opt.zero_grad()
            loss.backward()
            if fused:      # the step must have found every .grad inside the backbone's flat buffer
                assert opt._flat_grad() is m.backbone.last_flat_grad
            opt.step()
        out.append(({k: v.detach().clone() for k, v in m.state_dict().items()}, loss.item()))
    (sd0, l0), (sd1, l1) = out
    assert abs(l0 - l1) < 1e-5 * abs(l0)
    for k in sd0:
        if sd0[k].is_floating_point():
            if ".DeBlock" in k and k.endswith(("conv1.bias", "conv2.bias")):
                # a conv bias in front of a BatchNorm has a zero true gradient; what backward leaves is
                # rounding noise (|g| ~ 1e-9, not reproducible to the bit between two runs of the fp32
                # kernels) that Adam normalises to +-lr.  The reference has the same property.
                assert (sd1[k] - sd0[k]).abs().max().item() <= 2 * 0.002 * 1.0001, k
                continue
            assert rel_err(sd1[k], sd0[k]) < 2e-5, k
        else:
            assert torch.equal(sd0[k], sd1[k]), k             # num_batches_tracked
            assert int(sd0[k]) == 1


def test_train_loop_logs_every_step_without_per_step_sync():
    """devis_amd.train.train(): one line per iteration, in order, with the same values as loss.item() gives, and the
    epoch average of those losses (LossMeter defers the host copies instead of syncing every step)."""
    from devis_amd import TMSU
    from devis_amd.optim import FusedAdam
    from devis_amd.train import train
    from tests.util import brats_args
    args = brats_args(compute_dtype=torch.bfloat16)
    args.lr, args.mode = 0.002, 'train'
    g = torch.Generator().manual_seed(3)
    data = [(torch.randn((1, 4, 32, 32, 32), generator=g), torch.randint(0, 4, (1, 32, 32, 32), generator=g))
            for _ in range(12)]

    def run(log):
        torch.manual_seed(0)
        m = TMSU(args)
        opt = FusedAdam(m.parameters(), lr=args.lr, weight_decay=1e-5, module=m)
        return train(args, m, opt, 25, data, log=log)

    lines = []
    avg = run(lines.append)
    assert [int(l.split(",")[0]) for l in lines] == list(range(1, 13))
    vals = [float(l.split(":")[1]) for l in lines]
    assert abs(avg - sum(vals) / 12) < 2e-3          # the lines are rounded to 3 decimals
    assert abs(run(None) - avg) < 1e-6               # no logging: same training, same average
