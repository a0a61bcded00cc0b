"""Launch plans (devis_amd/launch_plan.py): a replayed training step must give the bits of the eager step - same
kernels, same arguments, same order - for several steps in a row, with new inputs at new addresses, and fall back to
eager where a replay would be wrong (two forwards before a backward, gradient accumulation)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(dtype):
    import argparse
    return argparse.Namespace(num_classes=4, dataset='BraTS', model_name='U', input_modality='four', end_epochs=99,
                              lambda_epochs=50, Uncertainty_Loss=False, compute_dtype=dtype, lr=0.002)


def _train(steps, plan, dtype=torch.bfloat16, shape=(2, 4, 32, 32, 32), model_name='U'):
    """losses, final parameters and BatchNorm buffers of `steps` Adam steps on fresh inputs each step"""
    from devis_amd import TMSU, launch_plan
    from devis_amd.optim import FusedAdam
    launch_plan.ENABLED = plan
    try:
        torch.manual_seed(0)
        a = _args(dtype)
        a.model_name = model_name
        model = TMSU(a)
        model.train()
        opt = FusedAdam(model.parameters(), lr=0.002, weight_decay=1e-5, module=model)
        g = torch.Generator().manual_seed(3)
        losses = []
        for i in range(steps):
            x = torch.randn(shape, generator=g).cuda()              # a new tensor (a new address) every step
            y = torch.randint(0, 4, (shape[0],) + shape[2:], generator=g).cuda()
            junk = torch.empty((1 + 37 * i,), device="cuda")         # shift the allocator's state between steps
            _, loss = model(x, y, 25, 'train', 'BraTS')
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.item())
            del junk
        plans = getattr(model.backbone, "_plans", {})
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        return losses, sd, plans
    finally:
        launch_plan.ENABLED = True


@pytest.mark.parametrize("model_name", ["U", "AU", "V"])
def test_replayed_steps_equal_eager_steps(model_name):
    l0, sd0, p0 = _train(5, False, model_name=model_name)
    l1, sd1, p1 = _train(5, True, model_name=model_name)
    assert not p0 and len(p1) == 1 and next(iter(p1.values())).ready()
    plan = next(iter(p1.values()))
    assert plan.gen == 5                                    # one recorded step + four replays
    # the launches every step must repeat are in the recording (the weight packing was once lost to a load-order bug)
    fwd = {getattr(fn, "__name__", "") for fn, _, _ in plan.fwd.ops}
    assert {"devis_pack_weights_batch", "devis_ncx_to_nxc", "devis_conv_fwd_stats"} <= fwd, fwd
    assert l0 == l1, (l0, l1)
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


def test_fp32_replay_equals_eager():
    """The fp32 (parity) path sums its weight gradients with atomics: two EAGER runs already differ in the last bits,
    and Adam turns that noise into +-lr on parameters whose true gradient is zero (conv biases in front of BatchNorm).
    So this one compares losses and gradients of three steps on fixed parameters, to 1e-5 of each tensor's largest
    entry."""
    from devis_amd import TMSU, launch_plan

    def series(plan):
        launch_plan.ENABLED = plan
        try:
            torch.manual_seed(0)
            model = TMSU(_args(torch.float32))
            model.train()
            g = torch.Generator().manual_seed(3)
            out = []
            for _ in range(3):
                x = torch.randn((1, 4, 16, 16, 16), generator=g).cuda()
                y = torch.randint(0, 4, (1, 16, 16, 16), generator=g).cuda()
                out.append(_grads(model, x, y))
            assert bool(getattr(model.backbone, "_plans", {})) == plan
            return out
        finally:
            launch_plan.ENABLED = True

    for (l0, g0), (l1, g1) in zip(series(False), series(True)):
        assert l0 == pytest.approx(l1, rel=1e-6)
        for a, b in zip(g0, g1):
            assert (a - b).abs().max().item() <= 1e-5 * max(a.abs().max().item(), 1e-3)


def _grads(model, x, y):
    for p in model.parameters():
        p.grad = None
    _, loss = model(x, y, 25, 'train', 'BraTS')
    loss.backward()
    return loss.item(), [p.grad.clone() for p in model.parameters()]


def test_two_forwards_before_backward_and_accumulation():
    from devis_amd import TMSU
    torch.manual_seed(0)
    model = TMSU(_args(torch.bfloat16))
    model.train()
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn((1, 4, 32, 32, 32), generator=g).cuda() for _ in range(2)]
    ys = [torch.randint(0, 4, (1, 32, 32, 32), generator=g).cuda() for _ in range(2)]
    # BatchNorm running statistics move with every training forward: compare gradients only (they do not read them)
    ref = [_grads(model, x, y) for x, y in zip(xs, ys)]          # step 1 records, step 2 replays
    plan = next(iter(model.backbone._plans.values()))
    assert plan.ready()
    # two forwards, then both backwards: the second forward must not touch the first one's activations
    for p in model.parameters():
        p.grad = None
    _, la = model(xs[0], ys[0], 25, 'train', 'BraTS')
    _, lb = model(xs[1], ys[1], 25, 'train', 'BraTS')
    assert la.item() == ref[0][0] and lb.item() == ref[1][0]
    la.backward()
    ga = [p.grad.clone() for p in model.parameters()]
    for a, r in zip(ga, ref[0][1]):
        assert torch.equal(a, r)
    lb.backward()                                                # accumulates into .grad: a + b, not 2 b
    for p, a, r in zip(model.parameters(), ref[0][1], ref[1][1]):
        assert torch.equal(p.grad, a + r)
    # accumulation over two replayed steps
    for p in model.parameters():
        p.grad = None
    for x, y in zip(xs, ys):
        _, loss = model(x, y, 25, 'train', 'BraTS')
        loss.backward()
    for p, a, r in zip(model.parameters(), ref[0][1], ref[1][1]):
        assert torch.equal(p.grad, a + r)


def test_unetdrop_is_not_planned():
    from devis_amd import Unetdrop
    torch.manual_seed(0)
    m = Unetdrop(4, 16, 4).cuda().train()
    x = torch.randn((1, 4, 16, 16, 16)).cuda()
    for _ in range(2):
        m(x).sum().backward()
    assert not getattr(m, "_plans", {})


def _drop_workspaces():
    """as in a fresh process: the first recorded step has to create (and grow) every workspace"""
    from devis_amd import ops
    ops._ws_retired.extend(ops._ws.values())
    ops._ws.clear()
    ops._gate_cache.clear()


@pytest.mark.parametrize("model_name", ["U", "AU"])
def test_buffers_that_outlive_the_step_are_not_in_the_plans_pool(model_name):
    """Workspaces and constant tables created WHILE a step is recorded must not come from the plan's pool: the replay
    hands that address to a tensor of the step again, and a later user of the buffer (the loss head between forward
    and backward) overwrote an activation - NaN gradients from the first replayed step of a fresh process on."""
    from devis_amd import ops
    _drop_workspaces()
    l1, sd1, p1 = _train(4, True, model_name=model_name)
    plan = next(iter(p1.values()))
    segs = [(s["address"], s["address"] + s["total_size"]) for s in plan.pool.snapshot()]
    assert segs
    held = list(ops._ws.values()) + list(ops._ws_retired) + [t for g in ops._gate_cache.values() for t in g]
    assert ops._ws
    for t in held:
        assert not any(a <= t.data_ptr() < b for a, b in segs), "a persistent buffer lies in the launch plan's pool"
    l0, sd0, _ = _train(4, False, model_name=model_name)
    assert l0 == l1, (l0, l1)
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


def test_first_replay_of_a_fresh_process_equals_eager_at_the_bench_shape():
    """tools/lag_diff.py in a process of its own (2 x 4 x 128^3, the default weight-gradient lag): record, replay,
    compare every gradient with an eager run.  This is the case bench.py and train() start with."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from devis_amd import unet3d
    lag = "%d:%d" % unet3d._WG_LAG
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "lag_diff.py"), lag], capture_output=True, text=True,
                         timeout=110, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "tensors differing: 0" in out.stdout, out.stdout[-2000:]
