"""BASELINE configs[3] (data-parallel BraTS training) on ONE GPU box.

  * one-rank RCCL group: the bucketed, overlapped all-reduce path (parallel.FlatGradSync over `nccl`, launched from
    inside the real backbone's backward at its three bucket boundaries) must leave gradients bit-identical to the
    step without a process group;
  * world_size 2 emulated: two seeded shards through the same replica, flat gradient buffers averaged, equal the
    DDP semantics of SURVEY.md 8(e) computed by the CPU oracle: every rank normalises (BatchNorm) and takes the soft
    dice over ITS OWN samples, gradients are averaged.

The 2/4/8-GPU RCCL runs themselves are the driver's (one process per GPU, `bench.py --gpus N`)."""
import os
import socket

import numpy as np
import pytest
import torch

from tests.util import GRAD_L2_TOL, brats_args, check, l2_rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_one_rank_rccl_bucketed_allreduce_is_bit_identical():
    import torch.distributed as dist
    from devis_amd import TMSU
    from devis_amd.parallel import GradientAllReducer, nccl_options
    torch.manual_seed(0)
    m = TMSU(brats_args(compute_dtype=torch.bfloat16)).train()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((2, 4, 64, 64, 64), generator=g).cuda()
    y = torch.randint(0, 4, (2, 64, 64, 64), generator=g).cuda()

    def step(reducer):
        _, loss = m(x, y, 25, 'train', 'BraTS')
        for p in m.parameters():
            p.grad = None
        loss.backward()
        if reducer is not None:
            reducer.all_reduce()
        torch.cuda.synchronize()
        return m.backbone.last_flat_grad.clone(), float(loss.item())

    ref, loss_ref = step(None)
    assert dist.is_available() and not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()),
                            pg_options=nccl_options())       # as bench.py / INTEGRATION.md set the group up
    try:
        reducer = GradientAllReducer(m.parameters(), module=m, force=True)
        assert reducer.active and reducer.flat_sync is not None and reducer.flat_sync._avg
        calls = []
        ready = reducer.flat_sync.bucket_ready
        reducer.flat_sync.bucket_ready = lambda i: (calls.append(i), ready(i))[1]
        for _ in range(2):
            got, loss = step(reducer)
            assert loss == loss_ref
            assert torch.equal(got, ref)
        assert calls == [0, 1, 2, 0, 1, 2]                # three buckets per backward, decoder first
        # a stale .grad would be summed with an un-reduced copy: the backward must refuse it
        _, loss = m(x, y, 25, 'train', 'BraTS')
        with pytest.raises(RuntimeError, match="zero_grad"):
            loss.backward()
    finally:
        m.backbone._grad_sync = None
        dist.destroy_process_group()
    got, _ = step(None)                                    # and the plain path is intact afterwards
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_two_rank_semantics_emulated_on_one_gpu_match_the_oracle(dtype):
    """W = 2 emulated on one GPU: the two ranks' shards (seeds 1234 + r) go through the same replica
    one after the other, each backward leaves its gradients in the backbone's flat buffer, the two flat buffers are
    averaged - what RCCL's AVG all-reduce of FlatGradSync leaves on every rank.  Oracle: the CPU restatement run on
    each shard separately (BatchNorm statistics and the soft dice over that rank's own samples), gradients averaged.
    (Two real processes cannot share this box's one GPU under RCCL; the two-process reducer itself runs under gloo
    in tests/test_cpu_distributed.py.)"""
    from oracle import devis_oracle as O
    from devis_amd import TMSU
    size, world = 32, 2
    fp32 = dtype == "fp32"
    torch.manual_seed(100)
    m = TMSU(brats_args(compute_dtype=torch.float32 if fp32 else torch.bfloat16)).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.backbone.state_dict().items()}
    names = [k for k, _ in m.backbone.named_parameters()]
    shards = []
    for r in range(world):
        g = torch.Generator().manual_seed(1234 + r)
        shards.append((torch.randn((2, 4, size, size, size), generator=g), torch.randint(0, 4, (2, size, size, size), generator=g)))
    flats, ours_loss = [], []
    for x, y in shards:
        _, loss = m(x.cuda(), y.cuda(), 25, 'train', 'BraTS')
        for p in m.parameters():
            p.grad = None
        loss.backward()
        torch.cuda.synchronize()
        flats.append(m.backbone.last_flat_grad.clone())
        ours_loss.append(float(loss.item()))
    flat = (flats[0] + flats[1]) / world
    layout = m.backbone._grad_layout
    ours = {k: flat[layout[k][0]:layout[k][0] + layout[k][1]].view(layout[k][2]).cpu() for k in names}
    avg, losses = {k: 0.0 for k in names}, []
    for x, y in shards:
        sd = {k: v.clone() for k, v in sd0.items()}
        for k in names:
            sd[k].requires_grad_(True)
        _, loss = O.tmsu_forward(lambda t: O.unet3d_forward(sd, t, True), x, y, 25, 'train', 'BraTS', 4, 50, 100)
        loss.backward()
        losses.append(float(loss.detach()))
        for k in names:
            avg[k] = avg[k] + sd[k].grad.double() / world
    for r in range(world):
        check(abs(ours_loss[r] - losses[r]), 1e-5 * abs(losses[r]) if fp32 else 1e-3, "rank %d loss %s" % (r, dtype))
    norms = np.array([ours[k].double().norm().item() for k in sorted(names)])
    ref = np.array([avg[k].norm().item() for k in sorted(names)])
    gtol, gatol = (GRAD_L2_TOL, 1e-6) if fp32 else (0.08, 1e-3)
    check(np.max((np.abs(norms - ref) - gatol * ref.max()) / (ref + 1e-30)), gtol, "averaged grad norms %s" % dtype)
    if not fp32:
        return
    worst = max(l2_rel_err(ours[k], avg[k].float()) for k in names if avg[k].norm() > 1e-4 * ref.max())
    check(worst, 4 * GRAD_L2_TOL, "averaged grads, worst per-tensor L2")
    # the semantics are PER-RANK statistics: the gradient of the concatenated batch (BatchNorm and dice over all four
    # samples) is a different vector, and ours must be the averaged one
    sd = {k: v.clone() for k, v in sd0.items()}
    for k in names:
        sd[k].requires_grad_(True)
    _, loss = O.tmsu_forward(lambda t: O.unet3d_forward(sd, t, True), torch.cat([s[0] for s in shards]),
                             torch.cat([s[1] for s in shards]), 25, 'train', 'BraTS', 4, 50, 100)
    loss.backward()
    k = "DeBlock2.conv1.weight"
    per_rank, joint = avg[k], sd[k].grad.double()
    assert (per_rank - joint).norm() > 3 * (ours[k].double() - per_rank).norm(), \
        ((per_rank - joint).norm().item(), (ours[k].double() - per_rank).norm().item())
