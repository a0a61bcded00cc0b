"""BASELINE configs[3] (data-parallel BraTS training) on ONE GPU box.

  * one-rank RCCL group: the bucketed, overlapped all-reduce path (parallel.FlatGradSync over `nccl`, launched from
    inside the real backbone's backward at its three bucket boundaries) must leave gradients bit-identical to the
    step without a process group;
  * world_size 2: two processes share cuda:0 (gloo carries the CUDA buffers - RCCL refuses two ranks on one device)
    and run the real train step on their own shards; the reduced gradients must be the same bits on both ranks and
    equal the DDP semantics of SURVEY.md 8(e) computed by the CPU oracle: every rank normalises (BatchNorm) and
    takes the soft dice over ITS OWN samples, gradients are averaged.

The 2/4/8-GPU RCCL runs themselves are the driver's (one process per GPU, `bench.py --gpus N`)."""
import os
import socket
import subprocess
import sys

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
    from devis_amd.parallel import GradientAllReducer
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
                            device_id=torch.device("cuda", torch.cuda.current_device()))
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
def test_two_ranks_on_one_gpu_match_ddp_semantics_of_the_oracle(tmp_path, dtype):
    from oracle import devis_oracle as O
    from devis_amd import TMSU
    size, world = 32, 2
    port = _free_port()
    # loopback only: gloo otherwise resolves the host name, which a GPU box need not be able to do
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), GLOO_SOCKET_IFNAME="lo")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), str(world), str(port),
                               str(tmp_path), dtype, str(size)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=150)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            tails = [q.communicate()[0].decode(errors="replace")[-1500:] for q in procs]
            pytest.fail("data-parallel workers timed out:\n" + "\n---\n".join(tails))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    # replicas: same start (rank 0's broadcast weights), same reduced gradients, same weights after the step
    assert torch.equal(res[0]["w0"], res[1]["w0"])
    for n in res[0]["grads"]:
        assert torch.equal(res[0]["grads"][n], res[1]["grads"][n]), n
    assert torch.equal(res[0]["w1"], res[1]["w1"]) and not torch.equal(res[0]["w1"], res[0]["w0"])
    # oracle: rank 0's initial replica (seed 100) on each shard separately, gradients averaged
    torch.manual_seed(100)
    m = TMSU(brats_args(compute_dtype=torch.float32))
    sd0 = {k: v.detach().cpu().clone() for k, v in m.backbone.state_dict().items()}
    names = [k for k, _ in m.backbone.named_parameters()]
    assert torch.equal(torch.cat([sd0[k].reshape(-1) for k in names]), res[0]["w0"])
    avg, losses = {k: 0.0 for k in names}, []
    for r in range(world):
        g = torch.Generator().manual_seed(1234 + r)
        x = torch.randn((2, 4, size, size, size), generator=g)
        y = torch.randint(0, 4, (2, size, size, size), generator=g)
        sd = {k: v.clone() for k, v in sd0.items()}
        for k in names:
            sd[k].requires_grad_(True)
        _, loss = O.tmsu_forward(lambda t: O.unet3d_forward(sd, t, True), x, y, 25, 'train', 'BraTS', 4, 50, 100)
        loss.backward()
        losses.append(float(loss.detach()))
        for k in names:
            avg[k] = avg[k] + sd[k].grad.double() / world
    fp32 = dtype == "fp32"
    for r in range(world):
        check(abs(res[r]["loss"] - losses[r]), 1e-5 * abs(losses[r]) if fp32 else 1e-3, "rank %d loss %s" % (r, dtype))
    norms = np.array([res[0]["grads"][k].double().norm().item() for k in sorted(names)])
    ref = np.array([avg[k].norm().item() for k in sorted(names)])
    gtol, gatol = (GRAD_L2_TOL, 1e-6) if fp32 else (0.08, 1e-3)
    check(np.max((np.abs(norms - ref) - gatol * ref.max()) / (ref + 1e-30)), gtol, "averaged grad norms %s" % dtype)
    if fp32:
        worst = max(l2_rel_err(res[0]["grads"][k], avg[k].float()) for k in names if avg[k].norm() > 1e-4 * ref.max())
        check(worst, 4 * GRAD_L2_TOL, "averaged grads, worst per-tensor L2")
    # the semantics are PER-RANK statistics: the gradient of the concatenated batch (BatchNorm and dice over all four
    # samples) is a different vector, and ours must be the averaged one
    xs, ys = [], []
    for r in range(world):
        g = torch.Generator().manual_seed(1234 + r)
        xs.append(torch.randn((2, 4, size, size, size), generator=g))
        ys.append(torch.randint(0, 4, (2, size, size, size), generator=g))
    sd = {k: v.clone() for k, v in sd0.items()}
    for k in names:
        sd[k].requires_grad_(True)
    _, loss = O.tmsu_forward(lambda t: O.unet3d_forward(sd, t, True), torch.cat(xs), torch.cat(ys), 25, 'train', 'BraTS',
                             4, 50, 100)
    loss.backward()
    k = "DeBlock2.conv1.weight"
    ours, per_rank, joint = res[0]["grads"][k].double(), avg[k], sd[k].grad.double()
    if fp32:
        assert (per_rank - joint).norm() > 3 * (ours - per_rank).norm(), \
            ((per_rank - joint).norm().item(), (ours - per_rank).norm().item())
