"""world_size-2 gloo tests (CPU) of the data-parallel pieces: gradient averaging and patch sharding."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from devis_amd.parallel import GradientAllReducer, broadcast_parameters
    torch.manual_seed(100 + rank)                       # different replicas on purpose
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    broadcast_parameters(net)                            # -> rank 0's weights everywhere
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    for p in net.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    net[1].bias.grad = None                               # a parameter without gradient must still line up
    GradientAllReducer(net.parameters(), bucket_bytes=64).all_reduce()
    g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    # overlapped path: a backbone-style flat gradient buffer reduced bucket by bucket as backward "produces" it
    from devis_amd.parallel import FlatGradSync
    sync = FlatGradSync(world)
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    view = flat[3:7]                                      # a parameter's .grad is a view of the flat buffer
    sync.begin(flat, [(0, 3), (3, 7), (7, 10)])
    for i in range(3):
        sync.bucket_ready(i)
    assert sync.finish() and not sync.finish()
    assert torch.allclose(flat, torch.arange(10, dtype=torch.float32) * 1.5)
    assert torch.allclose(view, torch.arange(3, 7, dtype=torch.float32) * 1.5)
    q.put((rank, w0, g))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_and_broadcast_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w_a, g_a), (_, w_b, g_b) = res
    assert torch.equal(w_a, w_b)                          # broadcast made the replicas identical
    assert torch.equal(g_a, g_b)
    n_bias = 3
    assert torch.allclose(g_a[:-n_bias], torch.full_like(g_a[:-n_bias], 1.5))   # mean of 1 and 2
    assert torch.allclose(g_a[-n_bias:], torch.zeros(n_bias))                   # missing grads count as zero


def test_patch_sharding_covers_all_patches():
    from devis_amd.parallel import shard_patches
    for world in (1, 2, 3, 4, 8):
        got = sorted(i for r in range(world) for i in shard_patches(8, r, world))
        assert got == list(range(8))


def _fake_patch_fn(patch):
    """Stands in for backbone + fused head on CPU: per-voxel maps that encode position and content."""
    s = patch.sum(1)                                              # (1,128,128,128)
    return dict(pred=(s.abs() * 3).to(torch.uint8) % 4, u=(s * 0.5).unsqueeze(1), umask=(s.abs() * 100).to(torch.int16),
                evidence=patch * 2 + 1, mne=None)


def _infer_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from devis_amd.predict import sliding_window_infer
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 2, 240, 240, 160), generator=g)
    out = sliding_window_infer(x, None, rank=rank, world=world, patch_fn=_fake_patch_fn)
    single = sliding_window_infer(x, None, patch_fn=_fake_patch_fn)
    ok = sorted(out) == sorted(single) and all(
        out[k].dtype == single[k].dtype and torch.equal(out[k], single[k]) for k in single)
    q.put((rank, bool(ok), sorted(out)))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sliding_window_gloo(world):
    """Patches sharded over the ranks + ONE all_gather of the packed sub-regions == the single-process stitch ==
    tailor_and_concat.  world 3: the ranks own 3 / 3 / 2 patches, so the gathered buffers are padded."""
    from devis_amd.predict import sliding_window_infer, tailor_and_concat
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 2, 240, 240, 160), generator=g)
    single = sliding_window_infer(x, None, patch_fn=_fake_patch_fn)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_infer_worker, args=(r, world, port + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, ok, keys in res:
        assert ok and keys == sorted(single)
    # and the evidence map is exactly what the reference's stitch gives for the same per-patch function
    ref = tailor_and_concat(torch.cat([x, x], 1), lambda t: t * 2 + 1)[:, :2]
    assert torch.equal(single["evidence"], ref)


def test_flat_grad_sync_sends_the_first_two_slices_together():
    """devis_amd/parallel.py: three marks in backward, two collectives (DEVIS_DP_BUCKETS, default 2)"""
    import torch
    from devis_amd import parallel
    s = parallel.FlatGradSync(2)
    flat = torch.zeros(12)
    s.begin(flat, [(0, 4), (4, 10), (10, 12)])
    assert s.bounds == ([(0, 0), (0, 10), (10, 12)] if parallel._MERGE_FIRST_TWO else [(0, 4), (4, 10), (10, 12)])
    s.bucket_ready(0) if parallel._MERGE_FIRST_TWO else None     # an empty slice launches nothing (no process group here)
    assert s.works == []


def test_nccl_options_ask_for_a_high_priority_stream(monkeypatch):
    """devis_amd/parallel.py:nccl_options - what bench.py passes to init_process_group"""
    from devis_amd import parallel
    monkeypatch.delenv("DEVIS_NCCL_HIGH_PRIORITY", raising=False)
    assert parallel.nccl_options().is_high_priority_stream is True
    monkeypatch.setenv("DEVIS_NCCL_HIGH_PRIORITY", "0")
    assert parallel.nccl_options() is None
