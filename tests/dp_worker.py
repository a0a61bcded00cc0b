"""One rank of the world_size-2 data-parallel train step that tests/test_gpu_dp.py runs on ONE GPU (both ranks share
cuda:0; gloo carries the CUDA tensors, because RCCL refuses two ranks on one device).  Everything else is the real
path: TMSU on the HIP kernels, gradients written into the backbone's flat buffer, FlatGradSync reducing the three
buckets as backward finishes them, FusedAdam reading the reduced buffer.

    python tests/dp_worker.py RANK WORLD PORT OUT_DIR DTYPE SIZE
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_dir, dtype, size = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4],
                                               sys.argv[5], int(sys.argv[6]))
    import torch
    import torch.distributed as dist
    from tests.util import brats_args
    from devis_amd import TMSU
    from devis_amd.optim import FusedAdam
    from devis_amd.parallel import GradientAllReducer, broadcast_parameters
    def say(msg):
        print("[dp_worker %d] %s" % (rank, msg), flush=True)

    torch.cuda.set_device(0)
    say("init gloo")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    say("group up")
    torch.manual_seed(100 + rank)                         # different initial replicas on purpose ...
    m = TMSU(brats_args(compute_dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)).train()
    broadcast_parameters(m)                               # ... made identical to rank 0's here
    say("parameters broadcast")
    w0 = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    opt = FusedAdam(m.parameters(), lr=0.002, weight_decay=1e-5, module=m)
    reducer = GradientAllReducer(m.parameters(), module=m)
    assert reducer.flat_sync is not None and m.backbone._grad_sync is reducer.flat_sync
    g = torch.Generator().manual_seed(1234 + rank)        # rank r draws its own shard
    x = torch.randn((2, 4, size, size, size), generator=g).cuda()
    y = torch.randint(0, 4, (2, size, size, size), generator=g).cuda()
    _, loss = m(x, y, 25, 'train', 'BraTS')
    opt.zero_grad()
    say("forward done")
    loss.backward()
    reducer.all_reduce()
    say("backward + all-reduce done")
    bb = m.backbone
    flat = bb.last_flat_grad
    lo = flat.data_ptr()
    for n, p in bb.named_parameters():                    # .grad still IS the reduced flat buffer
        assert p.grad.data_ptr() == lo + 4 * bb._grad_layout[n][0], n
    grads = {n: p.grad.detach().cpu().clone() for n, p in bb.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    w1 = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    torch.save(dict(loss=float(loss.item()), grads=grads, w0=w0, w1=w1), os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
