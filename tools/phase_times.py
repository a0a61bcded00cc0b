#!/usr/bin/env python3
"""GPU time of the phases of the bench step (HIP events recorded on the main stream, replayed with the launch plan):
forward, loss, backbone backward up to the join with the side stream, the wait at the join, optimizer.
With RANK / WORLD_SIZE / MASTER_* in the environment: the data-parallel step (one-rank RCCL group, bucketed all-reduce
from the side stream), with the phase "bwd->reduced" = GradientAllReducer.all_reduce().  (r02: this mode never came back
from its first synchronise; tools/dp_phase_probe.py is the non-blocking form of the same question.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from devis_amd import TMSU, launch_plan, unet3d  # noqa: E402
from devis_amd.optim import FusedAdam  # noqa: E402
from devis_amd.train import adjust_learning_rate, freeze_gc  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
DIST = "RANK" in os.environ
if DIST:
    import faulthandler
    import torch.distributed as dist
    from devis_amd.parallel import GradientAllReducer, nccl_options
    dist.init_process_group("nccl", device_id=dev, pg_options=nccl_options())
    faulthandler.dump_traceback_later(60, exit=True)
E = {k: torch.cuda.Event(enable_timing=True) for k in ("start", "fwd", "loss", "bwd0", "prejoin", "postjoin", "bwd", "reduced", "adam")}
main = torch.cuda.current_stream()
orig_end, orig_begin = unet3d._Run._end_backward, unet3d._Run._begin_backward


def begin(self, device):
    orig_begin(self, device)
    launch_plan.call(E["bwd0"].record, main)


def end(self):
    launch_plan.call(E["prejoin"].record, main)
    orig_end(self)
    launch_plan.call(E["postjoin"].record, main)


unet3d._Run._begin_backward, unet3d._Run._end_backward = begin, end
args = bench.brats_args(torch.bfloat16)
torch.manual_seed(0)
model = TMSU(args)
model.train()
opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)
reducer = GradientAllReducer(model.parameters(), module=model, force=True) if DIST else None
g = torch.Generator().manual_seed(1234)
x = torch.randn((2, 4, 128, 128, 128), generator=g).to(dev)
y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).to(dev)
bb = model.backbone
orig_fcl = bb.forward_channels_last


def fcl(xx):
    out = orig_fcl(xx)
    E["fwd"].record(main)
    return out


bb.forward_channels_last = fcl
freeze_gc()
acc = {}
N = 60
for i in range(N + 10):
    E["start"].record(main)
    adjust_learning_rate(opt, 25, args.end_epochs, args.lr)
    _, loss = model(x, y, 25, 'train', 'BraTS')
    E["loss"].record(main)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    E["bwd"].record(main)
    if reducer is not None:
        reducer.all_reduce()
    E["reduced"].record(main)
    opt.step()
    E["adam"].record(main)
    if i >= 10 and i % 4 == 0:          # a sync every fourth step only: the host stays ahead in between
        torch.cuda.synchronize()
        order = ["start", "fwd", "loss", "bwd0", "prejoin", "postjoin", "bwd", "reduced", "adam"]
        for a, b in zip(order, order[1:]):
            acc.setdefault(a + "->" + b, []).append(E[a].elapsed_time(E[b]))
for k, v in acc.items():
    v.sort()
    print("%-20s median %8.1f us   (min %.1f max %.1f)" % (k, v[len(v) // 2] * 1e3, v[0] * 1e3, v[-1] * 1e3))
print("total median %.1f us" % (sum(sorted(v)[len(v) // 2] for v in acc.values()) * 1e3))
if DIST:
    faulthandler.cancel_dump_traceback_later()
    dist.destroy_process_group()
