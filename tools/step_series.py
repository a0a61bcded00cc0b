#!/usr/bin/env python3
"""Per-step GPU times of the bench step inside ONE process (HIP events between steps): is the step time of a process
locked to one value, or does it move?  usage: step_series.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from devis_amd import TMSU  # noqa: E402
from devis_amd.optim import FusedAdam  # noqa: E402
from devis_amd.train import adjust_learning_rate, freeze_gc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
# SCHED="lag=1:2,tail=2,batch=2": the two-stream schedule constants of devis_amd/unet3d.py for this run (A/B sweeps)
for kv in filter(None, os.environ.get("SCHED", "").split(",")):
    from devis_amd import unet3d
    k, v = kv.split("=")
    if k == "lag":
        unet3d._WG_LAG = tuple(int(t) for t in v.split(":"))
    elif k == "tail":
        unet3d._WG_TAIL_MAIN = int(v)
    elif k == "batch":
        unet3d._WG_BATCH = int(v)
    else:
        sys.exit("SCHED: unknown key " + k)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = bench.brats_args(torch.bfloat16)
torch.manual_seed(0)
model = TMSU(args)
model.train()
opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)
g = torch.Generator().manual_seed(1234)
x = torch.randn((2, 4, 128, 128, 128), generator=g).to(dev)
y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).to(dev)


def step():
    adjust_learning_rate(opt, 25, args.end_epochs, args.lr)
    _, loss = model(x, y, 25, 'train', 'BraTS')
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


if os.environ.get("MAIN_HIGH_PRIORITY") == "1":          # experiment: the whole step on a high-priority stream
    torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
freeze_gc()
for _ in range(10):
    step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
import time
ev[0].record()
th = [time.perf_counter()]
for i in range(steps):
    step()
    ev[i + 1].record()
    th.append(time.perf_counter())
t_sync0 = time.perf_counter()
torch.cuda.synchronize()
t_sync1 = time.perf_counter()
hs = sorted(th[i + 1] - th[i] for i in range(steps))
print("host enqueue per step: min %.3f median %.3f max %.3f ms; final sync waited %.1f ms" %
      (hs[0] * 1e3, hs[steps // 2] * 1e3, hs[-1] * 1e3, (t_sync1 - t_sync0) * 1e3))
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
s = sorted(t)
print("steps %d  mean %.3f  min %.3f  p10 %.3f  median %.3f  p90 %.3f  max %.3f ms" %
      (steps, sum(t) / steps, s[0], s[steps // 10], s[steps // 2], s[steps * 9 // 10], s[-1]))
print(" ".join("%.2f" % v for v in t[:40]))
