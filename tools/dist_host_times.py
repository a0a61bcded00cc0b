#!/usr/bin/env python3
"""Host time of each part of a data-parallel training step (one-rank RCCL group): where does the host wait?
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29540 python tools/dist_host_times.py"""
import faulthandler, os, sys, time, torch
faulthandler.dump_traceback_later(40, exit=True)      # a hang names its line and ends
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import TMSU
from devis_amd.optim import FusedAdam
from devis_amd.parallel import GradientAllReducer, broadcast_parameters
from devis_amd.train import adjust_learning_rate, freeze_gc
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(0)
import bench
args = bench.brats_args(torch.bfloat16)
model = TMSU(args).train()
opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)
red = GradientAllReducer(model.parameters(), module=model, force=True)
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 4, 128, 128, 128), generator=g).to(dev)
y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).to(dev)
freeze_gc()
names = ("forward", "backward", "all_reduce", "opt.step")
acc = [[] for _ in names]
for it in range(40):
    t = [time.perf_counter()]
    _, loss = model(x, y, 25, 'train', 'BraTS'); t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True); loss.backward(); t.append(time.perf_counter())
    red.all_reduce(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if os.environ.get("SYNC_EVERY") and it % int(os.environ["SYNC_EVERY"]) == 0:
        torch.cuda.synchronize()
    if it >= 10:
        for k in range(4): acc[k].append((t[k + 1] - t[k]) * 1e3)
torch.cuda.synchronize()
for n, a in zip(names, acc):
    a.sort()
    print("%-10s host ms: median %.3f  p90 %.3f  max %.3f" % (n, a[len(a) // 2], a[int(len(a) * 0.9)], a[-1]))
dist.destroy_process_group()
