#!/usr/bin/env python3
"""Per-step wall time (synchronised) of the training step: exposes periodic stalls."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU
from devis_amd.train import adjust_learning_rate
args = bench.brats_args(torch.bfloat16)
torch.manual_seed(0)
m = TMSU(args); m.train()
from devis_amd.optim import FusedAdam
opt = FusedAdam(m.parameters(), lr=args.lr, weight_decay=1e-5, module=m)
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 4, 128, 128, 128), generator=g).cuda(); y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).cuda()
ts = []
import gc
for i in range(40):
    if i == 5 and os.environ.get('GCFREEZE'):
        gc.collect(); gc.freeze()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    adjust_learning_rate(opt, 25, 99, args.lr)
    _, loss = m(x, y, 25, 'train', 'BraTS'); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % t for t in ts))
print("reserved MB", torch.cuda.memory_reserved() >> 20, "allocated MB", torch.cuda.memory_allocated() >> 20,
      "num_alloc_retries", torch.cuda.memory_stats().get("num_alloc_retries"), "segments", torch.cuda.memory_stats().get("segment.all.allocated"))
