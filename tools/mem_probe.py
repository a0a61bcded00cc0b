#!/usr/bin/env python3
"""Device memory of the bench step after four steps: reserved / allocated / peak, with the launch plan
(DEVIS_LAUNCH_PLAN=1, default) or without (=0)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from devis_amd import TMSU
from devis_amd.optim import FusedAdam
args = bench.brats_args(torch.bfloat16)
torch.manual_seed(0)
m = TMSU(args).train()
opt = FusedAdam(m.parameters(), lr=0.002, weight_decay=1e-5, module=m)
x = torch.randn((2, 4, 128, 128, 128)).cuda(); y = torch.randint(0, 4, (2, 128, 128, 128)).cuda()
for i in range(4):
    _, loss = m(x, y, 25, 'train', 'BraTS'); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
torch.cuda.synchronize()
print("plan=%s reserved %.2f GB allocated %.2f GB peak allocated %.2f GB" % (os.environ.get("DEVIS_LAUNCH_PLAN", "1"), torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30))
