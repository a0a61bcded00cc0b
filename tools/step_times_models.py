#!/usr/bin/env python3
"""Training step time of the three 3-D backbones TMSU exposes (BraTS shape, B=2; argv[1] = bf16 (default) | fp32)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU
from devis_amd.optim import FusedAdam
from devis_amd.train import adjust_learning_rate, freeze_gc
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 4, 128, 128, 128), generator=g).cuda(); y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).cuda()
DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for name in ("U", "AU", "V"):
    args = bench.brats_args(torch.bfloat16 if DT == "bf16" else torch.float32); args.model_name = name
    torch.manual_seed(0)
    m = TMSU(args).train()
    opt = FusedAdam(m.parameters(), lr=args.lr, weight_decay=1e-5, module=m)
    def step():
        adjust_learning_rate(opt, 25, 99, args.lr)
        _, loss = m(x, y, 25, "train", "BraTS"); opt.zero_grad(); loss.backward(); opt.step(); return loss
    freeze_gc()
    for _ in range(4): l = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 15
    print("%-2s BraTS B=2 %s: %.2f ms/step, %.0f M voxels/s, loss %.6f" % (name, DT, dt * 1e3, 2 * 128**3 / dt / 1e6, l.item()))
    del m, opt
