#!/usr/bin/env python3
"""Time the halo-free weight gradients of the 3-D U-Net step (1x1 convs, ConvT k2 s2) at their BraTS shapes, HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
g = torch.Generator().manual_seed(1)
G1 = ops.ConvGeom((1, 1, 1), (1, 1, 1), (0, 0, 0))
rnd = lambda *s: torch.randn(s, generator=g).bfloat16().cuda()
cases = []
for name, sp, ci, co in (("1x1 16->16 @128 (dW3a)", 128, 16, 16), ("1x1 16->4 @128 (endconv)", 128, 16, 4), ("1x1 32->16 @64", 64, 32, 16),
                         ("1x1 32->32 @64", 64, 32, 32), ("1x1 64->32 @32", 32, 64, 32), ("1x1 128->64 @16", 16, 128, 64)):
    x, dy = rnd(2, sp, sp, sp, ci), rnd(2, sp, sp, sp, co)
    cases.append((name, lambda x=x, dy=dy, ci=ci, co=co: ops.conv_wgrad(x, None, dy, G1, (co, ci, 1, 1, 1))))
for name, sp, c in (("convT 16 @64->128 (G)", 64, 16), ("convT 32 @32->64", 32, 32), ("convT 64 @16->32", 16, 64)):
    x, dy = rnd(2, sp, sp, sp, c), rnd(2, 2 * sp, 2 * sp, 2 * sp, c)
    cases.append((name, lambda x=x, dy=dy, c=c: ops.convT_wgrad(x, dy, (2, 2, 2), (c, c, 2, 2, 2))))
for name, fn in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    dw, db = fn()
    print("%-28s %6.1f us  |dw| %.6e" % (name, e0.elapsed_time(e1) / 20 * 1e3, dw.double().norm().item()))
