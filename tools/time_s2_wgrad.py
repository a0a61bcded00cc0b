#!/usr/bin/env python3
"""Time the stride-2 3x3x3 weight gradient 16 -> 32 (EnDown1) at 2 x 128^3 (kernel + partial reduction), HIP events.
(A/B against the tile kernel: build the other variant with tools/build_alt.sh and DEVIS_HIP_LIB.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
g = torch.Generator().manual_seed(1)
geom = ops.ConvGeom((3, 3, 3), (2, 2, 2), (1, 1, 1))
for ci, co, sp in ((16, 32, 128), (32, 64, 64), (64, 128, 32)):
    x = torch.randn((2, sp, sp, sp, ci), generator=g).bfloat16().cuda()
    dy = torch.randn((2, sp // 2, sp // 2, sp // 2, co), generator=g).bfloat16().cuda()
    fn = lambda: ops.conv_wgrad(x, None, dy, geom, (co, ci, 3, 3, 3), None, None)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    dw, db = fn()
    print("%d->%d @%d: %.1f us per call (kernel + reduction), |dw| %.6e |db| %.6e" % (ci, co, sp, e0.elapsed_time(e1) / 20 * 1e3, dw.double().norm().item(), db.double().norm().item()))
