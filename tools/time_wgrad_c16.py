#!/usr/bin/env python3
"""Time the 16-channel 3x3x3 weight-gradient kernel (+ its partial reduction) at 2 x 128^3, HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
from devis_amd.unet3d import G3
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 128, 128, 128, 16), generator=g).bfloat16().cuda()
dy = torch.randn((2, 128, 128, 128, 16), generator=g).bfloat16().cuda()
sc = (torch.rand((2, 16), generator=g) + 0.5).cuda(); sh = (torch.randn((2, 16), generator=g) * 0.3).cuda()
for name, args in (("plain", (None, None)), ("prologue", (sc, sh))):
    fn = lambda: ops.conv_wgrad(x, None, dy, G3, (16, 16, 3, 3, 3), *args)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    dw, db = fn()
    print("%s: %.1f us per call (kernel + reduction), |dw| %.6e |db| %.6e" % (name, e0.elapsed_time(e1) / 20 * 1e3, dw.double().norm().item(), db.double().norm().item()))
