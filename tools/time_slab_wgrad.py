#!/usr/bin/env python3
"""Time the 3x3x3 weight gradients of the 32 / 64 / 128-channel layers (kernel + partial reduction) at the BraTS step's
shapes, with and without the norm prologue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
from devis_amd.unet3d import G3
g = torch.Generator().manual_seed(1)
for c, sp in ((32, 64), (64, 32), (128, 16)):
    x = torch.randn((2, sp, sp, sp, c), generator=g).bfloat16().cuda()
    dy = torch.randn((2, sp, sp, sp, c), generator=g).bfloat16().cuda()
    sc = (torch.rand((2, c), generator=g) + 0.5).cuda(); sh = (torch.randn((2, c), generator=g) * 0.3).cuda()
    for name, args in (("plain", (None, None)), ("prologue", (sc, sh))):
        fn = lambda: ops.conv_wgrad(x, None, dy, G3, (c, c, 3, 3, 3), *args)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        dw, db = fn()
        print("%3d ch @%d^3 %-8s %6.1f us  |dw| %.6e" % (c, sp, name, e0.elapsed_time(e1) / 20 * 1e3, dw.double().norm().item()))
