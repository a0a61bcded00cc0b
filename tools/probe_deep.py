#!/usr/bin/env python3
"""Time the deep-layer 3x3x3 convs (C=32@64^3, 64@32^3, 128@16^3; B=2): fwd, dgrad, wgrad."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
from devis_amd.unet3d import G3

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for C, S in ((32, 64), (64, 32), (128, 16)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, S, S, S, C), generator=g).bfloat16().cuda()
    dy = torch.randn((2, S, S, S, C), generator=g).bfloat16().cuda()
    w = (torch.randn((C, C, 3, 3, 3), generator=g) * 0.05).cuda()
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    sc = torch.ones((2, C), device="cuda"); sh = torch.zeros((2, C), device="cuda"); b = torch.zeros(C, device="cuda")
    fl = 2.0 * 27 * C * C * 2 * S ** 3
    tf = timeit(lambda: ops.conv_fwd(x, None, wp, b, G3, sc, sh))
    td = timeit(lambda: ops.conv_dgrad(dy, wp, G3, x.shape))
    tw = timeit(lambda: ops.conv_wgrad(x, None, dy, G3, w.shape, sc, sh))
    print("C=%3d S=%3d  fwd %6.1f us (%5.0f TF)  dgrad %6.1f us (%5.0f TF)  wgrad %6.1f us (%5.0f TF)"
          % (C, S, tf, fl / tf / 1e6, td, fl / td / 1e6, tw, fl / tw / 1e6))
