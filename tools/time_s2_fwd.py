#!/usr/bin/env python3
"""Time the three stride-2 (EnDown) forward convs of the BraTS step with epilogue statistics (B=2, bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
geom = ops.ConvGeom((3, 3, 3), (2, 2, 2), (1, 1, 1))
g = torch.Generator().manual_seed(0)
for ci, co, sz in ((16, 32, 128), (32, 64, 64), (64, 128, 32)):
    w = (torch.randn((co, ci, 3, 3, 3), generator=g) * 0.05).cuda()
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    bias = torch.randn((co,), generator=g).cuda()
    x = torch.randn((2, sz, sz, sz, ci), generator=g).bfloat16().cuda()
    f = lambda: ops.conv_fwd(x, None, wp, bias, geom, want_stats=True)
    y = f()[0]; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("s2 fwd %3d -> %3d @ %d^3: %.1f us  checksum %.6f" % (ci, co, sz, e0.elapsed_time(e1) / 20 * 1e3, y.float().abs().mean().item()))
