#!/usr/bin/env python3
"""Time the 64- and 128-channel 3x3x3 layers of the BraTS step (B=2, bf16): forward with prologue + statistics, dgrad
with statistics - as the network calls them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
geom = ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
g = torch.Generator().manual_seed(0)
def t(f):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for c, sz in ((64, 32), (128, 16)):
    w = (torch.randn((c, c, 3, 3, 3), generator=g) * 0.05).cuda()
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    wd = torch.empty((wp.shape[2], wp.shape[1], wp.shape[0]), dtype=wp.dtype, device=wp.device)
    x = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    dy = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    b = torch.zeros(c).cuda(); sc = torch.ones((2, c)).cuda(); sh = torch.zeros((2, c)).cuda()
    fw = t(lambda: ops.conv_fwd(x, None, wp, b, geom, sc, sh, want_stats=True))
    dg = t(lambda: ops.conv_dgrad(dy, wp, geom, x.shape, norm=(x, sc, sh)))
    print("%3d ch @ %d^3: fwd+pro+stats %.1f us, dgrad+stats (incl. repack) %.1f us" % (c, sz, fw, dg))
