#!/usr/bin/env python3
"""Time the 16-channel 3x3x3 forward / dgrad kernels at 2 x 128^3 (the variants the train step launches), HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
from devis_amd.unet3d import G3
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 128, 128, 128, 16), generator=g).bfloat16().cuda()
r = torch.randn((2, 128, 128, 128, 16), generator=g).bfloat16().cuda()
w = (torch.randn((16, 16, 3, 3, 3), generator=g) * 0.05).cuda()
b = torch.zeros(16).cuda()
wp = ops.pack_conv_weight(w, torch.bfloat16)
sc = (torch.rand((2, 16), generator=g) + 0.5).cuda(); sh = (torch.randn((2, 16), generator=g) * 0.3).cuda()
cases = (("fwd plain+stats", lambda: ops.conv_fwd(x, None, wp, b, G3, want_stats=True)),
         ("fwd prologue+stats", lambda: ops.conv_fwd(x, None, wp, b, G3, sc, sh, want_stats=True)),
         ("fwd prologue+residual", lambda: ops.conv_fwd(x, None, wp, b, G3, sc, sh, residual=r)),
         ("dgrad+norm moments", lambda: ops.conv_dgrad(x, wp, G3, x.shape, norm=(r, sc, sh))),
         ("dgrad+residual", lambda: ops.conv_dgrad(x, wp, G3, x.shape, residual=r)))
for name, fn in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-24s %.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3))
