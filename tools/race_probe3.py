#!/usr/bin/env python3
"""A wgrad on the side stream while the main stream runs LDS-free elementwise kernels (they can share its CUs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
from devis_amd.unet3d import _side_stream
geom = ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
g = torch.Generator().manual_seed(0)
dev = torch.device("cuda", 0)
side = _side_stream(dev)
big = torch.randn((2, 128, 128, 128, 16), generator=g).bfloat16().cuda()
bsc = torch.ones((2, 16)).cuda(); bsh = torch.zeros((2, 16)).cuda()
for c, sz, pro in ((16, 128, True), (32, 64, False), (32, 64, True), (64, 32, True), (128, 16, True), (64, 32, False)):
    w = (torch.randn((c, c, 3, 3, 3), generator=g) * 0.05).cuda()
    x = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    dy = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    sc = (torch.rand((2, c), generator=g) + 0.5).cuda(); sh = torch.zeros((2, c)).cuda()
    dw = torch.empty_like(w); db = torch.empty(c, device=dev)
    ref = None; bad = 0
    for it in range(24):
        concurrent = it >= 4
        torch.cuda.synchronize()
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        with ops.stream_override(side):
            ops.conv_wgrad(x, None, dy, geom, w.shape, sc if pro else None, sh if pro else None, out=(dw, db), slot="ws_side")
        if concurrent:
            for _ in range(3):
                ops.affine_act(big, bsc, bsh, relu=True)
        torch.cuda.synchronize()
        cur = torch.cat([dw.flatten(), db])
        if ref is None: ref = cur.clone()
        elif concurrent: bad += not torch.equal(cur, ref)
    print("pro=%d %3d ch @ %3d^3: wgrad mismatches with a concurrent elementwise kernel %d/20" % (pro, c, sz, bad))
