#!/usr/bin/env python3
"""wgrad on a side stream while the dgrad of the same layer runs on the main stream: are both outputs bit-stable?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
from devis_amd.unet3d import _side_stream
geom = ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
g = torch.Generator().manual_seed(0)
dev = torch.device("cuda", 0)
side = _side_stream(dev)
for c, sz, pro in ((16, 128, True), (32, 64, True), (64, 32, True), (128, 16, True)):
    w = (torch.randn((c, c, 3, 3, 3), generator=g) * 0.05).cuda()
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    x = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    dy = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    res = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    sc = (torch.rand((2, c), generator=g) + 0.5).cuda(); sh = torch.zeros((2, c)).cuda()
    dw = torch.empty_like(w); db = torch.empty(c, device=dev)
    ref = None
    bad_w = bad_d = bad_alone = 0
    for it in range(24):
        concurrent = it >= 4
        torch.cuda.synchronize()
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        if concurrent:
            with ops.stream_override(side):
                ops.conv_wgrad(x, None, dy, geom, w.shape, sc if pro else None, sh if pro else None, out=(dw, db), slot="ws_side")
            dx, _, part = ops.conv_dgrad(dy, wp, geom, x.shape, norm=(x, sc, sh))
        else:
            ops.conv_wgrad(x, None, dy, geom, w.shape, sc if pro else None, sh if pro else None, out=(dw, db))
            torch.cuda.synchronize()
            dx, _, part = ops.conv_dgrad(dy, wp, geom, x.shape, norm=(x, sc, sh))
        torch.cuda.synchronize()
        cur = (dw.clone(), db.clone(), torch.cat([dx.float().flatten(), part[0].float().flatten()]))
        if ref is None: ref = cur
        else:
            w_ok = torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]); d_ok = torch.equal(cur[2], ref[2])
            if concurrent: bad_w += not w_ok; bad_d += not d_ok
            else: bad_alone += not (w_ok and d_ok)
    print("pro=%d " % pro + "%3d ch @ %3d^3: serial mismatches %d; concurrent: wgrad %d/20, dgrad+stats %d/20" % (c, sz, bad_alone, bad_w, bad_d))
