#!/usr/bin/env python3
"""Is the OUTPUT of the elementwise norm-backward apply kernel bit-stable while MFMA kernels share the chip?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops
from devis_amd.unet3d import _side_stream
geom = ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
g = torch.Generator().manual_seed(0)
dev = torch.device("cuda", 0)
side = _side_stream(dev)
for c, sz in ((32, 64), (16, 128), (64, 32)):
    w = (torch.randn((c, c, 3, 3, 3), generator=g) * 0.05).cuda()
    x = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    dy = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    da = (torch.randn((2, sz, sz, sz, c), generator=g) * 1e-3).bfloat16().cuda()
    gamma = (torch.rand(c, generator=g) + 0.5).cuda(); beta = torch.zeros(c).cuda()
    sc, sh, mean, rstd = ops.bn_stats(x, gamma, beta, torch.zeros(c).cuda(), torch.ones(c).cuda(), True)
    dw = torch.empty_like(w); db = torch.empty(c, device=dev); dg = torch.empty(c, device=dev); dbt = torch.empty(c, device=dev)
    ref = None; bad = 0; nbad = 0
    for it in range(24):
        concurrent = it >= 4
        torch.cuda.synchronize()
        if concurrent:
            ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
            with ops.stream_override(side):
                for _ in range(3):
                    ops.conv_wgrad(x, None, dy, geom, w.shape, sc, sh, out=(dw, db), slot="ws_side")
        dh, _, _ = ops.norm_relu_bwd(da, x, sc, sh, gamma, mean, rstd, 0, True, out=(dg, dbt))
        torch.cuda.synchronize()
        if ref is None: ref = dh.clone()
        elif concurrent and not torch.equal(dh, ref):
            bad += 1; nbad = int((dh != ref).sum())
            diff_idx = (dh != ref).flatten().nonzero().flatten()
            par = (diff_idx % 2).float().mean().item()
    print("%3d ch @ %3d^3: apply output mismatches under concurrent MFMA kernels %d/20%s" % (c, sz, bad, "" if not bad else "  (last: %d elements, odd-channel share %.2f)" % (nbad, par)))
