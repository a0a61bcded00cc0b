#!/usr/bin/env python3
"""Graph-replayed sliding-window inference (graph_streams 1 and 3) against the eager run on a clean and a noised
volume, fp32 and bf16: mismatching evidence voxels per patch (all zeros expected; tests/test_gpu_tailor.py asserts it)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU
from devis_amd.predict import sliding_window_infer, patch_regions, CROPS
for dt in (torch.float32, torch.bfloat16):
    args = bench.brats_args(dt)
    torch.manual_seed(0)
    m = TMSU(args).eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 4, 240, 240, 160), generator=g).cuda()
    n = (x + 0.5 * torch.randn((1, 4, 240, 240, 160), generator=g).cuda())
    thr = (0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)
    with torch.no_grad():
        ea = sliding_window_infer(x, m.backbone, thr, want_evidence=True)
        eb = sliding_window_infer(n, m.backbone, thr, want_evidence=True)
        for gs in (1, 3):
            for rep in range(2):
                ga = sliding_window_infer(x, m.backbone, thr, want_evidence=True, graph_streams=gs)
                gb = sliding_window_infer(n, m.backbone, thr, want_evidence=True, graph_streams=gs)
                for name, e, gr in (("x", ea, ga), ("n", eb, gb)):
                    bad = (e["evidence"] != gr["evidence"]).any(dim=1)[0]
                    per = []
                    for i in range(len(CROPS)):
                        dst, _ = patch_regions(i)
                        per.append(int(bad[dst].sum()))
                    print(dt, "gs", gs, "rep", rep, name, "mismatch voxels per patch", per, flush=True)
