#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: sliding-window evidential inference of a 4x240x240x160 volume (8 patches of 128^3,
predict.py:159-191) through the backbone (eval mode) + the fused inference head: pred mask, u map, mne, 11 threshold masks."""
import argparse, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU
from devis_amd.predict import sliding_window_infer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--model", default="U")
ap.add_argument("--patch-batch", type=int, default=1)
ap.add_argument("--graph-streams", type=int, default=0)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
a = ap.parse_args()
args = bench.brats_args(torch.bfloat16 if a.dtype == "bf16" else torch.float32); args.model_name = a.model
torch.manual_seed(0)
m = TMSU(args).eval()
g = torch.Generator().manual_seed(1234)
x = torch.randn((1, 4, 240, 240, 160), generator=g).cuda()
thr = (0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)      # trainUMIS.py:521

def step():
    with torch.no_grad():
        return sliding_window_infer(x, m.backbone, thr, patch_batch=a.patch_batch, graph_streams=a.graph_streams)

for _ in range(2):
    out = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"workload": "configs[4]: sliding-window inference, 1 GPU, 4x240x240x160 volume = 8 patches of 128^3, model %s %s, patch_batch %d, graph_streams %d" % (a.model, a.dtype, a.patch_batch, a.graph_streams),
                  "ms_per_volume": dt * 1e3, "volumes_per_s": 1 / dt, "patch_voxels_per_s": 8 * 128 ** 3 / dt,
                  "outputs": sorted(k for k, v in out.items() if v is not None)}))
