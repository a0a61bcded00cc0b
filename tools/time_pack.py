#!/usr/bin/env python3
"""Time the per-step weight packing launch (devis_pack_weights_batch) of the BraTS 3-D U-Net and of Unet2D(1,2), HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU, ops, unet2d
torch.manual_seed(0)
for name, bb in (("3-D U-Net (4,16,4)", TMSU(bench.brats_args(torch.bfloat16)).backbone), ("Unet2D(1,2)", unet2d.Unet2D(1, 2).cuda())):
    ents = [(n, m.weight, "convT" if "Transpose" in type(m).__name__ else "conv") for n, m in bb.named_modules()
            if type(m).__name__.startswith("Conv")]
    pk = ops.WeightPacker(ents, torch.bfloat16, torch.device("cuda"))
    for _ in range(5):
        pk.pack()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        pk.pack()
    b.record()
    torch.cuda.synchronize()
    print("%-22s %4d tensors %6d chunks  %.1f us per launch" % (name, len(ents), pk.n_chunks, a.elapsed_time(b) * 20))
