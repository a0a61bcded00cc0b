#!/usr/bin/env python3
"""cProfile of the host side of the bench step (where do the milliseconds of enqueue time go?)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from devis_amd import TMSU  # noqa: E402
from devis_amd.optim import FusedAdam  # noqa: E402
from devis_amd.train import adjust_learning_rate, freeze_gc  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = bench.brats_args(torch.bfloat16)
torch.manual_seed(0)
model = TMSU(args)
model.train()
opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)
g = torch.Generator().manual_seed(1234)
x = torch.randn((2, 4, 128, 128, 128), generator=g).to(dev)
y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).to(dev)


def step():
    adjust_learning_rate(opt, 25, args.end_epochs, args.lr)
    _, loss = model(x, y, 25, 'train', 'BraTS')
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


# the backbone's backward runs on the autograd thread: give it a profiler of its own
from devis_amd import unet3d as _u  # noqa: E402
_bpr = cProfile.Profile()
_orig_bwd = _u._Run.backward


def _prof_bwd(self, *a, **k):
    _bpr.enable()
    try:
        return _orig_bwd(self, *a, **k)
    finally:
        _bpr.disable()


freeze_gc()
for _ in range(10):
    step()
torch.cuda.synchronize()
N = 30
_u._Run.backward = _prof_bwd
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(40)
print("==== backbone backward (autograd thread)")
sb = pstats.Stats(_bpr)
sb.sort_stats("tottime").print_stats(40)
sb.sort_stats("cumulative").print_stats(30)
