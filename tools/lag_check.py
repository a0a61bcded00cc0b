#!/usr/bin/env python3
"""Gradients of one training step at the bench shape under two weight-gradient hand-over lags, compared tensor by
tensor (eager, DEVIS_LAUNCH_PLAN=0), then the loss of a few steps under each.  python tools/lag_check.py [a:b]"""
import os, sys
os.environ.setdefault("DEVIS_LAUNCH_PLAN", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU, unet3d
from devis_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
lag = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1:2").split(":"))
g = torch.Generator().manual_seed(1234)
x = torch.randn((2, 4, 128, 128, 128), generator=g).to(dev)
y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).to(dev)

NOSYNC = False

def grads(l, steps=1):
    unet3d._WG_LAG = l
    torch.manual_seed(0)
    args = bench.brats_args(torch.bfloat16)
    model = TMSU(args).train()
    opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)
    losses = []
    if os.environ.get("LAG_FREEZE") == "1":
        from devis_amd.train import freeze_gc
        freeze_gc()
    for _ in range(steps):
        if os.environ.get("LAG_ADJ") == "1":
            from devis_amd.train import adjust_learning_rate
            adjust_learning_rate(opt, 25, args.end_epochs, args.lr)
        _, loss = model(x, y, 25, 'train', 'BraTS')
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if NOSYNC:
            out = None
            losses.append(loss if os.environ.get("LAG_NOCLONE") == "1" else loss.detach().clone())
        else:
            torch.cuda.synchronize()
            out = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            losses.append(float(loss.item()))
        opt.step()
    return out, [float(v) for v in losses]

SKIP = os.environ.get("LAG_SKIP_REF") == "1"
ref, _ = ({}, None) if SKIP else grads((0, 1))
got, _ = ({}, None) if SKIP else grads(lag)
bad = 0
for n in ref:
    a, b = ref[n], got[n]
    if not torch.equal(a, b):
        bad += 1
        fin = bool(torch.isfinite(b).all())
        print("DIFF %-40s max|d| %.3e  ref max %.3e  finite %s" % (n, float((a - b).abs().max()), float(a.abs().max()), fin))
print("tensors differing after step 1:", bad, "of", len(ref))
NOSYNC = os.environ.get("LAG_NOSYNC") == "1"
for l in ((lag, (0, 1)) if SKIP else ((0, 1), lag)):
    _, ls = grads(l, steps=int(os.environ.get("LAG_STEPS", "8")))
    print("lag", l, "losses", " ".join("%.6f" % v for v in ls))
