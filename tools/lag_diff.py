#!/usr/bin/env python3
"""Which gradients of the FIRST REPLAYED step differ from an eager step (fresh process, lag from argv)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU, unet3d, launch_plan
from devis_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
unet3d._WG_LAG = tuple(int(v) for v in sys.argv[1].split(":"))
g = torch.Generator().manual_seed(1234)
x = torch.randn((2, 4, 128, 128, 128), generator=g).to(dev)
y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).to(dev)

if os.environ.get("PREWS") == "1":
    from devis_amd import ops
    side = unet3d._side_stream(dev)
    with ops.stream_override(side):
        ops.workspace(256 << 20, dev, "ws_side")
    ops.workspace(256 << 20, dev, "ws")

def two_steps(plan):
    launch_plan.ENABLED = plan
    torch.manual_seed(0)
    args = bench.brats_args(torch.bfloat16)
    model = TMSU(args).train()
    opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)
    for it in range(2):
        _, loss = model(x, y, 25, 'train', 'BraTS')
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if it == 0:
            opt.step()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, float(loss)

a, la = two_steps(True)
b, lb = two_steps(False)
nbad = sum(0 if torch.equal(a[n], b[n]) else 1 for n in b)
print("SUMMARY", sys.argv[1], {k: v for k, v in os.environ.items() if k.startswith("DEVIS_") or k == "PREWS"}, "tensors differing:", nbad)
print("loss plan %.6f eager %.6f" % (la, lb))
for n in b:
    if not torch.equal(a[n], b[n]):
        d = (a[n] - b[n]).abs()
        print("DIFF %-36s n=%-7d bad=%-7d max|d| %.3e ref max %.3e finite %s" % (
            n, d.numel(), int((d > 0).sum() + (~torch.isfinite(d)).sum()), float(d.nan_to_num(1e30).max()),
            float(b[n].abs().max()), bool(torch.isfinite(a[n]).all())))
