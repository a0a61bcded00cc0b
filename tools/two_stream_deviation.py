#!/usr/bin/env python3
"""How far do the two-stream gradients stray from the one-stream (bit-reproducible) ones?  Same weights, same batch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU
args = bench.brats_args(torch.bfloat16)
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 4, 128, 128, 128), generator=g).cuda(); y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).cuda()
torch.manual_seed(0)
m = TMSU(args).train()
def grads():
    _, loss = m(x, y, 25, 'train', 'BraTS')
    for p in m.parameters(): p.grad = None
    loss.backward(); torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in m.named_parameters()}
os.environ["DEVIS_WGRAD_STREAM"] = "0"
ref = grads(); ref2 = grads()
print("one stream reproducible:", all(torch.equal(ref[n], ref2[n]) for n in ref))
os.environ["DEVIS_WGRAD_STREAM"] = "1"
worst_l2 = worst_max = 0.0; wn = ""
for it in range(8):
    cur = grads()
    for n in ref:
        d = (cur[n] - ref[n]).double()
        l2 = (d.norm() / ref[n].double().norm().clamp_min(1e-30)).item()
        mx = (d.abs().max() / ref[n].abs().max().clamp_min(1e-30)).item()
        if l2 > worst_l2: worst_l2, wn = l2, n
        worst_max = max(worst_max, mx)
print("two streams vs one stream over 8 backward passes: worst per-tensor L2-relative %.3e (%s), worst max-relative %.3e" % (worst_l2, wn, worst_max))
