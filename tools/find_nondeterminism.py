#!/usr/bin/env python3
"""Run the same seeded forward/backward several times and list the gradients that are not bit-identical."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU
args = bench.brats_args(torch.bfloat16)
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 4, 128, 128, 128), generator=g).cuda(); y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).cuda()
def run():
    torch.manual_seed(0)
    m = TMSU(args).train()
    out = []
    for it in range(3):
        _, loss = m(x, y, 25, 'train', 'BraTS')
        for p in m.parameters(): p.grad = None
        loss.backward()
        torch.cuda.synchronize()
        out.append({n: p.grad.clone() for n, p in m.named_parameters()})
    return out, loss.item()
a, la = run()
b, lb = run()
print("loss", la, lb)
for it in range(3):
    bad = [n for n in a[it] if not torch.equal(a[it][n], b[it][n])]
    print("iteration", it, "differing gradients:", len(bad), bad[:12])
# within one run: iterations have identical weights (no optimizer step) -> gradients must repeat
bad = [n for n in a[0] if not torch.equal(a[1][n], a[2][n])]
print("iteration 1 vs 2 of the same model:", len(bad))
for n in list(a[0])[::-1][:40]:
    d = (a[1][n] - a[2][n]).abs().max().item(); r = a[1][n].abs().max().item()
    print("  %-40s max|diff| %.3e  max|g| %.3e%s" % (n, d, r, "" if d == 0 else "   <--"))

order = list(a[0])[::-1]
first = next((n for n in order if not torch.equal(a[1][n], a[2][n])), None)
print("FIRST differing tensor in backward order:", first, "(index %s of %d)" % (order.index(first) if first else "-", len(order)))
