#!/usr/bin/env python3
"""Capture one whole training step (fwd + bwd + Adam) in a HIP graph and time replay vs eager."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from devis_amd import TMSU

args = bench.brats_args(torch.bfloat16)
torch.manual_seed(0)
m = TMSU(args); m.train()
from devis_amd.optim import FusedAdam
opt = FusedAdam(m.parameters(), lr=0.002, weight_decay=1e-5, module=m)   # probe only: step scalars are baked into the graph
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 4, 128, 128, 128), generator=g).cuda(); y = torch.randint(0, 4, (2, 128, 128, 128), generator=g).cuda()

def body():
    _, loss = m(x, y, 25, 'train', 'BraTS')
    loss.backward()
    opt.step()
    return loss

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        l0 = body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager loss", l0.item())
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    loss = body()
torch.cuda.synchronize()
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    graph.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("graph replay %.3f ms/step, loss %.6f" % (dt * 1e3, loss.item()))
