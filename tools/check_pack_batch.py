#!/usr/bin/env python3
"""devis_pack_weights_batch against the per-tensor packing calls on a few weight shapes (bit-exact)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops, _lib
print("library:", _lib.LIB_PATH, flush=True)
g = torch.Generator().manual_seed(3)
ents = []
for i, (kind, shape) in enumerate((("conv", (16, 4, 3, 3, 3)), ("conv", (32, 16, 3, 3, 3)), ("conv", (128, 128, 3, 3, 3)), ("conv", (4, 16, 1, 1, 1)),
                                   ("convT", (64, 32, 2, 2, 2)), ("conv", (32, 64, 1, 1, 1)), ("conv", (32, 3, 3, 3)), ("convT", (256, 128, 2, 2)))):
    ents.append(("w%d" % i, torch.randn(shape, generator=g).cuda(), kind))
pk = ops.WeightPacker(ents, torch.bfloat16, torch.device("cuda"))
views = pk.pack()
torch.cuda.synchronize()
print("packed", flush=True)
lib = _lib.load()
for name, w, kind in ents:
    a, b = w.shape[0], w.shape[1]
    t = w.numel() // (a * b)
    fwd, dg = views[name]
    if kind == "conv":
        ref = w.reshape(a, b, t).permute(0, 2, 1).contiguous().bfloat16()
        refd = w.reshape(a, b, t).flip(2).permute(1, 2, 0).contiguous().bfloat16()
    else:
        ref = w.reshape(a, b, t).permute(2, 1, 0).contiguous().bfloat16()
        refd = w.reshape(a, b, t).permute(0, 2, 1).contiguous().bfloat16().reshape(a, t * b)
    assert torch.equal(fwd, ref.view(fwd.shape)), name
    assert torch.equal(dg, refd.view(dg.shape)), name + " dgrad"
print("pack_weights_batch: bit-exact on %d tensors" % len(ents))
