#!/usr/bin/env python3
"""Time the fp32 matrix-core conv kernels (csrc/conv_f32_mfma.hip) on the 3-D U-Net's layer shapes at B = 2: forward with
prologue, data gradient, weight gradient (kernel + partial reduction).  HIP events, 10 calls each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops, _lib
print("library:", _lib.LIB_PATH)
g = torch.Generator().manual_seed(0)
G3 = ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
def timeit(fn, reps=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for c, sz in ((16, 128), (32, 64), (64, 32), (128, 16)):
    x = torch.randn((2, sz, sz, sz, c), generator=g).cuda()
    dy = torch.randn((2, sz, sz, sz, c), generator=g).cuda()
    w = (torch.randn((c, c, 3, 3, 3), generator=g) * 0.05).cuda()
    wp = ops.pack_conv_weight(w, torch.float32)
    sc = (torch.rand((2, c), generator=g) + 0.5).cuda(); sh = torch.randn((2, c), generator=g).cuda()
    fl = 2 * 27 * c * c * 2 * sz ** 3
    t_f = timeit(lambda: ops.conv_fwd(x, None, wp, None, G3, sc, sh))
    t_d = timeit(lambda: ops.conv_dgrad(dy, wp, G3, x.shape))
    t_w = timeit(lambda: ops.conv_wgrad(x, None, dy, G3, w.shape, sc, sh))
    print("%3d ch @ %3d^3: fwd %7.1f us (%5.1f TF/s)  dgrad %7.1f us (%5.1f)  wgrad %7.1f us (%5.1f)"
          % (c, sz, t_f, fl / t_f / 1e6, t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6))
