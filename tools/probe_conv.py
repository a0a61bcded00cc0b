#!/usr/bin/env python3
"""Launch the dominant kernels (3x3x3 conv 16->16 at 2x128^3: fwd, dgrad, wgrad) a few times - the target of
the rocprofv3 --pmc passes whose FETCH_SIZE / WRITE_SIZE feed bench.py's roofline.traffic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    torch.cuda.set_device(0)
    res = bench.kernel_probe(2, 128, torch.bfloat16)
    for k, v in res.items():
        print(k, "%.3f ms  %.1f TFLOP/s" % (v[0] * 1e3, v[1] / v[0] / 1e12))
