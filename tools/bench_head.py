#!/usr/bin/env python3
"""Time the fused evidential head + loss kernels alone at the BraTS size (B x 128^3, K = 4, channels-last logits,
int64 labels) with HIP events on the launch stream; also the program rocprofv3 --pmc runs for their counters.
    python tools/bench_head.py [--batch 2] [--size 128] [--dtype bf16] [--reps 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--labels", default="i64", choices=["i64", "u8"])
    a = ap.parse_args()
    from devis_amd import ops
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    S, B = a.size, a.batch
    z = torch.randn((B, S, S, S, 4), generator=g).to(dtype).to(dev)
    lab = torch.randint(0, 4, (B, S, S, S), generator=g).to(dev)
    if a.labels == "u8":
        lab = lab.to(torch.uint8)
    one = torch.ones((), device=dev)
    vox = B * S ** 3
    esz = z.element_size()
    lsz = lab.element_size()

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e-3

    scal, _, _ = ops.evid_loss_fwd(z, True, lab, 0.5, True)
    out = {}
    for name, fn, nbytes in (
            ("fwd+evidence", lambda: ops.evid_loss_fwd(z, True, lab, 0.5, True), vox * (4 * esz + lsz + 16)),
            ("fwd", lambda: ops.evid_loss_fwd(z, True, lab, 0.5, False), vox * (4 * esz + lsz)),
            ("bwd", lambda: ops.evid_loss_bwd(z, True, lab, 0.5, scal, one), vox * (4 * esz + lsz + 4 * esz))):
        t = timeit(fn)
        out[name] = {"us": t * 1e6, "GB/s_alg": nbytes / t / 1e9, "frac_of_8TBs": nbytes / t / 8e12}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
