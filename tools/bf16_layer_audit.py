#!/usr/bin/env python3
"""Per-layer bf16 error audit of the 2-D U-Net.

Runs Unet2D(1, 2) on the config-2 tensors (16 x 1 x 256 x 256 by default) three ways and compares the output of every
DoubleConv / ConvTranspose and the logits:
    ours   devis_amd's bf16 MFMA path on the GPU
    fp32   oracle.unet2d_forward: the reference's fp32 arithmetic (CPU)
    pts    oracle.unet2d_forward_bf16_points: the fp32 oracle with bf16 roundings at the points where ANY bf16
           tensor-core pipeline has them (weights, stored tensors, conv operands)
Columns: L2-relative error  ours-vs-fp32,  pts-vs-fp32,  ours-vs-pts  (and the same in max-norm for the logits).
ours-vs-pts is what the kernels add on top of the rounding points; pts-vs-fp32 is what the rounding points cost in
this network, whatever the implementation.

    python tools/bf16_layer_audit.py [--batch 16] [--size 256] [--out gpurun_out/r03_bf16_layer_audit.txt]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ORDER = ["conv1", "conv2", "conv3", "conv4", "conv5", "up6", "conv6", "up7", "conv7", "up8", "conv8", "up9", "conv9", "logits"]


def product_taps(net, x):
    """devis_amd forward with the output of every DoubleConv / ConvTranspose recorded (NCHW fp32 on the CPU)."""
    from devis_amd import unet2d
    taps = {}
    dc0, up0 = unet2d._Run2d.double_conv, unet2d._Run2d.up

    def dc(self, name, x1, x2=None):
        y = dc0(self, name, x1, x2)
        taps[name] = y.detach().float().permute(0, 3, 1, 2).cpu()
        return y

    def up(self, name, t):
        y = up0(self, name, t)
        taps[name] = y.detach().float().permute(0, 3, 1, 2).cpu()
        return y

    unet2d._Run2d.double_conv, unet2d._Run2d.up = dc, up
    try:
        with torch.no_grad():
            logits = net.forward_channels_last(x.cuda())
    finally:
        unet2d._Run2d.double_conv, unet2d._Run2d.up = dc0, up0
    taps["logits"] = logits.detach().float().permute(0, 3, 1, 2).cpu()
    return taps


def l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def mx(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def audit(batch=16, size=256, seed=4):
    from oracle import devis_oracle as O
    from devis_amd import Unet2D
    g = torch.Generator().manual_seed(2100)
    x = torch.randn((batch, 1, size, size), generator=g)
    torch.manual_seed(seed)
    net = Unet2D(1, 2, compute_dtype=torch.bfloat16)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().train()
    ours = product_taps(net, x)
    with torch.no_grad():
        t32, tpt = {}, {}
        t32["logits"] = O.unet2d_forward({k: v.clone() for k, v in sd.items()}, x, True, taps=t32)
        tpt["logits"] = O.unet2d_forward_bf16_points(sd, x, True, taps=tpt)
    rows = []
    for k in ORDER:
        rows.append((k, l2(ours[k], t32[k]), l2(tpt[k], t32[k]), l2(ours[k], tpt[k]),
                     mx(ours[k], t32[k]), mx(tpt[k], t32[k]), mx(ours[k], tpt[k])))
    return rows


def fmt(rows, batch, size):
    out = ["# Unet2D(1,2) bf16, %d x 1 x %d x %d, training-mode BatchNorm: error of every block's output" % (batch, size, size),
           "# ours = devis_amd bf16 (GPU), fp32 = oracle.unet2d_forward, pts = oracle.unet2d_forward_bf16_points",
           "%-8s | %-32s | %-32s" % ("", "L2-relative", "max |a-b| / max |b|"),
           "%-8s | %10s %10s %10s | %10s %10s %10s" % ("layer", "ours-fp32", "pts-fp32", "ours-pts", "ours-fp32", "pts-fp32", "ours-pts")]
    for r in rows:
        out.append("%-8s | %10.3e %10.3e %10.3e | %10.3e %10.3e %10.3e" % r)
    return "\n".join(out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_bf16_layer_audit.txt"))
    a = ap.parse_args()
    text = fmt(audit(a.batch, a.size), a.batch, a.size)
    print(text)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.write(text + "\n")
