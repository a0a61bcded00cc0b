"""Helpers shared by the parity tests."""
import hashlib
import inspect
import os

import numpy as np
import torch


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def rel_err(a, b):
    """max |a-b| / max |b| (b = reference)."""
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def brats_args(**kw):
    import argparse
    d = dict(num_classes=4, dataset='BraTS', model_name='U', input_modality='four', end_epochs=99,
             lambda_epochs=50, Uncertainty_Loss=False)
    d.update(kw)
    return argparse.Namespace(**d)


def l2_rel_err(a, b):
    """||a-b||_2 / ||b||_2 (b = reference)."""
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# End-to-end GRADIENT tolerances (fp32 path).  Every kernel is checked against torch fp64 to ~1e-6 in
# tests/test_gpu_ops.py; end to end, however, a ReLU gate whose pre-activation is within rounding of zero
# can open on one side and stay shut on the other (GPU fp32 vs CPU fp32 sum order).  One such flip moves
# that voxel's gradient by O(1) and a weight gradient by ~1/voxels: observed up to 3e-3 on the small
# golden volumes (measured: the oracle's own h.grad equals BN-backward of its inputs to 1e-7, ours equals
# fp64 BN-backward of ours to 1e-6, inputs agree to 1e-6, and the two outputs differ only at flipped gates).
GRAD_L2_TOL = 5e-3
GRAD_MAX_TOL = 2e-2


# ---------------------------------------------------------------- measured parity numbers
# Every parity assertion goes through check(): the measured value is recorded next to its bar and written to
# gpurun_out/parity_measured.json at the end of the session (tests/conftest.py), so the numbers actually achieved
# are on record (profiles/r02_parity_measured.json is a copy of one such run).  DEVIS_PARITY_MEASURE=1 records
# without asserting (used once to find what each bar can be).
# Bars of BASELINE.json's north_star: fp32 logits / alpha / u 1e-5, bf16 1e-2 (max |a-b| / max |b|), loss 1e-5
# relative in fp32 and 1e-3 in bf16.
FP32_BAR = 1e-5
BF16_BAR = 1e-2
LOSS_BF16_BAR = 1e-3
# bf16 exceptions to the 1e-2 MAX-norm bar (measured in round 2; each is asserted next to the
# L2-relative error, which stays under 1e-2 for every backbone): the 3-D U-Net of the headline config meets 1e-2
# (5.0e-3 at full size, 5.8e-3 on the goldens); the deeper / un-normalised-skip backbones do not in the worst voxel:
#   AttUnet 1.08e-2 (full size) - three more bf16 BatchNorm + sigmoid gates per skip
#   V-Net   2.2e-2 (full size), 1.6e-2 (golden) - additive skips carry un-normalised bf16 sums through nine stages
#   2-D U-Nets 2.3e-2 (16 x 256^2), 1.7e-2 (golden), 1.6e-2 (ISIC 224 x 300); their LOGITS 2.0e-2 in L2 as well ("2D_L2").
#     r03 audit (tools/bf16_layer_audit.py, profiles/r03_bf16_layer_audit.txt): the fp32 oracle with bf16 roundings put
#     in at the points where any bf16 tensor-core pipeline has them (oracle.unet2d_forward_bf16_points) differs from the
#     fp32 oracle by the SAME amount at every layer (5.3e-3 after conv1 ... 5.5e-2 after conv5 ... 1.99e-2 at the logits,
#     equal to ours to four digits); rounding the weights alone costs 9.6e-3.  So these caps are what bf16 costs in this
#     23-layer network without residual paths, and tests/test_gpu_unet2d.py asserts, layer by layer, that the kernels
#     add nothing on top (ours-vs-fp32 <= 1.05 x points-vs-fp32).  The evidence of the same runs (what the loss and u
#     consume) is within 1e-2 in L2 (asserted).
BF16_MAX_EXCEPTION = {"AU": 1.5e-2, "V": 3e-2, "2D": 3e-2, "2D_L2": 2.5e-2}
PARITY_LOG = {}


def check(value, bar, tag=""):
    """assert value <= bar, recording the measurement under '<test function>:<line>[:tag]'"""
    fr = inspect.stack()[1]
    fn = fr.function
    for outer in inspect.stack()[2:6]:          # a helper called from the test: name the test
        if outer.function.startswith("test_"):
            fn = outer.function + "/" + fn if not fn.startswith("test_") else fn
            break
    name = "%s:%s:%d%s" % (os.path.basename(fr.filename), fn, fr.lineno, (":" + str(tag)) if tag != "" else "")
    value = float(value)
    PARITY_LOG.setdefault(name, []).append([value, float(bar), os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]])
    if os.environ.get("DEVIS_PARITY_MEASURE") == "1":
        return
    assert value <= bar, "%s: measured %.3e > bar %.1e" % (name, value, bar)


def bf16_ulps(y, ref, slack=1e-5):
    """How far a bf16 kernel output is from the fp32-accumulated reference rounded once, in bf16 units in the
    last place of the reference: max over elements of (|y - ref| - slack * max|ref|) / ulp_bf16(ref).
    <= 1 means every element is the correctly rounded value or its neighbour; `slack` absorbs the fp32
    accumulation-order error where the result is a cancellation (tiny next to its addends)."""
    y = torch.as_tensor(np.asarray(y) if not torch.is_tensor(y) else y).double().cpu()
    ref = torch.as_tensor(np.asarray(ref) if not torch.is_tensor(ref) else ref).double().cpu()
    assert y.shape == ref.shape, (y.shape, ref.shape)
    _, e = torch.frexp(ref.abs().clamp_min(1e-30).float())            # |ref| = m * 2^e, m in [0.5, 1)
    ulp = torch.pow(torch.tensor(2.0, dtype=torch.float64), (e - 8).double())
    excess = ((y - ref).abs() - slack * ref.abs().max()).clamp_min(0.0)
    return (excess / ulp).max().item()


# the kernels may add this much to what the bf16 rounding points themselves cost (tests/test_gpu_unet2d.py audit)
BF16_OVER_ROUNDING_POINTS = 1.05

# End-to-end bf16 gradients, per parameter tensor, L2-relative against the oracle's fp32 autograd.
# Measured at 2 x 4 x 128^3 (r03, tests/test_gpu_unet3d.py full-size test): worst tensor 25 %, median 5.3 % - while
# every tensor's NORM agrees to 6.5e-4.  The fp32 oracle with bf16 roundings in its FORWARD only and an exact fp32
# backward (oracle.unet3d_forward_bf16_points, straight-through roundings) shows the same: worst 24 %, median 5.3 %.
# Cause: a forward that is 5e-3 off flips the ReLU gate of ~0.2 % of the voxels per layer (pre-activations within the
# error of zero), each flip moves that voxel's gradient by O(1), and the flips of ~20 layers add up on the way back.
# Per-element gradient agreement at the 2e-2 level is therefore not a property any bf16 forward has on this network;
# the test holds the worst and the median tensor to what the rounding points alone cost, times this factor:
BF16_GRAD_OVER_POINTS = 1.3
