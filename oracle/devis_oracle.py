"""CPU oracle for the DEviS evidential U-Net hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm.
It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``devis_amd/`` (the product) imports, calls or falls back to it.

Pinning: the reference holds no tests or golden vectors of its own,
so this restatement is pinned by running the real reference (imported from
/root/reference on CPU, see ``tests/golden/make_goldens.py``) on seeded inputs
and asserting bit-equality with this file; the resulting vectors are committed
under ``tests/golden/``.

It is written functionally (a ``state_dict`` in, tensors out) rather than as
``nn.Module`` classes; every function cites the reference lines it restates
(paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# 3D residual U-Net  (models/lib/UNet3DZoo.py:243-305)
# ----------------------------------------------------------------------------

def _conv(sd: StateDict, key: str, x: Tensor, stride: int = 1, pad: int = 1) -> Tensor:
    fn = F.conv3d if x.dim() == 5 else F.conv2d
    return fn(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=pad)


def _gn_relu(sd: StateDict, key: str, x: Tensor) -> Tensor:
    # GroupNorm(8, C) then ReLU: UNet3DZoo.py:12, :39-40
    return F.relu(F.group_norm(x, 8, sd[key + ".weight"], sd[key + ".bias"], 1e-5))


def _bn_relu(sd: StateDict, key: str, x: Tensor, training: bool) -> Tensor:
    # nn.BatchNorm{2,3}d defaults: eps 1e-5, momentum 0.1; running buffers are
    # updated in place inside ``sd`` when training (UNet3DZoo.py:149-162).
    if training:
        sd[key + ".num_batches_tracked"] += 1
    y = F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                     sd[key + ".weight"], sd[key + ".bias"], training, 0.1, 1e-5)
    return F.relu(y)


def _en_block(sd: StateDict, name: str, x: Tensor) -> Tensor:
    # EnBlock.forward, UNet3DZoo.py:47-56
    h = _conv(sd, name + ".conv1", _gn_relu(sd, name + ".bn1", x))
    h = _conv(sd, name + ".conv2", _gn_relu(sd, name + ".bn2", h))
    return h + x


def _de_block(sd: StateDict, name: str, x: Tensor, training: bool) -> Tensor:
    # DeBlock.forward, UNet3DZoo.py:156-165 (conv -> BN -> ReLU, twice, + x)
    h = _bn_relu(sd, name + ".bn1", _conv(sd, name + ".conv1", x), training)
    h = _bn_relu(sd, name + ".bn2", _conv(sd, name + ".conv2", h), training)
    return h + x


def _de_up_cat(sd: StateDict, name: str, x: Tensor, prev: Tensor) -> Tensor:
    # DeUp_Cat.forward, UNet3DZoo.py:124-130: 1x1 -> ConvT k2 s2 -> cat(prev, up) -> 1x1
    h = _conv(sd, name + ".conv1", x, pad=0)
    h = F.conv_transpose3d(h, sd[name + ".conv2.weight"], sd[name + ".conv2.bias"], stride=2)
    return _conv(sd, name + ".conv3", torch.cat((prev, h), dim=1), pad=0)


def unet3d_forward(sd: StateDict, x: Tensor, training: bool = True) -> Tensor:
    """Unet.forward, UNet3DZoo.py:276-305.  ``InitConv`` dropout has p=0 (:248)."""
    h = _conv(sd, "InitConv.conv", x)
    s1 = _en_block(sd, "EnBlock1", h)
    h = _conv(sd, "EnDown1.conv", s1, stride=2)
    h = _en_block(sd, "EnBlock2_1", h)
    s2 = _en_block(sd, "EnBlock2_2", h)
    h = _conv(sd, "EnDown2.conv", s2, stride=2)
    h = _en_block(sd, "EnBlock3_1", h)
    s3 = _en_block(sd, "EnBlock3_2", h)
    h = _conv(sd, "EnDown3.conv", s3, stride=2)
    for i in (1, 2, 3, 4):
        h = _en_block(sd, "EnBlock4_%d" % i, h)
    h = _de_block(sd, "DeBlock4", _de_up_cat(sd, "DeUpCat4", h, s3), training)
    h = _de_block(sd, "DeBlock3", _de_up_cat(sd, "DeUpCat3", h, s2), training)
    h = _de_block(sd, "DeBlock2", _de_up_cat(sd, "DeUpCat2", h, s1), training)
    return _conv(sd, "endconv", h, pad=0)


def unetdrop_masks(x_shape, training: bool = True, base: int = 16):
    """The five dropout masks of one ``Unetdrop.forward`` call, drawn from torch's global generator with the shapes
    and in the order the reference draws its noise (UNet3DZoo.py:30 ``F.dropout3d(y, 0.2)`` - training=True ALWAYS,
    also in eval mode - then nn.Dropout(0.5) at :352, :358, :362, :366, which are identity in eval mode).
    ATen's dropout is ``input * noise`` with noise = bernoulli(1 - p) / (1 - p) (feature dropout: noise of shape
    (N, C, 1, 1, 1)), so ``dropout(ones)`` IS the noise tensor the reference multiplies by after the same seeding."""
    n, _, d, h, w = x_shape
    m = {"init": F.dropout3d(torch.ones((n, base, d, h, w)), 0.2)[:, :, :1, :1, :1].contiguous()}
    shapes = {"d1": (n, base * 4, d // 4, h // 4, w // 4), "d2": (n, base * 8, d // 8, h // 8, w // 8),
              "u1": (n, base * 4, d // 4, h // 4, w // 4), "u2": (n, base * 2, d // 2, h // 2, w // 2)}
    for k in ("d1", "d2", "u1", "u2"):
        m[k] = F.dropout(torch.ones(shapes[k]), 0.5, training) if training else None
    return m


def unetdrop3d_forward(sd: StateDict, x: Tensor, training: bool = True, masks=None) -> Tensor:
    """Unetdrop.forward, UNet3DZoo.py:340-372, with the dropout noise passed in (``unetdrop_masks``): channel
    dropout p = 0.2 after InitConv (:30,:312), element dropout p = 0.5 after EnBlock3_2 (:352), EnBlock4_3 (:358),
    DeUpCat4 (:362) and DeUpCat3 (:366).  Note :350-361: the skip into DeUpCat4 is the output of EnBlock3_1, not of
    EnBlock3_2."""
    if masks is None:
        masks = unetdrop_masks(x.shape, training, sd["InitConv.conv.weight"].shape[0])
    mul = lambda t, k: t if masks.get(k) is None else t * masks[k]
    h = mul(_conv(sd, "InitConv.conv", x), "init")
    s1 = _en_block(sd, "EnBlock1", h)
    h = _conv(sd, "EnDown1.conv", s1, stride=2)
    h = _en_block(sd, "EnBlock2_1", h)
    s2 = _en_block(sd, "EnBlock2_2", h)
    h = _conv(sd, "EnDown2.conv", s2, stride=2)
    s3 = _en_block(sd, "EnBlock3_1", h)
    h = mul(_en_block(sd, "EnBlock3_2", s3), "d1")
    h = _conv(sd, "EnDown3.conv", h, stride=2)
    for i in (1, 2, 3):
        h = _en_block(sd, "EnBlock4_%d" % i, h)
    h = _en_block(sd, "EnBlock4_4", mul(h, "d2"))
    h = _de_block(sd, "DeBlock4", mul(_de_up_cat(sd, "DeUpCat4", h, s3), "u1"), training)
    h = _de_block(sd, "DeBlock3", mul(_de_up_cat(sd, "DeUpCat3", h, s2), "u2"), training)
    h = _de_block(sd, "DeBlock2", _de_up_cat(sd, "DeUpCat2", h, s1), training)
    return _conv(sd, "endconv", h, pad=0)


# ----------------------------------------------------------------------------
# 3D attention U-Net  (models/lib/UNet3DZoo.py:69-115,132-143,167-241)   SURVEY.md §8(f)-1
# ----------------------------------------------------------------------------

def _bn(sd: StateDict, key: str, x: Tensor, training: bool) -> Tensor:
    # nn.BatchNorm3d without activation (Attention_block, UNet3DZoo.py:72-86)
    if training:
        sd[key + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                        sd[key + ".weight"], sd[key + ".bias"], training, 0.1, 1e-5)


def _attention_block(sd: StateDict, name: str, g: Tensor, x: Tensor, training: bool) -> Tensor:
    # Attention_block.forward, UNet3DZoo.py:90-100
    g1 = _bn(sd, name + ".W_g.1", _conv(sd, name + ".W_g.0", g, pad=0), training)
    x1 = _bn(sd, name + ".W_x.1", _conv(sd, name + ".W_x.0", x, pad=0), training)
    psi = F.relu(g1 + x1)
    psi = torch.sigmoid(_bn(sd, name + ".psi.1", _conv(sd, name + ".psi.0", psi, pad=0), training))
    return x * psi


def _de_up(sd: StateDict, name: str, x: Tensor) -> Tensor:
    # DeUp.forward, UNet3DZoo.py:139-143: 1x1 -> ConvT k2 s2 -> 1x1
    h = _conv(sd, name + ".conv1", x, pad=0)
    h = F.conv_transpose3d(h, sd[name + ".conv2.weight"], sd[name + ".conv2.bias"], stride=2)
    return _conv(sd, name + ".conv3", h, pad=0)


def _de_cat(sd: StateDict, name: str, x: Tensor, prev: Tensor) -> Tensor:
    # De_Cat.forward, UNet3DZoo.py:109-115: cat((x, prev)) -> 1x1
    return _conv(sd, name + ".conv1", torch.cat((x, prev), dim=1), pad=0)


def attunet3d_forward(sd: StateDict, x: Tensor, training: bool = True) -> Tensor:
    """AttUnet.forward, UNet3DZoo.py:206-241.  Note the argument order of the three De_Cat calls: level 4
    concatenates (gated skip, upsampled), levels 3 and 2 (upsampled, gated skip) (:227, :232, :237)."""
    h = _conv(sd, "InitConv.conv", x)
    s1 = _en_block(sd, "EnBlock1", h)
    h = _conv(sd, "EnDown1.conv", s1, stride=2)
    h = _en_block(sd, "EnBlock2_1", h)
    s2 = _en_block(sd, "EnBlock2_2", h)
    h = _conv(sd, "EnDown2.conv", s2, stride=2)
    h = _en_block(sd, "EnBlock3_1", h)
    s3 = _en_block(sd, "EnBlock3_2", h)
    h = _conv(sd, "EnDown3.conv", s3, stride=2)
    for i in (1, 2, 3, 4):
        h = _en_block(sd, "EnBlock4_%d" % i, h)
    y4 = _de_up(sd, "DeUp4", h)
    a3 = _attention_block(sd, "Att4", y4, s3, training)
    y4 = _de_block(sd, "DeBlock4", _de_cat(sd, "DeUpCat4", a3, y4), training)
    y3 = _de_up(sd, "DeUp3", y4)
    a2 = _attention_block(sd, "Att3", y3, s2, training)
    y3 = _de_block(sd, "DeBlock3", _de_cat(sd, "DeUpCat3", y3, a2), training)
    y2 = _de_up(sd, "DeUp2", y3)
    a1 = _attention_block(sd, "Att2", y2, s1, training)
    y2 = _de_block(sd, "DeBlock2", _de_cat(sd, "DeUpCat2", y2, a1), training)
    return _conv(sd, "endconv", y2, pad=0)


# ----------------------------------------------------------------------------
# 3D V-Net  (models/lib/VNet3D.py:5-31,66-117,145-235; normalization='gn', has_dropout=False as
# TrustworthySeg.py:42-44,67-69 builds it)   SURVEY.md §8(f)-4
# ----------------------------------------------------------------------------

def _gn16_relu(sd: StateDict, key: str, x: Tensor) -> Tensor:
    # nn.GroupNorm(num_groups=16, C) then ReLU: VNet3D.py:18-24
    return F.relu(F.group_norm(x, 16, sd[key + ".weight"], sd[key + ".bias"], 1e-5))


def _v_conv_block(sd: StateDict, name: str, x: Tensor, n_stages: int) -> Tensor:
    # ConvBlock, VNet3D.py:5-31: (Conv3d k3 p1 -> GroupNorm(16) -> ReLU) x n_stages; Sequential indices 3i, 3i+1
    for i in range(n_stages):
        x = _gn16_relu(sd, "%s.conv.%d" % (name, 3 * i + 1), _conv(sd, "%s.conv.%d" % (name, 3 * i), x))
    return x


def _v_down(sd: StateDict, name: str, x: Tensor) -> Tensor:
    # DownsamplingConvBlock, VNet3D.py:66-89: Conv3d k2 s2 p0 -> GroupNorm(16) -> ReLU
    return _gn16_relu(sd, name + ".conv.1", _conv(sd, name + ".conv.0", x, stride=2, pad=0))


def _v_up(sd: StateDict, name: str, x: Tensor) -> Tensor:
    # UpsamplingDeconvBlock, VNet3D.py:92-117: ConvTranspose3d k2 s2 -> GroupNorm(16) -> ReLU
    h = F.conv_transpose3d(x, sd[name + ".conv.0.weight"], sd[name + ".conv.0.bias"], stride=2)
    return _gn16_relu(sd, name + ".conv.1", h)


def vnet3d_forward(sd: StateDict, x: Tensor) -> Tensor:
    """VNet.forward with has_dropout=False, VNet3D.py:180-235 (GroupNorm only: no train / eval difference)."""
    x1 = _v_conv_block(sd, "block_one", x, 1)
    x2 = _v_conv_block(sd, "block_two", _v_down(sd, "block_one_dw", x1), 2)
    x3 = _v_conv_block(sd, "block_three", _v_down(sd, "block_two_dw", x2), 3)
    x4 = _v_conv_block(sd, "block_four", _v_down(sd, "block_three_dw", x3), 3)
    x5 = _v_conv_block(sd, "block_five", _v_up(sd, "block_four_up", x4) + x3, 3)
    x6 = _v_conv_block(sd, "block_six", _v_up(sd, "block_five_up", x5) + x2, 3)
    x7 = _v_conv_block(sd, "block_seven", _v_up(sd, "block_six_up", x6) + x1, 1)
    return _conv(sd, "out_conv", x7, pad=0)


def _v_up2d(sd: StateDict, name: str, x: Tensor) -> Tensor:
    # UpsamplingDeconvBlock, VNet2D.py:94-118: ConvTranspose2d k2 s2 -> GroupNorm(16) -> ReLU
    h = F.conv_transpose2d(x, sd[name + ".conv.0.weight"], sd[name + ".conv.0.bias"], stride=2)
    return _gn16_relu(sd, name + ".conv.1", h)


def _offset_add(inputs: Tensor, down_outputs: Tensor, rand_log: Optional[List[Tuple[int, ...]]]) -> Tensor:
    # VNet2D.offsetAdd, VNet2D.py:234-255: pad the WIDTH of the second argument with torch.rand drawn from the
    # global CPU generator (shape (N,C,H) unsqueezed for an offset of 1, (N,C,H,offset) beyond), then add.
    outputs = down_outputs
    off = abs(inputs.size(3) - outputs.size(3))
    if off == 1:
        shape = (outputs.size(0), outputs.size(1), outputs.size(2))
        add = torch.rand(shape).unsqueeze(3)
    elif off > 1:
        shape = (outputs.size(0), outputs.size(1), outputs.size(2), off)
        add = torch.rand(shape)
    if off >= 1:
        if rand_log is not None:
            rand_log.append(tuple(shape))
        outputs = torch.cat([outputs, add.to(outputs.device)], dim=3)
    return inputs + outputs


def vnet2d_forward(sd: StateDict, x: Tensor, rand_log: Optional[List[Tuple[int, ...]]] = None) -> Tensor:
    """VNet2D.forward with has_dropout=False, VNet2D.py:182-265 (n_filters=32, GroupNorm(16): TrustworthySeg.py:54,
    78, 91, 104).  ConvBlock / DownsamplingConvBlock are the 2-D forms of the V-Net's (``_conv`` picks conv2d)."""
    x1 = _v_conv_block(sd, "block_one", x, 1)
    x2 = _v_conv_block(sd, "block_two", _v_down(sd, "block_one_dw", x1), 2)
    x3 = _v_conv_block(sd, "block_three", _v_down(sd, "block_two_dw", x2), 3)
    x4 = _v_conv_block(sd, "block_four", _v_down(sd, "block_three_dw", x3), 3)
    x5 = _v_conv_block(sd, "block_five", _v_down(sd, "block_four_dw", x4), 3)
    x5_up = _offset_add(x4, _v_up2d(sd, "block_five_up", x5), rand_log)
    x6 = _v_conv_block(sd, "block_six", x5_up, 3)
    x6_up = _offset_add(x3, _v_up2d(sd, "block_six_up", x6), rand_log)
    x7 = _v_conv_block(sd, "block_seven", x6_up, 3)
    x7_up = _offset_add(_v_up2d(sd, "block_seven_up", x7), x2, rand_log)
    x8 = _v_conv_block(sd, "block_eight", x7_up, 2)
    x8_up = _offset_add(_v_up2d(sd, "block_eight_up", x8), x1, rand_log)
    x9 = _v_conv_block(sd, "block_nine", x8_up, 2)
    return _conv(sd, "out_conv", x9, pad=0)


# ----------------------------------------------------------------------------
# 2D U-Net  (models/lib/UNet2DZoo.py:107-259)
# ----------------------------------------------------------------------------

def _double_conv(sd: StateDict, name: str, x: Tensor, training: bool) -> Tensor:
    # DoubleConv, UNet2DZoo.py:8-21: Sequential indices 0,1,(2),3,4,(5)
    h = _bn_relu(sd, name + ".conv.1", _conv(sd, name + ".conv.0", x), training)
    return _bn_relu(sd, name + ".conv.4", _conv(sd, name + ".conv.3", h), training)


def _offset_cat_w(skip: Tensor, up: Tensor, rand_log: Optional[List[Tuple[int, ...]]]) -> Tensor:
    # Unet2D.offsetCat, UNet2DZoo.py:162-182: pad *width* with torch.rand drawn
    # from the global CPU generator; (skip, up) channel order.
    off = abs(skip.size(3) - up.size(3))
    if off == 1:
        shape = (up.size(0), up.size(1), up.size(2))
        add = torch.rand(shape).unsqueeze(3)
    elif off > 1:
        shape = (up.size(0), up.size(1), up.size(2), off)
        add = torch.rand(shape)
    if off >= 1:
        if rand_log is not None:
            rand_log.append(tuple(shape))
        up = torch.cat([up, add.to(up.device)], dim=3)
    return torch.cat([skip, up], dim=1)


def _offset_cat_drive(skip: Tensor, up: Tensor, rand_log: Optional[List[Tuple[int, ...]]]) -> Tensor:
    # Unet2DDRIVE.offsetCat, UNet2DZoo.py:239-259: pads H if H differs, else W.
    if skip.shape[2] != up.shape[2]:
        shape = (up.size(0), up.size(1), abs(skip.size(2) - up.size(2)), up.size(3))
        dim = 2
    elif skip.shape[3] != up.shape[3]:
        shape = (up.size(0), up.size(1), up.size(2), abs(skip.size(3) - up.size(3)))
        dim = 3
    else:
        return torch.cat([skip, up], dim=1)
    if rand_log is not None:
        rand_log.append(tuple(shape))
    up = torch.cat([up, torch.rand(shape).to(up.device)], dim=dim)
    return torch.cat([skip, up], dim=1)


def unet2d_forward(sd: StateDict, x: Tensor, training: bool = True, drive: bool = False,
                   rand_log: Optional[List[Tuple[int, ...]]] = None, taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """Unet2D.forward (UNet2DZoo.py:130-160) / Unet2DDRIVE.forward (:207-237).

    Returns the raw logits ``c10`` (the ``nn.Softmax()`` at :159 is dead code).  ``taps``: a dict that receives the
    output of every DoubleConv / ConvTranspose (the per-layer audit of tests/test_gpu_unet2d.py reads it).
    """
    ocat = _offset_cat_drive if drive else _offset_cat_w

    def dc(name: str, t: Tensor) -> Tensor:
        y = _double_conv(sd, name, t, training)
        if taps is not None:
            taps[name] = y
        return y

    c1 = dc("conv1", x)
    c2 = dc("conv2", F.max_pool2d(c1, 2))
    c3 = dc("conv3", F.max_pool2d(c2, 2))
    c4 = dc("conv4", F.max_pool2d(c3, 2))
    c5 = dc("conv5", F.max_pool2d(c4, 2))

    def up(name: str, t: Tensor) -> Tensor:
        y = F.conv_transpose2d(t, sd[name + ".weight"], sd[name + ".bias"], stride=2)
        if taps is not None:
            taps[name] = y
        return y

    c6 = dc("conv6", ocat(c4, up("up6", c5), rand_log))
    c7 = dc("conv7", ocat(c3, up("up7", c6), rand_log))
    c8 = dc("conv8", torch.cat([up("up8", c7), c2], dim=1))
    u9 = up("up9", c8)
    m9 = ocat(c1, u9, rand_log) if drive else torch.cat([u9, c1], dim=1)
    c9 = dc("conv9", m9)
    return _conv(sd, "conv10", c9, pad=0)


# ---- the same network with a bf16 pipeline's roundings put in (NOT the reference's arithmetic: a measuring stick) ----
# The reference computes in fp32.  A bf16 tensor-core pipeline must round (i) every weight, (ii) every stored
# activation and (iii) every matrix operand to bf16; everything between two roundings stays fp32 (accumulators, bias,
# BatchNorm statistics and affine).  `unet2d_forward_bf16_points` restates Unet2D with exactly those roundings at the
# points where devis_amd's bf16 path has them: it separates what the ROUNDING POINTS cost (this function
# against unet2d_forward - a property of bf16 and of the network's conditioning, the same for any bf16 implementation)
# from what the KERNELS add on top (devis_amd against this function).

def _q(t: Tensor) -> Tensor:
    return t.bfloat16().float()


def _bn_affine(sd: StateDict, key: str, h: Tensor, training: bool) -> Tuple[Tensor, Tensor]:
    # BatchNorm as a per-channel (scale, shift) from the statistics of the STORED tensor h (UNet2DZoo.py:12,15)
    dims = (0, 2, 3)
    if training:
        mean = h.double().mean(dims)
        var = h.double().var(dims, unbiased=False)
    else:
        mean, var = sd[key + ".running_mean"].double(), sd[key + ".running_var"].double()
    scale = sd[key + ".weight"].double() / torch.sqrt(var + 1e-5)
    shift = sd[key + ".bias"].double() - mean * scale
    return scale.float().view(1, -1, 1, 1), shift.float().view(1, -1, 1, 1)


def unet2d_forward_bf16_points(sd: StateDict, x: Tensor, training: bool = True, drive: bool = False,
                               taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """Unet2D / Unet2DDRIVE (UNet2DZoo.py:130-160, 207-237) evaluated in fp32 with bf16 roundings of the weights, of
    every stored tensor (conv / ConvTranspose outputs, BatchNorm+ReLU outputs, logits) and of every conv operand.
    Does not touch the running statistics in ``sd``.  offsetCat's torch.rand padding is drawn like the reference's."""
    ocat = _offset_cat_drive if drive else _offset_cat_w

    def conv(key: str, t: Tensor, pad: int = 1) -> Tensor:
        return _q(F.conv2d(t, _q(sd[key + ".weight"]), sd[key + ".bias"], padding=pad))

    def dc(name: str, t: Tensor) -> Tensor:
        h1 = conv(name + ".conv.0", t)
        sc, sh = _bn_affine(sd, name + ".conv.1", h1, training)
        a = _q(F.relu(torch.addcmul(sh, h1, sc)))                  # the second conv's operand
        h2 = conv(name + ".conv.3", a)
        sc, sh = _bn_affine(sd, name + ".conv.4", h2, training)
        y = _q(F.relu(torch.addcmul(sh, h2, sc)))
        if taps is not None:
            taps[name] = y
        return y

    def up(name: str, t: Tensor) -> Tensor:
        y = _q(F.conv_transpose2d(t, _q(sd[name + ".weight"]), sd[name + ".bias"], stride=2))
        if taps is not None:
            taps[name] = y
        return y

    def cat_q(skip: Tensor, u: Tensor) -> Tensor:
        return _q(ocat(skip, u, None))                             # rounds the torch.rand padding; the rest is bf16 already

    c1 = dc("conv1", _q(x))
    c2 = dc("conv2", F.max_pool2d(c1, 2))
    c3 = dc("conv3", F.max_pool2d(c2, 2))
    c4 = dc("conv4", F.max_pool2d(c3, 2))
    c5 = dc("conv5", F.max_pool2d(c4, 2))
    c6 = dc("conv6", cat_q(c4, up("up6", c5)))
    c7 = dc("conv7", cat_q(c3, up("up7", c6)))
    c8 = dc("conv8", torch.cat([up("up8", c7), c2], dim=1))
    u9 = up("up9", c8)
    c9 = dc("conv9", cat_q(c1, u9) if drive else torch.cat([u9, c1], dim=1))
    return conv("conv10", c9, pad=0)


# ---- the 3-D U-Net with a bf16 pipeline's roundings put in (a measuring stick like unet2d_forward_bf16_points) ----

def _qs(t: Tensor) -> Tensor:
    """bf16 rounding with a straight-through gradient: autograd then differentiates the fp32 network at the ROUNDED
    forward values (its ReLU gates are those of a bf16 forward), while the backward arithmetic itself stays fp32."""
    return t + (t.bfloat16().float() - t).detach()


def unet3d_forward_bf16_points(sd: StateDict, x: Tensor, training: bool = True, fused_upcat: bool = True) -> Tensor:
    """Unet.forward (UNet3DZoo.py:276-305) in fp32 with bf16 roundings where devis_amd's bf16 path has them: weights,
    every stored tensor (conv outputs after bias and residual, DeBlock outputs, logits), every conv operand
    (norm + ReLU outputs).  Norm statistics come from the stored tensors; with `fused_upcat` DeUpCat3 / DeUpCat2 are
    evaluated as the one-pass form y = W3a prev + (W3b Wt[p]) t + (b3 + W3b bt) with the composite matrices rounded
    (csrc/upcat.hip), DeUpCat4 layer by layer.  Differentiable (straight-through roundings); running statistics in
    ``sd`` are not touched."""
    def conv(key: str, t: Tensor, stride: int = 1, pad: int = 1, res: Optional[Tensor] = None) -> Tensor:
        y = F.conv3d(t, _qs(sd[key + ".weight"]), sd[key + ".bias"], stride=stride, padding=pad)
        return _qs(y if res is None else y + res)

    def gn_op(key: str, h: Tensor) -> Tensor:            # GroupNorm(8) + ReLU as the next conv's operand
        return _qs(F.relu(F.group_norm(h, 8, sd[key + ".weight"], sd[key + ".bias"], 1e-5)))

    def bn(key: str, h: Tensor) -> Tensor:
        if training:
            return F.batch_norm(h, None, None, sd[key + ".weight"], sd[key + ".bias"], True, 0.1, 1e-5)
        return F.batch_norm(h, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"],
                            sd[key + ".bias"], False, 0.1, 1e-5)

    def en(name: str, t: Tensor) -> Tensor:
        h = conv(name + ".conv1", gn_op(name + ".bn1", t))
        return conv(name + ".conv2", gn_op(name + ".bn2", h), res=t)

    def de(name: str, t: Tensor) -> Tensor:
        h1 = conv(name + ".conv1", t)
        h2 = conv(name + ".conv2", _qs(F.relu(bn(name + ".bn1", h1))))
        return _qs(F.relu(bn(name + ".bn2", h2)) + t)

    def upcat(name: str, t: Tensor, prev: Tensor) -> Tensor:
        c = prev.shape[1]
        t = conv(name + ".conv1", t, pad=0)
        w3, b3 = sd[name + ".conv3.weight"], sd[name + ".conv3.bias"]
        wt, bt = sd[name + ".conv2.weight"], sd[name + ".conv2.bias"]
        if fused_upcat and c in (16, 32):
            w3a, w3b = w3[:, :c, 0, 0, 0], w3[:, c:, 0, 0, 0]
            m = _qs(torch.einsum("om,cmdhw->codhw", w3b, wt))              # composite ConvT weight [c_in][o][2][2][2]
            bias = b3 + w3b @ bt
            y = F.conv3d(prev, _qs(w3a)[:, :, None, None, None]) + F.conv_transpose3d(t, m, bias, stride=2)
            return _qs(y)
        up = _qs(F.conv_transpose3d(t, _qs(wt), bt, stride=2))
        return _qs(F.conv3d(torch.cat((prev, up), dim=1), _qs(w3), b3))

    h = conv("InitConv.conv", _qs(x))
    s1 = en("EnBlock1", h)
    h = en("EnBlock2_1", conv("EnDown1.conv", s1, stride=2))
    s2 = en("EnBlock2_2", h)
    h = en("EnBlock3_1", conv("EnDown2.conv", s2, stride=2))
    s3 = en("EnBlock3_2", h)
    h = conv("EnDown3.conv", s3, stride=2)
    for i in (1, 2, 3, 4):
        h = en("EnBlock4_%d" % i, h)
    h = de("DeBlock4", upcat("DeUpCat4", h, s3))
    h = de("DeBlock3", upcat("DeUpCat3", h, s2))
    h = de("DeBlock2", upcat("DeUpCat2", h, s1))
    return conv("endconv", h, pad=0)


# ----------------------------------------------------------------------------
# Evidential criteria  (models/criterions.py)
# ----------------------------------------------------------------------------

def kl_dirichlet(alpha: Tensor, c: int) -> Tensor:
    """KL(Dir(alpha) || Dir(1)) per row, criterions.py:117-127 (without the hard .cuda())."""
    s_alpha = torch.sum(alpha, dim=1, keepdim=True)
    beta = torch.ones((1, c), device=alpha.device)
    s_beta = torch.sum(beta, dim=1, keepdim=True)
    ln_b = torch.lgamma(s_alpha) - torch.sum(torch.lgamma(alpha), dim=1, keepdim=True)
    ln_b_uni = torch.sum(torch.lgamma(beta), dim=1, keepdim=True) - torch.lgamma(s_beta)
    dg0 = torch.digamma(s_alpha)
    dg1 = torch.digamma(alpha)
    return torch.sum((alpha - beta) * (dg1 - dg0), dim=1, keepdim=True) + ln_b + ln_b_uni


def soft_label(labels: Tensor, num_class: int) -> Tensor:
    """get_soft_label, criterions.py:353-370: (N,1,...) int -> (N,...,K) float one-hot."""
    t = labels.permute(0, 2, 3, 4, 1) if labels.ndim == 5 else labels.permute(0, 2, 3, 1)
    planes = [torch.eq(t, i * torch.ones_like(t)) for i in range(num_class)]
    return torch.cat(planes, dim=-1).float()


def sdice_loss(preds: Tensor, targets: Tensor) -> Tensor:
    """SDiceLoss.forward (weight_map=None), criterions.py:226-257."""
    k = preds.size(1)
    p = preds.permute(0, 2, 3, 4, 1) if preds.ndim == 5 else preds.permute(0, 2, 3, 1)
    pred = p.contiguous().view(-1, k)
    ground = targets.view(-1, k)
    ref_vol = torch.sum(ground, 0)
    intersect = torch.sum(ground * pred, 0)
    seg_vol = torch.sum(pred, 0)
    dice = (2.0 * intersect + 1e-5) / (ref_vol + seg_vol + 1.0 + 1e-5)
    return torch.mean(-torch.log(dice))


def tversky_dice_loss(preds: Tensor, targets: Tensor) -> Tensor:
    """DiceLoss.forward (weight=False, reduce, size_average), criterions.py:166-215."""
    c = preds.size(1)
    if preds.ndim == 5:
        preds = preds.permute(0, 2, 3, 4, 1).contiguous().view(-1, c)
    else:
        preds = preds.permute(0, 2, 3, 1).contiguous().view(-1, c)
    targets = targets.view(-1, 1)
    prob = torch.exp(F.log_softmax(preds, dim=1))
    smooth = torch.zeros(c, dtype=torch.float32).fill_(0.00001).to(preds.device)
    mask = torch.zeros(preds.shape).to(preds.device) + 1e-8
    mask.scatter_(1, targets, 1.)
    ones = torch.ones(preds.shape).to(preds.device)
    tp = prob * mask
    fp = prob * (ones - mask)
    fn = (ones - prob) * mask
    a = fp.sum(dim=0) / ((fp.sum(dim=0) + fn.sum(dim=0)) + smooth)
    a = torch.clamp(a, min=0.2, max=0.8)
    b = 1 - a
    num = torch.sum(tp, dim=0).float()
    den = num + a * torch.sum(fp, dim=0).float() + b * torch.sum(fn, dim=0).float()
    dice = num / (den + smooth)
    return (1 - dice).sum() / c


def _rows(alpha: Tensor) -> Tensor:
    # [N,C,...] -> [N*V, C]; criterions.py:436-438
    a = alpha.view(alpha.size(0), alpha.size(1), -1).transpose(1, 2)
    return a.contiguous().view(-1, a.size(2))


def dce_evidence_loss(p, alpha, c, current_step, lamda_step, total_step, eps,
                      disentangle, evidence, backbone_pred) -> Tensor:
    """criterions.py:421-456: returns (NV,1) = L_ace + L_dice + lambda*KL."""
    sp = p.unsqueeze(1) if alpha.ndim == 5 else p
    l_dice = sdice_loss(evidence, soft_label(sp, c))
    alpha = _rows(alpha)
    s = torch.sum(alpha, dim=1, keepdim=True)
    e = alpha - 1
    label = F.one_hot(p, num_classes=c).view(-1, c)
    l_ace = torch.sum(label * (torch.digamma(s) - torch.digamma(alpha)), dim=1, keepdim=True)
    coef = min(1, current_step / lamda_step)
    alp = e * (1 - label) + 1
    l_kl = coef * kl_dirichlet(alp, c)
    return l_ace + l_dice + l_kl


def dce_evidence_u_loss(p, alpha, c, current_step, lamda_step, total_step, eps,
                        disentangle, evidence, backbone_pred) -> Tensor:
    """criterions.py:373-419 (Uncertainty_Loss=True variant)."""
    sp = p.unsqueeze(1) if alpha.ndim == 5 else p
    l_dice = tversky_dice_loss(evidence, sp)
    alpha = _rows(alpha)
    s = torch.sum(alpha, dim=1, keepdim=True)
    e = alpha - 1
    label = F.one_hot(p, num_classes=c).view(-1, c)
    l_ace = torch.sum(label * (torch.digamma(s) - torch.digamma(alpha)), dim=1, keepdim=True)
    coef = min(1, current_step / lamda_step)
    a0 = torch.tensor(0.01, dtype=torch.float32)
    a_au = a0 * torch.exp(-torch.log(a0) / total_step * current_step)
    alp = e * (1 - label) + 1
    l_kl = coef * kl_dirichlet(alp, c)
    score, cls = torch.max(alpha / s, 1, keepdim=True)
    u = c / s
    match = torch.reshape(torch.eq(cls, p.view(-1, 1)).float(), (-1, 1))
    if disentangle:
        acc_unc = -torch.log(score * (1 - u) + eps)
        inacc_cert = -torch.log((1 - score) * u + eps)
    else:
        acc_unc = -score * torch.log(1 - u + eps)
        inacc_cert = -(1 - score) * torch.log(u + eps)
    l_au = a_au * match * acc_unc + (1 - a_au) * (1 - match) * inacc_cert
    return l_ace + l_kl + (1 - a_au) * l_dice + l_au


def ce_loss(p, alpha, c, global_step, annealing_step) -> Tensor:
    """criterions.py:130-142 (5-D NCDHW alpha; one-hot permuted to channel dim)."""
    s = torch.sum(alpha, dim=1, keepdim=True)
    e = alpha - 1
    label = F.one_hot(p, num_classes=c).permute(0, 4, 1, 2, 3)
    a = torch.sum(label * (torch.digamma(s) - torch.digamma(alpha)), dim=1, keepdim=True)
    coef = min(1, global_step / annealing_step)
    alp = e * (1 - label) + 1
    return a + coef * kl_dirichlet(alp, c)


def mse_loss(p, alpha, c, global_step, annealing_step=1) -> Tensor:
    """criterions.py:144-154 (row layout: alpha (NV,K), p (NV,))."""
    s = torch.sum(alpha, dim=1, keepdim=True)
    e = alpha - 1
    m = alpha / s
    label = F.one_hot(p, num_classes=c)
    a = torch.sum((label - m) ** 2, dim=1, keepdim=True)
    b = torch.sum(alpha * (s - alpha) / (s * s * (s + 1)), dim=1, keepdim=True)
    coef = min(1, global_step / annealing_step)
    alp = e * (1 - label) + 1
    return (a + b) + coef * kl_dirichlet(alp, c)


# ----------------------------------------------------------------------------
# TMSU wrapper + sliding window  (models/TrustworthySeg.py:124-166, predict.py:159-191)
# ----------------------------------------------------------------------------

def tailor_and_concat(x: Tensor, model: Callable[[Tensor], Tensor]) -> Tensor:
    """predict.py:159-191: eight 128^3 crops, eight disjoint writes, D cut to 155."""
    lo, hi = slice(0, 128), slice(112, 240)
    crops = [(lo, lo, slice(0, 128)), (lo, hi, slice(0, 128)), (hi, lo, slice(0, 128)),
             (hi, hi, slice(0, 128)), (lo, lo, slice(27, 155)), (lo, hi, slice(27, 155)),
             (hi, lo, slice(27, 155)), (hi, hi, slice(27, 155))]
    if x.shape[1] == 1:
        y = torch.cat((x.clone(), x.clone(), x.clone(), x.clone()), 1)
    elif x.shape[1] == 4:
        y = x.clone()
    else:
        y = torch.cat((x.clone(), x.clone()), 1)
    outs = [model(x[..., h, w, d]) for (h, w, d) in crops]
    dst_lo, dst_hi = slice(0, 128), slice(128, 240)
    src_lo, src_hi = slice(0, 128), slice(16, 128)
    for i, t in enumerate(outs):
        hh, ww, dd = (i >> 1) & 1, i & 1, (i >> 2) & 1
        dst = (dst_hi if hh else dst_lo, dst_hi if ww else dst_lo,
               slice(128, 155) if dd else slice(0, 128))
        src = (src_hi if hh else src_lo, src_hi if ww else src_lo,
               slice(96, 123) if dd else slice(0, 128))
        y[(Ellipsis,) + dst] = t[(Ellipsis,) + src]
    return y[..., :155]


def tmsu_forward(backbone: Callable[[Tensor], Tensor], x: Tensor, y: Optional[Tensor],
                 global_step, mode: str, dataset: str, classes: int, lambda_epochs: int,
                 total_epochs: int, u_loss: bool = False, eps: float = 1e-10):
    """TMSU.forward, TrustworthySeg.py:124-158.  ``total_epochs`` = end_epochs + 1 (:121)."""
    if mode != 'train' and dataset == 'BraTS':
        out = tailor_and_concat(x, backbone)
    else:
        out = backbone(x)
    evidence = F.softplus(out)
    pred = F.softmax(out, 1)
    alpha = evidence + 1
    if mode != 'train':
        return evidence
    fn = dce_evidence_u_loss if u_loss else dce_evidence_loss
    loss = fn(y.to(torch.int64), alpha, classes, global_step, lambda_epochs, total_epochs,
              eps, False, evidence, pred)
    return evidence, torch.mean(loss)


def infer_head(evidence: Tensor, classes: int):
    """val/test tensor math: trainUMIS.py:286-290, :446-451; predict.py:92-105."""
    alpha = evidence + 1
    s = torch.sum(alpha, dim=1, keepdim=True)
    prob = alpha / s
    _, predicted = torch.max(prob, 1)
    u = classes / s
    logp = torch.log(prob)
    mne = torch.sum((-prob * logp / math.log(classes))[:, 1:], dim=1)
    return predicted, u, prob, mne


UNCERTAINTY_THRESHOLDS = [0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95]  # trainUMIS.py:521


def poly_lr(init_lr: float, epoch: int, max_epoch: int, power: float = 0.9) -> float:
    """adjust_learning_rate, trainUMIS.py:217-219."""
    import numpy as np
    return round(init_lr * np.power(1 - epoch / max_epoch, power), 8)


# ---------------------------------------------------------------- evaluation counts
def uncertainty_counts(prediction, target, thresholded_uncertainty):
    """numpyfunctions.py:86-107 (mask=None): boolean arrays -> tp, tn, fp, fn, tpu, tnu, fpu, fnu."""
    import numpy as np
    tps = np.logical_and(target, prediction)
    tns = np.logical_and(~target, ~prediction)
    fps = np.logical_and(~target, prediction)
    fns = np.logical_and(target, ~prediction)
    tpu = np.logical_and(tps, thresholded_uncertainty).sum()
    tnu = np.logical_and(tns, thresholded_uncertainty).sum()
    fpu = np.logical_and(fps, thresholded_uncertainty).sum()
    fnu = np.logical_and(fns, thresholded_uncertainty).sum()
    return tps.sum(), tns.sum(), fps.sum(), fns.sum(), tpu, tnu, fpu, fnu


def dice_score(o, t, eps=1e-8):
    """predict.py:386-389."""
    num = 2 * (o * t).sum() + eps
    den = o.sum() + t.sum() + eps
    return num / den


def mIOU(o, t, eps=1e-8):
    """predict.py:296-299."""
    num = (o * t).sum() + eps
    den = (o | t).sum() + eps
    return num / den


def softmax_output_dice(output, target):
    """predict.py:401-416: BraTS whole / core / active."""
    return [dice_score(output > 0, target > 0),
            dice_score((output == 1) | (output == 3), (target == 1) | (target == 3)),
            dice_score(output == 3, target == 3)]


def softmax_output_litsdice(output, target):
    """predict.py:390-399: LiTS liver / tumor."""
    return [dice_score((output == 1) | (output == 2), (target == 1) | (target == 2)),
            dice_score(output == 2, target == 2)]


def eval_counts(pred, target, u, thresholds, classes: int):
    """What devis_eval_counts returns, from numpy arrays: conf[t, p], ucount[i] = uncertainty_counts(pred > 0,
    target > 0, u > thresholds[i]) (trainUMIS.py:521 thresholds), sums = (sum u, sum 1-u, sum exp(-u))."""
    import numpy as np
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    target = np.asarray(target).reshape(-1).astype(np.int64)
    conf = np.zeros((classes, classes), dtype=np.int64)
    np.add.at(conf, (target, pred), 1)
    u = np.asarray(u, dtype=np.float32).reshape(-1)
    ucount = np.array([uncertainty_counts(pred > 0, target > 0, u > np.float32(t)) for t in thresholds], dtype=np.int64)
    sums = np.array([u.astype(np.float64).sum(), (np.float32(1) - u).astype(np.float64).sum(),
                     np.exp(-u).astype(np.float64).sum()])
    return conf, ucount.reshape(len(thresholds), 8), sums
