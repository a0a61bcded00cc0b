"""3-D V-Net on MI355X: constructor, attribute names and ``state_dict`` keys of the reference ``VNet``
(models/lib/VNet3D.py:145-235) as TMSU builds it (``normalization='gn'``,
``has_dropout=False``: TrustworthySeg.py:42-44,67-69), executed by the HIP kernels.

Every block is (conv -> GroupNorm(16) -> ReLU) x n: the norm + ReLU of stage i runs in the prologue of stage
i+1's conv and its moments are gathered in stage i's conv epilogue, exactly as in the residual U-Net; only a
block's LAST activation is materialised (it feeds a k2-s2 conv, a transposed conv or an additive skip).
The k2-s2 down convolutions and the 3x3x3 convs with shapes outside the MFMA kernels run on the generic
direct kernels (correctness path); ConvTranspose k2-s2 runs on the gather-GEMM kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .ops import ConvGeom
from .unet3d import G1, G3, Unet, _Run

G2S2 = ConvGeom((2, 2, 2), (2, 2, 2), (0, 0, 0))
GROUPS = 16


def _norm(c):                                  # normalization == 'gn', VNet3D.py:18-19
    return nn.GroupNorm(num_groups=GROUPS, num_channels=c)


class _ConvBlock(nn.Module):                   # VNet3D.py:5-31 (parameter holder; Sequential indices as the reference)
    def __init__(self, n_stages, cin, cout):
        super().__init__()
        mods = []
        for i in range(n_stages):
            mods += [nn.Conv3d(cin if i == 0 else cout, cout, 3, padding=1), _norm(cout), nn.ReLU(inplace=True)]
        self.conv = nn.Sequential(*mods)
        self.n_stages = n_stages


class _Down(nn.Module):                        # VNet3D.py:66-89
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv3d(cin, cout, 2, padding=0, stride=2), _norm(cout), nn.ReLU(inplace=True))


class _Up(nn.Module):                          # VNet3D.py:92-117
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.ConvTranspose3d(cin, cout, 2, padding=0, stride=2), _norm(cout), nn.ReLU(inplace=True))


class _RunV(_Run):
    """Forward / backward schedule of the V-Net over channels-last tensors."""

    def _gn16(self, name, x, partials=None):
        return ops.gn_stats(x, self.P[name + ".weight"], self.P[name + ".bias"], groups=GROUPS, partials=partials)

    # ---- forward blocks
    def conv_block(self, name, x, n_stages, pro=None):
        """x: a materialised tensor (pro None) or a raw conv output whose GroupNorm + ReLU (scale, shift = pro[:2])
        is still to be applied.  -> the block's materialised output."""
        stages, h, p = [], x, pro
        for i in range(n_stages):
            cn, gn = "%s.conv.%d" % (name, 3 * i), "%s.conv.%d" % (name, 3 * i + 1)
            w = self._w(cn)
            raw, part = ops.conv_fwd(h, None, w, self.P[cn + ".bias"], G3, p[0] if p else None, p[1] if p else None,
                                     want_stats=True)
            st = self._gn16(gn, raw, part)
            stages.append((cn, gn, h, p, raw, st, w))
            h, p = raw, st
        out = ops.affine_act(h, p[0], p[1], relu=True)
        self._push("vblock", name, stages)
        return out

    def down(self, name, x):
        w = self._w(name + ".conv.0")
        raw = ops.conv_fwd(x, None, w, self.P[name + ".conv.0.bias"], G2S2)
        st = self._gn16(name + ".conv.1", raw)
        self._push("vdown", name, x, raw, st, w)
        return raw, st

    def up(self, name, x, skip):
        wt = self._w(name + ".conv.0")
        raw = ops.convT_fwd(x, wt, self.P[name + ".conv.0.bias"], (2, 2, 2))
        st = self._gn16(name + ".conv.1", raw)
        out = ops.affine_act(raw, st[0], st[1], residual=skip, relu=True)      # relu(gn(up)) + skip, VNet3D.py:208-218
        self._push("vup", name, x, raw, st, wt)
        return out

    def forward(self, x):
        x1 = self.conv_block("block_one", x, 1)
        r, st = self.down("block_one_dw", x1)
        x2 = self.conv_block("block_two", r, 2, st)
        r, st = self.down("block_two_dw", x2)
        x3 = self.conv_block("block_three", r, 3, st)
        r, st = self.down("block_three_dw", x3)
        x4 = self.conv_block("block_four", r, 3, st)
        x5 = self.conv_block("block_five", self.up("block_four_up", x4, x3), 3)
        x6 = self.conv_block("block_six", self.up("block_five_up", x5, x2), 3)
        x7 = self.conv_block("block_seven", self.up("block_six_up", x6, x1), 1)
        return self.conv("out_conv", x7, G1)

    # ---- backward blocks
    def _vblock_bwd(self, e, d, need_dx=True):
        """d = gradient w.r.t. the block's materialised output.  -> (gradient w.r.t. the block's input - the
        ACTIVATED input when it came with a prologue -, norm-backward moments for that prologue | None)."""
        _, name, stages = e
        g, P = self.grads, self.P
        pt = None
        for i in reversed(range(len(stages))):
            cn, gn, h_in, p_in, raw, st, w = stages[i]
            draw, g[gn + ".weight"], g[gn + ".bias"] = ops.norm_relu_bwd(
                d, raw, st[0], st[1], P[gn + ".weight"], st[2], st[3], GROUPS, False, out=self._o(gn), partials=pt)
            g[cn + ".weight"], g[cn + ".bias"] = self._wg(
                ops.conv_wgrad, h_in, None, draw, G3, P[cn + ".weight"].shape, p_in[0] if p_in else None,
                p_in[1] if p_in else None, out=self._o(cn))
            if p_in is not None:
                d, _, pt = ops.conv_dgrad(draw, w, G3, h_in.shape, wd=self._wd(cn), norm=(h_in, p_in[0], p_in[1]))
            elif i > 0 or need_dx:
                d, pt = ops.conv_dgrad(draw, w, G3, h_in.shape, wd=self._wd(cn))[0], None
            else:
                d, pt = None, None
        return d, pt

    def _vdown_bwd(self, e, d_act, pt, skip):
        _, name, x, raw, st, w = e
        g, P = self.grads, self.P
        gn, cn = name + ".conv.1", name + ".conv.0"
        draw, g[gn + ".weight"], g[gn + ".bias"] = ops.norm_relu_bwd(
            d_act, raw, st[0], st[1], P[gn + ".weight"], st[2], st[3], GROUPS, False, out=self._o(gn), partials=pt)
        g[cn + ".weight"], g[cn + ".bias"] = self._wg(
            ops.conv_wgrad, x, None, draw, G2S2, P[cn + ".weight"].shape, out=self._o(cn))
        return ops.conv_dgrad(draw, w, G2S2, x.shape, residual=skip, wd=self._wd(cn))[0]

    def _vup_bwd(self, e, d_out):
        """d_out = gradient w.r.t. relu(gn(up)) + skip.  -> (gradient w.r.t. the transposed conv's input, skip gradient)"""
        _, name, x, raw, st, wt = e
        g, P = self.grads, self.P
        gn, cn = name + ".conv.1", name + ".conv.0"
        draw, g[gn + ".weight"], g[gn + ".bias"] = ops.norm_relu_bwd(
            d_out, raw, st[0], st[1], P[gn + ".weight"], st[2], st[3], GROUPS, False, out=self._o(gn))
        g[cn + ".weight"], g[cn + ".bias"] = self._wg(
            ops.convT_wgrad, x, draw, (2, 2, 2), P[cn + ".weight"].shape, out=self._o(cn))
        return ops.convT_dgrad(draw, wt, (2, 2, 2), x.shape, wt=self._wd(cn)), d_out

    def backward(self, dlogits, need_dx):
        pop = self.tape.pop
        self._begin_backward(dlogits.device)
        d = self._conv_bwd(pop(), dlogits)                                     # out_conv
        skips = []
        for _ in range(3):                                                     # block_seven / six / five + their up convs
            d, _ = self._vblock_bwd(pop(), d)
            d, dsk = self._vup_bwd(pop(), d)
            skips.append(dsk)
        ds1, ds2, ds3 = skips
        self._bucket_done(0)
        d, pt = self._vblock_bwd(pop(), d)                                     # block_four (128 channels)
        self._bucket_done(1)
        d = self._vdown_bwd(pop(), d, pt, ds3)                                 # block_three_dw (+ skip gradient of x3)
        d, pt = self._vblock_bwd(pop(), d)
        d = self._vdown_bwd(pop(), d, pt, ds2)
        d, pt = self._vblock_bwd(pop(), d)
        d = self._vdown_bwd(pop(), d, pt, ds1)
        d, _ = self._vblock_bwd(pop(), d, need_dx)                             # block_one
        self._end_backward()
        self._bucket_done(2)
        return d


class VNet(Unet):
    """Drop-in for the reference ``VNet(n_channels, n_classes, n_filters, normalization='gn', has_dropout=False)``
    (VNet3D.py:145-235).  Other normalisations and ``has_dropout=True`` are off the DEviS path (TMSU never builds
    them) and raise."""

    _run_cls = _RunV
    _bucket_marks = ("block_four.", "block_four.")

    def __init__(self, n_channels=1, n_classes=4, n_filters=16, normalization='gn', has_dropout=False, compute_dtype=None):
        nn.Module.__init__(self)
        if normalization != 'gn' or has_dropout:
            raise NotImplementedError("VNet: only normalization='gn', has_dropout=False (what TMSU builds) is implemented")
        f = n_filters
        self.has_dropout = has_dropout
        self.block_one = _ConvBlock(1, n_channels, f)
        self.block_one_dw = _Down(f, f * 2)
        self.block_two = _ConvBlock(2, f * 2, f * 2)
        self.block_two_dw = _Down(f * 2, f * 4)
        self.block_three = _ConvBlock(3, f * 4, f * 4)
        self.block_three_dw = _Down(f * 4, f * 8)
        self.block_four = _ConvBlock(3, f * 8, f * 8)
        self.block_four_up = _Up(f * 8, f * 4)
        self.block_five = _ConvBlock(3, f * 4, f * 4)
        self.block_five_up = _Up(f * 4, f * 2)
        self.block_six = _ConvBlock(3, f * 2, f * 2)
        self.block_six_up = _Up(f * 2, f)
        self.block_seven = _ConvBlock(1, f, f)
        self.out_conv = nn.Conv3d(f, n_classes, 1, padding=0)
        self.dropout = nn.Dropout3d(p=0.5, inplace=False)
        self._in_channels = n_channels
        self._finish_init(compute_dtype if compute_dtype is not None else torch.bfloat16, n_classes)
