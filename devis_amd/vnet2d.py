"""2-D V-Net on MI355X: constructor, attribute names and ``state_dict`` keys of the reference ``VNet2D``
(models/lib/VNet2D.py:145-265) as TMSU builds it for ISIC / DRIVE / TOAR / HC with
``model_name='V'`` (``VNet2D(3, K, n_filters=32, normalization='gn', has_dropout=False)``,
TrustworthySeg.py:54,78,91,104), executed by the HIP kernels.

Every block is (conv -> GroupNorm(16) -> ReLU) x n, 32 ... 512 channels: the norm + ReLU of stage i runs in the
prologue of stage i+1's conv and its moments are gathered in stage i's conv epilogue (the 2-D MFMA kernels of
csrc/conv2d_mfma.hip); only a block's last activation is materialised.  ConvTranspose k2-s2 runs on the GEMM
kernels of the 2-D U-Net; the four k2-s2 'down' convolutions run on the generic direct kernels.  ``offsetAdd``
(VNet2D.py:234-255) pads the WIDTH of its second argument with ``torch.rand`` drawn from the global CPU generator
(same shapes and order as the reference), then adds.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from . import ops
from .ops import ConvGeom

Tensor = torch.Tensor

C3 = ConvGeom((1, 3, 3), (1, 1, 1), (0, 1, 1))
C1 = ConvGeom((1, 1, 1), (1, 1, 1), (0, 0, 0))
C2S2 = ConvGeom((1, 2, 2), (1, 2, 2), (0, 0, 0))
GROUPS = 16


def _norm(c):                                  # normalization == 'gn', VNet2D.py:19-20
    return nn.GroupNorm(num_groups=GROUPS, num_channels=c)


class _ConvBlock(nn.Module):                   # VNet2D.py:5-31 (parameter holder; Sequential indices as the reference)
    def __init__(self, n_stages, cin, cout):
        super().__init__()
        mods = []
        for i in range(n_stages):
            mods += [nn.Conv2d(cin if i == 0 else cout, cout, 3, padding=1), _norm(cout), nn.ReLU(inplace=True)]
        self.conv = nn.Sequential(*mods)


class _Down(nn.Module):                        # VNet2D.py:67-91
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin, cout, 2, padding=0, stride=2), _norm(cout), nn.ReLU(inplace=True))


class _Up(nn.Module):                          # VNet2D.py:94-118
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.ConvTranspose2d(cin, cout, 2, padding=0, stride=2), _norm(cout), nn.ReLU(inplace=True))


class _RunV2d:
    """Forward / backward schedule of the 2-D V-Net over channels-last (N, H, W, C) tensors."""

    def __init__(self, params, dtype, tape, packed):
        self.P = params
        self.W = packed                       # name.weight -> [w_packed, w_dgrad | None] (ops.WeightPacker: one launch)
        self.dtype, self.taping = dtype, tape
        self.tape: List = []
        self.grads: Dict[str, Tensor] = {}
        self.side = None

    def _push(self, *e):
        if self.taping:
            self.tape.append(e)

    def _w(self, name):
        return self.W[name + ".weight"][0]

    def _wd(self, name):
        return self.W[name + ".weight"][1]

    def _gn(self, name, x, partials=None):
        return ops.gn_stats(x, self.P[name + ".weight"], self.P[name + ".bias"], groups=GROUPS, partials=partials)

    # ---- forward blocks
    def conv_block(self, name, x, n_stages, pro=None):
        """x: a materialised tensor (pro None) or a raw conv output whose GroupNorm + ReLU (scale, shift = pro[:2]) is
        still to be applied.  -> the block's materialised output."""
        stages, h, p = [], x, pro
        for i in range(n_stages):
            cn, gn = "%s.conv.%d" % (name, 3 * i), "%s.conv.%d" % (name, 3 * i + 1)
            w = self._w(cn)
            raw, part = ops.conv_fwd(h, None, w, self.P[cn + ".bias"], C3, p[0] if p else None, p[1] if p else None,
                                     want_stats=True)
            st = self._gn(gn, raw, part)
            stages.append((cn, gn, h, p, raw, st, w))
            h, p = raw, st
        out = ops.affine_act(h, p[0], p[1], relu=True)
        self._push("vblock", name, stages)
        return out

    def down(self, name, x):
        if x.shape[1] % 2 or x.shape[2] % 2:                 # k2 s2 p0 drops an odd last row / column
            x = x[:, :x.shape[1] // 2 * 2, :x.shape[2] // 2 * 2, :].contiguous()
        w = self._w(name + ".conv.0")
        raw = ops.conv_fwd(x, None, w, self.P[name + ".conv.0.bias"], C2S2)
        st = self._gn(name + ".conv.1", raw)
        self._push("vdown", name, x, raw, st, w)
        return raw, st

    def up(self, name, x, skip, up_first: bool):
        """relu(gn(convT(x))) combined with `skip` by offsetAdd.  up_first: offsetAdd(up, skip) - the reference then
        pads the SKIP (VNet2D.py:220,224), which can only work when the widths already agree."""
        wt = self._w(name + ".conv.0")
        raw = ops.convT_fwd(x, wt, self.P[name + ".conv.0.bias"], (2, 2))
        st = self._gn(name + ".conv.1", raw)
        h, w = raw.shape[1], raw.shape[2]
        if skip.shape[1] != h:
            raise ValueError("VNet2D: skip and upsampled heights differ (%d vs %d): offsetAdd pads the width only "
                             "(VNet2D.py:237)" % (skip.shape[1], h))
        off = abs(skip.shape[2] - w)
        if off == 0:
            out = ops.affine_act(raw, st[0], st[1], residual=skip, relu=True)
        else:
            if up_first:
                raise ValueError("VNet2D: offsetAdd(up, skip) with different widths adds mismatched tensors in the "
                                 "reference too (VNet2D.py:220-224, 245-253)")
            act = ops.affine_act(raw, st[0], st[1], relu=True)
            n, c = raw.shape[0], raw.shape[3]
            add = torch.rand((n, c, h)).unsqueeze(3) if off == 1 else torch.rand((n, c, h, off))    # VNet2D.py:238-252
            padded = torch.cat([act, add.to(act.device).permute(0, 2, 3, 1).to(act.dtype)], dim=2).contiguous()
            out = ops.add(skip, padded)
        self._push("vup", name, x, raw, st, wt, w)
        return out

    def forward(self, x):
        x1 = self.conv_block("block_one", x, 1)
        r, st = self.down("block_one_dw", x1)
        x2 = self.conv_block("block_two", r, 2, st)
        r, st = self.down("block_two_dw", x2)
        x3 = self.conv_block("block_three", r, 3, st)
        r, st = self.down("block_three_dw", x3)
        x4 = self.conv_block("block_four", r, 3, st)
        r, st = self.down("block_four_dw", x4)
        x5 = self.conv_block("block_five", r, 3, st)
        x6 = self.conv_block("block_six", self.up("block_five_up", x5, x4, False), 3)
        x7 = self.conv_block("block_seven", self.up("block_six_up", x6, x3, False), 3)
        x8 = self.conv_block("block_eight", self.up("block_seven_up", x7, x2, True), 2)
        x9 = self.conv_block("block_nine", self.up("block_eight_up", x8, x1, True), 2)
        wp = self._w("out_conv")
        out = ops.conv_fwd(x9, None, wp, self.P["out_conv.bias"], C1)
        self._push("out", x9, wp)
        return out

    # ---- backward blocks
    def _wg(self, fn, *args, **kw):
        """A weight-gradient call on the side stream (see unet3d._Run._wg)."""
        side = self.side
        if side is None:
            return fn(*args, **kw)
        ev = self._events[self._nev % len(self._events)]
        self._nev += 1
        ev.record()
        side.wait_event(ev)
        with ops.stream_override(side):
            out = fn(*args, slot="ws_side", **kw)
        self._keep.append(args)
        return out

    def _vblock_bwd(self, d, need_dx=True):
        """d = gradient w.r.t. the block's materialised output.  -> (gradient w.r.t. the block's input - the ACTIVATED
        input when it came with a prologue -, norm-backward moments for that prologue | None)."""
        _, name, stages = self.tape.pop()
        g, P = self.grads, self.P
        pt = None
        for i in reversed(range(len(stages))):
            cn, gn, h_in, p_in, raw, st, w = stages[i]
            draw, g[gn + ".weight"], g[gn + ".bias"] = ops.norm_relu_bwd(
                d, raw, st[0], st[1], P[gn + ".weight"], st[2], st[3], GROUPS, False, partials=pt)
            g[cn + ".weight"], g[cn + ".bias"] = self._wg(
                ops.conv_wgrad, h_in, None, draw, C3, P[cn + ".weight"].shape, p_in[0] if p_in else None,
                p_in[1] if p_in else None)
            if p_in is not None:
                d, _, pt = ops.conv_dgrad(draw, w, C3, h_in.shape, wd=self._wd(cn), norm=(h_in, p_in[0], p_in[1]))
            elif i > 0 or need_dx:
                d, pt = ops.conv_dgrad(draw, w, C3, h_in.shape, wd=self._wd(cn))[0], None
            else:
                d, pt = None, None
        return d, pt

    def _vdown_bwd(self, d_act, pt, skip, full_shape):
        _, name, x, raw, st, w = self.tape.pop()
        g, P = self.grads, self.P
        gn, cn = name + ".conv.1", name + ".conv.0"
        draw, g[gn + ".weight"], g[gn + ".bias"] = ops.norm_relu_bwd(
            d_act, raw, st[0], st[1], P[gn + ".weight"], st[2], st[3], GROUPS, False, partials=pt)
        g[cn + ".weight"], g[cn + ".bias"] = self._wg(ops.conv_wgrad, x, None, draw, C2S2, P[cn + ".weight"].shape)
        if tuple(x.shape) == tuple(full_shape):
            return ops.conv_dgrad(draw, w, C2S2, x.shape, residual=skip, wd=self._wd(cn))[0]
        dx = ops.conv_dgrad(draw, w, C2S2, x.shape, wd=self._wd(cn))[0]           # an odd row / column the conv dropped: skip gradient only
        out = skip.clone()
        out[:, :x.shape[1], :x.shape[2], :] = ops.add(dx, skip[:, :x.shape[1], :x.shape[2], :].contiguous())
        return out

    def _vup_bwd(self, d_out):
        """d_out = gradient w.r.t. offsetAdd(skip, relu(gn(up))).  -> (gradient w.r.t. the transposed conv's input,
        skip gradient)"""
        _, name, x, raw, st, wt, w = self.tape.pop()
        g, P = self.grads, self.P
        gn, cn = name + ".conv.1", name + ".conv.0"
        d_up = d_out if d_out.shape[2] == w else d_out[:, :, :w, :].contiguous()     # the rand columns carry no gradient
        draw, g[gn + ".weight"], g[gn + ".bias"] = ops.norm_relu_bwd(
            d_up, raw, st[0], st[1], P[gn + ".weight"], st[2], st[3], GROUPS, False)
        g[cn + ".weight"], g[cn + ".bias"] = self._wg(ops.convT_wgrad, x, draw, (2, 2), P[cn + ".weight"].shape)
        return ops.convT_dgrad(draw, wt, (2, 2), x.shape, wt=self._wd(cn)), d_out

    def backward(self, dlogits, need_dx):
        from .unet3d import _side_events, _side_stream
        self.side = _side_stream(dlogits.device)
        self._keep, self._nev = [], 0
        if self.side is not None:
            self._events = _side_events(dlogits.device)
        _, x9, wp = self.tape.pop()
        self.grads["out_conv.weight"], self.grads["out_conv.bias"] = self._wg(
            ops.conv_wgrad, x9, None, dlogits, C1, self.P["out_conv.weight"].shape)
        d = ops.conv_dgrad(dlogits, wp, C1, x9.shape, wd=self._wd("out_conv"))[0]
        skips = []
        for _ in range(4):                                   # block_nine / eight / seven / six + their up convs
            d, _ = self._vblock_bwd(d)
            d, dsk = self._vup_bwd(d)
            skips.append(dsk)
        ds1, ds2, ds3, ds4 = skips
        d, pt = self._vblock_bwd(d)                          # block_five
        for dsk in (ds4, ds3, ds2, ds1):                     # block_four_dw + block_four ... block_one_dw + block_one
            d = self._vdown_bwd(d, pt, dsk, dsk.shape)
            d, pt = self._vblock_bwd(d, need_dx or dsk is not ds1)
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self._keep = []
        return d


class _VNet2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        names = model._param_names
        taping = any(ctx.needs_input_grad)
        P = dict(zip(names, params))
        run = _RunV2d(P, model.compute_dtype, taping, model._packed(P))
        logits = run.forward(ops.to_channels_last(x, model.compute_dtype))
        ctx.run, ctx.names, ctx.need_dx = run, names, ctx.needs_input_grad[1]
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        run = ctx.run
        if not run.taping or not run.tape:
            raise RuntimeError("devis_amd VNet2D: backward called twice or without a taped forward")
        dx_cl = run.backward(dlogits.contiguous(), ctx.need_dx)
        dx = ops.from_channels_last(dx_cl) if dx_cl is not None else None
        g = run.grads
        run.tape, run.grads = [], {}
        return (None, dx) + tuple(g.pop(n, None) for n in ctx.names)


class VNet2D(nn.Module):
    """Drop-in for the reference ``VNet2D(n_channels, n_classes, n_filters, normalization='gn', has_dropout=False)``.
    Other normalisations and ``has_dropout=True`` are off the DEviS path (TMSU never builds them) and raise."""

    def __init__(self, n_channels=3, n_classes=1, n_filters=32, normalization='gn', has_dropout=False,
                 compute_dtype=torch.bfloat16):
        super().__init__()
        if normalization != 'gn' or has_dropout:
            raise NotImplementedError("VNet2D: only normalization='gn', has_dropout=False (what TMSU builds) is implemented")
        f = n_filters
        self.has_dropout = has_dropout
        self.block_one = _ConvBlock(1, n_channels, f)
        self.block_one_dw = _Down(f, f * 2)
        self.block_two = _ConvBlock(2, f * 2, f * 2)
        self.block_two_dw = _Down(f * 2, f * 4)
        self.block_three = _ConvBlock(3, f * 4, f * 4)
        self.block_three_dw = _Down(f * 4, f * 8)
        self.block_four = _ConvBlock(3, f * 8, f * 8)
        self.block_four_dw = _Down(f * 8, f * 16)
        self.block_five = _ConvBlock(3, f * 16, f * 16)
        self.block_five_up = _Up(f * 16, f * 8)
        self.block_six = _ConvBlock(3, f * 8, f * 8)
        self.block_six_up = _Up(f * 8, f * 4)
        self.block_seven = _ConvBlock(3, f * 4, f * 4)
        self.block_seven_up = _Up(f * 4, f * 2)
        self.block_eight = _ConvBlock(2, f * 2, f * 2)
        self.block_eight_up = _Up(f * 2, f)
        self.block_nine = _ConvBlock(2, f, f)
        self.out_conv = nn.Conv2d(f, n_classes, 1, padding=0)
        self.dropout = nn.Dropout2d(p=0.5, inplace=False)
        self.compute_dtype = compute_dtype
        self.in_ch, self.num_classes = n_channels, n_classes
        self._param_names = [n for n, _ in self.named_parameters()]
        self._conv_modules = [(n + ".weight", m) for n, m in self.named_modules()
                              if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))]
        self._packer = None

    def _packed(self, P):
        """All conv weights in the kernels' layouts by ONE launch per step (ops.WeightPacker), as the other nets."""
        ent = [(n, P[n].detach(), "convT" if isinstance(m, nn.ConvTranspose2d) else "conv") for n, m in self._conv_modules]
        key = tuple(w.data_ptr() for _, w, _ in ent) + (self.compute_dtype,)
        if self._packer is None or self._packer.key != key:
            self._packer = ops.WeightPacker(ent, self.compute_dtype, ent[0][1].device)
        return self._packer.pack()

    def forward_channels_last(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_ch:
            raise ValueError("expected (N, %d, H, W), got %s" % (self.in_ch, tuple(x.shape)))
        if x.shape[2] < 16 or x.shape[3] < 16:
            raise ValueError("VNet2D needs H, W >= 16 (four stride-2 stages)")
        ops.require_device(x, self.out_conv.weight)
        return _VNet2dFn.apply(self, x, *[p for _, p in self.named_parameters()])

    def forward(self, x, turnoff_drop=False):
        from .unet3d import _FromChannelsLast
        return _FromChannelsLast.apply(self.forward_channels_last(x))
