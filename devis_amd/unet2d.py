"""2-D U-Nets on MI355X: ``Unet2D`` and ``Unet2DDRIVE`` with the reference's constructor,
attribute names and ``state_dict`` keys (models/lib/UNet2DZoo.py:107-259).

2-D tensors are the D = 1 case of the same channels-last kernels.  ``offsetCat`` pads the
upsampled tensor with ``torch.rand`` drawn from the GLOBAL CPU generator with the reference's
shapes and order (UNet2DZoo.py:162-182, 239-259), so a seeded run reproduces the reference.
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


class _DoubleConv(nn.Module):        # UNet2DZoo.py:8-21; keys conv.0 / conv.1 / conv.3 / conv.4
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
            nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class _Run2d:
    def __init__(self, params, buffers, dtype, training, tape, drive, packed):
        self.P, self.B = params, buffers
        self.W = packed                       # name.weight -> [w_packed, w_dgrad | None] (ops.WeightPacker: one launch)
        self.dtype, self.training, self.taping, self.drive = dtype, training, tape, drive
        self.tape: List = []
        self.grads: Dict[str, Tensor] = {}

    def _push(self, *e):
        if self.taping:
            self.tape.append(e)

    def _w(self, name):
        return self.W[name + ".weight"][0]

    def _wd(self, name):
        return self.W[name + ".weight"][1]

    def _bn(self, name, x, partials=None):
        return ops.bn_stats(x, self.P[name + ".weight"], self.P[name + ".bias"],
                            self.B[name + ".running_mean"], self.B[name + ".running_var"], self.training,
                            tracked=self.B[name + ".num_batches_tracked"], partials=partials)

    def _conv_stats(self, *a):
        """conv + the moments of its output for the BatchNorm that follows, gathered in the conv's epilogue
        when the kernel for this shape can (the MFMA layers in bf16); else the separate pass inside bn_stats."""
        if not self.training:
            return ops.conv_fwd(*a), None
        return ops.conv_fwd(*a, want_stats=True)

    def double_conv(self, name, x1, x2=None):
        P = self.P
        w0 = self._w(name + ".conv.0")
        h1, part = self._conv_stats(x1, x2, w0, P[name + ".conv.0.bias"], C3)
        st1 = self._bn(name + ".conv.1", h1, part)
        w3 = self._w(name + ".conv.3")
        h2, part = self._conv_stats(h1, None, w3, P[name + ".conv.3.bias"], C3, st1[0], st1[1])
        st2 = self._bn(name + ".conv.4", h2, part)
        out = ops.affine_act(h2, st2[0], st2[1], relu=True)
        self._push("dc", name, x1, x2, h1, h2, st1, st2, w0, w3)
        return out

    def pool(self, x):
        y, idx = ops.maxpool2d_fwd(x)
        self._push("pool", idx, tuple(x.shape))
        return y

    def up(self, name, x):
        wt = self._w(name)
        y = ops.convT_fwd(x, wt, self.P[name + ".bias"], (2, 2))
        self._push("up", name, x, wt)
        return y

    def offset_pad(self, skip, up):
        """Pad `up` (channels-last) to the skip's size with CPU-generator uniforms."""
        n, h, w, c = up.shape
        if self.drive and skip.shape[1] != h:                      # UNet2DZoo.py:244-250
            off = abs(skip.shape[1] - h)
            add = torch.rand((n, c, off, w))
            padded = torch.cat([up, add.to(up.device).permute(0, 2, 3, 1).to(up.dtype)], dim=1)
        elif skip.shape[2] != w:
            off = abs(skip.shape[2] - w)
            if off == 1 and not self.drive:                        # UNet2DZoo.py:166-173
                add = torch.rand((n, c, h)).unsqueeze(3)
            else:
                add = torch.rand((n, c, h, off))
            padded = torch.cat([up, add.to(up.device).permute(0, 2, 3, 1).to(up.dtype)], dim=2)
        else:
            padded = up
        self._push("pad", (h, w))
        return padded.contiguous()

    def forward(self, x):
        c1 = self.double_conv("conv1", x)
        c2 = self.double_conv("conv2", self.pool(c1))
        c3 = self.double_conv("conv3", self.pool(c2))
        c4 = self.double_conv("conv4", self.pool(c3))
        c5 = self.double_conv("conv5", self.pool(c4))
        c6 = self.double_conv("conv6", c4, self.offset_pad(c4, self.up("up6", c5)))      # (skip, up)
        c7 = self.double_conv("conv7", c3, self.offset_pad(c3, self.up("up7", c6)))
        c8 = self.double_conv("conv8", self.up("up8", c7), c2)                            # (up, skip)
        if self.drive:
            c9 = self.double_conv("conv9", c1, self.offset_pad(c1, self.up("up9", c8)))  # UNet2DZoo.py:233
        else:
            c9 = self.double_conv("conv9", self.up("up9", c8), c1)                        # UNet2DZoo.py:156
        wp = self._w("conv10")
        out = ops.conv_fwd(c9, None, wp, self.P["conv10.bias"], C1)
        self._push("c10", c9, wp)
        return out

    # ------------------------------------------------------------ backward
    def _wg(self, fn, *args, **kw):
        """A weight-gradient call on the side stream (see unet3d._Run._wg): nothing on backward's critical path
        reads it, and the deep 16x16 / 32x32 layers leave most of the chip idle for it."""
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

    def _dc_bwd(self, dout, need_dx1=True):
        _, name, x1, x2, h1, h2, st1, st2, w0, w3 = self.tape.pop()
        g, P = self.grads, self.P
        dh2, g[name + ".conv.4.weight"], g[name + ".conv.4.bias"] = ops.norm_relu_bwd(
            dout, h2, st2[0], st2[1], P[name + ".conv.4.weight"], st2[2], st2[3], 0, 1 if self.training else 2)
        g[name + ".conv.3.weight"], g[name + ".conv.3.bias"] = self._wg(
            ops.conv_wgrad, h1, None, dh2, C3, P[name + ".conv.3.weight"].shape, st1[0], st1[1])
        # this dgrad feeds bn1's backward: its epilogue leaves the two moments that backward needs
        da, _, part = ops.conv_dgrad(dh2, w3, C3, h1.shape, wd=self._wd(name + ".conv.3"), norm=(h1, st1[0], st1[1]))
        dh1, g[name + ".conv.1.weight"], g[name + ".conv.1.bias"] = ops.norm_relu_bwd(
            da, h1, st1[0], st1[1], P[name + ".conv.1.weight"], st1[2], st1[3], 0, 1 if self.training else 2, partials=part)
        g[name + ".conv.0.weight"], g[name + ".conv.0.bias"] = self._wg(
            ops.conv_wgrad, x1, x2, dh1, C3, P[name + ".conv.0.weight"].shape)
        if not need_dx1:
            return None, None
        return ops.conv_dgrad(dh1, w0, C3, x1.shape, 0 if x2 is None else x2.shape[-1], wd=self._wd(name + ".conv.0"))

    def _up_bwd(self, dy):
        _, name, x, wt = self.tape.pop()
        self.grads[name + ".weight"], self.grads[name + ".bias"] = self._wg(
            ops.convT_wgrad, x, dy, (2, 2), self.P[name + ".weight"].shape)
        return ops.convT_dgrad(dy, wt, (2, 2), x.shape, wt=self._wd(name))

    def _pad_bwd(self, d):
        _, (h, w) = self.tape.pop()
        return d[:, :h, :w, :].contiguous() if (d.shape[1] != h or d.shape[2] != w) else d

    def _pool_bwd(self, dy, skip):
        _, idx, shape = self.tape.pop()
        return ops.maxpool2d_bwd(dy, idx, shape, skip)

    def backward(self, dlogits, need_dx):
        from .unet3d import _side_stream, _side_events
        self.side = _side_stream(dlogits.device)
        self._keep, self._nev = [], 0
        if self.side is not None:
            self._events = _side_events(dlogits.device)
        _, c9, wp = self.tape.pop()
        self.grads["conv10.weight"], self.grads["conv10.bias"] = self._wg(
            ops.conv_wgrad, c9, None, dlogits, C1, self.P["conv10.weight"].shape)
        d = ops.conv_dgrad(dlogits, wp, C1, c9.shape, wd=self._wd("conv10"))[0]
        a, b = self._dc_bwd(d)                         # conv9
        if self.drive:
            dc1, d = a, self._up_bwd(self._pad_bwd(b))
        else:
            dc1, d = b, self._up_bwd(a)
        a, b = self._dc_bwd(d)                         # conv8: (up8, c2)
        dc2, d = b, self._up_bwd(a)
        a, b = self._dc_bwd(d)                         # conv7: (c3, up7 padded)
        dc3, d = a, self._up_bwd(self._pad_bwd(b))
        a, b = self._dc_bwd(d)                         # conv6: (c4, up6 padded)
        dc4, d = a, self._up_bwd(self._pad_bwd(b))
        d, _ = self._dc_bwd(d)                         # conv5
        d = self._pool_bwd(d, dc4)
        d, _ = self._dc_bwd(d)                         # conv4
        d = self._pool_bwd(d, dc3)
        d, _ = self._dc_bwd(d)                         # conv3
        d = self._pool_bwd(d, dc2)
        d, _ = self._dc_bwd(d)                         # conv2
        d = self._pool_bwd(d, dc1)
        d, _ = self._dc_bwd(d, need_dx)                # conv1
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self._keep = []                                # freed behind the join, on the stream that allocated them
        return d


class _Unet2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        names = model._param_names
        taping = any(ctx.needs_input_grad)
        P = dict(zip(names, params))
        run = _Run2d(P, dict(model.named_buffers()), model.compute_dtype, model.training, taping, model._drive,
                     model._packed(P))
        logits = run.forward(ops.to_channels_last(x, model.compute_dtype))
        ctx.run, ctx.names, ctx.need_dx = run, names, ctx.needs_input_grad[1]
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        run = ctx.run
        if not run.taping or not run.tape:
            raise RuntimeError("devis_amd Unet2D: backward called twice or without a taped forward")
        dx_cl = run.backward(dlogits.contiguous(), ctx.need_dx)
        dx = ops.from_channels_last(dx_cl) if dx_cl is not None else None
        # hand the gradients over without keeping another reference: autograd then adopts them as .grad instead of
        # cloning each one (82 device copies per step otherwise)
        g = run.grads
        run.tape, run.grads = [], {}
        return (None, dx) + tuple(g.pop(n, None) for n in ctx.names)


class Unet2D(nn.Module):
    """Drop-in for the reference ``Unet2D(in_ch, out_ch)``; returns raw logits ``c10``."""
    _drive = False

    def __init__(self, in_ch, out_ch, compute_dtype=torch.bfloat16):
        super().__init__()
        self.conv1 = _DoubleConv(in_ch, 32)
        self.conv2 = _DoubleConv(32, 64)
        self.conv3 = _DoubleConv(64, 128)
        self.conv4 = _DoubleConv(128, 256)
        self.conv5 = _DoubleConv(256, 512)
        self.up6 = nn.ConvTranspose2d(512, 256, 2, stride=2)
        self.conv6 = _DoubleConv(512, 256)
        self.up7 = nn.ConvTranspose2d(256, 128, 2, stride=2)
        self.conv7 = _DoubleConv(256, 128)
        self.up8 = nn.ConvTranspose2d(128, 64, 2, stride=2)
        self.conv8 = _DoubleConv(128, 64)
        self.up9 = nn.ConvTranspose2d(64, 32, 2, stride=2)
        self.conv9 = _DoubleConv(64, 32)
        self.conv10 = nn.Conv2d(32, out_ch, 1)
        self.compute_dtype = compute_dtype
        self.in_ch, self.num_classes = in_ch, out_ch
        self._param_names = [n for n, _ in self.named_parameters()]
        self._conv_modules = [(n + ".weight", m) for n, m in self.named_modules()
                              if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))]
        self._packer = None

    def _packed(self, P):
        """All conv weights in the kernels' layouts - forward and dgrad packings - by ONE launch per step (41 per-layer
        packing launches before: 0.23 ms of a 4.7 ms step); the tables are rebuilt only if a parameter moved."""
        ent = [(n, P[n].detach(), "convT" if isinstance(m, nn.ConvTranspose2d) else "conv") for n, m in self._conv_modules]
        key = tuple(w.data_ptr() for _, w, _ in ent) + (self.compute_dtype,)
        if self._packer is None or self._packer.key != key:
            self._packer = ops.WeightPacker(ent, self.compute_dtype, ent[0][1].device)
        return self._packer.pack()

    def forward_channels_last(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_ch:
            raise ValueError("expected (N, %d, H, W), got %s" % (self.in_ch, tuple(x.shape)))
        if x.shape[2] < 16 or x.shape[3] < 16:
            raise ValueError("Unet2D needs H, W >= 16 (four 2x2 poolings)")
        return _Unet2dFn.apply(self, x, *[p for _, p in self.named_parameters()])

    def forward(self, x):
        from .unet3d import _FromChannelsLast
        return _FromChannelsLast.apply(self.forward_channels_last(x))


class Unet2DDRIVE(Unet2D):
    """Drop-in for ``Unet2DDRIVE``: same layers/keys; pads H or W and merges conv9 as (skip, up)."""
    _drive = True
