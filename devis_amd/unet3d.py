"""3-D residual U-Net on MI355X: same constructor, attribute names and ``state_dict`` keys
as the reference ``Unet`` (models/lib/UNet3DZoo.py:243-305), executed by the HIP kernels.

The whole backbone is ONE autograd node (``_Unet3dFn``): forward and backward are explicit
schedules of C-ABI calls over channels-last activations, so that
  * GroupNorm/BatchNorm-apply + ReLU run in the consuming conv's prologue
    (the normalised tensor is never written to HBM),
  * residual adds and the skip-gradient adds run in conv / norm-backward epilogues,
  * ``torch.cat`` (UNet3DZoo.py:128) is a two-pointer K loop inside the 1x1 conv.
nn.Conv3d / nn.GroupNorm / nn.BatchNorm3d objects are used as parameter holders only
(identical init order => identical weights for a given torch seed); they are never called.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from . import launch_plan as _plan
from . import ops
from .ops import ConvGeom

Tensor = torch.Tensor

G3 = ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
G3S2 = ConvGeom((3, 3, 3), (2, 2, 2), (1, 1, 1))
G1 = ConvGeom((1, 1, 1), (1, 1, 1), (0, 0, 0))
G2S2 = ConvGeom((2, 2, 2), (2, 2, 2), (0, 0, 0))
# Schedule constants of the two-stream backward (measured in the step, DESIGN.md "two-stream backward"; tools/lag_check.py
# and tools/lag_diff.py set them on the module for A/B runs - no environment switches).
# weight-gradient calls handed to the side stream per event (see _Run._wg)
_WG_BATCH = 2
# (a, b): a weight-gradient call on tensors of >= _WG_BIG elements (the full-resolution layers) is handed to the side
# stream `a` schedule blocks after the one that issued it, the others `b` blocks after (_Run._tick)
_WG_LAG = (1, 2)
_WG_BIG = 1 << 24
# The main stream ends its backward before the side stream has worked off its queue (0.3 ms of weight gradients ran
# alone at the end of the step): the last _WG_TAIL_MAIN weight-gradient calls of the 3-D U-Net's backward
# (InitConv; EnBlock1.conv1; EnBlock1.conv2) run on the main stream instead, beside the side stream's queue.
_WG_TAIL_MAIN = 2


# ------------------------------------------------------------ parameter holders
class _InitConv(nn.Module):          # UNet3DZoo.py:21-32 (dropout p = 0 in Unet => identity, no RNG draw)
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size=3, padding=1)


class _EnBlock(nn.Module):           # UNet3DZoo.py:35-56
    def __init__(self, c):
        super().__init__()
        self.bn1 = nn.GroupNorm(8, c)
        self.conv1 = nn.Conv3d(c, c, kernel_size=3, padding=1)
        self.bn2 = nn.GroupNorm(8, c)
        self.conv2 = nn.Conv3d(c, c, kernel_size=3, padding=1)


class _EnDown(nn.Module):            # UNet3DZoo.py:59-67
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size=3, stride=2, padding=1)


class _DeUpCat(nn.Module):           # UNet3DZoo.py:117-130
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, cout, kernel_size=1)
        self.conv2 = nn.ConvTranspose3d(cout, cout, kernel_size=2, stride=2)
        self.conv3 = nn.Conv3d(cin, cout, kernel_size=1)


class _DeBlock(nn.Module):           # UNet3DZoo.py:145-165
    def __init__(self, c):
        super().__init__()
        self.bn1 = nn.BatchNorm3d(c)
        self.conv1 = nn.Conv3d(c, c, kernel_size=3, padding=1)
        self.conv2 = nn.Conv3d(c, c, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm3d(c)


# ------------------------------------------------------------ explicit schedule
class _Run:
    """One forward (and, if taped, backward) pass of the 3-D U-Net over channels-last tensors."""

    def __init__(self, params: Dict[str, Tensor], buffers: Dict[str, Tensor], dtype, training: bool, tape: bool,
                 packed: Dict[str, list]):
        self.P, self.B = params, buffers
        self.W = packed                       # name.weight -> [w_packed, w_dgrad | None] (ops.WeightPacker)
        self.G: Dict[str, Tensor] = {}        # name -> fp32 view of the flat gradient buffer (set by backward)
        self.sync = None                      # parallel.FlatGradSync or None
        self.dtype, self.training, self.taping = dtype, training, tape
        self.tape: List = []
        self.grads: Dict[str, Tensor] = {}
        self.side = None                      # stream the weight-gradient kernels run on during backward (or None)

    # ---- helpers
    def _w(self, name):
        return self.W[name + ".weight"][0]

    def _wd(self, name):
        return self.W[name + ".weight"][1]

    def _push(self, *entry):
        if self.taping:
            self.tape.append(entry)

    def _gn(self, name, x, partials=None):
        return ops.gn_stats(x, self.P[name + ".weight"], self.P[name + ".bias"], partials=partials)

    def _bn(self, name, x, partials=None):
        return ops.bn_stats(x, self.P[name + ".weight"], self.P[name + ".bias"],
                            self.B[name + ".running_mean"], self.B[name + ".running_var"], self.training,
                            tracked=self.B[name + ".num_batches_tracked"], partials=partials)

    # ---- forward blocks.  A conv whose output feeds a norm gathers that norm's moments in its epilogue
    # (want_stats); `xs` / the second return value carry those per-chunk partials to the consumer.
    def conv(self, name, x, geom, want_stats=False):
        wp = self._w(name)
        out = ops.conv_fwd(x, None, wp, self.P[name + ".bias"], geom, want_stats=want_stats)
        self._push("conv", name, x, wp, geom)
        return out

    def en_block(self, name, x, xs=None, out_stats=False):
        st1 = self._gn(name + ".bn1", x, xs)
        w1 = self._w(name + ".conv1")
        h, hs = ops.conv_fwd(x, None, w1, self.P[name + ".conv1.bias"], G3, st1[0], st1[1], want_stats=True)
        st2 = self._gn(name + ".bn2", h, hs)
        w2 = self._w(name + ".conv2")
        out, os_ = ops.conv_fwd(h, None, w2, self.P[name + ".conv2.bias"], G3, st2[0], st2[1], residual=x,
                                want_stats=True) if out_stats else \
            (ops.conv_fwd(h, None, w2, self.P[name + ".conv2.bias"], G3, st2[0], st2[1], residual=x), None)
        self._push("en", name, x, h, st1, st2, w1, w2)
        return out, os_

    def de_block(self, name, x):
        tr = self.training            # eval-mode BatchNorm uses the running statistics: nothing to gather
        w1 = self._w(name + ".conv1")
        h1, s1 = ops.conv_fwd(x, None, w1, self.P[name + ".conv1.bias"], G3, want_stats=True) if tr else \
            (ops.conv_fwd(x, None, w1, self.P[name + ".conv1.bias"], G3), None)
        st1 = self._bn(name + ".bn1", h1, s1)
        w2 = self._w(name + ".conv2")
        h2, s2 = ops.conv_fwd(h1, None, w2, self.P[name + ".conv2.bias"], G3, st1[0], st1[1], want_stats=True) if tr else \
            (ops.conv_fwd(h1, None, w2, self.P[name + ".conv2.bias"], G3, st1[0], st1[1]), None)
        st2 = self._bn(name + ".bn2", h2, s2)
        out = ops.affine_act(h2, st2[0], st2[1], residual=x, relu=True)
        self._push("de", name, x, h1, h2, st1, st2, w1, w2)
        return out

    def de_up_cat(self, name, x, prev):
        w1 = self._w(name + ".conv1")
        t = ops.conv_fwd(x, None, w1, self.P[name + ".conv1.bias"], G1)
        if ops.upcat_supported(prev, t):
            # ConvT + cat + 1x1 in one pass on composite weights (csrc/upcat.hip): `up` is never written
            m = ops.upcat_prepare(self.P[name + ".conv3.weight"], self.P[name + ".conv3.bias"],
                                  self.P[name + ".conv2.weight"], self.P[name + ".conv2.bias"])
            y = ops.upcat_fwd(prev, t, m)
            self._push("upf", name, x, prev, t, m, w1)
            return y
        wt = self._w(name + ".conv2")
        up = ops.convT_fwd(t, wt, self.P[name + ".conv2.bias"], (2, 2, 2))
        w3 = self._w(name + ".conv3")
        y = ops.conv_fwd(prev, up, w3, self.P[name + ".conv3.bias"], G1)     # virtual cat((prev, up))
        self._push("up", name, x, prev, t, up, w1, wt, w3)
        return y

    def forward(self, x: Tensor) -> Tensor:
        h, hs = self.conv("InitConv.conv", x, G3, True)
        s1, _ = self.en_block("EnBlock1", h, hs)
        h, hs = self.conv("EnDown1.conv", s1, G3S2, True)
        h, hs = self.en_block("EnBlock2_1", h, hs, True)
        s2, _ = self.en_block("EnBlock2_2", h, hs)
        h, hs = self.conv("EnDown2.conv", s2, G3S2, True)
        h, hs = self.en_block("EnBlock3_1", h, hs, True)
        s3, _ = self.en_block("EnBlock3_2", h, hs)
        h, hs = self.conv("EnDown3.conv", s3, G3S2, True)
        for i in (1, 2, 3, 4):
            h, hs = self.en_block("EnBlock4_%d" % i, h, hs, i < 4)
        h = self.de_block("DeBlock4", self.de_up_cat("DeUpCat4", h, s3))
        h = self.de_block("DeBlock3", self.de_up_cat("DeUpCat3", h, s2))
        h = self.de_block("DeBlock2", self.de_up_cat("DeUpCat2", h, s1))
        return self.conv("endconv", h, G1)

    # ---- backward blocks
    def _o(self, name, a=".weight", b=".bias"):
        """(buffer for name+a, buffer for name+b): where the kernels write that layer's gradients"""
        return (self.G[name + a], self.G[name + b])

    def _wg(self, fn, *args, **kw):
        """Run a weight-gradient call (every caller passes its `out` buffers).  Nothing on backward's critical path
        (dgrad -> norm backward -> dgrad ...) reads a weight gradient, so with a side stream they are launched there,
        behind an event that marks their inputs ready, and fill the CUs the small deep-layer kernels of the main
        stream leave idle.  The calls allocate nothing (gradient views and workspace are given), so the stream is
        handed to the kernels directly; their inputs are kept referenced until backward has joined the side stream,
        so the caching allocator cannot hand their memory out while it is still being read.  Calls are handed over
        in groups of `_WG_BATCH`: one event + one stream wait per group (a hipStreamWaitEvent costs the host ~15 us,
        three kernel launches' worth)."""
        if self.side is None:
            return fn(*args, **kw)
        if self._wg_left is not None:             # the last calls of the backward (see backward()): on the main stream
            self._wg_left -= 1
            if self._wg_left < _WG_TAIL_MAIN:
                return fn(*args, **kw)
        self._wg_seq += 1
        big = any(isinstance(a, Tensor) and a.numel() >= _WG_BIG for a in args)
        lag = (_WG_LAG[0] if big else _WG_LAG[1]) if self._lagged else 0
        due = max(self._blk + lag, self._last_due)          # never overtakes an earlier call
        self._last_due = due
        if due > self._blk:
            self._held.append((due, fn, args, kw))
            return kw["out"]
        self._pending.append((fn, args, kw))
        if len(self._pending) >= _WG_BATCH:
            self._flush_wg()
        return kw["out"]

    def _tick(self, final=False):
        """A block of the backward schedule is done (drivers that call this set `_lagged`): hand over the held
        weight-gradient calls that are due; all of them with `final`."""
        self._blk += 1
        if self._held:
            n = len(self._held) if final else next((i for i, h in enumerate(self._held) if h[0] > self._blk), len(self._held))
            if n:
                self._pending += [h[1:] for h in self._held[:n]]
                self._held = self._held[n:]
                self._flush_wg()

    def _flush_wg(self):
        pend = self._pending
        if not pend:
            return
        self._pending = []
        side = self.side
        ev = self._events[self._nev % len(self._events)]
        self._nev += 1
        _plan.call(ev.record, self._main)         # on the main (current) stream: the group's inputs are ready behind it
        _plan.call(side.wait_event, ev)
        with ops.stream_override(side):
            for fn, args, kw in pend:
                fn(*args, slot="ws_side", **kw)
                self._keep.append(args)
        self._wg_issued += len(pend)
        if self.sync is not None:
            self._launch_ready_buckets()

    def _next_norm(self):
        """(h2, scale, shift) of the DeBlock whose backward comes next on the tape: its bn2+ReLU backward needs
        {sum g, sum g*h2} of the gradient the CURRENT layer's dgrad is about to produce, so that dgrad gathers them."""
        e = self.tape[-1] if self.tape else None
        return (e[4], e[6][0], e[6][1]) if e is not None and e[0] == "de" else None

    def _conv_bwd(self, e, dy, need_dx=True, skip=None, norm=None):
        """-> dx, or (dx, partials) with `norm`"""
        _, name, x, wp, geom = e
        g = self.grads
        g[name + ".weight"], g[name + ".bias"] = self._wg(ops.conv_wgrad, x, None, dy, geom, self.P[name + ".weight"].shape,
                                                              out=self._o(name))
        if norm is not None:
            dx, _, pt = ops.conv_dgrad(dy, wp, geom, x.shape, residual=skip, wd=self._wd(name), norm=norm)
            return dx, pt
        return ops.conv_dgrad(dy, wp, geom, x.shape, residual=skip, wd=self._wd(name))[0] if need_dx else None

    def _norm_bwd_dgrad(self, gr, xn, st, gamma, groups, bn, out, partials, cname, wfwd, residual=None, norm=None):
        """Backward of relu(norm(xn)) followed by the dgrad of the conv `cname` that produced xn:
        -> (dx of that conv, dh = the gradient w.r.t. xn (its weight gradient needs it), epilogue moments | None, dgamma,
        dbeta).  At 16 channels the elementwise pass is fused into the dgrad (ops.conv_dgrad_normbwd; same bits)."""
        n, v, c = ops._nvc(xn)
        if partials is None:
            partials = ops._moment_partials(xn, gr, st[0], st[1])
        coef, dgam, dbet = ops.norm_bwd_coeffs(partials, n, v, c, gamma, st[2], st[3], groups, bn, out)
        # None: a layer the fused kernel does not cover (anything but the 16-channel 3x3x3 ones)
        fused = ops.conv_dgrad_normbwd(gr, xn, st[0], st[1], coef, wfwd, self._wd(cname), G3, residual=residual, norm=norm)
        if fused is not None:
            dx, dh, pt = fused
        else:
            dh = ops.norm_bwd_apply(gr, xn, st[0], st[1], coef)
            r = ops.conv_dgrad(dh, wfwd, G3, xn.shape, residual=residual, wd=self._wd(cname), norm=norm)
            dx, pt = r[0], (r[2] if norm is not None else None)
        return dx, dh, pt, dgam, dbet

    def _en_bwd(self, e, dout):
        _, name, x, h, st1, st2, w1, w2 = e
        g, P = self.grads, self.P
        g[name + ".conv2.weight"], g[name + ".conv2.bias"] = self._wg(
            ops.conv_wgrad, h, None, dout, G3, P[name + ".conv2.weight"].shape, st2[0], st2[1], out=self._o(name + ".conv2"))
        da, _, pt = ops.conv_dgrad(dout, w2, G3, h.shape, wd=self._wd(name + ".conv2"), norm=(h, st2[0], st2[1]))
        da, dh, pt, g[name + ".bn2.weight"], g[name + ".bn2.bias"] = self._norm_bwd_dgrad(
            da, h, st2, P[name + ".bn2.weight"], 8, False, self._o(name + ".bn2"), pt, name + ".conv1", w1,
            norm=(x, st1[0], st1[1]))
        g[name + ".conv1.weight"], g[name + ".conv1.bias"] = self._wg(
            ops.conv_wgrad, x, None, dh, G3, P[name + ".conv1.weight"].shape, st1[0], st1[1], out=self._o(name + ".conv1"))
        dx, g[name + ".bn1.weight"], g[name + ".bn1.bias"] = ops.norm_relu_bwd(
            da, x, st1[0], st1[1], P[name + ".bn1.weight"], st1[2], st1[3], 8, False, extra=dout,
            out=self._o(name + ".bn1"), partials=pt)
        return dx

    def _de_bwd(self, e, dout, partials=None):
        _, name, x, h1, h2, st1, st2, w1, w2 = e
        g, P = self.grads, self.P
        bnm = 1 if self.training else 2       # eval mode: BatchNorm used its running statistics (constants of x)
        da, dh2, pt, g[name + ".bn2.weight"], g[name + ".bn2.bias"] = self._norm_bwd_dgrad(
            dout, h2, st2, P[name + ".bn2.weight"], 0, bnm, self._o(name + ".bn2"), partials, name + ".conv2", w2,
            norm=(h1, st1[0], st1[1]))
        g[name + ".conv2.weight"], g[name + ".conv2.bias"] = self._wg(
            ops.conv_wgrad, h1, None, dh2, G3, P[name + ".conv2.weight"].shape, st1[0], st1[1], out=self._o(name + ".conv2"))
        dx, dh1, _, g[name + ".bn1.weight"], g[name + ".bn1.bias"] = self._norm_bwd_dgrad(
            da, h1, st1, P[name + ".bn1.weight"], 0, bnm, self._o(name + ".bn1"), pt, name + ".conv1", w1, residual=dout)
        g[name + ".conv1.weight"], g[name + ".conv1.bias"] = self._wg(
            ops.conv_wgrad, x, None, dh1, G3, P[name + ".conv1.weight"].shape, out=self._o(name + ".conv1"))
        return dx

    def _up_bwd(self, e, dy):
        g, P = self.grads, self.P
        if e[0] == "upf":
            _, name, x, prev, t, m, w1 = e
            c = m["C"]
            f = dict(dtype=torch.float32, device=dy.device)
            tmp = torch.empty((c * c * 8 + c * c + 2 * c,), **f)      # G, dW3a, s (twice: both kernels leave sum dy)
            gG, gA = tmp[:c * c * 8].view(c, c, 2, 2, 2), tmp[c * c * 8:c * c * 9].view(c, c, 1, 1, 1)
            s1, s2 = tmp[c * c * 9:c * c * 9 + c], tmp[c * c * 9 + c:]
            self._wg(ops.conv_wgrad, prev, None, dy, G1, gA.shape, out=(gA, s1))
            self._wg(ops.convT_wgrad, t, dy, (2, 2, 2), gG.shape, out=(gG, s2))
            self._wg(ops.upcat_grad_combine, gG, gA, s1, P[name + ".conv3.weight"], P[name + ".conv2.weight"],
                     P[name + ".conv2.bias"], out=self._o(name + ".conv3") + self._o(name + ".conv2"))
            for k in (".conv3.weight", ".conv3.bias", ".conv2.weight", ".conv2.bias"):
                g[name + k] = self.G[name + k]
            dprev, dt = ops.upcat_bwd_data(dy, m)     # d prev = W3a^T dy; d t = sum_p M[p]^T dy[fine(p)]
        else:
            _, name, x, prev, t, up, w1, wt, w3 = e
            g[name + ".conv3.weight"], g[name + ".conv3.bias"] = self._wg(
                ops.conv_wgrad, prev, up, dy, G1, P[name + ".conv3.weight"].shape, out=self._o(name + ".conv3"))
            dprev, dup = ops.conv_dgrad(dy, w3, G1, prev.shape, up.shape[-1], wd=self._wd(name + ".conv3"))
            g[name + ".conv2.weight"], g[name + ".conv2.bias"] = self._wg(
                ops.convT_wgrad, t, dup, (2, 2, 2), P[name + ".conv2.weight"].shape, out=self._o(name + ".conv2"))
            dt = ops.convT_dgrad(dup, wt, (2, 2, 2), t.shape, wt=self._wd(name + ".conv2"))
        g[name + ".conv1.weight"], g[name + ".conv1.bias"] = self._wg(
            ops.conv_wgrad, x, None, dt, G1, P[name + ".conv1.weight"].shape, out=self._o(name + ".conv1"))
        norm = self._next_norm()
        if norm is not None:
            dx, _, pt = ops.conv_dgrad(dt, w1, G1, x.shape, wd=self._wd(name + ".conv1"), norm=norm)
        else:
            dx, pt = ops.conv_dgrad(dt, w1, G1, x.shape, wd=self._wd(name + ".conv1"))[0], None
        return dx, dprev, pt

    def backward(self, dlogits: Tensor, need_dx: bool):
        """Mirror of forward(); returns d(input) (or None) and fills self.grads."""
        t = self.tape
        pop = t.pop
        self._begin_backward(dlogits.device)
        e = pop()
        tick = self._tick                                    # block boundaries: the clock of the weight-gradient lag
        self._lagged = True
        d, pt = self._conv_bwd(e, dlogits, norm=self._next_norm()); tick()    # endconv (+ moments for DeBlock2.bn2)
        d = self._de_bwd(pop(), d, pt); tick(); d, ds1, pt = self._up_bwd(pop(), d); tick()   # DeBlock2, DeUpCat2
        d = self._de_bwd(pop(), d, pt); tick(); d, ds2, pt = self._up_bwd(pop(), d); tick()   # DeBlock3, DeUpCat3
        d = self._de_bwd(pop(), d, pt); tick(); d, ds3, _ = self._up_bwd(pop(), d); tick()    # DeBlock4, DeUpCat4
        self._bucket_done(0)                                     # every decoder gradient is final
        for _ in range(4):
            d = self._en_bwd(pop(), d); tick()                   # EnBlock4_4..1
        self._bucket_done(1)                                     # the 128-channel stage (74 % of the parameters)
        d = self._conv_bwd(pop(), d, skip=ds3); tick()           # EnDown3 (+ skip gradient)
        d = self._en_bwd(pop(), d); tick(); d = self._en_bwd(pop(), d); tick()
        d = self._conv_bwd(pop(), d, skip=ds2); tick()           # EnDown2
        d = self._en_bwd(pop(), d); tick(); d = self._en_bwd(pop(), d); tick()
        d = self._conv_bwd(pop(), d, skip=ds1); tick()           # EnDown1
        self._wg_left = 3                                        # EnBlock1.conv2, .conv1, InitConv: the tail (_WG_TAIL_MAIN)
        d = self._en_bwd(pop(), d); tick()                       # EnBlock1
        d = self._conv_bwd(pop(), d, need_dx)                    # InitConv
        self._end_backward()
        self._bucket_done(2)
        return d

    def _begin_backward(self, device):
        self.side = _side_stream(device)
        self._keep, self._nev, self._pending, self._held, self._blk, self._last_due, self._lagged = [], 0, [], [], 0, 0, False
        self._wg_left = None
        self._joined = False
        self._wg_seq, self._wg_issued, self._bucket_wait = 0, 0, []
        if self.side is not None:
            self._main = torch.cuda.current_stream()
            self._events = _side_events(device)
            for v in self.G.values():             # views of one flat buffer: recording one records the block
                v.record_stream(self.side)
                break

    def _end_backward(self):
        if self.side is not None:
            self._tick(final=True)
            self._flush_wg()
            if self.sync is not None:
                self._launch_ready_buckets()
            _plan.call(self._main.wait_stream, self.side)
            self._joined = True
        self._keep = []                           # freed behind the join, on the stream that allocated them

    def _bucket_done(self, i):
        if self.sync is not None:
            if self.side is not None and not self._joined:
                # The bucket's weight gradients come from the side stream, which runs behind the main one, and some of
                # them are still held back by the hand-over lag (_wg).  The collective is therefore launched FROM the
                # side stream (RCCL's stream waits for the stream that is current at the call) once the last of the
                # bucket's weight-gradient calls has been handed over; the side stream first waits for the main
                # stream's part of the bucket (norm gradients), and the main stream waits for nobody.  (Joining the
                # side stream into the main one here stalled backward until the side queue had drained, and flushing
                # the held calls here gave up what the lag buys: DESIGN 5.)
                self._bucket_wait.append((i, self._wg_seq))
                self._launch_ready_buckets()
            else:
                _plan.call(self.sync.bucket_ready, i)

    def _launch_ready_buckets(self):
        while self._bucket_wait and self._bucket_wait[0][1] <= self._wg_issued:
            i, _ = self._bucket_wait.pop(0)
            ev = self._events[self._nev % len(self._events)]
            self._nev += 1
            _plan.call(ev.record, self._main)
            _plan.call(self.side.wait_event, ev)
            _plan.call(self._bucket_from_side, i)

    def _bucket_from_side(self, i):
        with torch.cuda.stream(self.side):
            self.sync.bucket_ready(i)


_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(device):
    """Per-device stream for the weight-gradient kernels (DEVIS_WGRAD_STREAM=0 keeps everything on one stream)."""
    import os
    if device.type != "cuda" or os.environ.get("DEVIS_WGRAD_STREAM", "1") == "0":
        return None
    s = _side_streams.get(device.index)
    if s is None:
        # A high-priority stream gets a hardware queue of its own.  A normal-priority one shares the process's few
        # queues (GPU_MAX_HW_QUEUES = 4) round-robin with every other stream; once RCCL has created its streams it
        # landed on the main stream's queue and the two serialised (one-rank RCCL rehearsal: 6.31 ms/step against
        # 5.58 with priority -1; without a process group the two priorities measure the same).
        s = _side_streams[device.index] = torch.cuda.Stream(device, priority=-1)
    return s


_side_event_pool: Dict[int, list] = {}


def _side_events(device):
    """Reusable events that order the side stream behind the main one (one per weight-gradient call of a step,
    more than a backward ever has in flight)."""
    ev = _side_event_pool.get(device.index)
    if ev is None:
        with torch.cuda.device(device):
            ev = _side_event_pool[device.index] = [torch.cuda.Event() for _ in range(64)]
    return ev


class _Unet3dFn(torch.autograd.Function):
    """x (NCDHW fp32) -> channels-last logits; the only autograd node of the backbone.  A training step at a shape
    the model has seen is replayed from its launch plan (devis_amd/launch_plan.py) instead of re-running the
    schedule's Python; the first step at a shape runs eagerly and is recorded."""

    @staticmethod
    def forward(ctx, model, x, *params):
        names = model._param_names
        taping = any(ctx.needs_input_grad)
        ctx.names, ctx.model = names, model
        ctx.need_dx = ctx.needs_input_grad[1]
        plan = model._step_plan(x, params, ctx.need_dx) if taping else None
        ctx.plan, ctx.run = plan, None
        if plan is not None and plan.ready():
            ctx.plan_gen = plan.begin_step(ctx)
            plan.fwd.run(x=x.data_ptr())
            return plan.logits.detach()
        P = dict(zip(names, params))

        def eager():
            run = model._run_cls(P, model._buffers_by_name(), model.compute_dtype, model.training, taping, model._packed(P))
            model._prepare_run(run, x)
            return run, run.forward(ops.to_channels_last(x, model.compute_dtype))

        if plan is None:
            ctx.run, logits = eager()
            return logits
        try:
            with plan.recording() as rec:
                ctx.run, logits = eager()
            plan.fwd = _plan.Segment(rec.ops, {"x": x.data_ptr()})
        except BaseException:
            model._drop_plan(plan)
            raise
        plan.logits = logits.detach()
        ctx.plan_gen = plan.begin_step(ctx)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model, plan = ctx.model, ctx.plan
        dlogits = dlogits.contiguous()
        if plan is not None and plan.bwd is not None and ctx.run is None:
            # replay.  The gradients land in the plan's flat buffer, which the .grad of an earlier step may still alias
            # (gradient accumulation): its contents are set aside and put back then, and autograd gets a copy to add
            if ctx.plan_gen != plan.gen:
                raise RuntimeError("devis_amd Unet: this forward's activations were overwritten by a later forward of "
                                   "the same launch plan (backward must run before the next forward at this shape)")
            clean = all(p.grad is None for p in model.parameters())
            if model._grad_sync is not None and not clean:
                raise RuntimeError(_DP_GRAD_MSG)
            held = None if clean else plan.flat.clone()
            plan.bwd.run(dlogits=dlogits.data_ptr())
            plan.end_step()
            flat = plan.flat
            if held is not None:
                flat = plan.flat.clone()
                plan.flat.copy_(held)
            model.last_flat_grad = plan.flat
            views = model._views_of(flat)
            dx = plan.dx.clone() if plan.dx is not None else None
            return (None, dx) + tuple(views[n] for n in ctx.names)
        run = ctx.run
        if run is None or not run.taping or not run.tape:
            raise RuntimeError("devis_amd Unet: backward called twice or without a taped forward")

        def eager():
            flat, run.G = model._grad_views(dlogits.device)
            run.sync = model._grad_sync
            if run.sync is not None:
                # the reducer averages the flat buffer in place and .grad must BE that buffer: a .grad left over from an
                # earlier backward (gradient accumulation, zero_grad(set_to_none=False)) would be summed by autograd with
                # an un-reduced copy - refuse instead of training on wrong gradients
                if any(p.grad is not None for p in model.parameters()):
                    raise RuntimeError(_DP_GRAD_MSG)
                _plan.call(run.sync.begin, flat, model._bucket_bounds)
            dx_cl = run.backward(dlogits, ctx.need_dx)
            dx = ops.from_channels_last(dx_cl) if dx_cl is not None else None
            return flat, dx

        recording = plan is not None and plan.bwd is None and plan.fwd is not None and ctx.plan_gen == plan.gen
        if recording:
            try:
                with plan.recording() as rec:
                    flat, dx = eager()
                plan.bwd = _plan.Segment(rec.ops, {"dlogits": dlogits.data_ptr()})
                plan.flat, plan.dx = flat, dx
                plan.end_step()
                dx = dx.clone() if dx is not None else None
            except BaseException:
                model._drop_plan(plan)
                raise
        else:
            flat, dx = eager()
        grads = tuple(run.G[n] for n in ctx.names)
        run.G = {}
        run.tape, run.grads = [], {}
        ctx.run = None
        return (None, dx) + grads


_DP_GRAD_MSG = ("data-parallel backward needs .grad = None on every backbone parameter "
                "(call optimizer.zero_grad() / zero_grad(set_to_none=True) first); gradient "
                "accumulation is not supported with the overlapped all-reduce")


class _FromChannelsLast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        ctx.dtype = y.dtype
        return ops.from_channels_last(y)

    @staticmethod
    def backward(ctx, g):
        return ops.to_channels_last(g, ctx.dtype)


class Unet(nn.Module):
    """Drop-in for the reference ``Unet(in_channels, base_channels, num_classes)``.

    ``forward(x)`` takes and returns NCDHW fp32 like the reference; ``forward_channels_last(x)``
    returns the channels-last logits the fused evidential head consumes.
    ``compute_dtype``: torch.bfloat16 (MFMA path) or torch.float32 (exact-arithmetic path).
    """

    def __init__(self, in_channels=1, base_channels=16, num_classes=4, compute_dtype=torch.bfloat16):
        super().__init__()
        b = base_channels
        self.InitConv = _InitConv(in_channels, b)
        self.EnBlock1 = _EnBlock(b)
        self.EnDown1 = _EnDown(b, b * 2)
        self.EnBlock2_1 = _EnBlock(b * 2)
        self.EnBlock2_2 = _EnBlock(b * 2)
        self.EnDown2 = _EnDown(b * 2, b * 4)
        self.EnBlock3_1 = _EnBlock(b * 4)
        self.EnBlock3_2 = _EnBlock(b * 4)
        self.EnDown3 = _EnDown(b * 4, b * 8)
        self.EnBlock4_1 = _EnBlock(b * 8)
        self.EnBlock4_2 = _EnBlock(b * 8)
        self.EnBlock4_3 = _EnBlock(b * 8)
        self.EnBlock4_4 = _EnBlock(b * 8)
        self.DeUpCat4 = _DeUpCat(b * 8, b * 4)
        self.DeBlock4 = _DeBlock(b * 4)
        self.DeUpCat3 = _DeUpCat(b * 4, b * 2)
        self.DeBlock3 = _DeBlock(b * 2)
        self.DeUpCat2 = _DeUpCat(b * 2, b)
        self.DeBlock2 = _DeBlock(b)
        self.endconv = nn.Conv3d(b, num_classes, kernel_size=1)
        self._finish_init(compute_dtype, num_classes)

    _run_cls = None       # set below: the schedule class that executes this backbone
    # gradient buckets for the overlapped all-reduce: (first parameter prefix of bucket 1 in reverse registration
    # order, prefix every member of bucket 1 shares)
    _bucket_marks = ("EnBlock4_4.", "EnBlock4_")

    def _finish_init(self, compute_dtype, num_classes):
        """Bookkeeping shared by the 3-D backbones: parameter order, conv modules to pack, flat gradient layout."""
        self.compute_dtype = compute_dtype
        self.num_classes = num_classes
        self._param_names = [n for n, _ in self.named_parameters()]
        self._conv_modules = [(n + ".weight", m) for n, m in self.named_modules()
                              if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d))]
        self._packer = None
        # gradient layout: reverse registration order ~ the order backward completes the modules.  Bucket 0 ends
        # where the decoder does (first EnBlock4_4 parameter), bucket 1 covers the EnBlock4_* stage.
        off, layout, bounds = 0, {}, []
        plist = list(self.named_parameters())
        for n, p_ in reversed(plist):
            if n.startswith(self._bucket_marks[0]) and len(bounds) == 0:
                bounds.append(off)                   # end of bucket 0: endconv ... first decoder module
            if len(bounds) == 1 and off > bounds[0] and not n.startswith(self._bucket_marks[1]):
                bounds.append(off)                   # end of bucket 1: the deepest stage
            layout[n] = (off, p_.numel(), tuple(p_.shape))
            off += p_.numel()
        self._grad_layout, self._grad_total = layout, off
        self._bucket_bounds = [(0, bounds[0]), (bounds[0], bounds[1]), (bounds[1], off)]
        self._grad_sync = None                       # parallel.FlatGradSync installs itself here
        self.last_flat_grad = None

    def _prepare_run(self, run, x):
        """Hook: per-call state a backbone hands to its schedule (Unetdrop: the dropout noise)."""

    _plan_ok = True             # the schedule issues nothing but C-ABI launches and launch_plan.call()s
    _grad_on = True

    def _param_list(self):
        """The parameters in named_parameters() order (module traversal done once: Parameter objects keep their identity
        through .to() / load_state_dict)."""
        pl = self.__dict__.get("_param_list_cache")
        if pl is None:
            pl = self.__dict__["_param_list_cache"] = [p for _, p in self.named_parameters()]
        return pl

    def _buffers_by_name(self):
        """{name: buffer} with the CURRENT buffer tensors (module.to() replaces them), without a module traversal"""
        slots = self.__dict__.get("_buffer_slots")
        if slots is None:
            slots = self.__dict__["_buffer_slots"] = [
                (m, bn, (mn + "." if mn else "") + bn) for mn, m in self.named_modules() for bn in m._buffers]
        return {full: m._buffers[bn] for m, bn, full in slots if m._buffers[bn] is not None}

    def _views_of(self, flat):
        return {n: flat[o:o + k].view(shape) for n, (o, k, shape) in self._grad_layout.items()}

    def _step_plan(self, x, params, need_dx):
        """The launch plan of a training step at this input shape (None: run eagerly).  Keyed by everything whose
        address or value the recorded launches carry."""
        if not (_plan.ENABLED and self._plan_ok and self.training and self._grad_on) or not x.is_cuda or x.dtype != torch.float32 \
                or not x.is_contiguous() or torch.cuda.is_current_stream_capturing():
            return None
        key = (tuple(x.shape), x.device.index, bool(need_dx), self.compute_dtype,
               torch._C._cuda_getCurrentRawStream(x.device.index), id(self._grad_sync),
               tuple(p.data_ptr() for p in params), tuple(b.data_ptr() for b in self._buffers_by_name().values()))
        plans = self.__dict__.setdefault("_plans", {})
        misses = self.__dict__.setdefault("_plan_misses", {})
        plan = plans.get(key)
        if plan is not None:
            if plan.busy():
                return None              # a second forward before the first one's backward: eager, on memory of its own
            if plan.ready():
                return plan
            self._drop_plan(plan)        # recorded half-way (its backward never ran): start over, twice at most
            misses[key] = misses.get(key, 0) + 1
        if misses.get(key, 0) >= 2:
            return None                  # forwards without backwards at this shape: not a training loop
        while len(plans) >= 2:           # each plan owns a step's worth of memory
            old = next((k for k, p in plans.items() if not p.busy()), None)
            if old is None:
                return None
            del plans[old]
        plan = plans[key] = _plan.StepPlan(x.device)
        return plan

    def _drop_plan(self, plan):
        plans = self.__dict__.get("_plans", {})
        for k in [k for k, p in plans.items() if p is plan]:
            del plans[k]

    def _grad_views(self, device):
        """One flat fp32 buffer per backward holding every parameter gradient, laid out in the order backward
        produces them (decoder first), plus {name: view}.  The kernels write into the views, autograd hands
        them out as .grad, and the data-parallel reducer all-reduces slices of the flat buffer in place -
        no gather, no copy-back.  `_bucket_bounds` = element ranges of the three overlap buckets."""
        flat = torch.empty((self._grad_total,), dtype=torch.float32, device=device)
        self.last_flat_grad = flat
        return flat, self._views_of(flat)

    def _packed(self, P: Dict[str, Tensor]):
        """All conv weights in the kernels' layouts, one launch (rebuilt tables only if a parameter moved)."""
        ent = [(n, P[n].detach(), "convT" if isinstance(m, nn.ConvTranspose3d) else "conv")
               for n, m in self._conv_modules]
        key = tuple(w.data_ptr() for _, w, _ in ent) + (self.compute_dtype,)
        if self._packer is None or self._packer.key != key:
            self._packer = ops.WeightPacker(ent, self.compute_dtype, ent[0][1].device)
            self._pack_frozen = False
        if getattr(self, "_pack_frozen", False):
            return self._packer.views                # frozen_weights(): packed once, read-only until it exits
        return self._packer.pack()

    def frozen_weights(self):
        """Context manager for several forwards on UNCHANGED parameters (the patches of one volume, possibly on
        several streams or replayed from HIP graphs): the weights are packed once on the current stream on entry
        and every forward inside reads that packing instead of rewriting it."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self._pack_frozen = False
            self._packed(dict(self.named_parameters()))
            self._pack_frozen = True
            try:
                yield self
            finally:
                self._pack_frozen = False
        return cm()

    def forward_channels_last(self, x: Tensor) -> Tensor:
        first = self._conv_modules[0][1]                     # the network's first conv
        if x.dim() != 5 or x.shape[1] != first.in_channels:
            raise ValueError("%s expects (N, %d, D, H, W), got %s" % (type(self).__name__, first.in_channels, tuple(x.shape)))
        if any(s % 8 for s in x.shape[2:]):
            raise ValueError("Unet needs D, H, W divisible by 8 (three stride-2 stages), got %s" % (tuple(x.shape[2:]),))
        ops.require_device(x, first.weight)
        self._grad_on = torch.is_grad_enabled()              # (inside the autograd node grad mode is always off)
        return _Unet3dFn.apply(self, x, *self._param_list())

    def forward(self, x: Tensor) -> Tensor:
        return _FromChannelsLast.apply(self.forward_channels_last(x))


Unet._run_cls = _Run


# ------------------------------------------------------------ Unetdrop (UNet3DZoo.py:308-372)
class _RunDrop(_Run):
    """Schedule of ``Unetdrop``: the U-Net with channel dropout (p = 0.2, ALWAYS on: ``F.dropout3d`` is called with its
    default training=True, UNet3DZoo.py:30) after InitConv and element dropout (p = 0.5, training mode only) after
    EnBlock3_2, EnBlock4_3, DeUpCat4 and DeUpCat3 (:352-366); the skip into DeUpCat4 is EnBlock3_1's output (:350,:361).
    Dropout is ``input * noise`` as ATen evaluates it; the noise tensors come in through ``self.masks``."""

    masks = None

    def _mask(self, h, key):
        m = self.masks.get(key)
        if m is None:
            return h
        self._push("mul", m)
        return ops.mul(h, m)

    def _mask_bwd(self, d):
        if self.tape and self.tape[-1][0] == "mul":
            d = ops.mul(d, self.tape.pop()[1])
        return d

    def forward(self, x: Tensor) -> Tensor:
        h = self.conv("InitConv.conv", x, G3)
        mi = self.masks["init"]
        h = ops.affine_act(h, mi, torch.zeros_like(mi), relu=False)       # per-(n, c) noise: moments are taken afterwards
        s1, _ = self.en_block("EnBlock1", h)
        h, hs = self.conv("EnDown1.conv", s1, G3S2, True)
        h, hs = self.en_block("EnBlock2_1", h, hs, True)
        s2, _ = self.en_block("EnBlock2_2", h, hs)
        h, hs = self.conv("EnDown2.conv", s2, G3S2, True)
        s3, hs = self.en_block("EnBlock3_1", h, hs, True)
        h, _ = self.en_block("EnBlock3_2", s3, hs)
        h, hs = self.conv("EnDown3.conv", self._mask(h, "d1"), G3S2, True)
        for i in (1, 2, 3):
            h, hs = self.en_block("EnBlock4_%d" % i, h, hs, i < 3 or self.masks.get("d2") is None)
        h, _ = self.en_block("EnBlock4_4", self._mask(h, "d2"), hs if self.masks.get("d2") is None else None)
        h = self.de_block("DeBlock4", self._mask(self.de_up_cat("DeUpCat4", h, s3), "u1"))
        h = self.de_block("DeBlock3", self._mask(self.de_up_cat("DeUpCat3", h, s2), "u2"))
        h = self.de_block("DeBlock2", self.de_up_cat("DeUpCat2", h, s1))
        return self.conv("endconv", h, G1)

    def backward(self, dlogits: Tensor, need_dx: bool):
        pop = self.tape.pop
        self._begin_backward(dlogits.device)
        d, pt = self._conv_bwd(pop(), dlogits, norm=self._next_norm())           # endconv
        d = self._de_bwd(pop(), d, pt); d, ds1, pt = self._up_bwd(pop(), d)      # DeBlock2, DeUpCat2
        d = self._mask_bwd(self._de_bwd(pop(), d, pt)); d, ds2, pt = self._up_bwd(pop(), d)   # DeBlock3, u2, DeUpCat3
        d = self._mask_bwd(self._de_bwd(pop(), d, pt)); d, ds3, _ = self._up_bwd(pop(), d)    # DeBlock4, u1, DeUpCat4
        self._bucket_done(0)
        d = self._mask_bwd(self._en_bwd(pop(), d))                               # EnBlock4_4, d2
        for _ in range(3):
            d = self._en_bwd(pop(), d)                                           # EnBlock4_3..1
        self._bucket_done(1)
        d = self._mask_bwd(self._conv_bwd(pop(), d))                             # EnDown3, d1
        d = ops.add(self._en_bwd(pop(), d), ds3)                                 # EnBlock3_2; + the skip gradient
        d = self._en_bwd(pop(), d)                                               # EnBlock3_1
        d = self._conv_bwd(pop(), d, skip=ds2)                                   # EnDown2
        d = self._en_bwd(pop(), d); d = self._en_bwd(pop(), d)
        d = self._conv_bwd(pop(), d, skip=ds1)                                   # EnDown1
        d = self._en_bwd(pop(), d)                                               # EnBlock1
        mi = self.masks["init"]
        d = ops.affine_act(d, mi, torch.zeros_like(mi), relu=False)
        d = self._conv_bwd(pop(), d, need_dx)                                    # InitConv
        self._end_backward()
        self._bucket_done(2)
        return d


class Unetdrop(Unet):
    """Drop-in for the reference ``Unetdrop(in_channels, base_channels, num_classes)`` (UNet3DZoo.py:308-372): same
    parameters and ``state_dict`` keys as ``Unet``.  ``forward(x)`` draws the dropout noise itself (torch's generator
    of x's device, bernoulli(1 - p) / (1 - p) like ATen's dropout); ``forward(x, masks=...)`` takes it from the
    caller instead: {'init': (N, C, 1, 1, 1), 'd1', 'd2', 'u1', 'u2': NCDHW or None} - the hook the parity tests use
    (oracle/devis_oracle.py:unetdrop_masks draws the reference's own noise)."""

    _run_cls = _RunDrop
    _masks_next = None
    _plan_ok = False            # the noise is drawn by torch each step: nothing to replay

    def forward(self, x: Tensor, masks=None) -> Tensor:
        self._masks_next = masks
        try:
            return super().forward(x)
        finally:
            self._masks_next = None

    def forward_channels_last(self, x: Tensor, masks=None) -> Tensor:
        if masks is not None:
            self._masks_next = masks
        try:
            return super().forward_channels_last(x)
        finally:
            self._masks_next = None

    def _prepare_run(self, run, x):
        n, _, d, h, w = x.shape
        b = self.InitConv.conv.out_channels
        dev, dt = x.device, self.compute_dtype
        shapes = {"d1": (n, d // 4, h // 4, w // 4, b * 4), "d2": (n, d // 8, h // 8, w // 8, b * 8),
                  "u1": (n, d // 4, h // 4, w // 4, b * 4), "u2": (n, d // 2, h // 2, w // 2, b * 2)}
        given = self._masks_next
        m = {}
        if given is None:
            m["init"] = torch.empty((n, b), dtype=torch.float32, device=dev).bernoulli_(0.8).div_(0.8)
            for k, shp in shapes.items():
                m[k] = torch.empty(shp, dtype=dt, device=dev).bernoulli_(0.5).mul_(2.0) if self.training else None
        else:
            m["init"] = given["init"].reshape(n, b).to(device=dev, dtype=torch.float32).contiguous()
            for k, shp in shapes.items():
                g = given.get(k)
                m[k] = None if g is None else ops.to_channels_last(g.to(dev), dt)
                if m[k] is not None and tuple(m[k].shape) != shp:
                    raise ValueError("Unetdrop mask %r: expected NCDHW of %s" % (k, (shp[0], shp[4]) + shp[1:4]))
        run.masks = m


Unetdrop._run_cls = _RunDrop
