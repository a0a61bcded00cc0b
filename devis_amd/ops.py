"""Thin, shape-checked wrappers: torch tensors in/out, HIP kernels underneath.

Activations are channels-last tensors ``[N, D, H, W, C]`` (3-D) or ``[N, H, W, C]``
(2-D), dtype float32 or bfloat16.  torch is used for memory (``torch.empty``) and the
current stream only; every computation is a libdevis_hip.so call.  None of these
functions is differentiable by autograd - the manual backward lives in
``devis_amd/unet3d.py`` / ``unet2d.py`` / ``criterions.py``.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, ConvTDesc, check

Tensor = torch.Tensor


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


_tls = threading.local()     # .override: raw stream handle that replaces torch's current stream (see stream_override)


def _stream() -> int:
    h = getattr(_tls, "override", None)
    if h is not None:
        return h
    # raw handle of torch's current stream on the current device; torch.cuda.current_stream() builds a Stream object
    # through several Python layers (about 5 us a call, ~200 calls a step)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class stream_override:
    """Launch the enclosed ops calls on `stream` (a torch.cuda.Stream) without switching torch's current stream:
    cheaper than `with torch.cuda.stream(...)` for calls that allocate nothing (outputs and workspace given).
    The override is per THREAD (forward runs on the caller's thread, backward on the autograd engine's)."""

    def __init__(self, stream):
        self.handle = stream.cuda_stream

    def __enter__(self):
        self.prev = getattr(_tls, "override", None)
        _tls.override = self.handle

    def __exit__(self, *exc):
        _tls.override = self.prev
        return False


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return _lib.DEVIS_F32
    if dtype == torch.bfloat16:
        return _lib.DEVIS_BF16
    raise _lib.DevisError("unsupported compute dtype %s (float32 or bfloat16)" % dtype)


def _need_cuda(*ts: Optional[Tensor]) -> None:
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise _lib.DevisError("devis_amd kernels need GPU tensors (got a %s tensor); "
                                      "there is no CPU fallback" % t.device)
            if not t.is_contiguous():
                raise _lib.DevisError("devis_amd kernels need contiguous tensors")


def require_device(*ts: Optional[Tensor]) -> None:
    """Model-level guard: inputs and parameters must be on the GPU (no CPU path exists)."""
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DevisError("devis_amd kernels need GPU tensors (got a %s tensor); "
                                  "there is no CPU fallback" % t.device)


def _spatial(x: Tensor) -> Tuple[int, int, int]:
    if x.dim() == 5:
        return x.shape[1], x.shape[2], x.shape[3]
    if x.dim() == 4:
        return 1, x.shape[1], x.shape[2]
    raise _lib.DevisError("expected a channels-last 4-D or 5-D tensor, got %d-D" % x.dim())


_ws = {}
_ws_retired = []


def _persistent(make, filled_on=None):
    """Run an allocation whose result OUTLIVES the step (workspaces, constant tables) so that it does not land in a
    launch plan's memory pool.  While a step is being recorded the thread's allocations come from the plan's private
    pool (launch_plan.StepPlan.recording), whose addresses the replay hands out again in the recorded order: a buffer
    allocated there at step time t sits where some tensor of the step lived BEFORE t, and a user of the buffer that
    runs earlier in a later step (the loss head's workspace between forward and backward, another model) would
    overwrite that tensor.  torch routes only the recording thread's allocations to the pool (torch.cuda.use_mem_pool),
    so the allocation is made on a helper thread; it is rare (first use, growth).  `filled_on` = the device to
    synchronise when `make` also launches fills (they run on the helper thread's stream)."""
    if getattr(_lib._rec_tls, "rec", None) is None:
        return make()
    import threading
    box = []

    def run():
        try:
            out = make()
            if filled_on is not None:
                torch.cuda.synchronize(filled_on)
            box.append((out, None))
        except BaseException as e:          # re-raised on the calling thread
            box.append((None, e))

    th = threading.Thread(target=run)
    th.start()
    th.join()
    out, err = box[0]
    if err is not None:
        raise err
    return out


def workspace(nbytes: int, device: torch.device, slot: str = "ws") -> Tensor:
    """Scratch that only ever grows (the library itself never allocates), one buffer per (device, slot, STREAM the
    calls launch on): kernels of different streams run concurrently and must not share it.  A buffer that is
    outgrown is retired, not freed - a recorded launch plan or a captured HIP graph may still hold its address.
    Never allocated from a launch plan's pool (_persistent)."""
    key = (device.index, slot, _stream())
    t = _ws.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            _ws_retired.append(t)
        size = max(nbytes, 1 << 20)
        t = _persistent(lambda: torch.empty(size, dtype=torch.uint8, device=device))
        _ws[key] = t
    return t


# ------------------------------------------------------------------ layout
def to_channels_last(x: Tensor, dtype: torch.dtype) -> Tensor:
    """NC(D)HW fp32 -> channels-last `dtype`."""
    x = x.contiguous().float()
    _need_cuda(x)
    n, c = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    v = 1
    for s in sp:
        v *= s
    y = torch.empty((n,) + sp + (c,), dtype=dtype, device=x.device)
    check(_lib.load().devis_ncx_to_nxc(_p(x), _p(y), n, c, v, dt_code(dtype), _stream()), "ncx_to_nxc")
    return y


def from_channels_last(y: Tensor) -> Tensor:
    """channels-last -> NC(D)HW fp32."""
    _need_cuda(y)
    n, c = y.shape[0], y.shape[-1]
    sp = tuple(y.shape[1:-1])
    v = 1
    for s in sp:
        v *= s
    x = torch.empty((n, c) + sp, dtype=torch.float32, device=y.device)
    check(_lib.load().devis_nxc_to_ncx(_p(y), _p(x), n, c, v, dt_code(y.dtype), _stream()), "nxc_to_ncx")
    return x


def pack_conv_weight(w: Tensor, dtype: torch.dtype) -> Tensor:
    """[Cout, Cin, *k] fp32 -> [Cout, T, Cin] `dtype`."""
    w = w.detach().contiguous()
    _need_cuda(w)
    co, ci = w.shape[0], w.shape[1]
    t = w.numel() // (co * ci)
    out = torch.empty((co, t, ci), dtype=dtype, device=w.device)
    check(_lib.load().devis_pack_conv_weight(_p(w), _p(out), co, ci, t, dt_code(dtype), _stream()), "pack_conv_weight")
    return out


def pack_convT_weight(w: Tensor, dtype: torch.dtype) -> Tensor:
    """[Cin, Cout, *k] fp32 -> [T, Cout, Cin] `dtype`."""
    w = w.detach().contiguous()
    _need_cuda(w)
    ci, co = w.shape[0], w.shape[1]
    t = w.numel() // (co * ci)
    out = torch.empty((t, co, ci), dtype=dtype, device=w.device)
    check(_lib.load().devis_pack_convT_weight(_p(w), _p(out), ci, co, t, dt_code(dtype), _stream()), "pack_convT_weight")
    return out


class WeightPacker:
    """Every conv / transposed-conv weight of a model re-packed by ONE launch per step
    (devis_pack_weights_batch): forward layout [Cout][T][Cin] and, in bf16, the dgrad layouts the MFMA
    kernels take.  The device tables are built once and stay valid while the parameters keep their addresses."""

    def __init__(self, entries, dtype: torch.dtype, device: torch.device):
        # entries: [(name, fp32 weight tensor, "conv" | "convT")]
        import numpy as np
        self.key = tuple(w.data_ptr() for _, w, _ in entries) + (dtype,)
        want_d = dtype in (torch.bfloat16, torch.float32)      # fp32: the flipped layout of csrc/conv_f32_mfma.hip's dgrad
        # Row chunks exist from library version 101 on.  An older build given one indexes its item table with the flag
        # bit set - a GPU memory fault (it happened: a same-box A/B with DEVIS_HIP_LIB pointing at the r02 build).
        rows_ok = int(_lib.load().devis_version()) >= 101
        plan, off = [], 0
        for name, w, kind in entries:
            if w.dtype != torch.float32 or not w.is_contiguous():
                raise _lib.DevisError("WeightPacker needs contiguous fp32 parameters (%s)" % name)
            a, b = int(w.shape[0]), int(w.shape[1])
            t = w.numel() // (a * b)
            n = a * b * t
            if kind == "conv":          # w [Cout][Cin][T]
                shapes = [(_lib.PACK_CONV, (a, t, b))] + ([(_lib.PACK_CONV_DGRAD, (b, t, a))] if want_d else [])
            else:                       # w [Cin][Cout][T]
                shapes = [(_lib.PACK_CONVT, (t, b, a))] + ([(_lib.PACK_CONVT_DGRAD, (a, t * b))] if want_d else [])
            for mode, shape in shapes:
                plan.append((name, w, a, b, t, mode, shape, off, n))
                off += (n + 7) // 8 * 8         # 16-byte aligned views
        self.arena = torch.empty((max(off, 8),), dtype=dtype, device=device)
        es = self.arena.element_size()
        items = np.zeros(len(plan), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"),
                                                    ("taps", "<i4"), ("mode", "<i4")]))
        chunks = []
        self.views = {}
        for i, (name, w, a, b, t, mode, shape, o, n) in enumerate(plan):
            items[i] = (w.data_ptr(), self.arena.data_ptr() + o * es, a, b, t, mode)
            # conv modes: one block per output row (coalesced through LDS, csrc/layout.hip) where a row fits
            row_len = (b if mode == _lib.PACK_CONV else a) * t
            if rows_ok and mode in (_lib.PACK_CONV, _lib.PACK_CONV_DGRAD) and t > 1 and row_len <= _lib.PACK_ROW_MAX:
                chunks += [(i | _lib.PACK_ROW_FLAG, r) for r in range(a if mode == _lib.PACK_CONV else b)]
            else:
                chunks += [(i, st) for st in range(0, n, _lib.PACK_CHUNK)]
            view = self.arena[o:o + n].view(shape)
            slot = self.views.setdefault(name, [None, None])
            slot[0 if mode in (_lib.PACK_CONV, _lib.PACK_CONVT) else 1] = view
        self.items = torch.from_numpy(items.view(np.uint8).copy()).to(device)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).to(device)
        self.n_chunks = len(chunks)
        self.dtype = dtype

    def pack(self):
        """-> {name: [w_packed, w_dgrad | None]} for the CURRENT parameter values."""
        check(_lib.load().devis_pack_weights_batch(_p(self.items), _p(self.chunks), self.n_chunks, dt_code(self.dtype),
                                                   _stream()), "pack_weights_batch")
        return self.views


# -------------------------------------------------------------------- conv
class ConvGeom:
    """kernel / stride / padding of one conv layer, each a (d, h, w) triple."""

    def __init__(self, kernel: Sequence[int], stride: Sequence[int], pad: Sequence[int]):
        def tri(v):
            v = tuple(int(a) for a in v)
            return (1,) * (3 - len(v)) + v if len(v) < 3 else v
        self.k, self.s = tri(kernel), tri(stride)
        p = tuple(int(a) for a in pad)
        self.p = (0,) * (3 - len(p)) + p if len(p) < 3 else p
        self._descs = {}

    def out_spatial(self, isp: Tuple[int, int, int]) -> Tuple[int, int, int]:
        return tuple((i + 2 * p - k) // s + 1 for i, k, s, p in zip(isp, self.k, self.s, self.p))

    def desc(self, n: int, isp, c1: int, c2: int, cout: int, dtype: torch.dtype) -> ConvDesc:
        """The C descriptor of this layer at this shape (cached: callers treat it as read-only)."""
        key = (n, tuple(isp), c1, c2, cout, dtype)
        d = self._descs.get(key)
        if d is None:
            osp = self.out_spatial(isp)
            d = ConvDesc(n, isp[0], isp[1], isp[2], osp[0], osp[1], osp[2], c1, c2, cout,
                         self.k[0], self.k[1], self.k[2], self.s[0], self.s[1], self.s[2],
                         self.p[0], self.p[1], self.p[2], dt_code(dtype))
            self._descs[key] = d
        return d


def _out_shape(x: Tensor, osp, cout: int):
    return (x.shape[0],) + (tuple(osp) if x.dim() == 5 else tuple(osp[1:])) + (cout,)


def _stats_request(lib, d, dgrad: int, mode: int, pro: bool, res: bool, n: int, c: int, device, xref=None, sc=None, sh=None):
    """-> (ConvStats | None, (partials, chunks) | None): a statistics request the kernel for this shape honours."""
    chunks = lib.devis_conv_stats_chunks(C.byref(d), dgrad, mode, int(pro), int(res))
    if chunks <= 0:
        return None, None
    part = torch.empty((n, chunks, c, 2), dtype=torch.float64, device=device)
    st = _lib.ConvStats(mode, 0, part.data_ptr(), part.numel() * 8, chunks, _p(xref), _p(sc), _p(sh))
    return st, (part, chunks)


def conv_fwd(x1: Tensor, x2: Optional[Tensor], wp: Tensor, bias: Optional[Tensor], geom: ConvGeom,
             scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
             residual: Optional[Tensor] = None, want_stats: bool = False):
    """-> y, or (y, partials | None) with want_stats: the per-chunk moments of y gathered in the conv's epilogue
    (pass them to gn_stats / bn_stats); None when the kernel for this shape cannot gather them."""
    _need_cuda(x1, x2, wp, bias, scale, shift, residual)
    isp = _spatial(x1)
    c1 = x1.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    cout = wp.shape[0]
    if wp.shape[2] != c1 + c2 or wp.shape[1] != geom.k[0] * geom.k[1] * geom.k[2]:
        raise _lib.DevisError("packed weight %s does not match input channels %d+%d / kernel %s"
                              % (tuple(wp.shape), c1, c2, geom.k))
    if x2 is not None and x2.shape[:-1] != x1.shape[:-1]:
        raise _lib.DevisError("virtual-concat inputs differ in shape")
    if wp.dtype != x1.dtype:
        raise _lib.DevisError("weight dtype %s != activation dtype %s" % (wp.dtype, x1.dtype))
    d = geom.desc(x1.shape[0], isp, c1, c2, cout, x1.dtype)
    y = torch.empty(_out_shape(x1, (d.OD, d.OH, d.OW), cout), dtype=x1.dtype, device=x1.device)
    if residual is not None and residual.shape != y.shape:
        raise _lib.DevisError("residual shape mismatch")
    if scale is not None and scale.numel() != x1.shape[0] * (c1 + c2):
        raise _lib.DevisError("prologue scale must be [N, Cin]")
    lib = _lib.load()
    if want_stats:
        st, part = _stats_request(lib, d, 0, 1, scale is not None, residual is not None, x1.shape[0], cout, x1.device)
        if st is not None:
            check(lib.devis_conv_fwd_stats(C.byref(d), _p(x1), _p(x2), _p(wp), _p(bias), _p(scale), _p(shift),
                                           _p(residual), _p(y), C.byref(st), _stream()), "conv_fwd_stats")
            return y, part
    check(lib.devis_conv_fwd(C.byref(d), _p(x1), _p(x2), _p(wp), _p(bias), _p(scale), _p(shift),
                             _p(residual), _p(y), _stream()), "conv_fwd")
    return (y, None) if want_stats else y


def conv_dgrad(dy: Tensor, wp: Tensor, geom: ConvGeom, in_shape1, c2: int = 0, residual: Optional[Tensor] = None,
               wd: Optional[Tensor] = None, norm=None):
    """-> (dx1, dx2|None): gradient w.r.t. the conv's (post-prologue) inputs.  `wd`: the dgrad packing
    of the weight when the caller already has it (WeightPacker), else it is derived from `wp` here.
    `norm` = (x, scale, shift) of the norm+ReLU whose output fed this conv: returns (dx1, dx2, partials | None)
    with the norm-backward moments gathered in the epilogue (pass them to norm_relu_bwd)."""
    _need_cuda(dy, wp)
    n = in_shape1[0]
    c1 = in_shape1[-1]
    isp = tuple(in_shape1[1:-1]) if len(in_shape1) == 5 else (1,) + tuple(in_shape1[1:-1])
    d = geom.desc(n, isp, c1, c2, wp.shape[0], dy.dtype)
    if _spatial(dy) != (d.OD, d.OH, d.OW) or dy.shape[-1] != wp.shape[0]:
        raise _lib.DevisError("dgrad: dy shape %s inconsistent with the layer" % (tuple(dy.shape),))
    dx1 = torch.empty(tuple(in_shape1), dtype=dy.dtype, device=dy.device)
    dx2 = torch.empty(tuple(in_shape1[:-1]) + (c2,), dtype=dy.dtype, device=dy.device) if c2 else None
    lib = _lib.load()
    # fp32 (the parity mode): csrc/conv_f32_mfma.hip runs every dgrad as a forward conv on the flipped weights
    if wd is None and (dy.dtype == torch.float32 or dy.dtype == torch.bfloat16 and (
            (c2 == 0 and geom.k == (3, 3, 3) and geom.s in ((1, 1, 1), (2, 2, 2)))
            or geom.k == (1, 1, 1)
            or (c2 == 0 and geom.k == (2, 2, 2) and geom.s == (2, 2, 2))
            or (geom.k == (1, 3, 3) and geom.s == (1, 1, 1)))):
        # tap-flipped [Cin][T][Cout] packing: dgrad runs on the MFMA forward kernel
        wd = torch.empty((wp.shape[2], wp.shape[1], wp.shape[0]), dtype=wp.dtype, device=wp.device)
        check(lib.devis_repack_conv_weight_dgrad(_p(wp), _p(wd), wp.shape[0], wp.shape[2], wp.shape[1],
                                                 dt_code(wp.dtype), _stream()), "repack_conv_weight_dgrad")
    if residual is not None and (c2 or residual.shape != dx1.shape):
        raise _lib.DevisError("dgrad residual needs C2 == 0 and the input's shape")
    if norm is not None:
        xref, sc, sh = norm
        _need_cuda(xref, sc, sh)
        if xref.shape != dx1.shape or xref.dtype != dx1.dtype:
            raise _lib.DevisError("dgrad statistics: the normalised tensor must have the input's shape and dtype")
        st, part = (None, None) if c2 else _stats_request(lib, d, 1, 2, False, residual is not None, n, c1, dy.device,
                                                          xref, sc, sh)
        if st is not None:
            check(lib.devis_conv_dgrad_stats(C.byref(d), _p(dy), _p(wp), _p(wd), _p(residual), _p(dx1), _p(dx2),
                                             C.byref(st), _stream()), "conv_dgrad_stats")
            return dx1, dx2, part
    check(lib.devis_conv_dgrad(C.byref(d), _p(dy), _p(wp), _p(wd), _p(residual), _p(dx1), _p(dx2), _stream()), "conv_dgrad")
    return (dx1, dx2, None) if norm is not None else (dx1, dx2)


def conv_dgrad_normbwd(g: Tensor, xn: Tensor, scale: Tensor, shift: Tensor, coef, wp: Tensor, wd: Tensor, geom: ConvGeom,
                       residual: Optional[Tensor] = None, norm=None, want_dh: bool = True):
    """norm_bwd_apply fused into the dgrad that consumes it (devis_conv_dgrad_normbwd): the gradient entering this
    conv's dgrad is dh = cA*g*[scale*xn+shift>0] + cB*xn + cC with coef = (cA, cB, cC) of norm_bwd_coeffs.
    -> (dx, dh | None, partials | None) - the bits norm_bwd_apply + conv_dgrad give - or None when the kernel does not
    take the layer (the caller runs the two calls)."""
    if (g.dtype != torch.bfloat16 or g.dim() != 5 or g.shape[-1] != 16 or wp.shape[0] != 16 or wp.shape[2] != 16 or wd is None
            or geom.k != (3, 3, 3) or geom.s != (1, 1, 1) or geom.p != (1, 1, 1) or g.shape[3] <= 16 or xn.shape != g.shape
            or xn.dtype != g.dtype or (residual is not None and norm is not None)):
        return None
    _need_cuda(g, xn, scale, shift, coef[0], coef[1], coef[2], wp, wd, residual)
    n = g.shape[0]
    d = geom.desc(n, tuple(g.shape[1:-1]), 16, 0, 16, g.dtype)
    lib = _lib.load()
    st, part = None, None
    if norm is not None:
        xref, sc, sh = norm
        _need_cuda(xref, sc, sh)
        if xref.shape != g.shape or xref.dtype != g.dtype:
            raise _lib.DevisError("dgrad statistics: the normalised tensor must have the input's shape and dtype")
        st, part = _stats_request(lib, d, 1, 2, False, False, n, 16, g.device, xref, sc, sh)
        if st is None:
            return None
    if residual is not None and residual.shape != g.shape:
        raise _lib.DevisError("dgrad residual needs the input's shape")
    dx = torch.empty_like(g)
    dh = torch.empty_like(g) if want_dh else None
    rc = lib.devis_conv_dgrad_normbwd(C.byref(d), _p(g), _p(xn), _p(scale), _p(shift), _p(coef[0]), _p(coef[1]), _p(coef[2]),
                                      _p(wd), _p(residual), _p(dx), _p(dh), C.byref(st) if st is not None else None, _stream())
    if rc == -2:                                   # unsupported shape: the caller runs the two calls
        return None
    check(rc, "conv_dgrad_normbwd")
    return dx, dh, part


def conv_wgrad(x1: Tensor, x2: Optional[Tensor], dy: Tensor, geom: ConvGeom, wshape,
               scale: Optional[Tensor] = None, shift: Optional[Tensor] = None, want_bias: bool = True, out=None,
               slot: str = "ws"):
    """-> (dw in the PyTorch layout `wshape` fp32, dbias fp32|None).  `out` = (dw, dbias) buffers to fill
    (views of a flat gradient bucket) instead of fresh tensors."""
    _need_cuda(x1, x2, dy, scale, shift)
    c1 = x1.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    cout = dy.shape[-1]
    d = geom.desc(x1.shape[0], _spatial(x1), c1, c2, cout, x1.dtype)
    if _spatial(dy) != (d.OD, d.OH, d.OW):
        raise _lib.DevisError("wgrad: dy shape inconsistent with the layer")
    dw, db = out if out is not None else (torch.empty(tuple(wshape), dtype=torch.float32, device=x1.device),
                                          torch.empty((cout,), dtype=torch.float32, device=x1.device) if want_bias else None)
    if dw.numel() != cout * (c1 + c2) * geom.k[0] * geom.k[1] * geom.k[2] or tuple(dw.shape) != tuple(wshape):
        raise _lib.DevisError("wgrad: weight shape mismatch")
    _check_out(dw, db)
    lib = _lib.load()
    nb = lib.devis_conv_wgrad_workspace_bytes(C.byref(d))
    ws = workspace(nb, x1.device, slot) if nb else None
    check(lib.devis_conv_wgrad(C.byref(d), _p(x1), _p(x2), _p(scale), _p(shift), _p(dy), _p(dw), _p(db),
                               _p(ws), nb, _stream()), "conv_wgrad")
    return dw, db


def _tdesc(x_shape, cin: int, cout: int, k, dtype) -> ConvTDesc:
    isp = tuple(x_shape[1:-1]) if len(x_shape) == 5 else (1,) + tuple(x_shape[1:-1])
    k = tuple(k)
    k = (1,) * (3 - len(k)) + k
    return ConvTDesc(x_shape[0], isp[0], isp[1], isp[2], cin, cout, k[0], k[1], k[2], dt_code(dtype))


def convT_fwd(x: Tensor, wp: Tensor, bias: Optional[Tensor], k) -> Tensor:
    _need_cuda(x, wp, bias)
    cout, cin = wp.shape[1], wp.shape[2]
    if cin != x.shape[-1]:
        raise _lib.DevisError("convT: channel mismatch")
    d = _tdesc(x.shape, cin, cout, k, x.dtype)
    osp = (d.ID * d.KD, d.IH * d.KH, d.IW * d.KW)
    y = torch.empty(_out_shape(x, osp, cout), dtype=x.dtype, device=x.device)
    check(_lib.load().devis_convT_fwd(C.byref(d), _p(x), _p(wp), _p(bias), _p(y), _stream()), "convT_fwd")
    return y


def convT_dgrad(dy: Tensor, wp: Tensor, k, x_shape, wt: Optional[Tensor] = None) -> Tensor:
    _need_cuda(dy, wp)
    d = _tdesc(x_shape, wp.shape[2], wp.shape[1], k, dy.dtype)
    dx = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device)
    lib = _lib.load()
    if wt is None and dy.dtype in (torch.bfloat16, torch.float32):
        t, co, ci = wp.shape
        wt = torch.empty((ci, t * co), dtype=wp.dtype, device=wp.device)
        check(lib.devis_repack_conv_weight_dgrad(_p(wp), _p(wt), t * co, ci, 1, dt_code(wp.dtype), _stream()),
              "repack convT weight")
    check(lib.devis_convT_dgrad(C.byref(d), _p(dy), _p(wp), _p(wt), _p(dx), _stream()), "convT_dgrad")
    return dx


def _check_out(*ts) -> None:
    for t in ts:
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda):
            raise _lib.DevisError("gradient output buffers must be contiguous fp32 device tensors")


def convT_wgrad(x: Tensor, dy: Tensor, k, wshape, out=None, slot: str = "ws"):
    _need_cuda(x, dy)
    cin, cout = wshape[0], wshape[1]
    d = _tdesc(x.shape, cin, cout, k, x.dtype)
    dw, db = out if out is not None else (torch.empty(tuple(wshape), dtype=torch.float32, device=x.device),
                                          torch.empty((cout,), dtype=torch.float32, device=x.device))
    if tuple(dw.shape) != tuple(wshape) or db.numel() != cout:
        raise _lib.DevisError("convT wgrad: output shape mismatch")
    _check_out(dw, db)
    lib = _lib.load()
    nb = lib.devis_convT_wgrad_workspace_bytes(C.byref(d))
    ws = workspace(nb, x.device, slot) if nb else None
    check(lib.devis_convT_wgrad(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws), nb, _stream()), "convT_wgrad")
    return dw, db


# ------------------------------------------------------- fused DeUp_Cat tail
def upcat_supported(prev: Tensor, t: Tensor) -> bool:
    """True when the fused ConvT(k2,s2) + cat + 1x1 kernel (csrc/upcat.hip) takes these tensors."""
    if prev.dtype != torch.bfloat16 or t.dtype != torch.bfloat16 or t.dim() != 5 or prev.dim() != 5:
        return False
    n, d, h, w, c = t.shape
    if tuple(prev.shape) != (n, 2 * d, 2 * h, 2 * w, c):
        return False
    return bool(_lib.load().devis_upcat_supported(n, d, h, w, c, dt_code(t.dtype)))


def upcat_prepare(w3: Tensor, b3: Tensor, wt: Tensor, bt: Tensor):
    """fp32 masters of DeUp_Cat.conv3 / conv2 -> dict of the composite weights in the kernels' layouts."""
    _need_cuda(w3, b3, wt, bt)
    c = wt.shape[1]
    if tuple(w3.shape[:2]) != (c, 2 * c) or wt.shape[0] != c or wt.numel() != c * c * 8:
        raise _lib.DevisError("upcat: conv3 must be (C, 2C, 1, 1, 1) and conv2 (C, C, 2, 2, 2)")
    bf = dict(dtype=torch.bfloat16, device=w3.device)
    arena = torch.empty((8 * c * 2 * c + 2 * c * c + 8 * c * c,), **bf)
    o1 = 8 * c * 2 * c
    m = {"wc": arena[:o1].view(8, c, 2 * c), "w3a": arena[o1:o1 + c * c].view(c, 1, c),
         "w3a_t": arena[o1 + c * c:o1 + 2 * c * c].view(c, 1, c), "wq": arena[o1 + 2 * c * c:].view(c, 8, c),
         "bias": torch.empty((c,), dtype=torch.float32, device=w3.device), "C": c}
    check(_lib.load().devis_upcat_prepare(_p(w3.detach()), _p(b3.detach()), _p(wt.detach()), _p(bt.detach()), c, _p(m["wc"]),
                                          _p(m["bias"]), _p(m["w3a"]), _p(m["w3a_t"]), _p(m["wq"]), _stream()), "upcat_prepare")
    return m


def upcat_fwd(prev: Tensor, t: Tensor, m) -> Tensor:
    """y = conv3(cat((prev, convT(t)), 1)) in one pass (UNet3DZoo.py:121-129)."""
    _need_cuda(prev, t)
    n, d, h, w, c = t.shape
    if tuple(prev.shape) != (n, 2 * d, 2 * h, 2 * w, c) or c != m["C"]:
        raise _lib.DevisError("upcat: prev must be the (2D, 2H, 2W) tensor with t's channels")
    y = torch.empty_like(prev)
    check(_lib.load().devis_upcat_fwd(_p(prev), _p(t), _p(m["wc"]), _p(m["bias"]), _p(y), n, d, h, w, c, dt_code(t.dtype),
                                      _stream()), "upcat_fwd")
    return y


def upcat_bwd_data(dy: Tensor, m):
    """(d prev, d t) of upcat_fwd in one pass over dy (csrc/upcat.hip)."""
    _need_cuda(dy)
    n, d2, h2, w2, c = dy.shape
    if c != m["C"] or d2 % 2 or h2 % 2 or w2 % 2:
        raise _lib.DevisError("upcat: dy must be the (2D, 2H, 2W) gradient with the block's channels")
    dprev = torch.empty_like(dy)
    dt = torch.empty((n, d2 // 2, h2 // 2, w2 // 2, c), dtype=dy.dtype, device=dy.device)
    check(_lib.load().devis_upcat_bwd_data(_p(dy), _p(m["w3a_t"]), _p(m["wq"]), _p(dprev), _p(dt), n, d2 // 2, h2 // 2,
                                           w2 // 2, c, dt_code(dy.dtype), _stream()), "upcat_bwd_data")
    return dprev, dt


def upcat_grad_combine(g: Tensor, dw3a: Tensor, s: Tensor, w3: Tensor, wt: Tensor, bt: Tensor, out, slot: str = "ws"):
    """(G, dW3a, s) of the fused backward -> (dw3, db3, dwt, dbt) written into the buffers `out`."""
    dw3, db3, dwt, dbt = out
    _check_out(dw3, db3, dwt, dbt, g, dw3a, s)
    c = wt.shape[1]
    check(_lib.load().devis_upcat_grad_combine(_p(g), _p(dw3a), _p(s), _p(w3.detach()), _p(wt.detach()), _p(bt.detach()), c,
                                               _p(dw3), _p(db3), _p(dwt), _p(dbt), _stream()), "upcat_grad_combine")
    return out


# -------------------------------------------------------------------- norm
def _nvc(x: Tensor):
    n, c = x.shape[0], x.shape[-1]
    return n, x.numel() // (n * c), c


def _moments(x: Tensor, dy: Optional[Tensor] = None, scale=None, shift=None) -> Tensor:
    n, v, c = _nvc(x)
    lib = _lib.load()
    nb = lib.devis_moments_workspace_bytes(n, v, c)
    ws = workspace(nb, x.device)
    out = torch.empty((n, c, 2), dtype=torch.float64, device=x.device)
    if dy is None:
        check(lib.devis_channel_moments(_p(x), n, v, c, dt_code(x.dtype), _p(out), _p(ws), nb, _stream()), "channel_moments")
    else:
        check(lib.devis_norm_bwd_moments(_p(dy), _p(x), _p(scale), _p(shift), n, v, c, dt_code(x.dtype), _p(out),
                                         _p(ws), nb, _stream()), "norm_bwd_moments")
    return out


def _moment_partials(x: Tensor, dy: Optional[Tensor] = None, scale=None, shift=None):
    """First reduction stage only -> (workspace holding partial[n][chunk][C][2] fp64, chunks per sample)."""
    n, v, c = _nvc(x)
    lib = _lib.load()
    nb = lib.devis_moments_workspace_bytes(n, v, c)
    ws = workspace(nb, x.device)
    chunks = C.c_int64(0)
    if dy is None:
        check(lib.devis_channel_moments_partials(_p(x), n, v, c, dt_code(x.dtype), _p(ws), nb, C.byref(chunks),
                                                 _stream()), "channel_moments_partials")
    else:
        check(lib.devis_norm_bwd_moments_partials(_p(dy), _p(x), _p(scale), _p(shift), n, v, c, dt_code(x.dtype),
                                                  _p(ws), nb, C.byref(chunks), _stream()), "norm_bwd_moments_partials")
    return ws, chunks.value


def gn_stats(x: Tensor, gamma: Tensor, beta: Tensor, groups: int = 8, eps: float = 1e-5, partials=None):
    """GroupNorm statistics -> (scale[N,C], shift[N,C], mean[N,G], rstd[N,G]).  `partials` = (buffer, chunks)
    when the producing conv already left the per-chunk moments."""
    _need_cuda(x, gamma, beta)
    n, v, c = _nvc(x)
    ws, chunks = partials if partials is not None else _moment_partials(x)
    f = dict(dtype=torch.float32, device=x.device)
    coef = torch.empty((2, n, c), **f)
    mr = torch.empty((2, n, groups), **f)
    check(_lib.load().devis_gn_finalize_partials(_p(ws), chunks, n, v, c, groups, _p(gamma), _p(beta), eps,
                                                 _p(coef[0]), _p(coef[1]), _p(mr[0]), _p(mr[1]), _stream()),
          "gn_finalize_partials")
    return coef[0], coef[1], mr[0], mr[1]


def bn_stats(x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor, training: bool,
             momentum: float = 0.1, eps: float = 1e-5, tracked: Optional[Tensor] = None, partials=None):
    """BatchNorm coefficients; in training mode updates the running buffers (and `tracked`, the
    num_batches_tracked counter, when given) in place."""
    _need_cuda(x, gamma, beta, running_mean, running_var, tracked)
    n, v, c = _nvc(x)
    f = dict(dtype=torch.float32, device=x.device)
    coef = torch.empty((2, n, c), **f)
    mr = torch.empty((2, c), **f)
    lib = _lib.load()
    if training:
        if tracked is not None and tracked.dtype != torch.int64:
            raise _lib.DevisError("num_batches_tracked must be int64")
        ws, chunks = partials if partials is not None else _moment_partials(x)
        check(lib.devis_bn_finalize_partials(_p(ws), chunks, n, v, c, _p(gamma), _p(beta), eps, momentum,
                                             _p(running_mean), _p(running_var), _p(tracked), _p(coef[0]), _p(coef[1]),
                                             _p(mr[0]), _p(mr[1]), _stream()), "bn_finalize_partials")
    else:
        check(lib.devis_bn_finalize(None, n, v, c, _p(gamma), _p(beta), eps, momentum, 0,
                                    _p(running_mean), _p(running_var), _p(coef[0]), _p(coef[1]), _p(mr[0]), _p(mr[1]),
                                    _stream()), "bn_finalize")
    return coef[0], coef[1], mr[0], mr[1]


def affine_act(x: Tensor, scale: Tensor, shift: Tensor, residual: Optional[Tensor] = None, relu: bool = True) -> Tensor:
    _need_cuda(x, scale, shift, residual)
    n, v, c = _nvc(x)
    y = torch.empty_like(x)
    check(_lib.load().devis_affine_act(_p(x), _p(scale), _p(shift), _p(residual), _p(y), n, v, c, dt_code(x.dtype),
                                       int(relu), _stream()), "affine_act")
    return y


_gate_cache = {}


def _open_gate(n: int, c: int, device):
    """(scale, shift) = (0, 1): makes the ReLU mask [scale*x + shift > 0] of the norm-backward kernels all ones,
    i.e. the backward of a norm WITHOUT activation (the attention gate's BatchNorms)."""
    key = (n, c, device.index)
    g = _gate_cache.get(key)
    if g is None:
        def make():                        # constants that outlive the step: not from a launch plan's pool
            return (torch.zeros((n, c), dtype=torch.float32, device=device),
                    torch.ones((n, c), dtype=torch.float32, device=device))
        g = _gate_cache[key] = _persistent(make, filled_on=device)
    return g


def norm_bwd_coeffs(partials, n: int, v: int, c: int, gamma: Tensor, mean: Tensor, rstd: Tensor, groups: int,
                    bn: bool, out=None):
    """Per-chunk moments {sum dy', sum dy'*x} -> (coef[3, n, c] = cA, cB, cC of dx = cA*dy' + cB*x + cC, dgamma, dbeta).
    bn: False / 0 GroupNorm, True / 1 BatchNorm on batch statistics, 2 BatchNorm on its running statistics (eval)."""
    ws, chunks = partials
    f = dict(dtype=torch.float32, device=gamma.device)
    coef = torch.empty((3, n, c), **f)
    dgamma, dbeta = out if out is not None else (torch.empty((c,), **f), torch.empty((c,), **f))
    _check_out(dgamma, dbeta)
    check(_lib.load().devis_norm_bwd_finalize_partials(_p(ws), chunks, n, v, c, groups, int(bn), _p(gamma), _p(mean),
                                                       _p(rstd), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(dgamma),
                                                       _p(dbeta), _stream()), "norm_bwd_finalize_partials")
    return coef, dgamma, dbeta


def norm_relu_bwd(dy: Tensor, x: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], gamma: Tensor, mean: Tensor,
                  rstd: Tensor, groups: int, bn: bool, extra: Optional[Tensor] = None, out=None, partials=None,
                  relu: bool = True):
    """Backward of relu(norm(x)) -> (dx (+extra), dgamma, dbeta).  `out` = (dgamma, dbeta) buffers to fill;
    `partials` = (buffer, chunks) when the kernel that produced dy already left the per-chunk moments;
    relu=False: backward of the bare norm (scale / shift are not needed)."""
    _need_cuda(dy, x, extra)
    n, v, c = _nvc(x)
    if not relu:
        scale, shift = _open_gate(n, c, x.device)
    if partials is None:
        partials = _moment_partials(x, dy, scale, shift)
    coef, dgamma, dbeta = norm_bwd_coeffs(partials, n, v, c, gamma, mean, rstd, groups, bn, out)
    return norm_bwd_apply(dy, x, scale, shift, coef, extra), dgamma, dbeta


def norm_bwd_apply(dy: Tensor, x: Tensor, scale: Tensor, shift: Tensor, coef, extra: Optional[Tensor] = None) -> Tensor:
    """dx = cA * dy * [scale*x+shift > 0] + cB * x + cC (+ extra), coef = (cA, cB, cC) of norm_bwd_coeffs."""
    n, v, c = _nvc(x)
    dx = torch.empty_like(x)
    check(_lib.load().devis_norm_bwd_apply(_p(dy), _p(x), _p(scale), _p(shift), _p(coef[0]), _p(coef[1]), _p(coef[2]),
                                           _p(extra), _p(dx), n, v, c, dt_code(x.dtype), _stream()), "norm_bwd_apply")
    return dx


# ---------------------------------------------------------- attention gate
def att_psi_fwd(g1: Tensor, x1: Tensor, st_g, st_x, w_psi: Tensor, b_psi: Tensor):
    """-> (s fp32 [N, *spatial, 1], (moment partials of s, chunks)).  st_g / st_x = BatchNorm coefficients of g1 / x1."""
    _need_cuda(g1, x1, w_psi, b_psi)
    if g1.shape != x1.shape or g1.dtype != x1.dtype:
        raise _lib.DevisError("attention gate: g1 and x1 must match")
    n, v, c = _nvc(g1)
    lib = _lib.load()
    chunks = lib.devis_att_chunks(v)
    s = torch.empty(tuple(g1.shape[:-1]) + (1,), dtype=torch.float32, device=g1.device)
    part = torch.empty((n, chunks, 1, 2), dtype=torch.float64, device=g1.device)
    check(lib.devis_att_psi_fwd(_p(g1), _p(x1), _p(st_g[0]), _p(st_g[1]), _p(st_x[0]), _p(st_x[1]), _p(w_psi), _p(b_psi),
                                n, v, c, dt_code(g1.dtype), _p(s), _p(part), part.numel() * 8, _stream()), "att_psi_fwd")
    return s, (part, chunks)


def att_gate_fwd(skip: Tensor, s: Tensor, st_p) -> Tensor:
    _need_cuda(skip, s)
    n, v, c = _nvc(skip)
    out = torch.empty_like(skip)
    check(_lib.load().devis_att_gate_fwd(_p(skip), _p(s), _p(st_p[0]), _p(st_p[1]), _p(out), n, v, c,
                                         dt_code(skip.dtype), _stream()), "att_gate_fwd")
    return out


def att_gate_bwd(dout: Tensor, skip: Tensor, s: Tensor, st_p):
    """-> (dskip = dout * p, dz fp32 like s, (partials {sum dz, sum dz*s}, chunks))"""
    _need_cuda(dout, skip, s)
    n, v, c = _nvc(skip)
    lib = _lib.load()
    chunks = lib.devis_att_chunks(v)
    dskip = torch.empty_like(skip)
    dz = torch.empty_like(s)
    part = torch.empty((n, chunks, 1, 2), dtype=torch.float64, device=skip.device)
    check(lib.devis_att_gate_bwd(_p(dout), _p(skip), _p(s), _p(st_p[0]), _p(st_p[1]), _p(dskip), _p(dz), _p(part),
                                 part.numel() * 8, n, v, c, dt_code(skip.dtype), _stream()), "att_gate_bwd")
    return dskip, dz, (part, chunks)


def att_psi_bwd(dz: Tensor, s: Tensor, coef_p: Tensor, g1: Tensor, x1: Tensor, st_g, st_x, w_psi: Tensor, out):
    """-> (dpre like g1, partials_g, partials_x): gradient w.r.t. both BatchNorm outputs and their backward moments;
    fills out = (dw_psi, db_psi)."""
    _need_cuda(dz, s, g1, x1, w_psi)
    n, v, c = _nvc(g1)
    lib = _lib.load()
    chunks = lib.devis_att_psi_bwd_chunks(v, c, dt_code(g1.dtype))
    if chunks <= 0:
        raise _lib.DevisError("attention gate: unsupported F_int = %d" % c)
    dpre = torch.empty_like(g1)
    mom = torch.empty((2, n, chunks, c, 2), dtype=torch.float64, device=g1.device)
    dw, db = out
    _check_out(dw, db)
    nb = lib.devis_att_psi_bwd_workspace_bytes(n, v, c, dt_code(g1.dtype))
    ws = workspace(nb, g1.device)
    check(lib.devis_att_psi_bwd(_p(dz), _p(s), _p(coef_p[0]), _p(coef_p[1]), _p(coef_p[2]), _p(g1), _p(x1),
                                _p(st_g[0]), _p(st_g[1]), _p(st_x[0]), _p(st_x[1]), _p(w_psi), n, v, c,
                                dt_code(g1.dtype), _p(dpre), _p(mom[0]), _p(mom[1]), mom[0].numel() * 8, _p(dw), _p(db),
                                _p(ws), nb, _stream()), "att_psi_bwd")
    return dpre, (mom[0], chunks), (mom[1], chunks)


def add(a: Tensor, b: Tensor) -> Tensor:
    _need_cuda(a, b)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise _lib.DevisError("add: shape/dtype mismatch")
    z = torch.empty_like(a)
    check(_lib.load().devis_add(_p(a), _p(b), _p(z), a.numel(), dt_code(a.dtype), _stream()), "add")
    return z


def mul(a: Tensor, b: Tensor) -> Tensor:
    """a * b elementwise (dropout with the noise tensor given)."""
    _need_cuda(a, b)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise _lib.DevisError("mul: shape/dtype mismatch")
    z = torch.empty_like(a)
    check(_lib.load().devis_mul(_p(a), _p(b), _p(z), a.numel(), dt_code(a.dtype), _stream()), "mul")
    return z


def maxpool2d_fwd(x: Tensor):
    _need_cuda(x)
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    idx = torch.empty((n, h // 2, w // 2, c), dtype=torch.uint8, device=x.device)
    check(_lib.load().devis_maxpool2d_fwd(_p(x), _p(y), _p(idx), n, h, w, c, dt_code(x.dtype), _stream()), "maxpool_fwd")
    return y, idx


def maxpool2d_bwd(dy: Tensor, idx: Tensor, x_shape, skip: Optional[Tensor] = None) -> Tensor:
    """-> gradient of the pooled tensor's input (+ `skip`, the gradient arriving through the skip connection)."""
    _need_cuda(dy, idx, skip)
    n, h, w, c = x_shape
    if skip is not None and (tuple(skip.shape) != tuple(x_shape) or skip.dtype != dy.dtype):
        raise _lib.DevisError("maxpool_bwd: skip gradient must have the input's shape and dtype")
    dx = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device)
    check(_lib.load().devis_maxpool2d_bwd_add(_p(dy), _p(idx), _p(skip), _p(dx), n, h, w, c, dt_code(dy.dtype), _stream()),
          "maxpool_bwd")
    return dx


# ------------------------------------------------------------ evidential
def _label_code(labels: Tensor) -> int:
    if labels.dtype == torch.int64:
        return _lib.LABEL_I64
    if labels.dtype == torch.uint8:
        return _lib.LABEL_U8
    if labels.dtype == torch.int32:
        return _lib.LABEL_I32
    raise _lib.DevisError("labels must be int64, int32 or uint8")


def logits_strides(logits: Tensor, channels_last: bool):
    """(N, V, K, sN, sV, sK) in elements for a contiguous logits tensor."""
    n = logits.shape[0]
    if channels_last:
        k = logits.shape[-1]
        v = logits.numel() // (n * k)
        return n, v, k, v * k, k, 1
    k = logits.shape[1]
    v = logits.numel() // (n * k)
    return n, v, k, v * k, 1, v


def evid_loss_fwd(logits: Tensor, channels_last: bool, labels: Tensor, lam: float, want_evidence: bool,
                  want_per_voxel: bool = False):
    _need_cuda(logits, labels)
    n, v, k, sn, sv, sk = logits_strides(logits, channels_last)
    if labels.numel() != n * v:
        raise _lib.DevisError("labels must hold one class index per voxel")
    f = dict(dtype=torch.float32, device=logits.device)
    sp = tuple(logits.shape[1:-1]) if channels_last else tuple(logits.shape[2:])
    evidence = torch.empty((n, k) + sp, **f) if want_evidence else None
    per_voxel = torch.empty((n * v, 1), **f) if want_per_voxel else None
    scal = torch.empty((_lib.EVID_SCAL_FLOATS,), **f)
    lib = _lib.load()
    nb = lib.devis_evid_workspace_bytes(n, v, k)
    ws = workspace(nb, logits.device)
    check(lib.devis_evid_loss_fwd(_p(logits), sn, sv, sk, dt_code(logits.dtype), _p(labels), _label_code(labels),
                                  n, v, k, float(lam), _p(evidence), _p(per_voxel), _p(scal), _p(ws), nb, _stream()),
          "evid_loss_fwd")
    return scal, evidence, per_voxel


def evid_loss_bwd(logits: Tensor, channels_last: bool, labels: Tensor, lam: float, scal: Tensor,
                  grad_loss: Optional[Tensor], grad_evidence: Optional[Tensor] = None,
                  grad_per_voxel: Optional[Tensor] = None, grad_dice: Optional[Tensor] = None) -> Tensor:
    _need_cuda(logits, labels, scal, grad_loss, grad_evidence, grad_per_voxel, grad_dice)
    n, v, k, sn, sv, sk = logits_strides(logits, channels_last)
    dz = torch.empty_like(logits)
    check(_lib.load().devis_evid_loss_bwd(_p(logits), sn, sv, sk, dt_code(logits.dtype), _p(labels), _label_code(labels),
                                          n, v, k, float(lam), _p(scal), _p(grad_loss), _p(grad_per_voxel),
                                          _p(grad_dice), _p(grad_evidence), _p(dz), _stream()), "evid_loss_bwd")
    return dz


def crit_params(kind: int, raw: bool, lam: float = 0.0, a_au: float = 0.0, eps: float = 0.0,
                disentangle: bool = False) -> "_lib.EvidParams":
    return _lib.EvidParams(int(kind), 1 if raw else 0, 1 if disentangle else 0, float(lam), float(a_au), float(eps))


def crit_fwd(params, x: Tensor, strides, labels: Optional[Tensor], evidence_in: Optional[Tensor] = None,
             want_evidence: bool = False, want_per_voxel: bool = True, want_scal: bool = True, ev_shape=None):
    """General evidential criterion forward (devis_evid_crit_fwd).  `strides` = (N, V, K, sN, sV, sK) of x."""
    _need_cuda(x, labels, evidence_in)
    n, v, k, sn, sv, sk = strides
    if labels is not None and labels.numel() != n * v:
        raise _lib.DevisError("labels must hold one class index per voxel")
    if evidence_in is not None and (evidence_in.dtype != torch.float32 or evidence_in.numel() != n * v * k):
        raise _lib.DevisError("evidence must be fp32 with the shape of alpha")
    f = dict(dtype=torch.float32, device=x.device)
    evidence = torch.empty(ev_shape if ev_shape is not None else (n, k, v), **f) if want_evidence else None
    per_voxel = torch.empty((n * v, 1), **f) if want_per_voxel else None
    scal = torch.empty((_lib.EVID_SCAL_FLOATS,), **f) if want_scal else None
    lib = _lib.load()
    nb = lib.devis_evid_workspace_bytes(n, v, k) if want_scal else 0
    ws = workspace(nb, x.device) if want_scal else None
    check(lib.devis_evid_crit_fwd(C.byref(params), _p(x), sn, sv, sk, dt_code(x.dtype), _p(evidence_in), _p(labels),
                                  _label_code(labels) if labels is not None else 0, n, v, k, _p(evidence),
                                  _p(per_voxel), _p(scal), _p(ws), nb, _stream()), "evid_crit_fwd")
    return scal, evidence, per_voxel


def crit_bwd(params, x: Tensor, strides, labels: Optional[Tensor], scal: Optional[Tensor],
             evidence_in: Optional[Tensor] = None, grad_loss: Optional[Tensor] = None,
             grad_per_voxel: Optional[Tensor] = None, grad_dice: Optional[Tensor] = None,
             grad_evidence: Optional[Tensor] = None):
    """-> (d_x, d_evidence_in | None)"""
    _need_cuda(x, labels, scal, evidence_in, grad_loss, grad_per_voxel, grad_dice, grad_evidence)
    n, v, k, sn, sv, sk = strides
    dx = torch.empty_like(x)
    dev = torch.empty_like(evidence_in) if evidence_in is not None else None
    check(_lib.load().devis_evid_crit_bwd(C.byref(params), _p(x), sn, sv, sk, dt_code(x.dtype), _p(evidence_in),
                                          _p(labels), _label_code(labels) if labels is not None else 0, n, v, k,
                                          _p(scal), _p(grad_loss), _p(grad_per_voxel), _p(grad_dice),
                                          _p(grad_evidence), _p(dx), _p(dev), _stream()), "evid_crit_bwd")
    return dx, dev


def evid_infer(logits: Tensor, channels_last: bool, thresholds: Sequence[float] = (), want_evidence: bool = True,
               want_mne: bool = False):
    """-> dict(evidence, pred uint8, u fp32, mne, umask uint16)."""
    _need_cuda(logits)
    n, v, k, sn, sv, sk = logits_strides(logits, channels_last)
    sp = tuple(logits.shape[1:-1]) if channels_last else tuple(logits.shape[2:])
    dev = logits.device
    evidence = torch.empty((n, k) + sp, dtype=torch.float32, device=dev) if want_evidence else None
    pred = torch.empty((n,) + sp, dtype=torch.uint8, device=dev)
    u = torch.empty((n, 1) + sp, dtype=torch.float32, device=dev)
    mne = torch.empty((n,) + sp, dtype=torch.float32, device=dev) if want_mne else None
    nthr = len(thresholds)
    umask = torch.empty((n,) + sp, dtype=torch.int16, device=dev) if nthr else None
    thr = (C.c_float * max(nthr, 1))(*[float(t) for t in thresholds])
    check(_lib.load().devis_evid_infer(_p(logits), sn, sv, sk, dt_code(logits.dtype), n, v, k, _p(evidence), _p(pred),
                                       _p(u), _p(mne), _p(umask), C.cast(thr, C.c_void_p), nthr, _stream()),
          "evid_infer")
    return dict(evidence=evidence, pred=pred, u=u, mne=mne, umask=umask)


def eval_counts(pred: Tensor, target: Tensor, num_classes: int, umask: Optional[Tensor] = None, n_thr: int = 0,
                u: Optional[Tensor] = None):
    """Device-side counts behind the reference's val()/test() metrics -> dict(conf (K,K) int64 [target, pred],
    ucount (n_thr, 8) int64 | None, sums (3,) fp64 | None, bad (1,) int32).  Nothing is copied to the host."""
    _need_cuda(pred, target, umask, u)
    if pred.dtype != torch.uint8:
        raise _lib.DevisError("pred must be uint8 (devis_evid_infer's output)")
    nv = pred.numel()
    if target.numel() != nv or (umask is not None and umask.numel() != nv) or (u is not None and u.numel() != nv):
        raise _lib.DevisError("eval_counts: pred, target, umask and u must have one entry per voxel")
    if umask is not None and umask.dtype != torch.int16:
        raise _lib.DevisError("umask must be int16 (devis_evid_infer's output)")
    if u is not None and u.dtype != torch.float32:
        raise _lib.DevisError("u must be fp32")
    dev = pred.device
    conf = torch.empty((num_classes, num_classes), dtype=torch.int64, device=dev)
    ucount = torch.empty((n_thr, 8), dtype=torch.int64, device=dev) if n_thr else None
    sums = torch.empty((3,), dtype=torch.float64, device=dev) if u is not None else None
    bad = torch.empty((1,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    nb = lib.devis_eval_counts_workspace_bytes(nv)
    ws = workspace(nb, dev)
    check(lib.devis_eval_counts(_p(pred), _p(target), _label_code(target), _p(umask) if n_thr else None, n_thr, _p(u), nv,
                                num_classes, _p(conf), _p(ucount), _p(sums), _p(bad), _p(ws), nb, _stream()),
          "eval_counts")
    return dict(conf=conf, ucount=ucount, sums=sums, bad=bad)


def softplus_bwd(z: Tensor, g: Tensor) -> Tensor:
    _need_cuda(z, g)
    dz = torch.empty_like(z)
    check(_lib.load().devis_softplus_bwd(_p(z), _p(g), _p(dz), z.numel(), _stream()), "softplus_bwd")
    return dz
