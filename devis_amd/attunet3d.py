"""3-D attention U-Net on MI355X: constructor, attribute names and ``state_dict`` keys of the reference
``AttUnet`` (models/lib/UNet3DZoo.py:69-115,132-143,167-241), executed by the HIP kernels.

Encoder, DeBlocks and endconv are the Unet's (devis_amd/unet3d.py).  The decoder differs: ``DeUp`` (1x1 ->
ConvT k2s2 -> 1x1), an attention gate on the skip tensor (csrc/attgate.hip: the gate's three BatchNorms, ReLU,
one-channel psi conv, sigmoid and product in two kernels forward and two backward; its W_g / W_x 1x1 convs on
the gather-GEMM kernels) and ``De_Cat`` (virtual concat + 1x1).  Like the reference, level 4 concatenates
(gated skip, upsampled) and levels 3 and 2 (upsampled, gated skip) (UNet3DZoo.py:227,232,237).
"""
from __future__ import annotations

import torch.nn as nn

from . import ops
from .unet3d import (G1, G3, G3S2, Unet, _DeBlock, _EnBlock, _EnDown, _InitConv, _Run)


class _AttentionBlock(nn.Module):      # UNet3DZoo.py:69-100 (parameter holder; Sequential indices as the reference)
    def __init__(self, f_g, f_l, f_int):
        super().__init__()
        self.W_g = nn.Sequential(nn.Conv3d(f_g, f_int, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm3d(f_int))
        self.W_x = nn.Sequential(nn.Conv3d(f_l, f_int, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm3d(f_int))
        self.psi = nn.Sequential(nn.Conv3d(f_int, 1, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm3d(1),
                                 nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)


class _DeCat(nn.Module):               # UNet3DZoo.py:102-115
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, cout, kernel_size=1)


class _DeUp(nn.Module):                # UNet3DZoo.py:132-143
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, cout, kernel_size=1)
        self.conv2 = nn.ConvTranspose3d(cout, cout, kernel_size=2, stride=2)
        self.conv3 = nn.Conv3d(cout, cout, kernel_size=1)


class _RunAtt(_Run):
    """Forward / backward schedule of the attention U-Net over channels-last tensors."""

    def _bn_plain(self, name, x, partials=None):
        return ops.bn_stats(x, self.P[name + ".weight"], self.P[name + ".bias"], self.B[name + ".running_mean"],
                            self.B[name + ".running_var"], self.training,
                            tracked=self.B[name + ".num_batches_tracked"], partials=partials)

    # ---- forward blocks
    def de_up(self, name, x):
        w1, wt, w3 = self._w(name + ".conv1"), self._w(name + ".conv2"), self._w(name + ".conv3")
        t = ops.conv_fwd(x, None, w1, self.P[name + ".conv1.bias"], G1)
        up = ops.convT_fwd(t, wt, self.P[name + ".conv2.bias"], (2, 2, 2))
        y = ops.conv_fwd(up, None, w3, self.P[name + ".conv3.bias"], G1)
        self._push("dup", name, x, t, up, w1, wt, w3)
        return y

    def att(self, name, g, x):
        wg, wx = self._w(name + ".W_g.0"), self._w(name + ".W_x.0")
        g1 = ops.conv_fwd(g, None, wg, self.P[name + ".W_g.0.bias"], G1)
        st_g = self._bn_plain(name + ".W_g.1", g1)
        x1 = ops.conv_fwd(x, None, wx, self.P[name + ".W_x.0.bias"], G1)
        st_x = self._bn_plain(name + ".W_x.1", x1)
        s, part = ops.att_psi_fwd(g1, x1, st_g, st_x, self.P[name + ".psi.0.weight"], self.P[name + ".psi.0.bias"])
        st_p = self._bn_plain(name + ".psi.1", s, part if self.training else None)
        out = ops.att_gate_fwd(x, s, st_p)
        self._push("att", name, g, x, g1, x1, s, st_g, st_x, st_p, wg, wx)
        return out

    def de_cat(self, name, first, second):
        w = self._w(name + ".conv1")
        y = ops.conv_fwd(first, second, w, self.P[name + ".conv1.bias"], G1)      # virtual cat((first, second))
        self._push("dcat", name, first, second, w)
        return y

    def forward(self, x):
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
        y = self.de_up("DeUp4", h)
        a = self.att("Att4", y, s3)
        y = self.de_block("DeBlock4", self.de_cat("DeUpCat4", a, y))       # cat((gated skip, up)), :227
        y = self.de_up("DeUp3", y)
        a = self.att("Att3", y, s2)
        y = self.de_block("DeBlock3", self.de_cat("DeUpCat3", y, a))       # cat((up, gated skip)), :232
        y = self.de_up("DeUp2", y)
        a = self.att("Att2", y, s1)
        y = self.de_block("DeBlock2", self.de_cat("DeUpCat2", y, a))       # :237
        return self.conv("endconv", y, G1)

    # ---- backward blocks
    def _dcat_bwd(self, e, dy):
        _, name, first, second, w = e
        g, P = self.grads, self.P
        g[name + ".conv1.weight"], g[name + ".conv1.bias"] = self._wg(
            ops.conv_wgrad, first, second, dy, G1, P[name + ".conv1.weight"].shape, out=self._o(name + ".conv1"))
        return ops.conv_dgrad(dy, w, G1, first.shape, second.shape[-1], wd=self._wd(name + ".conv1"))

    def _att_bwd(self, e, datt, dg_cat):
        """-> (d g  [= dg_cat + W_g^T d g1], d skip)"""
        _, name, gt, x, g1, x1, s, st_g, st_x, st_p, wg, wx = e
        g, P = self.grads, self.P
        n, v, c = ops._nvc(g1)
        dskip, dz, part = ops.att_gate_bwd(datt, x, s, st_p)
        coef_p, g[name + ".psi.1.weight"], g[name + ".psi.1.bias"] = ops.norm_bwd_coeffs(
            part, n, v, 1, P[name + ".psi.1.weight"], st_p[2], st_p[3], 0, 1 if self.training else 2, out=self._o(name + ".psi.1"))
        dpre, mom_g, mom_x = ops.att_psi_bwd(dz, s, coef_p, g1, x1, st_g, st_x, P[name + ".psi.0.weight"],
                                             out=self._o(name + ".psi.0"))
        g[name + ".psi.0.weight"], g[name + ".psi.0.bias"] = self._o(name + ".psi.0")
        dg1, g[name + ".W_g.1.weight"], g[name + ".W_g.1.bias"] = ops.norm_relu_bwd(
            dpre, g1, None, None, P[name + ".W_g.1.weight"], st_g[2], st_g[3], 0, 1 if self.training else 2, out=self._o(name + ".W_g.1"),
            partials=mom_g, relu=False)
        dx1, g[name + ".W_x.1.weight"], g[name + ".W_x.1.bias"] = ops.norm_relu_bwd(
            dpre, x1, None, None, P[name + ".W_x.1.weight"], st_x[2], st_x[3], 0, 1 if self.training else 2, out=self._o(name + ".W_x.1"),
            partials=mom_x, relu=False)
        g[name + ".W_g.0.weight"], g[name + ".W_g.0.bias"] = self._wg(
            ops.conv_wgrad, gt, None, dg1, G1, P[name + ".W_g.0.weight"].shape, out=self._o(name + ".W_g.0"))
        dg = ops.conv_dgrad(dg1, wg, G1, gt.shape, residual=dg_cat, wd=self._wd(name + ".W_g.0"))[0]
        g[name + ".W_x.0.weight"], g[name + ".W_x.0.bias"] = self._wg(
            ops.conv_wgrad, x, None, dx1, G1, P[name + ".W_x.0.weight"].shape, out=self._o(name + ".W_x.0"))
        dsk = ops.conv_dgrad(dx1, wx, G1, x.shape, residual=dskip, wd=self._wd(name + ".W_x.0"))[0]
        return dg, dsk

    def _dup_bwd(self, e, dy):
        """-> (dx, moments for the next DeBlock's bn2 backward | None)"""
        _, name, x, t, up, w1, wt, w3 = e
        g, P = self.grads, self.P
        g[name + ".conv3.weight"], g[name + ".conv3.bias"] = self._wg(
            ops.conv_wgrad, up, None, dy, G1, P[name + ".conv3.weight"].shape, out=self._o(name + ".conv3"))
        dup = ops.conv_dgrad(dy, w3, G1, up.shape, wd=self._wd(name + ".conv3"))[0]
        g[name + ".conv2.weight"], g[name + ".conv2.bias"] = self._wg(
            ops.convT_wgrad, t, dup, (2, 2, 2), P[name + ".conv2.weight"].shape, out=self._o(name + ".conv2"))
        dt = ops.convT_dgrad(dup, wt, (2, 2, 2), t.shape, wt=self._wd(name + ".conv2"))
        g[name + ".conv1.weight"], g[name + ".conv1.bias"] = self._wg(
            ops.conv_wgrad, x, None, dt, G1, P[name + ".conv1.weight"].shape, out=self._o(name + ".conv1"))
        norm = self._next_norm()
        if norm is not None:
            dx, _, pt = ops.conv_dgrad(dt, w1, G1, x.shape, wd=self._wd(name + ".conv1"), norm=norm)
            return dx, pt
        return ops.conv_dgrad(dt, w1, G1, x.shape, wd=self._wd(name + ".conv1"))[0], None

    def backward(self, dlogits, need_dx):
        pop = self.tape.pop
        self._begin_backward(dlogits.device)
        e = pop()
        d, pt = self._conv_bwd(e, dlogits, norm=self._next_norm())             # endconv
        skips = []
        for level in (2, 3, 4):
            d = self._de_bwd(pop(), d, pt)                                     # DeBlock
            dfirst, dsecond = self._dcat_bwd(pop(), d)                         # De_Cat
            datt, dg_cat = (dfirst, dsecond) if level == 4 else (dsecond, dfirst)
            dg, dsk = self._att_bwd(pop(), datt, dg_cat)                       # attention gate
            skips.append(dsk)
            d, pt = self._dup_bwd(pop(), dg)                                   # DeUp
        ds1, ds2, ds3 = skips
        self._bucket_done(0)
        for _ in range(4):
            d = self._en_bwd(pop(), d)                                         # EnBlock4_4..1
        self._bucket_done(1)
        d = self._conv_bwd(pop(), d, skip=ds3)                                 # EnDown3 (+ skip gradient)
        d = self._en_bwd(pop(), d); d = self._en_bwd(pop(), d)
        d = self._conv_bwd(pop(), d, skip=ds2)
        d = self._en_bwd(pop(), d); d = self._en_bwd(pop(), d)
        d = self._conv_bwd(pop(), d, skip=ds1)
        d = self._en_bwd(pop(), d)
        d = self._conv_bwd(pop(), d, need_dx)                                  # InitConv
        self._end_backward()
        self._bucket_done(2)
        return d


class AttUnet(Unet):
    """Drop-in for the reference ``AttUnet(in_channels, base_channels, num_classes)`` (UNet3DZoo.py:167-241).
    Modules are created in the reference's order, so ``torch.manual_seed(s)`` gives identical initial weights."""

    _run_cls = _RunAtt

    def __init__(self, in_channels=1, base_channels=16, num_classes=4, compute_dtype=None):
        import torch
        nn.Module.__init__(self)
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
        self.Att4 = _AttentionBlock(64, 64, 32)            # hard-coded channel counts, as UNet3DZoo.py:184-186
        self.Att3 = _AttentionBlock(32, 32, 16)
        self.Att2 = _AttentionBlock(16, 16, 16)
        self.EnBlock4_1 = _EnBlock(b * 8)
        self.EnBlock4_2 = _EnBlock(b * 8)
        self.EnBlock4_3 = _EnBlock(b * 8)
        self.EnBlock4_4 = _EnBlock(b * 8)
        self.DeUp4 = _DeUp(b * 8, b * 4)
        self.DeUpCat4 = _DeCat(b * 8, b * 4)
        self.DeBlock4 = _DeBlock(b * 4)
        self.DeUp3 = _DeUp(b * 4, b * 2)
        self.DeUpCat3 = _DeCat(b * 4, b * 2)
        self.DeBlock3 = _DeBlock(b * 2)
        self.DeUp2 = _DeUp(b * 2, b)
        self.DeUpCat2 = _DeCat(b * 2, b)
        self.DeBlock2 = _DeBlock(b)
        self.endconv = nn.Conv3d(b, num_classes, kernel_size=1)
        self._finish_init(compute_dtype if compute_dtype is not None else torch.bfloat16, num_classes)
