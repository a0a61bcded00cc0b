#!/usr/bin/env python3
"""norm_bwd_apply + conv_dgrad against the fused devis_conv_dgrad_normbwd at 2 x 128^3 x 16 (HIP events, 20 calls)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import ops, _lib
from devis_amd.unet3d import G3
g = torch.Generator().manual_seed(1)
bf = lambda *s: torch.randn(s, generator=g).bfloat16().cuda()
n, sp = 2, 128
gg, xn, res, xref = bf(n, sp, sp, sp, 16), bf(n, sp, sp, sp, 16), bf(n, sp, sp, sp, 16), bf(n, sp, sp, sp, 16)
sc = (torch.rand((n, 16), generator=g) + 0.5).cuda(); sh = (torch.randn((n, 16), generator=g) * 0.3).cuda()
coef = (torch.randn((3, n, 16), generator=g) * 0.5).cuda()
wp = ops.pack_conv_weight((torch.randn((16, 16, 3, 3, 3), generator=g) * 0.1).cuda(), torch.bfloat16)
wd = torch.empty((16, 27, 16), dtype=wp.dtype, device=wp.device)
lib = _lib.load()
_lib.check(lib.devis_repack_conv_weight_dgrad(wp.data_ptr(), wd.data_ptr(), 16, 16, 27, _lib.DEVIS_BF16, torch.cuda.current_stream().cuda_stream), "repack")
dh = torch.empty_like(gg)
def two(residual, norm):
    _lib.check(lib.devis_norm_bwd_apply(gg.data_ptr(), xn.data_ptr(), sc.data_ptr(), sh.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                        coef[2].data_ptr(), None, dh.data_ptr(), n, sp ** 3, 16, _lib.DEVIS_BF16, torch.cuda.current_stream().cuda_stream), "apply")
    return ops.conv_dgrad(dh, wp, G3, gg.shape, residual=residual, wd=wd, norm=norm)
for name, residual, norm in (("stats", None, (xref, sc, sh)), ("residual", res, None)):
    for label, fn in (("apply + dgrad", lambda: two(residual, norm)), ("fused", lambda: ops.conv_dgrad_normbwd(gg, xn, sc, sh, coef, wp, wd, G3, residual=residual, norm=norm))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print("%-9s %-14s %7.1f us" % (name, label, e0.elapsed_time(e1) / 20 * 1e3))
