#!/usr/bin/env python3
"""Which kernel of the norm backward is not bit-stable when MFMA kernels of another stream share the chip, in a
build WITH packed fp32 VALU instructions (tools/libdevis_hip_pk.so = the library with norm.hip compiled without
'-packed-fp32-ops')?  ADVICE r1: fix the inputs.  The moment partials and the coefficients cA/cB/cC are computed
ONCE and snapshotted; in the concurrent loop (a) only devis_norm_bwd_apply runs on the snapshot and its output is
compared bitwise, and (b) separately the partials + coefficients are recomputed and compared bitwise.
    python tools/race_probe5.py [pk|pknop]     (pk: load tools/libdevis_hip_pk.so; pknop: ..._pknop.so, the same
                                                with -DDEVIS_PK_NOP = two wait states between the ReLU-gate select
                                                and the packed FMA that reads it)
Build the diagnostic libraries (not shipped, not tracked):
    cd devis_amd/csrc && hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 [-DDEVIS_PK_NOP] -c norm.hip -o /tmp/norm_pk.o
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v /norm.o) /tmp/norm_pk.o -o ../../tools/libdevis_hip_pk.so
Result (profiles/r02_race_probe5_*.log): partials and coefficients never differ; the apply output differs in 8-9 of
10 runs at 64 ch x 32^3 with packed ops (with or without the extra wait states), never in the shipped build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devis_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] in ("pk", "pknop"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdevis_hip_%s.so" % sys.argv[1])
from devis_amd import ops
from devis_amd.ops import _p, _stream, dt_code
from devis_amd.unet3d import _side_stream
print("library:", _lib.LIB_PATH)
lib = _lib.load()
geom = ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
g = torch.Generator().manual_seed(0)
dev = torch.device("cuda", 0)
side = _side_stream(dev)
for c, sz in ((32, 64), (16, 128), (64, 32)):
    w = (torch.randn((c, c, 3, 3, 3), generator=g) * 0.05).cuda()
    x = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    dy = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
    da = (torch.randn((2, sz, sz, sz, c), generator=g) * 1e-3).bfloat16().cuda()
    gamma = (torch.rand(c, generator=g) + 0.5).cuda(); beta = torch.zeros(c).cuda()
    sc, sh, mean, rstd = ops.bn_stats(x, gamma, beta, torch.zeros(c).cuda(), torch.ones(c).cuda(), True)
    dw = torch.empty_like(w); db = torch.empty(c, device=dev)
    n, v = 2, sz ** 3
    # inputs of the apply kernel, computed once without anything else on the chip
    part = ops._moment_partials(x, da, sc, sh)
    part0 = part[0].clone()
    coef0, dg0, db0 = ops.norm_bwd_coeffs(part, n, v, c, gamma, mean, rstd, 0, True)
    coef0 = coef0.clone()
    torch.cuda.synchronize()

    def apply_only():
        dx = torch.empty_like(x)
        _lib.check(lib.devis_norm_bwd_apply(_p(da), _p(x), _p(sc), _p(sh), _p(coef0[0]), _p(coef0[1]), _p(coef0[2]),
                                            None, _p(dx), n, v, c, dt_code(x.dtype), _stream()), "apply")
        return dx
    ref = apply_only(); torch.cuda.synchronize()
    bad_apply = bad_part = bad_coef = 0; nbad = 0; par = 0.0
    for it in range(20):
        torch.cuda.synchronize()
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        with ops.stream_override(side):
            for _ in range(3):
                ops.conv_wgrad(x, None, dy, geom, w.shape, sc, sh, out=(dw, db), slot="ws_side")
        if it % 2 == 0:
            dh = apply_only()
            torch.cuda.synchronize()
            if not torch.equal(dh, ref):
                bad_apply += 1; nbad = int((dh != ref).sum())
                idx = (dh != ref).flatten().nonzero().flatten()
                par = (idx % 2).float().mean().item()
        else:
            p2 = ops._moment_partials(x, da, sc, sh)
            c2, _, _ = ops.norm_bwd_coeffs(p2, n, v, c, gamma, mean, rstd, 0, True)
            torch.cuda.synchronize()
            nel = part0.numel()
            bad_part += int(not torch.equal(p2[0].flatten()[:nel], part0.flatten()))
            bad_coef += int(not torch.equal(c2, coef0))
    print("%3d ch @ %3d^3 under concurrent MFMA kernels: apply(fixed inputs) differs %d/10%s; partials differ %d/10; coefficients differ %d/10"
          % (c, sz, bad_apply, "" if not bad_apply else " (last: %d elements, odd-channel share %.2f)" % (nbad, par), bad_part, bad_coef))
