#!/usr/bin/env python3
"""Bit-stability probes of kernels that share the chip with kernels of another stream (one file; was race_probe{,3,4,5}.py).
    python tools/race_probe.py wgrad         a weight gradient on the side stream beside the dgrad of the same layer
    python tools/race_probe.py elementwise   a weight gradient beside LDS-free elementwise kernels that share its CUs
    python tools/race_probe.py apply         the norm-backward apply kernel beside MFMA kernels
    python tools/race_probe.py packed [pk|pknop]   the same on fixed inputs in a build WITH packed fp32 VALU instructions
                                             (the r02 root-cause run; logs profiles/r02_race_probe5_*.log)
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def probe_wgrad(argv):
    """wgrad on a side stream while the dgrad of the same layer runs on the main stream: are both outputs bit-stable?"""

    from devis_amd import ops
    from devis_amd.unet3d import _side_stream
    geom = ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
    g = torch.Generator().manual_seed(0)
    dev = torch.device("cuda", 0)
    side = _side_stream(dev)
    for c, sz, pro in ((16, 128, True), (32, 64, True), (64, 32, True), (128, 16, True)):
        w = (torch.randn((c, c, 3, 3, 3), generator=g) * 0.05).cuda()
        wp = ops.pack_conv_weight(w, torch.bfloat16)
        x = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
        dy = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
        res = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
        sc = (torch.rand((2, c), generator=g) + 0.5).cuda(); sh = torch.zeros((2, c)).cuda()
        dw = torch.empty_like(w); db = torch.empty(c, device=dev)
        ref = None
        bad_w = bad_d = bad_alone = 0
        for it in range(24):
            concurrent = it >= 4
            torch.cuda.synchronize()
            ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
            if concurrent:
                with ops.stream_override(side):
                    ops.conv_wgrad(x, None, dy, geom, w.shape, sc if pro else None, sh if pro else None, out=(dw, db), slot="ws_side")
                dx, _, part = ops.conv_dgrad(dy, wp, geom, x.shape, norm=(x, sc, sh))
            else:
                ops.conv_wgrad(x, None, dy, geom, w.shape, sc if pro else None, sh if pro else None, out=(dw, db))
                torch.cuda.synchronize()
                dx, _, part = ops.conv_dgrad(dy, wp, geom, x.shape, norm=(x, sc, sh))
            torch.cuda.synchronize()
            cur = (dw.clone(), db.clone(), torch.cat([dx.float().flatten(), part[0].float().flatten()]))
            if ref is None: ref = cur
            else:
                w_ok = torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]); d_ok = torch.equal(cur[2], ref[2])
                if concurrent: bad_w += not w_ok; bad_d += not d_ok
                else: bad_alone += not (w_ok and d_ok)
        print("pro=%d " % pro + "%3d ch @ %3d^3: serial mismatches %d; concurrent: wgrad %d/20, dgrad+stats %d/20" % (c, sz, bad_alone, bad_w, bad_d))


def probe_elementwise(argv):
    """A wgrad on the side stream while the main stream runs LDS-free elementwise kernels (they can share its CUs)."""

    from devis_amd import ops
    from devis_amd.unet3d import _side_stream
    geom = ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
    g = torch.Generator().manual_seed(0)
    dev = torch.device("cuda", 0)
    side = _side_stream(dev)
    big = torch.randn((2, 128, 128, 128, 16), generator=g).bfloat16().cuda()
    bsc = torch.ones((2, 16)).cuda(); bsh = torch.zeros((2, 16)).cuda()
    for c, sz, pro in ((16, 128, True), (32, 64, False), (32, 64, True), (64, 32, True), (128, 16, True), (64, 32, False)):
        w = (torch.randn((c, c, 3, 3, 3), generator=g) * 0.05).cuda()
        x = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
        dy = torch.randn((2, sz, sz, sz, c), generator=g).bfloat16().cuda()
        sc = (torch.rand((2, c), generator=g) + 0.5).cuda(); sh = torch.zeros((2, c)).cuda()
        dw = torch.empty_like(w); db = torch.empty(c, device=dev)
        ref = None; bad = 0
        for it in range(24):
            concurrent = it >= 4
            torch.cuda.synchronize()
            ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
            with ops.stream_override(side):
                ops.conv_wgrad(x, None, dy, geom, w.shape, sc if pro else None, sh if pro else None, out=(dw, db), slot="ws_side")
            if concurrent:
                for _ in range(3):
                    ops.affine_act(big, bsc, bsh, relu=True)
            torch.cuda.synchronize()
            cur = torch.cat([dw.flatten(), db])
            if ref is None: ref = cur.clone()
            elif concurrent: bad += not torch.equal(cur, ref)
        print("pro=%d %3d ch @ %3d^3: wgrad mismatches with a concurrent elementwise kernel %d/20" % (pro, c, sz, bad))


def probe_apply(argv):
    """Is the OUTPUT of the elementwise norm-backward apply kernel bit-stable while MFMA kernels share the chip?"""

    from devis_amd import ops
    from devis_amd.unet3d import _side_stream
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
        dw = torch.empty_like(w); db = torch.empty(c, device=dev); dg = torch.empty(c, device=dev); dbt = torch.empty(c, device=dev)
        ref = None; bad = 0; nbad = 0
        for it in range(24):
            concurrent = it >= 4
            torch.cuda.synchronize()
            if concurrent:
                ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
                with ops.stream_override(side):
                    for _ in range(3):
                        ops.conv_wgrad(x, None, dy, geom, w.shape, sc, sh, out=(dw, db), slot="ws_side")
            dh, _, _ = ops.norm_relu_bwd(da, x, sc, sh, gamma, mean, rstd, 0, True, out=(dg, dbt))
            torch.cuda.synchronize()
            if ref is None: ref = dh.clone()
            elif concurrent and not torch.equal(dh, ref):
                bad += 1; nbad = int((dh != ref).sum())
                diff_idx = (dh != ref).flatten().nonzero().flatten()
                par = (diff_idx % 2).float().mean().item()
        print("%3d ch @ %3d^3: apply output mismatches under concurrent MFMA kernels %d/20%s" % (c, sz, bad, "" if not bad else "  (last: %d elements, odd-channel share %.2f)" % (nbad, par)))


def probe_packed(argv):
    """Which kernel of the norm backward is not bit-stable when MFMA kernels of another stream share the chip, in a
    build WITH packed fp32 VALU instructions (tools/libdevis_hip_pk.so = the library with norm.hip compiled without
    '-packed-fp32-ops')?  ADVICE r1: fix the inputs.  The moment partials and the coefficients cA/cB/cC are computed
    ONCE and snapshotted; in the concurrent loop (a) only devis_norm_bwd_apply runs on the snapshot and its output is
    compared bitwise, and (b) separately the partials + coefficients are recomputed and compared bitwise.
        python tools/race_probe.py packed [pk|pknop] (pk: load tools/libdevis_hip_pk.so; pknop: ..._pknop.so, the same
                                                    with -DDEVIS_PK_NOP = two wait states between the ReLU-gate select
                                                    and the packed FMA that reads it)
    Build the diagnostic libraries (not shipped, not tracked):
        cd devis_amd/csrc && hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 [-DDEVIS_PK_NOP] -c norm.hip -o /tmp/norm_pk.o
        hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v /norm.o) /tmp/norm_pk.o -o ../../tools/libdevis_hip_pk.so
    Result (profiles/r02_race_probe5_*.log): partials and coefficients never differ; the apply output differs in 8-9 of
    10 runs at 64 ch x 32^3 with packed ops (with or without the extra wait states), never in the shipped build."""

    from devis_amd import _lib
    if len(argv) > 0 and argv[0] in ("pk", "pknop"):
        _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdevis_hip_%s.so" % argv[0])
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



if __name__ == "__main__":
    modes = {"wgrad": probe_wgrad, "elementwise": probe_elementwise, "apply": probe_apply, "packed": probe_packed}
    if len(sys.argv) < 2 or sys.argv[1] not in modes:
        sys.exit(__doc__)
    modes[sys.argv[1]](sys.argv[2:])
