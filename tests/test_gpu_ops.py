"""GPU unit parity: every C-ABI op against the same torch op evaluated on the CPU in fp32/fp64
(the plain-PyTorch statement of what the reference's nn.Module would compute)."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import FP32_BAR, bf16_ulps, check, rel_err

pytestmark = pytest.mark.gpu


def cl(t, dtype=torch.float32):     # NC... (cpu) -> channels-last on GPU
    perm = (0,) + tuple(range(2, t.dim())) + (1,)
    return t.permute(*perm).contiguous().to(dtype).cuda()


def uncl(t):                        # channels-last GPU -> NC... cpu fp32
    perm = (0, t.dim() - 1) + tuple(range(1, t.dim() - 1))
    return t.float().permute(*perm).contiguous().cpu()


def close(y, ref, dtype, what):
    """A kernel OUTPUT TENSOR in the compute dtype.  fp32 mode: 1e-5 of the largest reference value.  bf16 mode:
    the kernels accumulate in fp32 and round once, so every element must be the reference (fp32 on the same
    rounded operands) rounded to bf16, or its neighbour: <= 1 bf16 ulp."""
    if dtype == torch.float32:
        check(rel_err(y, ref), FP32_BAR, what)
    else:
        check(bf16_ulps(y, ref), 1.0, what + " [bf16 ulps]")


def close32(y, ref, dtype, what):
    """An fp32 OUTPUT (weight / bias gradient, statistics) of a kernel: 1e-5 in fp32 mode; in bf16 mode the operands
    are bf16 but the sums are fp32 (fp64 for statistics), so 1e-4 - four orders tighter than a bf16 rounding."""
    check(rel_err(y, ref), FP32_BAR if dtype == torch.float32 else 1e-4, what)


CONV_CASES = [
    # (N, Cin1, Cin2, Cout, spatial, kernel, stride, pad)
    (2, 4, 0, 16, (6, 8, 10), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 16, 0, 32, (8, 8, 8), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 32, 0, 32, (4, 6, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 16, 16, 16, (4, 4, 6), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (2, 16, 0, 4, (4, 4, 6), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (2, 8, 24, 16, (9, 11), (3, 3), (1, 1), (1, 1)),
    (3, 3, 0, 32, (12, 7), (3, 3), (1, 1), (1, 1)),
    # shapes the MFMA implicit-GEMM kernels take in bf16 (incl. ragged tiles and stride 2)
    (2, 16, 0, 16, (8, 8, 16), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 16, 0, 16, (6, 10, 20), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 32, 0, 32, (8, 8, 16), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 32, 0, 32, (5, 9, 24), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 64, 0, 64, (4, 8, 16), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 128, 0, 128, (4, 4, 16), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 128, 0, 128, (2, 2, 2), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 64, 0, 64, (5, 6, 20), (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # weights-stationary kernel (conv_deepk.hip): ragged
    (2, 128, 0, 128, (3, 5, 18), (3, 3, 3), (1, 1, 1), (1, 1, 1)),    # tiles in D, H and W
    (1, 16, 0, 32, (8, 8, 32), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (2, 32, 0, 64, (8, 8, 16), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 64, 0, 128, (4, 8, 16), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 16, 0, 32, (5, 7, 9), (3, 3, 3), (2, 2, 2), (1, 1, 1)),       # odd sizes: ragged parity classes in the dgrad
    (2, 16, 0, 32, (40, 18, 36), (3, 3, 3), (2, 2, 2), (1, 1, 1)),    # stride-2 plane march: five D chunks, ragged tiles
    (2, 32, 0, 64, (22, 14, 40), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 32, 0, 64, (6, 10, 36), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    # 1x1 layers (gather-GEMM kernels in bf16): DeUpCat conv1 / conv3 (virtual concat) / endconv shapes
    (1, 128, 0, 64, (4, 4, 16), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (2, 64, 64, 64, (3, 5, 18), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (1, 32, 32, 32, (4, 4, 16), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (2, 16, 0, 4, (8, 8, 16), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (1, 4, 0, 16, (8, 8, 16), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # k2-s2-p0 'down' convs of the V-Net (gather-GEMM kernels in bf16), one with a ragged coarse row
    (2, 16, 0, 32, (8, 8, 32), (2, 2, 2), (2, 2, 2), (0, 0, 0)),
    (1, 32, 0, 64, (4, 8, 40), (2, 2, 2), (2, 2, 2), (0, 0, 0)),
    (1, 64, 0, 128, (4, 4, 32), (2, 2, 2), (2, 2, 2), (0, 0, 0)),
    # 3x3 layers of the 2-D U-Nets on the MFMA kernels (bf16): every (cout tile, tile height) variant,
    # ragged tiles, two-input (virtual concat) layers, many K chunks
    (2, 32, 0, 32, (16, 16), (3, 3), (1, 1), (1, 1)),
    (1, 32, 32, 32, (20, 40), (3, 3), (1, 1), (1, 1)),
    (2, 64, 64, 64, (9, 18), (3, 3), (1, 1), (1, 1)),
    (1, 256, 0, 128, (8, 16), (3, 3), (1, 1), (1, 1)),
    (3, 128, 0, 256, (16, 16), (3, 3), (1, 1), (1, 1)),
    (8, 32, 0, 64, (128, 128), (3, 3), (1, 1), (1, 1)),
    (8, 32, 0, 32, (128, 128), (3, 3), (1, 1), (1, 1)),
    # the 2-D nets' first layer (1 input channel; 3 is above) and 2-class head
    (2, 1, 0, 32, (20, 24), (3, 3), (1, 1), (1, 1)),
    (2, 32, 0, 2, (16, 24), (1, 1), (1, 1), (0, 0)),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("prologue", [False, True])
def test_conv_fwd_dgrad_wgrad(case, dtype, prologue):
    from devis_amd import ops
    n, c1, c2, co, sp, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    nd = len(sp)
    x = torch.randn((n, c1 + c2) + sp, generator=g)
    w = torch.randn((co, c1 + c2) + k, generator=g) * 0.2
    b = torch.randn((co,), generator=g)
    sc = torch.rand((n, c1 + c2), generator=g) + 0.5
    sh = torch.randn((n, c1 + c2), generator=g) * 0.3
    if dtype == torch.bfloat16:      # compare on the rounded operands
        x, w = x.bfloat16().float(), w.bfloat16().float()
    conv = F.conv3d if nd == 3 else F.conv2d
    res = dy = None

    def reference(round_a):
        """torch fp32 on the same (bf16-rounded) operands -> (y + res, d/d(post-prologue input), dw, db)"""
        nonlocal res, dy
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True)
        a = xr
        if prologue:
            shape = (n, c1 + c2) + (1,) * nd
            # the kernels' prologue is ONE fused multiply-add in fp32 (product exact in fp64, one rounding)
            a = F.relu((xr.double() * sc.view(shape).double() + sh.view(shape).double()).float())
            if round_a:
                a = a.bfloat16().float()
        yr = conv(a, wr, br, stride=s, padding=p)
        if res is None:
            res = torch.randn(yr.shape, generator=g)
            dy = torch.randn(yr.shape, generator=g)
            if dtype == torch.bfloat16:
                res, dy = res.bfloat16().float(), dy.bfloat16().float()
        a.retain_grad()
        (yr + res).backward(dy)
        return (yr + res).detach(), a.grad, wr.grad, br.grad

    # bf16 + prologue: the MFMA kernels round relu(scale * x + shift) to bf16 (it is their matrix operand); the
    # generic kernels, which take the (shape, prologue) pairs no network uses, keep it in fp32.  Either is a correct
    # evaluation; forward and weight gradient are separate C-ABI calls (possibly of different kernel families), so
    # each of the two outputs the prologue enters (y, dw) is held to the closer ONE of the two references.
    refs = [reference(False)]
    if prologue and dtype == torch.bfloat16:
        refs.append(reference(True))

    geom = ops.ConvGeom(k, s, p)
    x1 = cl(x[:, :c1], dtype)
    x2 = cl(x[:, c1:], dtype) if c2 else None
    wp = ops.pack_conv_weight(w.cuda(), dtype)
    scg, shg = (sc.cuda().contiguous(), sh.cuda().contiguous()) if prologue else (None, None)
    y = ops.conv_fwd(x1, x2, wp, b.cuda(), geom, scg, shg, residual=cl(res, dtype))
    measure = (lambda r: bf16_ulps(uncl(y), r[0])) if dtype == torch.bfloat16 else (lambda r: rel_err(uncl(y), r[0]))
    yref, da, _, db_ref = min(refs, key=measure)
    close(uncl(y), yref, dtype, "y")
    dx1, dx2 = ops.conv_dgrad(cl(dy, dtype), wp, geom, x1.shape, c2)
    close(uncl(dx1), da[:, :c1], dtype, "dx1")
    if c2:
        close(uncl(dx2), da[:, c1:], dtype, "dx2")
    dw, db = ops.conv_wgrad(x1, x2, cl(dy, dtype), geom, w.shape, scg, shg)
    dw_ref = min((r[2] for r in refs), key=lambda r: rel_err(dw, r))
    close32(dw, dw_ref, dtype, "dw")
    close32(db, db_ref, dtype, "db")
    if k == (2, 2, 2):            # the skip gradient the V-Net adds in this dgrad's epilogue
        rs = torch.randn(x[:, :c1].shape, generator=g)
        if dtype == torch.bfloat16:
            rs = rs.bfloat16().float()
        dxr, _ = ops.conv_dgrad(cl(dy, dtype), wp, geom, x1.shape, c2, residual=cl(rs, dtype))
        close(uncl(dxr), da[:, :c1] + rs, dtype, "dx+skip")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 16, 8, (3, 4, 5), (2, 2, 2)), (2, 32, 16, (5, 6), (2, 2)), (1, 64, 64, (2, 2, 2), (2, 2, 2)),
                                  (2, 256, 128, (4, 4), (2, 2)), (1, 512, 256, (4, 8), (2, 2)),    # wide 2-D up-convs: channel slabs
                                  (5, 256, 128, (16, 16), (2, 2)), (3, 128, 64, (20, 24), (2, 2)),  # ... on the GEMM kernel (>= 1024 pixels)
                                  (2, 64, 32, (24, 24), (2, 2))])                                   # (dgrad only: K = 4*32)
def test_convT(case, dtype):
    from devis_amd import ops
    n, ci, co, sp, k = case
    g = torch.Generator().manual_seed(4)
    x = torch.randn((n, ci) + sp, generator=g)
    w = torch.randn((ci, co) + k, generator=g) * 0.2
    b = torch.randn((co,), generator=g)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    fn = F.conv_transpose3d if len(sp) == 3 else F.conv_transpose2d
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = fn(xr, wr, br, stride=2)
    dy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    yr.backward(dy)
    wp = ops.pack_convT_weight(w.cuda(), dtype)
    xg = cl(x, dtype)
    y = ops.convT_fwd(xg, wp, b.cuda(), k)
    close(uncl(y), yr.detach(), dtype, "y")
    dx = ops.convT_dgrad(cl(dy, dtype), wp, k, xg.shape)
    close(uncl(dx), xr.grad, dtype, "dx")
    dw, db = ops.convT_wgrad(xg, cl(dy, dtype), k, w.shape)
    close32(dw, wr.grad, dtype, "dw")
    close32(db, br.grad, dtype, "db")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["gn", "bn"])
@pytest.mark.parametrize("shape", [(2, 16, 4, 6, 5), (3, 32, 7, 9), (1, 128, 2, 3, 4), (2, 512, 4, 4)])
def test_norm_relu_fwd_bwd(shape, kind, dtype):
    from devis_amd import ops
    g = torch.Generator().manual_seed(8)
    n, c = shape[0], shape[1]
    x = torch.randn(shape, generator=g) * 2 + 0.5
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    gamma = torch.rand((c,), generator=g) + 0.5
    beta = torch.randn((c,), generator=g) * 0.2
    rm0 = torch.randn((c,), generator=g) * 0.1
    rv0 = torch.rand((c,), generator=g) + 0.5
    extra = torch.randn(shape, generator=g)
    dy = torch.randn(shape, generator=g)
    if dtype == torch.bfloat16:
        extra, dy = extra.bfloat16().float(), dy.bfloat16().float()
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    if kind == "gn":
        yr = F.relu(F.group_norm(xr, 8, gr, br, 1e-5))
    else:
        yr = F.relu(F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5))
    (yr + xr * 0).backward(dy.double())
    dxr = xr.grad + extra.double()

    xg = cl(x, dtype)
    gg, bg = gamma.cuda(), beta.cuda()
    if kind == "gn":
        sc, sh, mean, rstd = ops.gn_stats(xg, gg, bg)
        groups, bn = 8, False
    else:
        rmg, rvg = rm0.cuda(), rv0.cuda()
        sc, sh, mean, rstd = ops.bn_stats(xg, gg, bg, rmg, rvg, True)
        groups, bn = 0, True
        assert rel_err(rmg, rm) < 1e-5 and rel_err(rvg, rv) < 1e-5
    y = ops.affine_act(xg, sc, sh, relu=True)
    close(uncl(y), yr.detach(), dtype, "y")
    dx, dgm, dbt = ops.norm_relu_bwd(cl(dy, dtype), xg, sc, sh, gg, mean, rstd, groups, bn, extra=cl(extra, dtype))
    close(uncl(dx), dxr, dtype, "dx")
    close32(dgm, gr.grad, dtype, "dgamma")
    close32(dbt, br.grad, dtype, "dbeta")


def test_bn_eval_mode():
    from devis_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn((2, 32, 5, 6, 4), generator=g)
    gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
    rm, rv = torch.randn(32, generator=g), torch.rand(32, generator=g) + 0.5
    yr = F.relu(F.batch_norm(x, rm.clone(), rv.clone(), gamma, beta, False, 0.1, 1e-5))
    rmg, rvg = rm.cuda(), rv.cuda()
    sc, sh, _, _ = ops.bn_stats(cl(x), gamma.cuda(), beta.cuda(), rmg, rvg, False)
    y = ops.affine_act(cl(x), sc, sh, relu=True)
    assert rel_err(uncl(y), yr) < 1e-6
    assert torch.equal(rmg.cpu(), rm) and torch.equal(rvg.cpu(), rv)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 32, 8, 10), (1, 64, 9, 7), (2, 16, 5, 12)])
def test_maxpool(shape, dtype):
    from devis_amd import ops
    g = torch.Generator().manual_seed(3)
    x = F.relu(torch.randn(shape, generator=g))          # ReLU zeros create ties: first max must win
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xg = cl(x, dtype)
    y, idx = ops.maxpool2d_fwd(xg)
    assert torch.equal(uncl(y), yr.detach())
    dx = ops.maxpool2d_bwd(cl(dy, torch.float32).to(dtype), idx, xg.shape)
    ref = xr.grad if dtype == torch.float32 else xr.grad.bfloat16().float()
    assert rel_err(uncl(dx), ref) < (1e-7 if dtype == torch.float32 else 1e-2)
    # with the skip connection's gradient added in the same pass (odd sizes: the last row / column gets skip only)
    skip = torch.randn(shape, generator=g)
    if dtype == torch.bfloat16:
        skip = skip.bfloat16().float()
    dxs = ops.maxpool2d_bwd(cl(dy, torch.float32).to(dtype), idx, xg.shape, cl(skip, dtype))
    dyq = dy if dtype == torch.float32 else dy.bfloat16().float()
    xq = x.clone().requires_grad_(True)
    F.max_pool2d(xq, 2).backward(dyq)
    want = xq.grad + skip
    if dtype == torch.bfloat16:
        assert torch.equal(uncl(dxs), want.bfloat16().float())       # one rounding, like bf16(pool grad) + skip
    else:
        assert rel_err(uncl(dxs), want) < 1e-7


def test_layout_roundtrip_and_add():
    from devis_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, 5, 3, 4, 6), generator=g).cuda()
    for dt in (torch.float32, torch.bfloat16):
        y = ops.to_channels_last(x, dt)
        assert y.shape == (2, 3, 4, 6, 5)
        assert torch.equal(y.float(), x.permute(0, 2, 3, 4, 1).to(dt).float())
        assert torch.equal(ops.from_channels_last(y), x.to(dt).float())
        z = ops.add(y, y)
        assert torch.equal(z.float(), (y.float() * 2).to(dt).float())
    # the 4-channel input of the 3-D nets has a kernel of its own in bf16 (one 8-byte store per voxel)
    x4 = torch.randn((3, 4, 5, 7, 9), generator=g).cuda()
    y4 = ops.to_channels_last(x4, torch.bfloat16)
    assert torch.equal(y4.float(), x4.permute(0, 2, 3, 4, 1).to(torch.bfloat16).float())


def test_bad_arguments_fail_loudly():
    from devis_amd import ops, _lib
    with pytest.raises(_lib.DevisError):
        ops.to_channels_last(torch.zeros(1, 2, 4, 4, 4), torch.float32)          # CPU tensor: no fallback
    x = torch.zeros((1, 4, 4, 4, 16), device="cuda")
    wp = torch.zeros((16, 27, 8), device="cuda")
    with pytest.raises(_lib.DevisError):
        ops.conv_fwd(x, None, wp, None, ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1)))   # channel mismatch


# ------------------------------------------------------- statistics gathered in conv epilogues
STATS_CASES = [
    # (N, Cin, Cout, spatial, stride): shapes of the bf16 MFMA kernels that gather norm moments
    (2, 16, 16, (8, 8, 32), 1),      # march 8x32, full tiles
    (1, 16, 16, (6, 10, 20), 1),     # march 8x32, ragged tiles
    (2, 16, 16, (20, 16, 16), 1),    # march 16x16, several D chunks
    (1, 4, 16, (8, 8, 16), 1),       # first layer (Cin = 4)
    (2, 32, 32, (5, 9, 24), 1),      # generic 32-channel kernel, ragged
    (2, 64, 64, (4, 8, 16), 1),      # deep 64
    (1, 128, 128, (4, 4, 16), 1),    # deep 128
    (2, 64, 64, (5, 6, 20), 1),      # deep 64 / 128, ragged tiles (two cout groups per block / four K chunks)
    (2, 128, 128, (3, 5, 18), 1),
    (1, 16, 32, (8, 8, 32), 2),      # stride-2 (EnDown) variants
    (2, 16, 32, (24, 20, 36), 2),    # stride-2 plane march: several D chunks, ragged tiles
    (2, 32, 64, (18, 12, 40), 2),
    (2, 32, 64, (8, 8, 16), 2),
    (1, 64, 128, (4, 8, 16), 2),
]


@pytest.mark.parametrize("case", STATS_CASES)
@pytest.mark.parametrize("prologue,residual", [(False, False), (True, True)])
def test_conv_epilogue_forward_moments(case, prologue, residual):
    """conv_fwd(want_stats) leaves per-chunk {sum y, sum y^2}; summed over chunks they equal the moments of
    the stored bf16 output (fp64 on the CPU) to fp32-accumulation accuracy, and the output is unchanged."""
    from devis_amd import ops
    n, ci, co, sp, s = case
    if ci == 4 and (prologue or residual):
        pytest.skip("the first layer has neither prologue nor residual")
    if s == 2 and residual:
        pytest.skip("stride-2 layers have no residual")
    g = torch.Generator().manual_seed(11)
    x = cl(torch.randn((n, ci) + sp, generator=g), torch.bfloat16)
    w = torch.randn((co, ci, 3, 3, 3), generator=g) * 0.1
    b = torch.randn((co,), generator=g).cuda()
    geom = ops.ConvGeom((3, 3, 3), (s,) * 3, (1, 1, 1))
    wp = ops.pack_conv_weight(w.cuda(), torch.bfloat16)
    sc = (torch.rand((n, ci), generator=g) + 0.5).cuda() if prologue else None
    sh = (torch.randn((n, ci), generator=g) * 0.3).cuda() if prologue else None
    osp = geom.out_spatial(sp)
    res = cl(torch.randn((n, co) + osp, generator=g), torch.bfloat16) if residual else None
    y0 = ops.conv_fwd(x, None, wp, b, geom, sc, sh, residual=res)
    y, part = ops.conv_fwd(x, None, wp, b, geom, sc, sh, residual=res, want_stats=True)
    assert part is not None, "this shape must gather statistics in the epilogue"
    assert torch.equal(y, y0)
    buf, chunks = part
    assert buf.shape == (n, chunks, co, 2)
    got = buf.sum(1).cpu()
    yd = y.double().cpu().reshape(n, -1, co)
    want = torch.stack([yd.sum(1), (yd * yd).sum(1)], dim=-1)
    assert rel_err(got[..., 0], want[..., 0]) < 2e-5
    assert rel_err(got[..., 1], want[..., 1]) < 2e-5
    # and through the finalize kernel: same scale/shift as the separate-pass statistics
    gamma, beta = torch.rand(co, generator=g).cuda() + 0.5, torch.randn(co, generator=g).cuda()
    a = ops.gn_stats(y, gamma, beta, partials=part)
    r = ops.gn_stats(y, gamma, beta)
    for u, v in zip(a, r):
        assert rel_err(u, v) < 1e-4


STATS2D_CASES = [
    # (N, C1, C2, Cout, (H, W)): the 2-D MFMA kernels' tile variants, a two-input layer, ragged tiles
    (2, 32, 0, 32, (16, 16)), (1, 32, 32, 32, (20, 40)), (2, 64, 64, 64, (9, 18)), (3, 128, 0, 256, (16, 16)),
    (8, 32, 0, 64, (128, 128)),
]


@pytest.mark.parametrize("case", STATS2D_CASES)
@pytest.mark.parametrize("prologue", [False, True])
def test_conv2d_epilogue_moments(case, prologue):
    """2-D 3x3 layers: forward leaves {sum y, sum y^2} of the stored output, dgrad (single-input layers) the
    norm-backward moments {sum g, sum g*x}; outputs are unchanged by gathering them."""
    from devis_amd import ops
    n, c1, c2, co, sp = case
    g = torch.Generator().manual_seed(13)
    x1 = cl(torch.randn((n, c1) + sp, generator=g), torch.bfloat16)
    x2 = cl(torch.randn((n, c2) + sp, generator=g), torch.bfloat16) if c2 else None
    w = torch.randn((co, c1 + c2, 3, 3), generator=g) * 0.1
    b = torch.randn((co,), generator=g).cuda()
    geom = ops.ConvGeom((3, 3), (1, 1), (1, 1))
    wp = ops.pack_conv_weight(w.cuda(), torch.bfloat16)
    sc = (torch.rand((n, c1 + c2), generator=g) + 0.5).cuda() if prologue else None
    sh = (torch.randn((n, c1 + c2), generator=g) * 0.3).cuda() if prologue else None
    y0 = ops.conv_fwd(x1, x2, wp, b, geom, sc, sh)
    y, part = ops.conv_fwd(x1, x2, wp, b, geom, sc, sh, want_stats=True)
    assert part is not None, "this shape must gather statistics in the epilogue"
    assert torch.equal(y, y0)
    buf, chunks = part
    assert buf.shape == (n, chunks, co, 2)
    yd = y.double().cpu().reshape(n, -1, co)
    want = torch.stack([yd.sum(1), (yd * yd).sum(1)], dim=-1)
    got = buf.sum(1).cpu()
    assert rel_err(got[..., 0], want[..., 0]) < 2e-5
    assert rel_err(got[..., 1], want[..., 1]) < 2e-5
    if c2 or prologue:
        return
    dy = cl(torch.randn((n, co) + sp, generator=g), torch.bfloat16)
    scn = (torch.rand((n, c1), generator=g) + 0.5).cuda()
    shn = (torch.randn((n, c1), generator=g) * 0.3).cuda()
    dx0, _ = ops.conv_dgrad(dy, wp, geom, x1.shape)
    dx, _, part = ops.conv_dgrad(dy, wp, geom, x1.shape, norm=(x1, scn, shn))
    assert part is not None and torch.equal(dx, dx0)
    got = part[0].sum(1).cpu()
    xd = x1.double().cpu().reshape(n, -1, c1)
    gate = torch.addcmul(shn.cpu()[:, None, :], x1.float().cpu().reshape(n, -1, c1), scn.cpu()[:, None, :]) > 0
    gd = dx.double().cpu().reshape(n, -1, c1) * gate
    want = torch.stack([gd.sum(1), (gd * xd).sum(1)], dim=-1)
    assert rel_err(got[..., 0], want[..., 0]) < 5e-5
    assert rel_err(got[..., 1], want[..., 1]) < 5e-5


# 1x1 layers whose dgrad feeds a DeBlock's bn2 backward: endconv (16 -> 4) and DeUpCat.conv1 (2C -> C)
PW_STATS_CASES = [(2, 16, 4, (8, 8, 16), 0), (1, 32, 16, (4, 8, 16), 0), (2, 64, 32, (4, 4, 16), 0), (2, 16, 4, (32, 32, 32), 0)]


@pytest.mark.parametrize("case", [c for c in STATS_CASES if c[4] == 1 and c[1] == c[2]] + PW_STATS_CASES)
def test_conv_epilogue_backward_moments(case):
    """conv_dgrad(norm=(x, scale, shift)) leaves {sum g, sum g*x} with g = dx*[scale*x+shift > 0]: the
    moments of norm+ReLU backward, equal to the separate pass (devis_norm_bwd_moments_partials)."""
    from devis_amd import ops
    n, ci, co, sp, s = case
    k = 3 if s else 1                  # s == 0 marks the 1x1 cases
    g = torch.Generator().manual_seed(12)
    dy = cl(torch.randn((n, co) + sp, generator=g), torch.bfloat16)
    xref = cl(torch.randn((n, ci) + sp, generator=g), torch.bfloat16)
    w = torch.randn((co, ci, k, k, k), generator=g) * 0.1
    geom = ops.ConvGeom((k, k, k), (1, 1, 1), (k // 2,) * 3)
    wp = ops.pack_conv_weight(w.cuda(), torch.bfloat16)
    sc = (torch.rand((n, ci), generator=g) + 0.5).cuda()
    sh = (torch.randn((n, ci), generator=g) * 0.3).cuda()
    dx0, _ = ops.conv_dgrad(dy, wp, geom, xref.shape)
    dx, _, part = ops.conv_dgrad(dy, wp, geom, xref.shape, norm=(xref, sc, sh))
    assert part is not None, "this shape must gather statistics in the epilogue"
    assert torch.equal(dx, dx0)
    buf, chunks = part
    got = buf.sum(1).cpu()
    xd = xref.double().cpu().reshape(n, -1, ci)
    gate = (xd * sc.double().cpu()[:, None, :] + sh.double().cpu()[:, None, :]) > 0
    # the kernel gates on fmaf(x, sc, sh) in fp32: evaluate the gate the same way
    gate = torch.addcmul(sh.cpu()[:, None, :], xref.float().cpu().reshape(n, -1, ci), sc.cpu()[:, None, :]) > 0
    gd = dx.double().cpu().reshape(n, -1, ci) * gate
    want = torch.stack([gd.sum(1), (gd * xd).sum(1)], dim=-1)
    assert rel_err(got[..., 0], want[..., 0]) < 5e-5
    assert rel_err(got[..., 1], want[..., 1]) < 5e-5
    gamma = torch.rand(ci, generator=g).cuda() + 0.5
    mean, rstd = torch.randn((n, 8), generator=g).cuda() * 0.1, torch.rand((n, 8), generator=g).cuda() + 0.5
    a = ops.norm_relu_bwd(dx, xref, sc, sh, gamma, mean, rstd, 8, False, partials=part)
    r = ops.norm_relu_bwd(dx, xref, sc, sh, gamma, mean, rstd, 8, False)
    close(a[0].float().cpu(), r[0].float().cpu(), torch.bfloat16, "dx via epilogue moments vs separate pass")
    close32(a[1], r[1], torch.bfloat16, "dgamma")
    close32(a[2], r[2], torch.bfloat16, "dbeta")


@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (1, 12, 20, 36), (2, 40, 10, 48)])
@pytest.mark.parametrize("variant", ["stats", "residual", "plain"])
def test_conv_dgrad_with_norm_backward_prologue(shape, variant):
    """devis_conv_dgrad_normbwd = devis_norm_bwd_apply + devis_conv_dgrad(_stats) in one kernel: dx, the dh it writes
    out for the weight gradient and the epilogue moments carry the bits of the two calls (several D chunks, ragged
    tiles, planes at the volume's border)."""
    from devis_amd import ops
    n, d, h, w = shape
    g = torch.Generator().manual_seed(d * 100 + w)
    bf = lambda t: t.bfloat16().cuda()
    gg, xn = bf(torch.randn((n, d, h, w, 16), generator=g)), bf(torch.randn((n, d, h, w, 16), generator=g))
    sc = (torch.rand((n, 16), generator=g) + 0.5).cuda()
    sh = (torch.randn((n, 16), generator=g) * 0.3).cuda()
    coef = (torch.randn((3, n, 16), generator=g) * 0.5).cuda()
    wt = (torch.randn((16, 16, 3, 3, 3), generator=g) * 0.1).cuda()
    G3 = ops.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
    wp = ops.pack_conv_weight(wt, torch.bfloat16)
    wd = torch.empty((wp.shape[2], wp.shape[1], wp.shape[0]), dtype=wp.dtype, device=wp.device)
    from devis_amd import _lib
    _lib.check(_lib.load().devis_repack_conv_weight_dgrad(wp.data_ptr(), wd.data_ptr(), 16, 16, 27, _lib.DEVIS_BF16,
                                                          torch.cuda.current_stream().cuda_stream), "repack")
    res = bf(torch.randn((n, d, h, w, 16), generator=g)) if variant == "residual" else None
    xref = bf(torch.randn((n, d, h, w, 16), generator=g))
    sc2 = (torch.rand((n, 16), generator=g) + 0.5).cuda()
    sh2 = (torch.randn((n, 16), generator=g) * 0.3).cuda()
    norm = (xref, sc2, sh2) if variant == "stats" else None
    # the two calls
    dh_ref = torch.empty_like(gg)
    nv = d * h * w
    _lib.check(_lib.load().devis_norm_bwd_apply(gg.data_ptr(), xn.data_ptr(), sc.data_ptr(), sh.data_ptr(), coef[0].data_ptr(),
                                                coef[1].data_ptr(), coef[2].data_ptr(), None, dh_ref.data_ptr(), n, nv, 16,
                                                _lib.DEVIS_BF16, torch.cuda.current_stream().cuda_stream), "apply")
    out = ops.conv_dgrad(dh_ref, wp, G3, gg.shape, residual=res, wd=wd, norm=norm)
    dx_ref, part_ref = out[0], (out[2] if norm is not None else None)
    fused = ops.conv_dgrad_normbwd(gg, xn, sc, sh, coef, wp, wd, G3, residual=res, norm=norm)
    assert fused is not None
    dx, dh, part = fused
    torch.cuda.synchronize()
    assert torch.equal(dh, dh_ref)
    assert torch.equal(dx, dx_ref)
    if norm is not None:
        assert part[1] == part_ref[1] and torch.equal(part[0], part_ref[0])


@pytest.mark.parametrize("kernel,pad", [((1, 1, 1), (0, 0, 0)), ((5, 5, 5), (2, 2, 2)), ((3, 3, 3), (1, 1, 1))])
def test_conv_dgrad_normbwd_refuses_other_geometries(kernel, pad):
    """devis_conv_dgrad_normbwd runs a 27-tap pad-1 kernel: a 16->16 stride-1 layer of any other kernel size keeps
    the spatial shape too and must come back as DEVIS_ERR_UNSUPPORTED (-2) through the raw C ABI, before any launch
    (ADVICE r2); the 3x3x3 descriptor on the same buffers is accepted."""
    import ctypes as C
    from devis_amd import _lib, ops
    lib = _lib.load()
    n, d, h, w = 1, 4, 8, 32
    z = lambda *s: torch.zeros(s, device="cuda")
    g = torch.zeros((n, d, h, w, 16), dtype=torch.bfloat16, device="cuda")
    xn, dx, dh = torch.zeros_like(g), torch.zeros_like(g), torch.zeros_like(g)
    wd = torch.zeros((16, 27, 16), dtype=torch.bfloat16, device="cuda")
    sc, sh, ca, cb, cc = z(n, 16), z(n, 16), z(n, 16), z(n, 16), z(n, 16)
    desc = ops.ConvGeom(kernel, (1, 1, 1), pad).desc(n, (d, h, w), 16, 0, 16, torch.bfloat16)
    rc = lib.devis_conv_dgrad_normbwd(C.byref(desc), g.data_ptr(), xn.data_ptr(), sc.data_ptr(), sh.data_ptr(), ca.data_ptr(),
                                      cb.data_ptr(), cc.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), dh.data_ptr(), None,
                                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == (0 if kernel == (3, 3, 3) else -2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_batch_weight_packing_equals_the_per_tensor_calls(dtype):
    """devis_pack_weights_batch (one launch per step; row chunks through LDS for the conv modes, element chunks for the
    rest) against devis_pack_conv_weight / devis_pack_convT_weight / devis_repack_conv_weight_dgrad per tensor, bit for
    bit: 3-D and 2-D shapes of the nets, the fragment-order layers (64 / 128 channels x 27 taps), 1x1 layers (element
    chunks), rows at and above DEVIS_PACK_ROW_MAX (1024 x 9 > 8192: element chunks), ragged row lengths."""
    from devis_amd import _lib, ops
    g = torch.Generator().manual_seed(3)
    shapes = (("conv", (16, 4, 3, 3, 3)), ("conv", (32, 16, 3, 3, 3)), ("conv", (64, 64, 3, 3, 3)), ("conv", (128, 128, 3, 3, 3)),
              ("conv", (4, 16, 1, 1, 1)), ("convT", (64, 32, 2, 2, 2)), ("conv", (32, 64, 1, 1, 1)), ("conv", (32, 3, 3, 3)),
              ("convT", (256, 128, 2, 2)), ("conv", (512, 512, 3, 3)), ("conv", (64, 1024, 3, 3)), ("conv", (7, 5, 3, 3, 3)),
              ("conv", (128, 64, 3, 3, 3)), ("conv", (16, 16, 2, 2, 2)))
    ents = [("w%d" % i, torch.randn(shape, generator=g).cuda(), kind) for i, (kind, shape) in enumerate(shapes)]
    pk = ops.WeightPacker(ents, dtype, torch.device("cuda"))
    flags = pk.chunks[:, 0].cpu()
    assert bool((flags & _lib.PACK_ROW_FLAG).bool().any()) and not bool((flags & _lib.PACK_ROW_FLAG).bool().all())
    views = pk.pack()
    lib = _lib.load()
    for name, w, kind in ents:
        a, b = w.shape[0], w.shape[1]
        t = w.numel() // (a * b)
        fwd, dg = views[name]
        if kind == "conv":
            ref = ops.pack_conv_weight(w, dtype)
            assert torch.equal(fwd.reshape(-1), ref.reshape(-1)), name
            if dg is not None:
                refd = torch.empty_like(dg)
                ops.check(lib.devis_repack_conv_weight_dgrad(ops._p(ref), ops._p(refd), a, b, t, ops.dt_code(dtype), ops._stream()), "repack")
                assert torch.equal(dg.reshape(-1), refd.reshape(-1)), name + " dgrad"
        else:
            ref = ops.pack_convT_weight(w, dtype)
            assert torch.equal(fwd.reshape(-1), ref.reshape(-1)), name
            if dg is not None:
                refd = w.reshape(a, b, t).permute(0, 2, 1).contiguous().to(dtype).reshape(a, t * b)
                assert torch.equal(dg, refd), name + " dgrad"


@pytest.mark.parametrize("case", [(2, 16, 0, 16, (12, 20, 36), (3, 3, 3), (1, 1, 1)), (1, 32, 32, 64, (6, 10, 18), (1, 1, 1), (1, 1, 1)),
                                  (2, 16, 0, 32, (11, 9, 20), (3, 3, 3), (2, 2, 2)), (1, 64, 0, 64, (5, 6, 20), (3, 3, 3), (1, 1, 1)),
                                  (3, 1, 0, 32, (1, 20, 24), (1, 3, 3), (1, 1, 1)), (2, 3, 0, 32, (1, 12, 7), (1, 3, 3), (1, 1, 1)),
                                  (2, 32, 0, 2, (1, 16, 24), (1, 1, 1), (1, 1, 1)), (2, 32, 32, 32, (1, 20, 40), (1, 3, 3), (1, 1, 1))])
def test_fp32_weight_gradient_is_deterministic(case):
    """The fp32 parity mode's weight gradients (csrc/conv_f32_mfma.hip) are sums in a fixed order: two calls on the same
    inputs give the same bits (the generic kernels they replaced accumulated with float atomics), for every slab
    configuration of the kernel (16 x 16 with the waves splitting K, 2 x 2 slabs, strided, two inputs; the 2-D nets'
    one-plane tiles, their 1- and 3-channel first layers and 2-class head - no prologue on the thin inputs)."""
    from devis_amd import ops
    n, c1, c2, co, sp, k, s = case
    g = torch.Generator().manual_seed(11)
    geom = ops.ConvGeom(k, s, tuple(kk // 2 for kk in k))
    x1 = torch.randn((n,) + sp + (c1,), generator=g).cuda()
    x2 = torch.randn((n,) + sp + (c2,), generator=g).cuda() if c2 else None
    osp = tuple((sp[i] + 2 * (k[i] // 2) - k[i]) // s[i] + 1 for i in range(3))
    dy = torch.randn((n,) + osp + (co,), generator=g).cuda()
    sc = (torch.rand((n, c1 + c2), generator=g) + 0.5).cuda()
    sh = torch.randn((n, c1 + c2), generator=g).cuda()
    if c1 < 16:                                     # first layers have no norm in front of them
        sc, sh = None, None
    runs = [ops.conv_wgrad(x1, x2, dy, geom, (co, c1 + c2) + k, sc, sh) for _ in range(3)]
    for dw, db in runs[1:]:
        assert torch.equal(dw, runs[0][0]) and torch.equal(db, runs[0][1])
    # and they are the right numbers
    xa = torch.cat([t for t in (x1, x2) if t is not None], -1)
    if sc is not None:
        xa = torch.relu(xa * sc[:, None, None, None, :] + sh[:, None, None, None, :])
    w = torch.zeros((co, c1 + c2) + k, device="cuda", requires_grad=True)
    b = torch.zeros((co,), device="cuda", requires_grad=True)
    F.conv3d(xa.permute(0, 4, 1, 2, 3), w, b, stride=s, padding=tuple(kk // 2 for kk in k)).backward(dy.permute(0, 4, 1, 2, 3))
    close32(runs[0][0], w.grad, torch.float32, "fp32 dw")
    close32(runs[0][1], b.grad, torch.float32, "fp32 db")
