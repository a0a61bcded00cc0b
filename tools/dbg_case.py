import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from tests.test_gpu_ops import CONV_CASES, cl, uncl
from devis_amd import ops
for ci in (5, 20, 34, 3):
    case = CONV_CASES[ci]
    n, c1, c2, co, sp, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    nd = len(sp)
    x = torch.randn((n, c1 + c2) + sp, generator=g)
    w = torch.randn((co, c1 + c2) + k, generator=g) * 0.2
    b = torch.randn((co,), generator=g)
    sc = torch.rand((n, c1 + c2), generator=g) + 0.5
    sh = torch.randn((n, c1 + c2), generator=g) * 0.3
    x, w = x.bfloat16().float(), w.bfloat16().float()
    conv = F.conv3d if nd == 3 else F.conv2d
    shape = (n, c1 + c2) + (1,) * nd
    a = F.relu((x.double() * sc.view(shape).double() + sh.view(shape).double()).float())
    y64 = conv(a.double(), w.double(), b.double(), stride=s, padding=p)
    y32 = conv(a, w, b, stride=s, padding=p)
    ab = a.bfloat16().float()
    y64b = conv(ab.double(), w.double(), b.double(), stride=s, padding=p)
    geom = ops.ConvGeom(k, s, p)
    x1 = cl(x[:, :c1], torch.bfloat16); x2 = cl(x[:, c1:], torch.bfloat16) if c2 else None
    wp = ops.pack_conv_weight(w.cuda(), torch.bfloat16)
    y = uncl(ops.conv_fwd(x1, x2, wp, b.cuda(), geom, sc.cuda().contiguous(), sh.cuda().contiguous()))
    # fp32 mode too
    x1f = cl(x[:, :c1]); x2f = cl(x[:, c1:]) if c2 else None
    wpf = ops.pack_conv_weight(w.cuda(), torch.float32)
    yf = uncl(ops.conv_fwd(x1f, x2f, wpf, b.cuda(), geom, sc.cuda().contiguous(), sh.cuda().contiguous()))
    print(ci, case)
    print("  torch32 vs 64:", (y32 - y64).abs().max().item(), " max|y|", y64.abs().max().item())
    print("  ours fp32 vs 64:", (yf - y64).abs().max().item())
    e = (y.double() - y64.bfloat16().double()).abs(); eb = (y.double() - y64b.bfloat16().double()).abs()
    print("  ours bf16 vs round(y64):", e.max().item(), " vs round(y64 with bf16 a):", eb.max().item())
    i = e.argmax(); print("   worst at", i.item(), "ours", y.flatten()[i].item(), "ref64", y64.flatten()[i].item(), "ref64b", y64b.flatten()[i].item())
