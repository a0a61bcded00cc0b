"""CPU suite (no GPU): the oracle against the golden vectors produced from the real reference, the
host-side logic, and the C-ABI library's symbol table."""
import hashlib
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import brats_args, rel_err, sd_hash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ oracle vs golden
@pytest.mark.parametrize("tag,K", [("head3d", 4), ("head2d", 2)])
@pytest.mark.parametrize("variant", ["dce", "dceu"])
@pytest.mark.parametrize("epoch", [1, 25, 50, 80])
def test_oracle_head_matches_reference_vectors(golden, tag, K, variant, epoch):
    from oracle import devis_oracle as O
    G = golden("head")
    z = torch.from_numpy(G[tag + "_logits"]).requires_grad_(True)
    lab = torch.from_numpy(G[tag + "_labels"])
    ev = F.softplus(z)
    fn = O.dce_evidence_loss if variant == "dce" else O.dce_evidence_u_loss
    pv = fn(lab, ev + 1, K, epoch, 50, 100, 1e-10, False, ev, torch.softmax(z, 1))
    loss = pv.mean()
    loss.backward()
    pre = "%s_%s_e%d_" % (tag, variant, epoch)
    assert rel_err(pv.detach(), G[pre + "per_voxel"]) < 1e-6
    assert abs(loss.item() - float(G[pre + "loss"])) < 1e-6
    assert rel_err(z.grad, G[pre + "grad"]) < 1e-5
    assert rel_err(ev.detach(), G[tag + "_evidence"]) < 1e-6


def test_oracle_kl_ce_mse(golden):
    from oracle import devis_oracle as O
    G = golden("head")
    ev = torch.from_numpy(G["head3d_evidence"])
    assert rel_err(O.kl_dirichlet(O._rows(ev + 1), 4), G["head3d_KL"]) < 1e-6
    al, lab = torch.from_numpy(G["ce_alpha"]), torch.from_numpy(G["ce_labels"])
    assert rel_err(O.ce_loss(lab, al, 4, 10, 50), G["ce_loss"]) < 1e-6
    assert rel_err(O.mse_loss(lab.view(-1), O._rows(al), 4, 10, 50), G["mse_loss"]) < 1e-6


def test_oracle_infer_head(golden):
    from oracle import devis_oracle as O
    G = golden("head")
    for tag, K in (("head3d", 4), ("head2d", 2)):
        pred, u, prob, mne = O.infer_head(torch.from_numpy(G[tag + "_evidence"]), K)
        assert np.array_equal(pred.numpy().astype(np.uint8), G[tag + "_pred"])
        assert rel_err(u, G[tag + "_u"]) < 1e-6 and rel_err(mne, G[tag + "_mne"]) < 1e-5


def test_oracle_unet3d_train_step(golden):
    """Seeded init of the PRODUCT module == the reference's; the oracle reproduces loss, grads, BN buffers."""
    from oracle import devis_oracle as O
    from devis_amd import Unet
    G = golden("unet3d")
    pre = "s1_e25_"
    shape = tuple(int(v) for v in G[pre + "shape"])
    torch.manual_seed(1)
    net = Unet(4, 16, 4)
    assert sd_hash(net.state_dict()) == str(G[pre + "sd_hash"])
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    for k in sd:
        if sd[k].is_floating_point():
            sd[k].requires_grad_(k.split('.')[-1] in ("weight", "bias"))
    g = torch.Generator().manual_seed(1001)
    x = torch.randn(shape, generator=g)
    y = torch.randint(0, 4, (shape[0],) + shape[2:], generator=g)
    ev, loss = O.tmsu_forward(lambda t: O.unet3d_forward(sd, t, True), x, y, 25, 'train', 'BraTS', 4, 50, 100)
    loss.backward()
    assert abs(loss.item() - float(G[pre + "loss"])) < 1e-6
    assert rel_err(ev.detach()[..., ::2], G[pre + "evidence"]) < 1e-5
    assert rel_err(sd["endconv.weight"].grad, G[pre + "grad_endconv.weight"]) < 1e-4
    assert rel_err(sd["DeBlock2.bn2.running_var"], G[pre + "buf_DeBlock2.bn2.running_var"]) < 1e-6


def test_oracle_unet2d_rand_padding(golden):
    from oracle import devis_oracle as O
    from devis_amd import Unet2D
    G = golden("unet2d")
    tag = "u2d_w76"
    shape = tuple(int(v) for v in G[tag + "_shape"])
    torch.manual_seed(3)
    net = Unet2D(3, 2)
    assert sd_hash(net.state_dict()) == str(G[tag + "_sd_hash"])
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(2000)
    x = torch.randn(shape, generator=g)
    torch.manual_seed(99)
    log = []
    with torch.no_grad():
        logits = O.unet2d_forward(sd, x, True, False, log)
    assert rel_err(logits, G[tag + "_logits"]) < 1e-5
    assert [list(s) + [0] * (4 - len(s)) for s in log] == G[tag + "_rand_shapes"].tolist()


def test_oracle_tailor_and_concat(golden):
    from oracle import devis_oracle as O
    G = golden("tailor")
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 4, 240, 240, 160), generator=g)
    a = O.tailor_and_concat(x, lambda t: t * 2 + 1)
    assert hashlib.sha256(a.contiguous().numpy().tobytes()).hexdigest() == str(G["fake_sha256"])
    idx = torch.from_numpy(G["fake_idx"])
    assert np.array_equal(a.reshape(-1)[idx].numpy(), G["fake_val"])


# ------------------------------------------------------------------ host logic
def test_tailor_regions_tile_the_volume():
    from devis_amd.predict import CROPS, patch_regions
    cover = np.zeros((240, 240, 155), dtype=np.int32)
    for i in range(len(CROPS)):
        dst, src = patch_regions(i)
        cover[dst] += 1
        for d, s in zip(dst, src):
            assert d.stop - d.start == s.stop - s.start
    assert (cover == 1).all()


def test_product_tailor_matches_oracle_on_cpu_callable():
    from oracle import devis_oracle as O
    from devis_amd.predict import tailor_and_concat
    g = torch.Generator().manual_seed(6)
    x = torch.randn((1, 4, 240, 240, 160), generator=g)
    fn = lambda t: t * 3 - 1            # noqa: E731  (plain torch callable: the stitch itself is only copies)
    assert torch.equal(tailor_and_concat(x, fn), O.tailor_and_concat(x, fn))
    x1 = x[:, :1].contiguous()
    fn4 = lambda t: torch.cat([t, t * 2, t * 3, t * 4], 1)   # noqa: E731
    assert torch.equal(tailor_and_concat(x1, fn4), O.tailor_and_concat(x1, fn4))


def test_poly_lr_and_args():
    from oracle import devis_oracle as O
    from devis_amd.train import adjust_learning_rate, getArgs
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=0.002)
    for ep in (0, 1, 25, 98):
        adjust_learning_rate(opt, ep, 99, 0.002)
        assert opt.param_groups[0]['lr'] == O.poly_lr(0.002, ep, 99)
    a = getArgs(["--dataset", "BraTS"])
    assert (a.lr, a.lambda_epochs, a.end_epochs, a.batch_size, a.num_classes) == (0.002, 50, 99, 2, 4)
    assert a.model_name == 'U' and a.input_modality == 'four' and a.Uncertainty_Loss is False


def test_tmsu_selection_and_errors():
    from devis_amd import TMSU, Unet, Unet2D, Unet2DDRIVE
    m = TMSU(brats_args())
    assert isinstance(m.backbone, Unet) and m.classes == 4 and m.total_epochs == 100 and m.eps == 1e-10
    assert m.disentangle is False and m.lambda_epochs == 50
    assert isinstance(TMSU(brats_args(dataset='DRIVE', num_classes=2)).backbone, Unet2DDRIVE)
    assert isinstance(TMSU(brats_args(dataset='HC', num_classes=2)).backbone, Unet2D)
    with pytest.raises(NameError):
        TMSU(brats_args(dataset='nope'))
    with pytest.raises(NameError):
        TMSU(brats_args(model_name='XYZ'))
    keys = list(m.state_dict().keys())
    assert keys[0] == 'backbone.InitConv.conv.weight' and 'backbone.DeBlock2.bn2.num_batches_tracked' in keys


def test_no_cpu_fallback():
    from devis_amd import Unet, _lib, ops
    with pytest.raises(_lib.DevisError):
        ops.to_channels_last(torch.zeros(1, 4, 8, 8, 8), torch.float32)
    with pytest.raises(_lib.DevisError):
        Unet(4, 16, 4)(torch.zeros(1, 4, 8, 8, 8))


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from devis_amd import _lib
    header = open(os.path.join(ROOT, "include", "devis_hip.h")).read()
    declared = set(re.findall(r"\b(devis_[a-zA-Z0-9_]+)\s*\(", header))
    declared -= {"devis_status", "devis_dtype", "devis_label_dtype"}
    assert len(declared) > 25
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libdevis_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "%s has no ctypes signature" % name
    assert lib.devis_version() >= 100
    assert lib.devis_evid_workspace_bytes(1, 8, 4) > 0       # pure host arithmetic, no GPU touched


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """include/devis_hip.h is the boundary a non-Python host binds: it must compile as C99, and a C
    program that includes it must link against libdevis_hip.so and run its host-only entry points (no GPU touched)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "devis_amd")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, "devis_hip.h")], check=True)
    src = tmp_path / "host.c"
    src.write_text(
        '#include <stdio.h>\n#include "devis_hip.h"\n'
        "int main(void) {\n"
        "  devis_conv_desc d = {0};\n"
        "  d.N = 1; d.ID = d.IH = d.IW = 8; d.OD = d.OH = d.OW = 8; d.C1 = 16; d.Cout = 16;\n"
        "  d.KD = d.KH = d.KW = 3; d.SD = d.SH = d.SW = 1; d.PD = d.PH = d.PW = 1; d.dtype = DEVIS_BF16;\n"
        '  printf("%d %zu %zu\\n", devis_version(), devis_conv_wgrad_workspace_bytes(&d), devis_evid_workspace_bytes(1, 512, 4));\n'
        "  return devis_conv_fwd(&d, 0, 0, 0, 0, 0, 0, 0, 0, 0) == DEVIS_ERR_BAD_ARG ? 0 : 1;\n"
        "}\n")
    exe = tmp_path / "host"
    subprocess.run([gcc, "-std=c99", "-Wall", "-I", inc, str(src), "-L", libdir, "-ldevis_hip", "-Wl,-rpath," + libdir,
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    ver, wg, ev = out.stdout.split()
    assert int(ver) >= 101 and int(wg) > 0 and int(ev) > 0


# ------------------------------------------------------------------ evaluation counts (§8(f)-3)
def _evalmaps_case(golden, tag):
    import numpy as np
    g = golden("evalmaps")
    shape, K = tuple(int(v) for v in g[tag + "_shape"]), int(g[tag + "_K"])
    gen = torch.Generator().manual_seed(21)
    target = torch.randint(0, K, shape, generator=gen)
    pred = torch.where(torch.rand(shape, generator=gen) < 0.7, target, torch.randint(0, K, shape, generator=gen))
    u = torch.rand(shape, generator=gen)
    return g, K, pred, target, u.numpy().astype(np.float32)


@pytest.mark.parametrize("tag", ["brats", "lits", "bin"])
def test_eval_counts_oracle_matches_reference_goldens(golden, tag):
    """oracle.eval_counts + the metric helpers reproduce what the reference's numpyfunctions.uncertainty /
    predict.dice_score / mIOU / softmax_output_dice returned on the same seeded volumes (integers exact)."""
    import numpy as np
    from oracle import devis_oracle as O
    from devis_amd import predict as P
    g, K, pred, target, u = _evalmaps_case(golden, tag)
    conf, ucount, sums = O.eval_counts(pred.numpy(), target.numpy(), u, O.UNCERTAINTY_THRESHOLDS, K)
    assert np.array_equal(conf, g[tag + "_conf"]) and np.array_equal(ucount, g[tag + "_ucount"])
    assert np.allclose(sums, g[tag + "_sums"], rtol=1e-12)
    # the product's host helpers work from the confusion counts alone
    assert abs(P.dice_score(conf, range(1, K)) - float(g[tag + "_fg_dice"])) < 1e-12
    assert abs(P.mIOU(conf, range(1, K)) - float(g[tag + "_fg_miou"])) < 1e-12
    if K == 4:
        assert np.allclose(P.softmax_output_dice(conf), g[tag + "_dice3"], rtol=0, atol=1e-12)
    if K == 3:
        assert np.allclose(P.softmax_output_litsdice(conf), g[tag + "_dice2"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_vnet2d_train_step(golden, seed):
    """oracle.vnet2d_forward against the numbers the REAL TMSU(dataset='DRIVE', model_name='V') produced; the
    product module's seeded init and state_dict keys equal the reference's."""
    from oracle import devis_oracle as O
    from devis_amd.vnet2d import VNet2D
    G = golden("vnet2d")
    pre = "s%d_" % seed
    shape = tuple(int(v) for v in G[pre + "shape"])
    torch.manual_seed(seed)
    net = VNet2D(3, 2, 32, compute_dtype=torch.float32)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert list(sd.keys()) == [str(k) for k in G[pre + "keys"]]
    assert sd_hash(sd) == str(G[pre + "sd_hash"])
    for k in sd:
        if sd[k].is_floating_point():
            sd[k].requires_grad_(True)
    g = torch.Generator().manual_seed(5000 + seed)
    x = torch.randn(shape, generator=g)
    y = (torch.rand((shape[0], 1) + shape[2:], generator=g) > 0.5).long()
    log = []
    torch.manual_seed(99)
    ev, loss = O.tmsu_forward(lambda t: O.vnet2d_forward(sd, t, log), x, y, 25, 'train', 'DRIVE', 2, 50, 100)
    loss.backward()
    assert abs(loss.item() - float(G[pre + "loss"])) < 1e-6
    assert rel_err(ev.detach(), G[pre + "evidence"]) < 1e-6
    want = [tuple(int(v) for v in r if v) for r in G[pre + "rand_shapes"] if any(r)]
    assert [tuple(t) for t in log] == want
    k = "block_five.conv.3.weight"
    assert rel_err(sd[k].grad.reshape(-1)[::37], G[pre + "grad_" + k]) < 1e-4


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_unetdrop_matches_reference_goldens(golden, seed):
    """oracle.unetdrop3d_forward, fed the dropout noise re-drawn from the same RNG state, against the numbers the REAL
    reference Unetdrop produced (make_goldens.py asserted bit equality); the product class initialises like it."""
    from oracle import devis_oracle as O
    from devis_amd import Unetdrop
    G = golden("unetdrop")
    pre = "s%d_" % seed
    shape = tuple(int(v) for v in G[pre + "shape"])
    train = bool(int(G[pre + "train"]))
    torch.manual_seed(seed)
    net = Unetdrop(4, 16, 4, compute_dtype=torch.float32)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert sd_hash(sd) == str(G[pre + "sd_hash"])
    for k in sd:
        if sd[k].is_floating_point() and k.split(".")[-1] in ("weight", "bias"):
            sd[k].requires_grad_(True)
    x = torch.randn(shape, generator=torch.Generator().manual_seed(5000 + seed))
    torch.manual_seed(1000 + seed)
    masks = O.unetdrop_masks(x.shape, train)
    assert np.array_equal((masks["init"].reshape(shape[0], 16) != 0).numpy(), G[pre + "keep_init"])
    assert (masks["d1"] is None) == (not train)
    y = O.unetdrop3d_forward(sd, x, train, masks)
    y.square().mean().backward()
    assert torch.equal(y.detach(), torch.from_numpy(G[pre + "logits"]))
    k = "EnBlock3_2.conv1.weight"
    assert np.array_equal(sd[k].grad.reshape(-1)[::37].numpy(), G[pre + "grad_" + k])
