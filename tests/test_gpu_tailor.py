"""GPU: sliding-window stitch (predict.py:159-191) index arithmetic vs golden, and TMSU val mode."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_tailor_index_arithmetic(golden):
    from devis_amd import tailor_and_concat
    G = golden("tailor")
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 4, 240, 240, 160), generator=g).cuda()
    a = tailor_and_concat(x, lambda t: t * 2 + 1).cpu()
    assert a.shape == (1, 4, 240, 240, 155)
    assert hashlib.sha256(a.contiguous().numpy().tobytes()).hexdigest() == str(G["fake_sha256"])

    def pos(t):
        d, h, w = t.shape[2:]
        base = (torch.arange(d).view(d, 1, 1) * 1e4 + torch.arange(h).view(1, h, 1) * 1e2
                + torch.arange(w).view(1, 1, w)).float().to(t.device)
        return t * 0 + base
    b = tailor_and_concat(x, pos).cpu()
    assert hashlib.sha256(b.contiguous().numpy().tobytes()).hexdigest() == str(G["pos_sha256"])


def test_tmsu_val_mode_matches_oracle_on_one_patch():
    """val-mode TMSU on a full volume; patch 0's region must equal the oracle's patch-0 forward."""
    from oracle import devis_oracle as O
    from devis_amd import TMSU
    from tests.util import brats_args, rel_err
    torch.manual_seed(0)
    m = TMSU(brats_args(compute_dtype=torch.float32))
    sd = {k: v.detach().cpu().clone() for k, v in m.backbone.state_dict().items()}
    m.eval()
    g = torch.Generator().manual_seed(9)
    x = torch.randn((1, 4, 240, 240, 160), generator=g)
    with torch.no_grad():
        ev = m(x.cuda(), None, 1, 'val', 'BraTS')
    assert ev.shape == (1, 4, 240, 240, 155)
    with torch.no_grad():
        ref = torch.nn.functional.softplus(O.unet3d_forward(sd, x[..., :128, :128, :128], False))
    from tests.util import FP32_BAR, check
    check(rel_err(ev[..., :128, :128, :128], ref), FP32_BAR, "evidence")


def test_sliding_window_infer_equals_stitch_then_head():
    """Fused per-patch head on channels-last logits + region placement == tailor_and_concat of NCDHW logits
    followed by the head (the reference's order, trainUMIS.py:283-290): bit-exact, also when the patches
    are split between two (emulated) ranks and the maps are summed."""
    from oracle.devis_oracle import UNCERTAINTY_THRESHOLDS as THR
    from devis_amd import TMSU, ops, tailor_and_concat
    from devis_amd.predict import CROPS, crop, patch_regions, sliding_window_infer
    from devis_amd.parallel import shard_patches
    from tests.util import brats_args
    for dtype in (torch.float32, torch.bfloat16):
        torch.manual_seed(1)
        m = TMSU(brats_args(compute_dtype=dtype))
        m.eval()
        g = torch.Generator().manual_seed(10)
        x = torch.randn((1, 4, 240, 240, 160), generator=g).cuda()
        with torch.no_grad():
            ref = ops.evid_infer(tailor_and_concat(x, m.backbone).contiguous(), False, THR, want_evidence=True,
                                 want_mne=True)
            one = sliding_window_infer(x, m.backbone, THR, want_evidence=True, want_mne=True)
            # two emulated ranks: each runs its shard only (no process group), then the maps are added
            halves = []
            for r in range(2):
                out = {}
                for i in shard_patches(len(CROPS), r, 2):
                    maps = ops.evid_infer(m.backbone.forward_channels_last(crop(x, i)), True, THR, want_evidence=False)
                    dst, src = patch_regions(i)
                    for k, t in maps.items():
                        if t is None:
                            continue
                        if k not in out:
                            out[k] = torch.zeros(tuple(t.shape[:t.dim() - 3]) + (240, 240, 155), dtype=t.dtype, device=t.device)
                        out[k][(Ellipsis,) + dst] = t[(Ellipsis,) + src]
                halves.append(out)
        for k in ("pred", "u", "mne", "umask", "evidence"):
            assert one[k].shape == ref[k].shape and torch.equal(one[k], ref[k]), (str(dtype), k)
        # all 8 patches as one batch: the same bits in fp32; in bf16 tile shapes / statistics chunks depend on the
        # batch size, so the maps agree to bf16 rounding (u within 2e-2 relative, argmax flips only at near-ties)
        with torch.no_grad():
            bat = sliding_window_infer(x, m.backbone, THR, want_evidence=True, want_mne=True, patch_batch=8)
        if dtype == torch.float32:
            for k in ("pred", "u", "mne", "umask", "evidence"):
                assert torch.equal(bat[k], ref[k]), (str(dtype), k, "batched")
        else:
            assert float(((bat["u"] - ref["u"]).abs() / ref["u"].abs().clamp_min(1e-3)).max()) < 2e-2
            assert float((bat["pred"] != ref["pred"]).float().mean()) < 2e-3
        for k in ("pred", "u", "umask"):
            both = halves[0][k].to(torch.int32) + halves[1][k].to(torch.int32) if k != "u" else halves[0][k] + halves[1][k]
            assert torch.equal(both.to(ref[k].dtype), ref[k]), (str(dtype), k)
        # supports are disjoint
        assert int(((halves[0]["u"] != 0) & (halves[1]["u"] != 0)).sum()) == 0


def test_test_mode_driver_loads_checkpoint_and_reports_device_side_metrics(tmp_path):
    """train.test() (trainUMIS.py:333-...): the checkpoint val() wrote is the one it loads (reference file name),
    and its dice / IoU / uncertainty means equal the oracle's numpy restatement on host copies of the same maps."""
    from oracle import devis_oracle as O
    from devis_amd import TMSU, train as T, predict as P
    from tests.util import brats_args
    torch.manual_seed(0)
    args = brats_args(compute_dtype=torch.float32, save_dir=str(tmp_path), folder='folder0', test_epoch=3, mode='test',
                      graph_streams=3)          # val()/test() replay the patches from HIP graphs; `want` below is eager
    m = TMSU(args)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 4, 240, 240, 160), generator=g)
    noised = x + 0.5 * torch.randn((1, 4, 240, 240, 160), generator=g)
    target = torch.randint(0, 4, (1, 240, 240, 160), generator=g)
    # val() with best_dice = -1 always checkpoints under the reference's name
    _, best = T.val(args, m, 3, -1.0, [(x, target)])
    assert (tmp_path / "U_BraTS_folder0_epoch_3.pth").exists()
    want = T.predict_volume(m, x.cuda(), 'BraTS', T.UEO_THRESHOLDS)
    want = {k: v.clone() for k, v in want.items() if torch.is_tensor(v)}
    with torch.no_grad():                                   # spoil the weights: test() must restore them
        for p in m.parameters():
            p.mul_(0.5)
    lines = []
    res = T.test(args, m, [(x, noised, target)], log=lines.append)
    assert args.start_epoch == 3 and lines[0].startswith('Successfully load checkpoint')
    got = T.predict_volume(m, x.cuda(), 'BraTS', T.UEO_THRESHOLDS)
    assert torch.equal(got["pred"], want["pred"]) and torch.equal(got["u"], want["u"])
    t = target[0, ..., :155].numpy()
    conf, ucount, sums = O.eval_counts(want["pred"].cpu().numpy().reshape(t.shape), t,
                                       want["u"].cpu().numpy().reshape(t.shape), T.UEO_THRESHOLDS, 4)
    assert np.array_equal(res["clean_ucount"], ucount)
    assert np.allclose(res["clean_dice"], P.softmax_output_dice(conf), atol=1e-12)
    assert abs(res["clean_dice"][0] - 2.0 * ((want["pred"].cpu().numpy().reshape(t.shape) > 0) & (t > 0)).sum()
               / ((want["pred"].cpu().numpy() > 0).sum() + (t > 0).sum())) < 1e-9
    assert abs(res["clean_mean_u"] - sums[0] / t.size) < 1e-6
    assert abs(res["clean_mne"] - float(want["mne"].double().mean())) < 1e-9
    assert res["regions"] == ('WT', 'TC', 'ET') and res["runtime"] > 0
    assert abs(res["noise_mean_u"] - res["clean_mean_u"]) > 0          # the noised input went through too


def test_full_volume_config5_inference_vs_oracle_live():
    """BASELINE configs[4] at full size: one 4x240x240x160 volume through predict_volume (8 patches, backbone in
    eval mode + fused head, fp32) against the oracle's tailor_and_concat + infer_head run live on the host cores.
    evidence 1e-5 rel-max; u and mne 1e-5 abs; the argmax mask and the eleven u-threshold masks are integers and
    must agree wherever the oracle's own decision margin exceeds the evidence tolerance (every voxel that
    disagrees must be such a near-tie, and there must be next to none: the counts are recorded - bit-exactness
    on a 8.9 M-voxel random volume would need two fp32 evaluation orders to round every near-tie alike; it is
    asserted on the golden volumes of tests/test_gpu_head.py, whose generator guards the margins)."""
    from oracle import devis_oracle as O
    from devis_amd import TMSU, train as T
    from tests.util import brats_args, rel_err
    torch.manual_seed(0)
    m = TMSU(brats_args(compute_dtype=torch.float32)).eval()
    sd = {k: v.detach().cpu().clone() for k, v in m.backbone.state_dict().items()}
    g = torch.Generator().manual_seed(31)
    x = torch.randn((1, 4, 240, 240, 160), generator=g)
    thr = O.UNCERTAINTY_THRESHOLDS
    out = T.predict_volume(m, x.cuda(), 'BraTS', thr)
    with torch.no_grad():
        ev = O.tmsu_forward(lambda t: O.unet3d_forward(sd, t, False), x, None, 0, 'test', 'BraTS', 4, 50, 100)
        pred, u, prob, mne = O.infer_head(ev, 4)
    assert out["evidence"].shape == ev.shape == (1, 4, 240, 240, 155)
    from tests.util import FP32_BAR, check
    check(rel_err(out["evidence"].cpu(), ev), FP32_BAR, "evidence")
    check((out["u"].cpu().reshape(u.shape) - u).abs().max(), 1e-5, "u abs")
    check(rel_err(out["u"].cpu().reshape(u.shape), u), FP32_BAR, "u")
    check((out["mne"].cpu().reshape(mne.shape) - mne).abs().max(), 1e-5, "mne abs")
    ours = out["pred"].cpu().reshape(pred.shape).long()
    diff = ours != pred
    top2 = prob.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    assert bool((margin[diff] < 1e-4).all()), int(diff.sum())
    check(int(diff.sum()) / diff.numel(), 1e-5, "argmax mask: fraction of voxels that differ (near-ties)")
    um = out["umask"].cpu().reshape(u.shape).int()
    for j, t in enumerate(thr):
        want = u > np.float32(t)
        d = ((um >> j) & 1).bool() != want
        assert bool(((u[d] - np.float32(t)).abs() < 1e-5).all()), (t, int(d.sum()))
        check(int(d.sum()) / d.numel(), 1e-5, "u > %.2f mask: fraction of voxels that differ (near-ties)" % t)
    # The shipped bf16 path on the same volume and weights: evidence / u to the bf16 bar, and the mask agreement as a
    # recorded figure.  A random-init network's class probabilities are close together, so the fraction that differs
    # here is an upper bound for a trained one; every voxel that differs must be one whose oracle margin is within
    # twice the largest probability error of the bf16 run (an argmax cannot flip otherwise).
    from tests.util import BF16_BAR
    mb = TMSU(brats_args(compute_dtype=torch.bfloat16)).eval()
    mb.load_state_dict(m.state_dict())
    outb = T.predict_volume(mb, x.cuda(), 'BraTS', thr)
    evb = outb["evidence"].cpu()
    check(rel_err(evb, ev), BF16_BAR, "bf16 evidence, full volume")
    check((outb["u"].cpu().reshape(u.shape) - u).abs().max(), BF16_BAR, "bf16 u abs, full volume")
    probb = (evb + 1) / (evb + 1).sum(1, keepdim=True)
    dp = float((probb - prob.reshape(probb.shape)).abs().max())
    diffb = outb["pred"].cpu().reshape(pred.shape).long() != pred
    assert bool((margin[diffb] <= 2 * dp).all()), (dp, float(margin[diffb].max()))
    check(int(diffb.sum()) / diffb.numel(), 1e-2, "bf16 argmax mask: fraction of voxels that differ from the fp32 oracle (random-init weights)")
    umb = outb["umask"].cpu().reshape(u.shape).int()
    worst = max(int((((umb >> j) & 1).bool() != (u > np.float32(t))).sum()) for j, t in enumerate(thr)) / u.numel()
    check(worst, 1e-2, "bf16 u-threshold masks: largest fraction of voxels that differ (random-init weights)")


@pytest.mark.parametrize("model_name,streams", [("U", 1), ("U", 3), ("AU", 3), ("V", 3)])
def test_graph_replayed_patches_are_bit_identical_and_follow_weight_updates(model_name, streams):
    """sliding_window_infer(graph_streams=n): the reference's batch-1 patches replayed from HIP graphs on n
    streams give the same maps bit for bit as the eager one-stream run, also after the parameters were updated
    in place (the volume's weight packing happens outside the graphs), with one cached graph per stream."""
    from devis_amd import TMSU
    from devis_amd.predict import sliding_window_infer
    from tests.util import brats_args
    torch.manual_seed(0)
    m = TMSU(brats_args(model_name=model_name, compute_dtype=torch.bfloat16)).eval()
    g = torch.Generator().manual_seed(21)
    x = torch.randn((1, 4, 240, 240, 160), generator=g).cuda()
    thr = (0.05, 0.5, 0.95)
    for round_ in range(3):
        with torch.no_grad():
            a = sliding_window_infer(x, m.backbone, thr, want_evidence=True)
            b = sliding_window_infer(x, m.backbone, thr, want_evidence=True, graph_streams=streams)
        assert sorted(a) == sorted(b)
        for k in a:
            assert torch.equal(a[k], b[k]), (round_, k)
        with torch.no_grad():
            for p in m.backbone.parameters():
                p.mul_(1.01)
    assert len(m.backbone._devis_patch_graphs) == streams and m.backbone._pack_frozen is False
    assert all(h.stream is not None for h in m.backbone._devis_patch_graphs.values())
    m.train()
    with pytest.raises(RuntimeError):
        sliding_window_infer(x, m.backbone, thr, graph_streams=streams)
