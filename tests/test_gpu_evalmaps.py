"""devis_eval_counts (csrc/evalmaps.hip) against the goldens the reference's metric helpers produced
(tests/golden/evalmaps.npz) and against the numpy oracle at a BraTS-sized volume.  Integers bit-exact;
the three fp64 sums to 1e-6 relative (fp32 expf on the GPU vs numpy)."""
import numpy as np
import pytest
import torch

from tests.test_cpu_oracle import _evalmaps_case

pytestmark = pytest.mark.gpu


def _umask(u, thr):
    m = torch.zeros(u.shape, dtype=torch.int32)
    for j, t in enumerate(thr):
        m |= (torch.from_numpy(u) > np.float32(t)).int() << j
    return m.to(torch.int16)


@pytest.mark.parametrize("tag", ["brats", "lits", "bin"])
@pytest.mark.parametrize("tdtype", [torch.int64, torch.uint8])
def test_eval_counts_match_reference_goldens(golden, tag, tdtype):
    from devis_amd import ops, predict as P
    from oracle import devis_oracle as O
    g, K, pred, target, u = _evalmaps_case(golden, tag)
    thr = O.UNCERTAINTY_THRESHOLDS
    r = ops.eval_counts(pred.to(torch.uint8).cuda(), target.to(tdtype).cuda(), K, _umask(u, thr).cuda(), len(thr),
                        torch.from_numpy(u).cuda())
    assert int(r["bad"].item()) == 0
    conf = r["conf"].cpu().numpy()
    assert np.array_equal(conf, g[tag + "_conf"])
    assert np.array_equal(r["ucount"].cpu().numpy(), g[tag + "_ucount"])
    assert np.allclose(r["sums"].cpu().numpy(), g[tag + "_sums"], rtol=1e-6)
    assert abs(P.dice_score(conf, range(1, K)) - float(g[tag + "_fg_dice"])) < 1e-12


def test_eval_counts_full_volume_and_head_output():
    """240x240x155 (the BraTS test volume): the inference head's own pred / u / umask feed the counts."""
    from devis_amd import ops
    from oracle import devis_oracle as O
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn((1, 4, 240, 240, 155), generator=g) * 3).cuda()
    target = torch.randint(0, 4, (240, 240, 155), generator=g)
    thr = O.UNCERTAINTY_THRESHOLDS
    out = ops.evid_infer(logits, False, thr, want_evidence=False)
    r = ops.eval_counts(out["pred"].reshape(-1), target.cuda().reshape(-1), 4, out["umask"].reshape(-1), len(thr),
                        out["u"].reshape(-1))
    conf, ucount, sums = O.eval_counts(out["pred"].cpu().numpy(), target.numpy(), out["u"].cpu().numpy(), thr, 4)
    assert np.array_equal(r["conf"].cpu().numpy(), conf)
    assert np.array_equal(r["ucount"].cpu().numpy(), ucount)
    assert np.allclose(r["sums"].cpu().numpy(), sums, rtol=1e-6)
    assert int(r["conf"].sum()) == 240 * 240 * 155


def test_eval_counts_flags_bad_labels():
    from devis_amd import ops
    pred = torch.zeros(1000, dtype=torch.uint8).cuda()
    target = torch.zeros(1000, dtype=torch.int64)
    target[17] = 9
    r = ops.eval_counts(pred, target.cuda(), 4)
    assert int(r["bad"].item()) == 1
