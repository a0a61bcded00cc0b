"""Host logic of devis_amd/train.py that needs no GPU: the per-dataset settings of trainUMIS.py:1359-1386 and the
dice val() selects checkpoints on (trainUMIS.py:294-310), against numbers the REFERENCE's own helpers produced
(tests/golden/evalmaps.npz, written by tests/golden/make_goldens.py)."""
import argparse

import numpy as np
import pytest


@pytest.mark.parametrize("dataset,classes,out_size", [
    ("BraTS", 4, (240, 240, 160)), ("ISIC", 2, (224, 300)), ("LiTS", 3, (16, 256, 256)),
    ("DRIVE", 2, (584, 565)), ("TOAR", 7, (512, 512)), ("HC", 10, (128, 1024))])
def test_dataset_overrides_follow_the_reference_table(dataset, classes, out_size):
    from devis_amd.train import getArgs
    a = getArgs(["--dataset", dataset])
    assert a.num_classes == classes and tuple(a.out_size) == out_size
    assert a.batch_size == (2 if dataset == "BraTS" else 8)          # :1360 forces 2 for BraTS only


def test_unknown_dataset_raises_name_error():
    from devis_amd.train import apply_dataset_overrides
    with pytest.raises(NameError):
        apply_dataset_overrides(argparse.Namespace(dataset="CHAOS"))


def test_selection_dice_is_the_references_checkpoint_metric(golden):
    """BraTS -> softmax_output_dice(...)[1], LiTS -> softmax_output_litsdice(...)[0], others -> Dice_isic."""
    from devis_amd.train import selection_dice, dice_isic_from_counts
    G = golden("evalmaps")
    assert selection_dice(G["brats_conf"], "BraTS") == float(G["brats_dice3"][1])
    assert selection_dice(G["lits_conf"], "LiTS") == float(G["lits_dice2"][0])
    for ds in ("ISIC", "DRIVE", "TOAR", "HC"):
        assert np.float32(selection_dice(G["bin_conf"], ds)) == G["bin_dice_isic"]
    # Dice_isic itself on the multi-class volumes (fp32 arithmetic of the reference)
    for tag in ("brats", "lits"):
        assert abs(dice_isic_from_counts(G[tag + "_conf"]) - float(G[tag + "_dice_isic"])) <= 1e-7
    # and it is NOT the mean per-class foreground dice round 1 used
    conf = G["brats_conf"]
    per_class = np.mean([2.0 * conf[k, k] / (conf[:, k].sum() + conf[k, :].sum()) for k in range(1, 4)])
    assert abs(selection_dice(conf, "BraTS") - per_class) > 1e-4


def test_stream_override_is_thread_local():
    """ops.stream_override must not leak into other threads (forward thread vs autograd thread)."""
    import threading
    from devis_amd import ops

    class S:
        cuda_stream = 1234

    seen = {}

    def other():
        seen["other"] = getattr(ops._tls, "override", None)

    with ops.stream_override(S()):
        assert ops._stream() == 1234
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert seen["other"] is None
    assert getattr(ops._tls, "override", None) is None
