"""Host logic of devis_amd/train.py that needs no GPU: the per-dataset settings of trainUMIS.py:1359-1386 and the
dice val() selects checkpoints on (trainUMIS.py:294-310), against numbers the REFERENCE's own helpers produced
(tests/golden/evalmaps.npz, written by tests/golden/make_goldens.py)."""
import argparse

import numpy as np
import pytest
import torch


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


def test_entry_point_parses_every_reference_flag():
    """`python -m devis_amd.train` takes the reference's command line (trainUMIS.py:35-91): every flag with a
    non-default value lands under the reference's attribute name, and the defaults are the reference's."""
    from devis_amd.train import getArgs
    a = getArgs(['--user', 'u', '--experiment', 'E', '--date', '2020-01-01', '--description', 'd', '--start_epoch', '3',
                 '--end_epochs', '7', '--submission', 's', '--lambda-epochs', '9', '--save_freq', '2', '--lr', '0.0002',
                 '--savepath', 'sp', '--save_dir', 'sd', '--mode', 'train', '--dataset', 'ISIC', '--folder', 'folder3',
                 '--input_H', '1', '--input_W', '2', '--input_D', '3', '--crop_H', '4', '--crop_W', '5', '--crop_D', '6',
                 '--output_D', '7', '--batch_size', '4', '--OOD_Condition', 'blur', '--OOD_Level', '2', '--snapshot', '',
                 '--Uncertainty_Loss', 'True', '--input_modality', 't1', '--model_name', 'V', '--test_epoch', '11'])
    want = dict(user='u', experiment='E', date='2020-01-01', description='d', start_epoch=3, end_epochs=7, submission='s',
                lambda_epochs=9, save_freq=2, lr=0.0002, savepath='sp', save_dir='sd', mode='train', dataset='ISIC',
                folder='folder3', input_H=1, input_W=2, input_D=3, crop_H=4, crop_W=5, crop_D=6, output_D=7, batch_size=4,
                OOD_Condition='blur', OOD_Level=2, snapshot=False, Uncertainty_Loss=True, input_modality='t1',
                model_name='V', test_epoch=11, num_classes=2, out_size=(224, 300))
    for k, v in want.items():
        assert getattr(a, k) == v, k
    d = getArgs([])
    ref = dict(start_epoch=1, end_epochs=99, lambda_epochs=50, lr=0.002, mode='train&test', dataset='LiTS', folder='folder0',
               input_H=240, input_W=240, input_D=160, crop_H=128, crop_W=128, crop_D=128, output_D=155, batch_size=8,
               OOD_Condition='normal', OOD_Level=1, snapshot=True, Uncertainty_Loss=False, input_modality='four',
               model_name='U', test_epoch=199, save_dir='./results', num_classes=3)
    for k, v in ref.items():
        assert getattr(d, k) == v, k
    assert getArgs(['--dataset', 'BraTS', '--batch_size', '8']).batch_size == 2      # trainUMIS.py:1360


def test_synthetic_loaders_follow_the_dataset_shapes_and_rank_seeds():
    from devis_amd.train import getArgs, synthetic_loaders
    a = getArgs(['--dataset', 'BraTS', '--synthetic', '2', '--crop_H', '16', '--crop_W', '16', '--crop_D', '16',
                 '--input_H', '24', '--input_W', '24', '--input_D', '16'])
    tr0, va0, te0 = synthetic_loaders(a, rank=0)
    tr1, va1, _ = synthetic_loaders(a, rank=1)
    assert len(tr0) == 2 and len(va0) == 1 and len(te0) == 1
    x, y = next(iter(tr0))
    assert x.shape == (2, 4, 16, 16, 16) and x.dtype == torch.float32
    assert y.shape == (2, 16, 16, 16) and y.dtype == torch.int64 and int(y.max()) <= 3
    assert not torch.equal(x, next(iter(tr1))[0])                    # rank r trains on seed + r
    assert torch.equal(next(iter(va0))[0], next(iter(va1))[0])       # every rank validates the same volume
    xt, xn, yt = next(iter(te0))
    assert xt.shape == (1, 4, 24, 24, 16) and xn.shape == xt.shape and yt.shape == (1, 24, 24, 16)
    b = getArgs(['--dataset', 'ISIC', '--synthetic', '1', '--batch_size', '3'])
    x, y = next(iter(synthetic_loaders(b)[0]))
    assert x.shape == (3, 3, 224, 300) and y.shape == (3, 1, 224, 300) and y.dtype == torch.float32
    assert set(y.unique().tolist()) <= {0.0, 1.0}


def test_entry_point_refuses_an_unknown_mode_like_the_reference():
    from devis_amd.train import getArgs, run
    a = getArgs(['--dataset', 'BraTS', '--mode', 'nonsense', '--synthetic', '1'])
    with pytest.raises(NameError):                                   # trainUMIS.py:1447-1449
        run(a)


def test_rotation_streams_are_kept_below_the_hardware_queue_count(monkeypatch):
    """sliding_window_infer refuses more rotation streams than the process has hardware queues to give them
    (GPU_MAX_HW_QUEUES, 4 by default: the caller's stream + 3); DESIGN.md section 5 says why."""
    from devis_amd import predict
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert predict.hw_queue_limit() == 4 and predict.max_patch_streams() == 3
    x = torch.zeros((1, 4, 240, 240, 160))
    with pytest.raises(ValueError, match="hardware queues"):
        predict.sliding_window_infer(x, None, patch_fn=lambda p: {}, graph_streams=4)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert predict.max_patch_streams() == 7
    predict.sliding_window_infer(x, None, patch_fn=lambda p: {}, graph_streams=4)     # patch_fn given: eager loop
