"""`python -m devis_amd.train` = the reference's `__main__` (trainUMIS.py:1357-1449) on synthetic tensors: two epochs
of two steps on 32^3 crops, validation on a full 240x240x160 volume through the sliding window, checkpoint by the
reference's trigger rule and file name, test() reloading it."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_and_test_mode_end_to_end(tmp_path):
    from devis_amd import train as T
    argv = ['--dataset', 'BraTS', '--mode', 'train&test', '--synthetic', '2', '--crop_H', '32', '--crop_W', '32',
            '--crop_D', '32', '--start_epoch', '4', '--end_epochs', '5', '--test_epoch', '5', '--save_dir', str(tmp_path)]
    lines = []
    args = T.getArgs(argv)
    out = T.run(args, log=lines.append)
    assert len(out['loss_list']) == 2 and all(v == v and abs(v) < 1e3 for v in out['loss_list'])
    # per-iteration loss lines in step order, as trainUMIS.py:257 prints them
    steps = [l.split(',')[0] for l in lines if 'train_loss' in l]
    assert steps == ['1', '2', '1', '2']
    # (5 + 1) % (end_epochs - 2) == 0: the reference's trigger saves epoch 5 under its file name (:314-330)
    path = os.path.join(str(tmp_path), 'U_BraTS_folder0_epoch_5.pth')
    assert os.path.exists(path)
    ck = torch.load(path, map_location='cpu', weights_only=True)
    assert ck['epoch'] == 5
    keys = list(ck['state_dict'].keys())
    assert keys[0] == 'backbone.InitConv.conv.weight' and 'backbone.endconv.bias' in keys
    assert 'backbone.DeBlock2.bn2.num_batches_tracked' in keys and 'backbone.EnBlock4_4.bn1.weight' in keys
    assert int(ck['state_dict']['backbone.DeBlock2.bn2.num_batches_tracked']) == 4       # 2 epochs x 2 steps
    # a fresh model of the reference's layout takes the file, and equals the trained one
    fresh = T.build(T.getArgs(argv))[0]
    fresh.load_state_dict(ck['state_dict'])
    for (k, a), (_, b) in zip(fresh.state_dict().items(), out['model'].state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    assert any('Successfully load checkpoint' in l for l in lines)
    t = out['test']
    assert t is not None and len(t['clean_dice']) == 3 and t['regions'] == ('WT', 'TC', 'ET')
    assert 0.0 < t['clean_mean_u'] <= 1.0 and 0.0 < t['noise_mean_u'] <= 1.0


def test_module_entry_runs_as_a_script(tmp_path):
    """`python -m devis_amd.train` in a process of its own (2-D config, one epoch)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'devis_amd.train', '--dataset', 'ISIC', '--mode', 'train', '--synthetic', '2',
                        '--batch_size', '2', '--end_epochs', '4', '--start_epoch', '4', '--save_dir', str(tmp_path)],
                       cwd=root, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'epoch 4 avg_loss' in r.stdout
