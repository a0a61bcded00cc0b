"""Train / val drivers with the call surface of the reference's trainUMIS.py (train :239, val :265,
adjust_learning_rate :217, AverageMeter :221, getArgs :35) and its entry point (`__main__` :1357-1449):

    python -m devis_amd.train --dataset BraTS --mode train --synthetic 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m devis_amd.train ...

Data loaders are the caller's (`run(args, loaders)`): anything yielding (input NC(D)HW fp32, target) batches
works; `--synthetic` supplies seeded tensors of the dataset's shapes.  The tensor work runs on the HIP kernels.
PNG / NIfTI dumps and the medpy metrics of the reference's test() are out of scope.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from . import ops


def getArgs(argv=None):
    """Same flag names and defaults as trainUMIS.py:35-91, plus --compute_dtype / --world_size."""
    local_time = time.strftime("%Y-%m-%d %H:%M:%S", time.localtime())
    p = argparse.ArgumentParser()
    p.add_argument('--user', default='name of user', type=str)
    p.add_argument('--experiment', default='UMIS', type=str)
    p.add_argument('--date', default=local_time.split(' ')[0], type=str)
    p.add_argument('--description', default='Trustworthy medical image segmentation by coco,training on train.txt!', type=str)
    p.add_argument('--start_epoch', type=int, default=1)
    p.add_argument('--end_epochs', type=int, default=99)
    p.add_argument('--submission', default='./results', type=str)
    p.add_argument('--lambda-epochs', type=int, default=50)
    p.add_argument('--save_freq', default=5, type=int)
    p.add_argument('--lr', type=float, default=0.002)
    p.add_argument('--savepath', default='./results/plot/output', type=str)
    p.add_argument('--save_dir', default='./results', type=str)
    p.add_argument('--mode', default='train&test', type=str)
    p.add_argument('--dataset', default='LiTS', type=str)
    p.add_argument('--folder', default='folder0', type=str)
    p.add_argument('--input_H', default=240, type=int)
    p.add_argument('--input_W', default=240, type=int)
    p.add_argument('--input_D', default=160, type=int)
    p.add_argument('--crop_H', default=128, type=int)
    p.add_argument('--crop_W', default=128, type=int)
    p.add_argument('--crop_D', default=128, type=int)
    p.add_argument('--output_D', default=155, type=int)
    p.add_argument('--batch_size', default=8, type=int)
    p.add_argument('--OOD_Condition', default='normal', type=str)
    p.add_argument('--OOD_Level', default=1, type=int)
    p.add_argument('--snapshot', default=True, type=bool)
    p.add_argument('--Uncertainty_Loss', default=False, type=bool)
    p.add_argument('--input_modality', default='four', type=str)
    p.add_argument('--model_name', default='U', type=str)
    p.add_argument('--test_epoch', type=int, default=199)
    # not in the reference: BraTS val/test patches replayed from HIP graphs on this many streams (0 = eager launches)
    p.add_argument('--graph_streams', type=int, default=0)
    p.add_argument('--compute_dtype', default='bf16', choices=['bf16', 'fp32'])
    # not in the reference: spread the eight BraTS patches of a val/test volume over the ranks (same volume on every rank)
    p.add_argument('--shard_patches', action='store_true')
    # not in the reference (its getDataset reads files from fixed E:/ paths; data loading is out of scope here):
    # seeded synthetic tensors of the dataset's shapes, this many training batches per epoch / volumes per val, test
    p.add_argument('--synthetic', type=int, default=0)
    p.add_argument('--synthetic_val', type=int, default=1)
    p.add_argument('--seed', type=int, default=1234, help='data seed; rank r draws from seed + r')
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='torch.distributed backend under torchrun')
    args = p.parse_args(argv)
    args.compute_dtype = torch.bfloat16 if args.compute_dtype == 'bf16' else torch.float32
    apply_dataset_overrides(args)
    return args


# trainUMIS.py:1359-1386: (num_classes, out_size) the reference sets after parsing; BraTS also forces batch_size 2
_DATASETS = {
    'BraTS': (4, (240, 240, 160)),
    'ISIC': (2, (224, 300)),
    'LiTS': (3, (16, 256, 256)),
    'DRIVE': (2, (584, 565)),
    'TOAR': (7, (512, 512)),
    'HC': (10, (128, 1024)),
}


def apply_dataset_overrides(args):
    """Per-dataset settings the reference applies after parsing (trainUMIS.py:1359-1386); an unknown dataset
    raises NameError as there (:1384-1386)."""
    if args.dataset not in _DATASETS:
        print('There is no this dataset')
        raise NameError(args.dataset)
    args.num_classes, args.out_size = _DATASETS[args.dataset]
    if args.dataset == 'BraTS':
        args.batch_size = 2
    return args


def adjust_learning_rate(optimizer, epoch, max_epoch, init_lr, power=0.9):
    # trainUMIS.py:217-219: recomputed from the EPOCH every iteration, rounded to 8 decimals
    lr = round(init_lr * np.power(1 - (epoch) / max_epoch, power), 8)
    for group in optimizer.param_groups:
        group['lr'] = lr


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


_gc_frozen = False


def freeze_gc() -> None:
    """Move everything allocated so far (torch, the model) out of the cyclic GC's reach.  A full Python
    collection walks millions of long-lived objects and stalls the launch thread for ~100 ms - ten training
    steps on MI355X; after this, collections only look at what the steps themselves allocate."""
    global _gc_frozen
    if not _gc_frozen:
        import gc
        gc.collect()
        gc.freeze()
        _gc_frozen = True


class LossMeter:
    """Per-iteration loss lines and the epoch average without a host sync per iteration (the
    reference calls loss.item() twice per step, trainUMIS.py:257-262).  Each step's loss is copied to its own
    pinned host slot behind an event; lines are emitted in step order as soon as their event has completed
    (normally one or two steps late), the rest at flush().  The running sum stays on the device in fp64."""

    def __init__(self, log=print, slots: int = 8):
        self.log, self.slots = log, slots
        self.pending = []            # (step, host slot, event)
        self.free = []
        self.total = None
        self.count = 0

    def update(self, step: int, loss: torch.Tensor) -> None:
        loss = loss.detach()
        self.total = loss.double() if self.total is None else self.total + loss.double()
        self.count += 1
        if self.log is not None:
            if len(self.pending) >= self.slots:
                self._emit(block=True)
            buf, ev = self.free.pop() if self.free else (torch.empty((), dtype=torch.float32).pin_memory(),
                                                         torch.cuda.Event())
            buf.copy_(loss.float(), non_blocking=True)
            ev.record()
            self.pending.append((step, buf, ev))
            self._emit(block=False)

    def _emit(self, block: bool) -> None:
        while self.pending and (block or self.pending[0][2].query()):
            step, buf, ev = self.pending.pop(0)
            ev.synchronize()
            self.log("%d,train_loss:%0.3f" % (step, float(buf)))
            self.free.append((buf, ev))
            block = False

    def flush(self) -> float:
        """Emit what is left; -> the average loss (one sync per epoch)."""
        while self.pending:
            self._emit(block=True)
        return float(self.total.item() / self.count) if self.count else 0.0


def train(args, model, optimizer, epoch, train_loader, reducer=None, log=print):
    """One epoch; same order of operations as trainUMIS.py:239-263.  `reducer` (devis_amd.parallel)
    averages gradients across ranks.  The per-iteration loss lines are kept, in order, but are printed
    through LossMeter: the launch thread never waits for the GPU inside the loop."""
    model.train()
    meter = LossMeter(log)
    step = 0
    freeze_gc()
    for data in train_loader:
        adjust_learning_rate(optimizer, epoch, args.end_epochs, args.lr)
        step += 1
        x, target = data[0], data[-1]
        x = x.cuda(non_blocking=True)
        target = target.cuda(non_blocking=True)
        args.mode = 'train'
        evidences, loss = model(x, target, epoch, args.mode, args.dataset)
        optimizer.zero_grad()
        loss.requires_grad_(True).backward()
        if reducer is not None:
            reducer.all_reduce()
        optimizer.step()
        meter.update(step, loss)
    return meter.flush()


def predict_volume(model, x, dataset, thresholds=(), graph_streams=0, shard=False, group=None):
    """The tensor part of val()/test() (trainUMIS.py:283-290, 446-461): evidence -> alpha, S, prob argmax,
    u = K/S, normalised entropy, u-threshold masks - one fused kernel after the backbone.

    `shard=True` (BraTS only) spreads the eight patches of THIS volume over the ranks of `group`
    and gathers the maps on every rank.  It is opt-in because it is only right when every rank passes the SAME
    volume: with a DistributedSampler on the loader each rank holds another subject, and stitching their patches
    would give a chimera without any error.  A cheap checksum of x is compared across ranks to catch that."""
    model.eval()
    with torch.no_grad():
        if dataset == 'BraTS':
            import torch.distributed as dist
            from .predict import sliding_window_infer
            on = shard and dist.is_available() and dist.is_initialized()
            rank, world = (dist.get_rank(group), dist.get_world_size(group)) if on else (0, 1)
            if on and world > 1:
                chk = x.double().sum().reshape(1)
                lo, hi = chk.clone(), chk.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
                if lo.item() != hi.item():
                    raise ValueError("predict_volume(shard=True): the ranks hold different volumes")
            return sliding_window_infer(x, model.backbone, thresholds, rank, world, group=group, want_evidence=True,
                                        want_mne=True, graph_streams=graph_streams)
        logits = model.backbone(x)
        return ops.evid_infer(logits.contiguous(), False, thresholds, want_evidence=True, want_mne=True)


def dice_per_class(pred, target, num_classes):
    """Per-class foreground dice from ONE device-side confusion-count pass (csrc/evalmaps.hip); the reference
    computes it with numpy after a D2H copy of both volumes."""
    conf = ops.eval_counts(pred.reshape(-1).contiguous(), target.reshape(-1).contiguous(), num_classes)["conf"].cpu()
    out = []
    for k in range(1, num_classes):
        inter, p, t = conf[k, k].item(), conf[:, k].sum().item(), conf[k, :].sum().item()
        out.append(2.0 * inter / max(p + t, 1))
    return out


def dice_isic_from_counts(conf):
    """predict.py:426-437 (`Dice_isic` on one-hot prediction and target) from the K x K confusion counts
    conf[target, pred]: mean over classes 1..K-1 of 2*|p&t| / (|t| + |p| + 1), in fp32 like the reference."""
    conf = np.asarray(conf)
    inter = np.diag(conf).astype(np.float32)
    ref_vol = conf.sum(1).astype(np.float32)
    seg_vol = conf.sum(0).astype(np.float32)
    dice = np.float32(2.0) * inter / (ref_vol + seg_vol + np.float32(1.0))
    return float(np.mean(dice[1:], dtype=np.float32))


def selection_dice(conf, dataset):
    """The dice val() accumulates into `aver_dice`, i.e. what decides `best_dice` and the checkpoint trigger
    (trainUMIS.py:294-310): BraTS -> softmax_output_dice(...)[1] (tumour core {1,3}), LiTS ->
    softmax_output_litsdice(...)[0] (liver {1,2}), every other dataset -> Dice_isic."""
    from .predict import softmax_output_dice, softmax_output_litsdice
    if dataset == 'BraTS':
        return softmax_output_dice(conf)[1]
    if dataset == 'LiTS':
        return softmax_output_litsdice(conf)[0]
    return dice_isic_from_counts(conf)


def _class_indices(target):
    """The 2-D loaders hand over float {0..K-1} masks (ts.ToTensor(), data/transform.py:234); the reference compares them
    with the argmax as they are, the count kernels take integer class indices."""
    return target if target.dtype in (torch.int64, torch.int32, torch.uint8) else target.to(torch.int64)


def val(args, model, current_epoch, best_dice, valid_loader, save=True):
    """trainUMIS.py:265-331: argmax(alpha/S), the dataset's selection dice (`selection_dice`), checkpoint
    {'epoch','state_dict'} with the reference's file name and trigger rule (:314-330).  In a process group only
    rank 0 writes the file."""
    model.eval()
    meter = AverageMeter()
    dice_total = 0.0
    for data in valid_loader:
        x, target = data[0].cuda(), data[-1].cuda()
        if args.dataset == 'BraTS':
            target = target[..., :155]
        out = predict_volume(model, x, args.dataset, graph_streams=getattr(args, 'graph_streams', 0),
                             shard=getattr(args, 'shard_patches', False))
        target = _class_indices(target)
        conf = ops.eval_counts(out["pred"].reshape(-1), target.reshape(-1).contiguous(), args.num_classes)["conf"]
        dice_total += selection_dice(conf.cpu().numpy(), args.dataset)
    aver_dice = dice_total / max(1, len(valid_loader))
    if aver_dice > best_dice \
            or (current_epoch + 1) % int(args.end_epochs - 1) == 0 \
            or (current_epoch + 1) % int(args.end_epochs - 2) == 0 \
            or (current_epoch + 1) % int(args.end_epochs - 3) == 0:
        best_dice = aver_dice
        import torch.distributed as dist
        rank0 = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        if save and rank0:
            tag = '_DUloss' if args.Uncertainty_Loss else ''
            name = '%s_%s_%s%s_epoch_%d.pth' % (args.model_name, args.dataset, args.folder, tag, current_epoch)
            os.makedirs(args.save_dir, exist_ok=True)
            torch.save({'epoch': current_epoch, 'state_dict': model.state_dict()}, os.path.join(args.save_dir, name))
    return meter.avg, best_dice


_REGIONS = {'BraTS': ('WT', 'TC', 'ET'), 'LiTS': ('liver', 'tumor')}
UEO_THRESHOLDS = (0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)      # trainUMIS.py:521


def _region_scores(conf, dataset, num_classes):
    """dice and IoU lists as predict.py:386-416 / 296-300 form them, from the K x K confusion counts."""
    from .predict import dice_score, mIOU
    if dataset == 'BraTS':
        regions = [range(1, num_classes), (1, 3), (3,)]
    elif dataset == 'LiTS':
        regions = [(1, 2), (2,)]
    else:
        regions = [(k,) for k in range(1, num_classes)]
    return [dice_score(conf, r) for r in regions], [mIOU(conf, r) for r in regions]


def test(args, model, test_loader, log=print):
    """The tensor part of trainUMIS.py:333-1000 (mode 'test'): load `<model>_<dataset>_<folder>[_DUloss]_epoch_<n>.pth`
    (:393-409), then for every (input, noised_input, target): evidence of both inputs, argmax(alpha/S), u = K/S,
    normalised entropy, 1-u / exp(-u) means, region dice / IoU and the per-threshold uncertainty confusion counts -
    all reduced on the GPU (predict_volume + evaluate_volume), one small host copy per volume.  Returns the averages
    the reference logs at the end.  Not computed: Hausdorff / ASSD (medpy), ECE (`ece_binary`) and the UEO score
    (`UncertaintyAndCorrectionEvalNumpy`, predict.py:20,53: its module is not in the reference) - the threshold
    counts they are formed from are returned instead (`ucount`)."""
    import time
    from .predict import evaluate_volume
    tag = '_DUloss' if args.Uncertainty_Loss else ''
    name = '%s_%s_%s%s_epoch_%s.pth' % (args.model_name, args.dataset, args.folder, tag, args.test_epoch)
    load_file = os.path.join(args.save_dir, name)
    if os.path.exists(load_file):
        checkpoint = torch.load(load_file, map_location='cpu', weights_only=True)
        model.load_state_dict(checkpoint['state_dict'])
        args.start_epoch = checkpoint['epoch']
        log('Successfully load checkpoint %s' % load_file)
    else:
        log('There is no resume file to load!')
    k = {'BraTS': 4, 'LiTS': 3, 'TOAR': 7, 'HC': 9}.get(args.dataset, 2)         # trainUMIS.py:419-428
    model.eval()
    args.mode = 'test'
    keys = ('dice', 'iou', 'mean_u', 'mean_conf', 'mean_confe', 'mne')
    tot = {c: {q: 0.0 for q in keys} for c in ('clean', 'noise')}
    ucount = {'clean': 0, 'noise': 0}
    runtimes, n = [], 0
    for data in test_loader:
        x, noised_x, target = data[0].cuda(), data[1].cuda(), data[-1].cuda()
        if args.dataset == 'BraTS':
            target = target[..., :155]
        torch.cuda.synchronize()
        t0 = time.time()
        gs = getattr(args, 'graph_streams', 0)
        sh = getattr(args, 'shard_patches', False)
        outs = {'clean': predict_volume(model, x, args.dataset, UEO_THRESHOLDS, gs, shard=sh),
                'noise': predict_volume(model, noised_x, args.dataset, UEO_THRESHOLDS, gs, shard=sh)}
        torch.cuda.synchronize()
        runtimes.append(time.time() - t0)
        n += 1
        target = _class_indices(target)
        for c, out in outs.items():
            r = evaluate_volume(out, target, k, len(UEO_THRESHOLDS))
            dice, iou = _region_scores(r['conf'], args.dataset, k)
            t = tot[c]
            t['dice'] = t['dice'] + np.asarray(dice)
            t['iou'] = t['iou'] + np.asarray(iou)
            for q in ('mean_u', 'mean_conf', 'mean_confe'):
                t[q] += float(r[q])
            t['mne'] += float(out['mne'].double().mean().item())
            ucount[c] = ucount[c] + r['ucount']
            log('%s subject %d: dice %s  U %.6f  conf %.6f  confe %.6f' %
                (c, n, np.round(dice, 4).tolist(), r['mean_u'], r['mean_conf'], r['mean_confe']))
    res = {}
    for c in tot:
        for q in keys:
            res[c + '_' + q] = tot[c][q] / max(n, 1)
        res[c + '_ucount'] = ucount[c]
    res['regions'] = _REGIONS.get(args.dataset)
    res['runtime'] = float(np.mean(runtimes)) if runtimes else 0.0
    return res


# ------------------------------------------------------------------ entry point (trainUMIS.py:1357-1449)
_IN_CHANNELS = {'BraTS': 4, 'LiTS': 1, 'ISIC': 3, 'DRIVE': 3, 'TOAR': 1, 'HC': 1}       # what TMSU's backbones take
OOD_CONDITIONS = ('noise', 'blur', 'mask')                                              # trainUMIS.py:1404


def _in_channels(args):
    if args.model_name == 'V' and args.dataset not in ('BraTS', 'LiTS'):
        return 3                                   # VNet2D(n_channels=3) on every 2-D dataset, TrustworthySeg.py:54,78,91,104
    return _IN_CHANNELS[args.dataset]


class SyntheticLoader:
    """`n` seeded batches shaped like the dataset's: x fp32 N(C)(D)HW, target int64 (3-D: N,D,H,W
    - BraTS crops, data/BraTS2019.py:46-57,135-146) or float {0..K-1} (2-D: N,1,H,W as ts.ToTensor() leaves it,
    data/transform.py:234); with `noised`, (x, x + 0.1 * noise, target) like the reference's test loaders.  The batches
    are drawn once on the CPU generator and live on the host, as a DataLoader's would."""

    def __init__(self, n, batch, channels, spatial, classes, seed, noised=False):
        g = torch.Generator().manual_seed(seed)
        self.items = []
        for _ in range(n):
            x = torch.randn((batch, channels) + tuple(spatial), generator=g)
            if len(spatial) == 3:
                y = torch.randint(0, classes, (batch,) + tuple(spatial), generator=g)
            else:
                y = torch.randint(0, classes, (batch, 1) + tuple(spatial), generator=g).float()
            if noised:
                self.items.append((x, x + 0.1 * torch.randn(x.shape, generator=g), y))
            else:
                self.items.append((x, y))

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


def synthetic_loaders(args, rank=0, salt=0):
    """(train, valid, test) loaders of seeded tensors: training batches of `batch_size` crops (BraTS: crop_H x crop_W x
    crop_D, trainUMIS.py:65-67; otherwise the dataset's out_size), validation / test volumes of batch 1 at the full
    size (BraTS: input_H x input_W x input_D = 240 x 240 x 160, which tailor_and_concat's fixed crops need).  Rank r
    draws its TRAINING data from seed + r; validation / test volumes are the same on every rank."""
    c, k = _in_channels(args), args.num_classes
    if args.dataset == 'BraTS':
        crop = (args.crop_H, args.crop_W, args.crop_D)
        full = (args.input_H, args.input_W, args.input_D)
    else:
        crop = full = tuple(args.out_size)
    seed = args.seed + 7919 * salt
    train_loader = SyntheticLoader(args.synthetic, args.batch_size, c, crop, k, seed + rank)
    valid_loader = SyntheticLoader(args.synthetic_val, 1, c, full, k, seed + 100003)
    test_loader = SyntheticLoader(args.synthetic_val, 1, c, full, k, seed + 200003, noised=True)
    return train_loader, valid_loader, test_loader


def init_distributed(args):
    """Under torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment): one process per GPU, the process group
    created BEFORE any GPU work with RCCL's stream taken from the high-priority pool (parallel.nccl_options: it
    must not share the main stream's hardware queue).  -> (rank, world, device)."""
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    if not torch.cuda.is_available():
        raise SystemExit("devis_amd.train needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if "RANK" in os.environ:
        import torch.distributed as dist
        if not dist.is_initialized():
            if args.backend == 'nccl':
                from .parallel import nccl_options
                dist.init_process_group("nccl", device_id=dev, pg_options=nccl_options())
            else:
                dist.init_process_group("gloo")
    return rank, world, dev


def build(args, rank=0):
    """model, optimizer, reducer as trainUMIS.py:1389-1395 builds them (TMSU, Adam(lr, weight_decay=1e-5)) - the
    optimizer as the one-launch FusedAdam, identical replicas on every rank (same seed + a broadcast from rank 0), the
    gradient all-reduce bucketed and overlapped with backward when there is a process group."""
    import torch.distributed as dist
    from . import TMSU
    from .optim import FusedAdam
    from .parallel import GradientAllReducer, broadcast_parameters
    torch.manual_seed(0)
    model = TMSU(args)
    model.cuda()
    total = sum(p.nelement() for p in model.parameters())
    if rank == 0:
        print("Number of model's Params: %.2fM" % (total / 1e6))
    on = dist.is_available() and dist.is_initialized()
    if on:
        broadcast_parameters(model, 0)
    optimizer = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)
    reducer = GradientAllReducer(model.parameters(), module=model) if on and dist.get_world_size() > 1 else None
    return model, optimizer, reducer


def run(args, loaders=None, log=print):
    """The reference's `__main__` after argument parsing (trainUMIS.py:1387-1449): model, Adam, then by `args.mode`
    'train&test' (epochs start_epoch..end_epochs of train + val, then test), 'train' (epochs 1..end_epochs of train +
    val) or 'test' (the OOD condition x level sweep of :1436-1446, three levels of 'noise' and 'blur').  `loaders` =
    (train, valid, test), or a callable(args) returning them (called again per OOD setting like getDataset); None takes
    the synthetic ones.  -> dict(loss_list, dice_list, best_dice, test)."""
    mode = args.mode                                 # train() / test() overwrite args.mode like the reference (:252)
    if mode not in ('train&test', 'train', 'test'):
        print('There is no this mode')
        raise NameError(mode)
    rank, world, _ = init_distributed(args)
    quiet = (lambda *a, **k: None)
    say = log if rank == 0 else quiet
    if loaders is None:
        if not args.synthetic:
            raise SystemExit("devis_amd.train: no data - pass --synthetic N (seeded tensors of the dataset's shapes) or "
                             "call devis_amd.train.run(args, loaders) with your own loaders; the reference's file "
                             "readers (data/*.py) are outside the scope of this package")
        get = lambda a, salt=0: synthetic_loaders(a, rank, salt)
    elif callable(loaders):
        get = lambda a, salt=0: loaders(a)
    else:
        get = lambda a, salt=0: loaders
    train_loader, valid_loader, test_loader = get(args)
    model, optimizer, reducer = build(args, rank)
    out = dict(loss_list=[], dice_list=[], best_dice=0, test=None)
    best_dice = 0
    if mode in ('train&test', 'train'):
        first = args.start_epoch if mode == 'train&test' else 1
        for epoch in range(first, args.end_epochs + 1):
            say('===========Train begining!===========')
            say('Epoch {}/{}'.format(epoch, args.end_epochs - 1))
            epoch_loss = train(args, model, optimizer, epoch, train_loader, reducer=reducer, log=say)
            say("epoch %d avg_loss:%0.3f" % (epoch, epoch_loss))
            _, best_dice = val(args, model, epoch, best_dice, valid_loader)
            out['loss_list'].append(epoch_loss)
            out['dice_list'].append(best_dice)
        if mode == 'train&test':
            out['test'] = test(args, model, test_loader, log=say)
    else:
        out['test'] = {}
        for j in range(0, 2):
            args.OOD_Condition = OOD_CONDITIONS[j]
            say("arg.OOD_Condition: %s" % args.OOD_Condition)
            for i in range(1, 4):
                say("arg.OOD_Level: %d" % i)
                args.OOD_Level = i
                _, _, test_loader = get(args, 3 * j + i)
                out['test'][(args.OOD_Condition, i)] = test(args, model, test_loader, log=say)
    out['best_dice'] = best_dice
    out['model'] = model
    return out


def main(argv=None):
    args = getArgs(argv)
    out = run(args)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
