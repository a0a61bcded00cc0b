"""Sliding-window stitch of predict.py:159-191 (`tailor_and_concat`).

Eight fixed 128^3 crops of a (1, C, 240, 240, >=155) volume go through the backbone; each
patch fills one of eight DISJOINT regions of the output, so the order (and the GPU a patch
runs on, see devis_amd/parallel.py) does not matter.  Crops and the stitch are plain copies.
"""
from __future__ import annotations

import torch

# (first-dim start, second-dim start, depth start) of the eight crops, predict.py:162-169
CROPS = [(0, 0, 0), (0, 112, 0), (112, 0, 0), (112, 112, 0),
         (0, 0, 27), (0, 112, 27), (112, 0, 27), (112, 112, 27)]


def patch_regions(i: int):
    """-> (dst slices in the 240x240x155 volume, src slices in patch i), predict.py:182-189."""
    a, b, d = CROPS[i]
    dst = (slice(128, 240) if a else slice(0, 128), slice(128, 240) if b else slice(0, 128),
           slice(128, 155) if d else slice(0, 128))
    src = (slice(16, 128) if a else slice(0, 128), slice(16, 128) if b else slice(0, 128),
           slice(96, 123) if d else slice(0, 128))
    return dst, src


def crop(x: torch.Tensor, i: int) -> torch.Tensor:
    a, b, d = CROPS[i]
    return x[..., a:a + 128, b:b + 128, d:d + 128]


def _seed_output(x: torch.Tensor) -> torch.Tensor:
    # predict.py:171-176: the output buffer starts as copies of the input, which fixes its channel count
    c = x.shape[1]
    reps = 4 if c == 1 else (1 if c == 4 else 2)
    return torch.cat([x.clone() for _ in range(reps)], 1) if reps > 1 else x.clone()


def tailor_and_concat(x, model):
    y = _seed_output(x)
    for i in range(len(CROPS)):
        out = model(crop(x, i))
        dst, src = patch_regions(i)
        y[(Ellipsis,) + dst] = out[(Ellipsis,) + src]
    return y[..., :155]


def _default_head(backbone, thresholds, want_evidence, want_mne):
    """patch (1,C,128,128,128) -> dict of per-voxel maps for the patch: backbone + fused inference head on the
    channels-last logits (no NCDHW copy of the logits is ever made)."""
    from . import ops

    def run(patch):
        logits = backbone.forward_channels_last(patch)
        return ops.evid_infer(logits, True, thresholds, want_evidence=want_evidence, want_mne=want_mne)
    return run


def sliding_window_infer(x, backbone, thresholds=(), rank: int = 0, world: int = 1, group=None,
                         want_evidence: bool = False, want_mne: bool = True, patch_fn=None, patch_batch: int = 1):
    """Sharded form of the BraTS val/test forward (predict.py:159-191 + trainUMIS.py:286-290,446-461,521):
    patch i runs on rank i % world, each rank applies the fused inference head to its own
    patches and writes their kept sub-regions into zero-initialised volume maps; the regions are disjoint, so
    one SUM all-reduce per map assembles the volume on every rank (no collective when world == 1).

    Returns dict(pred uint8 (1,240,240,155), u fp32 (1,1,...), mne fp32, umask int16, evidence fp32 (1,K,...))
    with the entries that were asked for.  `patch_fn(patch) -> dict` replaces backbone + head (tests).
    `patch_batch` > 1 sends that many of a rank's patches through the network as ONE batch (an eighth of the
    launches at 8).  In eval mode every sample of a batch is computed independently (GroupNorm is per sample,
    BatchNorm uses its running statistics), so fp32 maps are the same bits as the reference's one-by-one order;
    in bf16 the kernels' tile shapes and statistics chunks depend on the batch size, so values agree to bf16
    rounding only - the default stays 1 (the reference's order).
    """
    from .parallel import shard_patches
    run = patch_fn if patch_fn is not None else _default_head(backbone, thresholds, want_evidence, want_mne)
    vol = tuple(x.shape[2:4]) + (155,)
    out = {}
    mine = list(shard_patches(len(CROPS), rank, world))
    pb = max(1, int(patch_batch))
    for c0 in range(0, len(mine), pb):
        group_ids = mine[c0:c0 + pb]
        patches = [crop(x, i) for i in group_ids]
        maps = run(patches[0] if len(patches) == 1 else torch.cat(patches, dim=0))
        for j, i in enumerate(group_ids):
            dst, src = patch_regions(i)
            for key, t in maps.items():
                if t is None:
                    continue
                if key not in out:
                    lead = (1,) + tuple(t.shape[1:t.dim() - 3])
                    out[key] = torch.zeros(lead + vol, dtype=t.dtype, device=t.device)
                out[key][(Ellipsis,) + dst] = t[j:j + 1][(Ellipsis,) + src]
    if world > 1:
        import torch.distributed as dist
        keys = sorted(out)                       # same order on every rank (every rank owns >= 1 patch iff world <= 8)
        if len(shard_patches(len(CROPS), rank, world)) == 0:
            raise ValueError("sliding_window_infer: more ranks than patches (world must be <= 8)")
        for key in keys:
            t = out[key]
            # disjoint supports: SUM == placement.  Integer maps go through int32 (exact; gloo has no int16 sum)
            if t.dtype in (torch.uint8, torch.int16):
                w = t.to(torch.int32)
                dist.all_reduce(w, op=dist.ReduceOp.SUM, group=group)
                out[key] = w.to(t.dtype)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return out


# ------------------------------------------------------------------ evaluation metrics from device-side counts
# The reference computes these with numpy on host copies of the volumes (predict.py:296-300,386-416;
# numpyfunctions.py:86-107).  Here they are ratios of the integer counts ops.eval_counts() reduces on the
# GPU (csrc/evalmaps.hip): `conf[t, p]` = voxels with target class t and predicted class p.
def _region_counts(conf, classes):
    """(|o|, |t|, |o & t|, |o | t|) for the region {label in classes} of prediction o and target t."""
    idx = list(classes)
    o = conf[:, idx].sum()
    t = conf[idx, :].sum()
    both = conf[idx][:, idx].sum()
    return o, t, both, o + t - both


def dice_score(conf, classes, eps=1e-8):
    """predict.py:386-389 for the region `classes`: (2|o&t| + eps) / (|o| + |t| + eps)."""
    o, t, both, _ = _region_counts(conf, classes)
    return float((2 * both + eps) / (o + t + eps))


def mIOU(conf, classes, eps=1e-8):
    """predict.py:296-299 for the region `classes`."""
    _, _, both, union = _region_counts(conf, classes)
    return float((both + eps) / (union + eps))


def softmax_output_dice(conf):
    """predict.py:401-416 (BraTS): whole = labels > 0, core = {1, 3}, active = {3}."""
    k = conf.shape[0]
    return [dice_score(conf, range(1, k)), dice_score(conf, (1, 3)), dice_score(conf, (3,))]


def softmax_output_litsdice(conf):
    """predict.py:390-399 (LiTS): liver = {1, 2}, tumor = {2}."""
    return [dice_score(conf, (1, 2)), dice_score(conf, (2,))]


def evaluate_volume(out, target, num_classes, n_thr=0):
    """Counts behind val()/test() for one predict_volume() result, reduced on the device, ONE host copy of
    K*K + 8*n_thr + 3 numbers instead of the volumes -> dict(conf, ucount, mean_u, mean_conf, mean_confe)."""
    import numpy as np
    from . import ops
    pred = out["pred"].reshape(-1)
    r = ops.eval_counts(pred, target.reshape(-1).contiguous(), num_classes,
                        out["umask"].reshape(-1) if n_thr else None, n_thr,
                        out["u"].reshape(-1) if out.get("u") is not None else None)
    if int(r["bad"].item()):
        raise ValueError("evaluate_volume: a label or prediction is outside [0, %d)" % num_classes)
    nv = pred.numel()
    res = dict(conf=r["conf"].cpu().numpy(), ucount=r["ucount"].cpu().numpy() if n_thr else None)
    if r["sums"] is not None:
        s = r["sums"].cpu().numpy()
        res.update(mean_u=s[0] / nv, mean_conf=s[1] / nv, mean_confe=s[2] / nv)
    return res
