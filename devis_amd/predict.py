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


class GraphedPatchHead:
    """backbone + fused inference head of ONE patch on ONE stream, captured in a HIP graph and replayed per patch.
    Same kernels, same order: the maps are bit-identical to the eager run.  Captured inside
    `backbone.frozen_weights()`, so the graph holds no weight packing: the caller re-packs once per volume and the
    replays read that packing (parameter updates between volumes are seen).  `stale()` tells when to rebuild
    (mode flip, another patch shape, parameters / buffers / packing re-allocated).  The returned maps are the
    graph's own output buffers: consume them on the same stream before the next call."""

    def __init__(self, backbone, stream, thresholds, want_evidence, want_mne, shape, dtype, dev):
        if backbone.training:
            raise RuntimeError("GraphedPatchHead captures the eval-mode forward: call backbone.eval() first")
        self.backbone, self.stream = backbone, stream
        self._run = _default_head(backbone, thresholds, want_evidence, want_mne)
        self._ident = self._identity()
        cur = torch.cuda.current_stream(dev)
        stream.wait_stream(cur)
        with torch.cuda.stream(stream), torch.no_grad():
            self.static_in = torch.zeros(tuple(shape), dtype=dtype, device=dev)
            for _ in range(2):                                  # warm-up outside the capture
                self._run(self.static_in)
        stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=stream):
            self.static_out = self._run(self.static_in)
        cur.wait_stream(stream)

    def _identity(self):
        b = self.backbone
        return (id(b._packer),) + tuple(t.data_ptr() for t in list(b.parameters()) + list(b.buffers()))

    def stale_for(self, shape, dtype, device) -> bool:
        return (self.backbone.training or tuple(shape) != tuple(self.static_in.shape)
                or dtype != self.static_in.dtype or device != self.static_in.device
                or self._identity() != self._ident)

    def __call__(self, patch):
        """on self.stream (the caller has made it current)"""
        self.static_in.copy_(patch)
        self.graph.replay()
        return self.static_out


_PATCH_STREAMS = {}


def hw_queue_limit() -> int:
    """Hardware queues a process gets on one GPU (HIP's GPU_MAX_HW_QUEUES, 4 unless the environment says otherwise).
    Streams beyond that share an in-order queue with another stream: work of one then stands behind a WAIT of the
    other (seen in the data-parallel step, DESIGN.md section 5: RCCL's stream on the main stream's queue), and a
    rotation whose streams wait on events of each other's queue-mates can stop making progress (the 4-stream run
    of r02 that did not finish).  Every multi-stream path of the package keeps its streams below this number."""
    import os
    try:
        return max(1, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
    except ValueError:
        return 4


def max_patch_streams() -> int:
    """Rotation streams sliding_window_infer accepts: one hardware queue stays with the caller's stream."""
    return max(1, hw_queue_limit() - 1)


def _patch_streams(device, n):
    """n cached streams of `device` for the patch rotation."""
    if n > max_patch_streams():
        raise ValueError("graph_streams = %d: at most %d rotation streams (the caller's stream and the rotation "
                         "together must fit the process's %d hardware queues, GPU_MAX_HW_QUEUES)"
                         % (n, max_patch_streams(), hw_queue_limit()))
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    have = _PATCH_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


def _graphed_volume(x, backbone, thresholds, want_evidence, want_mne, groups, streams, out, vol):
    """The patch loop of sliding_window_infer with each patch (group) replayed from a HIP graph, rotated over
    `streams` streams.  One GraphedPatchHead per stream, cached on the backbone."""
    dev = x.device
    cur = torch.cuda.current_stream(dev)
    ns = max(1, min(int(streams), len(groups)))
    sts = _patch_streams(dev, ns)
    cache = backbone.__dict__.setdefault("_devis_patch_graphs", {})
    with backbone.frozen_weights():
        ready = torch.cuda.Event()
        ready.record(cur)                                   # x and the packed weights exist
        for gi, group_ids in enumerate(groups):
            st = sts[gi % ns]
            shape = (len(group_ids),) + tuple(crop(x, group_ids[0]).shape[1:])
            key = (gi % ns, tuple(thresholds), bool(want_evidence), bool(want_mne), shape, x.dtype)
            head = cache.get(key)
            if head is None or head.stale_for(shape, x.dtype, x.device):
                cache[key] = head = GraphedPatchHead(backbone, st, thresholds, want_evidence, want_mne, shape, x.dtype, dev)
            if not out:                                     # volume maps: allocated and zeroed on the caller's stream
                for k, t in head.static_out.items():
                    if t is not None:
                        lead = (1,) + tuple(t.shape[1:t.dim() - 3])
                        out[k] = torch.zeros(lead + vol, dtype=t.dtype, device=dev)
                ready = torch.cuda.Event()
                ready.record(cur)
            st.wait_event(ready)
            with torch.cuda.stream(st):
                # the crops are made on the stream that consumes them (x is complete: `ready`)
                patches = [crop(x, i) for i in group_ids]
                maps = head(patches[0] if len(patches) == 1 else torch.cat(patches, dim=0))
                for j, i in enumerate(group_ids):
                    dst, src = patch_regions(i)
                    for k, t in maps.items():
                        if t is not None:
                            out[k][(Ellipsis,) + dst] = t[j:j + 1][(Ellipsis,) + src]
        for st in sts:
            cur.wait_stream(st)
    return out


def sliding_window_infer(x, backbone, thresholds=(), rank: int = 0, world: int = 1, group=None,
                         want_evidence: bool = False, want_mne: bool = True, patch_fn=None, patch_batch: int = 1,
                         graph_streams: int = 0):
    """Sharded form of the BraTS val/test forward (predict.py:159-191 + trainUMIS.py:286-290,446-461,521):
    patch i runs on rank i % world, each rank applies the fused inference head to its own
    patches and writes their kept sub-regions into the volume maps; the regions are disjoint, so ONE all_gather
    of every rank's packed sub-regions (exchange_patch_regions) assembles the volume on every rank (no collective
    when world == 1).

    Returns dict(pred uint8 (1,240,240,155), u fp32 (1,1,...), mne fp32, umask int16, evidence fp32 (1,K,...))
    with the entries that were asked for.  `patch_fn(patch) -> dict` replaces backbone + head (tests).
    `patch_batch` > 1 sends that many of a rank's patches through the network as ONE batch (an eighth of the
    launches at 8).  In eval mode every sample of a batch is computed independently (GroupNorm is per sample,
    BatchNorm uses its running statistics), so fp32 maps are the same bits as the reference's one-by-one order;
    in bf16 the kernels' tile shapes and statistics chunks depend on the batch size, so values agree to bf16
    rounding only - the default stays 1 (the reference's order).
    `graph_streams` = n > 0 keeps the batch-1 launches (bit-identical maps in every dtype) but replays each
    patch from a cached HIP graph (GraphedPatchHead) and rotates the patches over n streams: the host no longer
    paces the launches and neighbouring patches fill the CUs the 32^3 / 16^3 layers leave idle.  n is refused
    (ValueError) above max_patch_streams() = hardware queues of the process - 1 (3 here; hw_queue_limit()).
    """
    from .parallel import shard_patches
    if int(graph_streams) > max_patch_streams():
        raise ValueError("graph_streams = %d: at most %d rotation streams (with the caller's stream they must fit the "
                         "process's %d hardware queues, GPU_MAX_HW_QUEUES)"
                         % (graph_streams, max_patch_streams(), hw_queue_limit()))
    run = patch_fn if patch_fn is not None else _default_head(backbone, thresholds, want_evidence, want_mne)
    vol = tuple(x.shape[2:4]) + (155,)
    out = {}
    mine = list(shard_patches(len(CROPS), rank, world))
    mine_all = list(mine)
    pb = max(1, int(patch_batch))
    if graph_streams and patch_fn is None and x.is_cuda and hasattr(backbone, "frozen_weights"):
        _graphed_volume(x, backbone, thresholds, want_evidence, want_mne,
                        [mine[c0:c0 + pb] for c0 in range(0, len(mine), pb)], graph_streams, out, vol)
        mine = []
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
        if len(mine_all) == 0:
            raise ValueError("sliding_window_infer: more ranks than patches (world must be <= 8)")
        exchange_patch_regions(out, rank, world, group)
    return out


def region_bytes(i: int, keys, out) -> int:
    """Bytes patch i's kept sub-region takes in the packed exchange buffer (all maps in `keys`)."""
    dst, _ = patch_regions(i)
    vox = 1
    for s in dst:
        vox *= s.stop - s.start
    return sum(vox * int(out[k].shape[1] if out[k].dim() == 5 else 1) * out[k].element_size() for k in keys)


def exchange_patch_regions(out, rank: int, world: int, group=None) -> None:
    """every rank holds the kept sub-regions of ITS patches in the volume maps `out`; after this call
    every rank holds all of them.  One collective: each rank packs its regions - pred uint8, threshold masks int16,
    u / entropy (and evidence, when asked for) fp32, in their own dtypes, widest first so every view stays aligned -
    into one byte buffer, the buffers are all-gathered (padded to the largest rank's size: the eight regions differ),
    and the other ranks' regions are copied into place.  On the wire: 11 B per voxel of the 240x240x155 volume
    (27 B with the K = 4 evidence), once - the four full-volume SUM all-reduces this replaces moved 16 B (32 B) per
    voxel PER RANK and widened the integer maps to int32."""
    import torch.distributed as dist
    from .parallel import shard_patches
    keys = sorted(out, key=lambda k: (-out[k].element_size(), k))       # same order on every rank
    owned = [list(shard_patches(len(CROPS), r, world)) for r in range(world)]
    sizes = [sum(region_bytes(i, keys, out) for i in owned[r]) for r in range(world)]
    dev = out[keys[0]].device
    buf = torch.zeros((max(sizes),), dtype=torch.uint8, device=dev)
    off = 0
    for i in owned[rank]:
        dst, _ = patch_regions(i)
        for k in keys:
            piece = out[k][(Ellipsis,) + dst].contiguous().view(-1).view(torch.uint8)
            buf[off:off + piece.numel()] = piece
            off += piece.numel()
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    for r in range(world):
        if r == rank:
            continue
        off = 0
        for i in owned[r]:
            dst, _ = patch_regions(i)
            shape3 = tuple(s.stop - s.start for s in dst)
            for k in keys:
                t = out[k]
                lead = tuple(t.shape[:t.dim() - 3])
                n = t.element_size()
                for d in lead + shape3:
                    n *= d
                t[(Ellipsis,) + dst] = gathered[r][off:off + n].view(t.dtype).view(lead + shape3)
                off += n


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
