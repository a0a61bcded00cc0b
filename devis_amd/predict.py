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
