#!/usr/bin/env python3
"""bench.py - BraTS-shaped evidential 3-D U-Net training throughput on MI355X.

One "step" = the reference's hot loop body (trainUMIS.py:244-262) on one synthetic batch:
poly-LR update, TMSU forward (backbone + fused evidential head/loss), backward, gradient
all-reduce over RCCL when WORLD_SIZE > 1, Adam step.  Metric: train voxels/s, whole job.

    python bench.py --gpus 1 --steps 30 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     - the dominant kernel timed live with HIP events on the launch stream,
  cpu_baseline - the CPU oracle (oracle/devis_oracle.py) timed on a bounded sample (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0          # HBM3E spec, same table
FLOP_PER_VOXEL_FWD_BWD = 428160  # conv FLOPs fwd+bwd per voxel of the 3-D U-Net


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2, help="samples per GPU (reference default 2, trainUMIS.py:1360)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    return ap.parse_args()


def brats_args(dtype):
    return argparse.Namespace(num_classes=4, dataset='BraTS', model_name='U', input_modality='four', end_epochs=99,
                              lambda_epochs=50, Uncertainty_Loss=False, compute_dtype=dtype, lr=0.002)


# ---------------------------------------------------------------------------------------------- roofline
# Which kernel the roofline line is about is READ, not hard-coded: the kernel name with the largest summed time in the
# latest tracked `rocprofv3 --kernel-trace --stats` summary of this command (profiles/*step_kernel_stats*two_stream*.csv,
# written by tools/prof_step.sh).  A kernel name is one template instance and may serve several layers of the step, so
# every family below lists the layer shapes it runs at and how often per step; each shape is launched through the C ABI
# into preallocated buffers, 20 launches between two HIP events on the launch stream, and
#   achieved = sum(calls * algorithmic work) / sum(calls * measured time).
def _latest_stats_csv():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*step_kernel_stats*two_stream*.csv")))
    return files[-1] if files else None


def _read_stats(path):
    """[(name, calls, total_ns)] sorted by total time, the TOTAL row dropped."""
    import csv
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Name"] != "TOTAL" and r["TotalDurationNs"]:
                rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])))
    rows.sort(key=lambda t: -t[2])
    return rows


def _conv_work(c, vox, cin=None):
    """(FLOPs, algorithmic bytes) of one pass of a 3x3x3 conv layer over `vox` voxels in bf16: forward and dgrad read one
    tensor and write one, the weight gradient reads two (C*s in + C*s out per voxel)."""
    cin = c if cin is None else cin
    return 2.0 * 27 * cin * c * vox, 2.0 * c * 2 * vox


class _Probes:
    """Live timing of one kernel family at the shapes the step runs it at."""

    def __init__(self, batch, size, dtype):
        import ctypes as C
        from devis_amd import _lib, ops
        from devis_amd.unet3d import G3
        self.C, self.lib, self._lib, self.ops, self.G3 = C, _lib.load(), _lib, ops, G3
        self.batch, self.size, self.dtype = batch, size, dtype
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.gen = torch.Generator(device="cpu").manual_seed(5)
        self.stream = torch.cuda.current_stream().cuda_stream
        self._t = {}

    def _tensors(self, c, s, cin=None):
        key = (c, s, cin)
        if key not in self._t:
            cin_ = c if cin is None else cin
            g, b = self.gen, self.batch
            x = torch.randn((b, s, s, s, cin_), generator=g).to(self.dtype).to(self.dev)
            dy = torch.randn((b, s, s, s, c), generator=g).to(self.dtype).to(self.dev)
            w = (torch.randn((c, cin_, 3, 3, 3), generator=g) * 0.05).to(self.dev)
            self._t[key] = dict(x=x, dy=dy, y=torch.empty_like(dy), wp=self.ops.pack_conv_weight(w, self.dtype),
                                b=torch.zeros(c, device=self.dev), sc=torch.ones((b, cin_), device=self.dev),
                                sh=torch.zeros((b, cin_), device=self.dev),
                                d=self.G3.desc(b, (s, s, s), cin_, 0, c, self.dtype))
        return self._t[key]

    def timeit(self, fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    def conv_fwd(self, c, s, prologue=True, cin=None):
        t, C = self._tensors(c, s, cin), self.C
        sc, sh = (t["sc"].data_ptr(), t["sh"].data_ptr()) if prologue else (None, None)

        def run():
            self._lib.check(self.lib.devis_conv_fwd(C.byref(t["d"]), t["x"].data_ptr(), None, t["wp"].data_ptr(), t["b"].data_ptr(),
                                                    sc, sh, None, t["y"].data_ptr(), self.stream), "conv_fwd")
        return self.timeit(run)

    def wgrad_stage1(self, c, s, prologue=True):
        """The partial-sum kernel of the weight gradient alone (devis_conv_wgrad_stage1: no second-stage reduction)."""
        t, C = self._tensors(c, s), self.C
        nb = self.lib.devis_conv_wgrad_workspace_bytes(C.byref(t["d"]))
        ws = self.ops.workspace(nb, self.dev, "bench_ws")
        sc, sh = (t["sc"].data_ptr(), t["sh"].data_ptr()) if prologue else (None, None)

        def run():
            self._lib.check(self.lib.devis_conv_wgrad_stage1(C.byref(t["d"]), t["x"].data_ptr(), sc, sh, t["dy"].data_ptr(), 1,
                                                             ws.data_ptr(), nb, self.stream), "conv_wgrad_stage1")
        return self.timeit(run)


def _families(batch, size):
    """kernel-name prefix -> (roof that binds it, [(label, calls per step, probe(P) -> seconds, flops, bytes)]).
    Shapes are the 3-D U-Net's levels: C channels at (size / 2^level)^3 voxels per sample (UNet3DZoo.py:243-305)."""
    v = [batch * (size >> l) ** 3 for l in range(4)]
    s = [size >> l for l in range(4)]
    fam = {}
    # 32 / 64 / 128-channel weight gradients: EnBlock2_x + DeBlock3 (6 layers), EnBlock3_x + DeBlock4 (6), EnBlock4_x (8)
    fam["k_wgrad3_slab_march"] = ("mfma", [("32ch @ %d^3" % s[1], 6, lambda P: P.wgrad_stage1(32, s[1]), *_conv_work(32, v[1])),
                                           ("64ch @ %d^3" % s[2], 6, lambda P: P.wgrad_stage1(64, s[2]), *_conv_work(64, v[2])),
                                           ("128ch @ %d^3" % s[3], 8, lambda P: P.wgrad_stage1(128, s[3]), *_conv_work(128, v[3]))])
    fam["k_wgrad3_c16_ksplit"] = ("hbm", [("16ch @ %d^3" % s[0], 4, lambda P: P.wgrad_stage1(16, s[0]), *_conv_work(16, v[0]))])
    fam["k_conv3_c16_march"] = ("hbm", [("16ch @ %d^3" % s[0], 8, lambda P: P.conv_fwd(16, s[0]), *_conv_work(16, v[0]))])
    fam["k_conv3_c32_march"] = ("mfma", [("32ch @ %d^3" % s[1], 12, lambda P: P.conv_fwd(32, s[1]), *_conv_work(32, v[1]))])
    fam["k_conv3_deepk<64"] = ("mfma", [("64ch @ %d^3" % s[2], 12, lambda P: P.conv_fwd(64, s[2]), *_conv_work(64, v[2]))])
    fam["k_conv3_deepk<128"] = ("mfma", [("128ch @ %d^3" % s[3], 16, lambda P: P.conv_fwd(128, s[3]), *_conv_work(128, v[3]))])
    return fam


def _family_of(name, fam):
    for k in fam:
        if k in name:
            return k
    return None


# algorithmic work per CALL of the other kernels that reach the step's top ten (elementwise passes: bytes read + written;
# glue kernels - partial reductions, finalize kernels - do no algorithmic work: their time counts, their work is zero)
def _other_work(name, batch, size):
    v = [batch * (size >> l) ** 3 for l in range(4)]
    if "k_norm_bwd_apply" in name:      # 6 calls at 32 ch, 6 at 64, 8 at 128, 1 at 16: dx = cA*dy + cB*x + cC, 3 tensors
        per = (6 * 32 * v[1] + 6 * 64 * v[2] + 8 * 128 * v[3] + 16 * v[0]) / 21.0
        return 0.0, 3 * 2 * per
    if "k_pw_wgrad" in name or "k_pw_mfma" in name or "k_s2_dgrad" in name or "k_wgrad3_mfma" in name:
        return None                      # gather GEMMs of several shapes: not modelled
    if "k_affine_act" in name:
        return 0.0, 3 * 2 * 16 * v[0]
    return 0.0, 0.0


def roofline(batch, size, dtype, world_rank=0):
    """-> (roofline dict for the JSON line, time-weighted dict over the top ten kernels | None)"""
    P = _Probes(batch, size, dtype)
    fam = _families(batch, size)
    path = _latest_stats_csv()
    rows = _read_stats(path) if path else []
    dominant = next((n for n, _, _ in rows if _family_of(n, fam)), None) if rows else None
    unprobed = rows[0][0] if rows and rows[0][0] != dominant else None
    key = _family_of(dominant, fam) if dominant else "k_conv3_c16_march"
    bound, shapes = fam[key]
    t_sum = f_sum = b_sum = 0.0
    detail = {}
    for label, calls, probe, flops, nbytes in shapes:
        t = probe(P)
        detail[label] = {"calls_per_step": calls, "avg_launch_ms": t * 1e3, "TFLOP/s": flops / t / 1e12, "GB/s_alg": nbytes / t / 1e9}
        t_sum += calls * t
        f_sum += calls * flops
        b_sum += calls * nbytes
    ncalls = sum(c for _, c, _, _, _ in shapes)
    if bound == "mfma":
        achieved, peak, unit = f_sum / t_sum / 1e12, PEAK_BF16_TFLOPS, "TFLOP/s"
    else:
        achieved, peak, unit = b_sum / t_sum / 1e9, PEAK_HBM_GBS, "GB/s"
    traffic, pmc_mfma = pmc_lookup((key,), batch * size ** 3, dtype)
    out = {"kernel": dominant or key, "family": key, "picked_from": os.path.relpath(path, ROOT) if path else None,
           "bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak, "traffic": traffic,
           "avg_launch_ms": t_sum / ncalls * 1e3, "algorithmic_bytes_per_launch": b_sum / ncalls,
           "algorithmic_flops_per_launch": f_sum / ncalls,
           "mfma_achieved_TFLOPs": f_sum / t_sum / 1e12, "mfma_frac": f_sum / t_sum / 1e12 / PEAK_BF16_TFLOPS,
           "hbm_achieved_GBs": b_sum / t_sum / 1e9, "hbm_frac": b_sum / t_sum / 1e9 / PEAK_HBM_GBS,
           "mfma_pipe_busy_frac_pmc": pmc_mfma, "shapes": detail}
    if unprobed:
        out["note"] = "largest name in the profile has no live probe: " + unprobed[:80]
    # the other families, for the record (one line each) + the fused evidential head (north_star's second roofline)
    others = {}
    for k, (bd, shp) in fam.items():
        if k == key:
            continue
        label, calls, probe, flops, nbytes = shp[-1] if k == "k_wgrad3_slab_march" else shp[0]
        t = probe(P)
        others[k + " " + label] = {"ms": t * 1e3, "TFLOP/s": flops / t / 1e12, "GB/s_alg": nbytes / t / 1e9}
    others.update(head_probe(P))
    out["all"] = others
    # time-weighted roofline fraction over the ten largest kernel names of the tracked profile: sum of the time each would
    # take AT its roof (max of FLOPs / MFMA peak and algorithmic bytes / HBM peak) over the sum of their measured times
    weighted = None
    if rows:
        roof_t = meas_t = 0.0
        unmodelled = []
        steps = max(1, next((c for n, c, _ in rows if "k_adam" in n), 1))
        for n, calls, tot in rows[:10]:
            k = _family_of(n, fam)
            if k:
                shp = fam[k][1]
                cs = float(sum(c for _, c, _, _, _ in shp))
                fl = sum(c * f for _, c, _, f, _ in shp) / cs
                by = sum(c * b for _, c, _, _, b in shp) / cs
                w = (fl, by)
            else:
                w = _other_work(n, batch, size)
            meas_t += tot * 1e-9
            if w is None:
                unmodelled.append(n[:60])
                continue
            roof_t += calls * max(w[0] / (PEAK_BF16_TFLOPS * 1e12), w[1] / (PEAK_HBM_GBS * 1e9))
        weighted = {"kernels": 10, "frac": roof_t / meas_t, "time_ms_per_step": meas_t / steps * 1e3,
                    "share_of_kernel_time": meas_t / (sum(t for _, _, t in rows) * 1e-9),
                    "unmodelled_counted_as_zero_work": unmodelled, "source": os.path.relpath(path, ROOT)}
    return out, weighted


def head_probe(P):
    """The fused evidential head + loss, launched through the C ABI into preallocated buffers (algorithmic bytes of
    SURVEY.md 8(d): forward reads K*s + 8 (int64 label) and writes the fp32 evidence the train mode returns; backward
    re-reads K*s + 8 and writes K*s)."""
    import ctypes as C
    lib, _lib, ops = P.lib, P._lib, P.ops
    b, s = P.batch, P.size
    vox = b * s ** 3
    esz = 2 if P.dtype == torch.bfloat16 else 4
    z = torch.randn((b, s, s, s, 4), generator=P.gen).to(P.dtype).to(P.dev)
    dz = torch.empty_like(z)
    lab = torch.randint(0, 4, (b, s, s, s), generator=P.gen).to(P.dev)
    lab8 = lab.to(torch.uint8)
    ev = torch.empty((b, 4, s, s, s), dtype=torch.float32, device=P.dev)
    scal = torch.empty((_lib.EVID_SCAL_FLOATS,), dtype=torch.float32, device=P.dev)
    one = torch.ones((), device=P.dev)
    nb = lib.devis_evid_workspace_bytes(b, s ** 3, 4)
    ws = ops.workspace(nb, P.dev, "bench_ws_head")
    dt = _lib.DEVIS_BF16 if P.dtype == torch.bfloat16 else _lib.DEVIS_F32
    v = s ** 3
    res = {}
    for tag, l, code, lb in (("", lab, _lib.LABEL_I64, 8), ("_u8_labels", lab8, _lib.LABEL_U8, 1)):
        def fwd():
            _lib.check(lib.devis_evid_loss_fwd(z.data_ptr(), v * 4, 4, 1, dt, l.data_ptr(), code, b, v, 4, C.c_float(0.5),
                                               ev.data_ptr(), None, scal.data_ptr(), ws.data_ptr(), nb, P.stream), "evid_fwd")

        def bwd():
            _lib.check(lib.devis_evid_loss_bwd(z.data_ptr(), v * 4, 4, 1, dt, l.data_ptr(), code, b, v, 4, C.c_float(0.5),
                                               scal.data_ptr(), one.data_ptr(), None, None, None, dz.data_ptr(), P.stream), "evid_bwd")
        tf, tb = P.timeit(fwd), P.timeit(bwd)
        bf, bb = vox * (4 * esz + lb + 16), vox * (4 * esz + lb + 4 * esz)
        res["evid_head_loss_fwd" + tag] = {"ms": tf * 1e3, "TFLOP/s": 0.0, "GB/s_alg": bf / tf / 1e9, "hbm_frac": bf / tf / 1e9 / PEAK_HBM_GBS}
        res["evid_head_loss_bwd" + tag] = {"ms": tb * 1e3, "TFLOP/s": 0.0, "GB/s_alg": bb / tb / 1e9, "hbm_frac": bb / tb / 1e9 / PEAK_HBM_GBS}
    return res


def pmc_lookup(name_parts, voxels, dtype):
    """(HBM bytes per launch, MFMA-pipe busy fraction) of the kernel whose name contains every string of `name_parts`,
    from the latest tracked counter passes (profiles/r*_pmc_step.json, tools/pmc_step.sh: taken at B=2 x 128^3; bytes
    scale with the voxels).  (None, None) when there is no such file or kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_step.json")))
    if dtype != torch.bfloat16 or not files:
        return None, None
    try:
        with open(files[-1]) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    best = None
    for k, v in kernels.items():
        if all(p_ in k for p_ in name_parts) and (best is None or v.get("us_per_step", 0) > best.get("us_per_step", 0)):
            best = v
    if best is None:
        return None, None
    return best["hbm_bytes"] * voxels / (2 * 128 ** 3), best["mfma_util"]


def cpu_baseline(steps, size):
    """The CPU oracle's train step (fwd + evidential loss + bwd + Adam) on B=1, timed on the host cores."""
    from oracle import devis_oracle as O
    from devis_amd import Unet
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in Unet(4, 16, 4).state_dict().items()}
    params = []
    for k in sd:
        if sd[k].is_floating_point() and k.split('.')[-1] in ("weight", "bias"):
            sd[k].requires_grad_(True)
            params.append(sd[k])
    opt = torch.optim.Adam(params, lr=0.002, weight_decay=1e-5)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((1, 4, size, size, size), generator=g)
    y = torch.randint(0, 4, (1, size, size, size), generator=g)
    cores = os.cpu_count() if "OMP_NUM_THREADS" not in os.environ else int(os.environ["OMP_NUM_THREADS"])
    torch.set_num_threads(cores)
    times = []
    for i in range(steps + 1):
        t0 = time.time()
        _, loss = O.tmsu_forward(lambda t: O.unet3d_forward(sd, t, True), x, y, 25, 'train', 'BraTS', 4, 50, 100)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if i > 0:
            times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return {"value": size ** 3 / dt, "unit": "voxels/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps of B=1 4x%d^3 K=4 fp32, oracle/devis_oracle.py (pinned bit-exact to the reference), "
                      "%.2f s/step" % (steps, size, dt)}


def fp32_step_ms(batch, size, x, y, steps=3, warmup=2):
    """ms per training step of the same model in the fp32 parity mode (same inputs, fresh replica)."""
    from devis_amd import TMSU
    from devis_amd.optim import FusedAdam
    from devis_amd.train import adjust_learning_rate
    args = brats_args(torch.float32)
    torch.manual_seed(0)
    model = TMSU(args)
    model.train()
    opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)

    def step():
        adjust_learning_rate(opt, 25, args.end_epochs, args.lr)
        _, loss = model(x, y, 25, 'train', 'BraTS')
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if not math.isfinite(float(loss.item())):
        return None
    return dt / steps * 1e3


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # DEVIS_BENCH_FORCE_DIST=1 (under torch.distributed.run with one rank): rehearse the RCCL path on a one-GPU box
    multi = world > 1 or (os.environ.get("DEVIS_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if os.environ.get("DEVIS_BENCH_PRESTREAM") == "1":
        from devis_amd.unet3d import _side_stream
        _side_stream(dev)
    if multi:
        import torch.distributed as dist
        from devis_amd.parallel import nccl_options
        dist.init_process_group("nccl", device_id=dev, pg_options=nccl_options())
    from devis_amd import TMSU
    from devis_amd.parallel import GradientAllReducer
    from devis_amd.optim import FusedAdam
    from devis_amd.train import adjust_learning_rate

    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    args = brats_args(dtype)
    torch.manual_seed(0)                       # identical replicas on every rank
    model = TMSU(args)
    model.train()
    opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=1e-5, module=model)   # trainUMIS.py:1394, one launch
    reducer = GradientAllReducer(model.parameters(), module=model, force=True) if multi else None
    g = torch.Generator().manual_seed(1234 + rank)
    S = a.size
    x = torch.randn((a.batch, 4, S, S, S), generator=g).to(dev)
    y = torch.randint(0, 4, (a.batch, S, S, S), generator=g).to(dev)
    epoch = 25

    def step():
        adjust_learning_rate(opt, epoch, args.end_epochs, args.lr)
        _, loss = model(x, y, epoch, 'train', 'BraTS')
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if reducer is not None:
            reducer.all_reduce()
        opt.step()
        return loss

    from devis_amd.train import freeze_gc
    freeze_gc()                                # as devis_amd.train.train() does before its loop
    trace = [] if os.environ.get("DEVIS_BENCH_TRACE_LOSS") == "1" else None   # debugging aid: every step's loss
    for _ in range(a.warmup):
        loss = step()
        if trace is not None:
            trace.append(loss.detach().clone())
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    host_dt = time.perf_counter() - t0         # the host has enqueued every step; the GPU may still be working
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = float(loss.item())
    if final_loss != final_loss or abs(final_loss) == float("inf"):
        raise SystemExit("bench.py: the training loss is not finite after %d steps (%r): the timing is void"
                         % (a.warmup + a.steps, final_loss))
    if trace is not None:
        print("warmup losses:", " ".join("%.6f" % float(v) for v in trace), file=sys.stderr)
    vox = world * a.batch * S ** 3 * a.steps
    out = {
        "metric": "train voxels/sec (fwd+bwd+evid loss), BraTS 4x128^3 K=4",
        "value": vox / dt, "unit": "voxels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "configs[2]/[3]: BraTS2019-shaped 3D U-Net(4,16,4) evidential training, "
                               "B=%d/GPU x 4x%d^3, K=4, epoch 25 (lambda 0.5), Adam+L2" % (a.batch, S),
                   "global_batch": world * a.batch, "parallelism": "dp%d" % world},
        "final_loss": final_loss,
        # host time to ENQUEUE a step (no sync inside the timed region): below ms_per_step = the GPU is the limit
        "host_enqueue_ms_per_step": host_dt / a.steps * 1e3,
        "launch_plan": bool(getattr(getattr(model, "backbone", None), "_plans", None)),
        "conv_mfma_frac_of_step": FLOP_PER_VOXEL_FWD_BWD * (vox / dt) / world / (PEAK_BF16_TFLOPS * 1e12),
    }
    if rank == 0:
        # The roofline line is about the kernel name with the largest summed time in the latest tracked kernel-stats
        # summary of this command (see roofline()); its probes run after the timed region, on the launch stream.
        if dtype == torch.bfloat16:
            out["roofline"], weighted = roofline(a.batch, S, dtype)
            if weighted is not None:
                out["roofline_top10_time_weighted"] = weighted
        else:
            # --dtype fp32 is the parity mode: generic VALU kernels (csrc/conv_direct.hip) with no roofline model;
            # the tracked kernel-stats summaries and the probes are about the bf16 product path
            out["roofline"] = None
            out["note"] = "fp32 parity mode (generic kernels): a secondary number, not the headline configuration"
        if world == 1 and dtype == torch.bfloat16 and not a.no_cpu_baseline:
            # secondary number: the same step at the reference's own precision (compute_dtype = float32: every conv on
            # v_mfma_f32_16x16x4_f32, csrc/conv_f32_mfma.hip; deterministic weight gradients) - not the headline
            out["fp32_parity_mode_ms_per_step"] = fp32_step_ms(a.batch, S, x, y)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_steps, S)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
