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


def kernel_probe(batch, size, dtype):
    """Time the dominant kernel family (3x3x3 conv, 16->16 channels at full resolution) with HIP events
    on the stream the kernels are launched on (torch's current stream)."""
    from devis_amd import ops
    from devis_amd.unet3d import G3
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((batch, size, size, size, 16), generator=g).to(dtype).to(dev)
    dy = torch.randn((batch, size, size, size, 16), generator=g).to(dtype).to(dev)
    w = (torch.randn((16, 16, 3, 3, 3), generator=g) * 0.05).to(dev)
    b = torch.zeros(16, device=dev)
    wp = ops.pack_conv_weight(w, dtype)
    sc = torch.ones((batch, 16), device=dev)
    sh = torch.zeros((batch, 16), device=dev)
    vox = batch * size ** 3
    flops = 2.0 * 27 * 16 * 16 * vox
    esz = 2 if dtype == torch.bfloat16 else 4
    res = {}

    def timeit(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    # the roofline kernel: launched through the C ABI into a preallocated output, 20 back-to-back launches between the
    # two events (the Python wrapper's allocation and checks would otherwise sit between launches)
    import ctypes as C
    from devis_amd import _lib
    lib = _lib.load()
    y = torch.empty_like(x)
    d = G3.desc(batch, (size, size, size), 16, 0, 16, dtype)
    stream = torch.cuda.current_stream().cuda_stream

    def fwd_direct():
        _lib.check(lib.devis_conv_fwd(C.byref(d), x.data_ptr(), None, wp.data_ptr(), b.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                      None, y.data_ptr(), stream), "conv_fwd")
    res["conv3d_fwd_c16"] = (timeit(fwd_direct, reps=20), flops, vox * 16 * esz * 2)
    res["conv3d_dgrad_c16"] = (timeit(lambda: ops.conv_dgrad(dy, wp, G3, x.shape)), flops, vox * 16 * esz * 2)
    res["conv3d_wgrad_c16"] = (timeit(lambda: ops.conv_wgrad(x, None, dy, G3, w.shape, sc, sh)), flops, vox * 16 * esz * 2)
    # the fused evidential head + loss (north_star's second roofline): channels-last K=4 logits, int64 labels.
    # Algorithmic bytes: forward reads K*s + 8 and writes the fp32 evidence the train mode returns
    # (K*4); backward re-reads K*s + 8 and writes K*s.  FLOP column: 0 (transcendental-bound).
    z = torch.randn((batch, size, size, size, 4), generator=g).to(dtype).to(dev)
    lab = torch.randint(0, 4, (batch, size, size, size), generator=g).to(dev)
    one = torch.ones((), device=dev)
    scal, _, _ = ops.evid_loss_fwd(z, True, lab, 0.5, True)
    res["evid_head_loss_fwd"] = (timeit(lambda: ops.evid_loss_fwd(z, True, lab, 0.5, True)), 0.0, vox * (4 * esz + 8 + 16))
    res["evid_head_loss_bwd"] = (timeit(lambda: ops.evid_loss_bwd(z, True, lab, 0.5, scal, one)), 0.0, vox * (4 * esz + 8 + 4 * esz))
    return res


def pmc_lookup(name_parts, voxels, dtype):
    """(HBM bytes per launch, MFMA-pipe busy fraction) of the kernel whose name contains every string of
    `name_parts`, from profiles/r02_pmc_step.json (counters taken at B=2 x 128^3; bytes scale with the voxels)."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_step.json")
    if dtype != torch.bfloat16 or not os.path.exists(path):
        return None, None
    try:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    for k, v in kernels.items():
        if all(p_ in k for p_ in name_parts):
            return v["hbm_bytes"] * voxels / (2 * 128 ** 3), v["mfma_util"]
    return None, None


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
        probe = kernel_probe(a.batch, S, dtype)
        # Dominant kernel family of the step by summed time (profiles/r01_*kernel_stats.csv): the 3x3x3 conv at
        # 16 channels and full resolution, k_conv3_c16_march<8,32,...> (forward and, with tap-flipped weights,
        # dgrad).  Its arithmetic intensity (27*16*16*2 FLOP per 64 B = 216 FLOP/B) is below the MI355X ridge
        # (2.5 PF / 8 TB/s = 312), so the roof that binds it is HBM; the MFMA fraction is reported next to it.
        name = "conv3d_fwd_c16"
        t, flops, alg_bytes = probe[name]
        # HBM bytes per launch of THIS kernel from the tracked counter passes (profiles/r02_pmc_step.json, made by
        # tools/pmc_step.sh on the shipped build: 2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md "HBM"), looked up by
        # kernel name and scaled with the voxel count; null when the file or the kernel is missing.
        traffic, pmc_mfma_util = pmc_lookup(("k_conv3_c16_march<8, 32, true, false, false, true, 0, false>",), a.batch * S ** 3, dtype)
        out["roofline"] = {"kernel": "k_conv3_c16_march<8, 32, true, false, false, true, 0, false> (" + name + ": 16->16 3x3x3, GroupNorm+ReLU prologue)", "bound": "hbm",
                           "achieved": alg_bytes / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": alg_bytes / t / 1e9 / PEAK_HBM_GBS, "traffic": traffic,
                           "avg_launch_ms": t * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
                           "mfma_achieved_TFLOPs": flops / t / 1e12, "mfma_frac": flops / t / 1e12 / PEAK_BF16_TFLOPS,
                           "mfma_pipe_busy_frac_pmc": pmc_mfma_util,
                           "all": {k: {"ms": v[0] * 1e3, "TFLOP/s": v[1] / v[0] / 1e12, "GB/s_alg": v[2] / v[0] / 1e9}
                                   for k, v in probe.items()}}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_steps, S)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
