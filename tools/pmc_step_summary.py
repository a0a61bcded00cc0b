#!/usr/bin/env python3
"""Merge the counter passes of tools/pmc_step.sh per kernel name -> JSON on stdout.
  hbm bytes = 2 x FETCH_SIZE (gfx950 tallies a wide coalesced read at half its bytes, MI355X_MICROARCH.md 'HBM')
              + WRITE_SIZE, both reported in KiB by rocprofv3;
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs): the share of the kernel's cycles in which a SIMD's
              matrix pipe is busy, averaged over the chip's 1024 SIMDs.  cycles = GRBM_GUI_ACTIVE / 8: rocprofv3 sums the
              counter over the 8 XCDs (one GRBM each); clock_GHz = cycles / duration is printed as the sanity check."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
steps = 5                                            # bench.py --steps 3 --warmup 2
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("FETCH_SIZE", "GRBM_GUI_ACTIVE") and r.get("End_Timestamp"):
            dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = {}
for k, c in acc.items():
    if not k.startswith(("void devis", "devis", "_ZN5devis")):
        continue
    mean = lambda name: (sum(c[name]) / len(c[name])) if c.get(name) else 0.0
    n = max(len(v) for v in c.values())
    us = sum(dur[k]) / len(dur[k]) if dur[k] else 0.0
    fetch, write = mean("FETCH_SIZE") * 1024.0 * 2.0, mean("WRITE_SIZE") * 1024.0
    cyc = mean("GRBM_GUI_ACTIVE") / 8.0
    out[k] = {"launches_per_step": n / steps, "mean_us": us, "us_per_step": us * n / steps,
              "fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write,
              "hbm_GBps": (fetch + write) / us / 1e3 if us else 0.0,
              "mfma_busy_cycles": mean("SQ_VALU_MFMA_BUSY_CYCLES"), "insts_mfma": mean("SQ_INSTS_MFMA"),
              "insts_valu": mean("SQ_INSTS_VALU"), "gui_active_cycles": cyc, "clock_GHz": cyc / us / 1e3 if us else 0.0,
              "mfma_util": mean("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0) if cyc else 0.0,
              "wave_quad_cycles": mean("SQ_WAVE_CYCLES"), "wait_any_quad_cycles": mean("SQ_WAIT_ANY"),
              "active_inst_any_quad_cycles": mean("SQ_ACTIVE_INST_ANY"), "active_inst_valu_quad_cycles": mean("SQ_ACTIVE_INST_VALU"),
              "active_inst_lds_quad_cycles": mean("SQ_ACTIVE_INST_LDS")}
json.dump({"workload": "bench.py --steps 3 --warmup 2, B=2 x 4x128^3 bf16, DEVIS_WGRAD_STREAM=0", "kernels": out}, sys.stdout, indent=1)
