#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of one bench.py run (kernel trace + separate --pmc passes) into the small
summaries committed under profiles/.  Usage (scripts/profile_round.sh does all of it on the GPU box):
  python profiles/summarize_rocprof.py gpurun_out/prof_c3 r01_c3 c3
Splits the fit launches of a step by kernel and grid size (phase A = starts, phase B = bootstraps: the largest grid).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md §HBM), so hbm_bytes_per_launch = 2*FETCH + WRITE is an upper-side estimate here
(this kernel's loads are 4- and 8-byte per lane, an uncalibrated width)."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

src, tag, workload = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
out = Path(__file__).resolve().parent

import os


def newest(pattern):   # gpurun merges every run's files into the same directories: take the latest
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


stats = newest(str(src / "kt" / "*" / "*_kernel_stats.csv"))
shutil.copy(stats, out / f"{tag}_kernel_stats.csv")

trace = newest(str(src / "kt" / "*" / "*_kernel_trace.csv"))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    name = r["Kernel_Name"]
    if "abn_fit_" in name:    # abn_fit_kernel / abn_fit_refill_kernel / abn_fit_spec_kernel
        name += f" grid={r['Grid_Size_X']}"
    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
summary = {"workload": workload, "kernels": {}}
for k, v in dur.items():
    summary["kernels"][k] = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = newest(str(src / d / "*" / "*_counter_collection.csv"))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "abn_fit_" in name:
            name += f" grid={r['Grid_Size']}"
        pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary["pmc_avg_per_launch"] = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in pmc.items()}
big = [k for k in summary["pmc_avg_per_launch"] if "abn_fit_" in k]   # phase B = the launch with the largest grid
if big:
    b = max(big, key=lambda k: int(k.split("grid=")[1]))
    p = summary["pmc_avg_per_launch"][b]
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        hbm = (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
        summary["phase_b_kernel"] = b
        summary["hbm_bytes_per_launch"] = hbm
        import datetime
        import hashlib

        h = hashlib.sha1()  # the same content hash bench.py computes: the figure belongs to these kernel sources
        root = Path(sys.argv[4]) if len(sys.argv) > 4 else out.parent
        for f in sorted((root / "alphabeta_rs_amd" / "csrc").glob("*")):
            if f.name == "abn_multi.hip":   # as bench.py: device orchestration, no kernel
                continue
            if f.suffix in {".hip", ".hpp", ".h"}:
                h.update(f.name.encode())
                h.update(f.read_bytes())
        (out / f"{tag.split('_')[0]}_pmc_fit_boot_{workload}.json").write_text(json.dumps(
            {"workload": workload, "kernel": b, "FETCH_SIZE_KiB": p["FETCH_SIZE"], "WRITE_SIZE_KiB": p["WRITE_SIZE"],
             "hbm_bytes_per_launch": hbm,
             # wavefront-instruction counts of the same launch (SQ pass): the vector pipe issues one per SIMD every 4 cycles
             **{k.lower() + "_per_launch": p[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES") if k in p},
             "source_sha1": h.hexdigest(),
             "collected": datetime.date.today().isoformat()}, indent=1))
(out / f"{tag}_rocprof_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1))
