#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of one bench.py run (kernel trace + separate --pmc passes) into the small summaries committed
under profiles/.  Usage (scripts/profile_round.sh does all of it on the GPU box):
  python profiles/summarize_rocprof.py gpurun_out/prof_c3 r04_c3 c3 <repo root>
Writes  <tag>_kernel_stats.csv, <tag>_rocprof_summary.json  and  <round>_pmc_<workload>.json: per PHASE of a step
(fit_starts = the fit launches before the selection kernel, fit_boot = those after it — two phases may run the same
kernel on the same grid, so the split is by dispatch order, not by name; the launches of a phase are added up) the averages
of every counter collected, the HBM bytes derived from them and the content hash of the kernel sources they belong to (bench.py quotes
them only for that very build).

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md §HBM): hbm_bytes_per_launch = 2 * FETCH + WRITE, as that section prescribes."""
import collections
import csv
import datetime
import glob
import hashlib
import json
import os
import shutil
import sys
from pathlib import Path

src, tag, workload = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
out = Path(__file__).resolve().parent
root = Path(sys.argv[4]) if len(sys.argv) > 4 else out.parent


def newest(pattern):   # gpurun merges every run's files into the same directories: take the latest
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


def phases_of(rows, name_key, order_key):
    """[(step, phase, row)] for the fit launches of `rows` in dispatch order.  A step of a plan: a buffer fill (the skip
    counters are cleared), the fit launches of phase A, the two selection kernels, fills, (abn_make_dstar_kernel,) the fit
    launches of phase B — a persistent launch may be followed by its tail's resume launch on abn_fit_spec_kernel, so a phase
    can hold two fit launches: phase B lasts from the selection until the first buffer fill behind one of its fit launches."""
    state, step, out_rows = "fit_starts", 0, []
    for r in sorted(rows, key=lambda r: int(r[order_key])):
        n = r[name_key]
        if "abn_select" in n:
            state = "fit_boot_pending"
        elif "fillBuffer" in n:
            if state == "fit_boot":
                state, step = "fit_starts", step + 1
        elif "abn_fit_" in n or "abn_make_dstar" in n:
            if state == "fit_boot_pending" and "abn_fit_" in n:
                state = "fit_boot"
            out_rows.append((step, "fit_boot" if state.startswith("fit_boot") else "fit_starts", r))
    return out_rows


stats = newest(str(src / "kt" / "*" / "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats, out / f"{tag}_kernel_stats.csv")

summary = {"workload": workload, "kernels": {}}
trace = newest(str(src / "kt" / "*" / "*_kernel_trace.csv"))
if trace:
    rows = list(csv.DictReader(open(trace)))
    dur = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        if "abn_fit_" in name:    # abn_fit_kernel / abn_fit_refill_kernel / abn_fit_spec_kernel
            name += f" grid={r['Grid_Size_X']}"
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        summary["kernels"][k] = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
    ph = collections.defaultdict(lambda: collections.defaultdict(float))
    for st, p, r in phases_of(rows, "Kernel_Name", "Dispatch_Id"):
        if "abn_fit_" in r["Kernel_Name"]:
            ph[p][st] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    summary["fit_launches_us_per_step_by_phase"] = {p: sum(v.values()) / len(v) for p, v in ph.items()}

# ---- counters, per phase
pmc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))  # phase, counter, step
kern = collections.defaultdict(list)
for d in sorted(glob.glob(str(src / "pmc_*"))):
    f = newest(d + "/*/*_counter_collection.csv")
    if not f:
        continue
    rows = list(csv.DictReader(open(f)))
    # one row per (dispatch, counter): split the dispatches into steps and phases once per file
    seen, disp = set(), []
    for r in rows:
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            disp.append(r)
    where = {r["Dispatch_Id"]: (st, p) for st, p, r in phases_of(disp, "Kernel_Name", "Dispatch_Id")
             if "abn_fit_" in r["Kernel_Name"]}
    for r in rows:
        sp = where.get(r["Dispatch_Id"])
        if sp:
            st, p = sp
            pmc[p][r["Counter_Name"]][st] += float(r["Counter_Value"])   # a phase's launches of one step add up
            k = f"{r['Kernel_Name']} grid={r['Grid_Size']}"
            if k not in kern[p]:
                kern[p].append(k)
summary["pmc_avg_per_phase_of_a_step"] = {p: {c: sum(x.values()) / len(x) for c, x in v.items()} for p, v in pmc.items()}

h = hashlib.sha1()  # the same content hash bench.py computes: the figures belong to these kernel sources
for f in sorted((root / "alphabeta_rs_amd" / "csrc").glob("*")):
    if f.name == "abn_multi.hip":   # as bench.py: device orchestration, no kernel
        continue
    if f.suffix in {".hip", ".hpp", ".h"}:
        h.update(f.name.encode())
        h.update(f.read_bytes())
phases = {}
for p, c in summary["pmc_avg_per_phase_of_a_step"].items():
    e = {"kernel": " + ".join(kern[p]), **{k: v for k, v in c.items()}}
    if "FETCH_SIZE" in c:
        e["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024
        e["hbm_bytes_has_write_pass"] = "WRITE_SIZE" in c
    phases[p] = e
if phases:
    (out / f"{tag.split('_')[0]}_pmc_{workload}.json").write_text(json.dumps(
        {"workload": workload, "phases": phases, "source_sha1": h.hexdigest(),
         "units": "averages per PHASE of a step (its fit launches added up: a persistent launch and its tail's resume launch "
                  "count together; the key hbm_bytes_per_launch is kept for the phase total); FETCH_SIZE / WRITE_SIZE in "
                  "KiB, hbm_bytes = (2 FETCH + WRITE) x 1024 (gfx950 "
                  "correction); SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES / SQ_BUSY_CYCLES in quad-cycles summed over "
                  "wavefronts (resp. SQs); SQ_INSTS_* in wavefront instructions",
         "collected": datetime.date.today().isoformat()}, indent=1))
(out / f"{tag}_rocprof_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1))
