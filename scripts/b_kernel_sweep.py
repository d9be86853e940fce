#!/usr/bin/env python3
"""Measurement aid (GPU box, knobs build: ABNEUTRAL_HIP_LIB=build/libabn_knobs.so): phase-B time of a pedigree
against the number of bootstraps for each kernel — speculative (four wavefronts per chain), one wavefront per chain,
packed — to place the thresholds of abn_api.hip.  usage: b_kernel_sweep.py <c2|c3|g351|sparse>"""
import json, os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic
import bench
which, B = sys.argv[1], int(sys.argv[2])
if which == "c3":
    ped, p0 = synthetic.c3_pedigree()
else:
    wl = bench.make_workload({"c2": "c2", "g351": "g351"}[which], 0, 1)
    ped = np.concatenate([wl["gens"], wl["D"][0][:, None]], axis=1); p0 = float(wl["p0"][0])
ctx = A.Context(0)
plan = A.Plan(ctx, ped[:, :3], 1, 10, B, options=A.default_options())
plan.set_windows(ped[:, 3][None, :], np.array([p0]))
plan.run(); plan.sync()
ms = []
for _ in range(5):
    plan.run_phase(1); ms.append(plan.kernel_ms()["fit_boot"])
print(json.dumps({"ms": min(ms)}))
''' % str(ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
for B in (500, 1000, 1500, 2000, 3000, 4000):
    row = {}
    for k in ("spec", "wide", "packed"):
        env = dict(os.environ, ABN_PHASE_B_KERNEL=k, ABNEUTRAL_HIP_LIB=str(ROOT / "build/libabn_knobs.so"))
        r = subprocess.run([sys.executable, "-c", code, which, str(B)], capture_output=True, text=True, env=env)
        try:
            row[k] = round(json.loads(r.stdout.strip().splitlines()[-1])["ms"], 3)
        except Exception:
            row[k] = r.stderr[-200:]
    print(which, B, row, flush=True)
