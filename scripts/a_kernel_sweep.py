#!/usr/bin/env python3
"""Measurement aid (GPU box, knobs build): phase-A time of the C3 topology against the number of start chains (W windows x 10
starts) for each kernel — speculative (four wavefronts per chain), one wavefront per chain, packed — to place
spec_max_chains for phase A (start chains differ far more in length than bootstrap chains).  usage: a_kernel_sweep.py [W ...]"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic
W = int(sys.argv[1])
S = 10
if len(sys.argv) > 2 and sys.argv[2] == "g351":   # the reference's 351-row golden pedigree: one window, W * 10 starts
    import bench
    wl = bench.make_workload("g351", 0, 1)
    gens, D, p0, S, W = wl["gens"], wl["D"], wl["p0"], W * 10, 1
else:
    gens, D, p0, _ = synthetic.c4_windows(W)
ctx = A.Context(0)
plan = A.Plan(ctx, gens, W, S, 0, options=A.default_options())
plan.set_windows(D, p0)
ms = []
for _ in range(5):
    plan.run_phase(0); ms.append(plan.kernel_ms()["fit_starts"])
print(json.dumps({"ms": min(ms)}))
''' % str(ROOT)
which = "g351" if "g351" in sys.argv[1:] else "c3-topology"
for W in [int(a) for a in sys.argv[1:] if a != "g351"] or (100, 125, 150, 200, 250, 300, 400):
    row = {}
    for k in ("spec", "wide", "packed"):
        env = dict(os.environ, ABN_PHASE_A_KERNEL=k, ABNEUTRAL_HIP_LIB=str(ROOT / "build/libabn_knobs.so"))
        r = subprocess.run([sys.executable, "-c", code, str(W), which], capture_output=True, text=True, env=env, cwd=str(ROOT))
        try:
            row[k] = round(json.loads(r.stdout.strip().splitlines()[-1])["ms"], 3)
        except Exception:
            row[k] = r.stderr[-200:]
    print(which, W * 10, "start chains", row, flush=True)
