#!/usr/bin/env python3
"""End-to-end time of the host-pointer entry points (abn_ab_neutral_run + abn_boot_model_run: allocation,
H2D, kernels, D2H) against the device-resident plan on BASELINE C3."""
import sys, time, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic
ctx = A.Context(0)
ped, p0 = synthetic.c3_pedigree()
o = A.default_options()
for _ in range(2):
    model, pred, resid, _ = ctx.ab_neutral_run(ped, p0, p0, 1.0, 10, options=o)
    raw, info = ctx.boot_model_run(ped, model, pred, resid, p0, p0, 1.0, 10000, options=o)
t = []
for _ in range(10):
    t0 = time.perf_counter()
    model, pred, resid, _ = ctx.ab_neutral_run(ped, p0, p0, 1.0, 10, options=o)
    t1 = time.perf_counter()
    raw, info = ctx.boot_model_run(ped, model, pred, resid, p0, p0, 1.0, 10000, options=o)
    t2 = time.perf_counter()
    t.append((t1 - t0, t2 - t1))
a = np.median([x[0] for x in t]) * 1e3; b = np.median([x[1] for x in t]) * 1e3
print(json.dumps(dict(ab_neutral_run_ms=round(a, 2), boot_model_run_ms=round(b, 2), total_ms=round(a + b, 2),
                      fits_per_s=round(10010 / ((a + b) * 1e-3)))))
