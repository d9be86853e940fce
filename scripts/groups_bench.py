#!/usr/bin/env python3
"""Development aid: C4 shard wall time per step versus the number of concurrent window groups."""
import sys, time, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic
ctx = A.Context(0)
gens, D, p0, _ = synthetic.c4_windows(25)
for g in (1, 2, 3, 4, 8):
    plan = A.Plan(ctx, gens, 25, 10, 1000, options=A.default_options(window_groups=g))
    plan.set_windows(D, p0)
    plan.run(); plan.sync()
    t0 = time.perf_counter()
    for _ in range(4):
        plan.run()
    plan.sync()
    dt = (time.perf_counter() - t0) / 4
    print(g, round(dt * 1e3, 2), "ms/step", {k: round(v, 2) for k, v in plan.kernel_ms().items()})
    plan.close()
