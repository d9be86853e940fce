#!/usr/bin/env python3
"""Development aid: distribution of Nelder-Mead iterations over the start fits of a metaprofile-shaped batch."""
import sys, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic
ctx = A.Context(0)
gens, D, p0, _ = synthetic.c4_windows(300)
plan = A.Plan(ctx, gens, 300, 100, 100, options=A.default_options())
plan.set_windows(D, p0)
plan.run()
d = plan.download()
it = d["info_a"]["iters"].ravel(); ev = d["info_a"]["evals"].ravel()
qs = [50, 90, 99, 99.9, 100]
print("phase A iters percentiles", {q: int(np.percentile(it, q)) for q in qs}, "mean", it.mean(), "evals mean", ev.mean())
for cap in (500, 1000, 2000, 5000, 9999):
    print("iters >", cap, int((it > cap).sum()), "of", it.size, " evals beyond cap (sum)", int(np.maximum(it - cap, 0).sum() * 1.7))
itb = d["info_b"]["iters"].ravel()
print("phase B iters percentiles", {q: int(np.percentile(itb, q)) for q in qs})
print("status A", np.bincount(d["info_a"]["status"].ravel()), "status B", np.bincount(d["info_b"]["status"].ravel()))
