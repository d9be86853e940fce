#!/usr/bin/env python3
"""Development aid: distribution of EXECUTED Nelder-Mead evaluations over the start fits of a metaprofile-shaped
batch (fits that hit argmin's fixed point end early under the default skip)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic
ctx = A.Context(0)
gens, D, p0, _ = synthetic.c4_windows(300)
res = {}
for ns in (1, 0):
    plan = A.Plan(ctx, gens, 300, 100, 100, options=A.default_options(no_fixed_point_skip=ns))
    plan.set_windows(D, p0)
    plan.run()
    res[ns] = plan.download()
    print("no_skip", ns, plan.kernel_ms(), plan.counters())
    plan.close()
it = res[1]["info_a"]["iters"].ravel(); ev = res[1]["info_a"]["evals"].ravel(); st = res[1]["info_a"]["status"].ravel()
# a stuck fit: runs to max_iters with exactly 2 evaluations per iteration at the end; executed evaluations under the skip
# are not in the outputs, so estimate them: fits with status MAX_ITERS are (almost all) stuck ones
stuck = st == 1
print("fits", it.size, "max_iters status", int(stuck.sum()))
qs = [50, 90, 99, 99.9, 100]
print("iterations of the fits that converge:", {q: int(np.percentile(it[~stuck], q)) for q in qs})
print("evaluations of the fits that converge:", {q: int(np.percentile(ev[~stuck], q)) for q in qs}, "mean", float(ev[~stuck].mean()))
