#!/usr/bin/env python3
"""GPU box: phase A of C3 (10 starts, the speculative kernel) — time per iteration of the longest chain."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic
ctx = A.Context(0)
ped, p0 = synthetic.c3_pedigree()
# a fixed amount of work whatever the arithmetic yields: 300 iterations, no convergence test, stuck fits executed
plan = A.Plan(ctx, ped[:, :3], 1, 10, 0, options=A.default_options(max_iters_start=300, sd_tolerance=0.0, no_fixed_point_skip=1))
plan.set_windows(ped[:, 3][None, :], np.array([p0]))
ms = []
for _ in range(6):
    plan.run_phase(0)
    ms.append(plan.kernel_ms()["fit_starts"])
it = plan.download(allow_failed_windows=True)["info_a"]["iters"][0]
print("ABN_SPEC_TABLES=" + os.environ.get("ABN_SPEC_TABLES", "-"), "phase A %.3f ms" % min(ms), "iters max", int(it.max()),
      "-> %.3f us per iteration" % (1e3 * min(ms) / it.max()))
