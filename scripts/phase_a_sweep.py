#!/usr/bin/env python3
"""Development aid (GPU box): phase-A time against the number of start chains for the speculative kernel
(auto) and the plain kernel with 16 and 64 lanes per chain — where should abn_plan switch?"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic

ctx = A.Context(0)
for W in [int(a) for a in sys.argv[1:]] or (10, 25, 50, 100, 200, 400):
    gens, D, p0, _ = synthetic.c4_windows(W)
    row = []
    for lanes in (0, 16, 32, 64):
        plan = A.Plan(ctx, gens, W, 10, 0, options=A.default_options(lanes_per_chain=lanes))
        plan.set_windows(D, p0)
        ms = []
        for _ in range(4):
            plan.run_phase(0)
            ms.append(plan.kernel_ms()["fit_starts"])
        row.append(f"lanes={lanes or 'auto'}: {min(ms):7.3f} ms")
        plan.close()
    print(f"{W * 10:5d} chains  " + "   ".join(row), flush=True)
