#!/usr/bin/env python3
"""Development aid: phase-B timing for mid-size pedigrees (N ~ 1e3) where the fit kernel streams rows."""
import sys, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic

ctx = A.Context(0)
for (nl, depth, every, B) in ((4, 60, 5, 4096), (6, 40, 2, 2048)):
    ped, p0 = synthetic.c5_pedigree(nl, depth, every)
    N = ped.shape[0]
    K = len({tuple(r) for r in ped[:, :3].astype(int).tolist()})
    plan = A.Plan(ctx, ped[:, :3], 1, 4, B, options=A.default_options())
    plan.set_windows(ped[:, 3][None, :], np.array([p0]))
    plan.run(); plan.run()
    ms = plan.kernel_ms()
    d = plan.download()
    ev = int(d["info_b"]["evals"].sum())
    print(json.dumps(dict(N=N, T=int(ped[:, :3].max()), K=K, B=B, lanes=int(d["info_b"]["lanes"][0, 0]),
                          ms={k: round(v, 2) for k, v in ms.items()}, evals_b=ev,
                          Mevals_per_s=round(ev / ms["fit_boot"] / 1e3, 1), max_evals=int(d["info_b"]["evals"].max()))))
    plan.close()
