#!/usr/bin/env python3
"""Development aid (GPU box): phase-B time against the pedigree size N (synthetic c5-style trees of depth 30,
sampled every 3 generations; 2000 bootstraps) — looks for performance cliffs at the kernel-variant boundaries
(lanes per chain, resident / stream, pair / deep stream loop)."""
import sys, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic

ctx = A.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 0
NLS = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20)
for nl in NLS:
    ped, p0 = synthetic.c5_pedigree(nl, 30, 3)
    N = ped.shape[0]
    K = len({tuple(r) for r in ped[:, :3].astype(int).tolist()})
    plan = A.Plan(ctx, ped[:, :3], 1, 4, B, options=A.default_options(lanes_per_chain=LANES))
    plan.set_windows(ped[:, 3][None, :], np.array([p0]))
    plan.run(); plan.run()
    ms = plan.kernel_ms()
    d = plan.download()
    ev = int(d["info_b"]["evals"].sum()) - plan.counters()["evals_skipped_boot"]
    code = int(d["info_b"]["lanes"][0, 0])
    print(json.dumps(dict(N=N, K=K, lanes=code & 0xff, stream=bool(code >> 8), fit_boot_ms=round(ms["fit_boot"], 3),
                          us_per_eval_row=round(1e3 * ms["fit_boot"] / ev / N * 1e3, 4), evals=ev,
                          Mevals_per_s=round(ev / ms["fit_boot"] / 1e3, 1), max_evals=int(d["info_b"]["evals"].max()))),
          flush=True)
    plan.close()
