#!/usr/bin/env python3
"""Branch traces of BASELINE C3's 10 000 bootstrap fits for scripts/sched_sim.py (CPU; uses the oracle, hence under
tests/): which branch every Nelder-Mead iteration took (oracle.boot_model_trace).  usage: gen_branch_traces.py out.npz"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import alphabeta_rs_amd as A
import oracle as O
from alphabeta_rs_amd import synthetic

ped, p0 = synthetic.c3_pedigree()
seed, tree = 20260101, A.reduction_tree(ped[:, :3])
s0 = np.stack([O.start_simplex(seed, 0, s, ped[:, 3].max()) for s in range(10)])
fits = O.fit_batch(ped, p0, p0, 1.0, s0, 10000, lanes=tree)
k, model, pred, resid, _ = O.select_best(ped, p0, fits["best"])
raw, res, tr = O.boot_model_trace(ped, model, pred, resid, p0, p0, 1.0, seed, 0, 0, 10000, lanes=tree)
np.savez_compressed(sys.argv[1] if len(sys.argv) > 1 else "c3_traces.npz", tr=tr, evals=res["evals"], iters=res["iters"])
kinds = np.concatenate([tr[i, : res["iters"][i]] for i in range(len(tr))])
print("evaluations", int(res["evals"].sum()), "iterations", int(res["iters"].sum()),
      "branch frequencies (accept, expansion, contraction ok, contraction rejected, shrink)",
      (np.bincount(kinds, minlength=5) / len(kinds)).round(3))
