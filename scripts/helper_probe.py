#!/usr/bin/env python3
"""GPU box: one wavefront of the packed kernel (16 lanes per chain) with ONE long chain and three chains that converge at
once (degenerate start simplices): how much faster does the long chain finish when the idle groups evaluate its
expansion / contraction points?  Run with ABNEUTRAL_HIP_LIB=build/libabn_knobs.so (-DABN_MEASUREMENT_KNOBS
-DABN_HELPER_GROUPS on the sources patched with scripts/attic/helper_groups_*_r03.patch, see scripts/helpers_ab.sh) and ABN_HELPERS=0 / 1.  Measured: 0.973 -> 0.690 ms."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic

ctx = A.Context(0)
ped, p0 = synthetic.c3_pedigree()
s0 = A.gen_start_simplices(7, 0, 4, ped[:, 3].max())
for k in (1, 2, 3):
    s0[k, :, :] = s0[k, 0, :]          # five equal vertices: SD of the costs is 0 -> converged at Solver::init
opts = A.default_options(lanes_per_chain=16)
for rep in range(3):
    t0 = time.perf_counter()
    best, info = ctx.fit_batch(ped, p0, p0, 1.0, s0, 10000, options=opts)
    dt = time.perf_counter() - t0
print(f"helpers={os.environ.get('ABN_HELPERS', 'default')}: {dt * 1e3:.3f} ms, evals {info['evals'].tolist()}, "
      f"iters {info['iters'].tolist()}, {dt * 1e6 / info['iters'][0]:.2f} us per iteration of the long chain")
