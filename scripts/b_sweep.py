#!/usr/bin/env python3
"""Development aid (GPU box): phase-B time of the C3 pedigree against the number of bootstraps — the wavefronts
per SIMD quantisation (1024 SIMDs; 4 chains per wavefront at 16 lanes per chain)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic
ctx = A.Context(0)
ped, p0 = synthetic.c3_pedigree()
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for B in (2048, 4096, 6144, 8192, 9000, 10000, 11000, 12288, 13000, 16384):
    plan = A.Plan(ctx, ped[:, :3], 1, 10, B, options=A.default_options(lanes_per_chain=lanes))
    plan.set_windows(ped[:, 3][None, :], np.array([p0]))
    ms = []
    for _ in range(4):
        plan.run()
        ms.append(plan.kernel_ms()["fit_boot"])
    code = int(plan.download()["info_b"]["lanes"][0, 0])
    print(f"B={B:6d} lanes={code} waves={B * code // 64:5d} fit_boot {min(ms):7.3f} ms  {B / min(ms) / 1e3:7.2f} M fits/s", flush=True)
    plan.close()
