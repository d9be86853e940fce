#!/usr/bin/env python3
"""Pairwise divergence (src/pedigree.rs:210-261) on the MI355X vs the CPU oracle: end-to-end call time
(host buffers in, PCIe included) for a genome-scale input."""
import sys, time, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import alphabeta_rs_amd as A
import oracle as O

ctx = A.Context(0)
rng = np.random.default_rng(1)
for n, L in ((15, 4_000_000), (50, 2_000_000)):
    status = rng.integers(0, 3, size=(n, L), dtype=np.uint8)
    pmax = rng.uniform(0.95, 1.0, size=(n, L))
    codes = (status | np.where(pmax < 0.99, 0x80, 0)).astype(np.uint8)
    ctx.pairwise_divergence(codes[:, :1000])
    t0 = time.perf_counter(); d, b, v = ctx.pairwise_divergence(codes); t_gpu = time.perf_counter() - t0
    Ls = L // 20
    t0 = time.perf_counter(); wd, wb, wv = O.pairwise_divergence(status[:, :Ls], pmax[:, :Ls], 0.99); t_cpu = (time.perf_counter() - t0) * 20
    gd, gb, gv = ctx.pairwise_divergence(codes[:, :Ls])
    assert np.array_equal(gd, wd) and np.array_equal(gb, wb)
    npairs = n * (n - 1) // 2
    print(json.dumps(dict(n=n, sites=L, pairs=npairs, gpu_call_ms=round(t_gpu * 1e3, 2), cpu_oracle_1thread_ms_est=round(t_cpu * 1e3),
                          site_pairs_per_s=f"{npairs * L / t_gpu:.3g}", codes_GBps_end_to_end=round(n * L / t_gpu / 1e9, 2))))
