#!/usr/bin/env python3
"""Development aid (GPU box): replays tests/fuzz/fuzz_plan.py's generator for a seed and runs chosen case indices many times,
reporting runs whose download fails (a persistent launch that did not finish every chain) — no oracle needed.
usage: repro_lost_chain.py <seed> <repeats> <index> [index ...]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "fuzz"))
import alphabeta_rs_amd as A
from fuzz_parity import rand_ped

seed, reps, want = int(sys.argv[1]), int(sys.argv[2]), set(int(a) for a in sys.argv[3:])
rng = np.random.default_rng(seed)
ctx = A.Context(0)
for ci in range(max(want) + 1):
    n = int(rng.choice([3, 6, 40, 105, 200, 600])); tmax = int(rng.choice([1, 4, 8, 16]))
    ped = rand_ped(rng, n, tmax)
    W = int(rng.choice([1, 2, 5, 60])); S = int(rng.choice([1, 3, 10, 90])) if W < 60 else 90
    B = int(rng.choice([1, 4, 16]))
    big = n <= 200 and rng.random() < 0.15
    if big:
        W, B = int(rng.choice([3, 7])), int(rng.choice([2000, 5000])); S = int(rng.choice([3, 10]))
    mid = (not big) and n in (40, 105) and rng.random() < 0.12
    if mid:
        W, S, B = 1, int(rng.choice([3, 10])), int(rng.choice([8300, 10000, 12000]))
    strict = int(rng.choice([-1, 0, 1], p=[0.3, 0.55, 0.15])); skip_off = int(rng.random() < 0.2)
    mi_a, mi_b = int(rng.choice([1500, 3000])), int(rng.choice([100, 400])); variant = int(rng.integers(0, 2))
    seed_o = int(rng.integers(1, 1 << 30))
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.7, 1.3, (W, 1)) + rng.normal(0, 1e-4, (W, n)))
    p0 = rng.uniform(0.55, 0.95, W)
    woff, boff = int(rng.integers(0, 1000)), int(rng.integers(0, 5000))
    sm = int(rng.integers(0, 2))
    rng.integers(0, W)
    if ci not in want:
        continue
    o = A.default_options(seed=seed_o, shrink_on_failed_contraction=variant, max_iters_start=mi_a, max_iters_boot=mi_b,
                          stream_mode=sm, no_fixed_point_skip=skip_off, strict_order=strict)
    bad = 0
    first = None
    k = th = None
    for r in range(reps):
        plan = A.Plan(ctx, ped[:, :3], W, S, B, window_offset=woff, boot_offset=boff, options=o)
        plan.set_windows(D, p0)
        try:
            plan.run()
            out = plan.download(allow_failed_windows=True)
            if first is None:
                first = out["raw"].copy()
            elif not np.array_equal(first, out["raw"], equal_nan=True):
                print("  run", r, "DIFFERENT TABLE")
        except A.AbnError as e:
            bad += 1
            import re as _re
            m = _re.search(r"finished (\d+) of (\d+) chains \(error word (\d+), (\d+) handed", str(e))
            if m and int(m.group(1)) >= 0x10000:      # build/libabn_diag.so: the resume launch counts in the upper half
                v, tot, h = int(m.group(1)), int(m.group(2)), int(m.group(4))
                P, R = v & 0xffff, v >> 16
                if P + R != tot or R != h:
                    print("  run", r, f"persistent finished {P}, resume finished {R}, handed {h}, total {tot}", flush=True)
                else:
                    bad -= 1
            else:
                print("  run", r, "ERROR", str(e)[:220], flush=True)
        try:
            k = plan.last_kernels()
            th = plan.tail_handed()
        except A.AbnError:
            pass
        plan.close()
    print(ci, dict(n=n, tmax=tmax, W=W, S=S, B=B, strict=strict, skip_off=skip_off, mi_b=mi_b, variant=variant), "kernels", k, "handed", th,
          "failed runs", bad, "of", reps, flush=True)
