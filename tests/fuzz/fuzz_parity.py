#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (development aid; the deterministic cases live in tests/).
Random pedigrees (rows, generations), lane counts, optimiser variants, start simplices incl. wild ones."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import alphabeta_rs_amd as A
import oracle as O

def rand_ped(rng, n, tmax):
    t0 = np.where(rng.random(n) < 0.3, rng.integers(0, max(1, tmax // 2 + 1), n), 0)
    t1 = t0 + rng.integers(0, tmax - t0 + 1)
    t2 = t0 + rng.integers(0, tmax - t0 + 1)
    d = np.abs(rng.normal(0.01, 0.004, n))
    return np.stack([t0, t1, t2, d], axis=1).astype(np.float64)

def main(seconds=120, seed=0, max_cases=None):
    ctx = A.Context(0)
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    cases = fails = 0
    t_note = time.time()
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        n = int(rng.choice([1, 2, 5, 16, 17, 64, 65, 105, 129, 300, 513, 700]))
        tmax = int(rng.choice([0, 1, 3, 8, 20, 40]))
        ped = rand_ped(rng, n, tmax)
        p0 = float(rng.uniform(0.5, 0.99))
        lanes = int(rng.choice([0, 8, 16, 32, 64]))
        variant = int(rng.integers(0, 2))
        f = int(rng.integers(1, 9))
        iters = int(rng.choice([50, 200, 600]))
        s0 = A.gen_start_simplices(int(rng.integers(1, 1 << 30)), 0, f, ped[:, 3].max())
        if rng.random() < 0.3:
            s0 *= rng.uniform(-3, 50, s0.shape)           # wild starts: negative rates, huge weights
        if rng.random() < 0.1:
            s0[0, int(rng.integers(0, 5)), int(rng.integers(0, 4))] = np.nan
        # serial row-order sums (1; and 0 = auto for pedigrees of up to 16 rows): info.lanes == 1, oracle lanes = 1; -1: the tree
        strict = int(rng.choice([-1, 0, 1], p=[0.25, 0.6, 0.15]))
        o = A.default_options(lanes_per_chain=lanes, shrink_on_failed_contraction=variant, strict_order=strict)
        eqp, ew = float(rng.uniform(0.3, 0.9)), float(rng.choice([0.0, 0.7, 1.0]))
        try:
            best, info = ctx.fit_batch(ped, p0, eqp, ew, s0, iters, options=o)
        except A.AbnError as e:                            # explicit lane count x many generations x many triples
            if "more LDS per workgroup" in str(e) and lanes in (8, 16, 32):
                continue
            raise
        code = int(info["lanes"][0])
        want = O.fit_batch(ped, p0, eqp, ew, s0, iters, shrink_variant=variant, lanes=code)
        ok = (np.array_equal(info["status"], want["status"]) and np.array_equal(info["iters"], want["iters"])
              and np.array_equal(info["evals"], want["evals"]))
        good = want["status"] != 2
        ok = ok and np.array_equal(best[good], want["best"][good], equal_nan=True)
        ok = ok and np.array_equal(info["best_cost"][good], want["best_cost"][good], equal_nan=True)
        cases += 1
        if time.time() - t_note > 60:                     # a sign of life for the GPU box's silence guard
            t_note = time.time()
            print(f"... {cases} cases, {fails} mismatches so far", flush=True)
        if not ok:
            fails += 1
            print("MISMATCH", dict(n=n, tmax=tmax, lanes=lanes, code=code, variant=variant, f=f, iters=iters, ew=ew),
                  info["status"], want["status"], info["iters"], want["iters"], flush=True)
    print(f"fuzz: {cases} cases, {fails} mismatches")
    return fails

if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
