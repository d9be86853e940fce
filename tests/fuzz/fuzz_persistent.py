#!/usr/bin/env python3
"""Race hunt on the persistent kernel (development aid, GPU box; no oracle: fast): random shapes that put phase B on the
time-sliced persistent launch with its tail hand-over — short and long chains, both optimiser variants, stuck fits executed
or skipped, deep and shallow queues — each plan run REPS times; every download must succeed (a lost chain is ABN_ERR_HIP) and
every bootstrap table must be byte-identical to the first run's.  usage: fuzz_persistent.py [seconds] [seed] [reps]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import alphabeta_rs_amd as A
from fuzz_parity import rand_ped


def main(seconds=300, seed=0, reps=4):
    ctx = A.Context(0)
    rng = np.random.default_rng(seed)
    t_end, t_note = time.time() + seconds, time.time()
    cases = launches = errors = diffs = handed_cases = 0
    while time.time() < t_end:
        n = int(rng.choice([3, 6, 40, 105, 200]))
        ped = rand_ped(rng, n, int(rng.choice([1, 4, 8, 16])))
        W = int(rng.choice([1, 3, 7, 25]))
        total = int(rng.choice([8300, 10000, 14000, 25000, 40000]))
        B, S = max(1, total // W), int(rng.choice([1, 3, 10]))
        o = A.default_options(seed=int(rng.integers(1, 1 << 30)), shrink_on_failed_contraction=int(rng.integers(0, 2)),
                              max_iters_start=int(rng.choice([200, 1500])), max_iters_boot=int(rng.choice([30, 100, 400, 1000])),
                              no_fixed_point_skip=int(rng.random() < 0.3), strict_order=int(rng.choice([-1, 0])))
        D = np.abs(ped[:, 3][None, :] * rng.uniform(0.7, 1.3, (W, 1)) + rng.normal(0, 1e-4, (W, n)))
        p0 = rng.uniform(0.55, 0.95, W)
        first, handed = None, 0
        for _ in range(reps):
            plan = A.Plan(ctx, ped[:, :3], W, S, B, options=o)
            plan.set_windows(D, p0)
            try:
                plan.run()
                out = plan.download(allow_failed_windows=True)
                handed = max(handed, plan.tail_handed()[1])
                if first is None:
                    first = out["raw"].copy()
                elif not np.array_equal(first, out["raw"], equal_nan=True):
                    diffs += 1
                    print("DIFFERENT TABLE", dict(n=n, W=W, B=B), flush=True)
            except A.AbnError as e:
                errors += 1
                print("ERROR", dict(n=n, W=W, B=B, mi_b=o.max_iters_boot, variant=o.shrink_on_failed_contraction), str(e)[:200], flush=True)
            plan.close()
            launches += 1
        cases += 1
        handed_cases += handed > 0
        if time.time() - t_note > 60:
            t_note = time.time()
            print(f"... {cases} shapes, {launches} launches, {errors} errors, {diffs} differing tables", flush=True)
    print(f"fuzz_persistent: {cases} shapes ({handed_cases} with a hand-over), {launches} launches, {errors} errors, {diffs} differing tables")
    return errors + diffs


if __name__ == "__main__":
    a = sys.argv[1:]
    sys.exit(1 if main(int(a[0]) if a else 300, int(a[1]) if len(a) > 1 else 0, int(a[2]) if len(a) > 2 else 4) else 0)
