#!/usr/bin/env python3
"""Randomised plan-level parity sweep (development aid): phase A (speculative / two-pass / plain / persistent),
selection, phase B (resident / stream / persistent) against per-window oracle runs."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))   # fuzz_parity, when loaded from tests/test_gpu_fuzz_slice.py
import alphabeta_rs_amd as A
import oracle as O
from fuzz_parity import rand_ped

def draw_case(rng):
    """One random plan of the sweep: every draw from `rng` that precedes the launch (replayable: tests/test_gpu_parity.py
    re-creates case 371 of seed 505, which once lost a chain of the tail hand-over)."""
    n = int(rng.choice([3, 6, 40, 105, 200, 600]))
    tmax = int(rng.choice([1, 4, 8, 16]))
    ped = rand_ped(rng, n, tmax)
    W = int(rng.choice([1, 2, 5, 60]))
    S = int(rng.choice([1, 3, 10, 90])) if W < 60 else 90       # 60 x 90 > 4096 -> two-pass phase A
    B = int(rng.choice([1, 4, 16]))
    big = n <= 200 and rng.random() < 0.15        # > 3072 wavefronts: the persistent (queue) kernel
    if big:
        W, B = int(rng.choice([3, 7])), int(rng.choice([2000, 5000]))
        S = int(rng.choice([3, 10]))
    mid = (not big) and n in (40, 105) and rng.random() < 0.12   # one window that just about fills the GPU: 2048 persistent wavefronts
    if mid:
        W, S, B = 1, int(rng.choice([3, 10])), int(rng.choice([8300, 10000, 12000]))
    # 1: serial row-order sums (oracle lanes = 1); 0: auto (serial up to 16 rows: n = 3, 6); -1: the tree whatever the size
    strict = int(rng.choice([-1, 0, 1], p=[0.3, 0.55, 0.15]))
    skip_off = int(rng.random() < 0.2)
    mi_a, mi_b = int(rng.choice([1500, 3000])), int(rng.choice([100, 400]))
    variant = int(rng.integers(0, 2))
    seed_o = int(rng.integers(1, 1 << 30))
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.7, 1.3, (W, 1)) + rng.normal(0, 1e-4, (W, n)))
    p0 = rng.uniform(0.55, 0.95, W)
    woff, boff = int(rng.integers(0, 1000)), int(rng.integers(0, 5000))
    o = A.default_options(seed=seed_o, shrink_on_failed_contraction=variant, max_iters_start=mi_a, max_iters_boot=mi_b,
                          stream_mode=int(rng.integers(0, 2)), no_fixed_point_skip=skip_off, strict_order=strict)
    return dict(n=n, tmax=tmax, ped=ped, W=W, S=S, B=B, strict=strict, mi_a=mi_a, mi_b=mi_b, variant=variant, seed_o=seed_o,
                D=D, p0=p0, woff=woff, boff=boff, options=o)


def main(seconds=180, seed=0, max_cases=None):
    ctx = A.Context(0)
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    cases = fails = 0
    t_note = time.time()
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        k_ = draw_case(rng)
        n, tmax, ped, W, S, B, strict = k_["n"], k_["tmax"], k_["ped"], k_["W"], k_["S"], k_["B"], k_["strict"]
        mi_a, mi_b, variant, seed_o, D, p0, woff, boff, o = (k_["mi_a"], k_["mi_b"], k_["variant"], k_["seed_o"], k_["D"], k_["p0"],
                                                             k_["woff"], k_["boff"], k_["options"])
        plan = A.Plan(ctx, ped[:, :3], W, S, B, window_offset=woff, boot_offset=boff, options=o)
        plan.set_windows(D, p0)
        plan.run()
        out = plan.download(allow_failed_windows=True)
        plan.close()
        la, lb = int(out["info_a"]["lanes"][0, 0]), int(out["info_b"]["lanes"][0, 0])
        ok = True
        for w in sorted(set([0, W - 1, int(rng.integers(0, W))])):
            pw = np.concatenate([ped[:, :3], D[w][:, None]], axis=1)
            s0 = A.gen_start_simplices(seed_o, woff + w, S, D[w].max())
            fits = O.fit_batch(pw, p0[w], p0[w], 1.0, s0, mi_a, shrink_variant=variant, lanes=la)
            ok = ok and np.array_equal(out["info_a"]["iters"][w], fits["iters"]) and np.array_equal(out["info_a"]["evals"][w], fits["evals"])
            k, model, pred, resid, _ = O.select_best(pw, p0[w], fits["best"])
            ok = ok and out["best_start"][w] == k
            if k >= 0:
                ok = ok and np.array_equal(out["models"][w], model)
                wraw, wres = O.boot_model(pw, model, pred, resid, p0[w], p0[w], 1.0, seed_o, woff + w, boff, B, max_iters=mi_b,
                                          shrink_variant=variant, lanes=lb)
                ok = ok and np.array_equal(out["raw"][w], wraw, equal_nan=True) and np.array_equal(out["info_b"]["evals"][w], wres["evals"])
        cases += 1
        if time.time() - t_note > 60:                     # a sign of life for the GPU box's silence guard
            t_note = time.time()
            print(f"... {cases} cases, {fails} mismatches so far", flush=True)
        if not ok:
            fails += 1
            print("MISMATCH", dict(n=n, tmax=tmax, W=W, S=S, B=B, variant=variant, la=la, lb=lb, mi_a=mi_a, strict=strict), flush=True)
    print(f"fuzz_plan: {cases} cases, {fails} mismatches")
    return fails

if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 180, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
