#!/usr/bin/env python3
"""Soak of the time-sliced persistent kernel (development aid, GPU box): the same multi-window plans run again and again —
which group parks and resumes which chain differs from run to run, the downloaded tables must not, byte for byte; the
first run is checked against the oracle on two windows.  usage: soak_timeslice.py [seconds]"""
import hashlib, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import alphabeta_rs_amd as A
import oracle as O
import bench

def digest(out):
    h = hashlib.sha1()
    for k in ("models", "pred", "resid", "raw", "best_start", "info_a", "info_b"):
        h.update(np.ascontiguousarray(out[k]).tobytes())
    return h.hexdigest()

def main(seconds=120):
    ctx = A.Context(0)
    t_end = time.time() + seconds
    names = ("c3", "c4", "mp")     # c3: one window on 2048 persistent wavefronts (round 3)
    for k, name in enumerate(names):
        wl = bench.make_workload(name, 0, 1)
        seed = 20260101
        plan = A.Plan(ctx, wl["gens"], wl["wr"], wl["S"], wl["B"], options=A.default_options(seed=seed))
        plan.set_windows(wl["D"], wl["p0"])
        plan.run()
        out = plan.download()
        ref = digest(out)
        lanes = int(out["info_b"]["lanes"][0, 0])
        for w in (0, wl["wr"] - 1):                      # oracle check of the first run
            pw = np.concatenate([wl["gens"], wl["D"][w][:, None]], axis=1)
            p0 = float(wl["p0"][w])
            raw, res = O.boot_model(pw, out["models"][w], out["pred"][w], out["resid"][w], p0, p0, 1.0, seed, w, 0,
                                    wl["B"], lanes=lanes)
            assert np.array_equal(out["raw"][w], raw, equal_nan=True), (name, w)
            assert np.array_equal(out["info_b"][w]["evals"], res["evals"]), (name, w)
        runs, t_note = 1, time.time()
        while time.time() < t_end - seconds * (len(names) - 1 - k) / len(names):
            plan.run()
            d = digest(plan.download())
            runs += 1
            if d != ref:
                print(f"MISMATCH {name} run {runs}", flush=True)
                return 1
            if time.time() - t_note > 30:
                t_note = time.time()
                print(f"... {name}: {runs} runs identical", flush=True)
        print(f"soak {name}: {runs} runs, all tables byte-identical, oracle-checked", flush=True)
        plan.close()
    return 0

if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 120))
