"""GPU box: evaluation counts of the metaprofile shape's 30 000 start chains (phase A) — how long is the tail that a
hand-over to the speculative kernel could shorten?  usage: python scripts/mp_tail_probe.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
import alphabeta_rs_amd as A
import bench

wl = bench.make_workload("mp", 0, 1)
with A.Context(0) as ctx:
    plan = A.Plan(ctx, wl["gens"], wl["wr"], wl["S"], wl["B"], options=A.default_options(seed=20260101))
    plan.set_windows(wl["D"], wl["p0"])
    plan.run()
    out = plan.download(allow_failed_windows=True)
    print("kernel_ms", plan.kernel_ms(), plan.last_kernels())
    plan.close()
ia = out["info_a"]
ev, st, it = ia["evals"].ravel(), ia["status"].ravel(), ia["iters"].ravel()
conv = st == 0
print("chains", ev.size, "converged", int(conv.sum()), "max_iters (stuck, skipped)", int((st == 1).sum()), "other", int(((st != 0) & (st != 1)).sum()))
e = ev[conv]
print("converged chains: evals mean %.0f, quantiles 50/90/99/99.9/max:" % e.mean(), np.percentile(e, [50, 90, 99, 99.9]).round(), e.max())
tot = e.sum()
for x in (1500, 2000, 2500, 3000, 4000):
    print(f"  chains beyond {x} evaluations: {(e > x).sum()} ; their evaluations beyond it: {np.maximum(e - x, 0).sum() / tot:.3%} of all")
np.save("gpurun_out/mp_start_evals.npy", e)
