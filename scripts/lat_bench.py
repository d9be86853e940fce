#!/usr/bin/env python3
"""Quick latency/throughput probe of the fit kernel (development aid, GPU only).
Prints per-configuration kernel times from HIP events: single-chain latency per evaluation, C3 phases."""
import sys, time, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import alphabeta_rs_amd as A
from alphabeta_rs_amd import synthetic

def main():
    ctx = A.Context(0)
    ped, p0 = synthetic.c3_pedigree()
    out = {}
    for lanes in (16, 32, 64):
        o = A.default_options(lanes_per_chain=lanes)
        plan = A.Plan(ctx, ped[:, :3], 1, 10, 10000, options=o)
        plan.set_windows(ped[:, 3][None, :], np.array([p0]))
        for _ in range(2):
            plan.run()
        ms = []
        for _ in range(5):
            plan.run()
            ms.append(plan.kernel_ms())
        d = plan.download()
        ea = d["info_a"]["evals"].max()
        eb = d["info_b"]["evals"].max()
        a = np.median([m["fit_starts"] for m in ms]); b = np.median([m["fit_boot"] for m in ms])
        out[lanes] = dict(A_ms=round(a, 3), B_ms=round(b, 3), A_us_per_eval=round(1e3 * a / ea, 3),
                          B_tail_us_per_eval=round(1e3 * b / eb, 3), maxevalsA=int(ea), maxevalsB=int(eb),
                          evals_B=int(d["info_b"]["evals"].sum()))
        plan.close()
    # C4 shard (25 windows x 1000 boots) phase B only timing
    gens, D, p0w, _ = synthetic.c4_windows(25)
    plan = A.Plan(ctx, gens, 25, 10, 1000, options=A.default_options())
    plan.set_windows(D, p0w)
    plan.run(); plan.run()
    out["c4"] = {k: round(v, 3) for k, v in plan.kernel_ms().items()}
    plan.close()
    ped5, p5 = synthetic.c5_pedigree()
    plan = A.Plan(ctx, ped5[:, :3], 1, 4, 256, options=A.default_options())
    plan.set_windows(ped5[:, 3][None, :], np.array([p5]))
    plan.run(); plan.run()
    ms = plan.kernel_ms(); cnt = plan.counters()
    d = plan.download()
    evb = int(d["info_b"]["evals"].sum())
    out["c5_256boots"] = dict(ms={k: round(v, 3) for k, v in ms.items()}, evals_b=evb,
                              idx_stream_GBps=round(evb * 20100 * 4 / (ms["fit_boot"] * 1e-3) / 1e9, 1),
                              max_evals_b=int(d["info_b"]["evals"].max()))
    plan.close()
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
