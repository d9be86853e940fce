#!/usr/bin/env python3
"""Scheduling model of a packed phase-B launch on the measured branch traces of every chain (oracle:
abo_boot_model_trace).  A wavefront holds 4 chains (16 lanes each) and steps once per evaluation; a SIMD that hosts k
wavefronts runs a step of each in t(k) microseconds (measured: 1.87 / 2.6 / 3.0 for k = 1 / 2 / 3).

  plain        every chain one evaluation per step; the wavefront lives as long as its longest chain
  speculate    lane groups whose chain has finished evaluate the contraction (first) and expansion (second) point of a
               running chain of the same wavefront in the same step as its reflection: an iteration whose second
               evaluation was speculated costs one step instead of two

usage: sched_sim.py traces.npz   (tr[nb, max_iters] u8, iters[nb]: written by tests/fuzz/gen_branch_traces.py)
"""
import sys

import numpy as np

T = {0: 0.0, 1: 1.87, 2: 2.6, 3: 3.0}


def chain_steps(kinds):
    """evaluation sequence of a chain: 'i' x5, then per iteration 'r' (+ 'e' or 'c')"""
    seq = ["i"] * 5
    for k in kinds:
        seq.append("r")
        if k == 1:
            seq.append("e")
        elif k in (2, 3):
            seq.append("c")
    return seq


def wave_trips(chains, speculate, order=("c", "e")):
    """number of steps a wavefront of these chains needs"""
    pos = [0] * len(chains)
    trips = 0
    while True:
        active = [i for i, c in enumerate(chains) if pos[i] < len(c)]
        if not active:
            return trips
        trips += 1
        idle = 4 - len(active)
        helped = {}
        if speculate and idle:
            at_r = [i for i in active if chains[i][pos[i]] == "r"]
            for kind in order:                     # every reflecting chain gets its first helper before any gets two
                for i in at_r:
                    if idle:
                        helped.setdefault(i, set()).add(kind)
                        idle -= 1
        for i in active:
            c = chains[i]
            if c[pos[i]] == "r" and pos[i] + 1 < len(c) and c[pos[i] + 1] in helped.get(i, ()):
                pos[i] += 2
            else:
                pos[i] += 1


def launch_time(trips, n_simd=1024):
    """waves dealt round-robin to the SIMDs, all resident from the start; processor sharing per SIMD"""
    worst = 0.0
    busy = 0.0
    for s in range(n_simd):
        tr = sorted(trips[s::n_simd])
        t, done = 0.0, 0
        k = len(tr)
        for j, x in enumerate(tr):
            t += (x - done) * T[min(k - j, 3)]
            done = x
        worst = max(worst, t)
    return worst


def main():
    z = np.load(sys.argv[1])
    tr, iters = z["tr"], z["iters"]
    chains = [chain_steps(tr[i, : iters[i]]) for i in range(len(iters))]
    for name, spec, order in (("plain", False, None), ("speculate c,e", True, ("c", "e")), ("speculate e,c", True, ("e", "c"))):
        trips = [wave_trips(chains[b : b + 4], spec, order or ("c", "e")) for b in range(0, len(chains), 4)]
        print(f"{name:16s} wave-steps {sum(trips):8d}  longest {max(trips):4d}  launch {launch_time(trips) / 1e3:.3f} ms")


if __name__ == "__main__":
    main()


def spread(argv):
    """the same chains over more wavefronts than they fill (some hold three chains and a helper group from the start)"""
    z = np.load(argv[1])
    tr, iters = z["tr"], z["iters"]
    chains = [chain_steps(tr[i, : iters[i]]) for i in range(len(iters))]
    for W in (2500, 2560, 2816, 3072):
        for order in (("c", "e"), ("e", "c")):
            trips = [wave_trips(chains[w::W], True, order) for w in range(W)]
            print(f"W={W} order={order} wave-steps {sum(trips):8d} longest {max(trips):4d} launch {launch_time(trips) / 1e3:.3f} ms")
