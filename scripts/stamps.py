#!/usr/bin/env python3
"""Diagnostic build with in-kernel s_memtime stamps: where one evaluation of a lone chain spends its
cycles.  Builds a separate libabneutral_hip_stamps.so (-DABN_STAMPS); never used by the product."""
import ctypes as C, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
lib = ROOT / "gpurun_out" / "libabneutral_hip_stamps.so"
lib.parent.mkdir(exist_ok=True)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-ldl",
                "-DABN_STAMPS", "-DABN_MEASUREMENT_KNOBS", "-o", str(lib), str(ROOT / "alphabeta_rs_amd/csrc/abn_api.hip"),
                str(ROOT / "alphabeta_rs_amd/csrc/abn_pairwise.hip"), str(ROOT / "alphabeta_rs_amd/csrc/abn_multi.hip")], check=True)
import alphabeta_rs_amd as A
A.LIB_PATH = lib
A._lib = None
L = A.load_library()
L.abn_plan_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
from alphabeta_rs_amd import synthetic
ctx = A.Context(0)
ped, p0 = synthetic.c3_pedigree()
if len(sys.argv) > 1 and sys.argv[1] == "g351":   # the reference's golden pedigree (351 rows, T = 32, K = 10)
    rows = [[float(t) for t in ln.replace("\t", " ").split()] for ln in
            (ROOT / "tests" / "golden" / "pedigree.txt").read_text().splitlines()[1:] if ln.strip()]
    ped, p0 = np.asarray(rows), 0.75
names = ["P1 bcast+genmatrix+puu", "P2 power table", "P3 triples", "P4 rows", "P5 reduce", "-", "P6 NM update", "evals"]
for lanes in (16, 64):
    plan = A.Plan(ctx, ped[:, :3], 1, 1, 0, options=A.default_options(lanes_per_chain=lanes))
    plan.set_windows(ped[:, 3][None, :], np.array([p0]))
    out = (C.c_uint64 * 8)()
    for _ in range(3):
        rc = L.abn_plan_debug_stamps(plan._h, out)
    assert rc == 0
    ev = out[7]
    print(f"lanes={lanes} evals={ev}")
    tot = sum(out[q] for q in range(7))
    for q in range(7):
        if names[q] != "-":
            print(f"  {names[q]:26s} {out[q] / ev:8.1f} cyc/eval  {100.0 * out[q] / tot:5.1f}%")
    print(f"  total {tot / ev:.1f} cyc/eval (stamps add ~40 each)")
    plan.close()

# speculative three-wavefront kernel (phase A): wavefront 0 of chain 0, per ITERATION
L.abn_plan_debug_stamps_spec.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
names = ["P1 bcast+genmatrix+puu", "P2 power table", "P3 triples", "P4+P5 rows, reduce", "exchange (LDS + barrier)",
         "decision, cost insert, termination", "candidate fetch, loop head", "iters"]
plan = A.Plan(ctx, ped[:, :3], 1, 1, 0, options=A.default_options())
plan.set_windows(ped[:, 3][None, :], np.array([p0]))
out = (C.c_uint64 * 8)()
for _ in range(3):
    rc = L.abn_plan_debug_stamps_spec(plan._h, out)
assert rc == 0
it = out[7]
tot = sum(out[q] for q in range(7))
print(f"spec kernel iters={it}")
for q in range(7):
    if names[q] != "-":
        print(f"  {names[q]:26s} {out[q] / it:8.1f} cyc/iter  {100.0 * out[q] / tot:5.1f}%")
print(f"  total {tot / it:.1f} cyc/iter (stamps add ~40 each)")
plan.close()
