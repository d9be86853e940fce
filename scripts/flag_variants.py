#!/usr/bin/env python3
"""Development aid (GPU box): build libabneutral_hip.so with extra compiler flags and run bench.py on each
build (ABNEUTRAL_HIP_LIB points the package at the variant).  Usage:
  python scripts/flag_variants.py [--workload c3] [--bench-args "--lanes 8"] -- "" "-mllvm -amdgpu-sched-strategy=max-ilp" ..."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"
OUT.mkdir(exist_ok=True)
args = sys.argv[1:]
workloads = ["c3"]
if args and args[0] == "--workload":
    workloads = args[1].split(",")
    args = args[2:]
bench_args = []
if args and args[0] == "--bench-args":
    bench_args = args[1].split()
    args = args[2:]
if args and args[0] == "--":
    args = args[1:]
for i, flags in enumerate(args or [""]):
    lib = OUT / f"libabn_flags_{i}.so"
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
           "-shared", "-ldl", *flags.split(), "-o", str(lib), str(ROOT / "alphabeta_rs_amd/csrc/abn_api.hip"),
           str(ROOT / "alphabeta_rs_amd/csrc/abn_pairwise.hip"), str(ROOT / "alphabeta_rs_amd/csrc/abn_multi.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(f"[{flags}] build failed: {r.stderr[-400:]}")
        continue
    for w in workloads:
        env = dict(os.environ, ABNEUTRAL_HIP_LIB=str(lib))
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", w, "--no-stream-probe",
                            "--no-cpu-baseline", *bench_args], capture_output=True, text=True, env=env)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"[{flags}] {w}: {d['value']:.0f} fits/s, {d['ms_per_step']:.3f} ms/step, "
                  + ", ".join(f"{k}={v:.3f}" for k, v in d["kernel_ms"].items()), flush=True)
        except Exception:
            print(f"[{flags}] {w}: failed {r.stderr[-300:]}")
