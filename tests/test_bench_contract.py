"""bench.py keeps the driver's contract: one JSON line with the required keys (GPU only)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_bench_json_contract():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--no-stream-probe",
                        "--no-cpu-baseline", "--no-c5-full"], capture_output=True, text=True, cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f64" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    ro = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert j["value"] > 1e5 and j["unit"] == "fits/s"
    assert "traffic_source" in ro and (ro["traffic"] is None or ro["traffic_source"].get("source_sha1"))
    # round 2: the PCIe-inclusive rate, short runs of the other BASELINE configurations and the pairwise scan ride along
    assert j["pcie_inclusive"]["fits_per_s"] > 1e5 and j["pcie_inclusive"]["fits_per_s"] < j["value"] * 1.05
    assert set(j["extra_workloads"]) == {"c2", "c4", "c4s", "mp", "g351", "ref1000_c3", "ref1000_g351"}   # (--no-c5-full)
    mp = j["extra_workloads"]["mp"]                    # the reference's default metaprofile shape, phase split and skipped evaluations
    assert set(mp["kernel_ms"]) == {"fit_starts", "select", "fit_boot"} and "evals_not_executed" in mp
    assert all(w["fits_per_s"] > 1e4 for w in j["extra_workloads"].values())
    # round 3: the reference's default shape (1000 starts + 1000 bootstraps) with its phase split, kernels and stuck fits
    for n in ("ref1000_c3", "ref1000_g351"):
        w = j["extra_workloads"][n]
        assert w["kernel_ms"]["fit_starts"] > 0 and w["kernel_ms"]["fit_boot"] > 0 and "starts" in w["kernels"]
        assert w["starts_at_max_iters"] >= 0 and "evals_not_executed" in w
    # ... and what the reference's summation order costs
    assert j["strict_order"]["price"] > 1.0 and j["strict_order_run"] is False
    assert all(sh["sane"] and sh["achieved_GBps"] > 100 for sh in j["pairwise"]["shapes"])


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts two ranks itself (gloo: they may share the one GPU of a
    test box) and prints exactly one JSON line with n_gpus = 2."""
    import os

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--no-stream-probe", "--no-cpu-baseline", "--no-extras"], capture_output=True,
                       text=True, cwd=str(ROOT), timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(r.stdout.strip().splitlines()) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["fits_per_step"] == 10 + 2 * 10000    # phase A counted once, bootstraps sharded
    # BASELINE C4 as strong scaling: 200 windows in all, 100 per rank, the gathered table checked against the shards
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "c4s",
                        "--steps", "2", "--warmup", "1", "--no-stream-probe", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, cwd=str(ROOT), timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["fits_per_step"] == 200 * 1010
    assert j["config"]["windows_per_gpu"] == 100


@pytest.mark.gpu
@pytest.mark.parametrize("workload,force", [("c3", "1"), ("c4", "2")])
def test_bench_single_process_form(workload, force):
    """`bench.py --single-process --devices 0`: abn_multi_* timed under the same JSON contract, the
    gather forced through RCCL on the one device of a test box (ABN_MULTI_FORCE_RCCL = 1: all-gather, 2: broadcasts)."""
    import os

    env = dict(os.environ, ABN_MULTI_FORCE_RCCL=force)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--single-process", "--devices", "0", "--workload", workload,
                        "--steps", "3", "--warmup", "1"], capture_output=True, text=True, cwd=str(ROOT), timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(r.stdout.strip().splitlines()) == 1     # RCCL's banner must not reach stdout
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["all_rows_finite"] and j["value"] > 1e5
    assert "abn_multi_* (one process)" in j["config"]["parallelism"] and "forced" in j["config"]["parallelism"]
    assert j["fits_per_step"] == (10010 if workload == "c3" else 25 * 1010)
    assert j["roofline"]["kernel_ms"] > 0


@pytest.mark.gpu
def test_bench_c4_strong_scaling_workload():
    """`--workload c4s`: BASELINE C4's 200 windows in all, sharded -> "scaling": "strong" (here: the whole job on one GPU)"""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "c4s", "--steps", "2", "--warmup", "1",
                        "--no-stream-probe", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["scaling"] == "strong" and j["fits_per_step"] == 200 * 1010 and j["config"]["windows_per_gpu"] == 200
