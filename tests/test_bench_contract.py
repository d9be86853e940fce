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
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=str(ROOT), timeout=600)
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
    assert set(j["extra_workloads"]) == {"c2", "c4", "g351"}
    assert all(w["fits_per_s"] > 1e4 for w in j["extra_workloads"].values())
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
