"""A bounded slice of the randomised GPU-vs-oracle sweeps under `-m gpu`: the long runs of
tests/fuzz/fuzz_parity.py and fuzz_plan.py (thousands of cases per round) are builder-run; this puts a
fixed-seed sample of the same generators where the driver's GPU tier sees it.  Bit-exact against the oracle: fit batches
(random pedigrees of 1 ... 700 rows, every lane packing, both optimiser variants, serial / auto / tree order, wild and NaN
starts) and whole plans (speculative, one-wavefront, packed, persistent and stream kernels, selection, window / bootstrap
offsets).  Bounded by cases and by time (about a minute together)."""
import importlib.util
from pathlib import Path

import pytest

FUZZ = Path(__file__).resolve().parent / "fuzz"


def _load(name):
    spec = importlib.util.spec_from_file_location(name, FUZZ / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
@pytest.mark.parametrize("seed", (20260401, 20260402))
def test_fuzz_parity_slice(abn, gpu_ctx, oracle, seed, capsys):
    fails = _load("fuzz_parity").main(seconds=40, seed=seed, max_cases=100)
    out = capsys.readouterr().out
    assert fails == 0, out
    assert "fuzz:" in out and " 0 mismatches" in out
    assert int(out.split("fuzz:")[1].split()[0]) >= 30, out     # it did run a meaningful number of cases in its time box


@pytest.mark.gpu
def test_fuzz_plan_slice(abn, gpu_ctx, oracle, capsys):
    fails = _load("fuzz_plan").main(seconds=60, seed=20260403, max_cases=50)
    out = capsys.readouterr().out
    assert fails == 0, out
    assert "fuzz_plan:" in out and " 0 mismatches" in out
    assert int(out.split("fuzz_plan:")[1].split()[0]) >= 10, out
