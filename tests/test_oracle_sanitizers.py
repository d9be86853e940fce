"""SURVEY.md §5 (race detection / sanitizers): the reference relies on Rust ownership; the checker's counterpart is the
CPU oracle rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile: libabn_oracle_asan.so) and run
over every entry point the parity tests use — all residual-tree modes, both optimiser variants with NaN starts, the
OpenMP fit batches, the bootstrap, selection, analysis and the pairwise divergence.  GPU sanitizers are not available
on the pool; this covers the test infrastructure's own memory safety."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import oracle as O
ped = O.load_pedigree(%r)
sparse = O.load_pedigree(%r)
x = np.array([0.0001179555, 0.0001180614, 0.03693534, 0.003023981])
assert O.cost(ped, 0.75, 0.5, 0.7, x, lanes=1) == 0.0006700888539608879
for lanes in (8, 16, 32, 64, 64 | (3 << 8), 0x10040):
    for table in (True, False):
        c = O.cost(ped, 0.75, 0.5, 0.7, x, lanes=lanes, table=table)
        assert abs(c - 0.0006700888539608879) < 1e-15
rng = np.random.default_rng(0)
s0 = np.stack([np.stack([O.start_simplex(7, 0, s, sparse[:, 3].max())]) for s in range(6)]).reshape(6, 5, 4)
s0[1, 2, 0] = np.nan
for variant in (0, 1):
    for lanes in (1, 16, 0x10040):
        fits = O.fit_batch(sparse, 0.99, 0.99, 1.0, s0, 300, shrink_variant=variant, lanes=lanes, threads=4)
        assert fits["iters"].max() <= 300
k, model, pred, resid, lse = O.select_best(sparse, 0.99, fits["best"])
raw, res = O.boot_model(sparse, model, pred, resid, 0.99, 0.99, 1.0, 7, 3, 5, 24, max_iters=200, lanes=0x10040, threads=4)
assert raw.shape == (24, 7)
an = O.analyze(raw)
status = rng.integers(0, 3, size=(5, 1001), dtype=np.uint8)
pmax = rng.uniform(0.9, 1.0, size=(5, 1001))
d, b, v = O.pairwise_divergence(status, pmax, 0.99)
assert d.shape == (10,)
idx = O.boot_indices(1, 2, 3, 77)
assert idx.max() < 77
print("sanitized oracle ok")
"""


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    r = subprocess.run(["make", "-C", str(ROOT / "oracle"), "-s", "libabn_oracle_asan.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, ABN_ORACLE_LIB=str(ROOT / "oracle" / "libabn_oracle_asan.so"), LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    script = _SCRIPT % (str(ROOT), str(ROOT / "tests" / "golden" / "pedigree.txt"),
                        str(ROOT / "tests" / "golden" / "pedigree_sparse.txt"))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
