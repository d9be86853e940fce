"""DMatrix::from (src/pedigree.rs:210-261), SURVEY.md §8f row 1: oracle against the reference's bundled
methylomes (data/methylome/*.txt -> the D column of data/pedigree_generated.txt), and the HIP kernel
(abn_pairwise_divergence) against the oracle, bit-exact (integer sums, one f64 division)."""
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"
STATUS = {"U": 0, "I": 1, "M": 2}


def load_methylome(path):
    """the CG rows of a methylome file as (status_numeric, posteriormax): first/second format of
    src/methylation_site.rs:146-232"""
    st, pm = [], []
    for ln in path.read_text().splitlines():
        f = ln.split("\t")
        if len(f) in (9, 10) and f[3] == "CG":
            st.append(STATUS.get(f[7][0], 0))
            pm.append(float(f[6]))
    return np.array(st, dtype=np.uint8), np.array(pm)


def bundled_samples():
    files = ["G0.txt", "G1_2.txt", "G4_2.txt", "G4_8.txt"]          # the "Y" rows of data/nodelist.txt, in order
    cols = [load_methylome(GOLDEN / "data" / "methylome" / f) for f in files]
    return np.stack([c[0] for c in cols]), np.stack([c[1] for c in cols])


def test_oracle_pairwise_reproduces_generated_pedigree(oracle, golden):
    status, pmax = bundled_samples()
    assert status.shape == (4, 500)
    diff, both, dval = oracle.pairwise_divergence(status, pmax, 0.99)
    assert np.array_equal(dval, golden["generated"][:, 3])      # D.value column, pair order of src/pedigree.rs:273-274
    assert np.all(both > 0) and np.all(diff <= 2 * both)


def _codes(status, pmax, flt):
    return (status | np.where(pmax < flt, 0x80, 0)).astype(np.uint8)


@pytest.mark.gpu
def test_gpu_pairwise_bundled_methylomes(abn, gpu_ctx, oracle, golden):
    status, pmax = bundled_samples()
    diff, both, dval = gpu_ctx.pairwise_divergence(_codes(status, pmax, 0.99))
    wd, wb, wv = oracle.pairwise_divergence(status, pmax, 0.99)
    assert np.array_equal(diff, wd) and np.array_equal(both, wb) and np.array_equal(dval, wv)
    assert np.array_equal(dval, golden["generated"][:, 3])


@pytest.mark.gpu
@pytest.mark.parametrize("n,L", [(2, 1), (2, 5), (3, 2047), (15, 2048), (15, 2049), (7, 100003), (70, 4100),
                                 (130, 1500), (16, 64), (17, 129), (33, 4097), (48, 1000), (50, 20011), (64, 5000),
                                 (65, 700), (100, 3001), (129, 1031), (200, 515)])
def test_gpu_pairwise_random(abn, gpu_ctx, oracle, n, L):
    rng = np.random.default_rng(n * 1000 + L)
    status = rng.integers(0, 3, size=(n, L), dtype=np.uint8)
    pmax = rng.uniform(0.9, 1.0, size=(n, L))
    pmax[0, : L // 2] = 0.5                      # a sample with a long filtered stretch
    if n > 2:
        pmax[2] = 0.1                            # a sample with no valid site at all -> 0/0 = NaN like the reference
    diff, both, dval = gpu_ctx.pairwise_divergence(_codes(status, pmax, 0.99))
    wd, wb, wv = oracle.pairwise_divergence(status, pmax, 0.99)
    assert np.array_equal(diff, wd) and np.array_equal(both, wb)
    assert np.array_equal(dval, wv, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("n,L", [(1100, 700), (2048, 300), (4096, 130)])
def test_gpu_pairwise_many_samples(abn, gpu_ctx, oracle, n, L):
    """The sample axis is tiled in groups of 64 (DMatrix::from has no limit on the number of samples, src/pedigree.rs:210-261;
    rounds 2-3 refused more than one LDS tile of ~2700 samples): 4096 samples are 64 groups, 2080 pairs of groups."""
    rng = np.random.default_rng(n + L)
    status = rng.integers(0, 3, size=(n, L), dtype=np.uint8)
    pmax = rng.uniform(0.95, 1.0, size=(n, L))
    diff, both, dval = gpu_ctx.pairwise_divergence(_codes(status, pmax, 0.99))
    wd, wb, wv = oracle.pairwise_divergence(status, pmax, 0.99)
    assert np.array_equal(diff, wd) and np.array_equal(both, wb) and np.array_equal(dval, wv, equal_nan=True)


@pytest.mark.gpu
def test_gpu_pairwise_more_jobs_than_one_launch_holds(abn, gpu_ctx, oracle):
    """8200 samples are 129 groups: 8256 pairs of groups, more than the 8192 jobs whose packed sums one launch keeps — the
    second family of launches runs in two slabs that reuse the partial rows (stream order).  33.6 M pairs, bit-exact."""
    n, L = 8200, 40
    rng = np.random.default_rng(99)
    status = rng.integers(0, 3, size=(n, L), dtype=np.uint8)
    pmax = rng.uniform(0.9, 1.0, size=(n, L))
    diff, both, dval = gpu_ctx.pairwise_divergence(_codes(status, pmax, 0.99))
    wd, wb, wv = oracle.pairwise_divergence(status, pmax, 0.99)
    assert np.array_equal(diff, wd) and np.array_equal(both, wb) and np.array_equal(dval, wv, equal_nan=True)


@pytest.mark.gpu
def test_gpu_pairwise_every_state_pair_and_operand_order(abn, gpu_ctx, oracle):
    """Exact-integer check of the matrix-instruction form: every (state_a, state_b) combination of {U, I, M, filtered U/I/M}
    in known counts, on samples in DIFFERENT 16-sample blocks and groups (an asymmetric layout: a transposed tile or a
    swapped table half would show)."""
    n, reps = 150, 37
    states = np.array([0, 1, 2, 0x80, 0x81, 0x82], dtype=np.uint8)
    rng = np.random.default_rng(7)
    codes = states[rng.integers(0, 6, size=(n, 36 * reps))]
    a, b = 3, 141                                       # samples of different groups: all 36 combinations, (k+1) times each
    cols = []
    for ia, sa in enumerate(states):
        for ib, sb in enumerate(states):
            cols += [(sa, sb)] * ((6 * ia + ib) % 5 + 1)
    cols = np.array(cols, dtype=np.uint8)
    codes[a, : len(cols)] = cols[:, 0]
    codes[b, : len(cols)] = cols[:, 1]
    diff, both, dval = gpu_ctx.pairwise_divergence(codes)
    status, pmax = codes & 3, np.where(codes & 0x80, 0.5, 1.0)
    wd, wb, wv = oracle.pairwise_divergence(status, pmax, 0.99)
    assert np.array_equal(diff, wd) and np.array_equal(both, wb) and np.array_equal(dval, wv, equal_nan=True)


@pytest.mark.gpu
def test_gpu_pairwise_degenerate(abn, gpu_ctx):
    d, b, v = gpu_ctx.pairwise_divergence(np.zeros((1, 10), dtype=np.uint8))   # one sample: no pairs
    assert d.size == 0
    d, b, v = gpu_ctx.pairwise_divergence(np.zeros((3, 0), dtype=np.uint8))    # no sites: 0/0
    assert np.all(b == 0) and np.all(np.isnan(v))


@pytest.mark.gpu
@pytest.mark.parametrize("n,L", [(15, 300_001), (50, 65_536), (9, 1_000_003), (95, 9_000)])
def test_gpu_pairwise_device_resident_entry(abn, gpu_ctx, oracle, n, L):
    """abn_pairwise_divergence_dev (codes and results stay in HBM) against the host entry and the oracle: persistent
    counters over many tiles, byte-misaligned rows (odd L), word slices per pair block (n = 9, 15), more than
    kPairThreads * kPairUnits pair blocks (n = 95: per-tile sums)."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")        # the HIP runtime the product library already holds (no second runtime: a
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]   # torch imported AFTER it would bring its own copy)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    rng = np.random.default_rng(n + L)
    status = rng.integers(0, 3, size=(n, L), dtype=np.uint8)
    pmax = rng.uniform(0.9, 1.0, size=(n, L))
    codes = _codes(status, pmax, 0.99)
    npairs = n * (n - 1) // 2
    bufs = [C.c_void_p() for _ in range(4)]
    for ptr, size in zip(bufs, (codes.nbytes + 16, 8 * npairs, 8 * npairs, 8 * npairs)):
        assert hip.hipMalloc(C.byref(ptr), size) == 0
    t0, d, b, v = bufs
    # the codes start at an ODD device address for the even site counts (the caller's buffer need not be aligned: the
    # kernel then takes its funnel-shift loader although L is a multiple of four)
    t = C.c_void_p(t0.value + (1 if L % 2 == 0 else 0))
    assert hip.hipMemcpy(t, codes.ctypes.data, codes.nbytes, 1) == 0
    ms = gpu_ctx.pairwise_divergence_dev(t.value, n, L, d.value, b.value, v.value)
    assert ms > 0
    dd, db, dv = np.zeros(npairs, np.uint64), np.zeros(npairs, np.uint64), np.zeros(npairs)
    for host, dev in ((dd, d), (db, b), (dv, v)):
        assert hip.hipMemcpy(host.ctypes.data, dev, 8 * npairs, 2) == 0
    for ptr in (t0, d, b, v):
        hip.hipFree(ptr)
    hd, hb, hv = gpu_ctx.pairwise_divergence(codes)
    assert np.array_equal(dd, hd) and np.array_equal(db, hb) and np.array_equal(dv, hv, equal_nan=True)
    Ls = min(L, 40_000)                                   # the oracle on a prefix it finishes quickly
    wd, wb, wv = oracle.pairwise_divergence(status[:, :Ls], pmax[:, :Ls], 0.99)
    gd, gb, gv = gpu_ctx.pairwise_divergence(np.ascontiguousarray(codes[:, :Ls]))
    assert np.array_equal(gd, wd) and np.array_equal(gb, wb) and np.array_equal(gv, wv)
    # linearity over the site axis: the whole equals the sum of two halves (size-independent property)
    h = L // 2
    d1, b1, _ = gpu_ctx.pairwise_divergence(np.ascontiguousarray(codes[:, :h]))
    d2, b2, _ = gpu_ctx.pairwise_divergence(np.ascontiguousarray(codes[:, h:]))
    assert np.array_equal(d1 + d2, hd) and np.array_equal(b1 + b2, hb)
