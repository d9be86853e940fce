"""Host-side logic of the product that needs no GPU: deterministic inputs and the bootstrap summary,
checked against the oracle's independent restatement."""
import numpy as np


def test_start_simplices_match_oracle(abn, oracle):
    for seed, window, mx in ((20260101, 0, 0.0123), (7, 5, 0.0), (2**40 + 3, 199, 0.5)):
        got = abn.gen_start_simplices(seed, window, 6, mx)
        want = np.stack([oracle.start_simplex(seed, window, s, mx) for s in range(6)])
        assert np.array_equal(got, want)


def test_boot_simplices_match_oracle(abn, oracle):
    p = np.array([1e-4, 5e-4, 0.03, 1e-3])
    for seed, window, b0 in ((20260101, 0, 0), (99, 3, 1000)):
        got = abn.gen_boot_simplices(seed, window, b0, 7, p)
        want = np.stack([oracle.boot_simplex(seed, window, b0 + i, p) for i in range(7)])
        assert np.array_equal(got, want)
    z = abn.gen_boot_simplices(1, 0, 0, 3, np.array([0.0, -2.0, 1.0, 1.0]))
    assert np.array_equal(z, np.stack([oracle.boot_simplex(1, 0, i, np.array([0.0, -2.0, 1.0, 1.0])) for i in range(3)]))


def test_analyze_matches_oracle(abn, oracle):
    rng = np.random.default_rng(11)
    for b in (10, 200, 1001):
        raw = np.abs(rng.normal(1.0, 0.2, size=(b, 7)))
        assert np.array_equal(abn.analyze(raw), oracle.analyze(raw))
