"""Pins the CPU oracle to the reference's own golden vectors.  CPU only."""
import numpy as np
import pytest

from conftest import COST_KNOWN_ANSWER, MODEL_DEFAULT


def test_same_as_r(oracle, golden):
    # src/divergence.rs:138-161: reference tolerance 1e-4 (src/macros.rs:15); the oracle reaches 1e-15
    for table in (False, True):
        dt, _ = oracle.divergence(golden["pedigree"], 0.25, 0.75, 3.974271e-09, 1.519045e-07, 0.06892953, table=table)
        r = golden["divergence"]
        assert dt.shape == r.shape
        assert np.all(np.isfinite(dt)) and np.all(dt != 0)
        assert np.max(np.abs(dt - r)) < 1e-15
        assert np.max(np.abs(dt - r) / r) < 1e-13


def test_cost_known_answer_bit_exact(oracle, golden):
    # src/structs.rs:225-240: assert_eq!(result, 0.0006700888539608879)
    for table in (False, True):
        c = oracle.cost(golden["pedigree"], 0.75, 0.5, 0.7, MODEL_DEFAULT, lanes=1, table=table)
        assert c == COST_KNOWN_ANSWER


def test_table_variant_is_bit_identical(oracle, golden):
    rng = np.random.default_rng(7)
    for ped, p0 in ((golden["pedigree"], 0.75), (golden["sparse"], golden["r_p0uu"]), (golden["generated"], 0.655)):
        for _ in range(20):
            a, b = 10 ** rng.uniform(-9, -2, 2)
            w = rng.uniform(0, 0.1)
            d0, p0_ = oracle.divergence(ped, 1 - p0, p0, a, b, w, table=False)
            d1, p1_ = oracle.divergence(ped, 1 - p0, p0, a, b, w, table=True)
            assert np.array_equal(d0, d1) and p0_ == p1_


def test_matrix_power_identity_and_rows(oracle):
    # src/divergence.rs:129-137
    m = np.arange(1.0, 10.0).reshape(3, 3)
    assert np.array_equal(oracle.matrix_power(m, 0), np.eye(3))
    # src/divergence.rs:163-209: e_r . G^k == row r of G^k, exact
    g = oracle.genmatrix(0.2, 0.5)
    for k in (1, 2):
        gk = oracle.matrix_power(g, k)
        for r in range(3):
            e = np.zeros(3)
            e[r] = 1.0
            dot = np.array([np.sum(e * gk[:, j]) for j in range(3)])
            assert np.array_equal(dot, gk[r])
    with pytest.raises(ValueError):
        oracle.matrix_power(g, -1)


def test_genmatrix_rows_are_stochastic(oracle):
    g = oracle.genmatrix(1e-4, 5e-4)
    assert np.allclose(g.sum(axis=1), 1.0, atol=1e-15)


def test_generated_pedigree_values(oracle, golden):
    # SURVEY.md §8c item 6 (scratch-derived; regenerated here with the oracle)
    pg, p0 = golden["generated"], golden["p0uu_generated"]
    dt, puu = oracle.divergence(pg, 1 - p0, p0, *MODEL_DEFAULT[:3])
    want = [0.003298421189504402, 0.006429850106554763, 0.006429850106554763, 0.003236883760296371,
            0.006746081022553649, 0.007269794893779856]
    assert dt.tolist() == want
    assert puu == 0.4999884706697511
    assert oracle.cost(pg, p0, p0, 1.0, MODEL_DEFAULT) == 1.3512094573699542


def test_r_optimum_anchor(oracle, golden):
    # data/desired_output/ABneutral_estimatats_...txt line 2: objective 5.26475086020599e-05
    xr = [5.7985750419976e-05, 0.00655710970515347, 0.0306958517646129, 5.96083073236131e-05]
    l = oracle.lse(golden["sparse"], golden["r_p0uu"], xr)
    assert abs(l - 5.26475086020599e-05) < 1e-12


def test_fit_lands_in_r_spread(oracle, golden):
    # R's ten best Nelder-Mead runs: alpha 5.798e-05..5.801e-05, beta 6.556e-03..6.559e-03
    ps, p0 = golden["sparse"], golden["r_p0uu"]
    s0 = np.stack([oracle.start_simplex(20260101, 0, s, ps[:, 3].max()) for s in range(16)])
    res = oracle.fit_batch(ps, p0, p0, 1.0, s0, 10000)
    k, model, pred, resid, lse = oracle.select_best(ps, p0, res["best"])
    assert k >= 0
    assert 5.79e-05 < model[0] < 5.81e-05
    assert 6.55e-03 < model[1] < 6.57e-03
    assert lse.min() <= 5.2648e-05
    assert np.array_equal(pred + resid, ps[:, 3]) or np.allclose(pred + resid, ps[:, 3], rtol=0, atol=1e-18)


def test_lane_tree_orders_agree_to_rounding(oracle, golden):
    ped = golden["pedigree"]
    ref = oracle.cost(ped, 0.75, 0.5, 0.7, MODEL_DEFAULT, lanes=1)
    for lanes in (8, 16, 32, 64):
        c = oracle.cost(ped, 0.75, 0.5, 0.7, MODEL_DEFAULT, lanes=lanes)
        assert abs(c - ref) <= 4 * np.finfo(float).eps * ref


def test_sd_termination_equals_threshold_form(oracle):
    # sqrt(v) < EPSILON  <=>  v < 2^-104 for correctly rounded sqrt (used to reason about the kernel)
    eps = np.finfo(float).eps
    for v in (2.0 ** -104, np.nextafter(2.0 ** -104, 0), np.nextafter(2.0 ** -104, 1), 0.0, 1e-300):
        assert (np.sqrt(v) < eps) == (v < 2.0 ** -104)


def test_as_i8_and_bad_pedigrees(oracle):
    L = oracle.lib()
    assert L.abo_as_i8(3.9) == 3 and L.abo_as_i8(-0.5) == 0 and L.abo_as_i8(1e9) == 127
    assert L.abo_as_i8(float("nan")) == 0 and L.abo_as_i8(-1e9) == -128
    bad = np.array([[2.0, 1.0, 3.0, 0.1]])
    with pytest.raises(ValueError):
        oracle.divergence(bad, 0.25, 0.75, 1e-4, 1e-4, 0.03)


def test_philox_known_answers(oracle):
    # Random123 kat_vectors: philox4x32-10
    assert oracle.philox((0, 0, 0, 0), (0, 0)).tolist() == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF)).tolist() == [
        0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)).tolist() == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_bootstrap_indices_and_simplices(oracle):
    idx = oracle.boot_indices(1234, 3, 17, 105)
    assert idx.dtype == np.uint32 and idx.max() < 105
    big = np.concatenate([oracle.boot_indices(1234, 0, b, 64) for b in range(400)])
    counts = np.bincount(big, minlength=64)
    assert counts.min() > 300 and counts.max() < 500  # uniform, mean 400
    s = oracle.start_simplex(1234, 0, 5, 0.02)
    assert np.all((s[:, :2] >= 1e-9) & (s[:, :2] <= 1e-2)) and np.all((s[:, 2] >= 0) & (s[:, 2] < 0.1))
    assert np.all((s[:, 3] >= 0) & (s[:, 3] < 0.02))
    p = np.array([1e-4, 5e-4, 0.03, -1e-3])
    b = oracle.boot_simplex(1234, 0, 9, p)
    assert np.array_equal(b[0], p)
    assert np.all(np.abs(b[1:] - p) <= 0.1 * np.abs(p) * (1 + 1e-12))
    z = oracle.boot_simplex(1234, 0, 9, np.array([0.0, 1.0, 1.0, 1.0]))
    assert np.all((z[1:, 0] >= 0.09) & (z[1:, 0] <= 0.11))  # zero is treated as 0.1, src/structs.rs:105-108


def test_analysis_against_numpy(oracle):
    rng = np.random.default_rng(3)
    raw = np.abs(rng.normal(1.0, 0.1, size=(257, 7)))
    out = oracle.analyze(raw)
    cols = [raw[:, 0], raw[:, 1], raw[:, 1] / raw[:, 0], raw[:, 2], raw[:, 3], raw[:, 4], raw[:, 5], raw[:, 6]]
    for k, col in enumerate(cols):
        assert np.isclose(out[0, k], col.mean(), rtol=1e-14)
        assert np.isclose(out[1, k], col.std(ddof=1), rtol=1e-12)
        assert np.isclose(out[2, k], np.quantile(col, 0.025), rtol=1e-13)
        assert np.isclose(out[3, k], np.quantile(col, 0.975), rtol=1e-13)
