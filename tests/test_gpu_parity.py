"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the reference's golden vectors.

Floating-point bar: the north star asks for 1e-6 on (alpha, beta, weight, predicted divergence).  The
tests hold the kernel to a much tighter bar — BIT-EXACT equality with the oracle on identical inputs
(same start simplices, same bootstrap indices, same reduction tree) — because Nelder-Mead is chaotic
under 1-ulp cost differences; wherever a tolerance is used instead it is written in the test.
"""
import numpy as np
import pytest

from conftest import COST_KNOWN_ANSWER, MODEL_DEFAULT

pytestmark = pytest.mark.gpu

LANES = (8, 16, 32, 64)


def synthetic_pedigree(rng, n, tmax, frac_t0=0.3):
    """random valid (t0,t1,t2,D) rows: t0 <= t1,t2 <= tmax"""
    t0 = np.where(rng.random(n) < frac_t0, rng.integers(0, max(1, tmax // 2), n), 0)
    t1 = t0 + rng.integers(0, tmax - t0 + 1)
    t2 = t0 + rng.integers(0, tmax - t0 + 1)
    d = np.abs(rng.normal(0.01, 0.004, n))
    return np.stack([t0, t1, t2, d], axis=1).astype(np.float64)


# ------------------------------------------------------------------------------------------------ cost
def test_cost_known_answer_strict_bit_exact(abn, gpu_ctx, golden):
    """src/structs.rs:225-240 on the GPU: serial row order -> 0.0006700888539608879 exactly."""
    o = abn.default_options(strict_order=1)
    c = gpu_ctx.cost_batch(golden["pedigree"], 0.75, 0.5, 0.7, MODEL_DEFAULT[None, :], options=o)
    assert c[0] == COST_KNOWN_ANSWER


def test_divergence_same_as_r(abn, gpu_ctx, golden, oracle):
    """src/divergence.rs:138-161 on the GPU, and bit-equality of dt1t2 / p_uu with the oracle."""
    x = np.array([[3.974271e-09, 1.519045e-07, 0.06892953, 0.0]])
    cost, dt, puu = gpu_ctx.cost_batch(golden["pedigree"], 0.75, 0.5, 0.7, x, want_dt=True, want_puu=True)
    r = golden["divergence"]
    assert np.max(np.abs(dt[0] - r)) < 1e-15          # reference tolerance: 1e-4 (src/macros.rs:15)
    want_dt, want_puu = oracle.divergence(golden["pedigree"], 0.25, 0.75, *x[0, :3])
    assert np.array_equal(dt[0], want_dt)
    assert puu[0] == want_puu


@pytest.mark.parametrize("lanes", LANES)
def test_cost_batch_matches_oracle_tree_order(abn, gpu_ctx, golden, oracle, lanes):
    rng = np.random.default_rng(100 + lanes)
    for ped, p0 in ((golden["pedigree"], 0.75), (golden["sparse"], golden["r_p0uu"]),
                    (golden["generated"], golden["p0uu_generated"]), (synthetic_pedigree(rng, 517, 40), 0.8)):
        m = 33
        cand = np.stack([10 ** rng.uniform(-9, -2, m), 10 ** rng.uniform(-9, -2, m), rng.uniform(0, 0.1, m),
                         rng.uniform(0, ped[:, 3].max(), m)], axis=1)
        cand[0] = [-1e-5, 2e-5, 0.5, 0.0]       # negative rate: still finite arithmetic
        o = abn.default_options(lanes_per_chain=lanes)
        got, dt, puu = gpu_ctx.cost_batch(ped, p0, 0.5, 0.7, cand, options=o, want_dt=True, want_puu=True)
        want = np.array([oracle.cost(ped, p0, 0.5, 0.7, x, lanes=lanes) for x in cand])
        assert np.array_equal(got, want)
        strict = gpu_ctx.cost_batch(ped, p0, 0.5, 0.7, cand, options=abn.default_options(strict_order=1))
        want_strict = np.array([oracle.cost(ped, p0, 0.5, 0.7, x, lanes=1, table=False) for x in cand])
        assert np.array_equal(strict, want_strict)
        for k in (0, 1, m - 1):
            wdt, wp = oracle.divergence(ped, 1 - p0, p0, *cand[k, :3])
            assert np.array_equal(dt[k], wdt) and puu[k] == wp


def test_cost_batch_nonfinite_candidates(abn, gpu_ctx, golden, oracle):
    ped = golden["sparse"]
    cand = np.array([[0.0, 0.0, 0.03, 0.0],          # alpha+beta = 0 -> 0/0 in p_uu_est
                     [np.inf, 1e-4, 0.03, 0.0], [np.nan, 1e-4, 0.03, 0.0], [1e-4, 1e-4, np.inf, 0.0]])
    got = gpu_ctx.cost_batch(ped, 0.9, 0.9, 1.0, cand, options=abn.default_options(lanes_per_chain=16))
    want = np.array([oracle.cost(ped, 0.9, 0.9, 1.0, x, lanes=16) for x in cand])
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])


def test_cost_batch_bootstrap_observations(abn, gpu_ctx, golden, oracle):
    ped, p0 = golden["sparse"], golden["r_p0uu"]
    n = ped.shape[0]
    rng = np.random.default_rng(5)
    pred, resid = rng.normal(0.005, 0.001, n), rng.normal(0, 1e-3, n)
    idx = np.stack([oracle.boot_indices(42, 0, b, n) for b in range(6)])
    cand = np.tile(np.array([[5.8e-05, 6.5e-03, 0.03, 6e-05]]), (9, 1)) * rng.uniform(0.9, 1.1, (9, 4))
    c2b = np.array([0, 1, 2, 3, 4, 5, 5, 0, 3], dtype=np.uint32)
    got = gpu_ctx.cost_batch(ped, p0, p0, 1.0, cand, pred=pred, resid=resid, idx=idx, cand_to_boot=c2b,
                             options=abn.default_options(lanes_per_chain=16))
    want = np.array([oracle.cost(ped, p0, p0, 1.0, cand[k], dobs=pred + resid[idx[c2b[k]]], lanes=16)
                     for k in range(9)])
    assert np.array_equal(got, want)


def test_bad_pedigree_is_rejected(abn, gpu_ctx):
    for rows in ([[2.0, 1.0, 3.0, 0.1]], [[-1.0, 1.0, 3.0, 0.1]]):
        with pytest.raises(abn.AbnError) as e:
            gpu_ctx.cost_batch(np.array(rows), 0.75, 0.5, 0.7, MODEL_DEFAULT[None, :])
        assert e.value.status == 2
    # generations saturate at 127 like Rust's `as i8` (src/divergence.rs:52)
    sat = np.array([[0.0, 500.0, 127.0, 0.1]])
    a = gpu_ctx.cost_batch(sat, 0.75, 0.5, 0.7, MODEL_DEFAULT[None, :])
    b = gpu_ctx.cost_batch(np.array([[0.0, 127.0, 127.0, 0.1]]), 0.75, 0.5, 0.7, MODEL_DEFAULT[None, :])
    assert a[0] == b[0]


# ------------------------------------------------------------------------------------------------ inputs
def test_bootstrap_indices_bit_exact(abn, gpu_ctx, oracle):
    for n in (6, 78, 105, 351, 1023):
        got = gpu_ctx.gen_boot_indices(20260101, 7, 1000, 5, n)
        want = np.stack([oracle.boot_indices(20260101, 7, 1000 + b, n) for b in range(5)])
        assert got.dtype == np.uint32 and np.array_equal(got, want)


# ------------------------------------------------------------------------------------------------ fits
def _assert_fits_equal(best, info, want):
    assert np.array_equal(info["status"], want["status"])
    assert np.array_equal(info["iters"], want["iters"])
    assert np.array_equal(info["evals"], want["evals"])
    ok = want["status"] != 2
    assert np.array_equal(best[ok], want["best"][ok])
    assert np.array_equal(info["best_cost"][ok], want["best_cost"][ok])


@pytest.mark.parametrize("lanes", LANES)
@pytest.mark.parametrize("variant", (0, 1))
def test_fit_batch_trajectories_bit_exact(abn, gpu_ctx, golden, oracle, lanes, variant):
    """Identical start simplices -> identical Nelder-Mead trajectory (iterations, evaluations, best
    parameters and cost all bit-equal to the oracle run with the same reduction tree)."""
    cases = (("sparse", golden["sparse"], golden["r_p0uu"], 24, 3000),
             ("generated", golden["generated"], golden["p0uu_generated"], 24, 1500),
             ("golden351", golden["pedigree"], 0.75, 8, 400))
    for name, ped, p0, f, iters in cases:
        s0 = abn.gen_start_simplices(11 + lanes, 0, f, ped[:, 3].max())
        o = abn.default_options(lanes_per_chain=lanes, shrink_on_failed_contraction=variant)
        best, info = gpu_ctx.fit_batch(ped, p0, p0, 1.0, s0, iters, options=o)
        code = int(info["lanes"][0])   # lanes, plus the row-block code when the pedigree is streamed
        assert np.all(info["lanes"] == code) and (code & 0xff) == lanes
        want = oracle.fit_batch(ped, p0, p0, 1.0, s0, iters, shrink_variant=variant, lanes=code)
        _assert_fits_equal(best, info, want)


def test_fit_batch_stream_mode_large_pedigree(abn, gpu_ctx, oracle):
    """More rows than the LDS-resident variants hold (8 per lane; 16 per lane with one wavefront per chain):
    the kernel re-streams rows every evaluation (stream mode).  N = 700 is streamed with 16 lanes per chain
    and resident (16 rows per lane) with 64; N = 1100 is streamed either way."""
    rng = np.random.default_rng(77)
    true = np.array([1e-4, 5e-4, 0.03, 1e-3])
    for n, expect in ((700, {16: 3, 64: 0}), (1100, {16: 3, 64: 3})):
        ped = synthetic_pedigree(rng, n, 12)
        dt, _ = oracle.divergence(ped, 0.25, 0.75, *true[:3], table=True)
        ped[:, 3] = np.maximum(true[3] + dt + rng.normal(0, 2e-4, n), 0)
        s0 = abn.gen_start_simplices(5, 0, 6, ped[:, 3].max())
        for lanes in (16, 64):
            o = abn.default_options(lanes_per_chain=lanes)
            best, info = gpu_ctx.fit_batch(ped, 0.75, 0.75, 1.0, s0, 300, options=o)
            code = int(info["lanes"][0])          # stream mode: lanes | (rows per block - 1) << 8
            assert code & 0xff == lanes and code >> 8 == expect[lanes]
            want = oracle.fit_batch(ped, 0.75, 0.75, 1.0, s0, 300, lanes=code)
            _assert_fits_equal(best, info, want)


def test_fit_batch_per_fit_observations(abn, gpu_ctx, golden, oracle):
    ped, p0 = golden["sparse"], golden["r_p0uu"]
    n = ped.shape[0]
    rng = np.random.default_rng(9)
    f = 10
    dobs = np.abs(ped[:, 3][None, :] + rng.normal(0, 2e-4, (f, n)))
    s0 = abn.gen_boot_simplices(3, 0, 0, f, np.array([5.8e-05, 6.5e-03, 0.03, 6e-05]))
    best, info = gpu_ctx.fit_batch(ped, p0, p0, 1.0, s0, 1000, dobs_rows=dobs,
                                   options=abn.default_options(lanes_per_chain=16))
    want = oracle.fit_batch(ped, p0, p0, 1.0, s0, 1000, dobs_rows=dobs, lanes=16)
    _assert_fits_equal(best, info, want)


def test_fit_nonfinite_start_reports_status(abn, gpu_ctx, golden, oracle):
    ped, p0 = golden["generated"], golden["p0uu_generated"]
    s0 = abn.gen_start_simplices(1, 0, 3, ped[:, 3].max())
    s0[1, :, 0] = np.nan                        # every vertex NaN -> never a finite best
    s0[2, 2, 1] = np.nan                        # one NaN vertex
    o = abn.default_options(lanes_per_chain=8)
    best, info = gpu_ctx.fit_batch(ped, p0, p0, 1.0, s0, 200, options=o)
    want = oracle.fit_batch(ped, p0, p0, 1.0, s0, 200, lanes=8)
    assert info["status"][1] == abn.FIT_NONFINITE
    _assert_fits_equal(best, info, want)


# ------------------------------------------------------------------------------------------------ runs
def _oracle_ab_neutral(oracle, abn, ped, p0, eqp, ew, n_starts, seed, lanes, max_iters=10000):
    s0 = abn.gen_start_simplices(seed, 0, n_starts, ped[:, 3].max())
    fits = oracle.fit_batch(ped, p0, eqp, ew, s0, max_iters, lanes=lanes)
    k, model, pred, resid, lse = oracle.select_best(ped, p0, fits["best"])
    return k, model, pred, resid, lse, fits


@pytest.mark.parametrize("case", ("generated", "sparse"))
def test_ab_neutral_and_boot_model_match_oracle(abn, gpu_ctx, golden, oracle, case):
    """src/ab_neutral.rs:13-142 then src/boot_model.rs:17-115, eqp = p0uu, eqp_weight = 1
    (src/alphabeta.rs:33-54): fitted model, predicted divergence, residuals and the bootstrap table."""
    ped = golden[case]
    p0 = golden["p0uu_generated"] if case == "generated" else golden["r_p0uu"]
    seed, n_starts, n_boot = 20260101, 10, 64
    o = abn.default_options(seed=seed)
    model, pred, resid, extra = gpu_ctx.ab_neutral_run(ped, p0, p0, 1.0, n_starts, options=o)
    lanes = int(extra["info"]["lanes"][0])
    k, wmodel, wpred, wresid, wlse, wfits = _oracle_ab_neutral(oracle, abn, ped, p0, p0, 1.0, n_starts, seed, lanes)
    _assert_fits_equal(extra["models"], extra["info"], wfits)
    assert np.array_equal(extra["lse"], wlse)
    assert np.array_equal(model, wmodel)                 # north-star bar: 1e-6; held: bit-exact
    assert np.array_equal(pred, wpred) and np.array_equal(resid, wresid)
    assert np.max(np.abs(pred - wpred)) <= 1e-6          # the stated tolerance, trivially met
    raw, info = gpu_ctx.boot_model_run(ped, model, pred, resid, p0, p0, 1.0, n_boot, options=o)
    lanes_b = int(info["lanes"][0])   # phase B may use a different lane count than phase A
    wraw, wres = oracle.boot_model(ped, wmodel, wpred, wresid, p0, p0, 1.0, seed, 0, 0, n_boot, lanes=lanes_b)
    assert np.array_equal(info["iters"], wres["iters"]) and np.array_equal(info["evals"], wres["evals"])
    assert np.array_equal(raw, wraw)


def test_plan_multi_window_matches_per_window_runs(abn, gpu_ctx, golden, oracle):
    """The batched plan (metaprofile shape: W windows, one topology) equals W independent oracle runs,
    and does not depend on how bootstraps are sharded (boot_offset)."""
    ped = golden["sparse"].copy()
    n = ped.shape[0]
    rng = np.random.default_rng(21)
    W, S, B, seed = 3, 4, 12, 77
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.8, 1.2, (W, 1)) + rng.normal(0, 1e-4, (W, n)))
    p0 = np.array([0.99, 0.985, 0.992])
    o = abn.default_options(seed=seed, lanes_per_chain=16, window_groups=2)   # two concurrent window groups
    plan = abn.Plan(gpu_ctx, ped[:, :3], W, S, B, options=o)
    plan.set_windows(D, p0)
    plan.run()
    out = plan.download()
    cnt = plan.counters()
    assert cnt["fits"] == W * (S + B)
    assert cnt["evals"] == out["info_a"]["evals"].sum() + out["info_b"]["evals"].sum()
    for w in range(W):
        pw = ped.copy()
        pw[:, 3] = D[w]
        s0 = abn.gen_start_simplices(seed, w, S, D[w].max())
        fits = oracle.fit_batch(pw, p0[w], p0[w], 1.0, s0, 10000, lanes=16)
        k, model, pred, resid, _ = oracle.select_best(pw, p0[w], fits["best"])
        assert out["best_start"][w] == k
        assert np.array_equal(out["models"][w], model)
        assert np.array_equal(out["pred"][w], pred) and np.array_equal(out["resid"][w], resid)
        wraw, _ = oracle.boot_model(pw, model, pred, resid, p0[w], p0[w], 1.0, seed, w, 0, B, lanes=16)
        assert np.array_equal(out["raw"][w], wraw)
    # shard of bootstraps [5, 12) of window 1 only
    pw = ped.copy()
    plan2 = abn.Plan(gpu_ctx, ped[:, :3], 1, S, 7, window_offset=1, boot_offset=5, options=o)
    plan2.set_windows(D[1:2], p0[1:2])
    plan2.run()
    out2 = plan2.download()
    assert np.array_equal(out2["models"][0], out["models"][1])
    assert np.array_equal(out2["raw"][0], out["raw"][1, 5:12])
    ms = plan.kernel_ms()
    assert ms["fit_boot"] > 0 and ms["fit_starts"] > 0
    plan.close()
    plan2.close()


def test_device_sqrt_and_division_are_correctly_rounded(abn, gpu_ctx, oracle):
    """The termination test uses sqrt, the cost uses one division: bootstrap rows (est_mm/um/uu) from
    the kernel must equal the host formulas bit for bit."""
    ped = np.array([[0.0, 1.0, 2.0, 0.01], [0.0, 2.0, 3.0, 0.02], [1.0, 2.0, 2.0, 0.015], [0.0, 3.0, 3.0, 0.03]])
    model = np.array([3.1e-4, 7.7e-4, 0.031, 1.1e-3])
    pred = np.array([0.011, 0.019, 0.016, 0.029])
    resid = ped[:, 3] - pred
    raw, info = gpu_ctx.boot_model_run(ped, model, pred, resid, 0.8, 0.8, 1.0, 50,
                                       options=abn.default_options(lanes_per_chain=8))
    for row in raw:
        assert np.array_equal(row, oracle.bootstrap_row(row[:4]))


def test_select_best_and_bootstrap_rows_standalone(abn, gpu_ctx, golden, oracle):
    """src/ab_neutral.rs:83-135 and src/boot_model.rs:86-91 as stand-alone entry points."""
    ped, p0 = golden["sparse"], golden["r_p0uu"]
    rng = np.random.default_rng(12)
    models = np.array([5.8e-05, 6.5e-03, 0.03, 6e-05]) * rng.uniform(0.5, 1.5, (9, 4))
    models[3] = models[7]                       # an exact tie: the lower index must win (stable sort)
    models[5, 0] = np.nan                       # NaN LSE never wins
    k, model, pred, resid, lse = gpu_ctx.select_best(ped, p0, models)
    wk, wmodel, wpred, wresid, wlse = oracle.select_best(ped, p0, models)
    assert k == wk and np.array_equal(model, wmodel)
    assert np.array_equal(pred, wpred) and np.array_equal(resid, wresid)
    assert np.array_equal(lse[~np.isnan(wlse)], wlse[~np.isnan(wlse)]) and np.isnan(lse[5])
    with pytest.raises(abn.AbnError) as e:
        gpu_ctx.select_best(ped, p0, np.full((2, 4), np.nan))
    assert e.value.status == 5
    raw = gpu_ctx.bootstrap_rows(models[:5])
    assert np.array_equal(raw, np.stack([oracle.bootstrap_row(m) for m in models[:5]]))


# ------------------------------------------------------------------------------------------------ properties
def test_full_size_properties_c3(abn, gpu_ctx, oracle):
    """BASELINE config C3 (105 rows, 10000 bootstraps) at full size through size-independent properties:
    determinism (two runs bit-identical), shard-independence, internal consistency of every row, and a
    sampled subset bit-equal to the oracle."""
    from alphabeta_rs_amd import synthetic

    ped, p0 = synthetic.c3_pedigree()
    n = ped.shape[0]
    B = 10000
    o = abn.default_options(seed=20260101)
    plan = abn.Plan(gpu_ctx, ped[:, :3], 1, 10, B, options=o)
    plan.set_windows(ped[:, 3][None, :], np.array([p0]))
    plan.run()
    a = plan.download()
    plan.run()
    b = plan.download()
    assert np.array_equal(a["raw"], b["raw"]) and np.array_equal(a["models"], b["models"])   # idempotence
    raw = a["raw"][0]
    assert np.all(np.isfinite(raw))
    assert np.all(a["info_b"]["status"] != abn.FIT_NONFINITE)
    assert np.all(a["info_b"]["iters"] <= 1000) and np.all(a["info_b"]["evals"] >= 5)
    # every row's equilibrium columns follow src/structs.rs:146-159 from its own (alpha, beta)
    for row in raw[:: B // 50]:
        assert np.array_equal(row, oracle.bootstrap_row(row[:4]))
    # pred + resid reproduces the observations
    assert np.allclose(a["pred"][0] + a["resid"][0], ped[:, 3], rtol=0, atol=1e-17)
    # sampled bootstraps against the oracle (bit-exact)
    lanes = int(a["info_b"]["lanes"][0, 0])
    for b0 in (0, 4321, 9990):
        wraw, wres = oracle.boot_model(ped, a["models"][0], a["pred"][0], a["resid"][0], p0, p0, 1.0, 20260101, 0,
                                       b0, 10, lanes=lanes)
        assert np.array_equal(raw[b0:b0 + 10], wraw)
        assert np.array_equal(a["info_b"]["evals"][0, b0:b0 + 10], wres["evals"])
    # the fitted rates recover the synthetic truth to within a factor of two (one noise realisation)
    an = abn.analyze(raw)
    assert 0.5 < an[0, 0] / synthetic.TRUE_PARAMS[0] < 2.0
    assert 0.5 < an[0, 1] / synthetic.TRUE_PARAMS[1] < 2.0
    plan.close()


# ------------------------------------------------------------------------------------------------ CLI
def test_alphabeta_cli_end_to_end(abn, gpu_ctx, golden, oracle, tmp_path):
    """The `alphabeta` binary (reference flags, src/arguments.rs:93-114) from raw nodelist/edgelist inputs:
    pedigree.txt, analysis.txt and raw.npy (src/cli/alphabeta.rs:28-35) against the oracle pipeline."""
    import subprocess
    from pathlib import Path

    from alphabeta_rs_amd import build as B

    cli = str(B.build_host())
    gold = Path(__file__).resolve().parent / "golden"
    seed, iters = 5, 16
    r = subprocess.run([cli, "-i", str(iters), "-n", "./data/nodelist.txt", "-e", "./data/edgelist.txt", "-o",
                        str(tmp_path), "--seed", str(seed)], capture_output=True, text=True, cwd=str(gold))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Results:" in r.stdout and "Estimated steady state" in r.stdout
    assert (tmp_path / "pedigree.txt").read_text() == (gold / "pedigree_generated.txt").read_text()
    raw = np.load(tmp_path / "raw.npy")
    assert raw.shape == (iters, 7)
    ped, p0 = golden["generated"], golden["p0uu_generated"]
    # auto options on the bundled six-row pedigree: the REFERENCE's serial row-order sums (pedigrees of up to 16 rows;
    # tree code 1), whichever kernel runs (here four wavefronts per chain in both phases)
    tree = abn.reduction_tree(ped[:, :3])
    assert tree == 1
    k, model, pred, resid, _, _ = _oracle_ab_neutral(oracle, abn, ped, p0, p0, 1.0, iters, seed, tree)
    wraw, _ = oracle.boot_model(ped, model, pred, resid, p0, p0, 1.0, seed, 0, 0, iters, lanes=tree)
    assert np.array_equal(raw, wraw)
    # --devices 0 goes through abn_multi_* (one device: no gather) and must write the same files
    multi = tmp_path / "multi"
    multi.mkdir()
    r2 = subprocess.run([cli, "-i", str(iters), "-n", "./data/nodelist.txt", "-e", "./data/edgelist.txt", "-o", str(multi),
                         "--seed", str(seed), "--devices", "0"], capture_output=True, text=True, cwd=str(gold))
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert np.array_equal(np.load(multi / "raw.npy"), raw)
    assert (multi / "analysis.txt").read_text() == (tmp_path / "analysis.txt").read_text()
    # --strict-order: the reference's serial summation order end to end (bit-equal to the oracle's lanes = 1)
    strict = tmp_path / "strict"
    strict.mkdir()
    r3 = subprocess.run([cli, "-i", str(iters), "-n", "./data/nodelist.txt", "-e", "./data/edgelist.txt", "-o", str(strict),
                         "--seed", str(seed), "--strict-order"], capture_output=True, text=True, cwd=str(gold))
    assert r3.returncode == 0, r3.stdout + r3.stderr
    k1, model1, pred1, resid1, _, _ = _oracle_ab_neutral(oracle, abn, ped, p0, p0, 1.0, iters, seed, 1)
    sraw, _ = oracle.boot_model(ped, model1, pred1, resid1, p0, p0, 1.0, seed, 0, 0, iters, lanes=1)
    assert np.array_equal(np.load(strict / "raw.npy"), sraw)
    # the reference's bootstrap.png (src/boot_model.rs:105-109) from the file the CLI wrote
    import sys
    pr = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "scripts" / "plot_bootstrap.py"),
                         str(tmp_path / "raw.npy")], capture_output=True, text=True)
    assert pr.returncode == 0 and (tmp_path / "bootstrap.png").stat().st_size > 5000, pr.stderr
    an = dict(ln.split("\t") for ln in (tmp_path / "analysis.txt").read_text().splitlines())
    want = oracle.analyze(wraw)
    assert float(an["Alpha"]) == want[0, 0] and float(an["SDBeta"]) == want[1, 1]
    if want[2, 0] > 0 and want[3, 0] > 0:      # "lo-hi": unambiguous when both bounds are positive
        lo, hi = an["CIAlpha"].split("-", 1)
        assert float(lo) == want[2, 0] and float(hi) == want[3, 0]


@pytest.mark.parametrize("stream_mode", (0, 1))
def test_deep_pedigree_stream_mode_c5_shape(abn, gpu_ctx, oracle, stream_mode):
    """BASELINE C5's pedigree (8 lineages x 125 generations, N = 20100 rows, T = 125, K = 950): the
    fit kernel streams the u32 bootstrap index row from HBM every evaluation.  A handful of starts and
    bootstraps, bit-equal to the oracle; the index-stream round trip is the property checked at size."""
    from alphabeta_rs_amd import synthetic

    ped, p0 = synthetic.c5_pedigree()
    n = ped.shape[0]
    assert n == 20100 and ped[:, :3].max() == 125
    seed = 31
    # stream_mode 0: bootstrap observations materialised once per fit and streamed (8 B/row/evaluation);
    # stream_mode 1: the u32 index row re-streamed and residuals gathered every evaluation.  Same bits.
    o = abn.default_options(seed=seed, max_iters_start=60, max_iters_boot=40, stream_mode=stream_mode)
    plan = abn.Plan(gpu_ctx, ped[:, :3], 1, 3, 6, options=o)
    plan.set_windows(ped[:, 3][None, :], np.array([p0]))
    plan.run()
    out = plan.download()
    la, lb = int(out["info_a"]["lanes"][0, 0]), int(out["info_b"]["lanes"][0, 0])
    s0 = abn.gen_start_simplices(seed, 0, 3, ped[:, 3].max())
    fits = oracle.fit_batch(ped, p0, p0, 1.0, s0, 60, lanes=la)
    _assert_fits_equal(out["models"][0:0], out["info_a"][0][:0], fits[:0])  # shape sanity only
    assert np.array_equal(out["info_a"]["evals"][0], fits["evals"])
    k, model, pred, resid, _ = oracle.select_best(ped, p0, fits["best"])
    assert out["best_start"][0] == k and np.array_equal(out["models"][0], model)
    assert np.array_equal(out["pred"][0], pred) and np.array_equal(out["resid"][0], resid)
    wraw, wres = oracle.boot_model(ped, model, pred, resid, p0, p0, 1.0, seed, 0, 0, 6, max_iters=40, lanes=lb)
    assert np.array_equal(out["raw"][0], wraw)
    assert np.array_equal(out["info_b"]["evals"][0], wres["evals"])
    plan.close()


@pytest.mark.parametrize("stream_mode", [0, 1])
def test_multi_window_stream_mode_matches_per_window_oracle(abn, gpu_ctx, oracle, stream_mode):
    """VERDICT r02, missing #2: several windows in stream mode — BASELINE C5's per-GPU shard is 63 windows of the deep
    pedigree; every window has its own observations, p0uu, index rows and materialised bootstrap observations
    (`dstar + w B N`).  Four windows of the deep topology sampled every 12th generation (3240 rows: streamed), a handful
    of starts and bootstraps per window, both stream variants, against one oracle run per window; then the same plan
    cut into two plans with window offsets (a shard boundary inside the job)."""
    from alphabeta_rs_amd import synthetic

    W, S, B, seed = 4, 3, 5, 77
    gens, D, p0, _ = synthetic.c5_windows(W, every=12)
    assert gens.shape[0] > 1024 and gens.max() == 120
    o = abn.default_options(seed=seed, max_iters_start=50, max_iters_boot=30, stream_mode=stream_mode)
    plan = abn.Plan(gpu_ctx, gens, W, S, B, options=o)
    plan.set_windows(D, p0)
    plan.run()
    out = plan.download()
    plan.close()
    assert ((int(out["info_b"]["lanes"][0, 0]) >> 8) & 0xff) == 3          # rows streamed in four-row blocks
    for w in range(W):
        ped = np.concatenate([gens, D[w][:, None]], axis=1)
        la, lb = int(out["info_a"]["lanes"][w, 0]), int(out["info_b"]["lanes"][w, 0])
        s0 = abn.gen_start_simplices(seed, w, S, D[w].max())
        fits = oracle.fit_batch(ped, p0[w], p0[w], 1.0, s0, 50, lanes=la)
        assert np.array_equal(out["info_a"]["evals"][w], fits["evals"])
        k, model, pred, resid, _ = oracle.select_best(ped, p0[w], fits["best"])
        assert out["best_start"][w] == k and np.array_equal(out["models"][w], model)
        assert np.array_equal(out["pred"][w], pred) and np.array_equal(out["resid"][w], resid)
        wraw, wres = oracle.boot_model(ped, model, pred, resid, p0[w], p0[w], 1.0, seed, w, 0, B, max_iters=30, lanes=lb)
        assert np.array_equal(out["raw"][w], wraw), w
        assert np.array_equal(out["info_b"]["evals"][w], wres["evals"])
    # windows 1..3 as a shard of their own: same bits as inside the four-window plan
    plan = abn.Plan(gpu_ctx, gens, 3, S, B, window_offset=1, options=o)
    plan.set_windows(D[1:], p0[1:])
    plan.run()
    part = plan.download()
    plan.close()
    assert np.array_equal(part["raw"], out["raw"][1:]) and np.array_equal(part["models"], out["models"][1:])


def test_many_distinct_triples_need_more_than_64k_of_lds(abn, gpu_ctx, oracle):
    """Maximum sizes: a pedigree with ~8 900 distinct (t0, t1-t0, t2-t0) triples over 40 generations needs
    10 (T+1) + K + 4 doubles = 72 KiB of LDS per chain — more than a launch gets without opting in; the
    one-chain-per-workgroup kernels (stream-mode fit, selection, 64-lane cost) opt in up to the CU's 160 KiB.
    Cost, fits, selection and bootstraps bit-equal to the oracle."""
    rng = np.random.default_rng(77)
    n, tmax = 12000, 40
    t0 = rng.integers(0, tmax // 2 + 1, n)
    t1 = t0 + rng.integers(0, tmax - t0 + 1)
    t2 = t0 + rng.integers(0, tmax - t0 + 1)
    d = np.abs(rng.normal(0.02, 0.006, n))
    ped = np.stack([t0, t1, t2, d], axis=1).astype(np.float64)
    k_distinct = len({(a, b, c) for a, b, c in zip(t0, t1 - t0, t2 - t0)})
    assert (10 * (tmax + 1) + k_distinct + 4) * 8 > 64 * 1024, k_distinct
    p0 = 0.8
    cand = np.array([[1e-3, 2e-3, 0.6, 0.01], [5e-4, 5e-4, 0.9, 0.0]])
    cost, dt = gpu_ctx.cost_batch(ped, p0, p0, 1.0, cand, want_dt=True)
    tree = abn.reduction_tree(ped[:, :3])
    assert np.array_equal(cost, np.array([oracle.cost(ped, p0, p0, 1.0, x, lanes=tree) for x in cand]))
    assert np.array_equal(dt[0], oracle.divergence(ped, 1 - p0, p0, *cand[0, :3])[0])
    seed = 5
    o = abn.default_options(seed=seed, max_iters_start=25, max_iters_boot=15)
    plan = abn.Plan(gpu_ctx, ped[:, :3], 1, 2, 3, options=o)
    plan.set_windows(ped[:, 3][None, :], np.array([p0]))
    plan.run()
    out = plan.download()
    la, lb = int(out["info_a"]["lanes"][0, 0]), int(out["info_b"]["lanes"][0, 0])
    s0 = abn.gen_start_simplices(seed, 0, 2, ped[:, 3].max())
    fits = oracle.fit_batch(ped, p0, p0, 1.0, s0, 25, lanes=la)
    assert np.array_equal(out["info_a"]["evals"][0], fits["evals"])
    k, model, pred, resid, _ = oracle.select_best(ped, p0, fits["best"])
    assert out["best_start"][0] == k and np.array_equal(out["models"][0], model)
    assert np.array_equal(out["pred"][0], pred) and np.array_equal(out["resid"][0], resid)
    wraw, wres = oracle.boot_model(ped, model, pred, resid, p0, p0, 1.0, seed, 0, 0, 3, max_iters=15, lanes=lb)
    assert np.array_equal(out["raw"][0], wraw)
    assert np.array_equal(out["info_b"]["evals"][0], wres["evals"])
    plan.close()


# ------------------------------------------------------------------------------------------------ edge cases
def test_edge_cases_tiny_and_degenerate_pedigrees(abn, gpu_ctx, oracle):
    """Smallest inputs and degenerate generations: one row, one start, one bootstrap; all generations 0
    (T = 0: only G^0 = I is ever used); generation 127 (the i8 ceiling); all-zero divergences
    (Model::new falls back to max = 0.1, src/structs.rs:80-83)."""
    cases = {
        "one_row": np.array([[0.0, 1.0, 2.0, 0.01]]),
        "all_gen_zero": np.array([[0.0, 0.0, 0.0, 0.02], [0.0, 0.0, 0.0, 0.03], [0.0, 0.0, 0.0, 0.01]]),
        "gen_127": np.array([[0.0, 127.0, 3.0, 0.2], [5.0, 127.0, 127.0, 0.3], [0.0, 1.0, 1.0, 0.01]]),
        "zero_divergence": np.array([[0.0, 1.0, 2.0, 0.0], [0.0, 2.0, 2.0, 0.0], [1.0, 2.0, 3.0, 0.0]]),
    }
    for name, ped in cases.items():
        for n_starts, n_boot in ((1, 1), (3, 5)):
            o = abn.default_options(seed=9, max_iters_start=400, max_iters_boot=200)
            plan = abn.Plan(gpu_ctx, ped[:, :3], 1, n_starts, n_boot, options=o)
            plan.set_windows(ped[:, 3][None, :], np.array([0.8]))
            plan.run()
            out = plan.download(allow_failed_windows=True)
            plan.close()
            la, lb = int(out["info_a"]["lanes"][0, 0]), int(out["info_b"]["lanes"][0, 0])
            s0 = abn.gen_start_simplices(9, 0, n_starts, ped[:, 3].max())
            fits = oracle.fit_batch(ped, 0.8, 0.8, 1.0, s0, 400, lanes=la)
            assert np.array_equal(out["info_a"]["evals"][0], fits["evals"]), name
            assert np.array_equal(out["info_a"]["status"][0], fits["status"]), name
            k, model, pred, resid, _ = oracle.select_best(ped, 0.8, fits["best"])
            assert out["best_start"][0] == k, name
            if k >= 0:
                assert np.array_equal(out["models"][0], model), name
                wraw, wres = oracle.boot_model(ped, model, pred, resid, 0.8, 0.8, 1.0, 9, 0, 0, n_boot, max_iters=200,
                                               lanes=lb)
                assert np.array_equal(out["raw"][0], wraw, equal_nan=True), name
                assert np.array_equal(out["info_b"]["evals"][0], wres["evals"]), name


def test_invalid_arguments_are_status_codes(abn, gpu_ctx, golden):
    import ctypes as C

    L = abn.load_library()
    ped = golden["generated"]
    dp = ped.ctypes.data_as(C.POINTER(C.c_double))
    out = np.zeros(8)
    op = out.ctypes.data_as(C.POINTER(C.c_double))
    # null pointers / non-positive sizes -> ABN_ERR_INVALID_ARG (1), never a crash
    assert L.abn_cost_batch(gpu_ctx._h, None, None, 6, 0.7, 0.7, 1.0, dp, 1, None, None, None, None, 0, op, None, None) == 1
    assert L.abn_cost_batch(gpu_ctx._h, None, dp, 0, 0.7, 0.7, 1.0, dp, 1, None, None, None, None, 0, op, None, None) == 1
    assert L.abn_fit_batch(gpu_ctx._h, None, dp, 6, 0.7, 0.7, 1.0, None, 1, None, 10, op, None) == 1
    assert L.abn_ab_neutral_run(gpu_ctx._h, None, dp, 6, 0.7, 0.7, 1.0, 0, op, op, op, None, None, None) == 1
    assert L.abn_boot_model_run(gpu_ctx._h, None, dp, 6, op, op, op, 0.7, 0.7, 1.0, 0, op, None) == 1
    h = C.c_void_p()
    assert L.abn_plan_create(gpu_ctx._h, None, dp, 6, 0, 1, 1, 0, 0, C.byref(h)) == 1
    assert L.abn_init(99, None, C.byref(h)) == 1
    # plan used out of order -> ABN_ERR_STATE (6)
    plan = abn.Plan(gpu_ctx, ped[:, :3], 1, 2, 2)
    with pytest.raises(abn.AbnError) as e:
        plan.run()
    assert e.value.status == 6
    plan.close()
    # options outside their range -> ABN_ERR_INVALID_ARG with a message
    for bad in (dict(max_iters_start=-1), dict(max_iters_boot=1 << 29), dict(lanes_per_chain=12),
                dict(sd_tolerance=float("nan")), dict(stream_mode=2)):
        with pytest.raises(abn.AbnError) as e:
            abn.Plan(gpu_ctx, ped[:, :3], 1, 2, 2, options=abn.default_options(**bad))
        assert e.value.status == 1, bad
    with pytest.raises(abn.AbnError):
        gpu_ctx.fit_batch(ped, 0.7, 0.7, 1.0, np.zeros((1, 5, 4)), -5)
    # zero candidates / fits are fine
    assert gpu_ctx.cost_batch(ped, 0.7, 0.7, 1.0, np.zeros((0, 4))).shape == (0,)


def test_full_size_properties_c2_and_c4(abn, gpu_ctx, golden, oracle):
    """BASELINE C2 (bundled pedigree x 1000 bootstraps) and a C4 shard (25 windows x 1000 bootstraps) at
    full size: shard-independence (two half-shards of the bootstraps equal the whole, bit for bit),
    idempotence, and sampled windows / bootstraps bit-equal to the oracle."""
    from alphabeta_rs_amd import synthetic

    # ---- C2
    ped, p0 = golden["generated"], golden["p0uu_generated"]
    o = abn.default_options(seed=20260101)
    whole = abn.Plan(gpu_ctx, ped[:, :3], 1, 10, 1000, options=o)
    whole.set_windows(ped[:, 3][None, :], np.array([p0]))
    whole.run()
    a = whole.download()
    halves = []
    for b0 in (0, 500):
        h = abn.Plan(gpu_ctx, ped[:, :3], 1, 10, 500, boot_offset=b0, options=o)
        h.set_windows(ped[:, 3][None, :], np.array([p0]))
        h.run()
        halves.append(h.download()["raw"][0])
        h.close()
    assert np.array_equal(np.concatenate(halves), a["raw"][0])
    lb = int(a["info_b"]["lanes"][0, 0])
    wraw, _ = oracle.boot_model(ped, a["models"][0], a["pred"][0], a["resid"][0], p0, p0, 1.0, 20260101, 0, 990, 10,
                                lanes=lb)
    assert np.array_equal(a["raw"][0, 990:], wraw)
    whole.close()

    # ---- C4 shard
    gens, D, p0w, _ = synthetic.c4_windows(25)
    plan = abn.Plan(gpu_ctx, gens, 25, 10, 1000, options=o)
    plan.set_windows(D, p0w)
    plan.run()
    r1 = plan.download()
    plan.run()
    r2 = plan.download()
    assert np.array_equal(r1["raw"], r2["raw"], equal_nan=True) and np.array_equal(r1["best_start"], r2["best_start"])
    assert np.all(r1["best_start"] >= 0)
    la, lb = int(r1["info_a"]["lanes"][0, 0]), int(r1["info_b"]["lanes"][0, 0])
    for w in (0, 13, 24):
        pw = np.concatenate([gens, D[w][:, None]], axis=1)
        s0 = abn.gen_start_simplices(20260101, w, 10, D[w].max())
        fits = oracle.fit_batch(pw, p0w[w], p0w[w], 1.0, s0, 10000, lanes=la)
        k, model, pred, resid, _ = oracle.select_best(pw, p0w[w], fits["best"])
        assert r1["best_start"][w] == k and np.array_equal(r1["models"][w], model)
        wraw, _ = oracle.boot_model(pw, model, pred, resid, p0w[w], p0w[w], 1.0, 20260101, w, 500, 8, lanes=lb)
        assert np.array_equal(r1["raw"][w, 500:508], wraw)
    plan.close()


def test_metaprofile_batch_driver(abn, gpu_ctx, golden, oracle, tmp_path):
    """`metaprofile_alphabeta` (src/cli/metaprofile.rs:33-114): window directories as src/setup.rs writes them, all
    windows fitted by batched plans; results.txt and the (iterations, 7, windows) raw.npy against per-window oracle
    runs.  A window without its nodelist IN THE MIDDLE of the enumeration is reported and skipped like :64-65, and one
    window has a different pedigree topology (a plan of its own): every window must still draw from the Philox streams
    of ITS position in the (region, window) enumeration.  The metaplot script runs on the real results.txt."""
    import shutil
    import subprocess
    import sys
    from pathlib import Path

    from alphabeta_rs_amd import build as B

    B.build_host()
    gold = Path(__file__).resolve().parent / "golden"
    dirs = [(r, w) for r in ("upstream", "gene", "downstream") for w in (0, 50)]   # enumeration index 0..5
    missing, other = ("gene", 0), ("upstream", 50)                                 # index 2 is skipped, index 1 differs
    for r, w in dirs:
        d = tmp_path / r / str(w)
        d.mkdir(parents=True)
        if (r, w) == missing:
            continue
        shutil.copy(gold / "data" / "edgelist.txt", d / "edgelist.txt")
        nodes = (gold / "data" / "nodelist.txt").read_text()
        if (r, w) == other:      # G4_8 not sampled: three samples, three pairs — another topology
            nodes = nodes.replace("./data/methylome/G4_8.txt,4_8,4,Y", "./data/methylome/G4_8.txt,4_8,4,N")
        (d / "nodelist.txt").write_text(nodes)
    iters, seed = 8, 123
    r = subprocess.run([str(B.META_CLI), "-o", str(tmp_path), "--name", "t", "-s", "50", "--iterations", str(iters),
                        "--seed", str(seed), "--devices", "0"], capture_output=True, text=True, cwd=str(gold))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Error: Error while building pedigree" in r.stdout           # the missing window
    raw = np.load(tmp_path / "raw.npy")
    assert raw.shape == (iters, 7, 5)
    lines = (tmp_path / "results.txt").read_text().splitlines()
    assert lines[0].startswith("run;window;cg_count;region;alpha;beta") and len(lines) == 1 + 5
    ped, p0_all = golden["generated"], golden["p0uu_generated"]
    ped3 = ped[[0, 1, 3]]                                                # pairs of (G0, G1_2, G4_2)
    p0_three = 0.642544651312622   # p0uu of that window (Pedigree::build over the three samples' common sites; host mirror)
    ok_index = [0, 1, 3, 4, 5]                                           # enumeration indices of the fitted windows
    for k, widx in enumerate(ok_index):
        pw, p0 = (ped3, p0_three) if widx == 1 else (ped, p0_all)
        tree = abn.reduction_tree(pw[:, :3])
        s0 = abn.gen_start_simplices(seed, widx, iters, pw[:, 3].max())
        fits = oracle.fit_batch(pw, p0, p0, 1.0, s0, 10000, lanes=tree)
        _, model, pred, resid, _ = oracle.select_best(pw, p0, fits["best"])
        wraw, _ = oracle.boot_model(pw, model, pred, resid, p0, p0, 1.0, seed, widx, 0, iters, lanes=tree)
        assert np.array_equal(raw[:, :, k], wraw), (k, widx)
        f = lines[1 + k].split(";")
        assert f[0] == "t" and int(f[1]) == k and f[3] == ["upstream", "upstream", "gene", "downstream", "downstream"][k]
        assert float(f[4]) == model[0] and float(f[5]) == model[1]
        assert float(f[8]) == 1.0 - p0
        an = oracle.analyze(wraw)
        assert float(f[9]) == an[1, 0] and float(f[10]) == an[1, 1]
    pr = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "scripts" / "plot_metaplot.py"),
                         str(tmp_path / "results.txt")], capture_output=True, text=True)
    assert pr.returncode == 0 and (tmp_path / "metaplot.png").stat().st_size > 5000, pr.stderr


@pytest.mark.parametrize("no_skip", (1, 0))
def test_two_pass_phase_a_is_bit_identical(abn, gpu_ctx, golden, oracle, no_skip):
    """Phase A with more than 4096 start chains runs in two passes (every chain for at most 1000 iterations,
    then the parked ones from their stored Nelder-Mead state): iterations, evaluations and fitted vectors
    must equal an uninterrupted run.  Many starts on the bundled pedigree run all 10000 iterations
    (no_skip = 1: the repetitions of a stuck fit are executed, so those chains go through the parking)."""
    ped, p0 = golden["generated"], golden["p0uu_generated"]
    W, S, seed = 52, 80, 77
    rng = np.random.default_rng(2)
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.7, 1.3, (W, 1)))
    p0w = rng.uniform(0.6, 0.8, W)
    o = abn.default_options(seed=seed, no_fixed_point_skip=no_skip)
    plan = abn.Plan(gpu_ctx, ped[:, :3], W, S, 2, options=o)
    plan.set_windows(D, p0w)
    plan.run()
    out = plan.download()
    cnt = plan.counters()
    plan.close()
    it = out["info_a"]["iters"]
    assert (it > 1000).sum() > 50 and (it == 10000).sum() > 10      # the cap was exercised
    assert (cnt["evals_skipped"] == 0) == (no_skip == 1)
    assert np.all(out["info_a"]["status"] != 4)                       # nothing left parked
    la = int(out["info_a"]["lanes"][0, 0])
    for w in (0, 17, 51):
        pw = np.concatenate([ped[:, :3], D[w][:, None]], axis=1)
        s0 = abn.gen_start_simplices(seed, w, S, D[w].max())
        fits = oracle.fit_batch(pw, p0w[w], p0w[w], 1.0, s0, 10000, lanes=la)
        assert np.array_equal(out["info_a"]["iters"][w], fits["iters"])
        assert np.array_equal(out["info_a"]["evals"][w], fits["evals"])
        assert np.array_equal(out["info_a"]["status"][w], fits["status"])
        k, model, _, _, _ = oracle.select_best(pw, p0w[w], fits["best"])
        assert out["best_start"][w] == k and np.array_equal(out["models"][w], model)


def test_fixed_point_skip_changes_no_output(abn, gpu_ctx, golden, oracle):
    """argmin 0.8.1 leaves the simplex untouched after a rejected contraction, so such a fit repeats the same
    two evaluations until max_iters.  By default the kernels finish it on the spot with the counters the
    repetitions would have produced; every output must equal the run that executes them (and the oracle,
    which always executes them).  Covers the speculative phase-A kernel, the resident and the stream kernel."""
    ped, p0 = golden["generated"], golden["p0uu_generated"]
    W, S, B, seed = 12, 40, 64, 5
    rng = np.random.default_rng(11)
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.7, 1.3, (W, 1)))
    p0w = rng.uniform(0.6, 0.8, W)
    outs, cnts = [], []
    for no_skip in (0, 1):
        plan = abn.Plan(gpu_ctx, ped[:, :3], W, S, B, options=abn.default_options(seed=seed, no_fixed_point_skip=no_skip))
        plan.set_windows(D, p0w)
        plan.run()
        outs.append(plan.download())
        cnts.append(plan.counters())
        plan.close()
    for k in ("models", "pred", "resid", "raw", "best_start"):
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k
    for k in ("info_a", "info_b"):
        assert outs[0][k].tobytes() == outs[1][k].tobytes(), k
    assert cnts[0]["evals"] == cnts[1]["evals"] and cnts[1]["evals_skipped"] == 0
    assert cnts[0]["evals_skipped_starts"] > 0
    stuck = outs[0]["info_a"]["iters"] == 10000
    assert stuck.sum() > 5
    # against the oracle (which executes the repetitions)
    la = int(outs[0]["info_a"]["lanes"][0, 0])
    w = int(np.argmax(stuck.sum(axis=1)))
    pw = np.concatenate([ped[:, :3], D[w][:, None]], axis=1)
    fits = oracle.fit_batch(pw, p0w[w], p0w[w], 1.0, abn.gen_start_simplices(seed, w, S, D[w].max()), 10000, lanes=la)
    for k in ("status", "iters", "evals", "best_cost"):
        assert np.array_equal(outs[0]["info_a"][w][k], fits[k], equal_nan=True), k
    # plain (several chains per wavefront) and stream kernels through abn_fit_batch
    big = synthetic_pedigree(np.random.default_rng(3), 700, 12)
    for pedx, lanes, iters in ((ped, 16, 3000), (big, 64, 400)):
        s0 = abn.gen_start_simplices(seed, 3, 24, pedx[:, 3].max())
        res = [gpu_ctx.fit_batch(pedx, 0.7, 0.7, 1.0, s0, iters,
                                 options=abn.default_options(lanes_per_chain=lanes, no_fixed_point_skip=ns))
               for ns in (0, 1)]
        assert np.array_equal(res[0][0], res[1][0], equal_nan=True)
        assert res[0][1].tobytes() == res[1][1].tobytes()
        want = oracle.fit_batch(pedx, 0.7, 0.7, 1.0, s0, iters, lanes=int(res[0][1]["lanes"][0]))
        _assert_fits_equal(res[0][0], res[0][1], want)


@pytest.mark.parametrize("order", (-1, 0))   # the canonical tree / the serial default of a six-row pedigree (STRICT variant)
@pytest.mark.parametrize("variant,no_skip,iters", ((1, 0, 10000), (0, 1, 400), (0, 0, 10000), (0, 0, 0), (0, 1, 1), (1, 0, 2)))
def test_speculative_phase_a_all_branches(abn, gpu_ctx, golden, oracle, variant, no_skip, iters, order):
    """Phase A of a small plan runs on abn_fit_spec_kernel (three evaluation wavefronts + the keeper).  Its
    rare branches — NelderMead::shrink after a NaN reflection cost or, in the textbook variant, after a
    rejected contraction; the repeated iterations of a stuck fit — must follow the oracle bit for bit."""
    ped, p0 = golden["generated"], golden["p0uu_generated"]
    W, S, seed = 5, 40, 91
    rng = np.random.default_rng(8)
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.7, 1.3, (W, 1)))
    D[3, 2] = np.nan                      # every cost of window 3 is NaN: a shrink per iteration
    p0w = rng.uniform(0.6, 0.8, W)
    o = abn.default_options(seed=seed, shrink_on_failed_contraction=variant, no_fixed_point_skip=no_skip,
                            max_iters_start=iters, strict_order=order)
    plan = abn.Plan(gpu_ctx, ped[:, :3], W, S, 0, options=o)
    plan.set_windows(D, p0w)
    plan.run_phase(0)
    with pytest.raises(abn.AbnError) as err:          # a window without a finite start is REPORTED (the reference
        plan.download()                                # panics, src/ab_neutral.rs:28,100), never passed on silently
    assert err.value.status == 5 and plan.failed_windows() == 1
    out = plan.download(allow_failed_windows=True)
    plan.close()
    la = int(out["info_a"]["lanes"][0, 0])
    assert la == (0x10040 if order < 0 else 1) == abn.reduction_tree(ped[:, :3], o)   # on four wavefronts per chain
    assert out["best_start"][3] == -1 and np.all(out["info_a"]["status"][3] == 2)
    assert np.all(np.isnan(out["models"][3])) and np.all(np.isnan(out["pred"][3]))
    for w in range(W):
        pw = np.concatenate([ped[:, :3], D[w][:, None]], axis=1)
        mx = D[w].max() if w != 3 else np.nanmax(D[w])
        s0 = abn.gen_start_simplices(seed, w, S, float(np.fmax.reduce(D[w])))
        fits = oracle.fit_batch(pw, p0w[w], p0w[w], 1.0, s0, iters, shrink_variant=variant, lanes=la)
        for k in ("status", "iters", "evals"):
            assert np.array_equal(out["info_a"][w][k], fits[k]), (w, k)
        ok = fits["status"] != 2
        assert np.array_equal(out["info_a"][w]["best_cost"][ok], fits["best_cost"][ok])
        if w != 3:
            k, model, _, _, _ = oracle.select_best(pw, p0w[w], fits["best"])
            assert out["best_start"][w] == k and np.array_equal(out["models"][w], model)


@pytest.mark.parametrize("case", ("generated", "sparse"))
def test_persistent_refill_kernel_is_schedule_independent(abn, gpu_ctx, golden, oracle, case):
    """More wavefronts than the GPU holds (> 3072) and several chains per wavefront: phase B runs on the persistent
    kernel whose lane groups take the next chain from an atomic queue.  Which group runs which chain depends
    on timing; the results must not: byte-identical to the static launch (two window groups use the plain
    kernel) and to the oracle."""
    if case == "generated":      # N = 6 -> 8 lanes per chain, 8 chains per wavefront
        ped, p0, W, S, B = golden["generated"], golden["p0uu_generated"], 13, 4, 2000
    else:                        # N = 78 -> 16 lanes per chain, 4 chains per wavefront
        ped, p0, W, S, B = golden["sparse"], golden["r_p0uu"], 8, 4, 1600
    seed = 404
    rng = np.random.default_rng(5)
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.8, 1.25, (W, 1)))
    p0w = np.clip(p0 * rng.uniform(0.9, 1.1, W), 0.05, 0.95)
    outs = []
    for groups in (1, 2):     # strict_order = -1: the tree kernels (the serial default of a six-row pedigree has no persistent form)
        plan = abn.Plan(gpu_ctx, ped[:, :3], W, S, B,
                        options=abn.default_options(seed=seed, window_groups=groups, strict_order=-1))
        plan.set_windows(D, p0w)
        plan.run()
        outs.append(plan.download())
        handed = plan.tail_handed()
        plan.close()
        if groups == 1:   # round 4: the last chains of the time-sliced launch finish on the speculative kernel (same bits)
            assert 0 < handed[1] <= 2048, handed   # at most 8 per CU (the 12-wavefronts-per-CU geometry of this launch)
        else:
            assert handed == (0, 0), handed
    lanes = int(outs[0]["info_b"]["lanes"][0, 0])          # the canonical tree code, whatever the packed lane count
    assert lanes == abn.reduction_tree(ped[:, :3], abn.default_options(strict_order=-1)) == 0x10040
    packed = 8 if case == "generated" else 16
    assert W * B // (64 // packed) > 3072                 # the persistent launch was taken
    for k in ("models", "pred", "resid", "raw", "best_start"):
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k
    for k in ("info_a", "info_b"):
        assert outs[0][k].tobytes() == outs[1][k].tobytes(), k
    out = outs[0]
    for w in (0, W - 1):
        pw = np.concatenate([ped[:, :3], D[w][:, None]], axis=1)
        raw, res = oracle.boot_model(pw, out["models"][w], out["pred"][w], out["resid"][w], p0w[w], p0w[w], 1.0,
                                     seed, w, 0, B, lanes=lanes)
        assert np.array_equal(out["raw"][w], raw, equal_nan=True)
        for k in ("iters", "evals", "status"):
            assert np.array_equal(out["info_b"][w][k], res[k]), k


def test_device_geometry_is_read_from_the_device_and_is_the_mi355x(abn, gpu_ctx):
    """VERDICT r03 hygiene: CU count and LDS size come from hipDeviceProp at abn_init (launch geometry = multiples of the CU
    count), asserted here to be the MI355X's: 256 CUs, 160 KiB of LDS per CU, 3072 / 2048 persistent wavefronts."""
    assert gpu_ctx.device_info() == {"compute_units": 256, "lds_kib_per_cu": 160, "persistent_wavefronts": 3072,
                                     "persistent_wavefronts_small": 2048}


def test_lost_fifo_entry_is_an_error_at_sync(abn):
    """a parked chain whose FIFO entry never appears must not end as uninitialised rows under ABN_OK on
    the paths that never call abn_plan_download — the torch.distributed shard runner reads a bound buffer after
    abn_plan_sync.  Fault injection on the -DABN_MEASUREMENT_KNOBS build (build/libabn_knobs.so, made by build()): the first
    entry of FIFO shard 0 is never published; the launch must still end (bounded spin) and abn_plan_sync,
    abn_plan_failed_windows and abn_plan_download must each report ABN_ERR_HIP.  A subprocess: the variant library is a
    second copy of the product library."""
    import subprocess
    import sys
    import textwrap
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    lib = root / "build" / "libabn_knobs.so"
    if not lib.exists():
        from alphabeta_rs_amd import build as B

        B.build_knobs()
    code = textwrap.dedent("""
        import numpy as np
        import alphabeta_rs_amd as A
        from alphabeta_rs_amd import synthetic
        ped, p0 = synthetic.c3_pedigree()
        W, S, B = 8, 4, 3000                 # 24 000 bootstrap chains of 16 lanes: the time-sliced persistent launch
        rng = np.random.default_rng(3)
        D = np.abs(ped[:, 3][None, :] * rng.uniform(0.9, 1.1, (W, 1)))
        with A.Context(0) as ctx:
            plan = A.Plan(ctx, ped[:, :3], W, S, B, options=A.default_options(seed=11))
            plan.set_windows(D, np.full(W, p0))
            plan.run()
            seen = []
            for what in (plan.sync, plan.failed_windows, plan.download):
                try:
                    what()
                    seen.append("ok")
                except A.AbnError as e:
                    seen.append(e.status_name if hasattr(e, "status_name") else str(e))
            print("KERNELS", plan.last_kernels())
            print("SEEN", seen)
            plan.close()
    """)
    import os

    def run(drop):
        env = dict(os.environ, ABNEUTRAL_HIP_LIB=str(lib), ABN_DROP_FIFO_ENTRY=drop, PYTHONPATH=str(root))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout

    out = run("0")                               # the same build without the fault: all three succeed
    assert out.count("ok") == 3, out
    out = run("1")
    seen = out.split("SEEN", 1)[1]
    assert seen.count("ABN_ERR_HIP") == 3 and "ok" not in seen, out
    assert "finished" in seen and "chains" in seen, out     # the message names the count


def test_reference_unit_tests_through_the_cpp_mirror(abn, gpu_ctx):
    """The reference's enabled unit tests on the path (same_as_r, test_cost_function, build_pedigree) restated
    against the C++ mirror of its API and executed on the GPU (alphabeta_rs_amd/host/reference_tests.cpp).
    The fixtures sit where the reference expects them relative to the working directory: ./data/..."""
    import shutil
    import subprocess
    import tempfile
    from pathlib import Path

    from alphabeta_rs_amd import build as B

    B.build_host()
    gold = Path(__file__).resolve().parent / "golden"
    with tempfile.TemporaryDirectory() as td:
        shutil.copytree(gold / "data", Path(td) / "data")
        shutil.copy(gold / "pedigree.txt", Path(td) / "data" / "pedigree.txt")
        shutil.copy(gold / "divergence.txt", Path(td) / "data" / "divergence.txt")
        shutil.copy(gold / "pedigree_generated.txt", Path(td) / "pedigree_generated.txt")
        r = subprocess.run([str(B.REF_TESTS)], capture_output=True, text=True, cwd=td)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "same_as_r ... ok" in r.stdout and "test_cost_function ... ok" in r.stdout
    assert "build_pedigree ... ok" in r.stdout


def test_kernel_choice_follows_the_launch_size(abn, gpu_ctx):
    """abn_plan_last_kernels: which fit kernel a launch of a given size gets.  The choice never changes a
    bit of the result (test_results_do_not_depend_on_launch_size); this pins the rules themselves on the C3 topology."""
    from alphabeta_rs_amd import synthetic

    ped, p0 = synthetic.c3_pedigree()
    deep, dp0 = synthetic.c5_pedigree(every=12)

    def kernels(gens, d, p, W, S, B, **opts):
        o = abn.default_options(max_iters_start=30, max_iters_boot=20, **opts)
        plan = abn.Plan(gpu_ctx, gens, W, S, B, options=o)
        plan.set_windows(np.tile(d, (W, 1)), np.full(W, p))
        plan.run()
        plan.download()
        k = plan.last_kernels()
        plan.close()
        return k

    k = kernels(ped[:, :3], ped[:, 3], p0, 1, 10, 10000)        # BASELINE C3: 2500 wavefronts of four chains
    assert k["starts"] == ("speculative", 64) and k["boot"] == ("persistent", 16)
    k = kernels(ped[:, :3], ped[:, 3], p0, 1, 10, 1000)         # few bootstraps: latency-bound like the starts
    assert k["boot"] == ("speculative", 64)
    k = kernels(ped[:, :3], ped[:, 3], p0, 1, 10, 1500)         # ... up to 1.5 x what that kernel keeps resident (6 per CU)
    assert k["boot"] == ("speculative", 64)
    k = kernels(ped[:, :3], ped[:, 3], p0, 1, 10, 2500)         # a wavefront per chain still beats packing
    assert k["boot"] == ("resident", 64)
    k = kernels(ped[:, :3], ped[:, 3], p0, 1, 10, 6000)         # packed, fits the GPU: the plain launch
    assert k["boot"] == ("resident", 16)
    k = kernels(ped[:, :3], ped[:, 3], p0, 25, 10, 1000)        # a multi-window shard: persistent with time slicing
    assert k["starts"] == ("speculative", 64) and k["boot"] == ("persistent", 16)
    k = kernels(ped[:, :3], ped[:, 3], p0, 200, 10, 2)          # BASELINE C4's 2000 start chains: still speculative (phase A: up to 16 per CU)
    assert k["starts"] == ("speculative", 64)
    k = kernels(ped[:, :3], ped[:, 3], p0, 500, 10, 2)          # 5000 start chains: a wavefront per chain
    assert k["starts"] == ("resident", 64)
    k = kernels(ped[:, :3], ped[:, 3], p0, 1, 10, 10000, strict_order=1)   # strict order: speculative or plain launches
    assert k["starts"] == ("speculative", 64) and k["boot"] == ("resident", 16)
    k = kernels(deep[:, :3], deep[:, 3], dp0, 1, 2, 4)          # 3240 rows: streamed
    assert k["starts"][0] == "stream" and k["boot"][0] == "stream"


def test_plan_may_outlive_its_context(abn):
    """a plan destroyed after abn_shutdown (easy from Python: ctx.close() before the Plan is collected) used
    to give its buffers back to a freed pool.  The pool is shared and closed now: the buffers are simply freed."""
    from alphabeta_rs_amd import synthetic

    ped, p0 = synthetic.c3_pedigree()
    for _ in range(3):
        ctx = abn.Context(0)
        plan = abn.Plan(ctx, ped[:, :3], 2, 3, 8, options=abn.default_options(max_iters_start=20, max_iters_boot=10))
        plan.set_windows(np.tile(ped[:, 3], (2, 1)), np.full(2, p0))
        plan.run()
        out = plan.download()
        ctx.close()                    # abn_shutdown first ...
        plan.close()                   # ... then abn_plan_destroy
        assert np.isfinite(out["raw"]).all()
