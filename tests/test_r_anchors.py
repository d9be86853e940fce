"""The reference's own R outputs (data/desired_output/) as distributional anchors for what no enabled Rust test pins:
the Nelder-Mead optimiser (a7) and the residual bootstrap (a9).  R's AlphaBeta ran the same objective with its own
optimiser and its own random streams, so the comparison is statistical: bootstrap standard errors of alpha, beta and
beta/alpha on data/pedigree_sparse.txt against Boutput_standard_errors_*.txt, the fitted optimum against R's ten best
optima (ABneutral_estimatats_...txt), for BOTH readings of argmin's rejected-contraction branch.

Stated tolerances: bootstrap SDs within 15 % of R's (R's number of bootstraps is not recorded; 1000 bootstraps here
have a sampling error of about 2 %, R's own sample presumably more), optimum inside R's run-to-run spread."""
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"
SD_TOL = 0.15
N_STARTS, N_BOOT, SEED = 100, 1000, 20260101


def r_anchors():
    out = {}
    for ln in (GOLDEN / "r_boutput.txt").read_text().splitlines():
        if ln and not ln.startswith("#"):
            k, v = ln.split()
            out[k] = float(v)
    est = np.loadtxt(GOLDEN / "r_abneutral_estimates.txt", skiprows=1, usecols=(1, 2, 3, 4, 5))
    return out, est


def check_against_r(raw, model, lse_sorted, best_sorted):
    r, est = r_anchors()
    a, b = raw[:, 0], raw[:, 1]
    got = {"se_alpha": a.std(ddof=1), "se_beta": b.std(ddof=1), "se_beta_alpha": (b / a).std(ddof=1)}
    for k, v in got.items():
        assert abs(v / r[k] - 1.0) < SD_TOL, (k, v, r[k])
    # the optimum: inside R's ten best runs' spread (alpha 5.798e-05..5.801e-05, beta 6.556e-03..6.559e-03), and the
    # bootstrap's base model is R's (Boutput_boot_base_*) to four digits
    assert est[:, 0].min() <= model[0] <= est[:, 0].max() and est[:, 1].min() <= model[1] <= est[:, 1].max()
    assert abs(model[0] / r["boot_base_alpha"] - 1) < 5e-4 and abs(model[1] / r["boot_base_beta"] - 1) < 5e-4
    # one more (weak) anchor the reference holds and no test reads: data/model_wt.txt, a dump of a Rust-side run on this
    # pedigree (alpha 5.814e-05, beta 6.575e-03: 0.3 % from R's optimum; which run produced it is not recorded, hence 0.5 %)
    wt = dict(ln.split() for ln in (GOLDEN / "model_wt.txt").read_text().splitlines() if ln.strip())
    assert abs(model[0] / float(wt["Alpha"]) - 1) < 5e-3 and abs(model[1] / float(wt["Beta"]) - 1) < 5e-3
    # our ten best starts agree with each other at least as well as R's ten best runs do (objective spread 3e-8 rel.)
    spread = (lse_sorted[9] - lse_sorted[0]) / lse_sorted[0]
    r_spread = (est[:, 4].max() - est[:, 4].min()) / est[:, 4].min()
    assert 0 <= spread <= r_spread
    assert np.ptp(best_sorted[:10, 0]) <= np.ptp(est[:, 0]) and np.ptp(best_sorted[:10, 1]) <= np.ptp(est[:, 1])
    return got


@pytest.mark.parametrize("variant", (0, 1))
def test_oracle_bootstrap_distribution_matches_r(oracle, abn, golden, variant):
    ped, p0 = golden["sparse"], golden["r_p0uu"]
    tree = abn.reduction_tree(ped[:, :3])
    s0 = abn.gen_start_simplices(SEED, 0, N_STARTS, ped[:, 3].max())
    fits = oracle.fit_batch(ped, p0, p0, 1.0, s0, 10000, lanes=tree, shrink_variant=variant)
    lse = np.array([oracle.lse(ped, p0, x) for x in fits["best"]])
    order = np.argsort(lse, kind="stable")
    k, model, pred, resid, _ = oracle.select_best(ped, p0, fits["best"])
    raw, res = oracle.boot_model(ped, model, pred, resid, p0, p0, 1.0, SEED, 0, 0, N_BOOT, lanes=tree, table=True,
                                 shrink_variant=variant)
    got = check_against_r(raw, model, lse[order], fits["best"][order])
    # what the two readings of argmin 0.8.1 differ in: starts that end in a rejected contraction stay there until
    # max_iters with variant 0 (a few per cent of random starts), never with the textbook shrink; no BOOTSTRAP fit of
    # this pedigree gets there, so the bootstrap table — and the R anchors — cannot tell the variants apart
    stuck_a = float((fits["iters"] == 10000).mean())
    stuck_b = float((res["iters"] == 1000).mean())
    assert stuck_b == 0.0
    assert (0.0 < stuck_a < 0.15) if variant == 0 else stuck_a == 0.0
    assert not np.any(fits["status"][order[:10]] == 1)          # no stuck start is among the ten best
    print(f"variant {variant}: SD(alpha) {got['se_alpha']:.4e} SD(beta) {got['se_beta']:.4e} "
          f"SD(beta/alpha) {got['se_beta_alpha']:.4f}; starts at max_iters {stuck_a:.3f}")


def test_both_variants_give_the_same_bootstrap_table_here(oracle, abn, golden):
    """the R anchors cannot discriminate: the two variants' bootstrap tables on this pedigree are byte-identical"""
    ped, p0 = golden["sparse"], golden["r_p0uu"]
    tree = abn.reduction_tree(ped[:, :3])
    s0 = abn.gen_start_simplices(SEED, 0, 30, ped[:, 3].max())
    tabs = []
    for variant in (0, 1):
        fits = oracle.fit_batch(ped, p0, p0, 1.0, s0, 10000, lanes=tree, shrink_variant=variant)
        k, model, pred, resid, _ = oracle.select_best(ped, p0, fits["best"])
        raw, _ = oracle.boot_model(ped, model, pred, resid, p0, p0, 1.0, SEED, 0, 0, 300, lanes=tree, table=True,
                                   shrink_variant=variant)
        tabs.append(raw)
    assert np.array_equal(tabs[0], tabs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", (0, 1))
def test_gpu_bootstrap_distribution_matches_r(abn, gpu_ctx, oracle, golden, variant):
    """the same anchors through the HIP path (one plan: 100 starts + 1000 bootstraps), plus bit-equality with the
    oracle's table"""
    ped, p0 = golden["sparse"], golden["r_p0uu"]
    o = abn.default_options(seed=SEED, shrink_on_failed_contraction=variant)
    plan = abn.Plan(gpu_ctx, ped[:, :3], 1, N_STARTS, N_BOOT, options=o)
    plan.set_windows(ped[:, 3][None, :], np.array([p0]))
    plan.run()
    out = plan.download()
    plan.close()
    tree = int(out["info_b"]["lanes"][0, 0])
    s0 = abn.gen_start_simplices(SEED, 0, N_STARTS, ped[:, 3].max())
    fits = oracle.fit_batch(ped, p0, p0, 1.0, s0, 10000, lanes=tree, shrink_variant=variant)
    lse = np.array([oracle.lse(ped, p0, x) for x in fits["best"]])
    order = np.argsort(lse, kind="stable")
    assert np.array_equal(out["info_a"]["iters"][0], fits["iters"])
    check_against_r(out["raw"][0], out["models"][0], lse[order], fits["best"][order])
    k, model, pred, resid, _ = oracle.select_best(ped, p0, fits["best"])
    wraw, _ = oracle.boot_model(ped, model, pred, resid, p0, p0, 1.0, SEED, 0, 0, N_BOOT, lanes=tree, shrink_variant=variant)
    assert np.array_equal(out["raw"][0], wraw)
