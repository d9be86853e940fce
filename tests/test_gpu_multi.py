"""GPU tests of what makes the N > 1 path correct by construction: results that do not depend on the size of a
launch (hence not on sharding), the sharded HIP path under a process group, the single-process multi-device entry
(abn_multi_*, through RCCL on the one GPU a test box has), per-window Philox ids, and the matrix-instruction proof."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _pedigree(kind, golden):
    from alphabeta_rs_amd import synthetic

    if kind == "c3":                      # N = 105: packed kernels with 16 lanes per chain (four accumulators each)
        ped, p0 = synthetic.c3_pedigree()
        return ped, p0
    if kind.startswith("generated"):      # N = 6: packed kernels with 8 lanes per chain
        return golden["generated"], golden["p0uu_generated"]
    if kind == "mid":                     # N = 200: packed kernels with 32 lanes per chain
        rng = np.random.default_rng(5)
        t0 = np.where(rng.random(200) < 0.3, rng.integers(0, 4, 200), 0)
        t1 = t0 + rng.integers(0, 10 - t0 + 1)
        t2 = t0 + rng.integers(0, 10 - t0 + 1)
        d = np.abs(rng.normal(0.01, 0.004, 200))
        return np.stack([t0, t1, t2, d], axis=1).astype(np.float64), 0.8
    return golden["pedigree"], 0.75       # N = 351: tree of 64


CANON = 0x10040   # the canonical tree of every LDS-resident pedigree: 64 accumulators, high lane bits first


@pytest.mark.parametrize("kind,tree", (("c3", CANON), ("generated", 1), ("generated_tree", CANON), ("mid", CANON),
                                       ("golden351", CANON)))
def test_results_do_not_depend_on_launch_size(abn, gpu_ctx, golden, oracle, kind, tree):
    """Auto options.  The same window fitted (1) alone — both phases on the four-wavefront speculative kernel —,
    (2) among 50 windows — phase B (2000 bootstraps) one wavefront per chain —, (3) among 200 windows — phase A one
    wavefront per chain, phase B packed / persistent —, and (4) among 800 windows — both phases packed — gives
    byte-identical models, residuals, bootstrap rows, iteration and evaluation counts: the reduction tree is the
    pedigree's, whichever kernel a launch picks.  Window 0 is also checked against the oracle."""
    ped, p0 = _pedigree(kind, golden)
    # the bundled six-row pedigree: serial row-order sums by default (tree code 1: pedigrees of up to 16 rows are summed in
    # the reference's order), the canonical tree with strict_order = -1 — both independent of the launch
    order = -1 if kind == "generated_tree" else 0
    assert abn.reduction_tree(ped[:, :3], abn.default_options(strict_order=order)) == tree
    n = ped.shape[0]
    S, B, seed = 8, 40, 31
    iters_a, iters_b = (300, 150) if n > 200 else (2000, 1000)
    o = abn.default_options(seed=seed, max_iters_start=iters_a, max_iters_boot=iters_b, strict_order=order)
    outs = []
    for W in (1, 50, 200, 800):
        rng = np.random.default_rng(17)
        D = np.tile(ped[:, 3], (W, 1))
        if W > 1:                          # other windows differ; window 0 is the same data everywhere
            D[1:] = np.abs(D[1:] * rng.uniform(0.7, 1.3, (W - 1, 1)))
        plan = abn.Plan(gpu_ctx, ped[:, :3], W, S, B, options=o)
        plan.set_windows(D, np.full(W, p0))
        plan.run()
        out = plan.download()
        plan.close()
        outs.append(out)
        assert np.all(out["info_a"]["lanes"] == tree) and np.all(out["info_b"]["lanes"] == tree)
    for out in outs[1:]:
        for k in ("models", "pred", "resid", "raw"):
            assert np.array_equal(out[k][0], outs[0][k][0], equal_nan=True), k
        for k in ("info_a", "info_b"):
            assert np.array_equal(out[k][0], outs[0][k][0]), k
    # 50, 200 and 800 windows share their leading windows as well
    assert np.array_equal(outs[1]["raw"], outs[2]["raw"][:50]) and np.array_equal(outs[2]["raw"], outs[3]["raw"][:200])
    s0 = abn.gen_start_simplices(seed, 0, S, ped[:, 3].max())
    fits = oracle.fit_batch(ped, p0, p0, 1.0, s0, iters_a, lanes=tree)
    k, model, pred, resid, _ = oracle.select_best(ped, p0, fits["best"])
    assert outs[0]["best_start"][0] == k and np.array_equal(outs[0]["models"][0], model)
    assert np.array_equal(outs[0]["info_a"]["evals"][0], fits["evals"])
    wraw, wres = oracle.boot_model(ped, model, pred, resid, p0, p0, 1.0, seed, 0, 0, B, max_iters=iters_b, lanes=tree)
    assert np.array_equal(outs[0]["raw"][0], wraw) and np.array_equal(outs[0]["info_b"]["evals"][0], wres["evals"])


def test_window_ids_give_every_window_its_own_streams(abn, gpu_ctx, golden, oracle):
    """abn_plan_set_window_ids: a plan over windows with NON-contiguous enumeration indices (a skipped window in the
    middle, a second topology group) draws each window's start simplices, jitter and bootstrap indices from the
    streams of ITS index — equal to single-window plans at that window_offset and to the oracle."""
    ped, p0 = golden["sparse"], golden["r_p0uu"]
    ids = np.array([0, 1, 3, 7], dtype=np.uint32)         # 2 was skipped; 4..6 belong to another topology
    W, S, B, seed = len(ids), 3, 6, 5
    rng = np.random.default_rng(3)
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.8, 1.2, (W, 1)))
    o = abn.default_options(seed=seed, max_iters_start=500, max_iters_boot=300)
    plan = abn.Plan(gpu_ctx, ped[:, :3], W, S, B, window_offset=0, options=o)
    plan.set_window_ids(ids)
    plan.set_windows(D, np.full(W, p0))
    plan.run()
    out = plan.download()
    plan.close()
    tree = int(out["info_b"]["lanes"][0, 0])
    for w, gid in enumerate(ids):
        single = abn.Plan(gpu_ctx, ped[:, :3], 1, S, B, window_offset=int(gid), options=o)
        single.set_windows(D[w:w + 1], np.array([p0]))
        single.run()
        so = single.download()
        single.close()
        assert np.array_equal(so["raw"][0], out["raw"][w]) and np.array_equal(so["models"][0], out["models"][w])
        pw = np.concatenate([ped[:, :3], D[w][:, None]], axis=1)
        s0 = abn.gen_start_simplices(seed, int(gid), S, D[w].max())
        fits = oracle.fit_batch(pw, p0, p0, 1.0, s0, 500, lanes=tree)
        k, model, pred, resid, _ = oracle.select_best(pw, p0, fits["best"])
        wraw, _ = oracle.boot_model(pw, model, pred, resid, p0, p0, 1.0, seed, int(gid), 0, B, max_iters=300, lanes=tree)
        assert np.array_equal(out["raw"][w], wraw)
    assert not np.array_equal(out["raw"][2], out["raw"][3])


@pytest.mark.parametrize("force", ("0", "1", "2"))
def test_multi_device_entry_matches_single_plan(abn, golden, force):
    """abn_multi_* on the one GPU of the test box: equal to a plain plan; with ABN_MULTI_FORCE_RCCL the table goes
    through RCCL (1: in-place ncclAllGather, 2: per-block ncclBroadcast) — librccl is bound at run time, a
    communicator is created, the collective runs on the device's stream behind the kernels."""
    ped, p0 = golden["sparse"], golden["r_p0uu"]
    W, S, B = 3, 4, 10
    rng = np.random.default_rng(9)
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.8, 1.2, (W, 1)))
    code = f"""
import sys, numpy as np
sys.path.insert(0, {str(ROOT)!r})
import alphabeta_rs_amd as A
ped = np.load(sys.argv[1]); D = np.load(sys.argv[2])
o = A.default_options(seed=11, max_iters_start=600, max_iters_boot=300)
assert A.rccl_available()
m = A.MultiPlan([0], ped[:, :3], {W}, {S}, {B}, options=o)
assert m.shard(0) == dict(window_offset=0, n_windows={W}, boot_offset=0, n_boot={B})
m.set_windows(D, np.full({W}, {p0!r}))
m.run(); m.run()
got = m.download(); cnt = m.counters(); m.close()
ctx = A.Context(0)
p = A.Plan(ctx, ped[:, :3], {W}, {S}, {B}, options=o)
p.set_windows(D, np.full({W}, {p0!r})); p.run()
want = p.download(); wc = p.counters(); p.close(); ctx.close()
for k in ("models", "pred", "resid", "raw", "info_a", "info_b", "best_start"):
    assert np.array_equal(got[k], want[k]), k
assert cnt == wc and np.isfinite(got["raw"]).all()
try:
    A.MultiPlan([0, 0], ped[:, :3], {W}, {S}, {B}, options=o)
    raise SystemExit("duplicate devices accepted")
except A.AbnError as e:
    assert e.status == 1
print("ok")
"""
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        np.save(f"{td}/ped.npy", ped)
        np.save(f"{td}/D.npy", D)
        r = subprocess.run([sys.executable, "-c", code, f"{td}/ped.npy", f"{td}/D.npy"], capture_output=True, text=True,
                           env=dict(os.environ, ABN_MULTI_FORCE_RCCL=force), timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_SHARD_WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, port, W, B = (int(x) for x in sys.argv[2:7])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
import torch, torch.distributed as dist
import alphabeta_rs_amd as A
from alphabeta_rs_amd import distributed as D_
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
ped = np.load(sys.argv[7]); D = np.load(sys.argv[8]); p0 = np.load(sys.argv[9])
S, seed = 4, 23
o = A.default_options(seed=seed, max_iters_start=600, max_iters_boot=300)      # AUTO lanes
ctx = A.Context(0, stream=torch.cuda.current_stream().cuda_stream)
full, shard = D_.run_sharded(D_.hip_shard_runner(ctx, ped[:, :3], D, p0, S, options=o), W, B, device="cpu")
if rank == 0:
    plan = A.Plan(ctx, ped[:, :3], W, S, B, options=o)
    plan.set_windows(D, p0); plan.run()
    want = plan.download()["raw"]; plan.close()
    assert full.shape == (W, B, 7) and np.array_equal(full.numpy(), want), "sharded table differs from the unsharded plan"
    print("ok", shard.mode)
dist.barrier(); dist.destroy_process_group(); ctx.close()
"""


@pytest.mark.parametrize("world,W,B,mode", ((2, 5, 24, "windows"), (2, 1, 700, "bootstraps"), (3, 2, 1500, "bootstraps")))
def test_hip_shard_runner_under_a_process_group(abn, golden, tmp_path, world, W, B, mode):
    """run_sharded(hip_shard_runner(...)) — the path bench.py --gpus N takes — with `world` gloo ranks sharing the one
    GPU: the gathered table equals the unsharded plan bit for bit under AUTO options.  The bootstrap-sharded cases
    put the shards (350 / 500 chains: speculative kernel) and the whole job (700 / 3000 chains: packed kernel) on
    different kernels — the sharding that changed results in round 1."""
    from alphabeta_rs_amd import synthetic

    ped, p0 = synthetic.c3_pedigree()
    rng = np.random.default_rng(2)
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.8, 1.2, (W, 1)))
    np.save(tmp_path / "ped.npy", ped)
    np.save(tmp_path / "D.npy", D)
    np.save(tmp_path / "p0.npy", np.full(W, p0))
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-c", _SHARD_WORKER, str(ROOT), str(r), str(world), str(port), str(W),
                               str(B), str(tmp_path / "ped.npy"), str(tmp_path / "D.npy"), str(tmp_path / "p0.npy")],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    assert f"ok {mode}" in outs[0][0]


def test_matrix_instruction_is_the_reference_fma_chain(tmp_path):
    """The proof the matrix-instruction power tables stand on (scripts/mfma_f64_probe.hip): v_mfma_f64_4x4x4 equals
    fma(a2,b2, fma(a1,b1, fma(a0,b0, 0))) bit for bit — random operands over 40 binades, 3x3 blocks padded with
    zeros, NaN / infinities / signed zeros / denormals / overflow."""
    exe = tmp_path / "mfma_probe"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", str(exe),
                        str(ROOT / "scripts" / "mfma_f64_probe.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert "3x3 padded: mismatches vs fma(a2,b2,fma(a1,b1,fma(a0,b0,0))): 0\n" in out, out
    assert "mismatches vs k-ascending fma chain 0," in out, out
    assert "special values (NaN compared as NaN): mismatches 0\n" in out, out
    assert "vs k-descending 0" not in out           # the order matters: the descending chain must NOT match
