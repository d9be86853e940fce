"""The N > 1 path on CPU: two gloo ranks shard the (window, bootstrap) space exactly as the GPU ranks
do, compute their shard (here with the oracle — test infrastructure — standing in for the HIP plan) and
all-gather the bootstrap tables; the result must be bit-identical to the unsharded run."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_tables(ped_gens, D, p0, S, seed, w0, wn, b0, bn):
    """the oracle standing in for one rank's plan.  The reduction tree is what the product reports for this pedigree
    under AUTO options (abn_reduction_tree: host arithmetic on the generations alone, so it cannot depend on the
    size of a rank's shard) — the GPU counterpart is tests/test_gpu_parity.py::test_results_do_not_depend_on_launch_size"""
    import alphabeta_rs_amd as A
    import oracle as O

    lanes = A.reduction_tree(ped_gens)
    out = np.empty((wn, bn, 7))
    for i in range(wn):
        w = w0 + i
        ped = np.concatenate([ped_gens, D[w][:, None]], axis=1)
        s0 = np.stack([O.start_simplex(seed, w, s, D[w].max()) for s in range(S)])
        fits = O.fit_batch(ped, p0[w], p0[w], 1.0, s0, 10000, lanes=lanes, threads=1)
        k, model, pred, resid, _ = O.select_best(ped, p0[w], fits["best"])
        raw, _ = O.boot_model(ped, model, pred, resid, p0[w], p0[w], 1.0, seed, w, b0, bn, lanes=lanes, threads=1)
        out[i] = raw
    return out


def _worker(rank, world, port, W, B, q):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist

    import oracle as O
    from alphabeta_rs_amd import distributed as D_

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ped = O.load_pedigree(ROOT / "tests" / "golden" / "pedigree_generated.txt")
    rng = np.random.default_rng(4)
    D = np.abs(ped[:, 3][None, :] * rng.uniform(0.7, 1.3, (W, 1)))
    p0 = rng.uniform(0.6, 0.8, W)
    S, seed = 3, 99

    def compute(shard):
        return _oracle_tables(ped[:, :3], D, p0, S, seed, shard.window_offset, shard.n_windows, shard.boot_offset,
                              shard.n_boot)

    full, shard = D_.run_sharded(compute, W, B)
    if rank == 0:
        want = _oracle_tables(ped[:, :3], D, p0, S, seed, 0, W, 0, B)
        q.put((np.array_equal(full.numpy(), want), shard.mode, tuple(full.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,B,mode", [(2, 3, 5, "windows"), (2, 1, 7, "bootstraps"), (2, 2, 4, "windows"),
                                            (8, 9, 3, "windows"), (8, 2, 11, "bootstraps")])
def test_gloo_shard_and_gather_matches_unsharded(world, W, B, mode):
    """world 2 and 8 (ragged shards included) against the unsharded (world 1) table, bit for bit"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, got_mode, shape = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok and got_mode == mode and shape == (W, B, 7)


def _failing_worker(rank, world, port, q, kind="nan_window"):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist

    import alphabeta_rs_amd as A
    from alphabeta_rs_amd import distributed as D_

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def compute(shard):  # window 2 (on the last rank) has no finite start: its rows are NaN, as the plan writes them
        out = np.full((shard.n_windows, shard.n_boot, 7), float(shard.window_offset + 1))
        for i in range(shard.n_windows):
            if shard.window_offset + i == 2:
                out[i] = np.nan
        return out

    if kind == "shard_error":  # the last rank's runner failed (what hip_shard_runner records when abn_plan_sync reports a
        def compute_err(shard):  # lost chain): finite rows, an `error` attribute — nothing the NaN scan could see
            if rank == world - 1:
                compute_err.error = A.AbnError(4, "persistent fit launch of phase B finished 23 999 of 24 000 chains")
            return np.ones((shard.n_windows, shard.n_boot, 7))

        compute_err.error = None
        compute = compute_err
    elif kind == "count_only":  # the runner reports a failed window the table does not show (a reporting runner is believed)
        def compute_cnt(shard):
            compute_cnt.failed_windows = 1 if rank == 0 else 0
            return np.ones((shard.n_windows, shard.n_boot, 7))

        compute = compute_cnt
    try:
        D_.run_sharded(compute, 3, 4)
        q.put((rank, "no error"))
    except A.AbnError as e:
        q.put((rank, f"AbnError {e.status}"))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failed_window_raises_on_every_rank():
    """a rank that raised before the gather left the others blocked in the collective.  Now the NaN rows
    travel through the gather and every rank raises ABN_ERR_NO_FINITE_FIT together."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, "AbnError 5"), (1, "AbnError 5")]


@pytest.mark.parametrize("kind,status", [("shard_error", 4), ("count_only", 5)])
def test_a_failed_shard_raises_on_every_rank(kind, status):
    """failures are exchanged explicitly (an all-reduce next to the gather), not inferred from NaN rows: a rank
    whose plan reported ABN_ERR_HIP at abn_plan_sync, or a failed-window count, makes EVERY rank raise."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, f"AbnError {status}"), (1, f"AbnError {status}")]


def test_reduction_tree_is_a_function_of_the_pedigree_only(abn):
    """abn_reduction_tree takes generations and options — no fit counts: the tree cannot change with sharding.
    Auto: the canonical 64-accumulator tree (0x10040) for every LDS-resident pedigree, whatever lane count its packed
    kernels use; streamed pedigrees 64 | 3 << 8; an explicit lanes_per_chain is the (legacy) tree itself."""
    import oracle as O
    from alphabeta_rs_amd import synthetic

    canon = 0x10040
    assert abn.reduction_tree(synthetic.c3_pedigree()[0][:, :3]) == canon
    bundled = O.load_pedigree(ROOT / "tests" / "golden" / "pedigree_generated.txt")[:, :3]
    assert abn.reduction_tree(bundled) == 1          # up to 16 rows: the reference's serial order by default (free there)
    assert abn.reduction_tree(bundled, abn.default_options(strict_order=-1)) == canon
    assert abn.reduction_tree(synthetic.c3_pedigree()[0][:17, :3]) == canon    # 17 rows: the tree
    assert abn.reduction_tree(O.load_pedigree(ROOT / "tests" / "golden" / "pedigree.txt")[:, :3]) == canon
    assert abn.reduction_tree(synthetic.c3_pedigree()[0][:, :3], abn.default_options(lanes_per_chain=32)) == 32
    assert abn.reduction_tree(synthetic.c3_pedigree()[0][:, :3], abn.default_options(lanes_per_chain=16)) == 16
    big = np.zeros((3000, 3))
    big[:, 1:] = 1.0
    assert abn.reduction_tree(big) == (64 | (3 << 8))


def test_bench_refuses_a_gpu_count_it_cannot_run():
    """`python bench.py --gpus 2` on a box without two GPUs exits non-zero and prints no JSON line (round 1 printed
    n_gpus = 1); a WORLD_SIZE that contradicts --gpus is refused as well."""
    import subprocess

    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")},
                       timeout=300)
    import torch

    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "{" not in r.stdout
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0"), timeout=300)
    assert r.returncode != 0 and "{" not in r.stdout and "WORLD_SIZE" in r.stderr


def test_shard_ranges_cover_everything():
    from alphabeta_rs_amd.distributed import plan_shard, shard_range

    for total in (0, 1, 7, 8, 200, 1001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, n = shard_range(total, world, r)
                seen += list(range(s, s + n))
            assert seen == list(range(total))
    s = plan_shard(200, 1000, 8, 3)
    assert (s.mode, s.window_offset, s.n_windows, s.boot_offset, s.n_boot) == ("windows", 75, 25, 0, 1000)
    s = plan_shard(1, 10000, 8, 7)
    assert (s.mode, s.n_windows, s.boot_offset, s.n_boot) == ("bootstraps", 1, 8750, 1250)


def test_single_process_entry_shards_as_the_process_per_gpu_path(abn):
    """abn_multi_* (one process, several GPUs) and distributed.py (one process per GPU) must cut a job the same way:
    abn_multi_plan_shard is host arithmetic, so the C++ rule is compared with the Python one here, without a device —
    windows in contiguous balanced blocks, bootstraps when there are fewer windows than devices, ragged remainders,
    more devices than items."""
    from alphabeta_rs_amd.distributed import plan_shard

    for W in (1, 2, 7, 8, 9, 25, 200, 301):
        for B in (1, 5, 1000, 10007):
            for n in (1, 2, 3, 4, 8, 16):
                covered_w, covered_b = [], []
                for r in range(n):
                    got = abn.multi_plan_shard(W, B, n, r)
                    want = plan_shard(W, B, n, r)
                    assert got == (want.window_offset, want.n_windows, want.boot_offset, want.n_boot), (W, B, n, r)
                    covered_w += list(range(got[0], got[0] + got[1])) if W >= n else []
                    covered_b += list(range(got[2], got[2] + got[3])) if W < n else []
                assert covered_w == (list(range(W)) if W >= n else [])
                assert covered_b == (list(range(B)) if W < n else [])
    with pytest.raises(abn.AbnError):
        abn.multi_plan_shard(4, 10, 2, 2)
    with pytest.raises(abn.AbnError):
        abn.multi_plan_shard(0, 10, 2, 0)
