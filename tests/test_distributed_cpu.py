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
    import oracle as O

    out = np.empty((wn, bn, 7))
    for i in range(wn):
        w = w0 + i
        ped = np.concatenate([ped_gens, D[w][:, None]], axis=1)
        s0 = np.stack([O.start_simplex(seed, w, s, D[w].max()) for s in range(S)])
        fits = O.fit_batch(ped, p0[w], p0[w], 1.0, s0, 10000, lanes=16, threads=1)
        k, model, pred, resid, _ = O.select_best(ped, p0[w], fits["best"])
        raw, _ = O.boot_model(ped, model, pred, resid, p0[w], p0[w], 1.0, seed, w, b0, bn, lanes=16, threads=1)
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


@pytest.mark.parametrize("W,B,mode", [(3, 5, "windows"), (1, 7, "bootstraps"), (2, 4, "windows")])
def test_two_rank_gloo_shard_and_gather(W, B, mode):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, got_mode, shape = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and got_mode == mode and shape == (W, B, 7)


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
