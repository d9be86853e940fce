"""Multi-GPU sharding of the (window, bootstrap) space — one process per GPU, `torch.distributed`
(backend "nccl" is RCCL on ROCm; "gloo" for the CPU rehearsal in tests).

Every fit is independent, so the data path has NO collective: ranks differ only in the
`window_offset` / `boot_offset` they hand to `abn_plan_create`, and because every random draw is a pure
function of the GLOBAL (window, bootstrap) index the union of the shards is bit-identical to a
single-GPU run.  The one exchange step is the all-gather of the bootstrap tables (56 B per fit).

Sharding rule: with at least as many windows as ranks, windows are dealt in contiguous blocks (phase A's
arg-min stays local).  With fewer windows than ranks every rank repeats the cheap phase A for all
windows (same inputs -> same bits everywhere) and takes a contiguous slice of the bootstraps.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable

import numpy as np


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """contiguous balanced split: the first (total % world) ranks get one extra item"""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


@dataclass(frozen=True)
class Shard:
    mode: str            # "windows" or "bootstraps"
    window_offset: int
    n_windows: int
    boot_offset: int
    n_boot: int


def plan_shard(n_windows: int, n_boot: int, world: int, rank: int) -> Shard:
    if n_windows >= world:
        w0, wn = shard_range(n_windows, world, rank)
        return Shard("windows", w0, wn, 0, n_boot)
    b0, bn = shard_range(n_boot, world, rank)
    return Shard("bootstraps", 0, n_windows, b0, bn)


def gather_tables(local, n_windows: int, n_boot: int, group=None):
    """all-gather of the per-rank bootstrap tables -> the full [W, B, 7] table on every rank.
    `local` is a torch tensor [Wl, Bl, 7] on the rank's device (f64)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    shards = [plan_shard(n_windows, n_boot, world, r) for r in range(world)]
    me = shards[dist.get_rank(group)]
    assert tuple(local.shape) == (me.n_windows, me.n_boot, 7), (tuple(local.shape), me)
    sizes = [s.n_windows * s.n_boot * 7 for s in shards]
    flat = local.contiguous().view(-1)
    if len(set(sizes)) == 1:
        out = torch.empty(world * sizes[0], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, flat, group=group)
        parts = [out[r * sizes[0]:(r + 1) * sizes[0]] for r in range(world)]
    else:  # ragged shards: pad to the largest
        mx = max(sizes)
        pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
        pad[: flat.numel()] = flat
        out = torch.empty(world * mx, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad, group=group)
        parts = [out[r * mx: r * mx + sizes[r]] for r in range(world)]
    full = torch.empty((n_windows, n_boot, 7), dtype=local.dtype, device=local.device)
    for s, p in zip(shards, parts):
        if s.n_windows * s.n_boot == 0:
            continue
        full[s.window_offset:s.window_offset + s.n_windows, s.boot_offset:s.boot_offset + s.n_boot] = p.view(
            s.n_windows, s.n_boot, 7)
    return full


def run_sharded(compute_shard: Callable[[Shard], "np.ndarray"], n_windows: int, n_boot: int, device=None, group=None):
    """Runs `compute_shard(shard) -> [Wl, Bl, 7]` on this rank's shard and gathers the full table.
    The HIP path passes `hip_shard_runner(...)`; the CPU test passes an oracle-backed runner."""
    import torch
    import torch.distributed as dist

    shard = plan_shard(n_windows, n_boot, dist.get_world_size(group), dist.get_rank(group))
    local = compute_shard(shard)
    if not isinstance(local, torch.Tensor):
        local = torch.as_tensor(np.ascontiguousarray(local), dtype=torch.float64)
    if device is not None:
        local = local.to(device)
    # What went wrong on this rank travels NEXT TO the gather, in a collective every rank enters: [windows without a finite
    # start, 1 if the shard's run failed].  A rank that raised before the gather would leave the others waiting in the
    # collective until its watchdog fires; so a runner records its failure (attributes `failed_windows`, `error`), still
    # returns a table of the right shape, and every rank raises here together.
    local_err = getattr(compute_shard, "error", None)
    flags = torch.tensor([int(getattr(compute_shard, "failed_windows", 0)), 1 if local_err is not None else 0],
                         dtype=torch.int64, device=local.device)
    dist.all_reduce(flags, op=dist.ReduceOp.SUM, group=group)
    full = gather_tables(local, n_windows, n_boot, group=group)
    from . import AbnError

    if int(flags[1].item()):
        raise AbnError(getattr(local_err, "status", 4),
                       f"{int(flags[1].item())} rank(s) failed in their shard" + (f"; this rank: {local_err}" if local_err else ""))
    # A window without a finite start has NaN rows (abn_plan_download: ABN_ERR_NO_FINITE_FIT; the reference panics,
    # src/ab_neutral.rs:28,100).  Primary signal: the exchanged count; the NaN scan of the gathered table names the windows
    # (and catches a runner that does not report).
    dead = torch.isnan(full[:, :, 0]).all(dim=1) if n_windows * n_boot else torch.zeros(n_windows, dtype=torch.bool)
    if int(flags[0].item()) or bool(dead.any().item()):
        ws = [int(w) for w in torch.nonzero(dead).flatten().tolist()]
        raise AbnError(5, f"{max(len(ws), int(flags[0].item()))} window(s) have no finite start (NaN rows): "
                          f"{ws[:8]}{'...' if len(ws) > 8 else ''}")
    return full, shard


def hip_shard_runner(ctx, generations, d_obs, p0uu, n_starts, options=None):
    """compute_shard for the MI355X path: one abn_plan per rank, bootstrap table written straight into a
    torch tensor (no copy before the RCCL gather).  The runner itself never raises an AbnError: it records it
    (`run.error`, `run.failed_windows`) for run_sharded, which raises on every rank after the collectives; a caller that
    uses the runner on its own must look at those two attributes."""
    import torch

    from . import Plan

    d_obs = np.asarray(d_obs, dtype=np.float64)
    p0uu = np.asarray(p0uu, dtype=np.float64)

    def run(shard: Shard):
        w0, wn = shard.window_offset, shard.n_windows
        raw = torch.empty((wn, shard.n_boot, 7), dtype=torch.float64, device=f"cuda:{ctx.device}")
        if wn == 0 or shard.n_boot == 0:
            return raw
        from . import AbnError

        plan = None
        try:
            plan = Plan(ctx, generations, wn, n_starts, shard.n_boot, window_offset=w0, boot_offset=shard.boot_offset,
                        options=options)
            plan.bind_raw(raw.data_ptr())
            plan.set_windows(d_obs[w0:w0 + wn], p0uu[w0:w0 + wn])
            plan.run()
            plan.sync()                                   # ABN_ERR_HIP if a persistent launch lost a chain
            run.failed_windows = plan.failed_windows()    # run_sharded exchanges the count and raises on EVERY rank
        except AbnError as e:                             # (with bootstraps sharded every rank sees the same windows:
            run.error = e                                 # the sum then counts them once per rank — it only has to be > 0)
            raw.fill_(float("nan"))                       # the table still has its shape: the gather must be entered
        finally:
            if plan is not None:
                plan.close()
        return raw

    run.failed_windows = 0
    run.error = None
    return run
