"""Exhaustive interleaving model of the WHOLE time-slicing protocol of the persistent fit kernel (CPU only).

`tests/test_fifo_protocol_model.py` models the credit counter alone.  This one models everything a wavefront of
`abn_fit_refill_kernel` (csrc/abn_fit_refill.hpp, "finished fits" section) does to shared memory, one atomic operation per
step, with the wavefront's lock-step order kept (a phase is over for all its groups before the next one starts, and a
wavefront that waits for an entry stalls ALL its groups):

  quantum boundary   one reading of (queue, credits, tail) -> park in the FIFO / park on the tail list / run on
  tail check         (queue empty, no credit, at most tail_cap chains unfinished) -> tail mode       [taken or not: both]
  finished           atomicAdd(finished)
  tail park          susp_list[atomicAdd(susp_count)] = chain
  FIFO park          pos = atomicAdd(tail); pk[pos] = chain; atomicAdd(avail)          (three phases, in this order)
  next chain         f = atomicAdd(queue); else claim a credit (atomicSub / give back / look again), h = atomicAdd(head),
                     wait for pk[h]

Every interleaving of the wavefronts' steps is explored (lanes of one phase: in lane order, except the claim loop whose lanes
run in any order).  Properties, checked in every reachable state:
  * no deadlock: a wavefront that waits for an entry always has a wavefront that can still publish it;
  * a ticket never runs past the FIFO's capacity (the park verdict's slack covers the tickets in flight);
  * no chain is ever taken up twice;
  * when all wavefronts have left, every chain is finished or on the tail list, exactly once, the counters add up to
    what `verify_persistent` (csrc/abn_api.hip) expects, no credit and no entry is left.
The protocol WITHOUT the look-again of round 4 must fail the last property (the model finds the stranded chain)."""
import pytest

IDLE, RUN, FINISHED, PARK_FIFO, PARK_TAIL = range(5)
P_TAILCHK, P_STEP, P_FIN, P_TAILPARK, P_TICKET, P_STORE, P_CREDIT, P_DRAW, P_CLAIM, P_HEAD, P_SPIN, P_ASSIGN, P_EXIT = range(13)
C_UNSTARTED, C_RUNNING, C_PARKED, C_DONE, C_TAIL = range(5)
DONE_STATES = (FINISHED, PARK_FIFO, PARK_TAIL)


class Violation(Exception):
    pass


def _applies(pc, g):
    chain, st, fd, nxt, tmp, cst = g
    if pc == P_STEP:
        return st == RUN
    if pc == P_FIN:
        return st == FINISHED
    if pc == P_TAILPARK:
        return st == PARK_TAIL
    if pc in (P_TICKET, P_STORE, P_CREDIT):
        return st == PARK_FIFO
    if pc == P_DRAW:
        return st in DONE_STATES and not fd
    if pc == P_CLAIM:
        return st in DONE_STATES and nxt < 0 and cst in (0, 1, 2)
    if pc in (P_HEAD, P_SPIN):
        return st in DONE_STATES and cst == 3 and nxt < 0
    return False


NEXT = {P_STEP: P_FIN, P_FIN: P_TAILPARK, P_TAILPARK: P_TICKET, P_TICKET: P_STORE, P_STORE: P_CREDIT, P_CREDIT: P_DRAW,
        P_DRAW: P_CLAIM, P_CLAIM: P_HEAD, P_HEAD: P_SPIN, P_SPIN: P_ASSIGN}
# a mutant for the model's own sake: the entries stored only AFTER the wavefront has looked for its next chains
NEXT_LATE_STORE = {**NEXT, P_TICKET: P_DRAW, P_SPIN: P_STORE, P_CREDIT: P_ASSIGN}


def _normalise(wf, cs, order=NEXT):
    """Advance (pc, idx) over everything that touches no shared word; -> (wavefront, chain states)."""
    pc, idx, tail_mode, wave_fd, groups = wf
    while True:
        if pc == P_EXIT or pc == P_TAILCHK:
            return (pc, 0, tail_mode, wave_fd, groups), cs
        if pc == P_ASSIGN:
            new, cs = [], list(cs)
            for chain, st, fd, nxt, tmp, cst in groups:
                wave_fd = wave_fd or fd
                if st in DONE_STATES:
                    if nxt >= 0:
                        if cs[nxt] not in (C_UNSTARTED, C_PARKED):
                            raise Violation(f"chain {nxt} taken up in state {cs[nxt]}")
                        cs[nxt] = C_RUNNING
                        new.append((nxt, RUN, fd, -1, -1, 0))
                    else:
                        new.append((-1, IDLE, fd, -1, -1, 0))
                else:
                    new.append((chain, st, fd, -1, -1, 0))
            groups, cs = tuple(new), tuple(cs)
            pc = P_EXIT if all(g[1] == IDLE for g in groups) else P_TAILCHK
            continue
        if pc == P_CLAIM:
            if any(_applies(pc, g) for g in groups):
                return (pc, 0, tail_mode, wave_fd, groups), cs
            pc, idx = order[P_CLAIM], 0
            continue
        while idx < len(groups) and not _applies(pc, groups[idx]):
            idx += 1
        if idx < len(groups):
            return (pc, idx, tail_mode, wave_fd, groups), cs
        if pc == P_STEP and not any(g[1] in DONE_STATES for g in groups):
            pc = P_EXIT if all(g[1] == IDLE for g in groups) else P_TAILCHK   # `while (__ballot(st != ST_IDLE))`
        else:
            pc = order[pc]
        idx = 0


def explore(n_waves, ng, work, recheck=True, tail_cap=0, cap=64, limit=3_000_000, order=NEXT, count_stranded=False,
            shards=1):
    """work[c] = quanta chain c runs for; wavefront w parks in and claims from FIFO shard w % shards (the kernel's
    blockIdx & (kParkShards - 1)); queue, finished count and tail list are shared.  -> dict of counts; raises Violation on a
    broken property."""
    total, base = len(work), n_waves * ng
    assert total >= base
    wf0 = [(P_TAILCHK, 0, False, False, tuple((w * ng + j, RUN, False, -1, -1, 0) for j in range(ng))) for w in range(n_waves)]
    cs0 = tuple(C_RUNNING if c < base else C_UNSTARTED for c in range(total))
    # shared words: queue, tail, head, avail, finished, the FIFO's entries, the tail list
    start = (0, (0,) * shards, (0,) * shards, (0,) * shards, 0, ((-1,) * cap,) * shards, (), tuple(work), cs0, tuple(wf0))
    seen, stack = {start}, [start]
    out = dict(states=0, terminals=0, stranded=0, tail_parks=0, fifo_parks=0, waits=0)
    while stack:
        Q, Ts, Hs, As, FIN, pks, susp, wk, cs, wfs = stack.pop()
        out["states"] += 1
        if out["states"] > limit:
            raise RuntimeError("state limit")
        if all(wf[0] == P_EXIT for wf in wfs):
            out["terminals"] += 1
            left = [c for c in range(total) if cs[c] not in (C_DONE, C_TAIL)]
            if left:
                out["stranded"] += 1
                if recheck and not count_stranded:
                    raise Violation(f"chains {left} neither finished nor handed over: states {[cs[c] for c in left]}")
                continue
            n_done, n_tail = cs.count(C_DONE), cs.count(C_TAIL)
            if FIN != n_done or sorted(susp) != [c for c in range(total) if cs[c] == C_TAIL] or FIN + len(susp) != total:
                raise Violation(f"counters: finished {FIN} of {n_done}, tail list {susp} of {n_tail}")
            if any(As) or Ts != Hs:
                raise Violation(f"left over: {As} credits, tails {Ts}, heads {Hs}")
            out["tail_parks"] = max(out["tail_parks"], n_tail)
            continue
        succ = []
        for w, wf in enumerate(wfs):
            pc, idx, tail_mode, wave_fd, groups = wf
            if pc == P_EXIT:
                continue
            sh = w % shards
            T, H, A, pk = Ts[sh], Hs[sh], As[sh], pks[sh]

            def put(tup, v):
                return tup[:sh] + (v,) + tup[sh + 1:]

            def emit(groups_, pc_=pc, idx_=idx, tm=tail_mode, Q=Q, T=T, H=H, A=A, FIN=FIN, pk=pk, susp=susp, wk=wk, cs=cs):
                nwf, ncs = _normalise((pc_, idx_, tm, wave_fd, groups_), cs, order)
                succ.append((Q, put(Ts, T), put(Hs, H), put(As, A), FIN, put(pks, pk), susp, wk, ncs,
                             wfs[:w] + (nwf,) + wfs[w + 1:]))

            def with_group(i, g):
                return groups[:i] + (g,) + groups[i + 1:]

            if pc == P_TAILCHK:
                emit(groups, pc_=P_STEP, idx_=0)   # not this step (the kernel looks every 64th)
                if tail_cap > 0 and wave_fd and not tail_mode and base + Q >= total and A <= 0 and total - FIN <= tail_cap:
                    emit(groups, pc_=P_STEP, idx_=0, tm=True)
                continue
            if pc == P_CLAIM:
                for i, g in enumerate(groups):
                    if not _applies(pc, g):
                        continue
                    chain, st, fd, nxt, tmp, cst = g
                    if cst == 0:      # atomicSub(avail, 1) > 0
                        emit(with_group(i, (chain, st, fd, nxt, tmp, 3 if A > 0 else 1)), A=A - 1)
                    elif cst == 1:    # give the credit back
                        emit(with_group(i, (chain, st, fd, nxt, tmp, 2 if recheck else 4)), A=A + 1)
                    else:             # look again
                        emit(with_group(i, (chain, st, fd, nxt, tmp, 0 if A > 0 else 4)))
                continue
            chain, st, fd, nxt, tmp, cst = groups[idx]
            if pc == P_STEP:
                nwk = wk[:chain] + (wk[chain] - 1,) + wk[chain + 1:]
                if nwk[chain] == 0:
                    nst = FINISHED
                elif (base + Q < total or A > 0) and T + base // shards + ng < cap:
                    nst = PARK_FIFO
                elif tail_mode:
                    nst = PARK_TAIL
                else:
                    nst = RUN
                emit(with_group(idx, (chain, nst, fd, nxt, tmp, cst)), idx_=idx + 1, wk=nwk)
            elif pc == P_FIN:
                ncs = cs[:chain] + (C_DONE,) + cs[chain + 1:]
                emit(groups, idx_=idx + 1, FIN=FIN + 1, cs=ncs)
            elif pc == P_TAILPARK:
                ncs = cs[:chain] + (C_TAIL,) + cs[chain + 1:]
                emit(groups, idx_=idx + 1, susp=susp + (chain,), cs=ncs)
            elif pc == P_TICKET:
                if T >= cap:
                    raise Violation(f"ticket {T} beyond the FIFO's {cap} entries")
                emit(with_group(idx, (chain, st, fd, nxt, T, cst)), idx_=idx + 1, T=T + 1)
            elif pc == P_STORE:
                ncs = cs[:chain] + (C_PARKED,) + cs[chain + 1:]
                emit(groups, idx_=idx + 1, pk=pk[:tmp] + (chain,) + pk[tmp + 1:], cs=ncs)
                out["fifo_parks"] += 1
            elif pc == P_CREDIT:
                emit(with_group(idx, (chain, st, fd, nxt, -1, cst)), idx_=idx + 1, A=A + 1)
            elif pc == P_DRAW:
                f = base + Q
                g = (chain, st, fd, f, tmp, cst) if f < total else (chain, st, True, nxt, tmp, cst)
                emit(with_group(idx, g), idx_=idx + 1, Q=Q + 1)
            elif pc == P_HEAD:
                emit(with_group(idx, (chain, st, fd, nxt, H, cst)), idx_=idx + 1, H=H + 1)
            elif pc == P_SPIN:
                if pk[tmp] >= 0:      # else: this wavefront waits (all its groups with it)
                    emit(with_group(idx, (chain, st, fd, pk[tmp], -1, cst)), idx_=idx + 1)
                else:
                    out["waits"] += 1
        if not succ:
            raise Violation(f"deadlock: {[(wf[0], wf[1]) for wf in wfs]}, entries {pks}, heads {Hs}, credits {As}")
        for s in succ:
            if s not in seen:
                seen.add(s)
                stack.append(s)
    return out


@pytest.mark.parametrize("n_waves,ng,work,tail_cap", [
    (2, 1, (2, 2, 1), 0),               # one chain behind two running ones: the credit race's shape
    (2, 1, (3, 2, 2, 1), 0),
    (2, 2, (2, 1, 2, 2, 1), 0),         # two groups a wavefront: the lock-step phases
    (2, 2, (2, 2, 1, 2, 1, 1), 0),
    (3, 1, (2, 3, 2, 1), 0),            # 466 022 states
    (2, 1, (3, 3, 1), 2),               # with the tail hand-over
    (2, 1, (3, 3, 2, 1), 2),
    (2, 2, (3, 2, 3, 2, 1), 2),         # 623 930 states
])
def test_whole_protocol_loses_no_chain_and_never_deadlocks(n_waves, ng, work, tail_cap):
    out = explore(n_waves, ng, work, recheck=True, tail_cap=tail_cap)
    assert out["terminals"] > 0 and out["stranded"] == 0, out
    assert out["fifo_parks"] > 0, out                       # the model does reach the FIFO
    if tail_cap:
        assert out["tail_parks"] > 0, out                   # ... and the tail list


@pytest.mark.parametrize("n_waves,ng,work,tail_cap", [
    (2, 1, (3, 2, 2, 1), 0),            # a shard each: a parked chain can only be taken up by its own wavefront
    (3, 1, (2, 3, 2, 1), 0),            # two wavefronts share shard 0, the third has its own
    (2, 2, (3, 2, 3, 2, 1), 2),
    (3, 1, (3, 3, 2, 1), 3),
])
def test_two_fifo_shards_lose_no_chain_either(n_waves, ng, work, tail_cap):
    out = explore(n_waves, ng, work, recheck=True, tail_cap=tail_cap, shards=2)
    assert out["terminals"] > 0 and out["stranded"] == 0 and out["fifo_parks"] > 0, out


def test_a_waiting_wavefront_is_a_reachable_state():
    # tickets are handed out before the entries are stored: a claimer does reach the wait for an entry of another
    # wavefront (the case the bounded spin of the kernel is for), and it always ends (no deadlock above)
    out = explore(2, 2, (2, 1, 2, 2, 1), recheck=True)
    assert out["waits"] > 0, out


def test_park_verdict_keeps_tickets_inside_a_tight_fifo():
    # capacity at the verdict's edge: parks still happen and no ticket runs past the entries
    out = explore(2, 2, (3, 3, 3, 3, 2, 1), recheck=True, cap=8)
    assert out["fifo_parks"] > 0 and out["stranded"] == 0, out


def test_entries_stored_after_the_look_for_a_next_chain_deadlock():
    # why "all of this wavefront's entries are out before any of its groups looks for one": with the order turned round a
    # wavefront can hold the ticket of the very entry it waits for (and a chain parked after the look is nobody's)
    with pytest.raises(Violation, match="deadlock"):
        explore(2, 1, (2, 2, 1), recheck=True, order=NEXT_LATE_STORE, count_stranded=True)
    with pytest.raises(Violation, match="neither finished nor handed over"):
        explore(2, 1, (2, 2, 1), recheck=True, order=NEXT_LATE_STORE)


def test_whole_model_reproduces_the_stranded_chain_of_the_old_protocol():
    out = explore(2, 1, (2, 2, 1), recheck=False)
    assert out["stranded"] > 0, out
