"""Exhaustive interleaving model of the credit protocol of the time-slicing FIFO (csrc/abn_fit_refill.hpp; CPU only).

A group that parks a chain publishes its entry and then adds a credit (`atomicAdd(avail, 1)`); a group that looks for a
parked chain claims a credit (`atomicSub(avail, 1) > 0`) and gives it back if there was none.  Between a failed claim and its
give-back the counter is one too low, so another claim — the parking group's own, right after its park — can fail although
a chain IS parked.  Round 4's plan fuzz found the consequence on the GPU: all groups of a FIFO shard idle, one chain still
parked, `abn_plan_download` -> ABN_ERR_HIP "finished 13999 of 14000 chains" ("Hardening" (iv)).  The fix: whoever
gives a credit back reads the counter again and retries while it is positive.

The model: every atomic operation is one step; `parkers` groups each park one chain and then claim, `claimers` groups (they
just finished a fit) claim once; a group whose claim fails goes idle for good — the end of a launch, when nobody comes by
later.  Every interleaving is explored.  Property: when all groups are idle, no chain is left parked.  The OLD protocol must
violate it (the model reproduces the bug), the NEW one must not."""
import sys

import pytest

sys.setrecursionlimit(100000)


def explore(parkers, claimers, recheck):
    """-> (terminal states seen, terminal states with a chain left parked).  A group is (pc, tries):
    pc 0 = park (parkers only), 1 = claim: atomicSub, 2 = give back: atomicAdd, 3 = re-read, 4 = idle."""
    start = (0, 0, 0, tuple([(0, 0)] * parkers + [(1, 0)] * claimers))
    seen, stack = {start}, [start]
    terminals = stranded = 0
    max_tries = 4 * (parkers + claimers)          # far more than the other groups can cause
    while stack:
        avail, parked, taken, groups = stack.pop()
        if all(g[0] == 4 for g in groups):
            terminals += 1
            stranded += parked > taken
            continue
        for i, (pc, tries) in enumerate(groups):
            if pc == 4:
                continue
            a, p, t = avail, parked, taken
            if pc == 0:                           # publish the entry, then the credit
                a, p, nxt = a + 1, p + 1, (1, 0)
            elif pc == 1:                         # claim
                old = a
                a -= 1
                if old > 0:
                    t += 1                        # a credit is only ever there against a published entry
                    nxt = (4, 0)
                else:
                    nxt = (2, tries)
            elif pc == 2:                         # give the credit back
                a += 1
                nxt = (3, tries) if recheck else (4, 0)
            else:                                 # pc == 3: look again
                nxt = (1, tries + 1) if (a > 0 and tries < max_tries) else (4, 0)
            state = (a, p, t, groups[:i] + (nxt,) + groups[i + 1:])
            if state not in seen:
                seen.add(state)
                stack.append(state)
    return terminals, stranded


@pytest.mark.parametrize("parkers,claimers", [(1, 1), (1, 2), (1, 3), (1, 4), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2)])
def test_a_failed_claim_that_looks_again_strands_no_chain(parkers, claimers):
    terminals, stranded = explore(parkers, claimers, recheck=True)
    assert terminals > 0 and stranded == 0, (terminals, stranded)


def test_the_model_reproduces_the_race_of_the_old_protocol():
    # one group parks a chain and claims, another one's failed claim is in flight: the old protocol can end with both idle
    # and the chain parked
    terminals, stranded = explore(1, 1, recheck=False)
    assert stranded > 0, (terminals, stranded)
    # and with nobody else around it cannot
    assert explore(1, 0, recheck=False)[1] == 0
