"""Synthetic pedigrees of BASELINE.md's configurations (C3, C4, C5).  Host-side input generation only.

A pedigree row is (t0, t1, t2, D): generation of the most recent common ancestor, generations of the
two sampled nodes, observed divergence (src/pedigree.rs:31-45).  D is drawn as
    D_i = c + dt1t2_i(alpha, beta, weight) + eps_i,  eps ~ N(0, (2e-4)^2), clipped at 0
with the true parameters below, noise from numpy's Philox generator (seed 20260101).
The model divergence used to *draw* D is plain numpy and makes no bit-level claim: whatever D comes out
is simply the input that both the HIP path and the oracle are given.
"""
from __future__ import annotations

import numpy as np

SEED = 20260101
TRUE_PARAMS = np.array([1e-4, 5e-4, 0.03, 1e-3])  # alpha, beta, weight, intercept
TRUE_P0UU = 0.75
NOISE_SD = 2e-4


class Tree:
    """Rooted tree of plants: node 0 is the G0 founder; each edge is one generation of selfing."""

    def __init__(self):
        self.parent = [-1]
        self.gen = [0]

    def add_child(self, p: int) -> int:
        self.parent.append(p)
        self.gen.append(self.gen[p] + 1)
        return len(self.parent) - 1

    def add_chain(self, p: int, length: int) -> list[int]:
        out = []
        for _ in range(length):
            p = self.add_child(p)
            out.append(p)
        return out

    @property
    def n_edges(self) -> int:
        return len(self.parent) - 1

    def lca_generation(self, a: int, b: int) -> int:
        seen = set()
        x = a
        while x >= 0:
            seen.add(x)
            x = self.parent[x]
        x = b
        while x not in seen:
            x = self.parent[x]
        return self.gen[x]

    def pair_rows(self, sampled: list[int]) -> np.ndarray:
        """(t0,t1,t2) for every unordered pair of sampled nodes, in the nested-loop order of
        src/pedigree.rs:273-274."""
        rows = []
        for i, a in enumerate(sampled):
            for b in sampled[i + 1:]:
                rows.append((self.lca_generation(a, b), self.gen[a], self.gen[b]))
        return np.asarray(rows, dtype=np.float64)


def model_divergence(gens: np.ndarray, p_uu: float, alpha: float, beta: float, weight: float) -> np.ndarray:
    """numpy restatement of the model used only to draw synthetic observations."""
    a, b = alpha, beta
    G = np.array([[(1 - a) ** 2, 2 * (1 - a) * a, a * a],
                  [0.25 * (b + 1 - a) ** 2, 0.5 * (b + 1 - a) * (a + 1 - b), 0.25 * (a + 1 - b) ** 2],
                  [b * b, 2 * (1 - b) * b, (1 - b) ** 2]])
    tmax = int(gens.max())
    pw = [np.eye(3)]
    for _ in range(tmax):
        pw.append(pw[-1] @ G)
    p_mm = 1 - p_uu
    sv0 = np.array([p_uu, weight * p_mm, (1 - weight) * p_mm])
    out = np.empty(gens.shape[0])
    for i, (t0, t1, t2) in enumerate(gens.astype(int)):
        s0 = sv0 @ pw[t0]
        A, B = pw[t1 - t0], pw[t2 - t0]
        d = [0.5 * (A[r, 0] * B[r, 1] + A[r, 1] * B[r, 0] + A[r, 1] * B[r, 2] + A[r, 2] * B[r, 1])
             + (A[r, 0] * B[r, 2] + A[r, 2] * B[r, 0]) for r in range(3)]
        out[i] = s0[0] * d[0] + s0[1] * d[1] + s0[2] * d[2]
    return out


def draw_observations(gens: np.ndarray, params, p_uu: float, rng: np.random.Generator, n_windows: int | None = None):
    params = np.asarray(params, dtype=np.float64)
    if n_windows is None:
        dt = model_divergence(gens, p_uu, *params[:3])
        return np.maximum(params[3] + dt + rng.normal(0.0, NOISE_SD, gens.shape[0]), 0.0)
    out = np.empty((n_windows, gens.shape[0]))
    for w in range(n_windows):
        dt = model_divergence(gens, p_uu, *params[w, :3])
        out[w] = np.maximum(params[w, 3] + dt + rng.normal(0.0, NOISE_SD, gens.shape[0]), 0.0)
    return out


def _rng(stream: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=SEED + stream))


def c3_tree() -> tuple[Tree, list[int]]:
    """2 lineages from a G0 founder, 8 generations, 100 edges, 15 sampled nodes -> 105 pairs."""
    t = Tree()
    sampled = [0]
    mains = []
    for _ in range(2):
        mains.append(t.add_chain(0, 8))                      # 2 x 8 = 16 edges
    # side branches: at generations 1..7 of each main line, a branch that runs on to generation 8
    for main in mains:
        for g in (1, 2, 3, 4, 5, 6):
            t.add_chain(main[g - 1], 8 - g)                  # 7+6+5+4+3+2 = 27 edges per lineage -> 54
    # 70 edges so far; 30 more: second-order branches off the generation-2 and generation-4 nodes
    for main in mains:
        t.add_chain(main[1], 6)                              # 6
        t.add_chain(main[3], 4)                              # 4
        t.add_chain(main[5], 2)                              # 2
        t.add_chain(main[0], 3)                              # 3 -> 15 per lineage -> 30
    assert t.n_edges == 100, t.n_edges
    # sampled: founder + 7 nodes per lineage (generations 1,2,4,6,8 on the main line, two branch tips)
    for li, main in enumerate(mains):
        sampled += [main[0], main[1], main[3], main[5], main[7]]
        base = 17 + li * 27                                  # first side-branch node of this lineage
        tips = [n for n in range(len(t.gen)) if t.gen[n] == 8 and n not in main]
        lineage_tips = [n for n in tips if _root_child(t, n) == main[0]]
        sampled += lineage_tips[:2]
    assert len(sampled) == 15 and len(set(sampled)) == 15
    return t, sampled


def _root_child(t: Tree, n: int) -> int:
    while t.parent[n] != 0:
        n = t.parent[n]
    return n


def c3_pedigree() -> tuple[np.ndarray, float]:
    """BASELINE C3: N = 105 rows, T = 8.  Returns (pedigree N x 4, p0uu)."""
    t, sampled = c3_tree()
    gens = t.pair_rows(sampled)
    d = draw_observations(gens, TRUE_PARAMS, TRUE_P0UU, _rng(3))
    return np.concatenate([gens, d[:, None]], axis=1), TRUE_P0UU


def c4_windows(n_windows: int = 200, window_offset: int = 0) -> tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """BASELINE C4: the C3 topology with window-specific rates.  Window w's data depends only on its
    GLOBAL index (window_offset + w), so any sharding of the windows sees the same inputs.
    Returns (generations N x 3, D [W x N], p0uu [W], true params [W x 4])."""
    t, sampled = c3_tree()
    gens = t.pair_rows(sampled)
    D = np.empty((n_windows, gens.shape[0]))
    p0 = np.empty(n_windows)
    params = np.empty((n_windows, 4))
    for w in range(n_windows):
        rng = _rng(1000 + window_offset + w)
        params[w] = TRUE_PARAMS * np.array([rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0), 1.0, 1.0])
        p0[w] = rng.uniform(0.6, 0.9)
        D[w] = draw_observations(gens, params[w], p0[w], rng)
    return gens, D, p0, params


def c5_tree(n_lineages: int = 8, depth: int = 125, every: int = 5) -> tuple[Tree, list[int]]:
    """BASELINE C5: 8 lineages x 125 generations = 1000 edges, every 5th generation sampled -> 201 nodes."""
    t = Tree()
    sampled = [0]
    for _ in range(n_lineages):
        chain = t.add_chain(0, depth)
        sampled += [chain[g - 1] for g in range(every, depth + 1, every)]
    return t, sampled


def c5_pedigree(n_lineages: int = 8, depth: int = 125, every: int = 5) -> tuple[np.ndarray, float]:
    t, sampled = c5_tree(n_lineages, depth, every)
    gens = t.pair_rows(sampled)
    d = draw_observations(gens, TRUE_PARAMS, TRUE_P0UU, _rng(5))
    return np.concatenate([gens, d[:, None]], axis=1), TRUE_P0UU


def _model_divergence_by_triple(gens: np.ndarray, p_uu: float, alpha: float, beta: float, weight: float) -> np.ndarray:
    """model_divergence evaluated once per distinct (t0, t1, t2) row (deep pedigrees: 20 100 rows, 950 distinct)"""
    uniq, inv = np.unique(gens.astype(np.int64), axis=0, return_inverse=True)
    return model_divergence(uniq.astype(np.float64), p_uu, alpha, beta, weight)[inv.reshape(-1)]


def c5_windows(n_windows: int, window_offset: int = 0, every: int = 5):
    """BASELINE C5: the deep pedigree (8 lineages x 125 generations, every 5th generation sampled: N = 20 100) with
    window-specific rates, as c4_windows.  Window w's data depends only on its GLOBAL index.
    Returns (generations N x 3, D [W x N], p0uu [W], true params [W x 4])."""
    t, sampled = c5_tree(every=every)
    gens = t.pair_rows(sampled)
    D = np.empty((n_windows, gens.shape[0]))
    p0 = np.empty(n_windows)
    params = np.empty((n_windows, 4))
    for w in range(n_windows):
        rng = _rng(5000 + window_offset + w)
        params[w] = TRUE_PARAMS * np.array([rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0), 1.0, 1.0])
        p0[w] = rng.uniform(0.6, 0.9)
        dt = _model_divergence_by_triple(gens, p0[w], *params[w, :3])
        D[w] = np.maximum(params[w, 3] + dt + rng.normal(0.0, NOISE_SD, gens.shape[0]), 0.0)
    return gens, D, p0, params
