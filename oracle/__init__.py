"""CPU ORACLE loader (ctypes) — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product path (``alphabeta_rs_amd``) never does; it fails loudly without its HIP library.

The arithmetic lives in ``abn_oracle.c`` (plain C restatement of the reference, pinned to the
reference's golden vectors by ``tests/test_oracle_golden.py``).  Optimizer-trajectory parity with
argmin 0.8.1 is UNPINNED (the crate is not under /root/reference and the reference has no enabled test
at that boundary); see ``abn_oracle.h``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libabn_oracle.so"
if os.environ.get("ABN_ORACLE_LIB"):  # tests/test_oracle_sanitizers.py: the address/UB-sanitizer build of the same source
    _LIB_PATH = Path(os.environ["ABN_ORACLE_LIB"])

FIT_CONVERGED, FIT_MAX_ITERS, FIT_NONFINITE, FIT_TARGET = 0, 1, 2, 3


class FitResult(C.Structure):
    _fields_ = [
        ("best", C.c_double * 4),
        ("best_cost", C.c_double),
        ("iters", C.c_int32),
        ("evals", C.c_int32),
        ("status", C.c_int32),
        ("pad", C.c_int32),
    ]


FIT_DTYPE = np.dtype(
    [("best", "<f8", (4,)), ("best_cost", "<f8"), ("iters", "<i4"), ("evals", "<i4"), ("status", "<i4"), ("pad", "<i4")]
)


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (make).  Building the checker is not using it."""
    src_newer = (not _LIB_PATH.exists()) or any(
        (_HERE / f).stat().st_mtime > _LIB_PATH.stat().st_mtime for f in ("abn_oracle.c", "abn_oracle.h", "Makefile")
    )
    if force or src_newer:
        subprocess.run(["make", "-C", str(_HERE), "-s", "libabn_oracle.so"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        L = C.CDLL(str(_LIB_PATH))
        dp = C.POINTER(C.c_double)
        u32p = C.POINTER(C.c_uint32)
        L.abo_genmatrix.argtypes = [C.c_double, C.c_double, dp]
        L.abo_matrix_power.argtypes = [dp, C.c_int, dp]
        L.abo_matrix_power.restype = C.c_int
        for name in ("abo_p_uu_est", "abo_est_mm", "abo_est_um", "abo_steady_state"):
            getattr(L, name).argtypes = [C.c_double, C.c_double]
            getattr(L, name).restype = C.c_double
        L.abo_as_i8.argtypes = [C.c_double]
        L.abo_as_i8.restype = C.c_int
        L.abo_check_pedigree.argtypes = [dp, C.c_int]
        L.abo_check_pedigree.restype = C.c_int
        for name in ("abo_divergence", "abo_divergence_table"):
            f = getattr(L, name)
            f.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]
            f.restype = C.c_int
        L.abo_cost.argtypes = [dp, C.c_int, dp, C.c_double, C.c_double, C.c_double, dp, C.c_int, C.c_int]
        L.abo_cost.restype = C.c_double
        L.abo_lse.argtypes = [dp, C.c_int, C.c_double, dp]
        L.abo_lse.restype = C.c_double
        L.abo_fit.argtypes = [dp, C.c_int, dp, C.c_double, C.c_double, C.c_double, dp, C.c_int, C.c_double,
                              C.c_int, C.c_int, C.c_int, C.POINTER(FitResult)]
        L.abo_fit_batch.argtypes = [dp, C.c_int, dp, C.c_int64, C.c_double, C.c_double, C.c_double, dp, C.c_int,
                                    C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.abo_select_best.argtypes = [dp, C.c_int, C.c_double, dp, C.c_int, dp, dp, dp, dp]
        L.abo_select_best.restype = C.c_int
        L.abo_bootstrap_row.argtypes = [dp, dp]
        L.abo_philox4x32.argtypes = [C.c_uint32] * 6 + [u32p]
        L.abo_boot_indices.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, u32p]
        L.abo_start_simplex.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, dp]
        L.abo_boot_simplex.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, dp, dp]
        L.abo_boot_model.argtypes = [dp, C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_double, C.c_uint64,
                                     C.c_uint32, C.c_uint32, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int,
                                     C.c_int, C.c_int, dp, C.c_void_p]
        L.abo_boot_model_trace.argtypes = L.abo_boot_model.argtypes + [C.POINTER(C.c_uint8), C.c_int]
        L.abo_analyze.argtypes = [dp, C.c_int64, dp]
        L.abo_pairwise_divergence.argtypes = [C.POINTER(C.c_uint8), dp, C.c_int, C.c_int64, C.c_double,
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), dp]
        L.abo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


EPS = float(np.finfo(np.float64).eps)


# ---------------------------------------------------------------- thin numpy wrappers
def genmatrix(alpha, beta):
    g = np.empty(9)
    lib().abo_genmatrix(alpha, beta, _dp(g))
    return g.reshape(3, 3)


def matrix_power(m, k):
    m = _f64(m).reshape(9)
    out = np.empty(9)
    rc = lib().abo_matrix_power(_dp(m), int(k), _dp(out))
    if rc:
        raise ValueError("negative power")
    return out.reshape(3, 3)


def p_uu_est(a, b):
    return lib().abo_p_uu_est(a, b)


def divergence(ped, p_mm, p_uu, alpha, beta, weight, table=False):
    ped = _f64(ped)
    n = ped.shape[0]
    dt = np.empty(n)
    puu = C.c_double()
    f = lib().abo_divergence_table if table else lib().abo_divergence
    rc = f(_dp(ped), n, p_mm, p_uu, alpha, beta, weight, _dp(dt), C.byref(puu))
    if rc:
        raise ValueError(f"bad pedigree (rc={rc})")
    return dt, puu.value


def cost(ped, p_uu, eqp, eqp_weight, x, dobs=None, lanes=1, table=True):
    ped = _f64(ped)
    x = _f64(x)
    d = None if dobs is None else _f64(dobs)
    return lib().abo_cost(_dp(ped), ped.shape[0], None if d is None else _dp(d), p_uu, eqp, eqp_weight, _dp(x),
                          lanes, int(table))


def lse(ped, p_uu, x):
    ped = _f64(ped)
    x = _f64(x)
    return lib().abo_lse(_dp(ped), ped.shape[0], p_uu, _dp(x))


def fit_batch(ped, p_uu, eqp, eqp_weight, simplex0, max_iters, dobs_rows=None, sd_tol=EPS, shrink_variant=0,
              lanes=1, table=True, threads=0):
    """simplex0: (F,5,4).  dobs_rows: None or (F,N).  Returns a structured array (FIT_DTYPE)."""
    ped = _f64(ped)
    s0 = _f64(simplex0).reshape(-1, 20)
    f = s0.shape[0]
    d = None if dobs_rows is None else _f64(dobs_rows).reshape(f, ped.shape[0])
    out = np.zeros(f, dtype=FIT_DTYPE)
    lib().abo_fit_batch(_dp(ped), ped.shape[0], None if d is None else _dp(d), f, p_uu, eqp, eqp_weight, _dp(s0),
                        max_iters, sd_tol, shrink_variant, lanes, int(table), threads, out.ctypes.data)
    return out


def select_best(ped, p_uu, models):
    ped = _f64(ped)
    m = _f64(models).reshape(-1, 4)
    n = ped.shape[0]
    lse_out = np.empty(m.shape[0])
    model = np.empty(4)
    pred = np.empty(n)
    resid = np.empty(n)
    k = lib().abo_select_best(_dp(ped), n, p_uu, _dp(m), m.shape[0], _dp(lse_out), _dp(model), _dp(pred), _dp(resid))
    return k, model, pred, resid, lse_out


def bootstrap_row(x):
    x = _f64(x)
    row = np.empty(7)
    lib().abo_bootstrap_row(_dp(x), _dp(row))
    return row


def philox(c, k):
    out = (C.c_uint32 * 4)()
    lib().abo_philox4x32(c[0], c[1], c[2], c[3], k[0], k[1], out)
    return np.array(list(out), dtype=np.uint32)


def boot_indices(seed, window, boot, n):
    idx = np.empty(n, dtype=np.uint32)
    lib().abo_boot_indices(seed, window, boot, n, idx.ctypes.data_as(C.POINTER(C.c_uint32)))
    return idx


def start_simplex(seed, window, start, max_div):
    s = np.empty(20)
    lib().abo_start_simplex(seed, window, start, max_div, _dp(s))
    return s.reshape(5, 4)


def boot_simplex(seed, window, boot, params):
    p = _f64(params)
    s = np.empty(20)
    lib().abo_boot_simplex(seed, window, boot, _dp(p), _dp(s))
    return s.reshape(5, 4)


def boot_model(ped, model, pred, resid, p_uu, eqp, eqp_weight, seed, window, b0, nb, max_iters=1000, sd_tol=EPS,
               shrink_variant=0, lanes=1, table=True, threads=0):
    ped = _f64(ped)
    model = _f64(model)
    pred = _f64(pred)
    resid = _f64(resid)
    raw = np.empty((nb, 7))
    res = np.zeros(nb, dtype=FIT_DTYPE)
    lib().abo_boot_model(_dp(ped), ped.shape[0], _dp(model), _dp(pred), _dp(resid), p_uu, eqp, eqp_weight, seed,
                         window, b0, nb, max_iters, sd_tol, shrink_variant, lanes, int(table), threads, _dp(raw),
                         res.ctypes.data)
    return raw, res


def boot_model_trace(ped, model, pred, resid, p_uu, eqp, eqp_weight, seed, window, b0, nb, max_iters=1000, sd_tol=EPS,
                     shrink_variant=0, lanes=1, table=True, threads=0):
    """boot_model plus the branch every Nelder-Mead iteration took: (raw, results, traces[nb, max_iters] u8;
    0 reflection accepted, 1 expansion tried, 2 contraction accepted, 3 contraction rejected, 4 shrink)."""
    ped, model, pred, resid = _f64(ped), _f64(model), _f64(pred), _f64(resid)
    raw = np.empty((nb, 7))
    res = np.zeros(nb, dtype=FIT_DTYPE)
    tr = np.full((nb, max_iters), 255, dtype=np.uint8)
    lib().abo_boot_model_trace(_dp(ped), ped.shape[0], _dp(model), _dp(pred), _dp(resid), p_uu, eqp, eqp_weight, seed,
                               window, b0, nb, max_iters, sd_tol, shrink_variant, lanes, int(table), threads, _dp(raw),
                               res.ctypes.data, tr.ctypes.data_as(C.POINTER(C.c_uint8)), max_iters)
    return raw, res, tr


def analyze(raw):
    raw = _f64(raw)
    out = np.empty(32)
    lib().abo_analyze(_dp(raw), raw.shape[0], _dp(out))
    return out.reshape(4, 8)


def pairwise_divergence(status, posteriormax, posterior_max_filter):
    """DMatrix::from (src/pedigree.rs:210-261): (diff, both, dvalue) per pair i < j."""
    st = np.ascontiguousarray(status, dtype=np.uint8)
    pm = _f64(posteriormax)
    n, L_ = st.shape
    npairs = n * (n - 1) // 2
    diff, both, dval = np.zeros(npairs, dtype=np.uint64), np.zeros(npairs, dtype=np.uint64), np.zeros(npairs)
    lib().abo_pairwise_divergence(st.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(pm), n, L_, posterior_max_filter,
                                  diff.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  both.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(dval))
    return diff, both, dval


def max_threads():
    return lib().abo_max_threads()


def load_pedigree(path):
    """src/pedigree.rs:62-79: skip the header; rows of four numbers (space- or tab-separated)."""
    rows = []
    with open(path) as fh:
        next(fh)
        for line in fh:
            line = line.strip()
            if line:
                rows.append([float(t) for t in line.replace("\t", " ").split(" ") if t != ""])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 4)
