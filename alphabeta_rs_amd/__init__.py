"""alphabeta_rs_amd — MI355X-native ABneutral hot path (ctypes binding of the C-ABI in include/abneutral.h).

The compute path is libabneutral_hip.so (hand-written HIP for gfx950).  There is no CPU fallback: if the
library is missing or no HIP device is usable, calls raise.  PyTorch is used by callers only for
`torch.distributed` (RCCL) and stream plumbing, never for the arithmetic.

Python-level names mirror the reference crate: `Pedigree`, `Model`, `ab_neutral.run`, `boot_model.run`
(see alphabeta_rs_amd/api.py); this module is the thin FFI layer.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import build as _build

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libabneutral_hip.so"
if os.environ.get("ABNEUTRAL_HIP_LIB"):  # development aid (scripts/flag_variants.py): another build of the same library
    LIB_PATH = Path(os.environ["ABNEUTRAL_HIP_LIB"])

ABN_OK = 0
STATUS_NAMES = {
    0: "ABN_OK",
    1: "ABN_ERR_INVALID_ARG",
    2: "ABN_ERR_BAD_PEDIGREE",
    3: "ABN_ERR_NO_DEVICE",
    4: "ABN_ERR_HIP",
    5: "ABN_ERR_NO_FINITE_FIT",
    6: "ABN_ERR_STATE",
}
FIT_CONVERGED, FIT_MAX_ITERS, FIT_NONFINITE, FIT_TARGET = 0, 1, 2, 3
KERNEL_NAMES = {0: "none", 1: "speculative", 2: "resident", 3: "persistent", 4: "stream", 5: "two_pass"}

FIT_INFO_DTYPE = np.dtype(
    [("best_cost", "<f8"), ("iters", "<i4"), ("evals", "<i4"), ("status", "<i4"), ("lanes", "<i4")]
)

# every symbol include/abneutral.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "abn_default_options", "abn_device_count", "abn_init", "abn_shutdown", "abn_device_info", "abn_last_error",
    "abn_status_string", "abn_version", "abn_cost_batch", "abn_fit_batch", "abn_gen_start_simplices",
    "abn_gen_boot_simplices", "abn_gen_boot_indices", "abn_ab_neutral_run", "abn_boot_model_run",
    "abn_analyze", "abn_select_best", "abn_bootstrap_rows", "abn_pairwise_divergence", "abn_plan_create", "abn_plan_destroy", "abn_plan_set_windows", "abn_plan_run",
    "abn_plan_run_phase", "abn_plan_sync", "abn_plan_tail_handed", "abn_plan_kernel_ms", "abn_plan_raw_device_ptr",
    "abn_plan_bind_raw", "abn_plan_download", "abn_plan_counters", "abn_plan_device_bytes",
    "abn_plan_set_window_ids", "abn_plan_failed_windows",
    "abn_multi_create", "abn_multi_destroy", "abn_multi_last_error", "abn_multi_set_windows", "abn_multi_run",
    "abn_multi_sync", "abn_multi_shard", "abn_multi_raw_device_ptr", "abn_multi_download", "abn_multi_counters",
    "abn_multi_rccl_available", "abn_reduction_tree", "abn_pairwise_divergence_dev", "abn_multi_set_window_ids",
    "abn_multi_plan_shard", "abn_plan_last_kernels", "abn_multi_kernel_ms",
]


class Options(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("lanes_per_chain", C.c_int32),
        ("strict_order", C.c_int32),
        ("shrink_on_failed_contraction", C.c_int32),
        ("max_iters_start", C.c_int32),
        ("max_iters_boot", C.c_int32),
        ("stream_mode", C.c_int32),
        ("sd_tolerance", C.c_double),
        ("window_groups", C.c_int32),
        ("no_fixed_point_skip", C.c_int32),
    ]


class AbnError(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {detail}")


_lib = None


def load_library(build_if_missing: bool = False) -> C.CDLL:
    """Load libabneutral_hip.so.  Raises (never falls back) when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        if build_if_missing:
            _build.build_hip()
        else:
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `python -m alphabeta_rs_amd.build` "
                "(the ABneutral path has no CPU fallback)"
            )
    L = C.CDLL(str(LIB_PATH))
    dp, u32p, vp = C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_void_p
    op = C.POINTER(Options)
    L.abn_default_options.argtypes = [op]
    L.abn_default_options.restype = None
    L.abn_device_count.argtypes = [C.POINTER(C.c_int)]
    L.abn_init.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.abn_shutdown.argtypes = [vp]
    L.abn_device_info.argtypes = [vp, C.POINTER(C.c_int32)]
    L.abn_last_error.argtypes = [vp]
    L.abn_last_error.restype = C.c_char_p
    L.abn_status_string.argtypes = [C.c_int]
    L.abn_status_string.restype = C.c_char_p
    L.abn_cost_batch.argtypes = [vp, op, dp, C.c_int32, C.c_double, C.c_double, C.c_double, dp, C.c_int64, dp, dp,
                                 u32p, u32p, C.c_int64, dp, dp, dp]
    L.abn_fit_batch.argtypes = [vp, op, dp, C.c_int32, C.c_double, C.c_double, C.c_double, dp, C.c_int64, dp,
                                C.c_int32, dp, vp]
    L.abn_gen_start_simplices.argtypes = [C.c_uint64, C.c_uint32, C.c_int32, C.c_double, dp]
    L.abn_gen_boot_simplices.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, dp, dp]
    L.abn_gen_boot_indices.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int32, u32p]
    L.abn_ab_neutral_run.argtypes = [vp, op, dp, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, dp, dp,
                                     dp, dp, vp, dp]
    L.abn_boot_model_run.argtypes = [vp, op, dp, C.c_int32, dp, dp, dp, C.c_double, C.c_double, C.c_double,
                                     C.c_int32, dp, vp]
    L.abn_analyze.argtypes = [dp, C.c_int64, dp]
    L.abn_select_best.argtypes = [vp, dp, C.c_int32, C.c_double, dp, C.c_int32, C.POINTER(C.c_int32), dp, dp, dp, dp]
    L.abn_bootstrap_rows.argtypes = [vp, dp, C.c_int64, dp]
    L.abn_pairwise_divergence.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int32, C.c_int64, C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64), dp]
    L.abn_pairwise_divergence_dev.argtypes = [vp, vp, C.c_int32, C.c_int64, vp, vp, vp, dp]
    L.abn_plan_create.argtypes = [vp, op, dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32,
                                  C.POINTER(vp)]
    L.abn_plan_destroy.argtypes = [vp]
    L.abn_plan_set_windows.argtypes = [vp, dp, dp, dp, dp]
    L.abn_plan_run.argtypes = [vp]
    L.abn_plan_run_phase.argtypes = [vp, C.c_int32]
    L.abn_plan_sync.argtypes = [vp]
    L.abn_plan_tail_handed.argtypes = [vp, C.POINTER(C.c_int64)]
    L.abn_plan_kernel_ms.argtypes = [vp, dp]
    L.abn_plan_raw_device_ptr.argtypes = [vp, C.POINTER(vp)]
    L.abn_plan_bind_raw.argtypes = [vp, vp]
    L.abn_plan_download.argtypes = [vp, dp, dp, dp, dp, vp, vp, C.POINTER(C.c_int32)]
    L.abn_plan_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    L.abn_plan_device_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
    L.abn_plan_set_window_ids.argtypes = [vp, u32p]
    L.abn_plan_failed_windows.argtypes = [vp, C.POINTER(C.c_int32)]
    L.abn_plan_last_kernels.argtypes = [vp, C.POINTER(C.c_int32)]
    i32p = C.POINTER(C.c_int32)
    L.abn_multi_create.argtypes = [i32p, C.c_int32, op, dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.abn_multi_destroy.argtypes = [vp]
    L.abn_multi_last_error.argtypes = [vp]
    L.abn_multi_last_error.restype = C.c_char_p
    L.abn_multi_set_windows.argtypes = [vp, dp, dp, dp, dp]
    L.abn_multi_set_window_ids.argtypes = [vp, u32p]
    L.abn_multi_run.argtypes = [vp]
    L.abn_multi_sync.argtypes = [vp]
    L.abn_multi_shard.argtypes = [vp, C.c_int32, i32p]
    L.abn_multi_plan_shard.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p]
    L.abn_multi_raw_device_ptr.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.abn_multi_kernel_ms.argtypes = [vp, C.c_int32, dp]
    L.abn_multi_download.argtypes = [vp, dp, dp, dp, dp, vp, vp, i32p]
    L.abn_multi_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    L.abn_multi_rccl_available.argtypes = [C.POINTER(C.c_int)]
    L.abn_reduction_tree.argtypes = [op, dp, C.c_int32, i32p]
    _lib = L
    return L


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _u32p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint32))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def default_options(**kw) -> Options:
    o = Options()
    load_library().abn_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k}")
        setattr(o, k, v)
    return o


def device_count() -> int:
    n = C.c_int(0)
    load_library().abn_device_count(C.byref(n))
    return n.value


def gen_start_simplices(seed: int, window: int, n_starts: int, max_divergence: float) -> np.ndarray:
    out = np.empty((n_starts, 5, 4))
    rc = load_library().abn_gen_start_simplices(seed, window, n_starts, max_divergence, _dp(out))
    if rc:
        raise AbnError(rc)
    return out


def gen_boot_simplices(seed: int, window: int, b0: int, nb: int, params) -> np.ndarray:
    p = _f64(params, (4,))
    out = np.empty((nb, 5, 4))
    rc = load_library().abn_gen_boot_simplices(seed, window, b0, nb, _dp(p), _dp(out))
    if rc:
        raise AbnError(rc)
    return out


def analyze(raw) -> np.ndarray:
    """src/analysis.rs:50-98 -> (4, 8): mean, sd, ci_lo, ci_hi x (alpha, beta, beta/alpha, weight,
    intercept, pr_mm, pr_um, pr_uu)."""
    raw = _f64(raw).reshape(-1, 7)
    with np.errstate(all="ignore"):
        bad = np.isnan(raw).any(axis=1) | np.isnan(raw[:, 1] / raw[:, 0])
    if bad.any():   # abn_analyze sorts with `<`: NaN-free columns only (the reference panics before it gets here)
        raise AbnError(5, f"bootstrap {int(np.flatnonzero(bad)[0])} has no finite fit: no analysis of this table")
    out = np.empty(32)
    rc = load_library().abn_analyze(_dp(raw), raw.shape[0], _dp(out))
    if rc:
        raise AbnError(rc)
    return out.reshape(4, 8)


class Context:
    """abn_ctx: one HIP device + stream.  `stream`: None = a private stream owned by the context; 0 = the
    device's default (null) stream (torch's current stream unless switched); otherwise a raw hipStream_t."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._L = load_library()
        h = C.c_void_p()
        if stream is None:
            sp = None
        elif stream == 0:
            sp = C.c_void_p(-1)  # ABN_STREAM_DEFAULT
        else:
            sp = C.c_void_p(stream)
        rc = self._L.abn_init(device, sp, C.byref(h))
        if rc:
            raise AbnError(rc, "abn_init failed (no HIP device? the ABneutral path has no CPU fallback)")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.abn_shutdown(self._h)
            self._h = None

    def device_info(self):
        """what abn_init read from hipDeviceProp and the persistent-launch geometry derived from it"""
        out = (C.c_int32 * 4)()
        self._check(self._L.abn_device_info(self._h, out))
        return {"compute_units": out[0], "lds_kib_per_cu": out[1], "persistent_wavefronts": out[2],
                "persistent_wavefronts_small": out[3]}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise AbnError(rc, (self._L.abn_last_error(self._h) or b"").decode())

    # ---- (1) Problem::cost for M candidates
    def cost_batch(self, pedigree, p_uu0, eqp, eqp_weight, candidates, *, pred=None, resid=None, idx=None,
                   cand_to_boot=None, options: Options | None = None, want_dt=False, want_puu=False):
        ped = _f64(pedigree).reshape(-1, 4)
        cand = _f64(candidates).reshape(-1, 4)
        n, m = ped.shape[0], cand.shape[0]
        cost = np.empty(m)
        dt = np.empty((m, n)) if want_dt else None
        puu = np.empty(m) if want_puu else None
        nb = 0
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1, n)
            nb = idx.shape[0]
            pred, resid = _f64(pred, (n,)), _f64(resid, (n,))
            if cand_to_boot is not None:
                cand_to_boot = np.ascontiguousarray(cand_to_boot, dtype=np.uint32).reshape(m)
        self._check(self._L.abn_cost_batch(self._h, C.byref(options) if options else None, _dp(ped), n, p_uu0, eqp,
                                           eqp_weight, _dp(cand), m, _dp(pred), _dp(resid), _u32p(idx),
                                           _u32p(cand_to_boot), nb, _dp(cost), _dp(dt), _dp(puu)))
        out = [cost]
        if want_dt:
            out.append(dt)
        if want_puu:
            out.append(puu)
        return out[0] if len(out) == 1 else tuple(out)

    # ---- Nelder-Mead fits from explicit start simplices
    def fit_batch(self, pedigree, p_uu0, eqp, eqp_weight, simplex0, max_iters, *, dobs_rows=None,
                  options: Options | None = None):
        ped = _f64(pedigree).reshape(-1, 4)
        s0 = _f64(simplex0).reshape(-1, 20)
        n, f = ped.shape[0], s0.shape[0]
        d = None if dobs_rows is None else _f64(dobs_rows, (f, n))
        best = np.empty((f, 4))
        info = np.zeros(f, dtype=FIT_INFO_DTYPE)
        self._check(self._L.abn_fit_batch(self._h, C.byref(options) if options else None, _dp(ped), n, p_uu0, eqp,
                                          eqp_weight, _dp(s0), f, _dp(d), max_iters, _dp(best), info.ctypes.data))
        return best, info

    def gen_boot_indices(self, seed, window, b0, nb, n_rows) -> np.ndarray:
        idx = np.empty((nb, n_rows), dtype=np.uint32)
        self._check(self._L.abn_gen_boot_indices(self._h, seed, window, b0, nb, n_rows, _u32p(idx)))
        return idx

    # ---- (2) ab_neutral::run
    def ab_neutral_run(self, pedigree, p0uu, eqp, eqp_weight, n_starts, *, options: Options | None = None):
        ped = _f64(pedigree).reshape(-1, 4)
        n = ped.shape[0]
        model, pred, resid = np.empty(4), np.empty(n), np.empty(n)
        allm = np.empty((n_starts, 4))
        info = np.zeros(n_starts, dtype=FIT_INFO_DTYPE)
        lse = np.empty(n_starts)
        self._check(self._L.abn_ab_neutral_run(self._h, C.byref(options) if options else None, _dp(ped), n, p0uu, eqp,
                                               eqp_weight, n_starts, _dp(model), _dp(pred), _dp(resid), _dp(allm),
                                               info.ctypes.data, _dp(lse)))
        return model, pred, resid, {"models": allm, "info": info, "lse": lse}

    def select_best(self, pedigree, p0uu, models):
        """src/ab_neutral.rs:83-135: (index, model, pred, resid, lse)"""
        ped = _f64(pedigree).reshape(-1, 4)
        m = _f64(models).reshape(-1, 4)
        n = ped.shape[0]
        k = C.c_int32(-1)
        model, pred, resid, lse = np.empty(4), np.empty(n), np.empty(n), np.empty(m.shape[0])
        self._check(self._L.abn_select_best(self._h, _dp(ped), n, p0uu, _dp(m), m.shape[0], C.byref(k), _dp(model),
                                            _dp(pred), _dp(resid), _dp(lse)))
        return k.value, model, pred, resid, lse

    def bootstrap_rows(self, best):
        """src/boot_model.rs:86-91"""
        b = _f64(best).reshape(-1, 4)
        raw = np.empty((b.shape[0], 7))
        self._check(self._L.abn_bootstrap_rows(self._h, _dp(b), b.shape[0], _dp(raw)))
        return raw

    def pairwise_divergence(self, codes):
        """DMatrix::from (src/pedigree.rs:210-261).  codes: (n_samples, n_sites) u8 = status | 0x80 if filtered.
        Returns (diff u64, both u64, dvalue f64), one entry per pair i < j in nested-loop order."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        n, L = codes.shape
        npairs = n * (n - 1) // 2
        diff, both = np.zeros(npairs, dtype=np.uint64), np.zeros(npairs, dtype=np.uint64)
        dval = np.zeros(npairs)
        self._check(self._L.abn_pairwise_divergence(self._h, codes.ctypes.data_as(C.POINTER(C.c_uint8)), n, L,
                                                    diff.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                    both.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(dval)))
        return diff, both, dval

    def pairwise_divergence_dev(self, codes_ptr: int, n_samples: int, n_sites: int, diff_ptr: int = 0,
                                both_ptr: int = 0, dvalue_ptr: int = 0) -> float:
        """The same on device-resident buffers (raw device pointers, e.g. torch tensors' data_ptr()): u8 codes
        [n x L] in, u64 diff / both and f64 dvalue [pairs] out (0 = not wanted).  Returns the kernels' HIP-event ms."""
        ms = C.c_double(0.0)
        self._check(self._L.abn_pairwise_divergence_dev(self._h, C.c_void_p(codes_ptr), n_samples, n_sites,
                                                        C.c_void_p(diff_ptr or None), C.c_void_p(both_ptr or None),
                                                        C.c_void_p(dvalue_ptr or None), C.byref(ms)))
        return ms.value

    # ---- (3) boot_model::run
    def boot_model_run(self, pedigree, model, pred, resid, p0uu, eqp, eqp_weight, n_boot, *,
                       options: Options | None = None):
        ped = _f64(pedigree).reshape(-1, 4)
        n = ped.shape[0]
        raw = np.empty((n_boot, 7))
        info = np.zeros(n_boot, dtype=FIT_INFO_DTYPE)
        self._check(self._L.abn_boot_model_run(self._h, C.byref(options) if options else None, _dp(ped), n,
                                               _dp(_f64(model, (4,))), _dp(_f64(pred, (n,))), _dp(_f64(resid, (n,))),
                                               p0uu, eqp, eqp_weight, n_boot, _dp(raw), info.ctypes.data))
        return raw, info


class Plan:
    """abn_plan: device-resident batch of W windows x (S starts + B bootstraps) over one pedigree topology."""

    def __init__(self, ctx: Context, generations, n_windows, n_starts, n_boot, *, window_offset=0, boot_offset=0,
                 options: Options | None = None):
        self.ctx = ctx
        self._L = ctx._L
        g = _f64(generations).reshape(-1, 3)
        self.N, self.W, self.S, self.B = g.shape[0], n_windows, n_starts, n_boot
        h = C.c_void_p()
        ctx._check(self._L.abn_plan_create(ctx._h, C.byref(options) if options else None, _dp(g), self.N, n_windows,
                                           n_starts, n_boot, window_offset, boot_offset, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.abn_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_window_ids(self, ids):
        """ids[W]: every window's index in the Philox counters (default window_offset + w); before set_windows"""
        a = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32).reshape(self.W)
        self.ctx._check(self._L.abn_plan_set_window_ids(self._h, _u32p(a)))

    def failed_windows(self) -> int:
        """windows whose selection found no finite start in the last phase-A run"""
        n = C.c_int32(0)
        self.ctx._check(self._L.abn_plan_failed_windows(self._h, C.byref(n)))
        return n.value

    def set_windows(self, d_obs, p0uu, eqp=None, eqp_weight=None):
        d = _f64(d_obs, (self.W, self.N))
        p = _f64(p0uu, (self.W,))
        e = None if eqp is None else _f64(eqp, (self.W,))
        ew = None if eqp_weight is None else _f64(eqp_weight, (self.W,))
        self.ctx._check(self._L.abn_plan_set_windows(self._h, _dp(d), _dp(p), _dp(e), _dp(ew)))

    def run(self):
        self.ctx._check(self._L.abn_plan_run(self._h))

    def run_phase(self, phase: int):
        self.ctx._check(self._L.abn_plan_run_phase(self._h, phase))

    def sync(self):
        self.ctx._check(self._L.abn_plan_sync(self._h))

    def tail_handed(self):
        """chains the last persistent launch of (phase A, phase B) handed to the speculative kernel for its tail"""
        out = (C.c_int64 * 2)()
        self.ctx._check(self._L.abn_plan_tail_handed(self._h, out))
        return int(out[0]), int(out[1])

    def kernel_ms(self):
        ms = np.zeros(3)
        self.ctx._check(self._L.abn_plan_kernel_ms(self._h, _dp(ms)))
        return {"fit_starts": ms[0], "select": ms[1], "fit_boot": ms[2]}

    def raw_device_ptr(self) -> int:
        p = C.c_void_p()
        self.ctx._check(self._L.abn_plan_raw_device_ptr(self._h, C.byref(p)))
        return p.value or 0

    def bind_raw(self, dev_ptr: int):
        self.ctx._check(self._L.abn_plan_bind_raw(self._h, C.c_void_p(dev_ptr)))

    def download(self, want_info=True, allow_failed_windows=False):
        """Raises AbnError(ABN_ERR_NO_FINITE_FIT) when a window has no finite start (the reference panics) unless
        allow_failed_windows: then the dict comes back with best_start = -1 and NaN rows for those windows."""
        W, N, S, B = self.W, self.N, self.S, self.B
        models, pred, resid = np.empty((W, 4)), np.empty((W, N)), np.empty((W, N))
        raw = np.empty((W, B, 7)) if B else None
        ia = np.zeros((W, S), dtype=FIT_INFO_DTYPE) if (S and want_info) else None
        ib = np.zeros((W, B), dtype=FIT_INFO_DTYPE) if (B and want_info) else None
        bs = np.full(W, -1, dtype=np.int32)
        rc = self._L.abn_plan_download(self._h, _dp(models), _dp(pred), _dp(resid), _dp(raw),
                                       None if ia is None else ia.ctypes.data,
                                       None if ib is None else ib.ctypes.data,
                                       bs.ctypes.data_as(C.POINTER(C.c_int32)))
        if not (rc == 5 and allow_failed_windows):  # ABN_ERR_NO_FINITE_FIT: every buffer is filled all the same
            self.ctx._check(rc)
        return {"models": models, "pred": pred, "resid": resid, "raw": raw, "info_a": ia, "info_b": ib,
                "best_start": bs}

    def counters(self):
        out = (C.c_int64 * 5)()
        self.ctx._check(self._L.abn_plan_counters(self._h, out))
        return {"fits": out[0], "evals": out[1], "iters": out[2], "evals_skipped": out[3] + out[4],
                "evals_skipped_starts": out[3], "evals_skipped_boot": out[4]}

    def device_bytes(self) -> int:
        b = C.c_int64()
        self.ctx._check(self._L.abn_plan_device_bytes(self._h, C.byref(b)))
        return b.value

    def last_kernels(self):
        """which fit kernel the last run of each phase used (ABN_KERNEL_* names) and its lanes per chain"""
        out = (C.c_int32 * 4)()
        self.ctx._check(self._L.abn_plan_last_kernels(self._h, out))
        return {"starts": (KERNEL_NAMES.get(out[0], out[0]), out[1]), "boot": (KERNEL_NAMES.get(out[2], out[2]), out[3])}


def reduction_tree(generations, options: Options | None = None) -> int:
    """The residual reduction tree of a pedigree (abn_fit_info.lanes): host arithmetic, no device needed."""
    g = _f64(generations).reshape(-1, 3)
    t = C.c_int32(0)
    rc = load_library().abn_reduction_tree(C.byref(options) if options else None, _dp(g), g.shape[0], C.byref(t))
    if rc:
        raise AbnError(rc)
    return t.value


def multi_plan_shard(n_windows: int, n_boot: int, n_devices: int, device_index: int):
    """(window_offset, n_windows, boot_offset, n_boot) of one device's shard in the one-process / several-GPUs entry
    points (abn_multi_plan_shard: host arithmetic, no device needed)."""
    out = (C.c_int32 * 4)()
    rc = load_library().abn_multi_plan_shard(n_windows, n_boot, n_devices, device_index, out)
    if rc:
        raise AbnError(rc)
    return tuple(int(v) for v in out)


def rccl_available() -> bool:
    ok = C.c_int(0)
    load_library().abn_multi_rccl_available(C.byref(ok))
    return bool(ok.value)


class MultiPlan:
    """abn_multi: one process, several GPUs — a plan per device, windows (or, with fewer windows than devices,
    bootstraps) sharded in contiguous blocks, the bootstrap tables gathered with RCCL over xGMI."""

    def __init__(self, devices, generations, n_windows, n_starts, n_boot, *, options: Options | None = None):
        self._L = load_library()
        g = _f64(generations).reshape(-1, 3)
        self.devices = [int(d) for d in devices]
        self.N, self.W, self.S, self.B = g.shape[0], n_windows, n_starts, n_boot
        devs = (C.c_int32 * len(self.devices))(*self.devices)
        h = C.c_void_p()
        rc = self._L.abn_multi_create(devs, len(self.devices), C.byref(options) if options else None, _dp(g), self.N,
                                      n_windows, n_starts, n_boot, C.byref(h))
        self._h = h if h.value else None
        if rc:
            msg = (self._L.abn_multi_last_error(h) or b"").decode() if h.value else ""
            self.close()
            raise AbnError(rc, msg)

    def _check(self, rc):
        if rc:
            raise AbnError(rc, (self._L.abn_multi_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.abn_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_window_ids(self, ids):
        a = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32).reshape(self.W)
        self._check(self._L.abn_multi_set_window_ids(self._h, _u32p(a)))

    def set_windows(self, d_obs, p0uu, eqp=None, eqp_weight=None):
        d = _f64(d_obs, (self.W, self.N))
        p = _f64(p0uu, (self.W,))
        e = None if eqp is None else _f64(eqp, (self.W,))
        ew = None if eqp_weight is None else _f64(eqp_weight, (self.W,))
        self._check(self._L.abn_multi_set_windows(self._h, _dp(d), _dp(p), _dp(e), _dp(ew)))

    def run(self):
        self._check(self._L.abn_multi_run(self._h))

    def sync(self):
        self._check(self._L.abn_multi_sync(self._h))

    def kernel_ms(self, device_index: int = 0):
        ms = np.zeros(3)
        self._check(self._L.abn_multi_kernel_ms(self._h, device_index, _dp(ms)))
        return {"fit_starts": ms[0], "select": ms[1], "fit_boot": ms[2]}

    def shard(self, device_index: int):
        out = (C.c_int32 * 4)()
        self._check(self._L.abn_multi_shard(self._h, device_index, out))
        return {"window_offset": out[0], "n_windows": out[1], "boot_offset": out[2], "n_boot": out[3]}

    def download(self, want_info=True, allow_failed_windows=False):
        W, N, S, B = self.W, self.N, self.S, self.B
        models, pred, resid = np.empty((W, 4)), np.empty((W, N)), np.empty((W, N))
        raw = np.empty((W, B, 7))
        ia = np.zeros((W, S), dtype=FIT_INFO_DTYPE) if want_info else None
        ib = np.zeros((W, B), dtype=FIT_INFO_DTYPE) if want_info else None
        bs = np.full(W, -1, dtype=np.int32)
        rc = self._L.abn_multi_download(self._h, _dp(models), _dp(pred), _dp(resid), _dp(raw),
                                        None if ia is None else ia.ctypes.data,
                                        None if ib is None else ib.ctypes.data,
                                        bs.ctypes.data_as(C.POINTER(C.c_int32)))
        if not (rc == 5 and allow_failed_windows):
            self._check(rc)
        return {"models": models, "pred": pred, "resid": resid, "raw": raw, "info_a": ia, "info_b": ib,
                "best_start": bs}

    def counters(self):
        out = (C.c_int64 * 5)()
        self._check(self._L.abn_multi_counters(self._h, out))
        return {"fits": out[0], "evals": out[1], "iters": out[2], "evals_skipped": out[3] + out[4],
                "evals_skipped_starts": out[3], "evals_skipped_boot": out[4]}
