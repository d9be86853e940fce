"""The C-ABI library loads and exports every symbol include/abneutral.h declares.  No compute calls
(there is no GPU in the CPU test tier and no CPU fallback in the product)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "abneutral.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(abn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(abn):
    L = abn.load_library(build_if_missing=True)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/abneutral.h but not exported"
    assert sorted(abn.EXPORTED_SYMBOLS) == names


def test_options_layout_and_defaults(abn):
    o = abn.default_options()
    assert ctypes.sizeof(abn.Options) == 48
    assert o.seed == 20260101 and o.lanes_per_chain == 0 and o.strict_order == 0
    assert o.max_iters_start == 10000 and o.max_iters_boot == 1000   # src/ab_neutral.rs:62, src/boot_model.rs:81
    assert o.sd_tolerance == np.finfo(np.float64).eps
    assert abn.FIT_INFO_DTYPE.itemsize == 24


def test_status_strings(abn):
    L = abn.load_library()
    assert L.abn_status_string(0) == b"ok"
    assert b"pedigree" in L.abn_status_string(2)
    assert L.abn_version() == 1


def test_no_silent_cpu_fallback(abn):
    """Without a device the product refuses to compute (ABN_ERR_NO_DEVICE), it never falls back."""
    if abn.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(abn.AbnError) as e:
        abn.Context(0)
    assert e.value.status == 3


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under alphabeta_rs_amd/ may reference it."""
    for p in (ROOT / "alphabeta_rs_amd").rglob("*"):
        if p.suffix in {".py", ".hip", ".hpp", ".h", ".cpp"}:
            t = p.read_text()
            assert "abn_oracle" not in t and "import oracle" not in t and "from oracle" not in t, p


def test_integration_doc_lists_every_symbol():
    """INTEGRATION.md's `extern "C"` block binds every entry point the header declares (and nothing else)."""
    doc = (ROOT / "INTEGRATION.md").read_text()
    block = doc[doc.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    bound = sorted(set(re.findall(r"pub fn (abn_[a-z0-9_]+)\(", block)))
    assert bound == declared_symbols()


def test_multi_device_entry_refuses_without_a_device(abn):
    """abn_multi_* has no CPU fallback either: without a HIP device creation fails with ABN_ERR_NO_DEVICE; bad arguments
    are status codes; librccl is found (the gather binds it at run time)."""
    assert abn.rccl_available()
    gens = np.array([[0.0, 1.0, 2.0], [0.0, 2.0, 2.0]])
    if abn.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(abn.AbnError) as e:
        abn.MultiPlan([0], gens, 1, 2, 2)
    assert e.value.status == 3
    L = abn.load_library()
    h = ctypes.c_void_p()
    assert L.abn_multi_create(None, 0, None, None, 0, 0, 0, 0, ctypes.byref(h)) == 1 and not h.value
