"""C++ host layer (alphabeta_rs_amd/host): Pedigree::build / from_file / to_file, number formatting and
the NPY writer — no GPU needed.  Golden: data/nodelist.txt + data/edgelist.txt + data/methylome/*.txt ->
data/pedigree_generated.txt (src/pedigree.rs:344-358 asserts the 6 x 4 shape and writes that file)."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"


@pytest.fixture(scope="module")
def hostlib():
    from alphabeta_rs_amd import build as B

    B.build_host()
    L = C.CDLL(str(B.PEDIGREE_LIB))
    L.abh_pedigree_build.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.POINTER(C.c_double), C.c_int,
                                     C.POINTER(C.c_double), C.c_char_p, C.c_int]
    L.abh_pedigree_roundtrip.argtypes = [C.c_char_p, C.c_char_p]
    L.abh_fmt_f64.argtypes = [C.c_double, C.c_char_p, C.c_int]
    L.abh_write_npy.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_longlong]
    return L


def test_pedigree_build_reproduces_generated_fixture(hostlib, golden):
    cwd = os.getcwd()
    os.chdir(GOLDEN)  # the nodelist names ./data/methylome/*.txt relative to the working directory
    try:
        rows = np.zeros((64, 4))
        p0 = C.c_double()
        err = C.create_string_buffer(256)
        n = hostlib.abh_pedigree_build(b"./data/nodelist.txt", b"./data/edgelist.txt", 0.99,
                                       rows.ctypes.data_as(C.POINTER(C.c_double)), 64, C.byref(p0), err, 256)
    finally:
        os.chdir(cwd)
    assert n == 4 * 3 // 2, err.value                      # assert_eq!(shape, [6, 4]), src/pedigree.rs:351
    assert np.array_equal(rows[:n], golden["generated"])    # bit-equal to data/pedigree_generated.txt
    assert p0.value == golden["p0uu_generated"]             # SURVEY.md §4 scratch value 0.6554051647850447


def test_pedigree_file_roundtrip(hostlib, tmp_path, golden):
    out = tmp_path / "pedigree.txt"
    n = hostlib.abh_pedigree_roundtrip(str(GOLDEN / "pedigree_generated.txt").encode(), str(out).encode())
    assert n == 6
    # to_file's tab-separated output equals the fixture the reference itself wrote (src/pedigree.rs:81-90)
    assert out.read_text() == (GOLDEN / "pedigree_generated.txt").read_text()
    n = hostlib.abh_pedigree_roundtrip(str(GOLDEN / "pedigree.txt").encode(), str(tmp_path / "p2.txt").encode())
    assert n == 351
    assert hostlib.abh_pedigree_roundtrip(b"/nonexistent", b"/tmp/x") == -1


def test_rust_display_formatting(hostlib):
    buf = C.create_string_buffer(512)
    cases = {0.10931174089068826: "0.10931174089068826", 1.0: "1", 0.0: "0", 1e-7: "0.0000001", 5.7985750419976e-05:
             "0.000057985750419976", 1e21: "1000000000000000000000", -2.5: "-2.5", float("inf"): "inf",
             float("nan"): "NaN"}
    for v, want in cases.items():
        hostlib.abh_fmt_f64(v, buf, 512)
        assert buf.value.decode() == want


def test_npy_writer(hostlib, tmp_path):
    raw = np.arange(21, dtype=np.float64).reshape(3, 7) / 7.0
    p = tmp_path / "raw.npy"
    hostlib.abh_write_npy(str(p).encode(), raw.ctypes.data_as(C.POINTER(C.c_double)), 3)
    assert np.array_equal(np.load(p), raw)


def test_cli_help_and_argument_errors():
    from alphabeta_rs_amd import build as B

    cli = str(B.build_host())
    r = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--posterior-max-filter" in r.stdout and "--iterations" in r.stdout
    r = subprocess.run([cli, "-n", "/nonexistent/nodes.txt"], capture_output=True, text=True, cwd=str(GOLDEN))
    assert r.returncode == 2 and "valid file path" in r.stderr


def test_analysis_refuses_a_table_with_non_finite_fits(hostlib, abn):
    # a bootstrap without a finite best vertex is a NaN row; the reference panics before its analysis
    # (src/boot_model.rs:86 best_param.unwrap()); quantiles of NaN have no order, so both mirrors refuse the table
    hostlib.abh_analyze.argtypes = [C.POINTER(C.c_double), C.c_longlong, C.POINTER(C.c_double), C.c_char_p, C.c_int]
    rng = np.random.default_rng(5)
    raw = np.abs(rng.normal(1.0, 0.2, size=(40, 7)))
    mean, err = C.c_double(), C.create_string_buffer(256)
    assert hostlib.abh_analyze(raw.ctypes.data_as(C.POINTER(C.c_double)), 40, C.byref(mean), err, 256) == 0
    assert mean.value == abn.analyze(raw)[0, 0]
    for r, c, v in ((17, 2, np.nan), (0, 0, np.nan), (39, 6, np.nan), (5, (0, 1), 0.0), (7, (0, 1), np.inf)):
        bad = raw.copy()
        bad[r, c] = v                      # (0, 0) / (inf, inf): beta / alpha is NaN though the row is not
        assert hostlib.abh_analyze(bad.ctypes.data_as(C.POINTER(C.c_double)), 40, C.byref(mean), err, 256) == -1
        assert f"bootstrap {r} has no finite fit".encode() in err.value
        with pytest.raises(abn.AbnError, match=f"ABN_ERR_NO_FINITE_FIT: bootstrap {r} "):
            abn.analyze(bad)
    inf = raw.copy()
    inf[3, 2] = np.inf                     # infinities do have an order
    assert hostlib.abh_analyze(inf.ctypes.data_as(C.POINTER(C.c_double)), 40, C.byref(mean), err, 256) == 0
    assert np.isinf(abn.analyze(inf)[0, 3])
