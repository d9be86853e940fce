"""SURVEY.md §5 (sanitizers), product side: the C++ host layer (alphabeta_rs_amd/host: the parsers of nodelist / edgelist /
methylome / pedigree files, Pedigree::build, the number formatter, the NPY writer) rebuilt with g++ AddressSanitizer +
UndefinedBehaviorSanitizer and run (a) over the reference's own data/ fixture (src/pedigree.rs:344-358) and (b) over ~250
seeded corruptions of those files — truncated, ragged, non-numeric, empty, binary, cyclic or unknown nodes, missing
methylomes.  The reference panics or returns Err on such input (src/pedigree.rs:92-208 `?` / `expect`); the host layer must
answer every one with a row count or an error text, never with a sanitizer report or a crash.  GPU sanitizers are not
available on the pool; the device side is held by the parity and fuzz tests.  No GPU needed (gpu_pairwise = false)."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
ASAN_LIB = ROOT / "build" / "libabneutral_host_asan.so"

_SCRIPT = r"""
import ctypes as C, os, random, shutil, sys
import numpy as np
lib, golden, work = sys.argv[1], sys.argv[2], sys.argv[3]
L = C.CDLL(lib)
L.abh_pedigree_build.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.POINTER(C.c_double), C.c_int,
                                 C.POINTER(C.c_double), C.c_char_p, C.c_int]
L.abh_pedigree_roundtrip.argtypes = [C.c_char_p, C.c_char_p]
L.abh_fmt_f64.argtypes = [C.c_double, C.c_char_p, C.c_int]
L.abh_write_npy.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_longlong]

def build(filt=0.99):
    rows = np.zeros((64, 4)); p0 = C.c_double(); err = C.create_string_buffer(512)
    n = L.abh_pedigree_build(b"./data/nodelist.txt", b"./data/edgelist.txt", filt,
                             rows.ctypes.data_as(C.POINTER(C.c_double)), 64, C.byref(p0), err, 512)
    return n, rows[:max(n, 0)].copy(), p0.value, err.value

# (a) the reference's fixture, bit-equal, under the sanitizers
os.chdir(golden)
n, rows, p0, err = build()
want = np.loadtxt(os.path.join(golden, "pedigree_generated.txt"), skiprows=1)
assert n == 6 and np.array_equal(rows, want), (n, err)
out = os.path.join(work, "rt.txt")
assert L.abh_pedigree_roundtrip(os.path.join(golden, "pedigree.txt").encode(), out.encode()) == 351
assert L.abh_pedigree_roundtrip(b"/nonexistent", out.encode()) == -1
buf = C.create_string_buffer(400)   # Rust `{}` prints no exponent: 5e-324 is 326 characters
for v in (0.0, -0.0, 1.0, 1e-7, 1e16, 123456789.125, 5e-324, 1.7976931348623157e308, float("inf"), float("-inf"), float("nan"),
          0.1, 1e21, 1e-5, 0.00001234):
    assert L.abh_fmt_f64(v, buf, 400) > 0
    if v == v:
        assert float(buf.value) == v, (v, buf.value)
assert L.abh_fmt_f64(1.0 / 3.0, buf, 4) == -1          # too small a buffer is an error, not an overrun
r7 = np.arange(21.0)
L.abh_write_npy(os.path.join(work, "raw.npy").encode(), r7.ctypes.data_as(C.POINTER(C.c_double)), 3)
assert np.load(os.path.join(work, "raw.npy")).size == 21
L.abh_write_npy(os.path.join(work, "raw0.npy").encode(), r7.ctypes.data_as(C.POINTER(C.c_double)), 0)
L.abh_analyze.argtypes = [C.POINTER(C.c_double), C.c_longlong, C.POINTER(C.c_double), C.c_char_p, C.c_int]
tab = np.abs(np.random.default_rng(3).normal(1.0, 0.2, size=(64, 7))); m = C.c_double(); e = C.create_string_buffer(256)
assert L.abh_analyze(tab.ctypes.data_as(C.POINTER(C.c_double)), 64, C.byref(m), e, 256) == 0
assert L.abh_analyze(tab.ctypes.data_as(C.POINTER(C.c_double)), 1, C.byref(m), e, 256) == 0     # one row: sd is NaN, no overrun
tab[9, 4] = np.nan
assert L.abh_analyze(tab.ctypes.data_as(C.POINTER(C.c_double)), 64, C.byref(m), e, 256) == -1 and b"bootstrap 9" in e.value

# (b) seeded corruptions of the fixture's files
rng = random.Random(20261005)
src = os.path.join(golden, "data")
files = ["nodelist.txt", "edgelist.txt"] + ["methylome/" + f for f in sorted(os.listdir(os.path.join(src, "methylome")))]
JUNK = [b"", b"\n", b"\t", b",", b"-", b"NaN", b"inf", b"-1", b"1e999", b"99999999999999999999", b"\x00", b"\xff\xfe", b"Y",
        b"0_0", b"9_9", b"M", b"U", b"I", b"X", b"1.5", b"\r\n", b" "]

def corrupt(data):
    lines = data.split(b"\n")
    kind = rng.randrange(12)
    if kind == 0:
        return b""
    if kind == 1:
        return data[:rng.randrange(len(data) + 1)]                       # cut anywhere, mid-field included
    if kind == 2:
        return lines[0]                                                  # header only, no newline
    if kind == 3:
        return b"\n".join(lines[1:])                                     # no header
    i = rng.randrange(len(lines))
    sep = b"," if b"," in lines[0] else b"\t"
    cells = lines[i].split(sep)
    if kind == 4:
        cells = cells[:rng.randrange(len(cells) + 1)]                    # ragged row
    elif kind == 5:
        cells[rng.randrange(len(cells))] = rng.choice(JUNK)
    elif kind == 6:
        cells = cells + [rng.choice(JUNK)] * rng.randrange(1, 4)         # extra columns
    elif kind == 7:
        lines.insert(i, lines[i])                                        # duplicated row
        return b"\n".join(lines)
    elif kind == 8:
        return bytes(rng.randrange(256) for _ in range(rng.randrange(1, 400)))   # binary noise
    elif kind == 9:
        return data.replace(b"\n", b"\r\n")
    elif kind == 10:
        return data.replace(sep, rng.choice([b" ", b";", b"\t\t"]))
    else:
        del lines[i]                                                     # a row gone (an edge, a node, a site)
        return b"\n".join(lines)
    lines[i] = sep.join(cells)
    return b"\n".join(lines)

cases = ok = refused = 0
for case in range(int(sys.argv[4])):
    d = os.path.join(work, "c")
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(src, os.path.join(d, "data"))
    for f in rng.sample(files, rng.choice([1, 1, 1, 2])):
        p = os.path.join(d, "data", f)
        if rng.randrange(14) == 0:
            os.remove(p)                                                 # a file the nodelist names is not there
        else:
            blob = open(p, "rb").read()
            if f.startswith("methylome") and rng.randrange(2):
                blob = b"\n".join(blob.split(b"\n")[:rng.randrange(1, 40)])   # short methylomes: ragged site counts
            open(p, "wb").write(corrupt(blob))
    os.chdir(d)
    n, rows, p0, err = build(rng.choice([0.99, 0.0, 1.0, 2.0, -1.0, float("nan")]))
    cases += 1
    if n >= 0:
        ok += 1
        assert rows.shape == (n, 4)
    else:
        refused += 1
        assert n in (-1, -2) and (n == -2 or len(err) > 0), (n, err)
os.chdir(work)
print("sanitized host ok", cases, ok, refused)
"""


def _build_asan():
    ASAN_LIB.parent.mkdir(exist_ok=True)
    hip = ROOT / "alphabeta_rs_amd"
    if not (hip / "libabneutral_hip.so").exists():
        pytest.skip("libabneutral_hip.so not built")
    src = hip / "host" / "host_capi.cpp"
    deps = [src, *sorted((hip / "host").glob("*.hpp")), ROOT / "include" / "abneutral.h"]
    if ASAN_LIB.exists() and ASAN_LIB.stat().st_mtime >= max(d.stat().st_mtime for d in deps):
        return
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not found")
    r = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
                        "-fno-omit-frame-pointer", "-I", str(ROOT / "include"), "-fPIC", "-shared", "-o", str(ASAN_LIB), str(src),
                        "-L", str(hip), "-labneutral_hip", f"-Wl,-rpath,{hip}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_host_layer_under_address_and_ub_sanitizers(tmp_path):
    _build_asan()
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    # libstdc++ beside the runtime: python itself does not link it, and the runtime's __cxa_throw interceptor must find it
    stdcxx = subprocess.run(["gcc", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=f"{asan} {stdcxx}", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", _SCRIPT, str(ASAN_LIB), str(GOLDEN), str(tmp_path), "250"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "sanitized host ok" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    cases, ok, refused = map(int, r.stdout.split("sanitized host ok")[1].split()[:3])
    assert cases == 250 and ok > 0 and refused > 0, (cases, ok, refused)   # both outcomes are exercised
