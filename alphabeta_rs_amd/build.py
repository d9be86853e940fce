"""Builds libabneutral_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

-ffp-contract=off is REQUIRED: parity with the reference depends on every multiply-add staying
unfused except the explicit __builtin_fma calls (see csrc/abn_device.hpp).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libabneutral_hip.so"
SOURCES = [CSRC / "abn_api.hip", CSRC / "abn_pairwise.hip", CSRC / "abn_multi.hip"]
DEPS = [*sorted(CSRC.glob("*.hpp")), CSRC / "abn_philox.h", PKG.parent / "include" / "abneutral.h"]
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-fPIC",
    "-shared",
    "-Wall",
    "-ldl",
]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the ABneutral HIP library cannot be built")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + DEPS)


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), *HIPCC_FLAGS, "-o", str(LIB), *map(str, SOURCES)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=str(PKG))
    return LIB


KNOBS_LIB = PKG.parent / "build" / "libabn_knobs.so"


def build_knobs(force: bool = False, verbose: bool = False) -> Path:
    """The same sources with -DABN_MEASUREMENT_KNOBS (environment switches for the sweeps under scripts/ and the fault
    injection of tests/test_gpu_parity.py::test_lost_fifo_entry_is_an_error_at_sync).  Never loaded by the product: the
    package takes it only through ABNEUTRAL_HIP_LIB."""
    if not force and KNOBS_LIB.exists() and all(p.stat().st_mtime <= KNOBS_LIB.stat().st_mtime for p in SOURCES + DEPS):
        return KNOBS_LIB
    KNOBS_LIB.parent.mkdir(exist_ok=True)
    cmd = [hipcc_path(), *HIPCC_FLAGS, "-DABN_MEASUREMENT_KNOBS", "-o", str(KNOBS_LIB), *map(str, SOURCES)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=str(PKG))
    return KNOBS_LIB


HOST = PKG / "host"
CLI = PKG / "bin" / "alphabeta"
META_CLI = PKG / "bin" / "metaprofile_alphabeta"
REF_TESTS = PKG / "bin" / "reference_tests"
PEDIGREE_LIB = PKG / "libabneutral_host.so"


def build_host(force: bool = False, verbose: bool = False) -> Path:
    """The C++ host layer: the `alphabeta` CLI (reference flags and output files) and a small shared
    library exposing Pedigree::build to the tests.  Both link libabneutral_hip.so via $ORIGIN rpaths."""
    build_hip()
    srcs = [HOST / "alphabeta_cli.cpp", HOST / "alphabeta.hpp", HOST / "pedigree_build.hpp", HOST / "host_capi.cpp",
            HOST / "metaprofile.hpp", HOST / "metaprofile_cli.cpp", HOST / "reference_tests.cpp"]
    newest = max(p.stat().st_mtime for p in srcs)
    CLI.parent.mkdir(exist_ok=True)
    common = ["-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-I", str(PKG.parent / "include")]
    if force or not CLI.exists() or CLI.stat().st_mtime < newest:
        cmd = [hipcc_path(), *common, "-o", str(CLI), str(HOST / "alphabeta_cli.cpp"), "-L", str(PKG),
               "-labneutral_hip", "-Wl,-rpath,$ORIGIN/.."]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=str(PKG))
    if force or not META_CLI.exists() or META_CLI.stat().st_mtime < newest:
        cmd = [hipcc_path(), *common, "-o", str(META_CLI), str(HOST / "metaprofile_cli.cpp"), "-L", str(PKG),
               "-labneutral_hip", "-Wl,-rpath,$ORIGIN/.."]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=str(PKG))
    if force or not REF_TESTS.exists() or REF_TESTS.stat().st_mtime < newest:
        cmd = [hipcc_path(), *common, "-o", str(REF_TESTS), str(HOST / "reference_tests.cpp"), "-L", str(PKG),
               "-labneutral_hip", "-Wl,-rpath,$ORIGIN/.."]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=str(PKG))
    if force or not PEDIGREE_LIB.exists() or PEDIGREE_LIB.stat().st_mtime < newest:
        cmd = [hipcc_path(), *common, "-fPIC", "-shared", "-o", str(PEDIGREE_LIB), str(HOST / "host_capi.cpp"), "-L",
               str(PKG), "-labneutral_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=str(PKG))
    return CLI


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
