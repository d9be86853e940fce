"""Checks on the EMITTED gfx950 ISA (hipcc -S --cuda-device-only: cross-compiles without a GPU, ~45 s).

The time-sliced persistent kernel hands a parked chain's state to a group on another XCD through memory: the state
stores (write-through, sc1) must have completed before the FIFO entry (another sc1 store) is published.  VERDICT r02
found that the workgroup-scope release fence the source used emits NO instruction and that only a compiler-inserted
wait (for an address dependency) ordered the two.  The wait is now explicit (inline `s_waitcnt vmcnt(0)` with a marker
comment); this test holds it in place: in every abn_fit_refill_kernel instantiation the marker sits after the sc1 state
stores and before the sc1 entry store, and no trap instruction is left in the kernel (a lost FIFO entry sets an
error word instead of aborting the process: the C-ABI never crashes).
"""
import hashlib
import re
import subprocess
import tempfile
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "alphabeta_rs_amd" / "csrc"
MARK = "abn: parked state written through"


@pytest.fixture(scope="module")
def device_isa():
    from alphabeta_rs_amd import build as B

    try:
        B.hipcc_path()
    except RuntimeError as e:     # a CPU-only box without ROCm: nothing to check here (the GPU tier builds with hipcc)
        pytest.skip(str(e))
    h = hashlib.sha1()
    for f in sorted(CSRC.glob("*")):
        h.update(f.read_bytes())
    out = Path(tempfile.gettempdir()) / f"abn_api_{h.hexdigest()[:16]}.s"
    if not out.exists():
        flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC", "-ldl")]
        subprocess.run([B.hipcc_path(), "-S", "--cuda-device-only", *flags, "-Wno-unused-command-line-argument", "-o",
                        str(out), str(CSRC / "abn_api.hip")], check=True, cwd=str(CSRC))
    return out.read_text()


def functions(isa, needle):
    """{mangled name: [instruction lines]} of the functions whose name contains `needle`"""
    out = {}
    for m in re.finditer(r"^(_Z\w*" + needle + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end", isa, re.S | re.M):
        out[m.group(1)] = [ln.strip() for ln in m.group(2).splitlines() if ln.strip() and not ln.strip().startswith(";")]
    return out


def test_parked_state_is_written_through_before_the_entry_is_published(device_isa):
    fns = functions(device_isa, "abn_fit_refill_kernel")
    assert len(fns) >= 12, sorted(fns)            # 4 lane counts x up to 4 rows-per-lane variants
    for name, body in fns.items():
        marks = [i for i, ln in enumerate(body) if MARK in ln]
        assert len(marks) == 1, (name, marks)
        assert body[marks[0]].startswith("s_waitcnt vmcnt(0)"), body[marks[0]]
        before, after = body[: marks[0]], body[marks[0] + 1:]
        # ORDER, not a count tied to one code generation: sc1 state stores exist before the wait (the 32 doubles: the
        # compiler may merge or split them), the sc1 entry store and no further state store come after it
        state_stores = [ln for ln in before if ln.startswith("global_store_dwordx") and " sc1" in ln]
        assert state_stores, name
        entry = [i for i, ln in enumerate(after) if ln.startswith("global_store_dword ") and " sc1" in ln]
        assert entry, name                         # the FIFO entry: a 32-bit sc1 store behind the wait
        assert not any(ln.startswith("global_store_dwordx") and " sc1" in ln for ln in after[: entry[0]]), name
        assert not any(ln.startswith("s_trap") for ln in body), name


def test_no_trap_in_any_fit_kernel(device_isa):
    for needle in ("abn_fit_kernel", "abn_fit_spec_kernel", "abn_fit_refill_kernel"):
        for name, body in functions(device_isa, needle).items():
            assert not any(ln.startswith("s_trap") for ln in body), name


def test_no_device_function_is_called_and_nothing_spills_in_the_speculative_kernel(device_isa):
    """Round 4: after a restructuring hipcc stopped inlining the evaluation lambda of abn_fit_spec_kernel<8, ...> — the kernel
    then CALLED it (s_swappc_b64), every LDS access of the callee became a flat load, and the 351-row golden pedigree lost
    30 % without any test noticing (the results are the same).  No fit kernel may contain a call, no lambda may survive as
    a function of its own, and the speculative kernel — whose four-workgroups-per-CU residency rests on it — has no scratch."""
    assert not re.search(r"^_ZZN3abn\w*:", device_isa, re.M), "a device lambda was emitted as a function (not inlined)"
    for needle in ("abn_fit_kernel", "abn_fit_spec_kernel", "abn_fit_refill_kernel", "abn_cost_kernel"):
        for name, body in functions(device_isa, needle).items():
            assert not any(ln.startswith("s_swappc_b64") for ln in body), name
    for m in re.finditer(r"^\s*\.amdhsa_kernel (_ZN3abn19abn_fit_spec_kernel\w+)\n(.*?)\.end_amdhsa_kernel", device_isa, re.S | re.M):
        priv = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2))
        vgpr = re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2))
        assert priv and int(priv.group(1)) == 0, (m.group(1), priv and priv.group(1))
        assert vgpr and int(vgpr.group(1)) <= 168, (m.group(1), vgpr and vgpr.group(1))
        if "ILi1E" in m.group(1) or "ILi2E" in m.group(1):     # up to two rows per lane: four workgroups per CU
            assert int(vgpr.group(1)) <= 128, (m.group(1), vgpr.group(1))
