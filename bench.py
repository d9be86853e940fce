#!/usr/bin/env python3
"""bench.py — ABneutral model fits/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path over one batch: phase A (S random-start Nelder-Mead fits per
window, src/ab_neutral.rs:37-78), selection (src/ab_neutral.rs:83-135), phase B (B residual-bootstrap
refits per window, src/boot_model.rs:41-100) and, for N > 1 GPUs, the RCCL all-gather of the bootstrap
tables.  Inputs (pedigree, observations, start simplices, bootstrap index buffer) are resident in HBM
before the timed region.  Each rank owns the windows [rank*Wr, (rank+1)*Wr) (weak scaling: per-GPU work
is fixed) and there is no data-path collective other than the final gather.

Workloads:  c3 (default) synthetic 100-edge / 8-generation pedigree, N=105 rows, 10 starts
+ 10000 bootstraps per window, one window per GPU;  c2 bundled pedigree (6 rows) x 1000 bootstraps;
c4 the C3 topology, 25 windows x 1000 bootstraps per GPU (= 200 windows over 8 GPUs);  g351 the
reference's golden pedigree (351 rows, T=32) x 1000 bootstraps;  c5s a single-GPU shard of C5 (20100 rows,
T=125, 4096 bootstraps) in stream mode.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
N_SIMDS, CLOCK_HZ = 256 * 4, 2.4e9   # MI355X: 256 CUs x 4 SIMDs; the clock the FP64 vector peak is quoted at


def make_workload(name: str, rank: int, world: int):
    """returns dict(gens Nx3, D [W,N], p0uu [W], S, B, label)"""
    from alphabeta_rs_amd import synthetic

    if name == "c3":
        # BASELINE C3 on every rank; for N > 1 GPUs the bootstraps of the one window are sharded: rank r owns
        # bootstraps [r*10000, (r+1)*10000) (weak scaling: per-GPU work fixed), phase A (10 starts) is repeated
        # on every rank (same inputs -> same bits) so that no exchange precedes phase B
        ped, p = synthetic.c3_pedigree()
        return dict(gens=ped[:, :3], D=ped[:, 3][None, :], p0=np.array([p]), S=10, B=10000, wr=1, shard="bootstraps",
                    label="C3: synthetic 100-edge/8-generation pedigree (N=105 rows, T=8, K=47 distinct triples), "
                          "10 starts + 10000 bootstraps per GPU (bootstraps of one window sharded over the GPUs)")
    if name == "c4":
        gens, D, p0, _ = synthetic.c4_windows(25, window_offset=25 * rank)
        return dict(gens=gens, D=D, p0=p0, S=10, B=1000, wr=25,
                    label="C4 shard: C3 topology, 25 windows x (10 starts + 1000 bootstraps) per GPU "
                          "(200 windows over 8 GPUs)")
    if name == "c4s":
        # BASELINE C4 as named: 200 windows in ALL, dealt to the GPUs in contiguous blocks -> strong scaling
        from alphabeta_rs_amd.distributed import shard_range
        w0, wn = shard_range(200, world, rank)
        gens, D, p0, _ = synthetic.c4_windows(wn, window_offset=w0)
        return dict(gens=gens, D=D, p0=p0, S=10, B=1000, wr=wn, w0=w0, scaling="strong",
                    label="C4: C3 topology, 200 windows x (10 starts + 1000 bootstraps) in all, windows sharded over the "
                          "GPUs (strong scaling)")
    if name == "c5":
        # BASELINE C5's per-GPU shard at its real size: 500 windows over 8 GPUs -> 63 windows per GPU
        gens, D, p0, _ = synthetic.c5_windows(63, window_offset=63 * rank)
        return dict(gens=gens, D=D, p0=p0, S=10, B=5000, wr=63,
                    label="C5 per-GPU shard at full size: deep pedigree (N=20100 rows, T=125, K=950), 63 windows x "
                          "(10 starts + 5000 bootstraps); rows streamed (materialised bootstrap observations, 8 B/row, "
                          "re-read from HBM every evaluation; 25 GB of indices + 51 GB of observations resident)")
    if name in ("ref1000_c3", "ref1000_g351"):
        # the reference's DEFAULT shape: `alphabeta -i 1000` sets n_starts = n_boot = iterations = 1000
        # (src/alphabeta.rs:33-54, src/arguments.rs:93-114), one pedigree
        base = make_workload("c3" if name.endswith("c3") else "g351", rank, world)
        base.update(S=1000, B=1000, label="reference default `-i 1000` (1000 starts + 1000 bootstraps, "
                    "src/alphabeta.rs:33-54) on " + base["label"].split(",")[0])
        return base
    if name == "mp":
        # the reference's default metaprofile shape (src/cli/metaprofile.rs:40-44 with -s 1: 3 regions x 100
        # windows; --iterations 100 -> 100 starts + 100 bootstraps per window), on the C3 topology
        gens, D, p0, _ = synthetic.c4_windows(300, window_offset=300 * rank)
        return dict(gens=gens, D=D, p0=p0, S=100, B=100, wr=300,
                    label="metaprofile shape: C3 topology, 300 windows x (100 starts + 100 bootstraps) per GPU")
    if name == "c5s":
        ped, p = synthetic.c5_pedigree()
        rng = np.random.Generator(np.random.Philox(key=synthetic.SEED + 500 + rank))
        D = np.maximum(ped[:, 3] + (rng.normal(0.0, synthetic.NOISE_SD, ped.shape[0]) if world > 1 else 0.0), 0.0)
        return dict(gens=ped[:, :3], D=D[None, :], p0=np.array([p]), S=4, B=4096, wr=1,
                    label="C5 shard: synthetic deep pedigree, 8 lineages x 125 generations (N=20100 rows, T=125, "
                          "K=950), 4 starts + 4096 bootstraps, 1 window per GPU; rows streamed (the materialised "
                          "bootstrap observations, 8 B/row, re-read from HBM every evaluation)")
    if name == "c5p":
        # the stream probe of every bench line (stream_probe) as a workload of its own, so that scripts/profile_round.sh can
        # take its counters: the C5 pedigree, 2 starts + 8192 bootstraps, one window
        ped, p = synthetic.c5_pedigree()
        return dict(gens=ped[:, :3], D=ped[:, 3][None, :], p0=np.array([p]), S=2, B=8192, wr=1,
                    label="C5 stream probe: deep pedigree (N=20100 rows, T=125, K=950), 2 starts + 8192 bootstraps, rows streamed")
    if name in ("c2", "g351"):
        # fixtures are data (tests/golden); read without the oracle package
        fn = "pedigree_generated.txt" if name == "c2" else "pedigree.txt"
        rows = [[float(t) for t in ln.replace("\t", " ").split()] for ln in
                (ROOT / "tests" / "golden" / fn).read_text().splitlines()[1:] if ln.strip()]
        ped = np.asarray(rows)
        p = 0.6554051647850447 if name == "c2" else 0.75
        lab = ("C2: bundled data/nodelist+edgelist pedigree (N=6 rows, T=4), 10 starts + 1000 bootstraps"
               if name == "c2" else "G: reference golden pedigree data/pedigree.txt (N=351 rows, T=32, K=10), "
                                    "10 starts + 1000 bootstraps")
        return dict(gens=ped[:, :3], D=ped[:, 3][None, :], p0=np.array([p]), S=10, B=1000, wr=1, label=lab)
    raise SystemExit(f"unknown workload {name}")


def usable_cpus(limit):
    """Host threads this process may really run at once: the affinity mask and the cgroup CPU quota, whichever
    is smaller (a GPU box hands one GPU's share of the host's cores to the job)."""
    n = min(limit, len(os.sched_getaffinity(0)))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(wl, model, pred, resid, lanes, seed, evals_per_fit, budget_s=12.0):
    """The CPU oracle (a port of the reference's algorithm: per-row repeated 3x3 multiplication,
    src/divergence.rs:51-90, one OpenMP thread per fit like the rayon par_iter) on a bounded sample.
    A probe of one truncated fit per thread sizes the sample; pedigrees whose full fits do not fit the
    budget are timed on truncated fits and converted with the measured evaluations per fit."""
    import oracle as O

    O.build()
    ped = np.concatenate([wl["gens"], wl["D"][0][:, None]], axis=1)
    p0 = float(wl["p0"][0])
    cores = usable_cpus(O.max_threads())

    def run(nb, iters, table):
        t0 = time.perf_counter()
        _, res = O.boot_model(ped, model, pred, resid, p0, p0, 1.0, seed, 0, 0, nb, max_iters=iters, lanes=1, table=table,
                              threads=cores)
        return time.perf_counter() - t0, float(res["evals"].sum())

    nb_next, it_next = cores, 4                         # probe: 5 + <= 8 evaluations per fit and thread
    for _ in range(5):
        nb, iters = nb_next, it_next                    # (nb, iters) always belong to the run that gave (dt, ev)
        dt, ev = run(nb, iters, False)
        if dt >= 0.75 * budget_s:
            break
        target = ev / dt * budget_s                     # evaluations the budget buys at the measured rate
        if target >= 4.0 * cores * evals_per_fit:       # whole fits: >= 4 per thread
            nb_next, it_next = int(min(8192.0 * cores, target / evals_per_fit)), 1000
        else:                                           # truncated fits, one per thread
            nb_next, it_next = cores, max(4, int(target / cores / 1.7) - 5)
        if (nb_next, it_next) == (nb, iters):
            break
    dt_t, ev_t = run(nb, iters, True)
    whole = iters == 1000
    fits = nb / dt if whole else ev / dt / evals_per_fit
    fits_t = nb / dt_t if whole else ev_t / dt_t / evals_per_fit
    what = (f"{nb} residual-bootstrap refits of window 0" if whole else
            f"{nb} residual-bootstrap refits of window 0 truncated at {iters} Nelder-Mead iterations ({int(ev)} "
            f"evaluations; fits/s = evaluations/s / {evals_per_fit:.1f} evaluations per fit measured on the GPU run)")
    return {
        "value": fits, "unit": "fits/s", "cores": cores, "kind": "port",
        "sample": f"{what} (same pedigree, same inputs), reference-shaped divergence (3 matrix_power per row), "
                  f"{cores} OpenMP threads, {dt:.1f} s",
        "evals_per_s": ev / dt,
        "power_table_variant_fits_per_s": fits_t,
    }


def kernel_source_sha1() -> str:
    """content hash of the device/host sources of libabneutral_hip.so: a PMC traffic figure is only quoted for
    the kernels it was measured on"""
    import hashlib

    h = hashlib.sha1()
    for f in sorted((ROOT / "alphabeta_rs_amd" / "csrc").glob("*")):
        if f.name == "abn_multi.hip":   # device orchestration over the same plans: no kernel, no launch geometry
            continue
        if f.suffix in {".hip", ".hpp", ".h"}:
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()


def measured_traffic(workload: str):
    """The committed rocprofv3 PMC passes of a workload (profiles/rNN_pmc_<workload>.json, written per phase by
    scripts/profile_round.sh + profiles/summarize_rocprof.py) — used only when that profile was taken on the kernel
    sources of this very build; a stale profile yields None and says so.  Returns (phases | None, source note)."""
    sha = kernel_source_sha1()
    for f in sorted((ROOT / "profiles").glob(f"r*_pmc_{workload}.json"), reverse=True):
        try:
            j = json.loads(f.read_text())
        except Exception:
            continue
        if j.get("workload") == workload and j.get("source_sha1") == sha and "phases" in j:
            return j["phases"], {"file": f"profiles/{f.name}", "source_sha1": sha, "collected": j.get("collected")}
    return None, {"note": "no rocprofv3 PMC profile of this build's kernel sources under profiles/ "
                          f"(source_sha1 {sha[:12]}); run scripts/profile_round.sh", "source_sha1": sha}


def valu_issue(ph, kern_s):
    """How busy the vector pipe was over a launch, from the SQ counters of the stamped profile of that launch:
    SQ_INSTS_VALU wavefront instructions, SQ_ACTIVE_INST_VALU quad-cycles a wavefront had a vector instruction in
    execution.  cycles_per_inst = 4 ACTIVE / INSTS is the MEASURED issue cost of this kernel's instruction mix (f64
    arithmetic 4 cycles, 32-bit moves / selects / DPP fewer, divisions and square roots more) — the peak instruction rate
    follows from it instead of from a flat 4 cycles; busy_frac = the share of SIMD-cycles of the launch spent issuing."""
    if not ph or "SQ_INSTS_VALU" not in ph or kern_s <= 0:
        return None
    nv = ph["SQ_INSTS_VALU"]
    out = {"insts_per_launch": nv, "achieved_ginst_per_s": nv / kern_s / 1e9, "kernel": ph.get("kernel")}
    act = ph.get("SQ_ACTIVE_INST_VALU")
    if act:
        cpi = 4.0 * act / nv
        peak = N_SIMDS * CLOCK_HZ / cpi
        out.update({"issue_cycles_per_inst": cpi, "peak_ginst_per_s": peak / 1e9, "frac": nv / kern_s / peak,
                    "what": "wavefront vector instructions / s against 1024 SIMDs x 2.4 GHz / (measured issue cycles per "
                            "instruction = 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU); frac = share of the launch's "
                            "SIMD-cycles spent issuing vector instructions (tail of long chains included)"})
    else:
        peak = N_SIMDS * CLOCK_HZ / 4.0
        out.update({"peak_ginst_per_s": peak / 1e9, "frac": nv / kern_s / peak,
                    "what": "wavefront vector instructions / s against 1024 SIMDs x 2.4 GHz / 4 (no SQ_ACTIVE_INST_VALU pass)"})
    for k_in, k_out in (("SQ_INSTS_SALU", "scalar_insts_per_launch"), ("SQ_INSTS_LDS", "lds_insts_per_launch"),
                        ("SQ_WAVES", "wavefronts")):
        if k_in in ph:
            out[k_out] = ph[k_in]
    if "SQ_WAVE_CYCLES" in ph and ph["SQ_WAVE_CYCLES"] > 0:   # where the resident wavefront-cycles went
        wc = ph["SQ_WAVE_CYCLES"]
        out["of_wavefront_cycles"] = {k: ph[c] / wc for k, c in (("issuing", "SQ_ACTIVE_INST_ANY"), ("waiting_to_issue", "SQ_WAIT_INST_ANY"),
                                                                 ("waiting_on_counter", "SQ_WAIT_ANY")) if c in ph}
    return out


def pcie_inclusive(A, ctx, wl, opts, reps=5):
    """The drop-in entry points on HOST buffers (abn_ab_neutral_run + abn_boot_model_run: allocation, H2D, start
    simplices, index generation, kernels, D2H) for a one-window workload: fits/s including PCIe."""
    ped = np.concatenate([wl["gens"], wl["D"][0][:, None]], axis=1)
    p0 = float(wl["p0"][0])
    S, B = wl["S"], wl["B"]

    def once():
        model, pred, resid, _ = ctx.ab_neutral_run(ped, p0, p0, 1.0, S, options=opts)
        ctx.boot_model_run(ped, model, pred, resid, p0, p0, 1.0, B, options=opts)

    once()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    dt = (time.perf_counter() - t0) / reps
    return {"fits_per_s": (S + B) / dt, "ms_per_step": 1e3 * dt, "reps": reps,
            "what": "abn_ab_neutral_run + abn_boot_model_run on host buffers, end to end"}


def quick_workload(A, ctx, name, seed, steps=3, **options):
    """a few steps of another BASELINE configuration on this GPU, so that its number is driver-run too"""
    wl = make_workload(name, 0, 1)
    plan = A.Plan(ctx, wl["gens"], wl["wr"], wl["S"], wl["B"], options=A.default_options(seed=seed, **options))
    plan.set_windows(wl["D"], wl["p0"])
    plan.run()
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run()
    plan.sync()
    dt = (time.perf_counter() - t0) / steps
    cnt = plan.counters()
    kms = plan.kernel_ms()
    kern = plan.last_kernels()
    out = plan.download(allow_failed_windows=True)
    plan.close()
    r = {"workload": wl["label"], "fits_per_s": cnt["fits"] / dt, "ms_per_step": 1e3 * dt, "steps": steps,
         "candidate_evals_per_s": (cnt["evals"] - cnt["evals_skipped"]) / dt, "kernel_ms": kms,
         "kernels": {k: f"{v[0]} ({v[1]} lanes per chain)" for k, v in kern.items()}}
    phases, src = measured_traffic(name)
    if phases:   # counters of the stamped rocprofv3 profile of this workload (scripts/profile_round.sh)
        r["pmc"] = {"source": src, **{p: {"traffic": ph.get("hbm_bytes_per_launch"), "valu_issue": valu_issue(ph, kms[p] * 1e-3)}
                                       for p, ph in phases.items() if p in kms}}
    if out["info_a"] is not None:   # starts that reached argmin's fixed point (ABN_FIT_MAX_ITERS after a rejected contraction)
        r["starts_at_max_iters"] = int((out["info_a"]["status"] == 1).sum())
        r["boots_at_max_iters"] = int((out["info_b"]["status"] == 1).sum())
        r["evals_not_executed"] = {"starts": cnt["evals_skipped_starts"], "boot": cnt["evals_skipped_boot"]}
    return r


def pairwise_bench(A, ctx, shapes=((15, 4_000_000), (15, 32_000_000), (50, 32_000_000), (50, 2_000_000)), reps=5):
    """SURVEY §8(f).1, `DMatrix::from` (src/pedigree.rs:210-261): pairwise divergence of n samples over L aligned
    sites, codes RESIDENT in HBM (one byte per sample and site).  One pass over the codes is the algorithmic traffic:
    n*L bytes / kernel time against the HBM peak; the pair arithmetic (n(n-1)/2 pairs x L sites) is reported as
    site-pairs/s next to it.  Shapes: the two of round 2 (the last one, 50 x 2 M, is the `pw` line's value) and the same
    sample counts at genome scale (32 M sites), where a workgroup sees ~40 tiles instead of 2.5."""
    import torch

    out = []
    pw_prof, pw_src = measured_traffic("pw")     # {"<n>x<L>": bytes} written by scripts/pw_profile.sh
    for n, L in shapes:
        g = torch.Generator(device="cuda")
        g.manual_seed(1234 + n)
        codes = torch.randint(0, 3, (n, L), dtype=torch.uint8, device="cuda", generator=g)
        codes |= (torch.rand((n, L), device="cuda", generator=g) < 0.05).to(torch.uint8) * 0x80  # 5 % filtered sites
        npairs = n * (n - 1) // 2
        diff = torch.zeros(npairs, dtype=torch.int64, device="cuda")
        both = torch.zeros(npairs, dtype=torch.int64, device="cuda")
        dval = torch.zeros(npairs, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ctx.pairwise_divergence_dev(codes.data_ptr(), n, L, diff.data_ptr(), both.data_ptr(), dval.data_ptr())
        ms = [ctx.pairwise_divergence_dev(codes.data_ptr(), n, L, diff.data_ptr(), both.data_ptr(), dval.data_ptr())
              for _ in range(reps)]
        best, avg = min(ms), sum(ms) / len(ms)
        # property check at full size: every pair compares the sites valid in both samples, 0 <= diff <= 2 * both
        ok = bool(((diff >= 0) & (diff <= 2 * both) & (both > 0) & (both <= L)).all().item())
        gbs = n * L / (avg * 1e-3) / 1e9
        out.append({"samples": n, "sites": L, "pairs": npairs, "kernel_ms_avg": avg, "kernel_ms_min": best,
                    "code_bytes": n * L, "achieved_GBps": gbs, "peak_GBps": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
                    "traffic": (pw_prof or {}).get(f"{n}x{L}"),   # HBM bytes of the scan launch (FETCH_SIZE x 2, stamped profile)
                    "site_pairs_per_s": npairs * L / (avg * 1e-3), "sane": ok})
        del codes
    return {"kernel": "abn_pairwise_mx_kernel (v_mfma_i32_16x16x64_i8 Gram products) + abn_pairwise_reduce_tiles_kernel", "bound": "hbm", "unit": "GB/s",
            "what": "n*L code bytes (one pass) / HIP-event time of the call's kernels, inputs resident in HBM",
            "traffic_source": pw_src, "shapes": out}


def stream_probe(A, ctx, seed, steps=2, every=5, B=8192, tag="C5 shard"):
    """Short run of the HBM-facing configuration (a C5 shard, stream mode: the bootstrap observations of a fit
    are re-read on every evaluation) so that the bench line also carries the roofline of the kernel variant the
    HBM roof actually applies to.  Phase B only is timed (HIP events on the launch stream).

    What the rate is a rate OF: the kernel streams 8N+40 B per evaluation from beyond L2.  The 256 MiB Infinity
    Cache (MALL) sits between L2 and DRAM, and the co-resident working set (co-resident chains x 8N bytes) decides
    how much of that stream it can serve; `beyond_l2_*` is therefore an L2-miss rate (fabric traffic), equal to the
    DRAM rate only when the working set is far above 256 MiB (--stream-sweep measures both ends)."""
    from alphabeta_rs_amd import synthetic

    ped, p = synthetic.c5_pedigree(every=every)
    N = ped.shape[0]
    plan = A.Plan(ctx, ped[:, :3], 1, 2, B, options=A.default_options(seed=seed))
    plan.set_windows(ped[:, 3][None, :], np.array([p]))
    plan.run()
    ms = 0.0
    for _ in range(steps):
        plan.run_phase(1)
        ms += plan.kernel_ms()["fit_boot"]
    ms /= steps
    out = plan.download()
    evals_b = int(out["info_b"]["evals"].sum()) - plan.counters()["evals_skipped_boot"]  # evaluations executed
    lanes = int(out["info_b"]["lanes"][0, 0])
    K = len({tuple(r) for r in ped[:, :3].astype(int).tolist()})
    plan.close()
    # co-resident chains: one wavefront per chain, limited by LDS (9(T+1) + K + 4 doubles per chain) and by the
    # kernel's two wavefronts per SIMD
    lds_chain = (9 * 126 + ((K + 1) & ~1) + 4) * 8
    per_cu = max(1, min(8, (160 * 1024) // lds_chain))
    resident = min(B, 256 * per_cu)
    alg = evals_b * (4 * N + 40) + B * 112 + N * 18
    streamed = evals_b * (8 * N + 40) + B * (12 * N + 112)
    achieved = alg / (ms * 1e-3) / 1e9
    # counters of the same launch (workload `c5p` = this probe) from the stamped profile, if there is one for this build
    phases, src = measured_traffic("c5p") if (every == 5 and B == 8192) else (None, {"note": "no profile of this variant"})
    ph_b = (phases or {}).get("fit_boot")
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": ph_b.get("hbm_bytes_per_launch") if ph_b else None, "traffic_source": src,
            "kernel": f"abn_fit_kernel<G={lanes & 0xff}, stream> phase B (+ abn_make_dstar_kernel)",
            "kernel_ms": ms, "algorithmic_bytes_per_launch": alg, "streamed_bytes_per_launch": streamed,
            "beyond_l2_GBps": streamed / (ms * 1e-3) / 1e9,
            "beyond_l2_frac_of_hbm_peak": streamed / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "coresident_chains": resident, "coresident_working_set_MiB": resident * 8 * N / 2**20,
            "infinity_cache_MiB": 256,
            "workload": f"{tag}: N={N} rows, T=125, K={K}, {B} bootstraps, {evals_b} evaluations; `achieved` at "
                        "SURVEY's 4N+40 B per evaluation; `beyond_l2_*` = the 8N+40 B per evaluation the kernel "
                        "streams past L2 (HBM + Infinity Cache hits: a working set below ~256 MiB can be served "
                        "partly by the Infinity Cache, so only the large-working-set figure is a DRAM rate)"}


def c5_full_shard(A, ctx, seed, steps=1):
    """BASELINE C5's per-GPU shard at its real size (63 windows x (10 starts + 5000 bootstraps) x 20100 rows: 25 GB of
    bootstrap indices and 51 GB of materialised observations resident in HBM), one warm-up pass and `steps` timed
    passes; the roofline of its phase-B launch like stream_probe's."""
    wl = make_workload("c5", 0, 1)
    N, W, S, B = wl["gens"].shape[0], wl["wr"], wl["S"], wl["B"]
    plan = A.Plan(ctx, wl["gens"], W, S, B, options=A.default_options(seed=seed))
    dev_bytes = plan.device_bytes()
    plan.set_windows(wl["D"], wl["p0"])
    plan.run()
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run()
    plan.sync()
    dt = (time.perf_counter() - t0) / steps
    kms = plan.kernel_ms()
    cnt = plan.counters()
    kern = plan.last_kernels()
    out = plan.download()
    plan.close()
    evals_b = int(out["info_b"]["evals"].sum()) - cnt["evals_skipped_boot"]
    fits_b = W * B
    ms = kms["fit_boot"]
    alg = evals_b * (4 * N + 40) + fits_b * 112 + W * N * 16 + N * 2
    streamed = evals_b * (8 * N + 40) + fits_b * (12 * N + 112)
    sane = bool(np.isfinite(out["raw"]).all() and (out["best_start"] >= 0).all())
    phases, src = measured_traffic("c5")
    ph_b = (phases or {}).get("fit_boot")
    return {"workload": wl["label"], "traffic_source": src, "fits_per_s": cnt["fits"] / dt, "ms_per_step": 1e3 * dt, "steps": steps,
            "candidate_evals_per_s": (cnt["evals"] - cnt["evals_skipped"]) / dt, "kernel_ms": kms,
            "kernels": {k: f"{v[0]} ({v[1]} lanes per chain)" for k, v in kern.items()},
            "plan_device_bytes": dev_bytes, "all_rows_finite": sane,
            "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": ph_b.get("hbm_bytes_per_launch") if ph_b else None, "kernel_ms": ms,
                         "kernel": "abn_fit_kernel<64, stream> phase B (+ abn_make_dstar_kernel)",
                         "algorithmic_bytes_per_launch": alg, "streamed_bytes_per_launch": streamed,
                         "beyond_l2_GBps": streamed / (ms * 1e-3) / 1e9,
                         "beyond_l2_frac_of_hbm_peak": streamed / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "evaluations": evals_b}}


def stream_sweep(A, ctx, seed):
    """The stream kernel with a co-resident working set far below and far above the 256 MiB Infinity Cache at a
    comparable number of co-resident wavefronts (the LDS scratch per chain decides how many fit): shorter / longer rows
    of the same deep pedigree.  If the large-working-set rate matches the standard probe's, that probe's stream comes
    from DRAM; if it is lower, the difference is what the Infinity Cache served."""
    return {"small_working_set": stream_probe(A, ctx, seed, steps=2, every=10, B=8192,
                                              tag="deep pedigree sampled every 10th generation"),
            "large_working_set": stream_probe(A, ctx, seed, steps=1, every=3, B=4096,
                                              tag="deep pedigree sampled every 3rd generation")}


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: one child process per GPU (fresh interpreters, started
    before this process initialises HIP; torch.cuda.device_count() does not), RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*
    in their environment, rank 0's JSON line passed through.  Fails loudly when the node has fewer GPUs than asked for
    (--backend gloo may oversubscribe one GPU to rehearse the multi-rank path)."""
    import socket
    import subprocess

    import torch

    ndev = torch.cuda.device_count()
    if ndev < args.gpus and args.backend == "nccl":
        print(f"bench.py: --gpus {args.gpus} asked for but only {ndev} GPU(s) are visible: no number is reported "
              "(RCCL needs one GPU per rank)", file=sys.stderr)
        return 2
    if ndev < 1:
        print("bench.py needs an MI355X: the ABneutral path has no CPU fallback", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [pr.wait() for pr in procs[1:]]
    if any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return next(c for c in codes if c) or 1
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]   # the ONE JSON line; libraries may chat on stdout
    if len(lines) != 1:
        print(f"bench.py: rank 0 printed {len(lines)} JSON lines", file=sys.stderr)
        return 1
    print(lines[0])
    return 0


def single_process_main(args) -> None:
    """`bench.py --single-process [--devices 0,1,..]`: the OTHER N > 1 form — one process, one host thread, a plan per
    device, the bootstrap tables gathered by RCCL inside the library (abn_multi_*, csrc/abn_multi.hip; what the
    `--devices` flag of the CLIs uses).  Same JSON contract; n_gpus = number of devices.  The job is the per-GPU
    workload times the number of devices (weak scaling), or `c4s`'s fixed 200 windows (strong)."""
    import alphabeta_rs_amd as A
    from alphabeta_rs_amd import synthetic

    # stdout carries exactly ONE line, the JSON: RCCL prints a version banner on file descriptor 1 when it is bound
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    n = len(devices)
    if args.devices and any(x == "--gpus" or x.startswith("--gpus=") for x in sys.argv) and args.gpus != n:   # both given explicitly: they must agree
        raise SystemExit(f"--gpus {args.gpus} and --devices {args.devices} ({n} devices) disagree")
    A.load_library(build_if_missing=True)
    if A.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the ABneutral path has no CPU fallback")
    seed = 20260101
    strong = args.workload == "c4s"
    if args.workload == "c3":                    # one window, the bootstraps sharded: 10000 per device
        ped, p = synthetic.c3_pedigree()
        gens, D, p0, S, B, W = ped[:, :3], ped[:, 3][None, :], np.array([p]), 10, 10000 * n, 1
        label = f"C3 pedigree, 10 starts + {B} bootstraps (10000 per device; bootstraps sharded)"
    elif args.workload in ("c4", "c4s"):
        W = 200 if strong else 25 * n
        gens, D, p0, _ = synthetic.c4_windows(W)
        S, B = 10, 1000
        label = f"C4: C3 topology, {W} windows x (10 starts + 1000 bootstraps), windows sharded over the devices"
    else:
        raise SystemExit("--single-process times the workloads c3, c4 and c4s")
    opts = A.default_options(seed=seed, lanes_per_chain=args.lanes, strict_order=1 if args.strict_order else (-1 if args.tree_order else 0),
                             no_fixed_point_skip=1 if args.execute_stuck_fits else 0)
    m = A.MultiPlan(devices, gens, W, S, B, options=opts)
    m.set_windows(D, p0)
    for _ in range(args.warmup):
        m.run()
    m.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.run()
    m.sync()
    elapsed = time.perf_counter() - t0
    cnt = m.counters()
    kms = [m.kernel_ms(i) for i in range(n)]
    out = m.download()
    N = gens.shape[0]
    # sanity at full size: every row finite; the table does not depend on how it was sharded (tests/test_gpu_multi.py
    # compares it with a plain plan byte for byte)
    sane = bool(np.isfinite(out["raw"]).all())
    sh = m.shard(0)
    fits_b0 = sh["n_windows"] * sh["n_boot"]
    kern_s = max(k["fit_boot"] for k in kms) * 1e-3
    alg0 = fits_b0 * (4 * N + 5 * 32 + 40) + sh["n_windows"] * N * 19
    dt = elapsed / args.steps
    # phase A is repeated on every device when the bootstraps are sharded: count those fits once
    reps = (n - 1) * W * S if W < n else 0
    ev_a = int(out["info_a"]["evals"].sum()) if reps else 0
    fits = cnt["fits"] - reps
    evals = cnt["evals"] - cnt["evals_skipped"] - (n - 1) * ev_a
    result = {
        "metric": "ABneutral model fits/sec (pedigree x bootstraps x windows)", "value": fits / dt, "unit": "fits/s",
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": label, "rows": N, "windows": W, "starts": S, "bootstraps": B, "devices": devices,
                   "parallelism": "abn_multi_* (one process): a plan per device from one host thread, RCCL "
                                  "all-gather / broadcast of the bootstrap tables inside the library"
                                  + (" (forced on one device: ABN_MULTI_FORCE_RCCL)" if os.environ.get("ABN_MULTI_FORCE_RCCL") else "")},
        "candidate_evals_per_s": evals / dt, "fits_per_step": fits, "evals_per_step": evals,
        "kernel_ms_per_device": kms, "all_rows_finite": sane,
        "roofline": {"bound": "hbm", "achieved": alg0 / kern_s / 1e9 if kern_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": (alg0 / kern_s / 1e9 if kern_s > 0 else 0.0) / HBM_PEAK_GBS,
                     "traffic": ((measured_traffic(args.workload)[0] or {}).get("fit_boot") or {}).get("hbm_bytes_per_launch")
                     if n == 1 else None,
                     "kernel": "phase-B fit kernel of device 0's shard (slowest device's HIP-event time)",
                     "kernel_ms": kern_s * 1e3, "algorithmic_bytes_per_launch": alg0,
                     "note": "LDS-resident fits: nominal HBM roofline, see the process-per-GPU line for valu_issue"},
    }
    m.close()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 200 for the millisecond workloads — long enough for a 5 s GPU-busy "
                         "sampler to see the run —, 20 for g351 / mp, 5 for c5s, 1 for the 6 s steps of c5)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c4", "c4s", "g351", "c5s", "c5p", "c5", "mp", "pw",
                                                         "ref1000_c3", "ref1000_g351"])
    ap.add_argument("--single-process", action="store_true",
                    help="time abn_multi_* — ONE process driving the devices of --devices, RCCL gather inside the library "
                         "— instead of one process per GPU; n_gpus = number of devices")
    ap.add_argument("--devices", default=None, help="device ordinals for --single-process, e.g. 0,1,2,3 (default 0..gpus-1)")
    ap.add_argument("--no-c5-full", action="store_true", help="skip the full-size C5 shard (76 GB, ~20 s) of the default line")
    ap.add_argument("--lanes", type=int, default=0, help="lanes of a wavefront per chain (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--execute-stuck-fits", action="store_true",
                    help="abn_options.no_fixed_point_skip = 1: execute the repeated iterations of fits that have reached "
                         "argmin's fixed point, as the reference does (same outputs; default: finish them at once)")
    ap.add_argument("--tree-order", action="store_true",
                    help="abn_options.strict_order = -1: the pedigree's reduction tree even for pedigrees of up to 16 rows "
                         "(which the default sums serially, in the reference's order): A/B of what that default costs on C2")
    ap.add_argument("--strict-order", action="store_true",
                    help="abn_options.strict_order = 1: every cost sums its residuals serially in row order, as the "
                         "reference does (src/structs.rs:206-213); fits are then bit-equal to the oracle's lanes = 1. "
                         "The default line reports the price of this mode as `strict_order`")
    ap.add_argument("--no-stream-probe", action="store_true",
                    help="skip the short C5-shard run that measures the stream-mode kernel against the HBM roof")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 ranks on one GPU)")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the process group and run the gather even with one rank (exercises the "
                         "torch.distributed / RCCL path on a one-GPU box)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the short extra workloads (C2, C4 shard, G351) and the PCIe-inclusive timing")
    ap.add_argument("--stream-sweep", action="store_true",
                    help="also run the stream-mode kernel with a co-resident working set far below / far above the "
                         "256 MiB Infinity Cache (slow to set up: a 123k-row pedigree)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"c5": 1, "c5s": 5, "c5p": 5, "g351": 20, "mp": 20, "pw": 20, "ref1000_c3": 20, "ref1000_g351": 20}.get(args.workload, 200)
    if args.warmup is None:
        args.warmup = 1 if args.workload == "c5" else 3
    if args.single_process:
        return single_process_main(args)

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks ourselves, BEFORE anything in this process touches the GPU
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for a "
                         "different GPU count than asked for")

    # stdout carries exactly ONE line, the JSON: libraries that chat on file descriptor 1 (RCCL prints a version banner,
    # gloo its connection count) go to stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the ABneutral path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    local_rank = local_rank % ndev          # gloo rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local_rank)
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    import alphabeta_rs_amd as A

    A.load_library(build_if_missing=True)
    seed = 20260101
    if args.workload == "pw":  # the pedigree-construction scan, its own JSON line
        ctx = A.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
        r = pairwise_bench(A, ctx, reps=max(3, args.steps))
        big = r["shapes"][-1]
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps({"metric": "pairwise divergence (DMatrix::from) code bytes/s", "value": big["achieved_GBps"],
                          "unit": "GB/s", "n_gpus": 1, "steps": max(3, args.steps), "warmup": 1,
                          "ms_per_step": big["kernel_ms_avg"], "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                          "config": {"workload": f"pairwise divergence, {big['samples']} samples x {big['sites']} sites, "
                                                 "codes resident in HBM"},
                          "roofline": {"bound": "hbm", "achieved": big["achieved_GBps"], "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": big["frac"], "traffic": big.get("traffic")},
                          "pairwise": r}), flush=True)
        os.dup2(2, 1)
        ctx.close()
        return
    wl = make_workload(args.workload, rank, world)
    N, Wr, S, B = wl["gens"].shape[0], wl["wr"], wl["S"], wl["B"]
    stream = torch.cuda.current_stream().cuda_stream
    ctx = A.Context(local_rank, stream=stream)
    opts = A.default_options(seed=seed, lanes_per_chain=args.lanes, strict_order=1 if args.strict_order else (-1 if args.tree_order else 0),
                             no_fixed_point_skip=1 if args.execute_stuck_fits else 0)
    by_boot = wl.get("shard") == "bootstraps"
    plan = A.Plan(ctx, wl["gens"], Wr, S, B, window_offset=0 if by_boot else wl.get("w0", rank * Wr),
                  boot_offset=rank * B if by_boot else 0, options=opts)
    # the bootstrap table lives in a torch tensor so that RCCL can gather it without a copy
    raw_local = torch.empty((Wr, B, 7), dtype=torch.float64, device="cuda")
    plan.bind_raw(raw_local.data_ptr())
    strong = wl.get("scaling") == "strong"
    if strong and collective and 200 % world:
        raise SystemExit("c4s: the in-place all-gather needs equal blocks: 200 windows must divide by the GPU count")
    raw_all = torch.empty((world * Wr, B, 7), dtype=torch.float64, device="cuda") if collective else raw_local
    plan.set_windows(wl["D"], wl["p0"])          # H2D + index-buffer generation: outside the timed region

    def step():
        plan.run()                               # phase A -> select -> phase B on torch's current stream
        if collective:                           # (same stream: the collective is ordered after the kernels)
            dist.all_gather_into_tensor(raw_all.view(-1), raw_local.view(-1))

    def fence():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kms = {"fit_starts": 0.0, "select": 0.0, "fit_boot": 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # per-kernel HIP-event durations (recorded on the launch stream): re-run K steps, reading the events
    # after each one (reading forces a sync, so this loop is not the timed region)
    for _ in range(args.steps):
        plan.run()
        ms = plan.kernel_ms()
        for k in kms:
            kms[k] += ms[k]
    for k in kms:
        kms[k] /= max(1, args.steps)
    cnt = plan.counters()
    out = plan.download()
    if collective:
        # the gathered table must hold every rank's shard: compare per-rank checksums
        step()
        fence()
        # checksums over the BIT PATTERNS (int64 sums wrap: exact and independent of the order of summation; a sum of
        # the doubles differs in the last bits between two reduction kernels once the shards are large)
        mine = raw_local.view(torch.int64).sum().reshape(1)
        sums = torch.empty(world, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(sums, mine)
        got = raw_all.view(world, -1).view(torch.int64).sum(dim=1)
        if not torch.equal(got, sums) or not torch.equal(raw_all[rank * Wr:(rank + 1) * Wr], raw_local):
            raise SystemExit(f"rank {rank}: gathered bootstrap table does not match the shards")
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # evaluations EXECUTED: fits that reach argmin's fixed point are finished without repeating it
        # (options.no_fixed_point_skip); those evaluations are in info.evals but were not computed
        c = torch.tensor([cnt["fits"], cnt["evals"] - cnt["evals_skipped"], cnt["iters"], cnt["evals_skipped"]],
                         dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot_fits, tot_evals, tot_iters, tot_skipped = (int(v) for v in c.tolist())
        if by_boot:  # the replicated phase-A fits are counted once
            ia = out["info_a"]
            tot_fits -= (world - 1) * ia.size
            tot_evals -= (world - 1) * (int(ia["evals"].sum()) - cnt["evals_skipped_starts"])
            tot_iters -= (world - 1) * int(ia["iters"].sum())
            tot_skipped -= (world - 1) * cnt["evals_skipped_starts"]
    else:
        tot_fits, tot_evals, tot_iters = cnt["fits"], cnt["evals"] - cnt["evals_skipped"], cnt["iters"]
        tot_skipped = cnt["evals_skipped"]

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        fits_per_s = tot_fits / (elapsed / args.steps)
        evals_per_s = tot_evals / (elapsed / args.steps)
        lanes = int(out["info_b"]["lanes"][0, 0])
        K = len({tuple(r) for r in wl["gens"].astype(int).tolist()})
        T = int(wl["gens"].max())
        # ---- roofline of the dominant kernel (phase-B fit kernel), per launch on this rank
        fits_b, evals_b = Wr * B, int(out["info_b"]["evals"].sum()) - cnt["evals_skipped_boot"]
        tree_code = lanes                 # abn_fit_info.lanes: the residual reduction tree of the pedigree
        stream = ((lanes >> 8) & 0xff) != 0   # row-block code set: the pedigree is streamed every evaluation
        lanes &= 0xff
        if stream:
            # SURVEY.md §8(d) per-evaluation figure (4N + 40: the u32 index stream).  The default stream variant
            # moves 8N + 40 per evaluation instead (materialised observations, no per-evaluation gather); its
            # real stream rate is reported beside it as `streamed_*`.
            alg_bytes = evals_b * (4 * N + 40) + fits_b * 112 + Wr * N * 16 + N * 2 + K * 4 + Wr * 56
            streamed_bytes = evals_b * (8 * N + 40) + fits_b * (12 * N + 112)  # tid (2 B/row) is L2-resident
        else:
            # 4N (index row) + 5 x 32 (start simplex) + 40 (candidate in, cost out) per LDS-resident fit,
            # shared bytes N (3 + 8 + 8) per window counted once per launch
            alg_bytes = fits_b * (4 * N + 5 * 32 + 40) + Wr * N * 19
        kern_s = kms["fit_boot"] * 1e-3
        achieved = alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
        flops_eval = 45 * T + 53 * K + 4 * N + 60
        valu_tflops = evals_b * flops_eval / kern_s / 1e12 if kern_s > 0 else 0.0
        phases, traffic_source = measured_traffic(args.workload)
        ph_b = (phases or {}).get("fit_boot")
        traffic = ph_b.get("hbm_bytes_per_launch") if ph_b else None
        if ph_b:
            traffic_source = dict(traffic_source, kernel=ph_b.get("kernel"))
        roofline = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "kernel": ("abn_fit_kernel<64, stream>" if stream else "abn_fit_kernel / abn_fit_refill_kernel (packed)") +
                      " phase B", "kernel_ms": kms["fit_boot"],
            "algorithmic_bytes_per_launch": alg_bytes,
            "mode": "stream" if stream else "resident",
            **({"streamed_bytes_per_launch": streamed_bytes,
                "beyond_l2_GBps": streamed_bytes / kern_s / 1e9,
                "beyond_l2_frac_of_hbm_peak": streamed_bytes / kern_s / 1e9 / HBM_PEAK_GBS} if stream else {}),
            "note": ("stream mode: `achieved` prices an evaluation at SURVEY's 4N+40 B (index stream); the kernel "
                     "streams the materialised observations, 8N+40 B per evaluation past L2 (HBM + Infinity Cache): "
                     "beyond_l2_*" if stream
                     else "LDS-resident fits: the index row is read once per fit, so the HBM roofline is nominal; "
                          "the kernel is FP64-VALU/latency bound (see valu_fp64)"),
            "valu_fp64": {"achieved_tflops": valu_tflops, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                          "frac": valu_tflops / FP64_VALU_PEAK_TFLOPS, "flops_per_eval": flops_eval},
        }
        # how busy the vector pipe is (the roof the resident kernels actually run under), per phase, from the stamped profile
        vi = valu_issue(ph_b, kern_s)
        if vi:
            roofline["valu_issue"] = vi
        vi_a = valu_issue((phases or {}).get("fit_starts"), kms["fit_starts"] * 1e-3)
        if vi_a:
            roofline["phase_a"] = {"kernel_ms": kms["fit_starts"], "valu_issue": vi_a,
                                   "traffic": (phases or {}).get("fit_starts", {}).get("hbm_bytes_per_launch")}
        result = {
            "metric": "ABneutral model fits/sec (pedigree x bootstraps x windows)",
            "value": fits_per_s, "unit": "fits/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["label"], "rows": N, "windows_per_gpu": Wr, "starts": S, "bootstraps": B,
                       "plan_device_bytes": plan.device_bytes(), "kernels": plan.last_kernels(),
                       "reduction_tree": hex(tree_code), "parallelism": f"{'bootstraps' if by_boot else 'windows'} sharded over {world} GPU(s), "
                                                                 "one RCCL all-gather of the bootstrap tables"},
            "candidate_evals_per_s": evals_per_s,
            "fits_per_step": tot_fits, "evals_per_step": tot_evals, "nm_iters_per_step": tot_iters,
            "evals_not_executed_per_step": tot_skipped,
            "fixed_point_skip": not args.execute_stuck_fits,
            "strict_order_run": bool(args.strict_order),
            "kernel_ms": kms,
            "roofline": roofline,
        }
        if world == 1 and not args.no_stream_probe and args.workload != "c5s":
            result["roofline_stream"] = stream_probe(A, ctx, seed)
            if args.stream_sweep:
                result["roofline_stream_sweep"] = stream_sweep(A, ctx, seed)
        if world == 1 and not args.no_extras:
            if Wr == 1:
                result["pcie_inclusive"] = pcie_inclusive(A, ctx, wl, opts)
            result["extra_workloads"] = {n: quick_workload(A, ctx, n, seed)
                                         for n in ("c2", "c4", "c4s", "mp", "g351", "ref1000_c3", "ref1000_g351")
                                         if n != args.workload}
            if not args.no_c5_full and args.workload != "c5":
                result["extra_workloads"]["c5_full_shard"] = c5_full_shard(A, ctx, seed)
            # what the reference's summation order costs on this workload (not on the streamed C5 shapes: N = 20 100 serial
            # additions per evaluation take minutes, and with --workload c5 a second 76 GB plan beside the open one)
            if not args.strict_order and args.workload not in ("c5", "c5s"):
                so = quick_workload(A, ctx, args.workload, seed, strict_order=1)
                so["price"] = fits_per_s / so["fits_per_s"]
                so["what"] = ("abn_options.strict_order = 1 (serial row-order residual sums, bit-equal to the oracle's "
                              "lanes = 1); price = default fits/s / strict fits/s")
                result["strict_order"] = so
            result["pairwise"] = pairwise_bench(A, ctx)
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is a one-GPU (rank 0) measurement
            bs = int(out["best_start"][0])
            if bs >= 0:
                result["cpu_baseline"] = cpu_baseline(wl, out["models"][0], out["pred"][0], out["resid"][0], lanes, seed,
                                                      float(out["info_b"]["evals"][0].mean()))
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(result), flush=True)
        os.dup2(2, 1)
    plan.close()
    ctx.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
