#!/usr/bin/env python3
"""Development aid (CPU): the two measurement tables of DESIGN.md §4 from the bench lines of a round
(gpurun_out/<tag>/bench_*.json, written by scripts/round_measure.sh) and the stamped counter profiles under profiles/.
usage: design_tables.py r04   -> prints the workload table and the pairwise table (markdown)"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
def J(n): return json.load(open(ROOT / f"gpurun_out/{tag}/bench_{n}.json"))
c3 = J("c3"); rows = []
def km(j): k = j["kernel_ms"]; return f"{j['ms_per_step']:.2f} ({k['fit_starts']:.2f} / {k['select']:.2f} / {k['fit_boot']:.2f})"
def kn(j): k = j["config"]["kernels"]; return f"{k['starts'][0]} {k['starts'][1]} / {k['boot'][0]} {k['boot'][1]}"
vi = c3["roofline"]["valu_issue"]; w = vi["of_wavefront_cycles"]
rows.append(f"| **C3** (default line, 200 steps) | **{c3['value']/1e6:.2f} M** (round 3: 2.67 M driver-run, 2.70–2.81 M over five boxes) | {km(c3)} | {kn(c3)} | {c3['candidate_evals_per_s']/1e9:.2f}·10⁹ candidate evaluations/s; phase B: {vi['insts_per_launch']/1e6:.0f} M vector instructions, vector issue {100*vi['frac']:.0f} % over the launch, of the wavefronts' resident cycles {100*w['issuing']:.0f} % issue / {100*w['waiting_on_counter']:.0f} % wait on a counter / {100*w['waiting_to_issue']:.0f} % wait to issue; counter traffic {c3['roofline']['traffic']/1e6:.1f} MB per launch against {c3['roofline']['algorithmic_bytes_per_launch']/1e6:.1f} MB algorithmic (the parked states of time slicing) |")
s = J("c3_strict"); rows.append(f"| C3 `--strict-order` | {s['value']/1e6:.2f} M | {km(s)} | {kn(s)} | bit-equal to the reference-order oracle; price {c3['value']/s['value']:.2f}× |")
for n, lab, note in (("c2", "C2 (bundled pedigree; serial order by default since round 4)", ""), ("c4", "C4 shard", ""),
                     ("c4s", "C4, 200 windows on one GPU (`c4s`)", "the strong-scaling job's N = 1 point"), ("mp", "metaprofile shape", ""),
                     ("g351", "G (351-row golden pedigree)", ""), ("ref1000_c3", "**reference default `-i 1000`, C3 pedigree**", ""),
                     ("ref1000_g351", "reference default, 351-row pedigree", ""), ("c5s", "C5 shard probe (`c5s`, 1 window × 4096)", "")):
    j = J(n); extra = note
    if n == "c2":
        t = J("c2_tree"); extra = f"`--tree-order` (round 3's default): {t['value']/1e3:.0f} k fits/s, {km(t)} — the serial default costs {100*(1-j['value']/t['value']):.1f} %"
    if n in ("c4s", "mp") and "pmc" in c3["extra_workloads"][n]:
        e = c3["extra_workloads"][n]["pmc"]; b = e["fit_boot"]["valu_issue"]; a = e["fit_starts"]["valu_issue"]
        extra = (extra + "; " if extra else "") + f"vector issue {100*a['frac']:.0f} % (phase A) / {100*b['frac']:.0f} % (phase B) of the launch's SIMD-cycles; counter traffic {e['fit_boot']['traffic']/1e6:.0f} MB per phase-B launch (parked states)"
    if j.get("evals_not_executed_per_step"):
        extra = (extra + "; " if extra else "") + f"{j['evals_not_executed_per_step']/1e6:.2f} M evaluations of stuck fits not executed"
    v = j["value"]; vs = f"{v/1e6:.2f} M" if v >= 1e6 else f"{v/1e3:.1f} k"
    rows.append(f"| {lab} | {vs} | {km(j)} | {kn(j)} | {extra} |")
c5 = c3["extra_workloads"]["c5_full_shard"]; r5 = c5["roofline"]
rows.append(f"| **C5 per-GPU shard at full size** (`c5`) | **{c5['fits_per_s']/1e3:.1f} k** | {c5['ms_per_step']:.0f} ({c5['kernel_ms']['fit_starts']:.1f} / {c5['kernel_ms']['select']:.1f} / {c5['kernel_ms']['fit_boot']:.0f}) | stream | 63 windows × (10 + 5000) × 20 100 rows; 76.0 GB resident; phase B: **{r5['traffic']/1e12:.2f} TB per launch by FETCH_SIZE × 2 + WRITE_SIZE** ({r5['streamed_bytes_per_launch']/1e12:.2f} TB by construction) = **{r5['traffic']/c5['kernel_ms']['fit_boot']/1e9:.2f} TB/s past L2**, {r5['achieved']/1e3:.2f} TB/s = **{r5['frac']:.2f} of the HBM roof at SURVEY's 4N+40 B** per evaluation |")
rs = c3["roofline_stream"]
rows.append(f"| stream probe of every bench line (`c5p`, 8192 bootstraps) | — | phase B {rs['kernel_ms']:.1f} | stream | counter traffic {rs['traffic']/1e9:.0f} GB per launch = {rs['traffic']/rs['kernel_ms']/1e9:.2f} TB/s past L2 ({rs['streamed_bytes_per_launch']/1e9:.0f} GB by construction); {rs['frac']:.2f} at SURVEY's bytes |")
rows.append(f"| C3 including PCIe (host-buffer entry points) | {c3['pcie_inclusive']['fits_per_s']/1e6:.2f} M | {c3['pcie_inclusive']['ms_per_step']:.2f} | — | `abn_ab_neutral_run` + `abn_boot_model_run` end to end |")
cb = c3["cpu_baseline"]
rows.append(f"| CPU oracle, reference-shaped, {cb['cores']} host cores | {cb['value']/1e3:.2f} k ({cb['power_table_variant_fits_per_s']/1e3:.1f} k with the power table) | — | — | `cpu_baseline` of the line: a reported baseline, not a target |")
print("| Workload | fits/s | ms/step (A / select / B) | kernels A / B (lanes) | note |\n|---|---|---|---|---|\n" + "\n".join(rows))
print()
r3 = {(15, 4000000): (28.6, 0.262), (15, 32000000): (135.0, 0.446), (50, 32000000): (722.0, 0.277), (50, 2000000): (52.9, 0.236)}
t = "  | samples × sites | code bytes | round 3 (vector ALU) | round 4 | fraction of 8 TB/s | FETCH_SIZE × 2 per scan |\n  |---|---|---|---|---|---|\n"
for s in c3["pairwise"]["shapes"]:
    o = r3[(s["samples"], s["sites"])]
    t += f"  | {s['samples']} × {s['sites']//1000000} M | {s['code_bytes']/1e6:.0f} MB | {o[0]:.1f} µs ({o[1]:.2f}) | **{s['kernel_ms_avg']*1e3:.1f} µs** | **{s['frac']:.2f}** ({s['achieved_GBps']/1e3:.2f} TB/s) | {s['traffic']/1e6:.1f} MB |\n"
print(t)
