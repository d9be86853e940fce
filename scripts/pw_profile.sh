# GPU box: rocprofv3 kernel trace of the pairwise workload (bench.py --workload pw); prints the abn_* kernel lines and
# writes <dir>/<tag>_pmc_pw.json (bytes fetched per scan launch and shape, stamped with the kernel-source hash)
#   usage: bash scripts/pw_profile.sh [dir] [tag]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/pw_prof}
rm -rf "$out" && mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --workload pw --steps 5 > "$out/bench.json" 2> "$out/kt.err"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tr = glob.glob(out + "/kt/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    if "abn" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:60], r["Grid_Size_X"], r.get("LDS_Block_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():      # the bench runs several site counts through the same kernel and grid: one line per cluster
    v = sorted(v)
    clusters = [[v[0]]]
    for x in v[1:]:
        if x < 1.5 * clusters[-1][0]:
            clusters[-1].append(x)
        else:
            clusters.append([x])
    for cl in clusters:
        print(k, "calls", len(cl), "avg_us %.1f min_us %.1f" % (sum(cl) / len(cl), min(cl)))
PY
# one separate --pmc pass: bytes fetched past L2 per launch (FETCH_SIZE is in KiB; x2 on gfx950 for wide coalesced reads,
# MI355X_MICROARCH.md) against the n x L code bytes — the kernel reads the codes once
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --workload pw --steps 5 > /dev/null 2> "$out/pmc.err"
python3 - "$out" "$GRAFT_REPO_ROOT" "${2:-r04}" <<'PY'
import csv, glob, sys, collections, json, hashlib, datetime
from pathlib import Path
out, root, tag = sys.argv[1], Path(sys.argv[2]), sys.argv[3]
f = glob.glob(out + "/pmc_fetch/*/*_counter_collection.csv")[0]
shapes = {(s["samples"], s["sites"]): s["code_bytes"] for s in json.load(open(out + "/bench.json"))["pairwise"]["shapes"]}
got = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "pairwise_mx" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        b = 2 * float(r["Counter_Value"]) * 1024                 # KiB; x2: the gfx950 correction for wide coalesced reads
        key = min(shapes, key=lambda k: abs(shapes[k] - b))     # the launches of a shape differ only in the site count: nearest
        got[key].append(b)
res = {}
for k, v in got.items():
    a = sum(v) / len(v)
    res[f"{k[0]}x{k[1]}"] = a
    print(f"abn_pairwise_mx_kernel {k[0]} samples x {k[1]} sites: FETCH_SIZE x 2 = {a / 1e6:.1f} MB per launch against {shapes[k] / 1e6:.1f} MB of codes ({len(v)} launches)")
h = hashlib.sha1()
for p in sorted((root / "alphabeta_rs_amd" / "csrc").glob("*")):
    if p.name != "abn_multi.hip" and p.suffix in {".hip", ".hpp", ".h"}:
        h.update(p.name.encode()); h.update(p.read_bytes())
Path(out, f"{tag}_pmc_pw.json").write_text(json.dumps({"workload": "pw", "phases": res, "source_sha1": h.hexdigest(),
    "units": "HBM bytes per scan launch = 2 x FETCH_SIZE x 1024 (gfx950 correction), by shape <samples>x<sites>",
    "collected": datetime.date.today().isoformat()}, indent=1))
PY
