# GPU box: rocprofv3 kernel trace of the pairwise workload (bench.py --workload pw); prints the abn_* kernel lines
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
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc_fetch/*/*_counter_collection.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "pairwise_mx" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
for k, v in d.items():
    a = sum(v) / len(v)
    print(k, "FETCH_SIZE avg %.0f KiB -> %.1f MB (x2 gfx950 correction: %.1f MB)" % (a, a * 1024 / 1e6, 2 * a * 1024 / 1e6))
PY
