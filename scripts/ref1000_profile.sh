#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of the reference's default shape (1000 starts + 1000 bootstraps) on the C3 and
# the 351-row pedigree; prints the abn_* kernel lines (the copies kept under profiles/ are r03_ref1000_kernel_trace.txt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in ref1000_c3 ref1000_g351; do
  out=gpurun_out/prof_$wl
  rm -rf "$out" && mkdir -p "$out"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-stream-probe --no-extras > "$out/bench.json" 2> "$out/kt.err"
  echo "== $wl"
  python3 - "$out" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
tr = max(glob.glob(out + "/kt/*/*_kernel_trace.csv"), key=os.path.getmtime)
d = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    if "abn" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:70], r["Grid_Size_X"], r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
j = json.load(open(out + "/bench.json"))
print("bench:", round(j["value"]), "fits/s", {k: round(v, 3) for k, v in j["kernel_ms"].items()}, j["config"]["kernels"])
for k, v in d.items():
    print(k, "calls", len(v), "avg_us %.1f min_us %.1f max_us %.1f" % (sum(v) / len(v), min(v), max(v)))
PY
done
