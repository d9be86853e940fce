# GPU box: SQ instruction-mix / utilisation counters of one bench workload (one --pmc pass, kernel names + averages)
#   usage: bash scripts/pmc_sq.sh <workload> [outdir]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
wl=${1:-c4}; out=${2:-gpurun_out/pmc_sq_$wl}
rm -rf "$out" && mkdir -p "$out"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc" -- python3 bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline --no-stream-probe --no-extras > "$out/bench.json" 2> "$out/err.txt"
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(out + "/pmc/*/*_counter_collection.csv")[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "abn_fit" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
j = json.load(open(out + "/bench.json"))
print("kernel_ms", j["kernel_ms"])
for k, v in d.items():
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, {c: round(x / 1e6, 1) for c, x in a.items()}, "M")
PY
