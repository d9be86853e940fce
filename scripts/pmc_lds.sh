#!/bin/bash
# LDS bank-conflict counters of the C3 fit kernels (development aid)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcl && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcl -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stream-probe --no-extras > /dev/null 2> gpurun_out/pmcl.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmcl/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "fit" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:44], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()}, "(millions per launch)")
PY
