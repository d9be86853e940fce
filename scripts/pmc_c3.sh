#!/bin/bash
# instruction counters of the C3 fit kernels (development aid): prints avg per launch by grid size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcq && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmcq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmcq.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmcq/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "fit_kernel" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:40], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()}, "(millions per launch)")
PY
