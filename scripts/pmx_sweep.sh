#!/bin/bash
# GPU box: the matrix-pipe pairwise scan (abn_pairwise_mx.hpp) over the workgroups per CU, on a -DABN_MEASUREMENT_KNOBS
# build (build/libabn_knobs.so; ABN_PMX_CU_JOBS).  profiles/r04_pmx_sweep.txt holds the runs of round 4, the vector-ALU
# kernel of rounds 2-3 (removed since) beside them.
run() {
  local label=$1; shift
  env ABNEUTRAL_HIP_LIB=$PWD/build/libabn_knobs.so "$@" python bench.py --workload pw --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$label', [(s['samples'], s['sites']//1000000, round(s['kernel_ms_avg']*1e3,1), round(s['frac'],3)) for s in j['pairwise']['shapes']])"
}
run "default" X=1
for j in 1 2 3 4; do run "mx cu_jobs=$j" ABN_PMX_CU_JOBS=$j; done
