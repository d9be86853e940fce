#!/bin/bash
# GPU box: helper groups (the round-3 experiment: scripts/attic/helper_groups_{device,api}_r03.patch, reverse diffs — apply
# with `patch -R` to a scratch copy of csrc/) compiled out (build/libabn_nohelp.so: the shipped code with
# -DABN_MEASUREMENT_KNOBS), compiled in and switched off / on (build/libabn_knobs.so: the patched copy with
# -DABN_MEASUREMENT_KNOBS -DABN_HELPER_GROUPS, ABN_HELPERS=0|1).  Cross-compile both before the call:
#   for v in "nohelp" "knobs -DABN_HELPER_GROUPS"; do set -- $v; hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off \
#     -fno-fast-math -fPIC -shared -ldl -DABN_MEASUREMENT_KNOBS $2 -o build/libabn_$1.so alphabeta_rs_amd/csrc/abn_{api,multi}.hip; done
#   bash scripts/helpers_ab.sh [workloads...]
for wl in ${@:-c3 c4}; do
  for cfg in "nohelp 0" "knobs 0" "knobs 1" "nohelp 0" "knobs 1"; do
    set -- $cfg
    ABNEUTRAL_HIP_LIB=$PWD/build/libabn_$1.so ABN_HELPERS=$2 python bench.py --workload $wl --steps 50 --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
      python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$wl lib=$1 helpers=$2', round(j['value']), 'fits/s', {k: round(v,3) for k,v in j['kernel_ms'].items()})"
  done
done
