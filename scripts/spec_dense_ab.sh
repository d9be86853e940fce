#!/bin/bash
# GPU box: the speculative kernel built for four workgroups per CU (DENSE) against three, on the launches it is chosen for
# (769 .. 1024 chains, up to two rows per lane): C2's 1000 bootstraps, the reference's default -i 1000 on the C3 pedigree.
# -DABN_MEASUREMENT_KNOBS build (build/libabn_knobs.so): ABN_SPEC_DENSE=0 keeps three per CU.
run() {
  local label=$1; shift
  env ABNEUTRAL_HIP_LIB=$PWD/build/libabn_knobs.so "$@" python bench.py --workload $WL --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$WL $label', round(j['value']), 'fits/s', {k: round(v,3) for k,v in j['kernel_ms'].items()})"
}
for WL in c2 ref1000_c3; do
  for rep in 1 2; do run "three per CU" ABN_SPEC_DENSE=0; run "four per CU (dense)" ABN_SPEC_DENSE=1; done
done
