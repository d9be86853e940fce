#!/bin/bash
# GPU box: time-slicing quantum of the persistent kernel with the tail hand-over in place, per workload (knobs build).
run() {
  local label=$1; shift
  env ABNEUTRAL_HIP_LIB=$PWD/build/libabn_knobs.so "$@" python bench.py --workload $WL --steps ${STEPS:-10} --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$WL $label', round(j['value']), 'fits/s', {k: round(v,3) for k,v in j['kernel_ms'].items()})"
}
for WL in mp c4 c4s; do
  for q in 128 256 384 512 768 1024 100000; do run "quantum=$q" ABN_QUANTUM_ENV=$q; done
done
