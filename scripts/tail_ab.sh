#!/bin/bash
# GPU box: tail hand-over of time-sliced persistent launches to the speculative kernel (FitArgs::tail_cap) on and off
# (ABN_TAIL_CAP=0) on a -DABN_MEASUREMENT_KNOBS build (build/libabn_knobs.so).
run() {
  local label=$1; shift
  env ABNEUTRAL_HIP_LIB=$PWD/build/libabn_knobs.so "$@" python bench.py --workload $WL --steps ${STEPS:-20} --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$WL $label', round(j['value']), 'fits/s', {k: round(v,3) for k,v in j['kernel_ms'].items()})"
}
for WL in mp c4 c3 c4s; do
  for rep in 1 2; do run "tail off" ABN_TAIL_CAP=0; run "tail on" X=1; done
done
