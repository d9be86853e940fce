#!/bin/bash
# GPU box: wave priority by chain age in the persistent kernel, on a -DABN_MEASUREMENT_KNOBS build
# (build/libabn_knobs.so, cross-compiled before the call).  ABN_PRIO = mode,t0,t1,t2 (abn_device.hpp: FitArgs::prio_mode).
run() {  # $1 = label, rest = env assignments
  local label=$1; shift
  env ABNEUTRAL_HIP_LIB=$PWD/build/libabn_knobs.so "$@" python bench.py --workload ${WL:-c3} --steps ${STEPS:-100} --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('${WL:-c3} $label', round(j['value']), 'fits/s', {k: round(v,3) for k,v in j['kernel_ms'].items()})"
}
run base X=1
run base X=1
for p in 1,400,600,750 1,500,650,780 1,600,700,800 1,300,500,700 1,700,780,840 2,400,600,750 2,300,500,700; do run "prio=$p" ABN_PRIO=$p; done
for w in 1536 2048 2560 3072; do
  run "waves=$w" ABN_PERSIST_WAVES_SMALL_ENV=$w
  run "waves=$w prio=1,500,650,780" ABN_PERSIST_WAVES_SMALL_ENV=$w ABN_PRIO=1,500,650,780
done
for q in 128 512; do run "quantum=$q prio=1,500,650,780" ABN_QUANTUM_ENV=$q ABN_PRIO=1,500,650,780; done
run base X=1
