#!/bin/bash
# GPU box: the reference's default shape (1000 starts + 1000 bootstraps) with each phase-A / phase-B kernel forced
# (build/libabn_nohelp.so = the shipped sources with -DABN_MEASUREMENT_KNOBS, cross-compiled before the call)
for wl in ref1000_c3 ref1000_g351; do
  for ka in default spec wide packed; do
    for kb in default; do
      env ABNEUTRAL_HIP_LIB=$PWD/build/libabn_nohelp.so $( [ $ka != default ] && echo ABN_PHASE_A_KERNEL=$ka ) python bench.py --workload $wl --steps 10 --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
        python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$wl A=$ka', round(j['value']), 'fits/s', {k: round(v,3) for k,v in j['kernel_ms'].items()}, j['config']['kernels'])"
    done
  done
  for kb in spec wide packed; do
    env ABNEUTRAL_HIP_LIB=$PWD/build/libabn_nohelp.so ABN_PHASE_B_KERNEL=$kb python bench.py --workload $wl --steps 10 --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
      python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$wl B=$kb', round(j['value']), 'fits/s', {k: round(v,3) for k,v in j['kernel_ms'].items()}, j['config']['kernels'])"
  done
done
