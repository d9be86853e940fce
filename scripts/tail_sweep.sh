#!/bin/bash
# GPU box: the persistent kernel's tail hand-over — chains left when the speculative kernel takes over (ABN_TAIL_CAP; above what
# that kernel keeps resident the later workgroups start as earlier ones end), wavefronts and quantum re-swept (knobs build).
run() {
  local label=$1; shift
  env ABNEUTRAL_HIP_LIB=$PWD/build/libabn_knobs.so "$@" python bench.py --workload $WL --steps ${STEPS:-30} --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$WL $label', round(j['value']), 'fits/s', {k: round(v,3) for k,v in j['kernel_ms'].items()})"
}
for WL in c3 c4 mp; do
  for t in 0 512 1024 1536 2048 3072 4096; do run "tail_cap=$t" ABN_TAIL_CAP=$t; done
done
WL=c3
for w in 1536 2048 2560 3072; do run "waves=$w" ABN_PERSIST_WAVES_SMALL_ENV=$w; done
for w in 2048 3072; do for t in 1536 2048; do run "waves=$w tail_cap=$t" ABN_PERSIST_WAVES_SMALL_ENV=$w ABN_TAIL_CAP=$t; done; done
