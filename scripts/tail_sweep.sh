run() {
  local label=$1; shift
  env ABNEUTRAL_HIP_LIB=$PWD/build/libabn_knobs.so "$@" python bench.py --workload $WL --steps 50 --no-extras --no-stream-probe --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$WL $label', round(j['value']), 'fits/s', {k: round(v,3) for k,v in j['kernel_ms'].items()})"
}
WL=c3
for w in 1536 1792 2048 2304 2560 3072; do run "waves=$w" ABN_PERSIST_WAVES_SMALL_ENV=$w; done
for q in 128 192 384 512; do run "quantum=$q" ABN_QUANTUM_ENV=$q; done
for t in 256 512 768 1024; do run "tail_cap=$t" ABN_TAIL_CAP=$t; done
run "default" X=1
