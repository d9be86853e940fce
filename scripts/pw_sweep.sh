# measurement aid (GPU box): pairwise kernel, persistent counters vs per-tile sums, slices per pair block
mkdir -p gpurun_out/pw_sweep
for cfg in default 1,1 1,2 0,1 0,2 0,4 0,8; do
  if [ "$cfg" = default ]; then unset ABN_PAIR_SLICES; else export ABN_PAIR_SLICES=$cfg; fi
  python bench.py --workload pw --steps 5 > gpurun_out/pw_sweep/pw_$cfg.json 2>/dev/null
  python -c "
import json; j=json.load(open('gpurun_out/pw_sweep/pw_$cfg.json'))
print('$cfg', [(s['samples'], round(s['kernel_ms_min']*1e3,1), round(s['kernel_ms_avg']*1e3,1), round(s['achieved_GBps'])) for s in j['pairwise']['shapes']])"
done
