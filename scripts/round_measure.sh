#!/bin/bash
# The measurements of a round on the GPU box (run through gpurun from the repository root), in stages so that each fits one
# gpurun call:   bash scripts/round_measure.sh r04 <stage>        stages: tests profiles bench bench2 c5 misc
# Writes under gpurun_out/<tag>/ (and gpurun_out/profiles_<wl>/ for the rocprofv3 summaries); the summaries worth keeping are
# copied to profiles/ afterwards.  The profiles come FIRST: the bench line quotes counters only from a PMC profile of these
# very kernel sources (source_sha1), so run `profiles`, copy gpurun_out/profiles_*/<tag>_* into profiles/, then `bench`.
tag=${1:-r04}; stage=${2:-all}
out=gpurun_out/$tag
mkdir -p $out
run_bench() {  # $1 = file stem, rest = bench args
  local stem=$1; shift
  timeout -k 10 ${BENCH_TIMEOUT:-600} python bench.py "$@" > $out/bench_$stem.json 2> $out/bench_$stem.err; echo "bench $stem rc=$?"
}
case $stage in
tests)
  timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log;;
profiles)
  for wl in c3 c4s mp c5p; do
    echo "== rocprofv3: $wl"; bash scripts/profile_round.sh $tag $wl > $out/profile_$wl.log 2>&1; tail -2 $out/profile_$wl.log
  done
  echo "== rocprofv3: pairwise"; bash scripts/pw_profile.sh $out/pw_prof $tag > $out/pw_profile.txt 2>&1; cat $out/pw_profile.txt;;
profile_c5)
  # the full per-GPU C5 shard (76 GB): one FETCH_SIZE and one WRITE_SIZE pass of one step
  PASSES="fetch write" STEPS=1 WARMUP=0 bash scripts/profile_round.sh $tag c5 > $out/profile_c5.log 2>&1; tail -2 $out/profile_c5.log;;
bench)
  echo "== default bench line (C3, cpu baseline, stream probe, extras, C5 full shard)"
  BENCH_TIMEOUT=900 run_bench c3;;
bench2)
  for wl in c2 c4 c4s g351 mp c5s pw ref1000_c3 ref1000_g351; do run_bench $wl --workload $wl --no-extras --no-stream-probe; done
  run_bench c3_strict --strict-order --no-extras --no-stream-probe --no-cpu-baseline
  run_bench c2_tree --workload c2 --tree-order --no-extras --no-stream-probe --no-cpu-baseline;;
c5)
  run_bench c5 --workload c5 --no-extras --no-stream-probe --no-cpu-baseline;;
misc)
  echo "== single-process form (abn_multi_*, gather forced through RCCL on the one device)"
  ABN_MULTI_FORCE_RCCL=1 timeout -k 10 300 python bench.py --single-process --devices 0 --workload c3 > $out/bench_c3_single_process.json 2> $out/bench_sp.err; echo "rc=$?"
  ABN_MULTI_FORCE_RCCL=2 timeout -k 10 300 python bench.py --single-process --devices 0 --workload c4 > $out/bench_c4_single_process.json 2>> $out/bench_sp.err; echo "rc=$?"
  echo "== 2 ranks (gloo) on the one GPU: the launcher path"
  run_bench c3_2rank_gloo --gpus 2 --backend gloo --steps 10 --no-extras --no-stream-probe --no-cpu-baseline
  timeout -k 10 300 python bench.py --gpus 2 --steps 2 > $out/bench_gpus2_nccl.out 2> $out/bench_gpus2_nccl.err; echo "nccl --gpus 2 on a one-GPU box: rc=$? (must be non-zero, no JSON)"
  echo "== stream-mode working-set sweep"
  BENCH_TIMEOUT=900 run_bench stream_sweep --stream-sweep --no-extras --no-cpu-baseline --steps 5
  echo "== fuzz (2 x ${FUZZ_S:-400} s) and soak (${SOAK_S:-300} s) on the final kernels"
  timeout -k 10 $((${FUZZ_S:-400} + 60)) python tests/fuzz/fuzz_parity.py ${FUZZ_S:-400} 404 > $out/fuzz_parity.log 2>&1; tail -1 $out/fuzz_parity.log
  timeout -k 10 $((${FUZZ_S:-400} + 60)) python tests/fuzz/fuzz_plan.py ${FUZZ_S:-400} 405 > $out/fuzz_plan.log 2>&1; tail -1 $out/fuzz_plan.log
  timeout -k 10 $((${SOAK_S:-300} + 120)) python tests/fuzz/soak_timeslice.py ${SOAK_S:-300} > $out/soak.log 2>&1; tail -3 $out/soak.log;;
*) echo "stage?";;
esac
echo done
