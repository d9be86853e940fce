#!/bin/bash
# The measurements of a round on the GPU box (run through gpurun from the repository root):
#   bash scripts/round_measure.sh r02
# Writes everything under gpurun_out/<tag>/; the summaries worth keeping are copied to profiles/ afterwards.
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
# profiles first: the bench line takes roofline.traffic only from a PMC profile of these very kernel sources
echo "== rocprofv3: c3"; bash scripts/profile_round.sh $tag c3 > $out/profile_c3.log 2>&1; tail -2 $out/profile_c3.log
echo "== rocprofv3: c5s"; bash scripts/profile_round.sh $tag c5s > $out/profile_c5s.log 2>&1; tail -2 $out/profile_c5s.log
cp gpurun_out/profiles_c3/${tag}_* gpurun_out/profiles_c5s/${tag}_* profiles/ 2>/dev/null
echo "== default bench line (C3, cpu baseline, stream probe, extras)"
timeout -k 10 600 python bench.py > $out/bench_c3.json 2> $out/bench_c3.err; echo "rc=$?"
for wl in c2 c4 c4s g351 mp c5s pw ref1000_c3 ref1000_g351; do
  echo "== bench $wl"; timeout -k 10 600 python bench.py --workload $wl --no-extras --no-stream-probe > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "rc=$?"
done
echo "== bench c5 (the full per-GPU shard: 63 windows x 5010 fits x 20100 rows)"; timeout -k 10 600 python bench.py --workload c5 --no-extras --no-stream-probe --no-cpu-baseline > $out/bench_c5.json 2> $out/bench_c5.err; echo "rc=$?"
echo "== bench c3 --strict-order"; timeout -k 10 300 python bench.py --strict-order --no-extras --no-stream-probe --no-cpu-baseline > $out/bench_c3_strict.json 2> $out/bench_c3_strict.err; echo "rc=$?"
echo "== single-process form (abn_multi_*, gather forced through RCCL on the one device)"
ABN_MULTI_FORCE_RCCL=1 timeout -k 10 300 python bench.py --single-process --devices 0 --workload c3 > $out/bench_c3_single_process.json 2> $out/bench_sp.err; echo "rc=$?"
ABN_MULTI_FORCE_RCCL=2 timeout -k 10 300 python bench.py --single-process --devices 0 --workload c4 > $out/bench_c4_single_process.json 2>> $out/bench_sp.err; echo "rc=$?"
echo "== 2 ranks (gloo) on the one GPU: the launcher path"
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --steps 10 --no-extras --no-stream-probe --no-cpu-baseline > $out/bench_c3_2rank_gloo.json 2> $out/bench_2rank.err; echo "rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --steps 2 > $out/bench_gpus2_nccl.out 2> $out/bench_gpus2_nccl.err; echo "nccl --gpus 2 on a one-GPU box: rc=$? (must be non-zero, no JSON)"
echo "== stream-mode working-set sweep"
timeout -k 10 900 python bench.py --stream-sweep --no-extras --no-cpu-baseline --steps 5 > $out/bench_stream_sweep.json 2> $out/bench_stream_sweep.err; echo "rc=$?"
echo "== rocprofv3: pairwise"; bash scripts/pw_profile.sh $out/pw_prof > $out/pw_profile.txt 2>&1; cat $out/pw_profile.txt
echo done
