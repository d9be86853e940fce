#!/bin/bash
# Collects the rocprofv3 evidence for one bench workload on the GPU box (run through gpurun):
#   kernel trace + stats of `bench.py`, then three separate --pmc passes (HBM read, HBM write, SQ instruction mix),
#   and summarises them into profiles/ (copied back through gpurun_out/).
# usage: scripts/profile_round.sh <tag> <workload> [bench args...]     e.g.  scripts/profile_round.sh r01 c3
set -e
tag=$1; wl=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$wl
rm -rf "$out" && mkdir -p "$out"
args="--workload $wl --steps 20 --warmup 2 --no-cpu-baseline --no-stream-probe --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py $args > "$out/bench_under_rocprof.json" 2> "$out/kt.err"
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py $args > /dev/null 2> "$out/pmc_fetch.err"
echo "pmc FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py $args > /dev/null 2> "$out/pmc_write.err"
echo "pmc WRITE_SIZE done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_sq" -- python3 bench.py $args > /dev/null 2> "$out/pmc_sq.err"
echo "pmc SQ done"
mkdir -p gpurun_out/profiles_$wl
cp profiles/summarize_rocprof.py gpurun_out/profiles_$wl/
python3 gpurun_out/profiles_$wl/summarize_rocprof.py "$out" "${tag}_${wl}" "$wl" "$GRAFT_REPO_ROOT" > /dev/null
cp "$out/bench_under_rocprof.json" gpurun_out/profiles_$wl/${tag}_${wl}_bench_under_rocprof.json
ls gpurun_out/profiles_$wl
