#!/bin/bash
# Collects the rocprofv3 evidence for one bench workload on the GPU box (run through gpurun):
#   kernel trace + stats of `bench.py`, then SEPARATE --pmc passes (HBM read, HBM write, two SQ passes: instruction mix and
#   busy / wait cycles), summarised per phase into profiles/ (copied back through gpurun_out/).
# usage: scripts/profile_round.sh <tag> <workload> [bench args...]     e.g.  scripts/profile_round.sh r04 c3
#   PASSES="kt fetch write sq sq2" (default: all) restricts the passes, STEPS / WARMUP the launches per pass
set -e
tag=$1; wl=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$wl
rm -rf "$out" && mkdir -p "$out"
passes=${PASSES:-kt fetch write sq sq2}
args="--workload $wl --steps ${STEPS:-20} --warmup ${WARMUP:-2} --no-cpu-baseline --no-stream-probe --no-extras $*"
for p in $passes; do
  case $p in
    kt) rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py $args > "$out/bench_under_rocprof.json" 2> "$out/kt.err";;
    fetch) rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py $args > /dev/null 2> "$out/pmc_fetch.err";;
    write) rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py $args > /dev/null 2> "$out/pmc_write.err";;
    sq) rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_sq" -- python3 bench.py $args > /dev/null 2> "$out/pmc_sq.err";;
    sq2) rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d "$out/pmc_sq2" -- python3 bench.py $args > /dev/null 2> "$out/pmc_sq2.err";;
  esac
  echo "pass $p done"
done
mkdir -p gpurun_out/profiles_$wl
cp profiles/summarize_rocprof.py gpurun_out/profiles_$wl/
python3 gpurun_out/profiles_$wl/summarize_rocprof.py "$out" "${tag}_${wl}" "$wl" "$GRAFT_REPO_ROOT" > /dev/null
[ -f "$out/bench_under_rocprof.json" ] && cp "$out/bench_under_rocprof.json" gpurun_out/profiles_$wl/${tag}_${wl}_bench_under_rocprof.json
ls gpurun_out/profiles_$wl
