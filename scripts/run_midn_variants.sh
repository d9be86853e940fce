set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "2 3" "4 2" "6 2"; do set -- $cfg
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -DABN_STREAM_BLOCKS=$1 -DABN_STREAM_WAVES=$2 -o gpurun_out/libmid_$1_$2.so alphabeta_rs_amd/csrc/abn_api.hip
  echo "blocks=$1 waves=$2"
  ABNEUTRAL_HIP_LIB=$PWD/gpurun_out/libmid_$1_$2.so python scripts/midn_bench.py
done
