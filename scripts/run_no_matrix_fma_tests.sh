set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -DABN_NO_MATRIX_FMA -o gpurun_out/libnomx.so alphabeta_rs_amd/csrc/abn_api.hip
ABNEUTRAL_HIP_LIB=$PWD/gpurun_out/libnomx.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not cli and not metaprofile and not reference_unit" 2>&1 | tail -3
