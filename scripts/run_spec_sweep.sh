#!/bin/bash
# Development aid (GPU box): phase-A sweep with the speculative kernel allowed for up to 4096 chains, against the
# default build (speculative <= 1024 chains)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -DABN_PHASE_A_SPEC_MAX=4096 -o gpurun_out/libspec4096.so alphabeta_rs_amd/csrc/abn_api.hip
echo "speculative kernel up to 4096 chains (lanes=auto column)"
ABNEUTRAL_HIP_LIB=$PWD/gpurun_out/libspec4096.so python scripts/phase_a_sweep.py 50 100 150 200 300 400
