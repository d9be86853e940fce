# measurement aid (GPU box): the pairwise kernel under rocprofv3 with its knobs — a library built with
# -DABN_MEASUREMENT_KNOBS reads ABN_PAIR_BLOCK (2|4 samples per block), ABN_PAIR_SLICES (word slices per pair block),
# ABN_PAIR_GRID (workgroups per CU) and ABN_PAIR_DEBUG (1 = staging only, 2 = pair phase only) from the environment.
#   usage: bash scripts/pw_sweep.sh
set -e
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -ldl -DABN_MEASUREMENT_KNOBS \
  -o build/variants/libabn_knobs.so alphabeta_rs_amd/csrc/abn_api.hip alphabeta_rs_amd/csrc/abn_multi.hip
export ABNEUTRAL_HIP_LIB=build/variants/libabn_knobs.so
echo "default"; bash scripts/pw_profile.sh gpurun_out/pw_sweep/default | grep abn
for dbg in 1 2; do echo "ABN_PAIR_DEBUG=$dbg"; ABN_PAIR_DEBUG=$dbg bash scripts/pw_profile.sh gpurun_out/pw_sweep/dbg$dbg | grep bits; done
for sl in 1 2 4 8; do echo "ABN_PAIR_SLICES=$sl"; ABN_PAIR_SLICES=$sl bash scripts/pw_profile.sh gpurun_out/pw_sweep/s$sl | grep bits; done
for g in 2 3 4; do echo "ABN_PAIR_GRID=$g"; ABN_PAIR_GRID=$g bash scripts/pw_profile.sh gpurun_out/pw_sweep/g$g | grep abn; done
echo "ABN_PAIR_BLOCK=2"; ABN_PAIR_BLOCK=2 bash scripts/pw_profile.sh gpurun_out/pw_sweep/b2 | grep bits
