#!/bin/bash
# Build a variant of the library with extra -D flags into build_variants/<name>/ (in-tree so that it travels to the GPU box;
# git-ignored).  Experiments only: scripts/gpu_ablate.sh times them.     usage: build_variant.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS="$2"
D=build_variants/$NAME
rm -rf $D; mkdir -p $D/crabml_amd
cp crabml_amd/*.py $D/crabml_amd/
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Xarch_host -mavx2 -Xarch_host -mfma -Iinclude $FLAGS -o $D/crabml_amd/libcrabml_amd.so \
    crabml_amd/csrc/crab_engine.cpp crabml_amd/csrc/crab_programs.cpp crabml_amd/csrc/crab_workplan.cpp crabml_amd/csrc/crab_alignment.cpp crabml_amd/csrc/crab_models.cpp crabml_amd/csrc/crab_oneshot.cpp crabml_amd/csrc/crab_eigen.cpp crabml_amd/csrc/crab_quantiles.cpp crabml_amd/csrc/crab_kernels.hip crabml_amd/csrc/crab_noalloc.hip
echo "built $D [$FLAGS]"
