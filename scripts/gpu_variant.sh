#!/bin/bash
# Build a variant of the library with extra -D flags into /tmp and time a workload with it (experiments only).
# usage: gpu_variant.sh "<flags>" [gpu_time_c2.py args...]
set -e
cd "$(dirname "$0")/.."
FLAGS="$1"; shift
D=/tmp/crabvar_$$
mkdir -p $D/crabml_amd
cp -r crabml_amd/*.py $D/crabml_amd/
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $FLAGS -o $D/crabml_amd/libcrabml_amd.so \
    crabml_amd/csrc/crab_engine.cpp crabml_amd/csrc/crab_programs.cpp crabml_amd/csrc/crab_workplan.cpp crabml_amd/csrc/crab_alignment.cpp crabml_amd/csrc/crab_models.cpp crabml_amd/csrc/crab_oneshot.cpp crabml_amd/csrc/crab_eigen.cpp crabml_amd/csrc/crab_quantiles.cpp crabml_amd/csrc/crab_kernels.hip
cp ${CRAB_SCRIPT:-scripts/gpu_time_c2.py} $D/t.py
sed -i "s#sys.path.insert(0, .*#sys.path.insert(0, \"$D\")#" $D/t.py
cd $D
echo "variant [$FLAGS]: $(timeout -k 10 300 python t.py "$@" 2>&1 | tail -1)"
