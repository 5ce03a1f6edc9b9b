#!/bin/bash
# An alternate build of the HIP library for A/B measurements:  tools/build_variant.sh <name> [-DFLAG ...]
# -> exp/lib_<name>.so (load it with PAIRCOUNT_LIB=exp/lib_<name>.so)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall "$@" \
    -shared -o $root/exp/lib_$name.so $root/particle-simulation_amd/csrc/paircount.hip -lpthread
echo built exp/lib_$name.so
