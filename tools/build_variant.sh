#!/bin/bash
# Developer A/B: a second library piet_metal_amd/lib/libpiet_metal_amd_<name>.so built with extra -D flags
#   tools/build_variant.sh <name> "<flags>"       (load it with PM_LIB_DEV=1 PM_LIB_VARIANT=<name>; never committed)
set -e
cd "$(dirname "$0")/../piet_metal_amd/csrc"
NAME=$1; FLAGS=$2
B=../_build/var_$NAME; mkdir -p $B
for f in pm_bin pm_coarse pm_fine pm_frame pm_flatten pm_context pm_gather; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Igfx950 -mllvm -amdgpu-kernarg-preload-count=13 -mllvm -structurizecfg-skip-uniform-regions $FLAGS -c $f.hip -o $B/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpiet_metal_amd_$NAME.so $B/*.o ../_build/pm_encoder.o ../_build/pm_svg.o -ldl
ls -la ../lib/libpiet_metal_amd_$NAME.so
