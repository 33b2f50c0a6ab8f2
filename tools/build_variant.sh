#!/bin/bash
# development aid: build comfystereo_amd/libcs_<name>.so with extra compiler flags on one translation unit (A/B timing with
# CS_LIB_PATH):   tools/build_variant.sh noslp cs_polypoint "-fno-slp-vectorize"
set -e
cd "$(dirname "$0")/../comfystereo_amd/csrc"
name=$1; unit=$2; extra=$3
make -s
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16"
[ "$unit" = cs_polypoint ] && FLAGS="$FLAGS -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS $extra -c $unit.hip -o /tmp/${unit}_$name.o
objs=""
for o in cs_*.o; do if [ "$o" = "$unit.o" ]; then objs="$objs /tmp/${unit}_$name.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcs_$name.so $objs
ls -la ../libcs_$name.so
