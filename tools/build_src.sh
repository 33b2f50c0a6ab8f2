#!/bin/bash
# development aid: build comfystereo_amd/libcs_<name>.so with ONE translation unit taken from another file
#   tools/build_src.sh <name> <unit, e.g. cs_polypoint> <file.hip>
set -e
cd "$(dirname "$0")/../comfystereo_amd/csrc"
name=$1; unit=$2; src=$3
make -s
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16 -I."
[ "$unit" = cs_polypoint ] && FLAGS="$FLAGS -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS -x hip -c "$src" -o /tmp/${unit}_$name.o
objs=""
for o in cs_*.o; do if [ "$o" = "$unit.o" ]; then objs="$objs /tmp/${unit}_$name.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcs_$name.so $objs
