#!/bin/bash
# development aid: build libcs_<name>.so with cs_polytile.hip taken from a git revision (A/B timing)
#   tools/build_variant.sh HEAD base   ->  comfystereo_amd/libcs_base.so
set -e
cd "$(dirname "$0")/../comfystereo_amd/csrc"
rev=$1; name=$2
git show "$rev:comfystereo_amd/csrc/cs_polytile.hip" > /tmp/cs_polytile_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16 -I. -c -x hip /tmp/cs_polytile_$name.hip -o /tmp/cs_polytile_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcs_$name.so cs_abi.o cs_blur.o cs_gpuwarp.o cs_rowwarp.o /tmp/cs_polytile_$name.o
ls -la ../libcs_$name.so
