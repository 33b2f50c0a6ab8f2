#!/bin/bash
# development aid: build comfystereo_amd/libcs_<name>.so from the sources of a git revision (A/B timing of two states of the
# kernels inside ONE gpurun session: the boxes of the pool differ by several percent)
#   tools/build_ref.sh HEAD base   ->  comfystereo_amd/libcs_base.so
set -e
cd "$(dirname "$0")/.."
rev=$1; name=$2
tmp=$(mktemp -d /tmp/csref.XXXXXX)
git archive "$rev" comfystereo_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/comfystereo_amd/csrc" -s -j8 TARGET="$PWD/comfystereo_amd/libcs_$name.so"
rm -rf "$tmp"
ls -la comfystereo_amd/libcs_$name.so
