#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CS_CHUNKS=1
LIBS="comfystereo_amd/libcs_base.so comfystereo_amd/libcomfystereo_hip.so comfystereo_amd/libcs_t480.so comfystereo_amd/libcs_occ8.so" tools/abn.sh --n 32 --blur 0 --iters 10
