#!/bin/bash
# round-4 session 24: k_gpuwarp column pass with ONE conservative pre-filter per round (product sign + magnitude) instead of a
# cascade of five exec-mask branches, division core + rare full-division branch: warp tests, A/B against the build before
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s24; mkdir -p $O
C=comfystereo_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_dropin.py tests/test_gpu_lazy_blur.py tests/test_gpu_fuzz.py -x -q -m gpu -k "warp or cfg4 or gpu_warp or fuzz" > $O/tests.log 2>&1; echo "warp tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=gpu_warp timeout 200 python tools/extended_fuzz.py 60 919000 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
LIBS="$C/libcs_gwprev.so $C/libcomfystereo_hip.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind radial --div 4.5 --blur 1 --iters 10 2>&1 | tee $O/ab_1080p.txt
LIBS="$C/libcs_gwprev.so $C/libcomfystereo_hip.so" tools/abn.sh --n 32 --fill gpu_warp --blur 1 --iters 10 2>&1 | tee $O/ab_4k.txt
LIBS="$C/libcs_gwprev.so $C/libcomfystereo_hip.so" tools/abn.sh --n 128 --h 1080 --w 1920 --fill gpu_warp --kind stepped --div 8 --blur 1 --iters 10 2>&1 | tee $O/ab_1080p_stepped.txt
