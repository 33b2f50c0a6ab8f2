#!/bin/bash
# round-5 session 3: (1) naive_interpolating with bit-row scans and the wave-parallel quirk replay: parity (every GPU test that
# names the technique + a fuzz slice), A/B against the previous kernel (libcs_fwdold.so); (2) polylines_sharp without the 21 spilled
# vector registers (80-register instantiation, 6 workgroups per CU: libcs_sharp6.so) A/B + scratch traffic; (3) the host tests the
# previous session did not reach; (4) first-call matrix of the host path (huge-page results, pinned cap)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s3; mkdir -p $O
C=comfystereo_amd
timeout 1200 python -m pytest tests -x -q -m gpu -k "naive or fwdtile or interp or chunk or fullsize or golden or node" > $O/tests_naive.log 2>&1; echo "naive_interpolating tests rc=$?"; tail -3 $O/tests_naive.log
CS_FUZZ_FILLS=naive_interpolating timeout 400 python tools/extended_fuzz.py 200 515151 > $O/fuzz_naive.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz_naive.log
LIBS="$C/libcs_fwdold.so $C/libcomfystereo_hip.so" tools/abn.sh --n 32 --fill naive_interpolating --blur 0 --iters 10 2>&1 | tee $O/ab_naive.txt
LIBS="$C/libcs_fwdold.so $C/libcomfystereo_hip.so" tools/abn.sh --n 32 --fill naive_interpolating --blur 0 --iters 10 --kind blobs 2>&1 | tee -a $O/ab_naive.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_sharp6.so $C/libcs_sharp5.so" PMC=k_polypoint tools/abn.sh --n 32 --fill polylines_sharp --blur 0 --iters 10 2>&1 | tee $O/ab_sharp.txt
LIBS="$C/libcomfystereo_hip.so $C/libcs_sharp6.so" tools/abn.sh --n 32 --fill polylines_sharp --blur 0 --iters 10 --kind blobs 2>&1 | tee -a $O/ab_sharp.txt
timeout 900 python -m pytest tests/test_gpu_chunks.py tests/test_gpu_dropin.py tests/test_gpu_sharded.py -x -q -m gpu > $O/tests_host.log 2>&1; echo "host-side tests rc=$?"; tail -3 $O/tests_host.log
bash tools/host_first_call.sh 2>&1 | tee $O/host_first_call.txt
