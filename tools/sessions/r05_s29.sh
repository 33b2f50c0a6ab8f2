#!/bin/bash
# round-5 session 29: 216 instead of 128 list slots for pixels under reversed segments in k_polypoint (16-bit tile-pixel flags): polylines /
# tie tests + fuzz, then A/B against 128 slots (libcs_dcap128.so): the headline workload (must not lose), saturated depth blur on / off,
# blobs; flagged rows per depth kind
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s29; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -k "polylines or tie or replay or parity or lean or saturated or stretch or fuzz or anaglyph or sharp or order or golden or cfg2 or dialect" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
CS_FUZZ_FILLS=polylines_soft,polylines_sharp timeout 400 python tools/extended_fuzz.py 200 525252 > $O/fuzz_poly.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_poly.log
C=comfystereo_amd
LIBS="$C/libcomfystereo_hip.so $C/libcs_dcap128.so" bash tools/abn.sh --n 64 --fill polylines_soft --kind stepped --blur 1 --iters 5 2>&1 | tee $O/ab_headline.txt
for k in clipped blobs; do for b in 0 1; do for L in libcomfystereo_hip.so libcs_dcap128.so; do
  printf "%-22s %-8s blur $b: " $L $k; CS_LIB_PATH=$PWD/$C/$L timeout 600 python tools/quick_bench.py --n 32 --fill polylines_soft --kind $k --blur $b --iters 3 2>&1 | tail -1 | sed 's/.*: //'
done; done; done 2>&1 | tee $O/ab_other.txt
for L in libcomfystereo_hip.so libcs_dcap128.so; do printf "%-22s sharp stepped / clipped blur 1: " $L; for k in stepped clipped; do CS_LIB_PATH=$PWD/$C/$L timeout 600 python tools/quick_bench.py --n 32 --fill polylines_sharp --kind $k --blur 1 --iters 3 2>&1 | tail -1 | sed 's/.*: //' | tr '\n' ' '; done; echo; done 2>&1 | tee -a $O/ab_other.txt
