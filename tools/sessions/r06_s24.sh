#!/bin/bash
# round-6 session 24: anaglyphs beyond the row kernel's stash form through the side-by-side form + composition (forward fills, none_post /
# inverse_post), the point-kernel test of numba's sweep with the corrected bound; then every -m gpu test, smoke, fuzz of the forward fills
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dialect.py tests/test_abi_exports.py -x -q -k "wide or refused or numba or abi or max_width" > $O/tests_a.log 2>&1; echo "new tests rc=$?"; tail -3 $O/tests_a.log
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
CS_FUZZ_FILLS=none,naive,naive_interpolating,inverse,none_post,inverse_post timeout 300 python tools/extended_fuzz.py 80 2401 > $O/fuzz_fwd.log 2>&1; echo "fuzz fwd rc=$?"; tail -1 $O/fuzz_fwd.log
