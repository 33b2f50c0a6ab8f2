cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# round-5 session 11b: the fixture sequence of session 11 again (three processes) and six fresh pytest processes of the fixture test: no failure
O=gpurun_out/r05_s11; mkdir -p $O
for i in 1 2 3; do timeout 600 python tools/sessions/r05_s11_debug.py 2>&1 | grep -v amdgpu.ids | tee -a $O/debug_sharp.txt; done
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_dialect.py -x -q -m gpu -k fixture 2>&1 | tail -1; done
