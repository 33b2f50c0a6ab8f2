#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s8
nproc
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_abi_exports.py -x -q -m gpu > gpurun_out/s8/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/s8/tests.log
timeout 600 python tools/node_host_bench.py --n 32 --iters 2 2>&1 | tail -4
timeout 600 python tools/node_host_bench.py --n 32 --iters 2 --h 1080 --w 1920 2>&1 | tail -4
python - <<'PY'
import sys, time, ctypes
sys.path.insert(0,'.')
import numpy as np, torch
from comfystereo_amd import _native
L=_native.lib()
codes=np.random.default_rng(0).integers(0,256,400_000_000,dtype=np.uint8)
for th in (8,16,32,64):
    out=torch.empty(codes.size,dtype=torch.float32)
    t=time.perf_counter(); L.cs_host_expand_u8(codes.ctypes.data,out.data_ptr(),codes.size,1,0,th); dt=time.perf_counter()-t
    t=time.perf_counter(); L.cs_host_expand_u8(codes.ctypes.data,out.data_ptr(),codes.size,1,0,th); dt2=time.perf_counter()-t
    print(f"expand threads={th}: fresh {codes.size*4/dt/1e9:.1f} GB/s, warm {codes.size*4/dt2/1e9:.1f} GB/s")
    del out
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read())
PY
