#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, time, ctypes
sys.path.insert(0,'.')
import numpy as np, torch
from comfystereo_amd import _native
L=_native.lib()
N=1_600_000_000   # 6.4 GB of float32 per tensor
codes=np.random.default_rng(0).integers(0,256,N,dtype=np.uint8)
def fill(t):
    t0=time.perf_counter(); L.cs_host_expand_u8(codes.ctypes.data, t.data_ptr(), N, 1, 0, 32); return round(N*4/(time.perf_counter()-t0)/1e9,1)
def thp():
    d={}
    for ln in open('/proc/meminfo'):
        if ln.startswith(('AnonHugePages','MemFree')): d[ln.split(':')[0]]=ln.split()[1]
    return d
print(thp())
kept=[]
r=[]
for i in range(6):
    t=torch.empty(N,dtype=torch.float32); r.append(fill(t)); kept.append(t)
print("fresh tensors, all kept:", r, thp())
kept.clear()
r=[]
for i in range(6):
    t=torch.empty(N,dtype=torch.float32); r.append(fill(t)); del t
print("fresh tensors, released each time:", r, thp())
r=[]
for i in range(6):
    t=torch.empty(N,dtype=torch.float32); r.append(fill(t)); kept.append(t)
print("kept again:", r, thp())
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip(), '|', open('/sys/kernel/mm/transparent_hugepage/defrag').read().strip())
PY
