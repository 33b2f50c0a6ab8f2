#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, time, threading, ctypes
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, numpy as np
from comfystereo_amd import _native
L=_native.lib()
h,w=2160,3840
src = torch.rand((32,h,w,3))
pin = [torch.empty((5,h,w,3),dtype=torch.float32,pin_memory=True) for _ in range(2)]
def stage(i, k, nt):
    s = src[5*i:5*i+5]
    t0=time.perf_counter(); L.cs_host_copy(pin[k].data_ptr(), s.data_ptr(), s.numel()*4, nt); return round(s.numel()*4/(time.perf_counter()-t0)/1e9,1)
def stage_t(i, k):
    t0=time.perf_counter(); pin[k].copy_(src[5*i:5*i+5]); return round(pin[k].numel()*4/(time.perf_counter()-t0)/1e9,1)
for nt in (8,16,32,64):
    print("cs_host_copy threads", nt, [stage(i%6, i%2, nt) for i in range(12)])
print("torch copy_", [stage_t(i%6, i%2) for i in range(12)])
pg = torch.empty((5,h,w,3),dtype=torch.float32)
def pp(i):
    t0=time.perf_counter(); pg.copy_(src[5*i:5*i+5]); return round(pg.numel()*4/(time.perf_counter()-t0)/1e9,1)
print("torch copy_ pageable->pageable", [pp(i%6) for i in range(12)])
PY
