#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, time, threading, ctypes
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, numpy as np
from comfystereo_amd import _native
L=_native.lib()
h,w=2160,3840
src = torch.rand((32,h,w,3))          # 3.2 GB pageable, touched
pin = [torch.empty((5,h,w,3),dtype=torch.float32,pin_memory=True) for _ in range(2)]
dev = [torch.empty((5,h,w,3),dtype=torch.float32,device='cuda') for _ in range(2)]
s2 = torch.cuda.Stream()
def stage(i, k): 
    t0=time.perf_counter(); pin[k].copy_(src[5*i:5*i+5]); return pin[k].numel()*4/(time.perf_counter()-t0)/1e9
print("alone (different source slices):", [round(stage(i, i%2),1) for i in range(6)])
# with an H2D in flight from the other pinned buffer
r=[]
for i in range(6):
    with torch.cuda.stream(s2):
        dev[(i+1)%2].copy_(pin[(i+1)%2], non_blocking=True)
    r.append(round(stage(i, i%2),1))
torch.cuda.synchronize()
print("with H2D of the other buffer in flight:", r)
# with expansion threads running
codes=np.random.default_rng(0).integers(0,256,200_000_000,dtype=np.uint8)
out=torch.empty(codes.size,dtype=torch.float32)
stop=False
def expander():
    while not stop:
        L.cs_host_expand_u8(codes.ctypes.data,out.data_ptr(),codes.size,1,0,32)
th=threading.Thread(target=expander); th.start()
time.sleep(0.1)
print("with 32 expansion threads running:", [round(stage(i, i%2),1) for i in range(6)])
stop=True; th.join()
for nt in (8,16,32,64):
    torch.set_num_threads(nt)
    print("torch threads", nt, [round(stage(i, i%2),1) for i in range(4)])
PY
