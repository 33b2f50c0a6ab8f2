#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, time, cProfile, pstats
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, synth
from comfystereo_amd import host_pipeline
n,h,w=32,2160,3840
img = torch.from_numpy(synth.image_f32(1, h, w, seed=1)).expand(n, -1, -1, -1).contiguous()
dep = torch.from_numpy(synth.depth_batch("stepped", n, h, w, channels=3))
hargs = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "polylines_soft", 20.0, 20.0, True, 2.0, 6, 12)
print("torch threads", torch.get_num_threads())
for compact in (True, False):
    host_pipeline.generate_host(img, dep, *hargs, compact=compact)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0=time.perf_counter()
    out = host_pipeline.generate_host(img, dep, *hargs, compact=compact)
    torch.cuda.synchronize()
    dt=time.perf_counter()-t0
    pr.disable()
    print(f"compact={compact}: {n/dt:.1f} fps")
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
    del out
# primitive: pageable -> pinned copy rate
pin = torch.empty((5,h,w,3),dtype=torch.float32,pin_memory=True)
for _ in range(3):
    t0=time.perf_counter(); pin.copy_(img[:5]); dt=time.perf_counter()-t0
    print(f"pageable->pinned copy {pin.numel()*4/dt/1e9:.1f} GB/s")
dev = torch.empty((5,h,w,3),dtype=torch.float32,device='cuda')
for _ in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); dev.copy_(pin, non_blocking=True); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"pinned->device {pin.numel()*4/dt/1e9:.1f} GB/s")
PY
