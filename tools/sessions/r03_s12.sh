#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -2
python - <<'PY'
import sys, time, cProfile, pstats
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, synth
from comfystereo_amd import host_pipeline
n,h,w=32,2160,3840
img = torch.from_numpy(synth.image_f32(1, h, w, seed=1)).expand(n, -1, -1, -1).contiguous()
dep = torch.from_numpy(synth.depth_batch("stepped", n, h, w, channels=3))
hargs = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "polylines_soft", 20.0, 20.0, True, 2.0, 6, 12)
for compact in (True, False):
    host_pipeline.generate_host(img, dep, *hargs, compact=compact)
    torch.cuda.synchronize()
    for rep in range(3):
        pr = cProfile.Profile(); pr.enable()
        t0=time.perf_counter()
        out = host_pipeline.generate_host(img, dep, *hargs, compact=compact)
        torch.cuda.synchronize()
        dt=time.perf_counter()-t0
        pr.disable()
        print(f"compact={compact}: {n/dt:.1f} fps")
        del out
    pstats.Stats(pr).sort_stats("tottime").print_stats(6)
PY
