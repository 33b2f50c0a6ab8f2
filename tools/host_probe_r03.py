import sys, time, cProfile, pstats
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, synth
from comfystereo_amd import host_pipeline
from comfystereo_amd.GenerateStereo import StereoImageNode
mode = sys.argv[1]
n,h,w=32,2160,3840
img = torch.from_numpy(synth.image_f32(1, h, w, seed=1)).expand(n, -1, -1, -1).contiguous()
dep = torch.from_numpy(synth.depth_batch("stepped", n, h, w, channels=3))
node = StereoImageNode()
nargs = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True, 2.0, 6, 12)
hargs = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "polylines_soft", 20.0, 20.0, True, 2.0, 6, 12)
f = {"node": lambda: node.generate(img, dep, *nargs), "host": lambda: host_pipeline.generate_host(img, dep, *hargs),
     "host_prog": lambda: host_pipeline.generate_host(img, dep, *hargs, progress=lambda k: None),
     "host_pageable": lambda: host_pipeline.generate_host(img, dep, *hargs, pinned_outputs=False)}[mode]
r=[]
keep = len(sys.argv) > 2 and sys.argv[2] == "keep"
out = None
for i in range(6):
    t0=time.perf_counter(); out = f(); torch.cuda.synchronize(); r.append(round(n/(time.perf_counter()-t0),1))
    if not keep: del out
print(mode, "keep" if keep else "release", r)
pr = cProfile.Profile(); pr.enable(); out = f(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(5)
