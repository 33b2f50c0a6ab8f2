"""The per-frame constants k_gpuwarp_flags wrote (tail of the workspace), and the statistics words (development aid, round 6)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd import engine, _native
np.set_printoptions(precision=6, linewidth=220, suppress=True)
dev = torch.device("cuda")
n, h, w = 1, 32, 64
img = torch.from_numpy(synth.image_f32(n, h, w, seed=5)).to(dev)
depth = torch.from_numpy(synth.depth_batch("blobs", n, h, w, channels=3)).to(dev)
p = engine.make_params(n, h, w, h, w, 3, "gpu_warp", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, False, 20.0, 20.0, 2.0, 6, 12)
plan = engine.Plan(p, dev)
plan.ws.zero_()
plan.run(img, depth); torch.cuda.synchronize()
ws = plan.ws.cpu().numpy()
print("ws bytes", plan.ws_bytes)
st = ws[:64].view(np.uint32)
def ord2f(u):
    u = np.uint32(u)
    v = np.uint32(u ^ 0x80000000) if (u & 0x80000000) else np.uint32(~u)
    return v.view(np.float32)
print("stats words", [hex(int(x)) for x in st], "L min/max", ord2f(st[2]), ord2f(st[3]))
tail = ws[-512:]
print("tail as f32", tail.view(np.float32)[64:64 + 16])
print("tail as u32", [hex(int(x)) for x in tail.view(np.uint32)[64:64 + 16]])
nz = np.nonzero(ws.view(np.uint32)[16:])[0]
print("nonzero words beyond stats: first", nz[:8] + 16, "last", nz[-24:] + 16, "total words", ws.size // 4)
