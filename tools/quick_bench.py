"""Ad-hoc timing of the fused batch path (development aid; bench.py is the contract)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd import engine, _native

# development switches from the environment (tools only: the library itself never reads the environment)
for _env, _key in (("CS_DBG", "dbg"), ("CS_NO_TILE", "no_tile"), ("CS_PT_VARIANT", "pt_variant"),
                   ("CS_BLUR_TWO_PASS", "blur_two_pass"), ("CS_BLUR_EDGES_SCALAR", "blur_edges_scalar"),
                   ("CS_BLUR_FULL_COPY", "blur_full_copy"), ("CS_CHUNKS", "chunks"), ("CS_NO_REPLAY_KERNEL", "no_replay_kernel")):
    if os.environ.get(_env):
        _native.debug_set(_key, int(os.environ[_env]))

if os.environ.get("CS_MESH"):
    engine.MESH_WARP = True   # gpu_warp as the mesh-quality rasteriser (cs_params.flags bit 2)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4)
ap.add_argument("--h", type=int, default=2160)
ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--fill", default="polylines_soft")
ap.add_argument("--mode", default="left-right")
ap.add_argument("--kind", default="stepped")
ap.add_argument("--div", type=float, default=8.0)
ap.add_argument("--blur", type=int, default=0)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dialect", default="D32", help="engine.DIALECT: D32 | f64-disparity | int64-sum | D64")
ap.add_argument("--tie-pool-mb", type=int, default=0, help="extra workspace for the stretch-replay pool (engine.Plan tie_pool_bytes)")
a = ap.parse_args()

engine.DIALECT = a.dialect
dev = torch.device("cuda:0")
img = torch.from_numpy(synth.image_f32(1, a.h, a.w, seed=1)).to(dev).expand(a.n, -1, -1, -1).contiguous()
depth = torch.from_numpy(synth.depth_batch(a.kind, a.n, a.h, a.w, channels=3)).to(dev)
p = engine.make_params(a.n, a.h, a.w, a.h, a.w, 3, a.fill, a.mode, a.div, 0.0, 0.0, 0.5, 2.0, bool(a.blur), 20.0, 20.0, 2.0, 6, 12)
plan = engine.Plan(p, dev, tie_pool_bytes=a.tie_pool_mb << 20)
plan.run(img, depth); torch.cuda.synchronize()
st = plan.stats()
import os
if os.environ.get("CS_DBG") == "20":
    nwg = 2 * a.n * ((a.h * ((a.w + 511) // 512) + 60) // 61)
    print("mean phase latency per workgroup that recorded it, us (phase = index):",
          [round(int(x) / max(int(c), 1) / 100.0, 2) for x, c in zip(st[:, 12].tolist(), st[:, 13].tolist())])
    print("workgroups per phase:", [int(c) for c in st[:, 13].tolist()])
print("chain px, generic px, waves with generic, waves:", st[:, 12].sum().item(), st[:, 13].sum().item(), st[:, 14].sum().item(), st[:, 15].sum().item())
print("stats word 12 (dev builds: hazard reason bits):", [hex(int(v)) for v in st[:, 12].tolist()])
print("seq-fallback rows:", st[:, 10].tolist(), "tile-redo rows:", st[:, 11].tolist(), "errors:", st[:, 9].tolist())
t0 = time.perf_counter()
for _ in range(a.iters):
    plan.run(img, depth)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
hw = a.h * a.w
print(f"{a.fill} {a.mode} {a.kind} n={a.n} {a.w}x{a.h} blur={a.blur}: {dt*1e3:.2f} ms/batch, {a.n/dt:.1f} fps, "
      f"{a.n*80*hw/dt/1e9:.1f} GB/s algorithmic")
