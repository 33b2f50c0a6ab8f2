"""End-to-end timing of StereoImageNode.generate with HOST tensors in and out (how ComfyUI calls the node):
PCIe-inclusive frames/s.  Development aid; the number is quoted in DESIGN.md, never as bench.py's `value`."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth
from comfystereo_amd.GenerateStereo import StereoImageNode

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--h", type=int, default=2160)
ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--fill", default="Fill - Polylines Soft")
ap.add_argument("--prewarm", type=int, default=1, help="1: what GenerateStereo.py does at import inside ComfyUI (host_pipeline."
                "prewarm of this shape, waited for: ComfyUI loads its models meanwhile); 0: cold caches")
ap.add_argument("--pin-cap-gb", type=float, default=-1.0, help="host_pipeline.PINNED_POOL_BYTES in GB (default: the module's)")
ap.add_argument("--first-only", type=int, default=0, help="1: stop after the first two calls (first-call measurements per shape)")
a = ap.parse_args()
from comfystereo_amd import host_pipeline as _hp
if a.pin_cap_gb >= 0:
    _hp.PINNED_POOL_BYTES = int(a.pin_cap_gb * (1 << 30))
from comfystereo_amd.GenerateStereo import FILL_TECHNIQUE_MAPPING as _FM
if a.prewarm:
    t0 = time.perf_counter()
    _hp.prewarm(a.n, a.h, a.w, fill=_FM[a.fill])
    print(f"prewarm (opt-in, off the first call's path): {time.perf_counter() - t0:.2f} s")
img = torch.from_numpy(synth.image_f32(1, a.h, a.w, seed=1)).expand(a.n, -1, -1, -1).contiguous()
dep = torch.from_numpy(synth.depth_batch("stepped", a.n, a.h, a.w, channels=3))
node = StereoImageNode()
args = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0, a.fill, 20.0, 20.0, True, 2.0, 6, 12)
t0 = time.perf_counter()
out = node.generate(img, dep, *args)
torch.cuda.synchronize()
dt_first = time.perf_counter() - t0
print(f"FIRST call of the process ({'prewarmed' if a.prewarm else 'cold'}): {dt_first*1e3:.1f} ms -> {a.n/dt_first:.1f} frames/s")
t0 = time.perf_counter()
out2 = node.generate(img, dep, *args)   # (the first call's results still held by the caller)
torch.cuda.synchronize()
dt_second = time.perf_counter() - t0
print(f"second call, first results still held: {dt_second*1e3:.1f} ms -> {a.n/dt_second:.1f} frames/s")
print(f"pinned cap {_hp.PINNED_POOL_BYTES / 2**30:.1f} GB")
print(f"results pinned: {out[0].is_pinned()}; pinned host memory reserved by PyTorch: "
      f"{torch.cuda.host_memory_stats().get('reserved_bytes.current', 0) / 1e9:.2f} GB; 4K-equivalent first call: "
      f"{a.n * (a.h * a.w) / (2160 * 3840) / dt_first:.1f} frames/s")
del out2
if a.first_only:
    sys.exit(0)
t0 = time.perf_counter()
for _ in range(a.iters):
    out = node.generate(img, dep, *args)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
pinned = out[0].is_pinned()
shapes = [tuple(t.shape) for t in out]
nbytes = sum(t.numel() * t.element_size() for t in out)
del out  # results released first: the next call finds their pinned blocks in PyTorch's cache
t0 = time.perf_counter()
for _ in range(a.iters):
    out = node.generate(img, dep, *args)
    del out
torch.cuda.synchronize()
dt_cached = (time.perf_counter() - t0) / a.iters
gb = nbytes / 1e9 + (img.numel() + dep.numel()) * 4 / 1e9
print(f"host tensors in/out: {a.n} frames {a.w}x{a.h}: {dt*1e3:.1f} ms -> {a.n/dt:.1f} frames/s, {gb/dt:.1f} GB/s over PCIe+host copies "
      f"(outputs are {'pinned' if pinned else 'pageable'} CPU tensors, freshly allocated); "
      f"with the previous results released first: {a.n/dt_cached:.1f} frames/s, {gb/dt_cached:.1f} GB/s")
# caller-provided pinned result tensors, kept across calls (host_pipeline.generate_host(..., out=...))
from comfystereo_amd import host_pipeline
from comfystereo_amd.GenerateStereo import FILL_TECHNIQUE_MAPPING
fill = FILL_TECHNIQUE_MAPPING[a.fill]
outs = tuple(torch.empty(sh, dtype=torch.float32).pin_memory() for sh in host_pipeline.result_shapes(img.shape, "left-right", fill))
hargs = (8.0, 0.0, "left-right", 0.0, 0.5, 2.0, fill, 20.0, 20.0, True, 2.0, 6, 12)
host_pipeline.generate_host(img, dep, *hargs, out=outs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    host_pipeline.generate_host(img, dep, *hargs, out=outs)
torch.cuda.synchronize()
dt_out = (time.perf_counter() - t0) / a.iters
print(f"results into caller-provided pinned tensors (out=): {a.n/dt_out:.1f} frames/s, {gb/dt_out:.1f} GB/s")
if fill != "gpu_warp":   # the float32 boundary of round 2 for comparison (compact=False)
    host_pipeline.generate_host(img, dep, *hargs, compact=False)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        o = host_pipeline.generate_host(img, dep, *hargs, compact=False)
        del o
    torch.cuda.synchronize()
    dt_f = (time.perf_counter() - t0) / a.iters
    print(f"float32 boundary (compact=False, pinned results): {a.n/dt_f:.1f} frames/s")
