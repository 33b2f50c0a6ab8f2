#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

  metric : SBS frames/s (+ achieved HBM GB/s) for 4K (3840x2160) forward warp + polylines_soft hole fill,
           divergence 8.0, left-right side-by-side, stepped synthetic depth, widget defaults otherwise
           (direction-aware depth blur ON: strength 20, threshold 20, falloff 2.0, vert 6).
  step   : one pass of the whole hot path (cs_generate: gray depth + min/max, depth blur, warp + fill +
           SBS/mask/depth-map assembly) over the batch of N_FRAMES frames, inputs resident in HBM;
           with --gpus N > 1 the batch is sharded by frame (strong scaling: total work fixed) and the step
           ends with the RCCL all-gather that reassembles the stereoscope tensor on every rank.

  python bench.py [--gpus N --steps K --warmup W] [--frames F] [--no-blur] [--no-cpu-baseline]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     : dominant kernel (k_polytile: warp + polylines fill + assembly) -- algorithmic bytes per launch / its mean
                 duration measured with HIP events on the launch stream inside the timed region
  cpu_baseline : the CPU oracle (C port of the reference's D32 arithmetic, 1 thread) on a bounded sample
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

H, W = 2160, 3840
B_ALG_PER_PIXEL = 80  # float32 node boundary, SBS, CPU-technique mask
HBM_PEAK_GBS = 8000.0


def make_inputs(torch, frames, first, device):
    """Synthetic batch on the device: 8-bit-origin random RGB; stepped radial depth with a moving centre."""
    g = torch.Generator(device=device)
    imgs, deps = [], []
    yy, xx = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float64),
                            torch.arange(W, device=device, dtype=torch.float64), indexing="ij")
    for i in range(first, first + frames):
        g.manual_seed(1000 + i)
        imgs.append(torch.randint(0, 256, (H, W, 3), generator=g, device=device, dtype=torch.int32).to(torch.float32) / 255.0)
        cx, cy = W / 2 + (17 * i) % (W // 4), H / 2 + (11 * i) % (H // 4)
        r = torch.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        d = torch.floor((1.0 - r / r.max()) * 6) / 6
        deps.append(d.to(torch.float32)[..., None].expand(H, W, 3))
    return torch.stack(imgs).contiguous(), torch.stack(deps).contiguous()


def cpu_baseline(frames_sample, blur):
    """The oracle (test infrastructure) timed as the reported CPU baseline: full node path, 1 thread."""
    import numpy as np
    import synth
    from oracle import node_oracle
    img = synth.image_f32(frames_sample, H, W, seed=1)
    depth = synth.depth_batch("stepped", frames_sample, H, W, channels=3)
    t0 = time.perf_counter()
    node_oracle.generate(img, depth, 8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, blur,
                         depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    dt = time.perf_counter() - t0
    return {"value": frames_sample / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{frames_sample} frame(s) of the same workload (4K, both eyes, full node path) through the C oracle, "
                      f"{dt:.1f} s on {os.cpu_count()} visible host cores (1 used)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=64, help="total frames in the batch (sharded over the GPUs)")
    ap.add_argument("--no-blur", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=10)
    ap.add_argument("--verify", action="store_true", help="N > 1: check the reassembled float32 batch against a local float32 run")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from comfystereo_amd import _native, engine, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (CS_BENCH_BACKEND=gloo: development -- exercises the N > 1 code path with several ranks on ONE GPU)
        backend = os.environ.get("CS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % torch.cuda.device_count()
            dist.init_process_group(backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    blur = not a.no_blur

    bounds = sharding.shard_bounds(a.frames, world)
    b0, b1 = bounds[rank], bounds[rank + 1]
    nloc = b1 - b0
    image, depth = make_inputs(torch, nloc, b0, device)
    p = engine.make_params(nloc, H, W, H, W, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, blur, 20.0, 20.0,
                           2.0, 6, 12)
    # N > 1: shards travel over xGMI as uint8 codes (the stereoscope is k/255 exactly: 4x fewer bytes) and are expanded
    # to float32 on every rank after the all-gather.  The block of a rank is produced in chunks; the all-gather of chunk c
    # (RCCL, its own stream) overlaps with the compute of the chunks after it, the expansion of chunk c with the
    # all-gathers still on the wire (sharding.ChunkedGather).
    gathered = torch.empty((a.frames, H, 2 * W, 3), dtype=torch.float32, device=device) if world > 1 else None
    n_chunks = next((k for k in (4, 2, 1) if world > 1 and sharding.ChunkedGather.usable(a.frames, world, k)), 0)
    if world > 1 and n_chunks == 0:
        raise SystemExit(f"--frames {a.frames} does not split evenly over {world} GPUs")
    if world > 1:
        cg = sharding.ChunkedGather(a.frames, n_chunks, (H, 2 * W, 3), torch.uint8, device)
        cp = engine.make_params(cg.cf, H, W, H, W, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, blur, 20.0,
                                20.0, 2.0, 6, 12)
        plans = [engine.Plan(cp, device, stereo_u8=True) for _ in range(n_chunks)]
        plan = plans[0]
    else:
        plan = engine.Plan(p, device)

    def sink(codes, first):
        engine.expand_u8(codes, gathered[first:first + codes.shape[0]])

    def step():
        if world == 1:
            plan.run(image, depth)
            return
        for c in range(n_chunks):
            lo, hi = cg.chunk_range(c)
            stereo, _, _, _ = plans[c].run(image[lo:hi], depth[lo:hi])
            cg.launch(c, stereo)
        for c in range(n_chunks):
            cg.finish(c, sink)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    L = _native.lib()
    for _ in range(a.warmup):
        step()
    fence()
    L.cs_profile(1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tot_ms, launches = ctypes.c_double(), ctypes.c_int()
    _native.check(L.cs_profile_read(ctypes.byref(tot_ms), ctypes.byref(launches)))
    L.cs_profile(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    all_plans = plans if world > 1 else [plan]
    fallback_rows = sum(int(q.stats()[:, 10].sum()) for q in all_plans)
    tile_redo_rows = sum(int(q.stats()[:, 11].sum()) for q in all_plans)
    err_flags = sum(int(q.stats()[:, 9].sum()) for q in all_plans)

    if a.verify and world > 1:
        # every rank: its own block of the reassembled batch == the float32 output computed locally in one piece
        ref = engine.Plan(p, device).run(image, depth)[0]
        ok = torch.equal(gathered[b0:b1], ref)
        other = (rank + 1) % world  # and one foreign frame: recompute it here
        fi, fd = make_inputs(torch, 1, bounds[other], device)
        p1 = engine.make_params(1, H, W, H, W, 3, "polylines_soft", "left-right", 8.0, 0.0, 0.0, 0.5, 2.0, blur, 20.0, 20.0, 2.0, 6, 12)
        ok = ok and torch.equal(gathered[bounds[other]:bounds[other] + 1], engine.Plan(p1, device).run(fi, fd)[0])
        print(f"[verify] rank {rank}: {'OK' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            raise SystemExit(1)
    if rank == 0:
        fps = a.frames * a.steps / dt
        kern_ms = tot_ms.value / max(launches.value, 1)
        frames_per_launch = cg.cf if world > 1 else nloc
        alg_bytes = frames_per_launch * B_ALG_PER_PIXEL * H * W  # per launch of the dominant kernel on this rank
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"polylines_soft_4k_blur{int(blur)}"
                if key in tj:
                    traffic = tj[key]["bytes_per_frame"] * frames_per_launch
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "SBS frames/sec, 4K warp+polylines_soft", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "4K 3840x2160, polylines_soft, left-right SBS, divergence 8.0, stepped depth, "
                                   f"depth blur {'on' if blur else 'off'} (widget defaults)",
                       "frames_total": a.frames, "frames_per_gpu": nloc, "sharding": "by frame, contiguous blocks",
                       "collective": f"all_gather(stereoscope as uint8 codes) over RCCL in {n_chunks} chunk(s) overlapped with compute"
                                     " + expand to float32 on every rank" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_polytile<soft> (+ k_rowwarp<polylines_soft> over the rows it flags)", "kernel_ms": kern_ms, "launches": launches.value,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "pipeline_achieved": a.frames * a.steps * B_ALG_PER_PIXEL * H * W / dt / 1e9 / world},
            "diagnostics": {"rows_redone_by_general_kernel": tile_redo_rows, "rows_replayed_sequentially": fallback_rows,
                            "kernel_error_flags": err_flags},
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(a.cpu_frames, blur)
        elif not a.no_cpu_baseline:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
