#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X (and, with --config, its other configurations).

  metric : SBS frames/s (+ achieved HBM GB/s) for 4K (3840x2160) forward warp + polylines_soft hole fill,
           divergence 8.0, left-right side-by-side, stepped synthetic depth, widget defaults otherwise
           (direction-aware depth blur ON: strength 20, threshold 20, falloff 2.0, vert 6).
  step   : one pass of the whole hot path (cs_generate: gray depth + min/max, depth blur, warp + fill +
           SBS/mask/depth-map assembly) over the batch of frames, inputs resident in HBM;
           with --gpus N > 1 the batch is sharded by frame (strong scaling: total work fixed) and the step
           ends with the RCCL all-gather that reassembles the stereoscope tensor on every rank.

  python bench.py [--gpus N --steps K --warmup W] [--config metric|cfg2|cfg3|cfg4|cfg5] [--frames F] [--no-blur]
                  [--no-cpu-baseline] [--verify]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with extra objects:
  roofline     : dominant kernel of the configuration -- algorithmic bytes per launch / its mean duration measured with HIP
                 events on the launch stream inside the timed region
  cpu_baseline : the CPU oracle (C port of the reference's D32 arithmetic) on a bounded sample of the same workload:
                 1 thread; all host cores as OpenMP over rows in one process (the analogue of the reference's numba prange);
                 64 single-threaded processes, a frame each
  value_blur_off (metric config, N = 1): the same workload with the depth blur switched off
  value_dialect_d64 (metric config, N = 1): the same workload under dialect D64 (what an install of the reference with numba computes)
  split        (N > 1): kernels only / kernels + all-gather / end to end
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

HBM_PEAK_GBS = 8000.0

# BASELINE.json configs.  bytes_px: algorithmic bytes per source pixel at the float32 node boundary
# (80: SBS + output-shaped mask, 76: gpu_warp SBS with the eye-shaped mask, 64: anaglyph).
CONFIGS = {
    "metric": dict(h=2160, w=3840, frames=64, fill="polylines_soft", mode="left-right", div=8.0, depth="stepped", bytes_px=80,
                   name="SBS frames/sec, 4K warp+polylines_soft", kernel="k_polypoint (+ k_rowwarp<polylines_soft> over the rows it flags)",
                   what="4K 3840x2160, polylines_soft, left-right SBS, divergence 8.0, stepped depth"),
    "cfg2": dict(h=1080, w=1920, frames=32, fill="polylines_soft", mode="left-right", div=3.5, depth="stepped", bytes_px=80,
                 name="SBS frames/sec, 1080p warp+polylines_soft (BASELINE cfg 2)", kernel="k_polypoint",
                 what="BASELINE.json configs[1]: 1080p, divergence 3.5, polylines_soft, left-right SBS, stepped depth, batch of 32 frames"),
    "cfg3": dict(h=2160, w=3840, frames=16, fill="hybrid_edge", mode="left-right", div=8.0, depth="stepped", bytes_px=80,
                 name="SBS frames/sec, 4K hybrid_edge + depth blur (BASELINE cfg 3)", kernel="k_hybrid_splat_tile<fused outputs> + k_hybrid_gaps (the HIP-event bracket covers both launches)",
                 what="BASELINE.json configs[2]: 4K, divergence 8.0, hybrid_edge fill + edge-aware depth blur, batch of 16 frames"),
    "cfg4": dict(h=1080, w=1920, frames=256, fill="gpu_warp", mode="left-right", div=4.5, depth="radial", bytes_px=76,
                 name="SBS frames/sec, 256x1080p gpu_warp (BASELINE cfg 4)", kernel="k_gpuwarp",
                 what="BASELINE.json configs[3]: batch of 256 1080p frames, gpu_warp fill, left-right SBS, radial depth with a moving centre"),
    "cfg5": dict(h=2160, w=3840, frames=64, fill="none", mode="red-cyan-anaglyph", div=8.0, depth="stepped", bytes_px=64,
                 name="anaglyph frames/sec, 64x4K no-fill + mask (BASELINE cfg 5)", kernel="k_fwdtile (halo-tile forward map, both eyes per workgroup)",
                 what="BASELINE.json configs[4]: batch of 64 4K frames, red-cyan-anaglyph + no_fill mask output, stepped depth"),
    # the two other fills north_star names, at the metric's size (profiles of their own)
    "naive_interp": dict(h=2160, w=3840, frames=64, fill="naive_interpolating", mode="left-right", div=8.0, depth="stepped", bytes_px=80,
                         name="SBS frames/sec, 4K warp+naive_interpolating", kernel="k_fwdtile<naive_interpolating>",
                         what="4K 3840x2160, naive_interpolating, left-right SBS, divergence 8.0, stepped depth"),
    "sharp": dict(h=2160, w=3840, frames=64, fill="polylines_sharp", mode="left-right", div=8.0, depth="stepped", bytes_px=80,
                  name="SBS frames/sec, 4K warp+polylines_sharp", kernel="k_polypoint<SHARP> (+ k_rowwarp<polylines_sharp> over the rows it flags)",
                  what="4K 3840x2160, polylines_sharp, left-right SBS, divergence 8.0, stepped depth"),
}
UI_FILL = {"naive_interpolating": "Fill - Naive interpolating", "polylines_sharp": "Fill - Polylines Sharp", "polylines_soft": "Fill - Polylines Soft", "hybrid_edge": "Imperfect fill - Hybrid Edge", "gpu_warp": "GPU Warp (Fast)",
           "none": "No fill"}


def make_inputs(torch, cfg, frames, first, device):
    """Synthetic batch on the device: 8-bit-origin random RGB; radial / stepped depth with a moving centre."""
    H, W = cfg["h"], cfg["w"]
    g = torch.Generator(device=device)
    imgs, deps = [], []
    yy, xx = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float64),
                            torch.arange(W, device=device, dtype=torch.float64), indexing="ij")
    for i in range(first, first + frames):
        g.manual_seed(1000 + i)
        imgs.append(torch.randint(0, 256, (H, W, 3), generator=g, device=device, dtype=torch.int32).to(torch.float32) / 255.0)
        cx, cy = W / 2 + (17 * i) % (W // 4), H / 2 + (11 * i) % (H // 4)
        r = torch.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        d = 1.0 - r / r.max()
        if cfg["depth"] == "stepped":
            d = torch.floor(d * 6) / 6
        deps.append(d.to(torch.float32)[..., None].expand(H, W, 3))
    return torch.stack(imgs).contiguous(), torch.stack(deps).contiguous()


def cpu_baseline(cfg, frames_sample, blur, seconds_budget=12.0):
    """The oracle (test infrastructure) timed as the reported CPU baseline on the host cores of the GPU box: the full node
    path on a bounded sample of the same workload, first on one thread, then on all cores (rows in parallel)."""
    import synth
    from oracle import node_oracle, oracle
    H, W = cfg["h"], cfg["w"]
    img = synth.image_f32(frames_sample, H, W, seed=1)
    depth = synth.depth_batch(cfg["depth"], frames_sample, H, W, channels=3)

    def run(n):
        t0 = time.perf_counter()
        node_oracle.generate(img[:n], depth[:n], cfg["div"], 0.0, cfg["mode"], 0.0, 0.5, 2.0, UI_FILL[cfg["fill"]], 20.0, 20.0, blur,
                             depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
        return time.perf_counter() - t0

    oracle.set_threads(1)
    t_one = run(1)
    n1 = max(1, min(frames_sample, int(seconds_budget / max(t_one, 1e-3))))
    dt1 = run(n1) if n1 > 1 else t_one
    # all cores, form 1 ("OpenMP over rows = the prange analogue, at all cores"): ONE process, the C oracle's row loops
    # on every visible core (oracle.set_threads(os.cpu_count())); the node glue around them (numpy) stays on one thread
    cores = os.cpu_count() or 1
    n_omp = max(1, min(frames_sample, 4))
    omp_runs = []   # (threads, frames/s): every visible core first, then fewer threads -- 256 threads on 2 160 short rows can cost more than they give
    for threads in sorted({cores, min(cores, 64), min(cores, 16)}, reverse=True):
        oracle.set_threads(threads)
        run(1)
        omp_runs.append((threads, n_omp / run(n_omp)))
    oracle.set_threads(1)
    omp_all = omp_runs[0]
    omp_best = max(omp_runs, key=lambda r: r[1])
    # all cores, form 2: frames in parallel, one single-threaded process per frame (the port is memory-bound when run in parallel:
    # more than 64 workers do not help on the 256-core boxes of the pool)
    import cpu_allcores
    workers = max(1, min(os.cpu_count() or 1, 64))
    fps_all, mean_s, dta = cpu_allcores.run(workers, workers, H, W, cfg["depth"], UI_FILL[cfg["fill"]], cfg["mode"], cfg["div"], blur)
    return {"value": n1 / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n1} frame(s) of the same workload (both eyes, full node path) through the C oracle, {dt1:.1f} s on 1 of "
                      f"{os.cpu_count()} visible host cores",
            "openmp_rows": {"value": omp_all[1], "unit": "frames/s", "cores": omp_all[0],
                            "sample": f"{n_omp} frame(s), one process, OpenMP over rows on all {cores} visible host cores (oracle.set_threads); the "
                                      "node glue around the row loops (numpy: depth blur, conversions) stays on one thread",
                            "by_threads": {str(t): v for t, v in omp_runs}, "best": {"threads": omp_best[0], "value": omp_best[1]}},
            "all_cores": {"value": fps_all, "unit": "frames/s", "cores": workers,
                          "sample": f"{workers} frame(s) in parallel, one single-threaded oracle process per frame on {workers} of "
                                    f"{os.cpu_count()} host cores, {dta:.1f} s ({mean_s:.2f} s per frame per core)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="metric", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=0, help="total frames in the batch (sharded over the GPUs); 0 = the config's")
    ap.add_argument("--no-blur", action="store_true")
    ap.add_argument("--depth", default="", choices=["", "clipped", "random8", "blobs", "scene8"],
                    help="replace the config's synthetic depth: clipped = saturated to exact 0 / 1 over large areas (exact closeness "
                         "ties: order-dependent rows, the stretch replay kernel), random8 = 8-bit noise (every row replayed whole), "
                         "scene8 = an 8-bit gradient with flat ellipses (quantised smooth depth with object silhouettes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--verify", action="store_true", help="N > 1: check the reassembled float32 batch against a local float32 run")
    ap.add_argument("--gather", default="collective", choices=["collective", "p2p", "none"],
                    help="N > 1: all_gather_into_tensor (RCCL), the direct peer fan-out (batched send / recv to every peer), or none: "
                         "every rank keeps the float32 results of its own frame block (sharding.generate_sharded(..., gather=()), the "
                         "deployment form of DESIGN.md section 6) -- `value` is then the kernels-only rate and the config says so")
    ap.add_argument("--no-other-depths", action="store_true", help="metric config, N = 1: skip the radial / blobs depth lines")
    a = ap.parse_args()
    cfg = dict(CONFIGS[a.config])
    frames = a.frames or cfg["frames"]
    H, W = cfg["h"], cfg["w"]

    import torch
    import torch.distributed as dist
    from comfystereo_amd import _native, engine, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (CS_BENCH_BACKEND=gloo: development / tests -- exercises the N > 1 code path with several ranks on ONE GPU)
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
    gpu_warp = cfg["fill"] == "gpu_warp"
    batch_size = 12

    def params(n, blur_on=blur):
        return engine.make_params(n, H, W, H, W, 3, cfg["fill"], cfg["mode"], cfg["div"], 0.0, 0.0, 0.5, 2.0, blur_on, 20.0, 20.0,
                                  2.0, 6, batch_size)

    bounds = sharding.shard_bounds(frames, world, min(batch_size, frames) if gpu_warp else 1)
    b0, b1 = bounds[rank], bounds[rank + 1]
    nloc = b1 - b0
    image, depth = make_inputs(torch, cfg, nloc, b0, device)
    if a.depth:   # (host-generated by tools/synth.py, eight distinct frames repeated)
        import numpy as np
        import synth
        base = np.stack([synth.DEPTHS[a.depth](H, W, seed=b0 + i) for i in range(min(8, nloc))])
        reps = (nloc + base.shape[0] - 1) // base.shape[0]
        depth = torch.from_numpy(np.tile(base, (reps, 1, 1))[:nloc]).to(device)[..., None].expand(nloc, H, W, 3).contiguous()
        cfg["what"] = cfg["what"].replace("stepped depth", f"{a.depth} depth").replace("radial depth with a moving centre", f"{a.depth} depth")
    out_h, out_w = engine.output_shape(params(1))[:2]

    # N > 1, CPU techniques: shards travel over xGMI as uint8 codes (the stereoscope is k/255 exactly: 4x fewer bytes), the
    # block of a rank is produced in chunks whose all-gathers overlap with the following compute, and every gathered chunk
    # is expanded to float32 on every rank (sharding.ShardedStereoJob).  gpu_warp colours are genuine floats: one float32
    # all-gather of the rank's block.
    job = None
    no_gather = world > 1 and a.gather == "none"
    if world > 1 and not gpu_warp and not no_gather:
        job = sharding.ShardedStereoJob(params, frames, (out_h, out_w, 3), device, method=a.gather)
        plans = job.plans
    else:
        # (--depth random8 with the blur off: every row is order-dependent and one stretch long -- room for all of them to export)
        tie_pool = nloc * H * W * 2 * 6 + (64 << 20) if (a.depth == "random8" and cfg["fill"].startswith("polylines")) else 0
        plan = engine.Plan(params(nloc), device, tie_pool_bytes=tie_pool)
        plans = [plan]
        if world > 1 and not no_gather:
            sizes = {bounds[r + 1] - bounds[r] for r in range(world)}
            if len(sizes) != 1:
                raise SystemExit(f"--frames {frames} does not split into equal sub-batch-aligned blocks over {world} GPUs")
    gathered_f32 = [None]

    def compute_only():
        if job is not None:
            job.compute(image, depth)
        else:
            plan.run(image, depth)

    def step(expand=True):
        if world == 1 or no_gather:
            plan.run(image, depth)
        elif job is not None:
            job.step(image, depth, expand=expand)
        else:
            gathered_f32[0] = sharding.all_gather_frames(plan.run(image, depth)[0], bounds, method=a.gather)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    ranks = None
    if world > 1:   # what the collective library really spans: an all-reduce of ones, every rank's device
        ones = torch.ones(1, dtype=torch.int32, device=device)
        dist.all_reduce(ones)
        devs = [None] * world
        dist.all_gather_object(devs, {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(local_rank),
                                      "pci": getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", None)})
        ranks = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "all_reduce_of_ones": int(ones.item()),
                 "devices": devs}
    L = _native.lib()
    for _ in range(a.warmup):
        step()
    fence()
    L.cs_profile(1)
    dt = timed(step, a.steps)
    tot_ms, launches = ctypes.c_double(), ctypes.c_int()
    _native.check(L.cs_profile_read(ctypes.byref(tot_ms), ctypes.byref(launches)))
    L.cs_profile(0)
    fallback_rows = sum(int(q.stats()[:, 10].sum()) for q in plans)
    tile_redo_rows = sum(int(q.stats()[:, 11].sum()) for q in plans)
    err_flags = sum(int(q.stats()[:, 9].sum()) for q in plans)

    split = None
    if world > 1 and not no_gather:  # kernels only / + all-gather / end to end, each timed like the headline
        dt_k = timed(compute_only, a.steps)
        dt_g = timed((lambda: step(expand=False)), a.steps) if job is not None else dt
        split = {"kernels_only_fps": frames * a.steps / dt_k, "kernels_plus_allgather_fps": frames * a.steps / dt_g,
                 "end_to_end_fps": frames * a.steps / dt,
                 "note": "end to end = kernels + all-gather + uint8->float32 expansion on every rank (device-resident inputs)"
                         if job is not None else "gpu_warp: float32 all-gather, no expansion step"}

    value_blur_off = None
    if world == 1 and a.config == "metric" and blur:
        plan_off = engine.Plan(params(nloc, False), device)
        plan_off.run(image, depth)
        value_blur_off = frames * a.steps / timed(lambda: plan_off.run(image, depth), a.steps)
        del plan_off

    value_other_depths = None
    if world == 1 and a.config == "metric" and not a.depth and not a.no_other_depths:
        # the same workload on depth maps without plateaus (a gain that only exists on the stepped depth's flat
        # regions must be visible as such): radial = the stepped map before quantisation, blobs = smooth random hills
        value_other_depths = {}
        import numpy as np
        import synth
        for kind in ("radial", "blobs", "scene8"):
            if kind == "radial":
                _, d2 = make_inputs(torch, dict(cfg, depth="radial"), nloc, b0, device)
            else:
                base = np.stack([synth.DEPTHS[kind](H, W, seed=b0 + i) for i in range(min(8, nloc))])
                reps = (nloc + base.shape[0] - 1) // base.shape[0]
                d2 = torch.from_numpy(np.tile(base, (reps, 1, 1))[:nloc]).to(device)[..., None].expand(nloc, H, W, 3).contiguous()
            plan.run(image, d2)
            value_other_depths[kind] = frames * a.steps / timed(lambda: plan.run(image, d2), a.steps)
            del d2
        plan.run(image, depth)

    # the same workload under dialect D64 -- what an install of the reference WITH numba computes (float64 disparity chain + numba's typing
    # of the polylines sweep; derived typing): in the point kernel since round 6.  Reported next to the metric, never as it.
    value_dialect_d64 = None
    if world == 1 and a.config == "metric" and not a.depth and not a.no_other_depths:
        try:
            engine.DIALECT = "D64"
            plan64 = engine.Plan(params(nloc), device)
            plan64.run(image, depth)
            value_dialect_d64 = frames * a.steps / timed(lambda: plan64.run(image, depth), a.steps)
            del plan64
        except Exception as ex:  # noqa: BLE001  (an extra, never a reason to lose the bench line)
            value_dialect_d64 = None
            print(f"bench.py: D64 leg skipped: {ex}", file=sys.stderr)
        finally:
            engine.DIALECT = "D32"

    if a.verify and world > 1 and no_gather:
        ref = engine.Plan(params(nloc), device).run(image, depth)[0]
        ok = torch.equal(plan.stereo, ref)
        print(f"[verify] rank {rank}: {'OK' if ok else 'MISMATCH'} (own block only: --gather none)", flush=True)
        if not ok:
            raise SystemExit(1)
    elif a.verify and world > 1:
        # every rank: its own block of the reassembled batch == the float32 output computed locally in one piece
        full = job.gathered if job is not None else gathered_f32[0]
        ref = engine.Plan(params(nloc), device).run(image, depth)[0]
        ok = torch.equal(full[b0:b1], ref)
        other = (rank + 1) % world  # and one foreign sub-batch: recompute it here
        no = min(batch_size, bounds[other + 1] - bounds[other]) if gpu_warp else 1
        fi, fd = make_inputs(torch, cfg, no, bounds[other], device)
        ok = ok and torch.equal(full[bounds[other]:bounds[other] + no], engine.Plan(params(no), device).run(fi, fd)[0])
        print(f"[verify] rank {rank}: {'OK' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            raise SystemExit(1)
    if world > 1:
        # (every rank's own prints are out before rank 0 writes the JSON line: the line is longer than a pipe's atomic write, and a
        # [verify] line of another rank landing inside it cost tests/test_gpu_sharded.py one run in ten sessions)
        sys.stdout.flush()
        dist.barrier()
    if rank == 0:
        fps = frames * a.steps / dt
        kern_ms = tot_ms.value / max(launches.value, 1)
        # launches of the dominant kernel per step: cs_generate cuts a batch into frame chunks (the pre-pass of chunk c + 1
        # runs under the warp of chunk c on a second stream), the sharded job into all-gather chunks
        launches_per_step = max(1, round(launches.value / max(a.steps, 1)))
        frames_per_launch = nloc / launches_per_step
        alg_bytes = frames_per_launch * cfg["bytes_px"] * H * W  # per launch of the dominant kernel on this rank
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{cfg['fill']}_{'4k' if H == 2160 else '1080p'}_blur{int(blur)}" if a.config != "metric" else f"polylines_soft_4k_blur{int(blur)}"
                if key in tj:
                    traffic = tj[key]["bytes_per_frame"] * frames_per_launch
                    traffic_source = f"profiles/pmc_traffic.json[{key}] (PMC passes of {tj[key].get('profile', 'the committed profile')}, not measured in this run)"
            except Exception:  # noqa: BLE001
                traffic = None
        # The dominant kernel's OWN algorithmic bytes: it reads the image (12 B/px) and one 4-byte depth value per eye (the
        # blurred maps of the two eyes; one shared gray map with the blur off) and writes every output; the 12 B/px RGB depth
        # input of the node is read by the gray / blur pre-pass, not by this kernel.  `frac_own_bytes` is quoted on these bytes;
        # `frac` (= `frac_node_bytes`) divides the whole node boundary's bytes by the same kernel time and
        # `pipeline_frac` by the whole step.
        # With the blur on, the tile kernels read the blurred maps only for the 64 x 32 tiles the blur touched (cs_profile_tiles:
        # their share of this run's batch); everywhere else both eyes read ONE shared gray value: 4 instead of 8 B/px.
        tile_frac = ctypes.c_double(-1.0)
        _native.check(L.cs_profile_tiles(ctypes.byref(tile_frac)))
        depth_px = 4.0 if not blur else (8.0 if tile_frac.value < 0 else 4.0 + 4.0 * tile_frac.value)
        own_px = cfg["bytes_px"] - 12 + depth_px
        own_bytes = frames_per_launch * own_px * H * W
        achieved_own = own_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        pipeline = frames * a.steps * cfg["bytes_px"] * H * W / dt / 1e9 / world
        # The vector-issue side: instruction counters and the issue floor of the dominant kernel from the
        # committed PMC passes (tools/make_valu.py -> profiles/pmc_valu.json; counters per wave do not depend on the box), set
        # against THIS run's kernel time.  `bound` stays the roof the contract prices (`peak` / `unit` are HBM figures);
        # `binding_roof` names the roof that actually binds: "valu" when the kernel's vector-issue floor exceeds its memory
        # floor (own bytes at the 6.29 TB/s copy ceiling of MI355X_MICROARCH.md).
        valu, binding = None, "hbm"
        vpath = os.path.join(ROOT, "profiles", "pmc_valu.json")
        if os.path.exists(vpath) and not a.depth:
            try:
                vj = json.load(open(vpath))
                key = f"{cfg['fill']}_{'4k' if H == 2160 else '1080p'}_blur{int(blur)}" if a.config != "metric" else f"polylines_soft_4k_blur{int(blur)}"
                if key in vj:
                    v = vj[key]
                    frames_profile = v.get("frames_per_dispatch") or frames_per_launch
                    floor_ms = v["issue_floor_us"] * 1e-3 * frames_per_launch / frames_profile if "issue_floor_us" in v else None
                    mem_floor_ms = own_bytes / 6.29e12 * 1e3
                    valu = {"valu_per_wave": v["valu_per_wave"], "salu_per_wave": v["salu_per_wave"], "lds_per_wave": v["lds_per_wave"],
                            "simd_busy": v["simd_busy"], "wait_any": v["wait_any"], "wait_inst": v["wait_inst"],
                            "mean_cycles_per_valu": v.get("mean_cycles_per_valu"), "issue_floor_ms": floor_ms,
                            "frac_of_issue_floor": (floor_ms / kern_ms) if (floor_ms and kern_ms > 0) else None,
                            "memory_floor_ms": mem_floor_ms, "kernel_ms_profile": v["kernel_us_profile"] * 1e-3 * frames_per_launch / frames_profile,
                            "source": f"profiles/pmc_valu.json[{key}] (PMC passes + ISA census of {v.get('profile', 'the committed profile')}: "
                                      "instruction counts per wave; the floor is priced at that profile's clock, not measured in this run)"}
                    if floor_ms and floor_ms > mem_floor_ms:
                        binding = "valu"
            except Exception:  # noqa: BLE001
                valu = None
        # `achieved` / `frac`: the CONTRACT formula -- SURVEY.md 8d's algorithmic bytes of the node boundary per launch over the
        # dominant kernel's measured time; the kernel's own bytes are `achieved_own_bytes` / `frac_own_bytes`
        roofline = {"bound": "hbm", "binding_roof": binding, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": cfg["kernel"], "kernel_ms": kern_ms, "launches": launches.value,
                    "launches_per_step": launches_per_step,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "algorithmic_bytes_note": f"{cfg['bytes_px']} B per source pixel at the float32 node boundary (SURVEY.md 8d) x the pixels of one launch; "
                                              f"the kernel itself reads and writes {own_px:.2f} B per source pixel once: image 12, depth "
                                              f"{depth_px:.2f} (one blurred value per eye in the {max(tile_frac.value, 0.0) * 100:.1f} % of the 64 x 32 tiles the "
                                              "blur touched, one shared gray value elsewhere; measured on this run's tile map), every output -- the 12 B/px "
                                              "RGB depth input is read by the gray / blur pre-pass",
                    "blurred_tile_fraction": tile_frac.value if tile_frac.value >= 0 else None,
                    "achieved_own_bytes": achieved_own, "frac_own_bytes": achieved_own / HBM_PEAK_GBS, "own_bytes_per_launch": own_bytes,
                    "frac_node_bytes": achieved / HBM_PEAK_GBS,
                    "pipeline_achieved": pipeline, "pipeline_frac": pipeline / HBM_PEAK_GBS, "valu": valu}
        line = {
            "metric": cfg["name"], "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg['what']}, depth blur {'on' if blur else 'off'} (widget defaults)",
                       "frames_total": frames, "frames_per_gpu": nloc, "sharding": "by frame, contiguous blocks",
                       "gather": a.gather if world > 1 else None,
                       "collective": "none" if world == 1 else (
                           "none: every rank keeps the float32 results of its own frame block (--gather none, sharding.generate_sharded"
                           "(gather=())); value = frames of all ranks / slowest rank's kernel time" if no_gather else
                           f"all_gather(stereoscope as uint8 codes) over RCCL in {job.n_chunks} chunk(s) overlapped with compute"
                           " + expand to float32 on every rank" if job is not None else "all_gather(stereoscope float32) over RCCL")},
            "roofline": roofline,
            "diagnostics": {"rows_redone_by_general_kernel": tile_redo_rows, "rows_replayed_sequentially": fallback_rows,
                            "kernel_error_flags": err_flags},
        }
        if value_blur_off is not None:
            line["value_blur_off"] = value_blur_off
        if value_other_depths is not None:
            line["value_other_depths"] = value_other_depths
        if value_dialect_d64 is not None:
            line["value_dialect_d64"] = value_dialect_d64
        if split is not None:
            line["split"] = split
        if ranks is not None:
            line["ranks"] = ranks
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, a.cpu_frames, blur)
        elif not a.no_cpu_baseline:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
