"""The frame-sharded (N > 1) leg with the REAL HIP kernels: two fresh processes share the one GPU of the test box (gloo
as the transport -- RCCL refuses two ranks on one device -- everything else is the code path `bench.py --gpus N` runs:
engine.Plan(stereo_u8=True) per chunk -> sharding.ChunkedGather -> cs_expand_u8), compared frame by frame with the CPU
oracle on the whole batch.  -m gpu.

BASELINE.json configs 4 and 5 are sharded workloads; the reference's sub-batch decisions the sharding must respect are
GenerateStereo.py:119-128 and stereoimage_generation.py:1045 / :315 (0..255 tests over a sub-batch)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(fill, n, h, w):
    import synth
    img = synth.image_f32(n, h, w, seed=12)
    dep = synth.depth_batch("blobs" if fill != "none" else "stepped", n, h, w, channels=3)
    if fill == "gpu_warp":
        # depth maxima straddle 1.0 across the reference's sub-batches (batch_size = 2): sub-batches 0 and 2 take the
        # `amax <= 1 -> x255` branch (:1045), sub-batch 1 does not (one of ITS frames exceeds 1.0, the other does not --
        # the decision is per sub-batch, not per frame, and a shard boundary must not split it)
        dep[2] = dep[2] * np.float32(1.5)
        dep[3] = dep[3] * np.float32(0.9)
    return img, dep


def _worker(rank, world, port, fill, mode, n, h, w, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from comfystereo_amd import engine, sharding
        from oracle import node_oracle
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        img, dep = _inputs(fill, n, h, w)
        ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}[fill]
        blur = fill != "none"
        batch = 2
        want = node_oracle.generate(img, dep, 6.0, 0.2, mode, 0.1, 0.5, 2.0, ui, 20.0, 20.0, blur, depth_blur_falloff=2.0,
                                    depth_blur_vert_smooth=3, batch_size=batch)
        timg, tdep = torch.from_numpy(img).to(dev), torch.from_numpy(dep).to(dev)

        def params(k):
            return engine.make_params(k, h, w, h, w, 3, fill, mode, 6.0, 0.2, 0.1, 0.5, 2.0, blur, 20.0, 20.0, 2.0, 3, batch)

        ok, note = True, ""
        if fill == "gpu_warp":
            out, bounds = sharding.generate_sharded(lambda ib, db: engine.Plan(params(ib.shape[0]), dev).run(ib, db), timg, tdep,
                                                    fill, batch, gather=("stereoscope", "mask"))
            assert all(b % batch == 0 for b in bounds[:-1]), bounds
            got, gm = out["stereoscope"].cpu().numpy(), out["mask"].cpu().numpy()
            err = float(np.abs(got - want[0]).max())
            ok = err <= 1e-4 and np.array_equal(gm, want[3])
            note = f"colour err {err:.2e}"
            # own block of the depth maps (not gathered)
            b0, b1 = bounds[rank], bounds[rank + 1]
            ok = ok and np.allclose(out["depth_left"].cpu().numpy(), want[1][b0:b1], atol=1e-6)
        else:
            oh, ow = engine.output_shape(params(1))[:2]
            job = sharding.ShardedStereoJob(params, n, (oh, ow, 3), dev, chunk_options=(2, 1))
            b0, b1 = job.bounds[rank], job.bounds[rank + 1]
            for _ in range(2):  # the staging buffers and plans are reused step after step
                full = job.step(timg[b0:b1], tdep[b0:b1])
            torch.cuda.synchronize()
            got = full.cpu().numpy()
            bad = [f for f in range(n) if not np.array_equal(got[f], want[0][f])]
            ok = not bad
            note = f"chunks {job.n_chunks}, mismatching frames {bad}"
            # mask / depth maps of the rank's own chunks come from the same plans: check the last chunk's
            lo, hi = job.cg.chunk_range(job.n_chunks - 1)
            p = job.plans[-1]
            ok = ok and np.array_equal(p.mask.cpu().numpy(), want[3][b0 + lo:b0 + hi])
            ok = ok and np.array_equal(p.depth_l.cpu().numpy(), want[1][b0 + lo:b0 + hi])
        q.put((rank, bool(ok), note))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 - reported to the parent
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:] + repr(e)))


@pytest.mark.parametrize("fill,mode,n", [("polylines_soft", "left-right", 8), ("none", "red-cyan-anaglyph", 4),
                                         ("hybrid_edge", "top-bottom", 4), ("gpu_warp", "left-right", 6)])
def test_two_ranks_with_the_hip_kernels_equal_the_oracle_on_the_whole_batch(fill, mode, n):
    ctx = mp.get_context("spawn")  # fresh processes: each initialises the GPU itself
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fill, mode, n, 40, 328, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(ok for _, ok, _ in res), res


@pytest.mark.parametrize("config,frames,gather", [("metric", 8, "collective"), ("cfg5", 4, "collective"), ("cfg4", 48, "collective"),
                                                  ("metric", 8, "p2p"), ("cfg4", 48, "p2p"), ("metric", 8, "none"), ("cfg4", 48, "none")])
def test_bench_two_ranks_verify(config, frames, gather):
    """`bench.py --gpus 2 --verify` (the N > 1 step the driver times) as two torchrun ranks sharing the GPU: every rank's
    block of the reassembled batch and one foreign sub-batch equal a local float32 run; the line carries the three-way
    split of BASELINE.md section 4."""
    import json
    env = dict(os.environ, CS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--config", config, "--frames", str(frames), "--verify", "--no-cpu-baseline", "--gather", gather]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("[verify] rank") == 2 and "MISMATCH" not in r.stdout
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["frames_total"] == frames
    if gather == "none":   # every rank keeps its own float32 block: no exchange step, the value IS the kernels-only rate
        assert "split" not in line and line["config"]["collective"].startswith("none")
    else:
        assert set(line["split"]) >= {"kernels_only_fps", "kernels_plus_allgather_fps", "end_to_end_fps"}
    assert line["diagnostics"]["kernel_error_flags"] == 0
    assert line["ranks"]["world_size"] == 2 and line["ranks"]["all_reduce_of_ones"] == 2 and len(line["ranks"]["devices"]) == 2
    assert line["config"]["gather"] == gather


def _visible_gpus():
    import torch
    return torch.cuda.device_count()   # (counting devices does not initialise the GPU: the launcher below starts fresh processes)


@pytest.mark.parametrize("config,frames,gather", [("metric", 8, "collective"), ("cfg4", 48, "collective"), ("cfg5", 4, "collective"),
                                                  ("metric", 8, "p2p"), ("cfg4", 48, "p2p"), ("metric", 8, "none")])
def test_bench_two_gpus_over_rccl_verify(config, frames, gather):
    """the same step over RCCL on TWO DEVICES -- the device `all_gather_into_tensor(async_op=True)` of
    sharding.ChunkedGather, the `batch_isend_irecv` fan-out, the single exchange at N = 2 and the `ranks` report have only ever run
    through gloo on one GPU.  Skips on a one-GPU box (the pool's); on the first multi-GPU box it runs with no code change: torchrun
    starts fresh processes (rendezvous on 127.0.0.1), backend nccl, one rank per device, --verify compares every rank's block of the
    reassembled batch and one foreign sub-batch with a local float32 run."""
    if _visible_gpus() < 2:
        pytest.skip("needs two visible GPUs (RCCL refuses two ranks on one device)")
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CS_BENCH_BACKEND", None)   # -> nccl
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--config", config, "--frames", str(frames), "--verify", "--no-cpu-baseline", "--gather", gather]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("[verify] rank") == 2 and "MISMATCH" not in r.stdout
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["frames_total"] == frames
    assert line["ranks"]["backend"] == "nccl" and line["ranks"]["world_size"] == 2 and line["ranks"]["all_reduce_of_ones"] == 2
    assert len({d["pci"] or d["local_rank"] for d in line["ranks"]["devices"]}) == 2   # two different devices
    assert line["diagnostics"]["kernel_error_flags"] == 0 and line["config"]["gather"] == gather
