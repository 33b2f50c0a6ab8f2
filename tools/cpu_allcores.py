"""The CPU oracle on many host cores: frames in parallel, one single-threaded C process per frame -- the companion of
bench.py's single-thread `cpu_baseline` (its "all_cores" entry).  Never touches the GPU.  Test infrastructure (oracle/)."""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(job):
    """job = (frame index, h, w, depth kind, ui fill string, mode, divergence, blur) -> seconds of oracle time."""
    for p in (ROOT, os.path.join(ROOT, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import synth
    from oracle import node_oracle
    i, h, w, kind, ui, mode, div, blur = job
    img = synth.image_f32(1, h, w, seed=1 + i)
    depth = synth.depth_batch(kind, 1, h, w, channels=3)
    t0 = time.perf_counter()
    node_oracle.generate(img, depth, div, 0.0, mode, 0.0, 0.5, 2.0, ui, 20.0, 20.0, blur, depth_blur_falloff=2.0,
                         depth_blur_vert_smooth=6, batch_size=12)
    return time.perf_counter() - t0


def run(workers, frames, h=2160, w=3840, kind="stepped", ui="Fill - Polylines Soft", mode="left-right", div=8.0, blur=True):
    """-> (frames per second over the pool, mean seconds per frame per core).  Spawned workers: safe after the parent has
    initialised the GPU."""
    jobs = [(i, h, w, kind, ui, mode, div, blur) for i in range(frames)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        pool.map(one, [(0, 64, 64, kind, ui, mode, div, blur)] * workers, chunksize=1)  # start the workers, load the library
        t0 = time.perf_counter()
        per = pool.map(one, jobs, chunksize=1)
        dt = time.perf_counter() - t0
    return frames / dt, sum(per) / len(per), dt


if __name__ == "__main__":
    cores = os.cpu_count()
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else min(cores, 64)
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else workers
    fps, mean_s, dt = run(workers, frames)
    print(f"{frames} 4K frames (polylines_soft SBS, blur on) through the C oracle on {workers} of {cores} cores: {dt:.1f} s "
          f"-> {fps:.1f} frames/s (mean {mean_s:.2f} s per frame per core)")
