"""The CPU oracle on ALL host cores (frames in parallel, one single-threaded C process per frame) -- the companion of
bench.py's single-thread `cpu_baseline`, quoted in DESIGN.md.  Never touches the GPU.  Test infrastructure (oracle/)."""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
H, W = 2160, 3840


def one(i):
    import synth
    from oracle import node_oracle
    img = synth.image_f32(1, H, W, seed=1 + i)
    depth = synth.depth_batch("stepped", 1, H, W, channels=3)
    t0 = time.perf_counter()
    node_oracle.generate(img, depth, 8.0, 0.0, "left-right", 0.0, 0.5, 2.0, "Fill - Polylines Soft", 20.0, 20.0, True,
                         depth_blur_falloff=2.0, depth_blur_vert_smooth=6, batch_size=12)
    return time.perf_counter() - t0


if __name__ == "__main__":
    cores = os.cpu_count()
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else cores
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * workers
    one(0)  # build / warm the oracle library once
    t0 = time.perf_counter()
    with mp.Pool(workers) as pool:
        per = pool.map(one, range(frames), chunksize=1)
    dt = time.perf_counter() - t0
    print(f"{frames} 4K frames (polylines_soft SBS, blur on) through the C oracle on {workers} of {cores} cores: {dt:.1f} s "
          f"-> {frames / dt:.1f} frames/s (mean {sum(per) / len(per):.2f} s per frame per core)")
