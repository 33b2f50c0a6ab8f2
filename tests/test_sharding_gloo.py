"""Frame sharding + all-gather with 2 gloo processes on CPU (the N>1 path of bench.py / sharding.py).
The local compute is stood in for by the CPU oracle -- the thing under test is the partition + collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from comfystereo_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds():
    assert sharding.shard_bounds(64, 8) == [0, 8, 16, 24, 32, 40, 48, 56, 64]
    assert sharding.shard_bounds(10, 4) == [0, 3, 6, 8, 10]
    assert sharding.shard_bounds(5, 8) == [0, 1, 2, 3, 4, 5, 5, 5, 5]
    b = sharding.shard_bounds(256, 8, align=12)  # gpu_warp: boundaries on reference sub-batch boundaries
    assert b[0] == 0 and b[-1] == 256 and all(x % 12 == 0 for x in b[:-1]) and b == sorted(b)
    assert sharding.shard_bounds(7, 2, align=12) == [0, 7, 7]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fill, n, q, method="collective"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from oracle import node_oracle
    h, w = 16, 48
    img = synth.image_f32(n, h, w, seed=2)
    dep = synth.depth_batch("stepped", n, h, w, channels=3)
    ui = {v: k for k, v in node_oracle.FILL_KEYS.items()}[fill]
    args = (6.0, 0.0, "left-right", 0.0, 0.5, 2.0, ui, 20.0, 20.0, False)

    def run_local(ib, db):
        return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in
                     node_oracle.generate(ib.numpy(), db.numpy(), *args, batch_size=2))

    expand = None
    if fill != "gpu_warp":  # the compact path bench.py uses for N > 1: uint8 codes over the wire, expanded afterwards
        inner = run_local

        def run_local(ib, db):  # noqa: F811
            s, dl, dr, m = inner(ib, db)
            k = torch.round(s * 255.0).to(torch.uint8)
            assert torch.equal(k.float() / 255.0, s)
            return k, dl, dr, m

        def expand(k):
            return k.to(torch.float32) / 255.0

    out, bounds = sharding.generate_sharded(run_local, torch.from_numpy(img), torch.from_numpy(dep), fill, 2, expand=expand,
                                            method=method)
    full = node_oracle.generate(img, dep, *args, batch_size=2)
    ok = np.array_equal(out["stereoscope"].numpy(), full[0]) and np.array_equal(out["mask"].numpy(), full[3])
    ok = ok and out["depth_left"].shape[0] == bounds[rank + 1] - bounds[rank]
    q.put((rank, bool(ok), bounds))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fill,n,method", [("polylines_soft", 5, "collective"), ("gpu_warp", 6, "collective"), ("none", 1, "collective"),
                                           ("polylines_soft", 5, "p2p"), ("none", 1, "p2p")])
def test_two_rank_gather_equals_unsharded(fill, n, method):
    """both forms of the all-gather: the collective and the direct peer fan-out (unequal blocks, an empty block)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fill, n, q, method)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def _chunk_worker(rank, world, port, n, chunks, q, method="collective"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape = (3, 5)
    full = torch.arange(n * 15, dtype=torch.int32).reshape(n, *shape).to(torch.uint8)  # frame f is recognisable
    g = sharding.ChunkedGather(n, chunks, shape, torch.uint8, "cpu", method=method)
    b0 = g.bounds[rank]
    out = torch.zeros((n,) + shape, dtype=torch.float32)
    for step in range(2):  # the staging buffers are reused step after step
        out.zero_()
        for c in range(chunks):
            lo, hi = g.chunk_range(c)
            g.launch(c, full[b0 + lo:b0 + hi].contiguous())

        def sink(src, first):
            out[first:first + src.shape[0]] = src.to(torch.float32)

        for c in range(chunks):
            g.finish(c, sink)
        ok = torch.equal(out, full.to(torch.float32))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,chunks,method", [(8, 2, "collective"), (12, 3, "collective"), (4, 1, "collective"), (12, 3, "p2p")])
def test_chunked_gather_reassembles_in_frame_order(n, chunks, method):
    """sharding.ChunkedGather (bench.py's N > 1 step): chunk c of every rank lands at bounds[r] + c * cf."""
    assert sharding.ChunkedGather.usable(64, 8, 4) and not sharding.ChunkedGather.usable(10, 4, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, n, chunks, q, method)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
