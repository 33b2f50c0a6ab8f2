"""Frame sharding of a video batch across the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Every quantity of the path is per frame (min/max normalisation per image, rows independent), so frames
shard with no halo and no exchange during compute; the only collective is the all-gather that
reassembles the stereoscope tensor (and, on request, the mask and depth maps) on every rank.
`gpu_warp` takes two decisions over a reference sub-batch (`batch_size` frames); shard boundaries are
therefore aligned to multiples of `batch_size` for that technique.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_frames, world_size, align=1):
    """Contiguous blocks frames[b[r]:b[r+1]] for rank r; every boundary is a multiple of `align`."""
    units = (n_frames + align - 1) // align
    base, extra = divmod(units, world_size)
    bounds = [0]
    for r in range(world_size):
        bounds.append(min(n_frames, bounds[-1] + (base + (1 if r < extra else 0)) * align))
    return bounds


def _host_staged(t, group=None):
    """gloo cannot all-gather device tensors: with that backend (CPU tests, several ranks sharing one GPU in the -m gpu
    tests) device tensors take a round trip through host memory.  RCCL ("nccl") gathers device memory directly."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_frames(local, bounds, group=None, method="collective"):
    """All-gather tensors whose dim 0 is this rank's frame block -> the full [N, ...] tensor on every rank.

    method "collective": one `all_gather_into_tensor` (RCCL picks ring / direct by message size).
    method "p2p": direct peer fan-out (SURVEY.md 8e's fallback) -- every rank sends its block to each peer and receives each
    peer's block straight into its slice of the result, all 2 (world - 1) transfers in one batch: on the fully connected
    xGMI node every pair has its own link, so a rank's block leaves over its 7 links at once instead of travelling a ring;
    blocks of different sizes need no padding."""
    if _host_staged(local, group):
        return all_gather_frames(local.cpu(), bounds, group, method).to(local.device)
    world = dist.get_world_size(group)
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    n = bounds[-1]
    if method == "p2p":
        rank = dist.get_rank(group)
        local = local.contiguous()
        out = torch.empty((n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        out[bounds[rank]:bounds[rank + 1]].copy_(local)
        ops = []
        for r in range(world):
            peer = dist.get_global_rank(group, r) if group is not None else r
            if r == rank:
                continue
            if sizes[rank]:
                ops.append(dist.P2POp(dist.isend, local, peer, group))
            if sizes[r]:
                ops.append(dist.P2POp(dist.irecv, out[bounds[r]:bounds[r + 1]], peer, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out
    if method != "collective":
        raise ValueError(f"unknown all-gather method {method!r}")
    if len(set(sizes)) == 1:
        out = torch.empty((n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def generate_sharded(run_local, image, depth_map, fill, batch_size, group=None, gather=("stereoscope", "mask"),
                     expand=None, method="collective"):
    """Run `run_local(image_block, depth_block) -> (stereo, depth_l, depth_r, mask)` on this rank's frame block
    and reassemble the requested outputs on every rank.  `image` / `depth_map` are the FULL batch (each rank
    slices its own block; nothing is sent before compute).

    `expand`: when `run_local` returns the stereoscope in its compact uint8 form (engine.Plan(stereo_u8=True):
    the CPU techniques' values are k/255 exactly), the shards cross xGMI as uint8 -- 4x fewer bytes -- and
    `expand` (engine.expand_u8) turns the gathered tensor into float32 on every rank."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = image.shape[0]
    align = min(batch_size, n) if fill == 'gpu_warp' else 1
    bounds = shard_bounds(n, world, align)
    b0, b1 = bounds[rank], bounds[rank + 1]
    names = ("stereoscope", "depth_left", "depth_right", "mask")
    if b1 > b0:
        local = run_local(image[b0:b1], depth_map[b0:b1])
    else:  # more ranks than (aligned) blocks: contribute an empty block of the right shape
        probe = run_local(image[:1], depth_map[:1])
        local = tuple(t[:0] for t in probe)
    out = {}
    for name, t in zip(names, local):
        out[name] = all_gather_frames(t, bounds, group, method) if name in gather else t
    if expand is not None and "stereoscope" in gather:
        out["stereoscope"] = expand(out["stereoscope"])
    return out, bounds


class ChunkedGather:
    """The all-gather of a rank's frame block, cut into `n_chunks` pieces so that it overlaps with the compute of
    the following pieces (the collective runs on the backend's own stream / thread: `async_op=True`).

    Per step:   for c in range(n_chunks):  g.launch(c, local_chunk_c)      # right after chunk c was produced
                for c in range(n_chunks):  g.finish(c, sink)               # sink(src_frames, first_frame_index)
    `finish` waits for chunk c only and hands every rank's piece of it to `sink` together with its position in the
    reassembled batch (rank r's frames are bounds[r] .. bounds[r+1], chunk c of them starts at bounds[r] + c * cf), so
    the consumer (bench.py: the uint8 -> float32 expansion) already works while later chunks are still on the wire.
    Requires equal blocks on all ranks and a block that divides into the chunks; `usable()` says whether that holds
    (fall back to `all_gather_frames` otherwise)."""

    @staticmethod
    def usable(n_frames, world, n_chunks):
        return n_frames % world == 0 and (n_frames // world) % n_chunks == 0 and n_frames >= world * n_chunks

    def __init__(self, n_frames, n_chunks, frame_shape, dtype, device, group=None, method="collective"):
        """method: "collective" (all_gather_into_tensor) or "p2p" (direct peer fan-out, see all_gather_frames)."""
        if method not in ("collective", "p2p"):
            raise ValueError(f"unknown all-gather method {method!r}")
        self.method = method
        self.group = group
        self.world = dist.get_world_size(group)
        assert self.usable(n_frames, self.world, n_chunks)
        self.bounds = shard_bounds(n_frames, self.world)
        self.n_chunks = n_chunks
        self.cf = n_frames // self.world // n_chunks  # frames per chunk
        self.staging = [torch.empty((self.world * self.cf,) + tuple(frame_shape), dtype=dtype, device=device)
                        for _ in range(n_chunks)]
        self.works = [None] * n_chunks
        # gloo + device tensors (several ranks on one GPU in the tests): the chunks cross the wire from pinned host copies
        self.host = torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"
        if self.host:
            self.h_local = [torch.empty((self.cf,) + tuple(frame_shape), dtype=dtype).pin_memory() for _ in range(n_chunks)]
            self.h_all = [torch.empty((self.world * self.cf,) + tuple(frame_shape), dtype=dtype).pin_memory()
                          for _ in range(n_chunks)]

    def chunk_range(self, c):
        """Local frame range of chunk c inside this rank's block."""
        return c * self.cf, (c + 1) * self.cf

    def _gather_async(self, dst, src):
        """Start the gather of `src` (this rank's chunk) into `dst` ([world * cf, ...]); returns the requests to wait for."""
        if self.method == "collective":
            return [dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)]
        rank = dist.get_rank(self.group)
        dst[rank * self.cf:(rank + 1) * self.cf].copy_(src)
        ops = []
        for r in range(self.world):
            if r == rank:
                continue
            peer = dist.get_global_rank(self.group, r) if self.group is not None else r
            ops.append(dist.P2POp(dist.isend, src, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, dst[r * self.cf:(r + 1) * self.cf], peer, self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def launch(self, c, local_chunk):
        assert local_chunk.shape[0] == self.cf and local_chunk.is_contiguous()
        if self.host:
            self.h_local[c].copy_(local_chunk, non_blocking=True)
            torch.cuda.current_stream().synchronize()  # the host copy must be complete before gloo reads it
            self.works[c] = self._gather_async(self.h_all[c], self.h_local[c])
            return
        self.works[c] = self._gather_async(self.staging[c], local_chunk)

    def finish(self, c, sink):
        for req in self.works[c]:
            req.wait()
        self.works[c] = None
        if self.host:
            self.staging[c].copy_(self.h_all[c], non_blocking=True)
        for r in range(self.world):
            sink(self.staging[c][r * self.cf:(r + 1) * self.cf], self.bounds[r] + c * self.cf)


class ShardedStereoJob:
    """One rank's part of a frame-sharded batch for the CPU techniques (what `bench.py --gpus N` times and the -m gpu
    sharding tests check): the rank's block is produced chunk by chunk with the stereoscope in its compact uint8 form
    (engine.Plan(stereo_u8=True): every value is k/255 exactly), chunk c is all-gathered (RCCL over xGMI; its own
    stream) while the chunks after it are computed, and each gathered chunk is expanded to float32 (cs_expand_u8) into
    the reassembled batch while later chunks are still on the wire.

    make_params(n_frames) -> engine params for a chunk of that many frames.  step(image_block, depth_block) returns the
    reassembled float32 stereoscope [N, out_h, out_w, 3] (the same tensor every step)."""

    def __init__(self, make_params, n_frames, out_shape, device, group=None, chunk_options=None, method="collective"):
        from . import engine
        self.engine = engine
        world = dist.get_world_size(group)
        if chunk_options is None:
            # Two ranks share ONE xGMI link: the half batch takes longer on the wire than its kernels save (DESIGN.md section 6:
            # 10.4 ms against 7.0 ms at 64 4K frames), so there is nothing for chunks to hide behind -- one exchange of the whole
            # block, without the per-chunk launches and waits.  From four ranks on the wire time per link falls below the
            # compute time and the chunked, overlapped form pays.
            chunk_options = (1,) if world == 2 else (4, 2, 1)
        self.n_chunks = next((k for k in chunk_options if ChunkedGather.usable(n_frames, world, k)), 0)
        if self.n_chunks == 0:
            raise ValueError(f"{n_frames} frames do not split evenly over {world} ranks")
        self.cg = ChunkedGather(n_frames, self.n_chunks, tuple(out_shape), torch.uint8, device, group, method)
        self.plans = [engine.Plan(make_params(self.cg.cf), device, stereo_u8=True) for _ in range(self.n_chunks)]
        self.gathered = torch.empty((n_frames,) + tuple(out_shape), dtype=torch.float32, device=device)
        self.bounds = self.cg.bounds

    def _sink(self, codes, first):
        self.engine.expand_u8(codes, self.gathered[first:first + codes.shape[0]])

    def compute(self, image_block, depth_block):
        """The kernels only (no collective): every chunk of this rank's block."""
        for c in range(self.n_chunks):
            lo, hi = self.cg.chunk_range(c)
            self.plans[c].run(image_block[lo:hi], depth_block[lo:hi])

    def step(self, image_block, depth_block, expand=True):
        for c in range(self.n_chunks):
            lo, hi = self.cg.chunk_range(c)
            stereo, _, _, _ = self.plans[c].run(image_block[lo:hi], depth_block[lo:hi])
            self.cg.launch(c, stereo)
        for c in range(self.n_chunks):
            self.cg.finish(c, self._sink if expand else (lambda codes, first: None))
        return self.gathered
