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


def all_gather_frames(local, bounds, group=None):
    """All-gather tensors whose dim 0 is this rank's frame block -> the full [N, ...] tensor on every rank."""
    world = dist.get_world_size(group)
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    n = bounds[-1]
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
                     expand=None):
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
        out[name] = all_gather_frames(t, bounds, group) if name in gather else t
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

    def __init__(self, n_frames, n_chunks, frame_shape, dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        assert self.usable(n_frames, self.world, n_chunks)
        self.bounds = shard_bounds(n_frames, self.world)
        self.n_chunks = n_chunks
        self.cf = n_frames // self.world // n_chunks  # frames per chunk
        self.staging = [torch.empty((self.world * self.cf,) + tuple(frame_shape), dtype=dtype, device=device)
                        for _ in range(n_chunks)]
        self.works = [None] * n_chunks

    def chunk_range(self, c):
        """Local frame range of chunk c inside this rank's block."""
        return c * self.cf, (c + 1) * self.cf

    def launch(self, c, local_chunk):
        assert local_chunk.shape[0] == self.cf and local_chunk.is_contiguous()
        self.works[c] = dist.all_gather_into_tensor(self.staging[c], local_chunk, group=self.group, async_op=True)

    def finish(self, c, sink):
        self.works[c].wait()
        self.works[c] = None
        for r in range(self.world):
            sink(self.staging[c][r * self.cf:(r + 1) * self.cf], self.bounds[r] + c * self.cf)
