"""Host staging pipeline around the fused device path (reference GenerateStereo.py:159-177,
286-342 moves every sub-batch through pageable memory synchronously and assembles the result by list + cat + copy).

ComfyUI hands the node CPU tensors and expects CPU tensors back.  A 4K frame is 0.2 GB in and 0.7 GB out, so the node
is bound by PCIe and host memory, not by the kernels.  Measured on the MI355X box (tools/host_primitives.py): device ->
pinned 57 GB/s, pinned -> device 31-57 GB/s, both directions together 87 GB/s, CPU copy pageable -> pinned 49 GB/s,
pinned allocation 24 GB/s (cached by PyTorch afterwards) -- but first-touch page faults of a fresh pageable tensor
run at ~5 GB/s, which is what bounds `tensor.cpu()`.  So `generate_host`

  * allocates the four result tensors in pinned memory (they are ordinary CPU tensors to the caller; PyTorch returns
    them to its pinned cache when freed) and lets the device write every chunk straight into its slice of them -- no
    staging copy, no list + cat; if pinning fails (or `pinned_outputs=False`) results go through pinned staging buffers
    into pageable tensors instead;
  * cuts the batch into chunks and runs three stages side by side, double-buffered:
        CPU: pageable input chunk -> pinned staging          (torch's parallel CPU copy)
        HIP stream "h2d": pinned -> device input buffers
        current stream:   cs_generate on the chunk            (engine.Plan)
        HIP stream "d2h": device outputs -> host
    Events order the stages and protect every buffer that is reused two chunks later.

Round 3 -- the COMPACT boundary (CPU techniques, i.e. everything but gpu_warp): every output value is one of 256 codes
(stereoscope k/255; depth maps trunc(d*255) mod 256 over 255 on three equal channels; mask 0/1), so the device hands over
one byte per value (cs_params.flags bit 1 for the stereoscope, cs_pack_u8 for depth maps and mask: 83 MB instead of 697 MB
per 4K frame over PCIe) and the float32 result tensors are written by the host cores (cs_host_expand_u8: worker threads,
table of the true quotients k/255.0f) while the next chunk is staged and computed.  The results are ordinary pageable
tensors -- no pinned allocation on the critical path -- and bit-identical to the float32 path (tested).

Results are bit-identical to processing the whole batch at once (every quantity of the path is per frame; gpu_warp's
two 0..255 decisions are per reference sub-batch, so chunks are multiples of `batch_size` for that technique).
"""
import ctypes
import os
from concurrent.futures import ThreadPoolExecutor

import torch

from . import _native, engine

# target size of one chunk's outputs in bytes: large enough for full-rate PCIe transfers, small enough that the
# pinned staging (2 x in + 2 x out) stays a few GB
CHUNK_OUT_BYTES = 3 << 30
# compact boundary: a chunk is sized by its float32 INPUTS instead (0.2 GB per 4K frame; outputs are 83 MB)
CHUNK_IN_BYTES = 1 << 30


def _chunk_frames(total, per_frame_out, fill, batch_size):
    chunk = max(1, min(total, CHUNK_OUT_BYTES // max(per_frame_out, 1)))
    if fill == 'gpu_warp':  # keep the reference's sub-batch boundaries (its 0..255 tests are per sub-batch)
        sub = max(1, min(batch_size, total))
        chunk = max(sub, (chunk // sub) * sub)
    return chunk


class _Stage:
    """Buffers of one pipeline slot: pinned + device inputs, a Plan (device outputs + workspace), pinned outputs."""

    def __init__(self, p, depth_shape, device, staged_outputs, compact=False):
        n, h, w = p.n, p.h, p.w
        self.plan = engine.Plan(p, device, stereo_u8=compact)
        self.compact = compact
        self.future = None
        if compact:   # uint8 codes: device buffers for the packed depth maps / mask, pinned staging for all four outputs
            u8 = dict(dtype=torch.uint8, device=device)
            self.dev_codes = [self.plan.stereo, torch.empty((n, h, w), **u8), torch.empty((n, h, w), **u8),
                              torch.empty(tuple(self.plan.mask.shape), **u8)]
            self.pin_codes = [torch.empty(t.shape, dtype=torch.uint8, pin_memory=True) for t in self.dev_codes]
            staged_outputs = False
        self.pin_img = torch.empty((n, h, w, 3), dtype=torch.float32, pin_memory=True)
        self.pin_dep = torch.empty((n,) + tuple(depth_shape), dtype=torch.float32, pin_memory=True)
        self.dev_img = torch.empty((n, h, w, 3), dtype=torch.float32, device=device)
        self.dev_dep = torch.empty((n,) + tuple(depth_shape), dtype=torch.float32, device=device)
        self.pin_out = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                        for t in (self.plan.stereo, self.plan.depth_l, self.plan.depth_r, self.plan.mask)] if staged_outputs else None
        self.used = False
        self.e_in = torch.cuda.Event()    # inputs of the slot's current chunk are on the device
        self.e_done = torch.cuda.Event()  # its kernels have finished
        self.e_out = torch.cuda.Event()   # its outputs are in pinned memory
        self.range = None


def result_shapes(image_shape, modes, fill="polylines_soft"):
    """Shapes of the four results for an image batch [N,H,W,3]: (stereoscope, depth_left, depth_right, mask)."""
    n, h, w = int(image_shape[0]), int(image_shape[1]), int(image_shape[2])
    p = engine.make_params(1, h, w, h, w, 1, fill, modes, 1.0, 0.0, 0.0, 0.5, 1.0, False, 0.0, 0.0, 1.0, 0, 1)
    oh, ow, mh, mw = engine.output_shape(p)
    return (n, oh, ow, 3), (n, h, w, 3), (n, h, w, 3), (n, mh, mw)


def generate_host(image, depth_map, divergence, separation, modes, stereo_balance, convergence_point,
                  stereo_offset_exponent, fill, depth_blur_edge_threshold, depth_blur_strength, depth_map_blur,
                  depth_blur_falloff=1.0, depth_blur_vert_smooth=0, batch_size=4, device=None, progress=None,
                  pinned_outputs=True, out=None, compact=None, expand_threads=0):
    """CPU tensors in (image [N,H,W,3], depth_map [N,H',W',C], float32) -> four CPU float32 tensors, like
    StereoImageNode.generate returns them.  `progress(k)` is called with the number of frames finished.
    out: (stereoscope, depth_left, depth_right, mask) CPU float32 tensors of the result shapes (`result_shapes`) to write
    into: a caller that keeps PINNED result tensors across calls (`tensor.pin_memory()` once) takes the
    pinned allocation -- 24 GB/s of page faulting and locking, the bound of the default path -- off the critical path;
    pageable tensors work too (through the pinned staging buffers)."""
    if not torch.cuda.is_available():
        raise RuntimeError("comfystereo_amd needs an MI355X (PyTorch-ROCm `cuda` device); there is no CPU fallback")
    device = device or torch.device("cuda", torch.cuda.current_device())
    image = image.contiguous().float()
    depth_map = depth_map.contiguous().float()
    total, h, w, c = image.shape
    if c != 3:
        raise ValueError("image must be [N,H,W,3]")
    dshape = tuple(depth_map.shape[1:])

    def params(n):
        return engine.make_params(n, h, w, dshape[0], dshape[1], dshape[2], fill, modes, divergence, separation,
                                  stereo_balance, convergence_point, stereo_offset_exponent, depth_map_blur,
                                  depth_blur_strength, depth_blur_edge_threshold, depth_blur_falloff,
                                  depth_blur_vert_smooth, batch_size)

    oh, ow, mh, mw = engine.output_shape(params(1))
    per_frame_out = 4 * (oh * ow * 3 + 2 * h * w * 3 + mh * mw)
    if compact is None:
        compact = fill != 'gpu_warp'
    if compact and fill == 'gpu_warp':
        raise ValueError("gpu_warp colours are not k/255: no compact boundary")
    if compact:
        per_frame_in = 4 * (h * w * 3 + dshape[0] * dshape[1] * dshape[2])
        chunk = max(1, min(total, CHUNK_IN_BYTES // max(per_frame_in, 1)))
        if total >= 4:   # at least four chunks so that staging, transfers, kernels and the host expansion overlap
            chunk = min(chunk, (total + 3) // 4)
    else:
        chunk = _chunk_frames(total, per_frame_out, fill, batch_size)
    shapes = ((total, oh, ow, 3), (total, h, w, 3), (total, h, w, 3), (total, mh, mw))
    final = None
    if out is not None:
        out = tuple(out)
        if len(out) != 4:
            raise ValueError("out: (stereoscope, depth_left, depth_right, mask)")
        for t, sh in zip(out, shapes):
            if t.device.type != "cpu" or t.dtype != torch.float32 or tuple(t.shape) != sh or not t.is_contiguous():
                raise ValueError(f"out tensors must be contiguous CPU float32 tensors of shapes {shapes}")
        if all(t.is_pinned() for t in out):
            final = out
    if final is None and out is None and pinned_outputs:
        # (compact boundary too: the host threads then write into pinned blocks that PyTorch's host allocator caches -- no page
        # faults on the way in, and dropping 15 GB of results is not a 0.7 s munmap on the caller's side, profiles/r03_host.txt)
        try:
            final = tuple(torch.empty(sh, dtype=torch.float32, pin_memory=True) for sh in shapes)
        except RuntimeError:  # not enough lockable memory: fall back to pageable results through staging buffers
            final = None
    direct = final is not None and not compact   # (float32 boundary: the device writes straight into pinned results)
    if final is None:
        final = out if out is not None else tuple(torch.empty(sh, dtype=torch.float32) for sh in shapes)
    ranges = [(b0, min(b0 + chunk, total)) for b0 in range(0, total, chunk)]
    slots = [_Stage(params(chunk), dshape, device, not direct, compact) for _ in range(min(2, len(ranges)))]
    tail = None  # a shorter last chunk gets its own (smaller) slot
    if ranges[-1][1] - ranges[-1][0] != chunk:
        tail = _Stage(params(ranges[-1][1] - ranges[-1][0]), dshape, device, not direct, compact)
    L = _native.lib()
    nthreads = expand_threads if expand_threads > 0 else max(1, min(32, (os.cpu_count() or 1)))
    copy_threads = max(1, min(16, (os.cpu_count() or 1)))
    pool = ThreadPoolExecutor(max_workers=1) if compact else None   # (one job at a time: each job is multi-threaded itself)

    def expand_chunk(slot, b0, b1):
        """Worker thread: wait for the chunk's codes in pinned memory, write the float32 results (the GIL is released
        inside the native calls)."""
        slot.e_out.synchronize()
        n = b1 - b0
        for k, (rep, mode) in enumerate(((1, 0), (3, 0), (3, 0), (1, 1))):
            dst = final[k][b0:b1]
            count = dst.numel() // rep
            rc = L.cs_host_expand_u8(ctypes.c_void_p(slot.pin_codes[k].data_ptr()), ctypes.c_void_p(dst.data_ptr()), count, rep,
                                     mode, nthreads)
            if rc:
                raise RuntimeError(f"cs_host_expand_u8 failed ({rc})")
        return n
    s_h2d, s_d2h = torch.cuda.Stream(device), torch.cuda.Stream(device)
    s_main = torch.cuda.current_stream(device)

    def drain(slot):  # the slot's chunk is complete on the host
        if compact:
            done = slot.future.result()
            slot.future = None
            slot.range = None
            if progress:
                progress(done)
            return
        slot.e_out.synchronize()
        b0, b1 = slot.range
        if not direct:
            for dst, src in zip(final, slot.pin_out):
                dst[b0:b1].copy_(src[: b1 - b0])
        slot.range = None
        if progress:
            progress(b1 - b0)

    for i, (b0, b1) in enumerate(ranges):
        slot = tail if (tail is not None and i == len(ranges) - 1) else slots[i % len(slots)]
        if slot.range is not None:  # the slot still holds the chunk of two iterations ago
            if compact:             # its codes must have been expanded out of the pinned staging
                drain(slot)
            elif direct:            # its inputs must have left the pinned staging; the rest is ordered on the device
                slot.e_in.synchronize()
            else:
                drain(slot)
        n = b1 - b0
        # staging copy pageable -> pinned by the library's own threads (torch's CPU copy_ is bimodal on the MI355X boxes:
        # 90-170 GB/s or, every few calls, 5 GB/s; cs_host_copy: 130-160 GB/s every time -- profiles/r03_host.txt)
        for dst, src in ((slot.pin_img, image[b0:b1]), (slot.pin_dep, depth_map[b0:b1])):
            rc = L.cs_host_copy(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), src.numel() * 4, copy_threads)
            if rc:
                raise RuntimeError(f"cs_host_copy failed ({rc})")
        with torch.cuda.stream(s_h2d):
            if slot.used:
                s_h2d.wait_event(slot.e_done)  # the kernels that read the device inputs two chunks ago
            slot.dev_img.copy_(slot.pin_img, non_blocking=True)
            slot.dev_dep.copy_(slot.pin_dep, non_blocking=True)
            slot.e_in.record(s_h2d)
        s_main.wait_event(slot.e_in)
        if slot.used:
            s_main.wait_event(slot.e_out)      # the device outputs of two chunks ago have been read out
        slot.plan.run(slot.dev_img, slot.dev_dep)
        if compact:   # depth maps (one code per pixel) and mask as bytes; the stereoscope already is (flags bit 1)
            st = ctypes.c_void_p(s_main.cuda_stream)
            for src, dst, stride, mode in ((slot.plan.depth_l, slot.dev_codes[1], 3, 0), (slot.plan.depth_r, slot.dev_codes[2], 3, 0),
                                           (slot.plan.mask, slot.dev_codes[3], 1, 1)):
                _native.check(L.cs_pack_u8(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), dst.numel(), stride, mode, st))
        slot.e_done.record(s_main)
        with torch.cuda.stream(s_d2h):
            s_d2h.wait_event(slot.e_done)
            if compact:
                for src, dst in zip(slot.dev_codes, slot.pin_codes):
                    dst.copy_(src, non_blocking=True)
            else:
                outs = (slot.plan.stereo, slot.plan.depth_l, slot.plan.depth_r, slot.plan.mask)
                for k, src in enumerate(outs):
                    dst = final[k][b0:b1] if direct else slot.pin_out[k]
                    dst.copy_(src, non_blocking=True)
            slot.e_out.record(s_d2h)
        if compact:
            slot.future = pool.submit(expand_chunk, slot, b0, b1)
        if direct and slot.range is not None and progress:
            progress(slot.range[1] - slot.range[0])
        slot.range = (b0, b1)
        slot.used = True
    for slot in slots + ([tail] if tail is not None else []):
        if slot.range is not None:
            drain(slot)
    if pool is not None:
        pool.shutdown(wait=True)
    return final
