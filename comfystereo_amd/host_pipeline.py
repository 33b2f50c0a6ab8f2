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
import threading
from concurrent.futures import ThreadPoolExecutor

import torch

from . import _native, engine

# target size of one chunk's outputs in bytes: large enough for full-rate PCIe transfers, small enough that the
# pinned staging (2 x in + 2 x out) stays a few GB
CHUNK_OUT_BYTES = 3 << 30
# compact boundary: a chunk is sized by its float32 INPUTS instead (0.2 GB per 4K frame; outputs are 83 MB)
CHUNK_IN_BYTES = 1 << 30
# Cap on the page-locked host memory the pipeline allocates for itself (staging buffers + result tensors).  A call whose
# results fit beside its staging buffers gets PINNED result tensors (the device / the expansion threads write straight into
# blocks PyTorch's host allocator caches); a larger call -- 32 x 4K frames are 15 GB of float32 results -- gets ordinary
# pageable result tensors like the reference's, and only the staging buffers (a few GB) are pinned.  The node shares its host
# with diffusion models and their offload buffers: the default keeps it below 8 GB whatever the batch.  A deployment with
# RAM to spare may raise it (steady state +5 % at 32 x 4K, profiles/r05_host.txt).
PINNED_POOL_BYTES = 8 << 30
_pinned_lock = threading.Lock()   # generate_host and the prewarm_async daemon thread both pass through _pinned_budget
_pinned_needs = set()             # distinct pinned needs (bytes) served since the cache was last released (module state only)
_warned_no_empty_cache = False


def _release_pinned_cache():
    """Give the pinned blocks PyTorch's caching host allocator holds (and nobody uses) back to the system.  The hook is a private
    one (torch._C._host_emptyCache); a PyTorch without it keeps its cache -- said once, not silently."""
    global _warned_no_empty_cache
    fn = getattr(torch._C, "_host_emptyCache", None)
    if fn is not None:
        fn()
    elif not _warned_no_empty_cache:
        _warned_no_empty_cache = True
        import warnings
        warnings.warn("comfystereo_amd: this PyTorch has no torch._C._host_emptyCache -- page-locked blocks of earlier shapes stay "
                      "cached beyond host_pipeline.PINNED_POOL_BYTES until the process ends")


def _pinned_budget(need):
    """Called once per generate_host / prewarm with the pinned bytes the call will ask for.  PyTorch never returns cached pinned
    blocks by itself, so blocks of shapes no longer in use would stay page-locked for nothing -- but releasing on every change of
    shape makes a caller that ALTERNATES between two shapes re-pin gigabytes per call at 24 GB/s.  Hysteresis: the
    distinct needs served since the last release bound what the cache can hold; it is released only when that bound, with this
    call's need, exceeds PINNED_POOL_BYTES."""
    with _pinned_lock:
        if need in _pinned_needs:
            return
        if sum(_pinned_needs) + need > PINNED_POOL_BYTES and _pinned_needs:
            _release_pinned_cache()
            _pinned_needs.clear()
        _pinned_needs.add(need)


def _plan_pinned(total, chunk, unit, per_frame_in, small_per_frame, f32_per_frame, per_frame_out, want_pinned_results):
    """(chunk, staging bytes, pin the results?) of a call -- ONE statement of it for generate_host and prewarm (they
    disagreed, and the float32 routes' pinned staging was not counted).  Staging = what the slots page-lock: inputs, compact landing
    buffers and -- when the RESULTS are not pinned -- a staging copy of each float32-route output (gpu_warp's colours: `pin_out`).
    Results are pinned while staging + results fit under PINNED_POOL_BYTES; otherwise the chunk shrinks (in multiples of `unit`, the
    gpu_warp sub-batch) until the staging alone does."""
    def slot_frames(ch):
        nranges = (total + ch - 1) // ch
        return ch * min(2, nranges) + (total % ch if total % ch and nranges > 1 else 0)
    base = per_frame_in + small_per_frame
    if want_pinned_results and slot_frames(chunk) * base + total * per_frame_out <= PINNED_POOL_BYTES:
        return chunk, slot_frames(chunk) * base, True
    per = base + (f32_per_frame if want_pinned_results is not None else 0)   # (None: caller-provided PINNED results, no staging copy)
    while chunk > unit and slot_frames(chunk) * per > PINNED_POOL_BYTES:
        chunk = max(unit, (chunk // 2) // unit * unit)
    return chunk, slot_frames(chunk) * per, False


def _chunk_frames(total, per_frame_out, fill, batch_size):
    chunk = max(1, min(total, CHUNK_OUT_BYTES // max(per_frame_out, 1)))
    if fill == 'gpu_warp':  # keep the reference's sub-batch boundaries (its 0..255 tests are per sub-batch)
        sub = max(1, min(batch_size, total))
        chunk = max(sub, (chunk // sub) * sub)
    return chunk


# how an output leaves the device: "u8" = one uint8 code per value (cs_pack_u8, or the kernels' own uint8 stereoscope), expanded
# by host threads (cs_host_expand_u8); "f1" = one float32 channel of three equal ones (cs_take_f32), replicated by host
# threads (cs_host_replicate_f32); "f32" = the float32 tensor itself, copied straight into the pinned result
ROUTES = {
    "compact": ("u8", "u8", "u8", "u8"),      # the CPU techniques: every value is one of 256 codes
    "warp": ("f32", "f1", "f1", "u8"),        # gpu_warp: genuine float colours; depth maps have three equal channels, the mask is a flag
    "float": ("f32", "f32", "f32", "f32"),    # round 2's boundary (comparison only)
}


class _Stage:
    """Buffers of one pipeline slot: pinned + device inputs, a Plan (device outputs + workspace), pinned outputs."""

    def __init__(self, p, depth_shape, device, routes, staged_f32):
        n, h, w = p.n, p.h, p.w
        self.plan = engine.Plan(p, device, stereo_u8=routes[0] == "u8")
        self.future = None
        outs = (self.plan.stereo, self.plan.depth_l, self.plan.depth_r, self.plan.mask)
        self.dev_small = [None] * 4   # the compact device form of an output (u8 codes / one float channel)
        self.pin_small = [None] * 4   # ... and its pinned landing buffer
        self.pin_out = [None] * 4     # pinned staging of a float32 output when the result tensor itself is not pinned
        for k, r in enumerate(routes):
            if r == "u8":
                shape = tuple(outs[k].shape) if k in (0, 3) else (n, h, w)
                self.dev_small[k] = outs[0] if k == 0 else torch.empty(shape, dtype=torch.uint8, device=device)
                self.pin_small[k] = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
            elif r == "f1":
                self.dev_small[k] = torch.empty((n, h, w), dtype=torch.float32, device=device)
                self.pin_small[k] = torch.empty((n, h, w), dtype=torch.float32, pin_memory=True)
            elif staged_f32:
                self.pin_out[k] = torch.empty(outs[k].shape, dtype=outs[k].dtype, pin_memory=True)
        self.pin_img = torch.empty((n, h, w, 3), dtype=torch.float32, pin_memory=True)
        self.pin_dep = torch.empty((n,) + tuple(depth_shape), dtype=torch.float32, pin_memory=True)
        self.dev_img = torch.empty((n, h, w, 3), dtype=torch.float32, device=device)
        self.dev_dep = torch.empty((n,) + tuple(depth_shape), dtype=torch.float32, device=device)
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


class _Lazy:
    """A value built on a helper thread (the second pipeline slot: its page-locked buffers are not needed before the second chunk)."""

    def __init__(self, fn, device):
        self.value, self.error = None, None

        def run():
            try:
                torch.cuda.set_device(device)
                self.value = fn()
            except BaseException as e:   # (re-raised by get() on the caller's thread)
                self.error = e
        self.thread = threading.Thread(target=run, name="comfystereo-slot", daemon=True)
        self.thread.start()

    def get(self):
        self.thread.join()
        if self.error is not None:
            raise self.error
        return self.value


def _small_bytes_per_frame(routes, oh, ow, h, w, mh, mw):
    """Bytes per frame of the compact output forms that travel through pinned landing buffers."""
    sizes = (oh * ow * 3, h * w, h * w, mh * mw)   # values per frame in compact form (depth maps: one channel)
    total = 0
    for k, r in enumerate(routes):
        if r == "u8":
            total += sizes[k]
        elif r == "f1":
            total += 4 * sizes[k]
    return total


class _Results:
    """The four result tensors, allocated by a helper thread while the first chunk is being staged and computed: page-locking
    15 GB of results (32 4K frames) takes 0.6 s at the 24 GB/s a pinned allocation runs at, three times the whole pipeline --
    nothing needs the tensors before the first chunk's codes are back on the host.  PyTorch's pinned-memory cache serves the
    request at once when a previous call's results have been released (or `prewarm` ran)."""

    def __init__(self, shapes, pinned):
        self.shapes, self.pinned, self.tensors, self.error = shapes, pinned, None, None
        self.thread = threading.Thread(target=self._run, name="comfystereo-results", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            if self.pinned:
                try:
                    self.tensors = tuple(torch.empty(sh, dtype=torch.float32, pin_memory=True) for sh in self.shapes)
                    return
                except RuntimeError:   # not enough lockable memory: pageable results
                    self.pinned = False
            self.tensors = tuple(torch.empty(sh, dtype=torch.float32) for sh in self.shapes)
        except BaseException as e:   # (re-raised by get() on the caller's thread)
            self.error = e

    def get(self):
        self.thread.join()
        if self.error is not None:
            raise self.error
        return self.tensors


def generate_host(image, depth_map, divergence, separation, modes, stereo_balance, convergence_point,
                  stereo_offset_exponent, fill, depth_blur_edge_threshold, depth_blur_strength, depth_map_blur,
                  depth_blur_falloff=1.0, depth_blur_vert_smooth=0, batch_size=4, device=None, progress=None,
                  pinned_outputs=True, out=None, compact=None, expand_threads=0):
    """CPU tensors in (image [N,H,W,3], depth_map [N,H',W',C], float32) -> four CPU float32 tensors, like
    StereoImageNode.generate returns them.  `progress(k)` is called with the number of frames finished.
    out: (stereoscope, depth_left, depth_right, mask) CPU float32 tensors of the result shapes (`result_shapes`) to write
    into; pageable tensors work too.  compact=False: the float32 boundary of round 2 (comparison)."""
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
    kind = "compact" if compact else ("warp" if fill == 'gpu_warp' else "float")
    routes = ROUTES[kind]
    per_frame_in = 4 * (h * w * 3 + dshape[0] * dshape[1] * dshape[2])
    unit = 1
    if kind == "compact":
        chunk = max(1, min(total, CHUNK_IN_BYTES // max(per_frame_in, 1)))
        if total >= 4:   # at least four chunks so that staging, transfers, kernels and the host expansion overlap
            chunk = min(chunk, (total + 3) // 4)
    else:
        chunk = _chunk_frames(total, per_frame_out, fill, batch_size)
        if fill == 'gpu_warp':
            unit = max(1, min(batch_size, total))
    shapes = ((total, oh, ow, 3), (total, h, w, 3), (total, h, w, 3), (total, mh, mw))
    # page-locked memory this call asks for (_plan_pinned): the staging buffers of its slots always (asynchronous copies need them)
    # -- including, when the results are pageable, the staging copies of the float32-route outputs --, the result tensors only
    # while everything stays under PINNED_POOL_BYTES
    f32_per_frame = sum(4 * sz for sz, r in zip((oh * ow * 3, h * w * 3, h * w * 3, mh * mw), routes) if r == "f32")
    if out is not None:
        out = tuple(out)
        if len(out) != 4:
            raise ValueError("out: (stereoscope, depth_left, depth_right, mask)")
        for t, sh in zip(out, shapes):
            if t.device.type != "cpu" or t.dtype != torch.float32 or tuple(t.shape) != sh or not t.is_contiguous():
                raise ValueError(f"out tensors must be contiguous CPU float32 tensors of shapes {shapes}")
    if out is None:
        want = bool(pinned_outputs)
    else:   # caller-provided results: pinned ones are written directly, pageable ones through the slots' staging copies
        want = None if all(t.is_pinned() for t, r in zip(tuple(out), routes) if r == "f32") else False
    chunk, staging_bytes, pinned_outputs = _plan_pinned(total, chunk, unit, per_frame_in,
                                                        _small_bytes_per_frame(routes, oh, ow, h, w, mh, mw), f32_per_frame,
                                                        per_frame_out, want)
    ranges = [(b0, min(b0 + chunk, total)) for b0 in range(0, total, chunk)]
    _pinned_budget(staging_bytes + (total * per_frame_out if pinned_outputs else 0))
    results = None   # helper thread that allocates the result tensors
    final = None
    if out is not None:
        final = out
    else:
        # (pinned for the compact boundary too: the host threads then write into blocks that PyTorch's host allocator caches -- no
        # page faults on the way in, and dropping 15 GB of results is not a 0.7 s munmap on the caller's side, profiles/r03_host.txt)
        results = _Results(shapes, pinned_outputs)
    has_f32 = "f32" in routes
    # float32 routes: the device writes straight into a pinned result tensor; otherwise through a pinned staging buffer.  Known
    # now for caller-provided tensors, for our own allocation when it has finished (staging buffers are allocated to be safe)
    direct_known = final is not None
    direct = direct_known and all(final[k].is_pinned() for k, r in enumerate(routes) if r == "f32")
    staged_f32 = has_f32 and direct_known and not direct
    L = _native.lib()
    nthreads = expand_threads if expand_threads > 0 else max(1, min(32, (os.cpu_count() or 1)))
    copy_threads = max(1, min(16, (os.cpu_count() or 1)))
    host_jobs = any(r != "f32" for r in routes)
    pool = ThreadPoolExecutor(max_workers=1) if host_jobs else None   # (one job at a time: each job is multi-threaded itself)
    state = {"final": final}

    def get_final():
        if state["final"] is None:
            state["final"] = results.get()
        return state["final"]

    def expand_chunk(slot, b0, b1):
        """Worker thread: wait for the chunk's compact outputs in pinned memory, write the float32 results (the GIL is
        released inside the native calls)."""
        fin = get_final()
        slot.e_out.synchronize()
        for k, r in enumerate(routes):
            if r == "f32":
                # pageable results: the chunk's float32 output leaves its pinned staging copy here, on the library's copy threads and
                # under the next chunk's transfers (round 6: it was a single-threaded torch copy_ on the caller's thread in drain() --
                # gpu_warp, the node's default technique, ran at 17 frames/s at 4K with the default pinned cap)
                if not state["direct"]:
                    dst, src = fin[k][b0:b1], slot.pin_out[k][: b1 - b0]
                    rc = L.cs_host_copy(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), src.numel() * 4, copy_threads)
                    if rc:
                        raise RuntimeError(f"cs_host_copy of output {k} failed ({rc})")
                continue
            dst = fin[k][b0:b1]
            rep = 1 if k in (0, 3) else 3
            count = dst.numel() // rep
            src = ctypes.c_void_p(slot.pin_small[k].data_ptr())
            if r == "u8":
                rc = L.cs_host_expand_u8(src, ctypes.c_void_p(dst.data_ptr()), count, rep, 1 if k == 3 else 0, nthreads)
            else:
                rc = L.cs_host_replicate_f32(src, ctypes.c_void_p(dst.data_ptr()), count, rep, nthreads)
            if rc:
                raise RuntimeError(f"host expansion of output {k} failed ({rc})")
        return b1 - b0

    s_h2d, s_d2h = torch.cuda.Stream(device), torch.cuda.Stream(device)
    s_main = torch.cuda.current_stream(device)
    slots = []

    def drain(slot):  # the slot's chunk is complete on the host
        b0, b1 = slot.range
        if slot.future is not None:
            slot.future.result()
            slot.future = None
        slot.e_out.synchronize()
        if has_f32 and not state["direct"] and not host_jobs:   # (no worker: the float32 boundary of round 2, comparison only)
            fin = get_final()
            for k, r in enumerate(routes):
                if r == "f32":
                    fin[k][b0:b1].copy_(slot.pin_out[k][: b1 - b0])
        slot.range = None
        if progress:
            progress(b1 - b0)

    state["direct"] = direct
    pending = {}   # slots being built by helper threads
    try:
        # the first slot now, the second one (and a shorter last chunk's own, smaller slot) on a helper thread: page-locking a slot's
        # buffers (2 GB for eight 4K frames) takes as long as the first chunk's staging + transfer + kernels, which do not need them
        nring = min(2, len(ranges))
        slots = [_Stage(params(chunk), dshape, device, routes, staged_f32)]
        tail_n = ranges[-1][1] - ranges[-1][0]
        if nring > 1 and (len(ranges) > 2 or tail_n == chunk):   # (two chunks of which the last is shorter: ring slot 1 is never used)
            pending[1] = _Lazy(lambda: _Stage(params(chunk), dshape, device, routes, staged_f32), device)
        if tail_n != chunk:
            pending["tail"] = _Lazy(lambda: _Stage(params(tail_n), dshape, device, routes, staged_f32), device)

        def take(key):   # a slot built by the helper thread, registered with `slots` so that the epilogue drains and waits for it
            st = pending.pop(key).get()
            slots.append(st)
            return st
        ring = {0: slots[0]}
        tail = None
        for i, (b0, b1) in enumerate(ranges):
            if "tail" in pending and i == len(ranges) - 1:
                tail = take("tail")
            if tail is not None and i == len(ranges) - 1:
                slot = tail
            else:
                if i % nring not in ring:
                    ring[i % nring] = take(i % nring)
                slot = ring[i % nring]
            if slot.range is not None:  # the slot still holds the chunk of two iterations ago
                drain(slot)
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
            outs = (slot.plan.stereo, slot.plan.depth_l, slot.plan.depth_r, slot.plan.mask)
            st = ctypes.c_void_p(s_main.cuda_stream)
            for k, r in enumerate(routes):   # the compact device forms (the uint8 stereoscope already is one: flags bit 1)
                if r == "u8" and k != 0:
                    _native.check(L.cs_pack_u8(ctypes.c_void_p(outs[k].data_ptr()), ctypes.c_void_p(slot.dev_small[k].data_ptr()),
                                               slot.dev_small[k].numel(), 3 if k in (1, 2) else 1, 1 if k == 3 else 0, st))
                elif r == "f1":
                    _native.check(L.cs_take_f32(ctypes.c_void_p(outs[k].data_ptr()), ctypes.c_void_p(slot.dev_small[k].data_ptr()),
                                                slot.dev_small[k].numel(), 3, st))
            slot.e_done.record(s_main)
            if has_f32 and not direct_known:   # our own result tensors: pinned (direct) or not is known once they exist
                fin = get_final()
                state["direct"] = direct = all(fin[k].is_pinned() for k, r in enumerate(routes) if r == "f32")
                direct_known = True
            with torch.cuda.stream(s_d2h):
                s_d2h.wait_event(slot.e_done)
                for k, r in enumerate(routes):
                    if r == "f32":
                        if not direct and slot.pin_out[k] is None:   # (pageable results only: a pinned staging buffer)
                            slot.pin_out[k] = torch.empty(outs[k].shape, dtype=outs[k].dtype, pin_memory=True)
                        dst = get_final()[k][b0:b1] if direct else slot.pin_out[k]
                        dst.copy_(outs[k], non_blocking=True)
                    else:
                        slot.pin_small[k].copy_(slot.dev_small[k], non_blocking=True)
                slot.e_out.record(s_d2h)
            if host_jobs:
                slot.future = pool.submit(expand_chunk, slot, b0, b1)
            slot.range = (b0, b1)
            slot.used = True
        for slot in slots:
            if slot.range is not None:
                drain(slot)
        return get_final()
    finally:
        # (also after an exception: no queued expansion may outlive the call -- it would write into `final` and hold the pinned
        # slots -- and no transfer may still be in flight when the slots' buffers go back to the allocators)
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
        for slot in slots:
            if slot.future is not None and not slot.future.cancelled():
                try:
                    slot.future.result()
                except BaseException:
                    pass
        s_h2d.synchronize(); s_d2h.synchronize()
        for lz in pending.values():   # (an exception before a helper-built slot was taken)
            lz.thread.join()
        if results is not None:
            results.thread.join()


def prewarm(frames, h, w, depth_shape=None, modes="left-right", fill="polylines_soft", batch_size=12, device=None, calls=1):
    """OPT-IN warm-up (nothing calls it unless the user asks: GenerateStereo.PREWARM / COMFYSTEREO_PREWARM): allocate -- and
    release into PyTorch's caching allocators -- the pinned staging buffers, the device buffers and, when they fit under
    PINNED_POOL_BYTES, the pinned result tensors a `generate_host` call of this shape needs, so that the FIRST call of a
    process finds them cached (a pinned allocation runs at 24 GB/s).  `calls`: result sets to warm (a caller that still holds the
    previous results needs a second set).  Blocking; run it on a thread (`prewarm_async`).  The memory stays in PyTorch's caches
    until a call with a smaller need releases it (`_pinned_budget`) or torch.cuda.empty_cache() / `_release_pinned_cache()`."""
    if not torch.cuda.is_available():
        return False
    device = device or torch.device("cuda", torch.cuda.current_device())
    dshape = tuple(depth_shape) if depth_shape else (h, w, 3)
    p = lambda n: engine.make_params(n, h, w, dshape[0], dshape[1], dshape[2], fill, modes, 4.5, 0.0, 0.0, 0.5, 2.0, True,
                                     20.0, 20.0, 2.0, 6, batch_size)
    oh, ow, mh, mw = engine.output_shape(p(1))
    kind = "warp" if fill == "gpu_warp" else "compact"
    routes = ROUTES[kind]
    per_frame_in = 4 * (h * w * 3 + dshape[0] * dshape[1] * dshape[2])
    per_frame_out = 4 * (oh * ow * 3 + 2 * h * w * 3 + mh * mw)
    unit = 1
    if kind == "compact":
        chunk = max(1, min(frames, CHUNK_IN_BYTES // max(per_frame_in, 1)))
        if frames >= 4:
            chunk = min(chunk, (frames + 3) // 4)
    else:
        chunk = _chunk_frames(frames, per_frame_out, fill, batch_size)
        unit = max(1, min(batch_size, frames))
    shapes = ((frames, oh, ow, 3), (frames, h, w, 3), (frames, h, w, 3), (frames, mh, mw))
    f32_per_frame = sum(4 * sz for sz, r in zip((oh * ow * 3, h * w * 3, h * w * 3, mh * mw), routes) if r == "f32")
    # (the decisions generate_host will take for this shape: same helper)
    chunk, staging_bytes, pin_results = _plan_pinned(frames, chunk, unit, per_frame_in, _small_bytes_per_frame(routes, oh, ow, h, w, mh, mw),
                                                     f32_per_frame, per_frame_out, True)
    nslots = min(2, (frames + chunk - 1) // chunk)
    staged_f32 = "f32" in routes and not pin_results   # (pageable results: the float32 routes go through the slots' staging copies)
    _pinned_budget(staging_bytes + (frames * per_frame_out if pin_results else 0))
    keep = []
    if pin_results:
        keep = [[torch.empty(sh, dtype=torch.float32, pin_memory=True) for sh in shapes] for _ in range(max(1, calls))]
    keep.append([_Stage(p(chunk), dshape, device, routes, staged_f32) for _ in range(nslots)])
    if frames % chunk and (frames + chunk - 1) // chunk > 1:
        keep.append(_Stage(p(frames % chunk), dshape, device, routes, staged_f32))
    torch.cuda.synchronize(device)
    del keep
    return True


_prewarm_thread = None


def prewarm_async(*args, **kw):
    """`prewarm` on a daemon thread (what GenerateStereo.py starts at import when PREWARM names a shape)."""
    global _prewarm_thread

    def run():
        try:
            prewarm(*args, **kw)
        except Exception:   # a warm-up must never take the process down (no GPU yet, not enough lockable memory ...)
            pass
    _prewarm_thread = threading.Thread(target=run, name="comfystereo-prewarm", daemon=True)
    _prewarm_thread.start()
    return _prewarm_thread


def prewarm_wait():
    if _prewarm_thread is not None:
        _prewarm_thread.join()
