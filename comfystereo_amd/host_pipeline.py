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

Results are bit-identical to processing the whole batch at once (every quantity of the path is per frame; gpu_warp's
two 0..255 decisions are per reference sub-batch, so chunks are multiples of `batch_size` for that technique).
"""
import torch

from . import engine

# target size of one chunk's outputs in bytes: large enough for full-rate PCIe transfers, small enough that the
# pinned staging (2 x in + 2 x out) stays a few GB
CHUNK_OUT_BYTES = 3 << 30


def _chunk_frames(total, per_frame_out, fill, batch_size):
    chunk = max(1, min(total, CHUNK_OUT_BYTES // max(per_frame_out, 1)))
    if fill == 'gpu_warp':  # keep the reference's sub-batch boundaries (its 0..255 tests are per sub-batch)
        sub = max(1, min(batch_size, total))
        chunk = max(sub, (chunk // sub) * sub)
    return chunk


class _Stage:
    """Buffers of one pipeline slot: pinned + device inputs, a Plan (device outputs + workspace), pinned outputs."""

    def __init__(self, p, depth_shape, device, staged_outputs):
        n, h, w = p.n, p.h, p.w
        self.plan = engine.Plan(p, device)
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
                  pinned_outputs=True, out=None):
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
        try:
            final = tuple(torch.empty(sh, dtype=torch.float32, pin_memory=True) for sh in shapes)
        except RuntimeError:  # not enough lockable memory: fall back to pageable results through staging buffers
            final = None
    direct = final is not None
    if not direct:
        final = out if out is not None else tuple(torch.empty(sh, dtype=torch.float32) for sh in shapes)
    ranges = [(b0, min(b0 + chunk, total)) for b0 in range(0, total, chunk)]
    slots = [_Stage(params(chunk), dshape, device, not direct) for _ in range(min(2, len(ranges)))]
    tail = None  # a shorter last chunk gets its own (smaller) slot
    if ranges[-1][1] - ranges[-1][0] != chunk:
        tail = _Stage(params(ranges[-1][1] - ranges[-1][0]), dshape, device, not direct)
    s_h2d, s_d2h = torch.cuda.Stream(device), torch.cuda.Stream(device)
    s_main = torch.cuda.current_stream(device)

    def drain(slot):  # the slot's chunk is complete on the host
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
            if direct:              # its inputs must have left the pinned staging; the rest is ordered on the device
                slot.e_in.synchronize()
            else:
                drain(slot)
        n = b1 - b0
        slot.pin_img[:n].copy_(image[b0:b1])
        slot.pin_dep[:n].copy_(depth_map[b0:b1])
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
        slot.e_done.record(s_main)
        with torch.cuda.stream(s_d2h):
            s_d2h.wait_event(slot.e_done)
            outs = (slot.plan.stereo, slot.plan.depth_l, slot.plan.depth_r, slot.plan.mask)
            for k, src in enumerate(outs):
                dst = final[k][b0:b1] if direct else slot.pin_out[k]
                dst.copy_(src, non_blocking=True)
            slot.e_out.record(s_d2h)
        if direct and slot.range is not None and progress:
            progress(slot.range[1] - slot.range[0])
        slot.range = (b0, b1)
        slot.used = True
    for slot in slots + ([tail] if tail is not None else []):
        if slot.range is not None:
            drain(slot)
    return final
