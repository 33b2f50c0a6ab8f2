/*
 * comfystereo_amd.h -- C ABI of the MI355X-native depth-to-stereo engine (libcomfystereo_hip.so).
 *
 * The reference (Dobidop/ComfyStereo) is pure Python and has no FFI; its only seam is the Python
 * module boundary `GenerateStereo.py` -> `stereoimage_generation.py`.  The entry points below are what
 * a binding for that seam binds: each one cites the reference interface it replaces.  They are
 * called by comfystereo_amd/_native.py (ctypes) on behalf of the drop-in module functions
 * `create_stereoimages` / `create_stereoimages_gpu` and of `StereoImageNode.generate`.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is caller-owned DEVICE memory (HIP), row-major,
 *     contiguous in the stated shape; no allocation happens inside the library
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it
 *   - return value: CS_OK (0) or a negative CS_E* code; cs_last_error() returns a thread-local
 *     message for the last failing call
 *   - scalar parameters are doubles because they are Python floats in the reference, which rounds
 *     them to float32 at specific points of the arithmetic
 *   - re-entrant per stream as long as each concurrent call gets its own workspace; the only process-wide state
 *     is opt-in (cs_profile's event pool, guarded by a mutex; cs_debug_set's development switches)
 */
#ifndef COMFYSTEREO_AMD_H
#define COMFYSTEREO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_ABI_VERSION 4

#if defined(__GNUC__)
#define CS_API __attribute__((visibility("default")))
#else
#define CS_API
#endif

enum cs_status {
    CS_OK = 0,
    CS_EINVAL = -1,    /* bad argument (null pointer, size <= 0, unknown enum) */
    CS_EWORKSPACE = -2,/* workspace too small: see cs_workspace_bytes */
    CS_ELIMIT = -3,    /* frame too wide for the LDS-resident row kernels (see cs_max_width) */
    CS_EHIP = -4       /* a HIP runtime call failed */
};

/* fill_technique keys: 0..7 are the UI-reachable ones of reference GenerateStereo.py:88-100; 8..10 are the remaining
 * branches of the dispatcher stereoimage_generation.py:1605-1610, reachable through its module functions only */
enum cs_fill {
    CS_FILL_NONE = 0,                /* 'none'                 stereoimage_generation.py:1850-1910 */
    CS_FILL_NAIVE = 1,               /* 'naive'                :1893-1908 */
    CS_FILL_NAIVE_INTERPOLATING = 2, /* 'naive_interpolating'  :1871-1892 */
    CS_FILL_POLYLINES_SOFT = 3,      /* 'polylines_soft'       :1912-1992 */
    CS_FILL_POLYLINES_SHARP = 4,     /* 'polylines_sharp'      :1912-1992 */
    CS_FILL_INVERSE = 5,             /* 'inverse'              :1715-1737 */
    CS_FILL_HYBRID_EDGE = 6,         /* 'hybrid_edge'          :1837-1848 */
    CS_FILL_GPU_WARP = 7,            /* 'gpu_warp'             forward_warp_gpu :277-450 */
    /* branches of the dispatcher that no UI string reaches (reference :1605-1610); not valid for gpu paths */
    CS_FILL_NONE_POST = 8,           /* 'none_post'            :1804-1817 (forward map + row-wise np.interp) */
    CS_FILL_INVERSE_POST = 9,        /* 'inverse_post'         :1820-1833 */
    CS_FILL_HYBRID_EDGE_PLUS = 10    /* 'hybrid_edge_plus'     :1778-1802 (hybrid_edge, black pixels from polylines_soft) */
};

/* output modes of reference stereoimage_generation.py:1543-1562 / :1093-1120 */
enum cs_mode {
    CS_MODE_LEFT_RIGHT = 0,
    CS_MODE_RIGHT_LEFT = 1,
    CS_MODE_TOP_BOTTOM = 2,
    CS_MODE_BOTTOM_TOP = 3,
    CS_MODE_RED_CYAN_ANAGLYPH = 4,
    CS_MODE_LEFT_ONLY = 5,
    CS_MODE_ONLY_RIGHT = 6,
    CS_MODE_CYAN_RED_REVERSEANAGLYPH = 7
};

/* One call of StereoImageNode.generate (reference GenerateStereo.py:79-80): widget values + shapes. */
typedef struct cs_params {
    int32_t n, h, w;            /* image batch [n][h][w][3] float32 0..1 (ComfyUI IMAGE)            */
    int32_t depth_h, depth_w;   /* depth batch [n][depth_h][depth_w][depth_c] float32               */
    int32_t depth_c;            /* 3 -> 0.2989 R + 0.5870 G + 0.1140 B; 1 -> as is; else channel 0  */
    int32_t fill;               /* enum cs_fill                                                     */
    int32_t mode;               /* enum cs_mode                                                     */
    int32_t batch_size;         /* gpu_warp only: frames per reference sub-batch (its 0..255 test is
                                   global over a sub-batch, stereoimage_generation.py:1045, :315)   */
    int32_t depth_map_blur;     /* bool: direction-aware depth blur on/off                          */
    int32_t depth_blur_vert_smooth;
    int32_t flags;              /* bit 0: gpu_warp depth outputs are NOT clamped to 0..1 (module-level
                                   create_stereoimages_gpu returns them unclamped, :1125-1126)
                                   bit 1: `stereo` receives the uint8 codes k (value = k/255) instead of
                                   float32 -- the compact form frame shards are all-gathered in; CPU
                                   techniques only (gpu_warp colours are genuine floats)
                                   bit 2: fill gpu_warp runs the mesh-quality warp (forward_warp_mesh,
                                   :453-689 -- what the reference does when moderngl is importable,
                                   :1068-1071) instead of forward_warp_gpu; see cs_forward_warp_mesh
                                   bits 3, 4: arithmetic dialect.  0 = D32, the reference WITHOUT numba
                                   (float32 disparities, uint8 pixel sums that wrap) -- the pinned
                                   contract.  bit 3: float64 disparity chain, bit 4: int64 pixel
                                   sums; both = D64, the typing numba gives the reference's kernels
                                   (derived).  polylines_soft / sharp: bit 3 =
                                   point coordinates from the float64 chain, rounded once into the
                                   float32 point array (pinned: tests/golden/dialect_f64.npz), bit 4 =
                                   numba's float64 typing of the sweep (derived; a literal one-lane
                                   replay per row in the general row kernel -- a compatibility path).
                                   hybrid_edge: bit 3 = dest_x, its distance to the column and the exp
                                   argument in float64 (pinned), bit 4 = float64 weight sums (derived).
                                   none / naive / naive_interpolating / inverse / polylines_* /
                                   hybrid_edge only, else CS_EINVAL                                */
    double divergence, separation, stereo_balance, convergence_point, stereo_offset_exponent;
    double depth_blur_strength, depth_blur_edge_threshold, depth_blur_falloff;
} cs_params;

CS_API int cs_version(void);
CS_API const char *cs_last_error(void);

/* Largest frame width the LDS-resident row kernels accept for `fill` (160 KiB LDS per CU). */
CS_API int cs_max_width(int fill);
/* The same for one output mode.  The row kernels' own anaglyph form keeps 2 more bytes of LDS per pixel; since round 6 every technique but
 * hybrid_edge_plus runs an anaglyph beyond that form's width side by side into scratch and composes afterwards, so the anaglyph limit equals
 * the side-by-side limit (cs_max_width is the anaglyph, i.e. smallest, limit). */
CS_API int cs_max_width_mode(int fill, int mode);
/* ABI 4: the widest frame cs_generate accepts with p's technique, mode, dialect flags (bits 3 / 4) and disparity parameters
 * (p->w, p->h, p->n are ignored): the same predicate the call itself applies, so pre-validation cannot disagree with it.
 * It can be lower than cs_max_width_mode -- e.g. an anaglyph of a polylines technique beyond the row kernel's own anaglyph
 * form only passes while the tile kernels take it (halo within their reach, no full-D64 flag). */
CS_API int cs_max_width_params(const cs_params *p);

/* Shape of the outputs of cs_generate for `p`: stereoscope [n][*out_h][*out_w][3],
 * mask [n][*mask_h][*mask_w] (output-shaped for the CPU techniques, eye-shaped for gpu_warp). */
CS_API int cs_output_shape(const cs_params *p, int *out_h, int *out_w, int *mask_h, int *mask_w);

/* Scratch bytes cs_generate needs for `p` (intermediate gray/blurred depth, per-frame statistics). */
CS_API size_t cs_workspace_bytes(const cs_params *p);

/*
 * The fused batch path.  Replaces StereoImageNode.generate's per-frame loop
 * (GenerateStereo.py:117-269: grayscale, resize, create_stereoimages[_gpu], convertResult,
 * generate_mask) for device-resident tensors:
 *   image     [n][h][w][3]            float32 in 0..1
 *   depth     [n][depth_h][depth_w][depth_c] float32
 *   stereo    [n][out_h][out_w][3]    float32   "stereoscope"
 *   depth_l/r [n][h][w][3]            float32   "blurred_depthmap_left/right"
 *   mask      [n][mask_h][mask_w]     float32   "no_fill_imperfect_mask"
 */
CS_API int cs_generate(const cs_params *p, const float *image, const float *depth, float *stereo, float *depth_l,
                float *depth_r, float *mask, void *workspace, size_t workspace_bytes, void *stream);

/*
 * apply_stereo_divergence (reference stereoimage_generation.py:1576-1620) for one eye of `n`
 * independent frames: per-frame min/max normalisation, convergence shift, percent -> pixels, row
 * kernel `fill` (any CPU technique).  image_u8 [n][h][w][3] uint8, depth [n][h][w] float32,
 * out_u8 [n][h][w][3].  workspace: cs_asd_workspace_bytes_for(n, h, w, fill) (cs_asd_workspace_bytes: enough for any technique).
 */
CS_API size_t cs_asd_workspace_bytes(int n, int h, int w);
CS_API size_t cs_asd_workspace_bytes_for(int n, int h, int w, int fill);
CS_API int cs_apply_stereo_divergence(const uint8_t *image_u8, const float *depth, int n, int h, int w, double divergence,
                               double separation, double stereo_offset_exponent, int fill, double convergence_point,
                               uint8_t *out_u8, void *workspace, size_t workspace_bytes, void *stream);
/* The same with the arithmetic dialect named: 0 = D32 (as above), 1 = float64 disparity chain, 2 = int64 pixel sums,
 * 3 = both = D64 (cs_params.flags bits 3 / 4). */
CS_API int cs_apply_stereo_divergence2(const uint8_t *image_u8, const float *depth, int n, int h, int w, double divergence,
                                double separation, double stereo_offset_exponent, int fill, double convergence_point,
                                int dialect, uint8_t *out_u8, void *workspace, size_t workspace_bytes, void *stream);

/*
 * directional_motion_blur_gpu (reference stereoimage_generation.py:1171-1251; its own callers pass
 * blur_mask_width = blur_strength, :1051-1054 / :1479-1482).  depth, out_l, out_r: [n][h][w]
 * float32 on the 0..255 scale.  blur_strength <= 0 copies the input (:1194).
 * workspace: cs_blur_workspace_bytes(n, h, w).
 */
CS_API size_t cs_blur_workspace_bytes(int n, int h, int w);
CS_API int cs_directional_blur(const float *depth, int n, int h, int w, double blur_strength, double edge_threshold,
                        double blur_mask_width, double falloff_exponent, int vert_smooth_px, float *out_l, float *out_r, void *workspace,
                        size_t workspace_bytes, void *stream);

/*
 * directional_motion_blur (reference stereoimage_generation.py:1346-1419): the scipy depth blur create_stereoimages applies to
 * numpy / PIL inputs (:1489-1494) -- scipy.ndimage's float64 accumulation, 'reflect' / 'nearest' borders, no x255 rescaling.
 * depth [n][h][w] float32 (as given) -> out_l / out_r [n][h][w] float32.  workspace: cs_blur_scipy_workspace_bytes(n, h, w).
 * CS_EINVAL when blur_strength rounds to 0 taps (the reference raises there as well).
 */
CS_API size_t cs_blur_scipy_workspace_bytes(int n, int h, int w);
CS_API int cs_directional_blur_scipy(const float *depth, int n, int h, int w, double blur_strength, double edge_threshold,
                              double blur_mask_width, double falloff_exponent, int vert_smooth_px, float *out_l, float *out_r,
                              void *workspace, size_t workspace_bytes, void *stream);

/*
 * forward_warp_gpu (reference stereoimage_generation.py:277-450) for a sub-batch:
 * image [n][3][h][w] float32, depth [n][h][w] float32 -> warped [n][3][h][w] float32,
 * gap_mask [n][h][w] uint8 (1 = disocclusion).  workspace: cs_warp_workspace_bytes(n, h, w).
 */
CS_API size_t cs_warp_workspace_bytes(int n, int h, int w);
CS_API int cs_forward_warp(const float *image, const float *depth, int n, int h, int w, double divergence_px,
                    double separation_px, double stereo_offset_exponent, double convergence_point, float *warped,
                    uint8_t *gap_mask, void *workspace, size_t workspace_bytes, void *stream);
/* the same with the reference's two keyword parameters (stereoimage_generation.py:277-279): gradient_threshold -- adjacent pixels
 * are connected when their offsets differ by less than it (:339-340), max_stretch -- scatter rounds (:365).  CS_ELIMIT when more
 * than 16 rounds could change a column (gradient_threshold > 13 together with max_stretch > 16). */
CS_API int cs_forward_warp2(const float *image, const float *depth, int n, int h, int w, double divergence_px,
                     double separation_px, double stereo_offset_exponent, double convergence_point, double gradient_threshold,
                     int max_stretch, float *warped, uint8_t *gap_mask, void *workspace, size_t workspace_bytes, void *stream);

/*
 * forward_warp_mesh (reference stereoimage_generation.py:453-689), the mesh-quality warp the reference uses whenever
 * `moderngl` is importable (:1068-1071): same tensors as cs_forward_warp plus the culling threshold (reference
 * default 1.5).  The reference rasterises through OpenGL; the parts OpenGL leaves to the implementation are fixed as
 * documented in oracle/stereo_oracle.c (oracle_forward_warp_mesh) -- no bit parity with a particular GL driver is
 * claimed.  Needs h >= 2 and w >= 2.  Through cs_generate the same warp is selected by cs_params.flags bit 2 with
 * fill CS_FILL_GPU_WARP.  workspace: cs_warp_mesh_workspace_bytes(n, h, w).
 */
CS_API size_t cs_warp_mesh_workspace_bytes(int n, int h, int w);
CS_API int cs_forward_warp_mesh(const float *image, const float *depth, int n, int h, int w, double divergence_px,
                         double separation_px, double stereo_offset_exponent, double convergence_point,
                         double gradient_threshold, float *warped, uint8_t *gap_mask, void *workspace,
                         size_t workspace_bytes, void *stream);

/*
 * out[i] = codes[i] / 255 (float32, true division): expands a uint8 stereoscope (cs_params.flags bit 1), e.g.
 * after the frame shards of a multi-GPU job were all-gathered in their compact form
 * (= convertResult / np2tensor, reference GenerateStereo.py:41-44, 365-378).
 */
CS_API int cs_expand_u8(const uint8_t *codes, float *out, size_t count, void *stream);

/*
 * The compact node boundary of the host pipeline (replaces the float32 device -> host copies around
 * convertResult / np2tensor / generate_mask, reference GenerateStereo.py:41-44, 159-177, 355-378).
 * cs_pack_u8 (device): codes[i] = the uint8 code of values[i * stride] -- mode 0: value k / 255 -> k; mode 1: non-zero -> 1
 * (mask).  stride 3 takes one code per pixel of a depth map with three equal channels.
 * cs_host_expand_u8 (HOST memory, blocking, `threads` worker threads, 0 = one per online core up to 64):
 * out[i * replicate + r] = codes[i] / 255.0f (mode 0, true division) or codes[i] != 0 (mode 1), r < replicate <= 4.
 * Bit-identical to the float32 outputs of cs_generate for the CPU techniques.
 */
CS_API int cs_pack_u8(const float *values, uint8_t *codes, size_t count, int stride, int mode, void *stream);
CS_API int cs_host_expand_u8(const uint8_t *codes, float *out, size_t count, int replicate, int mode, int threads);
/* cs_host_copy (HOST memory, blocking): memcpy in `threads` contiguous slices (0 = one per online core up to 64) -- the
 * staging copy of a pageable input tensor into a pinned buffer (reference GenerateStereo.py:126-131, .to(device)). */
CS_API int cs_host_copy(void *dst, const void *src, size_t bytes, int threads);
/* gpu_warp's depth-map outputs are genuine floats on three EQUAL channels (reference GenerateStereo.py:165-169: clamp(0, 1),
 * unsqueeze(-1).expand): cs_take_f32 (device) keeps one channel -- out[i] = values[i * stride] -- so that 4 instead of 12
 * bytes per pixel cross PCIe, cs_host_replicate_f32 (HOST memory, blocking) writes out[i * replicate + r] = values[i],
 * r < replicate <= 4, with `threads` worker threads (0 = one per online core up to 64). */
CS_API int cs_take_f32(const float *values, float *out, size_t count, int stride, void *stream);
CS_API int cs_host_replicate_f32(const float *values, float *out, size_t count, int replicate, int threads);

/*
 * stereo_shift_torch (reference stereo_utils.py:15-88; callers stereodiffusion_nodes.py:650, :664): the depth-driven
 * forward shift of the `none` technique on a float payload -- diffusion latents.  input [b][c][h][w] float32, depth
 * [b][h][w] float32 (normalised with its global min / max like the reference, :36-45) -> out [2b][c][h][w]: the left
 * views (the input itself unless shift_both) followed by the right views; destinations nothing lands on stay 0.
 * torch.pow is exact for exponents 1 (the callers' value), 2 and 0.5; other exponents go through libm powf.
 */
CS_API size_t cs_stereo_shift_workspace_bytes(void);
CS_API int cs_stereo_shift(const float *input, const float *depth, int b, int c, int h, int w, double scale_factor,
                    int shift_both, double stereo_offset_exponent, float *out, void *workspace, size_t workspace_bytes,
                    void *stream);

/*
 * Measurement hook for bench.py: while enabled, cs_generate brackets the launch of its dominant
 * kernel (the row warp + fill kernel of the selected technique) with HIP events on the caller's
 * stream.  cs_profile_read waits for the recorded events, returns the summed kernel time in
 * milliseconds and the number of launches, and clears the record.
 */
CS_API int cs_profile(int enable);
CS_API int cs_profile_read(double *total_ms, int *launches);
/* cs_profile_tiles (blocking): the share of 64 x 32 depth tiles the last profiled cs_generate call blurred (the warp kernel reads
 * the others from the gray depth, shared by both eyes); -1 when that call's warp kernel read complete blurred maps. */
CS_API int cs_profile_tiles(double *fraction);

/*
 * Development switches for the parity tests and profiling tools (compare two code paths of the same kernel,
 * count pixels per evaluation path).  Explicit, process-wide, opt-in state like cs_profile; the library never reads
 * the environment.  Release builds reject CS_DEBUG_DBG values that would leave outputs unwritten (phase cut-offs
 * exist in -DCS_DEV builds only).  No reference counterpart.
 */
enum cs_debug_key {
    CS_DEBUG_DBG = 0,               /* 14: count pixels per evaluation path into spare stats words; 17: no exponent shortcuts */
    CS_DEBUG_NO_TILE = 1,           /* polylines: general row kernel for every row instead of the tiled path */
    CS_DEBUG_PT_VARIANT = 2,        /* tile kernel: other list capacities / the previous kernel generation */
    CS_DEBUG_BLUR_TWO_PASS = 3,     /* depth blur: two-pass row kernels */
    CS_DEBUG_BLUR_EDGES_SCALAR = 4, /* depth blur: one-column-per-lane edge kernel */
    CS_DEBUG_BLUR_FULL_COPY = 5,    /* depth blur: write the edge-free tiles as well (no lazy tile map for the warp kernel) */
    CS_DEBUG_CHUNKS = 6,            /* cs_generate: k > 1 = k frame chunks, pre-passes on an auxiliary stream (+100: at default priority); default one chunk */
    CS_DEBUG_NO_REPLAY_KERNEL = 7,  /* polylines: order-dependent stretches replayed inside the row kernel (round-2 schedule) */
    CS_DEBUG_BLUR_NO_PRE_EDGES = 8, /* depth blur: k_gray + k_blur_edges4 instead of the one-pass k_gray_edges */
    CS_DEBUG_HYBRID_UNFUSED = 9,    /* hybrid_edge: splat result -> node outputs in a streaming pass of its own (k_hybrid_out4) */
    CS_DEBUG_GPUWARP_FULL_MAPS = 10, /* gpu_warp with the depth blur: complete blurred maps (k_blur_copy_tiles) instead of the tile map */
    CS_DEBUG_HYBRID_FULL_MAPS = 11, /* hybrid_edge with the depth blur: complete blurred maps instead of the tile map */
    CS_DEBUG_KEYS = 12
};
CS_API int cs_debug_set(int key, int value);

/* Device self-tests of the libm-exact scalar routines (used by the parity tests):
 * out[i] = powf(x[i], y) / out[i] = exp(x[i]) evaluated by the same device code the kernels use. */
CS_API int cs_test_powf(const float *x, float y, float *out, size_t count, void *stream);
CS_API int cs_test_exp(const double *x, double *out, size_t count, void *stream);
/* Host only (no GPU): the comparison value k_gray_edges uses for the depth blur's edge test -- the largest float t with
 * fl(t / den) <= 0.5, so that clamp(|g| / den, 0, 1) > 0.5 (reference stereoimage_generation.py:1213-1222) <=> |g| > t;
 * negative when den is not positive and finite (the kernels then keep the division). */
CS_API float cs_test_edge_threshold(float den);

#ifdef __cplusplus
}
#endif
#endif
