// cs_kernels.h -- internal interface between cs_abi.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/comfystereo_amd.h"

namespace cs {

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline int poly_npt(int w, int sharp) { return (sharp ? 2 * w : w) + 2; }

struct EyeArgs {
    const float* depth;  // [n][h][w] depth this eye warps with (unscaled when scale_from_stats)
    float div32, sep32;  // (float)divergence_px, (float)separation_px -- signed
    double div64, sep64; // the Python floats themselves (dialect D64)
    int enabled;         // 0: eye = source image (divergence < 0.001, quirk Q10)
    int asc;             // divergence_px < 0: sweep ascending (max source column wins)
    int naive_lim;       // abs(int(divergence_px)) + 2
    int csg_cap;         // 5 * int(abs(divergence_px)) + 25 (reference's active-list capacity)
    int st_min, st_max;  // stats words holding this eye's depth min / max
    int xoff, yoff;      // slot of this eye in the output layout
};

struct RowArgs {
    int n, h, w;
    const float* image_f32;   // [n][h][w][3] 0..1   (node path)   or null
    const uint8_t* image_u8;  // [n][h][w][3]        (asd path)    or null
    const uint32_t* stats;    // [n][ST_WORDS]
    uint32_t* stats_rw;
    int scale_from_stats;     // depth rows are multiplied by 255 when stats[ST_SCALE255]
    float e32, conv32;
    double e64;               // the exponent as the Python float (dialect D64)
    int d64;                  // dialect bits (cs_params.flags bits 3 / 4): 1 = float64 disparity chain, 2 = int64 pixel sums
    EyeArgs eye[2];
    int neyes;
    // outputs
    uint8_t* out_u8;  // asd path: [n][h][w][3]
    float* stereo;    // node path: [n][out_h][out_w][3] float32 (or uint8 codes k of k/255 when stereo_is_u8)
    int stereo_is_u8;
    int no_mask;      // (tile kernels, with stereo_is_u8) no mask output: the per-eye intermediate of the anaglyph modes
    float* mask;      //            [n][out_h][out_w]
    float* depth_l;   //            [n][h][w][3]
    float* depth_r;
    int out_h, out_w;
    int anaglyph;  // 0: eyes go to their slots; 1: R from eye0, GB from eye1; 2: R from eye1, GB from eye0
    int single;    // -1, or the only eye that is written (left-only / only-right)
    // hybrid_edge scratch (HBM): splat result of every eye, written by k_hybrid_splat
    uint8_t* hyb_base;  // [n][neyes][h][w][3]
    uint8_t* hyb_mask;  // [n][neyes][h][w]
    // lazy depth blur (cs_blur.hip): tiles of 64 x 32 pixels without an edge in reach are NOT written to the blurred depth
    // maps eye[].depth; bit (tile column) of tilemap[(frame * tile_rows + tile_row) * tm_words ..] is set for the tiles that
    // are, every other pixel is lazy_gray * (x255 scale of the frame).  Null: the blurred maps are complete.
    const uint32_t* tilemap;
    const float* lazy_gray;
    int tm_words;
    // Stretch replay OUTSIDE the row kernel (round 3, cs_rowwarp.hip k_poly_replay): a polylines row whose order-dependent
    // stretches fit the replay kernel's windows dumps its sorted points (perm, coord_d) into a slot of rp_dump and appends one
    // descriptor per stretch to rp_list; rp_ctr = {-, descriptors appended, replay cursor, -, pool units used (64 bits)}.  Null: the row kernel
    // replays its stretches itself (anaglyph modes, hybrid_edge_plus, no scratch).
    uint8_t* rp_dump;
    uint32_t* rp_list;
    uint32_t* rp_ctr;
    uint32_t rp_pool16, rp_cap;   // dump pool in 16-byte units, descriptors in the list
    const uint32_t* row_list;     // or null: process only these rows (frame * h + row), *row_count of them (tiled-path fallback)
    const uint32_t* row_count;
    // tile hints of k_polypoint (round 5): word [(frame * h + row) * 2 + eye], bit t = tile t (hint_T pixels wide) of that row-eye raised
    // the hazard, bit 31 = any tile from 31 on; the lean first pass over the flagged rows confines itself to those tiles' columns
    // plus a margin (and the sources within hint_S + 2 of them).  Null: whole rows.
    const uint32_t* hint;
    int hint_T, hint_S;
    int dbg;            // development only (cs_debug_set(CS_DEBUG_DBG, n)): see dev_switch() below
};

// Development switches (cs_debug_set, include/comfystereo_amd.h).  Release builds never read the environment; the
// switches are explicit process-wide state that only tests and profiling tools set.  In a release build CS_DEBUG_DBG
// only accepts the values that leave every output intact (14: count pixels per evaluation path, 17: no exponent
// shortcuts); the phase cut-offs of the tile and blur kernels need a -DCS_DEV build.
int dev_switch(int key);


// cs_rowwarp.hip
hipError_t launch_collect_rows(const uint8_t* flag, int total, uint32_t* count, uint32_t* list, hipStream_t stream);
hipError_t launch_rowwarp(int fill, const RowArgs& A, int threads, hipStream_t stream, int max_groups = 0, int lean = 0);
size_t rowwarp_lds_bytes(int fill, int w, int anaglyph = 1);   // anaglyph modes stash two channels of the first eye (2 B per pixel)
// scratch of the stretch replay kernel for a call of n frames (0: the frame is too wide for its windows); poly_replay_attach
// carves it into A.rp_* and zeroes the counters; launch_poly_replay runs the descriptors the row pass appended
size_t poly_replay_bytes(int n, int h, int w, int sharp);
hipError_t poly_replay_attach(RowArgs& A, int sharp, void* scratch, void* ctr_retry, hipStream_t stream, size_t surplus = 0);
hipError_t launch_poly_replay(int sharp, const RowArgs& A, int halo, hipStream_t stream);
uint8_t* poly_replay_retry_flags(const RowArgs& A);   // one byte per row: set by a stretch the replay kernel gave up on

// cs_polytile.hip: tiled fast path of polylines; flags rows it cannot do for the general kernel
hipError_t launch_polytile(int sharp, const RowArgs& A, int S, uint8_t* rowflag, hipStream_t stream);
int polytile_max_halo();

// cs_fwdtile.hip: fills 'none' / 'naive' / 'inverse' as a halo-tile kernel (node path); hipErrorNotSupported: not one of its
// cases.  'naive' flags the rows it cannot finish in `rowflag` (zeroed by the caller) for the row kernel.
hipError_t launch_fwdtile(int fill, const RowArgs& A, int S, uint8_t* rowflag, hipStream_t stream, const uint32_t* tier2_list = nullptr,
                          const uint32_t* tier2_count = nullptr);
int fwdtile_max_halo();

// cs_polypoint.hip: second generation of the tiled path (polylines_soft): one lane per polyline point
hipError_t launch_polypoint(const RowArgs& A, int S, uint8_t* rowflag, hipStream_t stream, int sharp = 0, uint32_t* hint = nullptr,
                            int* tile_width = nullptr);
// (round 6) second tier: the rows of `list` / `count` the first tier flagged, with more room for pixels under reversed segments
hipError_t launch_polypoint_tier2(const RowArgs& A, int S, uint8_t* rowflag2, hipStream_t stream, int sharp, uint32_t* hint2, int tile_width,
                                  const uint32_t* list, const uint32_t* count);
int polypoint_max_halo();
bool polypoint_sweep64_ok(int w, int halo);
// anaglyph modes behind the tile kernel: the eyes as uint8 codes side by side -> the composite (rows flagged in rowflag excepted)
hipError_t launch_anaglyph_compose(const uint8_t* sbs, const uint8_t* rowflag, int n, int h, int w, int anaglyph, float* stereo,
                                   int stereo_is_u8, float* mask, hipStream_t stream);
// depth-map output (code / 255 on three channels) of an eye the tile kernels do not visit (single-eye modes)
hipError_t launch_depth_codes(const float* depth, int n, int h, int w, const uint32_t* stats, int scale_from_stats, float* out,
                              hipStream_t stream);

// cs_blur.hip: directional depth blur; if `scale_from_stats`, the input is multiplied by 255 for frames
// whose stats say so, and the per-frame min/max of both outputs are accumulated into stats.
int launch_blur(const float* depth, int n, int h, int w, double strength, double edge_threshold, double mask_width,
                double falloff, int vert, float* out_l, float* out_r, float* wl, float* wr, uint32_t* stats, int node_path,
                hipStream_t stream, uint32_t* tilemap = nullptr, int* lazy_used = nullptr, int pre_edges = 0);
// One pass over an RGB depth map instead of k_gray + k_blur_edges4: gray depth, its per-frame min / max and the edge bit rows
// for both x255 hypotheses (cs_blur.hip k_gray_edges).  blur_pre_edges_ok: the parameters take launch_blur's lazy-tile path
// (the only consumer of the two-plane bit rows); launch_blur is then called with pre_edges = 1 and the same wl / wr.
float blur_edge_threshold_host(float den);   // the largest t with fl(t / den) <= 0.5, or < 0 (host arithmetic; no GPU)
bool blur_pre_edges_ok(int n, int h, int w, double strength, double edge_threshold, double mask_width, int vert, bool tilemap);
hipError_t launch_gray_edges(const float* rgb, float* gray, int n, int h, int w, uint32_t* stats, double strength,
                             double edge_threshold, double mask_width, int vert, float* wl, float* wr, hipStream_t stream);
// tilemap != nullptr (zeroed by launch_blur): the tiles are classified from the edge kernel's block summaries and the map names
// the tiles with an edge in reach.  lazy_used != nullptr: edge-free tiles are left UNWRITTEN (RowArgs::tilemap; *lazy_used = 0
// when the parameters took a path that writes everything); lazy_used == nullptr: a streaming copy completes the maps.
int blur_tilemap_words(int w);                 // 32-bit words per tile row, including the pad word the readers rely on
size_t blur_tilemap_bytes(int n, int h, int w);
// the rows of `list` (frame * h + row, *count of them; null: all `total` rows) complete in out_l / out_r: gray * scale for
// the unwritten tiles
hipError_t launch_lazy_rows(const uint32_t* list, const uint32_t* count, int total, const float* gray, float* out_l, float* out_r,
                            const uint32_t* tilemap, const uint32_t* stats, int h, int w, hipStream_t stream);

// cs_rowwarp.hip (hybrid_edge: k_hybrid_splat + the fill pass of k_rowwarp)
size_t hybrid_workspace_bytes(int n, int h, int w);
bool hybrid_fused_ok(int n, int w, int halo, int anaglyph, int single, int d64, int plus);
int hybrid_max_width();
int launch_hybrid(const RowArgs& A, void* workspace, hipStream_t stream, int plus = 0, int halo = -1);  // plus: hybrid_edge_plus; halo >= 0: bound of |offset| (tile splat)

// cs_scipyblur.hip: directional_motion_blur (the scipy depth blur of the numpy / PIL input path, reference :1346-1419)
size_t scipyblur_workspace_bytes(int n, int h, int w);
int launch_scipyblur(const float* depth, int n, int h, int w, double strength, double edge_threshold, double mask_width,
                     double falloff, int vert, float* out_l, float* out_r, void* workspace, hipStream_t stream);
// cs_gpuwarp.hip
size_t gpuwarp_workspace_bytes(int n, int h, int w, int group, int mesh);   // group: frames per reference sub-batch
int gpuwarp_max_width();
int meshwarp_max_width();
// mesh != 0: forward_warp_mesh (mesh-quality rasteriser) instead of forward_warp_gpu
int launch_gpuwarp_plain(const float* image, const float* depth, int n, int h, int w, double div_px, double sep_px,
                         double exponent, double convergence, float* warped, uint8_t* gap_mask, uint32_t* stats,
                         void* workspace, hipStream_t stream, int mesh = 0, double grad_thr = 1.5, int max_stretch = 8);
int launch_gpuwarp_node(const cs_params* p, const float* image, const float* dL, const float* dR, int scale_from_stats,
                        uint32_t* stats, float* stereo, float* depth_l, float* depth_r, float* mask, int out_h,
                        int out_w, void* workspace, hipStream_t stream, const uint32_t* tilemap = nullptr,
                        const float* gray = nullptr, int tm_words = 0);
// lazy depth-blur tiles in k_gpuwarp (tilemap != nullptr): rows of at most this many columns, not the mesh-quality warp
int gpuwarp_lazy_max_width();

}  // namespace cs
