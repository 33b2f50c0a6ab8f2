// cs_common.h -- shared definitions of the gfx950 kernels (device-side helpers, workspace layout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/comfystereo_amd.h"
#include "cs_math.h"

#define CS_LDS_BYTES 163840  // 160 KiB per CU on gfx950, all of it usable by one workgroup
#define CS_WAVE 64

// Per-frame statistics block in the workspace (uint32 words; floats stored through csm::f2ord so
// that integer atomics order them).  Written by the prep / blur kernels, read by the row kernels.
enum {
    ST_GRAY_MIN = 0,  // min / max of the gray depth as delivered (before the 0..255 scaling)
    ST_GRAY_MAX = 1,
    ST_L_MIN = 2,  // min / max of the left-eye depth the warp normalises with (scaled, post-blur)
    ST_L_MAX = 3,
    ST_R_MIN = 4,
    ST_R_MAX = 5,
    ST_SCALE255 = 6,  // 1 -> the reference multiplies this frame's depth by 255 (max <= 1)
    ST_WARP_DIV255_L = 7,  // gpu_warp: forward_warp_gpu divides the (sub-batch's) depth by 255 again
    ST_WARP_DIV255_R = 8,
    ST_ERROR = 9,  // set by kernels on internal capacity overflow (diagnostics)
    ST_FALLBACK_ROWS = 10,  // polylines: rows that took the sequential path (diagnostics)
    ST_TILE_REDO_ROWS = 11,  // polylines: rows the tiled fast path handed to the general row kernel
    ST_WORDS = 16
};

namespace cs {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

struct OpAdd {
    __device__ __forceinline__ int operator()(int a, int b) const { return a + b; }
};
struct OpMax {
    __device__ __forceinline__ int operator()(int a, int b) const { return a > b ? a : b; }
};
struct OpMin {
    __device__ __forceinline__ int operator()(int a, int b) const { return a < b ? a : b; }
};

// In-place inclusive scan of a[0..n) in LDS by the whole workgroup.  Each thread owns a contiguous
// chunk; chunk totals are combined with wave shuffles (64-wide) and one cross-wave step.
// `wsum` is LDS scratch of >= 16 ints.  If `reverse`, the scan runs from the last element down.
template <class T, class Op>
__device__ void block_scan_inclusive(T* a, int n, int identity, Op op, int* wsum, bool reverse = false) {
    const int nt = blockDim.x, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const int L = (n + nt - 1) / nt;
    const int b = tid * L, e = min(b + L, n);
    int acc = identity;
    for (int i = b; i < e; i++) {
        int j = reverse ? n - 1 - i : i;
        acc = op(acc, (int)a[j]);
        a[j] = (T)acc;
    }
    int v = acc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int u = __shfl_up(v, off);
        if (lane >= off) v = op(u, v);
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    if (wave == 0) {
        int nw = nt >> 6;
        int s = lane < nw ? wsum[lane] : identity;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            int u = __shfl_up(s, off);
            if (lane >= off) s = op(u, s);
        }
        if (lane < nw) wsum[lane] = s;
    }
    __syncthreads();
    int ex = __shfl_up(v, 1);
    if (lane == 0) ex = identity;
    int prefix = op(wave > 0 ? wsum[wave - 1] : identity, ex);
    for (int i = b; i < e; i++) {
        int j = reverse ? n - 1 - i : i;
        a[j] = (T)op(prefix, (int)a[j]);
    }
    __syncthreads();
}

// atomic add on element `i` of a uint16 array that lives in LDS as packed pairs; returns the old value.
__device__ __forceinline__ unsigned atomic_add_u16(uint16_t* a, int i, unsigned v) {
    unsigned* w = reinterpret_cast<unsigned*>(a) + (i >> 1);
    unsigned sh = (i & 1) * 16;
    unsigned old = atomicAdd(w, v << sh);
    return (old >> sh) & 0xffffu;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}


// Per-frame min/max into order-preserving uint keys with as little atomic traffic as possible: thousands of
// workgroups updating the same few words serialise in L2 (measured: 0.24 ms per 4K frame), so (1) reduce inside
// the workgroup through LDS, (2) one lane compares against the current value (agent-scope relaxed load; a stale
// value can only cause a redundant atomic, never a missed one, because min only falls and max only rises) and
// (3) only then issue the atomic.  `red` is LDS scratch of 2 * (blockDim.x / 64) floats.
__device__ __forceinline__ void block_minmax_update(float mn, float mx, uint32_t* st_min, uint32_t* st_max, float* red) {
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int nw = blockDim.x >> 6;
    if (lane_id() == 0) { red[2 * wave_id()] = mn; red[2 * wave_id() + 1] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; i++) { mn = fminf(mn, red[2 * i]); mx = fmaxf(mx, red[2 * i + 1]); }
        uint32_t kmn = csm::f2ord(mn), kmx = csm::f2ord(mx);
        if (kmn < __hip_atomic_load(st_min, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(st_min, kmn);
        if (kmx > __hip_atomic_load(st_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(st_max, kmx);
    }
    __syncthreads();
}

// ---- grid of the tile kernels that give a workgroup to one (tile, row, EYE): k_polypoint, k_polytile, k_hybrid_splat_tile ----
// blockIdx.x = tile * 8 + (row & 7): workgroup b runs on XCD b % 8 (observed dispatch order, MI355X_MICROARCH.md; speed only),
// so all tiles of a row share one L2.  blockIdx.y (two-eye launches): the two eyes of a GROUP of 8 << CS_EYE_GROUP rows follow
// each other in dispatch order -- rows 0..127 of the left eye, rows 0..127 of the right eye, rows 128..255 of the left eye ... --
// so the second eye finds the image rows (and the gray depth of edge-free blur tiles) in its XCD's L2 instead of fetching them
// again: FETCH_SIZE of k_polypoint halves (16.6 -> 8.3 GB per 64 4K frames, less than the algorithmic 10.6: the gray depth is
// shared as well) and the kernel is 3.5 % faster (profiles/r04_eye_group.txt).  Groups of 8 rows (the eyes adjacent) are 3 %
// SLOWER than eye-major order, as round 2 had found: the two eyes of a tile then run at the same time and their output streams
// collide; with 32 rows or more between them they do not.
#ifndef CS_EYE_GROUP
#define CS_EYE_GROUP 4
#endif
__host__ __device__ __forceinline__ int eye_group_grid_y(int h) {
    const int G = 1 << CS_EYE_GROUP, ny = (h + 7) / 8;
    return 2 * ((ny + G - 1) / G) * G;
}
// blockIdx.y of a two-eye launch -> the 8-row block and the eye (rows beyond h: the caller returns)
__device__ __forceinline__ void eye_group_decode(int yi, int& yrow, int& eye) {
    yrow = ((yi >> (CS_EYE_GROUP + 1)) << CS_EYE_GROUP) | (yi & ((1 << CS_EYE_GROUP) - 1));
    eye = (yi >> CS_EYE_GROUP) & 1;
}

// ---- lazy depth-blur tiles (cs_blur.hip k_blur_classify, RowArgs::tilemap) ------------------------------------------
// The blurred depth map of an eye only holds the 64 x 32 tiles the map names; everywhere else the value is
// gray * (the frame's x255 scale).  A tile kernel that stages the columns [s0, s0 + 2048) of one row sets a selector up
// once per workgroup (scalar work) and then selects per lane with three bit operations: the lane reads the LOWER of the
// two buffers, `delta` bytes further up when the bit of its tile is set, and multiplies by 1 (blurred: already scaled) or
// by the scale (gray).  The two buffers must lie within 4 GB of each other (checked on the host).
struct LazySel {
    const char* base;                    // row start (column s0) in the lower buffer
    uint32_t bits, delta, mul_set, mul_clr;
};
__device__ __forceinline__ LazySel lazy_select(const uint32_t* tilemap, int tm_words, int frame, int h, int row, int s0,
                                               const char* blurred_row, const char* gray_row, uint32_t scale255) {
    const uint32_t* tm = tilemap + ((uint32_t)frame * (uint32_t)((h + 31) >> 5) + (uint32_t)(row >> 5)) * (uint32_t)tm_words;
    const int t0 = s0 >> 6;
    const uint32_t w0 = tm[t0 >> 5], w1 = tm[(t0 >> 5) + 1];   // (the map rows end with a pad word)
    const uint32_t low = (1u << (t0 & 31)) - 1u;                // tiles of the second word wrap into the low bits
    const uint32_t bits = (w0 & ~low) | (w1 & low);             // bit (t & 31) set: tile t was written to the blurred map
    const uint32_t sc = scale255 ? 0x437f0000u /* 255.0f */ : 0x3f800000u;
    const bool gray_low = gray_row < blurred_row;
    LazySel Z;
    Z.base = gray_low ? gray_row : blurred_row;
    Z.delta = (uint32_t)(gray_low ? blurred_row - gray_row : gray_row - blurred_row);
    Z.bits = gray_low ? bits : ~bits;
    Z.mul_set = gray_low ? 0x3f800000u : sc;
    Z.mul_clr = gray_low ? sc : 0x3f800000u;
    return Z;
}
// depth of column `col` = s0 + j (not yet multiplied by `mul`)
__device__ __forceinline__ float lazy_load(const LazySel& Z, uint32_t col, uint32_t j, float& mul) {
    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)Z.bits, col >> 6, 1u);   // 0 / ~0
    mul = __builtin_bit_cast(float, (m & Z.mul_set) | (~m & Z.mul_clr));
    return *reinterpret_cast<const float*>(Z.base + (4u * j + (m & Z.delta)));
}

#if defined(__HIPCC__)
// a / b, correctly rounded, for operands whose quotient, reciprocal and residuals stay far from the float32 range limits
// (pixel coordinates and their differences).  The FMA core of the IEEE expansion hipcc emits for `a / b`
// (v_rcp_f32, one Newton step, quotient, two residual corrections) without v_div_scale / v_div_fixup, which only act on
// operands near the range limits: bit-identical results, 8 instead of 11 VALU.
__device__ __forceinline__ float div_core(float a, float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, y0, 1.0f);
    const float y1 = __builtin_fmaf(e0, y0, y0);
    const float q0 = a * y1;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y1, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}
// the same with the refined reciprocal y1 of b supplied (several numerators over one denominator)
__device__ __forceinline__ float rcp_refined(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e0, y0, y0);
}
__device__ __forceinline__ float div_with(float a, float b, float y1) {
    const float q0 = a * y1;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y1, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}

#endif

}  // namespace cs
