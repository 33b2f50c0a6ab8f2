// cs_blur.hip -- direction-aware depth blur (reference stereoimage_generation.py:1171-1251,
// `directional_motion_blur_gpu` + `_edge_distance_weight_gpu` :1131-1168).
//
// Default pipeline (further down): k_blur_edges4 (edge masks of the frame as bit rows) -> k_blur_copy (tiles without
// an edge in reach are a scaled copy; the others go onto a worklist) -> k_blur_fused (persistent workgroups over the
// worklist: weights from the bit rows, both boxes and the blend from one staged tile).  The two row kernels described
// next (pass A / pass B) are the general fallback for blur kernels too wide for the fused tile.
//
// The parity target is the reference as CPU torch executes it: each conv2d
// is a raster-order fmaf accumulation from 0 with zero padding, everything else is separate
// float32 elementwise ops.  So the kernels keep exactly that order:
//   pass A (one workgroup per row): Sobel-x over rows y-1..y+1 (9 fmaf, kh outer / kw inner),
//           edge masks, nearest-edge distance by a prefix-max and a suffix-min scan in LDS,
//           weight = clamp(1 - dist/radius, 0, 1) ** falloff  ->  wl, wr (float32, HBM scratch)
//   pass B (one workgroup per row): vertical (2v+1) box over the weights (fmaf chain, top to
//           bottom), horizontal k-tap box over the depth row staged in LDS (fmaf chain, left to
//           right), blend w*blur + (1-w)*depth, per-frame min/max of both outputs for the warp.
// torch.pow(x, falloff) is exact for 0.5 / 1 / 2 / 3 (special-cased by torch); other exponents go
// through the libm-exact powf and only approximate torch's vectorised pow.
#include "cs_common.h"
#include "cs_kernels.h"
#include <stdlib.h>

namespace cs {

struct BlurArgs {
    const float* depth;  // [n][h][w]
    int n, h, w;
    const uint32_t* stats;  // node path: ST_SCALE255 decides the x255 scaling; else null
    uint32_t* stats_rw;     // node path: min/max of outputs accumulate here; else null
    float den;              // (float)(10 * edge_threshold)
    int bs, radius, vert;
    int fall_mode;  // 0: x, 1: sqrt, 2: x*x, 3: x*x*x, 4: general powf, 5: ones
    float fall32;
    float* wl; float* wr;  // [n][h][w]
    float* out_l; float* out_r;
    int dbg;  // development only (env CS_DBG)
    // edge bit rows made by k_gray_edges: two planes (x1 / x255 hypothesis) `mask_plane` 64-bit words apart, the frame's
    // ST_SCALE255 word picks; block summaries unscaled with one "any bit" flag per plane.  0: one plane (k_blur_edges4)
    size_t mask_plane;
};

__constant__ csm::PowfTables c_blur_powf_tables = CS_POWF_TABLES_INIT;

__device__ __forceinline__ float falloff_pow(float t, int mode, float e32, const csm::PowfTables* T) {
    switch (mode) {
    case 0: return t;
    case 1: return sqrtf(t);
    case 2: return t * t;
    case 3: return (t * t) * t;
    case 5: return 1.0f;
    default: return csm::powf_exact(t, e32, T);
    }
}

__global__ void __launch_bounds__(1024) k_blur_weights(BlurArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int y = blockIdx.x, frame = blockIdx.y, w = A.w, h = A.h;
    int* ll = (int*)smem;          // last left-edge column <= c   (prefix max)
    int* lr = ll + w;              // next left-edge column >= c   (suffix min)
    int* rl = lr + w;
    int* rr = rl + w;
    int* ws = rr + w;              // scan scratch [32]
    csm::PowfTables* T = (csm::PowfTables*)(ws + 32);
    if (A.fall_mode == 4) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_blur_powf_tables);
        for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += nt) reinterpret_cast<uint32_t*>(T)[i] = src[i];
    }
    const float scale = (A.stats && A.stats[frame * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f;
    const float* d = A.depth + (size_t)frame * h * w;
    const int BIG = 1 << 29;
    for (int x = tid; x < w; x += nt) {
        float g = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            int yy = y + ky - 1;
            bool rowok = yy >= 0 && yy < h;
            const float* r = d + (size_t)(rowok ? yy : 0) * w;
            float vl = (rowok && x > 0) ? r[x - 1] * scale : 0.0f;
            float vr = (rowok && x + 1 < w) ? r[x + 1] * scale : 0.0f;
            float kl = ky == 1 ? -2.0f : -1.0f, kr = ky == 1 ? 2.0f : 1.0f;
            g = fmaf(kl, vl, g);  // (the centre tap has weight 0: fmaf(0, v, g) == g)
            g = fmaf(kr, vr, g);
        }
        float es = fabsf(g) / A.den;
        es = fminf(fmaxf(es, 0.0f), 1.0f);
        bool le = (g > 0.0f) && (es > 0.5f), re = (g < 0.0f) && (es > 0.5f);
        ll[x] = le ? x : -BIG; lr[x] = le ? x : BIG;
        rl[x] = re ? x : -BIG; rr[x] = re ? x : BIG;
    }
    __syncthreads();
    block_scan_inclusive(ll, w, -BIG, OpMax(), ws);
    block_scan_inclusive(lr, w, BIG, OpMin(), ws, true);
    block_scan_inclusive(rl, w, -BIG, OpMax(), ws);
    block_scan_inclusive(rr, w, BIG, OpMin(), ws, true);
    const float large = (float)(A.radius + 1), rad = (float)A.radius;
    float* wl = A.wl + ((size_t)frame * h + y) * w;
    float* wr = A.wr + ((size_t)frame * h + y) * w;
    for (int x = tid; x < w; x += nt) {
        float dl = ll[x] >= 0 ? (float)(x - ll[x]) : large;
        float dr = lr[x] < BIG ? (float)(lr[x] - x) : large;
        float t = 1.0f - fminf(dl, dr) / rad;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        wl[x] = falloff_pow(t, A.fall_mode, A.fall32, T);
        dl = rl[x] >= 0 ? (float)(x - rl[x]) : large;
        dr = rr[x] < BIG ? (float)(rr[x] - x) : large;
        t = 1.0f - fminf(dl, dr) / rad;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        wr[x] = falloff_pow(t, A.fall_mode, A.fall32, T);
    }
}

// pass B, tiled: a workgroup of 256 threads produces a TW x TR tile of both outputs.  The weight
// rows y0-v .. y0+TR-1+v and the depth columns x0-pad .. x0+TW-1+(bs-1-pad) of the tile are staged in
// LDS once (zero outside the frame == the reference's zero padding; fmaf(k, 0, acc) == acc exactly),
// so every weight is fetched from HBM/L2 (TR+2v)/TR times instead of 2v+1 times.
// (-DCS_DEV builds: cs_debug_set(CS_DEBUG_DBG, 21 .. 24) cuts k_blur_fused's tile function short after a phase; a release build
// has no such tests -- four loop-invariant conditions are eight scalar registers in a kernel that spills them)
#ifdef CS_DEV
#define BLUR_DEV_IS(n) (A.dbg == (n))
#else
#define BLUR_DEV_IS(n) false
#endif
#define BLUR_TW 64
#define BLUR_TR 32
__global__ void __launch_bounds__(256) k_blur_apply(BlurArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * BLUR_TW, y0 = blockIdx.y * BLUR_TR, frame = blockIdx.z;
    const int w = A.w, h = A.h, v = A.vert, bs = A.bs, pad = A.bs / 2;
    const int wrows = BLUR_TR + 2 * v, dcols = BLUR_TW + bs - 1;
    float* wlt = (float*)smem;                    // [wrows][TW]
    float* wrt = wlt + wrows * BLUR_TW;           // [wrows][TW]
    float* dt = wrt + wrows * BLUR_TW;            // [TR][dcols]
    const float scale = (A.stats && A.stats[frame * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f;
    const float* wl = A.wl + (size_t)frame * h * w;
    const float* wr = A.wr + (size_t)frame * h * w;
    const float* d = A.depth + (size_t)frame * h * w;
    for (int i = tid; i < wrows * BLUR_TW; i += 256) {
        int r = i / BLUR_TW, c = i - r * BLUR_TW;
        int yy = y0 - v + r, xx = x0 + c;
        bool ok = yy >= 0 && yy < h && xx < w;
        wlt[i] = ok ? wl[(size_t)yy * w + xx] : 0.0f;
        wrt[i] = ok ? wr[(size_t)yy * w + xx] : 0.0f;
    }
    for (int i = tid; i < BLUR_TR * dcols; i += 256) {
        int r = i / dcols, c = i - r * dcols;
        int yy = y0 + r, xx = x0 - pad + c;
        bool ok = yy < h && xx >= 0 && xx < w;
        dt[i] = ok ? d[(size_t)yy * w + xx] * scale : 0.0f;
    }
    __syncthreads();
    const int tx = tid & (BLUR_TW - 1), ty = tid >> 6;
    const float kb = 1.0f / (float)bs, kv = 1.0f / (float)(2 * v + 1);
    float lmin = INFINITY, lmax = -INFINITY, rmin = INFINITY, rmax = -INFINITY;
    const int x = x0 + tx;
    for (int rr = 0; rr < BLUR_TR / 4; rr++) {
        const int r = ty * (BLUR_TR / 4) + rr, y = y0 + r;
        if (y >= h || x >= w) continue;
        float a, b;
        if (v > 0) {
            a = 0.0f; b = 0.0f;
            const float* pl = wlt + r * BLUR_TW + tx;
            const float* pr = wrt + r * BLUR_TW + tx;
            for (int ky = 0; ky < 2 * v + 1; ky++) {
                a = fmaf(kv, pl[ky * BLUR_TW], a);
                b = fmaf(kv, pr[ky * BLUR_TW], b);
            }
        } else {
            a = wlt[r * BLUR_TW + tx];
            b = wrt[r * BLUR_TW + tx];
        }
        const float* pd = dt + r * dcols + tx;
        float acc = 0.0f;
        for (int k = 0; k < bs; k++) acc = fmaf(kb, pd[k], acc);
        float dv = pd[pad];
        float ol = a * acc + (1.0f - a) * dv;
        float orr = b * acc + (1.0f - b) * dv;
        A.out_l[((size_t)frame * h + y) * w + x] = ol;
        A.out_r[((size_t)frame * h + y) * w + x] = orr;
        lmin = fminf(lmin, ol); lmax = fmaxf(lmax, ol);
        rmin = fminf(rmin, orr); rmax = fmaxf(rmax, orr);
    }
    if (A.stats_rw) {
        __shared__ float red[2 * 16];
        uint32_t* st = A.stats_rw + frame * ST_WORDS;
        block_minmax_update(lmin, lmax, &st[ST_L_MIN], &st[ST_L_MAX], red);
        block_minmax_update(rmin, rmax, &st[ST_R_MIN], &st[ST_R_MAX], red);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused single-kernel blur (the default whenever its tiles fit in LDS; the two-pass kernels above
// are the general fallback for very wide kernels).  One 256-thread workgroup produces a 64 x 32
// tile of both outputs from ONE staged depth tile:
//   1. depth tile with halos (rows +-(v+1), cols +-(R+1), zero outside the frame) -> LDS
//   2. Sobel-x + edge tests for every (weight row, column in +-R); the two edge masks of a row are
//      built 64 columns at a time with wave ballots and kept as bit rows in LDS
//   3. nearest-edge distance = clz/ctz on the bit row (only edges closer than R can give a non-zero
//      weight, so a +-R window is exact), weight = clamp(1 - d/R, 0, 1) ** falloff -> LDS
//   4. vertical (2v+1) box of the weights and k-tap box of the depth as fmaf chains (8 independent
//      chains per lane, operands fetched 8 at a time), blend, store, per-frame min/max.
// The weight maps never exist in HBM: traffic per pixel is ~10 B read + 8 B written.
// ---------------------------------------------------------------------------------------------
// Edge masks of the whole frame as bit rows (2 x w/64 words per image row), one wave per 64 columns:
// Sobel-x (6 fmaf, raster order, zero padding), the two edge tests, __ballot.  Streams the depth once.
#define BLUR_ER 4  // image rows per thread of k_blur_edges (sliding 3-row window: 2 x (ER + 2) loads for ER results)
__global__ void __launch_bounds__(256) k_blur_edges(BlurArgs A, unsigned long long* mask_l, unsigned long long* mask_r,
                                                    int MW) {
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 256 + threadIdx.x, yb = blockIdx.y * BLUR_ER, frame = blockIdx.z;
    const int w = A.w, h = A.h;
    const float scale = (A.stats && A.stats[frame * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f;
    const float* d = A.depth + (size_t)frame * h * w;
    float vl[BLUR_ER + 2], vr[BLUR_ER + 2];
#pragma unroll
    for (int i = 0; i < BLUR_ER + 2; i++) {
        const int yy = yb - 1 + i;
        const bool rowok = yy >= 0 && yy < h && x < w;
        const float* r = d + (size_t)(yy >= 0 && yy < h ? yy : 0) * w;
        vl[i] = (rowok && x > 0) ? r[x - 1] * scale : 0.0f;
        vr[i] = (rowok && x + 1 < w) ? r[x + 1] * scale : 0.0f;
    }
    const int word = x >> 6;  // wave-uniform
#pragma unroll
    for (int j = 0; j < BLUR_ER; j++) {
        const int y = yb + j;
        // rows y-1, y, y+1 in raster order: (-1, +1), (-2, +2), (-1, +1); the centre taps have weight 0
        float g = 0.0f;
        g = fmaf(-1.0f, vl[j], g); g = fmaf(1.0f, vr[j], g);
        g = fmaf(-2.0f, vl[j + 1], g); g = fmaf(2.0f, vr[j + 1], g);
        g = fmaf(-1.0f, vl[j + 2], g); g = fmaf(1.0f, vr[j + 2], g);
        const float es = fminf(fmaxf(fabsf(g) / A.den, 0.0f), 1.0f);
        const bool le = x < w && (g > 0.0f) && (es > 0.5f);
        const bool re = x < w && (g < 0.0f) && (es > 0.5f);
        const unsigned long long bl = __ballot(le), br = __ballot(re);
        if (lane == 0 && word < MW && y < h) {
            mask_l[((size_t)frame * h + y) * MW + word] = bl;
            mask_r[((size_t)frame * h + y) * MW + word] = br;
        }
    }
}

// The same for widths that are multiples of 4 (every real frame): FOUR columns per lane, loaded as one float4 per image
// row (8x fewer load instructions -- the scalar version is bound by the L1 request rate, not by HBM); the two missing
// neighbours come from the adjacent lanes (shuffles; the first / last lane of a wave loads them).  A lane's four edge bits
// form a nibble, eight lanes' nibbles a 32-bit half word (OR over xor-shuffles), two halves a word of the bit row.
#define BLUR_ER4 4  // image rows per thread (8: 0.66 ms, 16: 0.85, 2: 0.71 -- the halo rows come from L2; fewer registers, more waves)
// cross-lane moves as DPP modifiers (2-4 cycles each) instead of ds_bpermute round trips through the LDS crossbar (the
// kernel was bound by those: 80 per thread).  Lanes without a source keep their own value.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, dpp_u32<CTRL>(__builtin_bit_cast(unsigned, v))); }
enum { DPP_XOR1 = 0xB1 /* quad_perm [1,0,3,2] */, DPP_XOR2 = 0x4E /* quad_perm [2,3,0,1] */, DPP_HALF_MIRROR = 0x141 /* i <-> 7 - i */,
       DPP_ROW_MIRROR = 0x140 /* i <-> 15 - i */, DPP_ROW_SHL8 = 0x108 /* i <- i + 8 (row of 16) */, DPP_WAVE_SHL1 = 0x130 /* i <- i + 1 */,
       DPP_WAVE_SHR1 = 0x138 /* i <- i - 1 */ };
// `blk` (lazy mode, or null): per 64-column x BLUR_ER4-row block {min, max of the scaled depth, any edge bit, -} for k_blur_classify
__global__ void __launch_bounds__(256) k_blur_edges4(BlurArgs A, unsigned long long* mask_l, unsigned long long* mask_r, int MW,
                                                     float4* blk) {
    const int lane = threadIdx.x & 63;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, yb = blockIdx.y * BLUR_ER4, frame = blockIdx.z;
    const int w = A.w, h = A.h;
    const float scale = (A.stats && A.stats[frame * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f;
    const float* d = A.depth + (size_t)frame * h * w;
    float4 v[BLUR_ER4 + 2];
    float nl[BLUR_ER4 + 2], nr[BLUR_ER4 + 2];
#pragma unroll
    for (int i = 0; i < BLUR_ER4 + 2; i++) {
        const int yy = yb - 1 + i;
        const bool rowok = yy >= 0 && yy < h;
        const float* r = d + (size_t)(rowok ? yy : 0) * w;
        v[i] = (rowok && x < w) ? *reinterpret_cast<const float4*>(r + x) : make_float4(0.f, 0.f, 0.f, 0.f);
        // columns x - 1 and x + 4: only the wave's first / last lane has to load them (zero padding outside the frame)
        nl[i] = (lane == 0 && rowok && x > 0 && x < w) ? r[x - 1] : 0.0f;
        nr[i] = (lane == 63 && rowok && x + 4 < w) ? r[x + 4] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < BLUR_ER4 + 2; i++) {
        v[i].x *= scale; v[i].y *= scale; v[i].z *= scale; v[i].w *= scale;
        const float fl = dpp_f32<DPP_WAVE_SHR1>(v[i].w), fr = dpp_f32<DPP_WAVE_SHL1>(v[i].x);
        nl[i] = lane == 0 ? nl[i] * scale : fl;
        nr[i] = lane == 63 ? nr[i] * scale : fr;
        if (x + 4 >= w) nr[i] = 0.0f;  // (the neighbour lane holds zeros anyway; explicit for the frame's last column)
    }
    const int word = x >> 6;
    unsigned long long anybits = 0ull;
#pragma unroll
    for (int j = 0; j < BLUR_ER4; j++) {
        const int y = yb + j;
        unsigned nib_l = 0, nib_r = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // rows y-1, y, y+1 in raster order: (-1, +1), (-2, +2), (-1, +1); the centre taps have weight 0
            float L[3], R[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const float4 q = v[j + t];
                L[t] = c == 0 ? nl[j + t] : (c == 1 ? q.x : (c == 2 ? q.y : q.z));
                R[t] = c == 0 ? q.y : (c == 1 ? q.z : (c == 2 ? q.w : nr[j + t]));
            }
            float g = 0.0f;
            g = fmaf(-1.0f, L[0], g); g = fmaf(1.0f, R[0], g);
            g = fmaf(-2.0f, L[1], g); g = fmaf(2.0f, R[1], g);
            g = fmaf(-1.0f, L[2], g); g = fmaf(1.0f, R[2], g);
            const float es = fminf(fmaxf(fabsf(g) / A.den, 0.0f), 1.0f);
            const bool in = x + c < w;
            nib_l |= (in && (g > 0.0f) && (es > 0.5f)) ? 1u << c : 0u;
            nib_r |= (in && (g < 0.0f) && (es > 0.5f)) ? 1u << c : 0u;
        }
        unsigned hl = nib_l << (4 * (lane & 7)), hr = nib_r << (4 * (lane & 7));
        // OR over the eight lanes of a half row: xor 1, xor 2, then the two quads swapped (i <-> 7 - i)
        hl |= dpp_u32<DPP_XOR1>(hl); hr |= dpp_u32<DPP_XOR1>(hr);
        hl |= dpp_u32<DPP_XOR2>(hl); hr |= dpp_u32<DPP_XOR2>(hr);
        hl |= dpp_u32<DPP_HALF_MIRROR>(hl); hr |= dpp_u32<DPP_HALF_MIRROR>(hr);
        const unsigned hl_hi = dpp_u32<DPP_ROW_SHL8>(hl), hr_hi = dpp_u32<DPP_ROW_SHL8>(hr);
        if ((lane & 15) == 0 && word < MW && y < h) {
            mask_l[((size_t)frame * h + y) * MW + word] = (unsigned long long)hl | ((unsigned long long)hl_hi << 32);
            mask_r[((size_t)frame * h + y) * MW + word] = (unsigned long long)hr | ((unsigned long long)hr_hi << 32);
            anybits |= (unsigned long long)(hl | hr) | ((unsigned long long)(hl_hi | hr_hi) << 32);
        }
    }
    if (blk) {
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < BLUR_ER4; j++) {
            if (yb + j < h && x < w) {   // (w % 4 == 0: a lane's four columns are inside together)
                const float4 q = v[j + 1];
                mn = fminf(fminf(mn, fminf(q.x, q.y)), fminf(q.z, q.w));
                mx = fmaxf(fmaxf(mx, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
            }
        }
        mn = fminf(mn, dpp_f32<DPP_XOR1>(mn)); mx = fmaxf(mx, dpp_f32<DPP_XOR1>(mx));
        mn = fminf(mn, dpp_f32<DPP_XOR2>(mn)); mx = fmaxf(mx, dpp_f32<DPP_XOR2>(mx));
        mn = fminf(mn, dpp_f32<DPP_HALF_MIRROR>(mn)); mx = fmaxf(mx, dpp_f32<DPP_HALF_MIRROR>(mx));
        mn = fminf(mn, dpp_f32<DPP_ROW_MIRROR>(mn)); mx = fmaxf(mx, dpp_f32<DPP_ROW_MIRROR>(mx));
        if ((lane & 15) == 0 && word < MW && yb < h)
            blk[((size_t)frame * gridDim.y + blockIdx.y) * MW + word] = make_float4(mn, mx, anybits ? 1.0f : 0.0f, 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------
// k_gray_edges: the gray conversion of an RGB depth map (cs_abi.hip k_gray: GenerateStereo.py:134-139) and the edge bit rows
// of k_blur_edges4 in ONE pass over the depth input -- the gray depth is not read back for the Sobel (4 B/px less HBM
// traffic, one launch less: 0.58 ms per 64 4K frames).  The x255 decision of the frame (reference :1475, :1045) needs the
// frame's maximum, which only exists when this kernel has finished, so the bit rows are built for BOTH hypotheses (plane 0:
// scale 1, plane 1: scale 255; the Sobel of the scaled values is not the scaled Sobel in float32) and the consumers pick the
// plane by the frame's ST_SCALE255 word.  One wave per 256 columns x GE_RB rows, marching down in blocks of BLUR_ER4 rows
// with a two-row overlap; four columns per lane (three 16-byte loads per row), neighbours through DPP; the block summaries
// hold the UNSCALED min / max (x -> fl(x * s) is monotone: the scaled extremes are the extremes scaled) and one
// "any edge bit" flag per hypothesis.
// `thr`: the largest float t with fl(t / den) <= 0.5, so that  clamp(|g| / den, 0, 1) > 0.5  <=>  |g| > thr  (the division
// is monotone in |g|; NaN fails both); computed on the host by blur_edge_threshold().
// ---------------------------------------------------------------------------------------------
#ifndef GE_RB
#define GE_RB 32
#endif
#ifndef GE_WAVES
#define GE_WAVES 0   // (development: > 0 = amdgpu_waves_per_eu)
#endif
static_assert(GE_RB % BLUR_ER4 == 0, "strips are whole summary blocks");
#if GE_WAVES > 0
__attribute__((amdgpu_waves_per_eu(GE_WAVES)))
#endif
__global__ void __launch_bounds__(64) k_gray_edges(const float* __restrict__ rgb, float* __restrict__ gray, int h, int w,
                                                   uint32_t* stats, float thr, unsigned long long* mask_l,
                                                   unsigned long long* mask_r, size_t plane, int MW, float4* blk, int HB) {
    const int lane = threadIdx.x;
    const int x = (blockIdx.x * 64 + lane) * 4, yb0 = blockIdx.y * GE_RB, frame = blockIdx.z;
    const float* src = rgb + (size_t)frame * h * w * 3;
    float* dst = gray + (size_t)frame * h * w;
    const bool incol = x < w;   // (w % 4 == 0: a lane's four columns are inside together)
    // the one neighbour column a wave has to load itself: x - 1 for its first lane, x + 4 for its last
    const int xn = lane == 0 ? x - 1 : x + 4;
    const bool nb_lane = (lane == 0 && x > 0 && incol) || (lane == 63 && x + 4 < w);
    struct Row { float4 v; float nb, lo, hi; };   // lo / hi: extremes of everything the row contributes to this lane's Sobel taps
    const bool pad0 = x == 0 || x + 4 >= w;   // a zero-padded neighbour column
    auto gray_of = [](float r, float g, float b) { return (0.2989f * r + 0.5870f * g) + 0.1140f * b; };
    auto load_row = [&](int yy, float4& a, float4& b, float4& c, float& n0, float& n1, float& n2) {
        // (loads from a clamped, always valid address; the VALUES are selected: a conditional load of a float4 becomes a
        // load through a selected pointer with the zero vector in scratch memory)
        const bool rowok = yy >= 0 && yy < h;
        const float* r = src + ((size_t)(rowok ? yy : 0) * w) * 3;
        const float4* q = reinterpret_cast<const float4*>(r + (size_t)(incol ? x : 0) * 3);
#ifdef GE_NT_LOAD   // (experiment, round 5: the RGB depth is read exactly once -- nontemporal loads)
        typedef float ge_v4 __attribute__((ext_vector_type(4)));
        const ge_v4* qv = reinterpret_cast<const ge_v4*>(q);
        const ge_v4 va = __builtin_nontemporal_load(qv), vb = __builtin_nontemporal_load(qv + 1), vc = __builtin_nontemporal_load(qv + 2);
        const float4 ta = make_float4(va.x, va.y, va.z, va.w), tb = make_float4(vb.x, vb.y, vb.z, vb.w), tc = make_float4(vc.x, vc.y, vc.z, vc.w);
#else
        const float4 ta = q[0], tb = q[1], tc = q[2];
#endif
        const bool ok = rowok && incol;
        a = make_float4(ok ? ta.x : 0.f, ok ? ta.y : 0.f, ok ? ta.z : 0.f, ok ? ta.w : 0.f);
        b = make_float4(ok ? tb.x : 0.f, ok ? tb.y : 0.f, ok ? tb.z : 0.f, ok ? tb.w : 0.f);
        c = make_float4(ok ? tc.x : 0.f, ok ? tc.y : 0.f, ok ? tc.z : 0.f, ok ? tc.w : 0.f);
        n0 = n1 = n2 = 0.0f;
        if (rowok && nb_lane) { const float* pn = r + (size_t)xn * 3; n0 = pn[0]; n1 = pn[1]; n2 = pn[2]; }
    };
    auto to_row = [&](const float4& a, const float4& b, const float4& c, float n0, float n1, float n2) {
        Row R;
        R.v.x = gray_of(a.x, a.y, a.z); R.v.y = gray_of(a.w, b.x, b.y); R.v.z = gray_of(b.z, b.w, c.x); R.v.w = gray_of(c.y, c.z, c.w);
        R.nb = gray_of(n0, n1, n2);   // (zeros in, zero out)
        const float e = nb_lane ? R.nb : (pad0 ? 0.0f : R.v.x);
        R.lo = fminf(fminf(fminf(R.v.x, R.v.y), fminf(R.v.z, R.v.w)), e);
        R.hi = fmaxf(fmaxf(fmaxf(R.v.x, R.v.y), fmaxf(R.v.z, R.v.w)), e);
        return R;
    };
    Row G[BLUR_ER4 + 2];   // rows y0 - 1 .. y0 + ER4 of the current block (gray, unscaled; zero outside the frame)
    {
        float4 a[2], b[2], c[2]; float n0[2], n1[2], n2[2];
#pragma unroll
        for (int i = 0; i < 2; i++) load_row(yb0 - 1 + i, a[i], b[i], c[i], n0[i], n1[i], n2[i]);
#pragma unroll
        for (int i = 0; i < 2; i++) G[BLUR_ER4 + i] = to_row(a[i], b[i], c[i], n0[i], n1[i], n2[i]);
    }
    float mn_f = INFINITY, mx_f = -INFINITY;
    const int word = x >> 6;
    // the RGB rows of the NEXT block are requested before the current block is worked on (the wave's own loads stay in flight
    // under its Sobel and its stores: with 2-3 waves per SIMD the other waves alone do not cover the memory latency)
    float4 pa[BLUR_ER4], pb[BLUR_ER4], pc[BLUR_ER4]; float p0[BLUR_ER4], p1[BLUR_ER4], p2[BLUR_ER4];
#pragma unroll
    for (int i = 0; i < BLUR_ER4; i++) load_row(yb0 + 1 + i, pa[i], pb[i], pc[i], p0[i], p1[i], p2[i]);
    for (int y0 = yb0; y0 < yb0 + GE_RB && y0 < h; y0 += BLUR_ER4) {
        G[0] = G[BLUR_ER4]; G[1] = G[BLUR_ER4 + 1];
#pragma unroll
        for (int i = 0; i < BLUR_ER4; i++) G[2 + i] = to_row(pa[i], pb[i], pc[i], p0[i], p1[i], p2[i]);
        if (y0 + BLUR_ER4 < yb0 + GE_RB && y0 + BLUR_ER4 < h) {
#pragma unroll
            for (int i = 0; i < BLUR_ER4; i++) load_row(y0 + BLUR_ER4 + 1 + i, pa[i], pb[i], pc[i], p0[i], p1[i], p2[i]);
        }
        // the block's own rows are G[1 .. ER4]: store the gray depth, min / max
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < BLUR_ER4; j++) {
            const int y = y0 + j;
            if (y < h && incol) {
                const float4 q = G[1 + j].v;
#ifndef GE_PLAIN_STORE   // (round 5: the gray map is not read back by this kernel -- nontemporal stores, -5 % on the kernel)
                typedef float ge_v4s __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(ge_v4s{q.x, q.y, q.z, q.w}, reinterpret_cast<ge_v4s*>(dst + (size_t)y * w + x));
#else
                *reinterpret_cast<float4*>(dst + (size_t)y * w + x) = q;
#endif
                mn = fminf(fminf(mn, fminf(q.x, q.y)), fminf(q.z, q.w));
                mx = fmaxf(fmaxf(mx, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
            }
        }
        mn_f = fminf(mn_f, mn); mx_f = fmaxf(mx_f, mx);
        unsigned any[2] = {0u, 0u};
        // |g| <= 4 s (max - min) + rounding (< 32 ulp of s max|v|) over the six rows of the window, neighbours and zero padding
        // included: where that stays below the threshold no pixel of the block is an edge, and the Sobel is skipped (flat
        // regions -- most of a depth map; with depth in 0..1 the whole x1 plane).  Wave-uniform.
        float wlo = G[0].lo, whi = G[0].hi;
#pragma unroll
        for (int i = 1; i < BLUR_ER4 + 2; i++) { wlo = fminf(wlo, G[i].lo); whi = fmaxf(whi, G[i].hi); }
        wlo = wave_min(wlo); whi = wave_max(whi);
        const float span = 4.0f * (whi - wlo) * 1.0001f + 1e-5f * fmaxf(fabsf(wlo), fabsf(whi));
#pragma unroll
        for (int hyp = 0; hyp < 2; hyp++) {
            const float scale = hyp ? 255.0f : 1.0f;
            if (span * scale < thr) {   // (NaN / infinite spans compute)
                if ((lane & 15) == 0 && word < MW) {
#pragma unroll
                    for (int j = 0; j < BLUR_ER4; j++) {
                        if (y0 + j < h) {
                            const size_t o = (size_t)hyp * plane + ((size_t)frame * h + y0 + j) * MW + word;
                            mask_l[o] = 0ull; mask_r[o] = 0ull;
                        }
                    }
                }
                continue;
            }
            // the six rows scaled, with both neighbours (k_blur_edges4's v / nl / nr)
            float4 v[BLUR_ER4 + 2]; float nl[BLUR_ER4 + 2], nr[BLUR_ER4 + 2];
#pragma unroll
            for (int i = 0; i < BLUR_ER4 + 2; i++) {
                v[i] = make_float4(G[i].v.x * scale, G[i].v.y * scale, G[i].v.z * scale, G[i].v.w * scale);
                const float nbs = G[i].nb * scale;
                const float fl = dpp_f32<DPP_WAVE_SHR1>(v[i].w), fr = dpp_f32<DPP_WAVE_SHL1>(v[i].x);
                nl[i] = lane == 0 ? nbs : fl;
                nr[i] = lane == 63 ? nbs : fr;
                if (x + 4 >= w) nr[i] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < BLUR_ER4; j++) {
                const int y = y0 + j;
                unsigned nib_l = 0, nib_r = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float L[3], R[3];
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        const float4 q = v[j + t];
                        L[t] = c == 0 ? nl[j + t] : (c == 1 ? q.x : (c == 2 ? q.y : q.z));
                        R[t] = c == 0 ? q.y : (c == 1 ? q.z : (c == 2 ? q.w : nr[j + t]));
                    }
                    float g = 0.0f;
                    g = fmaf(-1.0f, L[0], g); g = fmaf(1.0f, R[0], g);
                    g = fmaf(-2.0f, L[1], g); g = fmaf(2.0f, R[1], g);
                    g = fmaf(-1.0f, L[2], g); g = fmaf(1.0f, R[2], g);
                    // (g > 0 and |g| > thr) == g > thr for thr >= 0; likewise on the negative side
                    nib_l |= (incol && g > thr) ? 1u << c : 0u;
                    nib_r |= (incol && g < -thr) ? 1u << c : 0u;
                }
                unsigned hl = nib_l << (4 * (lane & 7)), hr = nib_r << (4 * (lane & 7));
                hl |= dpp_u32<DPP_XOR1>(hl); hr |= dpp_u32<DPP_XOR1>(hr);
                hl |= dpp_u32<DPP_XOR2>(hl); hr |= dpp_u32<DPP_XOR2>(hr);
                hl |= dpp_u32<DPP_HALF_MIRROR>(hl); hr |= dpp_u32<DPP_HALF_MIRROR>(hr);
                const unsigned hl_hi = dpp_u32<DPP_ROW_SHL8>(hl), hr_hi = dpp_u32<DPP_ROW_SHL8>(hr);
                if ((lane & 15) == 0 && word < MW && y < h) {
                    const size_t o = (size_t)hyp * plane + ((size_t)frame * h + y) * MW + word;
                    mask_l[o] = (unsigned long long)hl | ((unsigned long long)hl_hi << 32);
                    mask_r[o] = (unsigned long long)hr | ((unsigned long long)hr_hi << 32);
                    any[hyp] |= hl | hr | hl_hi | hr_hi;
                }
            }
        }
        mn = fminf(mn, dpp_f32<DPP_XOR1>(mn)); mx = fmaxf(mx, dpp_f32<DPP_XOR1>(mx));
        mn = fminf(mn, dpp_f32<DPP_XOR2>(mn)); mx = fmaxf(mx, dpp_f32<DPP_XOR2>(mx));
        mn = fminf(mn, dpp_f32<DPP_HALF_MIRROR>(mn)); mx = fmaxf(mx, dpp_f32<DPP_HALF_MIRROR>(mx));
        mn = fminf(mn, dpp_f32<DPP_ROW_MIRROR>(mn)); mx = fmaxf(mx, dpp_f32<DPP_ROW_MIRROR>(mx));
        if ((lane & 15) == 0 && word < MW)
            blk[((size_t)frame * HB + y0 / BLUR_ER4) * MW + word] = make_float4(mn, mx, any[0] ? 1.0f : 0.0f, any[1] ? 1.0f : 0.0f);
    }
    mn_f = wave_min(mn_f); mx_f = wave_max(mx_f);
    if (lane == 0) {
        uint32_t* st_min = &stats[frame * ST_WORDS + ST_GRAY_MIN];
        uint32_t* st_max = &stats[frame * ST_WORDS + ST_GRAY_MAX];
        const uint32_t kmn = csm::f2ord(mn_f), kmx = csm::f2ord(mx_f);
        if (kmn < __hip_atomic_load(st_min, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(st_min, kmn);
        if (kmx > __hip_atomic_load(st_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(st_max, kmx);
    }
}

// 64 bits of a frame-wide bit row starting at (possibly negative / out of range) bit position `fb`
__device__ __forceinline__ unsigned long long mask_window(const unsigned long long* row, int MW, int fb) {
    if (fb <= -64) return 0ull;
    if (fb < 0) return row[0] << (-fb);
    const int wi = fb >> 6, sh = fb & 63;
    const unsigned long long lo = wi < MW ? row[wi] : 0ull, hi = wi + 1 < MW ? row[wi + 1] : 0ull;
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

__device__ __forceinline__ int mask_dist_left(const unsigned long long* m, int p) {
    // distance from bit p to the nearest set bit at or below p, or -1
    int wi = p >> 6;
    unsigned long long cur = m[wi] & (~0ull >> (63 - (p & 63)));
    while (true) {
        if (cur) return p - (wi * 64 + 63 - __clzll((long long)cur));
        if (--wi < 0) return -1;
        cur = m[wi];
    }
}
__device__ __forceinline__ int mask_dist_right(const unsigned long long* m, int nw, int p) {
    int wi = p >> 6;
    unsigned long long cur = m[wi] & (~0ull << (p & 63));
    while (true) {
        if (cur) return wi * 64 + __ffsll((long long)cur) - 1 - p;
        if (++wi >= nw) return -1;
        cur = m[wi];
    }
}

// Edges are sparse, so most 64 x 32 tiles have no edge bit within reach (+-v rows, +-R columns): every weight there is
// exactly 0 and 0*blur + (1-0)*depth == depth -- the tile is a scaled copy.  k_blur_copy walks the frame in 256 x 32
// pieces (four tiles) with no LDS tile and few registers (high occupancy): the depth is loaded with 16-byte accesses
// TOGETHER with the tiles' windows of the edge bit rows (one memory round trip), edge-free tiles are stored straight
// from the registers, and the others are appended to `worklist` for k_blur_fused.
#define BLUR_CW 4  // tiles per k_blur_copy workgroup
__global__ void __launch_bounds__(256) k_blur_copy(BlurArgs A, const unsigned long long* mask_l, const unsigned long long* mask_r,
                                                   int MW, uint32_t* work_count, uint32_t* worklist) {
    const int tid = threadIdx.x;
    const int X0 = blockIdx.x * (BLUR_CW * BLUR_TW), y0 = blockIdx.y * BLUR_TR, frame = blockIdx.z;
    const int w = A.w, h = A.h, v = A.vert, R = A.radius;
    __shared__ float red[2 * 16];
    constexpr int C4 = BLUR_CW * BLUR_TW / 4;     // float4 columns of the piece
    constexpr int RPT = BLUR_TR * C4 / 256;       // rows per thread
    const int c4 = tid % C4, rb = tid / C4;       // float4 column, first row (rows rb, rb + 256 / C4, ..)
    const int x = X0 + 4 * c4;
    float4 vv[RPT];
#pragma unroll
    for (int i = 0; i < RPT; i++) {
        const int y = y0 + rb + (256 / C4) * i;
        vv[i] = (y < h && x < w) ? *reinterpret_cast<const float4*>(A.depth + ((size_t)frame * h + y) * w + x)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the tiles' windows of the edge bit rows: rows y0-v .., bits x0-R .. x0+TW+R-1 (exactly k_blur_fused's reach).
    // No barrier between the two groups of loads: they travel together.
    // Every word of the bit rows that overlaps the piece's reach is loaded ONCE (the two masks OR-ed) and tested against
    // the reach of each of the four tiles -- 2.7x fewer loads than a window per tile.
    const int WR = BLUR_TR + 2 * v;
    const int WB = ((BLUR_CW * BLUR_TW + 2 * R + 63) >> 6) + 1;   // words that can overlap [X0 - R, X0 + 256 + R)
    const int w0 = (X0 - R) >> 6;                                   // (arithmetic shift: floor for the negative start)
    unsigned found = 0;  // bit t: an edge within reach of tile t seen by this thread
    for (int item = tid; item < WR * WB; item += 256) {
        const int r = item / WB, k = item - r * WB;
        const int yy = y0 - v + r, wi = w0 + k;
        if (yy >= 0 && yy < h && wi >= 0 && wi < MW) {
            const size_t ro = ((size_t)frame * h + yy) * MW + wi;
            const unsigned long long bits = mask_l[ro] | mask_r[ro];
            if (bits) {
                const int b0 = wi * 64;   // frame column of bit 0
#pragma unroll
                for (int t = 0; t < BLUR_CW; t++) {
                    const int x0 = X0 + t * BLUR_TW;
                    const int lo = max(x0 - R, b0), hi = min(x0 + BLUR_TW - 1 + R, b0 + 63);
                    if (x0 < w && lo <= hi && (bits & (~0ull >> (63 - (hi - b0))) & (~0ull << (lo - b0)))) found |= 1u << t;
                }
            }
        }
    }
    unsigned any = 0;
#pragma unroll
    for (int t = 0; t < BLUR_CW; t++) any |= __syncthreads_or((found >> t) & 1u) ? 1u << t : 0u;
    if (A.fall_mode == 5) any = (1u << BLUR_CW) - 1;
    const int mytile = c4 / (BLUR_TW / 4);
    const bool edge = (any >> mytile) & 1u;
    if (tid < BLUR_CW && X0 + tid * BLUR_TW < w && ((any >> tid) & 1u))
        worklist[atomicAdd(work_count, 1u)] = ((uint32_t)frame << 20) | ((uint32_t)blockIdx.y << 10) | (uint32_t)(blockIdx.x * BLUR_CW + tid);
    const float scale = (A.stats && A.stats[frame * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f;
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < RPT; i++) {
        const int y = y0 + rb + (256 / C4) * i;
        if (!edge && y < h && x < w) {
            float4 o = make_float4(vv[i].x * scale, vv[i].y * scale, vv[i].z * scale, vv[i].w * scale);
            const size_t off = ((size_t)frame * h + y) * w + x;
            *reinterpret_cast<float4*>(A.out_l + off) = o;
            *reinterpret_cast<float4*>(A.out_r + off) = o;
            mn = fminf(fminf(mn, fminf(o.x, o.y)), fminf(o.z, o.w));
            mx = fmaxf(fmaxf(mx, fmaxf(o.x, o.y)), fmaxf(o.z, o.w));
        }
    }
    if (A.stats_rw) {
        uint32_t* st = A.stats_rw + frame * ST_WORDS;
        block_minmax_update(mn, mx, &st[ST_L_MIN], &st[ST_L_MAX], red);
        block_minmax_update(mn, mx, &st[ST_R_MIN], &st[ST_R_MAX], red);
    }
}

// Lazy mode (the consumer reads edge-free tiles from the gray depth itself, RowArgs::tilemap): nothing is copied, so the
// classification works from k_blur_edges4's block summaries alone -- one thread per tile: edge bits within reach (exact
// test on the bit rows, only inside blocks that have bits) -> worklist + tile map; otherwise the tile's min / max joins the
// frame's output statistics.  No pass over the depth.
__global__ void __launch_bounds__(256) k_blur_classify(BlurArgs A, const unsigned long long* mask_l, const unsigned long long* mask_r,
                                                       int MW, const float4* blk, int HB, uint32_t* work_count, uint32_t* worklist,
                                                       uint32_t* tilemap, int tm_words) {
    __shared__ float red[2 * 16];
    const int w = A.w, h = A.h, v = A.vert, R = A.radius, frame = blockIdx.y;
    const int gx = (w + BLUR_TW - 1) / BLUR_TW, gy = (h + BLUR_TR - 1) / BLUR_TR;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool live = t < gx * gy;
    const int ty = live ? t / gx : 0, tx = live ? t - ty * gx : 0;
    const int x0 = tx * BLUR_TW, y0 = ty * BLUR_TR;
    const int ya = max(y0 - v, 0), yz = min(y0 + BLUR_TR - 1 + v, h - 1);          // rows within reach
    const int xa = max(x0 - R, 0), xz = min(x0 + BLUR_TW - 1 + R, w - 1);          // columns within reach
    bool edge = live && A.fall_mode == 5;
    const float4* fb = blk + (size_t)frame * HB * MW;
    const bool second = A.mask_plane && A.stats && A.stats[frame * ST_WORDS + ST_SCALE255];
    const float bscale = second ? 255.0f : 1.0f;   // (two planes: the summaries are unscaled)
    mask_l += second ? A.mask_plane : 0; mask_r += second ? A.mask_plane : 0;
    // (round 5) first the block summaries within reach, all of them, with nothing between the loads that depends on a loaded value
    // (twelve 16-byte loads in flight per lane): four tiles in five have no edge bit in any of them and are done.  The exact test on
    // the bit rows below, with its early exits, used to run for every tile -- up to 33 dependent round trips per lane.
    bool cand = live && !edge;
    float pmn = INFINITY, pmx = -INFINITY;
    unsigned long long cmask = 0ull;   // candidate blocks: bit 3 (b - b0) + k <-> block row b, word wa0 + k has edge bits
    const bool compact = (xz >> 6) - (xa >> 6) <= 2 && yz / BLUR_ER4 - ya / BLUR_ER4 < 21;
    if (cand && compact) {
        const int b0 = ya / BLUR_ER4, b1 = yz / BLUR_ER4, wa0 = xa >> 6, wz0 = xz >> 6;
        for (int bb = b0; bb <= b1; bb += 4) {
            float4 sm[4][3];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k < 3; k++) sm[i][k] = fb[(size_t)min(bb + i, b1) * MW + min(wa0 + k, wz0)];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k < 3; k++)
                    if (bb + i <= b1 && wa0 + k <= wz0 && (second ? sm[i][k].w : sm[i][k].z) != 0.0f) cmask |= 1ull << (3 * (bb + i - b0) + k);
            // (the tile's own blocks are among them: its extremes for the edge-free case, below)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k < 3; k++)
                    if (wa0 + k == tx && bb + i >= y0 / BLUR_ER4 && bb + i <= min(y0 + BLUR_TR - 1, h - 1) / BLUR_ER4) {
                        pmn = fminf(pmn, sm[i][k].x * bscale); pmx = fmaxf(pmx, sm[i][k].y * bscale);
                    }
        }
        // the exact test on the bit rows of the candidate blocks, one block per round trip (its rows' words of both eyes are loaded
        // together; the first version walked them one dependent load at a time, and the slowest lane sets the kernel's time)
        while (cmask && !edge) {
            const int bit = __ffsll((long long)cmask) - 1;
            cmask &= cmask - 1ull;
            const int b = b0 + bit / 3, wi = wa0 + bit % 3;
            const int lo = max(xa, wi * 64) - wi * 64, hi = min(xz, wi * 64 + 63) - wi * 64;
            const unsigned long long colmask = (~0ull >> (63 - hi)) & (~0ull << lo);
            unsigned long long acc = 0ull;
#pragma unroll
            for (int j = 0; j < BLUR_ER4; j++) {
                const int y = b * BLUR_ER4 + j;
                const bool in = y >= ya && y <= yz;
                const size_t ro = ((size_t)frame * h + (in ? y : ya)) * MW + wi;
                const unsigned long long bits = mask_l[ro] | mask_r[ro];
                acc |= in ? bits : 0ull;
            }
            if (acc & colmask) edge = true;
        }
        cand = false;
    }
    if (cand) {   // (mask radii beyond 64 columns / more than 80 rows of vertical reach: the plain walk)
        for (int b = ya / BLUR_ER4; b <= yz / BLUR_ER4 && !edge; b++)
            for (int wi = xa >> 6; wi <= (xz >> 6) && !edge; wi++) {
                const float4 sm = fb[(size_t)b * MW + wi];
                if ((second ? sm.w : sm.z) == 0.0f) continue;
                const int lo = max(xa, wi * 64) - wi * 64, hi = min(xz, wi * 64 + 63) - wi * 64;
                const unsigned long long colmask = (~0ull >> (63 - hi)) & (~0ull << lo);
                for (int y = max(ya, b * BLUR_ER4); y <= min(yz, b * BLUR_ER4 + BLUR_ER4 - 1); y++) {
                    const size_t ro = ((size_t)frame * h + y) * MW + wi;
                    if ((mask_l[ro] | mask_r[ro]) & colmask) { edge = true; break; }
                }
            }
    }
    float mn = INFINITY, mx = -INFINITY;
    // (round 5) one atomicAdd per WAVE on the list counter: 56 000 tiles of the metric workload queueing on one address were most of
    // this kernel's 0.12 ms
    const unsigned long long em = __ballot(live && edge);
    uint32_t slot = 0;
    if (em) {
        const int leader = __ffsll((long long)em) - 1, lane = threadIdx.x & 63;
        if (lane == leader) slot = atomicAdd(work_count, (uint32_t)__popcll(em));
        slot = __shfl(slot, leader) + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
    }
    if (live && edge) {
        worklist[slot] = ((uint32_t)frame << 20) | ((uint32_t)ty << 10) | (uint32_t)tx;
        atomicOr(&tilemap[((size_t)frame * gy + ty) * tm_words + (tx >> 5)], 1u << (tx & 31));
    } else if (live && compact) {
        mn = pmn; mx = pmx;
    } else if (live) {
        for (int b = y0 / BLUR_ER4; b <= min(y0 + BLUR_TR - 1, h - 1) / BLUR_ER4; b++) {
            const float4 q = fb[(size_t)b * MW + tx];
            mn = fminf(mn, q.x * bscale); mx = fmaxf(mx, q.y * bscale);
        }
    }
    if (A.stats_rw) {
        uint32_t* st = A.stats_rw + frame * ST_WORDS;
        block_minmax_update(mn, mx, &st[ST_L_MIN], &st[ST_L_MAX], red);
        block_minmax_update(mn, mx, &st[ST_R_MIN], &st[ST_R_MAX], red);
    }
}

// k_blur_fused: persistent workgroups over the tiles of `worklist` (work_count entries, packed frame | tile row | tile
// column); worklist == nullptr: plain grid over all tiles (blockIdx = tile).
// `tstat` (worklist form): the tile's output extremes {min L, max L, min R, max R} per wave, 4 x float4 per worklist entry --
// k_blur_tile_stats folds them into the frame statistics afterwards (two workgroup reductions with four barriers and the
// atomic pre-checks per tile were 10 % of this kernel)
// workgroup barrier that only waits for the wave's LDS traffic: __syncthreads() also drains the vector-memory counter, which would
// stall on the NEXT tile's loads that k_blur_fused<FAST> keeps in flight across its phases
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// one tile's global inputs in registers (FAST): the depth tile as four 16-byte loads per lane, one bit-row window item per lane
struct BlurPre { float4 t4[4]; unsigned long long llo, lhi, rlo, rhi; };   // (raw: frame-border selects, the x255 scale and the window shifts happen when they are stored to LDS)

// FAST (round 5): the instantiation for the usual case -- worklist form, w % 4 == 0 with a 16-byte aligned depth, mask radius 1 .. 31 --
// without the other forms' code (the one kernel for everything kept 106 scalar registers live and spilled 50 more)
template <bool FAST>
__global__ void __launch_bounds__(256, 4) k_blur_fused(BlurArgs A, const unsigned long long* mask_l,
                                                    const unsigned long long* mask_r, int MW, const uint32_t* __restrict__ work_count,
                                                    const uint32_t* __restrict__ worklist, float4* tstat) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (FAST) the frames' x255 decisions as a bit row in LDS: a tile's scale is an LDS read instead of a global load that the fetch
    // of the next tile would have to wait for (worklist entries hold 12 bits of frame index)
    __shared__ uint32_t scale_bits[FAST ? 128 : 1];
    if (FAST) {
        if (tid < 128) scale_bits[tid] = 0u;
        __syncthreads();
        if (A.stats)
            for (int f = tid; f < A.n; f += 256)
                if (A.stats[f * ST_WORDS + ST_SCALE255]) atomicOr(&scale_bits[f >> 5], 1u << (f & 31));
    }
        // weight as a function of the nearest-edge distance d = 0 .. R (d >= R: clamp(1 - d/R) == 0), once per workgroup
    {
        const int v = A.vert, R = A.radius, WR = BLUR_TR + 2 * v, NW = (BLUR_TW + 2 * R + 63) >> 6, DC = ((BLUR_TW + A.bs - 1 + 3) & ~3) + 4;
        float* wl0 = (float*)smem + 8 + ((BLUR_TR * DC + 3) & ~3);
        csm::PowfTables* T = (csm::PowfTables*)((unsigned long long*)(wl0 + 2 * WR * BLUR_TW) + 2 * WR * NW);
        float* wtab = (float*)((int*)(T + 1) + 4);
        if (A.fall_mode == 4) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_blur_powf_tables);
            for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += 256) reinterpret_cast<uint32_t*>(T)[i] = src[i];
            __syncthreads();
        }
        const float rad = (float)R;
        for (int dd = tid; dd <= R; dd += 256) {
            const float t = dd < R ? 1.0f - (float)dd / rad : 0.0f;
            wtab[dd] = falloff_pow(fminf(fmaxf(t, 0.0f), 1.0f), A.fall_mode, A.fall32, T);
        }
        __syncthreads();
    }
    // FAST: `P` holds this tile's inputs (fetched while the previous tile was computed); `next` (a worklist entry, ~0u: none) is
    // fetched into it as soon as they have been stored to LDS -- its loads are in flight during the weights and the boxes
    auto tile = [&](const int x0, const int y0, const int frame, float4* tile_stat, BlurPre& P, const uint32_t next, const bool fetch_only = false) {
        const int w = A.w, h = A.h, v = A.vert, bs = A.bs, pad = A.bs / 2, R = A.radius;
        const int WR = BLUR_TR + 2 * v;          // weight rows: frame rows y0 - v ..
        const int EW = BLUR_TW + 2 * R;          // edge columns: frame cols x0 - R ..
        // depth tile: rows y0 .., cols x0 - pad .., stored in ROW PAIRS: element (r, c) at ((r >> 1) * DC + c) * 2 + (r & 1) -- the
        // two rows a lane of the box phase owns lie side by side, so that every tap is one packed FMA over the pair (v_pk_fma_f32;
        // round 5).  DC: a multiple of 4 with four spare columns per row, and eight floats in front of pair 0, so that the 16-byte
        // loader below stores every element it loaded without a bounds test: what falls left of a pair's first column lands in the
        // previous pair's spare columns.
        const int DC = ((BLUR_TW + bs - 1 + 3) & ~3) + 4;
        auto d_at = [&](int r, int c) -> int { return (((r >> 1) * DC + c) << 1) + (r & 1); };
        const int NW = (EW + 63) >> 6;           // 64-bit words per mask row
        float* D = (float*)smem + 8;                               // [TR / 2][DC][2]
        float* wlt = D + ((BLUR_TR * DC + 3) & ~3);                // [WR][TW]
        float* wrt = wlt + WR * BLUR_TW;                           // [WR][TW]
        unsigned long long* mL = (unsigned long long*)(wrt + WR * BLUR_TW);  // [WR][NW]
        unsigned long long* mR = mL + WR * NW;                     // [WR][NW]
        csm::PowfTables* T = (csm::PowfTables*)(mR + WR * NW);
        int* any_edge = (int*)(T + 1);
        const float* wtab = (const float*)(any_edge + 4);          // [R + 1] weight of nearest-edge distance d (set up by the caller)
        if (!FAST && tid == 0) *any_edge = 0;
        const float scale = FAST ? ((scale_bits[frame >> 5] >> (frame & 31)) & 1u ? 255.0f : 1.0f)
                                 : ((A.stats && A.stats[frame * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f);
        const float* d = A.depth + (size_t)frame * h * w;
        const size_t po = (A.mask_plane && scale != 1.0f) ? A.mask_plane : 0;
        // 1. (round 5) the depth tile's global loads go out FIRST, as 16-byte accesses: 8 lanes per tile row, the row read from
        // the 16-byte boundary at or left of x0 - pad (w % 4 == 0: a float4 lies inside the frame or outside it as a whole).  They
        // are in flight while the bit-row windows are fetched (one memory round trip per tile instead of two), and there is no
        // division per element (the scalar form below spent 80 instructions on each of its 11 loads per lane: a third of the kernel).
        const bool vec = FAST || ((w & 3) == 0 && (reinterpret_cast<uintptr_t>(A.depth) & 15) == 0);
        const int pad4 = (pad + 3) & ~3, shl = pad4 - pad;
        const int NQ = (pad4 + BLUR_TW + (bs - 1 - pad) + 3) >> 2;   // float4 per tile row
        constexpr int QU = 4;
        float4 t4[QU];
        const int drow_i = tid >> 3, dq = tid & 7;   // (the generic 16-byte loader: 8 lanes per row)
        const int pp = tid >> 4, pq = tid & 15;      // (FAST: 16 lanes per row pair)
        auto load_qf = [&](const float* fd, float fscale, int fx0, int fy0, int q) -> float4 {
            const int yy = fy0 + drow_i, xx = fx0 - pad4 + 4 * q;
            const bool ok = q < NQ && yy < h && xx >= 0 && xx < w;
            // (a clamped, always valid address and a select on the values: see k_gray_edges)
            const float4 t = *reinterpret_cast<const float4*>(fd + (size_t)(ok ? yy : 0) * w + (ok ? xx : 0));
            return make_float4(ok ? t.x * fscale : 0.0f, ok ? t.y * fscale : 0.0f, ok ? t.z * fscale : 0.0f, ok ? t.w * fscale : 0.0f);
        };
        auto load_q = [&](int q) -> float4 { return load_qf(d, scale, x0, y0, q); };
        const unsigned long long lastmask = (EW & 63) ? (~0ull >> (64 - (EW & 63))) : ~0ull;
        // (FAST) every global input of tile (fx0, fy0, fframe) into registers, RAW: nothing here consumes a loaded value, so no wait
        // is placed before the phases that follow (NQ <= 32 and WR * NW <= 256: launch_blur checks).  Clamped, always valid addresses.
        auto fscale_of = [&](int fframe) -> float { return (scale_bits[fframe >> 5] >> (fframe & 31)) & 1u ? 255.0f : 1.0f; };
        auto fetch = [&](int fx0, int fy0, int fframe, BlurPre& Q) {
            const float* fd = A.depth + (size_t)fframe * h * w;
            const size_t fpo = (A.mask_plane && fscale_of(fframe) != 1.0f) ? A.mask_plane : 0;
            // (16 lanes per row pair: t4[0], t4[1] = rows 2p, 2p + 1 at float4 column `pq`; t4[2], t4[3] = the same at pq + 16)
#pragma unroll
            for (int k = 0; k < QU; k++) {
                const int q = pq + 16 * (k >> 1), yy = fy0 + 2 * pp + (k & 1), xx = fx0 - pad4 + 4 * q;
                const bool ok = q < NQ && yy < h && xx >= 0 && xx < w;
                Q.t4[k] = *reinterpret_cast<const float4*>(fd + (size_t)(ok ? yy : 0) * w + (ok ? xx : 0));
            }
            const int r = tid >> 1, k = tid & 1, yy = fy0 - v + r;   // (NW == 2)
            const bool rowok = r < WR && yy >= 0 && yy < h;
            const int wi = (fx0 - R + 64 * k) >> 6;                  // (-1 for the first tile's left reach)
            const int w0 = min(max(wi, 0), MW - 1), w1 = min(max(wi + 1, 0), MW - 1);
            const size_t ro = fpo + ((size_t)fframe * h + (rowok ? yy : 0)) * MW;
            Q.llo = mask_l[ro + w0]; Q.lhi = mask_l[ro + w1];
            Q.rlo = mask_r[ro + w0]; Q.rhi = mask_r[ro + w1];
        };
        // ... and into LDS, with everything fetch() left out (the tile is (x0, y0, frame) again)
        auto commit = [&](const BlurPre& Q) {
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int q = pq + 16 * m, xx = x0 - pad4 + 4 * q;
                const bool okx = xx >= 0 && xx < w, ok0 = okx && y0 + 2 * pp < h, ok1 = okx && y0 + 2 * pp + 1 < h;
                if (q < NQ) {
                    float2* const dst = reinterpret_cast<float2*>(D) + (pp * DC + (4 * q - shl));   // one 8-byte store per column
                    const float4 a = Q.t4[2 * m], b = Q.t4[2 * m + 1];
                    dst[0] = make_float2(ok0 ? a.x * scale : 0.0f, ok1 ? b.x * scale : 0.0f);
                    dst[1] = make_float2(ok0 ? a.y * scale : 0.0f, ok1 ? b.y * scale : 0.0f);
                    dst[2] = make_float2(ok0 ? a.z * scale : 0.0f, ok1 ? b.z * scale : 0.0f);
                    dst[3] = make_float2(ok0 ? a.w * scale : 0.0f, ok1 ? b.w * scale : 0.0f);
                }
            }
            const int r = tid >> 1, k = tid & 1, yy = y0 - v + r;
            if (r < WR) {
                const int fb = x0 - R + 64 * k, wi = fb >> 6, sh = fb & 63;   // mask_window(): bits fb .. fb + 63 of the row
                unsigned long long bl = 0ull, br = 0ull;
                if (yy >= 0 && yy < h) {
                    if (fb < 0) { bl = Q.lhi << (-fb); br = Q.rhi << (-fb); }   // (fb > -64: R <= 31; both loads were word 0)
                    else {
                        const unsigned long long ll = wi < MW ? Q.llo : 0ull, lh = wi + 1 < MW ? Q.lhi : 0ull;
                        const unsigned long long rl = wi < MW ? Q.rlo : 0ull, rh = wi + 1 < MW ? Q.rhi : 0ull;
                        bl = sh ? (ll >> sh) | (lh << (64 - sh)) : ll;
                        br = sh ? (rl >> sh) | (rh << (64 - sh)) : rl;
                    }
                    if (k == 1) { bl &= lastmask; br &= lastmask; }
                }
                mL[tid] = bl; mR[tid] = br;
            }
        };
        auto store_q = [&](int q, const float4& t) {   // (columns -shl .. 4 NQ - shl - 1 <= BLUR_TW + bs + 1 < DC: see DC)
            if (q >= NQ) return;
            float* const dst = D + d_at(drow_i, 4 * q - shl);
            dst[0] = t.x; dst[2] = t.y; dst[4] = t.z; dst[6] = t.w;
        };
        if (FAST && fetch_only) { fetch(x0, y0, frame, P); return; }   // (the workgroup's first tile)
        if (FAST) {
            // (the previous tile's last LDS reads are behind the barrier at the end of the worklist loop)
            commit(P);
            if (next != ~0u) fetch((int)(next & 1023u) * BLUR_TW, (int)((next >> 10) & 1023u) * BLUR_TR, (int)(next >> 20), P);
            lds_barrier();
        }
        if (!FAST && vec) {
#pragma unroll
            for (int k = 0; k < QU; k++) t4[k] = load_q(dq + 8 * k);
        }
        if (!FAST) __syncthreads();
        // 2. the tile's window of the frame-wide edge bit rows (k_blur_edges): rows y0-v .., bits x0-R ..
        if (!FAST)
        for (int item = tid; item < WR * NW; item += 256) {
            const int r = item / NW, k = item - r * NW;
            const int yy = y0 - v + r;
            unsigned long long bl = 0ull, br = 0ull;
            if (yy >= 0 && yy < h) {
                const size_t ro = ((size_t)frame * h + yy) * MW;
                bl = mask_window(mask_l + po + ro, MW, x0 - R + 64 * k);
                br = mask_window(mask_r + po + ro, MW, x0 - R + 64 * k);
                if (k == NW - 1) { bl &= lastmask; br &= lastmask; }
            }
            mL[item] = bl; mR[item] = br;
            if (bl | br) *any_edge = 1;
        }
        if (!FAST) __syncthreads();   // (the bit-row windows are read by every thread in step 3)
        if (BLUR_DEV_IS(21)) return;
        if (!FAST && *any_edge == 0 && A.fall_mode != 5) {   // (FAST: the worklist only names tiles with an edge in reach)
            // no edge within reach of the tile: every weight is exactly 0, so 0*blur + (1-0)*depth == depth
            float mn = INFINITY, mx = -INFINITY;
            for (int i = tid; i < BLUR_TR * BLUR_TW; i += 256) {
                const int r = i >> 6, c = i & 63, y = y0 + r, x = x0 + c;
                if (y < h && x < w) {
                    float dvv = d[(size_t)y * w + x] * scale;
                    A.out_l[((size_t)frame * h + y) * w + x] = dvv;
                    A.out_r[((size_t)frame * h + y) * w + x] = dvv;
                    mn = fminf(mn, dvv); mx = fmaxf(mx, dvv);
                }
            }
            if (A.stats_rw && tile_stat) {
                mn = wave_min(mn); mx = wave_max(mx);
                if (lane == 0) tile_stat[wave] = make_float4(mn, mx, mn, mx);
            } else if (A.stats_rw) {
                __shared__ float red2[2 * 16];
                uint32_t* st = A.stats_rw + frame * ST_WORDS;
                block_minmax_update(mn, mx, &st[ST_L_MIN], &st[ST_L_MAX], red2);
                block_minmax_update(mn, mx, &st[ST_R_MIN], &st[ST_R_MAX], red2);
            }
            return;
        }
        // depth tile for the box filter (zero outside the frame == the reference's zero padding); the loads of a chunk
        // are all issued before the first LDS store so that a chunk costs one memory round trip, not one per element
        if (!FAST && vec) {
#pragma unroll
            for (int k = 0; k < QU; k++) store_q(dq + 8 * k, t4[k]);
            for (int qb = 8 * QU; qb < NQ; qb += 8 * QU) {   // (box widths beyond 100 columns)
#pragma unroll
                for (int k = 0; k < QU; k++) t4[k] = load_q(qb + dq + 8 * k);
#pragma unroll
                for (int k = 0; k < QU; k++) store_q(qb + dq + 8 * k, t4[k]);
            }
        } else if (!FAST) {
            constexpr int CH = 12;
            const int total = BLUR_TR * DC;
            for (int base = 0; base < total; base += CH * 256) {
                float tmp[CH];
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    const int i = base + k * 256 + tid;
                    const int r = i / DC, c = i - r * DC;
                    const int yy = y0 + r, xx = x0 - pad + c;
                    const bool ok = i < total && yy < h && xx >= 0 && xx < w;
                    tmp[k] = ok ? d[(size_t)yy * w + xx] * scale : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    const int i = base + k * 256 + tid;
                    if (i < total) D[d_at(i / DC, i % DC)] = tmp[k];
                }
            }
        }
        if (BLUR_DEV_IS(22)) return;
        // 3. weights from the bit rows.  Only edges within R columns give a non-zero weight, so a pixel looks at the 2R + 1
        // bits around its own column: one funnel shift puts them into a 64-bit word (R <= 31), the nearest set bit on
        // either side is a clz / ctz of its two halves.  (Wider masks: word-by-word searches over the whole window.)
        const bool narrow = 2 * R + 1 <= 63;
        auto nearest = [&](const unsigned long long* m, int c) -> int {   // min(distance to the nearest edge, R + 1)
            if (narrow) {
                const unsigned long long lo = m[0], hi = NW > 1 ? m[1] : 0ull;
                unsigned long long W = c ? (lo >> c) | (hi << (64 - c)) : lo;   // bit k = window bit c + k: the pixel sits at bit R
                W &= (2ull << (2 * R)) - 1ull;
                const unsigned long long left = W & ((2ull << R) - 1ull), right = W >> R;
                const int dl = left ? R - (63 - __clzll((long long)left)) : R + 1;
                const int dr = right ? __ffsll((long long)right) - 1 : R + 1;
                return min(dl, dr);
            }
            const int p = c + R;
            const int a = mask_dist_left(m, p), b = mask_dist_right(m, NW, p);
            return min(a >= 0 ? a : R + 1, b >= 0 ? b : R + 1);
        };
        if (FAST || (narrow && NW == 2)) {
            // (round 5) 32-bit form: a wave owns a weight row, the lane a column -- the row's four dwords come from one broadcast
            // 16-byte LDS read, two v_alignbit put window bits c .. c + 63 into a register pair, the nearest edge on the left is a
            // clz of its low R + 1 bits (clz(0) == 32 gives R + 1 by itself), on the right an ffs of bits R .. 2R: 15 instructions
            // per pixel and eye instead of ~40 in 64-bit arithmetic
            const int kq = lane >> 5, sh = lane & 31;
            const unsigned lmask = R == 31 ? ~0u : (2u << R) - 1u;
            auto dist32 = [&](const uint4& q) -> int {
                const unsigned a = kq ? q.y : q.x, b = kq ? q.z : q.y, c2 = kq ? q.w : q.z;
                const unsigned w0 = __builtin_amdgcn_alignbit(b, a, sh), w1 = __builtin_amdgcn_alignbit(c2, b, sh);
                const unsigned dl = (unsigned)(R - 31 + __clz((int)(w0 & lmask)));
                const unsigned rr = __builtin_amdgcn_alignbit(w1, w0, R) & lmask;
                const unsigned dr = (unsigned)(__ffs((int)rr) - 1);   // (no bit: 0xffffffff)
                return (int)min(min(dl, dr), (unsigned)R);
            };
            const float w_none = wtab[R];
            for (int r = wave; r < WR; r += 4) {
                const int yy = y0 - v + r;
                float wl = 0.0f, wr = 0.0f;
                if (yy >= 0 && yy < h && x0 + lane < w) {
                    // (a window row without bits -- most rows have them for one eye at most: an edge is rising or falling -- is all
                    // "no edge in reach": wtab[R].  The test is the same for every lane: the row is a broadcast read.)
                    const uint4 ql = *reinterpret_cast<const uint4*>(mL + r * 2), qr = *reinterpret_cast<const uint4*>(mR + r * 2);
                    wl = wr = w_none;
                    if (ql.x | ql.y | ql.z | ql.w) wl = wtab[dist32(ql)];
                    if (qr.x | qr.y | qr.z | qr.w) wr = wtab[dist32(qr)];
                }
                wlt[r * BLUR_TW + lane] = wl; wrt[r * BLUR_TW + lane] = wr;
            }
        } else if (!FAST)
        for (int i = tid; i < WR * BLUR_TW; i += 256) {
            const int r = i >> 6, c = i & 63;
            const int yy = y0 - v + r;
            float wl = 0.0f, wr = 0.0f;
            if (yy >= 0 && yy < h && x0 + c < w) {
                // an edge at distance >= R gives clamp(1 - d/R) == 0 exactly, like "no edge"
                wl = wtab[min(nearest(mL + r * NW, c), R)];
                wr = wtab[min(nearest(mR + r * NW, c), R)];
            }
            wlt[i] = wl; wrt[i] = wr;
        }
        if (FAST) lds_barrier(); else __syncthreads();
        if (BLUR_DEV_IS(23)) return;
        // 4. boxes + blend: a thread owns 4 consecutive columns of 2 consecutive rows and reads its operands as float4 --
        // 13x fewer LDS instructions than one column per lane (the phase was bound by LDS issue, not by the fmaf chains);
        // every chain keeps the reference's order (taps ascending from an accumulator of 0)
        const int cx = 4 * (tid & 15), r0 = 2 * (tid >> 4);
        const float kb = 1.0f / (float)bs, kv = 1.0f / (float)(2 * v + 1);
        // (round 5) packed float32 FMAs: the vertical weight sums pair neighbouring columns, the horizontal box sums pair the lane's
        // two rows (the depth tile's row-pair layout) -- v_pk_fma_f32 is two IEEE fmaf per issue slot, same results
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 kb2 = {kb, kb}, kv2 = {kv, kv};
        float wa[2][4], wb[2][4], acc[2][4];
        if (v > 0) {
            f2 wa2[2][2], wb2[2][2];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int c = 0; c < 2; c++) { wa2[j][c] = f2{0.0f, 0.0f}; wb2[j][c] = f2{0.0f, 0.0f}; }
            const int nv = 2 * v + 1;
            for (int kk = 0; kk <= nv; kk++) {   // weight row r0 + kk is tap kk of output row r0 and tap kk - 1 of row r0 + 1
                const int rr = min(r0 + kk, WR - 1);
                const float4 l4 = *reinterpret_cast<const float4*>(wlt + rr * BLUR_TW + cx);
                const float4 r4 = *reinterpret_cast<const float4*>(wrt + rr * BLUR_TW + cx);
                const f2 lv[2] = {f2{l4.x, l4.y}, f2{l4.z, l4.w}}, rv[2] = {f2{r4.x, r4.y}, f2{r4.z, r4.w}};
                if (kk < nv) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        wa2[0][c] = __builtin_elementwise_fma(kv2, lv[c], wa2[0][c]);
                        wb2[0][c] = __builtin_elementwise_fma(kv2, rv[c], wb2[0][c]);
                    }
                }
                if (kk >= 1) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        wa2[1][c] = __builtin_elementwise_fma(kv2, lv[c], wa2[1][c]);
                        wb2[1][c] = __builtin_elementwise_fma(kv2, rv[c], wb2[1][c]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int c = 0; c < 4; c++) { wa[j][c] = wa2[j][c >> 1][c & 1]; wb[j][c] = wb2[j][c >> 1][c & 1]; }
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float4 l4 = *reinterpret_cast<const float4*>(wlt + (r0 + j) * BLUR_TW + cx);
                const float4 r4 = *reinterpret_cast<const float4*>(wrt + (r0 + j) * BLUR_TW + cx);
                wa[j][0] = l4.x; wa[j][1] = l4.y; wa[j][2] = l4.z; wa[j][3] = l4.w;
                wb[j][0] = r4.x; wb[j][1] = r4.y; wb[j][2] = r4.z; wb[j][3] = r4.w;
            }
        }
        if (BLUR_DEV_IS(24)) return;
        // horizontal box: tap k of column c is D column c + k (the tile starts at frame column x0 - pad); dp[c] = {row r0, row r0 + 1}
        const float2* const dp = reinterpret_cast<const float2*>(D) + ((tid >> 4) * DC + cx);
        float lmin = INFINITY, lmax = -INFINITY, rmin = INFINITY, rmax = -INFINITY;
        {
            f2 acc2[4];
#pragma unroll
            for (int c = 0; c < 4; c++) acc2[c] = f2{0.0f, 0.0f};
            float4 na = *reinterpret_cast<const float4*>(dp), nb = *reinterpret_cast<const float4*>(dp + 2);
            for (int k0 = 0; k0 < bs; k0 += 4) {
                const float4 ca = na, cb = nb;
                na = *reinterpret_cast<const float4*>(dp + k0 + 4);   // (the last chunk may read the row pair's spare columns: unused)
                nb = *reinterpret_cast<const float4*>(dp + k0 + 6);
                const f2 d8[8] = {f2{ca.x, ca.y}, f2{ca.z, ca.w}, f2{cb.x, cb.y}, f2{cb.z, cb.w},
                                  f2{na.x, na.y}, f2{na.z, na.w}, f2{nb.x, nb.y}, f2{nb.z, nb.w}};
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    if (k0 + kk < bs) {
#pragma unroll
                        for (int c = 0; c < 4; c++) acc2[c] = __builtin_elementwise_fma(kb2, d8[c + kk], acc2[c]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) { acc[0][c] = acc2[c][0]; acc[1][c] = acc2[c][1]; }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int y = y0 + r0 + j, x = x0 + cx;
            if (y < h && x < w) {
                float ol[4], orr[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float2 dv2 = dp[c + pad];
                    const float dvv = j ? dv2.y : dv2.x;
                    ol[c] = wa[j][c] * acc[j][c] + (1.0f - wa[j][c]) * dvv;
                    orr[c] = wb[j][c] * acc[j][c] + (1.0f - wb[j][c]) * dvv;
                }
                const size_t off = ((size_t)frame * h + y) * w + x;
                if (x + 3 < w && (w & 3) == 0) {
                    *reinterpret_cast<float4*>(A.out_l + off) = make_float4(ol[0], ol[1], ol[2], ol[3]);
                    *reinterpret_cast<float4*>(A.out_r + off) = make_float4(orr[0], orr[1], orr[2], orr[3]);
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (x + c < w) {
                        if (!(x + 3 < w && (w & 3) == 0)) { A.out_l[off + c] = ol[c]; A.out_r[off + c] = orr[c]; }
                        lmin = fminf(lmin, ol[c]); lmax = fmaxf(lmax, ol[c]);
                        rmin = fminf(rmin, orr[c]); rmax = fmaxf(rmax, orr[c]);
                    }
                }
            }
        }
        if (A.stats_rw && tile_stat) {
            lmin = wave_min(lmin); lmax = wave_max(lmax); rmin = wave_min(rmin); rmax = wave_max(rmax);
            if (lane == 0) tile_stat[wave] = make_float4(lmin, lmax, rmin, rmax);
        } else if (!FAST && A.stats_rw) {
            __shared__ float red[2 * 16];
            uint32_t* st = A.stats_rw + frame * ST_WORDS;
            block_minmax_update(lmin, lmax, &st[ST_L_MIN], &st[ST_L_MAX], red);
            block_minmax_update(rmin, rmax, &st[ST_R_MIN], &st[ST_R_MAX], red);
        }
    };
    BlurPre P;
    if (!FAST && !worklist) {
        tile(blockIdx.x * BLUR_TW, blockIdx.y * BLUR_TR, blockIdx.z, nullptr, P, ~0u);
        return;
    }
    const uint32_t count = *work_count;
    if (FAST) {
        // software pipeline over the workgroup's tiles: tile i's inputs were fetched during tile i - stride
        uint32_t i = blockIdx.x;
        if (i >= count) return;
        uint32_t e = worklist[i];
        tile((int)(e & 1023u) * BLUR_TW, (int)((e >> 10) & 1023u) * BLUR_TR, (int)(e >> 20), nullptr, P, ~0u, true);
        for (; i < count; i += gridDim.x) {
            const uint32_t nx = i + gridDim.x;
            const uint32_t e2 = nx < count ? worklist[nx] : ~0u;
            tile((int)(e & 1023u) * BLUR_TW, (int)((e >> 10) & 1023u) * BLUR_TR, (int)(e >> 20), tstat ? tstat + 4 * (size_t)i : nullptr, P, e2);
            lds_barrier();  // the tile's LDS is reused by the next one
            e = e2;
        }
        return;
    }
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t e = worklist[i];
        tile((int)(e & 1023u) * BLUR_TW, (int)((e >> 10) & 1023u) * BLUR_TR, (int)(e >> 20), tstat ? tstat + 4 * (size_t)i : nullptr, P, ~0u);
        __syncthreads();  // the tile's LDS is reused by the next one
    }
}

// The listed tiles' output extremes -> the frame statistics: workgroup (b, frame) folds the entries of its frame among every
// gridDim.x-th slice of the worklist, then one pair of pre-checked atomics per workgroup.
__global__ void __launch_bounds__(256) k_blur_tile_stats(const uint32_t* work_count, const uint32_t* worklist, const float4* tstat,
                                                         uint32_t* stats) {
    __shared__ float red[2 * 16];
    const uint32_t count = *work_count, frame = blockIdx.y;
    float lmin = INFINITY, lmax = -INFINITY, rmin = INFINITY, rmax = -INFINITY;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
        if ((worklist[i] >> 20) != frame) continue;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 q = tstat[4 * (size_t)i + k];
            lmin = fminf(lmin, q.x); lmax = fmaxf(lmax, q.y); rmin = fminf(rmin, q.z); rmax = fmaxf(rmax, q.w);
        }
    }
    uint32_t* st = stats + frame * ST_WORDS;
    block_minmax_update(lmin, lmax, &st[ST_L_MIN], &st[ST_L_MAX], red);
    block_minmax_update(rmin, rmax, &st[ST_R_MIN], &st[ST_R_MAX], red);
}

int blur_tilemap_words(int w) { return ((w + BLUR_TW - 1) / BLUR_TW + 31) / 32 + 1; }
size_t blur_tilemap_bytes(int n, int h, int w) {
    return (size_t)n * ((h + BLUR_TR - 1) / BLUR_TR) * blur_tilemap_words(w) * 4;
}

// flagged rows of the tiled warp kernels are redone by row kernels that read complete depth rows: write the lazy tiles' part
__global__ void __launch_bounds__(256) k_lazy_rows(const uint32_t* list, const uint32_t* count, int total, const float* gray,
                                                   float* out_l, float* out_r, const uint32_t* tilemap, const uint32_t* stats,
                                                   int h, int w, int tm_words) {
    const uint32_t nrows = list ? *count : (uint32_t)total;
    const int gy = (h + BLUR_TR - 1) / BLUR_TR;
    for (uint32_t i = blockIdx.x; i < nrows; i += gridDim.x) {
        const uint32_t fr = list ? (list[i] & 0x3fffffffu) : i;   // (bits 30-31 of a list entry: the eyes the row kernel redoes, k_collect_rows)
        const int frame = (int)(fr / (uint32_t)h), row = (int)(fr - (uint32_t)frame * (uint32_t)h);
        const float scale = stats[frame * ST_WORDS + ST_SCALE255] ? 255.0f : 1.0f;
        const uint32_t* tm = tilemap + ((size_t)frame * gy + row / BLUR_TR) * tm_words;
        const size_t base = (size_t)fr * w;
        for (int x = threadIdx.x; x < w; x += 256) {
            const int t = x / BLUR_TW;
            if (!((tm[t >> 5] >> (t & 31)) & 1u)) {
                const float v = gray[base + x] * scale;
                out_l[base + x] = v; out_r[base + x] = v;
            }
        }
    }
}
hipError_t launch_lazy_rows(const uint32_t* list, const uint32_t* count, int total, const float* gray, float* out_l, float* out_r,
                            const uint32_t* tilemap, const uint32_t* stats, int h, int w, hipStream_t stream) {
    hipLaunchKernelGGL(k_lazy_rows, dim3(list ? 1024 : 4096), dim3(256), 0, stream, list, count, total, gray, out_l, out_r, tilemap, stats, h, w,
                       blur_tilemap_words(w));
    return hipGetLastError();
}

// Complete maps for consumers that read whole rows (gpu_warp, hybrid_edge, the row kernels): the tiles the map does not name
// are a scaled copy of the gray depth.  A pure streaming kernel -- the classification (and the frame statistics of these
// tiles) is k_blur_classify's: k_blur_copy, which does both, is bound by its classification chain (1.35 ms per 64 4K frames).
__global__ void __launch_bounds__(256) k_blur_copy_tiles(BlurArgs A, const uint32_t* tilemap, int tm_words) {
    const int tid = threadIdx.x;
    const int X0 = blockIdx.x * (BLUR_CW * BLUR_TW), y0 = blockIdx.y * BLUR_TR, frame = blockIdx.z;
    const int w = A.w, h = A.h;
    constexpr int C4 = BLUR_CW * BLUR_TW / 4, RPT = BLUR_TR * C4 / 256;
    const int c4 = tid % C4, rb = tid / C4;
    const int x = X0 + 4 * c4;
    const int t = x / BLUR_TW;
    const uint32_t word = tilemap[((size_t)frame * gridDim.y + blockIdx.y) * tm_words + (t >> 5)];
    if (x >= w || ((word >> (t & 31)) & 1u)) return;
    const float scale = (A.stats && A.stats[frame * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f;
    float4 vv[RPT];
#pragma unroll
    for (int i = 0; i < RPT; i++) {
        const int y = y0 + rb + (256 / C4) * i;
        vv[i] = y < h ? *reinterpret_cast<const float4*>(A.depth + ((size_t)frame * h + y) * w + x) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < RPT; i++) {
        const int y = y0 + rb + (256 / C4) * i;
        if (y < h) {
            const float4 o = make_float4(vv[i].x * scale, vv[i].y * scale, vv[i].z * scale, vv[i].w * scale);
            const size_t off = ((size_t)frame * h + y) * w + x;
            *reinterpret_cast<float4*>(A.out_l + off) = o;
            *reinterpret_cast<float4*>(A.out_r + off) = o;
        }
    }
}

static size_t blur_fused_lds(int v, int R, int bs) {
    int WR = BLUR_TR + 2 * v, EW = BLUR_TW + 2 * R, NW = (EW + 63) >> 6;
    (void)EW;
    return 32 + (size_t)(BLUR_TR * (((BLUR_TW + bs - 1 + 3) & ~3) + 4)) * 4 + 2 * (size_t)WR * BLUR_TW * 4 + 2 * (size_t)WR * NW * 8 +
           sizeof(csm::PowfTables) + 64 + 4 * (size_t)(R + 2);
}

// The geometry launch_blur and launch_gray_edges share: which path the parameters take and where the bit rows, the worklist
// and the block summaries lie inside the two weight scratch buffers (each n * h * w floats).
struct BlurPlan {
    bool fused, listed, lazy;
    int MW, HB, gx, gy;
    size_t plane_bytes, mask_bytes, list_bytes, blk_bytes;
};
static BlurPlan blur_plan(int n, int h, int w, double strength, double mask_width, int vert, bool tilemap, bool node_path, int planes) {
    BlurPlan P;
    const int bs = (int)nearbyint(strength), radius = (int)mask_width, v = vert > 0 ? vert : 0;
    P.fused = bs >= 1 && blur_fused_lds(v, radius, bs) <= 64 * 1024 && radius >= 1 && !dev_switch(CS_DEBUG_BLUR_TWO_PASS);
    P.MW = (w + 63) / 64;
    P.gy = (h + BLUR_TR - 1) / BLUR_TR; P.gx = (w + BLUR_TW - 1) / BLUR_TW; P.HB = (h + BLUR_ER4 - 1) / BLUR_ER4;
    P.plane_bytes = ((size_t)n * h * P.MW * 8 + 255) & ~(size_t)255;
    P.mask_bytes = P.plane_bytes * planes;
    {   // counter, worklist (4 B per tile, the extremes behind it start on a 16-byte boundary: launch_blur), 4 x float4 of output
        // extremes per entry
        const size_t total = (size_t)n * P.gy * P.gx;
        P.list_bytes = 256 + ((total * 4 + 15) & ~(size_t)15) + total * 64;
    }
    P.blk_bytes = (size_t)n * P.HB * P.MW * 16;
    P.listed = P.fused && (w & 3) == 0 && P.gx < 1024 && P.gy < 1024 && n < 4096 && P.mask_bytes + P.list_bytes <= (size_t)n * h * w * 4;
    // lazy mode: block summaries behind the second bit-row buffer
    P.lazy = tilemap && node_path && P.listed && !dev_switch(CS_DEBUG_BLUR_EDGES_SCALAR) && BLUR_TR % BLUR_ER4 == 0 &&
             P.mask_bytes + P.blk_bytes <= (size_t)n * h * w * 4;
    return P;
}

// the largest float t >= 0 with fl(t / den) <= 0.5 (k_gray_edges); < 0: no such value (den not positive and finite)
static float blur_edge_threshold(float den) {
    if (!(den > 0.0f) || !std::isfinite(den)) return -1.0f;
    volatile float d = den;
    float t = 0.5f * den;
    if (!std::isfinite(t)) return -1.0f;
    auto quot = [&](float a) { volatile float q = a / d; return (float)q; };
    for (int i = 0; i < 8 && quot(t) > 0.5f; i++) t = nextafterf(t, 0.0f);
    for (int i = 0; i < 8 && quot(nextafterf(t, INFINITY)) <= 0.5f; i++) t = nextafterf(t, INFINITY);
    if (quot(t) > 0.5f || quot(nextafterf(t, INFINITY)) <= 0.5f || !(t > 0.0f)) return -1.0f;
    return t;
}

float blur_edge_threshold_host(float den) { return blur_edge_threshold(den); }   // (cs_test_edge_threshold)

bool blur_pre_edges_ok(int n, int h, int w, double strength, double edge_threshold, double mask_width, int vert, bool tilemap) {
    if (dev_switch(CS_DEBUG_BLUR_NO_PRE_EDGES)) return false;
    const BlurPlan P = blur_plan(n, h, w, strength, mask_width, vert, tilemap, true, 2);
    return P.lazy && (w & 3) == 0 && blur_edge_threshold((float)(10.0 * edge_threshold)) >= 0.0f;
}

hipError_t launch_gray_edges(const float* rgb, float* gray, int n, int h, int w, uint32_t* stats, double strength,
                             double edge_threshold, double mask_width, int vert, float* wl, float* wr, hipStream_t stream) {
    const BlurPlan P = blur_plan(n, h, w, strength, mask_width, vert, true, true, 2);
    hipLaunchKernelGGL(k_gray_edges, dim3((w + 255) / 256, (h + GE_RB - 1) / GE_RB, n), dim3(64), 0, stream, rgb, gray, h, w, stats,
                       blur_edge_threshold((float)(10.0 * edge_threshold)), reinterpret_cast<unsigned long long*>(wl),
                       reinterpret_cast<unsigned long long*>(wr), P.plane_bytes / 8, P.MW,
                       reinterpret_cast<float4*>(reinterpret_cast<char*>(wr) + P.mask_bytes), P.HB);
    return hipGetLastError();
}

int launch_blur(const float* depth, int n, int h, int w, double strength, double edge_threshold, double mask_width,
                double falloff, int vert, float* out_l, float* out_r, float* wl, float* wr, uint32_t* stats, int node_path,
                hipStream_t stream, uint32_t* tilemap, int* lazy_used, int pre_edges) {
    // (lazy_used == nullptr with a tile map: the caller wants COMPLETE maps; the map is then only the classification's output)
    const bool want_lazy = lazy_used != nullptr;
    if (lazy_used) *lazy_used = 0;
    BlurArgs A;
    A.depth = depth; A.n = n; A.h = h; A.w = w;
    A.stats = node_path ? stats : nullptr;
    A.stats_rw = node_path ? stats : nullptr;
    A.den = (float)(10.0 * edge_threshold);
    A.bs = (int)nearbyint(strength);  // Python round(): half to even
    A.radius = (int)mask_width;  // mask_radius = int(blur_mask_width), reference :1209
    A.vert = vert > 0 ? vert : 0;
    if (A.bs < 1) return CS_EINVAL;  // torch raises on a zero-width kernel
    A.fall32 = (float)falloff;
    A.fall_mode = falloff == 1.0 ? 0 : falloff == 0.5 ? 1 : falloff == 2.0 ? 2 : falloff == 3.0 ? 3 : falloff == 0.0 ? 5 : 4;
    A.wl = wl; A.wr = wr; A.out_l = out_l; A.out_r = out_r;
    A.dbg = dev_switch(CS_DEBUG_DBG);
    A.mask_plane = 0;
    const BlurPlan P = blur_plan(n, h, w, strength, mask_width, A.vert, tilemap != nullptr, node_path != 0, pre_edges ? 2 : 1);
    if (pre_edges && !P.lazy) return CS_EINVAL;   // (the caller asked blur_pre_edges_ok)
    size_t ldsF = blur_fused_lds(A.vert, A.radius, A.bs);
    if (P.fused) {
        hipError_t e = hipFuncSetAttribute((const void*)k_blur_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsF);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_blur_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsF);
        if (e != hipSuccess) return CS_EHIP;
        // the weight scratch buffers double as the frame-wide edge bit rows (2 x ceil(w/64) words per image row)
        const int MW = P.MW;
        unsigned long long* mask_l = reinterpret_cast<unsigned long long*>(wl);
        unsigned long long* mask_r = reinterpret_cast<unsigned long long*>(wr);
        // edge-free tiles are copied by k_blur_copy; the others reach k_blur_fused through a worklist behind the bit rows
        const int gy = P.gy, gx = P.gx, HB = P.HB;
        const size_t mask_bytes = P.mask_bytes;
        const bool listed = P.listed, lazy = P.lazy;
        float4* blk = lazy ? reinterpret_cast<float4*>(reinterpret_cast<char*>(wr) + mask_bytes) : nullptr;
        if (pre_edges) A.mask_plane = P.plane_bytes / 8;   // (k_gray_edges wrote the bit rows and the summaries)
        else if ((w & 3) == 0 && !dev_switch(CS_DEBUG_BLUR_EDGES_SCALAR))
            hipLaunchKernelGGL(k_blur_edges4, dim3((w + 1023) / 1024, HB, n), dim3(256), 0, stream, A, mask_l, mask_r, MW, blk);
        else
            hipLaunchKernelGGL(k_blur_edges, dim3((w + 255) / 256, (h + BLUR_ER - 1) / BLUR_ER, n), dim3(256), 0, stream, A, mask_l, mask_r, MW);
        if (listed) {
            uint32_t* work_count = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(wl) + mask_bytes);
            uint32_t* worklist = work_count + 64;
            if (hipMemsetAsync(work_count, 0, 4, stream) != hipSuccess) return CS_EHIP;
            if (lazy) {
                if (hipMemsetAsync(tilemap, 0, blur_tilemap_bytes(n, h, w), stream) != hipSuccess) return CS_EHIP;
                hipLaunchKernelGGL(k_blur_classify, dim3((gx * gy + 255) / 256, n), dim3(256), 0, stream, A,
                                   (const unsigned long long*)mask_l, (const unsigned long long*)mask_r, MW, (const float4*)blk, HB,
                                   work_count, worklist, tilemap, blur_tilemap_words(w));
                if (want_lazy) *lazy_used = 1;
                else hipLaunchKernelGGL(k_blur_copy_tiles, dim3((gx + BLUR_CW - 1) / BLUR_CW, gy, n), dim3(256), 0, stream, A,
                                        (const uint32_t*)tilemap, blur_tilemap_words(w));
            } else {
                hipLaunchKernelGGL(k_blur_copy, dim3((gx + BLUR_CW - 1) / BLUR_CW, gy, n), dim3(256), 0, stream, A,
                                   (const unsigned long long*)mask_l, (const unsigned long long*)mask_r, MW, work_count, worklist);
            }
            const size_t total = (size_t)n * gy * gx;
            const int pg = (int)(total < 2048 ? total : 2048);
            float4* tstat = A.stats_rw ? reinterpret_cast<float4*>(reinterpret_cast<char*>(worklist) + ((total * 4 + 15) & ~(size_t)15)) : nullptr;
            // (one float4 chunk per lane: box widths up to 100 columns; one bit-row item per lane: vertical smoothing up to 48 rows)
            const int pad4 = (A.bs / 2 + 3) & ~3;
            const bool fast = (w & 3) == 0 && (reinterpret_cast<uintptr_t>(depth) & 15) == 0 && A.radius >= 1 && A.radius <= 31 &&
                              ((pad4 + BLUR_TW + (A.bs - 1 - A.bs / 2) + 3) >> 2) <= 32 && (BLUR_TR + 2 * A.vert) * 2 <= 256 &&
                              !dev_switch(CS_DEBUG_BLUR_EDGES_SCALAR);
            if (fast)
                hipLaunchKernelGGL(k_blur_fused<true>, dim3(pg), dim3(256), ldsF, stream, A, (const unsigned long long*)mask_l,
                                   (const unsigned long long*)mask_r, MW, (const uint32_t*)work_count, (const uint32_t*)worklist, tstat);
            else
                hipLaunchKernelGGL(k_blur_fused<false>, dim3(pg), dim3(256), ldsF, stream, A, (const unsigned long long*)mask_l,
                                   (const unsigned long long*)mask_r, MW, (const uint32_t*)work_count, (const uint32_t*)worklist, tstat);
            if (tstat)
                hipLaunchKernelGGL(k_blur_tile_stats, dim3(4, n), dim3(256), 0, stream, (const uint32_t*)work_count,
                                   (const uint32_t*)worklist, (const float4*)tstat, A.stats_rw);
        } else {
            hipLaunchKernelGGL(k_blur_fused<false>, dim3(gx, gy, n), dim3(256), ldsF, stream, A, (const unsigned long long*)mask_l,
                               (const unsigned long long*)mask_r, MW, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (float4*)nullptr);
        }
        return CS_OK;
    }
    int threads = w <= 256 ? 256 : (w <= 1024 ? 512 : 1024);
    size_t ldsA = 4 * (size_t)w * 4 + 32 * 4 + sizeof(csm::PowfTables) + 64;
    size_t ldsB = ((size_t)(BLUR_TR + 2 * A.vert) * BLUR_TW * 2 + (size_t)BLUR_TR * (BLUR_TW + A.bs - 1)) * 4;
    if (ldsA > CS_LDS_BYTES || ldsB > CS_LDS_BYTES) return CS_ELIMIT;
    hipError_t e = hipFuncSetAttribute((const void*)k_blur_weights, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA);
    if (e != hipSuccess) return CS_EHIP;
    e = hipFuncSetAttribute((const void*)k_blur_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB);
    if (e != hipSuccess) return CS_EHIP;
    hipLaunchKernelGGL(k_blur_weights, dim3(h, n), dim3(threads), ldsA, stream, A);
    hipLaunchKernelGGL(k_blur_apply, dim3((w + BLUR_TW - 1) / BLUR_TW, (h + BLUR_TR - 1) / BLUR_TR, n), dim3(256), ldsB, stream, A);
    return CS_OK;
}

}  // namespace cs
