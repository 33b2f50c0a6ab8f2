// cs_polytile.hip -- tiled fast path of the polylines technique (reference
// stereoimage_generation.py:1912-1992), the kernel behind the headline metric.
//
// The row kernel in cs_rowwarp.hip keeps a whole 4K row (~77 KB) in LDS, so only two workgroups fit
// on a CU and every phase is a block-wide barrier.  Polylines is LOCAL: output pixel p only depends on
// source pixels within S = ceil(|divergence_px| * max(c, 1-c)^e + |separation_px|) + 1 columns
// (c = convergence point; normalised depth lies in [0,1]).  So the row is cut into tiles of T output
// pixels; one 256-thread workgroup owns one tile of one eye and only looks at the source columns
// [o0-S-1, o0+T+S+1): 21.6 KB of LDS and 72 VGPRs -> 7 workgroups per CU, five barriers, no prefix sums
// (DESIGN.md section 5 has the measurements behind each choice):
//   1. all global loads of the tile first (depth row into registers, image row as float4), LDS set-up in their shadow
//   2. stage the source pixels (f32 -> u8) and the libm-exact disparity of the halo'ed range -> point x's; the same
//      loop finds out whether the polyline FOLDS anywhere in the tile (neighbour x by shuffle)
//   3. REGISTER.  Fold-free tile: all segments are forward and the segments over a pixel are the chain around its
//      points, so the first point of each pixel writes one 16-bit word (first id | count) -- no atomics.  Tile with
//      a fold: every polyline point drops its index into the (fixed-capacity) slot list of the output pixel it
//      falls in, every forward segment into the lists of the pixels it overlaps (returning LDS atomics; segments
//      longer than 3 pixels -- disocclusion bridges -- are written by the whole wave, 64 pixels per step)
//   4. EVALUATE pass 1: two pixels per lane; pixels with one point and two segments (~93 %) in straight-line code,
//      the others go onto a list
//   5. EVALUATE pass 2: the list packed onto lanes -- chain path (consecutive points, np + 1 segments: sub-interval
//      k belongs to chain segment k, verified by two compares), else the general search: the <= KP points sorted by
//      (x, index) in registers (== the reference's stable insertion sort restricted to the pixel), every listed
//      segment tested per sub-interval, largest interpolated |disparity| wins exactly like the reference
//      (float32/float64 typing of SURVEY.md Appendix A).  The idle waves write the depth-map output meanwhile.
//   6. results staged in LDS are written as full float4 rows into the SBS/TB slot together with the mask.
// Anything the fast path cannot prove order-independent -- exact closeness ties, no qualifying
// candidate, non-monotone centres, a slot list overflowing -- flags the ROW; flagged rows are redone
// by the general row kernel (cs_rowwarp.hip, `row_list`), whose results are authoritative.
// The neighbours of the first/last point of a pixel never enter the arithmetic (max(col, x) and
// min(col+1, x) discard them), which is why no global sort is needed.
#include "cs_common.h"
#include "cs_kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace cs {

#define PT_T 512      // output pixels per tile
// -DCS_PT_TIMESTAMPS=1: development build with per-phase timestamps of sampled workgroups (CS_DBG=20).
// PT_DEV_CUTOFF(n): CS_DBG=n returns after that phase (8 image staged, 9 LDS set up, 11 disparity staged, 12 registered,
// 18 pass 1, 13 before the stores).  The cut-offs stay in release builds on purpose: each is one scalar compare, and the
// basic-block boundaries they create keep the register allocator from spilling scalar state that is live across the
// phases (measured: 8 instead of 30 spilled VGPRs, all of them then in the rare general search; 6.2 vs 6.8 ms per
// 16 frames).  PT_DEV_IS(n): development-only behaviours (15 skips the evaluation, 41 memory-only pass).
#ifndef CS_PT_TIMESTAMPS
#define CS_PT_TIMESTAMPS 0
#endif
#define PT_DEV_CUTOFF(n) if (A.dbg == (n)) return
#if CS_PT_TIMESTAMPS
#define PT_DEV_IS(n) (A.dbg == (n))
#else
#define PT_DEV_IS(n) false
#endif
#define PT_THREADS 256  // == the 256 entries of the byte -> float table, one per thread

__constant__ csm::PowfTables c_pt_powf_tables = CS_POWF_TABLES_INIT;

struct TSub {
    bool sig64;
    double sig_d;
    float sig_f, center;
};
// sub-interval [max(col, a), min(col+1, b)] shrunk by EPSILON on both sides (reference :1957-1960, D32 typing)
__device__ __forceinline__ TSub pt_subinterval(int col, float a, float b) {
    const float eps32 = (float)1e-7;
    TSub s;
    bool from64 = !(a > (float)col), to64 = !(b < (float)(col + 1));
    if (from64 && to64) {
        double from_d = (double)col + 1e-7, to_d = (double)(col + 1) - 1e-7;
        s.sig64 = true;
        s.sig_d = to_d - from_d;
        s.sig_f = 0.0f;
        s.center = (float)(from_d + 0.5 * s.sig_d);
    } else {
        float ff = from64 ? (float)((double)col + 1e-7) : a + eps32;
        float tf = to64 ? (float)((double)(col + 1) - 1e-7) : b - eps32;
        s.sig64 = false;
        s.sig_d = 0.0;
        s.sig_f = tf - ff;
        s.center = ff + 0.5f * s.sig_f;
    }
    return s;
}

struct PolyTileArgs {
    int n, h, w, S;
    const float* image_f32;
    const uint8_t* image_u8;
    const uint32_t* stats;
    uint32_t* stats_rw;
    int scale_from_stats;
    float e32, conv32;
    EyeArgs eye[2];
    int neyes, single;
    uint8_t* out_u8;
    float* stereo; float* mask; float* depth_l; float* depth_r;
    int stereo_is_u8;
    int out_h, out_w;
    uint8_t* rowflag;  // [n][h] set to 1 when the row must be redone by the general kernel
    const uint32_t* tilemap; const float* gray; int tm_words;   // lazy depth-blur tiles (cs_common.h) or null
    int dbg;           // env CS_DBG: 14 = count the pixels per evaluation path into the spare stats words, 17 = no exponent
                       // shortcuts (tests compare the two); more in development builds, see CS_PT_TIMESTAMPS
};

// PT_KP / PT_KS: polyline points / forward segments per output pixel the fast path can hold (more -> row redo)
template <int SHARP, int PT_KP, int PT_KS, int MINW>
// The leading scalar arguments are the ones the first global loads depend on: with -amdgpu-kernarg-preload-count=16 they
// arrive in SGPRs with the wave instead of through scalar-memory round trips (a by-value struct is not preloaded).
__global__ void __launch_bounds__(PT_THREADS, MINW)
k_polytile(const float* __restrict__ hot_image, const float* __restrict__ hot_depth0, const float* __restrict__ hot_depth1,
           int hot_w, int hot_h, int hot_S, int hot_single, PolyTileArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // development (-DCS_PT_TIMESTAMPS=1 and CS_DBG=20): per-phase latency of sampled workgroups, summed into the spare stats
    // words of frame `phase`.  Compiled out by default: even the untaken branches delay the kernel-argument loads.
#if CS_PT_TIMESTAMPS
    const bool rec_wg = A.dbg == 20 && blockIdx.x % 61 == 0;  // sampled: the hot atomics perturb
    long long t_prev = wall_clock64();
    auto stamp = [&](int phase) {
        if (rec_wg && A.stats_rw && threadIdx.x == 0) {
            long long t = wall_clock64();
            atomicAdd(&A.stats_rw[(size_t)(phase % A.n) * ST_WORDS + 12], (unsigned)(t - t_prev));
            atomicAdd(&A.stats_rw[(size_t)(phase % A.n) * ST_WORDS + 13], 1u);
            t_prev = t;
        }
    };
#else
    constexpr bool rec_wg = false;
    auto stamp = [](int) {};
#endif
    const int tiles = (hot_w + PT_T - 1) / PT_T;
    // (eye = slowest grid dimension: pairing the two eyes of a tile on one XCD so that the second finds the image row in
    // that L2 was measured 13 % SLOWER -- the eyes' output streams then hit the same HBM channels at the same time)
    // grid = (tiles x 8 rows, rows / 8, frames x eyes), decoded with shifts: no integer division (~30 scalar instructions),
    // and all tiles of a row run on one XCD (workgroup b -> XCD b % 8), so the halo columns neighbouring tiles share come
    // from that L2 (cs_polypoint.hip has the measurements)
    const int bx = blockIdx.x;
    // (two-eye launches: blockIdx.y interleaves the eyes by row groups, cs_common.h eye_group_decode; z = frame)
    int yrow = blockIdx.y, eyei = hot_single;
    if (hot_single < 0) eye_group_decode((int)blockIdx.y, yrow, eyei);
    const int row = yrow * 8 + (bx & 7);
    if (row >= hot_h) return;
    const int tile = bx >> 3;
    const int frame = blockIdx.z;
    // (the eye's arguments are selected field by field: a dynamically indexed kernel-argument array costs a second,
    // dependent scalar-memory round trip before the first global load can be issued)
    EyeArgs E;
    E.depth = eyei ? hot_depth1 : hot_depth0;
    E.div32 = eyei ? A.eye[1].div32 : A.eye[0].div32;
    E.sep32 = eyei ? A.eye[1].sep32 : A.eye[0].sep32;
    E.enabled = eyei ? A.eye[1].enabled : A.eye[0].enabled;
    E.st_min = eyei ? A.eye[1].st_min : A.eye[0].st_min;
    E.st_max = eyei ? A.eye[1].st_max : A.eye[0].st_max;
    E.xoff = eyei ? A.eye[1].xoff : A.eye[0].xoff;
    E.yoff = eyei ? A.eye[1].yoff : A.eye[0].yoff;
    const bool eye_on = E.enabled && !PT_DEV_IS(41);
    const int w = hot_w, h = hot_h;
    const int o0 = tile * PT_T, wt = min(PT_T, w - o0);
    const int s0 = max(0, (o0 - hot_S - 1) & ~3), s1 = min(w, o0 + wt + hot_S + 1), ns = s1 - s0;  // s0 % 4 == 0: float4 staging
    const int nsmax = PT_T + 2 * hot_S + 6;

    // Local point ids: 0 = left sentinel (x = -w), then the points of source s0 + j in reference order
    // (soft: 1 + j; sharp: 1 + 2j and 2 + 2j), last = right sentinel (x = 2w).  The sentinels only take
    // part when the staged range touches the frame border.
    const int npts = (SHARP ? 2 * ns : ns) + 2;
    const int nptmax = (SHARP ? 2 * nsmax : nsmax) + 2;

    // LDS carve
    float* lut = (float*)smem;                                                  // [256] k / 255, filled after the staging (shares
    csm::PowfTables* tabs = (csm::PowfTables*)smem;                             //       its first half with the powf tables)
    static_assert(sizeof(csm::PowfTables) <= 1024, "lut / tables overlay");
    float* px = (float*)(smem + 1024);                                          // [nptmax] x of point o
    float* pz = px + ((nptmax + 3) & ~3);                                       // [nptmax] |coord_d| of point o
    uint32_t* img = (uint32_t*)(pz + ((nptmax + 3) & ~3));                      // [nsmax] R | G<<8 | B<<16
    uint16_t* cnt = (uint16_t*)(img + ((nsmax + 3) & ~3));                      // [PT_T] per-pixel registration word (see REGISTER)
    uint16_t* plist = cnt + PT_T;                                               // [PT_T] pixels the fastest path left over
    uint16_t* pts = plist + PT_T;                                               // [PT_T][PT_KP]  (fold tiles only)
    uint16_t* sgs = pts + PT_T * PT_KP;                                         // [PT_T][PT_KS]  (fold tiles only)
    uint16_t* pseg = sgs;                                                       // [PT_T] monotone tiles: a segment passing through the pixel
    uint8_t* res = (uint8_t*)(sgs + PT_T * PT_KS);                              // [3*PT_T]
    uint8_t* dep8 = res + align16(3 * PT_T);                                    // [PT_T] depth-map output code of this eye
    int* flags = (int*)(dep8 + PT_T);                                           // [4]

    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    const size_t rowpix = ((size_t)frame * h + row) * w;
    // ---- all global loads of the tile are issued first (one memory round trip): the depth row of the halo'ed range into
    // registers (consumed after the image conversion), the first 4-pixel group of the image row per thread; the LDS
    // initialisation and the copy of the powf tables (another global load) follow in their shadow
    constexpr int PT_PF = 3;
    const float* drow = E.depth + rowpix;
    float dpre[PT_PF];
    // lazy depth-blur tiles: edge-free tiles come from the gray depth (times the frame's x255 scale)
    const bool lazy = A.tilemap != nullptr;
    LazySel Z;
    if (lazy) Z = lazy_select(A.tilemap, A.tm_words, frame, h, row, s0, reinterpret_cast<const char*>(drow + s0),
                              reinterpret_cast<const char*>(A.gray + rowpix + s0), st[ST_SCALE255]);
    auto depth_at = [&](int j) {   // column s0 + j of this eye's depth row
        if (!lazy) return drow[s0 + j];
        float mul;
        const float v = lazy_load(Z, (uint32_t)(s0 + j), (uint32_t)j, mul);
        return v * mul;
    };
#pragma unroll
    for (int k = 0; k < PT_PF; k++) {
        const int j = tid + k * PT_THREADS;
        dpre[k] = j < ns ? depth_at(j) : 0.0f;
    }
    const int nq = (hot_image && (w & 3) == 0) ? ns / 4 : 0;  // (rowpix + s0) % 4 == 0 -> 16-byte aligned groups of 4 pixels
    const float4* s4 = reinterpret_cast<const float4*>(hot_image + (rowpix + s0) * 3);
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
    if (tid < nq) { q0 = s4[3 * tid]; q1 = s4[3 * tid + 1]; q2 = s4[3 * tid + 2]; }
    stamp(10);  // (development) loads issued
    if (rec_wg) { __builtin_amdgcn_s_waitcnt(0); stamp(11); }  // loads arrived
    const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
    const float dmin = eye_on ? csm::ord2f(st[E.st_min]) : 0.0f, dmax = eye_on ? csm::ord2f(st[E.st_max]) : 0.0f;

    // the powf tables go to LDS only when every point needs the clone (exponents other than 1 and 2); the 0.4 % risky
    // squares read them from constant memory (cs_polypoint.hip: +2 % / +6 %)
    const bool all_powf = A.dbg == 17 || !(A.e32 == 2.0f || A.e32 == 1.0f);
    if (all_powf) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_pt_powf_tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(tabs);
        for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += PT_THREADS) dst[i] = src[i];
    }
    for (int i = tid; i < PT_T / 2; i += PT_THREADS) reinterpret_cast<uint32_t*>(cnt)[i] = 0;
    if (tid == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0; }
    PT_DEV_CUTOFF(9);
    // stage the source pixels of the halo'ed range as packed uint8 RGB (reference :1508)
    if (hot_image) {
        const float* src = hot_image + (rowpix + s0) * 3;
        auto pack4 = [&](int i, float4 v0, float4 v1, float4 v2) {
            float f[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
            uint32_t pk[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t r = (uint32_t)(int)fminf(fmaxf(f[3 * k] * 255.0f, 0.0f), 255.0f);
                uint32_t g = (uint32_t)(int)fminf(fmaxf(f[3 * k + 1] * 255.0f, 0.0f), 255.0f);
                uint32_t b = (uint32_t)(int)fminf(fmaxf(f[3 * k + 2] * 255.0f, 0.0f), 255.0f);
                pk[k] = r | (g << 8) | (b << 16);
            }
            reinterpret_cast<uint4*>(img)[i] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        };
        if (tid < nq) pack4(tid, q0, q1, q2);
        for (int i = tid + PT_THREADS; i < nq; i += PT_THREADS) pack4(i, s4[3 * i], s4[3 * i + 1], s4[3 * i + 2]);
        for (int j = 4 * nq + tid; j < ns; j += PT_THREADS) {
            uint32_t r = (uint32_t)(int)fminf(fmaxf(src[3 * j] * 255.0f, 0.0f), 255.0f);
            uint32_t g = (uint32_t)(int)fminf(fmaxf(src[3 * j + 1] * 255.0f, 0.0f), 255.0f);
            uint32_t b = (uint32_t)(int)fminf(fmaxf(src[3 * j + 2] * 255.0f, 0.0f), 255.0f);
            img[j] = r | (g << 8) | (b << 16);
        }
    } else {
        const uint8_t* src = A.image_u8 + (rowpix + s0) * 3;
        for (int j = tid; j < ns; j += PT_THREADS)
            img[j] = (uint32_t)src[3 * j] | ((uint32_t)src[3 * j + 1] << 8) | ((uint32_t)src[3 * j + 2] << 16);
    }
    __syncthreads();  // tables ready
    stamp(1);
    PT_DEV_CUTOFF(8);
    if (!eye_on) {
        for (int q = tid; q < wt; q += PT_THREADS) dep8[q] = csm::f32_to_u8_wrap((depth_at(o0 + q - s0) * scale) * 255.0f);
    }
    if (eye_on) {
        const bool flat = dmax == dmin;
        const float range = dmax - dmin;
        bool fold = false;
        // everything before the power: the depth-map output code, the normalised depth's sign and magnitude
        auto pre = [&](int j, float draw, float& sgn, float& ax) {
            const float d = draw * scale;
            // depth-map output of this column: (depth*255).astype(uint8) wraps mod 256 (quirk Q7)
            const int qq = s0 + j - o0;
            if (qq >= 0 && qq < wt) dep8[qq] = csm::f32_to_u8_wrap(d * 255.0f);
            const float nd = flat ? 0.0f - A.conv32 : ((d - dmin) / range) - A.conv32;
            sgn = nd >= 0.0f ? 1.0f : -1.0f;
            ax = fabsf(nd);
        };
        // everything after it: the point's x (and |disparity|) into LDS, fold detection
        auto post = [&](int j, float sgn, float pw) {
            const float cdj = (sgn * pw) * E.div32;                                     // coord_d   (:1926)
            const float x = ((float)(s0 + j) + 0.5f + cdj) + E.sep32;                   // coord_x   (:1927)
            const float z = fabsf(cdj);
            // fold detection (is the polyline strictly increasing in x?): the right neighbour's x sits in the next lane;
            // the pairs across wave chunks and the sentinel pairs are checked after the barrier
            const float xn = __shfl_down(x, 1);
            const bool has_next = lane != 63 && j + 1 < ns;
            if (SHARP) {
                const float xa = x - (float)0.45, xb = x + (float)0.45;
                px[1 + 2 * j] = xa; pz[1 + 2 * j] = z;
                px[2 + 2 * j] = xb; pz[2 + 2 * j] = z;
                fold = fold || !(xa < xb) || (has_next && !(xb < xn - (float)0.45));
            } else {
                px[1 + j] = x; pz[1 + j] = z;
                fold = fold || (has_next && !(x < xn));
            }
        };
        // |nd| ** exponent, bit-exact with glibc's powf.  Exponents 2.0 (widget default) and 1.0 take the shortcuts of
        // cs_math.h (square_or_flag): the 0.4 % "risky" arguments of the square go through the full routine, batched per
        // wave after the loop.
        const int pow_mode = all_powf ? 0 : (A.e32 == 2.0f ? 2 : 1);
        auto square = [&](float ax, bool& risky) { return csm::square_or_flag(ax, risky); };
        if (pow_mode == 0) {
#pragma unroll
            for (int k = 0; k < PT_PF; k++) {
                const int j = tid + k * PT_THREADS;
                if (j < ns) {
                    float sgn, ax;
                    pre(j, dpre[k], sgn, ax);
                    post(j, sgn, csm::powf_exact_simt(ax, A.e32, tabs));
                }
            }
        } else {
            float sg[PT_PF], axs[PT_PF], pw[PT_PF];
            unsigned risk = 0;
#pragma unroll
            for (int k = 0; k < PT_PF; k++) {
                const int j = tid + k * PT_THREADS;
                sg[k] = 1.0f; axs[k] = 0.0f; pw[k] = 0.0f;
                if (j < ns) {
                    pre(j, dpre[k], sg[k], axs[k]);
                    bool r = false;
                    pw[k] = pow_mode == 1 ? axs[k] : square(axs[k], r);
                    risk |= r ? 1u << k : 0u;
                }
            }
            if (__any(risk != 0u)) {  // the full routine for the risky arguments: one pass per wave, rarely two (behind its own
                asm volatile("" ::: "memory");   // branch: its constant set-up is otherwise hoisted in front of the test)
                do {
                    float xin = 1.0f;
                    int sel = -1;
#pragma unroll
                    for (int k = PT_PF - 1; k >= 0; k--) if (risk & (1u << k)) { xin = axs[k]; sel = k; }
                    const float r = csm::powf_exact_simt(xin, A.e32, &c_pt_powf_tables);
#pragma unroll
                    for (int k = 0; k < PT_PF; k++) if (sel == k) pw[k] = r;
                    risk &= risk - 1u;
                } while (__any(risk != 0u));
            }
#pragma unroll
            for (int k = 0; k < PT_PF; k++) {
                const int j = tid + k * PT_THREADS;
                if (j < ns) post(j, sg[k], pw[k]);
            }
        }
        for (int j = tid + PT_PF * PT_THREADS; j < ns; j += PT_THREADS) {  // (halos beyond 128 columns only)
            float sgn, ax;
            pre(j, depth_at(j), sgn, ax);
            bool r = false;
            float pwv = pow_mode == 1 ? ax : (pow_mode == 2 ? square(ax, r) : 0.0f);
            if (pow_mode == 0 || __any(r)) {
                const float full = pow_mode == 0 ? csm::powf_exact_simt(ax, A.e32, tabs) : csm::powf_exact_simt(ax, A.e32, &c_pt_powf_tables);
                pwv = (pow_mode == 0 || r) ? full : pwv;
            }
            post(j, sgn, pwv);
        }
        if (tid == 0) {
            px[0] = (float)(-1.0 * w); pz[0] = 0.0f;
            px[npts - 1] = (float)(2.0 * w); pz[npts - 1] = 0.0f;
        }
        if (fold) flags[3] = 1;
    }
    __syncthreads();
    stamp(2);

    PT_DEV_CUTOFF(11);
    lut[tid] = (float)tid / 255.0f;  // (PT_THREADS == 256) the powf tables underneath are dead now; read in the store phase
    const bool left_edge = s0 == 0, right_edge = s1 == w;
    bool hazard = false;

    // ---- REGISTER ----------------------------------------------------------------------------------------
    // Tiles whose staged polyline is strictly increasing in x (no fold: the common case away from occluding depth
    // edges) take the MONO pass: every segment is forward and the segments over a pixel are exactly the chain around
    // its points, which are consecutive ids -- so each pixel only needs its first point id and the number of points,
    // written with one plain 16-bit store by the point that sees a different pixel to its left (cnt = first id |
    // points << 12, 0 = no point), and pseg = the segment bridging a pixel without points.  No atomics, nothing per
    // segment.  Tiles with a fold take the FULL pass: cnt = points in the pixel (low 8 bits) | forward segments
    // overlapping it (high 8) and the fixed-capacity id lists pts / sgs, filled through returning LDS atomics (on the
    // 32-bit word that holds two pixels' counters).
    const int ofirst = left_edge ? 0 : 1, olast = right_edge ? npts - 1 : npts - 2;
    bool mono = flags[3] == 0;
    if (eye_on) {  // the pairs the staging loop could not see; every wave computes the same answer, no barrier
        const int nb = (ns - 1) >> 6;  // pairs (j, j+1) with j = 64 b + 63
        bool fold = false;
        for (int b0 = 0; b0 < nb + 2; b0 += 64) {
            const int b = b0 + lane;
            int o = -1;
            if (b < nb) o = SHARP ? 2 + 2 * (64 * b + 63) : 1 + 64 * b + 63;
            else if (b == nb && left_edge) o = 0;
            else if (b == nb + 1 && right_edge) o = npts - 2;
            fold = fold || (o >= 0 && !(px[o] < px[o + 1]));
        }
        mono = mono && !__any(fold);
    }
    if (eye_on && mono) {
        for (int o = ofirst + tid; o <= olast; o += PT_THREADS) {
            const float x0 = px[o];
            const float f0 = floorf(x0);
            const float fm = floorf(px[max(o - 1, 0)]), f1 = floorf(px[min(o + 1, npts - 1)]);
            if (x0 >= (float)o0 && x0 < (float)(o0 + wt) && fm != f0) {   // the first point of its pixel (never a sentinel)
                int np = 1;  // points of the pixel = the run of ids with the same floor(x); the right sentinel ends every run
                float fn = f1;
                while (fn == f0 && np < 15) { np++; fn = floorf(px[min(o + np, npts - 1)]); }
                cnt[(int)x0 - o0] = (uint16_t)(o | (np << 12));   // ids < 4096: the halo limit keeps npts below that
            }
            // pixels strictly between the end pixels of segment o -> o+1 (disocclusion bridges)
            if (o < olast && f1 - f0 >= 2.0f && !(f1 <= (float)o0 || f0 >= (float)(o0 + wt - 1))) {
                const int pa = f0 < (float)o0 ? o0 : (int)f0 + 1;
                const int pb = f1 > (float)(o0 + wt - 1) ? o0 + wt - 1 : (int)f1 - 1;
                for (int p = pa; p <= pb; p++) pseg[p - o0] = (uint16_t)o;
            }
        }
    } else if (eye_on) {
        unsigned* cntw = reinterpret_cast<unsigned*>(cnt);
        const int niter = (olast - ofirst + 1 + PT_THREADS - 1) / PT_THREADS;
        for (int it = 0; it < niter; it++) {
            const int o = ofirst + it * PT_THREADS + tid;
            const bool live = o <= olast;
            int p0 = 1, p1 = 0;
            float f0 = 0.0f, f1 = 0.0f;
            if (live) {
                const float x0 = px[o];
                if (x0 >= (float)o0 && x0 < (float)(o0 + wt)) {
                    const int q = (int)x0 - o0;
                    unsigned idx = (atomicAdd(&cntw[q >> 1], 1u << ((q & 1) * 16)) >> ((q & 1) * 16)) & 0xffu;
                    if (idx < PT_KP) pts[q * PT_KP + idx] = (uint16_t)o;
                    else hazard = true;
                }
                if (o < olast) {  // segment o -> o+1
                    const float x1 = px[o + 1];
                    if (x0 < x1) {  // reversed / degenerate segments are never active
                        f0 = floorf(x0); f1 = floorf(x1);
                        if (!(f1 < (float)o0 || f0 > (float)(o0 + wt - 1))) {
                            p0 = f0 < (float)o0 ? o0 : (int)f0;
                            p1 = f1 > (float)(o0 + wt - 1) ? o0 + wt - 1 : (int)f1;
                        }
                    }
                }
            }
            const bool is_long = p1 - p0 > 3;
            if (!is_long) {
#pragma unroll
                for (int t = 0; t < 4; t++) {  // at most 4 pixels: predicated, no divergent loop
                    const int p = p0 + t;
                    if (p <= p1) {
                        const int qp = p - o0;
                        unsigned idx = (atomicAdd(&cntw[qp >> 1], 0x100u << ((qp & 1) * 16)) >> ((qp & 1) * 16 + 8)) & 0xffu;
                        if (idx < PT_KS) sgs[(p - o0) * PT_KS + idx] = (uint16_t)o;
                        else hazard = true;
                    }
                }
            }
            // long segments (disocclusion bridges): the whole wave writes them, 64 pixels per step
            unsigned long long m = __ballot(is_long);
            while (m) {
                int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                int lp0 = __shfl(p0, src), lp1 = __shfl(p1, src), lo = __shfl(o, src);
                for (int p = lp0 + lane; p <= lp1; p += 64) {
                    const int qp = p - o0;
                    unsigned idx = (atomicAdd(&cntw[qp >> 1], 0x100u << ((qp & 1) * 16)) >> ((qp & 1) * 16 + 8)) & 0xffu;
                    if (idx < PT_KS) sgs[(p - o0) * PT_KS + idx] = (uint16_t)lo;
                    else hazard = true;
                }
            }
        }
    }
    __syncthreads();
    stamp(mono ? 4 : 3);  // fold tiles are accounted separately

    // ---- EVALUATE ----------------------------------------------------------------------------------------
    // per-pixel constants of the float64 ("Python float") branch of the sub-interval arithmetic
    struct PixC { double sig_dd; float ff64, tf64, center64; };
    auto pix_consts = [](int col) {
        PixC P;
        const double from_d = (double)col + 1e-7, to_d = (double)(col + 1) - 1e-7;
        P.sig_dd = to_d - from_d;
        P.ff64 = (float)from_d; P.tf64 = (float)to_d; P.center64 = (float)(from_d + 0.5 * P.sig_dd);
        return P;
    };
    const float eps32 = (float)1e-7;
    auto put = [&](int q, uint32_t rgb) {
        res[3 * q] = (uint8_t)rgb; res[3 * q + 1] = (uint8_t)(rgb >> 8); res[3 * q + 2] = (uint8_t)(rgb >> 16);
    };
    // ---- chain path: the pixel's np points are CONSECUTIVE polyline points o1 .. o1+np-1 with strictly increasing x
    // inside the pixel and exactly np+1 forward segments overlap it -- then those are the chain segments (incoming,
    // internal, outgoing) and sub-interval k can only be covered by chain segment k = (o1-1+k -> o1+k): two compares
    // verify it, no list search.  np == 0: the one segment passing through.  Returns false when the pixel needs the
    // general search.  Called by whole waves (ballots inside); `act` masks the lanes that hold a pixel.
    auto eval_chain = [&](bool act, int q, uint32_t& rgb) -> bool {
        const int col = o0 + q;
        const unsigned c = cnt[q];
        int npr, cover, o1;  // points in the pixel, forward segments over it, its smallest point id (np == 0: the segment's end)
        if (mono) {
            const int pf = (int)(c & 0xfffu);
            npr = (int)(c >> 12);
            cover = npr + 1;
            o1 = npr ? pf : (int)pseg[q] + 1;
        } else {
            npr = (int)(c & 0xffu); cover = (int)(c >> 8);
            int om = 0xffff;
#pragma unroll
            for (int k = 0; k < PT_KP; k++) om = min(om, k < npr ? (int)pts[q * PT_KP + k] : 0xffff);
            o1 = npr ? om : (int)sgs[q * PT_KS] + 1;
        }
        const int np = act ? min(npr, PT_KP) : 0;
        const PixC P = pix_consts(col);
        const double sig_dd = P.sig_dd;
        const float ff64 = P.ff64, tf64 = P.tf64, center64 = P.center64;
        // wave-uniform bound: unrolled bodies beyond it are skipped by scalar branches
        // (ballots, not shuffles: a shuffle reduction is a chain of LDS-crossbar round trips)
        int wnp = 0;
#pragma unroll
        for (int t = 1; t <= PT_KP; t++) wnp = __any(np >= t) ? t : wnp;
        bool chain = act && cover == npr + 1 && npr < PT_KP;
        o1 = chain ? o1 : 1;
        float cx[PT_KP + 2];     // cx[k] = x of point o1 - 1 + k, k = 0 .. np + 1
        uint32_t cc[PT_KP + 2];  // colour of the source pixel that point refers to
        int cj[PT_KP + 2];
        if (__any(chain)) {
#pragma unroll
            for (int k = 0; k < PT_KP + 2; k++) {
                if (k <= wnp + 1) {
                    const int o = chain && k <= np + 1 ? o1 - 1 + k : 0;
                    cx[k] = px[o];
                    cj[k] = min(max(SHARP ? (o - 1) >> 1 : o - 1, 0), ns - 1);
                    cc[k] = img[cj[k]];
                } else { cx[k] = 0.0f; cj[k] = 0; cc[k] = 0; }
            }
            // every chain segment must be forward (then it is registered for this pixel, and with nsg == np + 1
            // the registered set IS the chain: no other layer passes through the pixel)
#pragma unroll
            for (int k = 0; k <= PT_KP; k++)
                if (k <= wnp) chain = chain && (k > np || cx[k] < cx[k + 1]) && (k < 1 || k > np || cx[k] < (float)(col + 1));
            float color0 = 0.5f, color1 = 0.5f, color2 = 0.5f;
            float prev = (float)col;
#pragma unroll
            for (int k = 0; k <= PT_KP; k++) {
                if (k <= wnp) {
                    const bool live = chain && k <= np;
                    const float a = k == 0 ? -INFINITY : cx[k];
                    const float b = k < np ? cx[k + 1] : INFINITY;
                    const bool from64 = !(a > (float)col), to64 = !(b < (float)(col + 1));
                    const bool sig64 = from64 && to64;
                    const float ff = from64 ? ff64 : a + eps32;
                    const float tf = to64 ? tf64 : b - eps32;
                    const float sig_f = tf - ff;
                    const float center = sig64 ? center64 : ff + 0.5f * sig_f;
                    const bool work = live && (sig64 ? sig_dd != 0.0 : sig_f != 0.0f);
                    // chain segment k must be the active one: x0 < centre <= x1, centres monotone inside the pixel
                    const bool ok = (cx[k] < center) && !(cx[k + 1] < center) && !(center < prev) &&
                                    !(center > (float)(col + 1));
                    chain = chain && (!live || !work || ok) && (!live || (!(center < prev) && !(center > (float)(col + 1))));
                    prev = live ? center : prev;
                    const float ip_k = (center - cx[k]) / (cx[k + 1] - cx[k]);
                    const float om = 1.0f - ip_k;
                    const float sg = sig64 ? (float)sig_dd : sig_f;
                    const uint32_t il = cc[k], ir = cc[k + 1];
                    const float l0 = (float)(il & 0xffu), l1 = (float)((il >> 8) & 0xffu), l2 = (float)((il >> 16) & 0xffu);
                    const float r0 = (float)(ir & 0xffu), r1 = (float)((ir >> 8) & 0xffu), r2 = (float)((ir >> 16) & 0xffu);
                    float n0 = color0 + (l0 * om + r0 * ip_k) * sg;
                    float n1 = color1 + (l1 * om + r1 * ip_k) * sg;
                    float n2 = color2 + (l2 * om + r2 * ip_k) * sg;
                    const bool flat = cj[k] == cj[k + 1];
                    if (__any(work && flat)) {
                        if (flat) {
                            if (sig64) {
                                n0 = (float)((double)color0 + (double)l0 * sig_dd);
                                n1 = (float)((double)color1 + (double)l1 * sig_dd);
                                n2 = (float)((double)color2 + (double)l2 * sig_dd);
                            } else {
                                n0 = color0 + l0 * sig_f;
                                n1 = color1 + l1 * sig_f;
                                n2 = color2 + l2 * sig_f;
                            }
                        }
                    }
                    color0 = work ? n0 : color0;
                    color1 = work ? n1 : color1;
                    color2 = work ? n2 : color2;
                }
            }
            rgb = (uint32_t)csm::f32_to_u8_wrap(color0) | ((uint32_t)csm::f32_to_u8_wrap(color1) << 8) |
                  ((uint32_t)csm::f32_to_u8_wrap(color2) << 16);
        }
        return chain;
    };

    PT_DEV_CUTOFF(12);
    // ---- pass 1, one output pixel per lane.  soft: the fastest path per lane -- exactly ONE polyline point o in
    // the pixel and two segments over it (flat and gently sloped regions): the two pieces [col, x] and [x, col+1]
    // belong to the segments (o-1 -> o) and (o -> o+1), verified below; straight-line code.  Every other pixel goes
    // onto the tile's list `plist`, which pass 2 works off densely packed (a few percent of the pixels, but spread
    // over a fifth of the waves).  sharp (two points per source pixel): the chain path directly.
    // ---- general search (fold tiles): the pixel's points sorted, every registered segment tested per sub-interval.
    // Called by whole waves; `pend` masks the lanes that hold a pixel.
    auto eval_generic = [&](bool pend, int q) {
        const int col = o0 + q;
        const unsigned c = pend ? cnt[q] : 0u;
        const int np = min((int)(c & 0xffu), PT_KP), nsg = min((int)(c >> 8), PT_KS);
        int wnp = 0, wns = 0;
#pragma unroll
        for (int t = 1; t <= PT_KP; t++) wnp = __any(np >= t) ? t : wnp;
#pragma unroll
        for (int t = 1; t <= PT_KS; t++) wns = __any(nsg >= t) ? t : wns;
        const PixC P = pix_consts(col);
        const double sig_dd = P.sig_dd;
        const float ff64 = P.ff64, tf64 = P.tf64, center64 = P.center64;
        // the pixel's points sorted by (x, id) == the reference's stable insertion sort inside the pixel
        float xs[PT_KP];
        int os[PT_KP];
#pragma unroll
        for (int k = 0; k < PT_KP; k++) { xs[k] = INFINITY; os[k] = 0x7fffffff; }
#pragma unroll
        for (int k = 0; k < PT_KP; k++) {
            if (k < wnp) {
                int o = k < np ? (int)pts[q * PT_KP + k] : 0x7fffffff;
                float x = k < np ? px[o] : INFINITY;
#pragma unroll
                for (int m2 = 0; m2 <= k; m2++) {
                    bool lt = x < xs[m2] || (x == xs[m2] && o < os[m2]);
                    float tx = lt ? xs[m2] : x; int to = lt ? os[m2] : o;
                    xs[m2] = lt ? x : xs[m2]; os[m2] = lt ? o : os[m2];
                    x = tx; o = to;
                }
            }
        }
        // forward segments overlapping this pixel: only their ids stay in registers (their end points are re-read from
        // LDS in the scan: ten registers less, which the allocator otherwise spills to scratch on this waited-for path)
        int so[PT_KS];
#pragma unroll
        for (int k = 0; k < PT_KS; k++) so[k] = (k < wns && k < nsg) ? (int)sgs[q * PT_KS + k] : -1;
        float color0 = 0.5f, color1 = 0.5f, color2 = 0.5f;
        float prev = (float)col, a = -INFINITY;
        for (int k = 0; k <= wnp; k++) {   // wave-uniform trip count; lanes with k > np idle
            const bool live = pend && k <= np;
            float b = INFINITY;
#pragma unroll
            for (int m2 = 0; m2 < PT_KP; m2++) b = (m2 == k && k < np) ? xs[m2] : b;
            // sub-interval [max(col, a), min(col+1, b)] shrunk by EPSILON (reference :1957-1960, D32 typing)
            const bool from64 = !(a > (float)col), to64 = !(b < (float)(col + 1));
            const bool sig64 = from64 && to64;
            const float ff = from64 ? ff64 : a + eps32;
            const float tf = to64 ? tf64 : b - eps32;
            const float sig_f = tf - ff;
            const float center = sig64 ? center64 : ff + 0.5f * sig_f;
            a = live ? b : a;
            if (live && (center < prev || center > (float)(col + 1))) hazard = true;
            prev = live ? center : prev;
            const bool work = live && (sig64 ? sig_dd != 0.0 : sig_f != 0.0f);  // a zero-length piece adds exactly 0
            // ONE scan of the pixel's listed segments: which are active at the centre (x0 < centre <= x1); with overlapping
            // layers (or none) the reference picks the largest interpolated |disparity| among candidates with
            // 0 < ip_k < 1, the first one on ties -> ties are order-dependent: flag.
            int nact = 0, pick = -1, nqual = 0, best = -1;
            float bc = (float)(-1e-7);
            bool tie = false;
#pragma unroll
            for (int e = 0; e < PT_KS; e++) {
                if (e < wns) {
                    const bool have = so[e] >= 0;
                    const int oe = have ? so[e] : 0;
                    const float e0 = px[oe], e1 = px[oe + 1];
                    const bool act = have && (e0 < center) && !(e1 < center);
                    nact += act ? 1 : 0;
                    pick = act ? e : pick;
                    const float ip_e = (center - e0) / (e1 - e0);
                    const bool qual = act && 0.0f < ip_e && ip_e < 1.0f;
                    const float cl = (1.0f - ip_e) * pz[oe] + ip_e * pz[oe + 1];
                    nqual += qual ? 1 : 0;
                    const bool better = qual && bc < cl;
                    tie = better ? false : (tie || (qual && cl == bc));
                    best = better ? e : best;
                    bc = better ? cl : bc;
                }
            }
            const bool multi = work && nact != 1;
            if (multi && (nqual == 0 || tie)) hazard = true;
            pick = (multi && best >= 0) ? best : pick;
            const bool contrib = work && pick >= 0;
            // colour contribution (reference :1981-1989, D32 typing); idle lanes compute on dummy operands
            int o = 1;
#pragma unroll
            for (int e = 0; e < PT_KS; e++)
                if (e < wns) o = (e == pick && so[e] >= 0) ? so[e] : o;
            const float x0 = contrib ? px[o] : 0.0f, x1 = contrib ? px[o + 1] : 1.0f;
            const int jl = min(max(SHARP ? (o - 1) >> 1 : o - 1, 0), ns - 1);
            const int jr = min(max(SHARP ? o >> 1 : o, 0), ns - 1);
            const uint32_t il = img[jl], ir = img[jr];
            // (the compiler folds these into v_cvt_f32_ubyte0/1/2)
            const float l0 = (float)(il & 0xffu), l1 = (float)((il >> 8) & 0xffu), l2 = (float)((il >> 16) & 0xffu);
            const float r0 = (float)(ir & 0xffu), r1 = (float)((ir >> 8) & 0xffu), r2 = (float)((ir >> 16) & 0xffu);
            const float ip_k = (center - x0) / (x1 - x0);
            const float om = 1.0f - ip_k;
            const float sg = sig64 ? (float)sig_dd : sig_f;
            float n0 = color0 + (l0 * om + r0 * ip_k) * sg;
            float n1 = color1 + (l1 * om + r1 * ip_k) * sg;
            float n2 = color2 + (l2 * om + r2 * ip_k) * sg;
            if (__any(contrib && jl == jr)) {  // segment inside one source pixel (sentinel pieces; every other 'sharp' piece)
                if (jl == jr) {
                    if (sig64) {
                        n0 = (float)((double)color0 + (double)l0 * sig_dd);
                        n1 = (float)((double)color1 + (double)l1 * sig_dd);
                        n2 = (float)((double)color2 + (double)l2 * sig_dd);
                    } else {
                        n0 = color0 + l0 * sig_f;
                        n1 = color1 + l1 * sig_f;
                        n2 = color2 + l2 * sig_f;
                    }
                }
            }
            color0 = contrib ? n0 : color0;
            color1 = contrib ? n1 : color1;
            color2 = contrib ? n2 : color2;
        }
        if (pend) put(q, (uint32_t)csm::f32_to_u8_wrap(color0) | ((uint32_t)csm::f32_to_u8_wrap(color1) << 8) |
                         ((uint32_t)csm::f32_to_u8_wrap(color2) << 16));
    };
    // straight-line code without branches or ballots: the two pixels of a lane are evaluated back to back so that the
    // scheduler overlaps their LDS round trips
    auto fast_px = [&](int q, bool cand, unsigned c, uint32_t& rgb) -> bool {
        const int col = o0 + q;
        // the float64 ("Python float") constants of pix_consts, rounded to float32 -- closed forms, checked against the
        // double arithmetic for every col < 2^24 (tests/test_cs_math_host.py): (float)(col + 1e-7) == col and
        // (float)(col + 1 - 1e-7) == col + 1 from col = 2 on, the centre is col + 0.5, the length rounds to 1 - 3 * 2^-24
        const float ff64 = col == 0 ? 0x1.ad7f2ap-24f : (col == 1 ? 0x1.000002p+0f : (float)col);
        const float tf64 = col == 0 ? 0x1.fffffcp-1f : (col == 1 ? 0x1.fffffep+0f : (float)(col + 1));
        const float center64 = (float)col + 0.5f;
        const float sig_dd = 0x1.fffffap-1f;  // (float)(to - from); never zero
        bool done = cand;
        const int o = cand ? (mono ? (int)(c & 0xfffu) : (int)pts[q * PT_KP]) : 1;
        const float xm = px[o - 1], x = px[o], xp = px[o + 1];
        const int j = o - 1;  // source column (local) of point o; o-1 >= 1 and o+1 <= npts-2 checked via jok
        const bool jok = j >= 1 && j + 1 <= ns - 1;
        const uint32_t ia = img[jok ? j - 1 : 0], ib = img[jok ? j : 0], ic = img[jok ? j + 1 : 0];
        // piece 0: [col, x]   (from = col + eps as Python float, to = x - eps as float32)
        const float tf0 = x - eps32;
        const float sig0 = tf0 - ff64;
        const float c0 = ff64 + 0.5f * sig0;
        // piece 1: [x, col+1] (from = x + eps if x > col else the Python-float col + eps; to = col+1-eps)
        const bool f64_1 = !(x > (float)col);
        const float ff1 = f64_1 ? ff64 : x + eps32;
        const float sig1 = tf64 - ff1;
        const float c1 = f64_1 ? center64 : ff1 + 0.5f * sig1;
        const float sg1 = f64_1 ? (float)sig_dd : sig1;
        const bool w0 = sig0 != 0.0f, w1 = f64_1 ? sig_dd != 0.0 : sig1 != 0.0f;
        // chain segments forward and covering their piece, centres monotone inside the pixel
        bool ok = jok && xm < x && x < xp;
        ok = ok && !(c0 < (float)col) && !(c1 < c0) && !(c1 > (float)(col + 1));
        ok = ok && (!w0 || (xm < c0 && !(x < c0))) && (!w1 || (x < c1 && !(xp < c1)));
        done = done && ok;
        const float ip0 = (c0 - xm) / (x - xm), ip1 = (c1 - x) / (xp - x);
        const float om0 = 1.0f - ip0, om1 = 1.0f - ip1;
        const float a0 = (float)(ia & 0xffu), a1 = (float)((ia >> 8) & 0xffu), a2 = (float)((ia >> 16) & 0xffu);
        const float b0 = (float)(ib & 0xffu), b1 = (float)((ib >> 8) & 0xffu), b2 = (float)((ib >> 16) & 0xffu);
        const float e0 = (float)(ic & 0xffu), e1 = (float)((ic >> 8) & 0xffu), e2 = (float)((ic >> 16) & 0xffu);
        float k0 = 0.5f, k1 = 0.5f, k2 = 0.5f;
        const float p0 = k0 + (a0 * om0 + b0 * ip0) * sig0, p1 = k1 + (a1 * om0 + b1 * ip0) * sig0,
                    p2 = k2 + (a2 * om0 + b2 * ip0) * sig0;
        k0 = w0 ? p0 : k0; k1 = w0 ? p1 : k1; k2 = w0 ? p2 : k2;
        const float r0 = k0 + (b0 * om1 + e0 * ip1) * sg1, r1 = k1 + (b1 * om1 + e1 * ip1) * sg1,
                    r2 = k2 + (b2 * om1 + e2 * ip1) * sg1;
        k0 = w1 ? r0 : k0; k1 = w1 ? r1 : k1; k2 = w1 ? r2 : k2;
        rgb = (uint32_t)csm::f32_to_u8_wrap(k0) | ((uint32_t)csm::f32_to_u8_wrap(k1) << 8) |
              ((uint32_t)csm::f32_to_u8_wrap(k2) << 16);
        return done;
    };
    // sharp: two points per source pixel, so the common pixel holds exactly TWO consecutive points o, o+1 and three
    // pieces [col, x_o] [x_o, x_o+1] [x_o+1, col+1] on the chain segments (o-1 -> o) (o -> o+1) (o+1 -> o+2), alternately
    // flat (inside one source pixel: colour * length) and interpolating.  Same checks as the chain path, straight-line.
    auto fast_px2 = [&](int q, bool cand, unsigned c, uint32_t& rgb) -> bool {
        const int col = o0 + q;
        const float ff64 = col == 0 ? 0x1.ad7f2ap-24f : (col == 1 ? 0x1.000002p+0f : (float)col);
        const float tf64 = col == 0 ? 0x1.fffffcp-1f : (col == 1 ? 0x1.fffffep+0f : (float)(col + 1));
        int o = 1;
        bool ok = cand;
        if (mono) o = cand ? (int)(c & 0xfffu) : 1;
        else {
            const int pa = pts[q * PT_KP], pb = pts[q * PT_KP + 1];
            o = cand ? min(pa, pb) : 1;
            ok = ok && max(pa, pb) == o + 1;
        }
        ok = ok && o >= 1 && o + 2 <= npts - 1;
        o = ok ? o : 1;
        const float x0 = px[o - 1], x1 = px[o], x2 = px[o + 1], x3 = px[o + 2];
        const int j0 = min(max((o - 2) >> 1, 0), ns - 1), j1 = min((o - 1) >> 1, ns - 1);
        const int j2 = min(o >> 1, ns - 1), j3 = min((o + 1) >> 1, ns - 1);
        const uint32_t i0 = img[j0], i1 = img[j1], i2 = img[j2], i3 = img[j3];
        const float colf = (float)col, col1 = (float)(col + 1);
        ok = ok && x0 < x1 && x1 < x2 && x2 < x3 && x1 < col1 && x2 < col1 && x2 > colf;
        float k0 = 0.5f, k1 = 0.5f, k2 = 0.5f, prev = colf;
        auto piece = [&](float xa, float xb, uint32_t ia, uint32_t ib, bool flat, float ff, float tf) {
            const float sig = tf - ff;
            const float cen = ff + 0.5f * sig;
            const bool work = sig != 0.0f;
            ok = ok && (!work || (xa < cen && !(xb < cen))) && !(cen < prev) && !(cen > col1);
            prev = cen;
            const float ip = (cen - xa) / (xb - xa);
            const float om = 1.0f - ip;
            const float l0 = (float)(ia & 0xffu), l1 = (float)((ia >> 8) & 0xffu), l2 = (float)((ia >> 16) & 0xffu);
            const float r0 = (float)(ib & 0xffu), r1 = (float)((ib >> 8) & 0xffu), r2 = (float)((ib >> 16) & 0xffu);
            const float m0 = flat ? l0 : l0 * om + r0 * ip, m1 = flat ? l1 : l1 * om + r1 * ip,
                        m2 = flat ? l2 : l2 * om + r2 * ip;
            const float n0 = k0 + m0 * sig, n1 = k1 + m1 * sig, n2 = k2 + m2 * sig;
            k0 = work ? n0 : k0; k1 = work ? n1 : k1; k2 = work ? n2 : k2;
        };
        piece(x0, x1, i0, i1, j0 == j1, ff64, x1 - eps32);
        piece(x1, x2, i1, i2, j1 == j2, !(x1 > colf) ? ff64 : x1 + eps32, x2 - eps32);
        piece(x2, x3, i2, i3, j2 == j3, x2 + eps32, tf64);
        rgb = (uint32_t)csm::f32_to_u8_wrap(k0) | ((uint32_t)csm::f32_to_u8_wrap(k1) << 8) |
              ((uint32_t)csm::f32_to_u8_wrap(k2) << 16);
        return ok;
    };
    auto fast_cand = [&](unsigned c) {
        if (SHARP) return mono ? (c >> 12) == 2u : c == 0x0302u;
        return mono ? (c >> 12) == 1u : c == 0x0201u;
    };
    if (!eye_on || PT_DEV_IS(15)) {
        for (int q = tid; q < wt; q += PT_THREADS) put(q, img[o0 + q - s0]);
    } else {
        for (int qa = tid; qa < wt; qa += 2 * PT_THREADS) {
            const bool vb = qa + PT_THREADS < wt;
            const int qb = vb ? qa + PT_THREADS : qa;
            const unsigned ca = cnt[qa], cb = cnt[qb];
            uint32_t ra = 0, rb = 0;
            bool da, db;
            if (SHARP) {
                da = fast_px2(qa, fast_cand(ca), ca, ra);
                db = fast_px2(qb, vb && fast_cand(cb), cb, rb);
            } else {
                da = fast_px(qa, fast_cand(ca), ca, ra);
                db = fast_px(qb, vb && fast_cand(cb), cb, rb);
            }
            if (da) put(qa, ra);
            else plist[atomicAdd((unsigned*)&flags[2], 1u)] = (uint16_t)qa;
            if (db) put(qb, rb);
            else if (vb) plist[atomicAdd((unsigned*)&flags[2], 1u)] = (uint16_t)qb;
        }
    }
    PT_DEV_CUTOFF(18);
    // this eye's depth-map output (byte code -> k/255, replicated over the three channels): depends on the staged depth only
    auto store_depth = [&](int t0, int nthr) {
        float* dd = (eyei == 0 ? A.depth_l : A.depth_r);
        if (!dd || A.out_u8) return;
        dd += (rowpix + o0) * 3;
        if ((wt & 3) == 0 && (w & 3) == 0) {
            float4* d4 = reinterpret_cast<float4*>(dd);
            for (int i = t0; i < (3 * wt) / 4; i += nthr) {  // 4 consecutive floats of the 3-channel row
                const int e = 4 * i, q0 = e / 3, r0 = e - 3 * q0;   // element e belongs to pixel e / 3
                const float va = lut[dep8[q0]], vb = lut[dep8[q0 + 1]];
                // r0 = 0: a a a b | r0 = 1: a a b b | r0 = 2: a b b b
                d4[i] = make_float4(va, r0 == 2 ? vb : va, r0 == 0 ? va : vb, vb);
            }
        } else {
            for (int q = t0; q < wt; q += nthr) {
                float v = lut[dep8[q]];
                dd[3 * q] = v; dd[3 * q + 1] = v; dd[3 * q + 2] = v;
            }
        }
    };
    {
        __syncthreads();
        // the leftovers below usually fit one wave; the other three write the depth-map output meanwhile
        if (tid >= 64) store_depth(tid - 64, PT_THREADS - 64);
        stamp(mono ? 5 : 8);
        // ---- pass 2 (soft): the chain path over the listed pixels, the general search for those it cannot do
        const int nlist = flags[2];
        for (int base = tid & ~63; base < nlist; base += PT_THREADS) {
            const int i = base + lane;
            const bool act = i < nlist;
            const int q = plist[act ? i : 0];
            uint32_t rgb = 0;
            bool ok = false;
            if (mono) {
                ok = eval_chain(act, q, rgb);
                if (ok) put(q, rgb);
                if (__any(act && !ok)) {
                    if (mono) hazard = hazard || (act && !ok);  // (not seen) -> row redo
                    else eval_generic(act && !ok, q);
                }
            } else {
                // fold tile: the lists exist, and the general search also does the pixels the chain path could do -- one
                // pass over the list instead of two dependent ones on the one wave the workgroup is waiting for
                eval_generic(act, q);
            }
            if (A.dbg == 14 && A.stats_rw) {  // development: how many pixels take which path
                unsigned long long ma = __ballot(act), mg = __ballot(act && !ok);
                if (lane == 0) {
                    atomicAdd(&A.stats_rw[(size_t)frame * ST_WORDS + 12], (unsigned)__popcll(ma));
                    atomicAdd(&A.stats_rw[(size_t)frame * ST_WORDS + 13], (unsigned)__popcll(mg));
                    atomicAdd(&A.stats_rw[(size_t)frame * ST_WORDS + 14], mg ? 1u : 0u);
                    atomicAdd(&A.stats_rw[(size_t)frame * ST_WORDS + 15], 1u);
                }
            }
        }
    }
    if (hazard) flags[0] = 1;
    __syncthreads();
    stamp(mono ? 7 : 6);
    if (flags[0]) {
        // the general kernel redoes this row (both eyes) and overwrites whatever is stored below
        if (tid == 0) A.rowflag[(size_t)frame * h + row] = 1;
    }
    PT_DEV_CUTOFF(13);
    // ---- store the tile ------------------------------------------------------------------------------
    if (A.out_u8) {
        uint8_t* dst = A.out_u8 + (rowpix + o0) * 3;
        if ((wt & 3) == 0 && (w & 3) == 0)  // (rowpix + o0) * 3 is then a multiple of 4
            for (int i = tid; i < (3 * wt) / 4; i += PT_THREADS)
                reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(res)[i];
        else
            for (int i = tid; i < 3 * wt; i += PT_THREADS) dst[i] = res[i];
    } else {
        const size_t o = ((size_t)frame * A.out_h + row + E.yoff) * A.out_w + E.xoff + o0;
        float* dst = A.stereo + o * 3;
        if (A.stereo_is_u8) {  // compact output for the multi-GPU all-gather: the uint8 codes k (value = k / 255)
            uint8_t* d8 = reinterpret_cast<uint8_t*>(A.stereo) + o * 3;
            if ((wt & 3) == 0 && (w & 3) == 0)
                for (int i = tid; i < (3 * wt) / 4; i += PT_THREADS)
                    reinterpret_cast<uint32_t*>(d8)[i] = reinterpret_cast<const uint32_t*>(res)[i];
            else
                for (int i = tid; i < 3 * wt; i += PT_THREADS) d8[i] = res[i];
        } else if ((wt & 3) == 0 && (w & 3) == 0) {
            float4* d4 = reinterpret_cast<float4*>(dst);
            for (int i = tid; i < (3 * wt) / 4; i += PT_THREADS) {
                uint32_t pk = reinterpret_cast<const uint32_t*>(res)[i];
                d4[i] = make_float4(lut[pk & 0xff], lut[(pk >> 8) & 0xff], lut[(pk >> 16) & 0xff], lut[pk >> 24]);
            }
        } else {
            for (int i = tid; i < 3 * wt; i += PT_THREADS) dst[i] = lut[res[i]];
        }
        float* m = A.mask + o;
        if (!A.mask) {   // (the per-eye intermediate of the anaglyph modes has no mask: k_anaglyph_compose writes the composite's)
        } else if ((wt & 3) == 0 && (w & 3) == 0) {
            float4* m4 = reinterpret_cast<float4*>(m);
            for (int i = tid; i < wt / 4; i += PT_THREADS) {
                const uint8_t* r = res + 12 * i;
                m4[i] = make_float4((r[0] | r[1] | r[2]) == 0 ? 1.0f : 0.0f, (r[3] | r[4] | r[5]) == 0 ? 1.0f : 0.0f,
                                    (r[6] | r[7] | r[8]) == 0 ? 1.0f : 0.0f, (r[9] | r[10] | r[11]) == 0 ? 1.0f : 0.0f);
            }
        } else {
            for (int q = tid; q < wt; q += PT_THREADS)
                m[q] = ((int)res[3 * q] + (int)res[3 * q + 1] + (int)res[3 * q + 2]) == 0 ? 1.0f : 0.0f;
        }
    }
    if (rec_wg) __builtin_amdgcn_s_waitcnt(0);
    stamp(9);
}

static size_t polytile_lds(int S, int sharp, int PT_KP, int PT_KS) {
    int nsmax = PT_T + 2 * S + 6;
    int nptmax = (sharp ? 2 * nsmax : nsmax) + 2;
    return 1024 + 2 * 4 * (size_t)((nptmax + 3) & ~3) + 4 * (size_t)((nsmax + 3) & ~3) +
           2 * PT_T + 2 * PT_T + 2 * PT_T * (PT_KP + PT_KS) + align16(3 * PT_T) + PT_T + 64;
}

// Largest halo the tiled path accepts.  The staged range is then 3.7x the tile (LDS 37 KB soft / 55 KB sharp: 2-4
// workgroups per CU), still several times faster than a whole row per workgroup; the 12-bit point ids of the fold-free
// registration word allow 2 * (512 + 2 * 764 + 6) + 2 points.
int polytile_max_halo() { return 700; }

// Launch the tiled fast path for the eyes of `A0` (SBS / TB / single-eye / uint8 outputs; no anaglyph).
// `rowflag` must be zeroed by the caller; afterwards the general kernel is run over the flagged rows.
hipError_t launch_polytile(int sharp, const RowArgs& R, int S, uint8_t* rowflag, hipStream_t stream) {
    PolyTileArgs A;
    A.n = R.n; A.h = R.h; A.w = R.w; A.S = S;
    A.image_f32 = R.image_f32; A.image_u8 = R.image_u8;
    A.stats = R.stats; A.stats_rw = R.stats_rw;
    A.scale_from_stats = R.scale_from_stats;
    A.e32 = R.e32; A.conv32 = R.conv32;
    A.eye[0] = R.eye[0]; A.eye[1] = R.eye[1];
    A.neyes = R.neyes; A.single = R.neyes == 1 ? 0 : R.single;
    A.stereo_is_u8 = R.stereo_is_u8;
    A.out_u8 = R.out_u8; A.stereo = R.stereo; A.mask = R.mask; A.depth_l = R.depth_l; A.depth_r = R.depth_r;
    A.out_h = R.out_h; A.out_w = R.out_w;
    A.rowflag = rowflag;
    A.dbg = R.dbg;
    A.tilemap = R.tilemap; A.gray = R.lazy_gray; A.tm_words = R.tm_words;
    const int tiles = (A.w + PT_T - 1) / PT_T;
    dim3 grid(tiles * 8, A.single >= 0 ? (A.h + 7) / 8 : eye_group_grid_y(A.h), A.n), block(PT_THREADS);
    const int variant = dev_switch(CS_DEBUG_PT_VARIANT);
#define PT_LAUNCH(SH, KP, KS, MW)                                                                                   \
    {                                                                                                               \
        size_t lds = polytile_lds(S, SH, KP, KS);                                                                   \
        hipError_t e = hipFuncSetAttribute((const void*)k_polytile<SH, KP, KS, MW>,                                 \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL((k_polytile<SH, KP, KS, MW>), grid, block, lds, stream, A.image_f32, A.eye[0].depth,     \
                           A.eye[1].depth, A.w, A.h, A.S, A.single, A);                              \
    }
    // soft default: 4 points / 5 segments per pixel in the lists -> 21.6 KB of LDS at the bench halo, 72 VGPRs: 7 workgroups
    // per CU (LDS is handed out in 2 KB granules; 6 segments would be 128 bytes over).  CS_PT_VARIANT: development.
    // sharp (two points per source pixel): 5 points / 7 segments per pixel (+16 % over 6 / 8 at the same redo rate on the bench)
    if (sharp && variant == 1) PT_LAUNCH(1, 6, 8, 4)
    else if (sharp && variant == 2) PT_LAUNCH(1, 4, 6, 5)
    else if (sharp) PT_LAUNCH(1, 5, 7, 5)
    else if (variant == 1) PT_LAUNCH(0, 4, 6, 6)
    else if (variant == 9) PT_LAUNCH(0, 4, 5, 7)
    else if (variant == 2) PT_LAUNCH(0, 3, 4, 7)
    else if (variant == 13) PT_LAUNCH(0, 6, 8, 6)
    else PT_LAUNCH(0, 4, 5, 7)
#undef PT_LAUNCH
    return hipGetLastError();
}

}  // namespace cs
