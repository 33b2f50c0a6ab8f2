// cs_rowwarp.hip -- the per-row forward-warp + hole-fill kernels (gfx950).
//
// One workgroup owns one image row of one frame and produces BOTH eyes of it: the source row
// (uint8 RGB) is staged once in LDS, the disparity of every source pixel is computed with the
// libm-exact powf (cs_math.h), and the fill technique runs entirely out of LDS:
//   none / naive / naive_interpolating  reference stereoimage_generation.py:1850-1910
//   inverse                              reference :1715-1737
//   polylines_soft / polylines_sharp     reference :1912-1992
// The eye row is then converted and written straight into its slot of the output layout
// (side-by-side, top-bottom, anaglyph; reference :1543-1562) together with the no-fill mask
// (GenerateStereo.py:355-361) and the two depth-map outputs (:1511-1516, :365-378), so no
// per-eye intermediate image ever touches HBM.
//
// The sequential CPU sweeps are restated as data-parallel steps with identical results:
//   * forward map: "later write wins" == max (div<0) / min (div>=0) source column per
//     destination -> LDS atomicMax/atomicMin
//   * naive fill: nearest filled pixel, right before left -> prefix-max / suffix-min scans
//   * naive_interpolating: the sweep only couples pixels between two consecutive "good" pixels,
//     so every such interval is replayed literally by one lane
//   * inverse: z-buffer with strict '>' and ascending x == argmax over (closeness, -x) ->
//     64-bit LDS atomicMax
//   * polylines: counting-sort of the polyline points by output pixel + in-bin ranking (== the
//     reference's stable insertion sort), per-pixel segment lists (CSR) and one lane per output
//     pixel; rows whose result would depend on the reference's active-list ORDER (exact ties,
//     non-monotone centres, list overflow) are detected and replayed sequentially by lane 0.
//
// Compile with -ffp-contract=off (see cs_math.h).
#include "cs_common.h"
#include "cs_kernels.h"
#include <stdlib.h>

namespace cs {
static inline size_t al256r(size_t x) { return (x + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------
// LDS carving
// ---------------------------------------------------------------------------------------------
struct Lds {
    float* lut;                // [256] k/255
    csm::PowfTables* tabs;     // libm tables
    int* misc;                 // [32] flags / counters / scan scratch
    uint8_t* img;              // [3w]
    uint8_t* res;              // [3w] eye result
    uint8_t* ana;              // [2w] channels stashed from the first eye of an anaglyph
    float* nd;                 // [w] normalised depth (forward/inverse) or coord_d (polylines)
    char* tech;                // technique-specific area
};


__host__ __device__ constexpr bool fill_uses_res(int fill) {
    // polylines emits pixels directly; hybrid_edge's LDS kernel is the splat only (its fill pass is k_hybrid_fill)
    return fill != CS_FILL_POLYLINES_SOFT && fill != CS_FILL_POLYLINES_SHARP && fill != CS_FILL_HYBRID_EDGE;
}
__host__ __device__ inline size_t lds_common_bytes(int fill, int w, int anaglyph) {
    return 1024 + align16(sizeof(csm::PowfTables)) + 128 + align16(3 * (size_t)w) +
           (fill_uses_res(fill) ? align16(3 * (size_t)w) : 0) + (anaglyph ? align16(2 * (size_t)w) : 0) +
           align16(4 * (size_t)w);
}
#define CS_ROW_LDS_STATIC 64   // static LDS of k_rowwarp next to its dynamic request (16 bytes today: the request may not take all 160 KB)
// capacity of the per-pixel segment lists of the polylines row kernel (16-bit entries; also the scratch of its counting sort:
// >= npt).  A row whose lists need more is not evaluated in parallel at all: it goes to the replay (as one whole-row stretch) or to
// the literal sequential sweep -- slower, same pixels.  polylines_sharp keeps two points per source: beyond 6 954 columns the full
// capacity (4 w + 64) no longer fits the 160 KB next to the row's other arrays, and the capacity is what the LDS leaves (round 4:
// 7 680-wide rows are accepted; the tile kernel k_polypoint<SHARP> has no such limit, this only concerns the rows it flags).
__host__ __device__ inline int poly_cap(int w, int sharp) {
    const int full = (sharp ? 4 : 3) * w + 64;
    if (!sharp) return full;
    const long long other = (long long)lds_common_bytes(CS_FILL_POLYLINES_SHARP, w, 0) + (long long)align16(2 * (size_t)(2 * w + 2)) +
                            (long long)align16(2 * ((size_t)w + 4)) + (long long)align16(2 * ((size_t)w + 2)) + 2048;
    const long long fit = (((long long)CS_LDS_BYTES - CS_ROW_LDS_STATIC - other) / 2) & ~7LL;
    if ((long long)full <= fit) return full;
    // (round 6: 2 048 entries beyond the point count are enough to accept the row -- 4 096 until then, which put the limit at 7 990 columns;
    // rows whose lists need more than the capacity are evaluated in column ranges, RW_MAX_RANGES, as they are at 7 680: 8 206 columns)
    return fit >= 2LL * w + 2 + 2048 ? (int)fit : full;   // (too wide even so: the full figure makes the width check fail)
}
// The LEAN instantiation (first pass over the rows the tile kernel flags) wants TWO rows per CU.  polylines_sharp at 4K asks for 92 KB with
// its full list capacity (4 w + 64 entries) -- one row per CU, and the first pass fell to the full kernel (64 x 4K saturated depth: 13 of
// 22 ms in it, tools/sessions/r06_s27.sh).  With the capacity that half the LDS leaves (2.67 entries per column at 3 840; a sharp row
// needs ~3) the row is evaluated in two column ranges, two rows at a time (round 6).
__host__ __device__ inline int poly_cap_lean(int w, int sharp) {
    const int full = poly_cap(w, sharp);
    if (!sharp) return full;
    const long long other = (long long)lds_common_bytes(CS_FILL_POLYLINES_SHARP, w, 0) + (long long)align16(2 * (size_t)(2 * w + 2)) +
                            (long long)align16(2 * ((size_t)w + 4)) + (long long)align16(2 * ((size_t)w + 2)) + 2048;
    const long long fit2 = (((long long)CS_LDS_BYTES / 2 - CS_ROW_LDS_STATIC - other) / 2) & ~7LL;
    if ((long long)full <= fit2) return full;
    return fit2 >= 2LL * w + 2 + 2048 ? (int)fit2 : full;
}
__host__ __device__ inline size_t lds_tech_bytes(int fill, int w) {
    switch (fill) {
    case CS_FILL_NONE: return align16(4 * (size_t)w);                       // winner
    case CS_FILL_NAIVE: return align16(4 * (size_t)w);   // winner (L, R: 16-bit columns over the dead normalised depth, round 6)
    case CS_FILL_NAIVE_INTERPOLATING:  // winner, flags, interval starts (the new colours: over the dead normalised depth, round 6)
        return align16(4 * (size_t)w) + align16((size_t)w) + align16(2 * (size_t)w);
    case CS_FILL_INVERSE: return align16(8 * (size_t)w);
    // (round 6: the nearest-valid-column arrays L, R of the post-fill overlay the normalised depth and the winner / key words: 11 578 / 9 004
    // instead of 7 368 / 6 234 columns)
    case CS_FILL_NONE_POST: return align16(4 * (size_t)w);                       // winner
    case CS_FILL_INVERSE_POST: return align16(8 * (size_t)w);                   // keys
    case CS_FILL_HYBRID_EDGE:  // splat kernel: dest_x, bin offsets, scratch, sorted ids, exp table
        return align16(4 * (size_t)w) + 3 * align16(2 * ((size_t)w + 4)) + 2048;
    case CS_FILL_HYBRID_EDGE_PLUS: {  // the hybrid fill pass, then the polylines row technique (same region), + its pixels
        size_t a = lds_tech_bytes(CS_FILL_HYBRID_EDGE, w), b = lds_tech_bytes(CS_FILL_POLYLINES_SOFT, w);
        return (a > b ? a : b) + align16(3 * (size_t)w);
    }
    case CS_FILL_POLYLINES_SOFT:
    case CS_FILL_POLYLINES_SHARP: {
        int sharp = fill == CS_FILL_POLYLINES_SHARP;
        size_t npt = poly_npt(w, sharp);
        return align16(2 * npt) + align16(2 * ((size_t)w + 4)) + align16(2 * ((size_t)w + 2)) +
               align16(2 * (size_t)poly_cap(w, sharp)) + align16(2 * 1024);
    }
    default: return 0;
    }
}

__device__ inline Lds carve(char* base, int fill, int w, int anaglyph) {
    Lds L;
    size_t o = 0;
    L.lut = (float*)(base + o); o += 1024;
    L.tabs = (csm::PowfTables*)(base + o); o += align16(sizeof(csm::PowfTables));
    L.misc = (int*)(base + o); o += 128;
    L.img = (uint8_t*)(base + o); o += align16(3 * (size_t)w);
    L.res = (uint8_t*)(base + o); if (fill_uses_res(fill)) o += align16(3 * (size_t)w);
    L.ana = (uint8_t*)(base + o); if (anaglyph) o += align16(2 * (size_t)w);
    L.nd = (float*)(base + o); o += align16(4 * (size_t)w);
    L.tech = base + o;
    return L;
}

__constant__ csm::PowfTables c_powf_tables = CS_POWF_TABLES_INIT;

// `sign_d * (abs(d) ** e) * divergence_px` in the reference's float32 dialect.
__device__ __forceinline__ float disparity(float d, float e32, float div32, const csm::PowfTables* T) {
    float s = d >= 0.0f ? 1.0f : -1.0f;
    float p = csm::powf_exact_simt(fabsf(d), e32, T);
    return (s * p) * div32;
}

// The same in dialect D64 (numba typing, SURVEY.md Appendix A -- what the reference computes when numba is installed):
// the float32 depth meets the float64 exponent, so `abs(d) ** e`, the products and the sums run in float64.  `pow` is the
// device library's (<= 1 ulp from libm's; the callers only take int() / floor() of the result plus a pixel coordinate).
__device__ __forceinline__ double disparity64(float d, double e64, double div64) {
    const double s = d >= 0.0f ? 1.0 : -1.0;
    // (exponents 2 and 1 exactly, like the tile kernels: the device library's pow(0.25, 2.0) is one ulp short of 0.0625, which
    // moves int() at offsets that are whole numbers -- found by the dialect fuzz, round 5)
    const double ax = (double)fabsf(d);
    const double p = e64 == 2.0 ? ax * ax : (e64 == 1.0 ? ax : pow(ax, e64));
    return (s * p) * div64;
}

// ---------------------------------------------------------------------------------------------
// forward map shared by none / naive / naive_interpolating
// ---------------------------------------------------------------------------------------------
__device__ void forward_map(const Lds& L, int w, const EyeArgs& E, float e32, int* winner, int d64 = 0, double e64 = 0.0) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int init = E.asc ? -1 : 0x7fffffff;
    for (int c = tid; c < w; c += nt) winner[c] = init;
    __syncthreads();
    for (int c = tid; c < w; c += nt) {
        int io;
        if (d64 & 1) {
            const double off = disparity64(L.nd[c], e64, E.div64) + E.sep64;
            io = off >= 2147483520.0 ? 0x7fffff00 : (off <= -2147483520.0 ? -0x7fffff00 : (int)off);
        } else {
            const float off = disparity(L.nd[c], e32, E.div32, L.tabs) + E.sep32;
            // int(): truncation toward zero; keep the conversion defined for absurd offsets
            io = off >= 2147483520.0f ? 0x7fffff00 : (off <= -2147483520.0f ? -0x7fffff00 : (int)off);
        }
        long long cd = (long long)c + io;
        if (cd >= 0 && cd < w) {
            if (E.asc) atomicMax(&winner[(int)cd], c);
            else atomicMin(&winner[(int)cd], c);
        }
    }
    __syncthreads();
    for (int c = tid; c < w; c += nt) {
        int s = winner[c];
        bool f = s != init;
        L.res[3 * c + 0] = f ? L.img[3 * s + 0] : 0;
        L.res[3 * c + 1] = f ? L.img[3 * s + 1] : 0;
        L.res[3 * c + 2] = f ? L.img[3 * s + 2] : 0;
    }
    __syncthreads();
}

// sum() of a uint8 pixel: wraps mod 256 in dialect D32 (quirk Q5), int64 under numba (D64)
__device__ __forceinline__ unsigned sum8(const uint8_t* p, unsigned mask = 0xffu) { return (unsigned)(p[0] + p[1] + p[2]) & mask; }

template <int FILL>
__device__ void technique_forward(const Lds& L, int w, const EyeArgs& E, float e32, int d64 = 0, double e64 = 0.0) {
    const int tid = threadIdx.x, nt = blockDim.x;
    int* winner = (int*)L.tech;
    const int init = E.asc ? -1 : 0x7fffffff;
    const unsigned smask = (d64 & 2) ? 0xffffu : 0xffu;
    forward_map(L, w, E, e32, winner, d64, e64);
    if (FILL == CS_FILL_NAIVE) {
        // nearest filled pixel: Lf[c] = last filled <= c (-1: none), Rf[c] = first filled >= c (0x7fff: none); 16-bit
        // columns (w < 32767 is far beyond what fits the LDS anyway) keep 8K-wide rows inside the 160 KB
        // (round 6: both live in the normalised depth's 4 w bytes, dead once the forward map is made: the row takes the LDS of 'none')
        int16_t* Lf = (int16_t*)L.nd;
        int16_t* Rf = Lf + w;
        const int NONE_R = 0x7fff, BIG = 1 << 29;
        for (int c = tid; c < w; c += nt) {
            bool f = winner[c] != init;
            Lf[c] = (int16_t)(f ? c : -1);
            Rf[c] = (int16_t)(f ? c : NONE_R);
        }
        __syncthreads();
        block_scan_inclusive(Lf, w, -1, OpMax(), L.misc + 8);
        block_scan_inclusive(Rf, w, NONE_R, OpMin(), L.misc + 8, true);
        for (int c = tid; c < w; c += nt) {
            if (winner[c] != init) continue;
            const int r = Rf[c], l = Lf[c];
            int dr = r == NONE_R ? BIG : r - c, dl = l < 0 ? BIG : c - l;
            int src = -1;
            if (dr <= dl) { if (dr < E.naive_lim) src = r; }
            else if (dl < E.naive_lim) src = l;
            if (src >= 0) {  // sources are filled pixels, which this loop never modifies
                L.res[3 * c + 0] = L.res[3 * src + 0];
                L.res[3 * c + 1] = L.res[3 * src + 1];
                L.res[3 * c + 2] = L.res[3 * src + 2];
            }
        }
        __syncthreads();
    } else if (FILL == CS_FILL_NAIVE_INTERPOLATING) {
        // "good" = filled and channel sum != 0 (mod 256): never overwritten, bounds every fill.
        uint8_t* flags = (uint8_t*)(L.tech + align16(4 * (size_t)w));  // bit0 filled, bit1 good
        for (int c = tid; c < w; c += nt) {
            bool f = winner[c] != init;
            bool g = f && sum8(&L.res[3 * c], smask) != 0;
            flags[c] = (uint8_t)((f ? 1 : 0) | (g ? 2 : 0));
        }
        __syncthreads();
        // The reference walks a row left to right; inside an interval between two good pixels the FIRST unfilled
        // pixel l0 triggers one linear ramp over l0 .. g-1 (g = the next good pixel, or w) from the colour left of l0 to
        // the colour at g, in uint8 wrap arithmetic.  Every later pixel of the interval is then skipped -- unless the
        // ramp left an unfilled pixel with channel sum 0 (mod 256), which triggers again with the freshly written
        // left neighbour (quirk).  Parallel form: every non-good pixel finds its interval's l0 and g by a bounded walk
        // over the immutable flags and computes its own ramp value; intervals that are too long for the walk or hit
        // the quirk are replayed literally by the sequential code below (bit 3 of the interval start's flag).
        // (round 6: the new colours live in the normalised depth's 4 w bytes, dead once the forward map is made: 17 instead of 20 bytes of
        // LDS per column, 9 536 instead of 8 104 columns)
        uint8_t* tmpc = (uint8_t*)L.nd;                                                                          // [3w]
        uint16_t* istart = (uint16_t*)(L.tech + align16(4 * (size_t)w) + align16((size_t)w));                    // [w]
        constexpr int NI_WALK = 160;
        auto flag_interval = [&](int s0) { atomicOr((unsigned*)flags + (s0 >> 2), 8u << ((s0 & 3) * 8)); };
        for (int c = tid; c < w; c += nt) {
            if (flags[c] & 2) continue;
            // walk left to the last good pixel (or the row start), remembering the leftmost unfilled pixel on the way
            int lg = c, l0 = -1, steps = 0;
            while (lg >= 0 && !(flags[lg] & 2) && steps <= NI_WALK) { if (!(flags[lg] & 1)) l0 = lg; lg--; steps++; }
            const bool left_ok = lg < 0 || (flags[lg] & 2);
            int g = c + 1;
            steps = 0;
            // (one step less than the walk to the left: an interval the start pixel finds short enough on its right is
            // then short enough for the left walk of every pixel in it -- the start pixel is the one that flags long ones)
            while (g < w && !(flags[g] & 2) && steps < NI_WALK) { g++; steps++; }
            const bool right_ok = g >= w || (flags[g] & 2);
            const int s0 = lg + 1;
            if (left_ok) istart[c] = (uint16_t)s0;
            if (!left_ok || !right_ok) {  // too long for the walk: the sequential code does this interval
                if (left_ok) flag_interval(s0);
                else istart[c] = 0xffff;   // start unknown: resolved below
                continue;
            }
            if (l0 < 0 || c < l0) continue;  // no unfilled pixel in the interval up to here: untouched
            uint8_t lb[3] = {0, 0, 0}, rb[3] = {0, 0, 0};
            if (l0 > 0) { lb[0] = L.res[3 * l0 - 3]; lb[1] = L.res[3 * l0 - 2]; lb[2] = L.res[3 * l0 - 1]; }
            if (g < w) { rb[0] = L.res[3 * g]; rb[1] = L.res[3 * g + 1]; rb[2] = L.res[3 * g + 2]; }
            if (sum8(lb, smask) == 0) { lb[0] = rb[0]; lb[1] = rb[1]; lb[2] = rb[2]; }
            else if (sum8(rb, smask) == 0) { rb[0] = lb[0]; rb[1] = lb[1]; rb[2] = lb[2]; }
            const float total = (float)(1 + g - l0);
            const float k = (float)(c - l0 + 1);
            uint8_t v[3];
            v[0] = (uint8_t)(lb[0] + csm::f32_to_u8_wrap((((float)rb[0] - (float)lb[0]) / total) * k));
            v[1] = (uint8_t)(lb[1] + csm::f32_to_u8_wrap((((float)rb[1] - (float)lb[1]) / total) * k));
            v[2] = (uint8_t)(lb[2] + csm::f32_to_u8_wrap((((float)rb[2] - (float)lb[2]) / total) * k));
            tmpc[3 * c] = v[0]; tmpc[3 * c + 1] = v[1]; tmpc[3 * c + 2] = v[2];
            atomicOr((unsigned*)flags + (c >> 2), 4u << ((c & 3) * 8));  // has a new colour (atomic: bit 3 of the same byte may be set concurrently)
            if (c > l0 && !(flags[c] & 1) && sum8(v, smask) == 0) flag_interval(s0);  // the quirk: re-trigger -> literal replay
        }
        __syncthreads();
        for (int c = tid; c < w; c += nt) {
            if (!(flags[c] & 4)) continue;
            if (flags[istart[c]] & 8) continue;  // its interval is replayed below
            L.res[3 * c] = tmpc[3 * c]; L.res[3 * c + 1] = tmpc[3 * c + 1]; L.res[3 * c + 2] = tmpc[3 * c + 2];
        }
        __syncthreads();
        for (int s = tid; s < w; s += nt) {
            if ((flags[s] & 2) || (s > 0 && !(flags[s - 1] & 2))) continue;  // not the start of an interval
            if (!(flags[s] & 8)) continue;                                      // done in parallel above
            for (int l = s; l < w && !(flags[l] & 2); l++) {
                if (sum8(&L.res[3 * l], smask) != 0 || (flags[l] & 1)) continue;
                uint8_t lb[3] = {0, 0, 0}, rb[3] = {0, 0, 0};
                if (l > 0) { lb[0] = L.res[3 * l - 3]; lb[1] = L.res[3 * l - 2]; lb[2] = L.res[3 * l - 1]; }
                int r = l + 1;
                while (r < w) {
                    if (sum8(&L.res[3 * r], smask) != 0 && (flags[r] & 1)) {
                        rb[0] = L.res[3 * r]; rb[1] = L.res[3 * r + 1]; rb[2] = L.res[3 * r + 2];
                        break;
                    }
                    r++;
                }
                if (sum8(lb, smask) == 0) { lb[0] = rb[0]; lb[1] = rb[1]; lb[2] = rb[2]; }
                else if (sum8(rb, smask) == 0) { rb[0] = lb[0]; rb[1] = lb[1]; rb[2] = lb[2]; }
                float total = (float)(1 + r - l);
                float st0 = ((float)rb[0] - (float)lb[0]) / total;
                float st1 = ((float)rb[1] - (float)lb[1]) / total;
                float st2 = ((float)rb[2] - (float)lb[2]) / total;
                for (int c = l; c < r; c++) {
                    float k = (float)(c - l + 1);
                    L.res[3 * c + 0] = (uint8_t)(lb[0] + csm::f32_to_u8_wrap(st0 * k));
                    L.res[3 * c + 1] = (uint8_t)(lb[1] + csm::f32_to_u8_wrap(st1 * k));
                    L.res[3 * c + 2] = (uint8_t)(lb[2] + csm::f32_to_u8_wrap(st2 * k));
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// none_post / inverse_post (reference :1804-1833): after the mapping, every row that has a valid pixel is
// np.interp'ed channel by channel over its valid pixels: float64 inside numpy -- slope = (y1 - y0) / (x1 - x0),
// slope * (x - x0) + y0, the sample itself on a valid pixel, the end values outside -- then stored into a float32
// array and truncated by astype(uint8).  Nearest valid pixel to the left / right = prefix-max / suffix-min scans.
// ---------------------------------------------------------------------------------------------
template <class Valid>
__device__ void technique_post_interp(const Lds& L, int w, int* Lf, int* Rf, const Valid& valid) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int BIG = 1 << 29;
    // (round 6: Lf / Rf overlay arrays that are dead once the validity of every column is known -- the normalised depth and the forward
    // map's winner / key words -- so the flags are collected in a register first: at most 32 columns per lane, checked by the launcher)
    unsigned fm = 0;
    {
        int i = 0;
        for (int c = tid; c < w; c += nt, i++) fm |= valid(c) ? 1u << i : 0u;
    }
    __syncthreads();
    {
        int i = 0;
        for (int c = tid; c < w; c += nt, i++) {
            const bool f = (fm >> i) & 1u;
            Lf[c] = f ? c : -BIG;
            Rf[c] = f ? c : BIG;
        }
    }
    __syncthreads();
    block_scan_inclusive(Lf, w, -BIG, OpMax(), L.misc + 8);
    block_scan_inclusive(Rf, w, BIG, OpMin(), L.misc + 8, true);
    for (int c = tid; c < w; c += nt) {
        const int l = Lf[c], r = Rf[c];
        if (l == c || (l < 0 && r >= BIG)) continue;  // valid pixel keeps its value; a row without valid pixels stays black
        // (sources are valid pixels, which this loop never modifies)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            double res;
            if (l < 0) res = (double)L.res[3 * r + ch];
            else if (r >= BIG) res = (double)L.res[3 * l + ch];
            else {
                const double y0 = (double)L.res[3 * l + ch], y1 = (double)L.res[3 * r + ch];
                const double slope = (y1 - y0) / ((double)r - (double)l);
                res = slope * ((double)c - (double)l) + y0;
            }
            L.res[3 * c + ch] = (uint8_t)(int)(float)res;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// inverse: two-column z-buffered splat
// ---------------------------------------------------------------------------------------------
__device__ void technique_inverse(const Lds& L, int w, const EyeArgs& E, float e32, int d64 = 0, double e64 = 0.0) {
    const int tid = threadIdx.x, nt = blockDim.x;
    unsigned long long* key = (unsigned long long*)L.tech;
    const unsigned long long init = ((unsigned long long)csm::f2ord(-1.0f) << 32) | 0xffffffffull;
    for (int c = tid; c < w; c += nt) key[c] = init;
    __syncthreads();
    for (int x = tid; x < w; x += nt) {
        float d = L.nd[x];
        float fl;
        if (d64 & 1) fl = (float)floor((((double)x + 0.5) + disparity64(d, e64, E.div64)) + E.sep64);   // (only the range test and int() use it)
        else {
            float off = disparity(d, e32, E.div32, L.tabs);
            float dest = ((float)x + 0.5f + off) + E.sep32;
            fl = floorf(dest);
        }
        if (!(fl >= -2.0f && fl <= (float)w)) continue;
        int j = (int)fl;
        unsigned long long k = ((unsigned long long)csm::f2ord(d) << 32) | (unsigned long long)(0xffffffffu - (unsigned)x);
        if (j >= 0 && j < w) atomicMax(&key[j], k);
        if (j + 1 >= 0 && j + 1 < w) atomicMax(&key[j + 1], k);
    }
    __syncthreads();
    for (int c = tid; c < w; c += nt) {
        unsigned long long k = key[c];
        bool f = k > init;
        int s = (int)(0xffffffffu - (unsigned)(k & 0xffffffffull));
        L.res[3 * c + 0] = f ? L.img[3 * s + 0] : 0;
        L.res[3 * c + 1] = f ? L.img[3 * s + 1] : 0;
        L.res[3 * c + 2] = f ? L.img[3 * s + 2] : 0;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// polylines
// ---------------------------------------------------------------------------------------------
struct Poly {
    int w, npt, sharp, cap;
    float sep32;        // x of polyline point o is recomputed from coord_d on every use (saves 4*npt bytes of LDS)
    uint16_t* perm;     // [npt] sorted position -> point
    uint16_t* binoff;   // [w+3] after the fill pass: END of bin b (bin 0: x<0, bin c+1: [c,c+1), bin w+1: x>=w)
    uint16_t* segoff;   // [w+1] after the fill pass: END of pixel p's segment list
    uint16_t* entries;  // [cap] segment ids (also: in-bin scratch during sorting, csg during the fallback)
    uint16_t* longs;    // [1024] long segments
    const float* cd;    // [w] coord_d
    // dialect bit 0 (float64 disparity chain, cs_params.flags bit 3): col + 0.5 + coord_d + separation_px as the float64 value
    // the reference holds before it is stored into the float32 `pt` array; cd then holds (float)coord_d.  null: D32
    const double* xd;   // [w]
};

// x of polyline point o in reference order (:1921-1935): sentinels at -w and 2w, else
// col + 0.5 + coord_d + separation_px (float32 step by step), -/+ 0.45 for the two 'sharp' points.
__device__ __forceinline__ float poly_x(const Poly& P, int o) {
    if (o <= 0) return (float)(-1.0 * P.w);
    if (o >= P.npt - 1) return (float)(2.0 * P.w);
    const float half32 = (float)0.45;
    int c = P.sharp ? (o - 1) >> 1 : o - 1;
    if (P.xd) {   // one rounding, from float64 (the half width is the Python float 0.45 there)
        const double b = P.xd[c];
        return P.sharp ? (float)(((o - 1) & 1) ? b + 0.45 : b - 0.45) : (float)b;
    }
    float x = ((float)c + 0.5f + P.cd[c]) + P.sep32;
    if (P.sharp) x = ((o - 1) & 1) ? x + half32 : x - half32;
    return x;
}
__device__ __forceinline__ int poly_col(const Poly& P, int o) {
    if (o <= 0) return 0;
    if (o >= P.npt - 1) return P.w - 1;
    return P.sharp ? (o - 1) >> 1 : o - 1;
}
__device__ __forceinline__ float poly_z(const Poly& P, int o) {
    if (o <= 0 || o >= P.npt - 1) return 0.0f;
    return fabsf(P.cd[P.sharp ? (o - 1) >> 1 : o - 1]);
}
__device__ __forceinline__ int poly_bin(const Poly& P, float x) {
    if (x < 0.0f) return 0;
    if (x >= (float)P.w) return P.w + 1;
    return (int)x + 1;
}

// colour contribution of one sub-interval (reference :1981-1989 with the D32 typing of Appendix A)
__device__ __forceinline__ void poly_accumulate(const Poly& P, const uint8_t* img, int seg, float center, bool sig64,
                                                double sig_d, float sig_f, float color[3]) {
    int col_l = poly_col(P, seg), col_r = poly_col(P, seg + 1);
    if (col_l == col_r) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (sig64) color[c] = (float)((double)color[c] + (double)img[3 * col_l + c] * sig_d);
            else color[c] = color[c] + (float)img[3 * col_l + c] * sig_f;
        }
    } else {
        float x0 = poly_x(P, seg), x1 = poly_x(P, seg + 1);
        float ip_k = (center - x0) / (x1 - x0);
        float om = 1.0f - ip_k;
        float s = sig64 ? (float)sig_d : sig_f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float a = (float)img[3 * col_l + c] * om;
            float b = (float)img[3 * col_r + c] * ip_k;
            color[c] = color[c] + (a + b) * s;
        }
    }
}

// sub-interval [max(col, a), min(col+1, b)] shrunk by EPSILON on both sides (reference :1957-1960)
struct SubInt {
    bool sig64;
    double sig_d;
    float sig_f, center;
};
__device__ __forceinline__ SubInt poly_subinterval(int col, float a, float b) {
    const float eps32 = (float)1e-7;
    SubInt s;
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

// literal replay of the reference sweep for one row by one lane (rows flagged as order-dependent)
template <class Emit>
__device__ int poly_sequential(const Poly& P, const Lds& L, int csg_cap_ref, const Emit& emit) {
    const int w = P.w, sg_end = P.npt - 1;
    uint16_t* csg = P.entries;
    int cap = min(csg_cap_ref, P.cap);
    int csg_end = 0, sg_pointer = 0, pt_i = 0;
    for (int col = 0; col < w; col++) {
        float color[3] = {0.5f, 0.5f, 0.5f};
        while (poly_x(P, P.perm[pt_i]) < (float)col) pt_i++;
        pt_i--;
        while (poly_x(P, P.perm[pt_i]) < (float)(col + 1)) {
            SubInt s = poly_subinterval(col, poly_x(P, P.perm[pt_i]), poly_x(P, P.perm[pt_i + 1]));
            while (sg_pointer < sg_end && poly_x(P, P.perm[sg_pointer]) < s.center) {
                if (csg_end >= cap) return -1;
                csg[csg_end++] = P.perm[sg_pointer++];
            }
            int ci = 0;
            while (ci < csg_end) {
                if (poly_x(P, csg[ci] + 1) < s.center) { csg[ci] = csg[csg_end - 1]; csg_end--; }
                else ci++;
            }
            int best = 0;
            if (csg_end != 1) {
                float bc = (float)(-1e-7);
                for (ci = 0; ci < csg_end; ci++) {
                    int o = csg[ci];
                    float x0 = poly_x(P, o), x1 = poly_x(P, o + 1);
                    float ip_k = (s.center - x0) / (x1 - x0);
                    float cl = (1.0f - ip_k) * poly_z(P, o) + ip_k * poly_z(P, o + 1);
                    if (bc < cl && 0.0f < ip_k && ip_k < 1.0f) { bc = cl; best = ci; }
                }
            }
            // csg_end == 0 cannot happen (the polyline is connected from -w to 2w); slot 0 then
            // still holds a valid (stale) id because the list is never empty after its first fill.
            poly_accumulate(P, L.img, csg[best], s.center, s.sig64, s.sig_d, s.sig_f, color);
            pt_i++;
        }
        emit(col, csm::f32_to_u8_wrap(color[0]), csm::f32_to_u8_wrap(color[1]), csm::f32_to_u8_wrap(color[2]));
    }
    return 0;
}

// The sweep with numba's typing (dialect bit 1; derived, oracle/stereo_oracle.c oracle_polylines holds the
// same statements): the ends of a sub-interval, its significance and its centre are float64 -- max(col, x) + EPSILON with
// the epsilon always alive --, float32 array elements meet them in float64 compares, ip_k is a float64 numerator over the
// float32 difference x1 - x0, closeness and the colour products are float64, and every sub-interval rounds once into the
// float32 `color`.  One lane replays a row literally; the dialect is a compatibility path, not a fast one.
// (round 6, second session) The same statements over a STRETCH c0 .. c1 of the row from a known state -- behind a reset pixel the
// active list is [seg0] whatever happened before (poly_replay_stretch's argument), `sgp0` sorted points lie left of that pixel's last
// centre, `pt0` points left of pixel c0 -- with the list in `csg` (capacity `csg_room`).  The whole row: c0 = 0, c1 = w - 1, seg0 < 0.
// Until then an order-dependent row under numba's sweep typing was swept whole by one lane (3 ms per 4K row); now a lane per
// stretch, a few dozen columns each.  Returns 0, -1 (the reference's list would overflow), -3 (`csg_room` exceeded: the whole-row form).
// (a real call: inlined at its two call sites it took the dialect row kernels from 113 to 328 spilled vector registers)
template <class Emit>
__device__ __noinline__ int poly_stretch64(const Poly& P, const Lds& L, int csg_cap_ref, const Emit& emit, int c0, int c1, int seg0, int sgp0, int pt0,
                                           uint16_t* csg, int csg_room) {
    const int sg_end = P.npt - 1;
    const int cap = min(csg_cap_ref, P.cap);
    int csg_end = 0, sg_pointer = sgp0, pt_i = max(pt0 - 1, 0);
    if (seg0 >= 0) { csg[0] = (uint16_t)seg0; csg_end = 1; }
    for (int col = c0; col <= c1; col++) {
        float color[3] = {0.5f, 0.5f, 0.5f};
        while (poly_x(P, P.perm[pt_i]) < (float)col) pt_i++;
        pt_i--;
        while (poly_x(P, P.perm[pt_i]) < (float)(col + 1)) {
            const double from_d = fmax((double)col, (double)poly_x(P, P.perm[pt_i])) + 1e-7;
            const double to_d = fmin((double)(col + 1), (double)poly_x(P, P.perm[pt_i + 1])) - 1e-7;
            const double sig = to_d - from_d;
            const double center = from_d + 0.5 * sig;
            while (sg_pointer < sg_end && (double)poly_x(P, P.perm[sg_pointer]) < center) {
                if (csg_end >= cap) return -1;
                if (csg_end >= csg_room) return -3;
                csg[csg_end++] = P.perm[sg_pointer++];
            }
            int ci = 0;
            while (ci < csg_end) {
                if ((double)poly_x(P, csg[ci] + 1) < center) { csg[ci] = csg[csg_end - 1]; csg_end--; }
                else ci++;
            }
            int best = 0;
            if (csg_end != 1) {
                double bc = -1e-7;
                for (ci = 0; ci < csg_end; ci++) {
                    const int o = csg[ci];
                    const float x0 = poly_x(P, o), x1 = poly_x(P, o + 1);
                    const double ip_k = (center - (double)x0) / (double)(x1 - x0);
                    const double cl = (1.0 - ip_k) * (double)poly_z(P, o) + ip_k * (double)poly_z(P, o + 1);
                    if (bc < cl && 0.0 < ip_k && ip_k < 1.0) { bc = cl; best = ci; }
                }
            }
            const int seg = csg[best];
            const int col_l = poly_col(P, seg), col_r = poly_col(P, seg + 1);
            if (col_l == col_r) {
#pragma unroll
                for (int c = 0; c < 3; c++) color[c] = (float)((double)color[c] + (double)L.img[3 * col_l + c] * sig);
            } else {
                const float x0 = poly_x(P, seg), x1 = poly_x(P, seg + 1);
                const double ip_k = (center - (double)x0) / (double)(x1 - x0);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const double v = ((double)L.img[3 * col_l + c] * (1.0 - ip_k) + (double)L.img[3 * col_r + c] * ip_k) * sig;
                    color[c] = (float)((double)color[c] + v);
                }
            }
            pt_i++;
        }
        emit(col, csm::f32_to_u8_wrap(color[0]), csm::f32_to_u8_wrap(color[1]), csm::f32_to_u8_wrap(color[2]));
    }
    return 0;
}
template <class Emit>
__device__ __forceinline__ int poly_sequential64(const Poly& P, const Lds& L, int csg_cap_ref, const Emit& emit) {
    // the whole row: from column 0 with an empty list, the list in the row's per-pixel segment lists (dead by then)
    return poly_stretch64(P, L, csg_cap_ref, emit, 0, P.w - 1, -1, 0, 0, P.entries, P.cap);
}

// The same replay by a whole WAVE.  The sweep itself is sequential (the active list carries its order from column to
// column), but the work inside a step is not: with noisy depth the list holds tens to hundreds of segments and the single
// lane spends its time in dependent LDS reads over it (4K, random 8-bit depth: 0.8 frames/s).  Here the 64 lanes share
// every step:
//   * removal.  The reference's scan `if dead(csg[ci]): csg[ci] = csg[end-1]; end -= 1 else ci += 1` has a closed form:
//     survivors in the first (end - removed) slots stay where they are, and the dead slots among those are filled, in
//     ascending order, by the survivors of the tail taken from the END backwards (a dead element moved into a hole is
//     re-tested and dropped again).  Dead flags by ballot, ranks by population counts, one LDS hand-over.
//   * selection: `best < closeness` with a strict compare keeps the FIRST maximum -> wave maximum, lowest lane that holds it.
// Returns 0, -1 (the reference's list would overflow) or -2 (more holes than the scratch holds: the single-lane replay runs).
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <class Emit>
__device__ int poly_sequential_wave(const Poly& P, const Lds& L, int csg_cap_ref, const Emit& emit, const float* pxs) {
    const int lane = threadIdx.x & 63;
    const int w = P.w, sg_end = P.npt - 1;
    uint16_t* csg = P.entries;
    unsigned long long* deadw = (unsigned long long*)P.longs;   // [64] dead flags of list chunk ch (P.longs is idle here)
    uint16_t* holepos = P.longs + 256;                          // [768] dead slots of the surviving prefix, ascending
    const int cap_ref = min(csg_cap_ref, P.cap), cap = min(cap_ref, 64 * 64);
    // `pxs` (soft only: npt = w + 2 floats fit the idle bin / segment offset arrays): x of every point, filled by the caller --
    // one LDS read instead of coord_d + arithmetic on every use
    auto PX = [&](int o) { return pxs ? pxs[o] : poly_x(P, o); };
    int csg_end = 0, sg_pointer = 0, pt_i = 0;
    for (int col = 0; col < w; col++) {
        float color[3] = {0.5f, 0.5f, 0.5f};
        while (PX(P.perm[pt_i]) < (float)col) pt_i++;
        pt_i--;
        while (PX(P.perm[pt_i]) < (float)(col + 1)) {
            const SubInt s = poly_subinterval(col, PX(P.perm[pt_i]), PX(P.perm[pt_i + 1]));
            while (sg_pointer < sg_end && PX(P.perm[sg_pointer]) < s.center) {
                if (csg_end >= cap) return cap == cap_ref ? -1 : -2;
                if (lane == 0) csg[csg_end] = P.perm[sg_pointer];
                csg_end++; sg_pointer++;
            }
            wave_lds_sync();
            // ---- removal
            const int n = csg_end, nch = (n + 63) >> 6;
            int removed = 0;
            for (int ch = 0; ch < nch; ch++) {
                const int i = ch * 64 + lane;
                const bool dead = i < n && PX((int)csg[i] + 1) < s.center;
                const unsigned long long m = __ballot(dead);
                if (lane == 0) deadw[ch] = m;
                removed += __popcll(m);
            }
            if (removed) {
                wave_lds_sync();
                const int ns = n - removed;
                int before = 0, nholes = 0;   // dead slots among the first ns, in ascending order
                for (int ch = 0; ch * 64 < ns; ch++) {
                    const unsigned long long m = deadw[ch];
                    const int i = ch * 64 + lane;
                    const bool hole = i < ns && ((m >> lane) & 1ull);
                    const int r = before + __popcll(m & ((1ull << lane) - 1ull));
                    if (hole) { if (r < 768) holepos[r] = (uint16_t)i; }
                    const unsigned long long inpref = (ch + 1) * 64 <= ns ? ~0ull : ((1ull << (ns - ch * 64)) - 1ull);
                    before += __popcll(m & inpref);
                }
                nholes = before;
                if (nholes > 768) return -2;
                wave_lds_sync();
                int after = 0;   // survivors behind the current chunk (towards the end of the list)
                for (int ch = nch - 1; ch >= 0 && ch * 64 + 63 >= ns; ch--) {
                    const unsigned long long m = deadw[ch];
                    const int i = ch * 64 + lane;
                    const unsigned long long valid = (ch + 1) * 64 <= n ? ~0ull : ((1ull << (n - ch * 64)) - 1ull);
                    const unsigned long long tail = ch * 64 >= ns ? ~0ull : ~((1ull << (ns - ch * 64)) - 1ull);
                    const unsigned long long surv = ~m & valid & tail;
                    const bool mine = (surv >> lane) & 1ull;
                    const int r = after + __popcll(lane == 63 ? 0ull : (surv >> (lane + 1)));
                    if (mine) csg[holepos[r]] = csg[i];
                    after += __popcll(surv);
                }
                csg_end = ns;
                wave_lds_sync();
            }
            // ---- selection: the reference's scan over the list, with its strict compare, over the lanes that can still win
            int best = 0;
            if (csg_end != 1) {
                float bc = (float)(-1e-7);
                for (int ch = 0; ch * 64 < csg_end; ch++) {
                    const int i = ch * 64 + lane;
                    float cl = -INFINITY;
                    if (i < csg_end) {
                        const int o = csg[i];
                        const float x0 = PX(o), x1 = PX(o + 1);
                        const float ip_k = (s.center - x0) / (x1 - x0);
                        const float c = (1.0f - ip_k) * poly_z(P, o) + ip_k * poly_z(P, o + 1);
                        if (0.0f < ip_k && ip_k < 1.0f) cl = c;
                    }
                    unsigned long long m = __ballot(bc < cl);
                    if (__popcll(m) <= 6) {
                        while (m) {   // (usually one to three candidates: a scalar pass over them beats six cross-lane steps)
                            const int b = __ffsll((long long)m) - 1;
                            m &= m - 1;
                            const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cl), b));
                            if (bc < v) { bc = v; best = ch * 64 + b; }
                        }
                    } else {          // wave maximum, lowest lane that holds it (strict compare == first maximum)
                        float mx = cl;
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                        if (bc < mx) {
                            bc = mx;
                            best = ch * 64 + __ffsll((long long)__ballot(cl == mx)) - 1;
                        }
                    }
                }
            }
            // csg_end == 0 cannot happen (the polyline is connected from -w to 2w); slot 0 then still holds a valid (stale) id
            poly_accumulate(P, L.img, csg[best], s.center, s.sig64, s.sig_d, s.sig_f, color);
            pt_i++;
        }
        if (lane == 0) emit(col, csm::f32_to_u8_wrap(color[0]), csm::f32_to_u8_wrap(color[1]), csm::f32_to_u8_wrap(color[2]));
    }
    return 0;
}

// Replay of a STRETCH of the row by one wave (round 2, end).  The order of the reference's active list is a function of its
// history -- except where the list holds exactly one segment: then the state is `[that segment]` whatever happened before.
// The parallel evaluation marks those pixels ("reset": one active segment at the pixel's last sub-interval) next to the
// order-dependent ones, and only the stretches from the pixel after a reset to the last order-dependent pixel before the
// next reset are replayed -- tens to hundreds of columns instead of the row, independent of each other, one wave each.
// The list of a stretch is short: one 64-entry chunk (more -> -1, and the row takes the whole-row replay).
// State at the top of pixel c0: the list is [seg0] (seg0 < 0: empty, c0 == 0), segments perm[0 .. sgp0) have been added.
// The list lives in REGISTERS: lane i holds entry i (segment id, both end points' x and |disparity|), so the dead test and
// the closeness of a step need no memory at all, and the 64 sorted points around the sweep position sit in a register
// window read with v_readlane.  LDS is touched when a segment enters the list (its far end point), when a removal leaves
// holes in front of survivors (their lane numbers are handed over, then cross-lane moves) and for the winner's colours.
// `slide` (round 4): the caller may hold only WINDOWS of the row's arrays (k_poly_replay: a few hundred sorted points and source
// columns around the sweep position in 4 KB of LDS, refilled from the row's dump as the sweep advances).  Before the function
// reads sorted positions [lo, hi) it calls slide.points(lo, hi), before it reads the coord_d / colour of source columns
// [lo, hi] slide.columns(lo, hi) -- both may re-base P.perm / P.cd / L.img; false: the range does not fit the windows (-2).
// What a pixel can touch: the 64 sorted points of the register window (and their right neighbours' columns), and the
// segments in the list -- every ACTIVE segment (o -> o + 1) covers the sweep position x, and x - column(o) lies in
// [coord_d(o), 1 + coord_d(o + 1)]: within `halo` columns of the pixel.  NoSlide: the whole row is resident (the row kernel).
// wave maximum of finite / -inf floats, the same value in every lane: four row_shr steps inside the rows of 16, lanes 15 / 31 of
// the lower rows broadcast into the upper ones (DPP; lanes without a source keep their own value), lane 63 holds the total
__device__ __forceinline__ float wave_max_dpp(float v) {
    int b = __builtin_bit_cast(int, v);
#define CS_DPP_FMAX(CTRL, ROWMASK)                                                                                          \
    b = __builtin_bit_cast(int, __builtin_fmaxf(__builtin_bit_cast(float, b),                                                \
                                                  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, CTRL, ROWMASK, 0xf, false))))
    CS_DPP_FMAX(0x111, 0xf);
    CS_DPP_FMAX(0x112, 0xf);
    CS_DPP_FMAX(0x114, 0xf);
    CS_DPP_FMAX(0x118, 0xf);
    CS_DPP_FMAX(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    CS_DPP_FMAX(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef CS_DPP_FMAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 63));
}
// Everything the replay wants to know about point `o` (per lane or uniform) without branches: the sentinels (o <= 0, o >= npt - 1)
// are selected, their column is the clamped one -- poly_x / poly_z / poly_col with their early returns cost exec-mask regions per
// accessor and, evaluated per lane for 64 window points, enough registers to spill.  D32 only for x (P.xd: the accessor).
// (values through references: a struct result went through scratch memory)
__device__ __forceinline__ void poly_point(const Poly& P, const Lds& L, int o, float& px, float& pz, uint32_t& prgb, int& pcol) {
    const int last = P.npt - 1;
    const int c = min(max(P.sharp ? (o - 1) >> 1 : o - 1, 0), P.w - 1);
    const float cdv = P.cd[c];
    float x = ((float)c + 0.5f + cdv) + P.sep32;
    if (P.sharp) x = ((o - 1) & 1) ? x + (float)0.45 : x - (float)0.45;
    x = o <= 0 ? (float)(-1.0 * P.w) : (o >= last ? (float)(2.0 * P.w) : x);
    if (P.xd) x = poly_x(P, o);
    px = x;
    pz = (o <= 0 || o >= last) ? 0.0f : fabsf(cdv);
    pcol = c;
    prgb = (uint32_t)L.img[3 * c] | ((uint32_t)L.img[3 * c + 1] << 8) | ((uint32_t)L.img[3 * c + 2] << 16);
}
struct NoSlide {
    static constexpr bool active = false;
    __device__ __forceinline__ bool points(int, int) const { return true; }
    __device__ __forceinline__ bool columns(int, int) const { return true; }
    __device__ __forceinline__ bool holds(int, int) const { return true; }
    __device__ __forceinline__ void rebase_points(Poly&) const {}
    __device__ __forceinline__ void rebase_columns(Poly&, Lds&) const {}
};
template <class Emit, class Slide>
__device__ int poly_replay_stretch(Poly& P, Lds& L, int csg_cap_ref, const Emit& emit, int c0, int c1, int seg0,
                                   int sgp0, uint16_t* srcpos, int pts_left_of_c0, Slide& slide, int halo) {
    const int lane = threadIdx.x & 63;
    if (!slide.columns(c0 - halo, c0 + halo)) return -2;   // (the start segment's end points, read below)
    slide.rebase_columns(P, L);
    const int sg_end = P.npt - 1;
    const int cap = min(min(csg_cap_ref, P.cap), 64);
    auto rl_f = [](float v, int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i)); };
    // entry `lane` of the list
    int e_o = 0;
    float e_x0 = 0.0f, e_x1 = 0.0f, e_z0 = 0.0f, e_z1 = 0.0f;
    // ... and the colour codes of its two end points' source pixels (r | g << 8 | b << 16; bit 24 of the first: both ends are the
    // same source pixel -- the flat typing of :1981-1984): every entry works out its own contribution in parallel and the winner's is
    // read with v_readlane, instead of a wave-uniform walk (column tests, six LDS reads) on the scalar unit that bounds this kernel
    uint32_t e_c0 = 0, e_c1 = 0;
    int csg_end = 0, sg_pointer = sgp0;
    if (seg0 >= 0) {
        int ca, cb;
        e_o = seg0;
        poly_point(P, L, seg0, e_x0, e_z0, e_c0, ca);
        poly_point(P, L, seg0 + 1, e_x1, e_z1, e_c1, cb);
        e_c0 |= ca == cb ? 1u << 24 : 0u;
        csg_end = 1;
    }
    // window of the sorted points: lane k holds point perm[wbase + k]
    // (round 4, end: the kernel is bound by the CU's scalar unit.  What a step reads about the point at sorted position k -- its x,
    // and when the point becomes the start of a list entry: the x of its right neighbour in SOURCE order and both |disparities| --
    // is fetched per lane when the window is loaded (vector instructions, 64 points at once) and picked up with v_readlane,
    // instead of three accessor calls per added segment with their wave-uniform sentinel branches and LDS round trips.)
    int wbase = 0, wo = 0;
    float wx = 0.0f, wx1 = 0.0f, wz0 = 0.0f, wz1 = 0.0f;
    uint32_t wc0 = 0, wc1 = 0;
    bool lost = false;   // the sweep position and the add pointer drifted more than a window apart (long runs of equal x)
    auto window = [&](int lo) {   // (two points of slack below: the sweep steps back by one at every pixel)
        wbase = max(lo - 2, 0);
        if (!slide.points(wbase, wbase + 64)) lost = true;
        slide.rebase_points(P);
        wo = P.perm[min(wbase + lane, P.npt - 1)];
        if (Slide::active) {   // the columns of these points and of their right neighbours, next to the active segments'
            int cl = poly_col(P, wo), ch = poly_col(P, min(wo + 1, P.npt - 1));
            if (!__all(slide.holds(cl, ch))) {   // (one ballot in the usual case; the wave-wide range only for a refill)
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { cl = min(cl, __shfl_xor(cl, d)); ch = max(ch, __shfl_xor(ch, d)); }
                if (!slide.columns(cl, ch)) lost = true;
                slide.rebase_columns(P, L);
            }
        }
        if (lost) { wx = wx1 = wz0 = wz1 = 0.0f; wc0 = wc1 = 0u; return; }
        int ca, cb;
        poly_point(P, L, wo, wx, wz0, wc0, ca);
        poly_point(P, L, min(wo + 1, P.npt - 1), wx1, wz1, wc1, cb);
        wc0 |= ca == cb ? 1u << 24 : 0u;
    };
    // positions [lo, hi] (lo <= hi) must be in the register window: one unsigned compare (lo >= wbase and hi < wbase + 64); the
    // "cannot be held" exit sits inside the rare refill path, so the steady-state step carries no `lost` test at all
#define RP_NEED(LO, HI)                                                                                     \
    do {                                                                                                    \
        const int lo_ = (LO), hi_ = (HI);                                                                   \
        if ((unsigned)(lo_ - wbase) > (unsigned)(63 - (hi_ - lo_)) || hi_ - lo_ > 63) {                     \
            window(lo_);                                                                                    \
            if (lost || hi_ >= wbase + 64) return -2;                                                       \
        }                                                                                                   \
    } while (0)
    int pt_i = pts_left_of_c0 - 1;   // (binoff[c0]: the number of points left of pixel c0; the sweep's own loop settles it)
    bool first_step = true;
    window(seg0 < 0 ? pt_i : min(pt_i, sg_pointer));
    if (lost) return -2;
    for (int col = c0; col <= c1; col++) {
        float color[3] = {0.5f, 0.5f, 0.5f};
        // (no column window to keep around the pixel any more: since every list entry carries its end points' data -- x, |disparity|,
        // colour codes, fetched when its point was in the register window -- the sweep reads source columns only in window())
        RP_NEED(pt_i, pt_i + 1);
        while (rl_f(wx, pt_i - wbase) < (float)col) { pt_i++; RP_NEED(pt_i, pt_i + 1); }
        pt_i--;
        RP_NEED(pt_i, pt_i + 1);
        while (rl_f(wx, pt_i - wbase) < (float)(col + 1)) {
            const SubInt s = poly_subinterval(col, rl_f(wx, pt_i - wbase), rl_f(wx, pt_i + 1 - wbase));
            if (first_step && seg0 < 0) {
                // ---- the very first step of a row (stretch from column 0, empty list, nothing added yet): the reference
                // appends EVERY segment that starts left of the centre -- all the points an eye shifts out of the frame, more
                // than the 64-entry list and window hold at 4K -- and removes the ended ones in the same step.  Both at once,
                // in closed form: of the K segments added (sorted positions 0 .. K-1) the ns survivors end up as
                // [alive positions < ns in place; the k-th dead position < ns takes the k-th survivor from the end].
                int K = 0, ns_total = 0;
                unsigned long long alive0 = 0ull;
                int alive_before = 0;   // alive entries in the chunks before this lane's chunk ... per chunk below
                // pass 1: K and the number of survivors
                for (int base = 0; base < sg_end; base += 64) {
                    const int k = base + lane;
                    if (!slide.points(base, min(base + 64, sg_end))) return -2;
                    slide.rebase_points(P);
                    const int o = P.perm[min(k, sg_end - 1)];
                    if (Slide::active) {   // (points left of the first centre: the columns an eye shifts out of the frame)
                        int ch = poly_col(P, min(o + 1, P.npt - 1));
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) ch = max(ch, __shfl_xor(ch, d));
                        if (!slide.columns(0, max(ch, halo))) return -2;
                        slide.rebase_columns(P, L);
                    }
                    const bool in = k < sg_end && poly_x(P, o) < s.center;
                    const unsigned long long mi = __ballot(in);
                    const unsigned long long ma = __ballot(in && !(poly_x(P, o + 1) < s.center));
                    if (base == 0) alive0 = ma;
                    K += __popcll(mi); ns_total += __popcll(ma);
                    if (mi != ~0ull) break;   // (sorted: the first point right of the centre ends the run)
                }
                if (ns_total > cap) return -1;
                // pass 2: every survivor finds its slot
                int my_o = -1;
                for (int base = 0; base < K; base += 64) {
                    const int k = base + lane;
                    if (!slide.points(base, min(base + 64, sg_end))) return -2;
                    slide.rebase_points(P);
                    const int o = P.perm[min(k, sg_end - 1)];
                    const bool al = k < K && !(poly_x(P, o + 1) < s.center);
                    const unsigned long long ma = __ballot(al);
                    if (al) {
                        if (k < ns_total) my_o = o;   // (k < ns_total <= 64: chunk 0, lane k: stays in place)
                        else {                         // a survivor behind the new end: the r-th from the end fills the r-th hole
                            const int after_in_chunk = __popcll(lane == 63 ? 0ull : (ma >> (lane + 1)));
                            const int r = ns_total - alive_before - __popcll(ma) + after_in_chunk;   // survivors behind this one
                            srcpos[r] = (uint16_t)o;
                        }
                    }
                    alive_before += __popcll(ma);
                }
                wave_lds_sync();
                {
                    const unsigned long long pref = ns_total >= 64 ? ~0ull : ((1ull << ns_total) - 1ull);
                    const unsigned long long holes = ~alive0 & pref;
                    if ((holes >> lane) & 1ull) my_o = (int)srcpos[__popcll(holes & ((1ull << lane) - 1ull))];
                }
                wave_lds_sync();
                if (lane < ns_total) {
                    int ca, cb;
                    e_o = my_o;
                    poly_point(P, L, my_o, e_x0, e_z0, e_c0, ca);
                    poly_point(P, L, my_o + 1, e_x1, e_z1, e_c1, cb);
                    e_c0 |= ca == cb ? 1u << 24 : 0u;
                }
                csg_end = ns_total; sg_pointer = K;
                window(min(pt_i, sg_pointer));
                if (lost) return -2;
            }
            first_step = false;
            RP_NEED(min(pt_i, sg_pointer), max(pt_i + 1, sg_pointer));
            while (sg_pointer < sg_end && rl_f(wx, sg_pointer - wbase) < s.center) {
                if (csg_end >= cap) return -1;
                const int wk = sg_pointer - wbase;
                const int o = __builtin_amdgcn_readlane(wo, wk);
                const float nx0 = rl_f(wx, wk), nx1 = rl_f(wx1, wk), nz0 = rl_f(wz0, wk), nz1 = rl_f(wz1, wk);
                const uint32_t nc0 = (uint32_t)__builtin_amdgcn_readlane((int)wc0, wk), nc1 = (uint32_t)__builtin_amdgcn_readlane((int)wc1, wk);
                if (lane == csg_end) { e_o = o; e_x0 = nx0; e_x1 = nx1; e_z0 = nz0; e_z1 = nz1; e_c0 = nc0; e_c1 = nc1; }
                csg_end++; sg_pointer++;
                RP_NEED(min(pt_i, sg_pointer), max(pt_i + 1, sg_pointer));
            }
            // ---- removal: the closed form of the swap-remove scan (poly_sequential_wave) on ballots
            const int n = csg_end;
            const unsigned long long m = __ballot(lane < n && e_x1 < s.center);
            if (m) {
                const int ns = n - __popcll(m);
                const unsigned long long pref = ns >= 64 ? ~0ull : ((1ull << ns) - 1ull);
                const unsigned long long valid = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
                const unsigned long long holes = m & pref, surv = ~m & valid & ~pref;
                if (holes) {   // survivors behind the new end move into the holes: k-th hole <- k-th survivor from the end
                    if ((surv >> lane) & 1ull) srcpos[__popcll(lane == 63 ? 0ull : (surv >> (lane + 1)))] = (uint16_t)lane;
                    wave_lds_sync();
                    const bool hole = (holes >> lane) & 1ull;
                    const int src = hole ? (int)srcpos[__popcll(holes & ((1ull << lane) - 1ull))] : lane;
                    const int mo = __shfl(e_o, src);
                    const float mx0 = __shfl(e_x0, src), mx1 = __shfl(e_x1, src), mz0 = __shfl(e_z0, src), mz1 = __shfl(e_z1, src);
                    const uint32_t mc0 = (uint32_t)__shfl((int)e_c0, src), mc1 = (uint32_t)__shfl((int)e_c1, src);
                    if (hole) { e_o = mo; e_x0 = mx0; e_x1 = mx1; e_z0 = mz0; e_z1 = mz1; e_c0 = mc0; e_c1 = mc1; }
                    wave_lds_sync();
                }
                csg_end = ns;
            }
            // ---- selection: first maximum of the closeness over the list (strict compare).  Every entry computes its interpolation
            // weight once -- for the closeness and for its colour contribution
            int best = 0;
            const float ip_e = (s.center - e_x0) / (e_x1 - e_x0);
            if (csg_end != 1) {
                float cl = -INFINITY;
                if (lane < csg_end) {
                    const float c = (1.0f - ip_e) * e_z0 + ip_e * e_z1;
                    if (0.0f < ip_e && ip_e < 1.0f) cl = c;
                }
                float bc = (float)(-1e-7);
                unsigned long long mm = __ballot(bc < cl);
                if (__popcll(mm) <= 4) {
                    while (mm) {   // (a few candidates: a scalar pass over them)
                        const int b = __ffsll((long long)mm) - 1;
                        mm &= mm - 1;
                        const float v = rl_f(cl, b);
                        if (bc < v) { bc = v; best = b; }
                    }
                } else {
                    // Long lists (depth noise: 25-50 entries): the scan's strict compare keeps the FIRST maximum, i.e. the lowest
                    // lane that holds the wave maximum -- by DPP, not by 8 scalar instructions per candidate (round 4: with 20 such
                    // waves per CU the replay kernel was bound by the CU's one scalar unit: 203 of 222 ms per 8 noise frames)
                    const float clc = cl > -INFINITY ? cl : -INFINITY;   // (NaN -> -inf: the scan skips what does not compare greater)
                    const float mx = wave_max_dpp(clc);
                    if (bc < mx) best = __ffsll((long long)__ballot(clc == mx)) - 1;
                }
            }
            // ---- colour contribution of the winner (poly_accumulate's arithmetic, :1981-1989): worked out by every entry for itself
            // (vector instructions, no LDS), the winner's three values picked up with v_readlane
            {
                const bool flat_e = (e_c0 >> 24) != 0u;
                const float om = 1.0f - ip_e;
                const float sg = s.sig64 ? (float)s.sig_d : s.sig_f;
                float t[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float il = (float)((e_c0 >> (8 * c)) & 0xffu), ir = (float)((e_c1 >> (8 * c)) & 0xffu);
                    const float a = il * om;
                    const float b = ir * ip_e;
                    t[c] = flat_e ? il * s.sig_f : (a + b) * sg;
                }
                const bool flat_w = (((uint32_t)__builtin_amdgcn_readlane((int)e_c0, best)) >> 24) != 0u;
                if (flat_w && s.sig64) {   // (both ends one source pixel, whole-pixel piece: the float64 typing of :1983)
                    const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)e_c0, best);
#pragma unroll
                    for (int c = 0; c < 3; c++) color[c] = (float)((double)color[c] + (double)(uint8_t)((cw >> (8 * c)) & 0xffu) * s.sig_d);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; c++) color[c] = color[c] + rl_f(t[c], best);
                }
            }
            pt_i++;
            RP_NEED(pt_i, pt_i + 1);   // (the loop condition and the next step's sub-interval read positions pt_i and pt_i + 1)
        }
        if (lane == 0) emit(col, csm::f32_to_u8_wrap(color[0]), csm::f32_to_u8_wrap(color[1]), csm::f32_to_u8_wrap(color[2]));
    }
    return 0;
#undef RP_NEED
}

// rasterise the forward segments into per-pixel lists; PASS 0 counts, PASS 1 fills
template <int PASS>
__device__ __forceinline__ void poly_seg_pixels(const Poly& P, int o, int& p0, int& p1) {
    float x0 = poly_x(P, o), x1 = poly_x(P, o + 1);
    p0 = 1; p1 = 0;
    if (!(x0 < x1)) return;  // reversed / degenerate segments are never active
    float f0 = floorf(x0), f1 = floorf(x1);
    if (f1 < 0.0f || f0 > (float)(P.w - 1)) return;
    p0 = f0 < 0.0f ? 0 : (int)f0;
    p1 = f1 > (float)(P.w - 1) ? P.w - 1 : (int)f1;
}

// windows of the stretch replay kernel (k_poly_replay): sorted points / source columns one stretch may touch
#define RP_PWS 512    // sorted positions / source columns a replay wave holds in LDS at a time (sliding windows, k_poly_replay:
#define RP_CWS 1024   // 8.3 KB per wave -- 19 waves per CU by LDS, 20 by registers)
#define RP_CWS_WIDE 4096   // ... for halos beyond 230 columns
#define RP_PW 1024    // a typical stretch's windows in the dump (pool sizing only: the windows have no upper limit any more)
#define RP_CW 1024
#define RP_DESC 8   // words per stretch descriptor
#ifndef RPL_K
#define RPL_K 16   // list entries a lane of k_poly_replay_lanes holds (18 B each in LDS)
#endif
#ifndef RPL_KL
#define RPL_KL 64  // ... of its second instantiation (lists of dozens: noise depth; 74 KB of LDS per wave, two waves per CU)
#endif
struct RpCtx {      // where a row exports its stretches to (dump == null: replay inside the row kernel)
    uint8_t* dump; uint32_t* list; uint32_t* ctr; uint32_t pool16, cap; uint32_t rowid; int eye;   // pool16: dump bytes / 16
    uint8_t* retry;   // LEAN: one byte per row, set when the row needs the full kernel (its inline replay)
};
// what a stretch leaves in the dump pool: ITS windows of the row's sorted order (perm[pw0 .. pw1]) and of coord_d
// (cd[cmin .. cmax]) -- a few hundred entries each, not the whole row (round 3 dumped 6-8 B per pixel of the row per slot: 3.2 GB
// of workspace for 64 4K frames; the windows average ~2 KB per stretch)
__host__ __device__ inline uint32_t rp_win16(int npw, int ncw) { return (uint32_t)((align16(2 * (size_t)npw) + align16(4 * (size_t)ncw)) >> 4); }

// DIALECT: the instantiation that can run the dialect bits d64 (separate kernels, k_rowwarp<FILL, true>: the D32 kernel keeps
// its registers and its code as they were)
// LEAN (round 4): the instantiation the FIRST pass over the flagged rows runs when the stretch replay kernel is attached: the
// parallel evaluation, the stretch list and the export only -- a row that would need the in-row replay (export impossible,
// no reset points, list overflow) is flagged for the second pass (the full kernel, export off) instead.  Without the replay
// code the kernel fits 64 registers: two 1024-thread workgroups per CU instead of one.
// RANGE (round 5, LEAN only): the first pass over a flagged row-eye confined to the columns [cA, cB) around the tiles that raised
// the hazard (k_polypoint's tile hints) and to the sources [sA, sB) that can reach them (cA - halo - 2 .. cB + halo + 2): every
// phase -- disparities, sort, segment lists, evaluation -- then costs what the range costs, not what the row costs (on saturated
// depth a flagged row-eye has 1.4 of 5 tiles flagged, tools/sessions/r05_s20.sh).  The row kernel as a generalised tile: points of
// sources outside the range cannot lie within 4.5 columns of it, so the sorted order, the segment lists and every sub-interval
// inside [cA, cB) are the whole row's (a predecessor / successor point outside only enters through max(col, a) / min(col + 1, b)).
// Pixels outside keep what the tile kernel wrote: they are final there.  A stretch must start behind a reset pixel INSIDE the
// range; a row that has none in front of its first order-dependent pixel goes to the retry pass (whole row).  cA < 0: whole row.
struct PolyRange { int cA, cB, sA, sB; };
template <int SHARP, bool DIALECT, class Emit, bool LEAN = false>
__device__ void technique_polylines(const Lds& L, int w, const EyeArgs& E, float e32, uint32_t* stats_rw,
                                    const Emit& emit, int dbg, const RpCtx* X = nullptr, int d64_ = 0, double e64 = 0.0,
                                    const PolyRange RG = PolyRange{-1, 0, 0, 0}) {
    const int d64 = DIALECT ? d64_ : 0;
    const bool restricted = LEAN && !DIALECT && RG.cA >= 0;
    // restricted: the points of sources sA .. sB - 1 (ids pfirst .. plast) plus the two sentinels; segments between them (plus the
    // sentinels' segments where the range touches the row's ends)
    const int srcA = restricted ? RG.sA : 0, srcB = restricted ? RG.sB : w;
    const int pfirst = SHARP ? 1 + 2 * srcA : 1 + srcA, plast = SHARP ? 2 * srcB : srcB;   // (whole row: 1 .. npt - 2)
    const int npts_r = plast - pfirst + 3;
    const int tid = threadIdx.x, nt = blockDim.x, lane = lane_id(), wave = wave_id(), nwaves = nt >> 6;
    Poly P;
    P.w = w; P.sharp = SHARP; P.npt = poly_npt(w, SHARP); P.cap = LEAN ? poly_cap_lean(w, SHARP) : poly_cap(w, SHARP);
    P.sep32 = E.sep32;
    char* t = L.tech;
    P.perm = (uint16_t*)t; t += align16(2 * (size_t)P.npt);
    P.binoff = (uint16_t*)t; t += align16(2 * ((size_t)w + 4));
    P.segoff = (uint16_t*)t; t += align16(2 * ((size_t)w + 2));
    P.entries = (uint16_t*)t; t += align16(2 * (size_t)P.cap);
    P.longs = (uint16_t*)t; t += align16(2 * 1024);
    P.cd = L.nd;
    double* xd = (DIALECT && (d64 & 1)) ? (double*)t : nullptr;   // (launch_rowwarp adds the 8 w bytes for the dialect)
    P.xd = xd;
    const int npt = P.npt, nbin = w + 2, LONGCAP = 1024;
    int* flag_hazard = L.misc + 0;
    int* nlong = L.misc + 1;
    int* ntotal = L.misc + 2;
    int* scan_ws = L.misc + 8;
    // P1: coord_d (in place over nd), point x's, histogram of bins (count of bin b at binoff[b+1])
    if (DIALECT && xd) {
        for (int c = tid; c < w; c += nt) {
            const double cd = disparity64(L.nd[c], e64, E.div64);
            xd[c] = (((double)c + 0.5) + cd) + E.sep64;
            L.nd[c] = (float)cd;   // (only |coord_d| as float32 is read from here on: the z of the point)
        }
    } else {
        // (one source beyond either end as well: the replay's window loader fetches the right neighbour of every window point)
        for (int c = max(srcA - 1, 0) + tid; c < min(srcB + 1, w); c += nt) L.nd[c] = disparity(L.nd[c], e32, E.div32, L.tabs);
    }
    for (int i = tid; i < (w + 4) / 2; i += nt) ((unsigned*)P.binoff)[i] = 0;
    for (int i = tid; i < (w + 2) / 2; i += nt) ((unsigned*)P.segoff)[i] = 0;
    // (round 6) rows evaluated in column ranges: the longest per-pixel segment list of the row -- the lists of a range are gone when the
    // stretches are exported, and without a bound on the active list every stretch of such a row went to the wave replay
    int* maxlist = L.misc + 24;
    if (tid == 0) { *flag_hazard = 0; *nlong = 0; *ntotal = 0; *maxlist = 0; }
    __syncthreads();
    if (dbg == 1) return;
    // (the point set as one index space: k = 0 .. npts_r - 1 -> left sentinel, pfirst .. plast, right sentinel)
    auto point_of = [&](int k) -> int { return k == 0 ? 0 : (k == npts_r - 1 ? npt - 1 : pfirst + k - 1); };
    for (int k = tid; k < npts_r; k += nt) atomic_add_u16(P.binoff, poly_bin(P, poly_x(P, point_of(k))) + 1, 1);
    __syncthreads();
    if (dbg == 2) return;
    // P2: counting sort by bin, then rank inside the bin by (x, reference index) == the reference's
    // stable insertion sort (:1941-1946).  After the scatter binoff[b] = END of bin b.
    block_scan_inclusive(P.binoff, nbin + 1, 0, OpAdd(), scan_ws);
    uint16_t* scratch = P.entries;
    for (int k = tid; k < npts_r; k += nt) {
        const int o = point_of(k);
        unsigned slot = atomic_add_u16(P.binoff, poly_bin(P, poly_x(P, o)), 1);
        scratch[slot] = (uint16_t)o;
    }
    __syncthreads();
    for (int k = tid; k < npts_r; k += nt) {
        int o = scratch[k];
        float x = poly_x(P, o);
        int b = poly_bin(P, x);
        int bs = b > 0 ? P.binoff[b - 1] : 0, be = P.binoff[b];
        int r = 0;
        for (int j = bs; j < be; j++) {
            int o2 = scratch[j];
            float x2 = poly_x(P, o2);
            r += (x2 < x || (x2 == x && o2 < o)) ? 1 : 0;
        }
        P.perm[bs + r] = (uint16_t)o;
    }
    __syncthreads();
    if (dbg == 3) return;
    // P3a/b: per-pixel lists of the forward segments that can be active inside the pixel (CSR).
    // Segments spanning > 3 pixels (disocclusion bridges) are rasterised cooperatively, 64 pixels
    // per wave step, instead of serialising one lane.
#ifndef RW_MAX_RANGES
#define RW_MAX_RANGES 4   // (32 measured on 4K noise depth, whose rows need ~20: the rows DO tie nearly everywhere, so the evaluation in
#endif                    // ranges is paid on top of the whole-row replay: 69.8 -> 63.7 frames/s; 4 sends such rows straight to the replay)
    // Column RANGES (round 4): the lists of all pixels together may not fit `entries` (polylines_sharp rows beyond 6 950 columns:
    // 3.0 entries per column needed, 2.9 left by the LDS) -- then the row is evaluated in 2 .. RW_MAX_RANGES ranges of columns, each with its
    // own count / scan / fill / evaluate over the same arrays (segments clipped to the range; boundaries on multiples of 64 so that
    // every hazard / reset word belongs to one range).  The long-segment list of the ranges after the first lives in the (dead
    // since P1) table block, because the first range's hazard / reset words already sit in P.longs.  One range = the old flow.
    int nr = 1;
    bool overflow = false;
    bool ranged = false;
    for (int r = 0; r < nr && !overflow; r++) {
        const int cA = restricted ? RG.cA : (nr == 1 ? 0 : (int)(((long long)w * r / nr) & ~63LL));
        const int cB = restricted ? RG.cB : ((nr == 1 || r == nr - 1) ? w : (int)(((long long)w * (r + 1) / nr) & ~63LL));
        uint16_t* const longs_r = r == 0 ? P.longs : (uint16_t*)L.tabs;
        const int longcap_r = r == 0 ? LONGCAP : (int)(sizeof(csm::PowfTables) / 2);
        if (r > 0 || ranged) {   // (a fresh count: the whole-row pass below only told us that ranges are needed)
            for (int i = tid; i < (w + 2) / 2; i += nt) ((unsigned*)P.segoff)[i] = 0;
            if (tid == 0) { *nlong = 0; *ntotal = 0; }
            __syncthreads();
        }
        int local = 0;
        // (the segment set as one index space: k = 0 .. nseg_r - 1 -> the left sentinel's segment (if the range starts at source 0),
        // pfirst .. plast - 1, the right sentinel's (if it ends at source w - 1))
        const int nseg_r = plast - pfirst + 2;
        auto seg_of = [&](int k) -> int {
            return k == 0 ? (srcA == 0 ? 0 : -1) : (k == nseg_r - 1 ? (srcB == w ? npt - 2 : -1) : pfirst + k - 1);
        };
        for (int k = tid; k < nseg_r; k += nt) {
            const int o = seg_of(k);
            if (o < 0) continue;
            int p0, p1;
            poly_seg_pixels<0>(P, o, p0, p1);
            p0 = max(p0, cA); p1 = min(p1, cB - 1);
            if (p0 > p1) continue;
            if (p1 - p0 <= 2) {
                for (int p = p0; p <= p1; p++) atomic_add_u16(P.segoff, p + 1, 1);
                local += p1 - p0 + 1;
            } else {
                int idx = atomicAdd(nlong, 1);
                if (idx < longcap_r) longs_r[idx] = (uint16_t)o;
                local += p1 - p0 + 1;
            }
        }
        atomicAdd(ntotal, local);
        __syncthreads();
        const int nl = min(*nlong, longcap_r);
        const bool over_r = *nlong > longcap_r || *ntotal > P.cap;
        if (over_r && nr == 1 && !ranged && !DIALECT && !restricted && *ntotal <= RW_MAX_RANGES * (P.cap - P.cap / 8) && w >= 256 && dbg != 31) {
            // too many entries for one pass, few enough for RW_MAX_RANGES: start over in ranges (the prefix sums are 16 bits wide, but
            // per range; the whole row's count of long segments does not matter either -- they are clipped and listed per range)
            nr = min(RW_MAX_RANGES, (*ntotal + (P.cap - P.cap / 8) - 1) / (P.cap - P.cap / 8));
            nr = max(nr, 2);
            ranged = true;
            r = -1;
            __syncthreads();
            continue;
        }
        if (!over_r && nr == 1 && !ranged && dbg == 30 && w >= 256 && !DIALECT && !restricted) {   // (development: two ranges although one would do)
            nr = 2; ranged = true; r = -1;
            __syncthreads();
            continue;
        }
        overflow = over_r;
        if (overflow) break;
        {
        for (int li = wave; li < nl; li += nwaves) {
            int p0, p1;
            poly_seg_pixels<0>(P, longs_r[li], p0, p1);
            p0 = max(p0, cA); p1 = min(p1, cB - 1);
            for (int p = p0 + lane; p <= p1; p += 64) atomic_add_u16(P.segoff, p + 1, 1);
        }
        __syncthreads();
        block_scan_inclusive(P.segoff, w + 1, 0, OpAdd(), scan_ws);
        for (int k = tid; k < nseg_r; k += nt) {
            const int o = seg_of(k);
            if (o < 0) continue;
            int p0, p1;
            poly_seg_pixels<1>(P, o, p0, p1);
            p0 = max(p0, cA); p1 = min(p1, cB - 1);
            if (p0 > p1 || p1 - p0 > 2) continue;
            for (int p = p0; p <= p1; p++) P.entries[atomic_add_u16(P.segoff, p, 1)] = (uint16_t)o;
        }
        for (int li = wave; li < nl; li += nwaves) {
            int p0, p1, o = longs_r[li];
            poly_seg_pixels<1>(P, o, p0, p1);
            p0 = max(p0, cA); p1 = min(p1, cB - 1);
            for (int p = p0 + lane; p <= p1; p += 64) P.entries[atomic_add_u16(P.segoff, p, 1)] = (uint16_t)o;
        }
        __syncthreads();
        if (dbg == 4) return;
        if (DIALECT && (d64 & 2)) {
            // P3c with numba's typing of the sweep (poly_sequential64 has the statements): one lane per output pixel; a pixel
            // whose choice depends on the order of the active list (several active segments, none or two equally close)
            // sends the row to the literal one-lane replay below
            // (round 6: next to the colour, the pixel's hazard and reset bits -- one active segment at its last sub-interval -- as bit rows over
            // the idle long-segment array, as in the float32 branch: the order-dependent stretches are then replayed one by one)
            unsigned long long* hzw64 = (unsigned long long*)P.longs;
            unsigned long long* rsw64 = hzw64 + ((w + 63) >> 6);
            for (int colb = 0; colb < w; colb += nt) {
                const int col = colb + tid;
                bool hazard = false, reset = false;
                if (col < w) {
                float color[3] = {0.5f, 0.5f, 0.5f};
                const int pos0 = P.binoff[col], pos1 = P.binoff[col + 1];
                const int ls = col > 0 ? P.segoff[col - 1] : 0, le = P.segoff[col];
                double prev = (double)col;
                float a = poly_x(P, P.perm[pos0 - 1]);
                for (int k = pos0 - 1; k < pos1; k++) {
                    const float b = poly_x(P, P.perm[k + 1]);
                    const double from_d = fmax((double)col, (double)a) + 1e-7, to_d = fmin((double)(col + 1), (double)b) - 1e-7;
                    const double sig = to_d - from_d, center = from_d + 0.5 * sig;
                    a = b;
                    if (center < prev || center > (double)(col + 1)) hazard = true;
                    prev = center;
                    int nact = 0, nqual = 0, best = -1, single = -1;
                    double bc = -1e-7;
                    bool tie = false;
                    for (int e = ls; e < le; e++) {
                        const int o = P.entries[e];
                        const float x0 = poly_x(P, o), x1 = poly_x(P, o + 1);
                        if (!((double)x0 < center) || (double)x1 < center) continue;
                        nact++;
                        single = o;
                        const double ip_k = (center - (double)x0) / (double)(x1 - x0);
                        if (0.0 < ip_k && ip_k < 1.0) {
                            const double cl = (1.0 - ip_k) * (double)poly_z(P, o) + ip_k * (double)poly_z(P, o + 1);
                            nqual++;
                            if (bc < cl) { bc = cl; best = o; tie = false; }
                            else if (cl == bc) tie = true;
                        }
                    }
                    int seg;
                    if (nact == 1) seg = single;
                    else { if (nqual == 0 || tie) hazard = true; seg = nqual ? best : single; }
                    reset = k == pos1 - 1 && nact == 1;
                    if (seg < 0) { hazard = true; continue; }
                    const int col_l = poly_col(P, seg), col_r = poly_col(P, seg + 1);
                    if (col_l == col_r) {
#pragma unroll
                        for (int c = 0; c < 3; c++) color[c] = (float)((double)color[c] + (double)L.img[3 * col_l + c] * sig);
                    } else {
                        const float x0 = poly_x(P, seg), x1 = poly_x(P, seg + 1);
                        const double ip_k = (center - (double)x0) / (double)(x1 - x0);
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const double v = ((double)L.img[3 * col_l + c] * (1.0 - ip_k) + (double)L.img[3 * col_r + c] * ip_k) * sig;
                            color[c] = (float)((double)color[c] + v);
                        }
                    }
                }
                reset = reset && !hazard;
                if (hazard) *flag_hazard = 1;
                else emit(col, csm::f32_to_u8_wrap(color[0]), csm::f32_to_u8_wrap(color[1]), csm::f32_to_u8_wrap(color[2]));
                }
                const unsigned long long hm = __ballot(hazard), rm = __ballot(reset);
                if (lane == 0 && colb + 64 * wave < w && w <= 8192) { hzw64[(colb >> 6) + wave] = hm; rsw64[(colb >> 6) + wave] = rm; }
            }
        } else {
        // P3c: one lane per output pixel (reference :1951-1991).  Next to the pixel's colour: is it ORDER-DEPENDENT (hazard), and
        // does the active list hold exactly one segment after its last sub-interval (reset, see poly_replay_stretch) -- as bit
        // rows over the (idle) long-segment array
        unsigned long long* hzw = (unsigned long long*)P.longs;
        unsigned long long* rsw = hzw + ((w + 63) >> 6);
        if (restricted)   // (nothing outside the range is evaluated: no events there)
            for (int wi = tid; wi < ((w + 63) >> 6); wi += nt)
                if (wi < (cA >> 6) || wi >= ((cB + 63) >> 6)) { hzw[wi] = 0ull; rsw[wi] = 0ull; }
        for (int colb = cA; colb < cB; colb += nt) {
            const int col = colb + tid;
            bool hazard = false, reset = false;
            if (col < cB) {
                float color[3] = {0.5f, 0.5f, 0.5f};
                const int pos0 = P.binoff[col], pos1 = P.binoff[col + 1];  // bin col+1 = [col, col+1)
                const int ls = col > 0 ? P.segoff[col - 1] : 0, le = P.segoff[col];
                float prev = (float)col;
                float a = poly_x(P, P.perm[pos0 - 1]);
                for (int k = pos0 - 1; k < pos1; k++) {
                    float b = poly_x(P, P.perm[k + 1]);
                    SubInt s = poly_subinterval(col, a, b);
                    a = b;
                    if (s.center < prev || s.center > (float)(col + 1)) hazard = true;
                    prev = s.center;
                    if (s.sig64 ? s.sig_d == 0.0 : s.sig_f == 0.0f) continue;  // adds exactly nothing
                    int nact = 0, nqual = 0, best = -1, single = -1;
                    float bc = (float)(-1e-7);
                    bool tie = false;
                    for (int e = ls; e < le; e++) {
                        int o = P.entries[e];
                        float x0 = poly_x(P, o), x1 = poly_x(P, o + 1);
                        if (!(x0 < s.center) || x1 < s.center) continue;
                        nact++;
                        single = o;
                        float ip_k = (s.center - x0) / (x1 - x0);
                        if (0.0f < ip_k && ip_k < 1.0f) {
                            float cl = (1.0f - ip_k) * poly_z(P, o) + ip_k * poly_z(P, o + 1);
                            nqual++;
                            if (bc < cl) { bc = cl; best = o; tie = false; }
                            else if (cl == bc) tie = true;
                        }
                    }
                    int seg;
                    if (nact == 1) seg = single;
                    else if (nqual == 0 || tie) { hazard = true; seg = single; }
                    else seg = best;
                    if (seg >= 0) poly_accumulate(P, L.img, seg, s.center, s.sig64, s.sig_d, s.sig_f, color);
                    reset = k == pos1 - 1 && nact == 1;
                }
                reset = reset && !hazard;
                if (hazard) *flag_hazard = 1;
                else if (dbg == 5) { if (color[0] == 12345.0f) emit(col, 1, 2, 3); }
                else emit(col, csm::f32_to_u8_wrap(color[0]), csm::f32_to_u8_wrap(color[1]), csm::f32_to_u8_wrap(color[2]));
            }
            const unsigned long long hm = __ballot(hazard), rm = __ballot(reset);
            if (lane == 0 && colb + 64 * wave < cB && w <= 8192) { hzw[(colb >> 6) + wave] = hm; rsw[(colb >> 6) + wave] = rm; }
            if (SHARP && ranged) {   // (wave-uniform; sharp only: in the soft lean kernel this cost 5 more spilled vector registers, - 2 % on saturated depth)
                int ll = col < cB ? (int)P.segoff[col] - (col > 0 ? (int)P.segoff[col - 1] : 0) : 0;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) ll = max(ll, __shfl_xor(ll, d));
                if (lane == 0 && ll > 0) atomicMax(maxlist, ll);
            }
        }
        }
        }
        __syncthreads();   // (the next range reuses segoff / entries)
    }
    if (ranged) {   // the table block served as the later ranges' long-segment list: the row's next eye reads it again (disparity())
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_powf_tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(L.tabs);
        for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += nt) dst[i] = src[i];
    }
    __syncthreads();
    if (restricted && overflow) {   // (the whole-row export below would need the whole row's sorted order)
        if (tid == 0) X->retry[X->rowid] = 1;
        return;
    }
    if (DIALECT && (d64 & 2)) {
        // order-dependent row under numba's typing of the sweep.  Round 6: the stretches between reset pixels, a LANE each (lane 0 of
        // wave si takes stretch si: poly_stretch64, the literal statements from a known state), their lists in the idle tail of the
        // per-pixel segment lists; what that cannot do (no room for the scratch, more than 256 stretches, a list beyond 128 entries,
        // lists that overflowed) is swept whole by one lane as before (poly_sequential64 rewrites every pixel).
        constexpr int NSTR64 = 256, ROOM64 = 128;
        int* nstretch64 = L.misc + 3;
        int* bad64 = L.misc + 4;
        const int tail64 = (*ntotal + 7) & ~7;
        bool done = false;
        if (!overflow && *flag_hazard && w <= 8192 && dbg != 26 && tail64 + 2 * NSTR64 + ROOM64 * nwaves + 8 <= P.cap) {
            uint32_t* slist = (uint32_t*)(P.entries + tail64);          // start | end << 16 (pixel columns)
            uint16_t* wscr = P.entries + tail64 + 2 * NSTR64;           // [nwaves][ROOM64]
            if (tid == 0) {
                const unsigned long long* hzw = (const unsigned long long*)P.longs;
                const unsigned long long* rsw = hzw + ((w + 63) >> 6);
                int count = 0, last_reset = -1, start = 0, end = 0;
                bool open = false;
                for (int wi = 0; wi < (w + 63) >> 6; wi++) {
                    const unsigned long long hz = hzw[wi], rs = rsw[wi];
                    unsigned long long ev = hz | rs;
                    while (ev) {
                        const int b = __ffsll((long long)ev) - 1;
                        ev &= ev - 1;
                        const int col = wi * 64 + b;
                        if ((hz >> b) & 1ull) {
                            if (!open) { open = true; start = last_reset + 1; }
                            end = col;
                        } else {
                            if (open) { if (count < NSTR64) slist[count] = (uint32_t)start | ((uint32_t)end << 16); count++; open = false; }
                            last_reset = col;
                        }
                    }
                }
                if (open) { if (count < NSTR64) slist[count] = (uint32_t)start | ((uint32_t)end << 16); count++; }
                *nstretch64 = count <= NSTR64 ? count : -1;
                *bad64 = 0;
            }
            __syncthreads();
            const int nstr = *nstretch64;
            if (nstr > 0) {
                for (int si = wave; si < nstr; si += nwaves) {
                    const int c0 = (int)(slist[si] & 0xffffu), c1 = (int)(slist[si] >> 16);
                    int rc = 0;
                    if (lane == 0) {
                        int seg0 = -1, sgp0 = 0;
                        if (c0 > 0) {   // the state after pixel c0 - 1: its single active segment, the points left of its last centre (float64 compares)
                            const int r = c0 - 1;
                            const int pos1 = P.binoff[r + 1];
                            const double from_d = fmax((double)r, (double)poly_x(P, P.perm[pos1 - 1])) + 1e-7;
                            const double to_d = fmin((double)(r + 1), (double)poly_x(P, P.perm[pos1])) - 1e-7;
                            const double center = from_d + 0.5 * (to_d - from_d);
                            const int ls = r > 0 ? P.segoff[r - 1] : 0, le = P.segoff[r];
                            for (int e = ls; e < le; e++) {
                                const int o = P.entries[e];
                                if ((double)poly_x(P, o) < center && !((double)poly_x(P, o + 1) < center)) seg0 = o;
                            }
                            sgp0 = pos1;
                            while (sgp0 > 0 && !((double)poly_x(P, P.perm[sgp0 - 1]) < center)) sgp0--;
                            if (seg0 < 0) rc = -3;   // (cannot happen: the pixel was marked because exactly one segment is active there)
                        }
                        if (rc == 0) rc = poly_stretch64(P, L, E.csg_cap, emit, c0, c1, seg0, sgp0, (int)P.binoff[c0], wscr + ROOM64 * wave, ROOM64);
                        if (rc) *bad64 = rc;
                    }
                }
            }
            __syncthreads();
            done = nstr > 0 && *bad64 == 0;
            if (done && stats_rw && tid == 0) atomicAdd(&stats_rw[ST_FALLBACK_ROWS], 1u);
            if (dbg == 14 && stats_rw && tid == 0) {   // diagnostics: row-eyes done in stretches / stretches / attempts that gave up (by reason)
                if (done) { atomicAdd(&stats_rw[12], 1u); atomicAdd(&stats_rw[13], (unsigned)nstr); }
                else atomicAdd(&stats_rw[15], nstr <= 0 ? 1u : (*bad64 == -3 ? 0x100u : 0x10000u));
            }
        }
        if (dbg == 14 && stats_rw && tid == 0 && !done && (overflow || *flag_hazard)) atomicAdd(&stats_rw[14], 1u);
        if (!done && (overflow || *flag_hazard) && tid == 0) {   // the literal replay of the whole row (it rewrites every pixel)
            const int rc = poly_sequential64(P, L, E.csg_cap, emit);
            if (stats_rw) {
                atomicAdd(&stats_rw[ST_FALLBACK_ROWS], 1u);
                if (rc) atomicOr(&stats_rw[ST_ERROR], 1u);
            }
        }
        __syncthreads();
        return;
    }
    // the one segment active at `center` (a reset pixel's last sub-interval) without the pixel's list: a wave-wide scan over the
    // row's segments (rows evaluated in column ranges keep only the last range's lists)
    auto active_segment_at = [&](float center) -> int {
        int found = -1;
        for (int o = lane; o < npt - 1; o += 64)
            if (poly_x(P, o) < center && !(poly_x(P, o + 1) < center)) found = o;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) found = max(found, __shfl_xor(found, d));
        return found;
    };
    // ---- order-dependent pixels: replay the stretches between reset points, one wave per stretch (poly_replay_stretch)
    constexpr int NSTR = 256;           // stretches per row
    int* nstretch = L.misc + 3;
    int* stretch_bad = L.misc + 4;
    const int tail0 = (*ntotal + 7) & ~7;   // the idle tail of the per-pixel segment lists: stretch list + 256 bytes of scratch per wave
    // Scratch of this phase: the stretch list, 256 bytes per wave for the in-row replay, the descriptors of the export.  The full
    // kernel takes it from the idle tail of the per-pixel segment lists -- which a row with many overlapping layers (the rows
    // that tie) does not have: half of the flagged rows of a saturated depth map could not export for that reason alone.  The
    // LEAN instantiation never replays in the row, so the source row's colour codes (3 w bytes) are dead here: its scratch.
    const bool img_scratch = LEAN && 3 * (size_t)w >= 4 * NSTR + 4 * RP_DESC * 64 + 16;
    // A row whose per-pixel segment lists overflowed (dozens of layers over every pixel: depth noise) was not evaluated at all --
    // but its sorted order exists, and that is all the replay needs: the row is ONE stretch from column 0 (empty list, nothing
    // added yet), exported like the others when the pool has the room (6 B per pixel).
    const bool whole_row = overflow && img_scratch && X && X->dump && dbg != 28;
    if (((!overflow && *flag_hazard) || whole_row) && w <= 8192 && dbg != 26 && dbg != 27 &&
        (img_scratch || tail0 + 2 * NSTR + 128 * nwaves + 8 <= P.cap)) {
        uint32_t* slist = img_scratch ? (uint32_t*)L.img : (uint32_t*)(P.entries + tail0);   // start | end << 16 (pixel columns)
        uint16_t* wscr = img_scratch ? nullptr : P.entries + tail0 + 2 * NSTR;
        if (tid == 0 && whole_row) { slist[0] = (uint32_t)(w - 1) << 16; *nstretch = 1; *stretch_bad = 0; }
        if (tid == 0 && !whole_row) {
            const unsigned long long* hzw = (const unsigned long long*)P.longs;
            const unsigned long long* rsw = hzw + ((w + 63) >> 6);
            int count = 0, last_reset = -1, start = 0, end = 0;
            bool open = false, badstart = false;   // (restricted: a stretch that would start in front of the range)
            const int cA0 = restricted ? RG.cA : 0;
            for (int wi = 0; wi < (w + 63) >> 6; wi++) {
                const unsigned long long hz = hzw[wi], rs = rsw[wi];
                if (hz == 0ull) {   // (the common word: no order-dependent pixel)
                    if (rs) {
                        if (open) { if (count < NSTR) slist[count] = (uint32_t)start | ((uint32_t)end << 16); count++; open = false; }
                        last_reset = wi * 64 + 63 - __clzll((long long)rs);
                    }
                    continue;
                }
                unsigned long long ev = hz | rs;
                while (ev) {
                    const int b = __ffsll((long long)ev) - 1;
                    ev &= ev - 1;
                    const int col = wi * 64 + b;
                    if ((hz >> b) & 1ull) {
                        if (!open) { open = true; start = last_reset + 1; badstart = badstart || start < cA0; }
                        end = col;
                    } else {
                        if (open) { if (count < NSTR) slist[count] = (uint32_t)start | ((uint32_t)end << 16); count++; open = false; }
                        last_reset = col;
                    }
                }
            }
            if (open) { if (count < NSTR) slist[count] = (uint32_t)start | ((uint32_t)end << 16); count++; }
            *nstretch = (count <= NSTR && !badstart) ? count : -1;
            *stretch_bad = 0;
        }
        __syncthreads();
        const int nstr = *nstretch;
        // ---- export: the stretches of this row go to the replay kernel (k_poly_replay: one wave per stretch, thousands in
        // flight) instead of being replayed here by 1-3 of this workgroup's 16 waves while the row's LDS stays allocated
        constexpr int RP_NSTR = 64;   // stretches per row the export handles (usually 1-3)
        if (nstr > 0 && nstr <= RP_NSTR && X && X->dump && dbg != 28 &&
            (img_scratch || tail0 + 2 * NSTR + 128 * nwaves + 2 * RP_DESC * RP_NSTR + 8 <= P.cap)) {
            int* rp_ok = L.misc + 5;
            int* rp_slot = L.misc + 6;
            int* rp_base = L.misc + 7;   // first descriptor in the small list
            uint32_t* sinfo = img_scratch ? (uint32_t*)(L.img + 4 * NSTR) : (uint32_t*)(wscr + 128 * nwaves);   // [RP_NSTR][RP_DESC]: the descriptor of a stretch
            if (tid == 0) *rp_ok = 1;
            __syncthreads();
            for (int si = wave; si < nstr; si += nwaves) {
                const int c0 = (int)(slist[si] & 0xffffu), c1 = (int)(slist[si] >> 16);
                int seg0 = -1, sgp0 = 0;
                if (c0 > 0) {   // the state after pixel c0 - 1: its single active segment, the points left of its last centre
                    const int r = c0 - 1;
                    const int pos1 = P.binoff[r + 1];
                    const SubInt sb = poly_subinterval(r, poly_x(P, P.perm[pos1 - 1]), poly_x(P, P.perm[pos1]));
                    if (ranged) seg0 = active_segment_at(sb.center);   // (the lists of the earlier column ranges are gone)
                    else {
                        const int ls = r > 0 ? P.segoff[r - 1] : 0, le = P.segoff[r];
                        for (int e = ls; e < le; e++) {
                            const int o = P.entries[e];
                            if (poly_x(P, o) < sb.center && !(poly_x(P, o + 1) < sb.center)) seg0 = o;
                        }
                    }
                    sgp0 = pos1;
                    while (sgp0 > 0 && !(poly_x(P, P.perm[sgp0 - 1]) < sb.center)) sgp0--;
                }
                // sorted points the replay can look at: from three below its two cursors to a 64-entry window past the last
                // point left of pixel c1 + 1; source columns: those points', their right neighbours', the start segment's
                const int pw0 = max(min((int)P.binoff[c0] - 1, sgp0) - 3, 0), pw1 = min((int)P.binoff[c1 + 1] + 66, npts_r - 1);
                int cmin = 0x7fffffff, cmax = -1;
                for (int i = pw0 + lane; i <= pw1; i += 64) {
                    const int o = P.perm[i];
                    cmin = min(cmin, poly_col(P, o)); cmax = max(cmax, poly_col(P, min(o + 1, npt - 1)));
                }
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { cmin = min(cmin, __shfl_xor(cmin, d)); cmax = max(cmax, __shfl_xor(cmax, d)); }
                if (seg0 >= 0) { cmin = min(cmin, poly_col(P, seg0)); cmax = max(cmax, poly_col(P, seg0 + 1)); }
                // (round 5) can a LANE replay this stretch (k_poly_replay_lanes: lists of at most RPL_K entries)?  The per-pixel segment
                // lists bound the active list; without them (whole-row export, column ranges) the answer is no: bit 31 of word 3
                int longest = (whole_row || (ranged && !SHARP)) ? 0x7fff : (ranged ? *maxlist : 0);   // (ranged, sharp: the row's longest list bounds every stretch's)
                if (!whole_row && !ranged)
                    for (int p = c0 + lane; p <= c1; p += 64) longest = max(longest, (int)P.segoff[p] - (p > 0 ? (int)P.segoff[p - 1] : 0));
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) longest = max(longest, __shfl_xor(longest, d));
                if (lane == 0) {
                    const bool good = c0 == 0 || seg0 >= 0;   // (always: the pixel before was marked because one segment is active)
                    if (!good) *rp_ok = 0;
                    uint32_t* q = sinfo + RP_DESC * si;
                    // (bit 31: not for the 16-entry lane kernel; bit 30 with it: for the 64-entry one -- lists of up to ~56, or unknown)
                    q[1] = slist[si]; q[2] = (uint32_t)seg0;
                    q[3] = (uint32_t)sgp0 | (longest > RPL_K - 3 ? 0x80000000u : 0u) | ((longest > RPL_K - 3 && (longest <= RPL_KL - 8 || longest == 0x7fff)) ? 0x40000000u : 0u);
                    q[4] = (uint32_t)pw0 | ((uint32_t)pw1 << 16);
                    q[5] = (uint32_t)cmin | ((uint32_t)cmax << 16); q[6] = X->rowid | ((uint32_t)X->eye << 31); q[7] = (uint32_t)P.binoff[c0];
                }
            }
            __syncthreads();
            if (*rp_ok && tid == 0) {
                // room in the pool for every window of the row (16-byte units; q[0] of a descriptor = where its windows start)
                uint32_t total16 = 0;
                for (int si = 0; si < nstr; si++) {
                    uint32_t* q = sinfo + RP_DESC * si;
                    q[0] = total16;
                    total16 += rp_win16((int)(q[4] >> 16) - (int)(q[4] & 0xffffu) + 1, (int)(q[5] >> 16) - (int)(q[5] & 0xffffu) + 1);
                }
                // Two independent atomic adds, back to back (their round trips to L2 overlap; a compare-and-swap loop starves
                // here -- a few hundred workgroups reserve at once and most attempts fail: 650 -> 490 frames/s on saturated
                // depth).  The pool counter is 64 bits wide (words 4-5 of the counter block): a full pool cannot make it wrap.
                const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long*>(X->ctr + 4), (unsigned long long)total16);
                const uint32_t base = atomicAdd(&X->ctr[1], (uint32_t)nstr);
                const uint32_t at16 = (uint32_t)at;
                *rp_slot = (int)at16; *rp_base = (int)base;
                const bool room = at + total16 <= (unsigned long long)X->pool16 && base + (uint32_t)nstr <= X->cap;
                if (!room)   // replay here; the reserved descriptors (those inside the list) say "skip"
                    for (uint32_t i = base; i < base + (uint32_t)nstr && i < X->cap; i++) X->list[(size_t)i * RP_DESC] = 0xffffffffu;
                // (give the units back when the POOL was what did not fit -- the counter only grew, so after one row had found
                // the pool full every later row, however small, took the retry pass as well.  ADVICE r5: a plain subtraction can hand
                // out overlapping regions -- A and B fail, A refunds, D succeeds above B's failed units, B refunds, and E lands
                // inside D.  The refund is therefore ONE compare-and-swap that only succeeds while this failed reservation is still
                // the topmost one (counter == at + total16 -> at); otherwise the units stay lost: a later row fails spuriously and
                // takes the retry pass, never an overlap.  One attempt, no loop: the starvation of the round-4 CAS loop does not arise)
                if (!room && at + total16 > (unsigned long long)X->pool16)
                    atomicCAS(reinterpret_cast<unsigned long long*>(X->ctr + 4), at + (unsigned long long)total16, at);
                if (!room) *rp_ok = 0;   // replay here
                if (!room && dbg == 14 && stats_rw) atomicAdd(&stats_rw[14], 1u);   // (diagnostics: rows that found the pool full)
            }
            __syncthreads();
            if (*rp_ok) {
                const uint32_t at16 = (uint32_t)*rp_slot, base = (uint32_t)*rp_base;
                for (int si = 0; si < nstr; si++) {   // (usually 1-3 stretches: the whole workgroup copies each one's windows)
                    const uint32_t* q = sinfo + RP_DESC * si;
                    const int pw0 = (int)(q[4] & 0xffffu), pw1 = (int)(q[4] >> 16), cmin = (int)(q[5] & 0xffffu), cmax = (int)(q[5] >> 16);
                    uint8_t* d = X->dump + ((size_t)(at16 + q[0]) << 4);
                    uint16_t* dperm = (uint16_t*)d;
                    float* dcd = (float*)(d + align16(2 * (size_t)(pw1 - pw0 + 1)));
                    for (int i = tid; i <= pw1 - pw0; i += nt) dperm[i] = P.perm[pw0 + i];
                    for (int c = tid; c <= cmax - cmin; c += nt) dcd[c] = P.cd[cmin + c];
                }
                for (int i = tid; i < nstr * RP_DESC; i += nt) {
                    const int si = i / RP_DESC, k = i - si * RP_DESC;
                    X->list[(size_t)(base + (uint32_t)si) * RP_DESC + k] = k == 0 ? at16 + sinfo[i] : sinfo[i];
                }
                if (stats_rw && tid == 0) atomicAdd(&stats_rw[ST_FALLBACK_ROWS], 1u);
                return;
            }
        }
        if (LEAN) {   // (not exported: the full kernel's business)
            if (tid == 0) X->retry[X->rowid] = 1;
            return;
        }
        if (nstr > 0) {
            bool bad = false;
            for (int si = wave; si < nstr && !bad; si += nwaves) {
                const int c0 = (int)(slist[si] & 0xffffu), c1 = (int)(slist[si] >> 16);
                int seg0 = -1, sgp0 = 0;
                if (c0 > 0) {   // the state after pixel c0 - 1: its single active segment, the points left of its last centre
                    const int r = c0 - 1;
                    const int pos1 = P.binoff[r + 1];
                    const SubInt s = poly_subinterval(r, poly_x(P, P.perm[pos1 - 1]), poly_x(P, P.perm[pos1]));
                    if (ranged) seg0 = active_segment_at(s.center);
                    else {
                        const int ls = r > 0 ? P.segoff[r - 1] : 0, le = P.segoff[r];
                        for (int e = ls; e < le; e++) {
                            const int o = P.entries[e];
                            if (poly_x(P, o) < s.center && !(poly_x(P, o + 1) < s.center)) seg0 = o;
                        }
                    }
                    sgp0 = pos1;
                    while (sgp0 > 0 && !(poly_x(P, P.perm[sgp0 - 1]) < s.center)) sgp0--;
                    if (seg0 < 0) bad = true;   // (cannot happen: the pixel was marked because exactly one segment is active there)
                }
                NoSlide whole;   // (the row is resident)
                Poly Pm = P; Lds Lm = L;
                if (!bad && poly_replay_stretch(Pm, Lm, E.csg_cap, emit, c0, c1, seg0, sgp0, wscr + 128 * wave, (int)P.binoff[c0], whole, 0)) bad = true;
            }
            if (bad && lane == 0) *stretch_bad = 1;
        }
        __syncthreads();
        if (dbg == 14 && stats_rw && tid == 0) {   // diagnostics: rows done in stretches / stretches / rows the stretch form gave up
            atomicAdd(&stats_rw[(nstr > 0 && !*stretch_bad) ? 12 : 14], 1u);
            atomicAdd(&stats_rw[13], (unsigned)max(nstr, 0));
        }
        if (nstr > 0 && !*stretch_bad) {
            if (stats_rw && tid == 0) atomicAdd(&stats_rw[ST_FALLBACK_ROWS], 1u);
            return;
        }
    }
    if (LEAN) {
        if ((overflow || *flag_hazard) && tid == 0) X->retry[X->rowid] = 1;
        return;
    }
    if (overflow || *flag_hazard) {
        // order-dependent row: replay the reference sweep literally -- by the first wave (its 64 lanes share the list work
        // of every step), or by one lane when the wave form runs out of scratch
        float* pxs = nullptr;
        if (!SHARP && (size_t)((char*)P.entries - (char*)P.binoff) >= 4 * (size_t)npt) {   // (bin / segment offsets are dead here)
            pxs = (float*)P.binoff;
            for (int o = tid; o < npt; o += nt) pxs[o] = poly_x(P, o);
        }
        __syncthreads();
        if (tid < 64) {
            int rc = dbg == 26 ? -2 : poly_sequential_wave(P, L, E.csg_cap, emit, pxs);
            if (rc == -2 && tid == 0) rc = poly_sequential(P, L, E.csg_cap, emit);
            if (stats_rw && tid == 0) {
                atomicAdd(&stats_rw[ST_FALLBACK_ROWS], 1u);
                if (rc) atomicOr(&stats_rw[ST_ERROR], 1u);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// hybrid_edge (reference :1837-1848)
//   k_hybrid_splat : enhanced_inverse_mapping_with_mask (:1622-1661).  Every source pixel adds
//     colour*w and w to destination columns j_c-1, j_c, j_c+1 IN SOURCE-x ORDER with a float64
//     intermediate per addition.  One lane per DESTINATION column replays exactly the additions that
//     target it: sources are counting-sorted by j_c (bins -1..w), ranked by x inside each bin, and the
//     lane merges bins j-1, j, j+1 in ascending x.  Result (uint8 image + touched mask) goes to HBM
//     scratch because the fill below needs the rows above and below.
//   technique_hybrid_fill : edge_aware_gap_fill (:1745-1774), 3x3 window over touched neighbours,
//     guidance = float64 luma of the UNWARPED source (:1740-1742, :1845).  Reads only the splat
//     result, so it is embarrassingly parallel.
// ---------------------------------------------------------------------------------------------
__device__ const unsigned long long d_hyb_exp_tab[256] = {CS_EXP_TAB_VALUES};
#define CS_EXP_M05 0x1.368b2fc6f960ap-1   // exp(-0.5)
#define CS_EXP_M10 0x1.78b56362cef38p-2   // exp(-1.0)

__device__ __forceinline__ uint8_t src_u8(const RowArgs& A, int frame, int y, int x, int c) {
    size_t o = (((size_t)frame * A.h + y) * A.w + x) * 3 + c;
    if (A.image_u8) return A.image_u8[o];
    float v = A.image_f32[o] * 255.0f;
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    return (uint8_t)(int)v;
}

__global__ void __launch_bounds__(1024) k_hybrid_splat(RowArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int row = blockIdx.x, frame = blockIdx.y, eyei = blockIdx.z;
    const int w = A.w, h = A.h;
    const EyeArgs& E = A.eye[eyei];
    Lds L = carve(smem, CS_FILL_HYBRID_EDGE, w, 0);
    char* t = L.tech;
    float* destx = (float*)t; t += align16(4 * (size_t)w);
    uint16_t* binoff = (uint16_t*)t; t += align16(2 * ((size_t)w + 4));   // bin b = j_c + 1, b in [0, w+1]
    uint16_t* scratch = (uint16_t*)t; t += align16(2 * ((size_t)w + 4));
    uint16_t* sorted = (uint16_t*)t; t += align16(2 * ((size_t)w + 4));
    unsigned long long* etab = (unsigned long long*)t; t += 2048;
    // dialect bit 0 (float64 disparity chain): dest_x as the float64 the reference holds (launch_hybrid adds the 8 w bytes)
    double* destx64 = (A.d64 & 1) ? (double*)t : nullptr;
    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    uint8_t* base = A.hyb_base + ((((size_t)frame * A.neyes + eyei) * h + row) * w) * 3;
    uint8_t* maskrow = A.hyb_mask + (((size_t)frame * A.neyes + eyei) * h + row) * w;
    if (!E.enabled) return;  // eye = source image; the fill pass never reads this slot
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_powf_tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(L.tabs);
        for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += nt) dst[i] = src[i];
        for (int i = tid; i < 256; i += nt) etab[i] = d_hyb_exp_tab[i];
    }
    for (int x = tid; x < w; x += nt)
        for (int c = 0; c < 3; c++) L.img[3 * x + c] = src_u8(A, frame, row, x, c);
    for (int i = tid; i < (w + 4) / 2 + 1; i += nt) ((unsigned*)binoff)[i] = 0;
    __syncthreads();
    const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
    const float dmin = csm::ord2f(st[E.st_min]), dmax = csm::ord2f(st[E.st_max]);
    const bool flat = dmax == dmin;
    const float range = dmax - dmin;
    const float* drow = E.depth + ((size_t)frame * h + row) * w;
    const int nbin = w + 2;
    for (int x = tid; x < w; x += nt) {
        float d = drow[x] * scale;
        float nd = flat ? 0.0f - A.conv32 : ((d - dmin) / range) - A.conv32;
        float fl;
        if (destx64) {
            const double dx = (((double)x + 0.5) + disparity64(nd, A.e64, E.div64)) + E.sep64;
            destx64[x] = dx;
            const double fd = floor(dx);
            fl = fd < -2.0 ? -2.0f : (fd > (double)w + 1.0 ? (float)w + 1.0f : (float)fd);   // (exact inside the range that matters)
        } else {
            float off = disparity(nd, A.e32, E.div32, L.tabs);
            float dx = ((float)x + 0.5f + off) + E.sep32;
            destx[x] = dx;
            fl = floorf(dx);
        }
        int b = fl < -1.0f ? -1 : (fl > (float)w ? -1 : (int)fl + 1);  // j_c in [-1, w] can still touch a column
        L.nd[x] = __int_as_float(b);
        if (b >= 0) atomic_add_u16(binoff, b + 1, 1);
    }
    __syncthreads();
    block_scan_inclusive(binoff, nbin + 1, 0, OpAdd(), L.misc + 8);
    for (int x = tid; x < w; x += nt) {
        int b = __float_as_int(L.nd[x]);
        if (b >= 0) scratch[atomic_add_u16(binoff, b, 1)] = (uint16_t)x;
    }
    __syncthreads();
    const int total = binoff[nbin - 1];
    for (int k = tid; k < total; k += nt) {
        int x = scratch[k];
        int b = __float_as_int(L.nd[x]);
        int bs = b > 0 ? binoff[b - 1] : 0, be = binoff[b];
        int r = 0;
        for (int j = bs; j < be; j++) r += scratch[j] < x ? 1 : 0;
        sorted[bs + r] = (uint16_t)x;
    }
    __syncthreads();
    for (int j = tid; j < w; j += nt) {
        // bins of j_c = j-1, j, j+1  ->  bin indices j, j+1, j+2
        int p0 = j > 0 ? binoff[j - 1] : 0, e0 = binoff[j];
        int p1 = e0, e1 = binoff[j + 1];
        int p2 = e1, e2 = binoff[j + 2];
        float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, ws = 0.0f;
        bool touched = false;
        while (p0 < e0 || p1 < e1 || p2 < e2) {
            int x0 = p0 < e0 ? sorted[p0] : 0x7fffffff;
            int x1 = p1 < e1 ? sorted[p1] : 0x7fffffff;
            int x2 = p2 < e2 ? sorted[p2] : 0x7fffffff;
            int x;
            if (x0 < x1 && x0 < x2) { x = x0; p0++; }
            else if (x1 < x2) { x = x1; p1++; }
            else { x = x2; p2++; }
            double wg;
            if (destx64) {
                const double diff = destx64[x] - (double)j;
                wg = csm::exp_exact_small(-(diff * diff) / 2.0, etab);
            } else {
                float diff = destx[x] - (float)j;
                float arg = -(diff * diff) / 2.0f;
                wg = csm::exp_exact_small((double)arg, etab);
            }
            acc0 = (float)((double)acc0 + (double)L.img[3 * x + 0] * wg);
            acc1 = (float)((double)acc1 + (double)L.img[3 * x + 1] * wg);
            acc2 = (float)((double)acc2 + (double)L.img[3 * x + 2] * wg);
            ws = (A.d64 & 2) ? (float)((double)ws + wg) : ws + (float)wg;   // (numba: float32 += float64 adds in float64)
            touched = true;
        }
        uint8_t o0 = 0, o1 = 0, o2 = 0;
        if (ws > 0.0f) {
            float v0 = acc0 / ws, v1 = acc1 / ws, v2 = acc2 / ws;
            v0 = v0 < 0.0f ? 0.0f : (v0 > 255.0f ? 255.0f : v0);
            v1 = v1 < 0.0f ? 0.0f : (v1 > 255.0f ? 255.0f : v1);
            v2 = v2 < 0.0f ? 0.0f : (v2 > 255.0f ? 255.0f : v2);
            o0 = (uint8_t)(int)v0; o1 = (uint8_t)(int)v1; o2 = (uint8_t)(int)v2;
        }
        base[3 * j + 0] = o0; base[3 * j + 1] = o1; base[3 * j + 2] = o2;
        maskrow[j] = touched ? 1 : 0;
    }
}

struct Px3f { float x, y, z; };
struct Px3b { uint8_t x, y, z; };
__device__ __forceinline__ float ax32_of(float nd) { return fabsf(nd); }

// -DHYB_WTAB (experiment, round 5): the Gaussian weight of a contribution is exp((double)(-(diff * diff) / 2.0f))
// with diff = dest_x - (float)column, an EXACT float32 difference (reference :1643-1644).  For dest_x >= 512 the float32 dest_x is
// a multiple of 2^-14, so |diff| < 2 is one of 32 768 multiples of 2^-14 (coarser binades use a subset): ONE table of the
// libm-exact doubles, indexed by (int)(|diff| * 2^14), replaces the ~25-instruction float64 evaluation by a gather.
#ifdef HYB_WTAB
__device__ double d_hyb_wtab[32768];
__global__ void __launch_bounds__(256) k_hyb_wtab_init() {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float d = (float)i * 0x1p-14f;
    const float arg = -(d * d) / 2.0f;
    d_hyb_wtab[i] = csm::exp_exact_small((double)arg, d_hyb_exp_tab);
}
#endif

// The same splat for an output TILE of one eye row (node path, float32 image): only the sources within S + 2 columns of
// the tile can touch it, so a 256-thread workgroup stages them three per lane and runs the counting sort, the in-bin
// ranking and the 3-way merge on ~770 sources out of 14 KB of LDS (8 workgroups per CU) instead of on a whole row out of
// 50 KB (2 per CU).  Exponents 2 / 1 take the exact shortcuts of cs_math.h.  Results identical to k_hybrid_splat's.
#define HYT_NT 256
#ifndef HYT_SLOTS
#define HYT_SLOTS 4
#endif
#define HYT_NPT (HYT_NT * HYT_SLOTS)
// FUSED (two-eye layouts of the node path): the tile's pixels go straight to the node outputs -- stereoscope value, no-fill
// mask, this eye's depth-map codes -- and the untouched pixels to the gap list of their row, i.e. everything k_hybrid_out4
// does in a pass of its own (2.1 ms per 16 4K frames at 4 TB/s) happens under the float64 arithmetic of this kernel, which
// leaves the memory pipes idle.  The 3 + 1 byte splat result is still written: k_hybrid_gaps reads the neighbours from it.
// DIA (round 5): the dialect bits A.d64 at tile speed -- bit 0: dest_x, its distance to the column and the exp argument in float64
// (8 more bytes of LDS per staged source; pinned by tests/golden/dialect_f64.npz), bit 1: the weight sum adds in float64 (derived)
template <bool FUSED, bool DIA = false>
__global__ void __launch_bounds__(HYT_NT) k_hybrid_splat_tile(RowArgs A, int S, int T, uint32_t* __restrict__ gap_count,
                                                              uint16_t* __restrict__ gap_list) {
    __shared__ unsigned long long etab[256];
    __shared__ double destx64[DIA ? HYT_NPT : 1];
    const bool f64chain = DIA && (A.d64 & 1);
    __shared__ uint32_t img[HYT_NPT];          // colour codes r | g << 8 | b << 16 of source s0 + j
    __shared__ float destx[HYT_NPT];
    __shared__ short bin[HYT_NPT];             // bin of source j: j_c - (o0 - 1), -1: cannot touch the tile
    __shared__ uint16_t binoff[HYT_NPT + 8];   // [T + 3]
    __shared__ uint16_t scratch[HYT_NPT];
    __shared__ uint16_t sorted[HYT_NPT];
    __shared__ int scan_ws[32];
    __shared__ uint8_t dcode[FUSED ? HYT_NPT : 4];   // depth-map code of source s0 + j (the tile's own columns are among them)
    __shared__ float lut255[FUSED ? 256 : 1];        // k / 255
    const int tid = threadIdx.x;
    const int xi = blockIdx.x;
    // (two-eye launches: blockIdx.y interleaves the eyes by row groups, cs_common.h eye_group_decode; z = frame)
    int yrow = blockIdx.y, eyei = 0;
    if (A.neyes == 2) eye_group_decode((int)blockIdx.y, yrow, eyei);
    const int row = yrow * 8 + (xi & 7);
    if (row >= A.h) return;
    const int tile = xi >> 3;
    const int frame = blockIdx.z;
    const int w = A.w, h = A.h;
    const EyeArgs& E = A.eye[eyei];
    const int o0 = tile * T, wt = min(T, w - o0);
    const size_t out0 = FUSED ? ((size_t)frame * A.out_h + row + E.yoff) * A.out_w + E.xoff : 0;   // output pixel of column 0
    // this eye's depth-map output: (depth * 255).astype(uint8) wraps mod 256 (quirk Q7), code / 255 on three channels
    if (FUSED) lut255[tid] = csm::code_over_255((float)tid);   // (HYT_NT == 256; read after the kernel's barriers)
    auto depth_code_out = [&](int jcol, uint8_t code) {
        float* dd = eyei == 0 ? A.depth_l : A.depth_r;
        const size_t pix = ((size_t)frame * h + row) * w + jcol;
        const float v = lut255[code];
        *reinterpret_cast<Px3f*>(dd + pix * 3) = Px3f{v, v, v};
    };
    auto colour_out = [&](int jcol, uint32_t r, uint32_t g, uint32_t b) {
        const size_t o = out0 + jcol;
        if (A.stereo_is_u8) *reinterpret_cast<Px3b*>(reinterpret_cast<uint8_t*>(A.stereo) + o * 3) = Px3b{(uint8_t)r, (uint8_t)g, (uint8_t)b};
        else *reinterpret_cast<Px3f*>(A.stereo + o * 3) = Px3f{lut255[r], lut255[g], lut255[b]};
        A.mask[o] = (r + g + b) == 0u ? 1.0f : 0.0f;   // GenerateStereo.py:355-361
    };
    if (!E.enabled) {
        if (FUSED) __syncthreads();   // the table  // eye = source image (quirk Q10); the fill pass never reads this slot
        if (FUSED) {
            const float scale0 = (A.scale_from_stats && A.stats[(size_t)frame * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f;
            for (int q = tid; q < wt; q += HYT_NT) {
                const int jcol = o0 + q;
                colour_out(jcol, src_u8(A, frame, row, jcol, 0), src_u8(A, frame, row, jcol, 1), src_u8(A, frame, row, jcol, 2));
                depth_code_out(jcol, csm::f32_to_u8_wrap((E.depth[((size_t)frame * h + row) * w + jcol] * scale0) * 255.0f));
            }
        }
        return;
    }
    const int s0 = max(0, o0 - S - 2), s1 = min(w, o0 + wt + S + 2), ns = s1 - s0;
    const int nbin = wt + 2;   // j_c = o0 - 1 .. o0 + wt
    etab[tid] = d_hyb_exp_tab[tid];
    for (int i = tid; i < (nbin + 4) / 2 + 1; i += HYT_NT) ((unsigned*)binoff)[i] = 0;
    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
    const float dmin = csm::ord2f(st[E.st_min]), dmax = csm::ord2f(st[E.st_max]);
    const bool flat = dmax == dmin;
    const float range = dmax - dmin;
    const size_t rowpix = ((size_t)frame * h + row) * w;
    const float* drow = E.depth + rowpix;
    const int pow_mode = A.dbg == 17 ? 0 : (A.e32 == 2.0f ? 2 : (A.e32 == 1.0f ? 1 : 0));
    const bool range_ok = range > 0x1p-40f && range < 0x1p40f;
    const float yr = range_ok ? rcp_refined(range) : 0.0f;
    __syncthreads();
    // all global loads of the lane's sources first (clamped index: no branches), one memory round trip for the four slots
    Px3f pxv[HYT_SLOTS];
    float dpv[HYT_SLOTS];
#pragma unroll
    for (int k = 0; k < HYT_SLOTS; k++) {
        const int jc = min(tid + k * HYT_NT, ns - 1);
        pxv[k] = *reinterpret_cast<const Px3f*>(A.image_f32 + (rowpix + s0 + jc) * 3);
    }
    float dmul[HYT_SLOTS];
    if (FUSED && A.tilemap) {   // lazy depth-blur tiles (cs_common.h): edge-free tiles come from the gray depth, times the x255 scale
        const LazySel Z = lazy_select(A.tilemap, A.tm_words, frame, h, row, s0, reinterpret_cast<const char*>(drow + s0),
                                      reinterpret_cast<const char*>(A.lazy_gray + rowpix + s0), st[ST_SCALE255]);
#pragma unroll
        for (int k = 0; k < HYT_SLOTS; k++) {
            const int jc = min(tid + k * HYT_NT, ns - 1);
            dpv[k] = lazy_load(Z, (uint32_t)(s0 + jc), (uint32_t)jc, dmul[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < HYT_SLOTS; k++) {
            dmul[k] = scale;
            dpv[k] = drow[s0 + min(tid + k * HYT_NT, ns - 1)];
        }
    }
#pragma unroll
    for (int k = 0; k < HYT_SLOTS; k++) {
        const int j = tid + k * HYT_NT;
        if (j >= ns) break;
        const int x = s0 + j;
        const uint32_t r = (uint32_t)(int)fminf(fmaxf(pxv[k].x * 255.0f, 0.0f), 255.0f);
        const uint32_t g = (uint32_t)(int)fminf(fmaxf(pxv[k].y * 255.0f, 0.0f), 255.0f);
        const uint32_t b = (uint32_t)(int)fminf(fmaxf(pxv[k].z * 255.0f, 0.0f), 255.0f);
        img[j] = r | g << 8 | b << 16;
        const float d = dpv[k] * dmul[k];
        if (FUSED) dcode[j] = csm::f32_to_u8_wrap(d * 255.0f);   // (no global load in the output loop: its latency is exposed there)
        // (d - dmin) / range through the refined reciprocal of the frame's range (cs_common.h div_with: the IEEE quotient for
        // numerators that are 0 or >= 2^-60 and a range within 2^+-40; anything else takes the full division, wave-uniformly)
        const float a = d - dmin;
        float qn = div_with(a, range, yr);
        if (__any(!(a == 0.0f || a >= 0x1p-60f)) || !range_ok) qn = a / range;
        const float nd = flat ? 0.0f - A.conv32 : qn - A.conv32;
        const float ax = fabsf(nd);
        float p;
        bool risky = pow_mode == 0;
        if (pow_mode == 1) p = ax;
        else if (pow_mode == 2) p = csm::square_or_flag(ax, risky);
        else p = 0.0f;
        if (risky) p = csm::powf_exact(ax, A.e32, &c_powf_tables);
        const float off = ((nd >= 0.0f ? 1.0f : -1.0f) * p) * E.div32;
        float dx = ((float)x + 0.5f + off) + E.sep32;
        if (f64chain) {   // (the float32 array then holds floor(dest_x), clamped: everything below only takes its floor)
            const double ax = (double)ax32_of(nd), p64 = A.e64 == 2.0 ? ax * ax : (A.e64 == 1.0 ? ax : pow(ax, A.e64));
            const double dx64 = (((double)x + 0.5) + ((nd >= 0.0f ? 1.0 : -1.0) * p64) * E.div64) + E.sep64;
            destx64[j] = dx64;
            const double fd = floor(dx64);
            dx = fd < (double)(o0 - 4) ? (float)(o0 - 4) : (fd > (double)(o0 + wt + 4) ? (float)(o0 + wt + 4) : (float)fd);
        }
        destx[j] = dx;
        const float fl = floorf(dx);
        int b_ = -1;
        if (fl >= (float)(o0 - 1) && fl <= (float)(o0 + wt)) b_ = (int)fl - (o0 - 1);   // j_c in [o0 - 1, o0 + wt] can touch a tile column
        bin[j] = (short)b_;
    }
    __syncthreads();
    // ---- sources grouped by destination column.  Where the staged polyline runs FORWARD -- floor(dest_x) does not decrease
    // from one source to the next: every tile except those on an occlusion fold -- the order by (column, source) IS the source
    // order, and all that is needed is where each column's sources start: every source writes its index into the columns
    // between its predecessor's and its own (round 4; no histogram, prefix sum, scatter and in-bin ranking: five barriers and
    // four passes over LDS less).  Fold tiles take the counting sort as before.
    bool fwd = true;
#pragma unroll
    for (int k = 0; k < HYT_SLOTS; k++) {
        const int j = tid + k * HYT_NT;
        if (j + 1 < ns) fwd = fwd && floorf(destx[j]) <= floorf(destx[j + 1]);
    }
    const bool mono = __syncthreads_and(fwd) != 0 && A.dbg != 29;   // (dbg 29, development: always the counting sort)
    if (mono) {
        // binoff[b] = the first source whose column bin (unclipped: floor(dest_x) - (o0 - 1)) is >= b, for b = 0 .. nbin
        const float base = (float)(o0 - 1);
#pragma unroll
        for (int k = 0; k < HYT_SLOTS; k++) {
            const int j = tid + k * HYT_NT;
            if (j >= ns) break;
            const float ub = floorf(destx[j]) - base;
            const float pb = j > 0 ? floorf(destx[j - 1]) - base : -1.0f;
            const int lo = pb < -1.0f ? 0 : (pb >= (float)nbin ? nbin + 1 : (int)pb + 1);
            const int hi = ub < 0.0f ? -1 : (ub >= (float)nbin ? nbin : (int)ub);
            for (int b = lo; b <= hi; b++) binoff[b] = (uint16_t)j;
            if (j == ns - 1) for (int b = max(hi + 1, 0); b <= nbin; b++) binoff[b] = (uint16_t)ns;
        }
    } else {
        for (int j = tid; j < ns; j += HYT_NT) {
            const int b_ = bin[j];
            if (b_ >= 0) atomic_add_u16(binoff, b_ + 1, 1);
        }
        __syncthreads();
        block_scan_inclusive(binoff, nbin + 1, 0, OpAdd(), scan_ws);
        for (int j = tid; j < ns; j += HYT_NT) {
            const int b_ = bin[j];
            if (b_ >= 0) scratch[atomic_add_u16(binoff, b_, 1)] = (uint16_t)j;
        }
        __syncthreads();
        const int total = binoff[nbin - 1];
        for (int k = tid; k < total; k += HYT_NT) {
            const int j = scratch[k];
            const int b_ = bin[j];
            const int bs = b_ > 0 ? binoff[b_ - 1] : 0, be = binoff[b_];
            int r = 0;
            for (int t = bs; t < be; t++) r += scratch[t] < j ? 1 : 0;
            sorted[bs + r] = (uint16_t)j;
        }
    }
    __syncthreads();
    uint8_t* base = A.hyb_base + ((((size_t)frame * A.neyes + eyei) * h + row) * w) * 3;
    uint8_t* maskrow = A.hyb_mask + (((size_t)frame * A.neyes + eyei) * h + row) * w;
    for (int q = tid; q < wt; q += HYT_NT) {
        const int jcol = o0 + q;
        // bins of j_c = jcol-1, jcol, jcol+1  ->  local bins q, q+1, q+2
        // (gathering the range into registers through a 6-element sorting network and evaluating the exps predicated, for
        // instruction-level parallelism, was measured SLOWER: 3.6 -> 3.9 ms per 16 frames -- the wave then runs as many
        // exps as its longest lane for every lane; the three weights of every SOURCE computed in the staging loop and read
        // here from an 18 KB float64 LDS array: 3.6 -> 4.4 ms -- fewer workgroups per CU and the 8-byte LDS traffic cost
        // more than the divergence)
        // (forward tiles: binoff[b] = START of bin b and the sources are their own sorted order; fold tiles: binoff[b] = END of bin b)
        int p0 = mono ? binoff[q] : (q > 0 ? binoff[q - 1] : 0), e0 = mono ? binoff[q + 1] : binoff[q];
        int p1 = e0, e1 = mono ? binoff[q + 2] : binoff[q + 1];
        int p2 = e1, e2 = mono ? binoff[q + 3] : binoff[q + 2];
        float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, ws = 0.0f;
        bool touched = false;
#ifdef HYB_WTAB
        const bool wtab = o0 - 1 >= 512;   // every dest_x that can touch the tile is >= o0 - 1 (tile-uniform)
#endif
        auto contribute = [&](int j) {
            const float diff = destx[j] - (float)jcol;
            const float arg = -(diff * diff) / 2.0f;
#ifdef HYB_WTAB
            double wg = wtab ? d_hyb_wtab[(uint32_t)(fabsf(diff) * 16384.0f)] : csm::exp_exact_small((double)arg, etab);
#else
            double wg = csm::exp_exact_small((double)arg, etab);   // (-8 < arg <= 0: no range test, cs_math.h)
#endif
            if (f64chain) {
                const double d64v = destx64[j] - (double)jcol;
                wg = csm::exp_exact_small(-(d64v * d64v) / 2.0, etab);
            }
            const uint32_t c = img[j];
            acc0 = (float)((double)acc0 + (double)(c & 0xffu) * wg);
            acc1 = (float)((double)acc1 + (double)((c >> 8) & 0xffu) * wg);
            acc2 = (float)((double)acc2 + (double)((c >> 16) & 0xffu) * wg);
            ws = (DIA && (A.d64 & 2)) ? (float)((double)ws + wg) : ws + (float)wg;   // (numba: float32 += float64 adds in float64)
            touched = true;
        };
        // The three bins are adjacent in `sorted`.  Where the polyline runs forward the sources of bin q all precede those of
        // bin q + 1 and those precede bin q + 2: the merged source order IS the concatenation sorted[p0 .. e2) -- checked at
        // the two junctions; a wave whose lanes all pass (everywhere but at occlusion folds) skips the 3-way merge.
        const bool ordered = mono || (!(p0 < e0 && e0 < e2 && sorted[e0 - 1] > sorted[e0]) && !(p0 < e1 && e1 < e2 && sorted[e1 - 1] > sorted[e1]));
        if (mono) {
            for (int p = p0; __any(p < e2); p++)
                if (p < e2) contribute(p);
        } else if (__all(ordered || q >= wt)) {
            for (int p = p0; __any(p < e2); p++)
                if (p < e2) contribute(sorted[p]);
        } else {
            while (p0 < e0 || p1 < e1 || p2 < e2) {
                const int x0 = p0 < e0 ? sorted[p0] : 0x7fffffff;
                const int x1 = p1 < e1 ? sorted[p1] : 0x7fffffff;
                const int x2 = p2 < e2 ? sorted[p2] : 0x7fffffff;
                int j;
                if (x0 < x1 && x0 < x2) { j = x0; p0++; }
                else if (x1 < x2) { j = x1; p1++; }
                else { j = x2; p2++; }
                contribute(j);
            }
        }
        uint8_t v[3] = {0, 0, 0};
        if (ws > 0.0f) {
            // (weights are exp(-d^2 / 2) with |d| < 2: ws >= 0.13, the sums are 0 or >= 0.13 -- div_with's proven range)
            const float yw = rcp_refined(ws);
            float v0 = div_with(acc0, ws, yw), v1 = div_with(acc1, ws, yw), v2 = div_with(acc2, ws, yw);
            v0 = v0 < 0.0f ? 0.0f : (v0 > 255.0f ? 255.0f : v0);
            v1 = v1 < 0.0f ? 0.0f : (v1 > 255.0f ? 255.0f : v1);
            v2 = v2 < 0.0f ? 0.0f : (v2 > 255.0f ? 255.0f : v2);
            v[0] = (uint8_t)(int)v0; v[1] = (uint8_t)(int)v1; v[2] = (uint8_t)(int)v2;
        }
        if (!FUSED) {
            base[3 * jcol + 0] = v[0]; base[3 * jcol + 1] = v[1]; base[3 * jcol + 2] = v[2];
            maskrow[jcol] = touched ? 1 : 0;
        } else {
            // (the splat result for k_hybrid_gaps<true>: one dword r | g << 8 | b << 16 | touched << 24 per pixel in the same scratch)
            reinterpret_cast<uint32_t*>(A.hyb_base)[(((size_t)frame * A.neyes + eyei) * h + row) * w + jcol] =
                (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | (touched ? 1u << 24 : 0u);
            colour_out(jcol, v[0], v[1], v[2]);
            depth_code_out(jcol, dcode[jcol - s0]);
            // untouched pixels -> the row's list: one atomic per wave on the row's own counter (the active lanes of this
            // iteration; readfirstlane reads the first ACTIVE lane, which is the one that holds the reservation)
            const unsigned long long gb = __ballot(!touched);
            if (gb) {
                const size_t rid = ((size_t)frame * A.neyes + eyei) * h + row;
                const unsigned long long act = __ballot(true);
                unsigned slot = 0;
                if ((tid & 63) == __ffsll((long long)act) - 1) slot = atomicAdd(&gap_count[rid], (unsigned)__popcll(gb));
                slot = (unsigned)__builtin_amdgcn_readfirstlane((int)slot);
                const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(gb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)gb, 0u));
                if (!touched) gap_list[rid * (size_t)w + slot + below] = (uint16_t)jcol;
            }
        }
    }
}

__device__ __forceinline__ double hyb_guidance(const RowArgs& A, int frame, int y, int x) {
    return (0.299 * (double)src_u8(A, frame, y, x, 0) + 0.587 * (double)src_u8(A, frame, y, x, 1)) +
           0.114 * (double)src_u8(A, frame, y, x, 2);
}

__device__ void technique_hybrid_fill(const Lds& L, const RowArgs& A, int frame, int row, int eyei) {
    const int tid = threadIdx.x, nt = blockDim.x, w = A.w, h = A.h;
    unsigned long long* etab = (unsigned long long*)(L.tech + align16(4 * (size_t)w) + 3 * align16(2 * ((size_t)w + 4)));
    for (int i = tid; i < 256; i += nt) etab[i] = d_hyb_exp_tab[i];
    __syncthreads();
    const uint8_t* base = A.hyb_base + (((size_t)frame * A.neyes + eyei) * h) * (size_t)w * 3;
    const uint8_t* mask = A.hyb_mask + (((size_t)frame * A.neyes + eyei) * h) * (size_t)w;
    for (int j = tid; j < w; j += nt) {
        const uint8_t* b = base + ((size_t)row * w + j) * 3;
        float r0 = (float)b[0], r1 = (float)b[1], r2 = (float)b[2];
        if (mask[(size_t)row * w + j] == 0) {
            float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
            double wt = 0.0;
            double g0 = 0.0;
            bool have_g0 = false;
            for (int di = -1; di <= 1; di++)
                for (int dj = -1; dj <= 1; dj++) {
                    int ni = row + di, nj = j + dj;
                    if (ni < 0 || ni >= h || nj < 0 || nj >= w) continue;
                    if (mask[(size_t)ni * w + nj] == 0) continue;
                    if (!have_g0) { g0 = hyb_guidance(A, frame, row, j); have_g0 = true; }
                    int dsq = di * di + dj * dj;
                    // math.exp(-dsq / 2) for dsq = 1, 2 (the centre is never a touched neighbour): the two values of glibc's exp,
                // i.e. of csm::exp_exact (tests/test_cs_math_host.py compares) -- no need to evaluate them per neighbour
                const double w_s = dsq == 1 ? CS_EXP_M05 : CS_EXP_M10;
                    double diff = g0 - hyb_guidance(A, frame, ni, nj);
                    double w_r = csm::exp_exact_small(-(diff * diff) / 200.0, etab);
                    double wg = w_s * w_r;
                    float wg32 = (float)wg;
                    const uint8_t* nb = base + ((size_t)ni * w + nj) * 3;
                    n0 = n0 + (float)nb[0] * wg32;
                    n1 = n1 + (float)nb[1] * wg32;
                    n2 = n2 + (float)nb[2] * wg32;
                    wt += wg;
                }
            if (wt > 0.0) {
                float wt32 = (float)wt;
                r0 = n0 / wt32; r1 = n1 / wt32; r2 = n2 / wt32;
            }
        }
        r0 = fminf(fmaxf(r0, 0.0f), 255.0f); r1 = fminf(fmaxf(r1, 0.0f), 255.0f); r2 = fminf(fmaxf(r2, 0.0f), 255.0f);
        L.res[3 * j + 0] = (uint8_t)(int)r0; L.res[3 * j + 1] = (uint8_t)(int)r1; L.res[3 * j + 2] = (uint8_t)(int)r2;
    }
    __syncthreads();
}

// hybrid_edge, second pass as a plain elementwise kernel (the row kernel's LDS staging, scans and barriers buy nothing
// here: edge_aware_gap_fill only reads the splat result around the pixel).  One lane per source column of one row: both
// eyes of the pixel, the layout / anaglyph composition of RowOut, the no-fill mask and both depth-map outputs.

// edge_aware_gap_fill (reference :1745-1774) only does arithmetic for untouched pixels with touched neighbours -- a few
// percent of the pixels, but spread over most waves (1-10 lanes each), and every (pixel, neighbour) pair costs a float64
// exp.  So the pairs of the workgroup's 256 pixels (both eyes) are COMPACTED: owners reserve slots with one LDS atomic and
// write descriptors, the weights are computed densely, one pair per lane, and the owners then accumulate their pairs in the
// reference's raster order (the float32 sums are order-dependent).
#define HYF_MAXP (256 * 8)
__global__ void __launch_bounds__(256) k_hybrid_fill(RowArgs A) {
    __shared__ unsigned long long etab[256];
    __shared__ double wgs[HYF_MAXP];       // weight w_s * w_r of pair p
    __shared__ uint16_t desc[HYF_MAXP];    // owner lane (8 bits) | neighbour k = 3 (di + 1) + (dj + 1) (4 bits)
    __shared__ unsigned npairs[2];
    const int tid = threadIdx.x;
    etab[tid] = d_hyb_exp_tab[tid];
    if (tid < 2) npairs[tid] = 0;
    __syncthreads();
    const int j = blockIdx.x * 256 + tid, row = blockIdx.y, frame = blockIdx.z;
    const int w = A.w, h = A.h;
    const bool live = j < w;
    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    const size_t pix = ((size_t)frame * h + row) * w + j;
    uint8_t px[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int e = 0; e < A.neyes; e++) {   // (uniform trip count and branches around the barriers)
        if (A.single >= 0 && A.single != e) continue;
        if (!A.eye[e].enabled) {
            if (live) { px[e][0] = src_u8(A, frame, row, j, 0); px[e][1] = src_u8(A, frame, row, j, 1); px[e][2] = src_u8(A, frame, row, j, 2); }
            continue;
        }
        const uint8_t* base = A.hyb_base + (((size_t)frame * A.neyes + e) * h) * (size_t)w * 3;
        const uint8_t* mask = A.hyb_mask + (((size_t)frame * A.neyes + e) * h) * (size_t)w;
        unsigned t = 0, off = 0;
        if (live) {
            const uint8_t* b = base + ((size_t)row * w + j) * 3;
            px[e][0] = b[0]; px[e][1] = b[1]; px[e][2] = b[2];   // (integer codes: the reference's clamp and cast change nothing)
            if (mask[(size_t)row * w + j] == 0 && A.dbg != 41) {
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const int ni = row + k / 3 - 1, nj = j + k % 3 - 1;
                    if (k != 4 && ni >= 0 && ni < h && nj >= 0 && nj < w && mask[(size_t)ni * w + nj] != 0) t |= 1u << k;
                }
                if (t) {
                    off = atomicAdd(&npairs[e], (unsigned)__popc(t));
                    unsigned r = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++)
                        if (t & (1u << k)) desc[off + r++] = (uint16_t)(tid | (k << 8));
                }
            }
        }
        __syncthreads();
        const unsigned np = npairs[e];
        for (unsigned p = tid; p < np; p += 256) {
            const unsigned d = desc[p];
            const int jo = blockIdx.x * 256 + (int)(d & 0xffu), k = (int)(d >> 8);
            const int ni = row + k / 3 - 1, nj = jo + k % 3 - 1;
            // math.exp(-dsq / 2) for dsq = 1, 2 (the centre is never a touched neighbour): the two values of glibc's exp, i.e.
            // of csm::exp_exact (tests/test_cs_math_host.py compares) -- no need to evaluate them per neighbour
            const double w_s = (k & 1) ? CS_EXP_M05 : CS_EXP_M10;
            const double diff = hyb_guidance(A, frame, row, jo) - hyb_guidance(A, frame, ni, nj);
            wgs[p] = w_s * csm::exp_exact_small(-(diff * diff) / 200.0, etab);
        }
        __syncthreads();
        if (t) {
            float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
            double wt = 0.0;
            unsigned r = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                if (!(t & (1u << k))) continue;
                const int ni = row + k / 3 - 1, nj = j + k % 3 - 1;
                const double wg = wgs[off + r++];
                const float wg32 = (float)wg;
                const uint8_t* nb = base + ((size_t)ni * w + nj) * 3;
                n0 = n0 + (float)nb[0] * wg32;
                n1 = n1 + (float)nb[1] * wg32;
                n2 = n2 + (float)nb[2] * wg32;
                wt += wg;
            }
            if (wt > 0.0) {
                const float wt32 = (float)wt;
                float r0 = n0 / wt32, r1 = n1 / wt32, r2 = n2 / wt32;
                r0 = fminf(fmaxf(r0, 0.0f), 255.0f); r1 = fminf(fmaxf(r1, 0.0f), 255.0f); r2 = fminf(fmaxf(r2, 0.0f), 255.0f);
                px[e][0] = (uint8_t)(int)r0; px[e][1] = (uint8_t)(int)r1; px[e][2] = (uint8_t)(int)r2;
            }
        }
    }
    if (!live) return;
    auto store = [&](int e, uint8_t r, uint8_t g, uint8_t b) {   // (RowOut's destination arithmetic)
        const EyeArgs& E = A.eye[e];
        const size_t o = ((size_t)frame * A.out_h + row + E.yoff) * A.out_w + E.xoff + j;
        if (A.stereo_is_u8) *reinterpret_cast<Px3b*>(reinterpret_cast<uint8_t*>(A.stereo) + o * 3) = Px3b{r, g, b};
        else *reinterpret_cast<Px3f*>(A.stereo + o * 3) = Px3f{csm::code_over_255((float)r), csm::code_over_255((float)g), csm::code_over_255((float)b)};
        A.mask[o] = ((int)r + (int)g + (int)b) == 0 ? 1.0f : 0.0f;  // GenerateStereo.py:355-361
    };
    if (A.out_u8) {
        *reinterpret_cast<Px3b*>(A.out_u8 + pix * 3) = Px3b{px[0][0], px[0][1], px[0][2]};
    } else if (A.anaglyph == 1) store(1, px[0][0], px[1][1], px[1][2]);       // R from eye 0, GB from eye 1 (:1996-2010)
    else if (A.anaglyph == 2) store(1, px[1][0], px[0][1], px[0][2]);
    else {
        for (int e = 0; e < A.neyes; e++) {
            if (A.single >= 0 && A.single != e) continue;
            store(e, px[e][0], px[e][1], px[e][2]);
        }
    }
    // depth-map outputs: (depth*255).astype(uint8) wraps mod 256 (quirk Q7), then /255, 3 channels
    if (A.depth_l) {
        const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
        for (int e = 0; e < 2; e++) {
            const float v = csm::code_over_255((float)csm::f32_to_u8_wrap((A.eye[e].depth[pix] * scale) * 255.0f));
            *reinterpret_cast<Px3f*>((e == 0 ? A.depth_l : A.depth_r) + pix * 3) = Px3f{v, v, v};
        }
    }
}

// ---------------------------------------------------------------------------------------------
// hybrid_edge, second pass in STREAMING form (round 3; eyes in separate slots -- side by side, top / bottom, one eye, the
// uint8 form of apply_stereo_divergence -- and w % 4 == 0).  edge_aware_gap_fill (reference :1745-1774) changes only the
// untouched pixels that have a touched neighbour: a few percent.  So
//   k_hybrid_out4 : four pixels per lane with 16-byte accesses: the splat result of EVERY pixel -> stereoscope slot, mask,
//                   both depth-map outputs (the elementwise 95+ % of the pass at the streaming rate); untouched pixels are
//                   appended to the list of their (frame, eye, row) -- 16-bit columns, one wave-aggregated atomic per wave
//                   on the row's own counter (no hot address);
//   k_hybrid_gaps : one wave per (frame, eye, row), one lane per listed pixel: the 3 x 3 window of touched neighbours in the
//                   reference's raster order (float32 sums are order-dependent), libm-exact float64 exp, and the pixel's
//                   stereoscope value and mask are overwritten where the window is not empty.
// k_hybrid_fill above remains for the anaglyph modes (the composite's mask needs both eyes of a pixel) and odd widths.
// ---------------------------------------------------------------------------------------------
struct U3 { uint32_t x, y, z; };
// (every store instruction of a wave writes ONE contiguous kilobyte: the byte -> float expansion of the stereoscope is
// elementwise in units of 4 bytes -> 16 bytes whatever the pixel boundaries; a first version that gave each lane four whole
// pixels and three 16-byte stores 48 bytes apart ran at 3.4 TB/s instead of the streaming rate)
__global__ void __launch_bounds__(256) k_hybrid_out4(RowArgs A, uint32_t* __restrict__ gap_count, uint16_t* __restrict__ gap_list) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int j0 = blockIdx.x * 1024, row = blockIdx.y, frame = blockIdx.z;   // this workgroup: pixels [j0, j0 + 1024) of the row
    const int w = A.w, h = A.h;
    const int npx = min(1024, w - j0);            // multiple of 4
    const int nq = npx * 3 / 4;                   // 4-byte groups of the npx * 3 colour bytes / floats
    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    const size_t pix0 = ((size_t)frame * h + row) * w + j0;
    using csm::code_over_255;
    for (int e = 0; e < A.neyes; e++) {   // (wave-uniform control flow: the ballots below need whole waves)
        if (A.single >= 0 && A.single != e) continue;
        const EyeArgs& E = A.eye[e];
        const bool on = E.enabled;
        const size_t epix0 = (((size_t)frame * A.neyes + e) * h + row) * (size_t)w + j0;
        const size_t o0 = A.out_u8 ? pix0 : ((size_t)frame * A.out_h + row + E.yoff) * A.out_w + E.xoff + j0;
        // ---- colours: byte group m (4 codes) -> 4 floats k / 255 (or the 4 codes)
        for (int m = tid; m < nq; m += 256) {
            uint32_t c;
            if (on) c = reinterpret_cast<const uint32_t*>(A.hyb_base + epix0 * 3)[m];
            else if (A.image_u8) c = reinterpret_cast<const uint32_t*>(A.image_u8 + pix0 * 3)[m];
            else {   // divergence < 0.001: the source image (quirk Q10), np.clip(x * 255, 0, 255).astype(uint8)
                const float4 v = reinterpret_cast<const float4*>(A.image_f32 + pix0 * 3)[m];
                c = (uint32_t)(int)fminf(fmaxf(v.x * 255.0f, 0.0f), 255.0f) | ((uint32_t)(int)fminf(fmaxf(v.y * 255.0f, 0.0f), 255.0f) << 8) |
                    ((uint32_t)(int)fminf(fmaxf(v.z * 255.0f, 0.0f), 255.0f) << 16) | ((uint32_t)(int)fminf(fmaxf(v.w * 255.0f, 0.0f), 255.0f) << 24);
            }
            if (A.out_u8) reinterpret_cast<uint32_t*>(A.out_u8 + o0 * 3)[m] = c;
            else if (A.stereo_is_u8) reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(A.stereo) + o0 * 3)[m] = c;
            else reinterpret_cast<float4*>(A.stereo + o0 * 3)[m] =
                     make_float4(code_over_255((float)(c & 0xffu)), code_over_255((float)((c >> 8) & 0xffu)),
                                 code_over_255((float)((c >> 16) & 0xffu)), code_over_255((float)(c >> 24)));
        }
        // ---- mask of four whole pixels per lane (GenerateStereo.py:355-361) and the untouched pixels -> the row's list
        const int j = 4 * tid;
        const bool live = j < npx;
        unsigned gap = 0;
        if (live) {
            uint32_t c0, c1, c2;
            if (on) { const U3 v = *reinterpret_cast<const U3*>(A.hyb_base + (epix0 + j) * 3); c0 = v.x; c1 = v.y; c2 = v.z; }
            else {
                uint8_t b[12];
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) b[3 * q + ch] = src_u8(A, frame, row, j0 + j + q, ch);
                c0 = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
                c1 = (uint32_t)b[4] | ((uint32_t)b[5] << 8) | ((uint32_t)b[6] << 16) | ((uint32_t)b[7] << 24);
                c2 = (uint32_t)b[8] | ((uint32_t)b[9] << 8) | ((uint32_t)b[10] << 16) | ((uint32_t)b[11] << 24);
            }
            if (on) {
                const uint32_t mk = *reinterpret_cast<const uint32_t*>(A.hyb_mask + epix0 + j);
#pragma unroll
                for (int q = 0; q < 4; q++) gap |= (((mk >> (8 * q)) & 0xffu) == 0u ? 1u : 0u) << q;
            }
            if (!A.out_u8) {
                const uint32_t s0 = (c0 & 0xff) + ((c0 >> 8) & 0xff) + ((c0 >> 16) & 0xff);
                const uint32_t s1 = (c0 >> 24) + (c1 & 0xff) + ((c1 >> 8) & 0xff);
                const uint32_t s2 = ((c1 >> 16) & 0xff) + (c1 >> 24) + (c2 & 0xff);
                const uint32_t s3 = ((c2 >> 8) & 0xff) + ((c2 >> 16) & 0xff) + (c2 >> 24);
                *reinterpret_cast<float4*>(A.mask + o0 + j) = make_float4(s0 == 0 ? 1.0f : 0.0f, s1 == 0 ? 1.0f : 0.0f, s2 == 0 ? 1.0f : 0.0f, s3 == 0 ? 1.0f : 0.0f);
            }
        }
        unsigned long long mb[4];
        unsigned total = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            mb[q] = __ballot((gap >> q) & 1u);
            total += (unsigned)__popcll(mb[q]);
        }
        if (total) {   // wave-uniform: one atomic per wave on the row's own counter
            const size_t rid = ((size_t)frame * A.neyes + e) * h + row;
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&gap_count[rid], total);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            uint16_t* lst = gap_list + rid * (size_t)w;
            unsigned before = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mb[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb[q], 0u));
                if ((gap >> q) & 1u) lst[base + before + below] = (uint16_t)(j0 + j + q);
                before += (unsigned)__popcll(mb[q]);
            }
        }
    }
    // ---- depth-map outputs: (depth*255).astype(uint8) wraps mod 256 (quirk Q7), then /255 on 3 channels.  Float group m of
    // the [pixel][3] output covers pixels p = 4m / 3 and (m % 3 != 0 ? ... : p + 1): its four floats are
    //   m % 3 == 0: p p p p+1     m % 3 == 1: p p p+1 p+1     m % 3 == 2: p p+1 p+1 p+1
    if (A.depth_l) {
        const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
        for (int e = 0; e < 2; e++) {
            const float* dsrc = A.eye[e].depth + pix0;
            float4* ddst = reinterpret_cast<float4*>((e == 0 ? A.depth_l : A.depth_r) + pix0 * 3);
            for (int m = tid; m < nq; m += 256) {
                const int p = (4 * m) / 3, r = m - 3 * (m / 3);
                const float va = code_over_255((float)csm::f32_to_u8_wrap((dsrc[p] * scale) * 255.0f));
                const float vb = code_over_255((float)csm::f32_to_u8_wrap((dsrc[p + 1] * scale) * 255.0f));   // (p + 1 < npx: 4m + 3 < 3 npx)
                ddst[m] = make_float4(va, r == 2 ? vb : va, r == 0 ? va : vb, vb);
            }
        }
    }
}

// PACKED: the splat result as k_hybrid_splat_tile<true> writes it (one dword per pixel: r | g << 8 | b << 16 | touched << 24)
template <bool PACKED>
__global__ void __launch_bounds__(64) k_hybrid_gaps(RowArgs A, const uint32_t* __restrict__ gap_count, const uint16_t* __restrict__ gap_list) {
    const int row = blockIdx.x, e = blockIdx.y, frame = blockIdx.z;
    const int w = A.w, h = A.h;
    const size_t rid = ((size_t)frame * A.neyes + e) * h + row;
    const unsigned count = gap_count[rid];
    if (count == 0) return;
    const uint16_t* lst = gap_list + rid * (size_t)w;
    const uint8_t* base = A.hyb_base + (((size_t)frame * A.neyes + e) * h) * (size_t)w * 3;
    const uint8_t* mask = A.hyb_mask + (((size_t)frame * A.neyes + e) * h) * (size_t)w;
    for (unsigned i = threadIdx.x; i < count; i += 64) {
        const int j = lst[i];
        float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
        double wt = 0.0, gc = 0.0;
        bool have = false;
        if (PACKED && A.image_f32) {
            // node path: the eight neighbour records first, then the source pixels of the whole window (clamped coordinates,
            // nine 12-byte loads in flight), then the sums in the reference's raster order out of registers -- the loop
            // below waits for two dependent global loads per touched neighbour, one after the other
            uint32_t pkv[9];
            const uint32_t* rec = reinterpret_cast<const uint32_t*>(A.hyb_base) + (((size_t)frame * A.neyes + e) * h) * (size_t)w;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int ni = row + k / 3 - 1, nj = j + k % 3 - 1;
                const bool in = k != 4 && ni >= 0 && ni < h && nj >= 0 && nj < w;
                const uint32_t v = rec[(size_t)min(max(ni, 0), h - 1) * w + min(max(nj, 0), w - 1)];
                pkv[k] = in ? v : 0u;
            }
            Px3f gv[9];
            const float* im = A.image_f32 + ((size_t)frame * h) * (size_t)w * 3;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int ni = min(max(row + k / 3 - 1, 0), h - 1), nj = min(max(j + k % 3 - 1, 0), w - 1);
                gv[k] = *reinterpret_cast<const Px3f*>(im + ((size_t)ni * w + nj) * 3);
            }
            auto code = [](float v) { return (double)(uint8_t)(int)fminf(fmaxf(v * 255.0f, 0.0f), 255.0f); };
            auto guid = [&](const Px3f& p) { return (0.299 * code(p.x) + 0.587 * code(p.y)) + 0.114 * code(p.z); };
            gc = guid(gv[4]);
#pragma unroll
            for (int k = 0; k < 9; k++) {
                if (k == 4 || (pkv[k] >> 24) == 0u) continue;
                const double w_s = (k & 1) ? CS_EXP_M05 : CS_EXP_M10;
                const double diff = gc - guid(gv[k]);
                const double wg = w_s * csm::exp_exact_small(-(diff * diff) / 200.0, d_hyb_exp_tab);
                const float wg32 = (float)wg;
                n0 = n0 + (float)(pkv[k] & 0xffu) * wg32;
                n1 = n1 + (float)((pkv[k] >> 8) & 0xffu) * wg32;
                n2 = n2 + (float)((pkv[k] >> 16) & 0xffu) * wg32;
                wt += wg;
            }
        } else
#pragma unroll
        for (int k = 0; k < 9; k++) {   // the reference's raster order over the window (:1757-1770)
            if (k == 4) continue;
            const int ni = row + k / 3 - 1, nj = j + k % 3 - 1;
            if (ni < 0 || ni >= h || nj < 0 || nj >= w) continue;
            uint32_t pk = 0;
            if (PACKED) {
                pk = reinterpret_cast<const uint32_t*>(A.hyb_base)[(((size_t)frame * A.neyes + e) * h + ni) * (size_t)w + nj];
                if ((pk >> 24) == 0u) continue;
            } else if (mask[(size_t)ni * w + nj] == 0) continue;
            if (!have) { gc = hyb_guidance(A, frame, row, j); have = true; }
            // math.exp(-dsq / 2) for dsq = 1, 2: the two values of glibc's exp (see k_hybrid_fill)
            const double w_s = (k & 1) ? CS_EXP_M05 : CS_EXP_M10;
            const double diff = gc - hyb_guidance(A, frame, ni, nj);
            const double wg = w_s * csm::exp_exact_small(-(diff * diff) / 200.0, d_hyb_exp_tab);
            const float wg32 = (float)wg;
            if (!PACKED) {
                const uint8_t* nb = base + ((size_t)ni * w + nj) * 3;
                pk = (uint32_t)nb[0] | ((uint32_t)nb[1] << 8) | ((uint32_t)nb[2] << 16);
            }
            n0 = n0 + (float)(pk & 0xffu) * wg32;
            n1 = n1 + (float)((pk >> 8) & 0xffu) * wg32;
            n2 = n2 + (float)((pk >> 16) & 0xffu) * wg32;
            wt += wg;
        }
        if (!(wt > 0.0)) continue;   // no touched neighbour: the pixel stays as k_hybrid_out4 wrote it ("imperfect" mask)
        const float wt32 = (float)wt;
        float r0 = n0 / wt32, r1 = n1 / wt32, r2 = n2 / wt32;
        r0 = fminf(fmaxf(r0, 0.0f), 255.0f); r1 = fminf(fmaxf(r1, 0.0f), 255.0f); r2 = fminf(fmaxf(r2, 0.0f), 255.0f);
        const uint8_t cr = (uint8_t)(int)r0, cg = (uint8_t)(int)r1, cb = (uint8_t)(int)r2;
        if (A.out_u8) {
            *reinterpret_cast<Px3b*>(A.out_u8 + (((size_t)frame * h + row) * w + j) * 3) = Px3b{cr, cg, cb};
        } else {
            const EyeArgs& E = A.eye[e];
            const size_t o = ((size_t)frame * A.out_h + row + E.yoff) * A.out_w + E.xoff + j;
            if (A.stereo_is_u8) *reinterpret_cast<Px3b*>(reinterpret_cast<uint8_t*>(A.stereo) + o * 3) = Px3b{cr, cg, cb};
            else *reinterpret_cast<Px3f*>(A.stereo + o * 3) = Px3f{csm::code_over_255((float)cr), csm::code_over_255((float)cg), csm::code_over_255((float)cb)};
            A.mask[o] = ((int)cr + (int)cg + (int)cb) == 0 ? 1.0f : 0.0f;  // GenerateStereo.py:355-361
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the row kernel
// ---------------------------------------------------------------------------------------------
// Destination of one eye row: converts uint8 pixels to the output layout as they are produced.
// (holds VALUES, not a pointer to the kernel's argument struct: an address-taken by-value kernel argument is copied to
// scratch memory -- 340 bytes per lane -- and a kernel with scratch pays for it at every wave launch, also when the
// flagged-row list is empty: 28 -> 210 us per launch)
struct RowOut {
    uint8_t* out_u8; float* stereo; float* mask;
    const uint8_t* ana_c; uint8_t* ana; const float* lut;   // (LDS) anaglyph stash, byte -> float table
    int h, out_h, out_w, xoff, yoff, anaglyph, stereo_is_u8;
    int frame, row, w;
    bool stash;  // first eye of an anaglyph: only remember the channels the composite takes from it
    __device__ __forceinline__ void operator()(int c, uint8_t r, uint8_t g, uint8_t b) const {
        if (out_u8) {
            uint8_t* d = out_u8 + (((size_t)frame * h + row) * w + c) * 3;
            d[0] = r; d[1] = g; d[2] = b;
            return;
        }
        if (stash) {
            if (anaglyph == 1) ana[c] = r;
            else { ana[2 * c] = g; ana[2 * c + 1] = b; }
            return;
        }
        if (anaglyph == 1) r = ana[c];
        else if (anaglyph == 2) { g = ana[2 * c]; b = ana[2 * c + 1]; }
        const size_t o = ((size_t)frame * out_h + row + yoff) * out_w + xoff + c;
        if (stereo_is_u8) {
            uint8_t* d8 = reinterpret_cast<uint8_t*>(stereo) + o * 3;
            d8[0] = r; d8[1] = g; d8[2] = b;
        } else {
            float* d = stereo + o * 3;
            d[0] = lut[r]; d[1] = lut[g]; d[2] = lut[b];
        }
        if (mask) mask[o] = ((int)r + (int)g + (int)b) == 0 ? 1.0f : 0.0f;  // GenerateStereo.py:355-361 (null: the per-eye intermediate of the anaglyph modes)
    }
};

template <int FILL, bool DIALECT, bool LEAN = false>
// eyemask (flagged rows): the eyes of the row the tile kernel could not finish (k_polypoint flags per eye since round 4: a row whose
// ties sit in one eye only is not evaluated again for the other -- 18 % of the eye rows of a saturated depth map); anaglyph
// layouts compose both eyes here and take both
__device__ __forceinline__ void rowwarp_row(const RowArgs& A, const int row, const int frame, char* smem, const int eyemask = 3) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int w = A.w, h = A.h;
    Lds L = carve(smem, FILL, w, A.anaglyph);
    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    uint32_t* st_rw = A.stats_rw ? A.stats_rw + (size_t)frame * ST_WORDS : nullptr;
    if (A.row_list && st_rw && tid == 0) atomicAdd(&st_rw[ST_TILE_REDO_ROWS], 1u);
    constexpr bool DIRECT = !fill_uses_res(FILL);  // the technique emits pixels itself

    // constants into LDS
    for (int i = tid; i < 256; i += nt) L.lut[i] = (float)i / 255.0f;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_powf_tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(L.tabs);
        for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += nt) dst[i] = src[i];
    }
    // source row -> uint8 RGB (np.clip(x*255, 0, 255).astype(uint8), reference :1508)
    const size_t rowpix = ((size_t)frame * h + row) * w;
    auto stage_image = [&](const int s0, const int s1) {   // source columns s0 .. s1 - 1 (the whole row: 0, w)
    if (A.image_f32) {
        const float* src = A.image_f32 + rowpix * 3;
        if ((w & 3) == 0) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            for (int i = ((3 * s0) >> 2) + tid; i < ((3 * s1 + 3) >> 2); i += nt) {
                float4 v = s4[i];
                float f[4] = {v.x, v.y, v.z, v.w};
                uint32_t pk = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float x = f[j] * 255.0f;
                    x = fminf(fmaxf(x, 0.0f), 255.0f);
                    pk |= (uint32_t)(uint8_t)(int)x << (8 * j);
                }
                reinterpret_cast<uint32_t*>(L.img)[i] = pk;
            }
        } else {
            for (int i = 3 * s0 + tid; i < 3 * s1; i += nt) {
                float x = src[i] * 255.0f;
                x = fminf(fmaxf(x, 0.0f), 255.0f);
                L.img[i] = (uint8_t)(int)x;
            }
        }
    } else {
        const uint8_t* src = A.image_u8 + rowpix * 3;
        for (int i = 3 * s0 + tid; i < 3 * s1; i += nt) L.img[i] = src[i];
    }
    __syncthreads();
    };
    if (!LEAN) stage_image(0, w);   // (LEAN: per eye below -- its export phase uses the colour row as scratch, and an eye may need a range only)

    const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
    for (int e = 0; e < A.neyes; e++) {
        // (field-by-field selects: a dynamically indexed -- or aggregate-selected -- kernel-argument array makes the compiler copy
        // the whole argument struct to scratch memory)
        EyeArgs E;
        E.depth = e ? A.eye[1].depth : A.eye[0].depth;
        E.div32 = e ? A.eye[1].div32 : A.eye[0].div32; E.sep32 = e ? A.eye[1].sep32 : A.eye[0].sep32;
        E.div64 = e ? A.eye[1].div64 : A.eye[0].div64; E.sep64 = e ? A.eye[1].sep64 : A.eye[0].sep64;
        E.enabled = e ? A.eye[1].enabled : A.eye[0].enabled; E.asc = e ? A.eye[1].asc : A.eye[0].asc;
        E.naive_lim = e ? A.eye[1].naive_lim : A.eye[0].naive_lim; E.csg_cap = e ? A.eye[1].csg_cap : A.eye[0].csg_cap;
        E.st_min = e ? A.eye[1].st_min : A.eye[0].st_min; E.st_max = e ? A.eye[1].st_max : A.eye[0].st_max;
        E.xoff = e ? A.eye[1].xoff : A.eye[0].xoff; E.yoff = e ? A.eye[1].yoff : A.eye[0].yoff;
        if (A.single >= 0 && A.single != e) continue;
        if (!A.anaglyph && !((eyemask >> e) & 1)) continue;
        // (round 5) the flagged tiles of this row-eye -> the column range the lean pass works on (technique_polylines, PolyRange):
        // first .. last flagged tile, 192 columns of margin for the reset pixel a stretch starts behind, multiples of 64
        PolyRange RG{-1, 0, 0, 0};
        if (LEAN && A.hint && E.enabled) {
            const uint32_t m = A.hint[((size_t)frame * h + row) * 2 + e];
            if (m) {
                const int t0 = __ffs((int)m) - 1, t1 = 31 - __clz((int)m);
                const int ra = max(t0 * A.hint_T - 192, 0) & ~63;
                const int rb = t1 >= 31 ? w : min(w, ((t1 + 1) * A.hint_T + 192 + 63) & ~63);
                if (rb - ra <= (3 * w) / 4) RG = PolyRange{ra, rb, max(0, ra - A.hint_S - 2), min(w, rb + A.hint_S + 2)};
            }
        }
        const int st0 = RG.cA >= 0 ? max(RG.sA - 1, 0) : 0, st1 = RG.cA >= 0 ? min(RG.sB + 1, w) : w;   // sources staged for this eye
        // (the lean polylines kernel uses the colour row as the scratch of its export phase: staged per eye; the barrier: the previous
        // eye's export may still be reading its scratch)
        if (LEAN) { __syncthreads(); stage_image(st0, st1); }
        const bool last = (e == A.neyes - 1) || A.single >= 0;
        RowOut out{A.out_u8, A.stereo, A.mask, L.ana, L.ana, L.lut, A.h, A.out_h, A.out_w, E.xoff, E.yoff, A.anaglyph, A.stereo_is_u8,
                   frame, row, w, A.anaglyph != 0 && !last};
        if (E.enabled) {
            // normalised depth: (d - min) / (max - min) - convergence (reference :1587-1600)
            float dmin = csm::ord2f(st[E.st_min]), dmax = csm::ord2f(st[E.st_max]);
            const float* drow = E.depth + rowpix;
            const bool flat = dmax == dmin;
            const float range = dmax - dmin;
            if (FILL != CS_FILL_HYBRID_EDGE && FILL != CS_FILL_HYBRID_EDGE_PLUS) {
                for (int c = st0 + tid; c < st1; c += nt) {
                    float d = drow[c] * scale;
                    L.nd[c] = flat ? 0.0f - A.conv32 : ((d - dmin) / range) - A.conv32;
                }
                __syncthreads();
            }
            if (FILL == CS_FILL_NONE || FILL == CS_FILL_NAIVE || FILL == CS_FILL_NAIVE_INTERPOLATING)
                technique_forward<FILL>(L, w, E, A.e32, A.d64, A.e64);
            else if (FILL == CS_FILL_NONE_POST) {
                technique_forward<CS_FILL_NONE>(L, w, E, A.e32, A.d64, A.e64);
                const int* winner = (const int*)L.tech;
                const int init = E.asc ? -1 : 0x7fffffff;
                technique_post_interp(L, w, (int*)L.nd, (int*)L.tech, [=](int c) { return winner[c] != init; });
            } else if (FILL == CS_FILL_INVERSE_POST) {
                technique_inverse(L, w, E, A.e32, A.d64, A.e64);
                const unsigned long long* key = (const unsigned long long*)L.tech;
                const unsigned long long init = ((unsigned long long)csm::f2ord(-1.0f) << 32) | 0xffffffffull;
                technique_post_interp(L, w, (int*)L.nd, (int*)L.tech, [=](int c) { return key[c] > init; });
            } else if (FILL == CS_FILL_INVERSE) technique_inverse(L, w, E, A.e32, A.d64, A.e64);
            else if (FILL == CS_FILL_POLYLINES_SOFT || FILL == CS_FILL_POLYLINES_SHARP) {
                // (eyes in separate output slots: the stretches of order-dependent rows may go to the replay kernel)
                const RpCtx X{A.anaglyph ? nullptr : A.rp_dump, A.rp_list, A.rp_ctr, A.rp_pool16, A.rp_cap,
                              (uint32_t)frame * (uint32_t)A.h + (uint32_t)row, e, A.rp_ctr ? (uint8_t*)A.rp_ctr + 256 : nullptr};
                // (lean instantiation: 64 VGPRs / 80 SGPRs, ~340 scalar and ~77 vector registers spilled.  The row's width and the
                // LDS base re-read through an opaque move per eye -- what the compiler derives from them is recomputed instead of
                // carried across the eye loop: 336 -> 278 scalar, 77 -> 48 vector spills; the same trick as in k_gpuwarp)
                int we = w, l0 = 0;
#ifndef RW_NO_LAUNDER
                if (LEAN) asm volatile("" : "+s"(we), "+s"(l0));
#endif
                const Lds Le = carve(smem + l0, FILL, we, A.anaglyph);
                technique_polylines<FILL == CS_FILL_POLYLINES_SHARP ? 1 : 0, DIALECT, decltype(out), LEAN>(Le, we, E, A.e32, st_rw, out, A.dbg, &X, A.d64, A.e64, RG);
            }
            else if (FILL == CS_FILL_HYBRID_EDGE_PLUS) {
                // hybrid_edge into `res`, then the polylines_soft row into `alt`; pixels that stayed black take the latter
                technique_hybrid_fill(L, A, frame, row, e);
                size_t ta = lds_tech_bytes(CS_FILL_HYBRID_EDGE, w), tb = lds_tech_bytes(CS_FILL_POLYLINES_SOFT, w);
                if (DIALECT && (A.d64 & 1)) tb += align16(8 * (size_t)w);   // Poly::xd behind the polylines arrays (launch_rowwarp adds the bytes)
                uint8_t* alt = (uint8_t*)(L.tech + (ta > tb ? ta : tb));
                for (int c = tid; c < w; c += nt) {
                    float d = drow[c] * scale;
                    L.nd[c] = flat ? 0.0f - A.conv32 : ((d - dmin) / range) - A.conv32;
                }
                __syncthreads();
                auto into_alt = [=](int c, uint8_t r, uint8_t g, uint8_t b) { alt[3 * c] = r; alt[3 * c + 1] = g; alt[3 * c + 2] = b; };
                technique_polylines<0, DIALECT>(L, w, E, A.e32, st_rw, into_alt, A.dbg, nullptr, A.d64, A.e64);
                __syncthreads();
                for (int c = tid; c < w; c += nt)
                    if (L.res[3 * c] == 0 && L.res[3 * c + 1] == 0 && L.res[3 * c + 2] == 0) {
                        L.res[3 * c] = alt[3 * c]; L.res[3 * c + 1] = alt[3 * c + 1]; L.res[3 * c + 2] = alt[3 * c + 2];
                    }
                __syncthreads();
            }
        } else if (DIRECT) {
            for (int c = tid; c < w; c += nt) out(c, L.img[3 * c], L.img[3 * c + 1], L.img[3 * c + 2]);
        } else {
            for (int i = tid; i < 3 * w; i += nt) L.res[i] = L.img[i];
            __syncthreads();
        }
        // ---- store the eye row held in LDS (techniques that do not emit directly) -----------------
        if (!DIRECT) {
            if (A.out_u8) {
                uint8_t* dst = A.out_u8 + rowpix * 3;
                for (int i = tid; i < 3 * w; i += nt) dst[i] = L.res[i];
            } else if (out.stash) {
                for (int c = tid; c < w; c += nt) out(c, L.res[3 * c], L.res[3 * c + 1], L.res[3 * c + 2]);
            } else {
                if (A.anaglyph) {
                    for (int c = tid; c < w; c += nt) {
                        if (A.anaglyph == 1) L.res[3 * c] = L.ana[c];
                        else { L.res[3 * c + 1] = L.ana[2 * c]; L.res[3 * c + 2] = L.ana[2 * c + 1]; }
                    }
                    __syncthreads();
                }
                const int oy = row + E.yoff, ox = E.xoff;
                float* dst = A.stereo + (((size_t)frame * A.out_h + oy) * A.out_w + ox) * 3;
                if (A.stereo_is_u8) {
                    uint8_t* d8 = reinterpret_cast<uint8_t*>(A.stereo) + (((size_t)frame * A.out_h + oy) * A.out_w + ox) * 3;
                    for (int i = tid; i < 3 * w; i += nt) d8[i] = L.res[i];
                } else if ((w & 3) == 0) {
                    float4* d4 = reinterpret_cast<float4*>(dst);
                    for (int i = tid; i < (3 * w) / 4; i += nt) {
                        uint32_t pk = reinterpret_cast<const uint32_t*>(L.res)[i];
                        d4[i] = make_float4(L.lut[pk & 0xff], L.lut[(pk >> 8) & 0xff], L.lut[(pk >> 16) & 0xff],
                                            L.lut[pk >> 24]);
                    }
                } else {
                    for (int i = tid; i < 3 * w; i += nt) dst[i] = L.lut[L.res[i]];
                }
                if (A.mask) {   // (null: the per-eye intermediate of an anaglyph too wide for the stash form, cs_abi.hip run_rows)
                    float* m = A.mask + ((size_t)frame * A.out_h + oy) * A.out_w + ox;
                    for (int c = tid; c < w; c += nt)
                        m[c] = ((int)L.res[3 * c] + (int)L.res[3 * c + 1] + (int)L.res[3 * c + 2]) == 0 ? 1.0f : 0.0f;
                }
            }
        }
        __syncthreads();
    }
    // depth-map outputs: (depth*255).astype(uint8) wraps mod 256 (quirk Q7), then /255, 3 channels
    // (not for the rows k_polypoint flagged -- `hint` says it was the tile kernel: it writes the depth maps of every row itself, whatever
    // it makes of the row's pixels; rewriting them here was 92 KB of stores per flagged row, round 5)
    if (A.depth_l && !(A.row_list && A.hint)) {
        for (int e = 0; e < 2; e++) {
            const float* drow = (e == 0 ? A.eye[0].depth : A.eye[1].depth) + rowpix;
            float* dst = (e == 0 ? A.depth_l : A.depth_r) + rowpix * 3;
            for (int c = tid; c < w; c += nt) {
                float v = L.lut[csm::f32_to_u8_wrap((drow[c] * scale) * 255.0f)];
                dst[3 * c + 0] = v; dst[3 * c + 1] = v; dst[3 * c + 2] = v;
            }
        }
    }
}

// One workgroup per row (both eyes); or, behind the tiled polylines path, a fixed number of workgroups working off the
// list of rows that path flagged (`row_list`: frame * h + row) -- usually empty, so nothing the size of the batch is launched.
template <int FILL, bool DIALECT = false, bool LEAN = false>
#ifndef RW_LEAN_NT
#define RW_LEAN_NT 1024   // threads of the lean instantiation's workgroups and the waves per SIMD its register budget is sized for
#define RW_LEAN_W 8       // (two workgroups share a CU either way: 1024 x 8 = 64 VGPRs; 768 x 6 = 85 VGPRs, five pixels per lane)
#endif
__global__ void __launch_bounds__(LEAN ? RW_LEAN_NT : 1024, LEAN ? RW_LEAN_W : 4) k_rowwarp(RowArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ONE call site of the (force-inlined) row function: as a real call it takes the argument struct by reference, i.e. a
    // 340-byte scratch copy per lane and scratch set-up at every wave launch (28 -> 210 us for the usual EMPTY flagged-row pass)
    const bool listed = A.row_list != nullptr;
    // rows are handed out dynamically (row_count[1] = next index): a row that needs the sequential replay takes 100x
    // longer than the others
    const uint32_t count = listed ? A.row_count[0] : 1u;
    __shared__ uint32_t s_next;
    for (uint32_t it = 0;; it++) {
        int row = blockIdx.x, frame = blockIdx.y, eyemask = 3;
        if (listed) {
            if (threadIdx.x == 0) s_next = atomicAdd(const_cast<uint32_t*>(&A.row_count[1]), 1u);
            __syncthreads();
            const uint32_t i = s_next;
            if (i >= count) break;
            const uint32_t e = A.row_list[i] & 0x3fffffffu;
            eyemask = (int)(A.row_list[i] >> 30);
            row = (int)(e % (uint32_t)A.h); frame = (int)(e / (uint32_t)A.h);
        } else if (it) break;
        rowwarp_row<FILL, DIALECT, LEAN>(A, row, frame, smem, eyemask);
        __syncthreads();  // the row's LDS (and s_next) is reused by the next one
    }
}

// ---------------------------------------------------------------------------------------------
// Stretch replay as a kernel of its own (round 3).  The row kernel holds a whole row in LDS (80 KB at 4K: two rows per CU)
// and an order-dependent row keeps 1-3 of its 16 waves busy for a millisecond: ~2 000 replaying waves on the chip, every
// row's workgroup alive until its longest stretch ends.  A stretch only looks at the sorted points around its pixels and
// at the source columns those points come from -- a few hundred of each -- so the row kernel dumps the row's sorted order
// and coord_d (RowArgs::rp_dump), appends a descriptor per stretch, and this kernel gives every stretch a WAVE with 7 KB of
// LDS windows: 22 waves per CU, 5 600 stretches in flight, pulled from the list through one atomic cursor.
// A stretch the wave form gives up on (list longer than 64, window drift) flags its row in `retry`; run_rows sends those
// rows through the row kernel once more with the export switched off (their whole-row replay).
// ---------------------------------------------------------------------------------------------
struct ReplayOut {   // RowOut's destination arithmetic without the anaglyph stash (eyes in separate slots only)
    uint8_t* out_u8; float* stereo; float* mask;
    int h, out_h, out_w, xoff, yoff, stereo_is_u8, frame, row, w;
    __device__ __forceinline__ void operator()(int c, uint8_t r, uint8_t g, uint8_t b) const {
        if (out_u8) {
            uint8_t* d = out_u8 + (((size_t)frame * h + row) * w + c) * 3;
            d[0] = r; d[1] = g; d[2] = b;
            return;
        }
        const size_t o = ((size_t)frame * out_h + row + yoff) * out_w + xoff + c;
        if (stereo_is_u8) {
            uint8_t* d8 = reinterpret_cast<uint8_t*>(stereo) + o * 3;
            d8[0] = r; d8[1] = g; d8[2] = b;
        } else {
            float* d = stereo + o * 3;
            d[0] = csm::code_over_255((float)r); d[1] = csm::code_over_255((float)g); d[2] = csm::code_over_255((float)b);
        }
        if (mask) mask[o] = ((int)r + (int)g + (int)b) == 0 ? 1.0f : 0.0f;  // GenerateStereo.py:355-361 (null: the per-eye intermediate of the anaglyph modes)
    }
};

// The windows a replay wave holds in LDS SLIDE along the stretch (round 4): PWS sorted positions and CWS source columns around the
// sweep position, refilled from the stretch's dump (global memory: perm[pw0 .. pw1], coord_d[cmin .. cmax]; the colours from the
// image) whenever poly_replay_stretch asks for a range they do not hold.  4.3 KB per wave instead of 7 (or 44 for a stretch that
// spans a row): 32 waves per CU whatever the stretch length -- a row without reset points (noise depth) is one stretch, and a
// wave, like any other.
template <int PWS, int CWS>
struct DumpSlide {
    static constexpr bool active = true;
    const uint16_t* gperm; int pw0, pw1;     // gperm[i] = perm[pw0 + i]
    const float* gcd; int cmin, cmax;        // gcd[c] = coord_d[cmin + c]
    const RowArgs* A; int frame, row;
    uint16_t* permw; float* cdw; uint8_t* imgw;
    int pb, cb;                              // first sorted position / source column in LDS (very negative: nothing loaded yet)
    // (the caller re-bases its own pointers after every call: a Poly* in here would keep the row descriptor in memory)
    __device__ __forceinline__ void rebase_points(Poly& P) const { P.perm = permw - pb; }
    __device__ __forceinline__ void rebase_columns(Poly& P, Lds& L) const { P.cd = cdw - cb; L.img = imgw - 3 * cb; }
    // (lo / hi are wave-uniform by construction, but some callers computed them through cross-lane shuffles, which the compiler
    // must treat as divergent: `lost` and with it EVERY loop-carried value of the replay then lived in vector registers and
    // every branch became an exec-mask branch -- on a kernel that is bound by the CU's scalar unit.  readfirstlane says so.)
    __device__ __forceinline__ bool points(int lo, int hi) {   // positions [lo, hi)
        lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
        if (lo >= pb && hi <= pb + PWS) return true;
        if (hi - lo > PWS) return false;
        const int lane = threadIdx.x & 63;
        wave_lds_sync();
        pb = max(lo - 8, pw0 - 8);            // (a little slack below: the sweep steps back by one point per pixel)
        for (int i = lane; i < PWS; i += 64) permw[i] = gperm[min(max(pb + i, pw0), pw1) - pw0];
        wave_lds_sync();
        return true;
    }
    __device__ __forceinline__ bool holds(int lo, int hi) const {   // (per lane: are the columns [lo, hi] in LDS?)
        return max(lo, cmin) >= cb && min(hi, cmax) < cb + CWS;
    }
    __device__ __forceinline__ bool columns(int lo, int hi) {   // columns [lo, hi], clipped to the stretch's window
        lo = __builtin_amdgcn_readfirstlane(max(lo, cmin)); hi = __builtin_amdgcn_readfirstlane(min(hi, cmax));
        if (lo >= cb && hi < cb + CWS) return true;
        if (hi - lo + 1 > CWS) return false;
        const int lane = threadIdx.x & 63;
        wave_lds_sync();
        cb = lo;
        for (int i = lane; i < CWS; i += 64) {
            const int c = min(cb + i, cmax);
            cdw[i] = gcd[c - cmin];
            imgw[3 * i] = src_u8(*A, frame, row, c, 0);
            imgw[3 * i + 1] = src_u8(*A, frame, row, c, 1);
            imgw[3 * i + 2] = src_u8(*A, frame, row, c, 2);
        }
        wave_lds_sync();
        return true;
    }
};

template <int SHARP, int PWS, int CWS>
#ifndef RP_MINW
#define RP_MINW 5   // waves per SIMD the register budget is sized for (96 VGPRs, 4 spilled; 4: 108 VGPRs -3 %, 6: -2 %, 8: -33 %)
#endif
__global__ void __launch_bounds__(64, CWS > 2048 ? 1 : RP_MINW) k_poly_replay(RowArgs A, uint8_t* __restrict__ retry, int halo) {   // (wide windows: LDS-bound)
    __shared__ uint16_t permw[PWS];
    __shared__ float cdw[CWS];
    __shared__ uint8_t imgw[3 * CWS];
    __shared__ uint16_t srcpos[128];
    const int lane = threadIdx.x;
    const uint32_t count = min(A.rp_ctr[1], A.rp_cap);
    if (count == 0) return;   // (the usual case: no order-dependent row in the batch -- no traffic on the cursor word)
    const int w = A.w, h = A.h;
    for (;;) {
        uint32_t idx = 0;
        if (lane == 0) idx = atomicAdd(&A.rp_ctr[2], 1u);
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx >= count) break;
        const uint32_t* q = A.rp_list + (size_t)idx * RP_DESC;
        const uint32_t slot = q[0];
        if (slot == 0xffffffffu) continue;
        const int c0 = (int)(q[1] & 0xffffu), c1 = (int)(q[1] >> 16), seg0 = (int)q[2], sgp0 = (int)(q[3] & 0x3fffffffu);   // (bits 30 / 31: which lane kernel may try it)
        const int pw0 = (int)(q[4] & 0xffffu), pw1 = (int)(q[4] >> 16), cmin = (int)(q[5] & 0xffffu), cmax = (int)(q[5] >> 16);
        const uint32_t rowid = q[6] & 0x7fffffffu;
        const int eye = (int)(q[6] >> 31), pt0 = (int)q[7];
        const int frame = (int)(rowid / (uint32_t)h), row = (int)(rowid - (uint32_t)frame * (uint32_t)h);
        const uint8_t* d = A.rp_dump + ((size_t)slot << 4);   // this stretch's windows (export in technique_polylines)
        Poly P;
        P.w = w; P.sharp = SHARP; P.npt = poly_npt(w, SHARP); P.cap = poly_cap(w, SHARP);
        P.sep32 = eye ? A.eye[1].sep32 : A.eye[0].sep32;
        P.perm = nullptr; P.binoff = nullptr; P.segoff = nullptr; P.entries = nullptr; P.longs = nullptr;
        P.cd = nullptr; P.xd = nullptr;   // (dialect D32 only: run_rows does not attach the replay kernel otherwise)
        Lds L;
        L.lut = nullptr; L.tabs = nullptr; L.misc = nullptr; L.res = nullptr; L.ana = nullptr; L.nd = nullptr; L.tech = nullptr;
        L.img = nullptr;
        DumpSlide<PWS, CWS> slide{(const uint16_t*)d, pw0, pw1, (const float*)(d + align16(2 * (size_t)(pw1 - pw0 + 1))), cmin, cmax,
                                  &A, frame, row, permw, cdw, imgw, -0x40000000, -0x40000000};
        const ReplayOut out{A.out_u8, A.stereo, A.mask, h, A.out_h, A.out_w, eye ? A.eye[1].xoff : A.eye[0].xoff,
                            eye ? A.eye[1].yoff : A.eye[0].yoff, A.stereo_is_u8, frame, row, w};
        int rc;
        if (pw1 - pw0 + 1 <= PWS - 8 && cmax - cmin + 1 <= CWS) {   // (8: the slack points() keeps below its first position)
            // the usual stretch: its windows fit at once -- loaded whole, replayed without the sliding checks (-10 % on saturated
            // depth against sliding everything; wave-uniform branch, the replay code exists twice)
            (void)slide.points(pw0, pw1 + 1);
            (void)slide.columns(cmin, cmax);
            slide.rebase_points(P);
            slide.rebase_columns(P, L);
            NoSlide whole;
            rc = poly_replay_stretch(P, L, eye ? A.eye[1].csg_cap : A.eye[0].csg_cap, out, c0, c1, seg0, sgp0, srcpos, pt0, whole, 0);
        } else
            rc = poly_replay_stretch(P, L, eye ? A.eye[1].csg_cap : A.eye[0].csg_cap, out, c0, c1, seg0, sgp0, srcpos, pt0, slide, halo);
        if (rc && lane == 0) retry[rowid] = 1;
        if (A.dbg == 14 && A.stats_rw && lane == 0)   // diagnostics: stretches replayed / given up (list > 64) / given up (window drift)
            atomicAdd(&A.stats_rw[(size_t)frame * ST_WORDS + (rc == 0 ? 12 : (rc == -1 ? 13 : 15))], 1u);
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// k_poly_replay_lanes (round 5): a LANE per stretch.  k_poly_replay gives a stretch a whole wave -- the list work of a step is
// spread over the lanes, everything else is wave-uniform and lands on the CU's one scalar unit, which is what bounds it (19 waves
// per CU x ~380 scalar instructions per sub-interval step).  On the inputs that matter most (depth saturated to 0 / 1: the active
// list holds two to four segments) there is no list work to spread: here every lane replays ITS stretch with the reference's
// literal sweep (:1951-1991, the statements of poly_sequential above) -- list of at most RPL_K segment ids per lane in LDS, points
// and coord_d straight from the stretch's dump windows, colours from the image -- 64 stretches per wave, no scalar work per step.
// A lane whose list outgrows RPL_K, or that would have to read outside its windows, gives up; a stretch that was finished is marked
// "skip" (slot word of its descriptor) and k_poly_replay, launched behind this kernel, does what is left.
// A stretch that starts at column 0 begins with the reference's bulk add of every point left of the first centre (77 at 4K and
// divergence 8) and the swap-with-last removal pass over them: simulated on the VIRTUAL list (position i holds perm[i] until it is
// written; the pass only ever writes position ci, and everything below ci is a survivor) -- no storage beyond the survivors.
// ---------------------------------------------------------------------------------------------
// Per step a lane's only dependent global loads are: the next sorted point (perm -> coord_d, issued at the top of the step), the
// right end of a segment that is added (coord_d), and the winner's colours.  The list entries keep both ends' x and |disparity|
// in LDS (18 B per entry and lane), so that removal, closeness scan and the winner's interpolation read no global memory; the sweep
// keeps x of the points around its cursor in registers (the first sub-interval of a column spans the same two points as the
// last one of the column before: the reference's `while x < col: pt_i++; pt_i--` never moves).
template <int SHARP, int K>
__global__ void __launch_bounds__(64) k_poly_replay_lanes(RowArgs A) {
    extern __shared__ __attribute__((aligned(16))) char lane_smem[];
    float4* const ent_all = reinterpret_cast<float4*>(lane_smem);                 // {x0, x1, z0, z1} of entry k of lane l at [64 k + l]
    uint16_t* const csg_all = reinterpret_cast<uint16_t*>(lane_smem + (size_t)K * 64 * 16);   // its segment id
    constexpr bool LONG = K > RPL_K;   // the second instantiation: the stretches flagged for it (bits 31 + 30 of word 3), its own cursor
    const int lane = threadIdx.x;
    const uint32_t count = min(A.rp_ctr[1], A.rp_cap);
    if (count == 0) return;
    const int w = A.w, h = A.h, npt = poly_npt(w, SHARP);
    uint16_t* const csg = csg_all + lane;
    float4* const ent = ent_all + lane;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&A.rp_ctr[LONG ? 0 : 3], 64u);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= count) break;
        const uint32_t idx = base + (uint32_t)lane;
        if (idx >= count) continue;
        uint32_t* const q = A.rp_list + (size_t)idx * RP_DESC;
        const uint32_t slot = q[0];
        if (slot == 0xffffffffu || (LONG ? (q[3] >> 30) != 3u : (q[3] >> 31) != 0u)) continue;   // (done / another kernel's)
        const int c0 = (int)(q[1] & 0xffffu), c1 = (int)(q[1] >> 16), seg0 = (int)q[2], sgp0 = (int)(q[3] & 0x3fffffffu);
        const int pw0 = (int)(q[4] & 0xffffu), pw1 = (int)(q[4] >> 16), cmin = (int)(q[5] & 0xffffu), cmax = (int)(q[5] >> 16);
        const uint32_t rowid = q[6] & 0x7fffffffu;
        const int eye = (int)(q[6] >> 31), pt0 = (int)q[7];
        const int frame = (int)(rowid / (uint32_t)h), row = (int)(rowid - (uint32_t)frame * (uint32_t)h);
        const uint8_t* d = A.rp_dump + ((size_t)slot << 4);
        const uint16_t* const gperm = (const uint16_t*)d - pw0;                                       // gperm[i], i in [pw0, pw1]
        const float* const gcd = (const float*)(d + align16(2 * (size_t)(pw1 - pw0 + 1))) - cmin;     // gcd[c], c in [cmin, cmax]
        const float sep32 = eye ? A.eye[1].sep32 : A.eye[0].sep32;
        const int cap = min(eye ? A.eye[1].csg_cap : A.eye[0].csg_cap, poly_cap(w, SHARP));
        const size_t rowpix = ((size_t)frame * h + row) * w;
        bool lost = false;   // a read outside the windows / a list beyond RPL_K: the wave kernel's business
        int err = 0;         // the reference's own list capacity exceeded (it would raise)
        auto perm_at = [&](int i) -> int { if (i < pw0 || i > pw1) { lost = true; return 0; } return (int)gperm[i]; };
        auto cd_at = [&](int c) -> float { if (c < cmin || c > cmax) { lost = true; return 0.0f; } return gcd[c]; };
        auto pcol = [&](int o) -> int { return o <= 0 ? 0 : (o >= npt - 1 ? w - 1 : (SHARP ? (o - 1) >> 1 : o - 1)); };
        // x and |coord_d| of point o (poly_x / poly_z)
        auto pxz = [&](int o, float& x, float& z) {
            if (o <= 0) { x = (float)(-1.0 * w); z = 0.0f; return; }
            if (o >= npt - 1) { x = (float)(2.0 * w); z = 0.0f; return; }
            const int c = SHARP ? (o - 1) >> 1 : o - 1;
            const float cd = cd_at(c);
            x = ((float)c + 0.5f + cd) + sep32;
            if (SHARP) x = ((o - 1) & 1) ? x + (float)0.45 : x - (float)0.45;
            z = fabsf(cd);
        };
        auto px = [&](int o) -> float { float x, z; pxz(o, x, z); return x; };
        auto code = [&](int col, int c) -> float {   // the source row as uint8 codes (rowwarp_row's stage_image), as a float
            if (A.image_f32) {
                float x = A.image_f32[(rowpix + col) * 3 + c] * 255.0f;
                x = fminf(fmaxf(x, 0.0f), 255.0f);
                return (float)(uint8_t)(int)x;
            }
            return (float)A.image_u8[(rowpix + col) * 3 + c];
        };
        auto push = [&](int k, int o, float x0, float z0) {   // entry k = segment (o, o + 1)
            float x1, z1;
            pxz(o + 1, x1, z1);
            csg[64 * k] = (uint16_t)o; ent[64 * k] = make_float4(x0, x1, z0, z1);
        };
        const ReplayOut out{A.out_u8, A.stereo, A.mask, h, A.out_h, A.out_w, eye ? A.eye[1].xoff : A.eye[0].xoff,
                            eye ? A.eye[1].yoff : A.eye[0].yoff, A.stereo_is_u8, frame, row, w};
        const int sg_end = npt - 1;
        int csg_end = 0, sg_pointer = c0 > 0 ? sgp0 : 0;
        if (c0 > 0) { float x0, z0; pxz(seg0, x0, z0); push(0, seg0, x0, z0); csg_end = 1; }
        // the add cursor: point `os` at sorted position sg_pointer, its x and |coord_d|
        int os = sg_pointer < sg_end ? perm_at(sg_pointer) : 0;
        float xs = 0.0f, zs = 0.0f;
        if (sg_pointer < sg_end) pxz(os, xs, zs);
        // the sweep cursor: pt_i = the last point left of the column; xa / xb = x of the points at pt_i / pt_i + 1
        int pt_i = pt0 - 1;
        float xa = px(perm_at(pt_i)), xb = px(perm_at(pt_i + 1)), xp = xa;
        for (int col = c0; col <= c1 && !lost && !err; col++) {
            float color[3] = {0.5f, 0.5f, 0.5f};
            while (!lost && !err && xa < (float)(col + 1)) {
                // (the point after next: on its way while the list is worked on)
                const int onext = perm_at(pt_i + 2 <= pw1 ? pt_i + 2 : pw1);
                const float xnext = px(onext);
                const SubInt s = poly_subinterval(col, xa, xb);
                if (csg_end == 0 && sg_pointer == 0) {
                    // the bulk add at column 0 and its removal pass, on the virtual list: N points lie left of the centre
                    int N = 0;
                    while (N < sg_end && !lost && px(perm_at(N)) < s.center) N++;
                    if (N > cap) err = 1;
                    int ci = 0, end = N;
                    int cur = N > 0 ? perm_at(0) : 0;
                    while (ci < end && !lost) {
                        if (px(cur + 1) < s.center) { end--; cur = end > ci ? perm_at(end) : cur; }
                        else {
                            if (ci >= K) { lost = true; break; }
                            float x0, z0; pxz(cur, x0, z0); push(ci, cur, x0, z0); ci++;
                            cur = ci < end ? perm_at(ci) : cur;
                        }
                    }
                    csg_end = end; sg_pointer = N;
                    os = sg_pointer < sg_end ? perm_at(sg_pointer) : 0;
                    if (sg_pointer < sg_end) pxz(os, xs, zs);
                } else {
                    while (sg_pointer < sg_end && !lost && xs < s.center) {
                        if (csg_end >= cap) { err = 1; break; }
                        if (csg_end >= K) { lost = true; break; }
                        push(csg_end, os, xs, zs); csg_end++; sg_pointer++;
                        if (sg_pointer < sg_end) { os = perm_at(sg_pointer); pxz(os, xs, zs); }
                    }
                    int ci = 0;
                    while (ci < csg_end) {
                        if (ent[64 * ci].y < s.center) { csg[64 * ci] = csg[64 * (csg_end - 1)]; ent[64 * ci] = ent[64 * (csg_end - 1)]; csg_end--; }
                        else ci++;
                    }
                }
                if (lost || err) break;
                if (csg_end == 0) { lost = true; break; }   // (cannot happen -- the polyline is connected from -w to 2w; slot 0 would be stale)
                int best = 0;
                if (csg_end != 1) {
                    float bc = (float)(-1e-7);
                    for (int ci = 0; ci < csg_end; ci++) {
                        const float4 e = ent[64 * ci];
                        const float ip_k = (s.center - e.x) / (e.y - e.x);
                        const float cl = (1.0f - ip_k) * e.z + ip_k * e.w;
                        if (bc < cl && 0.0f < ip_k && ip_k < 1.0f) { bc = cl; best = ci; }
                    }
                }
                {   // poly_accumulate
                    const int seg = csg[64 * best];
                    const float4 e = ent[64 * best];
                    const int col_l = pcol(seg), col_r = pcol(seg + 1);
                    if (col_l == col_r) {
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            if (s.sig64) color[c] = (float)((double)color[c] + (double)code(col_l, c) * s.sig_d);
                            else color[c] = color[c] + code(col_l, c) * s.sig_f;
                        }
                    } else {
                        const float ip_k = (s.center - e.x) / (e.y - e.x);
                        const float om = 1.0f - ip_k;
                        const float sg = s.sig64 ? (float)s.sig_d : s.sig_f;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const float a = code(col_l, c) * om;
                            const float b = code(col_r, c) * ip_k;
                            color[c] = color[c] + (a + b) * sg;
                        }
                    }
                }
                pt_i++;
                xp = xa; xa = xb; xb = xnext;
                if (pt_i + 1 > pw1) lost = true;   // (xb would be a point outside the window)
            }
            if (!lost && !err) out(col, csm::f32_to_u8_wrap(color[0]), csm::f32_to_u8_wrap(color[1]), csm::f32_to_u8_wrap(color[2]));
            // the next column starts one point back: its first sub-interval spans the points of this column's last one
            pt_i--;
            xb = xa; xa = xp;
        }
        if (!lost && !err) q[0] = 0xffffffffu;   // done: k_poly_replay skips it (a stretch given up here is replayed from its start there)
        if (A.dbg == 14 && A.stats_rw) atomicAdd(&A.stats_rw[(size_t)frame * ST_WORDS + ((lost || err) ? 15 : 12)], 1u);
    }
}

// The dump pool.  A stretch's windows average ~2 KB and an order-dependent eye row holds 1.3 stretches: 4 KB per image row carry
// a batch whose every row ties in places (saturated depth).  A row WITHOUT reset points (depth noise) is one stretch of the
// whole row: 6 B per pixel and eye.  Batches small enough get that much outright (up to 1 GiB: eleven 4K frames -- the chunks of
// the host pipeline are eight), larger ones the 4 KB per row; a row that finds the pool (or the descriptor list) full takes the
// second pass (in-row replay), and a caller can always give more (workspace beyond cs_workspace_bytes extends the pool).
static size_t rp_pool_bytes(size_t rows, int w, int sharp) {
    // (development: CS_DEBUG_PT_VARIANT 41 quadruples the per-row budget -- does a workload run out of pool?)
    // (48: a deliberately tiny pool -- most flagged rows find it full: tests/test_gpu_stress.py exercises the refund path)
    const int variant = dev_switch(CS_DEBUG_PT_VARIANT);
    const size_t per_row = variant == 41 ? 16384 : 4096;
    const size_t every_row = rows * 2 * ((size_t)rp_win16(poly_npt(w, sharp), w) << 4), budget = rows * per_row + (64u << 10);
    const size_t cap = (size_t)1 << 30;
    size_t pool = every_row <= cap ? every_row : (budget > cap ? budget : cap);
    if (variant == 48) pool = rows * 96 + 8192;
    if (pool > every_row) pool = every_row;
    return (pool + 15) & ~(size_t)15;
}
size_t poly_replay_bytes(int n, int h, int w, int sharp) {
    if (w > 8192) return 0;
    const size_t rows = (size_t)n * h;   // descriptors: four stretches per image row
    return al256r(rows * 4 * RP_DESC * 4) + rp_pool_bytes(rows, w, sharp) + 256;
}
// scratch: [descriptor list][dump pool]; `ctr_retry`: [counters 256 B][retry flags, one byte per row], zeroed by the caller
// (it lies in the flagged-row block that run_rows clears with one memset)
hipError_t poly_replay_attach(RowArgs& A, int sharp, void* scratch, void* ctr_retry, hipStream_t stream, size_t surplus) {
    const size_t rows = (size_t)A.n * A.h;
    char* b = (char*)scratch;
    A.rp_ctr = (uint32_t*)ctr_retry;
    A.rp_list = (uint32_t*)b;
    A.rp_dump = (uint8_t*)(b + al256r(rows * 4 * RP_DESC * 4));
    const size_t pool16 = (rp_pool_bytes(rows, A.w, sharp) + surplus) >> 4;   // (surplus: workspace the caller gave beyond cs_workspace_bytes)
    A.rp_pool16 = (uint32_t)(pool16 < 0xffffffffu ? pool16 : 0xffffffffu); A.rp_cap = (uint32_t)(rows * 4);
    (void)stream;
    return hipSuccess;
}
uint8_t* poly_replay_retry_flags(const RowArgs& A) { return (uint8_t*)A.rp_ctr + 256; }
hipError_t launch_poly_replay(int sharp, const RowArgs& A, int halo, hipStream_t stream) {
    // the column window holds the columns of the 64 sorted points around the sweep (their span: 64 + 2 halo) and of the active
    // segments (2 halo + 1): 1024 columns up to a halo of 230, 4096 beyond (a stretch that outgrows them goes to the retry pass)
    const bool wide = 4 * (halo + 2) + 80 > RP_CWS;
    const dim3 grid(256 * (wide ? 5 : 19)), block(64);   // (resident waves per CU: 29 KB / 8.3 KB of LDS each)
    uint8_t* retry = poly_replay_retry_flags(A);
    // first a lane per stretch (short lists: saturated depth), then a wave per stretch for what that kernel gave up
    // (CS_DEBUG_PT_VARIANT 45: the wave kernel alone, as in round 4)
    if (dev_switch(CS_DEBUG_PT_VARIANT) != 45) {
        const dim3 lgrid(256 * 8);
        const size_t lds16 = (size_t)RPL_K * 64 * 18, lds64 = (size_t)RPL_KL * 64 * 18;
        if (sharp) hipLaunchKernelGGL((k_poly_replay_lanes<1, RPL_K>), lgrid, block, lds16, stream, A);
        else hipLaunchKernelGGL((k_poly_replay_lanes<0, RPL_K>), lgrid, block, lds16, stream, A);
        // ... then, ONLY under CS_DEBUG_PT_VARIANT 46, the stretches with lists of dozens (noise depth) by the 64-entry instantiation, two
        // waves per CU.  Measured (tools/sessions/r05_s26.sh): it finishes them all, bit-exact, and is 2.4 x SLOWER than the wave kernel
        // (8 noise frames: 263 against 110 ms) -- one wave per SIMD walks two 40-entry loops of dependent LDS reads per step, 23 us
        // a step.  The scalar-bound wave kernel stays the path for long lists.
        if (dev_switch(CS_DEBUG_PT_VARIANT) == 46) {
            hipError_t e = sharp ? hipFuncSetAttribute((const void*)k_poly_replay_lanes<1, RPL_KL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64)
                                 : hipFuncSetAttribute((const void*)k_poly_replay_lanes<0, RPL_KL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
            if (e != hipSuccess) return e;
            const dim3 g64(256 * 2);
            if (sharp) hipLaunchKernelGGL((k_poly_replay_lanes<1, RPL_KL>), g64, block, lds64, stream, A);
            else hipLaunchKernelGGL((k_poly_replay_lanes<0, RPL_KL>), g64, block, lds64, stream, A);
        }
    }
    if (sharp && wide) hipLaunchKernelGGL((k_poly_replay<1, RP_PWS, RP_CWS_WIDE>), grid, block, 0, stream, A, retry, halo + 2);
    else if (sharp) hipLaunchKernelGGL((k_poly_replay<1, RP_PWS, RP_CWS>), grid, block, 0, stream, A, retry, halo + 2);
    else if (wide) hipLaunchKernelGGL((k_poly_replay<0, RP_PWS, RP_CWS_WIDE>), grid, block, 0, stream, A, retry, halo + 2);
    else hipLaunchKernelGGL((k_poly_replay<0, RP_PWS, RP_CWS>), grid, block, 0, stream, A, retry, halo + 2);
    return hipGetLastError();
}

// rows flagged by the tiled path -> compact list
__global__ void __launch_bounds__(256) k_collect_rows(const uint8_t* __restrict__ flag, int total, uint32_t* count, uint32_t* list) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    // flag byte: bit 0 = the row (both eyes: what every flagging kernel but k_polypoint writes), bits 1 / 2 = eye 0 / eye 1 only
    // (k_polypoint, round 4).  List entry: row index | eye mask << 30
    if (i < total && flag[i]) {
        const unsigned f = flag[i];
        const unsigned mask = (f & 1u) ? 3u : ((f >> 1) & 3u);
        list[atomicAdd(count, 1u)] = (uint32_t)i | (mask << 30);
    }
}

hipError_t launch_collect_rows(const uint8_t* flag, int total, uint32_t* count, uint32_t* list, hipStream_t stream) {
    hipLaunchKernelGGL(k_collect_rows, dim3((total + 255) / 256), dim3(256), 0, stream, flag, total, count, list);
    return hipGetLastError();
}

// host-side launcher (called from cs_abi.hip)
hipError_t launch_rowwarp(int fill, const RowArgs& A, int threads, hipStream_t stream, int max_groups, int lean) {
    size_t lds = lds_common_bytes(fill, A.w, A.anaglyph) + lds_tech_bytes(fill, A.w);
    if ((A.d64 & 1) && (fill == CS_FILL_POLYLINES_SOFT || fill == CS_FILL_POLYLINES_SHARP || fill == CS_FILL_HYBRID_EDGE_PLUS))
        lds += align16(8 * (size_t)A.w);   // Poly::xd
    if (lds + CS_ROW_LDS_STATIC > CS_LDS_BYTES) return hipErrorInvalidValue;
    if ((fill == CS_FILL_NONE_POST || fill == CS_FILL_INVERSE_POST) && (A.w + threads - 1) / threads > 32)
        return hipErrorInvalidValue;   // (technique_post_interp keeps a lane's validity flags in one 32-bit register)
    const long long rows = (long long)A.h * A.n;
    dim3 grid(A.h, A.n), block(threads);
    if (A.row_list) {
        // flagged rows (usually none): persistent workgroups, as many as are resident at once -- every wave of this kernel sets
        // up scratch (the replay code spills), so an empty launch costs per WORKGROUP: 2048 of them took 0.21 ms
        long long per_cu = 2048 / threads;
        if (lds > 0 && (long long)(CS_LDS_BYTES / lds) < per_cu) per_cu = (long long)(CS_LDS_BYTES / lds);
        if (per_cu < 1) per_cu = 1;
        const long long resident = 256 * per_cu;
        long long groups = rows < resident ? rows : resident;
        if (max_groups > 0 && groups > max_groups) groups = max_groups;   // (a pass that is almost always empty: fewer idle launches)
        grid = dim3((unsigned)groups, 1);
    }
#define CS_LAUNCH(F)                                                                                              \
    case F: {                                                                                                     \
        hipError_t e = hipFuncSetAttribute((const void*)k_rowwarp<F>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)lds);                                                             \
        if (e != hipSuccess) return e;                                                                            \
        hipLaunchKernelGGL(k_rowwarp<F>, grid, block, lds, stream, A);                                            \
        break;                                                                                                    \
    }
    // (only where two rows then share a CU: at 4K polylines_sharp's row takes 88 KB of LDS -- one per CU whatever the registers)
    // (... and for sharp rows so wide that their list capacity is reduced, poly_cap: the lean kernel exports the rows whose lists
    // overflow as whole-row stretches, the full kernel would sweep them sequentially)
    const bool reduced_cap = fill == CS_FILL_POLYLINES_SHARP && poly_cap(A.w, 1) < 4 * A.w + 64;
    // (round 6: the lean instantiation carves its per-pixel lists with poly_cap_lean -- what lets two sharp rows share a CU)
    size_t lds_lean = lds;
    if (fill == CS_FILL_POLYLINES_SHARP)
        lds_lean = lds - align16(2 * (size_t)poly_cap(A.w, 1)) + align16(2 * (size_t)poly_cap_lean(A.w, 1));
    if (lean && !(A.d64 & 3) && A.rp_dump && (2 * lds_lean <= CS_LDS_BYTES || reduced_cap) && (fill == CS_FILL_POLYLINES_SOFT || fill == CS_FILL_POLYLINES_SHARP)) {
        // first pass with the replay kernel attached: the instantiation without the in-row replay (64 registers, two rows per CU)
        hipError_t e = fill == CS_FILL_POLYLINES_SOFT
            ? hipFuncSetAttribute((const void*)k_rowwarp<CS_FILL_POLYLINES_SOFT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lean)
            : hipFuncSetAttribute((const void*)k_rowwarp<CS_FILL_POLYLINES_SHARP, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lean);
        if (e != hipSuccess) return e;
        if (A.row_list && lds_lean != lds) {   // (the persistent grid was sized with the full request: two rows per CU now)
            long long per_cu = (long long)(CS_LDS_BYTES / lds_lean);
            if (per_cu > 2048 / threads) per_cu = 2048 / threads;
            if (per_cu < 1) per_cu = 1;
            long long groups = rows < 256 * per_cu ? rows : 256 * per_cu;
            if (max_groups > 0 && groups > max_groups) groups = max_groups;
            grid = dim3((unsigned)groups, 1);
        }
        const dim3 lblock(threads < RW_LEAN_NT ? threads : RW_LEAN_NT);
        if (fill == CS_FILL_POLYLINES_SOFT) hipLaunchKernelGGL((k_rowwarp<CS_FILL_POLYLINES_SOFT, false, true>), grid, lblock, lds_lean, stream, A);
        else hipLaunchKernelGGL((k_rowwarp<CS_FILL_POLYLINES_SHARP, false, true>), grid, lblock, lds_lean, stream, A);
        return hipGetLastError();
    }
    if ((A.d64 & 3) && (fill == CS_FILL_POLYLINES_SOFT || fill == CS_FILL_POLYLINES_SHARP)) {   // the dialect instantiations
        hipError_t e = fill == CS_FILL_POLYLINES_SOFT
            ? hipFuncSetAttribute((const void*)k_rowwarp<CS_FILL_POLYLINES_SOFT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
            : hipFuncSetAttribute((const void*)k_rowwarp<CS_FILL_POLYLINES_SHARP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (fill == CS_FILL_POLYLINES_SOFT) hipLaunchKernelGGL((k_rowwarp<CS_FILL_POLYLINES_SOFT, true>), grid, block, lds, stream, A);
        else hipLaunchKernelGGL((k_rowwarp<CS_FILL_POLYLINES_SHARP, true>), grid, block, lds, stream, A);
        return hipGetLastError();
    }
    if ((A.d64 & 3) && fill == CS_FILL_HYBRID_EDGE_PLUS) {   // (its polylines half in the dialect instantiation)
        hipError_t e = hipFuncSetAttribute((const void*)k_rowwarp<CS_FILL_HYBRID_EDGE_PLUS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_rowwarp<CS_FILL_HYBRID_EDGE_PLUS, true>), grid, block, lds, stream, A);
        return hipGetLastError();
    }
    switch (fill) {
        CS_LAUNCH(CS_FILL_NONE)
        CS_LAUNCH(CS_FILL_NAIVE)
        CS_LAUNCH(CS_FILL_NAIVE_INTERPOLATING)
        CS_LAUNCH(CS_FILL_POLYLINES_SOFT)
        CS_LAUNCH(CS_FILL_POLYLINES_SHARP)
        CS_LAUNCH(CS_FILL_INVERSE)
        CS_LAUNCH(CS_FILL_NONE_POST)
        CS_LAUNCH(CS_FILL_INVERSE_POST)
        CS_LAUNCH(CS_FILL_HYBRID_EDGE_PLUS)
    default: return hipErrorInvalidValue;
    }
#undef CS_LAUNCH
    return hipGetLastError();
}

size_t rowwarp_lds_bytes(int fill, int w, int anaglyph) { return lds_common_bytes(fill, w, anaglyph) + lds_tech_bytes(fill, w) + CS_ROW_LDS_STATIC; }

// splat result (3 + 1 bytes per pixel and eye), then the gap lists of the streaming fill: a counter per (frame, eye, row) and
// 16-bit columns, w per row
static size_t hybrid_splat_bytes(int n, int h, int w) { return ((size_t)n * 2 * h * w * 4 + 255) & ~(size_t)255; }
static size_t hybrid_count_bytes(int n, int h) { return ((size_t)n * 2 * h * 4 + 255) & ~(size_t)255; }
size_t hybrid_workspace_bytes(int n, int h, int w) {
    return hybrid_splat_bytes(n, h, w) + hybrid_count_bytes(n, h) + (size_t)n * 2 * h * w * 2 + 256;
}
int hybrid_max_width() {
    int lo = 0, hi = 1 << 15;
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (rowwarp_lds_bytes(CS_FILL_HYBRID_EDGE, mid, 0) <= CS_LDS_BYTES) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// does launch_hybrid take the fused tile form (k_hybrid_splat_tile<true>) for a node-path call with these properties?  (the
// caller decides with it whether the depth blur may leave its edge-free tiles unwritten)
bool hybrid_fused_ok(int n, int w, int halo, int anaglyph, int single, int d64, int plus) {
    const int tmax = (HYT_NPT - 2 * (halo + 2) - 8) & ~3;
    (void)d64;   // (round 5: the tile kernel has a dialect instantiation)
    return halo >= 0 && tmax >= 128 && !dev_switch(CS_DEBUG_NO_TILE) && (size_t)n * 2 <= 65535 && !plus && !anaglyph && single < 0 &&
           w <= 65535 && !dev_switch(CS_DEBUG_HYBRID_UNFUSED);
}
int launch_hybrid(const RowArgs& A0, void* workspace, hipStream_t stream, int plus, int halo) {
    RowArgs A = A0;
    A.hyb_base = (uint8_t*)workspace;
    A.hyb_mask = A.hyb_base + (size_t)A.n * A.neyes * A.h * A.w * 3;
    size_t lds = rowwarp_lds_bytes(CS_FILL_HYBRID_EDGE, A.w, 0);
    if (A.d64 & 1) lds += align16(8 * (size_t)A.w);   // k_hybrid_splat: dest_x in float64
    if (lds > CS_LDS_BYTES) return CS_ELIMIT;
    int threads = A.w <= 256 ? 256 : (A.w <= 1024 ? 512 : 1024);
    hipError_t e = hipFuncSetAttribute((const void*)k_hybrid_splat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return CS_EHIP;
    // the tile form of the splat for the node path (float32 image) when a tile fits next to its halo
    const int tmax = (HYT_NPT - 2 * (halo + 2) - 8) & ~3;
    if (halo >= 0 && A.image_f32 && tmax >= 128 && !dev_switch(CS_DEBUG_NO_TILE) && (size_t)A.n * A.neyes <= 65535) {
        const int tiles = (A.w + tmax - 1) / tmax;
        const int T = ((A.w + tiles - 1) / tiles + 3) & ~3;
        // two-eye layouts: the splat writes the node outputs itself, only the gap pixels are left (CS_DEBUG_HYBRID_UNFUSED: the
        // separate streaming pass k_hybrid_out4)
        const bool fused = !A.out_u8 && A.neyes == 2 && A.depth_l && A.depth_r && A.mask &&
                           hybrid_fused_ok(A.n, A.w, halo, A.anaglyph, A.single, A.d64, plus);
        if (A.tilemap && !fused) return CS_EINVAL;   // (lazy blur tiles are only readable by the fused form)
        const dim3 grid(((A.w + T - 1) / T) * 8, A.neyes == 2 ? eye_group_grid_y(A.h) : (A.h + 7) / 8, A.n);
        if (fused) {
            uint32_t* cnt = (uint32_t*)((char*)workspace + hybrid_splat_bytes(A.n, A.h, A.w));
            uint16_t* lst = (uint16_t*)((char*)cnt + hybrid_count_bytes(A.n, A.h));
            e = hipMemsetAsync(cnt, 0, (size_t)A.n * A.neyes * A.h * 4, stream);
            if (e != hipSuccess) return CS_EHIP;
#ifdef HYB_WTAB
            hipLaunchKernelGGL(k_hyb_wtab_init, dim3(128), dim3(256), 0, stream);   // (experiment: every call; a product would do it once)
#endif
            if (A.d64) hipLaunchKernelGGL((k_hybrid_splat_tile<true, true>), grid, dim3(HYT_NT), 0, stream, A, halo, T, cnt, lst);
            else hipLaunchKernelGGL(k_hybrid_splat_tile<true>, grid, dim3(HYT_NT), 0, stream, A, halo, T, cnt, lst);
            hipLaunchKernelGGL(k_hybrid_gaps<true>, dim3(A.h, A.neyes, A.n), dim3(64), 0, stream, A, (const uint32_t*)cnt, (const uint16_t*)lst);
            return hipGetLastError() == hipSuccess ? CS_OK : CS_EHIP;
        }
        if (A.d64) hipLaunchKernelGGL((k_hybrid_splat_tile<false, true>), grid, dim3(HYT_NT), 0, stream, A, halo, T, (uint32_t*)nullptr, (uint16_t*)nullptr);
        else hipLaunchKernelGGL(k_hybrid_splat_tile<false>, grid, dim3(HYT_NT), 0, stream, A, halo, T, (uint32_t*)nullptr, (uint16_t*)nullptr);
    } else
        hipLaunchKernelGGL(k_hybrid_splat, dim3(A.h, A.n, A.neyes), dim3(threads), lds, stream, A);
    if (plus) e = launch_rowwarp(CS_FILL_HYBRID_EDGE_PLUS, A, threads, stream);
    else if (!A.anaglyph && A.w % 4 == 0 && A.w <= 65535 && A.h <= 65535 && A.n <= 65535 && !dev_switch(CS_DEBUG_NO_TILE)) {
        // streaming form: every pixel's splat result by 16-byte accesses, then the listed untouched pixels
        uint32_t* cnt = (uint32_t*)((char*)workspace + hybrid_splat_bytes(A.n, A.h, A.w));
        uint16_t* lst = (uint16_t*)((char*)cnt + hybrid_count_bytes(A.n, A.h));
        e = hipMemsetAsync(cnt, 0, (size_t)A.n * A.neyes * A.h * 4, stream);
        if (e != hipSuccess) return CS_EHIP;
        hipLaunchKernelGGL(k_hybrid_out4, dim3((A.w + 1023) / 1024, A.h, A.n), dim3(256), 0, stream, A, cnt, lst);
        hipLaunchKernelGGL(k_hybrid_gaps<false>, dim3(A.h, A.neyes, A.n), dim3(64), 0, stream, A, (const uint32_t*)cnt, (const uint16_t*)lst);
        e = hipGetLastError();
    } else {
        hipLaunchKernelGGL(k_hybrid_fill, dim3((A.w + 255) / 256, A.h, A.n), dim3(256), 0, stream, A);
        e = hipGetLastError();
    }
    return e == hipSuccess ? CS_OK : CS_EHIP;
}

}  // namespace cs
