// cs_fwdtile.hip -- fill techniques 'none' (reference stereoimage_generation.py:1850-1867, the forward map of
// apply_stereo_divergence_naive), 'naive' (:1893-1908, holes take the nearest filled pixel, right before left),
// 'naive_interpolating' (:1871-1892, linear ramps over the holes) and 'inverse' (:1715-1737, the z-buffered two-column
// splat) as a halo-tile kernel: the node path's float32 image in, both
// eyes of a tile out.
//
// The general row kernel (cs_rowwarp.hip) keeps a whole row of one frame in LDS (60 KB at 4K: two workgroups per CU,
// five barriers, every pixel through the full powf clone).  The forward map only moves a pixel by int(offset) columns,
// |offset| <= S = |div_px| max(c, 1-c)^e + |sep_px| (normalised depth in [0, 1]), so an output tile [o0, o0 + T) only
// needs the source columns [o0 - S, o0 + T + S): a 256-thread workgroup stages them three per lane, converts the
// colours once for BOTH eyes (packed uint8 codes in LDS), and per eye
//   * every source pixel proposes itself for its destination column: "later write wins" of the reference's sweep ==
//     the highest (divergence < 0) or lowest source column per destination -> LDS atomicMax / atomicMin ('none'); the
//     z-buffer with strict '>' and ascending x == the maximum over (closeness, -x) of the two columns a pixel lands on
//     -> 64-bit LDS atomicMax ('inverse'),
//   * every output column looks up its winner's colour (0 = hole) and keeps it in a register; 'naive': the forward map is
//     computed for R more columns on either side of the tile, the filled flags become bit rows (wave ballots), and a hole
//     finds its nearest filled neighbours with clz / ctz.  A hole that the window cannot decide (nothing filled within
//     its reach although the search limit |int(div_px)| + 1 goes further) flags the ROW for the row kernel;
//     'naive_interpolating': the row kernel's formulation (cs_rowwarp.hip: every pixel of an interval between two "good"
//     pixels finds the interval by walking the flags and computes its own ramp value; intervals that hit the re-trigger
//     quirk are replayed literally by one lane) on the window -- intervals that leave the window flag the row,
// then the eyes are written straight into their slots of the SBS / top-bottom layout or composed into the anaglyph
// (:1996-2010), together with the no-fill mask (GenerateStereo.py:355-361) and both depth-map outputs (:1511-1516).
// Dialect D32 only (other dialects, uint8 images and anaglyph-free single calls of apply_stereo_divergence take the row
// kernel).  Arithmetic as in cs_rowwarp.hip: (d - min) / range - conv, sign * powf(|nd|, e) * div + sep in float32 with
// the libm-exact powf (exponents 2 and 1 through the exact shortcuts of cs_math.h), int() truncation.
#include "cs_common.h"
#include "cs_kernels.h"

namespace cs {

__constant__ csm::PowfTables c_fw_powf_tables = CS_POWF_TABLES_INIT;

struct FwdTileArgs {
    int n, h, w, S, T;
    int R;                // 'naive': columns of the forward map computed beyond either end of the tile
    uint8_t* rowflag;     // 'naive': rows handed to the row kernel
    const float* image;
    const uint32_t* stats;
    int scale_from_stats;
    float e32, conv32;
    EyeArgs eye[2];
    int single, anaglyph;
    float* stereo; int stereo_is_u8;
    float* mask; float* depth_l; float* depth_r;
    int out_h, out_w;
    int dbg;
    const uint32_t* tilemap; const float* gray; int tm_words;   // lazy depth-blur tiles (cs_common.h) or null
    int d64; double e64;  // dialect bits (RowArgs::d64: 1 = float64 disparity chain, 2 = int64 pixel sums) and the exponent as a double
    // second tier (round 5): only the rows of this list (frame * h + row, *row_count of them), persistent workgroups -- the rows a first
    // launch flagged, with a wider window.  Null: every row, blockIdx = (tile x 8 rows, rows / 8, frame)
    const uint32_t* row_list; const uint32_t* row_count;
};

// `sign * (abs(d) ** e) * div` of dialect D64 (numba typing; cs_rowwarp.hip disparity64): float64 throughout.
// pow(x, 2.0) == x * x and pow(x, 1.0) == x exactly (the product of two float32 values is exact in float64; libm returns the
// correctly rounded -- here: exact -- result for both), any other exponent takes the device library's pow (<= 1 ulp from libm's:
// only int() / floor() of the offset plus a pixel coordinate is used).
__device__ __forceinline__ double fw_disparity64(float d, double e64, double div64) {
    const double s = d >= 0.0f ? 1.0 : -1.0, ax = (double)fabsf(d);
    const double p = e64 == 2.0 ? ax * ax : (e64 == 1.0 ? ax : pow(ax, e64));
    return (s * p) * div64;
}

struct FwF3 { float x, y, z; };
struct FwB3 { uint8_t x, y, z; };

// DIA (round 5): the instantiation that runs the dialect bits A.d64 -- D64 at tile speed; the D32 kernel keeps its registers
// LISTED (round 5): the second-tier instantiation -- persistent workgroups over a list of rows.  A template parameter, not a runtime
// test: with the loop around its body the one-pass kernel went from 44-63 to 59-137 registers and spilled 55-217 scalar ones (cfg 5
// 5.3 -> 6.6 ms per 64 frames, tools/sessions/r05_s36.sh)
template <int NT, int SLOTS, int FILL, bool DIA = false, bool LISTED = false>
__global__ void __launch_bounds__(NT) k_fwdtile(FwdTileArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    // grid = (tiles x 8 rows, rows / 8, frames): all tiles of a row on one XCD (see cs_polypoint.hip)
    // listed form (second tier): grid = (tiles, G): workgroup (t, g) does tile t of list entries g, g + G, ...
    constexpr bool listed = LISTED;
    const uint32_t lcount = listed ? A.row_count[0] : 0u;
    for (uint32_t it = 0; listed || it == 0; it++) {
    int row, tile, frame;
    if (listed) {
        const uint32_t li = blockIdx.y + it * gridDim.y;
        if (li >= lcount) break;
        const uint32_t e = A.row_list[li] & 0x3fffffffu;
        frame = (int)(e / (uint32_t)A.h); row = (int)(e - (uint32_t)frame * (uint32_t)A.h); tile = blockIdx.x;
        if (it) __syncthreads();   // (the previous row's LDS)
    } else {
        const int xi = blockIdx.x;
        row = blockIdx.y * 8 + (xi & 7);
        if (row >= A.h) return;
        tile = xi >> 3; frame = blockIdx.z;
    }
    const int w = A.w, h = A.h, T = A.T;
    const int o0 = tile * T, wt = min(T, w - o0);
    const int c0 = max(0, o0 - A.R), c1 = min(w, o0 + wt + A.R), nwin = c1 - c0;   // columns of the forward map (the tile + R)
    const int s0 = max(0, c0 - A.S), s1 = min(w, c1 + A.S), ns = s1 - s0;
    constexpr int NPT = NT * SLOTS;
    uint32_t* img = (uint32_t*)smem;        // [NPT] colour codes r | g << 8 | b << 16 of source column s0 + j
    int* winner = (int*)(img + NPT);        // [nwin] winning source column (local index) per column of the window ('none', 'naive')
    unsigned long long* key = (unsigned long long*)(img + NPT);   // [T] closeness << 32 | ~source index ('inverse')
    const unsigned long long key_init = ((unsigned long long)csm::f2ord(-1.0f) << 32) | 0xffffffffull;
    unsigned long long* fm = key + NPT;     // [NPT / 64] filled flags of the window as bit rows ('naive')
    // 'naive_interpolating': colours of the window (modified in place like the reference's derived_image), ramp values, the
    // interval start of a pixel, per-pixel flags (1 filled, 2 good, 4 has a ramp value, 8 interval needs the replay, 16 unresolved)
    uint32_t* colw = (uint32_t*)(fm + NPT / 64);
    uint32_t* tmpv = colw + NPT;
    uint16_t* istart = (uint16_t*)(tmpv + NPT);
    uint8_t* fl = (uint8_t*)(istart + NPT);
    bool giveup = false;
    // sum() of a pixel: uint8, wraps mod 256 in D32 (quirk Q5); int64 under numba (dialect bit 1)
    const unsigned smask = (DIA && (A.d64 & 2)) ? 0xffffu : 0xffu;
    auto sum8 = [&](uint32_t c) { return ((c & 0xffu) + ((c >> 8) & 0xffu) + ((c >> 16) & 0xffu)) & smask; };
    const bool f64chain = DIA && (A.d64 & 1);

    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    const uint32_t rowpix = ((uint32_t)frame * (uint32_t)h + (uint32_t)row) * (uint32_t)w;
    const char* const irow = reinterpret_cast<const char*>(reinterpret_cast<const FwF3*>(A.image) + rowpix + s0);
    const char* const drow0 = reinterpret_cast<const char*>(A.eye[0].depth + rowpix + s0);
    const char* const drow1 = reinterpret_cast<const char*>(A.eye[1].depth + rowpix + s0);
    FwF3 cpre[SLOTS];
    float dpre[2][SLOTS];
    if (A.tilemap) {   // lazy depth-blur tiles: edge-free tiles come from the gray depth
        const char* const grow = reinterpret_cast<const char*>(A.gray + rowpix + s0);
        const uint32_t s255 = st[ST_SCALE255];
        const LazySel Z0 = lazy_select(A.tilemap, A.tm_words, frame, h, row, s0, drow0, grow, s255);
        const LazySel Z1 = lazy_select(A.tilemap, A.tm_words, frame, h, row, s0, drow1, grow, s255);
        float mul[2][SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const uint32_t jc = (uint32_t)min(tid + k * NT, ns - 1);
            cpre[k] = *reinterpret_cast<const FwF3*>(irow + 12u * jc);
        }
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const uint32_t jc = (uint32_t)min(tid + k * NT, ns - 1);
            dpre[0][k] = lazy_load(Z0, (uint32_t)s0 + jc, jc, mul[0][k]);
            dpre[1][k] = lazy_load(Z1, (uint32_t)s0 + jc, jc, mul[1][k]);
        }
#pragma unroll
        for (int k = 0; k < SLOTS; k++) { dpre[0][k] *= mul[0][k]; dpre[1][k] *= mul[1][k]; }
    } else {
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const uint32_t jc = (uint32_t)min(tid + k * NT, ns - 1);
            cpre[k] = *reinterpret_cast<const FwF3*>(irow + 12u * jc);
            dpre[0][k] = *reinterpret_cast<const float*>(drow0 + 4u * jc);
            dpre[1][k] = *reinterpret_cast<const float*>(drow1 + 4u * jc);
        }
    }
    const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
    // colours: np.clip(x * 255, 0, 255).astype(uint8) (reference :1508), once for both eyes
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
        const uint32_t r = (uint32_t)(int)__builtin_amdgcn_fmed3f(cpre[k].x * 255.0f, 0.0f, 255.0f);
        const uint32_t g = (uint32_t)(int)__builtin_amdgcn_fmed3f(cpre[k].y * 255.0f, 0.0f, 255.0f);
        const uint32_t b = (uint32_t)(int)__builtin_amdgcn_fmed3f(cpre[k].z * 255.0f, 0.0f, 255.0f);
        img[tid + k * NT] = r | g << 8 | b << 16;
    }
    constexpr int OUTS = SLOTS;   // output columns per lane: T <= NT * SLOTS
    uint32_t res[2][OUTS];
    const int pow_mode = A.dbg == 17 ? 0 : (A.e32 == 2.0f ? 2 : (A.e32 == 1.0f ? 1 : 0));
#pragma unroll
    for (int e = 0; e < 2; e++) {
#pragma unroll
        for (int m = 0; m < OUTS; m++) res[e][m] = 0;
        if (A.single >= 0 && A.single != e) continue;
        const EyeArgs& E = A.eye[e];
        if (!E.enabled) {   // divergence < 0.001: the eye is the source image (quirk Q10)
            __syncthreads();
#pragma unroll
            for (int m = 0; m < OUTS; m++) {
                const int q = tid + m * NT;
                if (q < wt) res[e][m] = img[q + o0 - s0];
            }
            continue;
        }
        const int init = E.asc ? -1 : 0x7fffffff;
        for (int p = tid; p < nwin; p += NT) {
            if (FILL == CS_FILL_INVERSE) key[p] = key_init;
            else winner[p] = init;
        }
        __syncthreads();   // (also: img complete)
        const float dmin = csm::ord2f(st[E.st_min]), dmax = csm::ord2f(st[E.st_max]);
        const bool flat = dmax == dmin;
        const float range = dmax - dmin;
        float nd[SLOTS], pw[SLOTS];
        unsigned risk = 0;
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const float d = dpre[e][k] * scale;
            nd[k] = flat ? 0.0f - A.conv32 : ((d - dmin) / range) - A.conv32;   // (:1587-1600)
            const float ax = fabsf(nd[k]);
            bool r = pow_mode == 0;
            pw[k] = pow_mode == 1 ? ax : (pow_mode == 2 ? csm::square_or_flag(ax, r) : 0.0f);
            risk |= r ? 1u << k : 0u;
        }
        if (__any(risk != 0u)) {   // the full powf clone for the risky squares / for every point at other exponents
            asm volatile("" ::: "memory");
            do {
                float xin = 1.0f;
                int sel = -1;
#pragma unroll
                for (int k = SLOTS - 1; k >= 0; k--) if (risk & (1u << k)) { xin = fabsf(nd[k]); sel = k; }
                const float r = csm::powf_exact_simt(xin, A.e32, &c_fw_powf_tables);
#pragma unroll
                for (int k = 0; k < SLOTS; k++) if (sel == k) pw[k] = r;
                risk &= risk - 1u;
            } while (__any(risk != 0u));
        }
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int j = tid + k * NT;
            const float sg = nd[k] >= 0.0f ? 1.0f : -1.0f;
            if (FILL == CS_FILL_INVERSE) {
                const float off = (sg * pw[k]) * E.div32;
                const float dest = ((float)(s0 + j) + 0.5f + off) + E.sep32;   // (:1725)
                float fl = floorf(dest);
                if (f64chain) fl = (float)floor((((double)(s0 + j) + 0.5) + fw_disparity64(nd[k], A.e64, E.div64)) + E.sep64);   // (range test and int() only)
                if (j < ns && fl >= -2.0f && fl <= (float)w) {
                    const int q = (int)fl - o0;
                    const unsigned long long kk = ((unsigned long long)csm::f2ord(nd[k]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)j);
                    if (q >= 0 && q < wt) atomicMax(&key[q], kk);
                    if (q + 1 >= 0 && q + 1 < wt) atomicMax(&key[q + 1], kk);
                }
            } else {
                const float off = ((sg * pw[k]) * E.div32) + E.sep32;                  // (:1865)
                // int(): truncation toward zero; keep the conversion defined for absurd offsets
                int io = off >= 2147483520.0f ? 0x7fffff00 : (off <= -2147483520.0f ? -0x7fffff00 : (int)off);
                if (f64chain) {
                    const double o64 = fw_disparity64(nd[k], A.e64, E.div64) + E.sep64;
                    io = o64 >= 2147483520.0 ? 0x7fffff00 : (o64 <= -2147483520.0 ? -0x7fffff00 : (int)o64);
                }
                const long long q = (long long)(s0 + j - c0) + io;
                if (j < ns && q >= 0 && q < nwin) {
                    if (E.asc) atomicMax(&winner[(int)q], j);
                    else atomicMin(&winner[(int)q], j);
                }
            }
        }
        __syncthreads();
        if (FILL == CS_FILL_NAIVE_INTERPOLATING) {
            // (round 5) The filled / good flags of the window also exist as BIT ROWS (one wave ballot per 64 columns, in the idle upper
            // half of `key`): a pixel finds the good pixels around it and the first unfilled pixel of its interval with clz / ctz on
            // one or two words instead of walking the flag bytes (up to 2 S dependent LDS reads per hole pixel, the wave waiting
            // for its longest walker: SQ_WAIT_ANY was 65 % of the wave-cycles, profiles/r05a_naive_interp), and the intervals that
            // hit the re-trigger quirk are replayed by a whole WAVE each -- the triggers one after the other as in the reference,
            // every search and every ramp in parallel over the wave's lanes -- instead of literally by one lane.
            unsigned long long* const fbits = key + NPT / 2;      // [NPT / 64] filled
            unsigned long long* const gbits = fbits + NPT / 64;   // [NPT / 64] good = filled and a colour sum != 0
            unsigned long long* const qbits = gbits + NPT / 64;   // [NPT / 64] interval starting here needs the replay
            const int nw = (nwin + 63) >> 6;
            for (int pb = (tid >> 6) * 64; pb < nwin; pb += NT) {
                const int p = pb + (tid & 63);
                const bool in = p < nwin;
                const int s = in ? winner[p] : init;
                const bool f = in && s != init;
                const uint32_t c = f ? img[s] : 0u;
                const bool good = f && sum8(c) != 0u;
                if (in) { colw[p] = c; fl[p] = (uint8_t)((f ? 1 : 0) | (good ? 2 : 0)); }
                const unsigned long long fb = __ballot(f), gb = __ballot(good);
                if ((tid & 63) == 0) { fbits[pb >> 6] = fb; gbits[pb >> 6] = gb; qbits[pb >> 6] = 0ull; }
            }
            __syncthreads();
            // largest set bit below p (-1: none) / smallest set bit above p (nwin: none) / smallest CLEAR bit of `fbits` at or above p
            auto prev_set = [&](const unsigned long long* bits, int p) {
                if (p <= 0) return -1;
                int wi = (p - 1) >> 6;
                unsigned long long cur = bits[wi] & (~0ull >> (63 - ((p - 1) & 63)));
                while (true) {
                    if (cur) return wi * 64 + 63 - __clzll((long long)cur);
                    if (--wi < 0) return -1;
                    cur = bits[wi];
                }
            };
            auto next_set = [&](const unsigned long long* bits, int p) {
                int wi = (p + 1) >> 6;
                if (wi >= nw) return nwin;
                unsigned long long cur = bits[wi] & (~0ull << ((p + 1) & 63));
                while (true) {
                    if (cur) return min(wi * 64 + __ffsll((long long)cur) - 1, nwin);
                    if (++wi >= nw) return nwin;
                    cur = bits[wi];
                }
            };
            auto next_clear = [&](const unsigned long long* bits, int p) {   // (bits beyond nwin are clear: the result is clamped)
                int wi = p >> 6;
                if (wi >= nw) return nwin;
                unsigned long long cur = ~bits[wi] & (~0ull << (p & 63));
                while (true) {
                    if (cur) return min(wi * 64 + __ffsll((long long)cur) - 1, nwin);
                    if (++wi >= nw) return nwin;
                    cur = ~bits[wi];
                }
            };
            auto flag_or = [&](int p, unsigned bit) { atomicOr((unsigned*)fl + (p >> 2), bit << ((p & 3) * 8)); };
            auto ramp = [&](uint32_t lb, uint32_t rb, float total, float k) {   // l_border + (step * k).astype(uint8), per channel
                uint32_t v = 0;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const float a = (float)((lb >> (8 * ch)) & 0xffu), b = (float)((rb >> (8 * ch)) & 0xffu);
                    v |= (uint32_t)(uint8_t)(((lb >> (8 * ch)) & 0xffu) + csm::f32_to_u8_wrap(((b - a) / total) * k)) << (8 * ch);
                }
                return v;
            };
            for (int p = tid; p < nwin; p += NT) {
                if (fl[p] & 2) continue;
                const int lg = prev_set(gbits, p), g = next_set(gbits, p);
                if ((lg < 0 && c0 > 0) || (g >= nwin && c1 < w)) { flag_or(p, 16u); continue; }   // the interval leaves the window
                const int s0i = lg + 1;
                istart[p] = (uint16_t)s0i;
                const int l0 = next_clear(fbits, s0i);   // the first unfilled pixel of the interval: the first trigger
                if (l0 >= g || p < l0) continue;         // no unfilled pixel in the interval up to here: untouched
                uint32_t lb = l0 > 0 ? colw[l0 - 1] : 0u, rb = g < nwin ? colw[g] : 0u;   // (l0 == 0 only at the frame's first column)
                if (sum8(lb) == 0u) lb = rb;
                else if (sum8(rb) == 0u) rb = lb;
                const uint32_t v = ramp(lb, rb, (float)(1 + g - l0), (float)(p - l0 + 1));
                tmpv[p] = v;
                flag_or(p, 4u);
                // the quirk: an unfilled pixel whose ramp value sums to 0 triggers again -> the interval is replayed trigger by trigger
                if (p > l0 && !(fl[p] & 1) && sum8(v) == 0u) atomicOr(&qbits[s0i >> 6], 1ull << (s0i & 63));
            }
            __syncthreads();
            for (int p = tid; p < nwin; p += NT)
                if ((fl[p] & 4) && !((qbits[istart[p] >> 6] >> (istart[p] & 63)) & 1ull)) colw[p] = tmpv[p];
            __syncthreads();
            // ---- replay of the flagged intervals (reference :1873-1891 on [s, g)), one WAVE per interval, round-robin
            {
                const int lane = tid & 63, wave = tid >> 6;
                int turn = 0;
                for (int wi = 0; wi < nw; wi++) {
                    unsigned long long m = qbits[wi];
                    m = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)m);
                    while (m) {
                        const int s = wi * 64 + __ffsll((long long)m) - 1;
                        m &= m - 1;
                        if ((turn++ & (NT / 64 - 1)) != wave) continue;
                        const int g = next_set(gbits, s - 1);   // the good pixel that ends the interval (nwin: the frame's end)
                        int cur = s;
                        while (true) {
                            // the next trigger: the first unfilled pixel at or after `cur` whose CURRENT colour sums to 0
                            int l = -1;
                            for (int base = cur; base < g; base += 64) {
                                const int idx = base + lane;
                                const bool t = idx < g && !((fbits[idx >> 6] >> (idx & 63)) & 1ull) && sum8(colw[idx]) == 0u;
                                const unsigned long long b = __ballot(t);
                                if (b) { l = base + __ffsll((long long)b) - 1; break; }
                            }
                            if (l < 0) break;
                            // its right border: the first filled pixel after it whose current colour does not sum to 0 (ramps written
                            // over filled black pixels count: they are read as they are NOW)
                            int r = nwin;
                            for (int base = l + 1; base <= g && base < nwin; base += 64) {
                                const int idx = base + lane;
                                const bool t = idx <= g && idx < nwin && ((fbits[idx >> 6] >> (idx & 63)) & 1ull) && sum8(colw[idx]) != 0u;
                                const unsigned long long b = __ballot(t);
                                if (b) { r = base + __ffsll((long long)b) - 1; break; }
                            }
                            uint32_t lb = l > 0 ? colw[l - 1] : 0u, rb = r < nwin ? colw[r] : 0u;
                            if (sum8(lb) == 0u) lb = rb;
                            else if (sum8(rb) == 0u) rb = lb;
                            const float total = (float)(1 + r - l);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's reads of colw are done before its lanes overwrite it)
                            for (int c = l + lane; c < r; c += 64) colw[c] = ramp(lb, rb, total, (float)(c - l + 1));
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (LDS executes a wave's instructions in order: the next search reads these values)
                            cur = l + 1;
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (FILL == CS_FILL_NAIVE) {
            // filled flags of the window as bit rows: one ballot per 64 columns
            for (int pb = (tid >> 6) * 64; pb < nwin; pb += NT) {
                const int p = pb + (tid & 63);
                const unsigned long long bits = __ballot(p < nwin && winner[p] != init);
                if ((tid & 63) == 0) fm[pb >> 6] = bits;
            }
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < OUTS; m++) {
            const int q = tid + m * NT;
            if (q < wt) {
                if (FILL == CS_FILL_INVERSE) {
                    const unsigned long long kk = key[q];
                    res[e][m] = kk > key_init ? img[0xffffffffu - (unsigned)(kk & 0xffffffffull)] : 0u;
                } else {
                    const int p = q + (o0 - c0);
                    if (FILL == CS_FILL_NAIVE_INTERPOLATING) {
                        if (fl[p] & 16) giveup = true;
                        res[e][m] = colw[p];
                        continue;
                    }
                    int s = winner[p];
                    if (FILL == CS_FILL_NAIVE && s == init) {
                        // nearest filled column to the right (dr) and to the left (dl) inside the window, BIG: none there
                        const int BIG = 1 << 29, nw = (nwin + 63) >> 6;
                        int dr = BIG, dl = BIG;
                        {
                            int wi = (p + 1) >> 6;
                            unsigned long long cur = wi < nw ? fm[wi] & (~0ull << ((p + 1) & 63)) : 0ull;
                            while (wi < nw) {
                                if (cur) { dr = wi * 64 + __ffsll((long long)cur) - 1 - p; break; }
                                if (++wi < nw) cur = fm[wi];
                            }
                        }
                        if (p > 0) {
                            int wi = (p - 1) >> 6;
                            unsigned long long cur = fm[wi] & (~0ull >> (63 - ((p - 1) & 63)));
                            while (true) {
                                if (cur) { dl = p - (wi * 64 + 63 - __clzll((long long)cur)); break; }
                                if (--wi < 0) break;
                                cur = fm[wi];
                            }
                        }
                        // (:1899-1907) for o = 1 .. lim - 1: the right neighbour at distance o, else the left one.  What the window
                        // did not examine -- beyond its ends, unless they are the frame's -- must not be able to change the answer.
                        const int lim1 = E.naive_lim - 1;
                        const int cov_r = c1 == w ? BIG : nwin - 1 - p, cov_l = c0 == 0 ? BIG : p;
                        const int a = dr <= lim1 ? dr : BIG, b = dl <= lim1 ? dl : BIG;
                        if ((dr == BIG && cov_r < min(b, lim1)) || (dl == BIG && cov_l < min(a - 1, lim1))) giveup = true;
                        int sp = -1;
                        if (a <= b) { if (a != BIG) sp = p + a; }
                        else sp = p - b;
                        s = sp >= 0 ? winner[sp] : init;
                    }
                    res[e][m] = s != init ? img[s] : 0u;
                }
            }
        }
        __syncthreads();   // (winner is re-initialised for the next eye)
    }
    if ((FILL == CS_FILL_NAIVE || FILL == CS_FILL_NAIVE_INTERPOLATING) && giveup) A.rowflag[(uint32_t)frame * (uint32_t)h + (uint32_t)row] = 1;   // the row kernel redoes it
    // ---- outputs: eyes into their slots / the anaglyph composition, the no-fill mask, both depth-map outputs
    auto store = [&](int e, int q, uint32_t c) {
        const EyeArgs& E = A.eye[e];
        const uint32_t o = ((uint32_t)frame * (uint32_t)A.out_h + (uint32_t)(row + E.yoff)) * (uint32_t)A.out_w + (uint32_t)(E.xoff + o0 + q);
        const uint32_t r = c & 0xffu, g = (c >> 8) & 0xffu, b = (c >> 16) & 0xffu;
        if (A.stereo_is_u8) *reinterpret_cast<FwB3*>(reinterpret_cast<uint8_t*>(A.stereo) + (size_t)o * 3) = FwB3{(uint8_t)r, (uint8_t)g, (uint8_t)b};
        else *reinterpret_cast<FwF3*>(A.stereo + (size_t)o * 3) = FwF3{csm::code_over_255((float)r), csm::code_over_255((float)g), csm::code_over_255((float)b)};
        A.mask[o] = (c & 0xffffffu) == 0u ? 1.0f : 0.0f;   // sum of the channels == 0 (GenerateStereo.py:355-361)
    };
#pragma unroll
    for (int m = 0; m < OUTS; m++) {
        const int q = tid + m * NT;
        if (q >= wt) continue;
        if (A.anaglyph == 1) store(1, q, (res[0][m] & 0xffu) | (res[1][m] & 0xffff00u));        // R from eye 0, GB from eye 1
        else if (A.anaglyph == 2) store(1, q, (res[1][m] & 0xffu) | (res[0][m] & 0xffff00u));
        else {
            if (A.single < 0 || A.single == 0) store(0, q, res[0][m]);
            if (A.single < 0 || A.single == 1) store(1, q, res[1][m]);
        }
    }
    // depth-map outputs: (depth * 255).astype(uint8) wraps mod 256 (quirk Q7), value code / 255 on three channels
    const int qoff = s0 - o0;
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
        const int q = tid + k * NT + qoff;   // tile pixel of this source column
        if ((unsigned)q < (unsigned)wt) {
            const size_t p = (size_t)(rowpix + (uint32_t)(o0 + q)) * 3;
            const float v0 = csm::code_over_255((float)csm::f32_to_u8_wrap((dpre[0][k] * scale) * 255.0f));
            const float v1 = csm::code_over_255((float)csm::f32_to_u8_wrap((dpre[1][k] * scale) * 255.0f));
            *reinterpret_cast<FwF3*>(A.depth_l + p) = FwF3{v0, v0, v0};
            *reinterpret_cast<FwF3*>(A.depth_r + p) = FwF3{v1, v1, v1};
        }
    }
    }   // (rows of the list)
}

// Largest halo the tile kernel takes (beyond it: the row kernel)
int fwdtile_max_halo() { return (256 * 3 - 64) / 2; }

// `none` / `naive` / `inverse` through the tile kernel (`naive`: rows it flags in `rowflag` must be redone by the row kernel).  Returns hipErrorNotSupported when the call is not one of its cases (the caller then
// launches the row kernel).
hipError_t launch_fwdtile(int fill, const RowArgs& R, int S0, uint8_t* rowflag, hipStream_t stream, const uint32_t* tier2_list,
                          const uint32_t* tier2_count) {
    // tier2_list (round 5, naive_interpolating): the rows a first launch flagged, once more with a window of 2 halo + 16 on either side (every
    // hole fits: holes are at most 2 halo wide) -- persistent workgroups over the list; what it flags in `rowflag` goes to the row kernel
    const bool tier2 = tier2_list != nullptr;
    if (tier2 && fill != CS_FILL_NAIVE_INTERPOLATING) return hipErrorNotSupported;
    // source slots per lane: 3 for the pure streaming fills (none / inverse: HBM-bound, wider tiles only cost occupancy: -2..4 %),
    // 4 for naive / naive_interpolating, whose tiles carry a search window on top of the halo (+13 % / +10 %)
    constexpr int NT = 256;
    const int SLOTS = (fill == CS_FILL_NAIVE || fill == CS_FILL_NAIVE_INTERPOLATING) ? 4 : 3;
    if (fill != CS_FILL_NONE && fill != CS_FILL_INVERSE && fill != CS_FILL_NAIVE && fill != CS_FILL_NAIVE_INTERPOLATING) return hipErrorNotSupported;
    if (!R.image_f32 || R.out_u8 || R.neyes != 2 || !R.depth_l || !R.depth_r || (R.row_list && !tier2)) return hipErrorNotSupported;
    const int S = S0 + (fill == CS_FILL_INVERSE ? 2 : 0);   // (the splat also touches the column right of floor(dest))
    if (S > fwdtile_max_halo()) return hipErrorNotSupported;
    if ((size_t)R.n * R.h * R.w >= (1ull << 31) || (size_t)R.n * R.out_h * R.out_w >= (1ull << 31) || R.n > 65535) return hipErrorNotSupported;
    FwdTileArgs A;
    A.n = R.n; A.h = R.h; A.w = R.w; A.S = S;
    // 'naive': the source of column c lands within S of it, so a hole's nearest filled pixel is within S + 1 -- except next
    // to the frame borders, where the kernel flags what it cannot decide
    A.R = 0; A.rowflag = rowflag;
    if (fill == CS_FILL_NAIVE) {
        if (!rowflag) return hipErrorNotSupported;
        const int lim1 = max(R.eye[0].naive_lim, R.eye[1].naive_lim) - 1;
        A.R = min(S + 1, lim1 > 0 ? lim1 : 0);
    }
    // 'naive_interpolating': an interval runs from the good pixel before a hole to the good pixel after it; holes are at most
    // 2S wide but usually far narrower -- a window of S + 8 on either side, the rest (and intervals lengthened by black
    // pixels) goes to the row kernel
    if (fill == CS_FILL_NAIVE_INTERPOLATING) {
        if (!rowflag) return hipErrorNotSupported;
        A.R = tier2 ? 2 * S + 16 : S + 8;
    }
    int tmax = (NT * SLOTS - 2 * S - 2 * A.R) & ~3;
    if (tmax < 128) return hipErrorNotSupported;
    const int tiles = (R.w + tmax - 1) / tmax;
    A.T = ((R.w + tiles - 1) / tiles + 3) & ~3;
    if (A.T + 2 * S + 2 * A.R > NT * SLOTS) return hipErrorNotSupported;
    A.image = R.image_f32; A.stats = R.stats; A.scale_from_stats = R.scale_from_stats;
    A.e32 = R.e32; A.conv32 = R.conv32;
    A.eye[0] = R.eye[0]; A.eye[1] = R.eye[1];
    A.single = R.single; A.anaglyph = R.anaglyph;
    A.stereo = R.stereo; A.stereo_is_u8 = R.stereo_is_u8; A.mask = R.mask; A.depth_l = R.depth_l; A.depth_r = R.depth_r;
    A.out_h = R.out_h; A.out_w = R.out_w;
    A.dbg = R.dbg;
    A.tilemap = R.tilemap; A.gray = R.lazy_gray; A.tm_words = R.tm_words;
    A.d64 = R.d64; A.e64 = R.e64;
    A.row_list = tier2_list; A.row_count = tier2_count;
    const int ntiles = (R.w + A.T - 1) / A.T;
    const size_t npt = (size_t)NT * SLOTS;
    const size_t lds = npt * 4 + npt * 8 + (npt / 64) * 8 + (fill == CS_FILL_NAIVE_INTERPOLATING ? npt * (4 + 4 + 2 + 1) : 0) + 64;
    const dim3 grid = tier2 ? dim3(ntiles, (unsigned)((1536 + ntiles - 1) / ntiles)) : dim3(ntiles * 8, (R.h + 7) / 8, R.n);
    const dim3 block(NT);
    if (tier2) {
        if (R.d64) hipLaunchKernelGGL((k_fwdtile<NT, 4, CS_FILL_NAIVE_INTERPOLATING, true, true>), grid, block, lds, stream, A);
        else hipLaunchKernelGGL((k_fwdtile<NT, 4, CS_FILL_NAIVE_INTERPOLATING, false, true>), grid, block, lds, stream, A);
        return hipGetLastError();
    }
    if (R.d64) {   // dialect D64 (either bit): the same kernels with the float64 offset chain / unwrapped pixel sums compiled in
        if (fill == CS_FILL_INVERSE) hipLaunchKernelGGL((k_fwdtile<NT, 3, CS_FILL_INVERSE, true>), grid, block, lds, stream, A);
        else if (fill == CS_FILL_NAIVE) hipLaunchKernelGGL((k_fwdtile<NT, 4, CS_FILL_NAIVE, true>), grid, block, lds, stream, A);
        else if (fill == CS_FILL_NAIVE_INTERPOLATING) hipLaunchKernelGGL((k_fwdtile<NT, 4, CS_FILL_NAIVE_INTERPOLATING, true>), grid, block, lds, stream, A);
        else hipLaunchKernelGGL((k_fwdtile<NT, 3, CS_FILL_NONE, true>), grid, block, lds, stream, A);
    } else if (fill == CS_FILL_INVERSE) hipLaunchKernelGGL((k_fwdtile<NT, 3, CS_FILL_INVERSE>), grid, block, lds, stream, A);
    else if (fill == CS_FILL_NAIVE) hipLaunchKernelGGL((k_fwdtile<NT, 4, CS_FILL_NAIVE>), grid, block, lds, stream, A);
    else if (fill == CS_FILL_NAIVE_INTERPOLATING) hipLaunchKernelGGL((k_fwdtile<NT, 4, CS_FILL_NAIVE_INTERPOLATING>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((k_fwdtile<NT, 3, CS_FILL_NONE>), grid, block, lds, stream, A);
    return hipGetLastError();
}

}  // namespace cs
