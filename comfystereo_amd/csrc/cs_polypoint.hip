// cs_polypoint.hip -- polylines_soft, second generation of the tiled fast path (reference
// stereoimage_generation.py:1912-1992): the kernel behind the headline metric.
//
// What bounds the first generation (cs_polytile.hip, one lane per OUTPUT pixel) is VALU issue: ~1000 vector instructions
// per wave of 128 pixels, most of them not arithmetic of the reference but bookkeeping -- finding the polyline points of
// a pixel through a registration pass, uint8 <-> float conversions, selects, byte shuffles.  On gfx950 only the plain
// float32 add / mul / fma (and a few integer ops) issue in 2 cycles per wave; conversions, shifts, selects, min/max, DPP
// take 4, v_rcp 8, ds_bpermute 24 (tools/ubench/).  This kernel is organised so that the bookkeeping disappears:
//
//   * one lane per polyline POINT (= source pixel of the halo'ed range, SLOTS of them per lane).  A point knows its output
//     pixel (floor x); ~92 % of the pixels hold exactly one point, and their two pieces [col, x], [x, col+1] belong to
//     the two segments around that point -- the lane has everything it needs in registers plus the two neighbouring
//     points, read back from LDS as one float4 {x, R, G, B} each.  No registration pass, no per-pixel lists, no sort.
//   * colours are converted ONCE per source pixel to the float value of their uint8 code (trunc(clamp(v * 255))), kept as
//     floats in LDS; results go straight from registers to global memory (stereoscope slot, mask) through the k / 255
//     table -- no result staging, no store phase, two barriers.
//   * where the polyline FOLDS (a segment running backwards: x[j+1] <= x[j]) several layers overlap exactly over the
//     x-extent of the reversed segments (intermediate value theorem: a polyline from -w to 2w passes every x an odd number
//     of times).  Reversed segments are found while staging (neighbour x by DPP), the pixels under them get a slot in
//     small per-tile lists, and only those pixels (3 % on the bench) go through the general search of the first
//     generation (every listed segment tested per sub-interval, largest interpolated |disparity| wins).
//   * pixels without a point (disocclusion bridges) and pixels with several points of one layer are appended to a list
//     and evaluated densely packed afterwards (one piece per bridge pixel; the chain path for the others).
// Anything that cannot be proven order-independent flags the ROW for the general row kernel (cs_rowwarp.hip), exactly
// like the first generation.  Arithmetic: dialect D32, every float32 rounding explicit
// (-ffp-contract=off), divisions as the correctly rounded FMA sequence hipcc emits minus its range scaling (operands
// are pixel coordinates: no overflow / underflow; tools/ubench/issue_rate.hip checks the identity on the GPU).
#include "cs_common.h"
#include "cs_kernels.h"
#include <type_traits>

namespace cs {

#define PP_THREADS 256
// -DCS_DEV builds (make -C comfystereo_amd/csrc dev): cs_debug_set(CS_DEBUG_DBG, n) cuts the kernel short so that hardware
// counters can be attributed to its phases (31: after staging, 32: before phase C, 33: phase C without bridges and fold
// registration, 34: no pass 2, 35: no general search, 36: no powf for the risky squares, 37: no depth-map output).  Release builds compile the tests away.
#ifdef CS_DEV
#define PP_DEV_IS(n) (A.dbg == (n))
// (development) why a row was handed to the general kernel: reason bits OR-ed into the frame's spare stats word 12
// (CS_DEBUG_DBG 15: lane events per reason as well -- word 13: list / point capacities (1, 4, 256), 14: ties (128), 15: everything else)
#define PP_HAZARD(code) do { hazard = true; if (A.stats_rw) { atomicOr(&A.stats_rw[(size_t)frame * ST_WORDS + 12], (unsigned)(code)); \
    if (A.dbg == 15) atomicAdd(&A.stats_rw[(size_t)frame * ST_WORDS + (((code) & (1 | 4 | 256)) ? 13 : ((code) == 128 ? 14 : 15))], 1u); } } while (0)
#else
#define PP_DEV_IS(n) false
#define PP_HAZARD(code) do { hazard = true; } while (0)
#endif
#ifndef PP_DCAP
// (round 5: 128 -> 160, with 16-bit flags per tile pixel.  On depth saturated to 0 / 1 -- wide overlaps of a near and a far plateau -- the 128
// slots of the one-byte flags were what sent 47 % of the rows to the row kernel with the blur ON, where not one of them holds a tie
// (tools/sessions/r05_s28.sh): 64 x 4K saturated frames with the blur 2 770 -> 5 040 frames/s, polylines_sharp 1 060 -> 3 100.  160 is what fits:
// 21 584 B of LDS per workgroup at the bench geometry; from 176 slots on (21 936 B) the seventh workgroup per CU no longer fits and
// the kernel takes 9.88 instead of 9.65 ms per 64 frames on the headline workload (tools/sessions/r05_s29.sh ... r05_s32.sh))
#define PP_DCAP 160          // pixels under reversed segments a tile can hold in its lists (more -> row redo)
#endif
#ifndef PP_MINW
#define PP_MINW 7            // workgroups per CU the default geometry is compiled for (register budget)
#endif
// polylines_sharp: six workgroups per CU (80 VGPRs).  At the soft kernel's budget (7 per CU, 72 VGPRs) the sharp instantiation
// spilled 21 vector registers to scratch in its staging phase -- and scratch is memory: per 64 4K frames the kernel read 19.7
// instead of 8.9 GB and wrote 47.2 instead of 30.0 GB (profiles/r05a_sharp/pmc_fetch.txt, pmc_write.txt: 1.73 x the algorithmic
// traffic).  Without the spills: 4 015 -> 4 392 frames/s on stepped depth, 3 011 -> 3 357 on blobs (tools/sessions/r05_s3.sh).
#ifndef PP_SHARP_MINW
#define PP_SHARP_MINW 6
#endif
// polylines_sharp, lists of a pixel under a reversed segment (points / forward segments): 6 / 9 since round 6 (5 / 7 before -- what a HARD
// silhouette needs: three layers, two points per source; on depth with softened silhouettes, tools/synth.scene8, 48 % of the rows of a 4K
// frame overflowed them at the metric's divergence and went to the row kernel: 1 050 frames/s against 4 000 on stepped depth.  6 / 9
// flags 20 %: 1 690, with the second tier behind it 1 900; the price is 4 spilled vector registers at the 80-register budget,
// -2.5 % on stepped depth, tools/sessions/r06_s13.sh.  More dirty SLOTS -- PP_DCAP_SHARP 240 / 320 -- changed nothing: s12)
#ifndef PP_SHARP_KP
#define PP_SHARP_KP 6
#define PP_SHARP_KS 9
#endif
#ifndef PP_DCAP_SHARP
#define PP_DCAP_SHARP PP_DCAP
#endif
#ifndef PP_SOFT_PLCAP
#define PP_SOFT_PLCAP 0   // polylines_soft: entries of the pass-2 pixel list (0: one per tile pixel)
#endif
#ifndef PP_SOFT_KP
#define PP_SOFT_KP 4   // (polylines_soft: 72 registers, seven workgroups per CU, 21 584 bytes of LDS -- no room for more, section 10 of DESIGN.md)
#define PP_SOFT_KS 5
#endif
#ifndef PP_DCAP2
#define PP_DCAP2 192   // second tier (k_polypoint_listed): pixels under reversed segments a tile can hold
#endif
#ifndef PP_SW_MINW_SOFT
#define PP_SW_MINW_SOFT 5   // (soft first tier: 95 registers, no spill; 5 against 4 workgroups per CU: D64 3 607 -> 4 038 frames/s on stepped depth, + 9 % on scene8; 6: no better -- tools/sessions/r06_s35.sh)
#endif
#ifndef PP_SW_MINW
#define PP_SW_MINW 4   // workgroups per CU of the sweep-typing (SW) instantiations (sharp; soft: PP_SW_MINW_SOFT)
#endif
#define PP_DIRTY 0x8000u     // dflag (16 bits per tile pixel since round 5): pixel lies under a reversed segment; low 15 bits = its list slot

__constant__ csm::PowfTables c_pp_powf_tables = CS_POWF_TABLES_INIT;
// k / 255 by true division (convertResult / np2tensor, reference GenerateStereo.py:41-44): evaluated by the compiler in IEEE
// float32, copied to LDS by every workgroup
struct Lut255 {
    float v[256];
    constexpr Lut255() : v() { for (int i = 0; i < 256; i++) v[i] = (float)i / 255.0f; }
};
__constant__ Lut255 c_pp_lut255 = Lut255();

struct PolyPointArgs {
    int n, h, w, S, T;
    const float* image_f32;
    const uint8_t* image_u8;
    const uint32_t* stats;
    uint32_t* stats_rw;
    int scale_from_stats;
    float e32, conv32;
    EyeArgs eye[2];
    int single;
    uint8_t* out_u8;
    float* stereo; float* mask; float* depth_l; float* depth_r;
    int out_h, out_w;
    uint8_t* rowflag;
    int dbg;
    // lazy depth blur (RowArgs::tilemap): the blurred maps only hold the tiles the map names, the rest is gray * scale
    const uint32_t* tilemap;
    const float* gray;
    int tm_words;
    // the per-eye constants packed in 64-bit words, so that the eye of a workgroup is picked by three 64-bit scalar selects
    // instead of eight 32-bit ones: {div32, sep32}, {xoff, yoff}, {st_min, st_max | enabled << 16}
    unsigned long long epk[2][3];
    // dialect bit "float64 disparity chain" (RowArgs::d64 & 1; the DIA instantiations, soft and sharp): the exponent as a double
    int d64; double e64;
    // tile hints (round 5): bit t of word [(frame * h + row) * 2 + eye] = tile t of that row-eye raised a hazard (tiles from 31 on
    // share bit 31); null: not recorded.  The row kernel can then confine itself to those tiles' columns.
    uint32_t* hint;
};

struct F3 { float x, y, z; };
struct B3 { uint8_t x, y, z; };
// the float32 output stores of the tile (stereoscope / depth map: 12 bytes per pixel, mask: 4).  -DPP_NT_STORES (experiment, round 5):
// the nontemporal forms -- the outputs are never read back by this kernel
typedef float pp_f3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void out_store3(char* p, float a, float b, float c) {
#ifdef PP_NT_STORES
    const pp_f3v v{a, b, c};
    asm volatile("global_store_dwordx3 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
#else
    *reinterpret_cast<F3*>(p) = F3{a, b, c};
#endif
}
__device__ __forceinline__ void out_store1(char* p, float a) {
#ifdef PP_NT_STORES
    __builtin_nontemporal_store(a, reinterpret_cast<float*>(p));
#else
    *reinterpret_cast<float*>(p) = a;
#endif
}

// lane i receives lane i + 1's value, lane 63 +inf -- one DPP move instead of a ds_bpermute round trip
__device__ __forceinline__ float wave_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x7f800000, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
using csm::code_over_255;

// per-pixel constants of the float64 ("Python float") branch of the sub-interval arithmetic (reference :1957-1960)
struct PixC { double sig_dd; float ff64, tf64, center64; };
__device__ __forceinline__ PixC pix_consts(int col) {
    PixC P;
    const double from_d = (double)col + 1e-7, to_d = (double)(col + 1) - 1e-7;
    P.sig_dd = to_d - from_d;
    P.ff64 = (float)from_d; P.tf64 = (float)to_d; P.center64 = (float)(from_d + 0.5 * P.sig_dd);
    return P;
}

enum { PF_HAZARD = 0, PF_NLIST = 1, PF_NDIRTY = 2, PF_DLO = 3, PF_DHI = 4, PF_JLO = 5, PF_JHI = 6, PF_WORDS = 8 };
// output forms: float32 node outputs / the uint8 codes of the stereoscope (cs_params.flags bit 1) / apply_stereo_divergence
// (uint8 image in, uint8 image out, nothing else)
// (PO_U8NM: uint8 codes without the mask -- the per-eye intermediate of the anaglyph modes, composed by k_anaglyph_compose)
enum { PO_F32 = 0, PO_U8 = 1, PO_ASD = 2, PO_U8NM = 3 };
// list entries: kind << 28 | point id << 12 | pixel (tile-local)
enum { PK_CHAIN = 0u, PK_BRIDGE = 1u };

// SHARP (round 4): polylines_sharp -- every source pixel is TWO polyline points, x - 0.45 and x + 0.45 (reference :1929-1934), joined
// by a "flat" segment (both ends refer to one source pixel: colour * length, :1981-1984); the segments between neighbouring
// sources interpolate as in soft.  The lane still owns a SOURCE pixel: one 8-byte record {colour codes, x} per source, the two
// points are x -+ 0.45 on read.  Point ids: 0 = left sentinel, 1 + 2 j = left point of source j, 2 + 2 j = its right point,
// 2 ns + 1 = right sentinel.
// DIA (round 5): the float64 disparity chain of dialect D64 (what an install WITH numba computes for
// `abs(d) ** e * div + sep` and the point x, rounded ONCE into the float32 point array; pinned by tests/golden/dialect_f64.npz)
// in the staging phase -- everything after it works on the float32 points as in D32.  polylines_sharp: a point is
// (float)(x64 -+ 0.45), which one float32 centre per source cannot carry -- the dialect instantiation keeps both points of every
// source in a second LDS array `xq` (8 more bytes per record: five instead of six workgroups per CU) and reads them from there.
// The tile function: one tile of one row-eye.  Inlined into k_polypoint (every row once, blockIdx decoded below) and into
// k_polypoint_listed (round 6: the second tier -- the rows the first pass flagged, once more with room for more pixels under reversed
// segments and longer lists, before the general row kernel gets what is left).  DC: pixels under reversed segments a tile can hold.
// SW (round 6, second session): numba's typing of the SWEEP as well (dialect bit "int64-sum" of cs_params.flags = RowArgs::d64 & 2; with the
// float64 chain: full D64, what an install WITH numba computes, reference :1951-1991 under @njit, oracle_polylines `g_dialect & 2`) -- every
// sub-interval quantity (from, to, length, centre, the segment parameter, the closeness, the colour term) in float64, the colour sums rounded
// to float32 after every piece.  Derived typing like every D64 statement: checked against the oracle under the same
// setting.  SW instantiations are compiled with DIA = 1 and pick the chain at run time (A.d64 & 1).
template <int NT, int SLOTS, int OUT, int PT_KP, int PT_KS, int SHARP, int DIA, int DC, int SW = 0>
__device__ __forceinline__ void pp_tile(const float* __restrict__ hot_image, const float* __restrict__ hot_depth0, const float* __restrict__ hot_depth1,
                                        int hot_w, int hot_h, int hot_S, int hot_T, int hot_single, int hot_off_dflag, int hot_off_dcnt, int hot_pow_mode,
                                        int hot_npt, int hot_off_xq, const PolyPointArgs& A, char* const smem, const int row, const int eyei,
                                        const int tile, const int frame) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = hot_T;
    EyeArgs E;
    E.depth = eyei ? hot_depth1 : hot_depth0;
    {
        const unsigned long long ea = eyei ? A.epk[1][0] : A.epk[0][0], eb = eyei ? A.epk[1][1] : A.epk[0][1],
                                 ec = eyei ? A.epk[1][2] : A.epk[0][2];
        E.div32 = __builtin_bit_cast(float, (uint32_t)ea); E.sep32 = __builtin_bit_cast(float, (uint32_t)(ea >> 32));
        E.xoff = (int)(uint32_t)eb; E.yoff = (int)(uint32_t)(eb >> 32);
        E.st_min = (int)(uint32_t)ec; E.st_max = (int)((uint32_t)(ec >> 32) & 0xffffu); E.enabled = (int)((ec >> 48) & 1u);
    }
    const bool eye_on = E.enabled;
    const int w = hot_w, h = hot_h;
    const int o0 = tile * T, wt = min(T, w - o0);
    const int s0 = max(0, o0 - hot_S - 1), s1 = min(w, o0 + wt + hot_S + 1), ns = s1 - s0;
    // local point ids: 0 = left sentinel (x = -w), 1 + j = source column s0 + j, ns + 1 = right sentinel (x = 2w); the
    // sentinels only matter when the staged range touches the frame border.  Every lane stages SLOTS points j = tid + 256 k;
    // slots beyond the row end re-read the last column and get x = 2w + (j - ns): slot ns IS the right sentinel.
    const int npts = SHARP ? 2 * ns + 2 : ns + 2;
    const float HW = (float)0.45;   // PIXEL_HALF_WIDTH of polylines_sharp as the float32 the point array holds (:1915, :1933-1934)
    // point records allocated: what the tile can stage (T + 2 S + sentinels, host: polypoint_npt), not every slot -- the
    // slots past it are only ever the filler beyond the right sentinel and are not stored
    const int NPT = hot_npt;
    const bool left_edge = s0 == 0, right_edge = s1 == w;

    // ---- LDS carve ----
    // a point = its colour codes in one dword (r | g << 8 | b << 16) and x: 8 bytes instead of a float4 -- four point slots
    // per lane (1024 points, tiles of 768 pixels: halo 1.21 x instead of 1.29 x, the prologue amortised over 39 % more
    // pixels) still fit seven workgroups per CU; a reader pays one v_cvt_f32_ubyteN per channel
    struct PQ { uint32_t rgb; float x; };
    PQ* P = (PQ*)smem;                                                            // [NPT] point o
    float* pz = (float*)(P + NPT);                                                // [NPT] |coord_d|
    auto ch0 = [](uint32_t c) { return (float)(c & 0xffu); };
    auto ch1 = [](uint32_t c) { return (float)((c >> 8) & 0xffu); };
    auto ch2 = [](uint32_t c) { return (float)((c >> 16) & 0xffu); };
    // point o of the polyline: x, the staged source column its colour comes from, |coord_d| (the slow paths; the fast path
    // reads the records of its three sources directly)
    auto pcol = [&](int o) { return min(max(SHARP ? (o - 1) >> 1 : o - 1, 0), ns - 1); };
    // (sharp under the dialect: left / right point of record r at xq[2 r], xq[2 r + 1]; `hot_off_xq` = 0 otherwise)
    const float* const xq = reinterpret_cast<const float*>(smem + hot_off_xq);
    auto sl = [&](int r, float xc) -> float { return (SHARP && DIA) ? xq[2 * r] : xc - HW; };       // left point of record r (centre xc)
    auto sr = [&](int r, float xc) -> float { return (SHARP && DIA) ? xq[2 * r + 1] : xc + HW; };   // right point
    auto px = [&](int o) -> float {
        if (!SHARP) return P[o].x;
        const int rr = 1 + pcol(o);
        const float xc = P[rr].x;
        const float v = ((o - 1) & 1) ? sr(rr, xc) : sl(rr, xc);
        return o <= 0 ? (float)(-1.0 * w) : (o >= npts - 1 ? (float)(2.0 * w) : v);
    };
    auto prgb = [&](int o) -> uint32_t { return SHARP ? P[1 + pcol(o)].rgb : P[o].rgb; };
    auto pzv = [&](int o) -> float {
        if (!SHARP) return pz[o];
        return (o <= 0 || o >= npts - 1) ? 0.0f : pz[1 + ((o - 1) >> 1)];
    };
    uint32_t* plist = (uint32_t*)(pz + NPT);                                      // [max(plcap, 128)] pixels evaluated in pass 2
    const int plcap = (!SHARP && PP_SOFT_PLCAP && DC == PP_DCAP) ? min(T, PP_SOFT_PLCAP) : T;   // (the host's polypoint_plcap)
    // The powf tables: exponents 1 and 2 only run the clone for the 0.4 % risky squares and read the tables where they are,
    // in constant memory (512 bytes, L1-resident: no copy, nothing to wait for before barrier 0: +2 %); any other exponent
    // sends every point through the clone, and the tables are copied into LDS (overlaying plist until barrier 1: +6 % there)
    csm::PowfTables* const tabs_lds = (csm::PowfTables*)plist;
    const bool all_powf = hot_pow_mode == 0;   // (host: dbg == 17 or an exponent other than 1 and 2)
    static_assert(sizeof(csm::PowfTables) == 512, "tables overlay");
    // (the two offsets that depend on T come precomputed in preloaded kernel arguments: plist + max(T, 128) and + (2 T + 3 & ~3))
    uint16_t* dflag = (uint16_t*)(smem + hot_off_dflag);                          // [T] PP_DIRTY | slot
    uint16_t* dcnt = (uint16_t*)(smem + hot_off_dcnt);                            // [DCAP] points (low 8) | segments (high 8)
    uint16_t* dpix = dcnt + DC;                                              // [DCAP] pixel of the slot
    uint16_t* pts = dpix + DC;                                               // [DCAP][PT_KP]
    uint16_t* sgs = pts + DC * PT_KP;                                        // [DCAP][PT_KS]
    int* flags = (int*)(sgs + DC * PT_KS);                                   // [PF_WORDS]
    // k / 255 (convertResult / np2tensor, reference GenerateStereo.py:41-44) by table: the kernel is bound by the NUMBER of
    // vector instructions (one quad-cycle each whatever the type, profiles/r03_polypoint.txt), and a table read costs one
    // (the shift) where the arithmetic of cs_math.h code_over_255 costs three; the LDS pipe has room
    float* lut = (float*)(flags + PF_WORDS);                                      // [256]

    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    const uint32_t rowpix = ((uint32_t)frame * (uint32_t)h + (uint32_t)row) * (uint32_t)w;   // pixel index < 2^31 (checked on the host)
    // ---- all global loads first: depth and image of the lane's points (clamped to the staged range: no branches) ----
    // (32-bit byte offsets from wave-uniform row bases: one address instruction per load)
    const char* const drow = reinterpret_cast<const char*>(E.depth + rowpix + s0);
    const char* const irow = reinterpret_cast<const char*>(reinterpret_cast<const F3*>(hot_image) + rowpix + s0);
    const char* const irow8 = reinterpret_cast<const char*>(reinterpret_cast<const B3*>(A.image_u8) + rowpix + s0);
    float dpre[SLOTS];
    F3 cpre[SLOTS];
    B3 cpre8[SLOTS];
    // lazy depth-blur tiles (cs_common.h): edge-free tiles come from the gray depth, times the frame's x255 scale
    const bool lazy = A.tilemap != nullptr;
    float zmul[SLOTS];
    if (lazy) {
        const LazySel Z = lazy_select(A.tilemap, A.tm_words, frame, h, row, s0, drow,
                                      reinterpret_cast<const char*>(A.gray + rowpix + s0), st[ST_SCALE255]);
        // the image loads do not depend on the map: they go out while the selector's scalar loads are in flight
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const uint32_t jc = (uint32_t)min(tid + k * NT, ns - 1);
            if (OUT == PO_ASD) cpre8[k] = *reinterpret_cast<const B3*>(irow8 + 3u * jc);
            else cpre[k] = *reinterpret_cast<const F3*>(irow + 12u * jc);
        }
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const uint32_t jc = (uint32_t)min(tid + k * NT, ns - 1);
            dpre[k] = lazy_load(Z, (uint32_t)s0 + jc, jc, zmul[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const uint32_t jc = (uint32_t)min(tid + k * NT, ns - 1);
            zmul[k] = 1.0f;
            dpre[k] = *reinterpret_cast<const float*>(drow + 4u * jc);
            if (OUT == PO_ASD) cpre8[k] = *reinterpret_cast<const B3*>(irow8 + 3u * jc);
            else cpre[k] = *reinterpret_cast<const F3*>(irow + 12u * jc);
        }
    }
    const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
    const float dmin = csm::ord2f(st[E.st_min]), dmax = csm::ord2f(st[E.st_max]);
    // depth * 255 inside the int32 range everywhere in this frame?  (else: the exact x86 conversion for the depth-map codes)
    const bool code_wraps = !(fmaxf(fabsf(dmin), fabsf(dmax)) < 8.0e6f);

    // LDS set-up in the shadow of the loads
    if (all_powf && tid < (int)(sizeof(csm::PowfTables) / 4))
        reinterpret_cast<uint32_t*>(tabs_lds)[tid] = reinterpret_cast<const uint32_t*>(&c_pp_powf_tables)[tid];
    for (int i = tid; i < (T + 1) / 2; i += NT) reinterpret_cast<uint32_t*>(dflag)[i] = 0;   // (two 16-bit flags per word)
    if (tid < DC / 2) reinterpret_cast<uint32_t*>(dcnt)[tid] = 0;
    if (OUT == PO_F32) {
        for (int i = tid; i < 256; i += NT) lut[i] = code_over_255((float)i);
    }
    if (tid < PF_WORDS) {   // minima start at INT_MAX, maxima at -1, counters at 0
        constexpr unsigned is_min = (1u << PF_DLO) | (1u << PF_JLO), is_max = (1u << PF_DHI) | (1u << PF_JHI);
        flags[tid] = (int)(0x7fffffffu * ((is_min >> tid) & 1u)) | -(int)((is_max >> tid) & 1u);
    }
    __syncthreads();  // barrier 0: tables, flags
    if (lazy) {
#pragma unroll
        for (int k = 0; k < SLOTS; k++) dpre[k] *= zmul[k];
    }

    const float o0f = (float)o0, o1f = (float)(o0 + wt);   // tile = [o0f, o1f)
    bool hazard = false;
    // the float64 disparity chain: compiled in by DIA; the sweep-typing instantiations (SW) decide per call
    const bool chain64 = DIA && (!SW || (A.d64 & 1));
    // ---- pixels under the reversed segment (xa -> xb), xb <= xa: slots in the tile's lists.  Called by whole waves.
    auto mark_reversed = [&](unsigned long long m, float xa, float xb) {
        const bool rev = (m >> lane) & 1ull;
        int lo = 0, n = 0, base = 0;
        if (rev) {
            const float fl = fmaxf(floorf(xb), o0f), fh = fminf(floorf(xa), o1f - 1.0f);
            lo = (int)fl - o0;
            n = (int)fh - (int)fl + 1;
            if (n > 0) {
                base = atomicAdd(&flags[PF_NDIRTY], n);
                atomicMin(&flags[PF_DLO], lo);
                atomicMax(&flags[PF_DHI], lo + n - 1);
            } else n = 0;
        }
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int llo = __builtin_amdgcn_readlane(lo, src), ln = __builtin_amdgcn_readlane(n, src),
                      lbase = __builtin_amdgcn_readlane(base, src);
            for (int i = lane; i < ln; i += 64) {
                const int s = lbase + i;
                if (s < DC) { dflag[llo + i] = (uint16_t)(PP_DIRTY | s); dpix[s] = (uint16_t)(llo + i); }
                else PP_HAZARD(2);
            }
        }
    };
    // ---- output rows of this tile: 32-bit offsets from wave-uniform bases ----
    const uint32_t obase = OUT == PO_ASD ? (rowpix + (uint32_t)o0)
                                         : (((uint32_t)frame * (uint32_t)A.out_h + (uint32_t)(row + E.yoff)) * (uint32_t)A.out_w + (uint32_t)(E.xoff + o0));
    char* const st_row = OUT == PO_ASD ? (char*)A.out_u8 + (size_t)obase * 3 : ((OUT == PO_U8 || OUT == PO_U8NM) ? (char*)A.stereo + (size_t)obase * 3 : (char*)A.stereo + (size_t)obase * 12);
    char* const mk_row = (OUT == PO_ASD || OUT == PO_U8NM) ? nullptr : (char*)A.mask + (size_t)obase * 4;
    // colour codes of tile pixel q as integer-valued floats 0..255
    auto emit_f = [&](int q, float r, float g, float b) {
        const uint32_t uq = (uint32_t)q;
        if (OUT == PO_F32) out_store3(st_row + 12u * uq, code_over_255(r), code_over_255(g), code_over_255(b));
        else *reinterpret_cast<B3*>(st_row + 3u * uq) = B3{(uint8_t)(int)r, (uint8_t)(int)g, (uint8_t)(int)b};
        if (OUT != PO_ASD && OUT != PO_U8NM) out_store1(mk_row + 4u * uq, __builtin_fmaxf(__builtin_fmaxf(r, g), b) == 0.0f ? 1.0f : 0.0f);
    };
    auto emit = [&](int q, int r, int g, int b) { emit_f(q, (float)r, (float)g, (float)b); };   // the same from integer codes
    // the same from the colour sums 0.5 <= k < 255.5 before truncation (the two hot call sites): the conversion truncates
    auto emit_k = [&](int q, float k0, float k1, float k2) {
        const uint32_t uq = (uint32_t)q;
        const uint32_t r = (uint32_t)k0, g = (uint32_t)k1, b = (uint32_t)k2;
        if (OUT == PO_F32) out_store3(st_row + 12u * uq, lut[r], lut[g], lut[b]);
        else *reinterpret_cast<B3*>(st_row + 3u * uq) = B3{(uint8_t)r, (uint8_t)g, (uint8_t)b};
        if (OUT != PO_ASD && OUT != PO_U8NM) out_store1(mk_row + 4u * uq, (r | g | b) == 0u ? 1.0f : 0.0f);
    };

    // =====================================================================================================
    // phase B: stage the lane's points: colour codes as floats, the libm-exact disparity -> x, reversed segments
    // =====================================================================================================
    // colour: np.clip(x * 255, 0, 255).astype(uint8) (reference :1508) as the float value of the code -> P[o].rgb right away
    // (slots beyond the row end hold the last column's colour: slot ns is the right sentinel, :1935)
    const int qoff = s0 - o0;
    uint32_t rgbk[SLOTS];   // (kept in registers until x is known: one 8-byte LDS store per point)
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
        if (OUT == PO_ASD) rgbk[k] = (uint32_t)cpre8[k].x | ((uint32_t)cpre8[k].y << 8) | ((uint32_t)cpre8[k].z << 16);
        else rgbk[k] = (uint32_t)__builtin_amdgcn_fmed3f(cpre[k].x * 255.0f, 0.0f, 255.0f) |
                       ((uint32_t)__builtin_amdgcn_fmed3f(cpre[k].y * 255.0f, 0.0f, 255.0f) << 8) |
                       ((uint32_t)__builtin_amdgcn_fmed3f(cpre[k].z * 255.0f, 0.0f, 255.0f) << 16);
    }
    // this eye's depth-map output: (depth * 255).astype(uint8) wraps mod 256 (quirk Q7), value code / 255 on three channels
#if defined(PP_SKELETON) && PP_SKELETON >= 3
    if (false) {
#else
    if (OUT != PO_ASD && !PP_DEV_IS(37)) {
#endif
        float* const dd = eyei == 0 ? A.depth_l : A.depth_r;
        char* const dd_row = (char*)dd + (size_t)(rowpix + (uint32_t)o0) * 12;
        int code[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; k++) code[k] = (int)((dpre[k] * scale) * 255.0f);   // (the low byte is taken below)
        if (code_wraps) {
#pragma unroll
            for (int k = 0; k < SLOTS; k++) code[k] = (int)csm::f32_to_u8_wrap((dpre[k] * scale) * 255.0f);
        }
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int q = tid + k * NT + qoff;   // tile pixel of this source column
            if ((unsigned)q < (unsigned)wt) {
                const float v = OUT == PO_F32 ? lut[code[k] & 0xff] : code_over_255((float)(code[k] & 0xff));
                out_store3(dd_row + 12u * (uint32_t)q, v, v, v);
            }
        }
    }
    if (!eye_on) {   // divergence < 0.001 for this eye: the source image (quirk Q10); rare, kept out of the hot path
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int q = tid + k * NT + qoff;
            if ((unsigned)q < (unsigned)wt) emit_f(q, ch0(rgbk[k]), ch1(rgbk[k]), ch2(rgbk[k]));
        }
        return;
    }
#if defined(PP_SKELETON) && PP_SKELETON == 2
    {   // (memory skeleton without the staging arithmetic: x = column + a cheap function of the depth; records as in production)
        PQ* const Pw = P + 1 + tid;
        float* const pzw = pz + 1 + tid;
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int j = tid + k * NT;
            const float x = (float)(s0 + j) + 0.5f + dpre[k] * E.div32;
            if (SLOTS * NT + 4 == NPT || 1 + j < NPT) { Pw[k * NT] = PQ{rgbk[k], x}; pzw[k * NT] = dpre[k] - dmin; }
        }
        if (tid == 0) P[0] = PQ{rgbk[0], (float)(-1.0 * w)};
    }
#else
    {
        const bool flat = dmax == dmin;
        const float range = dmax - dmin;
        // (d - dmin) / range through the refined reciprocal of the frame's range, computed once.  The residual steps need
        // numerators that are 0 or >= 2^-60 and a range within 2^+-40 (depth maps are 0..255); otherwise: plain division.
        const bool range_ok = range > 0x1p-40f && range < 0x1p40f;
        const float yr = range_ok ? rcp_refined(range) : 0.0f;
        const int pow_mode = hot_pow_mode;
        float sg[SLOTS], axs[SLOTS], pw[SLOTS], av[SLOTS];
        uint32_t amin = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            av[k] = dpre[k] * scale - dmin;
            amin = min(amin, __builtin_bit_cast(uint32_t, av[k]) - 1u);   // 0 -> 0xffffffff; negative (never) -> huge
        }
        const bool slow_div = __any(amin < 0x21800000u - 1u) || !range_ok;   // some 0 < a < 2^-60
        unsigned risk = 0;
        float nqv[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; k++) nqv[k] = div_with(av[k], range, yr);
        if (slow_div) {
            asm volatile("" ::: "memory");  // (keeps the compiler from speculating the slow division into the hot path)
#pragma unroll
            for (int k = 0; k < SLOTS; k++) nqv[k] = av[k] / range;
        }
        if (flat) {
#pragma unroll
            for (int k = 0; k < SLOTS; k++) nqv[k] = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const float nd = nqv[k] - A.conv32;
            sg[k] = nd >= 0.0f ? 1.0f : -1.0f;
            axs[k] = fabsf(nd);
            bool r = false;
            pw[k] = pow_mode == 1 ? axs[k] : (pow_mode == 2 ? csm::square_or_flag(axs[k], r) : 0.0f);
            if (pow_mode == 0) r = true;
            risk |= r ? 1u << k : 0u;
        }
        if (PP_DEV_IS(36) || chain64) risk = 0;   // (float64 chain: the float32 pow is not used)
        // the full powf clone for the risky arguments (all of them for other exponents).  Behind its own branch: the compiler
        // otherwise hoists the clone's ~55 constant set-up instructions in front of the loop test, where every wave pays them.
        if (__any(risk != 0u)) {
            asm volatile("" ::: "memory");
            do {
                float xin = 1.0f;
                int sel = -1;
#pragma unroll
                for (int k = SLOTS - 1; k >= 0; k--) if (risk & (1u << k)) { xin = axs[k]; sel = k; }
                const float r = all_powf ? csm::powf_exact_simt(xin, A.e32, tabs_lds) : csm::powf_exact_simt(xin, A.e32, &c_pp_powf_tables);
#pragma unroll
                for (int k = 0; k < SLOTS; k++) if (sel == k) pw[k] = r;
                risk &= risk - 1u;
            } while (__any(risk != 0u));
        }
        const float tidf = (float)tid;
        const float jf0 = (float)s0 + tidf + 0.5f;  // exact: integers + 0.5 below 2^23
        int wjlo = 0x7fffffff, wjhi = -1;
        PQ* const Pw = P + 1 + tid;
        float* const pzw = pz + 1 + tid;
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int j = tid + k * NT;
            float cdj = (sg[k] * pw[k]) * E.div32;                                   // coord_d   (:1926)
            float x = ((jf0 + (float)(k * NT)) + cdj) + E.sep32;                   // coord_x   (:1927)
            double cx64 = 0.0;
            if (chain64) {   // the same two lines in float64, each result rounded once (pow(x, 2) == x * x, pow(x, 1) == x: exact in float64)
                const double ax = (double)axs[k];
                const double p64 = A.e64 == 2.0 ? ax * ax : (A.e64 == 1.0 ? ax : pow(ax, A.e64));
                const double cd64 = ((double)sg[k] * p64) * A.eye[eyei].div64;
                cx64 = (((double)(s0 + j) + 0.5) + cd64) + A.eye[eyei].sep64;
                x = (float)cx64;
                cdj = (float)cd64;
            }
            // slots beyond the staged range: x = 2w + (j - ns), strictly increasing, slot ns = the right sentinel (:1935)
            // (sharp: centres 2w + 1 + 2 (j - ns): their points lie beyond the sentinel, increasing; the sentinel itself is px())
            x = j < ns ? x : (SHARP ? 2.0f * tidf + (float)(2 * w + 1 - 2 * ns + 2 * k * NT) : tidf + (float)(2 * w - ns + k * NT));
            // the two points of a sharp source: x -+ 0.45 in float32 (:1933-1934); under the dialect (float)(x64 -+ 0.45), kept in xq
            float xl = x - HW, xr = x + HW;
            if (SHARP && chain64 && j < ns) { xl = (float)(cx64 - 0.45); xr = (float)(cx64 + 0.45); }
            if (SLOTS * NT + 4 == NPT || 1 + j < NPT) {   // (compile-time true for geometries whose every slot is allocated)
                Pw[k * NT] = PQ{rgbk[k], x};
                pzw[k * NT] = fabsf(cdj);
                if (SHARP && DIA) {
                    float* const xqw = const_cast<float*>(xq) + 2 * (1 + j);
                    xqw[0] = xl; xqw[1] = xr;
                }
            }
            // reversed segment (j -> j+1)?  The right neighbour sits in the next lane (lane 63: +inf; the pairs across wave
            // chunks and the left sentinel's pair are checked after the barrier).
            // (sharp: the segment between the sources runs from this source's right point to the next one's left point; the
            // segment inside a source, 0.9 long, is never reversed)
            const float xfrom = SHARP ? xr : x, xn = SHARP ? wave_next(xl) : wave_next(x);
            const unsigned long long mrev = __ballot(!(xfrom < xn));
            if (mrev) mark_reversed(mrev, xfrom, xn);
            // the range of points that can lie in the tile: first j with x >= o0, last j with x < o0 + wt (wave-uniform
            // candidates from ballots; one pair of atomics per wave below)
            const unsigned long long m1 = __ballot(xfrom >= o0f), m2 = __ballot((SHARP ? xl : x) < o1f);
            const int base = k * NT + wave * 64;
            wjlo = min(wjlo, m1 ? base + __ffsll((long long)m1) - 1 : 0x7fffffff);
            wjhi = max(wjhi, m2 ? base + 63 - __clzll((long long)m2) : -1);
        }
        if (lane == 0) { atomicMin(&flags[PF_JLO], wjlo); atomicMax(&flags[PF_JHI], wjhi); }
        if (tid == 0) {
            P[0] = PQ{rgbk[0], (float)(-1.0 * w)};   // left sentinel (:1921): the first column's colour (frame border only)
            if (SHARP && DIA) { float* const xq0 = const_cast<float*>(xq); xq0[0] = (float)(-1.0 * w); xq0[1] = (float)(-1.0 * w); }
        }
    }
#endif
    __syncthreads();  // barrier 1: points staged, in-wave reversed segments marked
    if (PP_DEV_IS(31)) return;
#ifdef PP_SKELETON
    // (measurement, round 5: the kernel's MEMORY SKELETON -- every global load, the LDS point records and their read-back, every
    // output store in today's form (lane = source slot: dwordx3 at the slot's tile pixel + the mask dword; the depth-map stores
    // above) -- with the polyline arithmetic removed: a pixel's colour is a cheap function of the three records the fast path
    // reads.  Results are meaningless; never defined in a release.  PP_SKELETON=2: without the staging arithmetic as well,
    // PP_SKELETON=3: like 2 with the loads only (no output store at all), 4: like 2 without the depth-map stores)
    {
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int j = min(tid + k * NT, ns - 1), o = 1 + j;
            const PQ pm = P[o - 1], pc = P[o], pp = P[o + 1];
            const int q = tid + k * NT + qoff;
            const uint32_t mix = (pm.rgb + pp.rgb) ^ pc.rgb ^ (uint32_t)(pc.x > pm.x);
#if PP_SKELETON == 3
            if (mix == 0x12345678u && q == 77777) emit_k(0, 1.0f, 2.0f, 3.0f);
#else
            if ((unsigned)q < (unsigned)wt) emit_k(q, (float)(mix & 0xffu), (float)((mix >> 8) & 0xffu), (float)((mix >> 16) & 0xffu));
#endif
        }
        return;
    }
#endif
    // (model experiment, tools/r04_model.sh: -DPP_PAD_S=n / -DPP_PAD_V=n add n independent scalar / vector adds per wave here,
    // so that the cost of one more instruction of either kind is measured on the production kernel; never defined in a release)
#if defined(PP_PAD_S) || defined(PP_PAD_V)
    {
        int pa = tid, pb = row, pc = tile, pd = frame;
#ifdef PP_PAD_S
        int sa = __builtin_amdgcn_readfirstlane(pa), sb = __builtin_amdgcn_readfirstlane(pb), sc = __builtin_amdgcn_readfirstlane(pc),
            sd = __builtin_amdgcn_readfirstlane(pd);
#pragma unroll
        for (int i = 0; i < PP_PAD_S / 4; i++)
            asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd));
        if (sa + sb + sc + sd == 0x7ffffff3) hazard = true;
#endif
#ifdef PP_PAD_V
#pragma unroll
        for (int i = 0; i < PP_PAD_V / 4; i++)
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));
        if (pa + pb + pc + pd == 0x7ffffff3) hazard = true;
#endif
    }
#endif
#ifdef PP_PAD_P   // packed float32 adds (two results per instruction)
    {
        double pa = (double)tid, pb = (double)row;
#pragma unroll
        for (int i = 0; i < PP_PAD_P / 2; i++)
            asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1" : "+v"(pa), "+v"(pb));
        if (pa + pb == 12345.678) hazard = true;
    }
#endif
#ifdef PP_PAD_D   // ONE dependent chain of vector adds (every instruction waits for the previous one's result)
    {
        int pa = tid;
#pragma unroll
        for (int i = 0; i < PP_PAD_D / 4; i++)
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1" : "+v"(pa));
        if (pa == 0x7ffffff3) hazard = true;
    }
#endif
#ifdef PP_PAD_C   // the same with an instruction of the "4-cycle class" of tools/ubench (a conversion)
    {
        int pa = tid, pb = row, pc = tile, pd = frame;
#pragma unroll
        for (int i = 0; i < PP_PAD_C / 4; i++)
            asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte0 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte0 %3, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));
        if (pa + pb + pc + pd == 0x7ffffff3) hazard = true;
    }
#endif

    // ---- the segment pairs the staging loop could not see: lane 63 of every 64-point chunk (its right neighbour was staged
    // by another wave) and the left sentinel's pair.  Every wave checks the chunks IT staged -- lanes 0 .. SLOTS-1, one pair
    // each; wave 0 also the sentinel's -- and marks what it finds; the barrier is unconditional (a first version let every
    // wave check all pairs so that the barrier could be skipped when none is reversed: 75 instructions per wave for a dozen
    // comparisons, and the kernel is bound by the number of instructions it issues).
    {
        int o = -1;
        if (lane < SLOTS) {
            const int j = lane * NT + wave * 64 + 63;
            if (j <= ns - 2) o = 1 + j;   // the pair (j, j + 1) of real points
        } else if (lane == SLOTS && wave == 0 && left_edge) o = 0;
        // (records o and o + 1; sharp: right point of the one, left point of the other -- record 0 holds the sentinel's x itself)
        const float xa = o >= 0 ? ((SHARP && o > 0) ? sr(o, P[o].x) : P[o].x) : 0.0f, xb = o >= 0 ? (SHARP ? sl(o + 1, P[o + 1].x) : P[o + 1].x) : 1.0f;
        const unsigned long long mrev = __ballot(o >= 0 && !(xa < xb));
        if (mrev) mark_reversed(mrev, xa, xb);
        __syncthreads();
    }
    const int ndirty = min(flags[PF_NDIRTY], DC);
    const bool fold_tile = flags[PF_NDIRTY] > 0;
    const int dlo = flags[PF_DLO], dhi = flags[PF_DHI];
    if (flags[PF_NDIRTY] > DC) PP_HAZARD(1);
    // points that can lie in the tile, plus the one before them (its segment may bridge into the tile); sentinels excluded
    const int jlo = max(flags[PF_JLO] - 1, 0), jhi = min(flags[PF_JHI], ns - 1);
    if (!SHARP && tid == 0) { pz[0] = 0.0f; pz[npts - 1] = 0.0f; }   // sentinels (:1921, :1935); read after barrier 2 (sharp: pzv())
    if (PP_DEV_IS(32)) return;

    auto list_push = [&](uint32_t kind, int o, int q) {
        const unsigned idx = atomicAdd((unsigned*)&flags[PF_NLIST], 1u);
        if (idx < (unsigned)plcap) plist[idx] = (kind << 28) | ((uint32_t)o << 12) | (uint32_t)q;
        else PP_HAZARD(4);
    };
    const float eps32 = (float)1e-7;
    const float sig_whole = 0x1.fffffap-1f;  // (float)((col + 1 - 1e-7) - (col + 1e-7)) for every col >= 2 (tests/test_cs_math_host.py)
    const float tlo = o0f - 0.5f, thi = o1f - 0.5f;
    const bool edge_tile = left_edge || right_edge;

    // =====================================================================================================
    // phase C: every point that can lie in the tile looks at its pixel (lanes densely packed over [jlo, jhi])
    // =====================================================================================================
    // ---- pixels strictly between the end pixels of the forward segment (os -> os + 1), x0 -> x1 with floors fl0 / fl1:
    // disocclusion bridges, one piece each, appended to the list (a run of up to 3 pixels by its lane, longer ones by the
    // whole wave).  Called by whole waves.
    auto bridges = [&](bool has_seg, int os, float fl0, float fl1, float den) {
        const float nbr = fmin3(den, (fl1 - fl0) - 1.5f, fminf(fl1 - o0f - 0.5f, (o1f - 1.5f) - fl0));
        const bool has_bridge = nbr > 0.0f && has_seg && !PP_DEV_IS(33);
        if (__any(has_bridge)) {
            int pa = 1, pb = 0;
            if (has_bridge) {
                pa = fl0 < o0f ? 0 : (int)fl0 + 1 - o0;
                pb = fl1 > o1f - 1.0f ? wt - 1 : (int)fl1 - 1 - o0;
            }
            const bool is_long = pb - pa + 1 > 3;
            if (pb >= pa && !is_long) {
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int p = pa + t;
                    if (p <= pb && !(fold_tile && (dflag[p] & PP_DIRTY))) list_push(PK_BRIDGE, os, p);
                }
            }
            unsigned long long m = __ballot(is_long);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int lpa = __builtin_amdgcn_readlane(pa, src), lpb = __builtin_amdgcn_readlane(pb, src),
                          lo = __builtin_amdgcn_readlane(os, src);
                for (int p = lpa + lane; p <= lpb; p += 64)
                    if (!(fold_tile && (dflag[p] & PP_DIRTY))) list_push(PK_BRIDGE, lo, p);
            }
        }
    };
    // ---- fold tiles: the point op (pixel floor fp, `in_tile`) and the forward segment (os -> os + 1) go into the lists of the
    // pixels under reversed segments they lie in / pass over.  Called by whole waves.
    auto fold_register = [&](bool pt_in_tile, int qp, int op, bool has_seg, int os, float fl0, float fl1, float den) {
        if (pt_in_tile && (dflag[qp] & PP_DIRTY)) {
            const int sl = dflag[qp] & 0x7fff;
            const unsigned idx = atomic_add_u16(dcnt, sl, 1u) & 0xffu;
            if (idx < PT_KP) pts[sl * PT_KP + idx] = (uint16_t)op;
            else PP_HAZARD(8);
        }
        int p0 = 1, p1 = 0;
        if (den > 0.0f && has_seg && !(fl1 < o0f || fl0 > o1f - 1.0f)) {
            p0 = fl0 < o0f ? 0 : (int)fl0 - o0;
            p1 = fl1 > o1f - 1.0f ? wt - 1 : (int)fl1 - o0;
            p0 = max(p0, dlo); p1 = min(p1, dhi);   // only the dirty stretch of the tile matters
        }
        auto reg_seg = [&](int p, int oo) {
            const unsigned fl = dflag[p];
            if (fl & PP_DIRTY) {
                const int sl = fl & 0x7fff;
                const unsigned idx = (atomic_add_u16(dcnt, sl, 0x100u) >> 8) & 0xffu;
                if (idx < PT_KS) sgs[sl * PT_KS + idx] = (uint16_t)oo;
                else PP_HAZARD(16);
            }
        };
        const bool seg_long = p1 - p0 > 3;
        if (__any(p1 >= p0)) {
            if (!seg_long) {
#pragma unroll
                for (int t = 0; t < 4; t++) if (p0 + t <= p1) reg_seg(p0 + t, os);
            }
            unsigned long long m = __ballot(seg_long);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int lp0 = __builtin_amdgcn_readlane(p0, src), lp1 = __builtin_amdgcn_readlane(p1, src),
                          lo = __builtin_amdgcn_readlane(os, src);
                for (int p = lp0 + lane; p <= lp1; p += 64) reg_seg(p, lo);
            }
        }
    };
    for (int jb = jlo; jb <= jhi; jb += NT) {
        // (every lane runs the body -- the cooperative loops below need whole waves -- lanes beyond jhi re-read point jhi
        // and are kept from acting by `act`)
        if (jb + wave * 64 > jhi) continue;   // (a wave without a point left has nothing to do or to cooperate on)
        const bool act = jb + tid <= jhi;
        const int j = min(jb + tid, jhi);
        const int o = 1 + j;   // record of source j (soft: also the id of its point)
        const PQ pm = P[o - 1], pc = P[o], pp = P[o + 1];
        const bool has_seg = act && (j + 1 < ns || right_edge);   // the segment towards source j + 1 (or the right sentinel) exists and is this lane's
        if (!SHARP) {
        const float x = pc.x, xm = pm.x, xp = pp.x;
        const float f0 = floorf(x), f1 = floorf(xp), f0p1 = f0 + 1.0f;
        // ---- the fast path: one point in the pixel, two pieces [col, x] and [x, col+1] on the segments (o-1 -> o),
        // (o -> o+1).  Piece 0: from = col + eps (-> col in float32 for col >= 2), to = x - eps; piece 1: from = x + eps,
        // to = col + 1 - eps (-> col + 1).
        const float tf0 = x - eps32, sig0 = tf0 - f0, c0 = f0 + 0.5f * sig0;
        const float ff1 = x + eps32, sig1 = f0p1 - ff1, c1 = ff1 + 0.5f * sig1;
        const float den0 = x - xm, den1 = xp - x;
        // Conditions as positive margins (a < b <=> b - a > 0 for finite floats), merged by min:
        //   first point of a pixel of this tile:  xm < col, o0 <= col < o0 + wt
        //   the only one:                         floor(xp) > col
        //   both chain segments forward:          xm < x < xp
        //   col >= 2 (closed-form float64 constants), x > col (else piece 1 starts at the Python-float col + eps), and
        //   piece 1's centre right of x (its segment is active there); everything else the reference checks follows:
        //   col <= c0 <= x - eps, c1 <= col + 1 <= xp  (monotone rounding of exact sums).
        const float g_first = fmin3(f0 - xm, f0 - tlo, thi - f0);
        float g = fmin3(fmin3(den0, den1, (f1 - f0) - 0.5f), fmin3(sig0, f0 - 1.5f, c1 - x), g_first);
        // numba's typing of the sweep: the two pieces in float64 -- from = max(col, a) + eps, to = min(col + 1, b) - eps with a < col <= x <
        // col + 1 <= b.  What the float32 form needs beyond "first and only point, both chain segments forward" (col >= 2, the centre of
        // piece 1 right of x: float32 absorption) becomes: both pieces of positive length -- then col < centre 0 <= x - eps and x + eps <=
        // centre 1 <= col + 1 - eps (+- an ulp of float64), each inside its own chain segment and in order.
        double sgd0 = 0.0, sgd1 = 0.0, cd0 = 0.0, cd1 = 0.0;
        if (SW) {
            const double xd = (double)x, cold = (double)f0;
            const double fr0 = cold + 1e-7, to0 = xd - 1e-7, fr1 = xd + 1e-7, to1 = (cold + 1.0) - 1e-7;
            sgd0 = to0 - fr0; cd0 = fr0 + 0.5 * sgd0;
            sgd1 = to1 - fr1; cd1 = fr1 + 0.5 * sgd1;
            g = fmin3(fmin3(den0, den1, (f1 - f0) - 0.5f), g_first, (sgd0 > 0.0 && sgd1 > 0.0) ? 1.0f : -1.0f);
        }
        // a sentinel neighbour makes a "flat" piece (other typing): frame-border tiles only
        if (edge_tile) g = fminf(g, fminf((float)j - 0.5f, (float)(ns - 1 - j) - 0.5f));
        const int q = (int)f0 - o0;
        bool dirty = false;
        if (fold_tile) dirty = g_first > 0.0f && (dflag[q] & PP_DIRTY) != 0;
        const bool fast = g > 0.0f && !dirty && act;
        if (SW) {
            // (center - x0) / (x1 - x0): the difference of the two float32 points is taken in float32 (:1986, both operands float32), the rest in float64
            const double ip0 = (cd0 - (double)xm) / (double)den0, ip1 = (cd1 - (double)x) / (double)den1;
            const double om0 = 1.0 - ip0, om1 = 1.0 - ip1;
            const double cr = (double)ch0(pc.rgb), cg = (double)ch1(pc.rgb), cb = (double)ch2(pc.rgb);
            float k0 = (float)(0.5 + ((double)ch0(pm.rgb) * om0 + cr * ip0) * sgd0);
            float k1 = (float)(0.5 + ((double)ch1(pm.rgb) * om0 + cg * ip0) * sgd0);
            float k2 = (float)(0.5 + ((double)ch2(pm.rgb) * om0 + cb * ip0) * sgd0);
            k0 = (float)((double)k0 + (cr * om1 + (double)ch0(pp.rgb) * ip1) * sgd1);
            k1 = (float)((double)k1 + (cg * om1 + (double)ch1(pp.rgb) * ip1) * sgd1);
            k2 = (float)((double)k2 + (cb * om1 + (double)ch2(pp.rgb) * ip1) * sgd1);
            if (fast) emit_k(q, k0, k1, k2);
        } else {
            const float ip0 = div_core(c0 - xm, den0), ip1 = div_core(c1 - x, den1);
            const float om0 = 1.0f - ip0, om1 = 1.0f - ip1;
            // (both pieces have positive length here; the lerp operands are finite)
            const float cr = ch0(pc.rgb), cg = ch1(pc.rgb), cb = ch2(pc.rgb);
            float k0 = 0.5f + (ch0(pm.rgb) * om0 + cr * ip0) * sig0;
            float k1 = 0.5f + (ch1(pm.rgb) * om0 + cg * ip0) * sig0;
            float k2 = 0.5f + (ch2(pm.rgb) * om0 + cb * ip0) * sig0;
            k0 = k0 + (cr * om1 + ch0(pp.rgb) * ip1) * sig1;
            k1 = k1 + (cg * om1 + ch1(pp.rgb) * ip1) * sig1;
            k2 = k2 + (cb * om1 + ch2(pp.rgb) * ip1) * sig1;
            // (0.5 <= k < 255.5: the lerp operands are codes 0..255, the piece lengths sum to less than 1)
            if (fast) emit_k(q, k0, k1, k2);
        }
        // the first point of a pixel that is not done yet: several points / special typing -> pass 2 (chain path)
        if (g_first > 0.0f && !fast && !dirty && act) list_push(PK_CHAIN, o, q);
        bridges(has_seg, o, f0, f1, den1);
        if (fold_tile && !PP_DEV_IS(33)) {
            const bool in_tile = fminf(f0 - tlo, thi - f0) > 0.0f && act;
            fold_register(in_tile, q, o, has_seg, o, f0, f1, den1);
        }
        } else {
        // ================= polylines_sharp: source j = points q1 = x - 0.45 (id 1 + 2 j) and q2 = x + 0.45 (id 2 + 2 j); its
        // neighbours' nearest points q0 = x[j-1] + 0.45, q3 = x[j+1] - 0.45 and q4 = x[j+1] + 0.45.  Segments: (q0 -> q1) and
        // (q2 -> q3) interpolate between two sources, (q1 -> q2) and (q3 -> q4) are flat.  The lane owns the pixel whose FIRST
        // point is one of its two: `A` -- q1 is (q0 left of floor(q1)); else q2 (q1 left of floor(q2)).  Fast path: the pixel's
        // points are q1? q2? q3? in a row and the next one lies beyond the pixel -- up to four pieces
        //   [col, q1] lerp(j-1, j)  (A only) | [.., min(q2, col+1)] flat(j) | [q2, min(q3, col+1)] lerp(j, j+1) | [q3, col+1] flat(j+1)
        // A flat piece adds colour * length, which IS the lerp formula with (1 - ip, ip) = (1, 0) (c * 1 + c' * 0 == c exactly);
        // an absent piece gets length 0 (adds +0 exactly).  Same margins as soft: pieces of positive length whose centres lie
        // right of their segment's start (to the left of its end they lie by monotone rounding).
        const int oa = 1 + 2 * j, ob = oa + 1;
        float q0 = sr(o - 1, pm.x), q3 = sl(o + 1, pp.x), q4 = sr(o + 1, pp.x);
        const float q1 = sl(o, pc.x), q2 = sr(o, pc.x);
        if (edge_tile) {   // the frame's sentinels instead of the filler records: exact -w / 2w, no point beyond
            if (left_edge && j == 0) q0 = (float)(-1.0 * w);
            if (right_edge && j == ns - 1) { q3 = (float)(2.0 * w); q4 = INFINITY; }
        }
        const float fa = floorf(q1), fb = floorf(q2), f3 = floorf(q3);
        const float ina = fminf(fa - tlo, thi - fa), inb = fminf(fb - tlo, thi - fb);   // > 0: that pixel belongs to this tile
        const bool ownA = fminf(fa - q0, ina) > 0.0f;                                      // q1 is the first point of a pixel of the tile
        const bool ownB = fminf(fb - q1, inb) > 0.0f;                                      // q2 is
        const float colf = ownA ? fa : fb, colp1 = colf + 1.0f;
        const bool e2 = q2 < colp1, e3 = q3 < colp1;                                     // q2 / q3 inside the pixel (ownA: q2 may be; ownB: q2 is)
        // piece bounds (from = col or point + eps, to = point - eps or col + 1: float32, col >= 2 checked below)
        const float to0 = q1 - eps32, sg0 = ownA ? to0 - colf : 0.0f, c0 = colf + 0.5f * sg0;
        const float fr1 = ownA ? q1 + eps32 : colf, to1 = e2 ? q2 - eps32 : colp1, sg1 = to1 - fr1, c1 = fr1 + 0.5f * sg1;
        const float fr2 = q2 + eps32, to2 = e3 ? q3 - eps32 : colp1, sg2r = to2 - fr2, c2 = fr2 + 0.5f * sg2r, sg2 = e2 ? sg2r : 0.0f;
        const float fr3 = q3 + eps32, sg3r = colp1 - fr3, c3 = fr3 + 0.5f * sg3r, sg3 = e3 ? sg3r : 0.0f;
        const float den0 = q1 - q0, den2 = q3 - q2;
        // margins: owner of a pixel with col >= 2; the interpolating segments forward; the point after the pixel's last one
        // beyond the pixel; every present piece of positive length with its centre right of its segment's start
        float g = fmin3(colf - 1.5f, den2, (floorf(q4) - colf) - 0.5f);
        g = fminf(g, ownA ? fmin3(den0, sg0, c1 - q1) : 1.0f);
        g = fminf(g, sg1);
        g = fminf(g, e2 ? fminf(sg2r, c2 - q2) : 1.0f);
        g = fminf(g, e3 ? fminf(sg3r, c3 - q3) : 1.0f);
        if (edge_tile) g = fminf(g, fminf((float)j - 0.5f, (float)(ns - 1 - j) - 0.5f));   // sentinel neighbours: other typing
        // numba's typing of the sweep: the up to four pieces in float64 (see the soft branch): owner, the interpolating segments forward, the
        // point after the pixel's last one beyond the pixel, every present piece of positive length
        double s0d = 0.0, s1d = 0.0, s2d = 0.0, s3d = 0.0, c0d = 0.0, c2d = 0.0;
        if (SW) {
            const double cold = (double)colf, top = (cold + 1.0) - 1e-7;
            const double q1d = (double)q1, q2d = (double)q2, q3d = (double)q3;
            const double fr0d = cold + 1e-7, s0r = (q1d - 1e-7) - fr0d;
            const double fr1d = (ownA ? q1d : cold) + 1e-7;
            s1d = (e2 ? q2d - 1e-7 : top) - fr1d;
            const double fr2d = q2d + 1e-7, s2r = (e3 ? q3d - 1e-7 : top) - fr2d;
            const double fr3d = q3d + 1e-7, s3r = top - fr3d;
            c0d = fr0d + 0.5 * s0r; c2d = fr2d + 0.5 * s2r;
            s0d = ownA ? s0r : 0.0; s2d = e2 ? s2r : 0.0; s3d = (e2 && e3) ? s3r : 0.0;
            const bool lens = s1d > 0.0 && (!ownA || s0r > 0.0) && (!e2 || s2r > 0.0) && (!(e2 && e3) || s3r > 0.0);
            g = fmin3(den2, (floorf(q4) - colf) - 0.5f, lens ? 1.0f : -1.0f);
            g = fminf(g, ownA ? den0 : 1.0f);
            if (edge_tile) g = fminf(g, fminf((float)j - 0.5f, (float)(ns - 1 - j) - 0.5f));
        }
        const bool own = ownA || ownB;
        const int q = (int)colf - o0;
        bool dirty = false;
        if (fold_tile) dirty = own && (dflag[own ? q : 0] & PP_DIRTY) != 0;
        const bool fast = g > 0.0f && own && !dirty && act;
        if (SW) {
            // lerp pieces: (cl * (1 - ip) + cr * ip) * sig; flat pieces (both ends one source, :1981-1984): c * sig; every term float64, the
            // sum rounded to float32 piece by piece; an absent piece has length 0 and adds +0 exactly
            const double ip0 = ((ownA ? c0d - (double)q0 : 0.0)) / (double)(ownA ? den0 : 1.0f), ip2 = (c2d - (double)q2) / (double)den2;
            const double om0 = 1.0 - ip0, om2 = 1.0 - ip2;
            const double mr = (double)ch0(pm.rgb), mg = (double)ch1(pm.rgb), mb = (double)ch2(pm.rgb);
            const double cr = (double)ch0(pc.rgb), cg = (double)ch1(pc.rgb), cb = (double)ch2(pc.rgb);
            const double nr = (double)ch0(pp.rgb), ng = (double)ch1(pp.rgb), nb = (double)ch2(pp.rgb);
            float k0 = (float)(0.5 + (mr * om0 + cr * ip0) * s0d);
            float k1 = (float)(0.5 + (mg * om0 + cg * ip0) * s0d);
            float k2 = (float)(0.5 + (mb * om0 + cb * ip0) * s0d);
            k0 = (float)((double)k0 + cr * s1d); k1 = (float)((double)k1 + cg * s1d); k2 = (float)((double)k2 + cb * s1d);
            k0 = (float)((double)k0 + (cr * om2 + nr * ip2) * s2d);
            k1 = (float)((double)k1 + (cg * om2 + ng * ip2) * s2d);
            k2 = (float)((double)k2 + (cb * om2 + nb * ip2) * s2d);
            k0 = (float)((double)k0 + nr * s3d); k1 = (float)((double)k1 + ng * s3d); k2 = (float)((double)k2 + nb * s3d);
            if (fast) emit_k(q, k0, k1, k2);
        } else {
            // (without piece 0 its quotient must still be finite -- the segment (q0 -> q1) may be reversed or empty then, and
            // NaN * 0 would poison the sums: 0 / 1)
            const float ip0 = div_core(ownA ? c0 - q0 : 0.0f, ownA ? den0 : 1.0f), ip2 = div_core(c2 - q2, den2);
            const float om0 = 1.0f - ip0, om2 = 1.0f - ip2;
            const float mr = ch0(pm.rgb), mg = ch1(pm.rgb), mb = ch2(pm.rgb);
            const float cr = ch0(pc.rgb), cg = ch1(pc.rgb), cb = ch2(pc.rgb);
            const float nr = ch0(pp.rgb), ng = ch1(pp.rgb), nb = ch2(pp.rgb);
            float k0 = 0.5f + (mr * om0 + cr * ip0) * sg0;
            float k1 = 0.5f + (mg * om0 + cg * ip0) * sg0;
            float k2 = 0.5f + (mb * om0 + cb * ip0) * sg0;
            k0 = k0 + cr * sg1; k1 = k1 + cg * sg1; k2 = k2 + cb * sg1;
            k0 = k0 + (cr * om2 + nr * ip2) * sg2;
            k1 = k1 + (cg * om2 + ng * ip2) * sg2;
            k2 = k2 + (cb * om2 + nb * ip2) * sg2;
            k0 = k0 + nr * sg3; k1 = k1 + ng * sg3; k2 = k2 + nb * sg3;
            if (fast) emit_k(q, k0, k1, k2);
        }
        // pixels this source's points are the first of, not done: pass 2 (chain path from that point)
        if (own && !fast && !dirty && act) list_push(PK_CHAIN, ownA ? oa : ob, q);
        if (ownA && ownB && act) {   // both points first in their (different) pixels: the second one's takes the chain path
            const int qb = (int)fb - o0;
            if (!(fold_tile && (dflag[qb] & PP_DIRTY))) list_push(PK_CHAIN, ob, qb);
        }
        bridges(has_seg, ob, fb, f3, den2);   // (the flat segment, 0.9 long, has no pixel strictly inside)
        if (fold_tile && !PP_DEV_IS(33)) {
            fold_register(ina > 0.0f && act, (int)fa - o0, oa, act, oa, fa, fb, 1.0f);
            fold_register(inb > 0.0f && act, (int)fb - o0, ob, has_seg, ob, fb, f3, den2);
        }
        }
    }
    // ---- the left sentinel's segment (0 -> 1), frame border only: the pixels 0 .. floor(x1) - 1 under it take the chain path
    // (their piece is "flat"); in fold tiles it is a listed segment like the others
    if (left_edge && tid == 0) {
        const float x1 = px(1), fs = floorf(x1);
        if ((float)(-1.0 * w) < x1) {
            const int e1 = min((int)fs - 1 - o0, wt - 1);
            if (fs - 1.0f >= o0f)
                for (int p = 0; p <= e1; p++)
                    if (!(fold_tile && (dflag[p] & PP_DIRTY))) list_push(PK_BRIDGE, 0, p);
            if (fold_tile && !(fs < o0f)) {
                const int e2 = min(min((int)fs - o0, wt - 1), dhi);
                for (int p = max(0, dlo); p <= e2; p++) {
                    const unsigned fl = dflag[p];
                    if (fl & PP_DIRTY) {
                        const int s = fl & 0x7fff;
                        const unsigned idx = (atomic_add_u16(dcnt, s, 0x100u) >> 8) & 0xffu;
                        if (idx < PT_KS) sgs[s * PT_KS + idx] = (uint16_t)0;
                        else PP_HAZARD(16);
                    }
                }
            }
        }
    }
    __syncthreads();  // barrier 2: lists complete
    if (PP_DEV_IS(34) || PP_DEV_IS(33)) return;

    // =====================================================================================================
    // pass 2: the listed pixels, densely packed
    // =====================================================================================================
    const int nlist = min(flags[PF_NLIST], plcap);
    // ---- chain path (first generation, cs_polytile.hip eval_chain): the pixel's np points are CONSECUTIVE polyline points
    // o1 .. o1+np-1 with strictly increasing x inside the pixel, the np+1 segments around them are forward, and no other
    // layer covers the pixel (it is not under a reversed segment) -- sub-interval k belongs to chain segment k.  np == 0:
    // the one segment passing through.  Returns false when the pixel cannot be done (-> row redo).
    auto eval_chain = [&](bool act, int q, int o1in, bool bridge, int& r8, int& g8, int& b8) -> bool {
        const int col = o0 + q;
        const float colf = (float)col, colp1 = (float)(col + 1);
        // points of the pixel: the run of ids from o1 with floor(x) == col (the right sentinel ends every run)
        int npr = 0;
        if (act && !bridge) {
            npr = 1;
            while (npr <= PT_KP && floorf(px(min(o1in + npr, npts - 1))) == colf) npr++;
        }
        const int o1 = bridge ? o1in + 1 : o1in;   // np == 0: the segment's END point
        const int np = act ? min(npr, PT_KP) : 0;
        const PixC C = pix_consts(col);
        int wnp = 0;
#pragma unroll
        for (int t = 1; t <= PT_KP; t++) wnp = __any(np >= t) ? t : wnp;
        // the chain o1-1 .. o1+np must exist (sentinels only at the frame border) and fit the registers
        bool chain = act && npr <= PT_KP && o1 - 1 >= (left_edge ? 0 : 1) && o1 + np <= (right_edge ? npts - 1 : npts - 2);
        float cx[PT_KP + 2];
        float c0[PT_KP + 2], c1[PT_KP + 2], c2[PT_KP + 2];
        int cj[PT_KP + 2];
#pragma unroll
        for (int k = 0; k < PT_KP + 2; k++) {
            if (k <= wnp + 1) {
                const int o = chain && k <= np + 1 ? o1 - 1 + k : 0;
                const uint32_t vrgb = prgb(o);
                cx[k] = px(o); c0[k] = ch0(vrgb); c1[k] = ch1(vrgb); c2[k] = ch2(vrgb);
                cj[k] = pcol(o);
            } else { cx[k] = 0.0f; c0[k] = c1[k] = c2[k] = 0.0f; cj[k] = 0; }
        }
        // every chain segment is forward, the pixel's points lie inside it, the chain enters from the left of the pixel and
        // leaves to its right (then, the pixel not being under a reversed segment, the chain is all that covers it)
        float cxlast = 0.0f;
#pragma unroll
        for (int k = 0; k <= PT_KP; k++) {
            if (k <= wnp) chain = chain && (k > np || cx[k] < cx[k + 1]) && (k < 1 || k > np || (cx[k] < colp1 && !(cx[k] < colf)));
            cxlast = (k == np) ? cx[k + 1] : cxlast;
        }
        chain = chain && cx[0] < colf && !(cxlast < colp1);
        float color0 = 0.5f, color1 = 0.5f, color2 = 0.5f;
        float prev = colf;
        if (SW) {   // numba's typing: every piece in float64 (no piece is skipped: a piece of length 0 adds +-0)
            double prevd = (double)colf;
#pragma unroll
            for (int k = 0; k <= PT_KP; k++) {
                if (k <= wnp) {
                    const bool live = chain && k <= np;
                    const double ad = k == 0 ? (double)colf : fmax((double)colf, (double)cx[k]);   // (the point before the pixel lies left of it: checked above)
                    const double bd = k < np ? fmin((double)colp1, (double)cx[k + 1]) : (double)colp1;
                    const double fr = ad + 1e-7, sgd = (bd - 1e-7) - fr, cd = fr + 0.5 * sgd;
                    const bool ok = ((double)cx[k] < cd) && !((double)cx[k + 1] < cd);
                    chain = chain && (!live || ok) && (!live || (!(cd < prevd) && !(cd > (double)colp1)));
                    prevd = live ? cd : prevd;
                    const double ip_k = (cd - (double)cx[k]) / (double)(cx[k + 1] - cx[k]);
                    const double om = 1.0 - ip_k;
                    const bool flatp = cj[k] == cj[k + 1];
                    const float n0 = (float)((double)color0 + (flatp ? (double)c0[k] * sgd : ((double)c0[k] * om + (double)c0[k + 1] * ip_k) * sgd));
                    const float n1 = (float)((double)color1 + (flatp ? (double)c1[k] * sgd : ((double)c1[k] * om + (double)c1[k + 1] * ip_k) * sgd));
                    const float n2 = (float)((double)color2 + (flatp ? (double)c2[k] * sgd : ((double)c2[k] * om + (double)c2[k + 1] * ip_k) * sgd));
                    color0 = live ? n0 : color0;
                    color1 = live ? n1 : color1;
                    color2 = live ? n2 : color2;
                }
            }
        } else {
#pragma unroll
        for (int k = 0; k <= PT_KP; k++) {
            if (k <= wnp) {
                const bool live = chain && k <= np;
                const float a = k == 0 ? -INFINITY : cx[k];
                const float b = k < np ? cx[k + 1] : INFINITY;
                const bool from64 = !(a > colf), to64 = !(b < colp1);
                const bool sig64 = from64 && to64;
                const float ff = from64 ? C.ff64 : a + eps32;
                const float tf = to64 ? C.tf64 : b - eps32;
                const float sig_f = tf - ff;
                const float center = sig64 ? C.center64 : ff + 0.5f * sig_f;
                const bool work = live && (sig64 ? C.sig_dd != 0.0 : sig_f != 0.0f);
                // chain segment k must be the active one: x0 < centre <= x1, centres monotone inside the pixel
                const bool ok = (cx[k] < center) && !(cx[k + 1] < center);
                chain = chain && (!live || !work || ok) && (!live || (!(center < prev) && !(center > colp1)));
                prev = live ? center : prev;
                const float ip_k = (center - cx[k]) / (cx[k + 1] - cx[k]);
                const float om = 1.0f - ip_k;
                const float sg = sig64 ? (float)C.sig_dd : sig_f;
                float n0 = color0 + (c0[k] * om + c0[k + 1] * ip_k) * sg;
                float n1 = color1 + (c1[k] * om + c1[k + 1] * ip_k) * sg;
                float n2 = color2 + (c2[k] * om + c2[k + 1] * ip_k) * sg;
                const bool flatp = cj[k] == cj[k + 1];
                if (__any(work && flatp)) {   // both ends refer to one source pixel (sentinel pieces): other typing (:1981-1984)
                    if (flatp) {
                        if (sig64) {
                            n0 = (float)((double)color0 + (double)c0[k] * C.sig_dd);
                            n1 = (float)((double)color1 + (double)c1[k] * C.sig_dd);
                            n2 = (float)((double)color2 + (double)c2[k] * C.sig_dd);
                        } else {
                            n0 = color0 + c0[k] * sig_f;
                            n1 = color1 + c1[k] * sig_f;
                            n2 = color2 + c2[k] * sig_f;
                        }
                    }
                }
                color0 = work ? n0 : color0;
                color1 = work ? n1 : color1;
                color2 = work ? n2 : color2;
            }
        }
        }
        r8 = csm::f32_to_u8_wrap(color0); g8 = csm::f32_to_u8_wrap(color1); b8 = csm::f32_to_u8_wrap(color2);
        return chain;
    };
    for (int base = wave * 64; base < nlist; base += NT) {
        const int i = base + lane;
        const bool act = i < nlist;
        const uint32_t e = plist[act ? i : 0];
        const int q = (int)(e & 0xfffu), o = (int)((e >> 12) & 0xffffu);
        const bool bridge = (e >> 28) == PK_BRIDGE;
        // bridge pixels of real segments away from the first two columns: one whole-pixel piece, closed-form constants
        const bool lean = act && bridge && o >= 1 && o + 1 <= npts - 2 && o0 + q >= 2;
        if (__any(lean)) {
            // (sharp: a bridge is always the segment between two sources -- right point of the one, left point of the other)
            const PQ a = P[lean ? (SHARP ? o >> 1 : o) : 1], b = P[lean ? (SHARP ? (o >> 1) + 1 : o + 1) : 2];
            const float ax = SHARP ? sr(lean ? (o >> 1) : 1, a.x) : a.x, bx = SHARP ? sl(lean ? (o >> 1) + 1 : 2, b.x) : b.x;
            const float colf = (float)(o0 + q);
            if (SW) {   // the whole pixel as one piece in float64: [col + eps, col + 1 - eps]
                const double fr = (double)colf + 1e-7, sgd = (((double)colf + 1.0) - 1e-7) - fr, cd = fr + 0.5 * sgd;
                const double ip = (cd - (double)ax) / (double)(bx - ax), om = 1.0 - ip;
                const float k0 = (float)(0.5 + ((double)ch0(a.rgb) * om + (double)ch0(b.rgb) * ip) * sgd);
                const float k1 = (float)(0.5 + ((double)ch1(a.rgb) * om + (double)ch1(b.rgb) * ip) * sgd);
                const float k2 = (float)(0.5 + ((double)ch2(a.rgb) * om + (double)ch2(b.rgb) * ip) * sgd);
                if (lean) emit_k(q, k0, k1, k2);
            } else {
            const float center = colf + 0.5f;
            const float ip = div_core(center - ax, bx - ax), om = 1.0f - ip;
            const float k0 = 0.5f + (ch0(a.rgb) * om + ch0(b.rgb) * ip) * sig_whole;
            const float k1 = 0.5f + (ch1(a.rgb) * om + ch1(b.rgb) * ip) * sig_whole;
            const float k2 = 0.5f + (ch2(a.rgb) * om + ch2(b.rgb) * ip) * sig_whole;
            // the segment is forward, starts left of the pixel and ends right of it (checked when it was listed)
            if (lean) emit_k(q, k0, k1, k2);
            }
        }
        const bool rest = act && !lean;
        if (__any(rest)) {
            int r8 = 0, g8 = 0, b8 = 0;
            const bool ok = eval_chain(rest, q, o, bridge, r8, g8, b8);
            if (rest && ok) emit(q, r8, g8, b8);
            if (rest && !ok) PP_HAZARD(32);
        }
    }
    // ---- general search over the pixels under reversed segments (first generation, eval_generic): the pixel's points
    // sorted by (x, id) == the reference's stable insertion sort inside the pixel; every listed segment tested per
    // sub-interval; with several (or no) active segments the largest interpolated |disparity| with 0 < ip < 1 wins,
    // ties are order-dependent -> row redo.
    for (int base = (NT / 64 - 1 - wave) * 64; base < (PP_DEV_IS(35) ? 0 : ndirty); base += NT) {
        const int s = base + lane;
        bool pend = s < ndirty;
        const int q = dpix[pend ? s : 0];
        pend = pend && dflag[q] == (uint16_t)(PP_DIRTY | s);   // (a slot that lost its pixel to an overlapping reversed segment)
        const int col = o0 + q;
        const unsigned c = pend ? dcnt[s] : 0u;
        if ((c & 0xffu) > PT_KP || (c >> 8) > PT_KS) PP_HAZARD(256);
        const int np = min((int)(c & 0xffu), PT_KP), nsg = min((int)(c >> 8), PT_KS);
        int wnp = 0, wns = 0;
#pragma unroll
        for (int t = 1; t <= PT_KP; t++) wnp = __any(np >= t) ? t : wnp;
#pragma unroll
        for (int t = 1; t <= PT_KS; t++) wns = __any(nsg >= t) ? t : wns;
        const PixC C = pix_consts(col);
        const float colf = (float)col, colp1 = (float)(col + 1);
        float xs[PT_KP];
        int os[PT_KP];
#pragma unroll
        for (int k = 0; k < PT_KP; k++) { xs[k] = INFINITY; os[k] = 0x7fffffff; }
#pragma unroll
        for (int k = 0; k < PT_KP; k++) {
            if (k < wnp) {
                int o = k < np ? (int)pts[s * PT_KP + k] : 0x7fffffff;
                float x = k < np ? px(o) : INFINITY;
#pragma unroll
                for (int m2 = 0; m2 <= k; m2++) {
                    const bool lt = x < xs[m2] || (x == xs[m2] && o < os[m2]);
                    const float tx = lt ? xs[m2] : x; const int to = lt ? os[m2] : o;
                    xs[m2] = lt ? x : xs[m2]; os[m2] = lt ? o : os[m2];
                    x = tx; o = to;
                }
            }
        }
        // x-extent of every listed segment, read once: the sub-interval loop below first finds the ACTIVE segments of a centre
        // from these (x0 < centre <= x1; a missing or reversed segment never is), then evaluates only those -- in list order,
        // which is all the reference's scan depends on: an inactive segment changes neither the count, nor the pick, nor the
        // best closeness.  Under one reversed segment two layers overlap, so two evaluations replace five.
        float se0[PT_KS], se1[PT_KS];
#pragma unroll
        for (int e = 0; e < PT_KS; e++) {
            se0[e] = INFINITY; se1[e] = -INFINITY;
            if (e < wns) {
                const bool have = e < nsg;
                const int oe = have ? (int)sgs[s * PT_KS + e] : 0;
                const float x0 = px(oe), x1 = px(oe + 1);
                se0[e] = have ? x0 : INFINITY;
                se1[e] = have ? x1 : -INFINITY;
            }
        }
        float color0 = 0.5f, color1 = 0.5f, color2 = 0.5f;
        if (SW) {
        // numba's typing of the same scan (oracle_polylines, `g_dialect & 2`; cs_rowwarp.hip P3c under the dialect): centre, segment parameter
        // and closeness in float64; no piece is skipped
        double prevd = (double)colf, ad = (double)colf;
        for (int k = 0; k <= wnp; k++) {   // wave-uniform trip count; lanes with k > np idle
            const bool live = pend && k <= np;
            float b = INFINITY;
#pragma unroll
            for (int m2 = 0; m2 < PT_KP; m2++) b = (m2 == k && k < np) ? xs[m2] : b;
            // (a: the previous point of the pixel, or anything left of it; b: the next one, or anything beyond the pixel)
            const double bd = fmin((double)colp1, (double)b);
            const double fr = ad + 1e-7, sgd = (bd - 1e-7) - fr, cd = fr + 0.5 * sgd;
            ad = live ? fmax((double)colf, (double)b) : ad;
            if (live && (cd < prevd || cd > (double)colp1)) PP_HAZARD(64);
            prevd = live ? cd : prevd;
            unsigned am = 0;
#pragma unroll
            for (int e = 0; e < PT_KS; e++)
                if (e < wns) am |= (((double)se0[e] < cd) && !((double)se1[e] < cd)) ? (1u << e) : 0u;
            const int nact = __popc(am);
            int nqual = 0, o_pick = -1, o_best = -1;
            double ip_pick = 0.0, ip_best = 0.0;
            double bc = -1e-7;
            bool tie = false;
            auto scan_step = [&](bool on, int oe) {
                const float x0 = px(oe), x1 = px(oe + 1);
                const double ip_e = (cd - (double)x0) / (double)(x1 - x0);
                o_pick = on ? oe : o_pick;
                ip_pick = on ? ip_e : ip_pick;
                const bool qual = on && 0.0 < ip_e && ip_e < 1.0;
                const double cl = (1.0 - ip_e) * (double)pzv(oe) + ip_e * (double)pzv(oe + 1);
                nqual += qual ? 1 : 0;
                const bool better = qual && bc < cl;
                tie = better ? false : (tie || (qual && cl == bc));
                o_best = better ? oe : o_best;
                ip_best = better ? ip_e : ip_best;
                bc = better ? cl : bc;
            };
            if (!__any(live && nact > 2)) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const bool on = am != 0u;
                    const int e = on ? __ffs((int)am) - 1 : 0;
                    am &= am - 1u;
                    scan_step(on, on ? (int)sgs[s * PT_KS + e] : 1);
                }
            } else {
#pragma unroll
                for (int e = 0; e < PT_KS; e++)
                    if (e < wns) {
                        const bool on = (am >> e) & 1u;
                        scan_step(on, on ? (int)sgs[s * PT_KS + e] : 1);
                    }
            }
            const bool multi = live && nact != 1;
            if (multi && (nqual == 0 || tie)) PP_HAZARD(128);
            const bool use_best = multi && o_best >= 0;
            const bool contrib = live && (use_best || o_pick >= 0);
            const int o = contrib ? (use_best ? o_best : o_pick) : 1;
            const double ip_k = use_best ? ip_best : ip_pick;
            const uint32_t rgb_l = prgb(o), rgb_r = prgb(o + 1);
            const bool flatp = pcol(o) == pcol(o + 1);   // segment inside one source pixel: c * sig (:1981-1984)
            const double om = 1.0 - ip_k;
            const float n0 = (float)((double)color0 + (flatp ? (double)ch0(rgb_l) * sgd : ((double)ch0(rgb_l) * om + (double)ch0(rgb_r) * ip_k) * sgd));
            const float n1 = (float)((double)color1 + (flatp ? (double)ch1(rgb_l) * sgd : ((double)ch1(rgb_l) * om + (double)ch1(rgb_r) * ip_k) * sgd));
            const float n2 = (float)((double)color2 + (flatp ? (double)ch2(rgb_l) * sgd : ((double)ch2(rgb_l) * om + (double)ch2(rgb_r) * ip_k) * sgd));
            color0 = contrib ? n0 : color0;
            color1 = contrib ? n1 : color1;
            color2 = contrib ? n2 : color2;
        }
        } else {
        float prev = colf, a = -INFINITY;
        for (int k = 0; k <= wnp; k++) {   // wave-uniform trip count; lanes with k > np idle
            const bool live = pend && k <= np;
            float b = INFINITY;
#pragma unroll
            for (int m2 = 0; m2 < PT_KP; m2++) b = (m2 == k && k < np) ? xs[m2] : b;
            const bool from64 = !(a > colf), to64 = !(b < colp1);
            const bool sig64 = from64 && to64;
            const float ff = from64 ? C.ff64 : a + eps32;
            const float tf = to64 ? C.tf64 : b - eps32;
            const float sig_f = tf - ff;
            const float center = sig64 ? C.center64 : ff + 0.5f * sig_f;
            a = live ? b : a;
            if (live && (center < prev || center > colp1)) PP_HAZARD(64);
            prev = live ? center : prev;
            const bool work = live && (sig64 ? C.sig_dd != 0.0 : sig_f != 0.0f);
            unsigned am = 0;
#pragma unroll
            for (int e = 0; e < PT_KS; e++)
                if (e < wns) am |= ((se0[e] < center) && !(se1[e] < center)) ? (1u << e) : 0u;
            const int nact = __popc(am);
            int nqual = 0, o_pick = -1, o_best = -1;
            float ip_pick = 0.0f, ip_best = 0.0f;
            float bc = (float)(-1e-7);
            bool tie = false;
            // one step of the reference's scan (:1972-1980) for a segment that is active (`on`), polyline point oe -> oe + 1.
            // 0 < centre - x0 <= x1 - x0 <= 3 w: div_core's proven range.
            auto scan_step = [&](bool on, int oe) {
                const float x0 = px(oe), x1 = px(oe + 1);
                const float ip_e = div_core(center - x0, x1 - x0);
                o_pick = on ? oe : o_pick;
                ip_pick = on ? ip_e : ip_pick;
                const bool qual = on && 0.0f < ip_e && ip_e < 1.0f;
                const float cl = (1.0f - ip_e) * pzv(oe) + ip_e * pzv(oe + 1);
                nqual += qual ? 1 : 0;
                const bool better = qual && bc < cl;
                tie = better ? false : (tie || (qual && cl == bc));
                o_best = better ? oe : o_best;
                ip_best = better ? ip_e : ip_best;
                bc = better ? cl : bc;
            };
            if (!__any(work && nact > 2)) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const bool on = am != 0u;
                    const int e = on ? __ffs((int)am) - 1 : 0;
                    am &= am - 1u;
                    scan_step(on, on ? (int)sgs[s * PT_KS + e] : 1);
                }
            } else {
#pragma unroll
                for (int e = 0; e < PT_KS; e++)
                    if (e < wns) {
                        const bool on = (am >> e) & 1u;
                        scan_step(on, on ? (int)sgs[s * PT_KS + e] : 1);
                    }
            }
            const bool multi = work && nact != 1;
            if (multi && (nqual == 0 || tie)) PP_HAZARD(128);
            const bool use_best = multi && o_best >= 0;
            const bool contrib = work && (use_best || o_pick >= 0);
            const int o = contrib ? (use_best ? o_best : o_pick) : 1;
            const float ip_k = use_best ? ip_best : ip_pick;   // (the chosen segment's parameter: the same division as :1986)
            const uint32_t rgb_l = prgb(o), rgb_r = prgb(o + 1);
            const F3 pl{ch0(rgb_l), ch1(rgb_l), ch2(rgb_l)}, pr{ch0(rgb_r), ch1(rgb_r), ch2(rgb_r)};
            const int jl = pcol(o), jr = pcol(o + 1);
            const float om = 1.0f - ip_k;
            const float sgm = sig64 ? (float)C.sig_dd : sig_f;
            float n0 = color0 + (pl.x * om + pr.x * ip_k) * sgm;
            float n1 = color1 + (pl.y * om + pr.y * ip_k) * sgm;
            float n2 = color2 + (pl.z * om + pr.z * ip_k) * sgm;
            if (__any(contrib && jl == jr)) {  // segment inside one source pixel (sentinel pieces)
                if (jl == jr) {
                    if (sig64) {
                        n0 = (float)((double)color0 + (double)pl.x * C.sig_dd);
                        n1 = (float)((double)color1 + (double)pl.y * C.sig_dd);
                        n2 = (float)((double)color2 + (double)pl.z * C.sig_dd);
                    } else {
                        n0 = color0 + pl.x * sig_f;
                        n1 = color1 + pl.y * sig_f;
                        n2 = color2 + pl.z * sig_f;
                    }
                }
            }
            color0 = contrib ? n0 : color0;
            color1 = contrib ? n1 : color1;
            color2 = contrib ? n2 : color2;
        }
        }
        if (pend) emit(q, csm::f32_to_u8_wrap(color0), csm::f32_to_u8_wrap(color1), csm::f32_to_u8_wrap(color2));
    }
    if (hazard) {   // the row kernel redoes this EYE of the row (flag byte: bit 1 / 2 = eye 0 / 1; the byte is shared by both eyes'
                    // workgroups: an atomic OR on its word -- flagged rows are rare)
        const uint32_t idx = (uint32_t)frame * (uint32_t)h + (uint32_t)row;
        atomicOr(reinterpret_cast<unsigned*>(A.rowflag + (idx & ~3u)), (eyei ? 4u : 2u) << (8u * (idx & 3u)));
        // (soft only: in the sharp instantiation, at its 80-register budget, this second atomic costs a spilled vector register --
        // scratch, +5 % HBM traffic, -3 % on ordinary depth; its flagged row-eyes leave the hint word 0 = the whole row)
        // (round 6, measured: hints from the SECOND tier of sharp -- no spill there -- change nothing: on saturated depth the flagged tiles
        // span more than three quarters of the row, so the lean row kernel takes the whole row anyway, profiles/r06_s23/ab_sharp.txt)
        if (!SHARP && A.hint) atomicOr(&A.hint[2u * idx + (uint32_t)eyei], 1u << min(tile, 31));
    }
}

template <int NT, int SLOTS, int OUT, int PT_KP, int PT_KS, int MINW, int SHARP, int DIA = 0, int SW = 0>
__global__ void __launch_bounds__(NT, MINW)
k_polypoint(const float* __restrict__ hot_image, const float* __restrict__ hot_depth0, const float* __restrict__ hot_depth1,
            int hot_w, int hot_h, int hot_S, int hot_T, int hot_single, int hot_off_dflag, int hot_off_dcnt, int hot_pow_mode, int hot_npt, int hot_off_xq,
            PolyPointArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // grid = (tiles x 8 rows, rows / 8 [x eyes], frames), decoded with shifts (a scalar division costs ~30 SALU instructions).
    // Workgroup b runs on XCD b % 8 (observed dispatch order, MI355X_MICROARCH.md; a speed assumption only): with blockIdx.x =
    // tile * 8 + (row & 7) all tiles of a row land on one XCD, back to back, and the halo columns two neighbouring tiles share
    // come from that XCD's L2 the second time instead of from HBM (FETCH_SIZE -21 %).  The two eyes: cs_common.h eye_group_decode
    // (round 4: row groups of the two eyes alternate, the second eye's image rows come from L2; the eyes of a tile directly
    // next to each other had been measured 4 % slower in round 2 and 3 % slower again in round 4).
    const int xi = blockIdx.x;
    // (two-eye launches: blockIdx.y interleaves the eyes by row groups, cs_common.h eye_group_decode; single-eye: z = frame)
    int yrow = blockIdx.y, eyei = hot_single;
    if (hot_single < 0) eye_group_decode((int)blockIdx.y, yrow, eyei);
    const int row = yrow * 8 + (xi & 7);
    if (row >= hot_h) return;
    pp_tile<NT, SLOTS, OUT, PT_KP, PT_KS, SHARP, DIA, (SHARP ? PP_DCAP_SHARP : PP_DCAP), SW>(hot_image, hot_depth0, hot_depth1, hot_w, hot_h, hot_S, hot_T, hot_single, hot_off_dflag,
                                                                hot_off_dcnt, hot_pow_mode, hot_npt, hot_off_xq, A, smem, row, eyei, xi >> 3, (int)blockIdx.z);
}

// Second tier (round 6): the rows k_polypoint flagged (list entries frame * h + row | eye mask << 30, k_collect_rows), every flagged
// eye of them through the same tile function with longer per-pixel lists (and PP_DCAP2 slots for pixels under reversed segments) -- what a
// depth map with soft silhouettes needs at the metric's divergence (tools/synth.scene8: under a fold three layers overlap, a softened
// silhouette is three steep segments instead of one).  A PLAIN launch: grid = (tiles x 2 eyes, the first list entries [polypoint_tier2_cap]); a workgroup
// beyond the list's count, or of an eye the entry does not name, returns at once (the count is only known on the device; an idle
// workgroup costs a dispatch slot).  The first version looped persistent workgroups over the list around the inlined tile function:
// 128 registers, 37-57 of them spilled, four workgroups per CU -- 3-4 x the first tier's time per row (tools/sessions/r06_s11.sh).
// Rows this pass cannot finish either (exact ties, lists beyond even these capacities) are flagged in `A.rowflag` -- a second flag
// array -- for the row kernel.
static unsigned polypoint_tier2_cap(long long rows) {
    const long long q = (rows + 3) / 4;
    return (unsigned)(q < 256 ? (rows < 256 ? rows : 256) : (q > 65535 ? 65535 : q));
}
// list entries beyond the second tier's grid: flagged for the row kernel as they are (flag byte: bits 1 / 2 = eye 0 / 1, k_collect_rows)
__global__ void __launch_bounds__(256) k_polypoint_carry(const uint32_t* __restrict__ row_list, const uint32_t* __restrict__ row_count, unsigned cap,
                                                         uint8_t* __restrict__ rowflag2) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x + cap;
    if (i >= row_count[0]) return;
    const uint32_t e = row_list[i];
    rowflag2[e & 0x3fffffffu] = (uint8_t)((e >> 30) << 1);   // (one list entry per row: no other writer of this byte)
}
#ifndef PP_MINW2
#define PP_MINW2 5
#endif
template <int NT, int SLOTS, int OUT, int PT_KP, int PT_KS, int MINW, int SHARP, int SW = 0>
__global__ void __launch_bounds__(NT, MINW)
k_polypoint_listed(const float* __restrict__ hot_image, const float* __restrict__ hot_depth0, const float* __restrict__ hot_depth1,
                   int hot_w, int hot_h, int hot_S, int hot_T, int hot_off_dflag, int hot_off_dcnt, int hot_pow_mode, int hot_npt, int hot_off_xq,
                   const uint32_t* __restrict__ row_list, const uint32_t* __restrict__ row_count, PolyPointArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t li = blockIdx.y;
    if (li >= row_count[0]) return;
    const uint32_t e = row_list[li];
    const int eyei = (int)(blockIdx.x & 1u);
    if (!((e >> (30 + eyei)) & 1u)) return;
    const uint32_t idx = e & 0x3fffffffu;
    const int frame = (int)(idx / (uint32_t)hot_h), row = (int)(idx - (uint32_t)frame * (uint32_t)hot_h);
    // (SW: numba's typing of the sweep, compiled with the float64 chain picked at run time like the first tier's SW instantiations)
    pp_tile<NT, SLOTS, OUT, PT_KP, PT_KS, SHARP, (SW ? 1 : 0), PP_DCAP2, SW>(hot_image, hot_depth0, hot_depth1, hot_w, hot_h, hot_S, hot_T, -1, hot_off_dflag,
                                                                              hot_off_dcnt, hot_pow_mode, hot_npt, hot_off_xq, A, smem, row, eyei, (int)(blockIdx.x >> 1), frame);
}

// Depth-map output of an eye the tile kernel does not visit (modes left-only / only-right still return both depth maps,
// reference stereoimage_generation.py:1511-1516): (depth * 255).astype(uint8) wraps mod 256 (quirk Q7), value code / 255.
__global__ void __launch_bounds__(256) k_depth_codes(const float* __restrict__ depth, int hw, const uint32_t* stats,
                                                     int scale_from_stats, float* __restrict__ out) {
    __shared__ float lut[256];
    lut[threadIdx.x] = c_pp_lut255.v[threadIdx.x];
    __syncthreads();
    const int frame = blockIdx.y;
    const float scale = (scale_from_stats && stats[frame * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f;
    const float* d = depth + (size_t)frame * hw;
    F3* o = reinterpret_cast<F3*>(out) + (size_t)frame * hw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const float v = lut[csm::f32_to_u8_wrap((d[i] * scale) * 255.0f)];
        o[i] = F3{v, v, v};
    }
}

hipError_t launch_depth_codes(const float* depth, int n, int h, int w, const uint32_t* stats, int scale_from_stats, float* out,
                              hipStream_t stream) {
    const int hw = h * w;
    const int gx = (hw + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(k_depth_codes, dim3(gx < 1 ? 1 : (gx > 4096 ? 4096 : gx), n), dim3(256), 0, stream, depth, hw, stats,
                       scale_from_stats, out);
    return hipGetLastError();
}

// Anaglyph composition of the two eyes the tile kernel wrote as uint8 codes side by side ([n][h][2w][3]): R from one eye,
// G and B from the other (overlap_red_cyan, reference :1996-2010), k / 255, the no-fill mask of the composite
// (GenerateStereo.py:355-361).  With `rowflag`: rows flagged there are skipped (rounds 2-4: the row kernel wrote them in final form);
// round 5 passes null -- the row kernel writes flagged rows into the same side-by-side scratch and every row is composed here.
__global__ void __launch_bounds__(256) k_anaglyph_compose(const uint8_t* __restrict__ sbs, const uint8_t* __restrict__ rowflag, int w,
                                                          int anaglyph, float* stereo, int stereo_is_u8, float* mask) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const uint32_t row = blockIdx.y;   // frame * h + row
    if (x >= w || (rowflag && rowflag[row])) return;   // (rowflag null: every row)
    const B3 l = *reinterpret_cast<const B3*>(sbs + ((size_t)row * 2 * w + x) * 3);
    const B3 r = *reinterpret_cast<const B3*>(sbs + ((size_t)row * 2 * w + w + x) * 3);
    const B3 c = anaglyph == 1 ? B3{l.x, r.y, r.z} : B3{r.x, l.y, l.z};
    const size_t o = (size_t)row * w + x;
    if (stereo_is_u8) *reinterpret_cast<B3*>(reinterpret_cast<uint8_t*>(stereo) + o * 3) = c;
    else *reinterpret_cast<F3*>(stereo + o * 3) = F3{code_over_255((float)c.x), code_over_255((float)c.y), code_over_255((float)c.z)};
    mask[o] = ((int)c.x + (int)c.y + (int)c.z) == 0 ? 1.0f : 0.0f;
}

hipError_t launch_anaglyph_compose(const uint8_t* sbs, const uint8_t* rowflag, int n, int h, int w, int anaglyph, float* stereo,
                                   int stereo_is_u8, float* mask, hipStream_t stream) {
    hipLaunchKernelGGL(k_anaglyph_compose, dim3((w + 255) / 256, n * h), dim3(256), 0, stream, sbs, rowflag, w, anaglyph, stereo,
                       stereo_is_u8, mask);
    return hipGetLastError();
}

// point records a workgroup allocates: the staged range of a tile (T + 2 S + 2 source columns), both sentinels and the one
// filler slot a boundary-pair check may read behind them -- never more than the slots there are
static int polypoint_npt(int nt, int slots, int T, int S) {
    const int all = slots * nt + 4, need = ((T + 2 * S + 2 + 2 + 1 + 3) & ~3) + 4;
    return need < all ? need : all;
}
// entries of the pass-2 pixel list (pp_tile's plcap): one per tile pixel; the first tier of polylines_soft may take fewer (PP_SOFT_PLCAP)
static int polypoint_plcap(int T, int sharp, int dcap) {
    const int cap = (!sharp && PP_SOFT_PLCAP && dcap == PP_DCAP) ? (T < PP_SOFT_PLCAP ? T : PP_SOFT_PLCAP) : T;
    return cap > 128 ? cap : 128;
}
static size_t polypoint_lds(int nt, int slots, int T, int S, int KP, int KS, int dcap = PP_DCAP, int sharp = 1) {
    const size_t npt = (size_t)polypoint_npt(nt, slots, T, S);
    return 8 * npt + 4 * npt + 4 * (size_t)polypoint_plcap(T, sharp, dcap) + (size_t)((2 * T + 3) & ~3) + 2 * (size_t)dcap * (2 + (size_t)KP + KS) +
           4 * PF_WORDS + 1024 + 64;
}

// Tile width for a row of `w` pixels with halo S: the staged range (T + 2S + 2 points) plus the right sentinel must fit the
// `nslots` point slots of a workgroup; equal tiles, multiples of 4.  0: the halo is too wide for this geometry.
static int polypoint_tile(int w, int S, int nslots, int nt) {
    if (nslots + 1 >= 4096) return 0;  // 12-bit point field of the list entries
    int tmax = (nslots - 2 * S - 4) & ~3;
    if (tmax > 4 * nt) tmax = 4 * nt;  // one pass zeroes the per-pixel flags
#ifdef PP_TMAX
    if (tmax > PP_TMAX) tmax = PP_TMAX;   // (development: narrower tiles)
#endif
    if (tmax < 64) return 0;
    const int tiles = (w + tmax - 1) / tmax;
    int t = ((w + tiles - 1) / tiles + 3) & ~3;
    return t < 4 ? 4 : t;
}

int polypoint_max_halo() { return (3 * 384 - 4 - 64) / 2; }
// numba's typing of the sweep (SW instantiations): the default geometry only (256 threads x 4 point slots)
bool polypoint_sweep64_ok(int w, int halo) { return polypoint_tile(w, halo, 256 * 4, 256) != 0; }

template <int NT, int SLOTS, int MINW, int SHARP, int DIA = 0, int SW = 0>
static hipError_t polypoint_launch(PolyPointArgs& A, int out, hipStream_t stream) {
    // points / forward segments a pixel under a reversed segment can hold in its lists (more: the row is redone); sharp has two
    // points per source (the values the first-generation kernel settled on)
    constexpr int KP = SHARP ? PP_SHARP_KP : PP_SOFT_KP, KS = SHARP ? PP_SHARP_KS : PP_SOFT_KS;
    const int tiles = (A.w + A.T - 1) / A.T;
    dim3 grid(tiles * 8, A.single >= 0 ? (A.h + 7) / 8 : eye_group_grid_y(A.h), A.n), block(NT);   // (see the kernel's prologue)
    size_t lds = polypoint_lds(NT, SLOTS, A.T, A.S, KP, KS, SHARP ? PP_DCAP_SHARP : PP_DCAP, SHARP);
    // (development: CS_DEBUG_PT_VARIANT 13..16 pads the LDS request so that only 3..6 workgroups fit a CU -- occupancy what-if)
    const int npt = polypoint_npt(NT, SLOTS, A.T, A.S);
    const int off_dflag = 8 * npt + 4 * npt + 4 * polypoint_plcap(A.T, SHARP, SHARP ? PP_DCAP_SHARP : PP_DCAP), off_dcnt = off_dflag + ((2 * A.T + 3) & ~3);
    // sharp under the dialect: both points of every record, behind everything else
    const int off_xq = (SHARP && DIA) ? (int)((lds + 15) & ~(size_t)15) : 0;
    if (SHARP && DIA) lds = (size_t)off_xq + 8 * (size_t)npt + 64;
    const int pow_mode = (A.dbg == 17 || !(A.e32 == 2.0f || A.e32 == 1.0f)) ? 0 : (A.e32 == 2.0f ? 2 : 1);
    for (int e = 0; e < 2; e++) {
        const EyeArgs& E = A.eye[e];
        A.epk[e][0] = (unsigned long long)__builtin_bit_cast(uint32_t, E.div32) | ((unsigned long long)__builtin_bit_cast(uint32_t, E.sep32) << 32);
        A.epk[e][1] = (unsigned long long)(uint32_t)E.xoff | ((unsigned long long)(uint32_t)E.yoff << 32);
        A.epk[e][2] = (unsigned long long)(uint32_t)E.st_min | ((unsigned long long)((uint32_t)E.st_max & 0xffffu) << 32) |
                      ((unsigned long long)(E.enabled ? 1u : 0u) << 48);
    }
    const int occ = dev_switch(CS_DEBUG_PT_VARIANT) - 10;
    if (occ >= 3 && occ <= 6) { const size_t pad = (size_t)(163840 / (occ + 1) + 1024) & ~(size_t)255; if (pad > lds) lds = pad; }
#define PP_LAUNCH(O)                                                                                                         \
    {                                                                                                                        \
        hipError_t e = hipFuncSetAttribute((const void*)k_polypoint<NT, SLOTS, O, KP, KS, MINW, SHARP, DIA, SW>,                 \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        if (e != hipSuccess) return e;                                                                                       \
        hipLaunchKernelGGL((k_polypoint<NT, SLOTS, O, KP, KS, MINW, SHARP, DIA, SW>), grid, block, lds, stream, A.image_f32, A.eye[0].depth, \
                           A.eye[1].depth, A.w, A.h, A.S, A.T, A.single, off_dflag, off_dcnt, pow_mode, npt, off_xq, A);                  \
    }
    if (out == PO_F32) PP_LAUNCH(PO_F32)
    else if (out == PO_U8) PP_LAUNCH(PO_U8)
    else if (out == PO_U8NM) PP_LAUNCH(PO_U8NM)
    else PP_LAUNCH(PO_ASD)
#undef PP_LAUNCH
    return hipGetLastError();
}

// Launch for the eyes of `R` (SBS / TB / single-eye / uint8 outputs; no anaglyph).  `rowflag` must be zeroed by the caller;
// afterwards the general kernel is run over the flagged rows.
hipError_t launch_polypoint(const RowArgs& R, int S, uint8_t* rowflag, hipStream_t stream, int sharp, uint32_t* hint, int* tile_width) {
    // workgroup geometry: threads x point slots per lane.  256 x 4 (7 workgroups per CU at the bench halo, 19.5 KB of LDS each) is
    // the default; development switch CS_DEBUG_PT_VARIANT: 3 = 256 x 3 (the default until the end of round 3), 4 = 256 x 4,
    // 5 = 384 x 3, 6 = 320 x 3
    const int forced = dev_switch(CS_DEBUG_PT_VARIANT);
    int geo = (forced >= 3 && forced <= 7) ? forced : 4;
    if (sharp && geo != 5) geo = 4;
    auto nt_of = [](int g) { return g == 5 ? 384 : (g == 6 ? 320 : 256); };
    auto sl_of = [](int g) { return g == 4 ? 4 : (g == 7 ? 5 : 3); };
    if (polypoint_tile(R.w, S, nt_of(geo) * sl_of(geo), nt_of(geo)) == 0 && (geo == 3 || geo == 4)) geo = 5;
    const int nt = nt_of(geo), slots = sl_of(geo);
    PolyPointArgs A;
    A.n = R.n; A.h = R.h; A.w = R.w; A.S = S;
    A.T = polypoint_tile(R.w, S, nt * slots, nt);
    if (A.T == 0 || A.T + 2 * S + 3 > nt * slots) return hipErrorInvalidValue;
    A.image_f32 = R.image_f32; A.image_u8 = R.image_u8;
    A.stats = R.stats; A.stats_rw = R.stats_rw;
    A.scale_from_stats = R.scale_from_stats;
    A.e32 = R.e32; A.conv32 = R.conv32;
    A.eye[0] = R.eye[0]; A.eye[1] = R.eye[1];
    A.single = R.neyes == 1 ? 0 : R.single;
    A.out_u8 = R.out_u8; A.stereo = R.stereo; A.mask = R.mask; A.depth_l = R.depth_l; A.depth_r = R.depth_r;
    A.out_h = R.out_h; A.out_w = R.out_w;
    A.rowflag = rowflag;
    A.dbg = R.dbg;
    A.tilemap = R.tilemap; A.gray = R.lazy_gray; A.tm_words = R.tm_words;
    A.d64 = R.d64; A.e64 = R.e64;
    A.hint = hint;
    if (tile_width) *tile_width = A.T;
    // (dialect bits: 1 = the float64 disparity chain alone -- the sweep's typing stays D32; 2 / 3 = numba's typing of the sweep, without /
    // with the chain: the SW instantiations, default geometry only)
    if ((R.d64 & 2) && geo != 4) return hipErrorInvalidValue;
    const int out = R.out_u8 ? PO_ASD : (R.stereo_is_u8 ? (R.no_mask ? PO_U8NM : PO_U8) : PO_F32);
    if ((out == PO_ASD) != (R.image_u8 != nullptr)) return hipErrorInvalidValue;  // uint8 image in <=> uint8 image out
    if ((size_t)A.n * A.h * A.w >= (1ull << 31) || (size_t)A.n * A.out_h * A.out_w >= (1ull << 31) || A.h > 4 * 65535 - 512 || A.n > 65535)
        return hipErrorInvalidValue;   // 32-bit pixel indices, grid limits
    if (R.d64 & 2) {   // numba's sweep (float64 pieces: 128 registers, four workgroups per CU)
        if (sharp) return polypoint_launch<256, 4, PP_SW_MINW, 1, 1, 1>(A, out, stream);
        return polypoint_launch<256, 4, PP_SW_MINW_SOFT, 0, 1, 1>(A, out, stream);
    }
    if (R.d64) {   // the float64 disparity chain (one geometry per form: the default one, or the wide-halo one)
        if (sharp) return geo == 5 ? polypoint_launch<384, 3, 5, 1, 1>(A, out, stream) : polypoint_launch<256, 4, 5, 1, 1>(A, out, stream);
        if (geo == 5) return polypoint_launch<384, 3, 7, 0, 1>(A, out, stream);
        return polypoint_launch<256, 4, PP_MINW, 0, 1>(A, out, stream);
    }
    if (sharp) {   // (two geometries: the default and the wide-halo one)
        if (geo == 5) return polypoint_launch<384, 3, 7, 1>(A, out, stream);
        return polypoint_launch<256, 4, PP_SHARP_MINW, 1>(A, out, stream);
    }
    switch (geo) {
    case 3: return polypoint_launch<256, 3, PP_MINW, 0>(A, out, stream);
    case 4: return polypoint_launch<256, 4, PP_MINW, 0>(A, out, stream);
    case 5: return polypoint_launch<384, 3, 7, 0>(A, out, stream);
    case 6: return polypoint_launch<320, 3, 6, 0>(A, out, stream);
    case 7: return polypoint_launch<256, 5, PP_MINW, 0>(A, out, stream);
    default: return polypoint_launch<256, 4, PP_MINW, 0>(A, out, stream);
    }
}

// Second tier (k_polypoint_listed): the flagged rows of `list` / `count` (k_collect_rows) once more, with PP_DCAP2 slots and longer
// lists; what it cannot finish either is flagged in `rowflag2` (zeroed by the caller), tile hints (soft) in `hint2`.  Same tile width
// as the first tier (the hints name tiles).  hipErrorNotSupported: not for this call (dialect, single-eye layouts, wide-halo geometry).
template <int SHARP, int SW = 0>
static hipError_t polypoint_launch_listed(PolyPointArgs& A, int out, const uint32_t* list, const uint32_t* count, hipStream_t stream) {
    constexpr int NT = 256, SLOTS = 4, MINW = SW ? PP_SW_MINW : PP_MINW2;
    constexpr int KP = SHARP ? 8 : 6, KS = SHARP ? 12 : 9;   // (first tier: 6 / 9 and 4 / 5)
    const int tiles = (A.w + A.T - 1) / A.T;
    const long long rows = (long long)A.n * A.h;
    // x: tile and eye; y: the list entries this tier takes -- the first quarter of the rows there are (at most 65535: grid.y); an idle
    // workgroup costs a dispatch slot (1.97 M of them, every row of 64 4K frames: 0.71 ms, 5 % of the step, tools/sessions/r06_s15.sh),
    // so the grid is sized for what the tier is for -- some rows of a frame -- and k_polypoint_carry hands the entries beyond it on
    const unsigned cap = polypoint_tier2_cap(rows);
    dim3 grid(tiles * 2, cap), block(NT);
    hipLaunchKernelGGL(k_polypoint_carry, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, list, count, cap, A.rowflag);
    size_t lds = polypoint_lds(NT, SLOTS, A.T, A.S, KP, KS, PP_DCAP2, SHARP);
    const int npt = polypoint_npt(NT, SLOTS, A.T, A.S);
    const int off_dflag = 8 * npt + 4 * npt + 4 * (A.T > 128 ? A.T : 128), off_dcnt = off_dflag + ((2 * A.T + 3) & ~3);
    // sharp under the dialect: both points of every record, behind everything else (as in polypoint_launch)
    const int off_xq = (SHARP && SW) ? (int)((lds + 15) & ~(size_t)15) : 0;
    if (SHARP && SW) lds = (size_t)off_xq + 8 * (size_t)npt + 64;
    const int pow_mode = (A.dbg == 17 || !(A.e32 == 2.0f || A.e32 == 1.0f)) ? 0 : (A.e32 == 2.0f ? 2 : 1);
    for (int e = 0; e < 2; e++) {
        const EyeArgs& E = A.eye[e];
        A.epk[e][0] = (unsigned long long)__builtin_bit_cast(uint32_t, E.div32) | ((unsigned long long)__builtin_bit_cast(uint32_t, E.sep32) << 32);
        A.epk[e][1] = (unsigned long long)(uint32_t)E.xoff | ((unsigned long long)(uint32_t)E.yoff << 32);
        A.epk[e][2] = (unsigned long long)(uint32_t)E.st_min | ((unsigned long long)((uint32_t)E.st_max & 0xffffu) << 32) |
                      ((unsigned long long)(E.enabled ? 1u : 0u) << 48);
    }
#define PP_LAUNCH2(O)                                                                                                        \
    {                                                                                                                        \
        hipError_t e = hipFuncSetAttribute((const void*)k_polypoint_listed<NT, SLOTS, O, KP, KS, MINW, SHARP, SW>,               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        if (e != hipSuccess) return e;                                                                                       \
        hipLaunchKernelGGL((k_polypoint_listed<NT, SLOTS, O, KP, KS, MINW, SHARP, SW>), grid, block, lds, stream, A.image_f32, A.eye[0].depth, \
                           A.eye[1].depth, A.w, A.h, A.S, A.T, off_dflag, off_dcnt, pow_mode, npt, off_xq, list, count, A);  \
    }
    if (out == PO_F32) PP_LAUNCH2(PO_F32)
    else if (out == PO_U8) PP_LAUNCH2(PO_U8)
    else if (out == PO_U8NM) PP_LAUNCH2(PO_U8NM)
    else return hipErrorNotSupported;
#undef PP_LAUNCH2
    return hipGetLastError();
}

hipError_t launch_polypoint_tier2(const RowArgs& R, int S, uint8_t* rowflag2, hipStream_t stream, int sharp, uint32_t* hint2, int tile_width,
                                  const uint32_t* list, const uint32_t* count) {
    // (dialect: numba's typing of the sweep -- d64 & 2, round 6 -- has its instantiations; the float64 chain alone has none)
    if (R.d64 == 1 || R.neyes != 2 || R.single >= 0 || R.out_u8 || !R.image_f32) return hipErrorNotSupported;
    const int nt = 256, slots = 4;
    PolyPointArgs A;
    A.n = R.n; A.h = R.h; A.w = R.w; A.S = S;
    A.T = polypoint_tile(R.w, S, nt * slots, nt);
    if (A.T == 0 || A.T != tile_width || A.T + 2 * S + 3 > nt * slots) return hipErrorNotSupported;   // (the first tier ran another geometry)
    A.image_f32 = R.image_f32; A.image_u8 = nullptr;
    A.stats = R.stats; A.stats_rw = R.stats_rw;
    A.scale_from_stats = R.scale_from_stats;
    A.e32 = R.e32; A.conv32 = R.conv32;
    A.eye[0] = R.eye[0]; A.eye[1] = R.eye[1];
    A.single = -1;
    A.out_u8 = nullptr; A.stereo = R.stereo; A.mask = R.mask; A.depth_l = R.depth_l; A.depth_r = R.depth_r;
    A.out_h = R.out_h; A.out_w = R.out_w;
    A.rowflag = rowflag2;
    A.dbg = R.dbg;
    A.tilemap = R.tilemap; A.gray = R.lazy_gray; A.tm_words = R.tm_words;
    A.d64 = R.d64; A.e64 = R.e64;
    A.hint = hint2;
    if ((size_t)A.n * A.h * A.w >= (1ull << 31) || (size_t)A.n * A.out_h * A.out_w >= (1ull << 31)) return hipErrorNotSupported;
    const int out = R.stereo_is_u8 ? (R.no_mask ? PO_U8NM : PO_U8) : PO_F32;
    if (R.d64 & 2) return sharp ? polypoint_launch_listed<1, 1>(A, out, list, count, stream) : polypoint_launch_listed<0, 1>(A, out, list, count, stream);
    return sharp ? polypoint_launch_listed<1>(A, out, list, count, stream) : polypoint_launch_listed<0>(A, out, list, count, stream);
}

}  // namespace cs
