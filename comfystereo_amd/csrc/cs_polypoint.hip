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
// registration, 34: no pass 2, 35: no general search).  Release builds compile the tests away.
#ifdef CS_DEV
#define PP_DEV_IS(n) (A.dbg == (n))
#else
#define PP_DEV_IS(n) false
#endif
#define PP_DCAP 128          // pixels under reversed segments a tile can hold in its lists (more -> row redo)
#define PP_DIRTY 0x80u       // dflag: pixel lies under a reversed segment; low 7 bits = its list slot

__constant__ csm::PowfTables c_pp_powf_tables = CS_POWF_TABLES_INIT;

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
    int stereo_is_u8;
    int out_h, out_w;
    uint8_t* rowflag;
    int dbg;
};

struct F3 { float x, y, z; };
struct B3 { uint8_t x, y, z; };

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

// lane i receives lane i + 1's value (lane 63: undefined) -- one DPP move instead of a ds_bpermute round trip
__device__ __forceinline__ float wave_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

// per-pixel constants of the float64 ("Python float") branch of the sub-interval arithmetic (reference :1957-1960)
struct PixC { double sig_dd; float ff64, tf64, center64; };
__device__ __forceinline__ PixC pix_consts(int col) {
    PixC P;
    const double from_d = (double)col + 1e-7, to_d = (double)(col + 1) - 1e-7;
    P.sig_dd = to_d - from_d;
    P.ff64 = (float)from_d; P.tf64 = (float)to_d; P.center64 = (float)(from_d + 0.5 * P.sig_dd);
    return P;
}

enum { PF_HAZARD = 0, PF_NLIST = 1, PF_NDIRTY = 2, PF_DLO = 3, PF_DHI = 4, PF_WORDS = 8 };
// list entries: kind << 28 | point id << 12 | pixel (tile-local)
enum { PK_CHAIN = 0u, PK_BRIDGE = 1u };

template <int SLOTS, int PT_KP, int PT_KS, int MINW>
__global__ void __launch_bounds__(PP_THREADS, MINW)
k_polypoint(const float* __restrict__ hot_image, const float* __restrict__ hot_depth0, const float* __restrict__ hot_depth1,
            int hot_w, int hot_h, int hot_S, int hot_T, int hot_single, PolyPointArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = hot_T;
    const int tiles = (hot_w + T - 1) / T;
    const int bx = blockIdx.x, eyei = hot_single >= 0 ? hot_single : (int)blockIdx.z;
    const int tile = bx % tiles, row = bx / tiles, frame = blockIdx.y;
    EyeArgs E;
    E.depth = eyei ? hot_depth1 : hot_depth0;
    E.div32 = eyei ? A.eye[1].div32 : A.eye[0].div32;
    E.sep32 = eyei ? A.eye[1].sep32 : A.eye[0].sep32;
    E.enabled = eyei ? A.eye[1].enabled : A.eye[0].enabled;
    E.st_min = eyei ? A.eye[1].st_min : A.eye[0].st_min;
    E.st_max = eyei ? A.eye[1].st_max : A.eye[0].st_max;
    E.xoff = eyei ? A.eye[1].xoff : A.eye[0].xoff;
    E.yoff = eyei ? A.eye[1].yoff : A.eye[0].yoff;
    const bool eye_on = E.enabled;
    const int w = hot_w, h = hot_h;
    const int o0 = tile * T, wt = min(T, w - o0);
    const int s0 = max(0, o0 - hot_S - 1), s1 = min(w, o0 + wt + hot_S + 1), ns = s1 - s0;
    const int nsmax = T + 2 * hot_S + 2;
    // local point ids: 0 = left sentinel (x = -w), 1 + j = source column s0 + j, ns + 1 = right sentinel (x = 2w); the
    // sentinels only exist when the staged range touches the frame border
    const int npts = ns + 2, nptmax = nsmax + 2;
    const bool left_edge = s0 == 0, right_edge = s1 == w;

    // ---- LDS carve ----
    float* lut = (float*)smem;                                                    // [256] k / 255
    float4* P = (float4*)(smem + 1024);                                           // [nptmax] {x, R, G, B} of point o
    float* pz = (float*)(P + nptmax);                                             // [nptmax] |coord_d| (fold tiles only)
    csm::PowfTables* tabs = (csm::PowfTables*)pz;                                 //   (the powf tables until barrier 1)
    static_assert(sizeof(csm::PowfTables) == 512, "tables overlay");
    uint32_t* plist = (uint32_t*)(pz + max((nptmax + 3) & ~3, 128));              // [T] pixels evaluated in pass 2
    uint8_t* dflag = (uint8_t*)(plist + T);                                       // [T] PP_DIRTY | slot
    uint16_t* dcnt = (uint16_t*)(dflag + ((T + 3) & ~3));                         // [DCAP] points (low 8) | segments (high 8)
    uint16_t* dpix = dcnt + PP_DCAP;                                              // [DCAP] pixel of the slot
    uint16_t* pts = dpix + PP_DCAP;                                               // [DCAP][PT_KP]
    uint16_t* sgs = pts + PP_DCAP * PT_KP;                                        // [DCAP][PT_KS]
    int* flags = (int*)(sgs + PP_DCAP * PT_KS);                                   // [PF_WORDS]

    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    const size_t rowpix = ((size_t)frame * h + row) * w;
    // ---- all global loads first: depth and image of the lane's points ----
    const float* drow = E.depth + rowpix;
    float dpre[SLOTS];
    F3 cpre[SLOTS];
    const F3* irow = reinterpret_cast<const F3*>(hot_image) + rowpix + s0;
    const B3* irow8 = reinterpret_cast<const B3*>(A.image_u8) + rowpix + s0;
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
        const int j = tid + k * PP_THREADS;
        dpre[k] = j < ns ? drow[s0 + j] : 0.0f;
        cpre[k] = F3{0.f, 0.f, 0.f};
        if (j < ns) {
            if (hot_image) cpre[k] = irow[j];
            else { const B3 b = irow8[j]; cpre[k] = F3{(float)b.x, (float)b.y, (float)b.z}; }
        }
    }
    const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
    const float dmin = eye_on ? csm::ord2f(st[E.st_min]) : 0.0f, dmax = eye_on ? csm::ord2f(st[E.st_max]) : 0.0f;

    // LDS set-up in the shadow of the loads
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_pp_powf_tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(tabs);
        if (tid < (int)(sizeof(csm::PowfTables) / 4)) dst[tid] = src[tid];
    }
    lut[tid] = (float)tid / 255.0f;  // PP_THREADS == 256
    for (int i = tid; i < (T + 3) / 4; i += PP_THREADS) reinterpret_cast<uint32_t*>(dflag)[i] = 0;
    if (tid < PP_DCAP / 2) reinterpret_cast<uint32_t*>(dcnt)[tid] = 0;
    if (tid < PF_WORDS) flags[tid] = tid == PF_DLO ? 0x7fffffff : (tid == PF_DHI ? -1 : 0);
    __syncthreads();  // tables, flags

    const float o0f = (float)o0, o1f = (float)(o0 + wt);   // tile = [o0f, o1f)
    bool hazard = false;
    // ---- pixels under the reversed segment (xa -> xb), xb <= xa: slots in the tile's lists.  Called by whole waves.
    auto mark_reversed = [&](bool rev, float xa, float xb) {
        unsigned long long m = __ballot(rev);
        if (!m) return;
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
                if (s < PP_DCAP) { dflag[llo + i] = (uint8_t)(PP_DIRTY | s); dpix[s] = (uint16_t)(llo + i); }
                else hazard = true;
            }
        }
    };

    // =====================================================================================================
    // phase B: stage the lane's points: colour codes as floats, the libm-exact disparity -> x, reversed segments
    // =====================================================================================================
    float px_[SLOTS], pzv[SLOTS], cr[SLOTS], cg[SLOTS], cb[SLOTS];
    int dcode[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; k++) { px_[k] = 0.f; pzv[k] = 0.f; cr[k] = cg[k] = cb[k] = 0.f; dcode[k] = 0; }
    // colour: np.clip(x * 255, 0, 255).astype(uint8) (reference :1508) as the float value of the code
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
        if (hot_image) {
            cr[k] = truncf(__builtin_amdgcn_fmed3f(cpre[k].x * 255.0f, 0.0f, 255.0f));
            cg[k] = truncf(__builtin_amdgcn_fmed3f(cpre[k].y * 255.0f, 0.0f, 255.0f));
            cb[k] = truncf(__builtin_amdgcn_fmed3f(cpre[k].z * 255.0f, 0.0f, 255.0f));
        } else { cr[k] = cpre[k].x; cg[k] = cpre[k].y; cb[k] = cpre[k].z; }
    }
    if (eye_on) {
        const bool flat = dmax == dmin;
        const float range = dmax - dmin;
        // (d - dmin) / range with the refined reciprocal of the frame's range computed once; ranges near the float32 limits
        // (never seen: depth maps are 0..255) take the plain division
        const bool range_ok = range > 0x1p-40f && range < 0x1p40f;
        const float yr = range_ok ? rcp_refined(range) : 0.0f;
        const int pow_mode = A.dbg == 17 ? 0 : (A.e32 == 2.0f ? 2 : (A.e32 == 1.0f ? 1 : 0));
        float sg[SLOTS], axs[SLOTS], pw[SLOTS];
        unsigned risk = 0;
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int j = tid + k * PP_THREADS;
            const float d = dpre[k] * scale;
            // depth-map output code of this column: (depth * 255).astype(uint8) wraps mod 256 (quirk Q7)
            dcode[k] = csm::f32_to_u8_wrap(d * 255.0f);
            const float a = d - dmin;
            // (numerators below 2^-60 other than 0 -- differences of denormal-sized depths -- would round inside the residuals)
            float nq = div_with(a, range, yr);
            if (__any(j < ns && !(range_ok && (a == 0.0f || (a >= 0x1p-60f && a < 0x1p60f))))) {
                asm volatile("" ::: "memory");  // (keeps the compiler from speculating the slow division into the hot path)
                nq = a / range;
            }
            const float nd = flat ? 0.0f - A.conv32 : nq - A.conv32;
            sg[k] = nd >= 0.0f ? 1.0f : -1.0f;
            axs[k] = fabsf(nd);
            bool r = false;
            pw[k] = pow_mode == 1 ? axs[k] : (pow_mode == 2 ? csm::square_or_flag(axs[k], r) : 0.0f);
            if (pow_mode == 0) r = true;
            risk |= (r && j < ns) ? 1u << k : 0u;
        }
        while (__any(risk != 0u)) {  // the full powf clone for the risky arguments (all of them for other exponents)
            float xin = 1.0f;
            int sel = -1;
#pragma unroll
            for (int k = SLOTS - 1; k >= 0; k--) if (risk & (1u << k)) { xin = axs[k]; sel = k; }
            const float r = csm::powf_exact_simt(xin, A.e32, tabs);
#pragma unroll
            for (int k = 0; k < SLOTS; k++) if (sel == k) pw[k] = r;
            risk &= risk - 1u;
        }
        const float jf0 = (float)(s0 + tid) + 0.5f;  // exact: integers + 0.5 below 2^23
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int j = tid + k * PP_THREADS;
            const float cdj = (sg[k] * pw[k]) * E.div32;                                   // coord_d   (:1926)
            const float x = ((jf0 + (float)(k * PP_THREADS)) + cdj) + E.sep32;             // coord_x   (:1927)
            px_[k] = x;
            pzv[k] = fabsf(cdj);
            if (j < ns) P[1 + j] = make_float4(x, cr[k], cg[k], cb[k]);
            // reversed segment (j -> j+1)?  The right neighbour sits in the next lane; the pairs across wave chunks and the
            // sentinel pairs are checked after the barrier.
            const float xn = wave_next(x);
            const bool rev = lane != 63 && j + 1 < ns && !(x < xn);
            mark_reversed(rev, x, xn);
        }
        if (tid == 0) {  // sentinels (:1921, :1935): they refer to the first / last source column
            if (left_edge) P[0] = make_float4((float)(-1.0 * w), cr[0], cg[0], cb[0]);
        }
        if (right_edge) {
#pragma unroll
            for (int k = 0; k < SLOTS; k++)
                if (tid + k * PP_THREADS == ns - 1) P[npts - 1] = make_float4((float)(2.0 * w), cr[k], cg[k], cb[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < SLOTS; k++) dcode[k] = csm::f32_to_u8_wrap((dpre[k] * scale) * 255.0f);
    }
    __syncthreads();  // barrier 1: points staged, in-wave reversed segments marked
    if (PP_DEV_IS(31)) return;

    // ---- output helpers -------------------------------------------------------------------------------
    const size_t obase = A.out_u8 ? (rowpix + o0) : (((size_t)frame * A.out_h + row + E.yoff) * A.out_w + E.xoff + o0);
    auto emit = [&](int q, int r, int g, int b) {   // colour codes 0..255 of tile pixel q
        if (A.out_u8) {
            reinterpret_cast<B3*>(A.out_u8)[obase + q] = B3{(uint8_t)r, (uint8_t)g, (uint8_t)b};
        } else {
            if (A.stereo_is_u8) reinterpret_cast<B3*>(A.stereo)[obase + q] = B3{(uint8_t)r, (uint8_t)g, (uint8_t)b};
            else reinterpret_cast<F3*>(A.stereo)[obase + q] = F3{lut[r], lut[g], lut[b]};
            A.mask[obase + q] = (r | g | b) == 0 ? 1.0f : 0.0f;
        }
    };
    // this eye's depth-map output (code -> k / 255 on the three channels) and, for a disabled eye, the source image
    {
        float* dd = A.out_u8 ? nullptr : (eyei == 0 ? A.depth_l : A.depth_r);
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int j = tid + k * PP_THREADS;
            const int q = s0 + j - o0;
            if (j < ns && q >= 0 && q < wt) {
                if (dd) { const float v = lut[dcode[k] & 0xff]; reinterpret_cast<F3*>(dd)[rowpix + o0 + q] = F3{v, v, v}; }
                if (!eye_on) emit(q, (int)cr[k], (int)cg[k], (int)cb[k]);
            }
        }
    }
    if (!eye_on) return;

    // ---- the segment pairs the staging loop could not see (lane 63 of every chunk, the sentinel pairs): every wave
    // computes the same answer; wave 0 marks, and only then a barrier is needed (rare)
    {
        const int nb = (ns - 1) >> 6;  // pairs (j, j+1) with j = 64 b + 63
        bool any = false;
        for (int b0 = 0; b0 < nb + 2; b0 += 64) {
            const int b = b0 + lane;
            int o = -1;
            if (b < nb) o = 1 + 64 * b + 63;
            else if (b == nb && left_edge) o = 0;
            else if (b == nb + 1 && right_edge) o = npts - 2;
            const float xa = o >= 0 ? P[o].x : 0.0f, xb = o >= 0 ? P[o + 1].x : 1.0f;
            const bool rev = o >= 0 && !(xa < xb);
            if (__any(rev)) {
                any = true;
                if (wave == 0) mark_reversed(rev, xa, xb);
            }
        }
        if (any) __syncthreads();
    }
    const int ndirty = min(flags[PF_NDIRTY], PP_DCAP);
    const bool fold_tile = flags[PF_NDIRTY] > 0;
    const int dlo = flags[PF_DLO], dhi = flags[PF_DHI];
    if (flags[PF_NDIRTY] > PP_DCAP) hazard = true;

    if (PP_DEV_IS(32)) return;
    auto list_push = [&](uint32_t kind, int o, int q) {
        const unsigned idx = atomicAdd((unsigned*)&flags[PF_NLIST], 1u);
        if (idx < (unsigned)T) plist[idx] = (kind << 28) | ((uint32_t)o << 12) | (uint32_t)q;
        else hazard = true;
    };
    const float eps32 = (float)1e-7;
    const float sig_whole = 0x1.fffffap-1f;  // (float)((col + 1 - 1e-7) - (col + 1e-7)) for every col >= 2 (tests/test_cs_math_host.py)

    // =====================================================================================================
    // phase C: every point looks at its pixel
    // =====================================================================================================
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
        const int j = tid + k * PP_THREADS;
        const int o = 1 + j;
        const bool valid = j < ns;
        const float x = px_[k];
        const float4 pm = P[valid ? o - 1 : 0], pp = P[valid ? o + 1 : 0];
        if (fold_tile && valid) pz[o] = pzv[k];
        if (fold_tile && k == 0 && tid == 0) { pz[0] = 0.0f; pz[npts - 1] = 0.0f; }   // sentinels (:1921, :1935)
        const float xm = pm.x, xp = pp.x;
        const float f0 = floorf(x), f0p1 = f0 + 1.0f;
        const bool has_m = valid && (j >= 1 || left_edge), has_p = valid && (j + 1 < ns || right_edge);
        const bool in_tile = valid && f0 >= o0f && f0 < o1f;
        const int q = in_tile ? (int)f0 - o0 : 0;
        // is this point the first one of its pixel / the only one?  (a pixel under a reversed segment is nobody's)
        const bool first = in_tile && has_m && xm < f0;
        const bool single = has_p && !(xp < f0p1);
        bool dirty = false;
        if (fold_tile) dirty = in_tile && (dflag[q] & PP_DIRTY) != 0;
        // ---- the fast path: one point, two pieces [col, x] and [x, col+1] on the segments (o-1 -> o), (o -> o+1).
        // Needs real neighbours (a sentinel piece is "flat": other typing), col >= 2 (closed-form float64 constants) and
        // x > col (else the second piece starts at the Python-float col + eps).
        bool fast = first && single && !dirty && j >= 1 && j + 1 < ns && f0 >= 2.0f && x > f0;
        {
            const float tf0 = x - eps32;                // piece 0: from = col + eps (-> col as float32), to = x - eps
            const float sig0 = tf0 - f0;
            const float c0 = f0 + 0.5f * sig0;
            const float ff1 = x + eps32;                // piece 1: from = x + eps, to = col + 1 - eps (-> col + 1)
            const float sig1 = f0p1 - ff1;
            const float c1 = ff1 + 0.5f * sig1;
            const bool w0 = sig0 != 0.0f, w1 = sig1 != 0.0f;
            // the two chain segments are forward and active at their piece's centre, centres monotone inside the pixel
            bool ok = xm < x && x < xp;
            ok = ok && !(c0 < f0) && !(c1 < c0) && !(c1 > f0p1);
            ok = ok && (!w0 || (xm < c0 && !(x < c0))) && (!w1 || (x < c1 && !(xp < c1)));
            const bool listed = first && !dirty && !(fast && ok);   // several points / special typing -> pass 2
            fast = fast && ok;
            if (__any(fast)) {
                const float ip0 = div_core(c0 - xm, x - xm), ip1 = div_core(c1 - x, xp - x);
                const float om0 = 1.0f - ip0, om1 = 1.0f - ip1;
                // (a piece of length zero adds exactly 0: no select needed; the lerp operands are finite)
                float k0 = 0.5f + (pm.y * om0 + cr[k] * ip0) * sig0;
                float k1 = 0.5f + (pm.z * om0 + cg[k] * ip0) * sig0;
                float k2 = 0.5f + (pm.w * om0 + cb[k] * ip0) * sig0;
                k0 = k0 + (cr[k] * om1 + pp.y * ip1) * sig1;
                k1 = k1 + (cg[k] * om1 + pp.z * ip1) * sig1;
                k2 = k2 + (cb[k] * om1 + pp.w * ip1) * sig1;
                if (fast) emit(q, (int)k0 & 0xff, (int)k1 & 0xff, (int)k2 & 0xff);
            }
            if (__any(listed)) { if (listed) list_push(PK_CHAIN, o, q); }
        }
        // ---- pixels strictly between the end pixels of the forward segment (o -> o+1): disocclusion bridges, one piece
        // each; appended to the list (a run of up to 3 pixels by its lane, longer ones by the whole wave)
        if (!PP_DEV_IS(33)) {
            const float f1 = floorf(xp);
            const bool fwd = has_p && x < xp;
            int pa = 1, pb = 0;
            if (fwd && f1 - f0 >= 2.0f && !(f1 <= o0f || f0 >= o1f - 1.0f)) {
                pa = f0 < o0f ? 0 : (int)f0 + 1 - o0;
                pb = f1 > o1f - 1.0f ? wt - 1 : (int)f1 - 1 - o0;
            }
            // the left sentinel's segment (0 -> 1) belongs to the lane of point 1
            int sa = 1, sb = 0;
            if (left_edge && j == 0 && valid) {
                const float fs = floorf(x);  // pixels 0 .. floor(x1) - 1 lie under the sentinel segment: this tile's share
                if (fs - 1.0f >= o0f) { sa = 0; sb = min((int)fs - 1 - o0, wt - 1); }
            }
            const int nrun = pb - pa + 1, nsen = sb - sa + 1;
            if (__any(nrun > 0 || nsen > 0)) {
                const bool is_long = nrun > 3;
                if (nrun > 0 && !is_long) {
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        const int p = pa + t;
                        if (p <= pb && !(fold_tile && (dflag[p] & PP_DIRTY))) list_push(PK_BRIDGE, o, p);
                    }
                }
                unsigned long long m = __ballot(is_long);
                while (m) {
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int lpa = __builtin_amdgcn_readlane(pa, src), lpb = __builtin_amdgcn_readlane(pb, src),
                              lo = __builtin_amdgcn_readlane(o, src);
                    for (int p = lpa + lane; p <= lpb; p += 64)
                        if (!(fold_tile && (dflag[p] & PP_DIRTY))) list_push(PK_BRIDGE, lo, p);
                }
                for (int p = sa; p <= sb; p++)   // (frame border only) sentinel pieces: the chain path knows their typing
                    if (!(fold_tile && (dflag[p] & PP_DIRTY))) list_push(PK_BRIDGE, 0, p);
            }
        }
        // ---- fold tiles: points and forward segments over the pixels under reversed segments go into those pixels' lists
        if (fold_tile && !PP_DEV_IS(33)) {
            if (dirty) {
                const int s = dflag[q] & 0x7f;
                const unsigned idx = atomic_add_u16(dcnt, s, 1u) & 0xffu;
                if (idx < PT_KP) pts[s * PT_KP + idx] = (uint16_t)o;
                else hazard = true;
            }
            int p0 = 1, p1 = 0;
            if (has_p && x < xp) {
                const float f1 = floorf(xp);
                if (!(f1 < o0f || f0 > o1f - 1.0f)) {
                    p0 = f0 < o0f ? 0 : (int)f0 - o0;
                    p1 = f1 > o1f - 1.0f ? wt - 1 : (int)f1 - o0;
                    p0 = max(p0, dlo); p1 = min(p1, dhi);   // only the dirty stretch of the tile matters
                }
            }
            auto reg_seg = [&](int p, int oo) {
                const unsigned fl = dflag[p];
                if (fl & PP_DIRTY) {
                    const int s = fl & 0x7f;
                    const unsigned idx = (atomic_add_u16(dcnt, s, 0x100u) >> 8) & 0xffu;
                    if (idx < PT_KS) sgs[s * PT_KS + idx] = (uint16_t)oo;
                    else hazard = true;
                }
            };
            const bool seg_long = p1 - p0 > 3;
            if (__any(p1 >= p0)) {
                if (!seg_long) {
#pragma unroll
                    for (int t = 0; t < 4; t++) if (p0 + t <= p1) reg_seg(p0 + t, o);
                }
                unsigned long long m = __ballot(seg_long);
                while (m) {
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int lp0 = __builtin_amdgcn_readlane(p0, src), lp1 = __builtin_amdgcn_readlane(p1, src),
                              lo = __builtin_amdgcn_readlane(o, src);
                    for (int p = lp0 + lane; p <= lp1; p += 64) reg_seg(p, lo);
                }
            }
            // the left sentinel's segment
            if (left_edge && j == 0 && valid && (float)(-1.0 * w) < x) {
                const int e1 = min(min((int)floorf(x) - o0, wt - 1), dhi);
                if (!(floorf(x) < o0f)) for (int p = max(0, dlo); p <= e1; p++) reg_seg(p, 0);
            }
        }
    }
    __syncthreads();  // barrier 2: lists complete
    if (PP_DEV_IS(34) || PP_DEV_IS(33)) return;

    // =====================================================================================================
    // pass 2: the listed pixels, densely packed
    // =====================================================================================================
    const int nlist = min(flags[PF_NLIST], T);
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
            while (npr <= PT_KP && floorf(P[min(o1in + npr, npts - 1)].x) == colf) npr++;
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
                const float4 v = P[o];
                cx[k] = v.x; c0[k] = v.y; c1[k] = v.z; c2[k] = v.w;
                cj[k] = min(max(o - 1, 0), ns - 1);
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
        r8 = csm::f32_to_u8_wrap(color0); g8 = csm::f32_to_u8_wrap(color1); b8 = csm::f32_to_u8_wrap(color2);
        return chain;
    };
    for (int base = wave * 64; base < nlist; base += PP_THREADS) {
        const int i = base + lane;
        const bool act = i < nlist;
        const uint32_t e = plist[act ? i : 0];
        const int q = (int)(e & 0xfffu), o = (int)((e >> 12) & 0xffffu);
        const bool bridge = (e >> 28) == PK_BRIDGE;
        // bridge pixels of real segments away from the first two columns: one whole-pixel piece, closed-form constants
        const bool lean = act && bridge && o >= 1 && o + 1 <= npts - 2 && o0 + q >= 2;
        if (__any(lean)) {
            const float4 a = P[lean ? o : 1], b = P[lean ? o + 1 : 2];
            const float colf = (float)(o0 + q);
            const float center = colf + 0.5f;
            const float ip = div_core(center - a.x, b.x - a.x), om = 1.0f - ip;
            const float k0 = 0.5f + (a.y * om + b.y * ip) * sig_whole;
            const float k1 = 0.5f + (a.z * om + b.z * ip) * sig_whole;
            const float k2 = 0.5f + (a.w * om + b.w * ip) * sig_whole;
            // the segment is forward, starts left of the pixel and ends right of it (checked when it was listed)
            if (lean) emit(q, (int)k0 & 0xff, (int)k1 & 0xff, (int)k2 & 0xff);
        }
        const bool rest = act && !lean;
        if (__any(rest)) {
            int r8 = 0, g8 = 0, b8 = 0;
            const bool ok = eval_chain(rest, q, o, bridge, r8, g8, b8);
            if (rest && ok) emit(q, r8, g8, b8);
            hazard = hazard || (rest && !ok);
        }
    }
    // ---- general search over the pixels under reversed segments (first generation, eval_generic): the pixel's points
    // sorted by (x, id) == the reference's stable insertion sort inside the pixel; every listed segment tested per
    // sub-interval; with several (or no) active segments the largest interpolated |disparity| with 0 < ip < 1 wins,
    // ties are order-dependent -> row redo.
    for (int base = (3 - wave) * 64; base < (PP_DEV_IS(35) ? 0 : ndirty); base += PP_THREADS) {
        const int s = base + lane;
        bool pend = s < ndirty;
        const int q = dpix[pend ? s : 0];
        pend = pend && dflag[q] == (uint8_t)(PP_DIRTY | s);   // (a slot that lost its pixel to an overlapping reversed segment)
        const int col = o0 + q;
        const unsigned c = pend ? dcnt[s] : 0u;
        if ((c & 0xffu) > PT_KP || (c >> 8) > PT_KS) hazard = true;
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
                float x = k < np ? P[o].x : INFINITY;
#pragma unroll
                for (int m2 = 0; m2 <= k; m2++) {
                    const bool lt = x < xs[m2] || (x == xs[m2] && o < os[m2]);
                    const float tx = lt ? xs[m2] : x; const int to = lt ? os[m2] : o;
                    xs[m2] = lt ? x : xs[m2]; os[m2] = lt ? o : os[m2];
                    x = tx; o = to;
                }
            }
        }
        int so[PT_KS];
#pragma unroll
        for (int k = 0; k < PT_KS; k++) so[k] = (k < wns && k < nsg) ? (int)sgs[s * PT_KS + k] : -1;
        float color0 = 0.5f, color1 = 0.5f, color2 = 0.5f;
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
            if (live && (center < prev || center > colp1)) hazard = true;
            prev = live ? center : prev;
            const bool work = live && (sig64 ? C.sig_dd != 0.0 : sig_f != 0.0f);
            int nact = 0, pick = -1, nqual = 0, best = -1;
            float bc = (float)(-1e-7);
            bool tie = false;
#pragma unroll
            for (int e = 0; e < PT_KS; e++) {
                if (e < wns) {
                    const bool have = so[e] >= 0;
                    const int oe = have ? so[e] : 0;
                    const float e0 = P[oe].x, e1 = P[oe + 1].x;
                    const bool actv = have && (e0 < center) && !(e1 < center);
                    nact += actv ? 1 : 0;
                    pick = actv ? e : pick;
                    const float ip_e = (center - e0) / (e1 - e0);
                    const bool qual = actv && 0.0f < ip_e && ip_e < 1.0f;
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
            int o = 1;
#pragma unroll
            for (int e = 0; e < PT_KS; e++)
                if (e < wns) o = (e == pick && so[e] >= 0) ? so[e] : o;
            const float4 pl = P[contrib ? o : 1], pr = P[contrib ? o + 1 : 2];
            const float x0 = contrib ? pl.x : 0.0f, x1 = contrib ? pr.x : 1.0f;
            const int jl = min(max(o - 1, 0), ns - 1), jr = min(max(o, 0), ns - 1);
            const float ip_k = (center - x0) / (x1 - x0);
            const float om = 1.0f - ip_k;
            const float sgm = sig64 ? (float)C.sig_dd : sig_f;
            float n0 = color0 + (pl.y * om + pr.y * ip_k) * sgm;
            float n1 = color1 + (pl.z * om + pr.z * ip_k) * sgm;
            float n2 = color2 + (pl.w * om + pr.w * ip_k) * sgm;
            if (__any(contrib && jl == jr)) {  // segment inside one source pixel (sentinel pieces)
                if (jl == jr) {
                    if (sig64) {
                        n0 = (float)((double)color0 + (double)pl.y * C.sig_dd);
                        n1 = (float)((double)color1 + (double)pl.z * C.sig_dd);
                        n2 = (float)((double)color2 + (double)pl.w * C.sig_dd);
                    } else {
                        n0 = color0 + pl.y * sig_f;
                        n1 = color1 + pl.z * sig_f;
                        n2 = color2 + pl.w * sig_f;
                    }
                }
            }
            color0 = contrib ? n0 : color0;
            color1 = contrib ? n1 : color1;
            color2 = contrib ? n2 : color2;
        }
        if (pend) emit(q, csm::f32_to_u8_wrap(color0), csm::f32_to_u8_wrap(color1), csm::f32_to_u8_wrap(color2));
    }
    if (hazard) A.rowflag[(size_t)frame * h + row] = 1;  // the general kernel redoes this row (both eyes)
}

static size_t polypoint_lds(int S, int T, int KP, int KS) {
    const int nptmax = T + 2 * S + 4;
    const int npz = ((nptmax + 3) & ~3) > 128 ? ((nptmax + 3) & ~3) : 128;  // (the powf tables overlay pz: 512 bytes at least)
    return 1024 + 16 * (size_t)nptmax + 4 * (size_t)npz + 4 * (size_t)T + (size_t)((T + 3) & ~3) +
           2 * PP_DCAP * (2 + (size_t)KP + KS) + 4 * PF_WORDS + 64;
}

// Tile width for a row of `w` pixels with halo S: the staged range (T + 2S + 2 points) must fit the SLOTS * 256 point
// slots of a workgroup; equal tiles, multiples of 4.  0: the halo is too wide for this kernel.
static int polypoint_tile(int w, int S, int slots) {
    int tmax = (slots * PP_THREADS - 2 * S - 2) & ~3;
    if (tmax > 4092) tmax = 4092;  // 12-bit pixel field of the list entries
    if (tmax < 64) return 0;
    const int tiles = (w + tmax - 1) / tmax;
    int t = ((w + tiles - 1) / tiles + 3) & ~3;
    return t < 4 ? 4 : t;
}

int polypoint_max_halo() { return (3 * PP_THREADS - 2 - 64) / 2; }

// Launch for the eyes of `R` (SBS / TB / single-eye / uint8 outputs; no anaglyph).  `rowflag` must be zeroed by the caller;
// afterwards the general kernel is run over the flagged rows.
hipError_t launch_polypoint(const RowArgs& R, int S, uint8_t* rowflag, hipStream_t stream) {
    constexpr int SLOTS = 3, KP = 4, KS = 5;
    PolyPointArgs A;
    A.n = R.n; A.h = R.h; A.w = R.w; A.S = S;
    A.T = polypoint_tile(R.w, S, SLOTS);
    if (A.T == 0 || A.T + 2 * S + 2 > SLOTS * PP_THREADS - 0 || A.T + 2 * S + 4 >= 4096) return hipErrorInvalidValue;
    A.image_f32 = R.image_f32; A.image_u8 = R.image_u8;
    A.stats = R.stats; A.stats_rw = R.stats_rw;
    A.scale_from_stats = R.scale_from_stats;
    A.e32 = R.e32; A.conv32 = R.conv32;
    A.eye[0] = R.eye[0]; A.eye[1] = R.eye[1];
    A.single = R.neyes == 1 ? 0 : R.single;
    A.stereo_is_u8 = R.stereo_is_u8;
    A.out_u8 = R.out_u8; A.stereo = R.stereo; A.mask = R.mask; A.depth_l = R.depth_l; A.depth_r = R.depth_r;
    A.out_h = R.out_h; A.out_w = R.out_w;
    A.rowflag = rowflag;
    A.dbg = R.dbg;
    const int tiles = (A.w + A.T - 1) / A.T;
    dim3 grid(tiles * A.h, A.n, A.single >= 0 ? 1 : 2), block(PP_THREADS);
    const size_t lds = polypoint_lds(S, A.T, KP, KS);
    hipError_t e = hipFuncSetAttribute((const void*)k_polypoint<SLOTS, KP, KS, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_polypoint<SLOTS, KP, KS, 7>), grid, block, lds, stream, A.image_f32, A.eye[0].depth, A.eye[1].depth,
                       A.w, A.h, A.S, A.T, A.single, A);
    return hipGetLastError();
}

}  // namespace cs
