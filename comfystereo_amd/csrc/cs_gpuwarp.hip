// cs_gpuwarp.hip -- the 'gpu_warp' technique: forward_warp_gpu (reference
// stereoimage_generation.py:277-450) and its batched driver create_stereoimages_gpu (:1005-1128).
//
// The reference issues ~1000 full-frame ATen launches per eye pair (8 gather/scatter rounds, two
// cummax scans, grid_sample).  Here one workgroup owns one image row and keeps the whole state of
// the row (normalised depth, pixel offsets, z-buffer, inverse source map) in LDS:
//   * the 8 rounds: every adjacent pixel pair proposes a target column per round; torch's CPU scatter_ is sequential, so
//     the HIGHEST pair index targeting a column decides it, and a non-improving winner writes back what it gathered
//     (quirk Q3).  The deciding pair of column c in round k is the highest pair with floor(min(dl, dr)) == c - k: one LDS
//     atomicMax pass over the pairs serves all rounds, then one lane per column replays its 8 z-tests from registers.
//   * gap fill: "left nearest" = prefix-max scan; "right nearest" is the row's RIGHTMOST filled
//     column (quirk Q2) = one block-wide max.
//   * sampling: the bilinear grid_sample through the [-1,1] coordinate round trip (both axes), read
//     straight from the source image in HBM (rows y, y+1 are L2-resident neighbours).
// Gap mask: bit-exact for exponents 2 / 1 / 0.5 (torch.pow is exact there); colours are float32 with
// a stated tolerance against torch's vectorised bilinear kernel (tests).
#include "cs_common.h"
#include "cs_kernels.h"
#include <string.h>

namespace cs {

struct GwEye {
    const float* depth;  // [n][h][w]
    float div32, sep32;
    int enabled;
    int st_min, st_max, st_div;  // stats words: depth min / max (pre-division), divide-by-255 flag
    int xoff, yoff;
    int chan_mask;  // which channels of the output pixel this eye writes (bit c)
};

struct GwArgs {
    int n, h, w;
    const float* image;            // pixel (f, y, x, c) at f*img_sf + y*img_sy + x*img_sx + c*img_sc
    long long img_sf, img_sy, img_sx, img_sc;
    const uint32_t* stats;
    int scale_from_stats;
    int pow_mode;  // 0: x, 1: sqrt, 2: x*x, 3: x*x*x, 4: powf, 5: ones
    float e32, conv32;
    GwEye eye[2];
    int neyes;
    float* out;                    // pixel (f, y, x, c) at f*out_sf + y*out_sy + x*out_sx + c*out_sc
    long long out_sf, out_sy, out_sx, out_sc;
    uint8_t* mask_u8;  // plain: [n][h][w]
    float* mask_f32;   // node:  [n][h][w] (left | right)
    float* depth_l; float* depth_r;  // node: [n][h][w][3]
    int noclamp;
    int dbg;           // development (-DCS_DEV builds): phase cut-offs 51-54 of k_gpuwarp
    int mesh;          // 1: mesh-quality warp (k_meshwarp) instead of forward_warp_gpu's scatter rounds
    float grad_thr;    // gradient_threshold: mesh -- the triangle culling (reference :455, 1.5); scatter rounds -- connectivity (:339-340)
    int rounds;        // scatter rounds that can change a column: min(max_stretch, floor(gradient_threshold + 2) + 1)  (:365)
    uint8_t* keep;     // mesh: [neyes][groups][h-1][w-1] keep bits (bit 0 triangle A, bit 1 triangle B)
    int group;         // mesh: frames per group (the tensor forward_warp_mesh is handed)
    // lazy depth-blur tiles (cs_common.h lazy_select; rows of at most 2048 columns, scatter-round warp only): the blurred maps
    // only hold the tiles the map names, everything else is gray * (the frame's x255 scale)
    const uint32_t* tilemap; const float* gray; int tm_words;
    // (round 6) per-frame constants of k_gpuwarp_q, written by k_gpuwarp_flags: GWC_WORDS floats per frame (see there)
    const float* fconst;
    float lin_step;    // 2 / (h - 1): the step of torch.linspace(-1, 1, h) (IEEE division on the host)
};

enum { GWC_EYE = 4, GWC_YSX = 8, GWC_SCALE = 9, GWC_S255 = 10, GWC_WORDS = 16 };   // per-frame constants of k_gpuwarp_q (k_gpuwarp_flags writes them)
struct Px3 { float x, y, z; };
#ifdef CS_DEV
#define GW_DEV_IS(n) (A.dbg == (n))
#else
#define GW_DEV_IS(n) false
#endif

__constant__ csm::PowfTables c_gw_powf_tables = CS_POWF_TABLES_INIT;

__device__ __forceinline__ float torch_pow(float x, int mode, float e32, const csm::PowfTables* T) {
    switch (mode) {
    case 0: return x;
    case 1: return sqrtf(x);
    case 2: return x * x;
    case 3: return (x * x) * x;
    case 5: return 1.0f;
    default: return csm::powf_exact(x, e32, T);
    }
}

// a / b, correctly rounded: the FMA core of the IEEE expansion hipcc emits for `a / b` without its range scaling
// (v_div_scale / v_div_fixup), which only acts near the range limits.  Callers guarantee a == 0 or 2^-60 <= |a| < 2^60
// and 2^-40 <= |b| <= 2^40 (checked on the GPU over random operands by tools/ubench/issue_rate.hip).
__device__ __forceinline__ float gw_div_core(float a, float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, y0, 1.0f);
    const float y1 = __builtin_fmaf(e0, y0, y0);
    const float q0 = a * y1;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y1, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}
__device__ __forceinline__ float gw_rcp_refined(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
}
__device__ __forceinline__ float gw_div_with(float a, float b, float y1) {   // gw_div_core with the refined reciprocal of b supplied
    const float q0 = a * y1;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y1, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}
// inclusive prefix maximum over the 64 lanes of a wave: four row_shr steps inside the rows of 16, then lane 15 / 31 of the
// lower rows broadcast into the upper ones (DPP; lanes without a source keep their own value)
__device__ __forceinline__ int wave_incl_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ bool gw_core_ok(float a) { const float m = fabsf(a); return a == 0.0f || (m >= 0x1p-60f && m < 0x1p60f); }
// The division core for every lane, the full division only for lanes outside its proven range -- behind a branch that is almost
// never taken (an if / else of the two forms costs two more exec-mask levels per quotient; the kernel is sensitive to those)
__device__ __forceinline__ float gw_div(float a, float b) {
    float q = gw_div_core(a, b);
    if (__builtin_expect(!gw_core_ok(a), 0)) { asm volatile("" ::: "memory"); q = a / b; }
    return q;
}
__device__ __forceinline__ float gw_div_y(float a, float b, float y1, bool b_ok) {   // (y1: refined reciprocal of b, valid when b_ok)
    float q = gw_div_with(a, b, y1);
    if (__builtin_expect(!(b_ok && gw_core_ok(a)), 0)) { asm volatile("" ::: "memory"); q = a / b; }
    return q;
}

// MINW: waves per SIMD the register budget is sized for -- 8 (64 VGPRs, 8 spilled) lets two 1024-thread workgroups share a
// CU at 4K; the 512-thread workgroups of narrower frames run 6 per SIMD without spills (+5 % at 1080p)
// POW: 2 = exponent 2 compiled in (the widget default: x * x, no mode dispatch per pixel), -1 = A.pow_mode at run time
// GEN: forward_warp_gpu's keyword parameters away from their defaults (gradient_threshold 1.5, max_stretch 8; reference :277-279):
// connectivity threshold and the number of rounds at run time, 16 instead of 8 slots for the rounds' border columns.
// NODE: the node's layout compiled in (image and output interleaved, every enabled eye writes the three channels): the strides
// and channel masks of the general form do not occupy scalar registers (round 4: the kernel spilled ~150 scalars per eye)
template <int MINW, int POW, bool GEN = false, bool NODE = false>
__attribute__((amdgpu_waves_per_eu(MINW, MINW)))   // (an upper bound as well: the scalar register budget follows the MAXIMUM -- 80 at the default 10)
__global__ void __launch_bounds__(1024) k_gpuwarp(GwArgs A) {
    constexpr int NR = GEN ? 16 : 8;              // rounds the border arrays W0 / W1 and the offset of M provide for
    const int RV = GEN ? A.rounds : 4;            // rounds that can pass the z-test (see the column pass)
    const float conn_thr = GEN ? A.grad_thr : 1.5f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, lane = lane_id(), wave = wave_id();
    // Workgroup b runs on XCD b % 8 (observed dispatch order; a speed assumption only).  With row = blockIdx.x the rows y and
    // y + 1 -- whose image rows BOTH feed output row y through the 4-corner blend of the grid_sample round trip (:441-448) --
    // sit on different XCDs, and each fetches both rows from the fabric: 2.8 x the algorithmic reads (profiles/r03g_cfg4_pmc).
    // -DGW_XCD_ROWS (measured in round 4, NOT the default): XCD k takes the rows [k * rpx, (k + 1) * rpx) in order, so the
    // neighbour row is in that XCD's L2 -- FETCH_SIZE x 2 of the kernel 15.1 -> 9.0 GB per 128 1080p frames (-41 %), and the
    // kernel is 1.3 % SLOWER at 1080p and at 4K (tools/sessions/r04_s7.sh): it is not the fabric that bounds this kernel.
#ifdef GW_XCD_ROWS
    const int rpx = (A.h + 7) >> 3;
    const int y = (int)(blockIdx.x & 7u) * rpx + (int)(blockIdx.x >> 3);
    if (y >= A.h) return;
    const int frame = blockIdx.y, w = A.w, h = A.h;
#else
    const int y = blockIdx.x, frame = blockIdx.y, w = A.w, h = A.h;
#endif
    float* ndn = (float*)smem;     // normalised depth (not convergence-shifted); after the column pass: the left-nearest scan
    float* po = ndn + w;           // pixel offset; from the column pass on: the source map
    float* D = po + w;             // x + offset = dl of pair x = dr of pair x-1
    float* zb = D + w;             // z-buffer
    int* M = (int*)(zb + w);       // [w + NR] highest pair index per floor(min(dl, dr)) = -(NR - 1) .. w-2
    int* W0 = M + w + NR;          // [NR] per round: highest pair index clamped to column 0
    int* W1 = W0 + NR;             // [NR] per round: highest pair index clamped to column w-1
    int* winner = (int*)ndn;
    float* sm = po;
    // (round 5) bit rows, one bit per column: fbits = filled by this eye's column pass (the gap fill finds its left neighbour with
    // clz on them instead of two DPP prefix-maximum passes over every column), gbits = gap in some eye (the mask output)
    uint32_t* fbits = (uint32_t*)(W1 + NR);
    uint32_t* gbits = fbits + ((w + 31) >> 5);
    uint8_t* flags = (uint8_t*)fbits;
    int* ws = (int*)(flags + align16((size_t)w));
    csm::PowfTables* T = (csm::PowfTables*)(ws + 32);
    if (A.pow_mode == 4) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_gw_powf_tables);
        for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += nt) reinterpret_cast<uint32_t*>(T)[i] = src[i];
    }
    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
    for (int x = tid; x < 2 * ((w + 31) >> 5); x += nt) fbits[x] = 0u;
    __syncthreads();
    // torch.linspace(-1, 1, H)[y] (symmetric two-sided fill, each value one fused multiply-add: bit-equal to CPU torch for
    // every H probed, 48 .. 2160) and its unnormalisation
    float gy;
    {
        float step = h > 1 ? 2.0f / (float)(h - 1) : 0.0f;
        gy = y < h / 2 ? fmaf(step, (float)y, -1.0f) : fmaf(-step, (float)(h - y - 1), 1.0f);
    }
    float yy = (gy + 1.0f) * ((float)(h - 1) / 2.0f);
    yy = fminf(fmaxf(yy, 0.0f), (float)(h - 1));
    const float yn = floorf(yy);
    const float wn = yy - yn, wsth = 1.0f - wn;
    const int iy0 = (int)yn, iy1 = min(iy0 + 1, h - 1);
    const float sxw = (float)(w - 1);
    const bool sxw_ok = w >= 2 && w <= (1 << 20);   // the division core's denominator range
    const float ysx = sxw_ok ? gw_rcp_refined(sxw) : 0.0f;
    const bool interleaved = NODE || (A.img_sc == 1 && A.img_sx == 3 && A.out_sc == 1 && A.out_sx == 3);

    // lazy depth-blur tiles (cs_common.h): the row's selector of one eye -- `hi`: columns 2048 .. 4095, a second word of tile bits
    // (the other fields of a selector do not depend on s0 when the lane's offset is taken from column 0).  Set up per eye at the
    // top of its iteration (round 3 held both eyes' selectors from the kernel's prologue on, so that their scalar loads would
    // not stand in front of the depth loads: 14 scalar registers carried across the loop -- spilled; per eye is 0.5 % faster)
    const bool lazy = A.tilemap != nullptr;
    auto lazy_select_eye = [&](int e, int wv, LazySel& Zs, uint32_t& hi) {
        Zs.base = nullptr; Zs.bits = 0; Zs.delta = 0; Zs.mul_set = Zs.mul_clr = 0; hi = 0;
        if (!lazy) return;
        const char* grow = reinterpret_cast<const char*>(A.gray + ((size_t)frame * h + y) * wv);
        const char* de = reinterpret_cast<const char*>(A.eye[e].depth + ((size_t)frame * h + y) * wv);
        Zs = lazy_select(A.tilemap, A.tm_words, frame, h, y, 0, de, grow, st[ST_SCALE255]);
        if (wv > 2048) hi = lazy_select(A.tilemap, A.tm_words, frame, h, y, 2048, de + 4 * 2048, grow + 4 * 2048, st[ST_SCALE255]).bits;
    };
    // depth of column x through a selector whose tile bits are `lo` for columns < 2048 and `hi` above
    auto lazy_load2 = [](LazySel Zs, uint32_t hi, uint32_t x, float& mul) {
        Zs.bits = x >= 2048u ? hi : Zs.bits;
        return lazy_load(Zs, x, x, mul);
    };
    const int w_row = w;
    for (int e = 0; e < A.neyes; e++) {
        // (round 4) The row's width and the LDS base are re-read through an opaque move at the top of every eye: what the compiler
        // derives from them (the carve-up, a dozen hoisted LDS addresses, loop bounds) is then recomputed per eye with a few
        // scalar instructions instead of being carried across the loop -- with 80 scalar registers at 8 waves per SIMD it spilled
        // ~100 of them into vector lanes before the loop and read them back in every phase (v_writelane / v_readlane: vector issue
        // slots).  Kernel -4.4 % at 1080p (tools/sessions/r04_s18.sh); doing the same at every phase boundary bought nothing more (s19).
        int w;
        float *ndn, *po, *D, *zb, *sm, sxw;
        int *M, *W0, *W1, *winner, *ws;
        uint8_t* flags;
        uint32_t *fbits, *gbits;
        csm::PowfTables* T;
        auto reread = [&]() {
            int w_eye = w_row, lds0 = 0;   // (an opaque OFFSET: the pointer itself must keep its LDS address space)
            asm volatile("" : "+s"(w_eye), "+s"(lds0));
            w = w_eye;
            ndn = (float*)(smem + lds0); po = ndn + w; D = po + w; zb = D + w;
            M = (int*)(zb + w); W0 = M + w + NR; W1 = W0 + NR;
            winner = (int*)ndn; sm = po;
            flags = (uint8_t*)(W1 + NR);
            fbits = (uint32_t*)flags; gbits = fbits + ((w + 31) >> 5);
            ws = (int*)(flags + align16((size_t)w));
            T = (csm::PowfTables*)(ws + 32);
            sxw = (float)(w - 1);
        };
        reread();
        const GwEye& E = A.eye[e];
        if (!E.enabled) {
            // eye = source image (divergence < 0.001): plain copy into the slot
            if (NODE) {
                const Px3* srow = reinterpret_cast<const Px3*>(A.image + frame * A.img_sf + y * A.img_sy);
                Px3* orow = reinterpret_cast<Px3*>(A.out + frame * A.out_sf + (y + E.yoff) * A.out_sy) + E.xoff;
                if (A.out && E.chan_mask)
                    for (int x = tid; x < w; x += nt) orow[x] = srow[x];
            } else if (A.out)
                for (int x = tid; x < w; x += nt)
                    for (int c = 0; c < 3; c++)
                        if (E.chan_mask & (1 << c))
                            A.out[frame * A.out_sf + (y + E.yoff) * A.out_sy + (x + E.xoff) * A.out_sx + c * A.out_sc] =
                                A.image[frame * A.img_sf + y * A.img_sy + x * A.img_sx + c * A.img_sc];
            continue;
        }
#ifdef GW_P1_RFL
        const bool div255 = __builtin_amdgcn_readfirstlane((int)st[E.st_div]) != 0;
        float dmin = csm::ord2f((uint32_t)__builtin_amdgcn_readfirstlane((int)st[E.st_min])),
              dmax = csm::ord2f((uint32_t)__builtin_amdgcn_readfirstlane((int)st[E.st_max]));
#else
        const bool div255 = st[E.st_div] != 0;
        float dmin = csm::ord2f(st[E.st_min]), dmax = csm::ord2f(st[E.st_max]);
#endif
        if (div255) { dmin = dmin / 255.0f; dmax = dmax / 255.0f; }
        const float range = dmax - dmin;
        const float crange = fmaxf(range, (float)1e-6);
        const bool has_range = range > (float)1e-6;
        const bool crange_ok = crange < 0x1p40f;   // (>= 1e-6 by construction)
        const float* const img_row0 = A.image + frame * A.img_sf + iy0 * A.img_sy;
        const float* const img_row1 = A.image + frame * A.img_sf + iy1 * A.img_sy;
        float* const out_row = A.out + frame * A.out_sf + (y + E.yoff) * A.out_sy + E.xoff * (NODE ? 3 : A.out_sx);
        const float* drow = E.depth + ((size_t)frame * h + y) * w;
        LazySel Z;
        uint32_t Zhi;
        lazy_select_eye(e, w, Z, Zhi);
        // ---- pass 1: normalised depth, pixel offset, x + offset (:300-328); four columns per thread with their loads first
        const float yr = crange_ok ? gw_rcp_refined(crange) : 0.0f;   // several numerators over one denominator
        float* const depth_out = !A.depth_l ? nullptr : (e == 0 ? A.depth_l : A.depth_r) + (((size_t)frame * h + y) * w) * 3;
        for (int xb = tid; xb < w; xb += 4 * nt) {
            float dv[4], dm[4];
            // (round 5, measured one by one, tools/sessions/r05_s7.sh: separate loops for the lazy / plain loads -DGW_P1_SPLIT - 1.4 %,
            // the statistics words through readfirstlane -DGW_P1_RFL - 1.3 % -- fewer vector instructions, more scalar registers live
            // across the eye loop at the 80-SGPR budget; both stay off.  32-bit offsets for the depth-map stores: + 0.4 %, on)
#ifdef GW_P1_SPLIT
            if (lazy) {
#pragma unroll
                for (int u = 0; u < 4; u++) dv[u] = lazy_load2(Z, Zhi, (uint32_t)min(xb + u * nt, w - 1), dm[u]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    dm[u] = scale;
                    dv[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(drow) + 4u * (uint32_t)min(xb + u * nt, w - 1));
                }
            }
#else
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int x = min(xb + u * nt, w - 1);
                dm[u] = scale;
                dv[u] = lazy ? lazy_load2(Z, Zhi, (uint32_t)x, dm[u]) : drow[x];
            }
#endif
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int x = xb + u * nt;
                if (x >= w) continue;
                float v = dv[u] * dm[u];
                if (div255) {
                    asm volatile("" ::: "memory");   // (a real branch: the division is not worth speculating)
                    v = v / 255.0f;
                }
                if (depth_out) {   // this eye's depth-map output (see the end of the kernel)
                    const float vo = A.noclamp ? v : fminf(fmaxf(v, 0.0f), 1.0f);
                    // (scalar base + 32-bit lane offset: + 0.4 % alone, tools/sessions/r05_s7.sh)
                    *reinterpret_cast<Px3*>(reinterpret_cast<char*>(depth_out) + 12u * (uint32_t)x) = Px3{vo, vo, vo};
                }
                const float num = v - dmin;
                float nrm = gw_div_y(num, crange, yr, crange_ok);
                nrm = has_range ? nrm : 0.0f;
                ndn[x] = nrm;
                const float s = nrm - A.conv32;
                const float sg = s > 0.0f ? 1.0f : (s < 0.0f ? -1.0f : 0.0f);
                const float ax = fabsf(s);
                const float od = sg * (POW == 2 ? ax * ax : torch_pow(ax, A.pow_mode, A.e32, T));
                const float p = od * E.div32 + E.sep32;
                po[x] = p;
                D[x] = (float)x + p;
            }
        }
        for (int x = tid; x < w + 3 * NR; x += nt) M[x] = 0;   // (M, W0, W1 are contiguous; 0 = no pair)
        __syncthreads();
        if (GW_DEV_IS(51)) continue;
        // ---- the 8 scatter rounds (:330-391).  In round k the pair (i, i+1) targets column clamp(fs_i + k, 0, w-1),
        // fs_i = floor(min(dl, dr)); the HIGHEST pair index targeting a column decides it (sequential scatter_), and only
        // that pair's z-test can change the column.  Columns are independent of each other, and the deciding pair of an
        // interior column c in round k is the highest i with fs_i == c - k, whatever k: ONE atomic pass builds
        // M[v] = max{i : fs_i == v}; columns 0 and w-1 collect the clamped pairs per round (W0 / W1).  Then one lane per
        // column replays its rounds in order out of registers -- no barrier between rounds.
        // (round 5) The table holds 2 (i + 1) + connected: the order by pair index is kept, 0 means "no pair" (it reads as an
        // unconnected pair: no `>= 0` test, no clamp of the index in the column pass), and the connected bit travels with the
        // index instead of through a flag byte per pair (one LDS read-modify-write here and a read + two tests per round there).
        for (int i = tid; i < w - 1; i += nt) {
            {
                const float dl = D[i], dr = D[i + 1];
                const float fs = floorf(fminf(dl, dr));
                const bool connected = fabsf(po[i + 1] - po[i]) < conn_thr;
                const int key = 2 * (i + 1) + (connected ? 1 : 0);
                if (fs >= -(float)(NR - 1) && fs <= (float)(w - 2)) atomicMax(&M[(int)fs + NR - 1], key);
                if (!(fs > 0.0f) || fs + (float)(NR - 1) >= sxw) {
#pragma unroll
                    for (int k = 0; k < NR; k++) {
                        const float cfl = fs + (float)k;
                        if (!(cfl > 0.0f)) atomicMax(&W0[k], key);          // fmaxf(NaN, 0) == 0 as well
                        else if (cfl >= sxw) atomicMax(&W1[k], key);
                    }
                }
            }
        }
        __syncthreads();
        if (GW_DEV_IS(52)) continue;
        // Only rounds 0..3 can pass the z-test: a valid proposal needs connected (|po[i+1] - po[i]| < 1.5, so
        // dr - dl < 2.51) and 0 <= frac < 1, i.e. dl <= fs + k < dr -- with fs >= dl - 1 that leaves k <= 3; a
        // deciding pair that is not valid changes nothing (it writes back what it gathered, quirk Q3).
        // One proposal: table entry `key` (2 (i + 1) + connected; 0: no pair) for the column value cfl (== fs_i + k).
        auto propose = [&](int key, float cfl, float& z, float& src) {
            const int i = max((key >> 1) - 1, 0);
            const float dl = D[i], dr = D[i + 1];
            const float sw = dr - dl;
            const float safe = fabsf(sw) < (float)1e-4 ? 1.0f : sw;
            const float num = cfl - dl;
            // (the quotient of operands of opposite sign is negative, and |num| >= 1.001 |safe| rounds to >= 1: such a
            // proposal fails `frac >= 0 && frac < 1` whatever the rounding -- three of four do, no division for them)
            const bool maybe = (key & 1) && cfl >= 0.0f && cfl < (float)w &&
                               !((num < 0.0f && safe > 0.0f) || (num > 0.0f && safe < 0.0f) || fabsf(num) >= 1.001f * fabsf(safe));
            if (!maybe) return;
            // (here 1e-4 <= |safe| < 2.51 and |num| < 2.6: inside the division core's range unless num is tiny)
            const float frac = gw_div(num, safe);
            const bool valid = frac >= 0.0f && frac < 1.0f;
            const float iz = ndn[i] * (1.0f - frac) + ndn[i + 1] * frac;
            if (valid && iz > z + (float)1e-6) {
                z = iz;
                src = (float)i + frac;
            }
        };
        int myright = -1;
        for (int x = tid; x < w; x += nt) {
            float z = -1.0f, src = -1.0f;
            if (GEN && x > 0 && x < w - 1) {
                for (int k = 0; k < RV; k++) {   // (general parameters: the rounds one after the other)
                    const int key = M[x - k + NR - 1];
                    if (key) propose(key, (float)x, z, src);
                }
            } else if (x > 0 && x < w - 1) {
                // interior column: the deciding pair of round k has fs == x - k, so its column value is x itself.  The LDS
                // reads of the four rounds go out together (entry 0 = "no pair" reads pair 0's columns and fails the connected
                // bit): the rounds are only sequential in the z-test
                int kk[4];
                float dlk[4], drk[4];
#pragma unroll
                for (int k = 0; k < 4; k++) kk[k] = M[x - k + NR - 1];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // (byte offset of D[i], i = (key >> 1) - 1: (key >> 1) * 4 - 4; key == 0 reads D[-1] = po[w - 1]: harmless)
                    const float* dp = D + (kk[k] >> 1);
                    dlk[k] = dp[-1]; drk[k] = dp[0];
                }
                const float cfl = (float)x;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float sw = drk[k] - dlk[k];
                    const float safe = fabsf(sw) < (float)1e-4 ? 1.0f : sw;
                    const float num = cfl - dlk[k];
                    // (round 4: ONE test instead of a cascade of five exec-mask branches per round.  The filter only has to be
                    // conservative -- the exact test is `valid` below: a product below zero means operands of opposite sign, a
                    // product that underflows to zero is let through)
                    const bool maybe = ((kk[k] & 1) != 0) & !(num * safe < 0.0f) & !(fabsf(num) >= 1.001f * fabsf(safe));
                    if (maybe) {
                        const int i = (kk[k] >> 1) - 1;
                        const float frac = gw_div(num, safe);
                        const bool valid = frac >= 0.0f && frac < 1.0f;
                        const float iz = ndn[i] * (1.0f - frac) + ndn[i + 1] * frac;
                        if (valid && iz > z + (float)1e-6) {
                            z = iz;
                            src = (float)i + frac;
                        }
                    }
                }
            } else {
                // the clamped columns: the deciding pair's own fs + k decides whether it is in range at all
                for (int k = 0; k < RV; k++) {
                    const int key = x == 0 ? W0[k] : W1[k];
                    if (key) {
                        const int i = (key >> 1) - 1;
                        propose(key, floorf(fminf(D[i], D[i + 1])) + (float)k, z, src);
                    }
                }
            }
            const bool filled = !(src < 0.0f);
            if (filled) myright = max(myright, x);
            zb[x] = z;
            sm[x] = src;   // (po's storage: the offsets were last read by the pair pass)
            // the wave's 64 consecutive columns as bits (lanes beyond the row end are inactive: zero bits)
            const unsigned long long fb = __ballot(filled), gb = __ballot(!filled);
            if (lane == 0) {
                const int wi = x >> 5;
                fbits[wi] = (uint32_t)fb; gbits[wi] |= (uint32_t)gb;
                if (x + 32 < w) { fbits[wi + 1] = (uint32_t)(fb >> 32); gbits[wi + 1] |= (uint32_t)(gb >> 32); }
            }
        }
        if (GW_DEV_IS(53)) { __syncthreads(); continue; }
        // gap fill (:393-438): "left nearest" = prefix max of the filled columns, "right nearest" = the row's RIGHTMOST
        // filled column (quirk Q2)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) myright = max(myright, __shfl_xor(myright, off));
        if (lane == 0) ws[16 + wave] = myright;
        __syncthreads();   // (every read of ndn is done: its storage takes the scan)
        int rightmost = -1;
        for (int i = 0; i < (nt >> 6); i++) rightmost = max(rightmost, ws[16 + i]);
        // "left nearest filled" of a gap pixel = the highest set bit of `fbits` below it: one or two words and a clz, only for the
        // gap pixels (rounds 3-4: two DPP prefix-maximum passes over every column of the row, 596 of the kernel's 2 970 vector
        // instructions per wave, profiles/r05_s5/phases.txt; rounds 2-3: every gap pixel walked left over the source map)
        auto left_filled = [&](int x) {
            if (x <= 0) return -1;
            int wi = (x - 1) >> 5;
            uint32_t cur = fbits[wi] & (0xffffffffu >> (31 - ((x - 1) & 31)));
            while (true) {
                if (cur) return wi * 32 + 31 - __clz((int)cur);
                if (--wi < 0) return -1;
                cur = fbits[wi];
            }
        };
        if (GW_DEV_IS(54)) { __syncthreads(); continue; }
        if (GW_DEV_IS(60)) {   // (development: the row's intermediate state instead of colours)
            for (int x = tid; x < w; x += nt) *reinterpret_cast<Px3*>(reinterpret_cast<char*>(out_row) + 12u * (uint32_t)x) = Px3{((float*)winner)[x], D[x], sm[x]};
            __syncthreads(); continue;
        }
        // final source position of column x (gap fill :393-438), then the bilinear taps of the grid_sample round trip (:440-448)
        struct Taps { int ix0, ix1; float nw, ne, sw2, se; };
        auto taps_of = [&](int x) {
            float s = sm[x];
            if (s < 0.0f) {
                int left = left_filled(x);
                int right = rightmost >= x ? rightmost : -1;
                bool hl = left >= 0, hr = right >= 0;
                int li = hl ? left : 0, ri = hr ? right : 0;
                float lsrc = sm[li], rsrc = sm[ri], lz = zb[li], rz = zb[ri];
                float ld = (float)(x - left), rd = (float)(right - x);
                float tot = fmaxf(ld + rd, 1.0f);
                float t = ld / tot;
                if (!hl) t = 1.0f;
                if (!hr) t = 0.0f;
                float tb = (lz < rz) ? sqrtf(t) : 1.0f - sqrtf(1.0f - t);
                float g = lsrc * (1.0f - tb) + rsrc * tb;
                if (hl || hr) s = g;
            }
            const float pos = fminf(fmaxf(s, 0.0f), sxw);
            const float p2 = pos * 2.0f;
            float gx = gw_div_y(p2, sxw, ysx, sxw_ok) - 1.0f;
            float xx = (gx + 1.0f) * (sxw / 2.0f);
            xx = fminf(fmaxf(xx, 0.0f), sxw);
            const float xw = floorf(xx);
            const float ww = xx - xw, we = 1.0f - ww;
            Taps t;
            t.ix0 = (int)xw; t.ix1 = min(t.ix0 + 1, w - 1);
            t.nw = wsth * we; t.ne = wsth * ww; t.sw2 = wn * we; t.se = wn * ww;
            return t;
        };
        if (NODE || (interleaved && E.chan_mask == 7)) {   // node layout, all channels: 12-byte accesses, 32-bit offsets, two columns
            const char* const r0 = reinterpret_cast<const char*>(img_row0);   // per step so that eight loads are in flight
            const char* const r1 = reinterpret_cast<const char*>(img_row1);
            constexpr bool TWO = MINW < 8;   // (the 64-register instantiation spills with two columns in flight: -3 % at 4K)
            for (int xb = tid; xb < w; xb += (TWO ? 2 : 1) * nt) {
                const int x1 = xb + nt;
                const bool two = TWO && x1 < w;
                const Taps t0 = taps_of(xb), t1 = taps_of(two ? x1 : xb);
                const uint32_t a0 = 12u * (uint32_t)t0.ix0, a1 = 12u * (uint32_t)t0.ix1, b0 = 12u * (uint32_t)t1.ix0, b1 = 12u * (uint32_t)t1.ix1;
                const Px3 pa = *reinterpret_cast<const Px3*>(r0 + a0), pb = *reinterpret_cast<const Px3*>(r0 + a1),
                          pc = *reinterpret_cast<const Px3*>(r1 + a0), pd = *reinterpret_cast<const Px3*>(r1 + a1);
                const Px3 qa = *reinterpret_cast<const Px3*>(r0 + b0), qb = *reinterpret_cast<const Px3*>(r0 + b1),
                          qc = *reinterpret_cast<const Px3*>(r1 + b0), qd = *reinterpret_cast<const Px3*>(r1 + b1);
                Px3 r;
                r.x = pa.x * t0.nw + pb.x * t0.ne + pc.x * t0.sw2 + pd.x * t0.se;
                r.y = pa.y * t0.nw + pb.y * t0.ne + pc.y * t0.sw2 + pd.y * t0.se;
                r.z = pa.z * t0.nw + pb.z * t0.ne + pc.z * t0.sw2 + pd.z * t0.se;
                *reinterpret_cast<Px3*>(reinterpret_cast<char*>(out_row) + 12u * (uint32_t)xb) = r;
                if (two) {
                    r.x = qa.x * t1.nw + qb.x * t1.ne + qc.x * t1.sw2 + qd.x * t1.se;
                    r.y = qa.y * t1.nw + qb.y * t1.ne + qc.y * t1.sw2 + qd.y * t1.se;
                    r.z = qa.z * t1.nw + qb.z * t1.ne + qc.z * t1.sw2 + qd.z * t1.se;
                    *reinterpret_cast<Px3*>(reinterpret_cast<char*>(out_row) + 12u * (uint32_t)x1) = r;
                }
            }
        } else if (!NODE) {
            for (int x = tid; x < w; x += nt) {
                const Taps t = taps_of(x);
                const float* p00 = img_row0 + (size_t)t.ix0 * A.img_sx;
                const float* p01 = img_row0 + (size_t)t.ix1 * A.img_sx;
                const float* p10 = img_row1 + (size_t)t.ix0 * A.img_sx;
                const float* p11 = img_row1 + (size_t)t.ix1 * A.img_sx;
                float* o = out_row + (size_t)x * A.out_sx;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (!(E.chan_mask & (1 << c))) continue;
                    o[c * A.out_sc] = p00[c * A.img_sc] * t.nw + p01[c * A.img_sc] * t.ne + p10[c * A.img_sc] * t.sw2 + p11[c * A.img_sc] * t.se;
                }
            }
        }
        __syncthreads();
    }
    if (A.mask_u8)
        for (int x = tid; x < w; x += nt) A.mask_u8[((size_t)frame * h + y) * w + x] = (uint8_t)((gbits[x >> 5] >> (x & 31)) & 1u);
    if (A.mask_f32)
        for (int x = tid; x < w; x += nt) A.mask_f32[((size_t)frame * h + y) * w + x] = ((gbits[x >> 5] >> (x & 31)) & 1u) ? 1.0f : 0.0f;
    if (A.depth_l) {
        // left_depth / 255 if its (sub-batch) max > 1 (:1125-1126), clamp(0,1), 3 channels (GenerateStereo.py:165-168).
        // Eyes that ran the warp wrote theirs in pass 1, where the depth is in registers anyway (and the stores overlap
        // with the rest of the row's work); here: the eyes the loop above skipped (eye = source image, or a single-eye call)
        for (int e = 0; e < 2; e++) {
            const GwEye& E = A.eye[e];
            if (e < A.neyes && E.enabled) continue;
            const bool div255 = st[E.st_div] != 0;
            const float* drow = E.depth + ((size_t)frame * h + y) * w;
            float* dst = (e == 0 ? A.depth_l : A.depth_r) + (((size_t)frame * h + y) * w) * 3;
            LazySel Z;
            uint32_t Zhi;
            lazy_select_eye(e, w, Z, Zhi);
            for (int xb = tid; xb < w; xb += 4 * nt) {   // (loads first; /255 behind a real branch)
                float dv[4], dm[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int x = min(xb + u * nt, w - 1);
                    dm[u] = scale;
                    dv[u] = A.tilemap ? lazy_load2(Z, Zhi, (uint32_t)x, dm[u]) : drow[x];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int x = xb + u * nt;
                    if (x >= w) continue;
                    float v = dv[u] * dm[u];
                    if (div255) {
                        asm volatile("" ::: "memory");
                        v = v / 255.0f;
                    }
                    if (!A.noclamp) v = fminf(fmaxf(v, 0.0f), 1.0f);
                    *reinterpret_cast<Px3*>(dst + 3 * x) = Px3{v, v, v};
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_gpuwarp_q (round 6): the same warp for the node's layout and forward_warp_gpu's default parameters, FOUR CONTIGUOUS
// COLUMNS PER LANE.  k_gpuwarp above is bound by its vector instructions (2 599 per wave at 1080p, the vector pipes ~80 % busy,
// profiles/r05d_cfg4_pmc), a third of them address arithmetic, loop control and single-dword LDS / global accesses of its
// column-strided loops (lane t: columns t, t + nt, ...).  Here lane t owns the columns 4t .. 4t + 3 in every phase:
//   * stage + pair pass in ONE phase: depth as one 16-byte load, ndn / D as one 16-byte LDS store each, the depth-map output as
//     three 16-byte stores; the pixel offsets never go to LDS -- the pairs (i, i + 1) of a lane's columns take their operands
//     from registers, the fourth from the next lane (DPP wave_shl:1).  A wave covers 63 groups and lane 63 stages the group that
//     lane 0 of the next wave owns (the same values written twice), so that no pair crosses a wave without both operands: one
//     barrier and a pass over LDS less per eye;
//   * column pass: the seven table entries the four rounds of four neighbouring columns can name (M[x0 - 3 .. x0 + 3]) are read
//     as two 16-byte LDS loads and decoded ONCE (pair index, dl, dr, the guarded width) instead of once per column and round;
//   * sampling: rows whose vertical blend weight is EXACTLY zero (the grid_sample round trip returns the integer row: 69 % of the
//     rows at 1080p and 4K) read one image row instead of two -- x + 0 * p == x for every finite pixel p (non-finite pixels of
//     the unweighted row do not propagate here; the reference would turn the column into NaN) -- and the four columns of a lane
//     leave as three 16-byte stores.
// Arithmetic: the expressions of k_gpuwarp, operation for operation (same tests: masks bit-exact, colours as before).
// Requirements (gw_launch falls back to k_gpuwarp otherwise): node layout, both eyes enabled, w % 4 == 0, w >= 8, 16-byte aligned
// buffers, default gradient_threshold / max_stretch.
// ---------------------------------------------------------------------------------------------------------------------
struct GwQuad { float x, y, z, w; };
// a scheduling fence between the columns of a lane: the loads of the next column are not hoisted above the arithmetic of this one
// (four columns' taps in flight do not fit the 64 registers of the 8-waves-per-SIMD instantiation)
#ifndef GWQ_NO_FENCE
#define GWQ_FENCE() asm volatile("" ::: "memory")
#else
#define GWQ_FENCE()
#endif
static size_t gwq_bit_words(int w) { return (size_t)((w + 127) >> 7) << 2; }   // one bit per column, padded to 16 bytes
static size_t gwq_lds_bytes(int w) { return 4 * (size_t)w * 4 + ((size_t)w + 24) * 4 + 2 * gwq_bit_words(w) * 4 + 32 * 4 + sizeof(csm::PowfTables) + 64; }

// LDS carve-up of k_gpuwarp_q, re-derived at the top of EVERY phase from an opaque copy of the width: what the compiler hoists out
// of the loops (a dozen LDS addresses, loop bounds, lane masks) is then recomputed per phase with a few scalar instructions instead of
// being carried across the kernel -- it held ~170 scalars live and spilled 90 of them into vector lanes (v_writelane / v_readlane are
// vector instructions: 50 in the prologue alone)
struct GwqLds {
    float *ndn, *D, *zb, *sm;   // [w] each: normalised depth (not convergence-shifted); x + offset (D[-1] is readable: table entry 0); z-buffer; source map
    int *M, *W0, *W1;           // [w + NR] highest (2 (i + 1) + connected) per floor(min(dl, dr)) = -(NR - 1) .. w - 2; [NR] each: the pairs clamped to column 0 / w - 1, per round
    uint32_t *fbits, *gbits;    // one bit per column: filled by this eye's column pass; gap in some eye (the mask output)
    int* ws;
    csm::PowfTables* T;
    int w, nbw, frame, y;
};
__device__ __forceinline__ GwqLds gwq_carve(char* smem, int w_arg) {
    constexpr int NR = 8;
    GwqLds L;
    int w = w_arg, lds0 = 0, fr = blockIdx.y, yy = blockIdx.x;   // (an opaque OFFSET: the pointer itself must keep its LDS address space)
    asm volatile("" : "+s"(w), "+s"(lds0), "+s"(fr), "+s"(yy));
    L.w = w; L.nbw = ((w + 127) >> 7) << 2; L.frame = fr; L.y = yy;
    L.ndn = (float*)(smem + lds0); L.D = L.ndn + w; L.zb = L.D + w; L.sm = L.zb + w;
    L.M = (int*)(L.sm + w); L.W0 = L.M + w + NR; L.W1 = L.W0 + NR;
    L.fbits = (uint32_t*)(L.W1 + NR); L.gbits = L.fbits + L.nbw;
    L.ws = (int*)(L.gbits + L.nbw);
    L.T = (csm::PowfTables*)(L.ws + 32);
    return L;
}

template <int MINW, int POW>
__attribute__((amdgpu_waves_per_eu(MINW, MINW)))
__global__ void __launch_bounds__(1024) k_gpuwarp_q(GwArgs A) {
    constexpr int NR = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, lane = lane_id(), wave = wave_id(), nwaves = nt >> 6;
    const int h = A.h;
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    float wn, wsth, sxw_half, ysx, scale;
    uint32_t s255;
    int iy0;
    {
        const GwqLds L = gwq_carve(smem, A.w);
        const int w = L.w, y = L.y;
        if (POW != 2 && A.pow_mode == 4) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_gw_powf_tables);
            for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += nt) reinterpret_cast<uint32_t*>(L.T)[i] = src[i];
        }
        // the frame's constants (k_gpuwarp_flags), by SCALAR loads: they are wave-uniform, and a vector load + readfirstlane per
        // value -- what the compiler makes of a load it cannot prove unclobbered -- is vector issue this kernel does not have to spare
        // (one 32-bit output per value: a 4-vector asm output whose elements are bit-cast to float is miscompiled by this hipcc -- the
        // constant-bus legalisation of an instruction with two of them reads element 0 twice; tools/sessions/r06_s7*.py found it)
        asm volatile("s_load_dword %0, %3, 0x20\n\ts_load_dword %1, %3, 0x24\n\ts_load_dword %2, %3, 0x28\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(ysx), "=&s"(scale), "=&s"(s255) : "s"(A.fconst + (size_t)L.frame * GWC_WORDS) : "memory");
        for (int q = tid; q < ((w + 3 * NR) >> 2) + (2 * L.nbw >> 2); q += nt) reinterpret_cast<int4*>(L.M)[q] = make_int4(0, 0, 0, 0);   // M, W0, W1, fbits, gbits
        // torch.linspace(-1, 1, H)[y] and its unnormalisation (as in k_gpuwarp; the step is an IEEE division on the host)
        const float step = A.lin_step;
        const float gy = y < h / 2 ? fmaf(step, (float)y, -1.0f) : fmaf(-step, (float)(h - y - 1), 1.0f);
        float yy = (gy + 1.0f) * ((float)(h - 1) / 2.0f);
        yy = fminf(fmaxf(yy, 0.0f), (float)(h - 1));
        const float yn = floorf(yy);
        wn = uni(yy - yn); wsth = uni(1.0f - wn);
        iy0 = __builtin_amdgcn_readfirstlane((int)yn);
        sxw_half = uni((float)(w - 1) / 2.0f);
    }
    const bool lazy = A.tilemap != nullptr;
    const bool one_row = wn == 0.0f;      // the second image row of the blend has weight exactly zero
    __syncthreads();

    for (int e = 0; e < 2; e++) {
        const GwEye& E = A.eye[e];
      {
        const GwqLds L = gwq_carve(smem, A.w);
        const int w = L.w, frame = L.frame, y = L.y;
        float* const ndn = L.ndn; float* const D = L.D; int* const M = L.M; int* const W0 = L.W0; int* const W1 = L.W1;
        csm::PowfTables* const T = L.T;
        const float sxw = (float)(w - 1);
        float dmin, crange, yr;
        uint32_t eflags;
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x4\n\ts_load_dword %2, %4, 0x8\n\ts_load_dword %3, %4, 0xc\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(dmin), "=&s"(crange), "=&s"(yr), "=&s"(eflags) : "s"(A.fconst + (size_t)frame * GWC_WORDS + GWC_EYE * e) : "memory");
        const bool has_range = (eflags & 1u) != 0, crange_ok = (eflags & 2u) != 0, div255 = (eflags & 4u) != 0;
        const char* const drow = reinterpret_cast<const char*>(E.depth + ((size_t)frame * h + y) * w);
        char* const depth_out = reinterpret_cast<char*>((e == 0 ? A.depth_l : A.depth_r) + (((size_t)frame * h + y) * w) * 3);
        LazySel Z;
        uint32_t Zhi = 0;
        Z.base = nullptr; Z.bits = 0; Z.delta = 0; Z.mul_set = Z.mul_clr = 0;
        if (lazy) {
            const char* grow = reinterpret_cast<const char*>(A.gray + ((size_t)frame * h + y) * w);
            Z = lazy_select(A.tilemap, A.tm_words, frame, h, y, 0, drow, grow, s255);
            if (w > 2048) Zhi = lazy_select(A.tilemap, A.tm_words, frame, h, y, 2048, drow + 4 * 2048, grow + 4 * 2048, s255).bits;
        }
        // ---- stage (:300-328) + the pair pass of the scatter rounds (:330-391, see k_gpuwarp): M[v] = max{2 (i + 1) + connected : fs_i == v}
        // Global accesses stay column-strided (lane = column inside a block of 64: a wave's load or store covers whole lines; with the
        // quad layout of the column pass every instruction touched four times as many lines, and the kernel ran 57 % slower,
        // tools/sessions/r06_s2.sh).  A wave stages 256 consecutive columns as four blocks and owns the pairs of the first 252: the
        // right operand of a pair is the next lane's column (DPP wave_shl:1), for lane 63 lane 0 of the wave's next block; the last
        // four columns are staged again by the next wave (same values), so no pair crosses a wave.
        for (int cb = wave * 252; cb < w; cb += nwaves * 252) {
            float p[4], d[4];
            float dv[4], dm[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t x = (uint32_t)min(cb + 64 * u + lane, w - 1);
                dm[u] = scale;
                if (lazy) {
                    LazySel Zs = Z;
                    Zs.bits = x >= 2048u ? Zhi : Z.bits;
                    dv[u] = lazy_load(Zs, x, x, dm[u]);
                } else
                    dv[u] = *reinterpret_cast<const float*>(drow + 4u * x);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int x = cb + 64 * u + lane;
                float v = dv[u] * dm[u];
                if (div255) {
                    asm volatile("" ::: "memory");   // (a real branch: the division is not worth speculating)
                    v = v / 255.0f;
                }
                const float num = v - dmin;
                float nrm = gw_div_y(num, crange, yr, crange_ok);
                nrm = has_range ? nrm : 0.0f;
                const float s = nrm - A.conv32;
                float od;
                if (POW == 2) od = s * fabsf(s);   // == sign(s) * (|s| * |s|): a product rounds independently of its sign
                else {
                    const float sg = s > 0.0f ? 1.0f : (s < 0.0f ? -1.0f : 0.0f);
                    od = sg * torch_pow(fabsf(s), A.pow_mode, A.e32, T);
                }
                p[u] = od * E.div32 + E.sep32;
                d[u] = (float)x + p[u];
                if (x < w) {
                    ndn[x] = nrm;
                    D[x] = d[u];
                    const float vo = A.noclamp ? v : fminf(fmaxf(v, 0.0f), 1.0f);
                    *reinterpret_cast<Px3*>(depth_out + 12u * (uint32_t)x) = Px3{vo, vo, vo};
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                // the pair's right column: the next lane's; lane 63: lane 0 of the wave's next block (block 3: not owned)
                float dr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d[u]), 0x130, 0xf, 0xf, false));
                float pr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p[u]), 0x130, 0xf, 0xf, false));
                if (u < 3) {
                    const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d[u < 3 ? u + 1 : 3]), 0));
                    const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p[u < 3 ? u + 1 : 3]), 0));
                    dr = lane == 63 ? d0 : dr; pr = lane == 63 ? p0 : pr;
                }
                const int i = cb + 64 * u + lane;
                if (i < w - 1 && (u < 3 || lane < 60)) {
                    const float dl = d[u], pl = p[u];
                    const float fs = floorf(fminf(dl, dr));
                    const bool connected = fabsf(pr - pl) < 1.5f;
                    const int key = 2 * (i + 1) + (connected ? 1 : 0);
                    if (fs >= -(float)(NR - 1) && fs <= (float)(w - 2)) atomicMax(&M[(int)fs + NR - 1], key);
                    if (!(fs > 0.0f) || fs + (float)(NR - 1) >= sxw) {   // (the first and the last columns of a row only: a rolled loop)
#pragma unroll 1
                        for (int k = 0; k < NR; k++) {
                            const float cfl = fs + (float)k;
                            if (!(cfl > 0.0f)) atomicMax(&W0[k], key);          // fmaxf(NaN, 0) == 0 as well
                            else if (cfl >= sxw) atomicMax(&W1[k], key);
                        }
                    }
                }
            }
        }
      }
        __syncthreads();
        if (GW_DEV_IS(52)) { __syncthreads(); __syncthreads(); continue; }
        // ---- column pass: the z-tests of rounds 0 .. 3 in order (see k_gpuwarp), four columns per lane
      {
        const GwqLds L = gwq_carve(smem, A.w);
        const int w = L.w, G = w >> 2;
        float* const ndn = L.ndn; float* const D = L.D; float* const zb = L.zb; float* const sm = L.sm;
        int* const M = L.M; int* const W0 = L.W0; int* const W1 = L.W1;
        auto propose = [&](int key, float cfl, float& z, float& src) {   // the clamped columns (k_gpuwarp's form)
            const int i = max((key >> 1) - 1, 0);
            const float dl = D[i], dr = D[i + 1];
            const float sw = dr - dl;
            const float safe = fabsf(sw) < (float)1e-4 ? 1.0f : sw;
            const float num = cfl - dl;
            const bool maybe = (key & 1) && cfl >= 0.0f && cfl < (float)w &&
                               !((num < 0.0f && safe > 0.0f) || (num > 0.0f && safe < 0.0f) || fabsf(num) >= 1.001f * fabsf(safe));
            if (!maybe) return;
            const float frac = gw_div(num, safe);
            const bool valid = frac >= 0.0f && frac < 1.0f;
            const float iz = ndn[i] * (1.0f - frac) + ndn[i + 1] * frac;
            if (valid && iz > z + (float)1e-6) {
                z = iz;
                src = (float)i + frac;
            }
        };
        int myright = -1;
        for (int g = tid; g < G; g += nt) {
            const int x0 = 4 * g;
            // entry j: the deciding pair with fs == x0 - 3 + j; column x0 + u meets it in round k = u + 3 - j
            const int4 e0 = *reinterpret_cast<const int4*>(M + x0 + 4), e1 = *reinterpret_cast<const int4*>(M + x0 + 8);
            const int key[7] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z};
            // decoded once per entry: dl and the guarded width -- ZERO for an unconnected pair or no pair (a real `safe` is never 0:
            // widths below 1e-4 become 1), which fails the pre-test below for every numerator.
            // (The clamped columns 0 and w - 1 run through the same code; their result is replaced below.)
            float dl[7], safe[7];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                // (entry 0 = "no pair" reads D[-1], D[0])
                const float* dp = D + (key[j] >> 1);
                const float a = dp[-1], b = dp[0];
                const float sw = b - a;
                dl[j] = a;
                const float sf = fabsf(sw) < (float)1e-4 ? 1.0f : sw;
                safe[j] = (key[j] & 1) ? sf : 0.0f;
            }
            float z[4], src[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int x = x0 + u;
                z[u] = -1.0f; src[u] = -1.0f;
                const float cfl = (float)x;
                // EXACT pre-test of a proposal (interior columns: x >= 1, so the numerator x - dl is 0 or at least 2^-24 in magnitude -- no
                // underflow in the quotient): frac = RN(num / safe) lies in [0, 1) iff num is zero or has safe's sign, and |num| < |safe|
                // (the quotient of two floats with |num| < |safe| is at most 1 - 2^-24: it does not round up to 1).  An unconnected pair
                // has safe == 0 and fails; NaN fails like the reference's comparison.
                float num[4];
                bool m[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = u + 3 - k;
                    num[k] = cfl - dl[j];
                    const bool opposite = (int)(__builtin_bit_cast(uint32_t, num[k]) ^ __builtin_bit_cast(uint32_t, safe[j])) < 0 && num[k] != 0.0f;
                    m[k] = !opposite & (fabsf(num[k]) < fabsf(safe[j]));
                }
                // One valid proposal per column is the rule (two: a fold -- foreground and background both cover the column); then
                // the rounds need no order: select the proposal per lane and divide ONCE, with every lane of the wave busy, instead of
                // once per round that has a taker somewhere in the wave (9 of the 16 round blocks of a lane's four columns ran).
                const bool two = (m[0] & (m[1] | m[2] | m[3])) | (m[1] & (m[2] | m[3])) | (m[2] & m[3]);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(two) != 0, 0)) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {   // the rounds in order (k_gpuwarp's form)
                        const int j = u + 3 - k;
                        if (m[k]) {
                            const int i = (key[j] >> 1) - 1;
                            const float frac = gw_div(num[k], safe[j]);
                            const bool valid = frac >= 0.0f && frac < 1.0f;
                            const float iz = ndn[i] * (1.0f - frac) + ndn[i + 1] * frac;
                            if (valid && iz > z[u] + (float)1e-6) {
                                z[u] = iz;
                                src[u] = (float)i + frac;
                            }
                        }
                    }
                } else {
                    const float nsel = m[0] ? num[0] : (m[1] ? num[1] : (m[2] ? num[2] : num[3]));
                    const float ssel = m[0] ? safe[u + 3] : (m[1] ? safe[u + 2] : (m[2] ? safe[u + 1] : safe[u]));
                    const int ksel = m[0] ? key[u + 3] : (m[1] ? key[u + 2] : (m[2] ? key[u + 1] : key[u]));
                    if (m[0] | m[1] | m[2] | m[3]) {
                        const int i = (ksel >> 1) - 1;
                        const float frac = gw_div(nsel, ssel);
                        const bool valid = frac >= 0.0f && frac < 1.0f;
                        const float iz = ndn[i] * (1.0f - frac) + ndn[i + 1] * frac;
                        if (valid && iz > -1.0f + (float)1e-6) {
                            z[u] = iz;
                            src[u] = (float)i + frac;
                        }
                    }
                }
            }
            if (g == 0 || g == G - 1) {   // the clamped columns 0 and w - 1: the deciding pair's own fs + k decides whether it is in range at all
                float zz = -1.0f, ss = -1.0f;
                for (int k = 0; k < 4; k++) {
                    const int kk = g == 0 ? W0[k] : W1[k];
                    if (kk) {
                        const int i = (kk >> 1) - 1;
                        propose(kk, floorf(fminf(D[i], D[i + 1])) + (float)k, zz, ss);
                    }
                }
                if (g == 0) { z[0] = zz; src[0] = ss; } else { z[3] = zz; src[3] = ss; }
            }
            *reinterpret_cast<GwQuad*>(zb + x0) = GwQuad{z[0], z[1], z[2], z[3]};
            *reinterpret_cast<GwQuad*>(sm + x0) = GwQuad{src[0], src[1], src[2], src[3]};
            uint32_t nib = 0;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (!(src[u] < 0.0f)) { nib |= 1u << u; myright = x0 + u; }
            const uint32_t sh = (uint32_t)x0 & 31u;
            if (nib) atomicOr(&L.fbits[x0 >> 5], nib << sh);
            if (nib != 15u) atomicOr(&L.gbits[x0 >> 5], (nib ^ 15u) << sh);
        }
        if (GW_DEV_IS(53)) { __syncthreads(); __syncthreads(); continue; }
        // the row's rightmost filled column: maximum over the wave (DPP), then over the waves
        myright = wave_incl_max(myright);
        if (lane == 63) L.ws[16 + wave] = myright;
      }
        __syncthreads();
      {
        const GwqLds L = gwq_carve(smem, A.w);
        const int w = L.w, frame = L.frame, y = L.y;
        const float* const zb = L.zb; const float* const sm = L.sm; const uint32_t* const fbits = L.fbits;
        const int rightmost = __builtin_amdgcn_readlane(wave_incl_max(lane < nwaves ? L.ws[16 + lane] : -1), 63);
        // the table of the next eye (every read of this eye's is behind the barrier above)
        for (int q = tid; q < ((w + 3 * NR) >> 2); q += nt) reinterpret_cast<int4*>(L.M)[q] = make_int4(0, 0, 0, 0);
        if (GW_DEV_IS(54)) { __syncthreads(); continue; }
        const float sxw = (float)(w - 1);
        const bool sxw_ok = w <= (1 << 20);   // the division core's denominator range (w >= 8 here)
        const int iy1 = min(iy0 + 1, h - 1);
        const char* const r0 = reinterpret_cast<const char*>(A.image + frame * A.img_sf + iy0 * A.img_sy);
        const char* const r1 = reinterpret_cast<const char*>(A.image + frame * A.img_sf + iy1 * A.img_sy);
        char* const out_row = reinterpret_cast<char*>(A.out + frame * A.out_sf + (y + E.yoff) * A.out_sy + E.xoff * 3);
        auto left_filled = [&](int x) {
            if (x <= 0) return -1;
            int wi = (x - 1) >> 5;
            uint32_t cur = fbits[wi] & (0xffffffffu >> (31 - ((x - 1) & 31)));
            while (true) {
                if (cur) return wi * 32 + 31 - __clz((int)cur);
                if (--wi < 0) return -1;
                cur = fbits[wi];
            }
        };
        // final source position of column x (gap fill :393-438), then the horizontal taps of the grid_sample round trip (:440-448)
        auto taps_of = [&](int x, float s, int& ix0, int& ix1, float& ww, float& we) {
            if (s < 0.0f) {
                int left = left_filled(x);
                int right = rightmost >= x ? rightmost : -1;
                bool hl = left >= 0, hr = right >= 0;
                int li = hl ? left : 0, ri = hr ? right : 0;
                float lsrc = sm[li], rsrc = sm[ri], lz = zb[li], rz = zb[ri];
                float ld = (float)(x - left), rd = (float)(right - x);
                float tot = fmaxf(ld + rd, 1.0f);
                float t = ld / tot;
                if (!hl) t = 1.0f;
                if (!hr) t = 0.0f;
                float tb = (lz < rz) ? sqrtf(t) : 1.0f - sqrtf(1.0f - t);
                float gg = lsrc * (1.0f - tb) + rsrc * tb;
                if (hl || hr) s = gg;
            }
            const float pos = fminf(fmaxf(s, 0.0f), sxw);
            const float p2 = pos * 2.0f;
            // (gw_div_y with its range test reduced to what can fail here: 0 <= p2 <= 2 (w - 1) < 2^60, so only 0 < p2 < 2^-60)
            float q = gw_div_with(p2, sxw, ysx);
            if (__builtin_expect(!sxw_ok || (p2 < 0x1p-60f && p2 != 0.0f), 0)) { asm volatile("" ::: "memory"); q = p2 / sxw; }
            float gx = q - 1.0f;
            float xx = (gx + 1.0f) * sxw_half;
            xx = fminf(fmaxf(xx, 0.0f), sxw);
            const float xw = floorf(xx);
            ww = xx - xw; we = 1.0f - ww;
            ix0 = (int)xw; ix1 = min(ix0 + 1, w - 1);
        };
        if (GW_DEV_IS(60)) {   // (development: the row's intermediate state instead of colours)
            for (int x = tid; x < w; x += nt) *reinterpret_cast<Px3*>(out_row + 12u * (uint32_t)x) = Px3{L.ndn[x], L.D[x], sm[x]};
            __syncthreads(); continue;
        }
        // (column-strided like the stage: a wave's gathers and 12-byte stores cover neighbouring columns)
        if (one_row) {
            // wn == 0: the taps of row iy1 carry the weights 0 * we and 0 * ww -- x + 0 * p == x for finite p
            constexpr int NC = MINW < 8 ? 2 : 1;   // columns in flight per lane
            for (int xb = tid; xb < w; xb += NC * nt) {
                int ix0[NC], ix1[NC]; float ww[NC], we[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) { const int x = min(xb + c * nt, w - 1); taps_of(x, sm[x], ix0[c], ix1[c], ww[c], we[c]); }
                Px3 pa[NC], pb[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    pa[c] = *reinterpret_cast<const Px3*>(r0 + 12u * (uint32_t)ix0[c]); pb[c] = *reinterpret_cast<const Px3*>(r0 + 12u * (uint32_t)ix1[c]);
                }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const int x = xb + c * nt;
                    if (x >= w) continue;
                    const float nw = we[c], ne = ww[c];   // (wsth == 1 - 0: the products wsth * we, wsth * ww are we and ww)
                    Px3 r;
                    r.x = pa[c].x * nw + pb[c].x * ne;
                    r.y = pa[c].y * nw + pb[c].y * ne;
                    r.z = pa[c].z * nw + pb[c].z * ne;
                    *reinterpret_cast<Px3*>(out_row + 12u * (uint32_t)x) = r;
                }
            }
        } else {
            for (int x = tid; x < w; x += nt) {
                int ix0, ix1; float ww, we;
                taps_of(x, sm[x], ix0, ix1, ww, we);
                const uint32_t a0 = 12u * (uint32_t)ix0, a1 = 12u * (uint32_t)ix1;
                const Px3 pa = *reinterpret_cast<const Px3*>(r0 + a0), pb = *reinterpret_cast<const Px3*>(r0 + a1),
                          pc = *reinterpret_cast<const Px3*>(r1 + a0), pd = *reinterpret_cast<const Px3*>(r1 + a1);
                const float nw = wsth * we, ne = wsth * ww, sw2 = wn * we, se = wn * ww;
                Px3 r;
                r.x = pa.x * nw + pb.x * ne + pc.x * sw2 + pd.x * se;
                r.y = pa.y * nw + pb.y * ne + pc.y * sw2 + pd.y * se;
                r.z = pa.z * nw + pb.z * ne + pc.z * sw2 + pd.z * se;
                *reinterpret_cast<Px3*>(out_row + 12u * (uint32_t)x) = r;
            }
        }
      }
        __syncthreads();
        if (e == 0) {   // (this eye's filled bits; the gap bits accumulate over the eyes)
            const GwqLds L = gwq_carve(smem, A.w);
            for (int q = tid; q < (L.nbw >> 2); q += nt) reinterpret_cast<int4*>(L.fbits)[q] = make_int4(0, 0, 0, 0);
        }
    }
    {
        const GwqLds L = gwq_carve(smem, A.w);
        const int w = L.w;
        float* const mrow = A.mask_f32 + ((size_t)L.frame * h + L.y) * w;
        for (int x = tid; x < w; x += nt) mrow[x] = ((L.gbits[x >> 5] >> (x & 31)) & 1u) ? 1.0f : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Mesh-quality warp: forward_warp_mesh (reference stereoimage_generation.py:453-689), what the reference runs whenever
// `moderngl` is importable (:1068-1071).  The reference hands a triangle mesh to OpenGL; here the rasteriser is written
// out.  Vertices move horizontally only, so output row k is one scanline through ONE row of quads (r = floor(wy),
// wy = (k + .5)(H-1)/H under the reference's clip-space mapping): a workgroup stages the offsets and depths of mesh rows
// r and r+1, every kept triangle scatters the pixels of its span into a 64-bit LDS z-buffer (interpolated depth | draw
// order: '<' on clip_z with the first-drawn triangle winning ties), and one lane per output pixel recomputes the winning
// triangle's interpolation for the colour.  Gaps take the nearest covered pixel on the side the eye's divergence names
// (:664-687).  The implementation-defined parts of OpenGL rasterisation are fixed as in oracle/stereo_oracle.c
// (`oracle_forward_warp_mesh`, the specification this kernel is tested against; no fixture of the reference can exist
// without an OpenGL context).
// ---------------------------------------------------------------------------------------------------------------------
// normalised depth and pixel offset of one mesh row (the same arithmetic as pass 1 of k_gpuwarp)
__device__ void mesh_stage_row(const GwArgs& A, const GwEye& E, const uint32_t* st, float scale, int frame, int y, float* nd,
                               float* po, const csm::PowfTables* T) {
    const int w = A.w;
    const bool div255 = st[E.st_div] != 0;
    float dmin = csm::ord2f(st[E.st_min]), dmax = csm::ord2f(st[E.st_max]);
    if (div255) { dmin = dmin / 255.0f; dmax = dmax / 255.0f; }
    const float range = dmax - dmin;
    const float crange = fmaxf(range, (float)1e-6);
    const bool has_range = range > (float)1e-6;
    const float* drow = E.depth + ((size_t)frame * A.h + y) * w;
    const bool crange_ok = crange < 0x1p40f;
    const float yr = crange_ok ? gw_rcp_refined(crange) : 0.0f;
    const int nt = blockDim.x;
    for (int xb = threadIdx.x; xb < w; xb += 4 * nt) {   // (four columns per thread with their loads first, as in k_gpuwarp)
        float dv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) dv[u] = xb + u * nt < w ? drow[xb + u * nt] : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int x = xb + u * nt;
            if (x >= w) continue;
            float v = dv[u] * scale;
            if (div255) {
                asm volatile("" ::: "memory");
                v = v / 255.0f;
            }
            const float num = v - dmin;
            float nrm = gw_div_y(num, crange, yr, crange_ok);
            nrm = has_range ? nrm : 0.0f;
            nd[x] = nrm;
            const float s = nrm - A.conv32;
            const float sg = s > 0.0f ? 1.0f : (s < 0.0f ? -1.0f : 0.0f);
            const float ax = fabsf(s);
            po[x] = (sg * (A.pow_mode == 2 ? ax * ax : torch_pow(ax, A.pow_mode, A.e32, T))) * E.div32 + E.sep32;
        }
    }
}

__device__ __forceinline__ unsigned mesh_keep_bits(float o00, float o10, float o01, float o11, float thr) {
    const float da = fmaxf(fmaxf(fabsf(o00 - o10), fabsf(o00 - o01)), fabsf(o10 - o01));
    const float db = fmaxf(fmaxf(fabsf(o11 - o10), fabsf(o11 - o01)), fabsf(o10 - o01));
    return (da < thr ? 1u : 0u) | (db < thr ? 2u : 0u);
}

// keep bits of quad row r for one eye, OR-ed over the frames of a group ("keep triangle if it passes in ANY batch item")
__global__ void __launch_bounds__(512) k_mesh_keep(GwArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int r = blockIdx.x, grp = blockIdx.y, e = blockIdx.z, w = A.w, h = A.h;
    float* nd = (float*)smem;      // (unused here, staged by the shared helper)
    float* o0 = nd + w;
    float* o1 = o0 + w;
    uint8_t* acc = (uint8_t*)(o1 + w);
    csm::PowfTables* T = (csm::PowfTables*)(acc + align16((size_t)w));
    const GwEye& E = A.eye[e];
    if (!E.enabled) return;
    if (A.pow_mode == 4) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_gw_powf_tables);
        for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += nt) reinterpret_cast<uint32_t*>(T)[i] = src[i];
    }
    for (int x = tid; x < w; x += nt) acc[x] = 0;
    __syncthreads();
    const int f0 = grp * A.group, f1 = min(f0 + A.group, A.n);
    for (int f = f0; f < f1; f++) {
        const uint32_t* st = A.stats + (size_t)f * ST_WORDS;
        const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
        mesh_stage_row(A, E, st, scale, f, r, nd, o0, T);
        __syncthreads();
        mesh_stage_row(A, E, st, scale, f, r + 1, nd, o1, T);
        __syncthreads();
        for (int x = tid; x < w - 1; x += nt) acc[x] |= (uint8_t)mesh_keep_bits(o0[x], o0[x + 1], o1[x], o1[x + 1], A.grad_thr);
        __syncthreads();
    }
    const int ngroups = (A.n + A.group - 1) / A.group;
    uint8_t* dst = A.keep + (((size_t)e * ngroups + grp) * (h - 1) + r) * (size_t)(w - 1);
    for (int x = tid; x < w - 1; x += nt) dst[x] = acc[x];
}

template <int MINW>   // (as for k_gpuwarp)
__global__ void __launch_bounds__(1024, MINW) k_meshwarp(GwArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, lane = lane_id(), wave = wave_id();
    const int k = blockIdx.x, frame = blockIdx.y, w = A.w, h = A.h;
    unsigned long long* key = (unsigned long long*)smem;   // [w] ordered depth << 32 | ~draw index; 0 = uncovered
    float* n0 = (float*)(key + w);   // normalised depth, mesh rows r / r+1
    float* n1 = n0 + w;
    float* o0 = n1 + w;              // pixel offset, mesh rows r / r+1
    float* o1 = o0 + w;
    int* fillcol = (int*)(o1 + w);   // nearest covered pixel (scan)
    uint8_t* flags = (uint8_t*)(fillcol + w);   // gap in some eye
    int* ws = (int*)(flags + align16((size_t)w));
    csm::PowfTables* T = (csm::PowfTables*)(ws + 32);
    if (A.pow_mode == 4) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_gw_powf_tables);
        for (int i = tid; i < (int)(sizeof(csm::PowfTables) / 4); i += nt) reinterpret_cast<uint32_t*>(T)[i] = src[i];
    }
    const uint32_t* st = A.stats + (size_t)frame * ST_WORDS;
    const float scale = (A.scale_from_stats && st[ST_SCALE255]) ? 255.0f : 1.0f;
    for (int x = tid; x < w; x += nt) flags[x] = 0;
    // the scanline of output row k in mesh coordinates
    const float sc = (float)(w - 1) / (float)w, isc = (float)w / (float)(w - 1), scy = (float)(h - 1) / (float)h;
    const float wy = ((float)k + 0.5f) * scy;
    const int r = (int)floorf(wy);
    const float t = wy - (float)r, omt = 1.0f - t;
    const int ngroups = (A.n + A.group - 1) / A.group, grp = frame / A.group;
    const int BIG = 1 << 29;
    __syncthreads();

    for (int e = 0; e < A.neyes; e++) {
        const GwEye& E = A.eye[e];
        if (!E.enabled) {
            if (A.out)
                for (int x = tid; x < w; x += nt)
                    for (int c = 0; c < 3; c++)
                        if (E.chan_mask & (1 << c))
                            A.out[frame * A.out_sf + (k + E.yoff) * A.out_sy + (x + E.xoff) * A.out_sx + c * A.out_sc] =
                                A.image[frame * A.img_sf + k * A.img_sy + x * A.img_sx + c * A.img_sc];
            continue;
        }
        mesh_stage_row(A, E, st, scale, frame, r, n0, o0, T);
        mesh_stage_row(A, E, st, scale, frame, r + 1, n1, o1, T);
        for (int x = tid; x < w; x += nt) key[x] = 0ull;
        __syncthreads();
        const uint8_t* keep = A.keep + (((size_t)e * ngroups + grp) * (h - 1) + r) * (size_t)(w - 1);
        // the span of triangle `type` of quad x on the scanline: end points a -> e2 (x), their depths
        auto span = [&](int x, int type, float& a, float& e2, float& za, float& ze) {
            const float P00 = (float)x + o0[x], P10 = (float)(x + 1) + o0[x + 1], P01 = (float)x + o1[x],
                        P11 = (float)(x + 1) + o1[x + 1];
            const float xl = omt * P00 + t * P01, xd = omt * P10 + t * P01, xr = omt * P10 + t * P11;
            const float zl = omt * n0[x] + t * n1[x], zd = omt * n0[x + 1] + t * n1[x], zr = omt * n0[x + 1] + t * n1[x + 1];
            a = type ? xd : xl; e2 = type ? xr : xd;
            za = type ? zd : zl; ze = type ? zr : zd;
        };
        // ---- scatter: every kept triangle into the pixels of its span
        for (int x = tid; x < w - 1; x += nt) {
            const unsigned kb = keep[x];
#pragma unroll
            for (int type = 0; type < 2; type++) {
                if (!(kb & (1u << type))) continue;
                float a, e2, za, ze;
                span(x, type, a, e2, za, ze);
                const float lo = fminf(a, e2), hi = fmaxf(a, e2);
                if (!(lo < hi)) continue;
                float f0 = floorf(lo * isc) - 1.0f, f1 = floorf(hi * isc) + 1.0f;
                f0 = fmaxf(f0, 0.0f); f1 = fminf(f1, (float)(w - 1));
                if (!(f0 <= f1)) continue;
                const unsigned order = 0xffffffffu - (unsigned)(type * (w - 1) + x);
                for (int px = (int)f0; px <= (int)f1; px++) {
                    const float u = ((float)px + 0.5f) * sc;
                    if (!(lo <= u && u < hi)) continue;
                    const float s = (u - a) / (e2 - a);
                    const float z = (1.0f - s) * za + s * ze;
                    atomicMax(&key[px], ((unsigned long long)csm::f2ord(z) << 32) | order);
                }
            }
        }
        __syncthreads();
        // ---- covered pixels; nearest covered pixel on the side this eye fills from
        const bool from_left = !(E.div32 < 0.0f);
        for (int x = tid; x < w; x += nt) {
            const bool covered = key[x] != 0ull;
            fillcol[x] = covered ? x : (from_left ? -1 : BIG);
            if (!covered) flags[x] = 1;
        }
        __syncthreads();
        if (from_left) block_scan_inclusive(fillcol, w, -1, OpMax(), ws);
        else block_scan_inclusive(fillcol, w, BIG, OpMin(), ws, true);
        // ---- colours
        const float* const img_r0 = A.image + frame * A.img_sf + r * A.img_sy;
        const float* const img_r1 = img_r0 + A.img_sy;
        float* const out_row = A.out + frame * A.out_sf + (k + E.yoff) * A.out_sy + E.xoff * A.out_sx;
        for (int px = tid; px < w; px += nt) {
            const int f = fillcol[px];
            float c3[3] = {0.0f, 0.0f, 0.0f};
            if (f >= 0 && f < w) {   // (the pixel itself when it is covered)
                const unsigned draw = 0xffffffffu - (unsigned)(key[f] & 0xffffffffull);
                const int type = draw >= (unsigned)(w - 1) ? 1 : 0, x = (int)draw - type * (w - 1);
                float a, e2, za, ze;
                span(x, type, a, e2, za, ze);
                const float u = ((float)f + 0.5f) * sc;
                const float s = (u - a) / (e2 - a), oms = 1.0f - s;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float c00 = img_r0[(size_t)x * A.img_sx + c * A.img_sc], c10 = img_r0[(size_t)(x + 1) * A.img_sx + c * A.img_sc],
                                c01 = img_r1[(size_t)x * A.img_sx + c * A.img_sc], c11 = img_r1[(size_t)(x + 1) * A.img_sx + c * A.img_sc];
                    const float cl = omt * c00 + t * c01, cd = omt * c10 + t * c01, cr = omt * c10 + t * c11;
                    c3[c] = oms * (type ? cd : cl) + s * (type ? cr : cd);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++)
                if (E.chan_mask & (1 << c)) out_row[(size_t)px * A.out_sx + c * A.out_sc] = c3[c];
        }
        __syncthreads();
    }
    if (A.mask_u8)
        for (int x = tid; x < w; x += nt) A.mask_u8[((size_t)frame * h + k) * w + x] = flags[x];
    if (A.mask_f32)
        for (int x = tid; x < w; x += nt) A.mask_f32[((size_t)frame * h + k) * w + x] = flags[x] ? 1.0f : 0.0f;
    if (A.depth_l) {
        for (int e = 0; e < 2; e++) {
            const GwEye& E = A.eye[e];
            const bool div255 = st[E.st_div] != 0;
            const float* drow = E.depth + ((size_t)frame * h + k) * w;
            float* dst = (e == 0 ? A.depth_l : A.depth_r) + (((size_t)frame * h + k) * w) * 3;
            for (int x = tid; x < w; x += nt) {
                float v = drow[x] * scale;
                if (div255) v = v / 255.0f;
                if (!A.noclamp) v = fminf(fmaxf(v, 0.0f), 1.0f);
                *reinterpret_cast<Px3*>(dst + 3 * x) = Px3{v, v, v};
            }
        }
    }
}

// forward_warp_gpu's `if (d_max_all > 1.0).any(): d = d / 255.0` is global over the tensor it is handed,
// i.e. over one reference sub-batch (`group` frames).
// (round 6) ... and the wave-uniform constants k_gpuwarp_q would otherwise derive in every wave of every row (two IEEE divisions, the
// refined reciprocals, the key decoding: ~170 of the 470 vector instructions a wave spent per eye on 3.75 columns): per frame
// GWC_WORDS floats -- per eye {dmin, clamped range, its refined reciprocal, flags: bit 0 has_range, bit 1 range inside the division
// core's reach, bit 2 divide by 255}, then the refined reciprocal of w - 1, the scale the kernel multiplies with, the frame's x255 flag.
__global__ void k_gpuwarp_flags(uint32_t* stats, int n, int group, float* fconst, int w, int scale_from_stats) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    int g0 = (f / group) * group, g1 = min(g0 + group, n);
    uint32_t fl = 0, fr = 0;
    for (int k = g0; k < g1; k++) {
        if (csm::ord2f(stats[k * ST_WORDS + ST_L_MAX]) > 1.0f) fl = 1;
        if (csm::ord2f(stats[k * ST_WORDS + ST_R_MAX]) > 1.0f) fr = 1;
    }
    stats[f * ST_WORDS + ST_WARP_DIV255_L] = fl;
    stats[f * ST_WORDS + ST_WARP_DIV255_R] = fr;
    if (!fconst) return;
    float* C = fconst + (size_t)f * GWC_WORDS;
    for (int e = 0; e < 2; e++) {
        const bool div255 = (e ? fr : fl) != 0;
        float dmin = csm::ord2f(stats[f * ST_WORDS + (e ? ST_R_MIN : ST_L_MIN)]), dmax = csm::ord2f(stats[f * ST_WORDS + (e ? ST_R_MAX : ST_L_MAX)]);
        if (div255) { dmin = dmin / 255.0f; dmax = dmax / 255.0f; }
        const float range = dmax - dmin;
        const float crange = fmaxf(range, (float)1e-6);
        const bool has_range = range > (float)1e-6;
        const bool crange_ok = crange < 0x1p40f;   // (>= 1e-6 by construction)
        C[GWC_EYE * e + 0] = dmin;
        C[GWC_EYE * e + 1] = crange;
        C[GWC_EYE * e + 2] = crange_ok ? gw_rcp_refined(crange) : 0.0f;
        C[GWC_EYE * e + 3] = __builtin_bit_cast(float, (has_range ? 1u : 0u) | (crange_ok ? 2u : 0u) | (div255 ? 4u : 0u));
    }
    const float sxw = (float)(w - 1);
    const bool sxw_ok = w >= 2 && w <= (1 << 20);
    C[GWC_YSX] = sxw_ok ? gw_rcp_refined(sxw) : 0.0f;
    C[GWC_SCALE] = (scale_from_stats && stats[f * ST_WORDS + ST_SCALE255]) ? 255.0f : 1.0f;
    C[GWC_S255] = __builtin_bit_cast(float, stats[f * ST_WORDS + ST_SCALE255]);   // (the lazy tiles' gray values take the scale whatever scale_from_stats says)
}

static size_t gw_lds_bytes(int w, int gen = 0) {
    return 5 * (size_t)w * 4 + (gen ? 48 : 24) * 4 + align16((size_t)w) + 32 * 4 + sizeof(csm::PowfTables) + 64;
}
static size_t mesh_lds_bytes(int w) { return 8 * (size_t)w + 5 * (size_t)w * 4 + align16((size_t)w) + 32 * 4 + sizeof(csm::PowfTables) + 64; }
static size_t mesh_keep_lds_bytes(int w) { return 3 * (size_t)w * 4 + align16((size_t)w) + sizeof(csm::PowfTables) + 64; }
// (the mesh variant's keep bits: 2 eyes x groups x (h-1)(w-1) bytes)
// [256 B][per-frame constants of k_gpuwarp_q: GWC_WORDS floats each][mesh: keep bits]
static size_t gw_const_bytes(int n) { return ((size_t)n * GWC_WORDS * 4 + 255) & ~(size_t)255; }
size_t gpuwarp_workspace_bytes(int n, int h, int w, int group, int mesh) {
    if (!mesh) return 256 + gw_const_bytes(n);
    const int g = group > 0 && group < n ? group : n;
    return 256 + gw_const_bytes(n) + 2 * (size_t)((n + g - 1) / g) * h * w;
}
int meshwarp_max_width() {
    int lo = 2, hi = 1 << 15;
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (mesh_lds_bytes(mid) <= CS_LDS_BYTES) lo = mid; else hi = mid - 1;
    }
    return lo;
}
int gpuwarp_max_width() {
    int lo = 2, hi = 1 << 15;
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (gw_lds_bytes(mid) <= CS_LDS_BYTES) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// Rounds of the scatter loop (:365) that can change a column.  A valid proposal of pair (i, i+1) in round k needs
// 0 <= frac < 1, i.e. the column fs + k inside [dl, dr) (or (dr, dl] for a reversed pair) with fs = floor(min(dl, dr)) > min - 1;
// connected pairs have |dr - dl| < 1 + gradient_threshold, so k < gradient_threshold + 2: rounds beyond that only write back
// what they gathered.  -1: more than the 16 rounds the general kernel provides for (gradient_threshold > 13).
static int gw_rounds(double gradient_threshold, int max_stretch) {
    if (max_stretch <= 0) return 0;
    if (!(gradient_threshold > 0.0)) return 0;   // nothing is connected (also NaN)
    const double kv = floor((double)(float)gradient_threshold + 2.0) + 1.0;
    const int rounds = kv < (double)max_stretch ? (int)kv : max_stretch;
    return rounds > 16 ? -1 : rounds;
}
static int pow_mode_of(double e) { return e == 1.0 ? 0 : e == 0.5 ? 1 : e == 2.0 ? 2 : e == 3.0 ? 3 : e == 0.0 ? 5 : 4; }

static int gw_launch(GwArgs& A, hipStream_t stream) {
    A.lin_step = A.h > 1 ? 2.0f / (float)(A.h - 1) : 0.0f;
    if (A.mesh) {
        if (A.h < 2 || A.w < 2) return CS_EINVAL;   // the reference divides by H - 1 and W - 1
        if (A.w > meshwarp_max_width()) return CS_ELIMIT;
        const int ngroups = (A.n + A.group - 1) / A.group;
        const int threads = A.w <= 1024 ? 256 : (A.w <= 2048 ? 512 : 1024);
        size_t lk = mesh_keep_lds_bytes(A.w), lm = mesh_lds_bytes(A.w);
        hipError_t e = hipFuncSetAttribute((const void*)k_mesh_keep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lk);
        if (e != hipSuccess) return CS_EHIP;
        e = hipFuncSetAttribute(threads > 512 ? (const void*)k_meshwarp<8> : (const void*)k_meshwarp<6>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
        if (e != hipSuccess) return CS_EHIP;
        hipLaunchKernelGGL(k_mesh_keep, dim3(A.h - 1, ngroups, A.neyes), dim3(threads > 512 ? 512 : threads), lk, stream, A);
        if (threads > 512) hipLaunchKernelGGL(k_meshwarp<8>, dim3(A.h, A.n), dim3(threads), lm, stream, A);
        else hipLaunchKernelGGL(k_meshwarp<6>, dim3(A.h, A.n), dim3(threads), lm, stream, A);
        return CS_OK;
    }
    // forward_warp_gpu's keyword parameters away from their defaults: the general instantiation
    const bool gen = !(A.rounds == 4 && A.grad_thr == 1.5f);
    if (gen && (A.rounds < 0 || A.rounds > 16)) return CS_ELIMIT;
    size_t lds = gw_lds_bytes(A.w, gen);
    if (A.pow_mode != 4) lds -= sizeof(csm::PowfTables) + 64;   // (the tables are the LAST item of the layout and only the general exponent reads them)
    // workgroup size: about 4 columns per thread (measured at 1080p: 512 threads 2.30 ms per 32 frames, 1024: 2.71, 256: 2.94)
    int threads = A.w <= 1024 ? 256 : (A.w <= 2048 ? 512 : 1024);
    const int forced = dev_switch(CS_DEBUG_PT_VARIANT);   // (development: workgroup size)
    if (forced == 21) threads = 512;
    if (forced == 22 && A.w <= 4 * 256) threads = 256;
    if (forced == 23) threads = 1024;
    if (forced == 26) threads = 256;   // (1080p: four workgroups of four waves per CU, 7.5 columns per lane)
    // 1080p: a row takes 40.5 KB without the tables -- FOUR 512-thread workgroups per CU instead of three if the kernel also
    // fits 64 registers (the 8-waves-per-SIMD instantiation; development switch 24: the 6-wave one)
    const bool four = threads == 512 && 4 * ((lds + 511) & ~(size_t)511) <= CS_LDS_BYTES && forced != 24;
    const bool wide = threads > 512 || four, pow2 = A.pow_mode == 2 && !gen;
    // the node's layout (interleaved image and output, every eye writes the three channels) has instantiations of its own
    bool node = !gen && A.out && A.img_sc == 1 && A.img_sx == 3 && A.out_sc == 1 && A.out_sx == 3 && forced != 25;
    for (int e = 0; e < A.neyes; e++) node = node && A.eye[e].chan_mask == 7;
    const void* fn = gen ? (wide ? (const void*)k_gpuwarp<8, -1, true> : (const void*)k_gpuwarp<6, -1, true>)
                   : node ? (wide ? (pow2 ? (const void*)k_gpuwarp<8, 2, false, true> : (const void*)k_gpuwarp<8, -1, false, true>)
                                  : (pow2 ? (const void*)k_gpuwarp<6, 2, false, true> : (const void*)k_gpuwarp<6, -1, false, true>))
                          : wide ? (pow2 ? (const void*)k_gpuwarp<8, 2> : (const void*)k_gpuwarp<8, -1>)
                                 : (pow2 ? (const void*)k_gpuwarp<6, 2> : (const void*)k_gpuwarp<6, -1>);
    // (round 6) the node's layout with four contiguous columns per lane: k_gpuwarp_q.  CS_DEBUG_PT_VARIANT 27: k_gpuwarp as before
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const bool quad = node && forced != 27 && A.neyes == 2 && A.eye[0].enabled && A.eye[1].enabled && (A.w & 3) == 0 && A.w >= 8 &&
                      A.depth_l && A.depth_r && A.mask_f32 && !A.mask_u8 && al16(A.out) && al16(A.depth_l) && al16(A.depth_r) &&
                      al16(A.mask_f32) && al16(A.eye[0].depth) && al16(A.eye[1].depth) && (!A.tilemap || al16(A.gray));
    if (quad) {
        size_t lq = gwq_lds_bytes(A.w);
        if (A.pow_mode != 4) lq -= sizeof(csm::PowfTables) + 64;
        const bool four_q = threads == 512 && 4 * ((lq + 511) & ~(size_t)511) <= CS_LDS_BYTES && forced != 24;
        const bool wide_q = threads > 512 || four_q;
        const void* fq = wide_q ? (pow2 ? (const void*)k_gpuwarp_q<8, 2> : (const void*)k_gpuwarp_q<8, -1>)
                                : (pow2 ? (const void*)k_gpuwarp_q<6, 2> : (const void*)k_gpuwarp_q<6, -1>);
        hipError_t eq = hipFuncSetAttribute(fq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lq);
        if (eq != hipSuccess) return CS_EHIP;
        const dim3 grid(A.h, A.n), block(threads);
        if (wide_q && pow2) hipLaunchKernelGGL((k_gpuwarp_q<8, 2>), grid, block, lq, stream, A);
        else if (wide_q) hipLaunchKernelGGL((k_gpuwarp_q<8, -1>), grid, block, lq, stream, A);
        else if (pow2) hipLaunchKernelGGL((k_gpuwarp_q<6, 2>), grid, block, lq, stream, A);
        else hipLaunchKernelGGL((k_gpuwarp_q<6, -1>), grid, block, lq, stream, A);
        return CS_OK;
    }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return CS_EHIP;
#ifdef GW_XCD_ROWS
    const dim3 grid(8 * ((A.h + 7) / 8), A.n), block(threads);
#else
    const dim3 grid(A.h, A.n), block(threads);
#endif
    if (gen && wide) hipLaunchKernelGGL((k_gpuwarp<8, -1, true>), grid, block, lds, stream, A);
    else if (gen) hipLaunchKernelGGL((k_gpuwarp<6, -1, true>), grid, block, lds, stream, A);
    else if (node && wide && pow2) hipLaunchKernelGGL((k_gpuwarp<8, 2, false, true>), grid, block, lds, stream, A);
    else if (node && wide) hipLaunchKernelGGL((k_gpuwarp<8, -1, false, true>), grid, block, lds, stream, A);
    else if (node && pow2) hipLaunchKernelGGL((k_gpuwarp<6, 2, false, true>), grid, block, lds, stream, A);
    else if (node) hipLaunchKernelGGL((k_gpuwarp<6, -1, false, true>), grid, block, lds, stream, A);
    else if (wide && pow2) hipLaunchKernelGGL((k_gpuwarp<8, 2>), grid, block, lds, stream, A);
    else if (wide) hipLaunchKernelGGL((k_gpuwarp<8, -1>), grid, block, lds, stream, A);
    else if (pow2) hipLaunchKernelGGL((k_gpuwarp<6, 2>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((k_gpuwarp<6, -1>), grid, block, lds, stream, A);
    return CS_OK;
}

int launch_gpuwarp_plain(const float* image, const float* depth, int n, int h, int w, double div_px, double sep_px,
                         double exponent, double convergence, float* warped, uint8_t* gap_mask, uint32_t* stats,
                         void* extra, hipStream_t stream, int mesh, double grad_thr, int max_stretch) {
    float* const fconst = reinterpret_cast<float*>((uint8_t*)extra + 256);
    hipLaunchKernelGGL(k_gpuwarp_flags, dim3((n + 63) / 64), dim3(64), 0, stream, stats, n, n, fconst, w, 0);
    GwArgs A;
    memset(&A, 0, sizeof(A));
    A.n = n; A.h = h; A.w = w;
    A.image = image;
    A.img_sf = 3LL * h * w; A.img_sc = (long long)h * w; A.img_sy = w; A.img_sx = 1;
    A.stats = stats;
    A.scale_from_stats = 0;
    A.pow_mode = pow_mode_of(exponent);
    A.e32 = (float)exponent; A.conv32 = (float)convergence;
    A.neyes = 1;
    A.eye[0].depth = depth;
    A.eye[0].div32 = (float)div_px; A.eye[0].sep32 = (float)sep_px;
    A.eye[0].enabled = 1;
    A.eye[0].st_min = ST_L_MIN; A.eye[0].st_max = ST_L_MAX; A.eye[0].st_div = ST_WARP_DIV255_L;
    A.eye[0].chan_mask = 7;
    A.out = warped;
    A.out_sf = A.img_sf; A.out_sc = A.img_sc; A.out_sy = w; A.out_sx = 1;
    A.mask_u8 = gap_mask;
    A.dbg = dev_switch(CS_DEBUG_DBG);
    A.mesh = mesh; A.grad_thr = (float)grad_thr; A.keep = (uint8_t*)extra + 256 + gw_const_bytes(n); A.group = n; A.fconst = fconst;
    A.rounds = gw_rounds(grad_thr, max_stretch);
    if (!mesh && A.rounds < 0) return CS_ELIMIT;
    return gw_launch(A, stream);
}

int launch_gpuwarp_node(const cs_params* p, const float* image, const float* dL, const float* dR, int scale_from_stats,
                        uint32_t* stats, float* stereo, float* depth_l, float* depth_r, float* mask, int out_h,
                        int out_w, void* extra, hipStream_t stream, const uint32_t* tilemap, const float* gray, int tm_words) {
    const int n = p->n, h = p->h, w = p->w;
    int group = p->batch_size > 0 ? (p->batch_size < n ? p->batch_size : n) : n;
    float* const fconst = reinterpret_cast<float*>((uint8_t*)extra + 256);
    hipLaunchKernelGGL(k_gpuwarp_flags, dim3((n + 63) / 64), dim3(64), 0, stream, stats, n, group, fconst, w, scale_from_stats);
    GwArgs A;
    memset(&A, 0, sizeof(A));
    A.n = n; A.h = h; A.w = w;
    A.image = image;
    A.img_sf = 3LL * h * w; A.img_sy = 3LL * w; A.img_sx = 3; A.img_sc = 1;
    A.stats = stats;
    A.scale_from_stats = scale_from_stats;
    A.pow_mode = pow_mode_of(p->stereo_offset_exponent);
    A.e32 = (float)p->stereo_offset_exponent; A.conv32 = (float)p->convergence_point;
    A.neyes = 2;
    const double left_div = p->divergence * (1 + p->stereo_balance), right_div = p->divergence * (1 - p->stereo_balance);
    const double lpx = (left_div / 100.0) * w, rpx = (right_div / 100.0) * w, spx = (p->separation / 100.0) * w;
    A.eye[0].depth = dL; A.eye[0].div32 = (float)(+lpx); A.eye[0].sep32 = (float)(-spx);
    A.eye[1].depth = dR; A.eye[1].div32 = (float)(-rpx); A.eye[1].sep32 = (float)(spx);
    A.eye[0].enabled = !(left_div < 0.001); A.eye[1].enabled = !(right_div < 0.001);
    A.eye[0].st_min = ST_L_MIN; A.eye[0].st_max = ST_L_MAX; A.eye[0].st_div = ST_WARP_DIV255_L;
    A.eye[1].st_min = ST_R_MIN; A.eye[1].st_max = ST_R_MAX; A.eye[1].st_div = ST_WARP_DIV255_R;
    A.eye[0].chan_mask = A.eye[1].chan_mask = 7;
    switch (p->mode) {
    case CS_MODE_LEFT_RIGHT: A.eye[1].xoff = w; break;
    case CS_MODE_RIGHT_LEFT: A.eye[0].xoff = w; break;
    case CS_MODE_TOP_BOTTOM: A.eye[1].yoff = h; break;
    case CS_MODE_BOTTOM_TOP: A.eye[0].yoff = h; break;
    case CS_MODE_RED_CYAN_ANAGLYPH: A.eye[0].chan_mask = 1; A.eye[1].chan_mask = 6; break;
    case CS_MODE_CYAN_RED_REVERSEANAGLYPH: A.eye[0].chan_mask = 6; A.eye[1].chan_mask = 1; break;
    case CS_MODE_LEFT_ONLY: A.eye[1].chan_mask = 0; break;
    case CS_MODE_ONLY_RIGHT: A.eye[0].chan_mask = 0; break;
    default: return CS_EINVAL;
    }
    A.out = stereo;
    A.out_sf = 3LL * out_h * out_w; A.out_sy = 3LL * out_w; A.out_sx = 3; A.out_sc = 1;
    A.mask_f32 = mask;
    A.depth_l = depth_l; A.depth_r = depth_r;
    A.noclamp = p->flags & 1;
    A.dbg = dev_switch(CS_DEBUG_DBG);
    A.mesh = (p->flags & 4) ? 1 : 0; A.grad_thr = 1.5f; A.rounds = 4; A.keep = (uint8_t*)extra + 256 + gw_const_bytes(n); A.group = group; A.fconst = fconst;   // (create_stereoimages_gpu calls the warp with its defaults, :1068-1083)
    if (tilemap && (A.mesh || w > gpuwarp_lazy_max_width())) return CS_EINVAL;   // (the caller asked gpuwarp_lazy_max_width)
    A.tilemap = tilemap; A.gray = gray; A.tm_words = tm_words;
    return gw_launch(A, stream);
}
int gpuwarp_lazy_max_width() { return 4096; }   // two words of tile bits = 64 tiles of 64 columns

}  // namespace cs
