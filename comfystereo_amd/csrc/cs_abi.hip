// cs_abi.hip -- the extern "C" boundary (include/comfystereo_amd.h) and the small streaming kernels
// around the row kernels: depth grayscale + per-frame min/max, bilinear depth resize, per-frame
// decisions (0..255 scaling), device self-tests of the libm-exact math.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>

#include "cs_common.h"
#include "cs_kernels.h"

namespace cs {

static thread_local char g_err[256] = "";
static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
static int fail_hip(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return CS_EHIP;
}

// launch_blur's three failure classes get their own message (CS_EHIP reads the HIP error that caused it)
static int fail_blur(int rc) {
    if (rc == CS_EHIP) return fail_hip(hipGetLastError(), "depth blur");
    if (rc == CS_ELIMIT) return fail(rc, "depth blur: frame or blur kernel too wide for the LDS of the two-pass path");
    return fail(rc, "depth blur: unsupported parameters (strength must round to >= 1)");
}

// development switches: explicit state set through cs_debug_set (never the environment)
static std::atomic<int> g_dev[CS_DEBUG_KEYS];
int dev_switch(int key) { return (key >= 0 && key < CS_DEBUG_KEYS) ? g_dev[key].load(std::memory_order_relaxed) : 0; }

// ---------------------------------------------------------------------------------------------
// streaming kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_stats_init(uint32_t* stats, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * ST_WORDS) return;
    int wd = i % ST_WORDS;
    bool is_min = wd == ST_GRAY_MIN || wd == ST_L_MIN || wd == ST_R_MIN;
    stats[i] = is_min ? 0xffffffffu : 0u;
}

__device__ __forceinline__ void block_minmax_commit(float mn, float mx, uint32_t* st_min, uint32_t* st_max) {
    __shared__ float red[2 * 16];
    block_minmax_update(mn, mx, st_min, st_max, red);
}

// gray = (0.2989 R + 0.5870 G) + 0.1140 B with separate float32 roundings (GenerateStereo.py:134-139,
// 206-209); C == 1 -> the channel; other C -> channel 0.  grid: (blocks, n).
__global__ void __launch_bounds__(256) k_gray(const float* __restrict__ depth, float* __restrict__ gray, int hw,
                                              int c, uint32_t* stats, int do_minmax) {
    const int frame = blockIdx.y;
    const float* src = depth + (size_t)frame * hw * c;
    float* dst = gray + (size_t)frame * hw;
    float mn = INFINITY, mx = -INFINITY;
    const int stride = gridDim.x * blockDim.x;
    if (c == 3 && (hw & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < hw / 4; q += stride) {
            float4 a = s4[3 * q], b = s4[3 * q + 1], cc = s4[3 * q + 2];
            float4 g;
            g.x = (0.2989f * a.x + 0.5870f * a.y) + 0.1140f * a.z;
            g.y = (0.2989f * a.w + 0.5870f * b.x) + 0.1140f * b.y;
            g.z = (0.2989f * b.z + 0.5870f * b.w) + 0.1140f * cc.x;
            g.w = (0.2989f * cc.y + 0.5870f * cc.z) + 0.1140f * cc.w;
#ifdef GRAY_NT_STORE   // (experiment, round 5: like k_gray_edges)
            typedef float gr_v4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(gr_v4{g.x, g.y, g.z, g.w}, reinterpret_cast<gr_v4*>(d4 + q));
#else
            d4[q] = g;
#endif
            mn = fminf(fminf(mn, g.x), fminf(g.y, fminf(g.z, g.w)));
            mx = fmaxf(fmaxf(mx, g.x), fmaxf(g.y, fmaxf(g.z, g.w)));
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += stride) {
            float g;
            if (c == 3) g = (0.2989f * src[3 * i] + 0.5870f * src[3 * i + 1]) + 0.1140f * src[3 * i + 2];
            else g = src[(size_t)i * c];
            dst[i] = g;
            mn = fminf(mn, g);
            mx = fmaxf(mx, g);
        }
    }
    if (do_minmax) block_minmax_commit(mn, mx, &stats[frame * ST_WORDS + ST_GRAY_MIN], &stats[frame * ST_WORDS + ST_GRAY_MAX]);
}

// per-frame min/max of a [n][hw] float array into the given stats words
__global__ void __launch_bounds__(256) k_minmax(const float* __restrict__ a, int hw, uint32_t* stats, int wmin, int wmax) {
    const int frame = blockIdx.y;
    const float* src = a + (size_t)frame * hw;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        float g = src[i];
        mn = fminf(mn, g);
        mx = fmaxf(mx, g);
    }
    block_minmax_commit(mn, mx, &stats[frame * ST_WORDS + wmin], &stats[frame * ST_WORDS + wmax]);
}

// F.interpolate(mode='bilinear', align_corners=False) of the gray depth (GenerateStereo.py:141-148, 214-220) + min/max of the
// result, bit for bit as CPU torch 2.10 computes it (probed against torch in the build container, tests/golden/resize.npz):
//   source index = fma(scale, dst + 0.5, -0.5) clamped at 0 (scale = (float)in / out; the compiler contracts the expression),
//   lambda1 = index - floor, lambda0 = 1 - lambda1, second tap = first + 1 unless at the border;
//   outputs with out_h + out_w > 128 take the separable loop  fma(t0, wy0, t1 * wy1), t = fma(a, wx0, b * wx1);
//   smaller ones the four-tap loop over weight products  fma(d, wy1*wx1, fma(c, wy1*wx0, fma(a, wy0*wx0, b * (wy0*wx1))))
//   (which of its two loops torch's TensorIterator picks depends on out_h + out_w only: scanned for 3 <= out_h <= 300 and every
//   out_w up to 4200 pixels per frame, independent of the input size, the batch and the thread count).
__global__ void __launch_bounds__(256) k_resize_bilinear(const float* __restrict__ in, int ih, int iw,
                                                         float* __restrict__ out, int oh, int ow, uint32_t* stats) {
    const int frame = blockIdx.y;
    const float* src = in + (size_t)frame * ih * iw;
    float* dst = out + (size_t)frame * oh * ow;
    const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
    const bool small = oh + ow <= 128;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < oh * ow; i += gridDim.x * blockDim.x) {
        const int y = i / ow, x = i - y * ow;
        const float fy = fmaxf(fmaf(sy, (float)y + 0.5f, -0.5f), 0.0f), fx = fmaxf(fmaf(sx, (float)x + 0.5f, -0.5f), 0.0f);
        const int y0 = min((int)fy, ih - 1), x0 = min((int)fx, iw - 1);
        const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float a = src[y0 * iw + x0], b = src[y0 * iw + x1], c = src[y1 * iw + x0], d = src[y1 * iw + x1];
        float v;
        if (small) {
            v = b * (hy * lx);
            v = fmaf(a, hy * hx, v);
            v = fmaf(c, ly * hx, v);
            v = fmaf(d, ly * lx, v);
        } else {
            const float t0 = fmaf(a, hx, b * lx), t1 = fmaf(c, hx, d * lx);
            v = fmaf(t0, hy, t1 * ly);
        }
        dst[i] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    block_minmax_commit(mn, mx, &stats[frame * ST_WORDS + ST_GRAY_MIN], &stats[frame * ST_WORDS + ST_GRAY_MAX]);
}

// per-frame decisions that the reference takes on host scalars (kept on device: no sync)
//   CPU techniques: `if depthmap.max() <= 1.0: depthmap *= 255`, per frame (:1475)
//   gpu_warp:       `if depth_tensor.amax() <= 1.0` over the sub-batch (:1045)
__global__ void k_finalize_stats(uint32_t* stats, int n, int group, int blur) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    float gmax = csm::ord2f(stats[f * ST_WORDS + ST_GRAY_MAX]);
    float gmin = csm::ord2f(stats[f * ST_WORDS + ST_GRAY_MIN]);
    float dec = gmax;
    if (group > 0) {
        int g0 = (f / group) * group, g1 = min(g0 + group, n);
        for (int k = g0; k < g1; k++) dec = fmaxf(dec, csm::ord2f(stats[k * ST_WORDS + ST_GRAY_MAX]));
    }
    uint32_t sc = dec <= 1.0f ? 1u : 0u;
    stats[f * ST_WORDS + ST_SCALE255] = sc;
    if (!blur) {
        float s = sc ? 255.0f : 1.0f;
        uint32_t mn = csm::f2ord(gmin * s), mx = csm::f2ord(gmax * s);
        stats[f * ST_WORDS + ST_L_MIN] = mn;
        stats[f * ST_WORDS + ST_R_MIN] = mn;
        stats[f * ST_WORDS + ST_L_MAX] = mx;
        stats[f * ST_WORDS + ST_R_MAX] = mx;
    }
}

// uint8 codes -> float32 k/255 (true division through a LUT), 4 values per thread
__global__ void __launch_bounds__(256) k_expand_u8(const uint8_t* __restrict__ in, float* __restrict__ out, size_t count) {
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
    __syncthreads();
    const size_t nq = count / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nq; i += stride) {
        uint32_t pk = reinterpret_cast<const uint32_t*>(in)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(lut[pk & 0xff], lut[(pk >> 8) & 0xff], lut[(pk >> 16) & 0xff], lut[pk >> 24]);
    }
    for (size_t i = 4 * nq + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += stride) out[i] = lut[in[i]];
}


// float32 node outputs -> their uint8 codes (the compact form the host pipeline moves over PCIe): mode 0: value k / 255 -> k
// (exact: fl(k / 255) * 255 lies within half an ulp-scale of k); mode 1: non-zero flag (mask).  `stride` floats between
// consecutive values (3: one code per pixel of a three-equal-channel depth map).
__global__ void __launch_bounds__(256) k_pack_u8(const float* __restrict__ in, uint8_t* __restrict__ out, size_t count, int stride,
                                                 int mode) {
    const size_t nq = count / 4, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nq; i += step) {
        uint32_t pk = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float v = in[(4 * i + j) * (size_t)stride];
            const uint32_t k = mode ? (v != 0.0f ? 1u : 0u) : (uint32_t)__builtin_rintf(v * 255.0f);
            pk |= (k & 0xffu) << (8 * j);
        }
        reinterpret_cast<uint32_t*>(out)[i] = pk;
    }
    for (size_t i = 4 * nq + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += step) {
        const float v = in[i * (size_t)stride];
        out[i] = (uint8_t)(mode ? (v != 0.0f ? 1u : 0u) : (uint32_t)__builtin_rintf(v * 255.0f));
    }
}

// every stride-th float (one channel of a depth-map output with equal channels)
__global__ void __launch_bounds__(256) k_take_f32(const float* __restrict__ in, float* __restrict__ out, size_t count, int stride) {
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += step) out[i] = in[i * (size_t)stride];
}

// ---------------------------------------------------------------------------------------------
// stereo_shift_torch (reference stereo_utils.py:15-88): the `none` forward map on a float payload (latents [b][c][h][w]).
// One workgroup per (row, batch item): winner source column per destination by LDS atomics (the reference's sweep makes the
// largest source column win for a negative shift and the smallest for a positive one, :56-67), then a gather of all channels.
// The depth is normalised with the GLOBAL min / max of the whole depth tensor (:36-45), unfilled destinations stay 0.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_stereo_shift(const float* __restrict__ in, const float* __restrict__ depth, int c, int h,
                                                      int w, const uint32_t* stats, float scale_px32, int asc, int pow_mode,
                                                      float e32, float* __restrict__ out) {
    extern __shared__ int win[];   // [w] winning source column per destination, -1 = none
    __shared__ csm::PowfTables T;
    const int row = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid == 0) { const csm::PowfTables init = CS_POWF_TABLES_INIT; T = init; }
    for (int x = tid; x < w; x += blockDim.x) win[x] = asc ? -1 : 0x7fffffff;
    __syncthreads();
    const float mn = csm::ord2f(stats[ST_L_MIN]), mx = csm::ord2f(stats[ST_L_MAX]);
    const float rng = mx - mn;
    const bool flat = !(rng > 1.1920929e-07f);   // torch.finfo(float32).eps (:40)
    const float* drow = depth + ((size_t)b * h + row) * w;
    for (int col = tid; col < w; col += blockDim.x) {
        const float nd = flat ? 0.0f : (1.0f * (drow[col] - mn)) / rng;
        float dv;
        if (pow_mode == 1) dv = nd;                      // torch.pow special-cases 1, 2 and 0.5
        else if (pow_mode == 2) dv = nd * nd;
        else if (pow_mode == 3) dv = sqrtf(nd);
        else dv = csm::powf_exact(nd, e32, &T);          // (other exponents: libm powf; torch's scalar pow may differ in the last ulp)
        const float prod = dv * scale_px32;
        if (fabsf(prod) < 1.0e9f) {
            const int cd = col + (int)prod;              // int() truncates toward zero (:64)
            if (cd >= 0 && cd < w) {
                if (asc) atomicMax(&win[cd], col);
                else atomicMin(&win[cd], col);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < c * w; i += blockDim.x) {
        const int ch = i / w, x = i - ch * w;
        const int src = win[x];
        const size_t base = (((size_t)b * c + ch) * h + row) * w;
        out[base + x] = (src >= 0 && src < w) ? in[base + src] : 0.0f;
    }
}

__global__ void k_test_powf(const float* x, float y, float* out, size_t n) {
    __shared__ csm::PowfTables T;
    const csm::PowfTables init = CS_POWF_TABLES_INIT;
    if (threadIdx.x == 0) T = init;
    __syncthreads();
    // both forms: the SIMT-shaped one the kernels use must equal the plain one bit for bit
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = csm::powf_exact(x[i], y, &T), b = csm::powf_exact_simt(x[i], y, &T);
        out[i] = csm::f2u(a) == csm::f2u(b) ? a : csm::u2f(0x7fc00001u);
        // the exponent shortcuts of the tile kernel (x >= 0 only, like |nd|) must agree as well
        if (x[i] >= 0.0f && y == 1.0f && csm::f2u(x[i]) != csm::f2u(a)) out[i] = csm::u2f(0x7fc00002u);
        if (x[i] >= 0.0f && y == 2.0f) {
            bool risky;
            const float sq = csm::square_or_flag(x[i], risky);
            if (!risky && csm::f2u(sq) != csm::f2u(a)) out[i] = csm::u2f(0x7fc00003u);
        }
    }
}

__device__ const unsigned long long d_exp_tab[256] = {CS_EXP_TAB_VALUES};

__global__ void k_test_exp(const double* x, double* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = csm::exp_exact(x[i], d_exp_tab);
}

// ---- measurement hook (cs_profile / cs_profile_read) ------------------------------------------
// The event pool is process-wide opt-in state (cs_profile); every access holds g_prof_mu, so concurrent cs_generate
// calls on different streams / threads may record into it.
static const int PROF_MAX = 4096;
static std::mutex g_prof_mu;
static std::atomic<bool> g_prof_on{false};
static hipEvent_t g_prof_ev[2 * PROF_MAX];
static int g_prof_made = 0, g_prof_used = 0;
// the lazy-tile map of the last profiled cs_generate call (null: the call's warp kernel read complete blurred maps)
// share of the lazy blur tiles the profiled calls wrote (cs_profile_tiles): counted on the device into a LIBRARY-OWNED pair of
// words {tiles set, tiles in all}, accumulated over the chunks of every profiled cs_generate -- no pointer into a caller's
// workspace outlives the call
static unsigned long long* g_prof_tiles = nullptr;
static bool g_prof_tiles_any = false;
static hipStream_t g_prof_tiles_stream = nullptr;
__global__ void __launch_bounds__(256) k_prof_count_tiles(const uint32_t* __restrict__ map, int rows, int words, int tcols,
                                                          unsigned long long* acc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned cnt = 0;
    if (i < rows * words) {
        const int wi = i % words, first = wi * 32;
        const int valid = tcols - first;   // tiles of the row this word holds (the pad word and the tail bits: none)
        const uint32_t m = valid >= 32 ? 0xffffffffu : (valid <= 0 ? 0u : (1u << valid) - 1u);
        cnt = __popc(map[i] & m);
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&acc[0], (unsigned long long)cnt);
    if (i == 0) atomicAdd(&acc[1], (unsigned long long)rows * (unsigned long long)tcols);
}
struct ProfScope {
    hipStream_t s; int slot;
    explicit ProfScope(hipStream_t stream) : s(stream), slot(-1) {
        if (!g_prof_on.load(std::memory_order_relaxed)) return;
        std::lock_guard<std::mutex> lock(g_prof_mu);
        if (g_prof_used >= PROF_MAX) return;
        while (g_prof_made <= g_prof_used) {
            if (hipEventCreate(&g_prof_ev[2 * g_prof_made]) != hipSuccess) return;
            if (hipEventCreate(&g_prof_ev[2 * g_prof_made + 1]) != hipSuccess) return;
            g_prof_made++;
        }
        slot = g_prof_used++;
        (void)hipEventRecord(g_prof_ev[2 * slot], s);
    }
    ~ProfScope() {
        if (slot < 0) return;
        std::lock_guard<std::mutex> lock(g_prof_mu);
        (void)hipEventRecord(g_prof_ev[2 * slot + 1], s);
    }
};

static int threads_for(int fill, int w);
static int grid_for(size_t items, int threads) {
    size_t b = (items + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// eye geometry / parameters for one call (reference :1533-1541, :1060-1065)
static void eye_setup(EyeArgs& E, double div_percent_signed, double sep_percent_signed, int w) {
    double div_px = (div_percent_signed / 100.0) * (double)w;
    double sep_px = (sep_percent_signed / 100.0) * (double)w;
    E.div32 = (float)div_px;
    E.sep32 = (float)sep_px;
    E.div64 = div_px;
    E.sep64 = sep_px;
    E.asc = div_px < 0 ? 1 : 0;
    E.naive_lim = abs((int)div_px) + 2;
    E.csg_cap = 5 * (int)fabs(div_px) + 25;
    E.enabled = 1;
    E.xoff = E.yoff = 0;
}

// halo of the tiled polylines path: every source column within S of an output pixel can reach it.
// |nd| <= max(|c|, |1-c|) because the normalised depth lies in [0, 1] (flat depth: nd = -c).
static int poly_halo(double div_percent_a, double div_percent_b, double sep_percent, double exponent, double conv, int w) {
    if (!(exponent >= 0.0)) return 1 << 20;  // |nd|^e is unbounded near 0 for e < 0: no finite halo
    double m = fmax(fabs(conv), fabs(1.0 - conv));
    double d = fmax(fabs(div_percent_a), fabs(div_percent_b)) / 100.0 * w;
    double s = d * pow(m, exponent) * 1.0001 + fabs(sep_percent / 100.0 * w);
    if (!(s < 1e6)) return 1 << 20;
    return (int)ceil(s) + 2;
}

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
// techniques with a D64 (numba typing) instantiation: the forward-map family, the z-buffered inverse map and -- through the
// general row kernel only, the sweep of full D64 as a literal one-lane replay -- the polylines techniques
// (round 5: also the three techniques no UI string reaches -- none_post / inverse_post: their mapping functions are @njit, the
// float64 offset chain is all numba changes; hybrid_edge_plus = hybrid_edge + polylines_soft, both of which have the dialect)
static bool dialect_d64_ok(int fill) { return fill >= 0 && fill <= CS_FILL_HYBRID_EDGE_PLUS && fill != CS_FILL_GPU_WARP; }
static const char* const DIALECT_MSG = "dialect D64 (flags bits 3/4) exists for the CPU techniques, not for gpu_warp (torch arithmetic in both installs)";
// flags of the rows the tiled polylines path hands to the row kernel + their compacted list (run_rows)
// flagged-row block: [row flags, one byte per row][count / cursor pairs, 256 B][stretch-replay counters, 256 B][replay retry
// flags, one byte per row] -- everything one memset clears -- then [row list, 4 B per row]
// (round 5: + [tile hints, one 32-bit word per row and eye: bit t = tile t of that row-eye could not be finished by k_polypoint]
// behind the retry flags, inside the cleared part)
// (round 6: + [second-tier row flags, one byte per row][second-tier tile hints] behind the hints, inside the cleared part: what
// k_polypoint_listed flags among the rows the first tier handed it)
static size_t rowflag_hint_off(size_t rows) { return 2 * al256(rows) + 512; }
static size_t rowflag_flag2_off(size_t rows) { return rowflag_hint_off(rows) + al256(rows * 8); }
static size_t rowflag_hint2_off(size_t rows) { return rowflag_flag2_off(rows) + al256(rows); }
static size_t rowflag_clear_bytes(size_t rows) { return rowflag_hint2_off(rows) + al256(rows * 8); }
static size_t rowflag_bytes(size_t rows) { return rowflag_clear_bytes(rows) + al256(rows * 4); }
static uint32_t* rowflag_list(uint8_t* rowflag, size_t rows) { return (uint32_t*)(rowflag + rowflag_clear_bytes(rows)); }

// polylines: tiled fast path + general row kernel over the rows it flagged; everything else: row kernel
// anaglyph scratch of the tiled polylines path: both eyes as uint8 codes side by side
static size_t poly_anaglyph_bytes(int n, int h, int w) { return al256((size_t)n * h * 2 * w * 3); }

// Anaglyph modes of the forward fills and the two `_post` techniques beyond the width their row kernel takes WITH its anaglyph stash (two
// more bytes of LDS per column): the row kernel runs in its side-by-side form into uint8 scratch -- every row, no tile kernel -- and
// k_anaglyph_compose makes the composite, as the polylines techniques do since round 5.  Row-kernel speed, for frames that were refused
// before (round 6: naive_interpolating 7 365 -> 8 104 columns, i.e. an 8K anaglyph; none 10 128 -> 11 578, naive / inverse 8 104 -> 9 004).
static bool ana_wide_fill(int fill) {
    return fill == CS_FILL_NONE || fill == CS_FILL_NAIVE || fill == CS_FILL_NAIVE_INTERPOLATING || fill == CS_FILL_INVERSE ||
           fill == CS_FILL_NONE_POST || fill == CS_FILL_INVERSE_POST;
}
static int stash_form_max_width(int fill) {   // widest row the row kernel takes in its own anaglyph form
    int lo = 0, hi = 1 << 16;
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (rowwarp_lds_bytes(fill, mid, 1) <= CS_LDS_BYTES && poly_npt(mid, 1) < 65535) lo = mid; else hi = mid - 1;
    }
    return lo;
}
static bool ana_wide_call(int fill, int anaglyph, int w) { return anaglyph && ana_wide_fill(fill) && w > stash_form_max_width(fill); }

static int run_rows(int fill, RowArgs& A, int halo, uint8_t* rowflag, hipStream_t stream, uint8_t* ana_sbs = nullptr,
                    void* replay_scratch = nullptr, size_t replay_surplus = 0) {
    const bool poly = fill == CS_FILL_POLYLINES_SOFT || fill == CS_FILL_POLYLINES_SHARP;
    // anaglyph modes of the tiled polylines path (round 5): BOTH kernels -- the tile kernel and the row kernel over the rows it
    // flags -- write the two eyes as uint8 codes side by side into scratch (no mask), and k_anaglyph_compose makes the composite
    // of every row afterwards.  The row kernel then runs in its side-by-side form: no anaglyph stash in LDS (the width limit of
    // the side-by-side modes: polylines_sharp 7 990 instead of 6 395 columns), and its order-dependent stretches go to the replay
    // kernel like those of the other modes.  (Rounds 2-4 let the row kernel write flagged rows in final anaglyph form.)
    // soft / sharp: the point-owner kernel (cs_polypoint.hip) unless the halo is too wide for it or the development switch
    // CS_DEBUG_PT_VARIANT asks for the first generation (cs_polytile.hip)
    const int variant = dev_switch(CS_DEBUG_PT_VARIANT);   // 0 / 3 .. 7: point-owner kernel; 41 - 43: tie-path what-ifs; other values: first generation
    auto polypoint_takes = [&](int hl) {
        return hl <= polypoint_max_halo() && (variant == 0 || (variant >= 3 && variant <= 7) || (variant >= 13 && variant <= 16) || (variant >= 41 && variant <= 50));
    };
    // (the tile kernels are dialect D32, plus -- round 5 -- the float64 disparity chain alone: k_polypoint<..., DIA>)
    // (round 6: numba's typing of the sweep as well -- d64 & 2, k_polypoint<..., SW> -- in the point kernel's default geometry)
    const bool tile_dialect = A.d64 == 0 || (A.d64 == 1 && polypoint_takes(halo)) ||
                              ((A.d64 & 2) && polypoint_takes(halo) && polypoint_sweep64_ok(A.w, halo) && !dev_switch(CS_DEBUG_PT_VARIANT));
    const bool ana_tiled = poly && tile_dialect && A.anaglyph && ana_sbs && A.image_f32 && !A.out_u8 && halo <= polytile_max_halo() && rowflag &&
                           !dev_switch(CS_DEBUG_NO_TILE);
    const bool ana_wide = ana_wide_call(fill, A.anaglyph, A.w) && ana_sbs && A.image_f32 && !A.out_u8;
    const RowArgs Afinal = A;
    if (ana_tiled || ana_wide) {
        A.anaglyph = 0; A.single = -1;
        A.stereo = reinterpret_cast<float*>(ana_sbs); A.stereo_is_u8 = 1; A.no_mask = 1; A.mask = nullptr;
        A.out_h = A.h; A.out_w = 2 * A.w;
        A.eye[0].xoff = 0; A.eye[0].yoff = 0; A.eye[1].xoff = A.w; A.eye[1].yoff = 0;
    }
    // polylines, eyes in separate output slots: the stretches of order-dependent rows are replayed by a kernel of their own
    // (cs_rowwarp.hip k_poly_replay) instead of inside the row kernel
    const bool replay = poly && !A.d64 && replay_scratch && rowflag && !A.anaglyph && !dev_switch(CS_DEBUG_NO_REPLAY_KERNEL) &&
                        poly_replay_bytes(A.n, A.h, A.w, fill == CS_FILL_POLYLINES_SHARP) > 0;
    int hint_T = 0;         // tile width of k_polypoint when it took the call (its hint words are then valid)
    bool cleared = false;   // the flagged-row block (row flags, counters, replay counters and retry flags) has been zeroed
    if (replay)
        (void)poly_replay_attach(A, fill == CS_FILL_POLYLINES_SHARP, replay_scratch, rowflag + al256((size_t)A.n * A.h) + 256, stream, replay_surplus);
    if (poly && tile_dialect && !A.anaglyph && halo <= polytile_max_halo() && rowflag &&
        !dev_switch(CS_DEBUG_NO_TILE)) {   // (anaglyph calls arrive here in their side-by-side form)
        // workspace: [n*h flag bytes][count, padded to 256][n*h list entries]
        const size_t rows = (size_t)A.n * A.h;
        uint32_t* count = (uint32_t*)(rowflag + al256(rows));
        uint32_t* list = rowflag_list(rowflag, rows);
        hipError_t e = hipMemsetAsync(rowflag, 0, rowflag_clear_bytes(rows), stream);
        if (e != hipSuccess) return fail_hip(e, "rowflag memset");
        cleared = true;
        const RowArgs& T = A;
        if (polypoint_takes(halo))
            e = launch_polypoint(T, halo, rowflag, stream, fill == CS_FILL_POLYLINES_SHARP, (uint32_t*)(rowflag + rowflag_hint_off(rows)), &hint_T);
        else
            e = launch_polytile(fill == CS_FILL_POLYLINES_SHARP, T, halo, rowflag, stream);
        if (e != hipSuccess) return fail_hip(e, "tiled polylines launch");
        // single-eye modes (left-only / only-right): the tile kernels visit one eye, both depth maps are outputs all the same
        if (A.neyes == 2 && A.single >= 0 && !A.out_u8) {
            const int other = 1 - A.single;
            float* dd = other == 0 ? A.depth_l : A.depth_r;
            if (dd) {
                e = launch_depth_codes(A.eye[other].depth, A.n, A.h, A.w, A.stats, A.scale_from_stats, dd, stream);
                if (e != hipSuccess) return fail_hip(e, "depth-map output launch");
            }
        }
        e = launch_collect_rows(rowflag, (int)rows, count, list, stream);
        if (e != hipSuccess) return fail_hip(e, "flagged-row collection");
        A.row_list = list; A.row_count = count;
        const uint32_t* hints = (const uint32_t*)(rowflag + rowflag_hint_off(rows));
        // (round 6) second tier: the flagged rows through the point kernel once more with 512 instead of 160 slots for pixels under
        // reversed segments and longer per-pixel lists (k_polypoint_listed); what THAT flags -- a second flag array and hint block,
        // a third {count, cursor} pair -- is collected into the same list for the row kernel.  Depth maps with strong silhouettes
        // (tools/synth.scene8) overflowed the first tier's lists in 15 % (soft) / 48 % (sharp) of the rows at the metric's divergence
        // and sent them to the row kernel: 4 200 / 1 060 frames/s against 5 380 / 4 050 on stepped depth.  polylines_sharp only:
        // for soft the lean row kernel on the hinted tiles' columns is the cheaper second stop (0.96 against 1.16 ms per 16 frames,
        // tools/sessions/r06_s11.sh, s12: 4 100 against 3 750 frames/s; CS_DEBUG_PT_VARIANT 50 forces the tier for soft, 49: off)
        // (numba's sweep, d64 & 2: both techniques -- what the tier does not finish takes the dialect row kernel whole, there is no lean pass)
        if (hint_T > 0 && (fill == CS_FILL_POLYLINES_SHARP || (A.d64 & 2) || dev_switch(CS_DEBUG_PT_VARIANT) == 50) && dev_switch(CS_DEBUG_PT_VARIANT) != 49) {
            uint8_t* flag2 = rowflag + rowflag_flag2_off(rows);
            uint32_t* hint2 = (uint32_t*)(rowflag + rowflag_hint2_off(rows));
            uint32_t* count3 = count + 16;
            e = launch_polypoint_tier2(T, halo, flag2, stream, fill == CS_FILL_POLYLINES_SHARP, hint2, hint_T, list, count);
            if (e == hipSuccess) {
                e = launch_collect_rows(flag2, (int)rows, count3, list, stream);
                if (e != hipSuccess) return fail_hip(e, "flagged-row collection (second tier)");
                A.row_count = count3;
                hints = hint2;
            } else if (e != hipErrorNotSupported) return fail_hip(e, "tiled polylines launch (second tier)");
        }
        // (the lean first pass works on the flagged tiles' column ranges; CS_DEBUG_PT_VARIANT 44: whole rows as in round 4)
        if (hint_T > 0 && dev_switch(CS_DEBUG_PT_VARIANT) != 44) {
            A.hint = hints; A.hint_T = hint_T; A.hint_S = halo;
        }
    }
    if ((fill == CS_FILL_NONE || fill == CS_FILL_INVERSE || fill == CS_FILL_NAIVE || fill == CS_FILL_NAIVE_INTERPOLATING) &&
        !dev_switch(CS_DEBUG_NO_TILE) && !ana_wide) {
        // the halo-tile kernel where it applies; 'naive' hands the rows it cannot decide to the row kernel
        const bool flagging = (fill == CS_FILL_NAIVE || fill == CS_FILL_NAIVE_INTERPOLATING) && rowflag;
        const size_t rows = (size_t)A.n * A.h;
        if (flagging) {
            hipError_t e = hipMemsetAsync(rowflag, 0, rowflag_clear_bytes(rows), stream);
            if (e != hipSuccess) return fail_hip(e, "rowflag memset");
        }
        hipError_t e = launch_fwdtile(fill, A, halo, flagging ? rowflag : nullptr, stream);
        if (e == hipSuccess) {
            if (!flagging) return CS_OK;
            uint32_t* count = (uint32_t*)(rowflag + al256(rows));
            uint32_t* list = rowflag_list(rowflag, rows);
            e = launch_collect_rows(rowflag, (int)rows, count, list, stream);
            if (e != hipSuccess) return fail_hip(e, "flagged-row collection");
            A.row_list = list; A.row_count = count;
            // naive_interpolating, second tier (round 5): the flagged rows through the tile kernel once more with a window that holds every
            // hole; what THAT flags (the retry-flag bytes of the block) is collected into the same list for the row kernel.  On depth
            // saturated to 0 / 1 the holes between a near and a far plateau outgrow the first window in most rows (1 830 frames/s at 4K)
            if (fill == CS_FILL_NAIVE_INTERPOLATING && dev_switch(CS_DEBUG_PT_VARIANT) != 47) {
                uint8_t* flag2 = rowflag + al256(rows) + 512;
                uint32_t* count2 = count + 8;
                e = launch_fwdtile(fill, A, halo, flag2, stream, list, count);
                if (e == hipSuccess) {
                    e = launch_collect_rows(flag2, (int)rows, count2, list, stream);
                    if (e != hipSuccess) return fail_hip(e, "flagged-row collection (second tier)");
                    A.row_count = count2;
                } else if (e != hipErrorNotSupported) return fail_hip(e, "forward tile kernel launch (second tier)");
            }
        } else if (e != hipErrorNotSupported) return fail_hip(e, "forward tile kernel launch");
    }
    if (A.tilemap) {
        // lazy depth-blur tiles: the row kernels read complete rows of the blurred maps -- fill in the unwritten tiles of the
        // rows they are about to visit (the rows a tile kernel flagged; every row when no tile kernel took the call)
        hipError_t e = launch_lazy_rows(A.row_list, A.row_count, A.n * A.h, A.lazy_gray, const_cast<float*>(A.eye[0].depth),
                                        const_cast<float*>(A.eye[1].depth), A.tilemap, A.stats, A.h, A.w, stream);
        if (e != hipSuccess) return fail_hip(e, "lazy depth rows");
    }
    if (replay && !cleared) {   // (no tile kernel took the call: the block has not been cleared yet)
        hipError_t e0 = hipMemsetAsync(rowflag, 0, rowflag_clear_bytes((size_t)A.n * A.h), stream);
        if (e0 != hipSuccess) return fail_hip(e0, "rowflag memset");
    }
    // (polylines with the replay kernel attached: the lean instantiation first -- evaluation, stretch list, export; the rows it
    // cannot export come back through the retry flags below.  CS_DEBUG_PT_VARIANT 43: the full kernel as in round 3)
    hipError_t e = launch_rowwarp(fill, A, threads_for(fill, A.w), stream, 0, replay && dev_switch(CS_DEBUG_PT_VARIANT) != 43);
    if (e != hipSuccess) return fail_hip(e, "row kernel launch");
    if (replay) {
        e = launch_poly_replay(fill == CS_FILL_POLYLINES_SHARP, A, halo, stream);
        if (e != hipSuccess) return fail_hip(e, "stretch replay launch");
        // rows with a stretch the replay kernel gave up on (usually none): once more through the row kernel, export off
        const size_t rows = (size_t)A.n * A.h;
        uint32_t* count2 = (uint32_t*)(rowflag + al256(rows)) + 8;   // (a second {count, cursor} pair in the cleared counter block)
        uint32_t* list = rowflag_list(rowflag, rows);
        e = launch_collect_rows(poly_replay_retry_flags(A), (int)rows, count2, list, stream);
        if (e != hipSuccess) return fail_hip(e, "replay retry collection");
        RowArgs R = A;
        R.rp_dump = nullptr; R.row_list = list; R.row_count = count2; R.hint = nullptr;
        // (one workgroup per CU: on saturated depth a few hundred rows per frame come back -- stretches whose list outgrows the
        // wave form's 64 entries --, 32 workgroups made them the tail of the call; an empty launch of 256 costs 0.03 ms)
        e = launch_rowwarp(fill, R, threads_for(fill, A.w), stream, dev_switch(CS_DEBUG_PT_VARIANT) == 42 ? 32 : 256);
        if (e != hipSuccess) return fail_hip(e, "row kernel launch (replay retry)");
    }
    if (ana_tiled || ana_wide) {   // R from one eye, G and B from the other, k / 255, the mask of the composite -- every row
        e = launch_anaglyph_compose(ana_sbs, nullptr, Afinal.n, Afinal.h, Afinal.w, Afinal.anaglyph, Afinal.stereo, Afinal.stereo_is_u8,
                                    Afinal.mask, stream);
        if (e != hipSuccess) return fail_hip(e, "anaglyph composition launch");
    }
    return CS_OK;
}

static int threads_for(int fill, int w) {
    if (w <= 256) return 256;
    if (w <= 1024) return 512;
    // wide rows: the row kernels need 110-128 VGPRs, i.e. 16 waves per CU either way -- as two 512-thread workgroups they
    // overlap each other's barrier phases (measured at 4K: none +7 %, inverse +9 %, naive_interpolating +43 %); naive and
    // the polylines row kernel prefer the single 1024-thread workgroup (naive: -22 % at 512)
    if (fill == CS_FILL_NONE || fill == CS_FILL_INVERSE || fill == CS_FILL_NAIVE_INTERPOLATING || fill == CS_FILL_NONE_POST ||
        fill == CS_FILL_INVERSE_POST)
        return 512;
    return 1024;
}

}  // namespace cs

using namespace cs;

extern "C" {

int cs_version(void) { return CS_ABI_VERSION; }
const char* cs_last_error(void) { return g_err; }

// widest frame the LDS-resident row kernels take; anaglyph modes keep two channels of the first eye per pixel as well
static int max_width_for(int fill, int anaglyph) {
    // (polylines, round 5: the anaglyph modes run both kernels in their side-by-side form and compose afterwards, run_rows above:
    // no anaglyph stash in LDS, the limit of the side-by-side modes.  A call whose halo is too wide for the tile kernels still takes
    // the row kernel's anaglyph form: cs_generate checks that case against `row_form_max_width`)
    if (fill == CS_FILL_POLYLINES_SOFT || fill == CS_FILL_POLYLINES_SHARP) anaglyph = 0;
    if (ana_wide_fill(fill)) anaglyph = 0;   // (round 6: beyond the stash form's width the row kernel runs side by side + composition, run_rows)
    if (fill == CS_FILL_GPU_WARP) return gpuwarp_max_width();   // (the mesh-quality variant, cs_params.flags bit 2: cs_forward_warp_mesh's limit)
    if (fill == CS_FILL_HYBRID_EDGE) return hybrid_max_width();
    if (fill < 0 || fill > CS_FILL_HYBRID_EDGE_PLUS) return 0;
    int lo = 0, hi = 1 << 16;
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (rowwarp_lds_bytes(fill, mid, anaglyph) <= CS_LDS_BYTES && poly_npt(mid, 1) < 65535) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
// widest row the row kernel takes in its own anaglyph form (two more bytes of LDS per column)
static int row_form_max_width(int fill, int anaglyph) {
    int lo = 0, hi = 1 << 16;
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (rowwarp_lds_bytes(fill, mid, anaglyph) <= CS_LDS_BYTES && poly_npt(mid, 1) < 65535) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
// polylines under the float64 disparity chain keep 8 more bytes per column in LDS (cs_rowwarp.hip Poly::xd)
static bool dialect_width_ok(int fill, int w, int anaglyph, int d64) {
    if (!(d64 & 1) || (fill != CS_FILL_POLYLINES_SOFT && fill != CS_FILL_POLYLINES_SHARP && fill != CS_FILL_HYBRID_EDGE && fill != CS_FILL_HYBRID_EDGE_PLUS)) return true;
    return rowwarp_lds_bytes(fill, w, fill == CS_FILL_HYBRID_EDGE ? 0 : anaglyph) + 8 * (size_t)w + 16 <= CS_LDS_BYTES;
}
// The ONE width predicate of cs_generate: nullptr when a frame of `w` columns with p's technique, mode, dialect flags
// and disparity parameters is accepted, else the reason.  cs_max_width_params searches it, so a caller that pre-validates gets
// exactly the answer the call itself would give.
static const char* width_refusal(const cs_params* p, int w) {
    const bool ana = p->mode == CS_MODE_RED_CYAN_ANAGLYPH || p->mode == CS_MODE_CYAN_RED_REVERSEANAGLYPH;
    if (w > max_width_for(p->fill, ana) || !dialect_width_ok(p->fill, w, ana, (p->flags >> 3) & 3))
        return "frame too wide for the LDS-resident row kernel";
    if ((p->fill == CS_FILL_POLYLINES_SOFT || p->fill == CS_FILL_POLYLINES_SHARP) && ana && w > row_form_max_width(p->fill, 1)) {
        // an anaglyph wider than the row kernel's anaglyph form: only through the tile kernels -- the predicate run_rows uses
        // for its tiled anaglyph path (D32, or the float64 disparity chain alone with a halo within the point kernel's reach;
        // cs_generate's images are float32)
        const int halo = poly_halo(p->divergence * (1 + p->stereo_balance), p->divergence * (1 - p->stereo_balance), p->separation,
                                   p->stereo_offset_exponent, p->convergence_point, w);
        const int d64 = (p->flags >> 3) & 3;
        const bool tile_dialect = d64 == 0 || (d64 == 1 && halo <= polypoint_max_halo()) ||
                                  ((d64 & 2) && halo <= polypoint_max_halo() && polypoint_sweep64_ok(w, halo) && !dev_switch(CS_DEBUG_PT_VARIANT));
        if (!tile_dialect || halo > polytile_max_halo() || dev_switch(CS_DEBUG_NO_TILE))
            return "frame too wide for the LDS-resident row kernel (anaglyph form)";
    }
    return nullptr;
}
int cs_max_width_params(const cs_params* p) {
    if (!p || p->fill < 0 || p->fill > CS_FILL_HYBRID_EDGE_PLUS) return 0;
    int lo = 0, hi = 1 << 16;   // (monotone: LDS bytes and the halo both grow with the width)
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (!width_refusal(p, mid)) lo = mid; else hi = mid - 1;
    }
    return lo;
}
int cs_max_width(int fill) { return max_width_for(fill, 1); }
int cs_max_width_mode(int fill, int mode) {
    return max_width_for(fill, mode == CS_MODE_RED_CYAN_ANAGLYPH || mode == CS_MODE_CYAN_RED_REVERSEANAGLYPH);
}

int cs_output_shape(const cs_params* p, int* out_h, int* out_w, int* mask_h, int* mask_w) {
    if (!p) return fail(CS_EINVAL, "null params");
    int oh = p->h, ow = p->w;
    switch (p->mode) {
    case CS_MODE_LEFT_RIGHT: case CS_MODE_RIGHT_LEFT: ow = 2 * p->w; break;
    case CS_MODE_TOP_BOTTOM: case CS_MODE_BOTTOM_TOP: oh = 2 * p->h; break;
    case CS_MODE_RED_CYAN_ANAGLYPH: case CS_MODE_CYAN_RED_REVERSEANAGLYPH:
    case CS_MODE_LEFT_ONLY: case CS_MODE_ONLY_RIGHT: break;
    default: return fail(CS_EINVAL, "Unknown mode");
    }
    if (out_h) *out_h = oh;
    if (out_w) *out_w = ow;
    // CPU techniques: the mask is computed from the assembled output (GenerateStereo.py:355-361);
    // gpu_warp: left_mask | right_mask, eye-shaped (stereoimage_generation.py:1090).
    if (mask_h) *mask_h = p->fill == CS_FILL_GPU_WARP ? p->h : oh;
    if (mask_w) *mask_w = p->fill == CS_FILL_GPU_WARP ? p->w : ow;
    return CS_OK;
}


struct WsLayout {
    size_t rowflag, gray_src, gray, L, R, wl, wr, tilemap, extra, total;
};
// Scratch of one chunk of frames (everything but the per-frame statistics words, which stay one array for the whole call
// at the start of the workspace so that callers find them at a fixed place).
static WsLayout ws_layout(const cs_params* p) {
    WsLayout W;
    size_t hw = (size_t)p->h * p->w, n = p->n, o = 0;
    W.rowflag = o; o += rowflag_bytes(n * (size_t)p->h);
    bool resize = p->depth_h != p->h || p->depth_w != p->w;
    W.gray_src = o; if (resize) o += al256(n * (size_t)p->depth_h * p->depth_w * 4);
    bool blur = p->depth_map_blur && p->depth_blur_strength > 0;  // strength <= 0: the reference returns the depth as is (:1194)
    // (the gray buffer sits BETWEEN the blurred maps: the lazy-tile readers address either of a pair with a 32-bit offset)
    W.L = o; if (blur) o += al256(n * hw * 4);
    W.gray = o; o += al256(n * hw * 4);
    W.R = o; if (blur) o += al256(n * hw * 4);
    W.wl = o; if (blur) o += al256(n * hw * 4);
    W.wr = o; if (blur) o += al256(n * hw * 4);
    W.tilemap = o; if (blur) o += al256(blur_tilemap_bytes(p->n, p->h, p->w));
    W.extra = o;
    if (p->fill == CS_FILL_HYBRID_EDGE || p->fill == CS_FILL_HYBRID_EDGE_PLUS) o += al256(hybrid_workspace_bytes(p->n, p->h, p->w));
    if (p->fill == CS_FILL_GPU_WARP) o += al256(gpuwarp_workspace_bytes(p->n, p->h, p->w, p->batch_size, p->flags & 4));
    if ((p->fill == CS_FILL_POLYLINES_SOFT || p->fill == CS_FILL_POLYLINES_SHARP) &&
        (p->mode == CS_MODE_RED_CYAN_ANAGLYPH || p->mode == CS_MODE_CYAN_RED_REVERSEANAGLYPH))
        o += poly_anaglyph_bytes(p->n, p->h, p->w);   // (both eyes as uint8 codes side by side; the replay scratch behind it)
    if (ana_wide_call(p->fill, p->mode == CS_MODE_RED_CYAN_ANAGLYPH || p->mode == CS_MODE_CYAN_RED_REVERSEANAGLYPH, p->w))
        o += poly_anaglyph_bytes(p->n, p->h, p->w);   // (the same scratch for an anaglyph too wide for the row kernel's stash form)
    if (p->fill == CS_FILL_POLYLINES_SOFT || p->fill == CS_FILL_POLYLINES_SHARP)   // scratch of the stretch replay kernel
        o += al256(poly_replay_bytes(p->n, p->h, p->w, p->fill == CS_FILL_POLYLINES_SHARP));
    W.total = o;
    return W;
}

// ---- frame chunks: the pre-pass of chunk c + 1 under the warp of chunk c ---------------------------------------------
// A call can cut its batch into chunks of frames -- every quantity of the path is per frame (gpu_warp: per reference
// sub-batch, so its chunks are multiples of batch_size) -- and issue the pre-pass of every chunk (gray depth, edge bit rows,
// tile classification, fused blur: HBM-bound) on an auxiliary stream, the warps on the caller's stream, one event per chunk
// in between, so that the pre-pass of chunk c + 1 runs under the warp of chunk c.  Fork / join is by events only (legal
// under stream capture); all work the call enqueues is ordered before whatever the caller enqueues next on its stream.
// MEASURED (round 3, profiles/r03_overlap.txt): the schedule is correct and overlaps, and it is a zero-sum game -- the warp
// kernels are bound by the latency of their resident workgroups (k_polypoint: time x workgroups per CU is constant from 3
// to 6 workgroups per CU), every co-resident pre-pass workgroup takes wave slots and memory queue from them: 64 4K frames
// polylines_soft 14.3 ms in one chunk, 14.5 / 14.8 / 15.1 / 15.6 ms in 2 / 4 / 8 / 16 chunks with the auxiliary stream at
// the highest priority, 14.3 / 14.5 / 14.8 in 2 / 4 / 8 at the caller's priority; cfg 3 / 4 / 5 gain 0.1 - 2 %.  The
// default is therefore ONE chunk on the caller's stream; cs_debug_set(CS_DEBUG_CHUNKS, k) with k > 1 cuts k chunks
// (+100: auxiliary stream at the default priority instead of the highest).
struct ChunkPlan { int nch, cf; };
static ChunkPlan plan_chunks(const cs_params* p) {
    const int n = p->n, forced = dev_switch(CS_DEBUG_CHUNKS) % 100;   // (+100: auxiliary stream at the default priority)
    int unit = 1;
    if (p->fill == CS_FILL_GPU_WARP) unit = p->batch_size > 0 ? (p->batch_size < n ? p->batch_size : n) : n;
    if (forced <= 1 || n < 2 * unit) return ChunkPlan{1, n};
    int cf = (n + forced - 1) / forced;
    cf = (cf + unit - 1) / unit * unit;
    const int nch = (n + cf - 1) / cf;
    if (nch < 2) return ChunkPlan{1, n};
    return ChunkPlan{nch, cf};
}
static size_t ws_total(const cs_params* p) {
    const ChunkPlan C = plan_chunks(p);
    cs_params q = *p;
    q.n = C.cf;
    return al256((size_t)p->n * ST_WORDS * 4) + (size_t)C.nch * ws_layout(&q).total;
}

// the auxiliary stream (highest priority the device offers) and its fork / ready events: one set per device, created on
// first use, shared by every caller (the events order each call's own work; sharing the stream only serialises the
// pre-passes of concurrent callers)
static const int AUX_MAX_DEV = 16, AUX_EVENTS = 8;
struct AuxStream { hipStream_t s; hipEvent_t fork, ready[AUX_EVENTS]; bool ok; };
static std::mutex g_aux_mu;
static AuxStream g_aux[AUX_MAX_DEV][2];
// low = 0: the highest priority the device offers, 1: the lowest (the pre-pass only fills what the warp leaves free)
static AuxStream* aux_stream(int low) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= AUX_MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lock(g_aux_mu);
    AuxStream& A = g_aux[dev][low ? 1 : 0];
    if (A.ok) return &A;
    int least = 0, greatest = 0;   // (the numerically lowest value is the highest priority)
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
    if (hipStreamCreateWithPriority(&A.s, hipStreamNonBlocking, low ? least : greatest) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&A.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    for (int i = 0; i < AUX_EVENTS; i++)
        if (hipEventCreateWithFlags(&A.ready[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    A.ok = true;
    return &A;
}

size_t cs_workspace_bytes(const cs_params* p) { return (p && p->n > 0) ? ws_total(p) : 0; }

// One chunk of frames: the pre-pass on `sp`, the warp on `stream` (the same stream, or `ready` recorded on sp and awaited
// by stream in between).  `stats`: the chunk's slice of the call's statistics words; `ws`: the chunk's scratch (ws_layout).
static int generate_chunk(const cs_params* p, const float* image, const float* depth, float* stereo, float* depth_l,
                          float* depth_r, float* mask, uint32_t* stats, char* ws, int out_h, int out_w, hipStream_t sp,
                          hipStream_t stream, hipEvent_t ready, size_t surplus = 0) {
    int rc;
    const WsLayout W = ws_layout(p);
    float* gray = (float*)(ws + W.gray);
    const int n = p->n, h = p->h, w = p->w, hw = h * w;
    const bool gpu_warp = p->fill == CS_FILL_GPU_WARP;
    const bool blur = p->depth_map_blur && p->depth_blur_strength > 0;  // strength <= 0 == blur off (reference :1194, :1050)

    hipLaunchKernelGGL(k_stats_init, dim3((n * ST_WORDS + 255) / 256), dim3(256), 0, sp, stats, n);
    const bool resize = p->depth_h != h || p->depth_w != w;
    // RGB depth + blur: the gray conversion also builds the blur's edge bit rows (one pass over the depth input)
    const bool blur_map = !dev_switch(CS_DEBUG_BLUR_FULL_COPY);
    const bool pre_edges = blur && !resize && p->depth_c == 3 &&
                           blur_pre_edges_ok(n, h, w, p->depth_blur_strength, p->depth_blur_edge_threshold, p->depth_blur_strength,
                                             p->depth_blur_vert_smooth, blur_map);
    if (resize) {
        float* gs = (float*)(ws + W.gray_src);
        int shw = p->depth_h * p->depth_w;
        hipLaunchKernelGGL(k_gray, dim3(grid_for(shw / 32 + 1, 256), n), dim3(256), 0, sp, depth, gs, shw, p->depth_c, stats, 0);
        hipLaunchKernelGGL(k_resize_bilinear, dim3(grid_for(hw, 256), n), dim3(256), 0, sp, gs, p->depth_h, p->depth_w, gray, h, w, stats);
    } else if (pre_edges) {
        hipError_t e = launch_gray_edges(depth, gray, n, h, w, stats, p->depth_blur_strength, p->depth_blur_edge_threshold,
                                         p->depth_blur_strength, p->depth_blur_vert_smooth, (float*)(ws + W.wl), (float*)(ws + W.wr), sp);
        if (e != hipSuccess) return fail_hip(e, "gray + edge pass");
    } else {
        // (>= 8 float4 groups per thread: the per-workgroup min/max reduction is amortised -- 3x faster at 1080p)
        hipLaunchKernelGGL(k_gray, dim3(grid_for(hw / 32 + 1, 256), n), dim3(256), 0, sp, depth, gray, hw, p->depth_c, stats, 1);
    }
    hipLaunchKernelGGL(k_finalize_stats, dim3((n + 63) / 64), dim3(64), 0, sp, stats, n,
                       gpu_warp ? (p->batch_size > 0 ? (p->batch_size < n ? p->batch_size : n) : n) : 0, blur ? 1 : 0);
    const float* dL = gray;
    const float* dR = gray;
    int scale_from_stats = 1, lazy = 0;
    const int halo = poly_halo(p->divergence * (1 + p->stereo_balance), p->divergence * (1 - p->stereo_balance), p->separation,
                               p->stereo_offset_exponent, p->convergence_point, w);
    if (blur) {
        float* L = (float*)(ws + W.L);
        float* R = (float*)(ws + W.R);
        // lazy tiles: the tile kernels (cs_polypoint / cs_polytile / cs_fwdtile) read the edge-free tiles of the blurred depth
        // -- a scaled copy of the gray depth -- from the gray depth itself, so the blur leaves them unwritten and hands over
        // the map of the tiles it did write; run_rows completes the rows that still go to a row kernel.  (Round 2 measured the
        // selector in k_gpuwarp and in the hybrid splat as dearer than the copy; with both kernels restructured at the end of
        // round 3 -- loads first, four workgroups per CU -- it is cheaper: see warp_lazy / hyb_lazy below.)
        const bool tile_fill = p->fill == CS_FILL_POLYLINES_SOFT || p->fill == CS_FILL_POLYLINES_SHARP || p->fill == CS_FILL_NONE ||
                               p->fill == CS_FILL_INVERSE || p->fill == CS_FILL_NAIVE || p->fill == CS_FILL_NAIVE_INTERPOLATING;
        // gpu_warp (scatter-round warp, rows of at most 4096 columns) reads the map too since the end of round 3: the kernel is
        // bound by its dependent chains, not by its instruction count -- the selector costs less than the copy of the tiles
        const bool warp_lazy = gpu_warp && !(p->flags & 4) && w <= gpuwarp_lazy_max_width() && !dev_switch(CS_DEBUG_GPUWARP_FULL_MAPS);
        // hybrid_edge through the fused splat tile kernel (two-eye layouts): the same; the disabled-eye path and the unfused
        // forms read complete maps, so the map is only used when both eyes run
        const bool two_eyes = !(p->divergence * (1 + p->stereo_balance) < 0.001) && !(p->divergence * (1 - p->stereo_balance) < 0.001);
        const bool ana = p->mode == CS_MODE_RED_CYAN_ANAGLYPH || p->mode == CS_MODE_CYAN_RED_REVERSEANAGLYPH;
        const bool hyb_lazy = p->fill == CS_FILL_HYBRID_EDGE && two_eyes && !dev_switch(CS_DEBUG_HYBRID_FULL_MAPS) &&
                              hybrid_fused_ok(n, w, halo, ana, -1, (p->flags >> 3) & 3, 0);
        // (round 6: the polylines techniques under a dialect flag too -- the point kernel's dialect instantiations share the tile function,
        // lazy loads included, and run_rows completes the rows of any call a row kernel ends up taking; the other techniques' dialect
        // kernels keep complete maps: 0.94 of 17.1 ms per 64 frames under D64, profiles/r06_s39/)
        const int d64f = (p->flags >> 3) & 3;
        const bool poly_fill = p->fill == CS_FILL_POLYLINES_SOFT || p->fill == CS_FILL_POLYLINES_SHARP;
        const bool dialect_lazy = d64f == 0 || (poly_fill && halo <= polypoint_max_halo() && (d64f == 1 || polypoint_sweep64_ok(w, halo)));
        const bool want_lazy = (tile_fill || warp_lazy || hyb_lazy) && dialect_lazy && !dev_switch(CS_DEBUG_NO_TILE) && !dev_switch(CS_DEBUG_BLUR_FULL_COPY) &&
                               p->mode != CS_MODE_LEFT_ONLY && p->mode != CS_MODE_ONLY_RIGHT &&
                               al256((size_t)n * hw * 4) < (1ull << 32) - (1u << 20);
        rc = launch_blur(gray, n, h, w, p->depth_blur_strength, p->depth_blur_edge_threshold, p->depth_blur_strength, p->depth_blur_falloff,
                         p->depth_blur_vert_smooth, L, R, (float*)(ws + W.wl), (float*)(ws + W.wr), stats, 1, sp,
                         blur_map ? (uint32_t*)(ws + W.tilemap) : nullptr, want_lazy ? &lazy : nullptr, pre_edges ? 1 : 0);
        if (rc) return fail_blur(rc);
        dL = L; dR = R;
        scale_from_stats = 0;  // the blur kernel already wrote scaled depth
        if (lazy && g_prof_on.load(std::memory_order_relaxed)) {   // (measurement: cs_profile_tiles; outside the timed kernel scope)
            std::lock_guard<std::mutex> lock(g_prof_mu);
            if (g_prof_tiles) {
                const int words = blur_tilemap_words(w), rows = (int)(blur_tilemap_bytes(n, h, w) / ((size_t)words * 4));
                hipLaunchKernelGGL(k_prof_count_tiles, dim3((rows * words + 255) / 256), dim3(256), 0, sp, (const uint32_t*)(ws + W.tilemap),
                                   rows, words, (w + 63) / 64, g_prof_tiles);
                g_prof_tiles_any = true; g_prof_tiles_stream = sp;
            }
        }
    }
    const double left_div = p->divergence * (1 + p->stereo_balance);
    const double right_div = p->divergence * (1 - p->stereo_balance);

    if (sp != stream) {   // the warp of this chunk waits for its pre-pass; the next chunk's pre-pass runs under it
        hipError_t e = hipEventRecord(ready, sp);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, ready, 0);
        if (e != hipSuccess) return fail_hip(e, "chunk hand-over");
    }
    ProfScope prof(stream);
    if (gpu_warp) {
        rc = launch_gpuwarp_node(p, image, dL, dR, scale_from_stats, stats, stereo, depth_l, depth_r, mask, out_h, out_w,
                                 ws + W.extra, stream, lazy ? (const uint32_t*)(ws + W.tilemap) : nullptr, lazy ? gray : nullptr,
                                 lazy ? blur_tilemap_words(w) : 0);
        if (rc) return fail(rc, "gpu_warp launch failed");
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? CS_OK : fail_hip(e, "cs_generate");
    }

    RowArgs A;
    memset(&A, 0, sizeof(A));
    A.n = n; A.h = h; A.w = w;
    A.image_f32 = image;
    A.stats = stats; A.stats_rw = stats;
    A.scale_from_stats = scale_from_stats;
    A.e32 = (float)p->stereo_offset_exponent;
    A.conv32 = (float)p->convergence_point;
    A.e64 = p->stereo_offset_exponent;
    A.d64 = (p->flags >> 3) & 3;
    A.neyes = 2;
    eye_setup(A.eye[0], +1 * left_div, -1 * p->separation, w);
    eye_setup(A.eye[1], -1 * right_div, p->separation, w);
    A.eye[0].enabled = !(left_div < 0.001);
    A.eye[1].enabled = !(right_div < 0.001);
    A.eye[0].depth = dL; A.eye[0].st_min = ST_L_MIN; A.eye[0].st_max = ST_L_MAX;
    A.eye[1].depth = dR; A.eye[1].st_min = ST_R_MIN; A.eye[1].st_max = ST_R_MAX;
    A.stereo = stereo; A.mask = mask; A.depth_l = depth_l; A.depth_r = depth_r;
    A.stereo_is_u8 = (p->flags & 2) ? 1 : 0;
    A.out_h = out_h; A.out_w = out_w;
    A.single = -1;
    A.dbg = dev_switch(CS_DEBUG_DBG);
    if (lazy) { A.tilemap = (const uint32_t*)(ws + W.tilemap); A.lazy_gray = gray; A.tm_words = blur_tilemap_words(w); }
    switch (p->mode) {
    case CS_MODE_LEFT_RIGHT: A.eye[1].xoff = w; break;
    case CS_MODE_RIGHT_LEFT: A.eye[0].xoff = w; break;
    case CS_MODE_TOP_BOTTOM: A.eye[1].yoff = h; break;
    case CS_MODE_BOTTOM_TOP: A.eye[0].yoff = h; break;
    case CS_MODE_RED_CYAN_ANAGLYPH: A.anaglyph = 1; break;
    case CS_MODE_CYAN_RED_REVERSEANAGLYPH: A.anaglyph = 2; break;
    case CS_MODE_LEFT_ONLY: A.single = 0; break;
    case CS_MODE_ONLY_RIGHT: A.single = 1; break;
    }
    if (p->fill == CS_FILL_HYBRID_EDGE || p->fill == CS_FILL_HYBRID_EDGE_PLUS) {
        rc = launch_hybrid(A, ws + W.extra, stream, p->fill == CS_FILL_HYBRID_EDGE_PLUS, halo);
        if (rc) return fail(rc, "hybrid_edge launch failed");
    } else {
        const bool poly_ana = A.anaglyph && (p->fill == CS_FILL_POLYLINES_SOFT || p->fill == CS_FILL_POLYLINES_SHARP);
        rc = run_rows(p->fill, A, halo, (uint8_t*)(ws + W.rowflag), stream, A.anaglyph ? (uint8_t*)(ws + W.extra) : nullptr,
                      (void*)(ws + W.extra + (poly_ana ? poly_anaglyph_bytes(p->n, p->h, p->w) : 0)), surplus);
        if (rc) return rc;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : fail_hip(e, "cs_generate");
}

int cs_generate(const cs_params* p, const float* image, const float* depth, float* stereo, float* depth_l,
                float* depth_r, float* mask, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p || !image || !depth || !stereo || !depth_l || !depth_r || !mask || !workspace) return fail(CS_EINVAL, "null pointer");
    if (p->n <= 0 || p->h <= 0 || p->w <= 0 || p->depth_h <= 0 || p->depth_w <= 0 || p->depth_c <= 0)
        return fail(CS_EINVAL, "non-positive size");
    if (p->fill < 0 || p->fill > CS_FILL_HYBRID_EDGE_PLUS) return fail(CS_EINVAL, "unknown fill technique");
    int out_h, out_w, mask_h, mask_w;
    int rc = cs_output_shape(p, &out_h, &out_w, &mask_h, &mask_w);
    if (rc) return rc;
    if (const char* why = width_refusal(p, p->w)) return fail(CS_ELIMIT, why);
    if (workspace_bytes < ws_total(p)) return fail(CS_EWORKSPACE, "workspace too small");
    if (p->fill == CS_FILL_GPU_WARP && (p->flags & 2)) return fail(CS_EINVAL, "gpu_warp colours are not k/255: no uint8 stereoscope output");
    if ((p->flags & 24) && !dialect_d64_ok(p->fill))
        return fail(CS_EINVAL, DIALECT_MSG);
    char* const ws = (char*)workspace;
    uint32_t* const stats = (uint32_t*)ws;
    char* const scratch = ws + al256((size_t)p->n * ST_WORDS * 4);
    const ChunkPlan C = plan_chunks(p);
    // (one chunk: whatever the caller's workspace holds beyond cs_workspace_bytes extends the stretch-replay pool -- the last item
    // of the layout -- so that a caller who expects tie-heavy depth maps can give every flagged row room to export)
    if (C.nch == 1) return generate_chunk(p, image, depth, stereo, depth_l, depth_r, mask, stats, scratch, out_h, out_w, stream, stream, nullptr,
                                          workspace_bytes - ws_total(p));

    // mode (CS_DEBUG_CHUNKS / 100): 0 = pre-passes on the highest-priority auxiliary stream, warps on the caller's stream;
    // 1 = pre-passes on an auxiliary stream of the default priority; 2 = pre-passes at the default priority AND the warps on
    // the highest-priority auxiliary stream (the warp keeps its workgroup slots, the pre-pass only fills what is left)
    const int mode = dev_switch(CS_DEBUG_CHUNKS) / 100;
    // The auxiliary streams and their fork / ready events are one set per device: two host threads interleaving their
    // hipEventRecord / hipStreamWaitEvent pairs on them would wait for each other's records.  The chunked schedule (a development
    // switch) therefore enqueues under one lock, from the fork to the join -- host-side enqueueing only, the GPU work itself
    // still overlaps; the shared events also make this path unfit for stream capture (the default single-chunk path is).
    static std::mutex chunk_mu;
    std::lock_guard<std::mutex> chunk_lock(chunk_mu);
    AuxStream* const X = aux_stream(mode >= 1);
    AuxStream* const Wp = mode == 2 ? aux_stream(0) : nullptr;
    if (!X || (mode == 2 && !Wp)) return fail_hip(hipGetLastError(), "auxiliary stream");
    hipStream_t warp_stream = Wp ? Wp->s : stream;
    // fork: the pre-passes start after everything the caller has enqueued so far (inputs, the previous call's use of the
    // workspace); join: the last chunk's warp waits for the last pre-pass, and the auxiliary stream holds nothing else
    hipError_t e = hipEventRecord(X->fork, stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(X->s, X->fork, 0);
    if (e == hipSuccess && Wp) e = hipStreamWaitEvent(Wp->s, X->fork, 0);
    if (e != hipSuccess) return fail_hip(e, "chunk fork");
    cs_params q = *p;
    q.n = C.cf;
    const size_t chunk_ws = ws_layout(&q).total;
    const size_t in_px = (size_t)p->h * p->w, d_px = (size_t)p->depth_h * p->depth_w * p->depth_c;
    const size_t st_px = (size_t)out_h * out_w * 3, mk_px = (size_t)mask_h * mask_w;
    const size_t st_bytes = (p->flags & 2) ? 1 : 4;   // uint8 stereoscope codes (flags bit 1) or float32
    rc = CS_OK;
    for (int c = 0; c < C.nch && rc == CS_OK; c++) {
        const int f0 = c * C.cf;
        q.n = (f0 + C.cf <= p->n) ? C.cf : p->n - f0;
        rc = generate_chunk(&q, image + f0 * in_px * 3, depth + f0 * d_px, (float*)((char*)stereo + f0 * st_px * st_bytes),
                            depth_l + f0 * in_px * 3, depth_r + f0 * in_px * 3, mask + f0 * mk_px, stats + (size_t)f0 * ST_WORDS,
                            scratch + (size_t)c * chunk_ws, out_h, out_w, X->s, warp_stream, X->ready[c % AUX_EVENTS]);
    }
    // join: whatever was enqueued on the auxiliary streams precedes the caller's next work (also after a failure)
    if (Wp) {
        if (hipEventRecord(Wp->fork, Wp->s) == hipSuccess) (void)hipStreamWaitEvent(stream, Wp->fork, 0);
    }
    if (rc != CS_OK || Wp) {
        if (hipEventRecord(X->ready[0], X->s) == hipSuccess) (void)hipStreamWaitEvent(stream, X->ready[0], 0);
    }
    return rc;
}

// [statistics][flagged-row block][technique scratch: the splat result of hybrid_edge / the replay pool of polylines]
static size_t asd_tech_bytes(int n, int h, int w, int fill) {
    if (fill == CS_FILL_HYBRID_EDGE || fill == CS_FILL_HYBRID_EDGE_PLUS) return al256(hybrid_workspace_bytes(n, h, w));
    if (fill == CS_FILL_POLYLINES_SOFT || fill == CS_FILL_POLYLINES_SHARP) return al256(poly_replay_bytes(n, h, w, fill == CS_FILL_POLYLINES_SHARP));
    return 0;
}
size_t cs_asd_workspace_bytes_for(int n, int h, int w, int fill) {
    return al256((size_t)n * ST_WORDS * 4) + rowflag_bytes((size_t)n * h) + asd_tech_bytes(n, h, w, fill);
}
size_t cs_asd_workspace_bytes(int n, int h, int w) {   // enough for any technique
    const size_t a = cs_asd_workspace_bytes_for(n, h, w, CS_FILL_HYBRID_EDGE), b = cs_asd_workspace_bytes_for(n, h, w, CS_FILL_POLYLINES_SHARP);
    return a > b ? a : b;
}

int cs_apply_stereo_divergence(const uint8_t* image_u8, const float* depth, int n, int h, int w, double divergence,
                               double separation, double exponent, int fill, double convergence, uint8_t* out_u8,
                               void* workspace, size_t workspace_bytes, void* stream_) {
    return cs_apply_stereo_divergence2(image_u8, depth, n, h, w, divergence, separation, exponent, fill, convergence, 0, out_u8,
                                       workspace, workspace_bytes, stream_);
}

int cs_apply_stereo_divergence2(const uint8_t* image_u8, const float* depth, int n, int h, int w, double divergence,
                                double separation, double exponent, int fill, double convergence, int dialect,
                                uint8_t* out_u8, void* workspace, size_t workspace_bytes, void* stream_) {
    if (dialect < 0 || dialect > 3) return fail(CS_EINVAL, "dialect: 0 (D32), 1 (float64 disparities), 2 (int64 sums), 3 (D64)");
    if (dialect && !dialect_d64_ok(fill))
        return fail(CS_EINVAL, DIALECT_MSG);
    hipStream_t stream = (hipStream_t)stream_;
    if (!image_u8 || !depth || !out_u8 || !workspace) return fail(CS_EINVAL, "null pointer");
    if (n <= 0 || h <= 0 || w <= 0) return fail(CS_EINVAL, "non-positive size");
    if (fill < 0 || fill > CS_FILL_HYBRID_EDGE_PLUS || fill == CS_FILL_GPU_WARP) return fail(CS_EINVAL, "unknown fill technique");
    if (w > max_width_for(fill, 0) || !dialect_width_ok(fill, w, 0, dialect)) return fail(CS_ELIMIT, "frame too wide for the LDS-resident row kernel");
    if (workspace_bytes < cs_asd_workspace_bytes_for(n, h, w, fill)) return fail(CS_EWORKSPACE, "workspace too small");
    uint32_t* stats = (uint32_t*)workspace;
    hipLaunchKernelGGL(k_stats_init, dim3((n * ST_WORDS + 255) / 256), dim3(256), 0, stream, stats, n);
    hipLaunchKernelGGL(k_minmax, dim3(grid_for((size_t)h * w, 256), n), dim3(256), 0, stream, depth, h * w, stats, ST_L_MIN, ST_L_MAX);
    RowArgs A;
    memset(&A, 0, sizeof(A));
    A.n = n; A.h = h; A.w = w;
    A.image_u8 = image_u8;
    A.stats = stats; A.stats_rw = stats;
    A.scale_from_stats = 0;
    A.e32 = (float)exponent;
    A.conv32 = (float)convergence;
    A.e64 = exponent;
    A.d64 = dialect;
    A.neyes = 1;
    eye_setup(A.eye[0], divergence, separation, w);
    A.eye[0].depth = depth; A.eye[0].st_min = ST_L_MIN; A.eye[0].st_max = ST_L_MAX;
    A.out_u8 = out_u8;
    A.single = -1;
    A.dbg = dev_switch(CS_DEBUG_DBG);
    if (fill == CS_FILL_HYBRID_EDGE || fill == CS_FILL_HYBRID_EDGE_PLUS) {
        int rc = launch_hybrid(A, (char*)workspace + al256((size_t)n * ST_WORDS * 4) + rowflag_bytes((size_t)n * h), stream,
                               fill == CS_FILL_HYBRID_EDGE_PLUS);
        if (rc) return fail(rc, "hybrid_edge launch failed");
    } else {
        int halo = poly_halo(divergence, divergence, separation, exponent, convergence, w);
        const bool poly = fill == CS_FILL_POLYLINES_SOFT || fill == CS_FILL_POLYLINES_SHARP;   // (the others have no replay scratch)
        int rc = run_rows(fill, A, halo, (uint8_t*)workspace + al256((size_t)n * ST_WORDS * 4), stream, nullptr,
                          poly ? (char*)workspace + al256((size_t)n * ST_WORDS * 4) + rowflag_bytes((size_t)n * h) : nullptr);
        if (rc) return rc;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : fail_hip(e, "cs_apply_stereo_divergence");
}

size_t cs_blur_workspace_bytes(int n, int h, int w) { return al256((size_t)n * ST_WORDS * 4) + 2 * al256((size_t)n * h * w * 4); }

int cs_directional_blur(const float* depth, int n, int h, int w, double blur_strength, double edge_threshold,
                        double blur_mask_width, double falloff_exponent, int vert_smooth_px, float* out_l, float* out_r, void* workspace,
                        size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!depth || !out_l || !out_r || !workspace) return fail(CS_EINVAL, "null pointer");
    if (n <= 0 || h <= 0 || w <= 0) return fail(CS_EINVAL, "non-positive size");
    if (workspace_bytes < cs_blur_workspace_bytes(n, h, w)) return fail(CS_EWORKSPACE, "workspace too small");
    size_t hw = (size_t)h * w;
    if (blur_strength <= 0) {  // reference :1194 returns the input twice
        hipError_t e1 = hipMemcpyAsync(out_l, depth, n * hw * 4, hipMemcpyDeviceToDevice, stream);
        hipError_t e2 = hipMemcpyAsync(out_r, depth, n * hw * 4, hipMemcpyDeviceToDevice, stream);
        return (e1 == hipSuccess && e2 == hipSuccess) ? CS_OK : fail_hip(e1 != hipSuccess ? e1 : e2, "cs_directional_blur");
    }
    char* ws = (char*)workspace;
    uint32_t* stats = (uint32_t*)ws;
    float* wl = (float*)(ws + al256((size_t)n * ST_WORDS * 4));
    float* wr = wl + al256(n * hw * 4) / 4;
    hipLaunchKernelGGL(k_stats_init, dim3((n * ST_WORDS + 255) / 256), dim3(256), 0, stream, stats, n);
    int rc = launch_blur(depth, n, h, w, blur_strength, edge_threshold, blur_mask_width, falloff_exponent, vert_smooth_px, out_l, out_r, wl,
                         wr, stats, 0, stream);
    if (rc) return fail_blur(rc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : fail_hip(e, "cs_directional_blur");
}

size_t cs_blur_scipy_workspace_bytes(int n, int h, int w) { return scipyblur_workspace_bytes(n, h, w); }

int cs_directional_blur_scipy(const float* depth, int n, int h, int w, double blur_strength, double edge_threshold,
                              double blur_mask_width, double falloff_exponent, int vert_smooth_px, float* out_l, float* out_r,
                              void* workspace, size_t workspace_bytes, void* stream_) {
    if (!depth || !out_l || !out_r || !workspace) return fail(CS_EINVAL, "null pointer");
    if (n <= 0 || h <= 0 || w <= 0) return fail(CS_EINVAL, "non-positive size");
    if (workspace_bytes < cs_blur_scipy_workspace_bytes(n, h, w)) return fail(CS_EWORKSPACE, "workspace too small");
    if (!(blur_strength > 0.0)) return fail(CS_EINVAL, "blur_strength <= 0: the reference returns the depth map itself (:1374); the caller does the same");
    if (!(blur_strength < 1e6) || !(blur_mask_width < 1e6)) return fail(CS_EINVAL, "blur_strength / blur_mask_width out of range");
    int rc = launch_scipyblur(depth, n, h, w, blur_strength, edge_threshold, blur_mask_width, falloff_exponent, vert_smooth_px,
                              out_l, out_r, workspace, (hipStream_t)stream_);
    if (rc == CS_EINVAL) return fail(rc, "blur_strength rounds to a box of 0 taps: the reference fails there too (scipy: no filter weights given)");
    return rc ? fail(rc, "cs_directional_blur_scipy launch failed") : CS_OK;
}

size_t cs_warp_workspace_bytes(int n, int h, int w) { return al256((size_t)n * ST_WORDS * 4) + al256(gpuwarp_workspace_bytes(n, h, w, n, 0)); }
size_t cs_warp_mesh_workspace_bytes(int n, int h, int w) { return al256((size_t)n * ST_WORDS * 4) + al256(gpuwarp_workspace_bytes(n, h, w, n, 1)); }

static int forward_warp_common(const float* image, const float* depth, int n, int h, int w, double divergence_px,
                               double separation_px, double exponent, double convergence, int mesh, double grad_thr,
                               float* warped, uint8_t* gap_mask, void* workspace, size_t workspace_bytes, hipStream_t stream,
                               int max_stretch = 8) {
    if (!image || !depth || !warped || !gap_mask || !workspace) return fail(CS_EINVAL, "null pointer");
    if (n <= 0 || h <= 0 || w <= 0) return fail(CS_EINVAL, "non-positive size");
    if (mesh && (h < 2 || w < 2)) return fail(CS_EINVAL, "the mesh warp needs at least 2 x 2 pixels");
    if (mesh && !(grad_thr >= 0.0)) return fail(CS_EINVAL, "gradient_threshold must be >= 0");
    if (w > (mesh ? meshwarp_max_width() : gpuwarp_max_width())) return fail(CS_ELIMIT, "frame too wide for the LDS-resident row kernel");
    if (workspace_bytes < (mesh ? cs_warp_mesh_workspace_bytes(n, h, w) : cs_warp_workspace_bytes(n, h, w)))
        return fail(CS_EWORKSPACE, "workspace too small");
    uint32_t* stats = (uint32_t*)workspace;
    hipLaunchKernelGGL(k_stats_init, dim3((n * ST_WORDS + 255) / 256), dim3(256), 0, stream, stats, n);
    hipLaunchKernelGGL(k_minmax, dim3(grid_for((size_t)h * w, 256), n), dim3(256), 0, stream, depth, h * w, stats, ST_L_MIN, ST_L_MAX);
    int rc = launch_gpuwarp_plain(image, depth, n, h, w, divergence_px, separation_px, exponent, convergence, warped,
                                  gap_mask, stats, (char*)workspace + al256((size_t)n * ST_WORDS * 4), stream, mesh, grad_thr, max_stretch);
    if (rc == CS_ELIMIT) return fail(rc, "gpu_warp: gradient_threshold above 13 with max_stretch above 16 is not supported (more than 16 effective scatter rounds)");
    if (rc) return fail(rc, "gpu_warp launch failed");
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : fail_hip(e, "cs_forward_warp");
}

int cs_forward_warp(const float* image, const float* depth, int n, int h, int w, double divergence_px,
                    double separation_px, double exponent, double convergence, float* warped, uint8_t* gap_mask,
                    void* workspace, size_t workspace_bytes, void* stream_) {
    return forward_warp_common(image, depth, n, h, w, divergence_px, separation_px, exponent, convergence, 0, 1.5, warped,
                               gap_mask, workspace, workspace_bytes, (hipStream_t)stream_);
}

int cs_forward_warp2(const float* image, const float* depth, int n, int h, int w, double divergence_px,
                     double separation_px, double exponent, double convergence, double gradient_threshold, int max_stretch,
                     float* warped, uint8_t* gap_mask, void* workspace, size_t workspace_bytes, void* stream_) {
    return forward_warp_common(image, depth, n, h, w, divergence_px, separation_px, exponent, convergence, 0, gradient_threshold,
                               warped, gap_mask, workspace, workspace_bytes, (hipStream_t)stream_, max_stretch);
}

int cs_forward_warp_mesh(const float* image, const float* depth, int n, int h, int w, double divergence_px,
                         double separation_px, double exponent, double convergence, double gradient_threshold,
                         float* warped, uint8_t* gap_mask, void* workspace, size_t workspace_bytes, void* stream_) {
    return forward_warp_common(image, depth, n, h, w, divergence_px, separation_px, exponent, convergence, 1,
                               gradient_threshold, warped, gap_mask, workspace, workspace_bytes, (hipStream_t)stream_);
}

int cs_expand_u8(const uint8_t* codes, float* out, size_t count, void* stream) {
    if (!codes || !out) return fail(CS_EINVAL, "null pointer");
    if (((uintptr_t)codes & 3) || ((uintptr_t)out & 15)) return fail(CS_EINVAL, "cs_expand_u8 needs 4-/16-byte aligned buffers");
    hipLaunchKernelGGL(k_expand_u8, dim3(grid_for(count / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, codes, out, count);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : fail_hip(e, "cs_expand_u8");
}


int cs_pack_u8(const float* values, uint8_t* codes, size_t count, int stride, int mode, void* stream) {
    if (!values || !codes) return fail(CS_EINVAL, "null pointer");
    if (stride < 1 || mode < 0 || mode > 1) return fail(CS_EINVAL, "cs_pack_u8: stride >= 1, mode 0 (k / 255 values) or 1 (mask flags)");
    if ((uintptr_t)codes & 3) return fail(CS_EINVAL, "cs_pack_u8 needs a 4-byte aligned code buffer");
    if (count == 0) return CS_OK;
    hipLaunchKernelGGL(k_pack_u8, dim3(grid_for(count / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, values, codes, count, stride, mode);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : fail_hip(e, "cs_pack_u8");
}

int cs_take_f32(const float* values, float* out, size_t count, int stride, void* stream) {
    if (!values || !out) return fail(CS_EINVAL, "null pointer");
    if (stride < 1) return fail(CS_EINVAL, "cs_take_f32: stride >= 1");
    if (count == 0) return CS_OK;
    hipLaunchKernelGGL(k_take_f32, dim3(grid_for(count, 256)), dim3(256), 0, (hipStream_t)stream, values, out, count, stride);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : fail_hip(e, "cs_take_f32");
}

size_t cs_stereo_shift_workspace_bytes(void) { return al256(ST_WORDS * 4); }

int cs_stereo_shift(const float* input, const float* depth, int b, int c, int h, int w, double scale_factor, int shift_both,
                    double stereo_offset_exponent, float* out, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!input || !depth || !out || !workspace) return fail(CS_EINVAL, "null pointer");
    if (b <= 0 || c <= 0 || h <= 0 || w <= 0) return fail(CS_EINVAL, "non-positive size");
    if ((size_t)w * 4 > 64 * 1024) return fail(CS_ELIMIT, "row too wide for the LDS-resident winner table");
    if (workspace_bytes < cs_stereo_shift_workspace_bytes()) return fail(CS_EWORKSPACE, "workspace too small");
    uint32_t* stats = (uint32_t*)workspace;
    hipLaunchKernelGGL(k_stats_init, dim3(1), dim3(256), 0, stream, stats, 1);
    // one "frame" = the whole depth tensor: the reference normalises with its global min / max (stereo_utils.py:36-45)
    const size_t total = (size_t)b * h * w;
    if (total >= (1ull << 31)) return fail(CS_ELIMIT, "depth tensor too large");
    hipLaunchKernelGGL(k_minmax, dim3(grid_for(total, 256), 1), dim3(256), 0, stream, depth, (int)total, stats, ST_L_MIN, ST_L_MAX);
    const double e = stereo_offset_exponent;
    const int pow_mode = e == 1.0 ? 1 : (e == 2.0 ? 2 : (e == 0.5 ? 3 : 0));
    const size_t half = (size_t)b * c * h * w;
    const double balance = shift_both ? 0.5 : 0.0;
    for (int eye = 0; eye < 2; eye++) {
        float* dst = out + eye * half;
        if (eye == 0 && !shift_both) {   // left = input (:75-77)
            hipError_t e1 = hipMemcpyAsync(dst, input, half * 4, hipMemcpyDeviceToDevice, stream);
            if (e1 != hipSuccess) return fail_hip(e1, "cs_stereo_shift");
            continue;
        }
        const double sf = eye == 0 ? +1 * scale_factor * balance : -1 * scale_factor * (1 - balance);
        const double scale_px = (sf / 100.0) * (double)w;   // (:54)
        hipLaunchKernelGGL(k_stereo_shift, dim3(h, b), dim3(256), (size_t)w * 4, stream, input, depth, c, h, w, stats,
                           (float)scale_px, scale_px < 0 ? 1 : 0, pow_mode, (float)e, dst);
    }
    hipError_t er = hipGetLastError();
    return er == hipSuccess ? CS_OK : fail_hip(er, "cs_stereo_shift");
}

int cs_profile(int enable) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof_on.store(enable != 0);
    if (enable) {   // a new measurement: events and tile counts start from zero (switching off keeps them readable)
        g_prof_used = 0;
        g_prof_tiles_any = false; g_prof_tiles_stream = nullptr;
        if (!g_prof_tiles && hipMalloc(&g_prof_tiles, 16) != hipSuccess) { g_prof_tiles = nullptr; (void)hipGetLastError(); }
        if (g_prof_tiles && hipMemset(g_prof_tiles, 0, 16) != hipSuccess) return fail_hip(hipGetLastError(), "cs_profile");
    }
    return CS_OK;
}

int cs_debug_set(int key, int value) {
    if (key < 0 || key >= CS_DEBUG_KEYS) return fail(CS_EINVAL, "cs_debug_set: unknown key");
#ifndef CS_DEV
    if (key == CS_DEBUG_DBG && value != 0 && value != 14 && value != 17 && value != 30 && value != 31)   // (30 / 31: column ranges of the polylines row kernel forced / off: same pixels)
        return fail(CS_EINVAL, "cs_debug_set: this CS_DEBUG_DBG value needs a -DCS_DEV build (phase cut-offs leave outputs unwritten)");
#endif
    g_dev[key].store(value, std::memory_order_relaxed);
    return CS_OK;
}

int cs_profile_read(double* total_ms, int* launches) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    double tot = 0.0;
    for (int i = 0; i < g_prof_used; i++) {
        hipError_t e = hipEventSynchronize(g_prof_ev[2 * i + 1]);
        if (e != hipSuccess) return fail_hip(e, "cs_profile_read");
        float ms = 0.0f;
        e = hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]);
        if (e != hipSuccess) return fail_hip(e, "cs_profile_read");
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = g_prof_used;
    g_prof_used = 0;
    return CS_OK;
}

// Which share of the 64 x 32 tiles of the cs_generate calls profiled since cs_profile(1) had been written to the blurred depth maps
// (the others are read from the shared gray depth by both eyes: 4 instead of 8 bytes per pixel of depth input for the warp
// kernel), over all their chunks.  Blocking (waits for the last such call's stream, copies two words from a buffer the library
// owns).  *fraction = -1: none of those calls used lazy tiles.
int cs_profile_tiles(double* fraction) {
    if (!fraction) return fail(CS_EINVAL, "null pointer");
    std::lock_guard<std::mutex> lock(g_prof_mu);
    *fraction = -1.0;
    if (!g_prof_tiles || !g_prof_tiles_any) return CS_OK;
    unsigned long long host[2] = {0, 0};
    hipError_t e = hipStreamSynchronize(g_prof_tiles_stream);
    if (e == hipSuccess) e = hipMemcpy(host, g_prof_tiles, 16, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail_hip(e, "cs_profile_tiles");
    if (host[1]) *fraction = (double)host[0] / (double)host[1];
    return CS_OK;
}

int cs_test_powf(const float* x, float y, float* out, size_t count, void* stream) {
    if (!x || !out) return fail(CS_EINVAL, "null pointer");
    hipLaunchKernelGGL(k_test_powf, dim3(grid_for(count, 256)), dim3(256), 0, (hipStream_t)stream, x, y, out, count);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : fail_hip(e, "cs_test_powf");
}

float cs_test_edge_threshold(float den) { return blur_edge_threshold_host(den); }

int cs_test_exp(const double* x, double* out, size_t count, void* stream) {
    if (!x || !out) return fail(CS_EINVAL, "null pointer");
    hipLaunchKernelGGL(k_test_exp, dim3(grid_for(count, 256)), dim3(256), 0, (hipStream_t)stream, x, out, count);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CS_OK : fail_hip(e, "cs_test_exp");
}

}  // extern "C"
