// cs_blur.hip -- direction-aware depth blur (reference stereoimage_generation.py:1171-1251,
// `directional_motion_blur_gpu` + `_edge_distance_weight_gpu` :1131-1168), two row kernels.
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
        uint32_t* st = A.stats_rw + frame * ST_WORDS;
        lmin = wave_min(lmin); lmax = wave_max(lmax); rmin = wave_min(rmin); rmax = wave_max(rmax);
        if (lane_id() == 0) {
            atomicMin(&st[ST_L_MIN], csm::f2ord(lmin)); atomicMax(&st[ST_L_MAX], csm::f2ord(lmax));
            atomicMin(&st[ST_R_MIN], csm::f2ord(rmin)); atomicMax(&st[ST_R_MAX], csm::f2ord(rmax));
        }
    }
}

int launch_blur(const float* depth, int n, int h, int w, double strength, double edge_threshold, double falloff,
                int vert, float* out_l, float* out_r, float* wl, float* wr, uint32_t* stats, int node_path,
                hipStream_t stream) {
    BlurArgs A;
    A.depth = depth; A.n = n; A.h = h; A.w = w;
    A.stats = node_path ? stats : nullptr;
    A.stats_rw = node_path ? stats : nullptr;
    A.den = (float)(10.0 * edge_threshold);
    A.bs = (int)nearbyint(strength);  // Python round(): half to even
    A.radius = (int)strength;
    A.vert = vert > 0 ? vert : 0;
    if (A.bs < 1) return CS_EINVAL;  // torch raises on a zero-width kernel
    A.fall32 = (float)falloff;
    A.fall_mode = falloff == 1.0 ? 0 : falloff == 0.5 ? 1 : falloff == 2.0 ? 2 : falloff == 3.0 ? 3 : falloff == 0.0 ? 5 : 4;
    A.wl = wl; A.wr = wr; A.out_l = out_l; A.out_r = out_r;
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
