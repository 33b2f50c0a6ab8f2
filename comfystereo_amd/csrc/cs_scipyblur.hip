// cs_scipyblur.hip -- the depth blur of the numpy / PIL input path: `directional_motion_blur`, reference
// stereoimage_generation.py:1346-1419, what create_stereoimages runs when it is handed numpy arrays or PIL images with the depth
// blur on (:1489-1494).  The node never takes this path (ComfyUI passes tensors -> directional_motion_blur_gpu, cs_blur.hip); it
// exists so that the module function is a drop-in for every input form.  Three plain kernels, one lane per pixel: this is not a
// hot path, the point is the ARITHMETIC, which differs from the tensor path's:
//   * scipy.ndimage (sobel, convolve1d; a dependency outside /root/reference, restated from ni_filters.c NI_Correlate1D):
//     every correlation converts the float32 line to float64, accumulates in float64 -- separate multiply and add, in the order
//     of the symmetric / antisymmetric / general loops -- and rounds ONCE into the float32 output;
//   * borders 'reflect' for the Sobel (scipy's default: d c b a | a b c d | d c b a) and 'nearest' for the box filters;
//   * an even box of k taps covers the columns [x - k/2 + 1, x + k/2] (convolve1d reverses the kernel and moves the origin by -1);
//   * no x255 rescaling of the depth map, nearest-edge distance by two cumulative maxima (= nearest, :1393-1403).
// oracle/scipy_blur_oracle.py is the checker (pinned by outputs of the reference, tests/golden/numpy_blur.npz).
#include "cs_common.h"
#include "cs_kernels.h"

namespace cs {

__constant__ csm::PowfTables c_sb_powf_tables = CS_POWF_TABLES_INIT;

__device__ __forceinline__ int sb_reflect(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }   // |i - [0, n)| <= n
__device__ __forceinline__ int sb_nearest(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// sobel(depth, axis=1) (:1381) and the two edge masks (:1383-1386): bit 0 = left mask (grad > 0), bit 1 = right mask (grad < 0)
__global__ void __launch_bounds__(256) k_sb_masks(const float* __restrict__ depth, int h, int w, float den32, uint8_t* __restrict__ mask) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= w) return;
    const float* d = depth + (size_t)f * h * w;
    const int xl = sb_reflect(x - 1, w), xr = sb_reflect(x + 1, w);
    // correlate1d([-1, 0, 1]) along x, antisymmetric loop: o = x0 * w[1]; o += (x[-1] - x[+1]) * w[0]; rounded to float32
    auto g1 = [&](int r) {
        const float* row = d + (size_t)r * w;
        double o = (double)row[x] * 0.0;
        o += ((double)row[xl] - (double)row[xr]) * -1.0;
        return (float)o;
    };
    const float gu = g1(sb_reflect(y - 1, h)), gc = g1(y), gd = g1(sb_reflect(y + 1, h));
    // correlate1d([1, 2, 1]) along y, symmetric loop: o = x0 * 2; o += (x[-1] + x[+1]) * 1
    double o = (double)gc * 2.0;
    o += ((double)gu + (double)gd) * 1.0;
    const float grad = (float)o;
    const float edge = fabsf(grad) / den32;          // float32 array / Python float (:1383); clip(., 0, 1) > 0.5 <=> . > 0.5
    const bool strong = edge > 0.5f;
    mask[((size_t)f * h + y) * w + x] = (uint8_t)((strong && grad > 0.0f ? 1u : 0u) | (strong && grad < 0.0f ? 2u : 0u));
}

// _dist_weight (:1393-1404): clip(1 - dist / mask_radius, 0, 1) ** falloff, dist = distance to the nearest mask pixel of the row
// (anything beyond mask_radius gives 0 whatever the exact distance, so the search stops there)
__global__ void __launch_bounds__(256) k_sb_weights(const uint8_t* __restrict__ mask, int h, int w, int radius, int pow_mode,
                                                    float falloff32, float* __restrict__ wl, float* __restrict__ wr) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= w) return;
    const uint8_t* m = mask + ((size_t)f * h + y) * w;
    const float large = (float)(radius + 1);
    float dl = large, dr = large;      // left / right eye mask
    for (int k = 0; k <= radius; k++) {
        unsigned b = 0;
        if (x - k >= 0) b |= m[x - k];
        if (x + k < w) b |= m[x + k];
        if ((b & 1u) && dl == large) dl = (float)k;
        if ((b & 2u) && dr == large) dr = (float)k;
        if (dl != large && dr != large) break;
    }
    auto weight = [&](float dist) {
        float v = 1.0f - dist / (float)radius;       // (radius 0: inf / NaN like NumPy's; np.clip keeps a NaN)
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        switch (pow_mode) {                          // a float32 array ** Python float: NumPy's exact shortcuts, powf otherwise
        case 2: return v * v;
        case 1: return v;
        case 3: return sqrtf(v);
        default: return csm::powf_exact(v, falloff32, &c_sb_powf_tables);
        }
    };
    const size_t o = ((size_t)f * h + y) * w + x;
    wl[o] = weight(dl);
    wr[o] = weight(dr);
}

// vertical smoothing of the weights (:1410-1413), the k-tap box of the depth (:1416-1418), the blend (:1420-1421)
__global__ void __launch_bounds__(256) k_sb_final(const float* __restrict__ depth, const float* __restrict__ wl, const float* __restrict__ wr,
                                                  int h, int w, int vert, int k, float* __restrict__ out_l, float* __restrict__ out_r) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= w) return;
    const size_t fo = (size_t)f * h * w, o = fo + (size_t)y * w + x;
    float lw = wl[o], rw = wr[o];
    if (vert > 0) {   // convolve1d(ones(2v + 1) / (2v + 1), axis=0, 'nearest'): symmetric loop, then clip(., 0, 1)
        const double wv = 1.0 / (double)(2 * vert + 1);
        double al = (double)lw * wv, ar = (double)rw * wv;
        for (int j = -vert; j < 0; j++) {
            const size_t a = fo + (size_t)sb_nearest(y + j, h) * w + x, b = fo + (size_t)sb_nearest(y - j, h) * w + x;
            al += ((double)wl[a] + (double)wl[b]) * wv;
            ar += ((double)wr[a] + (double)wr[b]) * wv;
        }
        lw = (float)al; rw = (float)ar;
        lw = lw < 0.0f ? 0.0f : (lw > 1.0f ? 1.0f : lw);
        rw = rw < 0.0f ? 0.0f : (rw > 1.0f ? 1.0f : rw);
    }
    const float* row = depth + fo + (size_t)y * w;
    const double wk = 1.0 / (double)k;
    double acc;
    if (k & 1) {   // symmetric loop
        const int s1 = k / 2;
        acc = (double)row[x] * wk;
        for (int j = -s1; j < 0; j++) acc += ((double)row[sb_nearest(x + j, w)] + (double)row[sb_nearest(x - j, w)]) * wk;
    } else {       // general loop with origin -1: columns x - k/2 + 1 .. x + k/2, the last one first
        const int s1 = k / 2, s2 = k - s1 - 1;
        acc = (double)row[sb_nearest(x + s2 + 1, w)] * wk;
        for (int j = -s1; j < s2; j++) acc += (double)row[sb_nearest(x + j + 1, w)] * wk;
    }
    const float blurred = (float)acc, dv = row[x];
    out_l[o] = lw * blurred + (1.0f - lw) * dv;
    out_r[o] = rw * blurred + (1.0f - rw) * dv;
}

size_t scipyblur_workspace_bytes(int n, int h, int w) {
    const size_t px = (size_t)n * h * w;
    return ((px + 255) & ~(size_t)255) + 2 * ((px * 4 + 255) & ~(size_t)255);
}

int launch_scipyblur(const float* depth, int n, int h, int w, double strength, double edge_threshold, double mask_width,
                     double falloff, int vert, float* out_l, float* out_r, void* workspace, hipStream_t stream) {
    const int k = (int)nearbyint(strength);   // int(round(blur_strength)), :1377 (Python rounds halves to even, like nearbyint)
    if (k < 1) return CS_EINVAL;              // (the reference fails in scipy: "no filter weights given")
    const int radius = (int)mask_width;       // :1378
    const size_t px = (size_t)n * h * w;
    uint8_t* mask = (uint8_t*)workspace;
    float* wl = (float*)((char*)workspace + ((px + 255) & ~(size_t)255));
    float* wr = (float*)((char*)wl + ((px * 4 + 255) & ~(size_t)255));
    const dim3 grid((w + 255) / 256, h, n), block(256);
    const int pow_mode = falloff == 2.0 ? 2 : (falloff == 1.0 ? 1 : (falloff == 0.5 ? 3 : 0));
    hipLaunchKernelGGL(k_sb_masks, grid, block, 0, stream, depth, h, w, (float)(10.0 * edge_threshold), mask);
    hipLaunchKernelGGL(k_sb_weights, grid, block, 0, stream, (const uint8_t*)mask, h, w, radius < 0 ? 0 : radius, pow_mode, (float)falloff, wl, wr);
    hipLaunchKernelGGL(k_sb_final, grid, block, 0, stream, depth, (const float*)wl, (const float*)wr, h, w, vert > 0 ? vert : 0, k, out_l, out_r);
    return hipGetLastError() == hipSuccess ? CS_OK : CS_EHIP;
}

}  // namespace cs
