// store_patterns.hip -- the memory skeleton of k_polypoint as a standalone micro-benchmark (round 5):
// the tile geometry, grid order (XCD-aware x, eye groups in y) and byte counts of the metric kernel -- 64 x 4K frames, tiles of
// 768 pixels, 256 threads -- with NO arithmetic, in several load / store forms:
//   cur   : lane = source slot j = tid + 256 k (today's kernel): 12-byte image load + 4-byte depth load per slot (halo included:
//           T + 2 S + 2 columns staged), per tile pixel one global_store_dwordx3 (stereoscope), one dword (mask), one dwordx3
//           (depth map) -- 12-byte stride between lanes
//   x4    : lane = 16 output bytes: the same bytes of the same tile moved with dwordx4 loads / stores (1 KB contiguous per wave
//           instruction), what a pixel-owned store phase out of LDS would issue
//   x4h   : x4 stores, `cur` loads (the loads must stay lane-per-source in any restructuring)
// each as plain and as nontemporal stores, and write-only / read-only.  Output values are a cheap function of what was loaded.
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip && ./store_patterns [frames]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct F3 { float x, y, z; };
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f3v __attribute__((ext_vector_type(3)));
constexpr int W = 3840, H = 2160, T = 768, S = 79, NT = 256, SLOTS = 4, EG = 4;

__host__ __device__ inline int eye_group_grid_y(int h) { const int G = 1 << EG, ny = (h + 7) / 8; return 2 * ((ny + G - 1) / G) * G; }
__device__ inline void eye_group_decode(int yi, int& yrow, int& eye) {
    yrow = ((yi >> (EG + 1)) << EG) | (yi & ((1 << EG) - 1));
    eye = (yi >> EG) & 1;
}

template <bool NTS> __device__ __forceinline__ void st(float* p, float v) {
    if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NTS> __device__ __forceinline__ void st(float4* p, float4 v) {
    if (NTS) __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(p)); else *p = v;
}
template <bool NTS> __device__ __forceinline__ void st3(char* p, float a, float b, float c) {
    if (NTS) {   // (no 12-byte nontemporal builtin: three dword nt stores would change the instruction count -- use asm)
        const f3v v{a, b, c};
        asm volatile("global_store_dwordx3 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    } else *reinterpret_cast<F3*>(p) = F3{a, b, c};
}

// LOADS: 0 none, 1 lane-per-source (12 B + 4 B, halo), 2 dwordx4 of the tile's bytes.  STORES: 0 none, 1 lane-per-pixel dwordx3 /
// dword / dwordx3, 2 dwordx4 contiguous.  EYE_ORDER: 1 eye groups (production), 0 eye-major
template <int LOADS, int STORES, bool NTS>
__global__ void __launch_bounds__(NT, 7)
k_skel(const float* __restrict__ image, const float* __restrict__ depth0, const float* __restrict__ depth1, float* stereo, float* mask,
       float* dl, float* dr, float* sink) {
    __shared__ float lds[T * 4 + 8];
    const int tid = threadIdx.x;
    const int xi = blockIdx.x;
    int yrow, eye;
    eye_group_decode((int)blockIdx.y, yrow, eye);
    const int row = yrow * 8 + (xi & 7);
    if (row >= H) return;
    const int tile = xi >> 3, frame = blockIdx.z;
    const int o0 = tile * T, wt = min(T, W - o0);
    const int s0 = max(0, o0 - S - 1), s1 = min(W, o0 + wt + S + 1), ns = s1 - s0;
    const uint32_t rowpix = ((uint32_t)frame * H + row) * W;
    const float* depth = eye ? depth1 : depth0;
    float acc[SLOTS * 4];
#pragma unroll
    for (int k = 0; k < SLOTS * 4; k++) acc[k] = (float)(tid + k);
    if (LOADS == 1) {
        const char* irow = reinterpret_cast<const char*>(reinterpret_cast<const F3*>(image) + rowpix + s0);
        const char* drow = reinterpret_cast<const char*>(depth + rowpix + s0);
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const uint32_t jc = (uint32_t)min(tid + k * NT, ns - 1);
            const F3 c = *reinterpret_cast<const F3*>(irow + 12u * jc);
            const float d = *reinterpret_cast<const float*>(drow + 4u * jc);
            acc[4 * k] = c.x; acc[4 * k + 1] = c.y; acc[4 * k + 2] = c.z; acc[4 * k + 3] = d;
        }
        // the point records of the real kernel: one 8-byte LDS store per slot, read back by the neighbours after a barrier
#pragma unroll
        for (int k = 0; k < SLOTS; k++) { lds[2 * (tid + k * NT)] = acc[4 * k]; lds[2 * (tid + k * NT) + 1] = acc[4 * k + 3]; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SLOTS; k++) acc[4 * k + 1] += lds[2 * ((tid + k * NT + 1) & 1023)];
    } else if (LOADS == 2) {
        const float4* irow = reinterpret_cast<const float4*>(reinterpret_cast<const F3*>(image) + rowpix + o0);
        const float4* drow = reinterpret_cast<const float4*>(depth + rowpix + o0);
        const int n16i = wt * 3 / 4, n16d = wt / 4;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = tid + k * NT;
            if (i < n16i) { const float4 v = irow[i]; acc[4 * k] = v.x; acc[4 * k + 1] = v.y; acc[4 * k + 2] = v.z; acc[4 * k + 3] = v.w; }
        }
        if (tid < n16d) { const float4 v = drow[tid]; acc[12] = v.x; acc[13] = v.y; acc[14] = v.z; acc[15] = v.w; }
    }
    const int xoff = eye ? W : 0;
    const uint32_t obase = ((uint32_t)frame * H + row) * (2 * W) + xoff + o0;
    float* const dd = eye ? dr : dl;
    if (STORES == 1) {
        char* st_row = (char*)stereo + (size_t)obase * 12;
        char* mk_row = (char*)mask + (size_t)obase * 4;
        char* dd_row = (char*)dd + (size_t)(rowpix + o0) * 12;
        const int qoff = s0 - o0;
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
            const int q = tid + k * NT + (LOADS == 1 ? qoff : 0);
            if ((unsigned)q < (unsigned)wt) {
                st3<NTS>(dd_row + 12u * q, acc[4 * k + 3], acc[4 * k + 3], acc[4 * k + 3]);
                st3<NTS>(st_row + 12u * q, acc[4 * k] * 0.5f, acc[4 * k + 1] * 0.5f, acc[4 * k + 2] * 0.5f);
                st<NTS>(reinterpret_cast<float*>(mk_row + 4u * q), acc[4 * k] == 0.0f ? 1.0f : 0.0f);
            }
        }
    } else if (STORES == 2) {
        float4* st_row = reinterpret_cast<float4*>((char*)stereo + (size_t)obase * 12);
        float4* mk_row = reinterpret_cast<float4*>((char*)mask + (size_t)obase * 4);
        float4* dd_row = reinterpret_cast<float4*>((char*)dd + (size_t)(rowpix + o0) * 12);
        const int n16 = wt * 3 / 4, n16m = wt / 4;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = tid + k * NT;
            if (i < n16) {
                st<NTS>(dd_row + i, float4{acc[4 * k + 3], acc[(4 * k + 7) & 15], acc[(4 * k + 11) & 15], acc[4 * k + 3]});
                st<NTS>(st_row + i, float4{acc[4 * k] * 0.5f, acc[4 * k + 1] * 0.5f, acc[4 * k + 2] * 0.5f, acc[4 * k + 3] * 0.5f});
            }
        }
        if (tid < n16m) st<NTS>(mk_row + tid, float4{acc[0] == 0.0f ? 1.0f : 0.0f, acc[4] == 0.0f ? 1.0f : 0.0f, acc[8] == 0.0f ? 1.0f : 0.0f, acc[12] == 0.0f ? 1.0f : 0.0f});
    } else {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < SLOTS * 4; k++) s += acc[k];
        if (s == 123.456f) sink[0] = s;
    }
}

template <int LOADS, int STORES, bool NTS>
static int run(const char* name, int frames, const float* image, const float* d0, const float* d1, float* stereo, float* mask, float* dl,
               float* dr, float* sink) {
    const int tiles = (W + T - 1) / T;
    dim3 grid(tiles * 8, eye_group_grid_y(H), frames), block(NT);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_skel<LOADS, STORES, NTS>), grid, block, 0, 0, image, d0, d1, stereo, mask, dl, dr, sink);
    CK(hipDeviceSynchronize());
    const int iters = 8;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k_skel<LOADS, STORES, NTS>), grid, block, 0, 0, image, d0, d1, stereo, mask, dl, dr, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double px = (double)frames * W * H;
    // bytes that must cross the HBM interface once: image 12 + depth 2 x 4 (each eye its own map here) read, 56 written
    const double rd = LOADS ? px * (12.0 + 8.0) : 0.0, wr = STORES ? px * 56.0 : 0.0;
    printf("%-34s %8.3f ms per %d frames = %7.3f ms per 64 | %6.0f GB/s (read %.1f + written %.1f GB)\n", name, ms, frames,
           ms * 64.0 / frames, (rd + wr) / ms / 1e6, rd / 1e9, wr / 1e9);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 16;
    const size_t px = (size_t)frames * W * H;
    float *image, *d0, *d1, *stereo, *mask, *dl, *dr, *sink;
    CK(hipMalloc(&image, px * 12)); CK(hipMalloc(&d0, px * 4)); CK(hipMalloc(&d1, px * 4));
    CK(hipMalloc(&stereo, px * 24)); CK(hipMalloc(&mask, px * 8)); CK(hipMalloc(&dl, px * 12)); CK(hipMalloc(&dr, px * 12));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(image, 0x3c, px * 12)); CK(hipMemset(d0, 0x3c, px * 4)); CK(hipMemset(d1, 0x3c, px * 4));
    printf("store_patterns: %d frames of %d x %d, tiles of %d (+ halo %d), %d threads x %d slots, eye groups of %d rows\n", frames, W, H, T, S, NT,
           SLOTS, 8 << EG);
#define RUN(L, St, N) if (run<L, St, N>(#L " loads / " #St " stores / nt=" #N, frames, image, d0, d1, stereo, mask, dl, dr, sink)) return 1;
    RUN(1, 1, false)   // today's skeleton
    RUN(1, 1, true)
    RUN(1, 2, false)   // lane-per-source loads, full-line stores
    RUN(1, 2, true)
    RUN(2, 2, false)   // pure dwordx4 copy of the same bytes
    RUN(2, 2, true)
    RUN(0, 1, false)   // write-only
    RUN(0, 1, true)
    RUN(0, 2, false)
    RUN(0, 2, true)
    RUN(1, 0, false)   // read-only
    RUN(2, 0, false)
    RUN(1, 1, false)   // (again: drift check)
    return 0;
}
