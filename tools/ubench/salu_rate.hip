// salu_rate.hip -- is the scalar ALU a per-SIMD or a per-CU resource on gfx950, and do SALU instructions co-issue with VALU?
// (k_polypoint executes about as many SALU as VALU instructions per wave.)  hipcc --offload-arch=gfx950 -O2 -o salu_rate salu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define S8 "s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n s_add_u32 s44, s44, 1\n s_add_u32 s45, s45, 1\n s_add_u32 s46, s46, 1\n s_add_u32 s47, s47, 1\n"
#define A8 "s_and_b64 s[40:41], s[40:41], s[42:43]\n s_or_b64 s[44:45], s[44:45], s[46:47]\n s_and_b64 s[42:43], s[40:41], s[46:47]\n s_or_b64 s[46:47], s[44:45], s[40:41]\n s_and_b64 s[40:41], s[40:41], s[42:43]\n s_or_b64 s[44:45], s[44:45], s[46:47]\n s_and_b64 s[42:43], s[40:41], s[46:47]\n s_or_b64 s[46:47], s[44:45], s[40:41]\n"
#define V8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define I8 "v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n"
// interleaved: one VALU, one SALU
#define VS8 "v_fma_f32 %0, %0, %8, %9\n s_add_u32 s40, s40, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s41, s41, 1\n v_fma_f32 %2, %2, %8, %9\n s_add_u32 s42, s42, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s43, s43, 1\n v_fma_f32 %4, %4, %8, %9\n s_add_u32 s44, s44, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 s45, s45, 1\n v_fma_f32 %6, %6, %8, %9\n s_add_u32 s46, s46, 1\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 s47, s47, 1\n"
#define IS8 "v_lshl_add_u32 %0, %0, 1, %8\n s_add_u32 s40, s40, 1\n v_lshl_add_u32 %1, %1, 1, %8\n s_add_u32 s41, s41, 1\n v_lshl_add_u32 %2, %2, 1, %8\n s_add_u32 s42, s42, 1\n v_lshl_add_u32 %3, %3, 1, %8\n s_add_u32 s43, s43, 1\n v_lshl_add_u32 %4, %4, 1, %8\n s_add_u32 s44, s44, 1\n v_lshl_add_u32 %5, %5, 1, %8\n s_add_u32 s45, s45, 1\n v_lshl_add_u32 %6, %6, 1, %8\n s_add_u32 s46, s46, 1\n v_lshl_add_u32 %7, %7, 1, %8\n s_add_u32 s47, s47, 1\n"
#define CLOB "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc"

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f + 1.0f, c = 0.25f;
    asm volatile("s_mov_b32 s40, 0\n s_mov_b32 s41, 0\n s_mov_b32 s42, 0\n s_mov_b32 s43, 0\n s_mov_b32 s44, 0\n s_mov_b32 s45, 0\n s_mov_b32 s46, 0\n s_mov_b32 s47, 0" ::: CLOB);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) asm volatile(S8 ::: CLOB);
            if (MODE == 1) asm volatile(A8 ::: CLOB);
            if (MODE == 2) asm volatile(V8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if (MODE == 3) asm volatile(VS8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : CLOB);
            if (MODE == 4) asm volatile(I8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if (MODE == 5) asm volatile(IS8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : CLOB);
        }
    }
    int sres;
    asm volatile("s_add_u32 s40, s40, s44\n v_mov_b32 %0, s40" : "=v"(sres) :: CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)sres;
}

template <int MODE>
static int run(const char* name, int per_body, int wps, float* dout) {
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, dout, 10, 1.0f); CK(hipGetLastError()); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, dout, iters, 1.0f); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s waves/SIMD=%d %8.3f ms  %6.2f cycles per instruction per SIMD (%.2f per CU)\n", name, wps, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 8 * per_body * wps), ms * 1e-3 * 2.4e9 / ((double)iters * 8 * per_body * wps * 4));
    return 0;
}

int main() {
    setvbuf(stdout, 0, _IONBF, 0);
    float* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    for (int w : {1, 2, 4, 8}) {
        run<0>("8 s_add_u32", 8, w, dout);
        run<1>("8 s_and/or_b64", 8, w, dout);
        run<2>("8 v_fma_f32", 8, w, dout);
        run<3>("8 v_fma_f32 + 8 s_add_u32 (per 16)", 16, w, dout);
        run<4>("8 v_lshl_add_u32", 8, w, dout);
        run<5>("8 v_lshl_add_u32 + 8 s_add_u32 (per 16)", 16, w, dout);
    }
    return 0;
}
