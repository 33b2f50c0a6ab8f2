// issue_rate.hip -- gfx950 micro-benchmarks behind the design choices of k_polytile:
// which instruction classes share an issue slot, what the non-full-rate VALU ops cost, what the f32->u8 pack
// instruction does.  Standalone: hipcc --offload-arch=gfx950 -O2 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// 8 independent accumulators; REP8(X) emits X for chains 0..7
#define V8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    float b = seed * 0.5f + 1.0f, c = 0.25f;
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
    __shared__ float lds[1024];
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1; lds[threadIdx.x + 512] = a2; lds[threadIdx.x + 768] = a3;
    __syncthreads();
    int la = (threadIdx.x * 4) & 4095;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) {  // 8 v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 1) {  // 8 v_fma_f32 + 8 s_add (interleaved)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 %11, %11, 1\n"
                             "v_fma_f32 %2, %2, %8, %9\n s_add_u32 %12, %12, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 %13, %13, 1\n"
                             "v_fma_f32 %4, %4, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 %11, %11, 1\n"
                             "v_fma_f32 %6, %6, %8, %9\n s_add_u32 %12, %12, 1\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 %13, %13, 1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c),
                               "s"(s0), "s"(s1), "s"(s2), "s"(s3));
            } else if (MODE == 2) {  // 8 s_add only
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
            } else if (MODE == 3) {  // 8 v_fma_f32 + 2 ds_read_b32
                float t0, t1;
                asm volatile("v_fma_f32 %0, %0, %10, %11\n ds_read_b32 %8, %12\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11\n"
                             "v_fma_f32 %4, %4, %10, %11\n ds_read_b32 %9, %12 offset:1024\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n"
                             "s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1)
                             : "v"(b), "v"(c), "v"(la));
            } else if (MODE == 4) {  // 8 v_fma_f64 (4 chains x2)
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)b), "v"((double)c));
            } else if (MODE == 5) {  // 8 v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 6) {  // 8 v_pk_fma_f32 (4 chains of register pairs x2)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                             "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)b));
            } else if (MODE == 7) {  // 8 DPP moves (row_shr:1) feeding adds -> 8 VALU total (dpp folded into v_add)
                asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 8) {  // 8 ds_bpermute_b32
                asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                             "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(la & 255));
            } else if (MODE == 9) {  // 8 v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                             "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)b));
            } else if (MODE == 10) {  // 4 v_cvt_f64_f32 + 4 v_cvt_f32_f64
                asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n"
                             "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            } else if (MODE == 11) {  // 8 x (v_cmp_lt_f32 -> sgpr pair) + s_and_b64: VALU compare + SALU combine
                asm volatile("v_cmp_lt_f32 s[40:41], %0, %8\n v_cmp_lt_f32 s[42:43], %1, %8\n s_and_b64 s[40:41], s[40:41], s[42:43]\n"
                             "v_cmp_lt_f32 s[44:45], %2, %8\n v_cmp_lt_f32 s[46:47], %3, %8\n s_and_b64 s[44:45], s[44:45], s[46:47]\n"
                             "v_cmp_lt_f32 s[48:49], %4, %8\n v_cmp_lt_f32 s[50:51], %5, %8\n s_and_b64 s[48:49], s[48:49], s[50:51]\n"
                             "v_cmp_lt_f32 s[52:53], %6, %8\n v_cmp_lt_f32 s[54:55], %7, %8\n s_and_b64 s[52:53], s[52:53], s[54:55]\n"
                             "v_cndmask_b32 %0, %0, %8, s[40:41]\n v_cndmask_b32 %2, %2, %8, s[44:45]\n v_cndmask_b32 %4, %4, %8, s[48:49]\n v_cndmask_b32 %6, %6, %8, s[52:53]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
                             : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
            } else if (MODE == 12) {  // 8 mixed cheap VALU: floor, med3, cvt_f32_ubyte, cvt_u32, max, min, and, lshl_or
                asm volatile("v_floor_f32 %0, %0\n v_med3_f32 %1, %1, %8, %9\n v_cvt_f32_ubyte1 %2, %2\n v_cvt_u32_f32 %3, %3\n"
                             "v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %9\n v_and_b32 %6, %6, %3\n v_lshl_or_b32 %7, %7, 8, %3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 13) {  // 8 v_fma_f32 + 8 s_nop-free s_waitcnt (no-op waits): does a wait cost an issue slot?
                asm volatile("v_fma_f32 %0, %0, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %1, %1, %8, %9\n s_waitcnt lgkmcnt(0)\n"
                             "v_fma_f32 %2, %2, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %3, %3, %8, %9\n s_waitcnt lgkmcnt(0)\n"
                             "v_fma_f32 %4, %4, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %5, %5, %8, %9\n s_waitcnt lgkmcnt(0)\n"
                             "v_fma_f32 %6, %6, %8, %9\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %7, %7, %8, %9\n s_waitcnt lgkmcnt(0)\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 14) {  // 8 ds_read_b32 only (+ wait)
                float t0, t1, t2, t3;
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n"
                             "ds_read_b32 %0, %4 offset:1024\n ds_read_b32 %1, %4 offset:1280\n ds_read_b32 %2, %4 offset:1536\n ds_read_b32 %3, %4 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                             : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(la & 1023));
                a0 += t0 + t1 + t2 + t3;
            } else if (MODE == 15) {  // 8 v_fma_f32 + 4 v_cmp (to vcc) -- compare rate
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_cmp_lt_f32 vcc, %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_cmp_lt_f32 vcc, %3, %8\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_cmp_lt_f32 vcc, %5, %8\n v_fma_f32 %6, %6, %8, %9\n v_cmp_lt_f32 vcc, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(s0 + s1 + s2 + s3);
}

__global__ void k_cvt_pk(const float* in, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = 0xAABBCCDDu;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(in[i]));
    out[i] = r;
}

// division variants on a stream of operand pairs: [0] compiler's IEEE division, [1] the same FMA core without
// div_scale / div_fixup, [2] reciprocal (rcp + one Newton step) times numerator with ONE residual correction
__global__ void k_div(const float* a, const float* b, float* q, int n, int mode) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], r;
    if (mode == 0) r = x / y;
    else {
        float y0 = __builtin_amdgcn_rcpf(y);
        float e0 = fmaf(-y, y0, 1.0f);
        float y1 = fmaf(e0, y0, y0);
        float q0 = x * y1;
        float r0 = fmaf(-y, q0, x);
        float q1 = fmaf(r0, y1, q0);
        if (mode == 1) {
            float r1 = fmaf(-y, q1, x);
            r = fmaf(r1, y1, q1);
        } else r = q1;
    }
    q[i] = r;
}

template <int MODE>
static int run_rate(const char* name, int per_body, int wgs_per_cu, float* dout, double scale_ops) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 0, 0, dout, 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 0, 0, dout, iters, 1.0f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // instructions per wave = iters * 8 * per_body; waves per SIMD = wgs_per_cu; cycles at 2.4 GHz
    double cyc = ms * 1e-3 * 2.4e9;
    double instr_per_simd = (double)iters * 8 * per_body * wgs_per_cu;
    printf("%-44s waves/SIMD=%d  %8.3f ms  cycles/instr/SIMD = %6.3f\n", name, wgs_per_cu, ms, cyc / instr_per_simd * scale_ops);
    return 0;
}

int main() {
    float* dout;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    for (int w : {1, 2, 4, 8}) {
        run_rate<0>("8 v_fma_f32", 8, w, dout, 1);
        run_rate<1>("8 v_fma_f32 + 8 s_add (per instr of 16)", 16, w, dout, 1);
        run_rate<2>("8 s_add_u32", 8, w, dout, 1);
        run_rate<3>("8 v_fma + 2 ds_read + wait + 2 v_add (of 12)", 12, w, dout, 1);
        run_rate<4>("8 v_fma_f64", 8, w, dout, 1);
        run_rate<5>("8 v_rcp_f32", 8, w, dout, 1);
        run_rate<6>("8 v_pk_fma_f32", 8, w, dout, 1);
        run_rate<7>("8 v_add_f32_dpp row_shr:1", 8, w, dout, 1);
        run_rate<8>("8 ds_bpermute_b32", 8, w, dout, 1);
        run_rate<9>("8 v_mul_f64", 8, w, dout, 1);
        run_rate<10>("4 cvt_f64_f32 + 4 cvt_f32_f64", 8, w, dout, 1);
        run_rate<11>("8 v_cmp->sgpr + 4 s_and + 4 cndmask (of 16)", 16, w, dout, 1);
        run_rate<12>("8 cheap VALU (floor med3 cvt max min and or)", 8, w, dout, 1);
        run_rate<13>("8 v_fma + 8 s_waitcnt (of 16)", 16, w, dout, 1);
        run_rate<14>("8 ds_read_b32 + wait", 8, w, dout, 1);
        run_rate<15>("4 v_fma + 4 v_cmp vcc (of 8)", 8, w, dout, 1);
        printf("\n");
    }
    // v_cvt_pk_u8_f32 semantics
    {
        float h[16] = {0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 254.5f, 254.9f, 255.0f, 255.5f, 256.0f, 300.0f, -0.5f, -3.0f, NAN, 1e10f};
        float* din; uint32_t* dres; uint32_t res[16];
        CK(hipMalloc(&din, sizeof(h))); CK(hipMalloc(&dres, sizeof(res)));
        CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_cvt_pk, dim3(1), dim3(64), 0, 0, din, dres, 16);
        CK(hipMemcpy(res, dres, sizeof(res), hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; i++) printf("v_cvt_pk_u8_f32(%g, byte 1, 0xAABBCCDD) = 0x%08x -> %u\n", h[i], res[i], (res[i] >> 8) & 0xff);
    }
    // division variants: operand ranges of the polylines interpolation (numerator 0..4, denominator 2^-12..8)
    {
        const int n = 1 << 26;
        std::vector<float> a(n), b(n);
        uint64_t s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
        for (int i = 0; i < n; i++) {
            double den = exp2(-12.0 + 15.0 * rnd());
            b[i] = (float)den;
            a[i] = (float)(rnd() * den * ((i & 7) == 0 ? 1.2 : 1.0));
            if ((i & 1023) == 0) a[i] = 0.0f;
        }
        float *da, *db, *dq; std::vector<float> q0(n), q1(n);
        CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&dq, n * 4));
        CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_div, dim3(n / 256), dim3(256), 0, 0, da, db, dq, n, 0);
        CK(hipMemcpy(q0.data(), dq, n * 4, hipMemcpyDeviceToHost));
        long host_bad = 0;
        for (int i = 0; i < n; i++) { float h = a[i] / b[i]; if (memcmp(&h, &q0[i], 4)) host_bad++; }
        printf("division: device IEEE vs host IEEE mismatches: %ld / %d\n", host_bad, n);
        for (int mode = 1; mode <= 2; mode++) {
            hipLaunchKernelGGL(k_div, dim3(n / 256), dim3(256), 0, 0, da, db, dq, n, mode);
            CK(hipMemcpy(q1.data(), dq, n * 4, hipMemcpyDeviceToHost));
            long bad = 0;
            for (int i = 0; i < n; i++) if (memcmp(&q0[i], &q1[i], 4)) bad++;
            printf("division: variant %d (%s) vs IEEE mismatches: %ld / %d\n", mode, mode == 1 ? "FMA core without scale/fixup" : "one residual correction", bad, n);
        }
    }
    return 0;
}
