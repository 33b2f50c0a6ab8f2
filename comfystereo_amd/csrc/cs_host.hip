// cs_host.hip -- host half of the compact node boundary (reference GenerateStereo.py:41-44, :159-177,
// :355-378: convertResult / np2tensor / generate_mask produce float32 tensors from uint8 images on the CPU).
//
// Every output value of the CPU techniques is one of 256 codes (stereoscope k / 255, depth maps trunc(d * 255) mod 256 over
// 255 on three equal channels, mask 0 / 1), so a node call that must hand CPU float32 tensors to ComfyUI crosses PCIe with
// one BYTE per value (83 MB instead of 697 MB per 4K frame) and the float32 tensors are written by the host cores:
// cs_host_expand_u8, several threads, a 256-entry table of the true quotients k / 255.0f (IEEE division, what the device
// kernels produce by arithmetic, cs_math.h code_over_255).  No device code in this file.
#include <emmintrin.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <thread>
#include <vector>

#include "../../include/comfystereo_amd.h"

namespace {

struct Lut {
    float q[256], m[256];
    Lut() {
        for (int k = 0; k < 256; k++) {
            q[k] = (float)k / 255.0f;      // true division (np2tensor: astype(float32) / 255.0)
            m[k] = k ? 1.0f : 0.0f;        // mask flag
        }
    }
};
const Lut g_lut;

// The result tensors are written once and not read back by these threads: non-temporal 16-byte stores (SSE2, baseline x86-64)
// skip the read-for-ownership of every destination line, i.e. half of the memory traffic of a plain store loop.
void expand_range(const uint8_t* codes, float* out, size_t i0, size_t i1, int replicate, const float* lut) {
    const uint8_t* c = codes + i0;
    size_t n = i1 - i0;
    if (replicate == 1) {
        float* o = out + i0;
        while (n && ((uintptr_t)o & 15)) { *o++ = lut[*c++]; n--; }
        for (; n >= 8; n -= 8, c += 8, o += 8) {
            uint64_t pk;
            memcpy(&pk, c, 8);
            _mm_stream_ps(o, _mm_setr_ps(lut[pk & 0xff], lut[(pk >> 8) & 0xff], lut[(pk >> 16) & 0xff], lut[(pk >> 24) & 0xff]));
            _mm_stream_ps(o + 4, _mm_setr_ps(lut[(pk >> 32) & 0xff], lut[(pk >> 40) & 0xff], lut[(pk >> 48) & 0xff], lut[pk >> 56]));
        }
        for (; n; n--) *o++ = lut[*c++];
    } else if (replicate == 3) {
        float* o = out + i0 * 3;
        while (n && ((uintptr_t)o & 15)) { const float v = lut[*c++]; o[0] = v; o[1] = v; o[2] = v; o += 3; n--; }
        for (; n >= 4; n -= 4, c += 4, o += 12) {   // a a a b | b b c c | c d d d
            uint32_t pk;
            memcpy(&pk, c, 4);
            const float a = lut[pk & 0xff], b = lut[(pk >> 8) & 0xff], cc = lut[(pk >> 16) & 0xff], d = lut[pk >> 24];
            _mm_stream_ps(o, _mm_setr_ps(a, a, a, b));
            _mm_stream_ps(o + 4, _mm_setr_ps(b, b, cc, cc));
            _mm_stream_ps(o + 8, _mm_setr_ps(cc, d, d, d));
        }
        for (; n; n--) { const float v = lut[*c++]; o[0] = v; o[1] = v; o[2] = v; o += 3; }
    } else {
        for (size_t i = i0; i < i1; i++) {
            const float v = lut[codes[i]];
            float* o = out + i * (size_t)replicate;
            for (int r = 0; r < replicate; r++) o[r] = v;
        }
    }
    _mm_sfence();
}

void replicate_range(const float* in, float* out, size_t i0, size_t i1, int replicate) {
    const float* c = in + i0;
    size_t n = i1 - i0;
    if (replicate == 3) {
        float* o = out + i0 * 3;
        while (n && ((uintptr_t)o & 15)) { const float v = *c++; o[0] = v; o[1] = v; o[2] = v; o += 3; n--; }
        for (; n >= 4; n -= 4, c += 4, o += 12) {   // a a a b | b b c c | c d d d
            const float a = c[0], b = c[1], cc = c[2], d = c[3];
            _mm_stream_ps(o, _mm_setr_ps(a, a, a, b));
            _mm_stream_ps(o + 4, _mm_setr_ps(b, b, cc, cc));
            _mm_stream_ps(o + 8, _mm_setr_ps(cc, d, d, d));
        }
        for (; n; n--) { const float v = *c++; o[0] = v; o[1] = v; o[2] = v; o += 3; }
    } else {
        for (size_t i = i0; i < i1; i++) {
            float* o = out + i * (size_t)replicate;
            for (int r = 0; r < replicate; r++) o[r] = in[i];
        }
    }
    _mm_sfence();
}

// contiguous slices of [0, count), one per thread, boundaries on multiples of 1024 values
template <class F>
void run_slices(size_t count, int threads, size_t min_per_thread, F&& f) {
    long hw = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = threads > 0 ? threads : (int)(hw > 0 ? hw : 1);
    if (nt > 64) nt = 64;
    if ((size_t)nt > (count + min_per_thread - 1) / min_per_thread) nt = (int)((count + min_per_thread - 1) / min_per_thread);
    if (nt <= 1) { f((size_t)0, count); return; }
    const size_t per = ((count + nt - 1) / nt + 1023) & ~(size_t)1023;
    std::vector<std::thread> pool;
    pool.reserve(nt);
    for (int t = 0; t < nt; t++) {
        const size_t i0 = (size_t)t * per, i1 = i0 + per < count ? i0 + per : count;
        if (i0 >= count) break;
        pool.emplace_back(f, i0, i1);
    }
    for (auto& th : pool) th.join();
}

void advise_huge(void* p, size_t bytes) {
    // fresh result tensors are untouched anonymous memory: ask for huge pages before the first touch (512 x fewer page faults
    // where transparent huge pages are available on request; a no-op otherwise)
    const uintptr_t a = ((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
    const uintptr_t b = ((uintptr_t)p + bytes) & ~(uintptr_t)((2u << 20) - 1);
    if (b > a) (void)madvise((void*)a, b - a, MADV_HUGEPAGE);
}

}  // namespace

extern "C" {

int cs_host_replicate_f32(const float* values, float* out, size_t count, int replicate, int threads) {
    if (!values || !out || replicate < 1 || replicate > 4) return CS_EINVAL;
    if (count == 0) return CS_OK;
    advise_huge(out, count * (size_t)replicate * 4);
    run_slices(count, threads, (size_t)1 << 16, [=](size_t i0, size_t i1) { replicate_range(values, out, i0, i1, replicate); });
    return CS_OK;
}

int cs_host_expand_u8(const uint8_t* codes, float* out, size_t count, int replicate, int mode, int threads) {
    if (!codes || !out || replicate < 1 || replicate > 4 || mode < 0 || mode > 1) return CS_EINVAL;
    if (count == 0) return CS_OK;
    const float* lut = mode ? g_lut.m : g_lut.q;
    advise_huge(out, count * (size_t)replicate * 4);
    // (each thread first-touches its own pages)
    run_slices(count, threads, (size_t)1 << 16, [=](size_t i0, size_t i1) { expand_range(codes, out, i0, i1, replicate, lut); });
    return CS_OK;
}

// Parallel memcpy for the input staging (pageable caller tensor -> pinned buffer): contiguous slices, one per thread.
int cs_host_copy(void* dst, const void* src, size_t bytes, int threads) {
    if (!dst || !src) return CS_EINVAL;
    if (bytes == 0) return CS_OK;
    long hw = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = threads > 0 ? threads : (int)(hw > 0 ? hw : 1);
    if (nt > 64) nt = 64;
    const size_t min_per_thread = (size_t)4 << 20;
    if ((size_t)nt > (bytes + min_per_thread - 1) / min_per_thread) nt = (int)((bytes + min_per_thread - 1) / min_per_thread);
    if (nt <= 1) { memcpy(dst, src, bytes); return CS_OK; }
    const size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
    std::vector<std::thread> pool;
    pool.reserve(nt);
    for (int t = 0; t < nt; t++) {
        const size_t o = (size_t)t * per;
        if (o >= bytes) break;
        const size_t len = o + per < bytes ? per : bytes - o;
        pool.emplace_back([=] { memcpy((char*)dst + o, (const char*)src + o, len); });
    }
    for (auto& th : pool) th.join();
    return CS_OK;
}

}  // extern "C"
