// Do MFMA (wave A) and VALU (wave B) on the SAME SIMD overlap?  512 threads per block = 2 waves per SIMD; waves 0-3 run
// back-to-back MFMAs, waves 4-7 run packed-FMA chains (or exp).  Each reports its own cycle count; compare with each alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>  // 0: both, 1: MFMA waves only (others exit), 2: VALU waves only; bit 2 set: VALU = v_exp
__global__ __launch_bounds__(512, 1) void probe(uint64_t* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    const bool is_mfma = wave < 4;
    uint64_t t = 0;
    float sink = 0.f;
    if (is_mfma) {
        if ((MODE & 3) == 2) return;
        f32x16 acc[2];
        for (int i = 0; i < 2; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x); b[i] = (__bf16)(seed * 0.5f); }
        const uint64_t t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 1], 0, 0, 0);
        t = __builtin_readcyclecounter() - t0;
        sink = acc[0][0] + acc[1][0];
    } else {
        if ((MODE & 3) == 1) return;
        f32x2 v[8];
        for (int i = 0; i < 8; ++i) { v[i].x = seed * 1e-3f * (i + 1); v[i].y = seed * 2e-3f * (i + 1); }
        const uint64_t t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (MODE & 4) { v[j].x = __builtin_amdgcn_exp2f(v[j].x); v[j].y = __builtin_amdgcn_exp2f(v[j].y); }
                else { v[j] = v[j] * 1.0001f + 1e-7f; v[j] = v[j] * 0.9999f + 1e-7f; }
            }
        t = __builtin_readcyclecounter() - t0;
        for (int i = 0; i < 8; ++i) sink += v[i].x + v[i].y;
    }
    if (threadIdx.x % 64 == 0) {
        uint64_t* o = out + ((size_t)blockIdx.x * 8 + wave) * 2;
        o[0] = t; o[1] = sink == 1.f;
    }
}

template <int MODE>
void run(const char* name) {
    const int blocks = 256, iters = 4000;
    uint64_t* d;
    (void)hipMalloc(&d, (size_t)blocks * 8 * 16);
    (void)hipMemset(d, 0, (size_t)blocks * 8 * 16);
    probe<MODE><<<blocks, 512>>>(d, 10, 1.0f);
    probe<MODE><<<blocks, 512>>>(d, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> h((size_t)blocks * 8 * 2);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double tm = 0, tv = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? tm : tv) += h[(b * 8 + w) * 2];
    tm /= blocks * 4; tv /= blocks * 4;
    printf("%-40s MFMA wave: %6.2f cycles/MFMA   VALU wave: %6.2f cycles per 2 instructions\n", name, tm / (8.0 * iters),
           tv / (8.0 * iters));
    (void)hipFree(d);
}

int main() {
    run<1>("MFMA waves alone");
    run<2>("pk_fma waves alone");
    run<0>("MFMA + pk_fma waves together");
    run<6>("v_exp waves alone");
    run<4>("MFMA + v_exp waves together");
    return 0;
}
