// Clock / MFMA-rate / VALU-overlap probe (experiments; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
// Every wave runs N back-to-back v_mfma_f32_32x32x16_bf16 (optionally with K independent packed-fp32 FMAs in each gap)
// and reports s_memtime ticks, s_memrealtime ticks (100 MHz) and the kernel's wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VALU, bool DEP>
__global__ __launch_bounds__(256, 1) void probe(uint64_t* out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x); b[i] = (__bf16)(seed * 0.5f); }
    f32x2 v[4];
    for (int i = 0; i < 4; ++i) { v[i].x = seed + i; v[i].y = seed - i; }
    const f32x2 k1 = {1.0001f, 0.9999f}, k2 = {1e-6f, -1e-6f};
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            acc[DEP ? 0 : (j & 3)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[DEP ? 0 : (j & 3)], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < VALU; ++q) v[q & 3] = v[q & 3] * k1 + k2;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    for (int i = 0; i < 4; ++i) sink += acc[i][0] + v[i].x + v[i].y;
    if (threadIdx.x % 64 == 0) {
        uint64_t* o = out + ((size_t)blockIdx.x * 4 + threadIdx.x / 64) * 3;
        o[0] = t1 - t0; o[1] = r1 - r0; o[2] = (uint64_t)(sink == 123.f);
    }
}

template <int VALU, bool DEP>
void run(const char* name, int blocks, int iters) {
    uint64_t* d;
    hipMalloc(&d, (size_t)blocks * 4 * 3 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<VALU, DEP><<<blocks, 256>>>(d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<VALU, DEP><<<blocks, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h((size_t)blocks * 4 * 3);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double mt = 0, rt = 0;
    for (int i = 0; i < blocks * 4; ++i) { mt += h[3 * i]; rt += h[3 * i + 1]; }
    mt /= blocks * 4; rt /= blocks * 4;
    const double nm = 8.0 * iters;
    printf("%-34s blocks %4d: wall %8.1f us | memtime %10.0f ticks (%.3f GHz vs realtime 100 MHz) | %6.2f memtime ticks/MFMA | "
           "%6.2f ns/MFMA | %7.1f TF/s chip\n", name, blocks, ms * 1e3, mt, mt / (rt / 100e6) * 1e-9, mt / nm,
           rt / 100e6 / nm * 1e9, blocks * 4 * nm * 32768.0 / (ms * 1e-3) * 1e-12);
    hipFree(d);
}

int main() {
    const int it = 20000;
    run<0, false>("mfma only, 4 accumulators", 1, it);
    run<0, false>("mfma only, 4 accumulators", 256, it);
    run<0, true>("mfma only, 1 accumulator (dep)", 256, it);
    run<2, false>("mfma + 2 pk_fma per gap", 256, it);
    run<4, false>("mfma + 4 pk_fma per gap", 256, it);
    run<6, false>("mfma + 6 pk_fma per gap", 256, it);
    run<8, false>("mfma + 8 pk_fma per gap", 256, it);
    run<12, false>("mfma + 12 pk_fma per gap", 256, it);
    run<8, true>("dep mfma + 8 pk_fma per gap", 256, it);
    return 0;
}
