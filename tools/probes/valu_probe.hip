// VALU instruction cost probe (experiments): cycles per wave-instruction, one wave per SIMD and three.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(uint64_t* out, int iters, float seed) {
    float v[8];
    f32x2 w[4];
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1) * 1e-3f + threadIdx.x * 1e-6f;
    for (int i = 0; i < 4; ++i) { w[i].x = v[i]; w[i].y = v[i + 4]; }
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int i = j & 7;
            if constexpr (OP == 0) v[i] = __builtin_amdgcn_exp2f(v[i]);
            if constexpr (OP == 1) v[i] = fmaf(v[i], 1.0001f, 1e-7f);
            if constexpr (OP == 2) w[i & 3] = w[i & 3] * 1.0001f + 1e-7f;
            if constexpr (OP == 3) v[i] = fmaxf(fmaxf(v[i], v[(i + 1) & 7]), v[(i + 2) & 7]);
            if constexpr (OP == 4) acc += __builtin_bit_cast(uint32_t, __builtin_convertvector(w[i & 3], bf16x2_t));
            if constexpr (OP == 5) v[i] = __builtin_amdgcn_rcpf(v[i]);
            if constexpr (OP == 6) w[i & 3] = w[i & 3] * w[(i + 1) & 3];
            if constexpr (OP == 7) w[i & 3] = w[i & 3] + w[(i + 1) & 3];
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float sink = (float)acc;
    for (int i = 0; i < 8; ++i) sink += v[i];
    for (int i = 0; i < 4; ++i) sink += w[i].x + w[i].y;
    if (threadIdx.x % 64 == 0) {
        uint64_t* o = out + ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2;
        o[0] = t1 - t0; o[1] = sink == 1.f;
    }
}

template <int OP>
void run(const char* name, int threads) {
    const int blocks = 256, iters = 2000, waves = threads / 64;
    uint64_t* d;
    (void)hipMalloc(&d, (size_t)blocks * waves * 16);
    probe<OP><<<blocks, threads>>>(d, 10, 1.0f);
    probe<OP><<<blocks, threads>>>(d, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> h((size_t)blocks * waves * 2);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double t = 0;
    for (int i = 0; i < blocks * waves; ++i) t += h[2 * i];
    t /= blocks * waves;
    printf("%-22s %2d waves/SIMD: %6.2f cycles per instruction per wave  (%5.2f per SIMD)\n", name, waves / 4,
           t / (32.0 * iters), t / (32.0 * iters) / (waves / 4));
    (void)hipFree(d);
}

int main() {
    for (int th : {256, 768}) {
        run<0>("v_exp_f32", th);
        run<5>("v_rcp_f32", th);
        run<1>("v_fma_f32", th);
        run<2>("v_pk_fma_f32", th);
        run<6>("v_pk_mul_f32", th);
        run<7>("v_pk_add_f32", th);
        run<3>("v_max3_f32", th);
        run<4>("v_cvt_pk_bf16_f32+add", th);
    }
    return 0;
}
