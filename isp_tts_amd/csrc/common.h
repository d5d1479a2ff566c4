// Shared helpers for the gfx950 kernels of libispk.so (see include/ispk.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/ispk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// thread-local last-error text (set by ISPK_FAIL, read through ispk_last_error_string)
char* ispk_err_buf();

#define ISPK_FAIL(code, ...)                          \
    do {                                              \
        snprintf(ispk_err_buf(), 256, __VA_ARGS__);   \
        return (code);                                \
    } while (0)

#define ISPK_REQUIRE(cond, code, ...) \
    do {                              \
        if (!(cond)) ISPK_FAIL(code, __VA_ARGS__); \
    } while (0)

// Dynamic LDS above 64 KiB needs a one-time per-kernel opt-in.  Done once per call site (an idempotent cache, the only
// mutable state in the library) so that launch functions stay free of non-stream API calls and can be graph-captured.
#define ISPK_RESERVE_LDS(kernel, bytes, what)                                                                       \
    do {                                                                                                            \
        static std::atomic<size_t> reserved_{0};                                                                    \
        if ((size_t)(bytes) > 64 * 1024 && reserved_.load(std::memory_order_acquire) < (size_t)(bytes)) {           \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                              \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));          \
            if (e_ != hipSuccess)                                                                                   \
                ISPK_FAIL((int32_t)e_, what ": cannot reserve %zu B of LDS: %s", (size_t)(bytes),                   \
                          hipGetErrorString(e_));                                                                   \
            reserved_.store((size_t)(bytes), std::memory_order_release);                                            \
        }                                                                                                           \
    } while (0)

static inline int32_t ispk_launch_status() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(ispk_err_buf(), 256, "kernel launch failed: %s", hipGetErrorString(e));
        return (int32_t)e;
    }
    return 0;
}

static inline bool ispk_aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

// bf16 <-> fp32 (round to nearest even; a plain cast keeps NaN a NaN on gfx950)
__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU(erf) with the Abramowitz-Stegun 7.1.26 erf (|error| <= 1.5e-7): 1 rcp + 1 exp + 7 FMA instead of libm erff.
// Used where the result is rounded to bf16 anyway (relative step 2^-9); the fp32 parity path keeps erff.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = 1.0f - poly * t * e;
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

constexpr int kWave = 64;
