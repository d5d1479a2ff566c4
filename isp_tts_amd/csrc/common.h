// Shared helpers for the gfx950 kernels of libispk.so (see include/ispk.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <type_traits>

#include "../../include/ispk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // native vector: HIP's uint4 (a union type) can block SROA

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

// Experiment knobs (tools/stamp_*.py, tools/sweep_gemm.py): compiled in only with -DISPK_EXPERIMENTS
// (`python -m isp_tts_amd.build --experiments` -> libispk_exp.so).  In the product build every knob reads as "unset" at
// compile time, the branches that test one are removed, and nothing the library does depends on the process environment.
#ifdef ISPK_EXPERIMENTS
#include <stdlib.h>
static inline const char* ispk_knob(const char* name) { return getenv(name); }
#else
#define ispk_knob(name) (static_cast<const char*>(nullptr))
#endif

static inline bool ispk_aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

// bf16 <-> fp32 (round to nearest even; a plain cast keeps NaN a NaN on gfx950)
__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
typedef float f32x2 __attribute__((ext_vector_type(2)));

// GELU(erf) for outputs that are rounded to bf16 anyway, two elements at a time so the polynomial runs on packed fp32
// (v_pk_fma_f32 / v_pk_mul_f32).  erf by Abramowitz-Stegun 7.1.28: erf(z) = 1 - (1 + a1 z + .. + a6 z^6)^-16, z >= 0,
// |error| <= 3e-7; then gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) = 0.5 x + 0.5 |x| (1 - q^-16).
// 6 FMA + 4 squarings + 1 rcp per element (libm erff: ~40 instructions with branches).  The fp32 parity path keeps erff.
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    f32x2 ax;
    ax.x = fabsf(x.x);
    ax.y = fabsf(x.y);
    const f32x2 z = ax * 0.70710678118654752440f;
    f32x2 q = z * 0.0000430638f + 0.0002765672f;
    q = q * z + 0.0001520143f;
    q = q * z + 0.0092705272f;
    q = q * z + 0.0422820123f;
    q = q * z + 0.0705230784f;
    q = q * z + 1.0f;
    q = q * q;
    q = q * q;
    q = q * q;
    q = q * q;  // q^16 (overflows to +inf for |x| > ~25: the reciprocal is then exactly 0, as it should be)
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(q.x);
    r.y = __builtin_amdgcn_rcpf(q.y);
    const f32x2 hx = ax * 0.5f;   // max(x, 0) = 0.5 x + 0.5 |x|: three packed operations, no v_max pair with its canonicalising copies
    return x * 0.5f + (hx - hx * r);
}
__device__ __forceinline__ float gelu_fast(float x) {
    f32x2 v;
    v.x = x;
    v.y = x;
    return gelu_fast2(v).x;
}
// The same erf (Abramowitz-Stegun 7.1.28, |error| <= 3e-7) in plain scalar fp32, for fp32 outputs: |gelu error| <= 1.5e-7 |x|,
// two orders below the fp32 path's 1e-4 mel bar, at 16 instructions instead of libm erff's ~40 with branches.
__device__ __forceinline__ float gelu_as28(float x) {
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752440f;
    float q = fmaf(z, 0.0000430638f, 0.0002765672f);
    q = fmaf(q, z, 0.0001520143f);
    q = fmaf(q, z, 0.0092705272f);
    q = fmaf(q, z, 0.0422820123f);
    q = fmaf(q, z, 0.0705230784f);
    q = fmaf(q, z, 1.0f);
    q = q * q;
    q = q * q;
    q = q * q;
    q = q * q;
    const float r = 1.0f / q;           // (IEEE division: q^16 overflows to +inf for |x| > ~25, 1/inf = 0 as it should be)
    const float hx = 0.5f * ax;
    return fmaf(0.5f, x, fmaf(-hx, r, hx));
}
// d gelu / dx for bf16 tensors: cdf(x) + x pdf(x) with the 3e-7 erf above and v_exp_f32 (the bf16 GELU backward and the
// fused da -> du epilogue of the training step evaluate the SAME expression)
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float ax = fabsf(x), z = ax * 0.70710678118654752440f;
    float q = fmaf(z, 0.0000430638f, 0.0002765672f);
    q = fmaf(q, z, 0.0001520143f);
    q = fmaf(q, z, 0.0092705272f);
    q = fmaf(q, z, 0.0422820123f);
    q = fmaf(q, z, 0.0705230784f);
    q = fmaf(q, z, 1.0f);
    q = q * q; q = q * q; q = q * q; q = q * q;
    const float erf_abs = 1.0f - __builtin_amdgcn_rcpf(q);                      // erf(|x| / sqrt 2)
    const float cdf = 0.5f * (1.f + (x < 0.f ? -erf_abs : erf_abs));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

constexpr int kWave = 64;

// ---- hand-scheduled LDS reads (guide §5.7): hipcc sinks every ds_read next to its consumer and keeps at most two in
// flight, which makes short MFMA loops LDS-latency-bound.  These reads are opaque to its scheduler; the caller counts
// them with lds_wait<N>() (LDS operations complete in order, so "at most N outstanding" retires everything older)
// and fences the consumer with __builtin_amdgcn_sched_barrier(0) after the wait (rule 18).
template <int OFF>
__device__ __forceinline__ void lds_read_b128_asm(bf16x8& dst, uint32_t lds_byte_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF) : "memory");
}
// two 8-byte reads (byte offsets 8*O0 and 8*O1 from the address) into one 128-bit fragment
template <int O0, int O1>
__device__ __forceinline__ void lds_read2_b64_asm(bf16x8& dst, uint32_t lds_byte_addr) {
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(lds_byte_addr), "n"(O0), "n"(O1) : "memory");
}
// same read with the destination in the ACCUMULATOR half of the unified register file ("=a"): for kernels whose arch
// VGPRs are full, where hipcc otherwise loads to a VGPR and copies to an AGPR (4 v_accvgpr_write + hazard nops) before
// every MFMA.  MFMA A/B operands may be AGPRs on gfx950.
template <int OFF>
__device__ __forceinline__ void lds_read_b128_asm_acc(bf16x8& dst, uint32_t lds_byte_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(lds_byte_addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
