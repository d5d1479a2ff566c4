// LayerNorm / AdaptiveLayerNorm (+ row mask), small Linear, fp32->bf16 cast.  HBM-bound row kernels:
// one wavefront per row, the row held in registers (D <= 1024 -> <= 16 values per lane), 256-B coalesced
// accesses per wave instruction, statistics by two in-register passes (mean, then centred variance) reduced
// across the wave with DPP/permute shuffles.  No LDS.
//
// Semantics: /root/reference/tts/modules/transformer/normalization.py:20-27, :37-61 (F.layer_norm, eps 1e-5,
// biased variance) and the row masking of transformer.py:101-102, :205-206.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename OutT>
__device__ __forceinline__ void store_out(OutT* p, float v);
template <>
__device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store_out<uint16_t>(uint16_t* p, float v) { *p = f32_to_bf16(v); }

// NV = D / 64 values per lane (compile-time so the row lives in registers)
template <int NV, typename OutT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ ada_scale,
                                                        const float* __restrict__ ada_shift, int64_t ada_stride,
                                                        int rows_per_batch, const uint8_t* __restrict__ row_mask,
                                                        OutT* __restrict__ y, int64_t ldy, int rows, float eps,
                                                        int64_t y_plane) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int D = NV * 64;
    const float* xr = x + (int64_t)row * ldx;
    float v[NV];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        v[c] = xr[lane + 64 * c];
        s += v[c];
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const float d = v[c] - mean;
        ss += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / D) + eps);
    const float mk = row_mask ? (row_mask[row] ? 1.0f : 0.0f) : 1.0f;
    const float* sc = gamma;
    const float* sh = beta;
    if (ada_scale) {
        const int64_t off = (int64_t)(row / rows_per_batch) * ada_stride;
        sc = ada_scale + off;
        sh = ada_shift ? ada_shift + off : nullptr;
    }
    OutT* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = lane + 64 * c;
        float o = (v[c] - mean) * rstd;
        if (sc) o *= sc[col];
        if (sh) o += sh[col];
        if constexpr (sizeof(OutT) == 2) {
            if (y_plane) {   // split fp16 planes (kernel-uniform)
                const float val = __builtin_amdgcn_fmed3f(o * mk, -65504.0f, 65504.0f);
                const _Float16 hh = (_Float16)val;
                yr[col] = __builtin_bit_cast(uint16_t, hh);
                yr[col + y_plane] = __builtin_bit_cast(uint16_t, (_Float16)(val - (float)hh));
                continue;
            }
        }
        store_out<OutT>(yr + col, o * mk);
    }
}

// D % 128 == 0 (the model's 256 / 384): TWO rows per wavefront, 32 lanes each, so a lane's share is NV4 = D/128 float4s -
// 16-byte loads, 8- (bf16) or 16-byte stores and a third of the instructions of the scalar kernel above, which moved 4
// bytes per lane per instruction (and 2-byte bf16 stores) and ran the 32,768-row decoder norms at 2.7 TB/s.
// Reductions stay inside a 32-lane half (xor 16 .. 1).
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int NV4, typename OutT>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ ada_scale,
                                                            const float* __restrict__ ada_shift, int64_t ada_stride,
                                                            int rows_per_batch, const uint8_t* __restrict__ row_mask,
                                                            OutT* __restrict__ y, int64_t ldy, int rows, float eps,
                                                            int64_t y_plane) {
    const int lane = threadIdx.x & 63, l = lane & 31;
    const int row_raw = blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + (lane >> 5);
    const int row = row_raw < rows ? row_raw : rows - 1;   // out-of-range halves recompute the last row and store nothing
    constexpr int D = NV4 * 128;
    const float* xr = x + (int64_t)row * ldx;
    float4 v[NV4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NV4; ++c) {
        v[c] = *reinterpret_cast<const float4*>(xr + 4 * (l + 32 * c));
        s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    }
    const float mean = half_sum(s) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NV4; ++c) {
        const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
        ss += (a * a + b * b) + (cc * cc + d * d);
    }
    const float rstd = 1.0f / sqrtf(half_sum(ss) * (1.0f / D) + eps);
    const float mk = row_mask ? (row_mask[row] ? 1.0f : 0.0f) : 1.0f;
    const float* sc = gamma;
    const float* sh = beta;
    if (ada_scale) {
        const int64_t off = (int64_t)(row / rows_per_batch) * ada_stride;
        sc = ada_scale + off;
        sh = ada_shift ? ada_shift + off : nullptr;
    }
    if (row_raw >= rows) return;
    OutT* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int c = 0; c < NV4; ++c) {
        const int col = 4 * (l + 32 * c);
        float4 o;
        o.x = (v[c].x - mean) * rstd; o.y = (v[c].y - mean) * rstd;
        o.z = (v[c].z - mean) * rstd; o.w = (v[c].w - mean) * rstd;
        if (sc) {
            const float4 g = *reinterpret_cast<const float4*>(sc + col);
            o.x *= g.x; o.y *= g.y; o.z *= g.z; o.w *= g.w;
        }
        if (sh) {
            const float4 bb = *reinterpret_cast<const float4*>(sh + col);
            o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w;
        }
        o.x *= mk; o.y *= mk; o.z *= mk; o.w *= mk;
        if constexpr (sizeof(OutT) == 4) {
            *reinterpret_cast<float4*>(yr + col) = o;
        } else if (y_plane) {   // split fp16 planes (kernel-uniform): hi = fp16(v), lo = fp16(v - hi)
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const float vv[4] = {__builtin_amdgcn_fmed3f(o.x, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(o.y, -65504.0f, 65504.0f),
                                 __builtin_amdgcn_fmed3f(o.z, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(o.w, -65504.0f, 65504.0f)};
            uint2 ph, pl;
            h2 a, b, c, d;
            a.x = (_Float16)vv[0]; a.y = (_Float16)vv[1]; b.x = (_Float16)vv[2]; b.y = (_Float16)vv[3];
            c.x = (_Float16)(vv[0] - (float)a.x); c.y = (_Float16)(vv[1] - (float)a.y);
            d.x = (_Float16)(vv[2] - (float)b.x); d.y = (_Float16)(vv[3] - (float)b.y);
            ph.x = __builtin_bit_cast(uint32_t, a); ph.y = __builtin_bit_cast(uint32_t, b);
            pl.x = __builtin_bit_cast(uint32_t, c); pl.y = __builtin_bit_cast(uint32_t, d);
            *reinterpret_cast<uint2*>(yr + col) = ph;
            *reinterpret_cast<uint2*>(yr + y_plane + col) = pl;
        } else {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
            pk.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
            *reinterpret_cast<uint2*>(yr + col) = pk;
        }
    }
}

template <typename OutT>
int32_t layernorm_dispatch(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* ada_scale,
                           const float* ada_shift, int64_t ada_stride, int32_t rows_per_batch, const uint8_t* row_mask,
                           OutT* y, int64_t ldy, int32_t rows, int32_t D, float eps, ispk_stream_t stream,
                           int64_t y_plane = 0) {
    ISPK_REQUIRE(x && y, ISPK_E_NULL, "layernorm: null x/y");
    ISPK_REQUIRE(y_plane % 4 == 0 && (y_plane == 0 || sizeof(OutT) == 2), ISPK_E_ALIGN, "layernorm: y_plane %% 4 (split output)");
    ISPK_REQUIRE(rows >= 0 && D >= 64 && D <= 1024 && D % 64 == 0, ISPK_E_SHAPE,
                 "layernorm: D=%d must be a multiple of 64 in [64, 1024]", D);
    ISPK_REQUIRE(ldx >= D && ldy >= D, ISPK_E_SHAPE, "layernorm: leading stride < D");
    ISPK_REQUIRE(!ada_scale || rows_per_batch >= 1, ISPK_E_SHAPE, "layernorm: rows_per_batch must be >= 1");
    if (rows == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // vector path: 16-byte aligned rows of D % 128 == 0 values (parameters 16-byte aligned as well)
    const bool vec_ok = D % 128 == 0 && D <= 512 && ldx % 4 == 0 && ldy % 4 == 0 && ispk_aligned(x, 16) &&
                        ispk_aligned(y, sizeof(OutT) * 4) && (!gamma || ispk_aligned(gamma, 16)) &&
                        (!beta || ispk_aligned(beta, 16)) &&
                        (!ada_scale || (ispk_aligned(ada_scale, 16) && ada_stride % 4 == 0)) &&
                        (!ada_shift || ispk_aligned(ada_shift, 16)) && ispk_knob("ISPK_LN_SCALAR") == nullptr;
    if (vec_ok) {
        dim3 grid8((rows + 7) / 8), block8(256);
#define ISPK_LNV_CASE(NV4)                                                                                               \
    case NV4:                                                                                                           \
        hipLaunchKernelGGL((layernorm_vec_kernel<NV4, OutT>), grid8, block8, 0, s, x, ldx, gamma, beta, ada_scale,      \
                           ada_shift, ada_stride, rows_per_batch, row_mask, y, ldy, rows, eps, y_plane);                \
        break;
        switch (D / 128) { ISPK_LNV_CASE(1) ISPK_LNV_CASE(2) ISPK_LNV_CASE(3) ISPK_LNV_CASE(4) }
#undef ISPK_LNV_CASE
        return ispk_launch_status();
    }
    dim3 grid((rows + 3) / 4), block(256);
#define ISPK_LN_CASE(NV)                                                                                              \
    case NV:                                                                                                          \
        hipLaunchKernelGGL((layernorm_kernel<NV, OutT>), grid, block, 0, s, x, ldx, gamma, beta, ada_scale, ada_shift, \
                           ada_stride, rows_per_batch, row_mask, y, ldy, rows, eps, y_plane);                         \
        break;
    switch (D / 64) {
        ISPK_LN_CASE(1) ISPK_LN_CASE(2) ISPK_LN_CASE(3) ISPK_LN_CASE(4) ISPK_LN_CASE(5) ISPK_LN_CASE(6)
        ISPK_LN_CASE(7) ISPK_LN_CASE(8) ISPK_LN_CASE(9) ISPK_LN_CASE(10) ISPK_LN_CASE(11) ISPK_LN_CASE(12)
        ISPK_LN_CASE(13) ISPK_LN_CASE(14) ISPK_LN_CASE(15) ISPK_LN_CASE(16)
        default: ISPK_FAIL(ISPK_E_SHAPE, "layernorm: unsupported D=%d", D);
    }
#undef ISPK_LN_CASE
    return ispk_launch_status();
}

// ------------------------------------------------------------------------------------------------ small Linear
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ a, int64_t lda,
                                                           const float* __restrict__ w, int64_t ldw,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ resid, int64_t ldr,
                                                           float* __restrict__ out, int64_t ldo, int M, int N, int K,
                                                           uint32_t act) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
    const float* ar = a + (int64_t)i * lda;
    const float* wr = w + (int64_t)j * ldw;
    // one FMA chain in k order (the summation order of the CPU oracle); the loads of 8 steps are issued together - with a
    // plain loop every step waited for its own two loads, which made these tiny launches 20-50 us long
    float acc = 0.f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
        float av[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = ar[k + u]; wv[u] = wr[k + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(av[u], wv[u], acc);
    }
    if (k < K) {   // tail of up to 7 steps, loads again issued together (clamped index, zero weight past the end:
        float av[8], wv[8];   // fma(a, +0, acc) leaves acc unchanged for finite a)
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int kk = k + u < K ? k + u : K - 1;
            av[u] = ar[kk];
            wv[u] = k + u < K ? wr[kk] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) acc = fmaf(av[u], wv[u], acc);
    }
    if (bias) acc += bias[j];
    if (act & ISPK_EP_GELU) acc = gelu_erf(acc);
    if (act & ISPK_EP_SILU) acc = silu(acc);
    if (resid) acc += resid[(int64_t)i * ldr + j];
    out[(int64_t)i * ldo + j] = acc;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, int64_t ldx,
                                                        uint16_t* __restrict__ y, int64_t ldy, int rows, int cols4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * cols4) return;
    const int r = (int)(idx / cols4), c = (int)(idx - (int64_t)r * cols4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + c);
    uint2 o;
    o.x = f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    o.y = f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(y + (int64_t)r * ldy + c) = o;
}

}  // namespace

extern "C" int32_t ispk_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                      const float* ada_scale, const float* ada_shift, int64_t ada_stride,
                                      int32_t rows_per_batch, const uint8_t* row_mask, float* y, int64_t ldy,
                                      int32_t rows, int32_t D, float eps, ispk_stream_t stream) {
    return layernorm_dispatch<float>(x, ldx, gamma, beta, ada_scale, ada_shift, ada_stride, rows_per_batch, row_mask, y,
                                     ldy, rows, D, eps, stream);
}

extern "C" int32_t ispk_layernorm_f32_bf16(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                           const float* ada_scale, const float* ada_shift, int64_t ada_stride,
                                           int32_t rows_per_batch, const uint8_t* row_mask, uint16_t* y, int64_t ldy,
                                           int32_t rows, int32_t D, float eps, ispk_stream_t stream) {
    return layernorm_dispatch<uint16_t>(x, ldx, gamma, beta, ada_scale, ada_shift, ada_stride, rows_per_batch, row_mask,
                                        y, ldy, rows, D, eps, stream);
}

extern "C" int32_t ispk_layernorm_f32_split(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                            const float* ada_scale, const float* ada_shift, int64_t ada_stride,
                                            int32_t rows_per_batch, const uint8_t* row_mask, uint16_t* y_hi, int64_t ldy,
                                            int64_t y_plane, int32_t rows, int32_t D, float eps, ispk_stream_t stream) {
    ISPK_REQUIRE(y_plane > 0, ISPK_E_SHAPE, "layernorm_split: y_plane must be positive");
    return layernorm_dispatch<uint16_t>(x, ldx, gamma, beta, ada_scale, ada_shift, ada_stride, rows_per_batch, row_mask,
                                        y_hi, ldy, rows, D, eps, stream, y_plane);
}

extern "C" int32_t ispk_linear_small_f32(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias,
                                         const float* resid, int64_t ldr, float* out, int64_t ldo, int32_t M, int32_t N,
                                         int32_t K, uint32_t act, ispk_stream_t stream) {
    ISPK_REQUIRE(a && w && out, ISPK_E_NULL, "linear_small: null pointer");
    ISPK_REQUIRE(M >= 0 && N >= 1 && K >= 1, ISPK_E_SHAPE, "linear_small: bad shape M=%d N=%d K=%d", M, N, K);
    ISPK_REQUIRE(lda >= K && ldw >= K && ldo >= N && (!resid || ldr >= N), ISPK_E_SHAPE, "linear_small: bad strides");
    if (M == 0) return 0;
    const int64_t total = (int64_t)M * N;
    hipLaunchKernelGGL(linear_small_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a, lda, w, ldw, bias, resid, ldr, out, ldo, M, N, K, act);
    return ispk_launch_status();
}

extern "C" int32_t ispk_cast_f32_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int32_t rows, int32_t cols,
                                      ispk_stream_t stream) {
    ISPK_REQUIRE(x && y, ISPK_E_NULL, "cast: null pointer");
    ISPK_REQUIRE(rows >= 0 && cols >= 4 && cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, ISPK_E_SHAPE,
                 "cast: cols and strides must be multiples of 4");
    ISPK_REQUIRE(ispk_aligned(x, 16) && ispk_aligned(y, 8), ISPK_E_ALIGN, "cast: unaligned pointer");
    if (rows == 0) return 0;
    const int64_t total = (int64_t)rows * (cols / 4);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, ldx, y, ldy, rows, cols / 4);
    return ispk_launch_status();
}
