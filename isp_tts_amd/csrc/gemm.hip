// Linear layers on the matrix cores: C = epilogue(A · Wᵀ), A [M][K], W [N][K] (both K-contiguous: an "NT" GEMM,
// which is what nn.Linear's [out][in] weight layout gives for free).
//
// fp32 path: v_mfma_f32_32x32x2_f32 — exact fp32 products with fp32 accumulation (bitwise an fmaf chain), so this
// path carries the 1e-4 mel parity bar.  Its rate is 64 FLOP/clk/SIMD (157 TF/chip), i.e. it is MFMA-bound by a wide
// margin, so the design spends nothing on clever staging: register-staged double-buffered LDS tiles, one barrier per
// 32-deep K step, and every LDS read is a conflict-free ds_read_b128:
//   * tile rows are padded to 36 dwords: row r starts at bank 36r mod 64 = {0,36,8,44,...}, which puts the 16 lanes
//     of every ds_read_b128 lane group on 16 distinct 4-bank slots;
//   * a lane does not read k, k+2, k+4.. (the MFMA's natural k pairing) but 4 CONSECUTIVE k (one b128): lane half h
//     of step s supplies k = 8*kq + 4*h + s for BOTH operands.  A sum over k does not care which k meets which MFMA
//     step as long as A and B agree, so 1 LDS instruction feeds 4 MFMAs per operand tile.
// Wave layout: 4 waves as 2x2, each wave TM x TN tiles of 32x32 (block = 64*TM x 64*TN), accumulators in registers.
// C/D fragment (guide §3): col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5): a store instruction
// writes two 128-B row segments, or — for to_mel with the operands swapped — 32 consecutive mel frames.
//
// Replaces the nn.Linear call sites listed in include/ispk.h (attention.py:105,111,168; feedforward.py:33-36;
// transformer.py:170; model.py:167-168 of the reference) and fuses the surrounding bias / GELU / residual / mask ops.
#include <stdlib.h>

#include "common.h"

namespace {

struct GemmParams {
    const void* A;
    int64_t lda;
    const void* W;
    int64_t ldw;
    void* C;
    int64_t ldc;
    const float* bias;
    const void* resid;
    int64_t ldr;
    const uint8_t* mask;
    int M, N, K;
    uint32_t flags;
    int cpb;
    int64_t bstride;
    // fused LayerNorm of the OUTPUT rows (wide bf16 kernel only, ispk_gemm_bf16_ln)
    const float* ln_gamma = nullptr;
    const float* ln_beta = nullptr;
    void* ln_out = nullptr;
    int64_t ln_ld = 0;
    float ln_eps = 1e-5f;
    uint32_t ln_flags = 0;
};

constexpr int kLdt = 36;  // padded LDS row length in dwords (32 + 4)

__device__ __forceinline__ void epilogue_store(const GemmParams& p, int i, int j, float v) {
    if (i >= p.M || j >= p.N) return;
    if (p.bias) v += p.bias[(p.flags & ISPK_EP_BIAS_ROW) ? i : j];
    if (p.flags & ISPK_EP_GELU) v = (p.flags & ISPK_EP_OUT_BF16) ? gelu_fast(v) : gelu_erf(v);
    if (p.flags & ISPK_EP_SILU) v = silu(v);
    float mk = 1.0f;
    if (p.mask) mk = p.mask[(p.flags & ISPK_EP_MASK_COL) ? j : i] ? 1.0f : 0.0f;
    if (p.flags & ISPK_EP_MASK_ACC) v *= mk;
    int64_t off;
    if (p.cpb > 0) {
        const int bb = j / p.cpb;
        off = (int64_t)bb * p.bstride + (int64_t)i * p.ldc + (j - bb * p.cpb);
    } else {
        off = (int64_t)i * p.ldc + j;
        if (p.resid) {
            const int64_t ro = (int64_t)i * p.ldr + j;
            v += (p.flags & ISPK_EP_RESID_BF16) ? bf16_to_f32(static_cast<const uint16_t*>(p.resid)[ro])
                                                : static_cast<const float*>(p.resid)[ro];
        }
    }
    if (p.flags & ISPK_EP_MASK_OUT) v *= mk;
    if (p.flags & ISPK_EP_OUT_BF16)
        static_cast<uint16_t*>(p.C)[off] = f32_to_bf16(v);
    else
        static_cast<float*>(p.C)[off] = v;
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* As = reinterpret_cast<float*>(smem_raw);  // [2][BM][kLdt]
    float* Bs = As + 2 * BM * kLdt;                  // [2][BN][kLdt]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* A = static_cast<const float*>(p.A);
    const float* W = static_cast<const float*>(p.W);

    const int r0 = tid >> 3, c4 = (tid & 7) * 4;  // staging: 8 lanes cover one 128-B row segment
    float4 ra[BM / 32], rb[BN / 32];
    auto gload = [&](int kt) {
        const int k = kt * 32 + c4;
        const bool kin = k < p.K;
#pragma unroll
        for (int q = 0; q < BM / 32; ++q) {
            const int row = m0 + r0 + 32 * q;
            ra[q] = (kin && row < p.M) ? *reinterpret_cast<const float4*>(A + (int64_t)row * p.lda + k)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < BN / 32; ++q) {
            const int row = n0 + r0 + 32 * q;
            rb[q] = (kin && row < p.N) ? *reinterpret_cast<const float4*>(W + (int64_t)row * p.ldw + k)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int q = 0; q < BM / 32; ++q)
            *reinterpret_cast<float4*>(As + ((buf * BM) + r0 + 32 * q) * kLdt + c4) = ra[q];
#pragma unroll
        for (int q = 0; q < BN / 32; ++q)
            *reinterpret_cast<float4*>(Bs + ((buf * BN) + r0 + 32 * q) * kLdt + c4) = rb[q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nk = (p.K + 31) / 32;
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float* Ab = As + (buf * BM + wm * 32 * TM + l31) * kLdt + h * 4;
        const float* Bb = Bs + (buf * BN + wn * 32 * TN + l31) * kLdt + h * 4;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(Ab + mi * 32 * kLdt + kq * 8);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(Bb + ni * 32 * kLdt + kq * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][s], b[ni][s], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int j = n0 + (wn * TN + ni) * 32 + l31;
            const int ib = m0 + (wm * TM + mi) * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) epilogue_store(p, ib + (r & 3) + 8 * (r >> 2), j, acc[mi][ni][r]);
        }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 path: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Same skeleton; a K step is 64 deep (128-B rows padded to
// 144 B = 36 dwords, so the ds_read_b128 fragment reads stay conflict-free), a lane's A/B fragment is 8 consecutive k
// (lane half h owns k = 16*ks + 8h .. +7: the MFMA's natural operand map, guide §3), 16 MFMAs per wave and K step for
// the 128x128 block.  Epilogue identical (fp32 math), output/residual fp32 or bf16 by flag.
constexpr int kLdtH = 72;  // padded LDS row length in bf16 elements (64 + 8)

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint16_t* As = reinterpret_cast<uint16_t*>(smem_raw);  // [2][BM][kLdtH]
    uint16_t* Bs = As + 2 * BM * kLdtH;                    // [2][BN][kLdtH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const uint16_t* A = static_cast<const uint16_t*>(p.A);
    const uint16_t* W = static_cast<const uint16_t*>(p.W);

    const int r0 = tid >> 3, c8 = (tid & 7) * 8;  // staging: 8 lanes x 16 B cover one 128-B row segment
    uint4 ra[BM / 32], rb[BN / 32];
    auto gload = [&](int kt) {
        const int k = kt * 64 + c8;
        const bool kin = k < p.K;
#pragma unroll
        for (int q = 0; q < BM / 32; ++q) {
            const int row = m0 + r0 + 32 * q;
            ra[q] = (kin && row < p.M) ? *reinterpret_cast<const uint4*>(A + (int64_t)row * p.lda + k)
                                       : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int q = 0; q < BN / 32; ++q) {
            const int row = n0 + r0 + 32 * q;
            rb[q] = (kin && row < p.N) ? *reinterpret_cast<const uint4*>(W + (int64_t)row * p.ldw + k)
                                       : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int q = 0; q < BM / 32; ++q)
            *reinterpret_cast<uint4*>(As + ((buf * BM) + r0 + 32 * q) * kLdtH + c8) = ra[q];
#pragma unroll
        for (int q = 0; q < BN / 32; ++q)
            *reinterpret_cast<uint4*>(Bs + ((buf * BN) + r0 + 32 * q) * kLdtH + c8) = rb[q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nk = (p.K + 63) / 64;
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const uint16_t* Ab = As + (buf * BM + wm * 32 * TM + l31) * kLdtH + h * 8;
        const uint16_t* Bb = Bs + (buf * BN + wn * 32 * TN + l31) * kLdtH + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(Ab + mi * 32 * kLdtH + ks * 16);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(Bb + ni * 32 * kLdtH + ks * 16);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int j = n0 + (wn * TN + ni) * 32 + l31;
            const int ib = m0 + (wm * TM + mi) * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) epilogue_store(p, ib + (r & 3) + 8 * (r >> 2), j, acc[mi][ni][r]);
        }
}

template <int TM, int TN>
int32_t launch_bf16(const GemmParams& p, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr size_t lds = (size_t)2 * (BM + BN) * kLdtH * sizeof(uint16_t);
    static_assert(lds <= 64 * 1024 || true, "");
    ISPK_RESERVE_LDS((&gemm_bf16_kernel<TM, TN>), lds, "gemm");
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN>), grid, dim3(256), lds, s, p);
    return ispk_launch_status();
}

// Epilogue for the transposed-compute kernels: 4 consecutive output features n..n+3 of activation row m.
// Same operation order as epilogue_store; bias/residual/output move as 8- or 16-byte vectors.
__device__ __forceinline__ void epilogue_vec4(const GemmParams& p, int m, int n, float (&v)[4], float mk) {
    if (p.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
    }
    if (p.flags & ISPK_EP_GELU) {  // bf16-operand kernels only: the packed A&S erf (|err| <= 3e-7) is far below their noise
        f32x2 a, b;
        a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3];
        a = gelu_fast2(a);
        b = gelu_fast2(b);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
    if (p.flags & ISPK_EP_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu(v[e]);
    }
    if (p.flags & ISPK_EP_MASK_ACC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= mk;
    }
    if (p.resid) {
        const int64_t ro = (int64_t)m * p.ldr + n;
        if (p.flags & ISPK_EP_RESID_BF16) {
            const uint2 rr = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p.resid) + ro);
            v[0] += bf16_to_f32((uint16_t)(rr.x & 0xffffu)); v[1] += bf16_to_f32((uint16_t)(rr.x >> 16));
            v[2] += bf16_to_f32((uint16_t)(rr.y & 0xffffu)); v[3] += bf16_to_f32((uint16_t)(rr.y >> 16));
        } else {
            const float4 rr = *reinterpret_cast<const float4*>(static_cast<const float*>(p.resid) + ro);
            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
    }
    if (p.flags & ISPK_EP_MASK_OUT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= mk;
    }
    const int64_t co = (int64_t)m * p.ldc + n;
    if (p.flags & ISPK_EP_OUT_BF16) {
        uint2 o;
        o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.C) + co) = o;
    } else {
        *reinterpret_cast<float4*>(static_cast<float*>(p.C) + co) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- row-coalescing epilogue for the transposed-compute kernels.
// In the MFMA C/D fragment a lane owns ONE activation row and scattered groups of 4 output features, so direct stores
// write 16-byte pieces of 32 different rows per instruction — the memory system then handles 8 partial writes per 128-B
// line and the epilogue, not the MFMA loop, bounds the kernel (measured: 0.53 vs 1.0 PFLOP/s incremental).  Here a
// wave passes its tile through a private 32 x 144-B LDS patch (no barrier: LDS operations of one wave complete in order)
// and comes out with lane = (row, 16-byte chunk), so the residual is read and the output written as full 128-B rows.
constexpr int kStageRow = 144;               // bytes per staged row (128 + 16: keeps ds_read_b128 aligned, spreads banks)
constexpr int kStageBytes = 32 * kStageRow;  // per wave

__device__ __forceinline__ void pre_stage(const GemmParams& p, int n, float (&v)[4], float mk) {
    if (p.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
    }
    if (p.flags & ISPK_EP_GELU) {  // bf16-operand kernels only: the packed A&S erf (|err| <= 3e-7) is far below their noise
        f32x2 a, b;
        a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3];
        a = gelu_fast2(a);
        b = gelu_fast2(b);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
    if (p.flags & ISPK_EP_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu(v[e]);
    }
    if (p.flags & ISPK_EP_MASK_ACC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= mk;
    }
}

// fp32 output: one 32-feature tile (features n0 .. n0+31) of the wave's 32 rows (m0 .. m0+31)
__device__ __forceinline__ void store_rows_f32(const GemmParams& p, char* stage, int m0, int n0, const f32x16& acc,
                                               float mk, int lane, float4* keep = nullptr) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e];
        const int n = n0 + 8 * g + 4 * h;
        pre_stage(p, n < p.N ? n : 0, v, mk);
        *reinterpret_cast<float4*>(stage + l31 * kStageRow + (8 * g + 4 * h) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    const int c = lane & 7, n = n0 + 4 * c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), m = m0 + r;
        float4 v = *reinterpret_cast<const float4*>(stage + r * kStageRow + c * 16);
        if (m < p.M && n < p.N) {
            if (p.resid) {
                const int64_t ro = (int64_t)m * p.ldr + n;
                if (p.flags & ISPK_EP_RESID_BF16) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p.resid) + ro);
                    v.x += bf16_to_f32((uint16_t)(rr.x & 0xffffu)); v.y += bf16_to_f32((uint16_t)(rr.x >> 16));
                    v.z += bf16_to_f32((uint16_t)(rr.y & 0xffffu)); v.w += bf16_to_f32((uint16_t)(rr.y >> 16));
                } else {
                    const float4 rr = *reinterpret_cast<const float4*>(static_cast<const float*>(p.resid) + ro);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
            }
            if (p.flags & ISPK_EP_MASK_OUT) {
                const float mo = p.mask[m] ? 1.0f : 0.0f;
                v.x *= mo; v.y *= mo; v.z *= mo; v.w *= mo;
            }
            *reinterpret_cast<float4*>(static_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = v;
        }
        if (keep) keep[i] = v;  // final values in row layout: rows 8i + (lane>>3), features n0 + 4(lane&7) .. +3
    }
}

// bf16 output: two adjacent 32-feature tiles (features n0 .. n0+63); no residual on this path
__device__ __forceinline__ void store_rows_bf16(const GemmParams& p, char* stage, int m0, int n0, const f32x16& acc0,
                                                const f32x16& acc1, float mk, int lane) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = t ? acc1[4 * g + e] : acc0[4 * g + e];
            const int n = n0 + t * 32 + 8 * g + 4 * h;
            pre_stage(p, n < p.N ? n : 0, v, mk);
            if (p.flags & ISPK_EP_MASK_OUT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= mk;
            }
            uint2 o;
            o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
            o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<uint2*>(stage + l31 * kStageRow + (t * 32 + 8 * g + 4 * h) * 2) = o;
        }
    const int c = lane & 7, n = n0 + 8 * c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), m = m0 + r;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + r * kStageRow + c * 16);
        if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p.C) + (int64_t)m * p.ldc + n) = v;
    }
}

// can the row-coalescing epilogue be used?  (else: epilogue_vec4)
inline bool rows_epilogue_ok(const GemmParams& p) {
    if (p.flags & ISPK_EP_OUT_BF16)
        return !p.resid && p.N % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0;
    return true;  // fp32 out: vec_epilogue_ok() already guarantees 16-byte alignment of C / resid rows
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 "wide" GEMM for the long reductions (FFN second Linear: K = 1536 -> N = 384; adaptor: K = 1024 -> N = 256).
// One workgroup = 128 activation rows x ALL N output features, so the big operand (the [rows, K] hidden activations,
// 100 MB per decoder layer) is read exactly once; 8 waves as 4 (rows) x 2 (feature halves), a wave holds 32 rows x
// TN 32-wide feature tiles in accumulators (TN = 6 -> 96 registers).  K advances in 64-deep chunks through
// double-buffered, padded (conflict-free) LDS tiles filled by fully coalesced 128-B row segments; computed transposed
// (D = W_chunk · Xᵀ) for the vector epilogue.  Per chunk a wave issues 4*TN MFMAs for 4*(TN+1) ds_read_b128.
template <int TN, int WM, bool LN = false>  // WM row-waves (32 rows each) x 2 feature-waves (TN 32-wide tiles each)
__global__ __launch_bounds__(WM * 128) void gemm_bf16_wide_kernel(GemmParams p) {
    constexpr int BM = 32 * WM, BN = 64 * TN, LD = 72, NT = WM * 128;  // LD: 64 + 8 bf16 per LDS row
    constexpr int XCH = (BM * 8 + NT - 1) / NT, WCH = (BN * 8 + NT - 1) / NT;  // 16-B chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint16_t* Xs = reinterpret_cast<uint16_t*>(smem_raw);  // [2][BM][LD]
    uint16_t* Ws = Xs + 2 * BM * LD;                       // [2][BN][LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int nb0 = blockIdx.y * BN;  // feature offset of this workgroup (grid.y > 1 only for small M)
    const uint16_t* A = static_cast<const uint16_t*>(p.A);
    const uint16_t* W = static_cast<const uint16_t*>(p.W);

    u32x4 rx[XCH], rw[WCH];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto gload = [&](int kt) {
        const int k = kt * 64;
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = tid + NT * i, r = id >> 3, c = (id & 7) * 8;
            const int row = m0 + r;
            rx[i] = (id < BM * 8 && row < p.M && k + c < p.K)
                        ? *reinterpret_cast<const u32x4*>(A + (int64_t)row * p.lda + k + c) : zero4;
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int id = tid + NT * i, r = id >> 3, c = (id & 7) * 8;
            rw[i] = (id < BN * 8 && nb0 + r < p.N && k + c < p.K)
                        ? *reinterpret_cast<const u32x4*>(W + (int64_t)(nb0 + r) * p.ldw + k + c) : zero4;
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = tid + NT * i, r = id >> 3, c = (id & 7) * 8;
            if (id < BM * 8) *reinterpret_cast<u32x4*>(Xs + (buf * BM + r) * LD + c) = rx[i];
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int id = tid + NT * i, r = id >> 3, c = (id & 7) * 8;
            if (id < BN * 8) *reinterpret_cast<u32x4*>(Ws + (buf * BN + r) * LD + c) = rw[i];
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nk = (p.K + 63) / 64;
    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const uint16_t* xp = Xs + (buf * BM + wm * 32 + l31) * LD + 8 * h;
        const uint16_t* wp = Ws + (buf * BN + wn * 32 * TN + l31) * LD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp + 16 * ks);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wp + t * 32 * LD + 16 * ks);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[t], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }

    const int m = m0 + wm * 32 + l31;
    const float mk = (p.mask && m < p.M) ? (p.mask[m] ? 1.0f : 0.0f) : 1.0f;
    if constexpr (LN) {
        // ---- fused LayerNorm of the finished rows (this workgroup holds ALL N features of its rows: BN == N).
        // Writes C = the GEMM result (fp32 residual stream) AND ln_out = LN(C) * [mask], so the consumer GEMM needs no
        // separate normalisation pass (normalization.py:20-27 + transformer.py:101-102 / :205-206 of the reference).
        // Two-pass statistics in fp32 from registers; a row is spread over 8 lanes x TN tiles x 2 waves.
        char* stage = smem_raw + wave * kStageBytes;
        float* red = reinterpret_cast<float*>(smem_raw + WM * 2 * kStageBytes);  // [2 passes][WM][2 waves][32 rows]
        float4 yv[TN][4];
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            store_rows_f32(p, stage, m0 + wm * 32, (wn * TN + t) * 32, acc[t], mk, lane, yv[t]);
#pragma unroll
            for (int i = 0; i < 4; ++i) rs[i] += (yv[t][i].x + yv[t][i].y) + (yv[t][i].z + yv[t][i].w);
        }
        auto row_total = [&](float (&v)[4], int pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] += __shfl_xor(v[i], 1, 64);
                v[i] += __shfl_xor(v[i], 2, 64);
                v[i] += __shfl_xor(v[i], 4, 64);
                if ((lane & 7) == 0) red[((pass * WM + wm) * 2 + wn) * 32 + 8 * i + (lane >> 3)] = v[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 8 * i + (lane >> 3);
                v[i] = red[((pass * WM + wm) * 2 + 0) * 32 + r] + red[((pass * WM + wm) * 2 + 1) * 32 + r];
            }
        };
        row_total(rs, 0);
        float mean[4], qs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mean[i] = rs[i] * (1.0f / (float)BN);
            qs[i] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = yv[t][i].x - mean[i], b = yv[t][i].y - mean[i], c = yv[t][i].z - mean[i],
                            d = yv[t][i].w - mean[i];
                qs[i] += (a * a + b * b) + (c * c + d * d);
            }
        row_total(qs, 1);
        const int c4 = (lane & 7) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mr = m0 + wm * 32 + 8 * i + (lane >> 3);
            if (mr >= p.M) continue;
            const float rstd = 1.0f / sqrtf(qs[i] * (1.0f / (float)BN) + p.ln_eps);
            const float mo = ((p.ln_flags & 1u) && p.mask) ? (p.mask[mr] ? 1.0f : 0.0f) : 1.0f;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int n = (wn * TN + t) * 32 + c4;
                const float4 g = *reinterpret_cast<const float4*>(p.ln_gamma + n);
                const float4 be = *reinterpret_cast<const float4*>(p.ln_beta + n);
                float4 o;
                o.x = ((yv[t][i].x - mean[i]) * rstd * g.x + be.x) * mo;
                o.y = ((yv[t][i].y - mean[i]) * rstd * g.y + be.y) * mo;
                o.z = ((yv[t][i].z - mean[i]) * rstd * g.z + be.z) * mo;
                o.w = ((yv[t][i].w - mean[i]) * rstd * g.w + be.w) * mo;
                const int64_t off = (int64_t)mr * p.ln_ld + n;
                if (p.ln_flags & 2u) {
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
                    pk.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
                    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.ln_out) + off) = pk;
                } else {
                    *reinterpret_cast<float4*>(static_cast<float*>(p.ln_out) + off) = o;
                }
            }
        }
    } else if (!(p.flags & ISPK_EP_OUT_BF16)) {
        // the K loop is over: its LDS tiles are dead, every wave takes a private patch of them for the row transpose
        char* stage = smem_raw + wave * kStageBytes;
#pragma unroll
        for (int t = 0; t < TN; ++t)
            store_rows_f32(p, stage, m0 + wm * 32, nb0 + (wn * TN + t) * 32, acc[t], mk, lane);
    } else if (m < p.M) {
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb0 + (wn * TN + t) * 32 + 8 * g + 4 * h;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[t][4 * g + e];
                epilogue_vec4(p, m, n, v, mk);
            }
    }
}

template <int TN, int WM, bool LN = false>
int32_t launch_wide(const GemmParams& p, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 64 * TN;
    constexpr size_t lds_tiles = (size_t)2 * (BM + BN) * 72 * sizeof(uint16_t);
    constexpr size_t lds_epi = (size_t)WM * 2 * kStageBytes + (LN ? (size_t)2 * WM * 2 * 32 * sizeof(float) : 0);
    constexpr size_t lds = lds_tiles > lds_epi ? lds_tiles : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    ISPK_RESERVE_LDS((&gemm_bf16_wide_kernel<TN, WM, LN>), lds, "gemm");
    hipLaunchKernelGGL((gemm_bf16_wide_kernel<TN, WM, LN>), dim3((p.M + BM - 1) / BM, (p.N + BN - 1) / BN),
                       dim3(WM * 128), lds, s, p);
    return ispk_launch_status();
}

bool vec_epilogue_ok(const GemmParams& p) {
    const bool out16 = p.flags & ISPK_EP_OUT_BF16, res16 = p.flags & ISPK_EP_RESID_BF16;
    return p.N % 4 == 0 && p.cpb <= 0 && !(p.flags & (ISPK_EP_BIAS_ROW | ISPK_EP_MASK_COL)) && p.ldc % 4 == 0 &&
           ispk_aligned(p.C, out16 ? 8 : 16) && (!p.resid || (p.ldr % 4 == 0 && ispk_aligned(p.resid, res16 ? 8 : 16))) &&
           (!p.bias || ispk_aligned(p.bias, 16));
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 "panel" GEMM for this model's small reduction dims (K = 256 or 384: to_q/to_kv, to_out, FFN1, adaptor stacks).
// The generic tile loop above is latency-bound here: with K = 384 a 128x128 tile has six dependent
// load -> barrier -> compute steps and then a scalar epilogue.  This kernel is built around the shape instead:
//   * a wave owns 32 activation rows and keeps them, for the FULL K, as MFMA operand fragments in registers
//     (K/16 fragments of 8 bf16: 96 VGPRs at K = 384), loaded by ONE burst of K/16 independent 16-B loads per lane —
//     the activation matrix is read exactly once from HBM with all of a wave's loads in flight together;
//   * weights stream through LDS as [64 out-features][K] tiles (50 KB, conflict-free padded rows), the next tile
//     prefetched into registers while the current one feeds the MFMAs, so a step is 2 x K/16 MFMAs per wave with one
//     ds_read_b128 per MFMA;
//   * the tile is computed TRANSPOSED (D = W_tile · Xᵀ): the MFMA C/D fragment then has the activation row on the lane
//     and 4 consecutive output features in consecutive registers, so bias / residual / output are 8- or 16-byte
//     vector accesses (4 store instructions per 32x32 tile instead of 16 scalar ones);
//   * the N range is split over blockIdx.x so that >= 512 workgroups exist (2 per CU).
template <int KC>  // K = 64 * KC
__global__ __launch_bounds__(256, 2) void gemm_bf16_panel_kernel(GemmParams p, int tiles_per_wg, int nsplit, int mblocks) {
    constexpr int K = 64 * KC, LDW = K + 8, KS = K / 16, CPR = K / 8;  // CPR: 16-B chunks per row
    constexpr int HCH = 32 * CPR / 256;                                 // chunks per thread per 32-row half tile (= KC)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint16_t* Ws = reinterpret_cast<uint16_t*>(smem_raw);  // [64][LDW]: weight tile as two 32-row halves

    // XCD-aware block mapping: blocks b and b+8 share an XCD (and its L2).  The nsplit blocks that re-read the same 128
    // activation rows are placed on ONE XCD, back to back, so only the first of them goes to HBM for those rows.
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int ns = jj % nsplit, mb = (jj / nsplit) * 8 + xcd;
    if (mb >= mblocks) return;  // whole workgroup, before any barrier

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int m = mb * 128 + wave * 32 + l31;
    const int ntiles = (p.N + 63) / 64;
    const int nt0 = ns * tiles_per_wg;
    const int nt1 = nt0 + tiles_per_wg < ntiles ? nt0 + tiles_per_wg : ntiles;
    if (nt0 >= nt1) return;
    const uint16_t* A = static_cast<const uint16_t*>(p.A);
    const uint16_t* W = static_cast<const uint16_t*>(p.W);

    // ---- half-tile (32 weight rows) staging: registers <-> LDS.  Separately named register arrays and statically
    // indexed, unconditional loads (row index clamped; rows >= N only feed outputs that are never stored): anything
    // runtime-indexed or predicated here ends up in scratch or behind per-load branches.
    u32x4 wr0[HCH], wr1[HCH];
    const int ntc = nt1 - 1;
    auto wload = [&](u32x4 (&dst)[HCH], int nt, int half) {
        nt = nt < ntc ? nt : ntc;
#pragma unroll
        for (int i = 0; i < HCH; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPR, c = id - r * CPR;
            int n = nt * 64 + half * 32 + r;
            n = n < p.N ? n : p.N - 1;
            dst[i] = *reinterpret_cast<const u32x4*>(W + (int64_t)n * p.ldw + c * 8);
        }
    };
    auto wstore = [&](const u32x4 (&src)[HCH], int half) {
#pragma unroll
        for (int i = 0; i < HCH; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPR, c = id - r * CPR;
            *reinterpret_cast<u32x4*>(Ws + (half * 32 + r) * LDW + c * 8) = src[i];
        }
    };

    // ---- prologue: ONE burst of K/16 independent 16-B loads per lane puts the wave's 32 activation rows, for the whole
    // K, straight into MFMA B-operand fragments (lane = row, half h = k offset 8h).  Four consecutive k-steps share a
    // 128-B line and are issued back to back, so the line is fetched once.  (Staging the panel through LDS instead
    // was tried: the conditional per-wave fragment reads made hipcc spill the staging registers to scratch.)
    bf16x8 xf[KS];
    {
        const int mrow = m < p.M ? m : p.M - 1;
        const uint16_t* xp = A + (int64_t)mrow * p.lda + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xp + 16 * ks);
    }
    wload(wr0, nt0, 0);
    wload(wr1, nt0, 1);
    wstore(wr0, 0);
    wstore(wr1, 1);
    wload(wr0, nt0 + 1, 0);
    wload(wr1, nt0 + 1, 1);
    __syncthreads();

    const float mk = (p.mask && m < p.M) ? (p.mask[m] ? 1.0f : 0.0f) : 1.0f;
    const uint32_t wbase = lds_addr(Ws + l31 * LDW + 8 * h);
    char* stage = smem_raw + (size_t)64 * LDW * sizeof(uint16_t) + wave * kStageBytes;  // wave-private epilogue patch
    const int mw0 = mb * 128 + wave * 32;
    for (int nt = nt0; nt < nt1; ++nt) {
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
        // The 2*KS weight-fragment reads of a tile form ONE stream through a D-deep register ring: the read for step
        // s+D is issued right after step s's MFMA, so D ds_read_b128 stay in flight (left to itself hipcc keeps two and
        // every MFMA waits on LDS latency; it also dissolves a ring written in plain C++, hence the opaque asm reads
        // with hand-counted waits).  Half 0 (steps < KS) is refilled with tile nt+1 after barrier 1 while half 1
        // computes, and vice versa; half 1's reads may be issued before barrier 1 (that half is stable by then).
        constexpr int D = 3, NS = 2 * KS;
        bf16x8 wq[D];
        static_for<0, D>([&](auto ic) {
            constexpr int st = decltype(ic)::value;
            lds_read_b128_asm<((st / KS) * 32 * LDW + 16 * (st % KS)) * 2>(wq[st], wbase);
        });
        static_for<0, NS>([&](auto ic) {
            constexpr int st = decltype(ic)::value;
            if constexpr (st == KS) {
                __syncthreads();  // barrier 1: every wave is done with half 0
                if (nt + 1 < nt1) {
                    wstore(wr0, 0);
                    wload(wr0, nt + 2, 0);
                }
            }
            constexpr int younger = (NS - 1 - st) < (D - 1) ? (NS - 1 - st) : (D - 1);
            lds_wait<younger>();
            __builtin_amdgcn_sched_barrier(0);
            acc[st / KS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[st % D], xf[st % KS], acc[st / KS], 0, 0, 0);
            if constexpr (st + D < NS) {
                constexpr int nx = st + D;
                lds_read_b128_asm<((nx / KS) * 32 * LDW + 16 * (nx % KS)) * 2>(wq[st % D], wbase);
            }
        });
        __syncthreads();  // barrier 2: every wave is done with half 1
        if (nt + 1 < nt1) {
            wstore(wr1, 1);
            wload(wr1, nt + 2, 1);
        }
        // epilogue: register 4g+e of tile half t is output feature n = nt*64 + t*32 + 8g + 4h + e, row m (this lane)
        if (p.cpb == -7) {  // ablation (experiments only): keep the accumulators live, skip the epilogue
            if (acc[0][0] + acc[1][5] == 123.456f) static_cast<float*>(p.C)[0] = 1.f;
        } else if (p.flags & ISPK_EP_OUT_BF16) {
            store_rows_bf16(p, stage, mw0, nt * 64, acc[0], acc[1], mk, lane);
        } else {
            store_rows_f32(p, stage, mw0, nt * 64, acc[0], mk, lane);
            store_rows_f32(p, stage, mw0, nt * 64 + 32, acc[1], mk, lane);
        }
    }
}

template <int KC>
int32_t launch_panel(const GemmParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)64 * (64 * KC + 8) * sizeof(uint16_t) + 4 * kStageBytes;
    const int ntiles = (p.N + 63) / 64, mblocks = (p.M + 127) / 128;
    int nsplit = (512 + mblocks - 1) / mblocks;          // aim for >= 512 workgroups (2 per CU)
    if (const char* e = getenv("ISPK_PANEL_NSPLIT")) nsplit = atoi(e);  // experiments only
    nsplit = nsplit < 1 ? 1 : (nsplit > ntiles ? ntiles : nsplit);
    const int per = (ntiles + nsplit - 1) / nsplit;
    nsplit = (ntiles + per - 1) / per;
    const int mb8 = (mblocks + 7) / 8 * 8;
    ISPK_RESERVE_LDS((&gemm_bf16_panel_kernel<KC>), lds, "gemm");
    hipLaunchKernelGGL((gemm_bf16_panel_kernel<KC>), dim3(mb8 * nsplit), dim3(256), lds, s, p, per, nsplit, mblocks);
    return ispk_launch_status();
}

bool panel_ok(const GemmParams& p) {
    return (p.K == 256 || p.K == 384) && vec_epilogue_ok(p) && rows_epilogue_ok(p) && getenv("ISPK_NO_PANEL") == nullptr;
}

bool wide_ok(const GemmParams& p) {
    return (p.K >= 512 || getenv("ISPK_FORCE_WIDE")) && (p.N == 384 || p.N == 256 || p.N % 192 == 0) && p.M >= 128 * 16 &&
           vec_epilogue_ok(p) &&
           !(p.flags & ISPK_EP_OUT_BF16) && getenv("ISPK_NO_WIDE") == nullptr;
}

template <int TM, int TN>
int32_t launch_f32(const GemmParams& p, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr size_t lds = (size_t)2 * (BM + BN) * kLdt * sizeof(float);
    ISPK_RESERVE_LDS((&gemm_f32_kernel<TM, TN>), lds, "gemm");
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_f32_kernel<TM, TN>), grid, dim3(256), lds, s, p);
    return ispk_launch_status();
}

int32_t check_common(const GemmParams& p, int elt) {
    ISPK_REQUIRE(p.A && p.W && p.C, ISPK_E_NULL, "gemm: null A/W/C");
    ISPK_REQUIRE(p.M >= 0 && p.N >= 1 && p.K >= 1, ISPK_E_SHAPE, "gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    const int vec = 16 / elt;
    ISPK_REQUIRE(p.K % 8 == 0, ISPK_E_SHAPE, "gemm: K=%d must be a multiple of 8", p.K);
    ISPK_REQUIRE(p.lda % vec == 0 && p.ldw % vec == 0 && p.lda >= 1 && p.ldw >= p.K, ISPK_E_ALIGN,
                 "gemm: lda=%lld / ldw=%lld must be multiples of %d (ldw >= K)", (long long)p.lda, (long long)p.ldw, vec);
    ISPK_REQUIRE(ispk_aligned(p.A, 16) && ispk_aligned(p.W, 16), ISPK_E_ALIGN, "gemm: A/W must be 16-byte aligned");
    ISPK_REQUIRE(!((p.flags & (ISPK_EP_MASK_ACC | ISPK_EP_MASK_OUT)) && !p.mask), ISPK_E_NULL,
                 "gemm: mask flag set but mask is NULL");
    ISPK_REQUIRE(!(p.cpb > 0 && p.resid), ISPK_E_UNSUPPORTED, "gemm: resid with a batched (transposed) store");
    ISPK_REQUIRE(p.cpb >= -7 && (p.cpb <= 0 || p.N % p.cpb == 0), ISPK_E_SHAPE, "gemm: N %% cols_per_batch != 0");
    ISPK_REQUIRE((p.flags & ISPK_EP_GELU) == 0 || (p.flags & ISPK_EP_SILU) == 0, ISPK_E_UNSUPPORTED,
                 "gemm: GELU and SILU together");
    return 0;
}

}  // namespace

// tile choice: the largest tile that still gives every one of the 256 CUs a workgroup.  Returns TM*10 + TN.
extern "C" int32_t ispk_gemm_f32_tile(int32_t M, int32_t N, int32_t K) {
    (void)K;
    if (const char* e = getenv("ISPK_GEMM_TILE")) return atoi(e);  // experiments only (tools/bench_kernels.py)
    const int64_t wg128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
    const int64_t wg64x128 = (int64_t)((M + 63) / 64) * ((N + 127) / 128);
    if (wg128 >= 256) return 22;
    if (wg64x128 >= 256 || N > 64) return 12;
    return 11;
}

extern "C" int32_t ispk_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                                 const float* bias, const float* resid, int64_t ldr, const uint8_t* mask, int32_t M,
                                 int32_t N, int32_t K, uint32_t flags, int32_t cols_per_batch, int64_t batch_stride,
                                 ispk_stream_t stream) {
    GemmParams p{A, lda, W, ldw, C, ldc, bias, resid, ldr, mask, M, N, K, flags, cols_per_batch, batch_stride};
    if (int32_t rc = check_common(p, 4)) return rc;
    ISPK_REQUIRE((flags & (ISPK_EP_OUT_BF16 | ISPK_EP_RESID_BF16)) == 0, ISPK_E_UNSUPPORTED,
                 "gemm_f32: bf16 output/residual flags belong to ispk_gemm_bf16");
    if (M == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (ispk_gemm_f32_tile(M, N, K)) {
        case 22: return launch_f32<2, 2>(p, s);
        case 12: return launch_f32<1, 2>(p, s);
    }
    return launch_f32<1, 1>(p, s);
}

// Which kernel instance the last ispk_gemm_bf16 call of THIS thread dispatched (for profilers' labels):
// 1000 + KC -> gemm_bf16_panel_kernel<KC>; 2000 + 10*TN + WM -> gemm_bf16_wide_kernel<TN, WM>;
// 3000 + 10*TM + TN -> gemm_bf16_kernel<TM, TN>; 0 = none yet.
static thread_local int32_t g_last_bf16_variant = 0;
extern "C" int32_t ispk_gemm_bf16_last_variant(void) { return g_last_bf16_variant; }

extern "C" int32_t ispk_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, void* C, int64_t ldc,
                                  const float* bias, const void* resid, int64_t ldr, const uint8_t* mask, int32_t M,
                                  int32_t N, int32_t K, uint32_t flags, int32_t cols_per_batch, int64_t batch_stride,
                                  ispk_stream_t stream) {
    GemmParams p{A, lda, W, ldw, C, ldc, bias, resid, ldr, mask, M, N, K, flags, cols_per_batch, batch_stride};
    if (getenv("ISPK_PANEL_NOEPI")) p.cpb = -7;  // experiments only
    if (int32_t rc = check_common(p, 2)) return rc;
    if (M == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (panel_ok(p) && !(getenv("ISPK_FORCE_WIDE") && (N == 384 || N == 256) && !(flags & ISPK_EP_OUT_BF16))) {
        g_last_bf16_variant = 1000 + K / 64;
        return K == 256 ? launch_panel<4>(p, s) : launch_panel<6>(p, s);
    }
    if (wide_ok(p)) {
        // all N features per workgroup when there are enough rows to fill the chip with 128-row blocks; for the short
        // sequences (encoder: 6,400 rows) 64-row blocks x half the features so that >= 200 workgroups exist
        if (M >= 128 * 160 && (N == 384 || N == 256)) {
            g_last_bf16_variant = 2000 + (N == 384 ? 64 : 44);
            return N == 384 ? launch_wide<6, 4>(p, s) : launch_wide<4, 4>(p, s);
        }
        g_last_bf16_variant = 2000 + (N % 192 == 0 ? 32 : 22);   // 192- or 128-feature column blocks over grid.y
        return N % 192 == 0 ? launch_wide<3, 2>(p, s) : launch_wide<2, 2>(p, s);
    }
    const int tile = ispk_gemm_f32_tile(M, N, K);  // same occupancy rule as the fp32 path
    g_last_bf16_variant = 3000 + tile;
    switch (tile) {
        case 22: return launch_bf16<2, 2>(p, s);
        case 12: return launch_bf16<1, 2>(p, s);
    }
    return launch_bf16<1, 1>(p, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Fused feed-forward block (bf16):  out = mask * ( resid + gelu(x · W1ᵀ + b1) · W2ᵀ + b2 )     feedforward.py:33-40 plus
// the residual add and row mask of transformer.py:105-110 — ONE kernel, the [rows, inner] hidden activations never
// leave the CU (unfused they cost 2 x rows x inner x 2 B of HBM traffic: 200 MB per decoder layer at the benchmark shape,
// more than everything else the layer moves).
//
// A workgroup owns 128 rows (4 waves x 32 rows, one wave per SIMD with the whole 512-register file):
//   xf   : the wave's 32 input rows x D as MFMA fragments, loaded once                            (D/16 x 4 VGPRs)
//   acc2 : the wave's 32 rows x D outputs, transposed (feature on the row axis, row on the lane)   (D/32 x 16 regs)
// and walks the inner dimension in chunks of 32 hidden units.  Per chunk:
//   1. acc1 = W1[chunk] · xfᵀ          (D/16 MFMAs; W1 chunk [32][D] streamed through LDS)
//   2. GELU on the 16 accumulator registers, packed pairwise to bf16 — which IS the B operand of the next product
//      (register 8s+j of lane half h = hidden 16s + 8(j>>2) + 4h + (j&3): accumulator-as-operand, guide §3)
//   3. acc2[nt] += W2[nt-th 32 features][chunk] · Pᵀ   (D/32 x 2 MFMAs; W2 chunk [D][32] in LDS, its 32 hidden columns
//      stored permuted into that same order so that each fragment is ONE conflict-free ds_read_b128)
// Weight chunks are double-buffered in LDS and prefetched one chunk ahead through registers; one barrier per chunk.
// Epilogue: the row-coalescing transpose (store_rows_f32) with residual and mask.
template <int KC, bool B1>  // D = 64 * KC; B1: first Linear has a bias (recipes: no)
__global__ __launch_bounds__(256, 1) void ffn_bf16_kernel(GemmParams p, const uint16_t* __restrict__ W2, int64_t ldw2,
                                                          const float* __restrict__ bias1, int F) {
    constexpr int D = 64 * KC, KS = D / 16, NT = D / 32, HC = 32;
    constexpr int LD1 = D + 8, LD2 = HC + 8;         // padded LDS rows (bf16 elements)
    constexpr int C1 = HC * (D / 8) / 256;            // 16-B chunks per thread: W1 chunk (32 rows x D/8)
    constexpr int C2 = D * (HC / 8) / 256;            //                          W2 chunk (D rows x 4)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint16_t* W1s = reinterpret_cast<uint16_t*>(smem_raw);   // [2][HC][LD1]
    uint16_t* W2s = W1s + 2 * HC * LD1;                      // [2][D][LD2]
    char* stage = smem_raw + (size_t)(2 * HC * LD1 + 2 * D * LD2) * 2 + (threadIdx.x >> 6) * kStageBytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int mw0 = blockIdx.x * 128 + wave * 32;
    const int m = mw0 + l31;
    const uint16_t* X = static_cast<const uint16_t*>(p.A);
    const uint16_t* W1 = static_cast<const uint16_t*>(p.W);
    const int nchunks = F / HC;

    // per-thread element offsets of its staging pieces inside chunk 0 (32-bit, computed once); a chunk then only adds a
    // wave-uniform step, so the loads are "uniform base + 32-bit lane offset" with no 64-bit arithmetic in the loop
    u32x4 r1[C1], r2[C2];
    int o1[C1], o2[C2];
#pragma unroll
    for (int i = 0; i < C1; ++i) {
        const int id = tid + 256 * i, r = id / (D / 8), cc = id - r * (D / 8);
        o1[i] = r * (int)p.ldw + cc * 8;
    }
#pragma unroll
    for (int i = 0; i < C2; ++i) {
        const int id = tid + 256 * i, n = id >> 2, cc = id & 3;
        o2[i] = n * (int)ldw2 + cc * 8;
    }
    const int step1 = HC * (int)p.ldw;
    auto load_chunk = [&](int c) {
        c = c < nchunks ? c : nchunks - 1;
        const uint16_t* b1 = W1 + (int64_t)c * step1;
        const uint16_t* b2 = W2 + c * HC;
#pragma unroll
        for (int i = 0; i < C1; ++i) r1[i] = *reinterpret_cast<const u32x4*>(b1 + o1[i]);
#pragma unroll
        for (int i = 0; i < C2; ++i) r2[i] = *reinterpret_cast<const u32x4*>(b2 + o2[i]);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const int id = tid + 256 * i, r = id / (D / 8), cc = id - r * (D / 8);
            *reinterpret_cast<u32x4*>(W1s + (buf * HC + r) * LD1 + cc * 8) = r1[i];
        }
        // W2 chunk rows are stored PERMUTED in the hidden order of the accumulator fragment (LDS position 16s + 8h + j
        // holds hidden 16s + 8(j>>2) + 4h + (j&3)), so a lane's k-step fragment is one aligned 16-byte run: the global
        // 16-byte piece cc (hidden 8cc .. 8cc+7; s = cc>>1, a = cc&1) lands as two 8-byte halves at 16s + 4a (h = 0)
        // and 16s + 8 + 4a (h = 1).
#pragma unroll
        for (int i = 0; i < C2; ++i) {
            const int id = tid + 256 * i, n = id >> 2, cc = id & 3;
            uint16_t* row = W2s + (buf * D + n) * LD2 + 16 * (cc >> 1) + 4 * (cc & 1);
            uint2 lo, hi;
            lo.x = r2[i][0]; lo.y = r2[i][1]; hi.x = r2[i][2]; hi.y = r2[i][3];
            *reinterpret_cast<uint2*>(row) = lo;
            *reinterpret_cast<uint2*>(row + 8) = hi;
        }
    };

    load_chunk(0);
    bf16x8 xf[KS];
    {
        const int mrow = m < p.M ? m : p.M - 1;
        const uint16_t* xp = X + (int64_t)mrow * p.lda + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xp + 16 * ks);
    }
    store_chunk(0);
    load_chunk(1);
    f32x16 acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    __syncthreads();

    // One wave per SIMD: nothing hides an LDS round trip, so the KS + 2*NT operand reads of a chunk run as ONE stream
    // through an RD-deep register ring of opaque asm reads with hand-counted waits (see gemm_bf16_panel_kernel); the
    // W2 reads are issued while the GELU of the chunk is still running.
    constexpr int RD = 4, NS = KS + 2 * NT;
    const uint32_t w1base = lds_addr(W1s + l31 * LD1 + 8 * h);
    const uint32_t w2base = lds_addr(W2s + l31 * LD2 + 8 * h);
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const uint32_t a1 = w1base + buf * (HC * LD1 * 2);
        const uint32_t a2 = w2base + buf * (D * LD2 * 2);
        bf16x8 q[RD];
        auto issue = [&](auto ic) {
            constexpr int st = decltype(ic)::value;
            if constexpr (st < KS) {
                lds_read_b128_asm_acc<st * 32>(q[st % RD], a1);
            } else {
                constexpr int nt = (st - KS) / 2, s2 = (st - KS) % 2;
                lds_read_b128_asm_acc<(nt * 32 * LD2 + 16 * s2) * 2>(q[st % RD], a2);
            }
        };
        static_for<0, RD>(issue);
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
        union { uint32_t u[4]; bf16x8 f; } pf[2];
        static_for<0, NS>([&](auto ic) {
            constexpr int st = decltype(ic)::value;
            if constexpr (st == KS) {
                // bias, GELU, pack: accumulator registers 8s .. 8s+7 become the B fragment of k-step s
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f32x2 v;
                        v.x = acc1[8 * s + 2 * e];
                        v.y = acc1[8 * s + 2 * e + 1];
                        if constexpr (B1) {
                            const int hid = c * HC + ((2 * e) & 3) + 8 * ((8 * s + 2 * e) >> 2) + 4 * h;
                            v.x += bias1[hid];
                            v.y += bias1[hid + 1];
                        }
                        v = gelu_fast2(v);
                        pf[s].u[e] = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
                    }
            }
            constexpr int younger = (NS - 1 - st) < (RD - 1) ? (NS - 1 - st) : (RD - 1);
            lds_wait<younger>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (st < KS) {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[st % RD], xf[st], acc1, 0, 0, 0);
            } else {
                constexpr int nt = (st - KS) / 2, s2 = (st - KS) % 2;
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[st % RD], pf[s2].f, acc2[nt], 0, 0, 0);
            }
            if constexpr (st + RD < NS) issue(std::integral_constant<int, st + RD>{});
        });
        // ---- stage chunk c+1 (already in registers) into the other buffer, fetch chunk c+2
        if (c + 1 < nchunks) {
            store_chunk(buf ^ 1);
            load_chunk(c + 2);
        }
        __syncthreads();
    }

    const float mk = (p.mask && m < p.M) ? (p.mask[m] ? 1.0f : 0.0f) : 1.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) store_rows_f32(p, stage, mw0, nt * 32, acc2[nt], mk, lane);
}

extern "C" int32_t ispk_ffn_bf16(const uint16_t* x, int64_t ldx, const uint16_t* W1, int64_t ldw1, const float* bias1,
                                 const uint16_t* W2, int64_t ldw2, const float* bias2, const float* resid, int64_t ldr,
                                 const uint8_t* mask, float* out, int64_t ldo, int32_t rows, int32_t D, int32_t F,
                                 uint32_t flags, ispk_stream_t stream) {
    ISPK_REQUIRE(x && W1 && W2 && out, ISPK_E_NULL, "ffn: null pointer");
    ISPK_REQUIRE(D == 384 || D == 256, ISPK_E_UNSUPPORTED, "ffn: dim %d (built for 256 / 384)", D);
    ISPK_REQUIRE(rows >= 0 && F >= 64 && F % 32 == 0, ISPK_E_SHAPE, "ffn: bad shape rows=%d inner=%d", rows, F);
    ISPK_REQUIRE((flags & ~(ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) == 0, ISPK_E_UNSUPPORTED, "ffn: unsupported flags");
    ISPK_REQUIRE(!((flags & (ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) && !mask), ISPK_E_NULL, "ffn: mask flag without mask");
    ISPK_REQUIRE(ldx % 8 == 0 && ldw1 % 8 == 0 && ldw2 % 8 == 0 && ldo % 4 == 0 && (!resid || ldr % 4 == 0) && ldx >= D &&
                     ldw1 >= D && ldw2 >= F && ldo >= D,
                 ISPK_E_ALIGN, "ffn: leading strides must be multiples of 8 (bf16) / 4 (fp32)");
    ISPK_REQUIRE(ispk_aligned(x, 16) && ispk_aligned(W1, 16) && ispk_aligned(W2, 16) && ispk_aligned(out, 16) &&
                     (!resid || ispk_aligned(resid, 16)) && (!bias2 || ispk_aligned(bias2, 16)),
                 ISPK_E_ALIGN, "ffn: pointers must be 16-byte aligned");
    if (rows == 0) return 0;
    GemmParams p{x, ldx, W1, ldw1, out, ldo, bias2, resid, ldr, mask, rows, D, D, flags & ~ISPK_EP_GELU, 0, 0};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((rows + 127) / 128);
    ISPK_REQUIRE((int64_t)F * ldw1 < (1ll << 30) && (int64_t)D * ldw2 < (1ll << 30), ISPK_E_SHAPE, "ffn: weights too large");
#define ISPK_FFN_LAUNCH(KC_, B1_)                                                                         \
    do {                                                                                                  \
        constexpr size_t lds = (size_t)(2 * 32 * (64 * KC_ + 8) + 2 * 64 * KC_ * 40) * 2 + 4 * kStageBytes; \
        ISPK_RESERVE_LDS((&ffn_bf16_kernel<KC_, B1_>), lds, "ffn");                                       \
        hipLaunchKernelGGL((ffn_bf16_kernel<KC_, B1_>), grid, dim3(256), lds, s, p, W2, ldw2, bias1, F);   \
    } while (0)
    if (D == 384) {
        if (bias1) ISPK_FFN_LAUNCH(6, true); else ISPK_FFN_LAUNCH(6, false);
    } else {
        if (bias1) ISPK_FFN_LAUNCH(4, true); else ISPK_FFN_LAUNCH(4, false);
    }
#undef ISPK_FFN_LAUNCH
    return ispk_launch_status();
}

extern "C" int32_t ispk_gemm_bf16_ln(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, float* C, int64_t ldc,
                                     const float* bias, const void* resid, int64_t ldr, const uint8_t* mask, int32_t M,
                                     int32_t N, int32_t K, uint32_t flags, const float* ln_gamma, const float* ln_beta,
                                     float ln_eps, void* ln_out, int64_t ln_ld, uint32_t ln_flags, ispk_stream_t stream) {
    GemmParams p{A, lda, W, ldw, C, ldc, bias, resid, ldr, mask, M, N, K, flags, 0, 0};
    if (int32_t rc = check_common(p, 2)) return rc;
    ISPK_REQUIRE(ln_gamma && ln_beta && ln_out, ISPK_E_NULL, "gemm_ln: null LayerNorm argument");
    ISPK_REQUIRE(N == 384 || N == 256, ISPK_E_UNSUPPORTED, "gemm_ln: N=%d (a workgroup must hold whole rows: 256 or 384)", N);
    ISPK_REQUIRE(!(flags & (ISPK_EP_OUT_BF16 | ISPK_EP_BIAS_ROW | ISPK_EP_MASK_COL)) && vec_epilogue_ok(p), ISPK_E_UNSUPPORTED,
                 "gemm_ln: needs an fp32 row-major output with 16-byte aligned rows");
    ISPK_REQUIRE(ln_ld % 4 == 0 && ispk_aligned(ln_out, (ln_flags & 2u) ? 8 : 16) && ispk_aligned(ln_gamma, 16) &&
                     ispk_aligned(ln_beta, 16), ISPK_E_ALIGN, "gemm_ln: LayerNorm buffers must be 16-byte aligned");
    ISPK_REQUIRE(!((ln_flags & 1u) && !mask), ISPK_E_NULL, "gemm_ln: ln mask flag set but mask is NULL");
    if (M == 0) return 0;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_out = ln_out; p.ln_ld = ln_ld; p.ln_eps = ln_eps; p.ln_flags = ln_flags;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (M >= 128 * 160) return N == 384 ? launch_wide<6, 4, true>(p, s) : launch_wide<4, 4, true>(p, s);
    return N == 384 ? launch_wide<6, 2, true>(p, s) : launch_wide<4, 2, true>(p, s);
}
