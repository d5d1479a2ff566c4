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
};

constexpr int kLdt = 36;  // padded LDS row length in dwords (32 + 4)

__device__ __forceinline__ void epilogue_store(const GemmParams& p, int i, int j, float v) {
    if (i >= p.M || j >= p.N) return;
    if (p.bias) v += p.bias[(p.flags & ISPK_EP_BIAS_ROW) ? i : j];
    if (p.flags & ISPK_EP_GELU) v = gelu_erf(v);
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
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) ISPK_FAIL((int32_t)e, "gemm: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    }
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_bf16_kernel<TM, TN>), grid, dim3(256), lds, s, p);
    return ispk_launch_status();
}

template <int TM, int TN>
int32_t launch_f32(const GemmParams& p, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr size_t lds = (size_t)2 * (BM + BN) * kLdt * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) ISPK_FAIL((int32_t)e, "gemm: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    }
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_f32_kernel<TM, TN>), grid, dim3(256), lds, s, p);
    return ispk_launch_status();
}

int32_t check_common(const GemmParams& p, int elt) {
    ISPK_REQUIRE(p.A && p.W && p.C, ISPK_E_NULL, "gemm: null A/W/C");
    ISPK_REQUIRE(p.M >= 0 && p.N >= 1 && p.K >= 1, ISPK_E_SHAPE, "gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    const int vec = 16 / elt;
    ISPK_REQUIRE(p.K % 8 == 0, ISPK_E_SHAPE, "gemm: K=%d must be a multiple of 8", p.K);
    ISPK_REQUIRE(p.lda % vec == 0 && p.ldw % vec == 0 && p.lda >= p.K && p.ldw >= p.K, ISPK_E_ALIGN,
                 "gemm: lda=%lld / ldw=%lld must be >= K and multiples of %d", (long long)p.lda, (long long)p.ldw, vec);
    ISPK_REQUIRE(ispk_aligned(p.A, 16) && ispk_aligned(p.W, 16), ISPK_E_ALIGN, "gemm: A/W must be 16-byte aligned");
    ISPK_REQUIRE(!((p.flags & (ISPK_EP_MASK_ACC | ISPK_EP_MASK_OUT)) && !p.mask), ISPK_E_NULL,
                 "gemm: mask flag set but mask is NULL");
    ISPK_REQUIRE(!(p.cpb > 0 && p.resid), ISPK_E_UNSUPPORTED, "gemm: resid with a batched (transposed) store");
    ISPK_REQUIRE(p.cpb >= 0 && (p.cpb == 0 || p.N % p.cpb == 0), ISPK_E_SHAPE, "gemm: N %% cols_per_batch != 0");
    ISPK_REQUIRE((p.flags & ISPK_EP_GELU) == 0 || (p.flags & ISPK_EP_SILU) == 0, ISPK_E_UNSUPPORTED,
                 "gemm: GELU and SILU together");
    return 0;
}

}  // namespace

// tile choice: the largest tile that still gives every one of the 256 CUs a workgroup.  Returns TM*10 + TN.
extern "C" int32_t ispk_gemm_f32_tile(int32_t M, int32_t N, int32_t K) {
    (void)K;
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

extern "C" int32_t ispk_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, void* C, int64_t ldc,
                                  const float* bias, const void* resid, int64_t ldr, const uint8_t* mask, int32_t M,
                                  int32_t N, int32_t K, uint32_t flags, int32_t cols_per_batch, int64_t batch_stride,
                                  ispk_stream_t stream) {
    GemmParams p{A, lda, W, ldw, C, ldc, bias, resid, ldr, mask, M, N, K, flags, cols_per_batch, batch_stride};
    if (int32_t rc = check_common(p, 2)) return rc;
    if (M == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (ispk_gemm_f32_tile(M, N, K)) {  // same occupancy rule as the fp32 path
        case 22: return launch_bf16<2, 2>(p, s);
        case 12: return launch_bf16<1, 2>(p, s);
    }
    return launch_bf16<1, 1>(p, s);
}
