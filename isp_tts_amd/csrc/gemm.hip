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

#include "gemm_common.h"

namespace {


constexpr int kLdt = 36;  // padded LDS row length in dwords (32 + 4)


template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams pin) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    GemmParams p = pin;
    if (gridDim.z > 1) {          // batched: one product per blockIdx.z
        p.A = static_cast<const float*>(pin.A) + (int64_t)blockIdx.z * pin.za;
        p.W = static_cast<const float*>(pin.W) + (int64_t)blockIdx.z * pin.zw;
        p.C = static_cast<float*>(pin.C) + (int64_t)blockIdx.z * pin.zc;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* As = reinterpret_cast<float*>(smem_raw);  // [2][BM][kLdt]
    float* Bs = As + 2 * BM * kLdt;                  // [2][BN][kLdt]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // Interior tiles with a plain row-major fp32 output (every hot call of the fp32 path): no per-element bounds / layout
    // branches, the 16 residual values and row masks of a tile fetched together BEFORE the arithmetic.  (The generic
    // per-element path below issues its residual load, waits for it, stores, 64 times per lane: it took the
    // out-projection to 51 TFLOP/s against 86 for the same GEMM without a residual.)
    const bool plain = m0 + BM <= p.M && n0 + BN <= p.N && p.cpb <= 0 &&
                       !(p.flags & (ISPK_EP_BIAS_ROW | ISPK_EP_MASK_COL | ISPK_EP_OUT_BF16 | ISPK_EP_RESID_BF16));
    if (plain) {
        const bool gelu = p.flags & ISPK_EP_GELU, silu_f = p.flags & ISPK_EP_SILU, mask_acc = p.flags & ISPK_EP_MASK_ACC,
                   mask_out = p.flags & ISPK_EP_MASK_OUT;
        const float* resid = static_cast<const float*>(p.resid);
        float* C = static_cast<float*>(p.C);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int ib = m0 + (wm * TM + mi) * 32 + 4 * h;
            float mk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) mk[r] = (p.mask && (mask_acc || mask_out)) ? (p.mask[ib + (r & 3) + 8 * (r >> 2)] ? 1.0f : 0.0f) : 1.0f;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int j = n0 + (wn * TN + ni) * 32 + l31;
                const float bj = p.bias ? p.bias[j] : 0.0f;
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = resid ? resid[(int64_t)(ib + (r & 3) + 8 * (r >> 2)) * p.ldr + j] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[mi][ni][r] + bj;
                    if (gelu) v = gelu_as28(v);
                    if (silu_f) v = silu(v);
                    if (mask_acc) v *= mk[r];
                    v += rv[r];
                    if (mask_out) v *= mk[r];
                    C[(int64_t)(ib + (r & 3) + 8 * (r >> 2)) * p.ldc + j] = v;
                }
            }
        }
        return;
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

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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


// ---------------------------------------------------------------------------------------------------------------
// bf16 "wide" GEMM for the long reductions (FFN second Linear: K = 1536 -> N = 384; adaptor: K = 1024 -> N = 256).
// One workgroup = 128 activation rows x ALL N output features, so the big operand (the [rows, K] hidden activations,
// 100 MB per decoder layer) is read exactly once; 8 waves as 4 (rows) x 2 (feature halves), a wave holds 32 rows x
// TN 32-wide feature tiles in accumulators (TN = 6 -> 96 registers).  K advances in 64-deep chunks through
// double-buffered, padded (conflict-free) LDS tiles filled by fully coalesced 128-B row segments; computed transposed
// (D = W_chunk · Xᵀ) for the vector epilogue.  Per chunk a wave issues 4*TN MFMAs for 4*(TN+1) ds_read_b128.

// Operand staging: the K loop of this kernel is a chain of short steps (4*TN MFMAs per wave), and with register staging
// one step ahead every step waited out a global-load round trip (the encoder's 6,400-row FFN2 ran at 1.7 us per
// 64-deep chunk).  Both operand tiles now stream by LDS-DMA (global_load_lds_dwordx4) into a ring of S slots with up to
// S-1 chunks in flight and no staging registers.  A DMA instruction writes 8 rows x 128 B linearly, so rows are
// unpadded and the bank spread comes from an XOR swizzle applied to the SOURCE address: LDS (row r, 16-byte slot pc)
// holds logical chunk pc ^ ((r >> 1) & 7), which makes the 16 rows of a ds_read_b128 lane group hit 16 distinct
// (bank half, slot) pairs.  Each wave issues IPL = (BM + BN) / 8 / waves DMAs per chunk and waits for its own with a
// counted vmcnt; one raw s_barrier per chunk publishes the chunk and retires the slot the next DMA overwrites.
template <int TN, int WM, bool LN = false>  // WM row-waves (32 rows each) x 2 feature-waves (TN 32-wide tiles each)
__global__ __launch_bounds__(WM * 128) void gemm_bf16_wide_kernel(GemmParams pin) {
    constexpr int BM = 32 * WM, BN = 64 * TN, NT = WM * 128, NWV = NT / 64;
    constexpr int kSlot = (BM + BN) * 128;                       // bytes per ring slot: X tile then W tile, 128-B rows
    constexpr int S = 4 * kSlot <= 128 * 1024 ? 4 : (3 * kSlot <= 144 * 1024 ? 3 : 2);
    constexpr int IPL = (BM + BN) / 8 / NWV;
    static_assert((BM + BN) / 8 % NWV == 0 && (S - 1) * IPL <= 63, "DMA split / vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int nb0 = blockIdx.y * BN;  // feature offset of this workgroup (grid.y > 1 only for small M)
    // split-K (ispk_gemm_bf16_splitk: few rows, long K): workgroup z reduces K slice z - both operands advance za = zw = p.K
    // elements, the raw fp32 partial product goes to slab z of the workspace (zc elements apart); no epilogue flags then
    const GemmParams p = [&]() { GemmParams q = pin; q.C = static_cast<float*>(pin.C) + (int64_t)blockIdx.z * pin.zc; return q; }();
    const uint16_t* A = static_cast<const uint16_t*>(p.A) + (int64_t)blockIdx.z * p.za;
    const uint16_t* W = static_cast<const uint16_t*>(p.W) + (int64_t)blockIdx.z * p.zw;

    // this lane's part of DMA instruction j: row (8-row group wave*IPL + j, row lane>>3), LDS slot lane&7
    const uint16_t* src_row[IPL];
    int src_chunk[IPL];
#pragma unroll
    for (int j = 0; j < IPL; ++j) {
        const int r = (wave * IPL + j) * 8 + (lane >> 3);       // row of the concatenated [X; W] tile
        const int rr = r < BM ? r : r - BM;
        src_chunk[j] = ((lane & 7) ^ ((rr >> 1) & 7)) * 8;      // logical k offset (elements) inside the chunk
        if (r < BM) {
            const int row = m0 + r < p.M ? m0 + r : p.M - 1;     // rows past the end: a valid row, never stored
            src_row[j] = A + (int64_t)row * p.lda;
        } else {
            const int n = nb0 + rr < p.N ? nb0 + rr : p.N - 1;
            src_row[j] = W + (int64_t)n * p.ldw;
        }
    }
    auto issue = [&](int kt) {
        char* slot = smem_raw + (kt % S) * kSlot + wave * (IPL * 1024);
#pragma unroll
        for (int j = 0; j < IPL; ++j) {
            const int k = kt * 64 + src_chunk[j];
            const uint16_t* src = k < p.K ? src_row[j] + k : g_zero16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(slot + j * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nk = (p.K + 63) / 64;
#pragma unroll 1
    for (int kt = 0; kt < nk && kt < S - 1; ++kt) issue(kt);

    // fragment byte offsets inside a slot: k-step ks, logical chunk 2ks + h of row l31 (tile bases are multiples of 16 rows)
    uint32_t xoff[4], woff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint32_t sw = (uint32_t)(((2 * ks + h) ^ ((l31 >> 1) & 7)) << 4);
        xoff[ks] = (wm * 32 + l31) * 128 + sw;
        woff[ks] = (BM + wn * 32 * TN + l31) * 128 + sw;
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int after = (nk - 1 - kt) < (S - 2) ? (nk - 1 - kt) : (S - 2);   // younger chunks this wave has in flight
        if (S >= 4 && after >= 2) vm_wait<2 * IPL>();
        else if (S >= 3 && after == 1) vm_wait<IPL>();
        else vm_wait<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // publishes chunk kt; every wave is done with chunk kt-1
        asm volatile("" ::: "memory");
        // Operand fragments of k-step ks: 1 + TN ds_read_b128, requested two k-steps ahead through a double register
        // set of opaque asm reads with counted waits (hipcc sinks every plain read next to its MFMA and waits there;
        // with one or two waves per SIMD nothing else hides that round trip - 16 of them made up most of a chunk).
        const uint32_t sl = lds_addr(smem_raw) + (uint32_t)((kt % S) * kSlot);
        bf16x8 fr[2][TN + 1];
        auto rd = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            lds_read_b128_asm<0>(fr[g & 1][0], sl + xoff[g]);
            static_for<0, TN>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                lds_read_b128_asm<t * 32 * 128>(fr[g & 1][1 + t], sl + woff[g]);
            });
        };
        rd(std::integral_constant<int, 0>{});
        rd(std::integral_constant<int, 1>{});
        if (kt + S - 1 < nk) issue(kt + S - 1);  // into the slot chunk kt-1 just left
        static_for<0, 4>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            if constexpr (ks < 3) lds_wait<TN + 1>(); else lds_wait<0>();   // group ks is in; group ks+1 may be in flight
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks & 1][1 + t], fr[ks & 1][0], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ks + 2 < 4) rd(std::integral_constant<int, ks + 2>{});
        });
    }
    __syncthreads();   // the epilogue's transposition patches alias the ring

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
        float mo4[4];
        mask_rows(p, m0 + wm * 32, lane, mo4);
        float4 rres[TN][4];
#pragma unroll
        for (int t = 0; t < TN; ++t) resid_prefetch(p, m0 + wm * 32, nb0 + (wn * TN + t) * 32, lane, rres[t]);
#pragma unroll
        for (int t = 0; t < TN; ++t)
            store_rows_f32(p, stage, m0 + wm * 32, nb0 + (wn * TN + t) * 32, acc[t], mk, lane, nullptr, rres[t], mo4);
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
    constexpr size_t slot = (size_t)(BM + BN) * 128;
    constexpr size_t lds_tiles = (4 * slot <= 128 * 1024 ? 4 : (3 * slot <= 144 * 1024 ? 3 : 2)) * slot;
    constexpr size_t lds_epi = (size_t)WM * 2 * kStageBytes + (LN ? (size_t)2 * WM * 2 * 32 * sizeof(float) : 0);
    constexpr size_t lds = lds_tiles > lds_epi ? lds_tiles : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    ISPK_RESERVE_LDS((&gemm_bf16_wide_kernel<TN, WM, LN>), lds, "gemm");
    hipLaunchKernelGGL((gemm_bf16_wide_kernel<TN, WM, LN>), dim3((p.M + BM - 1) / BM, (p.N + BN - 1) / BN),
                       dim3(WM * 128), lds, s, p);
    return ispk_launch_status();
}

// ---- split-K for the row-block kernel at FEW rows (a rank's share under strong scaling: 8 - 16 utterances).  With a handful of
// row blocks the K loop of one workgroup IS the kernel - 24 dependent 64-deep steps at ~0.8 us each for 800 x 384 x 1536 - while
// nine CUs in ten idle.  ksplit workgroups per tile reduce one K slice each into fp32 slabs; gemm_splitk_combine_kernel adds the
// slabs in slice order (deterministic) and applies the epilogue (bias, activation, masks, residual, output type).
__global__ __launch_bounds__(256) void gemm_splitk_combine_kernel(GemmParams p, const float* __restrict__ parts, int ksplit) {
    const int n4 = p.N / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (int64_t)m * n4) * 4;
    const int64_t slab = (int64_t)p.M * p.N;
    const float4 a = *reinterpret_cast<const float4*>(parts + (int64_t)m * p.N + n);
    float v[4] = {a.x, a.y, a.z, a.w};
    for (int z = 1; z < ksplit; ++z) {
        const float4 b = *reinterpret_cast<const float4*>(parts + z * slab + (int64_t)m * p.N + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    const float mk = p.mask ? (p.mask[m] ? 1.0f : 0.0f) : 1.0f;
    epilogue_vec4(p, m, n, v, mk);
}

template <int TN, int WM>
int32_t launch_wide_splitk(const GemmParams& p, hipStream_t s, int ksplit, float* workspace) {
    constexpr int BM = 32 * WM, BN = 64 * TN;
    constexpr size_t slot = (size_t)(BM + BN) * 128;
    constexpr size_t lds_tiles = (4 * slot <= 128 * 1024 ? 4 : (3 * slot <= 144 * 1024 ? 3 : 2)) * slot;
    constexpr size_t lds_epi = (size_t)WM * 2 * kStageBytes;
    constexpr size_t lds = lds_tiles > lds_epi ? lds_tiles : lds_epi;
    GemmParams q = p;
    q.K = p.K / ksplit;
    q.za = q.zw = q.K;
    q.C = workspace; q.ldc = p.N; q.zc = (int64_t)p.M * p.N;
    q.bias = nullptr; q.resid = nullptr; q.mask = nullptr; q.flags = 0; q.ldr = 0;
    ISPK_RESERVE_LDS((&gemm_bf16_wide_kernel<TN, WM, false>), lds, "gemm");
    hipLaunchKernelGGL((gemm_bf16_wide_kernel<TN, WM, false>), dim3((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, ksplit),
                       dim3(WM * 128), lds, s, q);
    if (int32_t rc = ispk_launch_status()) return rc;
    const int64_t total = (int64_t)p.M * (p.N / 4);
    hipLaunchKernelGGL(gemm_splitk_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, workspace, ksplit);
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
//     (K/16 fragments of 8 bf16: 96 VGPRs at K = 384), fetched with coalesced 16-byte loads and re-shaped through a
//     wave-private LDS patch (see the prologue) — the activation matrix is read once per N-split from HBM / L2;
//   * weights stream through LDS as [64 out-features][K] tiles (50 KB, conflict-free padded rows) in two 32-row halves,
//     the next half prefetched into registers ONE load per second MFMA gap while the current one feeds the MFMAs, so a
//     step is 2 x K/16 MFMAs per wave with one ds_read_b128 per MFMA;
//   * the tile is computed TRANSPOSED (D = W_tile · Xᵀ): the MFMA C/D fragment then has the activation row on the lane
//     and 4 consecutive output features in consecutive registers, so bias / residual / output are 8- or 16-byte
//     vector accesses (4 store instructions per 32x32 tile instead of 16 scalar ones);
//   * the N range is split over blockIdx.x so that >= 512 workgroups exist (2 per CU).
// LNP: fp32 A + LayerNorm in the prologue - 1: the rows' (mean, rstd) are given, 2: the wave computes them itself
template <int KC, int EP = kEpDyn, bool ST = false, int LNP = 0>  // K = 64 * KC; EP: compile-time epilogue; ST: stamps
__global__ __launch_bounds__(256, 2) void gemm_bf16_panel_kernel(GemmParams p, int tiles_per_wg, int nsplit, int mblocks) {
    [[maybe_unused]] uint64_t tsum[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, tA = 0, tB = 0, tC = 0, tD = 0;
    if constexpr (ST) t0 = __builtin_readcyclecounter();
    constexpr int K = 64 * KC, LDW = K + 8, KS = K / 16, CPR = K / 8;  // CPR: 16-B chunks per row
    constexpr int HCH = 32 * CPR / 256;                                 // chunks per thread per 32-row half tile (= KC)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint16_t* Ws = reinterpret_cast<uint16_t*>(smem_raw);  // [64][LDW]: weight tile as two 32-row halves

    // XCD-aware block mapping: blocks b and b+8 share an XCD (and its L2).  The nsplit blocks that re-read the same 128
    // activation rows are placed on ONE XCD, back to back, so only the first of them goes to HBM for those rows.
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int ns = jj % nsplit, mb = (jj / nsplit) * 8 + xcd;
    if (mb >= mblocks) return;  // whole workgroup, before any barrier

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    // piece i of a half tile: weight row 32*half + (tid + 256 i) / CPR, 16-byte column chunk (tid + 256 i) % CPR
    auto wload1 = [&](u32x4& dst, int i, int nt, int half) {
        nt = nt < ntc ? nt : ntc;
        const int id = tid + 256 * i;
        const int r = id / CPR, c = id - r * CPR;
        int n = nt * 64 + half * 32 + r;
        n = n < p.N ? n : p.N - 1;
        dst = *reinterpret_cast<const u32x4*>(W + (int64_t)n * p.ldw + c * 8);
    };
    auto wload = [&](u32x4 (&dst)[HCH], int nt, int half) {
#pragma unroll
        for (int i = 0; i < HCH; ++i) wload1(dst[i], i, nt, half);
    };
    auto wstore = [&](const u32x4 (&src)[HCH], int half) {
#pragma unroll
        for (int i = 0; i < HCH; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPR, c = id - r * CPR;
            *reinterpret_cast<u32x4*>(Ws + (half * 32 + r) * LDW + c * 8) = src[i];
        }
    };

    // ---- prologue.  The first weight tile's loads go out first (they are needed first and fly during the rest).  Then
    // the wave's 32 activation rows x K - one contiguous 32*K*2-byte block when lda == K - are fetched with fully
    // coalesced 16-byte loads (64 lanes = 1 KB = 8 whole lines per instruction) and turned into MFMA B-operand
    // fragments (lane = row, half h = k offset 8h) through a wave-private LDS patch, one K-half at a time: patch rows
    // are padded by 16 bytes so that both the row-major writes and the ds_read_b128 fragment reads are conflict-free.
    // (Fragment-shaped global loads - 16 bytes from each of 32 rows per instruction - cost 4x the tag lookups and
    // made this prologue as long as four weight tiles.)  The patches alias the weight-tile area, hence the barrier.
    constexpr int KH = K / 2, CPH = KH / 8, XCH = 32 * CPH / 64, XLD = KH * 2 + 16;   // per K-half; XLD in bytes
    static_assert(4 * 32 * XLD <= 64 * LDW * 2 + 4 * kStageBytes, "x staging patches must fit the workgroup's LDS");
    if constexpr (LNP != 2) {   // (the self-statistics prologue holds the whole fp32 panel in registers: loads these later)
        wload(wr0, nt0, 0);
        wload(wr1, nt0, 1);
    }
    bf16x8 xf[KS];
    if constexpr (!LNP) {
        char* xs = smem_raw + wave * (32 * XLD);
        const int mwave = mb * 128 + wave * 32;
        u32x4 t[2][XCH];   // both K-halves in flight at once: one memory round trip for the whole panel
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int id = lane + 64 * j, r = id / CPH, c = id - r * CPH;
                const int row = mwave + r < p.M ? mwave + r : p.M - 1;
                t[half][j] = *reinterpret_cast<const u32x4*>(A + (int64_t)row * p.lda + half * KH + c * 8);
            }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int id = lane + 64 * j, r = id / CPH, c = id - r * CPH;
                *reinterpret_cast<u32x4*>(xs + r * XLD + c * 16) = t[half][j];
            }
#pragma unroll
            for (int ks = 0; ks < KS / 2; ++ks)
                xf[half * (KS / 2) + ks] = *reinterpret_cast<const bf16x8*>(xs + l31 * XLD + ks * 32 + h * 16);
        }
    } else if constexpr (LNP == 2) {
        // LayerNorm in the prologue with NO producer hand-off (ispk_gemm_bf16_lnin, row_stats == NULL): a wave owns whole
        // rows, so it computes their statistics itself, exactly as the fused feed-forward kernel does for its pre-norm
        // (ffn_bf16_kernel, LX): the 32 rows x K fp32 stay in registers (K/2 VGPRs; no accumulator is live yet) through two
        // passes - per-lane float4 partials of one K-quarter -> a wave-private LDS table -> one lane per row half adds
        // them in fixed order - then (x - mean) * rstd * gamma + beta is rounded to bf16 on the way into the fragment
        // patch.  With N split over several workgroups each of them repeats the statistics (a few hundred cycles).
        constexpr int KQ = K / 4, CPQ = KQ / 4, XQ = 32 * CPQ / 64, XLQ = KQ * 2 + 16;   // per K-quarter; XLQ in bytes
        constexpr int GC = (64 % CPQ == 0) ? CPQ : (CPQ == 24 ? 8 : 1), NG = CPQ / GC;       // gcd(64, CPQ); groups per lane
        constexpr int PLD = CPQ + 1;                                                         // one quarter's partials per row
        static_assert(K == 384 || K == 256, "quarter staging is laid out for K = 256 / 384");
        static_assert(4 * 32 * XLQ + 4 * 32 * PLD * 4 + 4 * 64 * 4 <= 64 * (K + 8) * 2, "pre-norm staging aliases the weight tile");
        const float* Af = static_cast<const float*>(p.A);
        char* xs = smem_raw + wave * (32 * XLQ);
        float* part = reinterpret_cast<float*>(smem_raw + 4 * (32 * XLQ)) + wave * (32 * PLD);
        float* sst = reinterpret_cast<float*>(smem_raw + 4 * (32 * XLQ) + 4 * 32 * PLD * 4) + wave * 64;
        const int mwave = mb * 128 + wave * 32;
        float4 t[4][XQ];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                const int row = mwave + r < p.M ? mwave + r : p.M - 1;
                t[q][j] = *reinterpret_cast<const float4*>(Af + (int64_t)row * p.lda + q * KQ + c * 4);
            }
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                const float4 v = t[q][j];
                part[r * PLD + c] = (v.x + v.y) + (v.z + v.w);
            }
#pragma unroll
            for (int i = 0; i < CPQ / 2; ++i) tot += part[l31 * PLD + h * (CPQ / 2) + i];
        }
        tot += __shfl_xor(tot, 32, 64);
        if (h == 0) sst[2 * l31] = tot * (1.0f / (float)K);
        tot = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                const float mu = sst[2 * r];
                const float4 v = t[q][j];
                const float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
                part[r * PLD + c] = (a * a + b * b) + (cc * cc + d * d);
            }
#pragma unroll
            for (int i = 0; i < CPQ / 2; ++i) tot += part[l31 * PLD + h * (CPQ / 2) + i];
        }
        tot += __shfl_xor(tot, 32, 64);
        if (h == 0) sst[2 * l31 + 1] = 1.0f / sqrtf(tot * (1.0f / (float)K) + p.ln_eps);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 2) wload(wr0, nt0, 0);   // registers of the first two quarters are free again
            if (q == 3) wload(wr1, nt0, 1);
            float4 g[NG], be[NG];
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int c = (lane + 64 * u) % CPQ;
                g[u] = *reinterpret_cast<const float4*>(p.ln_gamma + q * KQ + c * 4);
                be[u] = *reinterpret_cast<const float4*>(p.ln_beta + q * KQ + c * 4);
            }
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                const float mean = sst[2 * r], rstd = sst[2 * r + 1];
                const float4 v = t[q][j], gg = g[j % NG], bb = be[j % NG];
                uint2 o;
                o.x = pack_bf16x2((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y);
                o.y = pack_bf16x2((v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
                *reinterpret_cast<uint2*>(xs + r * XLQ + c * 8) = o;
            }
#pragma unroll
            for (int ks = 0; ks < KS / 4; ++ks)
                xf[q * (KS / 4) + ks] = *reinterpret_cast<const bf16x8*>(xs + l31 * XLQ + ks * 32 + h * 16);
        }
    } else {
        // LayerNorm in the prologue (ispk_gemm_bf16_lnin): A is the fp32 residual stream; every row's (mean, rstd) comes
        // from the kernel that produced it (p.ln_out, float [M][2]); this wave normalises its 32 rows x K while it turns
        // them into fragments - (x - mean) * rstd * gamma + beta, rounded to bf16 - so the separate LayerNorm launch and
        // its write + re-read of a bf16 copy disappear.  fp32 rows are twice as long: four K-quarters through the patch,
        // two of them (one K-half) in flight at a time.  A lane meets only NG distinct 4-column groups per quarter
        // (chunk = (lane + 64 j) % CPQ), so gamma / beta are 2 NG float4 registers per quarter.
        constexpr int KQ = K / 4, CPQ = KQ / 4, XQ = 32 * CPQ / 64, XLQ = KQ * 2 + 16;   // per K-quarter; XLQ in bytes
        constexpr int GC = (64 % CPQ == 0) ? CPQ : (CPQ == 24 ? 8 : 1), NG = CPQ / GC;       // gcd(64, CPQ); groups per lane
        static_assert(K == 384 || K == 256, "quarter staging is laid out for K = 256 / 384");
        const float* Af = static_cast<const float*>(p.A);
        const float* stats = static_cast<const float*>(p.ln_out);
        char* xs = smem_raw + wave * (32 * XLQ);
        float* sst = reinterpret_cast<float*>(smem_raw + 4 * (32 * XLQ)) + wave * 64;      // this wave's 32 x (mean, rstd)
        const int mwave = mb * 128 + wave * 32;
        if (lane < 32) {
            const int row = mwave + lane < p.M ? mwave + lane : p.M - 1;
            const float2 ms = *reinterpret_cast<const float2*>(stats + 2 * (int64_t)row);
            sst[2 * lane] = ms.x;
            sst[2 * lane + 1] = ms.y;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 t[2][XQ];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int j = 0; j < XQ; ++j) {
                    const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                    const int row = mwave + r < p.M ? mwave + r : p.M - 1;
                    t[qq][j] = *reinterpret_cast<const float4*>(Af + (int64_t)row * p.lda + (2 * half + qq) * KQ + c * 4);
                }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * half + qq;
                float4 g[NG], be[NG];
#pragma unroll
                for (int u = 0; u < NG; ++u) {
                    const int c = (lane + 64 * u) % CPQ;
                    g[u] = *reinterpret_cast<const float4*>(p.ln_gamma + q * KQ + c * 4);
                    be[u] = *reinterpret_cast<const float4*>(p.ln_beta + q * KQ + c * 4);
                }
#pragma unroll
                for (int j = 0; j < XQ; ++j) {
                    const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                    const float mean = sst[2 * r], rstd = sst[2 * r + 1];
                    const float4 v = t[qq][j], gg = g[j % NG], bb = be[j % NG];
                    uint2 o;
                    o.x = pack_bf16x2((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y);
                    o.y = pack_bf16x2((v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
                    *reinterpret_cast<uint2*>(xs + r * XLQ + c * 8) = o;
                }
#pragma unroll
                for (int ks = 0; ks < KS / 4; ++ks)
                    xf[q * (KS / 4) + ks] = *reinterpret_cast<const bf16x8*>(xs + l31 * XLQ + ks * 32 + h * 16);
            }
        }
    }
    __syncthreads();   // every wave has its fragments: the patches may be overwritten by the weight tile
    wstore(wr0, 0);
    wstore(wr1, 1);
    wload(wr0, nt0 + 1, 0);
    __syncthreads();

    float mk = 1.0f;
    if (EP < 0 || ((uint32_t)EP & (ISPK_EP_MASK_ACC | ISPK_EP_MASK_OUT))) mk = (p.mask && m < p.M) ? (p.mask[m] ? 1.0f : 0.0f) : 1.0f;
    const uint32_t wbase = lds_addr(Ws + l31 * LDW + 8 * h);
    char* stage = smem_raw + (size_t)64 * LDW * sizeof(uint16_t) + wave * kStageBytes;  // wave-private epilogue patch
    const int mw0 = mb * 128 + wave * 32;
    float mo4[4];
    mask_rows<EP>(p, mw0, lane, mo4);
    float* ct = nullptr;   // ISPK_EP_ROWS_T: &C[batch of this lane's row][0][frame]
    if (ep_flag<EP>(p, ISPK_EP_ROWS_T) && m < p.M) {
        const int bi = m / p.cpb;
        ct = static_cast<float*>(p.C) + (int64_t)bi * p.bstride + (m - bi * p.cpb);
    }
    if constexpr (ST) { tA = __builtin_readcyclecounter(); tsum[0] = tA - t0; }
    for (int nt = nt0; nt < nt1; ++nt) {
        if constexpr (ST) { t0 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
        // The 2*KS weight-fragment reads of a tile form ONE stream through a D-deep register ring: the read for step
        // s+D is issued right after step s's MFMA, so D ds_read_b128 stay in flight (left to itself hipcc keeps two and
        // every MFMA waits on LDS latency; it also dissolves a ring written in plain C++, hence the opaque asm reads
        // with hand-counted waits).
        // Weight prefetch: a burst of HCH global loads per wave blocks instruction issue for ~16 cycles x HCH x 4 waves
        // (the CU's one texture-address path), so the loads are spread ONE per second MFMA gap instead:
        //   half-0 steps: wr1 <- tile nt+1, half 1   (stored after barrier 2, >= 24 MFMAs later)
        //   half-1 steps: wr0 <- tile nt+2, half 0   (stored after barrier 1 of the next tile)
        // Barrier 1 (every wave is done with half 0) is followed by the first D reads of half 1 - which must not be issued
        // earlier: the other waves' stores of that half (after barrier 2 of the previous tile) are only ordered by this
        // barrier - and then by the HCH LDS stores of wr0.  LDS operations retire in order, so the waits of those D
        // steps count the stores as younger operations instead of draining them, and the stores' issue time covers
        // most of the reads' latency.
        constexpr int D = 3, NS = 2 * KS;
        static_assert(2 * HCH <= KS && D - 1 + HCH <= 15, "prefetch slots / lgkmcnt range");
        bf16x8 wq[D];
        auto rd = [&](auto ic) {
            constexpr int st = decltype(ic)::value;
            lds_read_b128_asm<((st / KS) * 32 * LDW + 16 * (st % KS)) * 2>(wq[st % D], wbase);
        };
        static_for<0, D>(rd);
        static_for<0, NS>([&](auto ic) {
            constexpr int st = decltype(ic)::value;
            if constexpr (st == KS) {
                if constexpr (ST) { __builtin_amdgcn_sched_barrier(0); tA = __builtin_readcyclecounter(); }
                __syncthreads();  // barrier 1
                static_for<KS, KS + D>(rd);
                wstore(wr0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ST) { tB = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
            }
            constexpr int hend = st < KS ? KS : NS;
            constexpr int reads_after = (hend - 1 - st) < (D - 1) ? (hend - 1 - st) : (D - 1);
            constexpr int younger = (st >= KS && st < KS + D) ? D - 1 + HCH : reads_after;
            lds_wait<younger>();
            __builtin_amdgcn_sched_barrier(0);
            acc[st / KS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[st % D], xf[st % KS], acc[st / KS], 0, 0, 0);
            if constexpr (st + D < hend) rd(std::integral_constant<int, st + D>{});
            if constexpr (st % 2 == 0 && (st % KS) / 2 < HCH) {
                constexpr int i = (st % KS) / 2;
                if constexpr (st < KS) wload1(wr1[i], i, nt + 1, 1); else wload1(wr0[i], i, nt + 2, 0);
            }
        });
        if constexpr (ST) { __builtin_amdgcn_sched_barrier(0); tC = __builtin_readcyclecounter(); }
        __syncthreads();  // barrier 2: every wave is done with half 1
        wstore(wr1, 1);
        if constexpr (ST) { __builtin_amdgcn_sched_barrier(0); tD = __builtin_readcyclecounter(); }
        // epilogue: register 4g+e of tile half t is output feature n = nt*64 + t*32 + 8g + 4h + e, row m (this lane)
        if (p.cpb == -7) {  // ablation (experiments only): keep the accumulators live, skip the epilogue
            if (acc[0][0] + acc[1][5] == 123.456f) static_cast<float*>(p.C)[0] = 1.f;
        } else if (ep_flag<EP>(p, ISPK_EP_ROWS_T)) {
            store_rows_t<EP>(p, ct, nt * 64, acc[0], ep_flag<EP>(p, ISPK_EP_MASK_OUT) ? mk : 1.0f, h);
            store_rows_t<EP>(p, ct, nt * 64 + 32, acc[1], ep_flag<EP>(p, ISPK_EP_MASK_OUT) ? mk : 1.0f, h);
        } else if (ep_flag<EP>(p, ISPK_EP_OUT_BF16)) {
            store_rows_bf16<EP>(p, stage, mw0, nt * 64, acc[0], acc[1], mk, lane);
        } else if constexpr (EP >= 0) {   // (the generic instance is out of registers: it loads the residual in place)
            float4 r0[4], r1[4];
            resid_prefetch<EP>(p, mw0, nt * 64, lane, r0);
            resid_prefetch<EP>(p, mw0, nt * 64 + 32, lane, r1);
            store_rows_f32<EP>(p, stage, mw0, nt * 64, acc[0], mk, lane, nullptr, r0, mo4);
            store_rows_f32<EP>(p, stage, mw0, nt * 64 + 32, acc[1], mk, lane, nullptr, r1, mo4);
        } else {
            store_rows_f32<EP>(p, stage, mw0, nt * 64, acc[0], mk, lane, nullptr, nullptr, mo4);
            store_rows_f32<EP>(p, stage, mw0, nt * 64 + 32, acc[1], mk, lane, nullptr, nullptr, mo4);
        }
        if constexpr (ST) {
            __builtin_amdgcn_sched_barrier(0);
            const uint64_t tE = __builtin_readcyclecounter();
            tsum[1] += tA - t0; tsum[2] += tB - tA; tsum[3] += tC - tB; tsum[4] += tD - tC; tsum[5] += tE - tD;
        }
    }
    if constexpr (ST) {
        if (lane == 0) {
            uint64_t* dbg = static_cast<uint64_t*>(p.ln_out) + ((int64_t)blockIdx.x * 4 + wave) * 6;
            for (int i = 0; i < 6; ++i) dbg[i] = tsum[i];
        }
    }
}

template <int KC>
int32_t launch_panel(const GemmParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)64 * (64 * KC + 8) * sizeof(uint16_t) + 4 * kStageBytes;
    const int ntiles = (p.N + 63) / 64, mblocks = (p.M + 127) / 128;
    int nsplit = (512 + mblocks - 1) / mblocks;          // aim for >= 512 workgroups (2 per CU)
    if (const char* e = ispk_knob("ISPK_PANEL_NSPLIT")) nsplit = atoi(e);  // experiments only
    nsplit = nsplit < 1 ? 1 : (nsplit > ntiles ? ntiles : nsplit);
    const int per = (ntiles + nsplit - 1) / nsplit;
    nsplit = (ntiles + per - 1) / per;
    const int mb8 = (mblocks + 7) / 8 * 8;
#define ISPK_PANEL_GO(EP_, ST_, P_)                                                                                   \
    do {                                                                                                               \
        ISPK_RESERVE_LDS((&gemm_bf16_panel_kernel<KC, EP_, ST_>), lds, "gemm");                                      \
        hipLaunchKernelGGL((gemm_bf16_panel_kernel<KC, EP_, ST_>), dim3(mb8 * nsplit), dim3(256), lds, s, P_, per, nsplit, \
                           mblocks);                                                                                   \
        return ispk_launch_status();                                                                                   \
    } while (0)
    // the model's hot epilogues get branch-free instances (tools/trace_gemms.py lists what a forward launches)
    constexpr int kQkv = ISPK_EP_OUT_BF16, kFfn1 = ISPK_EP_OUT_BF16 | ISPK_EP_GELU, kProj = ISPK_EP_MASK_ACC | kEpResid;
    constexpr int kMelT = ISPK_EP_ROWS_T | ISPK_EP_MASK_OUT | kEpBias;   // to_mel (model.py:167-168)
    constexpr int kFfn1T = ISPK_EP_OUT_BF16 | ISPK_EP_DUAL_GELU;          // training: u and dropout(gelu(u)) from one launch
    constexpr int kFfn2B = ISPK_EP_OUT_BF16 | ISPK_EP_GELU_BWD, kFfn2Bm = kFfn2B | ISPK_EP_MASK_OUT;   // training: da -> du
    const int key = p.cpb == -7 ? -2 : ep_key(p);
#ifdef ISPK_EXPERIMENTS
    if (const char* e = ispk_knob("ISPK_PANEL_STAMP")) {   // experiments only: per-wave phase cycle sums -> uint64[grid*4][6]
        GemmParams q = p;
        q.ln_out = reinterpret_cast<void*>(strtoull(e, nullptr, 16));
        if (key == kQkv) ISPK_PANEL_GO(kQkv, true, q);
        if (key == kFfn1) ISPK_PANEL_GO(kFfn1, true, q);
        ISPK_PANEL_GO(kEpDyn, true, q);
    }
#endif
    if (p.ln_flags & 0x100u) {   // fp32 A + LayerNorm in the prologue (ispk_gemm_bf16_lnin)
#define ISPK_PANEL_GO_LN(EP_, LNP_)                                                                                     \
    do {                                                                                                               \
        ISPK_RESERVE_LDS((&gemm_bf16_panel_kernel<KC, EP_, false, LNP_>), lds, "gemm");                               \
        hipLaunchKernelGGL((gemm_bf16_panel_kernel<KC, EP_, false, LNP_>), dim3(mb8 * nsplit), dim3(256), lds, s, p, per,  \
                           nsplit, mblocks);                                                                           \
        return ispk_launch_status();                                                                                   \
    } while (0)
        if (p.ln_out) {   // the rows' statistics are given (decoder layers 2..: from the fused feed-forward kernel)
            if (key == kQkv) ISPK_PANEL_GO_LN(kQkv, 1);
            ISPK_PANEL_GO_LN(kEpDyn, 1);
        }
        if (key == kQkv) ISPK_PANEL_GO_LN(kQkv, 2);
        if (key == kFfn1) ISPK_PANEL_GO_LN(kFfn1, 2);
        ISPK_PANEL_GO_LN(kEpDyn, 2);
#undef ISPK_PANEL_GO_LN
    }
    if (ispk_knob("ISPK_EP_DYN") == nullptr) {   // (set: experiments, forces the generic epilogue)
        if (key == kFfn1T) ISPK_PANEL_GO(kFfn1T, false, p);
        if (key == kFfn2B) ISPK_PANEL_GO(kFfn2B, false, p);
        if (key == kFfn2Bm) ISPK_PANEL_GO(kFfn2Bm, false, p);
        if (key == kQkv) ISPK_PANEL_GO(kQkv, false, p);
        if (key == kFfn1) ISPK_PANEL_GO(kFfn1, false, p);
        if (key == kProj) ISPK_PANEL_GO(kProj, false, p);
        if (key == kMelT) ISPK_PANEL_GO(kMelT, false, p);
    }
    ISPK_PANEL_GO(kEpDyn, false, p);
#undef ISPK_PANEL_GO
}

bool panel_ok(const GemmParams& p) {
    return (p.K == 256 || p.K == 384) && vec_epilogue_ok(p) && rows_epilogue_ok(p) && ispk_knob("ISPK_NO_PANEL") == nullptr;
}

// The row-block kernel fits (a) the long reductions and (b) skinny outputs (N <= 192: the aligner's convolutions over
// 33,000 mel frames with 80-160 output channels), where one 64-row workgroup covers every output feature and the
// activations stream through exactly once; the generic 128x128 tiling wastes half its columns there.
bool wide_ok(const GemmParams& p) {
    const bool long_k = (p.K >= 512 || ispk_knob("ISPK_FORCE_WIDE")) && (p.N == 384 || p.N == 256 || p.N % 192 == 0);
    const bool skinny = p.N <= 192 && ispk_knob("ISPK_NO_SKINNY") == nullptr;
    // (rows: from one 64-row block on.  Below 2,048 rows the generic tile kernel used to take these shapes: with a handful of
    // workgroups its two-loads-in-flight K loop is pure latency - 800 x 384 x 1536, the text encoder's second feed-forward
    // Linear at 8 utterances per GPU, took 39 us on 39 workgroups; the row-block kernel's LDS-DMA ring streams the same
    // reduction several k-steps deep.  ISPK_WIDE_MIN_M: experiments.)
    const char* mm = ispk_knob("ISPK_WIDE_MIN_M");
    const int min_m = mm ? atoi(mm) : 64;
    return (long_k || skinny) && p.M >= min_m && vec_epilogue_ok(p) && !(p.flags & ISPK_EP_OUT_BF16) &&
           ispk_knob("ISPK_NO_WIDE") == nullptr;
}

template <int TM, int TN>
int32_t launch_f32(const GemmParams& p, hipStream_t s, int batch = 1) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr size_t lds = (size_t)2 * (BM + BN) * kLdt * sizeof(float);
    ISPK_RESERVE_LDS((&gemm_f32_kernel<TM, TN>), lds, "gemm");
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, batch);
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
    if (p.flags & ISPK_EP_ROWS_T) {
        ISPK_REQUIRE(p.cpb > 0 && p.M % p.cpb == 0, ISPK_E_SHAPE, "gemm: ROWS_T needs M %% cols_per_batch == 0");
    } else {
        ISPK_REQUIRE(p.cpb >= -7 && (p.cpb <= 0 || p.N % p.cpb == 0), ISPK_E_SHAPE, "gemm: N %% cols_per_batch != 0");
    }
    ISPK_REQUIRE((p.flags & ISPK_EP_GELU) == 0 || (p.flags & ISPK_EP_SILU) == 0, ISPK_E_UNSUPPORTED,
                 "gemm: GELU and SILU together");
    return 0;
}

}  // namespace

// tile choice: the largest tile that still gives every one of the 256 CUs a workgroup.  Returns TM*10 + TN.
extern "C" int32_t ispk_gemm_f32_tile(int32_t M, int32_t N, int32_t K) {
    (void)K;
    if (const char* e = ispk_knob("ISPK_GEMM_TILE")) return atoi(e);  // experiments only (tools/bench_kernels.py)
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
    ISPK_REQUIRE((flags & (ISPK_EP_OUT_BF16 | ISPK_EP_RESID_BF16 | ISPK_EP_ROWS_T)) == 0, ISPK_E_UNSUPPORTED,
                 "gemm_f32: bf16 output/residual and ROWS_T flags belong to ispk_gemm_bf16");
    if (M == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (ispk_gemm_f32_tile(M, N, K)) {
        case 22: return launch_f32<2, 2>(p, s);
        case 12: return launch_f32<1, 2>(p, s);
    }
    return launch_f32<1, 1>(p, s);
}

extern "C" int32_t ispk_gemm_f32_batched(const float* A, int64_t lda, int64_t stride_a, const float* W, int64_t ldw, int64_t stride_w,
                                         float* C, int64_t ldc, int64_t stride_c, int32_t batch, int32_t M, int32_t N, int32_t K,
                                         ispk_stream_t stream) {
    GemmParams p{A, lda, W, ldw, C, ldc, nullptr, nullptr, 0, nullptr, M, N, K, 0u, 0, 0};
    p.za = stride_a;
    p.zw = stride_w;
    p.zc = stride_c;
    if (int32_t rc = check_common(p, 4)) return rc;
    ISPK_REQUIRE(batch >= 0 && batch <= 65535 && stride_a % 4 == 0 && stride_w % 4 == 0 && ldc >= N, ISPK_E_SHAPE,
                 "gemm_f32_batched: batch=%d (<= 65535), operand strides multiples of 4, ldc >= N", batch);
    if (M == 0 || batch == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t wg128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * batch;
    if (wg128 >= 256) return launch_f32<2, 2>(p, s, batch);
    return launch_f32<1, 2>(p, s, batch);
}

static int32_t gelu_train_common(GemmParams& p, float dropout_p, uint64_t seed, const char* who) {
    ISPK_REQUIRE(p.A && p.W && p.C && p.aux, ISPK_E_NULL, "%s: null pointer", who);
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, ISPK_E_SHAPE, "%s: dropout_p must be in [0, 1)", who);
    if (int32_t rc = check_common(p, 2)) return rc;
    ISPK_REQUIRE(panel_ok(p) && p.N % 8 == 0 && p.ldc % 8 == 0 && p.ld_aux % 8 == 0 && ispk_aligned(p.C, 16) && ispk_aligned(p.aux, 16) &&
                     (int64_t)p.M * p.N < ((int64_t)1 << 32),
                 ISPK_E_UNSUPPORTED, "%s: K = 256 / 384, N %% 8 == 0, 16-byte aligned bf16 rows, M N < 2^32", who);
    p.drop_thresh = drop_thresh(dropout_p);
    p.drop_inv_keep = 1.0f / (1.0f - dropout_p);
    p.drop_seed = mix_seed(seed);
    p.drop_src = ispk_seed_source();
    return 0;
}

extern "C" int32_t ispk_gemm_bf16_gelu_train(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, uint16_t* u, int64_t ldu,
                                             uint16_t* a, int64_t ld_a, int32_t M, int32_t N, int32_t K, float dropout_p,
                                             uint64_t seed, ispk_stream_t stream) {
    GemmParams p{A, lda, W, ldw, u, ldu, nullptr, nullptr, 0, nullptr, M, N, K, ISPK_EP_OUT_BF16 | ISPK_EP_DUAL_GELU, 0, 0};
    p.aux = a;
    p.ld_aux = ld_a;
    if (int32_t rc = gelu_train_common(p, dropout_p, seed, "gemm_bf16_gelu_train")) return rc;
    if (M == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return K == 256 ? launch_panel<4>(p, s) : launch_panel<6>(p, s);
}

extern "C" int32_t ispk_gemm_bf16_gelu_bwd(const uint16_t* dY, int64_t lddy, const uint16_t* W2t, int64_t ldw, const uint16_t* u,
                                           int64_t ldu, uint16_t* du, int64_t lddu, const uint8_t* row_mask, int32_t M, int32_t N,
                                           int32_t K, float dropout_p, uint64_t seed, ispk_stream_t stream) {
    GemmParams p{dY, lddy, W2t, ldw, du, lddu, nullptr, nullptr, 0, row_mask, M, N, K,
                 ISPK_EP_OUT_BF16 | ISPK_EP_GELU_BWD | (row_mask ? ISPK_EP_MASK_OUT : 0u), 0, 0};
    p.aux = const_cast<uint16_t*>(u);
    p.ld_aux = ldu;
    if (int32_t rc = gelu_train_common(p, dropout_p, seed, "gemm_bf16_gelu_bwd")) return rc;
    if (M == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return K == 256 ? launch_panel<4>(p, s) : launch_panel<6>(p, s);
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
#ifdef ISPK_EXPERIMENTS
    if (ispk_knob("ISPK_PANEL_NOEPI")) p.cpb = -7;  // skips the epilogue: timing probes only, WRONG results
#endif
    if (int32_t rc = check_common(p, 2)) return rc;
    if (M == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (flags & ISPK_EP_ROWS_T) {
        ISPK_REQUIRE((K == 256 || K == 384) && !resid && N % 4 == 0 && ldc >= cols_per_batch &&
                         !(flags & (ISPK_EP_OUT_BF16 | ISPK_EP_BIAS_ROW | ISPK_EP_MASK_COL | ISPK_EP_GELU | ISPK_EP_SILU |
                                    ISPK_EP_MASK_ACC)) && (!bias || ispk_aligned(bias, 16)) && ispk_aligned(C, 4),
                     ISPK_E_UNSUPPORTED, "gemm_bf16: ROWS_T is built for K = 256 / 384, fp32 C, bias + MASK_OUT only");
        g_last_bf16_variant = 1000 + K / 64;
        return K == 256 ? launch_panel<4>(p, s) : launch_panel<6>(p, s);
    }
    if (panel_ok(p) && !(ispk_knob("ISPK_FORCE_WIDE") && (N == 384 || N == 256) && !(flags & ISPK_EP_OUT_BF16))) {
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
        const bool w192 = N <= 192 ? N > 128 : N % 192 == 0;
        // long K with few rows (the aligner's key convolution: 6,656 x 768 x 1920): 64-row blocks make 416 workgroups =
        // 1.6 rounds, each streaming its 0.7 MB of weights; 128-row blocks are ONE round of 208 that streams them half as often
        const int blocks128 = ((M + 127) / 128) * (N / 192);
        if (w192 && N > 192 && p.K >= 1536 && blocks128 >= 160 && blocks128 <= 256) {
            g_last_bf16_variant = 2000 + 34;
            return launch_wide<3, 4>(p, s);
        }
        g_last_bf16_variant = 2000 + (w192 ? 32 : 22);   // 192- or 128-feature column blocks over grid.y
        return w192 ? launch_wide<3, 2>(p, s) : launch_wide<2, 2>(p, s);
    }
    const int tile = ispk_gemm_f32_tile(M, N, K);  // same occupancy rule as the fp32 path
    g_last_bf16_variant = 3000 + tile;
    switch (tile) {
        case 22: return launch_bf16<2, 2>(p, s);
        case 12: return launch_bf16<1, 2>(p, s);
    }
    return launch_bf16<1, 1>(p, s);
}

// ---- split-K entry points (see launch_wide_splitk)
static int splitk_choice(int M, int N, int K, uint32_t flags) {
    if (M >= 2048 || M < 1 || N % 4 != 0 || K % 64 != 0 || K == 256 || K == 384) return 1;     // (K 256 / 384: the panel kernel)
    if (flags & (ISPK_EP_BIAS_ROW | ISPK_EP_MASK_COL | ISPK_EP_ROWS_T | ISPK_EP_DUAL_GELU | ISPK_EP_GELU_BWD | ISPK_EP_OUT_SPLIT)) return 1;
    const bool long_k = K >= 512 && (N == 384 || N == 256 || N % 192 == 0);
    const bool skinny = N <= 192 && K >= 512;
    if (!long_k && !skinny) return 1;
    const bool w192 = N <= 192 ? N > 128 : N % 192 == 0;
    const int bn = w192 ? 192 : 128;
    const int blocks = ((M + 63) / 64) * ((N + bn - 1) / bn);
    int best = 1;
    for (int s : {2, 3, 4, 5, 6, 8, 10, 12})       // slices of >= 3 steps of 64, one round of the chip's 256 CUs at most
        if (K % (64 * s) == 0 && K / (64 * s) >= 3 && blocks * s <= 256) best = s;
    return best;
}

extern "C" int32_t ispk_gemm_bf16_splitk_plan(int32_t M, int32_t N, int32_t K, uint32_t flags) {
    return splitk_choice(M, N, K, flags);
}

extern "C" int32_t ispk_gemm_bf16_splitk(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, void* C, int64_t ldc,
                                         const float* bias, const void* resid, int64_t ldr, const uint8_t* mask, int32_t M,
                                         int32_t N, int32_t K, uint32_t flags, float* workspace, int32_t ksplit,
                                         ispk_stream_t stream) {
    GemmParams p{A, lda, W, ldw, C, ldc, bias, resid, ldr, mask, M, N, K, flags, 0, 0};
    if (int32_t rc = check_common(p, 2)) return rc;
    ISPK_REQUIRE(workspace && ispk_aligned(workspace, 16), ISPK_E_NULL, "gemm_bf16_splitk: workspace (ksplit * M * N floats, 16-byte aligned)");
    ISPK_REQUIRE(ksplit >= 2 && ksplit <= 12 && K % (64 * ksplit) == 0 && splitk_choice(M, N, K, flags) > 1 && vec_epilogue_ok(p),
                 ISPK_E_UNSUPPORTED, "gemm_bf16_splitk: M=%d N=%d K=%d ksplit=%d is not a split-K shape (ispk_gemm_bf16_splitk_plan)", M, N, K, ksplit);
    if (M == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool w192 = N <= 192 ? N > 128 : N % 192 == 0;
    g_last_bf16_variant = 2000 + (w192 ? 32 : 22);
    return w192 ? launch_wide_splitk<3, 2>(p, s, ksplit, workspace) : launch_wide_splitk<2, 2>(p, s, ksplit, workspace);
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
// and walks the inner dimension in chunks of 32 hidden units, software-pipelined by one chunk.  Iteration c:
//   phase A  acc1' = W1[chunk c+1] · xfᵀ   (D/16 MFMAs; W1 chunk [32][D] streamed through LDS)  - and, in the gaps
//            between those MFMAs, the GELU of chunk c's accumulators, packed pairwise to bf16: which IS the B operand
//            of phase B (register 8s+j of lane half h = hidden 16s + 8(j>>2) + 4h + (j&3): accumulator-as-operand)
//   phase B  acc2[nt] += W2[nt-th 32 features][chunk c] · Pᵀ   (D/32 x 2 MFMAs; W2 chunk [D][32] in LDS, its 32 hidden
//            columns in that same permuted order so that each fragment is ONE conflict-free ds_read_b128)
// With one wave per SIMD nothing else hides anything, so every non-MFMA instruction is placed by hand in an MFMA gap
// (measured with in-kernel stamps, tools/stamp_ffn.py: unscheduled, GELU and the weight staging bursts each took as
// long as a 24-MFMA phase):
//   * GELU: 8 register pairs x 3 stages of ~7 packed-fp32 instructions, one stage per phase-A gap;
//   * LDS stores of the prefetched W1 chunk c+2 / W2 chunk c+1 (already in registers): one per gap at the start of
//     phase A; they retire in order with the operand reads, so the hand-counted lgkmcnt waits count them as younger
//     operations instead of draining them;
//   * the global loads of W1 chunk c+3 / W2 chunk c+2 into those registers: one per gap at the start of phase B (a
//     burst of 12 loads blocks the wave's issue for 12 x 16 cycles x 4 waves on the CU's one address path).
// The operand reads of both phases run as ONE stream through an RD-deep ring of opaque asm reads.  One barrier per chunk.
// Epilogue: the row-coalescing transpose (store_rows_f32) with residual and mask.
// LX: the input is the fp32 residual stream and the kernel applies the LayerNorm that precedes the block itself.
// PJ (with LX): the rows it normalises are produced here too - x1 = x + mask * (attention output · Woᵀ), transformer.py:91.
template <int KC, bool B1, bool PK, int EP = kEpDyn, bool ST = false, bool LN = false, bool LX = false, bool PJ = false>  // D = 64 KC; B1: Linear 1 bias; PK: packed W2; LN: + LayerNorm of the result
__global__ __launch_bounds__(256, 1) void ffn_bf16_kernel(GemmParams p, const uint16_t* __restrict__ W2, int64_t ldw2,
                                                          const float* __restrict__ bias1, int F) {
    [[maybe_unused]] uint64_t tk0 = 0;
    if constexpr (ST) tk0 = __builtin_readcyclecounter();
    constexpr int D = 64 * KC, KS = D / 16, NT = D / 32, HC = 32;
    constexpr int LD1 = D + 8, LD2 = HC + 8;         // padded LDS rows (bf16 elements)
    constexpr int C1 = HC * (D / 8) / 256;            // 16-B pieces per thread: W1 chunk (32 rows x D/8)
    constexpr int C2 = D * (HC / 8) / 256;            //                          W2 chunk (D rows x 4)
    constexpr int W2OPS = PK ? 1 : 2;                 // LDS stores per W2 piece
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint16_t* W1s = reinterpret_cast<uint16_t*>(smem_raw);   // [2][HC][LD1]
    uint16_t* W2s = W1s + 2 * HC * LD1;                      // [2][D][LD2]
    char* stage = smem_raw + (size_t)(2 * HC * LD1 + 2 * D * LD2) * 2 + (threadIdx.x >> 6) * kStageBytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int mw0 = blockIdx.x * 128 + wave * 32;
    const int m = mw0 + l31;
    [[maybe_unused]] const uint16_t* X = static_cast<const uint16_t*>(p.A);
    const uint16_t* W1 = static_cast<const uint16_t*>(p.W);
    const int nchunks = F / HC;

    // Weight staging: ONE set of C = D/64 16-byte registers per thread alternates between the two operands - in phase A
    // gap i it holds W2 piece i of chunk c+1 (stored to LDS there, then reloaded with W1 piece i of chunk c+2), in phase
    // B gap i that W1 piece (stored, then reloaded with W2 piece i of chunk c+2): each load has one 24-MFMA phase to
    // land (the weights are L2-resident), and only C x 4 registers are tied up instead of 2C x 4 - this kernel sits at
    // the edge of the 256 + 256 register file.  Addresses are "uniform base + one lane offset" (W1 rows are contiguous,
    // ldw1 == D, checked by the launcher): piece i of a chunk is 16-byte unit tid + 256 i.
    constexpr int C = C1;
    static_assert(C1 == C2, "D/64 pieces of either operand per thread");
    u32x4 R[C];
    const uint32_t lane_off1 = tid * 8;                                                    // elements
    const uint32_t lane_off2 = PK ? tid * 8 : (tid >> 2) * (uint32_t)ldw2 + (tid & 3) * 8;
    const int step1 = HC * D;
    const int step2 = PK ? D * HC : HC;              // packed: chunk c is one contiguous [D][32] block
    const int pstep2 = PK ? 2048 : 64 * (int)ldw2;   // piece i -> i + 1 (uniform)
    auto load1 = [&](int i, int c) {      // chunk indices past the end re-read the last chunk (never consumed)
        c = c < nchunks ? c : nchunks - 1;
        R[i] = *reinterpret_cast<const u32x4*>(W1 + ((int64_t)c * step1 + i * 2048) + lane_off1);
    };
    auto load2 = [&](int i, int c) {
        c = c < nchunks ? c : nchunks - 1;
        R[i] = *reinterpret_cast<const u32x4*>(W2 + ((int64_t)c * step2 + (int64_t)i * pstep2) + lane_off2);
    };
    uint32_t s1off[C];   // LDS byte offset of W1 piece i inside a buffer: row (tid + 256 i) / (D/8), 16-byte column
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int id = tid + 256 * i, r = id / (D / 8), cc = id - r * (D / 8);
        s1off[i] = (r * LD1 + cc * 8) * 2;
    }
    auto store1 = [&](int i, int buf) {
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(W1s) + buf * (HC * LD1 * 2) + s1off[i]) = R[i];
    };
    // Row-major W2: chunk rows are stored PERMUTED into the hidden order of the accumulator fragment (LDS position
    // 16s + 8h + j holds hidden 16s + 8(j>>2) + 4h + (j&3)), so a lane's k-step fragment is one aligned 16-byte run: the
    // global 16-byte piece cc (hidden 8cc .. 8cc+7; s = cc>>1, a = cc&1) lands as two 8-byte halves at 16s + 4a (h = 0)
    // and 16s + 8 + 4a (h = 1).  The packed image is already in that order.
    const int s2cc = tid & 3;
    uint16_t* const s2row = W2s + (tid >> 2) * LD2 + (PK ? s2cc * 8 : 16 * (s2cc >> 1) + 4 * (s2cc & 1));
    auto store2 = [&](int i, int buf) {
        uint16_t* row = s2row + (buf * D + 64 * i) * LD2;
        if constexpr (PK) {
            *reinterpret_cast<u32x4*>(row) = R[i];
        } else {
            uint2 lo, hi;
            lo.x = R[i][0]; lo.y = R[i][1]; hi.x = R[i][2]; hi.y = R[i][3];
            *reinterpret_cast<uint2*>(row) = lo;
            *reinterpret_cast<uint2*>(row + 8) = hi;
        }
    };

    // ---- prologue: W1 chunk 0 on its way, then the wave's 32 rows x D as MFMA fragments through a wave-private LDS
    // patch (coalesced 16-byte loads; see gemm_bf16_panel_kernel), one D-half at a time
    bf16x8 xf[KS];
    if constexpr (PJ) {
        // ---- the attention block's output projection, residual add and mask (attention.py:168-172, transformer.py:91) in
        // front of the pre-norm: x1 = x + mask * (o · Woᵀ).  The wave's 32 rows of o become fragments like any bf16 input;
        // Wo streams through the W1 buffers in 32-feature tiles (same shape as a W1 chunk) into the accumulators the main
        // loop uses later; the finished rows go to HBM once (the epilogue re-reads them as the residual) and stay in
        // registers in row layout for the LayerNorm statistics - the normalised bf16 rows reach the fragment patch
        // without ever being read back.  One launch and 100 MB of traffic less per layer than out-projection + FFN.
        const uint16_t* O = p.pj_o;
        const uint16_t* Wo = p.pj_w;
#pragma unroll
        for (int i = 0; i < C; ++i) R[i] = *reinterpret_cast<const u32x4*>(Wo + i * 2048 + lane_off1);
        {
            constexpr int KH = D / 2, CPH = KH / 8, XCH = 32 * CPH / 64, XLD = KH * 2 + 16;
            char* xs = smem_raw + wave * (32 * XLD);
            u32x4 t[2][XCH];
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int j = 0; j < XCH; ++j) {
                    const int id = lane + 64 * j, r = id / CPH, c = id - r * CPH;
                    const int row = mw0 + r < p.M ? mw0 + r : p.M - 1;
                    t[half][j] = *reinterpret_cast<const u32x4*>(O + (int64_t)row * p.pj_ldo + half * KH + c * 8);
                }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int j = 0; j < XCH; ++j) {
                    const int id = lane + 64 * j, r = id / CPH, c = id - r * CPH;
                    *reinterpret_cast<u32x4*>(xs + r * XLD + c * 16) = t[half][j];
                }
#pragma unroll
                for (int ks = 0; ks < KS / 2; ++ks)
                    xf[half * (KS / 2) + ks] = *reinterpret_cast<const bf16x8*>(xs + l31 * XLD + ks * 32 + h * 16);
            }
        }
        __syncthreads();   // the patches alias the weight buffers
#pragma unroll
        for (int i = 0; i < C; ++i) store1(i, 0);
        __syncthreads();
        f32x16 accp[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) accp[t][r] = 0.f;
        // (plain loop: one tile of Wo ahead, a barrier per tile.  Measured 35 us - 288 MFMAs are 5 - which is why this
        // entry point is not the default: it needs the main loop's hand-placed loads / ring reads to pay.)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) {
#pragma unroll
                for (int i = 0; i < C; ++i)
                    R[i] = *reinterpret_cast<const u32x4*>(Wo + ((int64_t)(t + 1) * step1 + i * 2048) + lane_off1);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(W1s + (t & 1) * (HC * LD1) + l31 * LD1 + 8 * h + 16 * ks);
                accp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[ks], accp[t], 0, 0, 0);
            }
            if (t + 1 < NT) {
#pragma unroll
                for (int i = 0; i < C; ++i) store1(i, (t + 1) & 1);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < C; ++i) load1(i, 0);   // W1 chunk 0 on its way
        // x1 rows: to HBM (the residual of the epilogue) and, in row layout, into registers
        GemmParams q = p;
        q.C = const_cast<void*>(p.resid); q.ldc = p.ldr; q.resid = p.pj_x; q.ldr = p.pj_ldx; q.bias = nullptr; q.N = D;
        q.flags = p.mask ? (uint32_t)ISPK_EP_MASK_ACC : 0u; q.cpb = 0;
        const float mkp = (p.mask && m < p.M) ? (p.mask[m] ? 1.0f : 0.0f) : 1.0f;
        float4 yv[NT][4];
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            store_rows_f32<kEpDyn>(q, stage, mw0, nt * 32, accp[nt], mkp, lane, yv[nt]);
#pragma unroll
            for (int i = 0; i < 4; ++i) rs[i] += (yv[nt][i].x + yv[nt][i].y) + (yv[nt][i].z + yv[nt][i].w);
        }
        auto row_total = [&](float (&v)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] += __shfl_xor(v[i], 1, 64);
                v[i] += __shfl_xor(v[i], 2, 64);
                v[i] += __shfl_xor(v[i], 4, 64);
            }
        };
        row_total(rs);
        float mean[4], qs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mean[i] = rs[i] * (1.0f / (float)D);
            qs[i] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = yv[t][i].x - mean[i], b = yv[t][i].y - mean[i], c = yv[t][i].z - mean[i],
                            d = yv[t][i].w - mean[i];
                qs[i] += (a * a + b * b) + (c * c + d * d);
            }
        row_total(qs);
        constexpr int XLP = D * 2 + 16;   // bytes per patch row: the whole normalised row in bf16
        static_assert(4 * 32 * XLP <= (2 * HC * LD1 + 2 * D * LD2) * 2, "normalised-row patches alias the weight buffers");
        char* xs = smem_raw + wave * (32 * XLP);
#pragma unroll
        for (int i = 0; i < 4; ++i) qs[i] = 1.0f / sqrtf(qs[i] * (1.0f / (float)D) + p.lx_eps);
        const int c4 = (lane & 7) * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 g = *reinterpret_cast<const float4*>(p.lx_gamma + t * 32 + c4);
            const float4 be = *reinterpret_cast<const float4*>(p.lx_beta + t * 32 + c4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 o;
                o.x = pack_bf16x2((yv[t][i].x - mean[i]) * qs[i] * g.x + be.x, (yv[t][i].y - mean[i]) * qs[i] * g.y + be.y);
                o.y = pack_bf16x2((yv[t][i].z - mean[i]) * qs[i] * g.z + be.z, (yv[t][i].w - mean[i]) * qs[i] * g.w + be.w);
                *reinterpret_cast<uint2*>(xs + (8 * i + (lane >> 3)) * XLP + (t * 32 + c4) * 2) = o;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xs + l31 * XLP + ks * 32 + h * 16);
    } else if constexpr (LX) {
#pragma unroll
        for (int i = 0; i < C; ++i) load1(i, 0);
        // Pre-norm in the prologue (ispk_ffn_bf16_prenorm; transformer.py:101-105: feed_forward(feed_forward_norm(x))):
        // a wave owns whole rows, so it computes their LayerNorm statistics itself - the 32 rows x D fp32 stay in
        // registers (D/2 VGPRs; the accumulators are not live yet) through two passes in fixed summation order (each
        // lane's float4 partial -> a wave-private LDS table -> one lane per row half adds them up: deterministic), then
        // (x - mean) * rstd * gamma + beta is rounded to bf16 on the way into the fragment patch, a K-quarter at a time.
        constexpr int KQ = D / 4, CPQ = KQ / 4, XQ = 32 * CPQ / 64, XLQ = KQ * 2 + 16;     // per K-quarter; XLQ in bytes
        constexpr int GC = (64 % CPQ == 0) ? CPQ : (CPQ == 24 ? 8 : 1), NG = CPQ / GC;       // gcd(64, CPQ); groups per lane
        constexpr int PLD = 4 * CPQ + 1;                                                     // partials per row, padded
        static_assert(4 * 32 * XLQ + 4 * 32 * PLD * 4 + 4 * 64 * 4 <= (2 * HC * LD1 + 2 * D * LD2) * 2,
                      "pre-norm staging aliases the weight buffers");
        const float* Xf = static_cast<const float*>(p.A);
        char* xs = smem_raw + wave * (32 * XLQ);
        float* part = reinterpret_cast<float*>(smem_raw + 4 * (32 * XLQ)) + wave * (32 * PLD);
        float* sst = reinterpret_cast<float*>(smem_raw + 4 * (32 * XLQ) + 4 * 32 * PLD * 4) + wave * 64;
        float4 t[4][XQ];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                const int row = mw0 + r < p.M ? mw0 + r : p.M - 1;
                t[q][j] = *reinterpret_cast<const float4*>(Xf + (int64_t)row * p.lda + q * KQ + c * 4);
            }
        auto row_total = [&]() {   // sum of this lane's row (l31) over the partial table; both halves end with the total
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 2 * CPQ; ++i) a += part[l31 * PLD + h * (2 * CPQ) + i];
            return a + __shfl_xor(a, 32, 64);
        };
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                const float4 v = t[q][j];
                part[r * PLD + q * CPQ + c] = (v.x + v.y) + (v.z + v.w);
            }
        const float mean_l = row_total() * (1.0f / (float)D);
        if (h == 0) sst[2 * l31] = mean_l;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                const float mu = sst[2 * r];
                const float4 v = t[q][j];
                const float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
                part[r * PLD + q * CPQ + c] = (a * a + b * b) + (cc * cc + d * d);
            }
        const float rstd_l = 1.0f / sqrtf(row_total() * (1.0f / (float)D) + p.lx_eps);
        if (h == 0) sst[2 * l31 + 1] = rstd_l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 g[NG], be[NG];
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int c = (lane + 64 * u) % CPQ;
                g[u] = *reinterpret_cast<const float4*>(p.lx_gamma + q * KQ + c * 4);
                be[u] = *reinterpret_cast<const float4*>(p.lx_beta + q * KQ + c * 4);
            }
#pragma unroll
            for (int j = 0; j < XQ; ++j) {
                const int id = lane + 64 * j, r = id / CPQ, c = id - r * CPQ;
                const float mean = sst[2 * r], rstd = sst[2 * r + 1];
                const float4 v = t[q][j], gg = g[j % NG], bb = be[j % NG];
                uint2 o;
                o.x = pack_bf16x2((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y);
                o.y = pack_bf16x2((v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
                *reinterpret_cast<uint2*>(xs + r * XLQ + c * 8) = o;
            }
#pragma unroll
            for (int ks = 0; ks < KS / 4; ++ks)
                xf[q * (KS / 4) + ks] = *reinterpret_cast<const bf16x8*>(xs + l31 * XLQ + ks * 32 + h * 16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i) load1(i, 0);
        constexpr int KH = D / 2, CPH = KH / 8, XCH = 32 * CPH / 64, XLD = KH * 2 + 16;
        static_assert(4 * 32 * XLD <= (2 * HC * LD1 + 2 * D * LD2) * 2, "x staging patches alias the weight buffers");
        char* xs = smem_raw + wave * (32 * XLD);
        u32x4 t[2][XCH];
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int id = lane + 64 * j, r = id / CPH, c = id - r * CPH;
                const int row = mw0 + r < p.M ? mw0 + r : p.M - 1;
                t[half][j] = *reinterpret_cast<const u32x4*>(X + (int64_t)row * p.lda + half * KH + c * 8);
            }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int id = lane + 64 * j, r = id / CPH, c = id - r * CPH;
                *reinterpret_cast<u32x4*>(xs + r * XLD + c * 16) = t[half][j];
            }
#pragma unroll
            for (int ks = 0; ks < KS / 2; ++ks)
                xf[half * (KS / 2) + ks] = *reinterpret_cast<const bf16x8*>(xs + l31 * XLD + ks * 32 + h * 16);
        }
    }
    __syncthreads();   // the patches alias the weight buffers
    // LDS <- W1 chunks 0 and 1, W2 chunk 0; R <- W2 chunk 1 (stored in phase A of iteration 0)
#pragma unroll
    for (int i = 0; i < C; ++i) store1(i, 0);
#pragma unroll
    for (int i = 0; i < C; ++i) load2(i, 0);
#pragma unroll
    for (int i = 0; i < C; ++i) store2(i, 0);
#pragma unroll
    for (int i = 0; i < C; ++i) load1(i, 1);
#pragma unroll
    for (int i = 0; i < C; ++i) store1(i, 1);
#pragma unroll
    for (int i = 0; i < C; ++i) load2(i, 1);
    f32x16 acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    __syncthreads();

    const uint32_t w1base = lds_addr(W1s + l31 * LD1 + 8 * h);
    const uint32_t w2base = lds_addr(W2s + l31 * LD2 + 8 * h);

    // GELU of one accumulator register pair in three stages of ~7 packed-fp32 instructions (= gelu_fast2, same operation
    // order): stage 0 reads the accumulators, stage 2 writes the packed bf16 pair into the next phase's B operand.
    union Frag { uint32_t u[4]; bf16x8 f; };
    f32x16 acc1;
    Frag pfC[2], pfN[2];                  // B operand of the running phase B / being produced for the next one
    f32x2 gx[8], gax[8], gz[8], gq[8];   // per-pair state between stages (one or two pairs live at a time)
    auto gelu_stage = [&](auto gc, int c) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, pr = g / 3, sg = g % 3;   // register pair, stage
        constexpr int r0 = 8 * (pr >> 2) + 2 * (pr & 3);                  // accumulator registers r0, r0 + 1
        if constexpr (sg == 0) {
            f32x2 v;
            v.x = acc1[r0]; v.y = acc1[r0 + 1];
            if constexpr (B1) {
                const int hid = c * HC + (r0 & 3) + 8 * (r0 >> 2) + 4 * h;
                v.x += bias1[hid]; v.y += bias1[hid + 1];
            }
            gx[pr] = v;
            f32x2 ax;
            ax.x = fabsf(v.x); ax.y = fabsf(v.y);
            gax[pr] = ax;
            const f32x2 z = ax * 0.70710678118654752440f;
            gz[pr] = z;
            f32x2 qq = z * 0.0000430638f + 0.0002765672f;
            qq = qq * z + 0.0001520143f;
            qq = qq * z + 0.0092705272f;
            gq[pr] = qq * z + 0.0422820123f;
        } else if constexpr (sg == 1) {
            const f32x2 z = gz[pr];
            f32x2 qq = gq[pr] * z + 0.0705230784f;
            qq = qq * z + 1.0f;
            qq = qq * qq; qq = qq * qq; qq = qq * qq; qq = qq * qq;
            f32x2 r;
            r.x = __builtin_amdgcn_rcpf(qq.x); r.y = __builtin_amdgcn_rcpf(qq.y);
            gq[pr] = r;
        } else {
            const f32x2 hx = gax[pr] * 0.5f;                  // max(x, 0) = 0.5 x + 0.5 |x|
            const f32x2 o = gx[pr] * 0.5f + (hx - hx * gq[pr]);
            pfN[pr >> 2].u[pr & 3] = pack_bf16x2(o.x, o.y);
        }
    };

    {   // pipeline fill: acc1 = W1[chunk 0] · xfᵀ and its GELU (plain reads, no overlap)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(W1s + l31 * LD1 + 8 * h + 16 * ks);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[ks], acc1, 0, 0, 0);
        }
        static_for<0, 24>([&](auto gc) { gelu_stage(gc, 0); });
        pfC[0].f = pfN[0].f;
        pfC[1].f = pfN[1].f;
    }
    __syncthreads();   // iteration 1 overwrites W1s[0]

    // Iteration c = 1 .. nchunks:  phase A  acc1 = W1[chunk c] · xfᵀ            (reads W1s[c & 1]; chunk nchunks: a re-run
    //                                                                           of the last chunk, never consumed)
    //                              phase B  acc2 += W2[chunk c-1] · pfCᵀ        (reads W2s[(c-1) & 1]) + GELU(chunk c) -> pfN
    constexpr int RD = 4, NB = 2 * NT, NS = KS + NB;
    static_assert(C <= KS && C <= NB && RD - 1 + RD * W2OPS <= 15, "staging slots / lgkmcnt range");
    [[maybe_unused]] uint64_t tsum[5] = {0, 0, 0, 0, 0}, t0 = 0, tA = 0, tB = 0;
    if constexpr (ST) tsum[3] = __builtin_readcyclecounter() - tk0;
    for (int c = 1; c <= nchunks; ++c) {
        if constexpr (ST) t0 = __builtin_readcyclecounter();
        const int pc = c & 1;
        const uint32_t a1 = w1base + pc * (HC * LD1 * 2);
        const uint32_t a2 = w2base + (pc ^ 1) * (D * LD2 * 2);
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
        static_for<0, NS>([&](auto ic) {
            constexpr int st = decltype(ic)::value;
            if constexpr (ST && st == KS) { __builtin_amdgcn_sched_barrier(0); tA = __builtin_readcyclecounter(); }
            // LDS operations younger than read(st): the later reads of the ring plus the stores of gaps st-RD .. st-1
            // (W2 pieces in phase-A gaps 0 .. C-1, W1 pieces in phase-B gaps KS .. KS+C-1)
            constexpr int reads_after = (NS - 1 - st) < (RD - 1) ? (NS - 1 - st) : (RD - 1);
            constexpr int g0 = st - RD < 0 ? 0 : st - RD;
            constexpr int n2 = (st < C ? st : C) - (g0 < C ? g0 : C);
            constexpr int hi1 = st < KS ? KS : (st < KS + C ? st : KS + C), lo1 = g0 < KS ? KS : (g0 < KS + C ? g0 : KS + C);
            lds_wait<reads_after + n2 * W2OPS + (hi1 - lo1)>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (st == 0) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[st % RD], xf[st], z, 0, 0, 0);
            } else if constexpr (st < KS) {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[st % RD], xf[st], acc1, 0, 0, 0);
            } else {
                constexpr int nt = (st - KS) / 2, s2 = (st - KS) % 2;
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[st % RD], pfC[s2].f, acc2[nt], 0, 0, 0);
            }
            if constexpr (st + RD < NS) issue(std::integral_constant<int, st + RD>{});
            // ---- gap work
            if constexpr (st < C) {                        // phase A: W2 chunk c -> LDS, W1 chunk c+1 -> register
                store2(st, pc);
                load1(st, c + 1);
            }
            if constexpr (st >= KS && st - KS < C) {        // phase B: W1 chunk c+1 -> LDS, W2 chunk c+1 -> register
                store1(st - KS, pc ^ 1);
                load2(st - KS, c + 1);
            }
            if constexpr (st >= KS) {                      // phase B: the GELU stages that fall into this gap (the first
                static_for<0, 24>([&](auto gc) {           // gap is left to the last phase-A MFMA's latency)
                    constexpr int g = decltype(gc)::value;
                    constexpr int gap = (g * NB / 24 + 1) < NB ? (g * NB / 24 + 1) : NB - 1;
                    if constexpr (gap == st - KS) gelu_stage(gc, c);
                });
            }
        });
        pfC[0].f = pfN[0].f;
        pfC[1].f = pfN[1].f;
        if constexpr (ST) { __builtin_amdgcn_sched_barrier(0); tB = __builtin_readcyclecounter(); }
        __syncthreads();
        if constexpr (ST) {
            const uint64_t tE = __builtin_readcyclecounter();
            tsum[0] += tA - t0; tsum[1] += tB - tA; tsum[2] += tE - tB;
        }
    }
    if constexpr (ST) tk0 = __builtin_readcyclecounter();

    float mk = 1.0f;
    if (EP < 0 || ((uint32_t)EP & (ISPK_EP_MASK_ACC | ISPK_EP_MASK_OUT))) mk = (p.mask && m < p.M) ? (p.mask[m] ? 1.0f : 0.0f) : 1.0f;
    // epilogue: the residual rows of tile nt+PF are requested while tile nt is transposed and stored
    constexpr int PF = LN ? 1 : 3;   // (the LayerNorm variant keeps all final values in registers)
    float mo4[4];
    mask_rows<EP>(p, mw0, lane, mo4);
    float4 rres[PF + 1][4];
    static_for<0, PF>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t < NT) resid_prefetch<EP>(p, mw0, t * 32, lane, rres[t]);
    });
    if constexpr (!LN) {
        static_for<0, NT>([&](auto tc) {
            constexpr int nt = decltype(tc)::value;
            if constexpr (nt + PF < NT) resid_prefetch<EP>(p, mw0, (nt + PF) * 32, lane, rres[(nt + PF) % (PF + 1)]);
            store_rows_f32<EP>(p, stage, mw0, nt * 32, acc2[nt], mk, lane, nullptr, rres[nt % (PF + 1)], mo4);
        });
    } else {
        // + LayerNorm of the finished rows for the next Linear (normalization.py:20-27; transformer.py:79 of the next
        // layer, or :205-206 after the last): a wave holds ALL D features of its 32 rows, so the statistics need no LDS
        // or barrier - two passes in fp32 over the final values kept in registers (D/32 x 4 float4, the accumulators
        // they replace die tile by tile), reduced over the 8 lanes that share a row.  Saves the separate LayerNorm
        // launch and its 50-MB re-read of the residual stream per decoder layer.
        float4 yv[NT][4];
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
        static_for<0, NT>([&](auto tc) {
            constexpr int nt = decltype(tc)::value;
            if constexpr (nt + PF < NT) resid_prefetch<EP>(p, mw0, (nt + PF) * 32, lane, rres[(nt + PF) % (PF + 1)]);
            store_rows_f32<EP>(p, stage, mw0, nt * 32, acc2[nt], mk, lane, yv[nt], rres[nt % (PF + 1)], mo4);
#pragma unroll
            for (int i = 0; i < 4; ++i) rs[i] += (yv[nt][i].x + yv[nt][i].y) + (yv[nt][i].z + yv[nt][i].w);
        });
        auto row_total = [&](float (&v)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] += __shfl_xor(v[i], 1, 64);
                v[i] += __shfl_xor(v[i], 2, 64);
                v[i] += __shfl_xor(v[i], 4, 64);
            }
        };
        row_total(rs);
        float mean[4], qs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mean[i] = rs[i] * (1.0f / (float)D);
            qs[i] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = yv[t][i].x - mean[i], b = yv[t][i].y - mean[i], c = yv[t][i].z - mean[i],
                            d = yv[t][i].w - mean[i];
                qs[i] += (a * a + b * b) + (c * c + d * d);
            }
        row_total(qs);
        const int c4 = (lane & 7) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mr = mw0 + 8 * i + (lane >> 3);
            if (mr >= p.M) continue;
            const float rstd = 1.0f / sqrtf(qs[i] * (1.0f / (float)D) + p.ln_eps);
            if (p.ln_flags & 4u) {   // statistics only: the consumer GEMM normalises in its prologue (ispk_gemm_bf16_lnin)
                if ((lane & 7) == 0)
                    *reinterpret_cast<float2*>(static_cast<float*>(p.ln_out) + 2 * (int64_t)mr) = make_float2(mean[i], rstd);
                continue;
            }
            const float mo = ((p.ln_flags & 1u) && p.mask) ? (p.mask[mr] ? 1.0f : 0.0f) : 1.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = t * 32 + c4;
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
                    pk.x = pack_bf16x2(o.x, o.y);
                    pk.y = pack_bf16x2(o.z, o.w);
                    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.ln_out) + off) = pk;
                } else {
                    *reinterpret_cast<float4*>(static_cast<float*>(p.ln_out) + off) = o;
                }
            }
        }
    }
    if constexpr (ST) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tsum[4] = __builtin_readcyclecounter() - tk0;
        if (lane == 0) {
            uint64_t* dbg = static_cast<uint64_t*>(p.ln_out) + (blockIdx.x * 4 + wave) * 5;
            for (int i = 0; i < 5; ++i) dbg[i] = tsum[i];
        }
    }
}

// W2 [D][F] (nn.Linear layout) -> [F/32][D][32] with each chunk's 32 hidden units in accumulator-fragment order
// (position 16s + 8h + 4a + b holds hidden 16s + 8a + 4h + b): a chunk becomes ONE contiguous 64*D-byte block, so the
// fused kernel streams it with full-line loads spread over every L2 channel (the [D][F] layout reads 64 bytes from each
// of D rows 2*F bytes apart) and stages it with straight 16-byte LDS stores.
__global__ void ffn_pack_w2_kernel(const uint16_t* __restrict__ W2, int64_t ldw2, uint16_t* __restrict__ out, int D, int F) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one output element
    if (idx >= (int64_t)D * F) return;
    const int pos = idx & 31, n = (idx >> 5) % D, c = (idx >> 5) / D;
    const int hid = (pos & 16) | ((pos & 4) << 1) | ((pos & 8) >> 1) | (pos & 3);
    out[idx] = W2[(int64_t)n * ldw2 + c * 32 + hid];
}

extern "C" int32_t ispk_ffn_pack_w2_bf16(const uint16_t* W2, int64_t ldw2, int32_t D, int32_t F, uint16_t* packed,
                                         ispk_stream_t stream) {
    ISPK_REQUIRE(W2 && packed, ISPK_E_NULL, "ffn_pack_w2: null pointer");
    ISPK_REQUIRE(D >= 1 && F >= 32 && F % 32 == 0 && ldw2 >= F, ISPK_E_SHAPE, "ffn_pack_w2: bad shape D=%d inner=%d", D, F);
    const int64_t n = (int64_t)D * F;
    hipLaunchKernelGGL(ffn_pack_w2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), W2, ldw2, packed, D, F);
    return ispk_launch_status();
}

namespace {
struct FfnLn {   // optional LayerNorm of the result (ispk_ffn_bf16_ln)
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float eps = 0.f;
    void* out = nullptr;
    int64_t ld = 0;
    uint32_t flags = 0;
};
struct FfnPj {   // attention output projection + residual + mask in front of the pre-norm
    const uint16_t* o = nullptr;
    int64_t ldo = 0;
    const uint16_t* Wo = nullptr;
    const float* x = nullptr;
    int64_t ldx = 0;
};
struct FfnLx {   // LayerNorm applied to the (fp32) input in the prologue
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float eps = 1e-5f;
};
int32_t ffn_launch(const void* x, int64_t ldx, const uint16_t* W1, int64_t ldw1, const float* bias1,
                   const uint16_t* W2, int64_t ldw2, const float* bias2, const float* resid, int64_t ldr,
                   const uint8_t* mask, float* out, int64_t ldo, int32_t rows, int32_t D, int32_t F, uint32_t flags,
                   const FfnLn* ln, ispk_stream_t stream, const FfnLx* lx = nullptr, const FfnPj* pj = nullptr) {
    ISPK_REQUIRE(x && W1 && W2 && out, ISPK_E_NULL, "ffn: null pointer");
    ISPK_REQUIRE(D == 384 || D == 256, ISPK_E_UNSUPPORTED, "ffn: dim %d (built for 256 / 384)", D);
    ISPK_REQUIRE(rows >= 0 && F >= 64 && F % 32 == 0, ISPK_E_SHAPE, "ffn: bad shape rows=%d inner=%d", rows, F);
    ISPK_REQUIRE((flags & ~(ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) == 0, ISPK_E_UNSUPPORTED, "ffn: unsupported flags");
    ISPK_REQUIRE(!((flags & (ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) && !mask), ISPK_E_NULL, "ffn: mask flag without mask");
    ISPK_REQUIRE(ldx % (lx ? 4 : 8) == 0 && ldw1 % 8 == 0 && ldw2 % 8 == 0 && ldo % 4 == 0 && (!resid || ldr % 4 == 0) && ldx >= D &&
                     ldw1 >= D && (ldw2 >= F || ldw2 == 0) && ldo >= D,
                 ISPK_E_ALIGN, "ffn: leading strides must be multiples of 8 (bf16) / 4 (fp32)");
    ISPK_REQUIRE(ispk_aligned(x, 16) && ispk_aligned(W1, 16) && ispk_aligned(W2, 16) && ispk_aligned(out, 16) &&
                     (!resid || ispk_aligned(resid, 16)) && (!bias2 || ispk_aligned(bias2, 16)),
                 ISPK_E_ALIGN, "ffn: pointers must be 16-byte aligned");
    if (rows == 0) return 0;
    GemmParams p{x, ldx, W1, ldw1, out, ldo, bias2, resid, ldr, mask, rows, D, D, flags & ~ISPK_EP_GELU, 0, 0};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((rows + 127) / 128);
    ISPK_REQUIRE((int64_t)F * ldw1 < (1ll << 30) && (int64_t)D * ldw2 < (1ll << 30), ISPK_E_SHAPE, "ffn: weights too large");
    ISPK_REQUIRE(ldw1 == D, ISPK_E_UNSUPPORTED, "ffn: W1 rows must be contiguous (ldw1 == dim)");
    constexpr int kHot = ISPK_EP_MASK_OUT | kEpResid;   // the transformer layer's call (transformer.py:105-110), no bias2
    const bool hot = ep_key(p) == kHot && ispk_knob("ISPK_EP_DYN") == nullptr;
    const bool packed = ldw2 == 0;   // W2 laid out by ispk_ffn_pack_w2_bf16
    if (ln) {
        const bool stats_only = ln->flags & 4u;   // ln_out = float [rows][2] (mean, rstd); gamma / beta unused
        ISPK_REQUIRE(ln->out && (stats_only || (ln->gamma && ln->beta)), ISPK_E_NULL, "ffn_ln: null LayerNorm argument");
        ISPK_REQUIRE(stats_only ? ispk_aligned(ln->out, 8)
                                : (ln->ld % 4 == 0 && ln->ld >= D && ispk_aligned(ln->out, (ln->flags & 2u) ? 8 : 16) &&
                                   ispk_aligned(ln->gamma, 16) && ispk_aligned(ln->beta, 16)),
                     ISPK_E_ALIGN, "ffn_ln: LayerNorm buffers must be 16-byte aligned, ln_ld a multiple of 4");
        ISPK_REQUIRE(!((ln->flags & 1u) && !mask), ISPK_E_NULL, "ffn_ln: ln mask flag set but mask is NULL");
        ISPK_REQUIRE(packed && !bias1, ISPK_E_UNSUPPORTED, "ffn_ln: needs the packed W2 image and no first-Linear bias");
        p.ln_gamma = ln->gamma; p.ln_beta = ln->beta; p.ln_out = ln->out; p.ln_ld = ln->ld; p.ln_eps = ln->eps;
        p.ln_flags = ln->flags;
    }
    if (lx) {
        ISPK_REQUIRE(lx->gamma && lx->beta, ISPK_E_NULL, "ffn_prenorm: null LayerNorm argument");
        ISPK_REQUIRE(ispk_aligned(lx->gamma, 16) && ispk_aligned(lx->beta, 16), ISPK_E_ALIGN,
                     "ffn_prenorm: gamma / beta must be 16-byte aligned");
        ISPK_REQUIRE(packed && !bias1, ISPK_E_UNSUPPORTED, "ffn_prenorm: needs the packed W2 image and no first-Linear bias");
        p.lx_gamma = lx->gamma; p.lx_beta = lx->beta; p.lx_eps = lx->eps;
    }
    if (pj) {
        ISPK_REQUIRE(lx && pj->o && pj->Wo && pj->x && resid && mask, ISPK_E_NULL, "attn_out_ffn: null pointer");
        ISPK_REQUIRE(pj->ldo % 8 == 0 && pj->ldo >= D && pj->ldx % 4 == 0 && pj->ldx >= D && ispk_aligned(pj->o, 16) &&
                         ispk_aligned(pj->Wo, 16) && ispk_aligned(pj->x, 16),
                     ISPK_E_ALIGN, "attn_out_ffn: attention output / Wo / x must be 16-byte aligned with strides %% 8 / %% 4");
        ISPK_REQUIRE(hot, ISPK_E_UNSUPPORTED, "attn_out_ffn: built for the transformer layer's call (row mask, no biases)");
        p.pj_o = pj->o; p.pj_ldo = pj->ldo; p.pj_w = pj->Wo; p.pj_x = pj->x; p.pj_ldx = pj->ldx;
    }
    void* stamp = nullptr;
#ifdef ISPK_EXPERIMENTS
    if (const char* e = ispk_knob("ISPK_FFN_STAMP")) {   // experiments only: per-wave phase cycle sums -> uint64[grid*4][3]
        stamp = reinterpret_cast<void*>(strtoull(e, nullptr, 16));
        ISPK_REQUIRE(D == 384 && packed && !bias1 && hot && !ln && !lx && !pj, ISPK_E_UNSUPPORTED, "ffn stamps: the hot instance only");
        p.ln_out = stamp;
    }
#endif
#define ISPK_FFN_GO(KC_, B1_, PK_, EP_, ST_)                                                                          \
    do {                                                                                                              \
        constexpr size_t lds = (size_t)(2 * 32 * (64 * KC_ + 8) + 2 * 64 * KC_ * 40) * 2 + 4 * kStageBytes;             \
        ISPK_RESERVE_LDS((&ffn_bf16_kernel<KC_, B1_, PK_, EP_, ST_>), lds, "ffn");                                    \
        hipLaunchKernelGGL((ffn_bf16_kernel<KC_, B1_, PK_, EP_, ST_>), grid, dim3(256), lds, s, p, W2, ldw2, bias1, F); \
        return ispk_launch_status();                                                                                  \
    } while (0)
#define ISPK_FFN_GO_LN(KC_, EP_)                                                                                      \
    do {                                                                                                              \
        constexpr size_t lds = (size_t)(2 * 32 * (64 * KC_ + 8) + 2 * 64 * KC_ * 40) * 2 + 4 * kStageBytes;             \
        ISPK_RESERVE_LDS((&ffn_bf16_kernel<KC_, false, true, EP_, false, true>), lds, "ffn");                         \
        hipLaunchKernelGGL((ffn_bf16_kernel<KC_, false, true, EP_, false, true>), grid, dim3(256), lds, s, p, W2, ldw2, \
                           bias1, F);                                                                                 \
        return ispk_launch_status();                                                                                  \
    } while (0)
#define ISPK_FFN_GO_LX(KC_, EP_, LN_)                                                                                 \
    do {                                                                                                              \
        constexpr size_t lds = (size_t)(2 * 32 * (64 * KC_ + 8) + 2 * 64 * KC_ * 40) * 2 + 4 * kStageBytes;             \
        ISPK_RESERVE_LDS((&ffn_bf16_kernel<KC_, false, true, EP_, false, LN_, true>), lds, "ffn");                    \
        hipLaunchKernelGGL((ffn_bf16_kernel<KC_, false, true, EP_, false, LN_, true>), grid, dim3(256), lds, s, p, W2,  \
                           ldw2, bias1, F);                                                                           \
        return ispk_launch_status();                                                                                  \
    } while (0)
#define ISPK_FFN_GO_PJ(KC_, LN_)                                                                                      \
    do {                                                                                                              \
        constexpr size_t lds = (size_t)(2 * 32 * (64 * KC_ + 8) + 2 * 64 * KC_ * 40) * 2 + 4 * kStageBytes;             \
        ISPK_RESERVE_LDS((&ffn_bf16_kernel<KC_, false, true, kHot, false, LN_, true, true>), lds, "ffn");             \
        hipLaunchKernelGGL((ffn_bf16_kernel<KC_, false, true, kHot, false, LN_, true, true>), grid, dim3(256), lds, s, p,  \
                           W2, ldw2, bias1, F);                                                                       \
        return ispk_launch_status();                                                                                  \
    } while (0)
#ifdef ISPK_EXPERIMENTS
#define ISPK_FFN_STAMPED() do { if (stamp) ISPK_FFN_GO(6, false, true, kHot, true); } while (0)
#else
#define ISPK_FFN_STAMPED() (void)stamp
#endif
    // Variants that measured slower than what the module mirror calls (projection prologue `pj`; LayerNorm epilogue without the
    // pre-norm prologue) are compiled into the experiments build only (libispk_exp.so); the product ABI does not reach them.
#ifdef ISPK_EXPERIMENTS
#define ISPK_FFN_KC_EXP(KC_)                                               \
    do {                                                                   \
        if (pj && ln) ISPK_FFN_GO_PJ(KC_, true);                           \
        if (pj) ISPK_FFN_GO_PJ(KC_, false);                                \
        if (!lx && ln && hot) ISPK_FFN_GO_LN(KC_, kHot);                   \
        if (!lx && ln) ISPK_FFN_GO_LN(KC_, kEpDyn);                        \
    } while (0)
#else
#define ISPK_FFN_KC_EXP(KC_) ISPK_REQUIRE(!pj && (lx || !ln), ISPK_E_UNSUPPORTED, "ffn: variant of the experiments build only")
#endif
#define ISPK_FFN_KC(KC_)                                                   \
    do {                                                                   \
        ISPK_FFN_KC_EXP(KC_);                                              \
        if (lx && ln && hot) ISPK_FFN_GO_LX(KC_, kHot, true);              \
        if (lx && ln) ISPK_FFN_GO_LX(KC_, kEpDyn, true);                   \
        if (lx && hot) ISPK_FFN_GO_LX(KC_, kHot, false);                   \
        if (lx) ISPK_FFN_GO_LX(KC_, kEpDyn, false);                        \
        ISPK_FFN_STAMPED();                                                \
        if (!bias1 && packed && hot) ISPK_FFN_GO(KC_, false, true, kHot, false);  \
        if (!bias1 && packed) ISPK_FFN_GO(KC_, false, true, kEpDyn, false);  \
        if (!bias1) ISPK_FFN_GO(KC_, false, false, kEpDyn, false);          \
        if (packed) ISPK_FFN_GO(KC_, true, true, kEpDyn, false);            \
        ISPK_FFN_GO(KC_, true, false, kEpDyn, false);                       \
    } while (0)
    if (D == 384) ISPK_FFN_KC(6); else ISPK_FFN_KC(4);
#undef ISPK_FFN_KC
#undef ISPK_FFN_KC_EXP
#undef ISPK_FFN_STAMPED
#undef ISPK_FFN_GO_PJ
#undef ISPK_FFN_GO_LX
#undef ISPK_FFN_GO_LN
#undef ISPK_FFN_GO
    return ispk_launch_status();
}
}  // namespace

extern "C" int32_t ispk_ffn_bf16(const uint16_t* x, int64_t ldx, const uint16_t* W1, int64_t ldw1, const float* bias1,
                                 const uint16_t* W2, int64_t ldw2, const float* bias2, const float* resid, int64_t ldr,
                                 const uint8_t* mask, float* out, int64_t ldo, int32_t rows, int32_t D, int32_t F,
                                 uint32_t flags, ispk_stream_t stream) {
    return ffn_launch(x, ldx, W1, ldw1, bias1, W2, ldw2, bias2, resid, ldr, mask, out, ldo, rows, D, F, flags, nullptr, stream);
}

#ifdef ISPK_EXPERIMENTS   // round 2, the four-wave kernel writing x1 and re-reading it: measured slower than two launches; experiments
                          // build only (the product's ispk_attn_out_ffn_bf16 is csrc/ffn2.hip's projection mode: x1 stays in the accumulators)
extern "C" int32_t ispk_ffn_bf16_ln(const uint16_t* x, int64_t ldx, const uint16_t* W1, int64_t ldw1,
                                    const uint16_t* W2_packed, const float* bias2, const float* resid, int64_t ldr,
                                    const uint8_t* mask, float* out, int64_t ldo, int32_t rows, int32_t D, int32_t F,
                                    uint32_t flags, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                    void* ln_out, int64_t ln_ld, uint32_t ln_flags, ispk_stream_t stream) {
    FfnLn ln{ln_gamma, ln_beta, ln_eps, ln_out, ln_ld, ln_flags};
    return ffn_launch(x, ldx, W1, ldw1, nullptr, W2_packed, 0, bias2, resid, ldr, mask, out, ldo, rows, D, F, flags, &ln,
                      stream);
}
#endif

#ifdef ISPK_EXPERIMENTS   // round 2, the four-wave kernel writing x1 and re-reading it: measured slower than two launches; experiments
                          // build only (the product's ispk_attn_out_ffn_bf16 is csrc/ffn2.hip's projection mode: x1 stays in the accumulators)
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
#endif

extern "C" int32_t ispk_gemm_bf16_lnin(const float* x, int64_t ldx, const float* row_stats, const float* ln_gamma,
                                       const float* ln_beta, float ln_eps, const uint16_t* W, int64_t ldw, void* C, int64_t ldc,
                                       const float* bias, const void* resid, int64_t ldr, const uint8_t* mask, int32_t M,
                                       int32_t N, int32_t K, uint32_t flags, ispk_stream_t stream) {
    ISPK_REQUIRE(x && ln_gamma && ln_beta && W && C, ISPK_E_NULL, "gemm_lnin: null pointer");
    ISPK_REQUIRE(K == 256 || K == 384, ISPK_E_UNSUPPORTED, "gemm_lnin: K=%d (built for 256 / 384)", K);
    ISPK_REQUIRE(M >= 0 && N >= 1, ISPK_E_SHAPE, "gemm_lnin: bad shape M=%d N=%d", M, N);
    ISPK_REQUIRE(ldx % 4 == 0 && ldx >= K && ldw % 8 == 0 && ldw >= K && ispk_aligned(x, 16) && ispk_aligned(W, 16) &&
                     ispk_aligned(row_stats, 8) && ispk_aligned(ln_gamma, 16) && ispk_aligned(ln_beta, 16),
                 ISPK_E_ALIGN, "gemm_lnin: x / gamma / beta / W must be 16-byte aligned (ldx %% 4, ldw %% 8)");
    GemmParams p{x, ldx, W, ldw, C, ldc, bias, resid, ldr, mask, M, N, K, flags, 0, 0};
    ISPK_REQUIRE(!((flags & (ISPK_EP_MASK_ACC | ISPK_EP_MASK_OUT)) && !mask), ISPK_E_NULL, "gemm_lnin: mask flag without mask");
    ISPK_REQUIRE(!(flags & (ISPK_EP_ROWS_T | ISPK_EP_BIAS_ROW | ISPK_EP_MASK_COL)) && vec_epilogue_ok(p) && rows_epilogue_ok(p),
                 ISPK_E_UNSUPPORTED, "gemm_lnin: needs a row-major output with 16-byte aligned rows");
    if (M == 0) return 0;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_out = const_cast<float*>(row_stats); p.ln_flags = 0x100u;
    p.ln_eps = ln_eps;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return K == 256 ? launch_panel<4>(p, s) : launch_panel<6>(p, s);
}

extern "C" int32_t ispk_ffn_bf16_prenorm(const float* x, int64_t ldx, const float* norm_gamma, const float* norm_beta,
                                         float norm_eps, const uint16_t* W1, int64_t ldw1, const uint16_t* W2_packed,
                                         const float* bias2, const uint8_t* mask, float* out, int64_t ldo, int32_t rows,
                                         int32_t D, int32_t F, uint32_t flags, float* row_stats, float stats_eps,
                                         ispk_stream_t stream) {
    ISPK_REQUIRE(x && ispk_aligned(x, 16), ISPK_E_ALIGN, "ffn_prenorm: x must be a 16-byte aligned fp32 pointer");
    FfnLx lx{norm_gamma, norm_beta, norm_eps};
    FfnLn ln{nullptr, nullptr, stats_eps, row_stats, 0, 4u};
    return ffn_launch(x, ldx, W1, ldw1, nullptr, W2_packed, 0, bias2, x, ldx, mask, out, ldo, rows, D, F, flags,
                      row_stats ? &ln : nullptr, stream, &lx);
}

#ifdef ISPK_EXPERIMENTS   // round 2, the four-wave kernel writing x1 and re-reading it: measured slower than two launches; experiments
                          // build only (the product's ispk_attn_out_ffn_bf16 is csrc/ffn2.hip's projection mode: x1 stays in the accumulators)
extern "C" int32_t ispk_attn_out_ffn_bf16_v1(const uint16_t* attn_out, int64_t ldao, const uint16_t* Wo, const float* x,
                                          int64_t ldx, const float* norm_gamma, const float* norm_beta, float norm_eps,
                                          const uint16_t* W1, int64_t ldw1, const uint16_t* W2_packed, const uint8_t* mask,
                                          float* x1, int64_t ldx1, float* out, int64_t ldo, int32_t rows, int32_t D,
                                          int32_t F, float* row_stats, float stats_eps, ispk_stream_t stream) {
    ISPK_REQUIRE(x1 && ispk_aligned(x1, 16) && ldx1 % 4 == 0 && ldx1 >= D, ISPK_E_ALIGN,
                 "attn_out_ffn: x1 must be a 16-byte aligned fp32 buffer [rows][dim]");
    FfnPj pj{attn_out, ldao, Wo, x, ldx};
    FfnLx lx{norm_gamma, norm_beta, norm_eps};
    FfnLn ln{nullptr, nullptr, stats_eps, row_stats, 0, 4u};
    // the kernel writes x1 in its prologue and re-reads it as the residual of its epilogue
    return ffn_launch(x1, ldx1, W1, ldw1, nullptr, W2_packed, 0, nullptr, x1, ldx1, mask, out, ldo, rows, D, F,
                      ISPK_EP_MASK_OUT, row_stats ? &ln : nullptr, stream, &lx, &pj);
}
#endif
