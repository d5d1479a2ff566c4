// The fp32-grade fast path: every fp32 operand value v travels as TWO fp16 terms, hi = fp16(v) and lo = fp16(v - hi)
// (22 significant bits together), and a product of two such operands is three v_mfma_f32_32x32x16_f16 instructions,
//     a * b  ~=  hi_a hi_b + hi_a lo_b + lo_a hi_b          (the lo lo term is below 2^-22 of the product),
// each fp16 x fp16 product exact in the MFMA's fp32 accumulator.  That is three matrix instructions at the fp16 / bf16
// rate (2.5 PF dense) where the exact-fp32 path (v_mfma_f32_32x32x2_f32, 157 TF) needs sixteen times the cycles of one:
// 5.3x the fp32 MFMA rate at fp32-grade results.  (Split into bf16 terms, the same three products keep only 16 bits: mel
// L-inf 5e-5 against 9e-6 on the oracle's forward, tools/split_numerics.py - fp16's 11-bit terms are what makes two terms
// enough.)  Domain: |v| <= 65504 (clamped); terms below fp16's normal range keep fp16's subnormal spacing (2^-24
// absolute), which MFMA inputs do not flush.
//
// Operand format in HBM ("split planes"): a [rows][cols] fp32 matrix becomes two fp16 matrices of the same shape and
// leading stride, the lo plane `plane` elements behind the hi plane - 4 bytes per value, like the fp32 it replaces, so
// that a producer splits each value ONCE (LayerNorm, GELU epilogue, attention epilogue) and every consumer GEMM streams
// fp16 tiles by LDS-DMA without touching a VALU.
//
// Replaces, on the parity path, the exact-fp32 kernels behind the same reference call sites: nn.Linear (attention.py:105,
// 111,168; feedforward.py:33-36; transformer.py:170; model.py:167-168; the aligner's Conv1d as a GEMM, alignment.py:69-83)
// and Attend.efficient_attn (attend.py:49-122).
#include <stdlib.h>

#include "gemm_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t su32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x8 as_f16x8(const bf16x8& v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ void keep_alive(const bf16x8& v) { asm volatile("" ::"v"(v)); }

// two fp32 values -> packed fp16 hi terms and packed fp16 lo terms (round to nearest even both times; v - hi is exact)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    a = __builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f);
    b = __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f);
    f16x2 h;
    h.x = (_Float16)a;
    h.y = (_Float16)b;
    f16x2 l;
    l.x = (_Float16)(a - (float)h.x);
    l.y = (_Float16)(b - (float)h.y);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
// the same without the range clamp, for values known to lie inside fp16's range (softmax probabilities, scaled queries)
__device__ __forceinline__ void split_pair_nc(float a, float b, uint32_t& hi, uint32_t& lo) {
    f16x2 h;
    h.x = (_Float16)a;
    h.y = (_Float16)b;
    f16x2 l;
    l.x = (_Float16)(a - (float)h.x);
    l.y = (_Float16)(b - (float)h.y);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// ------------------------------------------------------------------------------------------------ fp32 -> split planes
// One thread per 4 consecutive values: 16-byte load, two 8-byte stores.
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ hi,
                                                        uint16_t* __restrict__ lo, int64_t ldy, int rows, int cols4) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * cols4) return;
    const int r = (int)(idx / cols4), c = (int)(idx - (int64_t)r * cols4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + c);
    uint2 h, l;
    split_pair(v.x, v.y, h.x, l.x);
    split_pair(v.z, v.w, h.y, l.y);
    *reinterpret_cast<uint2*>(hi + (int64_t)r * ldy + c) = h;
    *reinterpret_cast<uint2*>(lo + (int64_t)r * ldy + c) = l;
}

// The epilogue of the split GEMMs: a wave's RT x TN accumulator tiles (rows m0 + (wm RT + rt) 32 .., features nb0 + (wn TN + t)
// 32 ..) through the wave's private LDS patches - fp32 rows (+ bias, GELU / SiLU, masks, fp32 residual), split planes, or the
// transposed [batch][feature][frame] store of to_mel.
template <int TN, int RT>
__device__ __forceinline__ void split_store_tiles(const GemmParams& p, char* smem_raw, f32x16 (&acc)[RT][TN], int m0, int nb0,
                                                  int wm, int wn, int wave, int lane) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int mw = m0 + (wm * RT + rt) * 32;    // first row of this wave's tile rt
        const int m = mw + l31;
        const float mk = (p.mask && m < p.M) ? (p.mask[m] ? 1.0f : 0.0f) : 1.0f;
        if (p.flags & ISPK_EP_ROWS_T) {
            // rows are [batch][T] frames, T = cpb; output feature n of frame (b, t) goes to C[b][n][t] (to_mel + transpose)
            float* cbp = nullptr;
            if (m < p.M) {
                const int bb = m / p.cpb;
                cbp = static_cast<float*>(p.C) + (int64_t)bb * p.bstride + (m - bb * p.cpb);
            }
            const float mo = (p.flags & ISPK_EP_MASK_OUT) ? mk : 1.0f;
#pragma unroll
            for (int t = 0; t < TN; ++t) store_rows_t(p, cbp, nb0 + (wn * TN + t) * 32, acc[rt][t], mo, h);
        } else if (p.flags & ISPK_EP_OUT_SPLIT) {
            // split-plane output: two adjacent 32-feature tiles per pass through a pair of wave-private LDS patches
            char* stage = smem_raw + wave * (2 * kStageBytes);
            static_assert(TN % 2 == 0 || TN == 3, "tile pairing");
            uint16_t* Chi = static_cast<uint16_t*>(p.C);
#pragma unroll
            for (int t = 0; t < TN; t += 2) {
                const bool pair = t + 1 < TN;
                const int n0 = nb0 + (wn * TN + t) * 32;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    if (tt == 1 && !pair) break;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[rt][t + tt < TN ? t + tt : t][4 * g + e];
                        const int n = n0 + tt * 32 + 8 * g + 4 * h;
                        pre_stage(p, n < p.N ? n : 0, v, mk);
                        if (p.flags & ISPK_EP_MASK_OUT) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= mk;
                        }
                        uint2 oh, ol;
                        split_pair(v[0], v[1], oh.x, ol.x);
                        split_pair(v[2], v[3], oh.y, ol.y);
                        *reinterpret_cast<uint2*>(stage + l31 * kStageRow + (tt * 32 + 8 * g + 4 * h) * 2) = oh;
                        *reinterpret_cast<uint2*>(stage + kStageBytes + l31 * kStageRow + (tt * 32 + 8 * g + 4 * h) * 2) = ol;
                    }
                }
                const int c = lane & 7, n = n0 + 8 * c;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 8 * i + (lane >> 3), mr = mw + r;
                    const uint4 vh = *reinterpret_cast<const uint4*>(stage + r * kStageRow + c * 16);
                    const uint4 vl = *reinterpret_cast<const uint4*>(stage + kStageBytes + r * kStageRow + c * 16);
                    if (mr < p.M && n < p.N && (pair || c < 4)) {
                        *reinterpret_cast<uint4*>(Chi + (int64_t)mr * p.ldc + n) = vh;
                        *reinterpret_cast<uint4*>(Chi + p.c_plane + (int64_t)mr * p.ldc + n) = vl;
                    }
                }
            }
        } else {
            char* stage = smem_raw + wave * kStageBytes;
            float mo4[4];
            mask_rows(p, mw, lane, mo4);
            float4 rres[TN][4];
#pragma unroll
            for (int t = 0; t < TN; ++t) resid_prefetch(p, mw, nb0 + (wn * TN + t) * 32, lane, rres[t]);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                store_rows_f32(p, stage, mw, nb0 + (wn * TN + t) * 32, acc[rt][t], mk, lane, nullptr, rres[t], mo4);
        }
    }
}

// ------------------------------------------------------------------------------------------------ Linear on split planes
// C = epilogue(A · Wᵀ) with A [M][K] and W [N][K] given as split planes.  Skeleton of gemm_bf16_wide_kernel (gemm.hip): one
// workgroup = 32 WM RT activation rows x 64 TN output features on WM x 2 waves, a wave holds RT 32-row tiles x TN 32-feature
// tiles in accumulators, computed transposed (D = W_tile · Xᵀ) so that a lane owns one activation row for the
// row-coalescing epilogue; operand tiles stream by LDS-DMA (global_load_lds_dwordx4) into a ring of S slots, one raw
// s_barrier per K chunk.  What is different:
//   * a K chunk is 32 deep and an LDS row (128 B) holds BOTH terms of it: 16-byte slots 0-3 = hi k 0..31, slots 4-7 = lo.
//     A DMA lane picks its plane with its slot, so in HBM the planes stay separate matrices (and a leading stride SMALLER
//     than K - the sliding-window view that turns a padded channel-last Conv1d into a GEMM - keeps working per plane);
//     the XOR swizzle (physical slot = logical slot ^ ((row >> 1) & 7)) is applied on the source side as before;
//   * per 16-deep k-step a wave reads X hi / X lo and W hi / W lo fragments (2 RT + 2 TN ds_read_b128) for 3 RT TN MFMAs.
//     Both the LDS fill (4 bytes per operand value, every workgroup streaming its X rows and W rows through LDS) and the
//     fragment reads (every wave re-reading them) are co-critical with the MFMAs at 128 x 256 blocks (measured:
//     tools/ablate_split.py), hence RT = 2: 256 x 256 blocks with 64 x 128 wave tiles halve the fill per product and take
//     the reads from 0.83 to 0.5 per MFMA;
//     (Tried and dropped, tools measured: a 16-deep-chunk variant with 128 x 256 tiles and TWO workgroups per CU, so that one's
//     epilogue would run under the other's MFMAs - 72 / 63 / 183 / 201 us against 56 / 47 / 180 / 150 for q/kv, out, FFN1,
//     FFN2 at 32,768 rows: its LDS rows hold 32-byte pieces of each plane, and LDS-DMA from 32-byte pieces runs at half rate.)
//   * XCD-aware block order: workgroups are dealt to the 8 XCDs round-robin in launch order, so workgroup `lin` serves row
//     block 8 (seq / NCB) + lin % 8, column block seq % NCB with seq = lin / 8: the column blocks of one row block run
//     back to back on ONE XCD and find the X rows in that XCD's L2 after the first fetch.
// AB: timing probes of the experiments build (tools/ablate_split.py; WRONG results) - 1 no MFMAs, 2 no operand DMA, 3 DMA
// and barriers only, 5 every workgroup streams the SAME X rows (L2-resident), 6 plain (not XCD-aware) block order
template <int TN, int WM, int RT, int AB = 0>
__global__ __launch_bounds__(WM * 128) void gemm_split_f16_kernel(GemmParams p, int nrb, int ncb) {
    constexpr int BM = 32 * WM * RT, BN = 64 * TN, NT = WM * 128, NWV = NT / 64;
    constexpr int kSlot = (BM + BN) * 128;                       // bytes per ring slot: X rows then W rows, 128 B each
    // ring depth: as many slots as fit; small tiles stop at three (72 KB) so that TWO workgroups share a CU - their phases
    // (prologue fetch, K loop, epilogue stores) then overlap, which is what the short 6,400-row launches lack
    constexpr int S = kSlot <= 24 * 1024 ? 3 : (4 * kSlot <= 128 * 1024 ? 4 : (3 * kSlot <= 152 * 1024 ? 3 : 2));
    constexpr int IPL = (BM + BN) / 8 / NWV;
    static_assert((BM + BN) / 8 % NWV == 0 && (S - 1) * IPL <= 63, "DMA split / vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    int rb, cb;
    {
        const int lin = blockIdx.x, full = AB == 6 ? 0 : (nrb / 8) * 8 * ncb;   // workgroups of the complete groups of 8 row blocks
        if (lin < full) {
            const int seq = lin >> 3;
            rb = (seq / ncb) * 8 + (lin & 7);
            cb = seq % ncb;
        } else {                                                  // ragged tail: plain order
            const int t = lin - full;
            rb = (AB == 6 ? 0 : (nrb / 8) * 8) + t / ncb;
            cb = t % ncb;
        }
    }
    const int m0 = rb * BM;
    const int nb0 = cb * BN;
    const uint16_t* A = static_cast<const uint16_t*>(p.A);
    const uint16_t* W = static_cast<const uint16_t*>(p.W);

    // this lane's part of DMA instruction j: row (8-row group wave*IPL + j, row lane>>3), physical LDS slot lane&7.
    // Kept as ONE 32-bit element offset per instruction (row, plane and k slot folded in; the launcher checks the range):
    // 64-bit pointers + k offsets cost 3 registers per instruction, which the 256 x 256 tile does not have.
    int src_off[IPL];
#pragma unroll
    for (int j = 0; j < IPL; ++j) {
        const int r = (wave * IPL + j) * 8 + (lane >> 3);       // row of the concatenated [X; W] tile
        const int rr = r < BM ? r : r - BM;
        const int ls = (lane & 7) ^ ((rr >> 1) & 7);             // logical slot: plane ls >> 2, k offset 8 (ls & 3)
        if (r < BM) {
            int row = m0 + r < p.M ? m0 + r : p.M - 1;           // rows past the end: a valid row, never stored
            if constexpr (AB == 5) row = r;
            src_off[j] = (int)((int64_t)row * p.lda + (ls >> 2) * p.a_plane) + (ls & 3) * 8;
        } else {
            const int n = nb0 + rr < p.N ? nb0 + rr : p.N - 1;
            src_off[j] = (int)((int64_t)n * p.ldw + (ls >> 2) * p.w_plane) + (ls & 3) * 8;
        }
    }
    auto issue = [&](int kt) {
        if constexpr (AB == 2) return;           // probe: no operand traffic
        char* slot = smem_raw + (kt % S) * kSlot + wave * (IPL * 1024);
#pragma unroll
        for (int j = 0; j < IPL; ++j) {
            const bool isx = (wave * IPL + j) * 8 < BM;                        // wave-uniform: whole 8-row groups
            const int rr = (wave * IPL + j) * 8 + (lane >> 3) - (isx ? 0 : BM);
            const int k = kt * 32 + ((((lane & 7) ^ ((rr >> 1) & 7)) & 3) << 3);
            const uint16_t* src = k < p.K ? (isx ? A : W) + src_off[j] + kt * 32 : g_zero16;
            if constexpr (AB == 9) {   // probe (WRONG data): the same bytes as whole 128-B lines - 8 lanes of a row contiguous
                const int r = (wave * IPL + j) * 8 + (lane >> 3);
                const int rowi = isx ? (m0 + r < p.M ? m0 + r : p.M - 1) : (nb0 + rr < p.N ? nb0 + rr : p.N - 1);
                src = (isx ? A + (int64_t)rowi * p.lda : W + (int64_t)rowi * p.ldw) + kt * 64 + (((lane & 7) ^ ((rr >> 1) & 7)) << 3);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(slot + j * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[RT][TN];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;

    const int nk = AB == 8 ? 0 : (p.K + 31) / 32;   // (probe 8: epilogue only)
#pragma unroll 1
    for (int kt = 0; kt < nk && kt < S - 1; ++kt) issue(kt);

    // fragment byte offsets inside a slot: group g = 2 * plane + k-step, logical slot 2g + h of row l31
    uint32_t xoff[4], woff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t sw = (uint32_t)(((2 * g + h) ^ ((l31 >> 1) & 7)) << 4);
        xoff[g] = (wm * 32 * RT + l31) * 128 + sw;
        woff[g] = (BM + wn * 32 * TN + l31) * 128 + sw;
    }
    [[maybe_unused]] uint64_t st_t[6] = {0, 0, 0, 0, 0, 0}, st_a = 0, st_b = 0, st_0 = 0;
    if constexpr (AB == 7) st_0 = __builtin_readcyclecounter();
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (AB == 7) st_a = __builtin_readcyclecounter();
        const int after = (nk - 1 - kt) < (S - 2) ? (nk - 1 - kt) : (S - 2);   // younger chunks this wave has in flight
        if (S >= 4 && after >= 2) vm_wait<2 * IPL>();
        else if (S >= 3 && after == 1) vm_wait<IPL>();
        else vm_wait<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // publishes chunk kt; every wave is done with chunk kt-1
        asm volatile("" ::: "memory");
        if constexpr (AB == 7) { st_b = __builtin_readcyclecounter(); st_t[0] += st_b - st_a; st_a = st_b; }
        const uint32_t sl = lds_addr(smem_raw) + (uint32_t)((kt % S) * kSlot);
        if constexpr (AB == 3 || AB == 9) {      // probe: operand traffic only
            if (kt + S - 1 < nk) issue(kt + S - 1);
            continue;
        }
        bf16x8 fr[4][RT + TN];                   // [group][X row tile 0 .. RT-1, W tile 0 .. TN-1]
        auto rd = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            static_for<0, RT>([&](auto rc) {
                constexpr int rt = decltype(rc)::value;
                lds_read_b128_asm<rt * 32 * 128>(fr[g][rt], sl + xoff[g]);
            });
            static_for<0, TN>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                lds_read_b128_asm<t * 32 * 128>(fr[g][RT + t], sl + woff[g]);
            });
        };
        rd(std::integral_constant<int, 0>{});    // hi, k-step 0
        rd(std::integral_constant<int, 2>{});    // lo, k-step 0
        rd(std::integral_constant<int, 1>{});    // hi, k-step 1
        rd(std::integral_constant<int, 3>{});    // lo, k-step 1
        if (kt + S - 1 < nk) issue(kt + S - 1);  // into the slot chunk kt-1 just left
        static_for<0, 2>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            if constexpr (ks == 0) lds_wait<2 * (RT + TN)>(); else lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AB == 7) { st_b = __builtin_readcyclecounter(); st_t[1 + 2 * ks] += st_b - st_a; st_a = st_b; __builtin_amdgcn_sched_barrier(0); }
            if constexpr (AB == 1) {             // probe: no MFMAs (the fragments must stay live until they have landed)
#pragma unroll
                for (int i = 0; i < RT + TN; ++i) {
                    keep_alive(fr[ks][i]);
                    keep_alive(fr[2 + ks][i]);
                }
                return;
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < TN; ++t)     // W lo · X hi
                    acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(fr[2 + ks][RT + t]), as_f16x8(fr[ks][rt]), acc[rt][t], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < TN; ++t)     // W hi · X lo
                    acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(fr[ks][RT + t]), as_f16x8(fr[2 + ks][rt]), acc[rt][t], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < TN; ++t)     // W hi · X hi
                    acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(fr[ks][RT + t]), as_f16x8(fr[ks][rt]), acc[rt][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AB == 7) { st_b = __builtin_readcyclecounter(); st_t[2 + 2 * ks] += st_b - st_a; st_a = st_b; __builtin_amdgcn_sched_barrier(0); }
        });
    }
    __syncthreads();   // the epilogue's transposition patches alias the ring
    if constexpr (AB == 7) { st_b = __builtin_readcyclecounter(); st_t[5] = st_b - st_0; st_a = st_b; }

    split_store_tiles<TN, RT>(p, smem_raw, acc, m0, nb0, wm, wn, wave, lane);
    if constexpr (AB == 7) {   // [workgroup][wave][8]: wait+barrier, reads ks0, mfma ks0, wait ks1, mfma ks1, main loop total, epilogue, start
        if (lane == 0 && p.ln_out) {
            uint64_t* o = static_cast<uint64_t*>(p.ln_out) + ((int64_t)blockIdx.x * NWV + wave) * 8;
            for (int i = 0; i < 6; ++i) o[i] = st_t[i];
            o[6] = __builtin_readcyclecounter() - st_a;
            o[7] = st_0;
        }
    }
}

template <int TN, int WM, int RT, int AB = 0>
int32_t launch_split(const GemmParams& p, hipStream_t s) {
    constexpr int BM = 32 * WM * RT, BN = 64 * TN;
    constexpr size_t slot = (size_t)(BM + BN) * 128;
    constexpr size_t lds_tiles = (slot <= 24 * 1024 ? 3 : (4 * slot <= 128 * 1024 ? 4 : (3 * slot <= 152 * 1024 ? 3 : 2))) * slot;
    constexpr size_t lds_epi = (size_t)WM * 2 * 2 * kStageBytes;
    constexpr size_t lds = lds_tiles > lds_epi ? lds_tiles : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    ISPK_RESERVE_LDS((&gemm_split_f16_kernel<TN, WM, RT, AB>), lds, "gemm_split");
    const int nrb = (p.M + BM - 1) / BM, ncb = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_split_f16_kernel<TN, WM, RT, AB>), dim3(nrb * ncb), dim3(WM * 128), lds, s, p, nrb, ncb);
    return ispk_launch_status();
}

}  // namespace

extern "C" int32_t ispk_split_f16(const float* x, int64_t ldx, uint16_t* hi, uint16_t* lo, int64_t ldy, int32_t rows,
                                  int32_t cols, ispk_stream_t stream) {
    ISPK_REQUIRE(x && hi && lo, ISPK_E_NULL, "split_f16: null pointer");
    ISPK_REQUIRE(rows >= 0 && cols >= 4 && cols % 4 == 0, ISPK_E_SHAPE, "split_f16: cols=%d must be a positive multiple of 4", cols);
    ISPK_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= cols && ldy >= cols, ISPK_E_ALIGN, "split_f16: strides must be multiples of 4");
    ISPK_REQUIRE(ispk_aligned(x, 16) && ispk_aligned(hi, 8) && ispk_aligned(lo, 8), ISPK_E_ALIGN, "split_f16: alignment");
    if (rows == 0) return 0;
    const int64_t n = (int64_t)rows * (cols / 4);
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, ldx, hi, lo, ldy, rows, cols / 4);
    return ispk_launch_status();
}

// Tile choice (TN * 100 + WM * 10 + RT = 64 TN features x 32 WM RT rows) by a small cost model fitted to sweeps of the
// nine variants over the 18 GEMM shapes of the B = 64 forward (tools/sweep_split_shapes.py): weighted by launches per
// forward its picks cost 2.79 ms where the per-shape best tiles cost 2.76 and the former divisibility rule 2.99.
// Per workgroup: a fixed start-up, K / 32 chunks that cost max(MFMA time, LDS-DMA time of the slot) plus a barrier, and the
// epilogue's share of the HBM write burst; the launch runs ~ workgroups / resident slots rounds of that - a ragged last
// round costs 0.4 + 0.6 of its fill, its workgroups having the chip to themselves - and is bounded below by the
// L2 -> LDS traffic of the whole grid.
extern "C" int32_t ispk_gemm_split_f16_tile(int32_t M, int32_t N, int32_t K) {
    if (const char* e = ispk_knob("ISPK_SPLIT_TILE")) return atoi(e);  // experiments only
    static const int kTiles[9] = {221, 241, 242, 321, 341, 342, 421, 441, 442};
    const double nk = (double)((K + 31) / 32);
    int best = 221;
    double best_us = 1e30;
    for (int tile : kTiles) {
        const int tn = tile / 100, wm = tile / 10 % 10, rt = tile % 10;
        if (64 * tn > (N + 63) / 64 * 64 + 64) continue;               // more than one idle 64-feature group per block
        const double bm = 32.0 * wm * rt, bn = 64.0 * tn;
        const double wgs = (double)((M + (int)bm - 1) / (int)bm) * (double)((N + (int)bn - 1) / (int)bn);
        const double slot = (bm + bn) * 128.0;
        const int ring = slot <= 24 * 1024 ? 3 : (4 * slot <= 128 * 1024 ? 4 : (3 * slot <= 152 * 1024 ? 3 : 2));
        const double occ = ring * slot <= 80 * 1024 ? 2.0 : 1.0;       // workgroups per CU by LDS
        const double eff = tile == 442 ? 0.6 : 0.7;                    // 256 x 256: 128 accumulator registers, fewer loads in flight
        const double mfma_us = bm * bn * 32.0 * 6.0 / 9.77e6 / eff, dma_us = slot / 40e3;
        const double chunk_us = (mfma_us > dma_us ? mfma_us : dma_us) * (occ > 1.0 ? 1.3 : 1.0) + 0.25;
        const double cap = 256.0 * occ;
        const double epi_us = (wgs < cap ? wgs : cap) * bm * bn * 4.0 / 4.5e6 + 1.0;
        const double wg_us = 3.0 + nk * chunk_us + epi_us;
        const double full = (double)(int64_t)(wgs / cap), frac = wgs / cap - full;
        double rounds = full + (frac > 0.0 ? 0.4 + 0.6 * frac : 0.0);
        if (full == 0.0) rounds = 1.0;
        const double traffic_us = wgs * (bm + bn) * (double)K * 4.0 / 9e6;
        const double us = rounds * wg_us > traffic_us ? rounds * wg_us : traffic_us;
        if (us < best_us) best_us = us, best = tile;
    }
    return best;
}

extern "C" int32_t ispk_gemm_split_f16(const uint16_t* A, int64_t lda, int64_t a_plane, const uint16_t* W, int64_t ldw,
                                       int64_t w_plane, void* C, int64_t ldc, int64_t c_plane, const float* bias,
                                       const float* resid, int64_t ldr, const uint8_t* mask, int32_t M, int32_t N, int32_t K,
                                       uint32_t flags, int32_t cols_per_batch, int64_t batch_stride, ispk_stream_t stream) {
    GemmParams p{A, lda, W, ldw, C, ldc, bias, resid, ldr, mask, M, N, K, flags, cols_per_batch, batch_stride};
    p.a_plane = a_plane;
    p.w_plane = w_plane;
    p.c_plane = c_plane;
    ISPK_REQUIRE(A && W && C, ISPK_E_NULL, "gemm_split: null A/W/C");
    ISPK_REQUIRE(M >= 0 && N >= 1 && K >= 1 && K % 8 == 0, ISPK_E_SHAPE, "gemm_split: bad shape M=%d N=%d K=%d (K %% 8)", M, N, K);
    ISPK_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && a_plane % 8 == 0 && w_plane % 8 == 0 && lda >= 1 && ldw >= K, ISPK_E_ALIGN,
                 "gemm_split: lda / ldw / plane offsets must be multiples of 8 (ldw >= K)");
    ISPK_REQUIRE(ispk_aligned(A, 16) && ispk_aligned(W, 16), ISPK_E_ALIGN, "gemm_split: A/W must be 16-byte aligned");
    ISPK_REQUIRE((int64_t)M * lda + a_plane + K < (int64_t)1 << 31 && (int64_t)N * ldw + w_plane + K < (int64_t)1 << 31 && a_plane >= 0 &&
                     w_plane >= 0, ISPK_E_SHAPE, "gemm_split: operands beyond 2^31 elements (32-bit in-kernel offsets)");
    ISPK_REQUIRE(!((flags & (ISPK_EP_MASK_ACC | ISPK_EP_MASK_OUT)) && !mask), ISPK_E_NULL, "gemm_split: mask flag without mask");
    ISPK_REQUIRE(!(flags & (ISPK_EP_OUT_BF16 | ISPK_EP_RESID_BF16 | ISPK_EP_BIAS_ROW | ISPK_EP_MASK_COL)), ISPK_E_UNSUPPORTED,
                 "gemm_split: bf16 output / residual, row bias and column mask are not built");
    ISPK_REQUIRE(!((flags & ISPK_EP_GELU) && (flags & ISPK_EP_SILU)), ISPK_E_UNSUPPORTED, "gemm_split: GELU and SILU together");
    ISPK_REQUIRE(N % 4 == 0 && (!bias || ispk_aligned(bias, 16)), ISPK_E_ALIGN, "gemm_split: N %% 4, 16-byte aligned bias");
    if (flags & ISPK_EP_ROWS_T) {
        ISPK_REQUIRE(cols_per_batch > 0 && M % cols_per_batch == 0 && !resid && ldc >= cols_per_batch && ispk_aligned(C, 4) &&
                         !(flags & (ISPK_EP_OUT_SPLIT | ISPK_EP_GELU | ISPK_EP_SILU | ISPK_EP_MASK_ACC)),
                     ISPK_E_UNSUPPORTED, "gemm_split: ROWS_T takes bias + MASK_OUT only, fp32 C, M %% cols_per_batch == 0");
    } else if (flags & ISPK_EP_OUT_SPLIT) {
        ISPK_REQUIRE(cols_per_batch <= 0 && !resid && N % 8 == 0 && ldc % 8 == 0 && c_plane % 8 == 0 && ispk_aligned(C, 16),
                     ISPK_E_UNSUPPORTED, "gemm_split: split output needs N, ldc, c_plane %% 8 == 0, no residual");
    } else {
        ISPK_REQUIRE(cols_per_batch <= 0 && ldc % 4 == 0 && ispk_aligned(C, 16) &&
                         (!resid || (ldr % 4 == 0 && ispk_aligned(resid, 16))),
                     ISPK_E_ALIGN, "gemm_split: fp32 C / resid rows must be 16-byte aligned");
    }
    if (M == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#ifdef ISPK_EXPERIMENTS
    if (const char* e = ispk_knob("ISPK_SPLIT_ABLATE")) {   // timing probes only: WRONG results
        const int ab = atoi(e), tile = ispk_gemm_split_f16_tile(M, N, K);
        if (const char* st = ispk_knob("ISPK_SPLIT_STAMPS")) p.ln_out = reinterpret_cast<void*>(strtoull(st, nullptr, 16));
#define ISPK_AB_CASE(T, TN_, WM_, RT_)                                                   \
        if (tile == T) {                                                                 \
            if (ab == 1) return launch_split<TN_, WM_, RT_, 1>(p, s);                    \
            if (ab == 2) return launch_split<TN_, WM_, RT_, 2>(p, s);                    \
            if (ab == 3) return launch_split<TN_, WM_, RT_, 3>(p, s);                    \
            if (ab == 5) return launch_split<TN_, WM_, RT_, 5>(p, s);                    \
            if (ab == 6) return launch_split<TN_, WM_, RT_, 6>(p, s);                    \
            if (ab == 7) return launch_split<TN_, WM_, RT_, 7>(p, s);                    \
            if (ab == 8) return launch_split<TN_, WM_, RT_, 8>(p, s);                    \
            if (ab == 9) return launch_split<TN_, WM_, RT_, 9>(p, s);                    \
        }
        ISPK_AB_CASE(441, 4, 4, 1) ISPK_AB_CASE(442, 4, 4, 2) ISPK_AB_CASE(341, 3, 4, 1) ISPK_AB_CASE(342, 3, 4, 2)
#undef ISPK_AB_CASE
    }
#endif
    switch (ispk_gemm_split_f16_tile(M, N, K)) {
        case 442: return launch_split<4, 4, 2>(p, s);
        case 342: return launch_split<3, 4, 2>(p, s);
        case 242: return launch_split<2, 4, 2>(p, s);
        case 441: return launch_split<4, 4, 1>(p, s);
        case 341: return launch_split<3, 4, 1>(p, s);
        case 241: return launch_split<2, 4, 1>(p, s);
        case 421: return launch_split<4, 2, 1>(p, s);
        case 321: return launch_split<3, 2, 1>(p, s);
    }
    return launch_split<2, 2, 1>(p, s);
}

namespace {

// ------------------------------------------------------------------------------------------------ attention on split terms
// ALiBi-biased multi-query attention (attend.py:49-122, embeddings.py:51-82, attention.py:128-152) with fp32-grade
// products at the fp16 MFMA rate.  q / k / v arrive as fp32 (the fused projection's [Q | K | V] rows); the decomposition is
// attn_f32_kernel's (attention.hip): a workgroup = one (batch item, 64-query tile) for ALL heads, wave = (head, 32-query
// half), Sᵀ = K·Qᵀ so that a lane owns one query's softmax row and the P accumulator is already the next product's operand.
//   * K / V tiles (64 keys) are split ONCE per workgroup while they are staged into LDS (hi and lo planes, row-major
//     [key][64] fp16, 128-B rows, XOR-swizzled 16-B slots as in attn_bf16_kernel: K by (row >> 1) & 7, V by
//     ((row >> 1) & 1) << 2) and shared by all 2 H waves;
//   * Q (pre-scaled by 1/8, exact) is split once per wave into 4 + 4 register fragments;
//   * Sᵀ = K hi Q hi + K hi Q lo + K lo Q hi: 12 MFMAs per 32 x 32 block; softmax in fp32 against a lazily raised reference
//     maximum (exp through v_exp_f32 on log2-domain arguments: relative error ~1e-6 on probabilities that are summed in fp32);
//   * P is split in registers (p in [0, 1]: no clamp) and Oᵀ += V hi P hi + V hi P lo + V lo P hi with V read through the
//     transposing ds_read_b64_tr_b16 in the accumulator's key order (guide T10): 12 MFMAs per block.
// Output: fp32 rows, or split planes for the out-projection GEMM.
constexpr int kSaKeys = 64;                         // keys per staged tile
constexpr int kSaPlane = kSaKeys * 128;             // bytes of one fp16 plane of one tile
constexpr int kSaBuf = 4 * kSaPlane;                // K hi, K lo, V hi, V lo

template <int OFF>
__device__ __forceinline__ void sa_read_tr16_b64(su32x2& dst, uint32_t lds_byte_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF) : "memory");
}

template <int MAXT, int NS, bool SPLIT_OUT>   // NS = ceil(2048 / threads): staged float4 per thread and tile
__global__ __launch_bounds__(MAXT) void attn_split_f16_kernel(const float* __restrict__ q, int64_t ldq,
                                                              const float* __restrict__ k, const float* __restrict__ v,
                                                              int64_t ldkv, const float* __restrict__ slopes,
                                                              const int64_t* __restrict__ key_len, void* __restrict__ out,
                                                              int64_t ldo, int64_t o_plane, int N, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = wave % H, qhalf = wave / H;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 64 + qhalf * 32;
    int klen = key_len ? (int)key_len[b] : N;
    klen = klen < 1 ? 1 : (klen > N ? N : klen);
    const float slope = slopes[head];
    const float ninf = -__builtin_huge_valf();
    constexpr float kLog2e = 1.4426950408889634f;

    // ---- Q fragments: lane (query l31, half h) holds head dims 16 ks + 8 h .. + 7 of k-step ks, scaled by 1/8, split
    const int qi = q0 + l31;
    const int qrow = qi < N ? qi : N - 1;
    f16x8 qh[4], ql[4];
    {
        const float* qp = q + ((int64_t)b * N + qrow) * ldq + head * 64 + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + ks * 16);
            const f32x4 c = *reinterpret_cast<const f32x4*>(qp + ks * 16 + 4);
            union { uint32_t u[4]; f16x8 f; } hh, ll;
            split_pair(a[0] * 0.125f, a[1] * 0.125f, hh.u[0], ll.u[0]);
            split_pair(a[2] * 0.125f, a[3] * 0.125f, hh.u[1], ll.u[1]);
            split_pair(c[0] * 0.125f, c[1] * 0.125f, hh.u[2], ll.u[2]);
            split_pair(c[2] * 0.125f, c[3] * 0.125f, hh.u[3], ll.u[3]);
            qh[ks] = hh.f;
            ql[ks] = ll.f;
        }
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    const float* kb = k + (int64_t)b * N * ldkv;
    const float* vb = v + (int64_t)b * N * ldkv;
    const int ntiles = (klen + kSaKeys - 1) / kSaKeys;

    // staging: the fp32 loads of tile t+1 are issued before tile t is computed; split + LDS writes happen after it
    f32x4 sreg[NS];
    auto stage_load = [&](int t, int round = 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int idx = tid + (round * NS + i) * nthreads;
            const int isv = idx >> 10, rem = idx & 1023;
            const int row = rem >> 4, c4 = (rem & 15) * 4;
            const int key = t * kSaKeys + row;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (idx < 2048 && key < N) val = *reinterpret_cast<const f32x4*>((isv ? vb : kb) + (int64_t)key * ldkv + c4);
            sreg[i] = val;
        }
    };
    auto stage_store = [&](int buf, int round = 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int idx = tid + (round * NS + i) * nthreads;
            const int isv = idx >> 10, rem = idx & 1023;
            const int row = rem >> 4, c4 = (rem & 15) * 4;
            if (idx < 2048) {
                const int sw = isv ? ((row >> 1) & 1) << 2 : (row >> 1) & 7;
                char* dst = smem_raw + buf * kSaBuf + isv * 2 * kSaPlane + row * 128 + (((c4 >> 3) ^ sw) << 4) + (c4 & 4) * 2;
                uint2 hh, ll;
                split_pair(sreg[i][0], sreg[i][1], hh.x, ll.x);
                split_pair(sreg[i][2], sreg[i][3], hh.y, ll.y);
                *reinterpret_cast<uint2*>(dst) = hh;
                *reinterpret_cast<uint2*>(dst + kSaPlane) = ll;
            }
        }
    };

    // per-lane LDS byte offsets inside a buffer.  K fragment of k-step ks: row l31 (+ 32 per block), logical slot 2 ks + h
    const uint32_t lbase = lds_addr(smem_raw);
    // ((2 ks + h) ^ sw) << 4 = ((h ^ sw) << 4) ^ (ks << 5): one register, the k-step is a constant XOR
    const uint32_t koff0 = l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4);
    // V transposing read (as attn_bf16_kernel): 16-lane group = (h, dim half dh); lane 4 qq + pp of the group points at key
    // row 4 h + qq, dims 4 pp .. 4 pp + 3 of the group's 16-dim block = logical slot 4 dt + 2 dh + (pp >> 1), byte 8 (pp & 1)
    const int qq = (lane & 15) >> 2, pp = lane & 3, dh = (lane >> 4) & 1;
    const uint32_t voff0 = 2 * kSaPlane + (4 * h + qq) * 128 + (((2 * dh + (pp >> 1)) ^ ((qq >> 1) << 2)) << 4) + 8 * (pp & 1);   // dim tile dt: ^ (dt << 6)

    // Softmax bookkeeping kept small, as in attn_bf16_kernel (the loop is otherwise VALU-bound: ~300 vector instructions
    // against 24 MFMAs per block with a plain online softmax):
    //   * the ALiBi bias enters the first score MFMA as its C operand: away from the diagonal block key - query has a fixed
    //     sign, so the bias is "a per-lane base -/+ slope * (register's key offset)" - the offsets are a constant vector,
    //     the base rides in the FMA that turns a score into the exp2 argument; only the key0 == q0 block pays for |.|;
    //   * m_ref is a LAZY reference maximum, raised (and O, l rescaled) only when a block exceeds it by more than 2^kLazy:
    //     probabilities stay <= 2^kLazy = 256 (inside fp16's range for the split), the true row maximum contributes >= 1;
    //   * keys beyond key_len are masked only in the one block that straddles it.
    constexpr float kLazy = 8.0f;
    const float nsl = -slope;                              // bias per unit of |key - query|, natural-log units
    float mref2 = 0.f;                                     // = -m_ref in exp2 units
    float l2a = 0.f, l2b = 0.f;                            // row sum, two chains

    // a tile is 2048 float4: ceil(2048 / (NS threads)) rounds of NS registers per thread (one round at the recipes' head
    // counts); round 0 of the next tile is prefetched under this tile's products
    const int rounds = (2048 + NS * nthreads - 1) / (NS * nthreads);
    for (int rd0 = 0; rd0 < rounds; ++rd0) {
        stage_load(0, rd0);
        stage_store(0, rd0);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) stage_load(t + 1);
#pragma unroll 1
        for (int kblk = 0; kblk < 2; ++kblk) {
            const int key0 = t * kSaKeys + kblk * 32;
            if (key0 >= klen) break;  // wave-uniform
            const uint32_t blk = lbase + (uint32_t)(buf * kSaBuf + kblk * 32 * 128);
            // ---- Sᵀ[key][query] (+ ALiBi bias through the C operand)
            f32x16 s;
            float base2;                                   // exp2 argument = fma(s, log2e, base2)
            const float d0 = (float)(key0 + 4 * h - qi);   // key - query of accumulator register 0
            if (key0 == q0) {                              // wave-uniform: the block that straddles the diagonal
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = fabsf(d0 + (float)((r & 3) + 8 * (r >> 2))) * nsl;
                base2 = mref2;
            } else {                                       // keys before (sgn = -1) or after the queries: |d| = sgn (d0 + off_r)
                const float sn = key0 < q0 ? slope : nsl;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = sn * (float)((r & 3) + 8 * (r >> 2));
                base2 = fmaf(sn * kLog2e, d0, mref2);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 kh, kl;
                lds_read_b128_asm<0>(kh, blk + (koff0 ^ (uint32_t)(ks << 5)));
                lds_read_b128_asm<kSaPlane>(kl, blk + (koff0 ^ (uint32_t)(ks << 5)));
                lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(kl), qh[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(kh), ql[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(kh), qh[ks], s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // V fragments of the first k-step of the second product: requested now, they land during the softmax
            su32x2 vh[2][2], vl[2][2];   // [dim tile][run]
            auto rdv = [&](auto sc) {
                constexpr int st = decltype(sc)::value;
                static_for<0, 4>([&](auto ic) {
                    constexpr int i4 = decltype(ic)::value, dt = i4 >> 1, run = i4 & 1;
                    sa_read_tr16_b64<(16 * st + 8 * run) * 128>(vh[dt][run], blk + (voff0 ^ (uint32_t)(dt << 6)));
                    sa_read_tr16_b64<(16 * st + 8 * run) * 128 + kSaPlane>(vl[dt][run], blk + (voff0 ^ (uint32_t)(dt << 6)));
                });
            };
            rdv(std::integral_constant<int, 0>{});
            if (key0 + 32 > klen) {   // the one block that straddles key_len (wave-uniform test)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    s[r] = key < klen ? s[r] : ninf;
                }
            }
            float bmax = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) bmax = fmaxf(fmaxf(bmax, s[r]), s[r + 1]);
            // this lane half's block maximum in exp2 units relative to m_ref (the base differs between the halves), then the row's
            bmax = fmaf(fmaxf(bmax, s[15]), kLog2e, base2);
            bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
            const bool first = key0 == 0;
            if (first || __builtin_amdgcn_ballot_w64(bmax > kLazy) != 0) {   // wave-uniform
                // raise the reference to this block's row maximum (block 0: set it), rescale what was accumulated
                const float delta = first ? bmax : fmaxf(bmax, 0.f);
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                mref2 -= delta;
                base2 -= delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
                l2a *= alpha;
                l2b *= alpha;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[2 * j] = __builtin_amdgcn_exp2f(fmaf(s[2 * j], kLog2e, base2));
                s[2 * j + 1] = __builtin_amdgcn_exp2f(fmaf(s[2 * j + 1], kLog2e, base2));
                l2a += s[2 * j];
                l2b += s[2 * j + 1];
            }
            // ---- P -> split B-operand fragments (k-step st = registers 8 st .. 8 st + 7), Oᵀ += Vᵀ P
            static_for<0, 2>([&](auto sc) {
                constexpr int st = decltype(sc)::value;
                union { uint32_t u[4]; f16x8 f; } ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) split_pair_nc(s[8 * st + 2 * e], s[8 * st + 2 * e + 1], ph.u[e], pl.u[e]);
                union { uint32_t u[4]; f16x8 f; } ah0, ah1, al0, al1;
                lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
                ah0.u[0] = vh[0][0][0]; ah0.u[1] = vh[0][0][1]; ah0.u[2] = vh[0][1][0]; ah0.u[3] = vh[0][1][1];
                ah1.u[0] = vh[1][0][0]; ah1.u[1] = vh[1][0][1]; ah1.u[2] = vh[1][1][0]; ah1.u[3] = vh[1][1][1];
                al0.u[0] = vl[0][0][0]; al0.u[1] = vl[0][0][1]; al0.u[2] = vl[0][1][0]; al0.u[3] = vl[0][1][1];
                al1.u[0] = vl[1][0][0]; al1.u[1] = vl[1][0][1]; al1.u[2] = vl[1][1][0]; al1.u[3] = vl[1][1][1];
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0.f, ph.f, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1.f, ph.f, o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0.f, pl.f, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1.f, pl.f, o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0.f, ph.f, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1.f, ph.f, o1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (st == 0) rdv(std::integral_constant<int, 1>{});   // (the MFMAs above have read their operands)
            });
        }
        if (t + 1 < ntiles) {
            stage_store(buf ^ 1);
            for (int rd1 = 1; rd1 < rounds; ++rd1) {
                stage_load(t + 1, rd1);
                stage_store(buf ^ 1, rd1);
            }
        }
        __syncthreads();
    }
    const float l_run = l2a + l2b;

    // ---- normalise and store: lane (query, half) holds d = tile * 32 + (r & 3) + 8 (r >> 2) + 4 h
    const float lsum = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / lsum;
    if (qi < N) {
        if constexpr (SPLIT_OUT) {
            uint16_t* op = static_cast<uint16_t*>(out) + ((int64_t)b * N + qi) * ldo + head * 64 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 h0, l0, h1, l1;
                split_pair(o0[4 * g] * inv, o0[4 * g + 1] * inv, h0.x, l0.x);
                split_pair(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv, h0.y, l0.y);
                split_pair(o1[4 * g] * inv, o1[4 * g + 1] * inv, h1.x, l1.x);
                split_pair(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv, h1.y, l1.y);
                *reinterpret_cast<uint2*>(op + 8 * g) = h0;
                *reinterpret_cast<uint2*>(op + 32 + 8 * g) = h1;
                *reinterpret_cast<uint2*>(op + o_plane + 8 * g) = l0;
                *reinterpret_cast<uint2*>(op + o_plane + 32 + 8 * g) = l1;
            }
        } else {
            float* op = static_cast<float*>(out) + ((int64_t)b * N + qi) * ldo + head * 64 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a, c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = o0[4 * g + e] * inv;
                    c[e] = o1[4 * g + e] * inv;
                }
                *reinterpret_cast<f32x4*>(op + 8 * g) = a;
                *reinterpret_cast<f32x4*>(op + 32 + 8 * g) = c;
            }
        }
    }
}

}  // namespace

extern "C" int32_t ispk_alibi_mqa_attn_split_f16(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                                 const float* slopes, const int64_t* key_len, void* out, int64_t ldo,
                                                 int64_t o_plane, int32_t B, int32_t N, int32_t H, ispk_stream_t stream) {
    ISPK_REQUIRE(q && k && v && slopes && out, ISPK_E_NULL, "attn_split: null pointer");
    ISPK_REQUIRE(B >= 0 && N >= 1 && H >= 1 && H <= 8, ISPK_E_SHAPE, "attn_split: bad shape B=%d N=%d H=%d (H <= 8)", B, N, H);
    ISPK_REQUIRE(B <= 65535, ISPK_E_SHAPE, "attn_split: B=%d exceeds the grid limit 65535", B);
    ISPK_REQUIRE(ldq >= H * 64 && ldo >= H * 64 && ldkv >= 64, ISPK_E_SHAPE, "attn_split: leading strides too small");
    ISPK_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && o_plane % 4 == 0, ISPK_E_ALIGN,
                 "attn_split: strides must be multiples of 4");
    ISPK_REQUIRE(ispk_aligned(q, 16) && ispk_aligned(k, 16) && ispk_aligned(v, 16) && ispk_aligned(out, 16), ISPK_E_ALIGN,
                 "attn_split: pointers must be 16-byte aligned");
    if (B == 0) return 0;
    if (H == 5 || H >= 7) {   // head counts the recipes do not use: two calls over head ranges (K / V are staged twice)
        const int h0 = H == 5 ? 4 : 6;
        const int64_t oe = o_plane != 0 ? 2 : 4;   // bytes per output element
        if (int32_t rc = ispk_alibi_mqa_attn_split_f16(q, ldq, k, v, ldkv, slopes, key_len, out, ldo, o_plane, B, N, h0, stream)) return rc;
        return ispk_alibi_mqa_attn_split_f16(q + h0 * 64, ldq, k, v, ldkv, slopes + h0, key_len,
                                             static_cast<char*>(out) + (int64_t)h0 * 64 * oe, ldo, o_plane, B, N, H - h0, stream);
    }
    constexpr size_t lds = (size_t)2 * kSaBuf;   // 64 KB
    dim3 grid((N + 63) / 64, B), block(2 * H * 64);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool so = o_plane != 0;
#define ISPK_SA_GO(MAXT_, NS_)                                                                                           \
    do {                                                                                                                 \
        if (so) {                                                                                                        \
            ISPK_RESERVE_LDS((&attn_split_f16_kernel<MAXT_, NS_, true>), lds, "attn_split");                             \
            hipLaunchKernelGGL((attn_split_f16_kernel<MAXT_, NS_, true>), grid, block, lds, st, q, ldq, k, v, ldkv,      \
                               slopes, key_len, out, ldo, o_plane, N, H);                                                \
        } else {                                                                                                         \
            ISPK_RESERVE_LDS((&attn_split_f16_kernel<MAXT_, NS_, false>), lds, "attn_split");                            \
            hipLaunchKernelGGL((attn_split_f16_kernel<MAXT_, NS_, false>), grid, block, lds, st, q, ldq, k, v, ldkv,     \
                               slopes, key_len, out, ldo, o_plane, N, H);                                                \
        }                                                                                                                \
    } while (0)
    if (H == 6) ISPK_SA_GO(768, 3);     // 12 waves: one staging round of 3 float4 per thread
    else ISPK_SA_GO(512, 4);            // H <= 4: 2 H waves; one round at H = 4, more with fewer threads
#undef ISPK_SA_GO
    return ispk_launch_status();
}
