// Backward kernels of the transformer stacks ("backward for attention / FFN / LN kernels"), fp32, gfx950.
// First correct cut: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere, operands fetched straight from global memory (L2
// resident at these sizes) in MFMA-fragment order, no LDS staging yet.  Every reduction has a fixed order (two-stage
// partial sums, no floating-point atomics), so gradients are reproducible run to run.
//
//   ispk_transpose_f32          weights for dX = dY . W  (the NT GEMM of gemm.hip wants W^T rows)
//   ispk_gemm_tn_f32            dW = dY^T . X            (reduction over the rows; split over row ranges, then summed)
//   ispk_layernorm_bwd_f32      normalization.py:20-31 backward (dx, d gamma, d beta), optional row mask
//   ispk_gelu_bwd_f32           d/du of the exact-erf GELU (layers.py:29)
//   ispk_alibi_mqa_attn_bwd_f32 attend.py:49-122 + embeddings.py:51-82 backward: dQ per head, dK / dV summed over the heads
//                               that share them, d log-slope
#include "common.h"
#include "dropout.h"

namespace {

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// Row of the 32x32 fp32 accumulator tile that register r of lane l holds (the column is l % 32); also the order in which
// a lane half walks the reduction index when an accumulator is fed back as an MFMA operand (step t <-> register t).
__device__ __forceinline__ int acc_row(int r, int hf) { return 8 * (r >> 2) + 4 * hf + (r & 3); }

// ------------------------------------------------------------------------------------------------ transpose
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                        int rows, int cols) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < rows && c0 + tx < cols) t[k][tx] = x[(int64_t)(r0 + k) * ldx + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < cols && r0 + tx < rows) y[(int64_t)(c0 + k) * ldy + r0 + tx] = t[tx][k];
}

// ------------------------------------------------------------------------------------------------ C = A^T . B  (over rows)
// A [M][N1], B [M][N2] row-major, C [N1][N2].  With the rows as the MFMA reduction index both operands are used in their
// natural layout: lane (c = l % 32, hf = l / 32) supplies A[m0 + 2t + hf][n1 + c] and B[m0 + 2t + hf][n2 + c] - no
// transposition anywhere.  A workgroup (2 x 2 waves) owns a 128 x 128 tile of C for one range of rows and stages 32-row
// chunks of both operands through LDS (16-byte global loads one chunk ahead, conflict-free 4-byte LDS reads); the ranges'
// partial tiles go to a workspace and a second kernel adds them in range order (deterministic; optional accumulation into
// C).  `mask` (uint8 per row, or NULL) zeroes rows of A.
constexpr int kTnRows = 32;          // rows (reduction steps) per LDS stage
constexpr int kTnLd = 160;           // floats per staged row: 128 + 32, so the two lane halves (rows 2t, 2t + 1) hit disjoint banks
constexpr int kTnStage = 2 * kTnRows * kTnLd;   // floats per stage: A chunk + B chunk

template <bool kMask>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                      int64_t ldb, float* __restrict__ part, int M, int N1, int N2,
                                                      int rows_per_split, const uint8_t* __restrict__ mask, int splits,
                                                      int64_t stride_a, int64_t stride_b) {
    // Staging: thread (r = tid / 32, q = tid % 32) fetches float4 q of rows r, r + 8, r + 16, r + 24 of the A chunk and of the
    // B chunk (512 contiguous bytes per row and wave half), one chunk ahead of the MFMAs, double-buffered in LDS.
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, c = l & 31, hf = l >> 5;
    const int n1 = blockIdx.x * 128, n2 = blockIdx.y * 128, wa = (wave >> 1) * 64, wb = (wave & 1) * 64;
    // blockIdx.z = batch item * splits + row range (batched form: independent products A_b^T B_b, e.g. one per utterance)
    const int bz = blockIdx.z / splits, split = blockIdx.z - bz * splits;
    A += (int64_t)bz * stride_a;
    B += (int64_t)bz * stride_b;
    if (kMask) mask += (int64_t)bz * M;
    const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    const int sr = tid >> 5, sq = (tid & 31) * 4;
    const bool a_ok = n1 + sq < N1, b_ok = n2 + sq < N2;     // (N1, N2 are multiples of 4: a float4 is in or out as a whole)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[4], rb[4];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + sr + 8 * j;
            const bool ok = m < m_end;
            float s = 1.f;
            if (kMask) s = (ok && mask[m]) ? 1.f : 0.f;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[j] = (ok && a_ok) ? *reinterpret_cast<const f32x4*>(A + (int64_t)m * lda + n1 + sq) : z;
            rb[j] = (ok && b_ok) ? *reinterpret_cast<const f32x4*>(B + (int64_t)m * ldb + n2 + sq) : z;
            if (kMask) ra[j] *= s;
        }
    };
    auto stash = [&](int buf) {
        float* sa = lds + buf * kTnStage;
        float* sb = sa + kTnRows * kTnLd;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4*>(sa + (sr + 8 * j) * kTnLd + sq) = ra[j];
            *reinterpret_cast<f32x4*>(sb + (sr + 8 * j) * kTnLd + sq) = rb[j];
        }
    };
    fetch(m_begin);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += kTnRows) {
        const bool more = m0 + kTnRows < m_end;
        if (more) fetch(m0 + kTnRows);
        const float* sa = lds + buf * kTnStage + wa + c;
        const float* sb = lds + buf * kTnStage + kTnRows * kTnLd + wb + c;
#pragma unroll
        for (int t = 0; t < kTnRows / 2; ++t) {
            const int row = (2 * t + hf) * kTnLd;
            const float a0 = sa[row], a1 = sa[row + 32], b0 = sb[row], b1 = sb[row + 32];
            acc[0][0] = mfma2(a0, b0, acc[0][0]);
            acc[0][1] = mfma2(a0, b1, acc[0][1]);
            acc[1][0] = mfma2(a1, b0, acc[1][0]);
            acc[1][1] = mfma2(a1, b1, acc[1][1]);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = part + (int64_t)blockIdx.z * N1 * N2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n2 + wb + 32 * j + c;
            if (col >= N2) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n1 + wa + 32 * i + acc_row(r, hf);
                if (row < N1) out[(int64_t)row * N2 + col] = acc[i][j][r];
            }
        }
}

// The same product with bf16 operands (the weight gradients of a step under autocast, recipes/default.yaml:56: the reference's
// Linear backward multiplies bf16 copies of dY and X): fp32 rows are rounded to bf16 on their way into LDS, the 32-row chunk of
// each operand is a [32][128] bf16 image with 256-byte rows, 16-byte chunk ch of row r at 256 r + 16 (ch ^ (((r & 3) << 2) |
// ((r >> 2) & 3))), and ds_read_b64_tr_b16 hands every lane 4 consecutive ROWS of its column - the k-major operand fragment
// of v_mfma_f32_32x32x16_bf16 without a transposing pass (both operands take rows 8g .. 8g+3 | 8g+4 .. 8g+7 of a 16-row
// step for lane half g: any k order is fine as long as A and B agree).  fp32 accumulation, the same workspace / ordered sum.
typedef uint32_t tn_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t tn_img_off(int row, int ch) { return 256u * row + 16u * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
template <int OFF>
__device__ __forceinline__ void tn_read_tr(tn_u32x2& dst, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
constexpr int kTnImg = kTnRows * 256;            // bytes of one operand's chunk image

// IN16: the operands are ALREADY bf16 in memory (activations an AMP step keeps in bf16 as their producers wrote them): half the
// operand bytes of a kernel that is bound by them (a 128 x 128 tile re-reads both operands once per tile of the other axis).
template <bool kMask, bool IN16 = false>
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const void* __restrict__ Av, int64_t lda, const void* __restrict__ Bv,
                                                           int64_t ldb, float* __restrict__ part, int M, int N1, int N2,
                                                           int rows_per_split, const uint8_t* __restrict__ mask, int splits,
                                                           int64_t stride_a, int64_t stride_b, int nz) {
    typedef typename std::conditional<IN16, uint16_t, float>::type in_t;
    const in_t* A = static_cast<const in_t*>(Av);
    const in_t* B = static_cast<const in_t*>(Bv);
    extern __shared__ __attribute__((aligned(16))) char ldsb[];      // [2 buffers][A image | B image]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, c = l & 31, hf = l >> 5;
    // XCD-aware order (1-D grid): workgroups go to the 8 XCDs round-robin in launch order, and the T tiles of one row range
    // read the SAME rows of A and B - so row range 8 g + (lin % 8) takes tiles (lin / 8) % T: a row range's tiles run on ONE
    // XCD and find its rows in that L2 after the first fetch (in plain order every L2 fetched every row range).
    const int tx = (N1 + 127) / 128, ty = (N2 + 127) / 128, T = tx * ty;
    int zz, tile;
    {
        const int lin = blockIdx.x, full = (nz / 8) * 8 * T;
        if (lin < full) {
            const int g = lin / (8 * T), r = lin - g * 8 * T;
            zz = g * 8 + (r & 7);
            tile = r >> 3;
        } else {
            const int t = lin - full;
            zz = (nz / 8) * 8 + t / T;
            tile = t % T;
        }
    }
    const int n1 = (tile % tx) * 128, n2 = (tile / tx) * 128, wa = (wave >> 1) * 64, wb = (wave & 1) * 64;
    const int bz = zz / splits, split = zz - bz * splits;
    A += (int64_t)bz * stride_a;
    B += (int64_t)bz * stride_b;
    if (kMask) mask += (int64_t)bz * M;
    const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    // staging roles.  fp32 operands: a thread takes 4 columns (one float4) of rows sr + 8 j; bf16 operands: 8 columns (16 bytes,
    // one whole image chunk) of rows sr + 16 j - 16-byte requests, half as many of them
    const int sr = IN16 ? tid >> 4 : tid >> 5, sq = IN16 ? (tid & 15) * 8 : (tid & 31) * 4;
    const bool a_ok = n1 + sq < N1, b_ok = n2 + sq < N2;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[IN16 ? 1 : 4], rb[IN16 ? 1 : 4];
    u32x4 qa[IN16 ? 2 : 1], qb[IN16 ? 2 : 1];     // IN16: eight bf16 per row piece, as loaded
    auto fetch = [&](int m0) {
        if constexpr (IN16) {
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = m0 + sr + 16 * j;
                const bool ok = m < m_end, keep = ok && (!kMask || mask[m]);
                // (N1, N2 multiples of 4: a chunk that straddles the edge is fetched as its first half)
                if (keep && n1 + sq + 8 <= N1) qa[j] = *reinterpret_cast<const u32x4*>(A + (int64_t)m * lda + n1 + sq);
                else if (keep && a_ok) { const uint2 t = *reinterpret_cast<const uint2*>(A + (int64_t)m * lda + n1 + sq); qa[j] = u32x4{t.x, t.y, 0u, 0u}; }
                else qa[j] = z;
                if (ok && n2 + sq + 8 <= N2) qb[j] = *reinterpret_cast<const u32x4*>(B + (int64_t)m * ldb + n2 + sq);
                else if (ok && b_ok) { const uint2 t = *reinterpret_cast<const uint2*>(B + (int64_t)m * ldb + n2 + sq); qb[j] = u32x4{t.x, t.y, 0u, 0u}; }
                else qb[j] = z;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + sr + 8 * j;
                const bool ok = m < m_end;
                float sc = 1.f;
                if (kMask) sc = (ok && mask[m]) ? 1.f : 0.f;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                ra[j] = (ok && a_ok) ? *reinterpret_cast<const f32x4*>(A + (int64_t)m * lda + n1 + sq) : z;
                rb[j] = (ok && b_ok) ? *reinterpret_cast<const f32x4*>(B + (int64_t)m * ldb + n2 + sq) : z;
                if (kMask) ra[j] *= sc;
            }
        }
    };
    auto pack2 = [](float lo, float hi) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v;
        v.x = lo; v.y = hi;
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
    };
    auto stash = [&](int buf) {       // fp32: float4 (columns sq .. sq+3) -> 8 bytes at chunk sq / 8, half (sq / 4) & 1
        char* sa = ldsb + buf * 2 * kTnImg;
        char* sb = sa + kTnImg;
        if constexpr (IN16) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t off = tn_img_off(sr + 16 * j, sq >> 3);
                *reinterpret_cast<u32x4*>(sa + off) = qa[j];
                *reinterpret_cast<u32x4*>(sb + off) = qb[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = sr + 8 * j;
                const uint32_t off = tn_img_off(row, sq >> 3) + 8 * ((sq >> 2) & 1);
                uint2 pa, pb;
                pa.x = pack2(ra[j][0], ra[j][1]); pa.y = pack2(ra[j][2], ra[j][3]);
                pb.x = pack2(rb[j][0], rb[j][1]); pb.y = pack2(rb[j][2], rb[j][3]);
                *reinterpret_cast<uint2*>(sa + off) = pa;
                *reinterpret_cast<uint2*>(sb + off) = pb;
            }
        }
    };
    // transposing reads: 16-lane group (half hf, 16-column block sub) takes the block of 4 rows x 16 columns whose lane
    // 4q + p addresses row r0 + q, columns 4p .. 4p+3, i.e. chunk c0 + (p >> 1), byte 8 (p & 1); lane i receives column i
    const int sub = (l >> 4) & 1, q = (l & 15) >> 2, pp = l & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)ldsb;
    // address of (16-row step ks, rows 8 hf + 4 run + q, 32-column tile at column cb): all tiles / steps / runs differ from
    // tile 0, step 0, run 0 by a row offset that changes the swizzle, so each gets its own address register (16)
    uint32_t aoff[2][2][2], boff[2][2][2];      // [step][run][tile]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int run = 0; run < 2; ++run)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 16 * ks + 8 * hf + 4 * run + q;
                aoff[ks][run][t] = tn_img_off(row, ((wa + 32 * t) >> 3) + 2 * sub + (pp >> 1)) + 8 * (pp & 1);
                boff[ks][run][t] = kTnImg + tn_img_off(row, ((wb + 32 * t) >> 3) + 2 * sub + (pp >> 1)) + 8 * (pp & 1);
            }
    fetch(m_begin);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += kTnRows) {
        const bool more = m0 + kTnRows < m_end;
        if (more) fetch(m0 + kTnRows);
        const uint32_t base = lds0 + buf * 2 * kTnImg;
        tn_u32x2 fa[2][2][2], fb[2][2][2];     // [step][tile][run]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int run = 0; run < 2; ++run) {
                    tn_read_tr<0>(fa[ks][t][run], base + aoff[ks][run][t]);
                    tn_read_tr<0>(fb[ks][t][run], base + boff[ks][run][t]);
                }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            union { uint32_t u[4]; bf16x8 f; } a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t].u[0] = fa[ks][t][0][0]; a[t].u[1] = fa[ks][t][0][1]; a[t].u[2] = fa[ks][t][1][0]; a[t].u[3] = fa[ks][t][1][1];
                b[t].u[0] = fb[ks][t][0][0]; b[t].u[1] = fb[ks][t][0][1]; b[t].u[2] = fb[ks][t][1][0]; b[t].u[3] = fb[ks][t][1][1];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].f, b[0].f, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].f, b[1].f, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].f, b[0].f, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].f, b[1].f, acc[1][1], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = part + (int64_t)zz * N1 * N2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n2 + wb + 32 * j + c;
            if (col >= N2) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n1 + wa + 32 * i + acc_row(r, hf);
                if (row < N1) out[(int64_t)row * N2 + col] = acc[i][j][r];
            }
        }
}

// The same product for bf16-stored operands WITHOUT a row mask, fills by LDS-DMA (global_load_lds_dwordx4: a wave instruction
// writes four 256-byte image rows, the XOR applied on the source side).  The register-staged kernel above moves every chunk
// through ds_write - 64 KB per round of a CU's sixteen waves at the ~79 B / clk the write path sustains (≈830 cycles) next to
// 128 KB of transposed reads (≈512) against 1,024 cycles of MFMAs: LDS-bound, and bound by its writes.  The DMA needs no
// registers and no store instructions; a four-stage ring (64 KB, two workgroups per CU) keeps three chunks in flight.
// Rows past the range and columns past N1 / N2 come from 16 zero bytes (a DMA cannot zero-fill); N1, N2 multiples of 8.
__device__ __attribute__((aligned(16))) const uint16_t g_tn_zero16[8] = {0, 0, 0, 0, 0, 0, 0, 0};
template <int kTnDmaStages>
__global__ __launch_bounds__(256) void gemm_tn_dma_kernel(const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ B,
                                                          int64_t ldb, float* __restrict__ part, int M, int N1, int N2,
                                                          int rows_per_split, int splits, int nz) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];      // [stages][A image | B image]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, c = l & 31, hf = l >> 5;
    const int tx = (N1 + 127) / 128, ty = (N2 + 127) / 128, T = tx * ty;
    int zz, tile;
    {
        const int lin = blockIdx.x, full = (nz / 8) * 8 * T;
        if (lin < full) {
            const int g = lin / (8 * T), r = lin - g * 8 * T;
            zz = g * 8 + (r & 7);
            tile = r >> 3;
        } else {
            const int t = lin - full;
            zz = (nz / 8) * 8 + t / T;
            tile = t % T;
        }
    }
    const int n1 = (tile % tx) * 128, n2 = (tile / tx) * 128, wa = (wave >> 1) * 64, wb = (wave & 1) * 64;
    const int split = zz % splits;
    const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    const int nchunks = (m_end - m_begin + kTnRows - 1) / kTnRows;
    // DMA roles: instruction i of a stage (16 of them, 4 per wave) fills image i >> 3 (A | B), rows 4 (i & 7) .. + 3; lane L
    // lands at row 4 (i & 7) + (L >> 4), 16-byte position L & 15, which holds logical chunk (L & 15) ^ swizzle(row)
    const int drow = l >> 4, dpos = l & 15;
    auto issue = [&](int chunk) {
        char* stage = ldsb + (chunk % kTnDmaStages) * 2 * kTnImg;
        const int m0 = m_begin + chunk * kTnRows;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave * 4 + j, isb = i >> 3, grp = i & 7, row = 4 * grp + drow, m = m0 + row;
            const int ch = dpos ^ (((row & 3) << 2) | ((row >> 2) & 3)), col = (isb ? n2 : n1) + 8 * ch;
            const bool ok = m < m_end && col < (isb ? N2 : N1);
            const uint16_t* src = ok ? (isb ? B + (int64_t)m * ldb : A + (int64_t)m * lda) + col : g_tn_zero16;
            char* dst = stage + isb * kTnImg + grp * 1024;                       // wave-uniform; lane L lands at + 16 L
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int sub = (l >> 4) & 1, q = (l & 15) >> 2, pp = l & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)ldsb;
    uint32_t aoff[2][2][2], boff[2][2][2];      // [step][run][tile]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int run = 0; run < 2; ++run)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 16 * ks + 8 * hf + 4 * run + q;
                aoff[ks][run][t] = tn_img_off(row, ((wa + 32 * t) >> 3) + 2 * sub + (pp >> 1)) + 8 * (pp & 1);
                boff[ks][run][t] = kTnImg + tn_img_off(row, ((wb + 32 * t) >> 3) + 2 * sub + (pp >> 1)) + 8 * (pp & 1);
            }
#pragma unroll 1
    for (int ch = 0; ch < kTnDmaStages - 1 && ch < nchunks; ++ch) issue(ch);
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        // chunk ch has landed once only this wave's younger DMAs are outstanding (4 per chunk); the barrier publishes it and
        // retires the slot chunk ch - 1 was read from, which the next DMA overwrites
        // (stages - 2 younger chunks of this wave may still be in flight)
        const int young = nchunks - 1 - ch < kTnDmaStages - 2 ? nchunks - 1 - ch : kTnDmaStages - 2;
        if (young >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (young == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ch + kTnDmaStages - 1 < nchunks) issue(ch + kTnDmaStages - 1);
        const uint32_t base = lds0 + (ch % kTnDmaStages) * 2 * kTnImg;
        tn_u32x2 fa[2][2][2], fb[2][2][2];     // [step][tile][run]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int run = 0; run < 2; ++run) {
                    tn_read_tr<0>(fa[ks][t][run], base + aoff[ks][run][t]);
                    tn_read_tr<0>(fb[ks][t][run], base + boff[ks][run][t]);
                }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            union { uint32_t u[4]; bf16x8 f; } a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t].u[0] = fa[ks][t][0][0]; a[t].u[1] = fa[ks][t][0][1]; a[t].u[2] = fa[ks][t][1][0]; a[t].u[3] = fa[ks][t][1][1];
                b[t].u[0] = fb[ks][t][0][0]; b[t].u[1] = fb[ks][t][0][1]; b[t].u[2] = fb[ks][t][1][0]; b[t].u[3] = fb[ks][t][1][1];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].f, b[0].f, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].f, b[1].f, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].f, b[0].f, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].f, b[1].f, acc[1][1], 0, 0, 0);
        }
    }
    float* out = part + (int64_t)zz * N1 * N2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n2 + wb + 32 * j + c;
            if (col >= N2) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n1 + wa + 32 * i + acc_row(r, hf);
                if (row < N1) out[(int64_t)row * N2 + col] = acc[i][j][r];
            }
        }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int splits, int64_t n, int cols,
                                                           float* __restrict__ C, int64_t ldc, int accumulate, int64_t stride_c) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    part += (int64_t)blockIdx.y * splits * n;       // blockIdx.y = batch item
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(int64_t)k * n + i];
    float* dst = C + (int64_t)blockIdx.y * stride_c + (i / cols) * ldc + (i % cols);
    *dst = accumulate ? *dst + s : s;
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// y = (x - mean) * rstd * gamma + beta (eps inside the root), optionally y *= mask[row] afterwards.
//   g = dy * mask * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  d gamma = sum_rows dy * mask * xhat;
//   d beta = sum_rows dy * mask.
// One wave per row at a time, NV float4s per lane (D = 128 * NV... D = 256 -> 1 float4 per lane, 384 -> 1.5: handled as
// D / 64 scalars per lane in column-interleaved order so that loads stay coalesced).  A workgroup of 4 waves walks 64 rows
// and leaves one partial (d gamma, d beta) pair; stage 2 adds the partials in order.
template <int NPL>   // floats per lane = D / 64
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                            int64_t lddy, const float* __restrict__ gamma,
                                                            const uint8_t* __restrict__ mask, float* __restrict__ dx,
                                                            int64_t lddx, int add_to_dx, float* __restrict__ part, int rows,
                                                            float eps) {
    constexpr int D = NPL * 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    float gm[NPL], dg[NPL], db[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        gm[k] = gamma ? gamma[l + 64 * k] : 1.f;
        dg[k] = 0.f;
        db[k] = 0.f;
    }
    const int row0 = blockIdx.x * 64;
    for (int rr = wave; rr < 64; rr += 4) {
        const int row = row0 + rr;
        if (row >= rows) break;
        const float mk = mask ? (mask[row] ? 1.f : 0.f) : 1.f;
        float xv[NPL], gv[NPL];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            xv[k] = x[(int64_t)row * ldx + l + 64 * k];
            gv[k] = dy[(int64_t)row * lddy + l + 64 * k] * mk;
            s += xv[k];
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s * (1.f / D);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            xv[k] -= mean;
            q += xv[k] * xv[k];
        }
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
        const float rstd = 1.f / sqrtf(q * (1.f / D) + eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            xv[k] *= rstd;                 // xhat
            dg[k] += gv[k] * xv[k];
            db[k] += gv[k];
            gv[k] *= gm[k];                // g
            c1 += gv[k];
            c2 += gv[k] * xv[k];
        }
        for (int off = 32; off > 0; off >>= 1) {
            c1 += __shfl_xor(c1, off, 64);
            c2 += __shfl_xor(c2, off, 64);
        }
        c1 *= (1.f / D);
        c2 *= (1.f / D);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const float v = rstd * (gv[k] - c1 - xv[k] * c2);
            float* d = dx + (int64_t)row * lddx + l + 64 * k;
            *d = add_to_dx ? *d + v : v;
        }
    }
    if (!part) return;
    __shared__ float red[4][2][D];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        red[wave][0][l + 64 * k] = dg[k];
        red[wave][1][l + 64 * k] = db[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int w = i / D, col = i % D;
        part[(int64_t)blockIdx.x * 2 * D + i] = (red[0][w][col] + red[1][w][col]) + (red[2][w][col] + red[3][w][col]);
    }
}

// The same backward with 16-byte accesses: TWO rows per wavefront (32 lanes each, NV4 = D / 128 float4 per lane), as the
// forward's layernorm_vec_kernel - the scalar kernel above moved 4 bytes per lane per instruction and ran the 32,768-row decoder
// norms at 2.5 TB/s (200 MB per launch with the accumulate-into-dx form).  Same arithmetic per element; the statistics' and
// the (d gamma, d beta) partial sums' orders differ from the scalar kernel's (fixed, deterministic).
template <int NV4, int NW = 8>   // NW waves per 64-row block: 8 (two blocks per CU = 16 waves keep more loads in flight than 4 did)
__global__ __launch_bounds__(NW * 64) void layernorm_bwd_vec_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                                int64_t lddy, const float* __restrict__ gamma,
                                                                const uint8_t* __restrict__ mask, float* __restrict__ dx,
                                                                int64_t lddx, int add_to_dx, float* __restrict__ part, int rows,
                                                                float eps, uint16_t* __restrict__ dx16, int64_t lddx16) {
    constexpr int D = NV4 * 128;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, l = lane & 31, hf = lane >> 5;
    auto hsum = [](float v) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
    f32x4 gm[NV4], dg[NV4], db[NV4];
#pragma unroll
    for (int c = 0; c < NV4; ++c) {
        gm[c] = gamma ? *reinterpret_cast<const f32x4*>(gamma + 4 * (l + 32 * c)) : f32x4{1.f, 1.f, 1.f, 1.f};
        dg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int row0 = blockIdx.x * 64;
    for (int rr = wave * 2 + hf; rr < 64; rr += 2 * NW) {
        const int row_raw = row0 + rr;
        const bool live = row_raw < rows;                         // (half-waves of the last block: computed on the last row, not stored)
        const int row = live ? row_raw : rows - 1;
        const float mk = mask ? (mask[row] ? 1.f : 0.f) : 1.f;
        f32x4 xv[NV4], gv[NV4], old[NV4];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NV4; ++c) {
            xv[c] = *reinterpret_cast<const f32x4*>(x + (int64_t)row * ldx + 4 * (l + 32 * c));
            gv[c] = *reinterpret_cast<const f32x4*>(dy + (int64_t)row * lddy + 4 * (l + 32 * c)) * mk;
            if (add_to_dx) old[c] = *reinterpret_cast<const f32x4*>(dx + (int64_t)row * lddx + 4 * (l + 32 * c));
            s += (xv[c][0] + xv[c][1]) + (xv[c][2] + xv[c][3]);
        }
        const float mean = hsum(s) * (1.f / D);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NV4; ++c) {
            xv[c] -= mean;
            q += (xv[c][0] * xv[c][0] + xv[c][1] * xv[c][1]) + (xv[c][2] * xv[c][2] + xv[c][3] * xv[c][3]);
        }
        const float rstd = 1.f / sqrtf(hsum(q) * (1.f / D) + eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int c = 0; c < NV4; ++c) {
            xv[c] *= rstd;                 // xhat
            if (live) {
                dg[c] += gv[c] * xv[c];
                db[c] += gv[c];
            }
            gv[c] *= gm[c];                // g
            c1 += (gv[c][0] + gv[c][1]) + (gv[c][2] + gv[c][3]);
            c2 += (gv[c][0] * xv[c][0] + gv[c][1] * xv[c][1]) + (gv[c][2] * xv[c][2] + gv[c][3] * xv[c][3]);
        }
        c1 = hsum(c1) * (1.f / D);
        c2 = hsum(c2) * (1.f / D);
        if (live) {
#pragma unroll
            for (int c = 0; c < NV4; ++c) {
                f32x4 v = (gv[c] - c1 - xv[c] * c2) * rstd;
                if (add_to_dx) v += old[c];
                *reinterpret_cast<f32x4*>(dx + (int64_t)row * lddx + 4 * (l + 32 * c)) = v;
                if (dx16) {      // the same rows as the bf16 operand the next dX GEMM and weight gradient take (an AMP step)
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2*>(dx16 + (int64_t)row * lddx16 + 4 * (l + 32 * c)) = pk;
                }
            }
        }
    }
    if (!part) return;
    __shared__ float red[2 * NW][2][D];   // [wave * 2 + half][d gamma | d beta][column]
#pragma unroll
    for (int c = 0; c < NV4; ++c) {
        *reinterpret_cast<f32x4*>(&red[wave * 2 + hf][0][4 * (l + 32 * c)]) = dg[c];
        *reinterpret_cast<f32x4*>(&red[wave * 2 + hf][1][4 * (l + 32 * c)]) = db[c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += NW * 64) {
        const int w = i / D, col = i % D;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * NW; ++k) t += red[k][w][col];
        part[(int64_t)blockIdx.x * 2 * D + i] = t;
    }
}

// d gamma / d beta: thread i of [2 * D] adds its column of the `nparts` partial rows, 8 interleaved running sums.
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ part, int nparts, int twoD,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta) {
    // 32 columns per workgroup, eight threads per column: thread (c, g) adds partial rows g, g + 8, ... with four running sums,
    // the eight results are added in g order.  (One thread per column - three workgroups for 768 columns - walked all 512
    // partial rows of a decoder-sized launch alone: 14 us per call, 26 calls per step.)
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5, i = blockIdx.x * 32 + c;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < twoD) {
        int p = g;
        for (; p + 24 < nparts; p += 32)
#pragma unroll
            for (int k = 0; k < 4; ++k) s4[k] += part[(int64_t)(p + 8 * k) * twoD + i];
        for (int k = 0; p < nparts; p += 8, ++k) s4[k] += part[(int64_t)p * twoD + i];
    }
    red[g][c] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    if (g == 0 && i < twoD) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[k][c];
        const int D = twoD / 2;
        if (i < D) {
            if (dgamma) dgamma[i] = v;
        } else if (dbeta) {
            dbeta[i - D] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ GELU backward
// With dropout (feedforward.py:35: Linear -> GELU -> Dropout -> Linear): a = gelu(u) * keep / (1 - p), and backward
// du = da * keep / (1 - p) * gelu'(u); keep = drop_keep(seed, element index).
template <bool IO16 = false, bool U16 = false>   // IO16: da and du are bf16 (both are only ever GEMM operands of an AMP step);
// U16: the pre-activation is bf16 too (under autocast the first Linear's output IS bf16)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const void* __restrict__ da, const void* __restrict__ u,
                                                       void* __restrict__ du, int64_t n4, uint32_t thresh, float inv_keep,
                                                       uint64_t seed,
        const uint64_t* __restrict__ seed_src) {
    seed = run_seed(seed, seed_src);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 x;
    if constexpr (U16) {
        const uint2 q = reinterpret_cast<const uint2*>(u)[i];
        x[0] = bf16_to_f32((uint16_t)(q.x & 0xffffu)); x[1] = bf16_to_f32((uint16_t)(q.x >> 16));
        x[2] = bf16_to_f32((uint16_t)(q.y & 0xffffu)); x[3] = bf16_to_f32((uint16_t)(q.y >> 16));
    } else {
        x = reinterpret_cast<const f32x4*>(u)[i];
    }
    f32x4 a;
    if constexpr (IO16) {
        const uint2 q = reinterpret_cast<const uint2*>(da)[i];
        a[0] = bf16_to_f32((uint16_t)(q.x & 0xffffu)); a[1] = bf16_to_f32((uint16_t)(q.x >> 16));
        a[2] = bf16_to_f32((uint16_t)(q.y & 0xffffu)); a[3] = bf16_to_f32((uint16_t)(q.y >> 16));
    } else {
        a = reinterpret_cast<const f32x4*>(da)[i];
    }
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float dgelu;
        if constexpr (IO16) {
            // bf16 in and out: erf by Abramowitz-Stegun 7.1.28 (|error| <= 3e-7) and v_exp_f32 (common.h) - two orders below the
            // rounding of du to bf16, at a third of libm's instructions (the kernel was issue-bound, not bandwidth-bound)
            dgelu = gelu_grad_fast(x[k]);
        } else {
            const float cdf = 0.5f * (1.f + erff(x[k] * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * x[k] * x[k]);
            dgelu = cdf + x[k] * pdf;
        }
        float g = a[k] * dgelu;
        if (thresh) g = drop_keep(seed, (uint32_t)(4 * i + k), thresh) ? g * inv_keep : 0.f;
        o[k] = g;
    }
    if constexpr (IO16) {
        uint2 pk;
        pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
        reinterpret_cast<uint2*>(du)[i] = pk;
    } else {
        reinterpret_cast<f32x4*>(du)[i] = o;
    }
}

// the training forward keeps the pre-activation u (for the line above), so its GELU is a pass of its own: a = gelu(u)
template <bool OUT16 = false, bool U16 = false>   // OUT16: a is bf16 (an AMP step keeps the activation its second Linear multiplies in bf16)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const void* __restrict__ u, void* __restrict__ a, int64_t n4,
                                                       uint32_t thresh, float inv_keep, uint64_t seed,
        const uint64_t* __restrict__ seed_src) {
    seed = run_seed(seed, seed_src);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 x;
    if constexpr (U16) {
        const uint2 q = reinterpret_cast<const uint2*>(u)[i];
        x[0] = bf16_to_f32((uint16_t)(q.x & 0xffffu)); x[1] = bf16_to_f32((uint16_t)(q.x >> 16));
        x[2] = bf16_to_f32((uint16_t)(q.y & 0xffffu)); x[3] = bf16_to_f32((uint16_t)(q.y >> 16));
    } else {
        x = reinterpret_cast<const f32x4*>(u)[i];
    }
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float g = OUT16 ? gelu_as28(x[k]) : gelu_erf(x[k]);      // bf16 output: the 3e-7 erf of common.h (the inference path's)
        if (thresh) g = drop_keep(seed, (uint32_t)(4 * i + k), thresh) ? g * inv_keep : 0.f;
        o[k] = g;
    }
    if constexpr (OUT16) {
        uint2 pk;
        pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
        reinterpret_cast<uint2*>(a)[i] = pk;
    } else {
        reinterpret_cast<f32x4*>(a)[i] = o;
    }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ out, int64_t n, uint32_t thresh, uint64_t seed,
        const uint64_t* __restrict__ seed_src) {
    seed = run_seed(seed, seed_src);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = drop_keep(seed, (uint32_t)i, thresh) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ attention backward
// Forward (attend.py:49-122, embeddings.py:51-82): S = Q K^T / 8 - slope_h |i - j|, keys j >= key_len masked, P = softmax
// over keys, O = P V; one K / V head shared by all H query heads.  Backward with dO:
//   delta_i = sum_d dO_id O_id;  dP = dO V^T;  dS = P (dP - delta);  dQ = dS K / 8;  dK = sum_h dS^T Q / 8;
//   dV = sum_h P^T dO;  d slope_h = sum_ij dS_ij (-|i - j|).
// A 32 x 64 operand tile is held as 32 registers per lane: lane (c = l % 32, hf = l / 32) keeps row c, columns
// 8a + 4hf + {0..3} (eight 16-byte loads).  Register t of that fragment is MFMA step t's operand - the reduction index
// then runs in the order acc_row(t, hf), the SAME order in which a lane holds the rows of an accumulator tile, so an
// accumulator (P, dS) feeds the next product as its B operand without moving (the trick the forward kernels use for P).
struct Frag { float v[32]; };

__device__ __forceinline__ Frag load_frag(const float* __restrict__ base, int64_t ld, int row, int nrows, int hf) {
    Frag f;
    const bool ok = row < nrows;
    const f32x4* p = reinterpret_cast<const f32x4*>(base + (int64_t)(ok ? row : 0) * ld + 4 * hf);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const f32x4 q = ok ? p[2 * a] : f32x4{0.f, 0.f, 0.f, 0.f};
        f.v[4 * a + 0] = q.x;
        f.v[4 * a + 1] = q.y;
        f.v[4 * a + 2] = q.z;
        f.v[4 * a + 3] = q.w;
    }
    return f;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

__device__ __forceinline__ f32x16 dot_frags(const Frag& a, const Frag& b) {
    f32x16 acc = zero16();
#pragma unroll
    for (int t = 0; t < 32; ++t) acc = mfma2(a.v[t], b.v[t], acc);
    return acc;
}

// acc[mt] += sum over the 32 reduction rows of rowsrc[row][c + 32 mt] * w[t]   (w: an accumulator tile, t <-> acc_row(t, hf))
__device__ __forceinline__ void acc_operand_update(f32x16 (&acc)[2], const float* __restrict__ rowsrc, int64_t ld, int row0, int nrows,
                                                   int c, int hf, const float (&w)[16]) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int row = row0 + acc_row(t, hf);
        const bool ok = row < nrows;
        const float* r = rowsrc + (int64_t)(ok ? row : 0) * ld;
        const float a0 = ok ? r[c] : 0.f, a1 = ok ? r[32 + c] : 0.f;
        acc[0] = mfma2(a0, w[t], acc[0]);
        acc[1] = mfma2(a1, w[t], acc[1]);
    }
}
// dQ kernel: one wave per (query tile of 32, head, batch item), transposed orientation S^T[key][query] (query on the lane).
// Pass 1: row maxima / sums -> LSE (kept, and written for the dK/dV kernel together with delta).  Pass 2: P, dP, dS,
// dQ^T += K^T dS^T, slope partial.
template <bool kDrop>
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ o,
                                                         const float* __restrict__ dout, int64_t ldo,
                                                         const float* __restrict__ slopes, const int64_t* __restrict__ key_len,
                                                         float* __restrict__ dqkv, float* __restrict__ lse,
                                                         float* __restrict__ delta, float* __restrict__ slope_part, int N,
                                                         int H, float scale, const float* __restrict__ lse_in, uint32_t thresh,
                                                         float inv_keep, uint64_t seed,
        const uint64_t* __restrict__ seed_src) {
    seed = run_seed(seed, seed_src);
    const int tile = blockIdx.x, h = blockIdx.y, b = blockIdx.z, l = threadIdx.x, c = l & 31, hf = l >> 5;
    const int ntiles = gridDim.x;
    const int klen = key_len ? (int)min((int64_t)N, max((int64_t)0, key_len[b])) : N;
    const int i = tile * 32 + c;
    const float* qb = qkv + (int64_t)b * N * ld;
    const float* kb = qb + H * 64;
    const float* vb = kb + 64;
    const float* ob = o + (int64_t)b * N * ldo + h * 64;
    const float* dob = dout + (int64_t)b * N * ldo + h * 64;
    const float slope = slopes[h];
    const Frag qf = load_frag(qb + h * 64, ld, i, N, hf);
    const Frag dof = load_frag(dob, ldo, i, N, hf);
    float dl = 0.f;
    {
        const Frag of = load_frag(ob, ldo, i, N, hf);
#pragma unroll
        for (int t = 0; t < 32; ++t) dl += of.v[t] * dof.v[t];
        dl += __shfl_xor(dl, 32, 64);
    }
    const int kt_end = (klen + 31) / 32;
    const uint32_t row_idx = (((uint32_t)b * H + h) * N + (uint32_t)(i < N ? i : 0)) * (uint32_t)N;   // dropout index of (i, 0)
    // pass 1 (skipped when the training forward kept the row statistics)
    float mx = -INFINITY, sum = 0.f;
    for (int kt = 0; kt < (lse_in ? 0 : kt_end); ++kt) {
        const Frag kf = load_frag(kb, ld, kt * 32 + c, N, hf);
        const f32x16 s = dot_frags(kf, qf);
        float sv[16], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = kt * 32 + acc_row(r, hf);
            sv[r] = j < klen ? s[r] * scale - slope * fabsf((float)(i - j)) : -INFINITY;
            tmax = fmaxf(tmax, sv[r]);
        }
        const float nm = fmaxf(mx, tmax);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) part += sv[r] == -INFINITY ? 0.f : expf(sv[r] - nm);
        sum = (mx == -INFINITY ? 0.f : sum * expf(mx - nm)) + part;
        mx = nm;
    }
    {   // merge the two lane halves of a query
        const float om = __shfl_xor(mx, 32, 64), os = __shfl_xor(sum, 32, 64);
        const float nm = fmaxf(mx, om);
        const float a = mx == -INFINITY ? 0.f : sum * expf(mx - nm), bb = om == -INFINITY ? 0.f : os * expf(om - nm);
        // add in half order (half 0 first) so that both lanes of a query hold identical bits
        sum = hf == 0 ? a + bb : bb + a;
        mx = nm;
    }
    float L = sum > 0.f ? mx + logf(sum) : INFINITY;   // no valid key: P = exp(s - inf) = 0
    if (lse_in) L = i < N ? lse_in[((int64_t)b * H + h) * N + i] : INFINITY;
    if (hf == 0 && i < N) {
        lse[((int64_t)b * H + h) * N + i] = L;
        delta[((int64_t)b * H + h) * N + i] = dl;
    }
    // pass 2
    f32x16 dq[2] = {zero16(), zero16()};
    float gs = 0.f;
    for (int kt = 0; kt < kt_end; ++kt) {
        const Frag kf = load_frag(kb, ld, kt * 32 + c, N, hf);
        const f32x16 s = dot_frags(kf, qf);
        const Frag vf = load_frag(vb, ld, kt * 32 + c, N, hf);
        const f32x16 dp = dot_frags(vf, dof);
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = kt * 32 + acc_row(r, hf);
            const float dist = fabsf((float)(i - j));
            const float p = (j < klen && i < N) ? expf(s[r] * scale - slope * dist - L) : 0.f;
            float dpr = dp[r];
            if constexpr (kDrop) dpr = drop_keep(seed, row_idx + (uint32_t)j, thresh) ? dpr * inv_keep : 0.f;   // through the dropout
            ds[r] = p * (dpr - dl);
            gs -= ds[r] * dist;
        }
        // dQ^T[d][i] += sum_j K[j][d] dS^T[j][i]: A = K^T, its reduction index j walked in accumulator-row order
        acc_operand_update(dq, kb, ld, kt * 32, N, c, hf, ds);
    }
    if (i < N) {
        float* dst = dqkv + ((int64_t)b * N + i) * ld + h * 64;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 w = {dq[mt][4 * a] * scale, dq[mt][4 * a + 1] * scale, dq[mt][4 * a + 2] * scale, dq[mt][4 * a + 3] * scale};
                *reinterpret_cast<f32x4*>(dst + 32 * mt + 8 * a + 4 * hf) = w;
            }
    }
    for (int off = 32; off > 0; off >>= 1) gs += __shfl_xor(gs, off, 64);
    if (l == 0) slope_part[((int64_t)h * gridDim.z + b) * ntiles + tile] = gs;
}

// Training forward (attention dropout, attend.py:118 / SDPA dropout_p): the dQ kernel's skeleton - pass 1 row statistics
// (kept for the backward), pass 2 P, dropout, O^T[d][i] += sum_j V[j][d] Pdrop^T[j][i] with the accumulator fed back as the
// operand.  fp32, one wave per (32 queries, head, batch item).
__global__ __launch_bounds__(64) void attn_train_fwd_kernel(const float* __restrict__ qkv, int64_t ld,
                                                            const float* __restrict__ slopes, const int64_t* __restrict__ key_len,
                                                            float* __restrict__ o, int64_t ldo, float* __restrict__ lse, int N, int H,
                                                            float scale, uint32_t thresh, float inv_keep, uint64_t seed,
        const uint64_t* __restrict__ seed_src) {
    seed = run_seed(seed, seed_src);
    const int tile = blockIdx.x, h = blockIdx.y, b = blockIdx.z, l = threadIdx.x, c = l & 31, hf = l >> 5;
    const int klen = key_len ? (int)min((int64_t)N, max((int64_t)0, key_len[b])) : N;
    const int i = tile * 32 + c;
    const float* qb = qkv + (int64_t)b * N * ld;
    const float* kb = qb + H * 64;
    const float* vb = kb + 64;
    const float slope = slopes[h];
    const Frag qf = load_frag(qb + h * 64, ld, i, N, hf);
    const int kt_end = (klen + 31) / 32;
    const uint32_t row_idx = (((uint32_t)b * H + h) * N + (uint32_t)(i < N ? i : 0)) * (uint32_t)N;
    float mx = -INFINITY, sum = 0.f;
    for (int kt = 0; kt < kt_end; ++kt) {
        const Frag kf = load_frag(kb, ld, kt * 32 + c, N, hf);
        const f32x16 s = dot_frags(kf, qf);
        float sv[16], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = kt * 32 + acc_row(r, hf);
            sv[r] = j < klen ? s[r] * scale - slope * fabsf((float)(i - j)) : -INFINITY;
            tmax = fmaxf(tmax, sv[r]);
        }
        const float nm = fmaxf(mx, tmax);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) part += sv[r] == -INFINITY ? 0.f : expf(sv[r] - nm);
        sum = (mx == -INFINITY ? 0.f : sum * expf(mx - nm)) + part;
        mx = nm;
    }
    {
        const float om = __shfl_xor(mx, 32, 64), os = __shfl_xor(sum, 32, 64);
        const float nm = fmaxf(mx, om);
        const float a = mx == -INFINITY ? 0.f : sum * expf(mx - nm), bb = om == -INFINITY ? 0.f : os * expf(om - nm);
        sum = hf == 0 ? a + bb : bb + a;
        mx = nm;
    }
    const float L = sum > 0.f ? mx + logf(sum) : INFINITY;
    if (hf == 0 && i < N) lse[((int64_t)b * H + h) * N + i] = L;
    f32x16 ot[2] = {zero16(), zero16()};
    for (int kt = 0; kt < kt_end; ++kt) {
        const Frag kf = load_frag(kb, ld, kt * 32 + c, N, hf);
        const f32x16 s = dot_frags(kf, qf);
        float pd[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = kt * 32 + acc_row(r, hf);
            float p = (j < klen && i < N) ? expf(s[r] * scale - slope * fabsf((float)(i - j)) - L) : 0.f;
            if (thresh) p = drop_keep(seed, row_idx + (uint32_t)j, thresh) ? p * inv_keep : 0.f;
            pd[r] = p;
        }
        acc_operand_update(ot, vb, ld, kt * 32, N, c, hf, pd);
    }
    if (i < N) {
        float* dst = o + ((int64_t)b * N + i) * ldo + h * 64;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int a = 0; a < 4; ++a)
                *reinterpret_cast<f32x4*>(dst + 32 * mt + 8 * a + 4 * hf) =
                    f32x4{ot[mt][4 * a], ot[mt][4 * a + 1], ot[mt][4 * a + 2], ot[mt][4 * a + 3]};
    }
}

// dK / dV kernel: one workgroup per (key tile of 32, batch item), one wave per head; a wave loops over the query tiles
// (S[query][key], key on the lane) with its head's sums in registers, then the H waves add their tiles into one LDS tile in
// head order (barrier between heads): no atomics, a fixed summation order, one write per element.
template <bool kDrop>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ dout,
                                                            int64_t ldo, const float* __restrict__ slopes,
                                                            const int64_t* __restrict__ key_len, const float* __restrict__ lse,
                                                            const float* __restrict__ delta, float* __restrict__ dqkv, int N,
                                                            int H, float scale, uint32_t thresh, float inv_keep, uint64_t seed,
        const uint64_t* __restrict__ seed_src) {
    seed = run_seed(seed, seed_src);
    __shared__ float red[2][2][16][64];     // [dk | dv][M tile][register][lane]
    const int kt = blockIdx.x, b = blockIdx.y, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, c = l & 31, hf = l >> 5;
    const int klen = key_len ? (int)min((int64_t)N, max((int64_t)0, key_len[b])) : N;
    const int j = kt * 32 + c;
    const float* qb = qkv + (int64_t)b * N * ld;
    const float* kb = qb + H * 64;
    const float* vb = kb + 64;
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    if (kt * 32 < klen) {
        const Frag kf = load_frag(kb, ld, j, N, hf);
        const Frag vf = load_frag(vb, ld, j, N, hf);
        const int qt_end = (N + 31) / 32;
        const float slope = slopes[h];
        const float* qh = qb + h * 64;
        const float* doh = dout + (int64_t)b * N * ldo + h * 64;
        const float* lh = lse + ((int64_t)b * H + h) * N;
        const float* dh = delta + ((int64_t)b * H + h) * N;
        for (int qt = 0; qt < qt_end; ++qt) {
            const Frag qf = load_frag(qh, ld, qt * 32 + c, N, hf);
            const f32x16 s = dot_frags(qf, kf);
            const Frag dof = load_frag(doh, ldo, qt * 32 + c, N, hf);
            const f32x16 dp = dot_frags(dof, vf);
            float p[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = qt * 32 + acc_row(r, hf);
                const bool ok = i < N && j < klen;
                const float L = i < N ? lh[i] : 0.f, dl = i < N ? dh[i] : 0.f;
                p[r] = ok ? expf(s[r] * scale - slope * fabsf((float)(i - j)) - L) : 0.f;
                if constexpr (kDrop) {
                    const bool keep = drop_keep(seed, (((uint32_t)b * H + h) * N + (uint32_t)(i < N ? i : 0)) * (uint32_t)N + (uint32_t)j, thresh);
                    ds[r] = p[r] * ((keep ? dp[r] * inv_keep : 0.f) - dl);
                    p[r] = keep ? p[r] * inv_keep : 0.f;          // dV takes the DROPPED probabilities
                } else {
                    ds[r] = p[r] * (dp[r] - dl);
                }
            }
            acc_operand_update(dv, doh, ldo, qt * 32, N, c, hf, p);
            acc_operand_update(dk, qh, ld, qt * 32, N, c, hf, ds);
        }
    }
    for (int hh = 0; hh < H; ++hh) {          // heads add in index order
        if (h == hh) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    red[0][mt][r][l] = hh == 0 ? dk[mt][r] : red[0][mt][r][l] + dk[mt][r];
                    red[1][mt][r][l] = hh == 0 ? dv[mt][r] : red[1][mt][r][l] + dv[mt][r];
                }
        }
        __syncthreads();
    }
    if (h == 0 && j < N) {
        float* dst = dqkv + ((int64_t)b * N + j) * ld + H * 64;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 wk = {red[0][mt][4 * a][l] * scale, red[0][mt][4 * a + 1][l] * scale, red[0][mt][4 * a + 2][l] * scale,
                            red[0][mt][4 * a + 3][l] * scale};
                f32x4 wv = {red[1][mt][4 * a][l], red[1][mt][4 * a + 1][l], red[1][mt][4 * a + 2][l], red[1][mt][4 * a + 3][l]};
                *reinterpret_cast<f32x4*>(dst + 32 * mt + 8 * a + 4 * hf) = wk;
                *reinterpret_cast<f32x4*>(dst + 64 + 32 * mt + 8 * a + 4 * hf) = wv;
            }
    }
}

// d log-slope_h = slope_h * sum of the (batch, tile) partials, in index order (the parameter is log-slope:
// slope = exp(learned_logslopes), embeddings.py:59-82)
__global__ __launch_bounds__(512) void slope_reduce_kernel(const float* __restrict__ part, int per_head, const float* __restrict__ slopes,
                                                           float* __restrict__ dlogslopes, int H) {
    const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;      // one wave per head; lane l adds partials l, l + 64, ... in order
    if (h >= H) return;
    float s = 0.f;
    for (int k = l; k < per_head; k += 64) s += part[(int64_t)h * per_head + k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (l == 0) dlogslopes[h] = s * slopes[h];
}

}  // namespace

extern "C" int32_t ispk_transpose_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t rows, int32_t cols,
                                      ispk_stream_t stream) {
    ISPK_REQUIRE(x && y, -1, "ispk_transpose_f32: null pointer");
    ISPK_REQUIRE(rows >= 1 && cols >= 1 && ldx >= cols && ldy >= rows, -2, "ispk_transpose_f32: bad shape %d x %d", rows, cols);
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, ldx, y, ldy, rows, cols);
    return ispk_launch_status();
}

static int32_t gemm_tn_launch(const void* A, int64_t lda, int64_t stride_a, const void* B, int64_t ldb, int64_t stride_b, float* C,
                              int64_t ldc, int64_t stride_c, int batch, int M, int N1, int N2, const uint8_t* row_mask,
                              int accumulate, float* workspace, int64_t workspace_floats, hipStream_t s, const char* who,
                              bool bf16_operands = false, bool in16 = false) {
    ISPK_REQUIRE(A && B && C && workspace, -1, "%s: null pointer", who);
    // (ldb < N2 is allowed: overlapping rows, i.e. the 5-tap windows of a padded convolution input)
    ISPK_REQUIRE(batch >= 1 && M >= 1 && N1 >= 4 && N2 >= 4 && N1 % 4 == 0 && N2 % 4 == 0 && lda >= N1 && ldb >= 4 && ldc >= N2 &&
                     lda % 4 == 0 && ldb % 4 == 0 && stride_a % 4 == 0 && stride_b % 4 == 0 && ispk_aligned(A, 16) &&
                     ispk_aligned(B, 16), -2,
                 "%s: bad shape batch=%d M=%d N1=%d N2=%d (N1, N2, leading dimensions and batch strides multiples of 4, 16-byte "
                 "aligned operands)", who, batch, M, N1, N2);
    const int64_t tile = (int64_t)N1 * N2;
    ISPK_REQUIRE(workspace_floats >= tile * batch, -3, "%s: workspace holds %lld floats, one partial per batch item needs %lld",
                 who, (long long)workspace_floats, (long long)(tile * batch));
    // row ranges: enough workgroups to fill the chip (>= 1024; the LDS-DMA kernel: ONE round of its two workgroups per CU - swept
    // in tools/sweep_tn.py: more row ranges only add partial sums to write and add up),
    // at least 64 rows each, bounded by the workspace
    const bool dma = bf16_operands && in16 && !row_mask && batch == 1 && N1 % 8 == 0 && N2 % 8 == 0;
    const int tiles = ((N1 + 127) / 128) * ((N2 + 127) / 128) * batch;
    int target = dma ? 512 : 1024;
    if (const char* e = ispk_knob("ISPK_TN_TARGET")) target = atoi(e);               // experiments only
    int64_t splits = dma ? (target / tiles > 0 ? target / tiles : 1) : (target + tiles - 1) / tiles;
    splits = splits < (M + 63) / 64 ? splits : (M + 63) / 64;
    splits = splits < workspace_floats / (tile * batch) ? splits : workspace_floats / (tile * batch);
    splits = splits < 1 ? 1 : (splits > 256 ? 256 : splits);
    int rows_per = (int)((M + splits - 1) / splits);
    rows_per = (rows_per + kTnRows - 1) / kTnRows * kTnRows;
    splits = (M + rows_per - 1) / rows_per;
    ISPK_REQUIRE(splits * batch <= 65535, -4, "%s: batch %d x %lld row ranges exceed the grid limit", who, batch, (long long)splits);
    const dim3 grid3((N1 + 127) / 128, (N2 + 127) / 128, (unsigned)(splits * batch));
    if (bf16_operands) {
        constexpr size_t lds16 = 2 * 2 * kTnImg;                  // 32 KB
        const int nz = (int)(splits * batch);
        const dim3 grid(grid3.x * grid3.y * grid3.z);             // 1-D: the kernel orders (row range, tile) itself
        if (dma) {
            int stages = 4;
            if (const char* e = ispk_knob("ISPK_TN_STAGES")) stages = atoi(e);       // experiments only
            const size_t lds_dma = (size_t)stages * 2 * kTnImg;                        // 64 KB: two workgroups per CU
            if (stages == 2)
                hipLaunchKernelGGL(gemm_tn_dma_kernel<2>, grid, dim3(256), lds_dma, s, static_cast<const uint16_t*>(A), lda,
                                   static_cast<const uint16_t*>(B), ldb, workspace, M, N1, N2, rows_per, (int)splits, nz);
            else if (stages == 4)
                hipLaunchKernelGGL(gemm_tn_dma_kernel<4>, grid, dim3(256), lds_dma, s, static_cast<const uint16_t*>(A), lda,
                                   static_cast<const uint16_t*>(B), ldb, workspace, M, N1, N2, rows_per, (int)splits, nz);
            else
                hipLaunchKernelGGL(gemm_tn_dma_kernel<3>, grid, dim3(256), lds_dma, s, static_cast<const uint16_t*>(A), lda,
                                   static_cast<const uint16_t*>(B), ldb, workspace, M, N1, N2, rows_per, (int)splits, nz);
        } else if (in16) {
            if (row_mask)
                hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, true>), grid, dim3(256), lds16, s, A, lda, B, ldb, workspace, M, N1, N2,
                                   rows_per, row_mask, (int)splits, stride_a, stride_b, nz);
            else
                hipLaunchKernelGGL((gemm_tn_bf16_kernel<false, true>), grid, dim3(256), lds16, s, A, lda, B, ldb, workspace, M, N1, N2,
                                   rows_per, row_mask, (int)splits, stride_a, stride_b, nz);
        } else if (row_mask)
            hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, false>), grid, dim3(256), lds16, s, A, lda, B, ldb, workspace, M, N1, N2, rows_per,
                               row_mask, (int)splits, stride_a, stride_b, nz);
        else
            hipLaunchKernelGGL((gemm_tn_bf16_kernel<false, false>), grid, dim3(256), lds16, s, A, lda, B, ldb, workspace, M, N1, N2, rows_per,
                               row_mask, (int)splits, stride_a, stride_b, nz);
        hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((tile + 255) / 256), batch), dim3(256), 0, s, workspace, (int)splits,
                           tile, N2, C, ldc, accumulate, stride_c);
        return ispk_launch_status();
    }
    constexpr size_t lds_bytes = 2 * kTnStage * sizeof(float);   // 80 KB: two workgroups per CU
    if (row_mask) {
        ISPK_RESERVE_LDS(gemm_tn_kernel<true>, lds_bytes, "ispk_gemm_tn_f32");
        hipLaunchKernelGGL(gemm_tn_kernel<true>, grid3, dim3(256), lds_bytes, s, static_cast<const float*>(A), lda,
                           static_cast<const float*>(B), ldb, workspace, M, N1, N2, rows_per,
                           row_mask, (int)splits, stride_a, stride_b);
    } else {
        ISPK_RESERVE_LDS(gemm_tn_kernel<false>, lds_bytes, "ispk_gemm_tn_f32");
        hipLaunchKernelGGL(gemm_tn_kernel<false>, grid3, dim3(256), lds_bytes, s, static_cast<const float*>(A), lda,
                           static_cast<const float*>(B), ldb, workspace, M, N1, N2, rows_per,
                           row_mask, (int)splits, stride_a, stride_b);
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((tile + 255) / 256), batch), dim3(256), 0, s, workspace, (int)splits,
                       tile, N2, C, ldc, accumulate, stride_c);
    return ispk_launch_status();
}

extern "C" int32_t ispk_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t M,
                                    int32_t N1, int32_t N2, const uint8_t* row_mask, int32_t accumulate, float* workspace,
                                    int64_t workspace_floats, ispk_stream_t stream) {
    return gemm_tn_launch(A, lda, 0, B, ldb, 0, C, ldc, 0, 1, M, N1, N2, row_mask, accumulate, workspace, workspace_floats,
                          reinterpret_cast<hipStream_t>(stream), "ispk_gemm_tn_f32");
}

extern "C" int32_t ispk_gemm_tn_bf16(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t M,
                                     int32_t N1, int32_t N2, const uint8_t* row_mask, int32_t accumulate, float* workspace,
                                     int64_t workspace_floats, ispk_stream_t stream) {
    return gemm_tn_launch(A, lda, 0, B, ldb, 0, C, ldc, 0, 1, M, N1, N2, row_mask, accumulate, workspace, workspace_floats,
                          reinterpret_cast<hipStream_t>(stream), "ispk_gemm_tn_bf16", true);
}

extern "C" int32_t ispk_gemm_tn_b16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int32_t M,
                                    int32_t N1, int32_t N2, const uint8_t* row_mask, int32_t accumulate, float* workspace,
                                    int64_t workspace_floats, ispk_stream_t stream) {
    ISPK_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, -2, "ispk_gemm_tn_b16: lda / ldb must be multiples of 8 (16-byte row pieces)");
    return gemm_tn_launch(A, lda, 0, B, ldb, 0, C, ldc, 0, 1, M, N1, N2, row_mask, accumulate, workspace, workspace_floats,
                          reinterpret_cast<hipStream_t>(stream), "ispk_gemm_tn_b16", true, true);
}

extern "C" int32_t ispk_gemm_tn_batched_f32(const float* A, int64_t lda, int64_t stride_a, const float* B, int64_t ldb,
                                            int64_t stride_b, float* C, int64_t ldc, int64_t stride_c, int32_t batch, int32_t M,
                                            int32_t N1, int32_t N2, const uint8_t* row_mask, int32_t accumulate, float* workspace,
                                            int64_t workspace_floats, ispk_stream_t stream) {
    return gemm_tn_launch(A, lda, stride_a, B, ldb, stride_b, C, ldc, stride_c, batch, M, N1, N2, row_mask, accumulate, workspace,
                          workspace_floats, reinterpret_cast<hipStream_t>(stream), "ispk_gemm_tn_batched_f32");
}

static int32_t layernorm_bwd_launch(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* gamma,
                                    const uint8_t* row_mask, float* dx, int64_t lddx, int32_t add_to_dx, float* dgamma, float* dbeta,
                                    float* workspace, int64_t workspace_floats, int64_t rows, int32_t dim, float eps,
                                    ispk_stream_t stream, uint16_t* dx16, int64_t lddx16) {
    ISPK_REQUIRE(x && dy && dx, -1, "ispk_layernorm_bwd_f32: null pointer");
    ISPK_REQUIRE(rows >= 1 && (dim == 256 || dim == 384) && ldx >= dim && lddy >= dim && lddx >= dim, -2,
                 "ispk_layernorm_bwd_f32: rows=%lld dim=%d (dim must be 256 or 384)", (long long)rows, dim);
    const int blocks = (int)((rows + 63) / 64);
    const bool want = dgamma || dbeta;
    ISPK_REQUIRE(!want || (workspace && workspace_floats >= (int64_t)blocks * 2 * dim), -3,
                 "ispk_layernorm_bwd_f32: workspace needs %lld floats", (long long)blocks * 2 * dim);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* part = want ? workspace : nullptr;
    const bool vec = ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ispk_aligned(x, 16) && ispk_aligned(dy, 16) &&
                     ispk_aligned(dx, 16) && (!gamma || ispk_aligned(gamma, 16));
    ISPK_REQUIRE(!dx16 || (vec && lddx16 % 4 == 0 && lddx16 >= dim && ispk_aligned(dx16, 8)), -4,
                 "ispk_layernorm_bwd_dual_f32: the bf16 copy needs 16-byte aligned fp32 operands and an 8-byte aligned copy");
    if (vec && dim == 384)
        hipLaunchKernelGGL((layernorm_bwd_vec_kernel<3, 8>), dim3(blocks), dim3(512), 0, s, x, ldx, dy, lddy, gamma, row_mask, dx, lddx,
                           add_to_dx, part, (int)rows, eps, dx16, lddx16);
    else if (vec)
        hipLaunchKernelGGL((layernorm_bwd_vec_kernel<2, 8>), dim3(blocks), dim3(512), 0, s, x, ldx, dy, lddy, gamma, row_mask, dx, lddx,
                           add_to_dx, part, (int)rows, eps, dx16, lddx16);
    else if (dim == 384)
        hipLaunchKernelGGL(layernorm_bwd_kernel<6>, dim3(blocks), dim3(256), 0, s, x, ldx, dy, lddy, gamma, row_mask, dx, lddx,
                           add_to_dx, part, (int)rows, eps);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, x, ldx, dy, lddy, gamma, row_mask, dx, lddx,
                           add_to_dx, part, (int)rows, eps);
    if (want)
        hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * dim + 31) / 32), dim3(256), 0, s, part, blocks, 2 * dim, dgamma,
                           dbeta);
    return ispk_launch_status();
}

extern "C" int32_t ispk_layernorm_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* gamma,
                                          const uint8_t* row_mask, float* dx, int64_t lddx, int32_t add_to_dx, float* dgamma,
                                          float* dbeta, float* workspace, int64_t workspace_floats, int64_t rows, int32_t dim,
                                          float eps, ispk_stream_t stream) {
    return layernorm_bwd_launch(x, ldx, dy, lddy, gamma, row_mask, dx, lddx, add_to_dx, dgamma, dbeta, workspace, workspace_floats, rows,
                                dim, eps, stream, nullptr, 0);
}

// ... with dx written a second time as bf16 rows (dx_bf16, ld_dx_bf16): the operand of the next dX GEMM / weight gradient
extern "C" int32_t ispk_layernorm_bwd_dual_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* gamma,
                                               const uint8_t* row_mask, float* dx, int64_t lddx, int32_t add_to_dx, float* dgamma,
                                               float* dbeta, float* workspace, int64_t workspace_floats, int64_t rows, int32_t dim,
                                               float eps, uint16_t* dx_bf16, int64_t ld_dx_bf16, ispk_stream_t stream) {
    ISPK_REQUIRE(dx_bf16, -1, "ispk_layernorm_bwd_dual_f32: null bf16 output");
    return layernorm_bwd_launch(x, ldx, dy, lddy, gamma, row_mask, dx, lddx, add_to_dx, dgamma, dbeta, workspace, workspace_floats, rows,
                                dim, eps, stream, dx_bf16, ld_dx_bf16);
}

extern "C" int32_t ispk_gelu_bwd_f32(const float* da, const float* u, float* du, int64_t n, float dropout_p, uint64_t seed,
                                     ispk_stream_t stream) {
    ISPK_REQUIRE(da && u && du, -1, "ispk_gelu_bwd_f32: null pointer");
    ISPK_REQUIRE(n >= 0 && n % 4 == 0 && ispk_aligned(da, 16) && ispk_aligned(u, 16) && ispk_aligned(du, 16), -2,
                 "ispk_gelu_bwd_f32: n must be a multiple of 4 and the arrays 16-byte aligned");
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, -3, "ispk_gelu_bwd_f32: dropout_p must be in [0, 1)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gelu_bwd_kernel<false>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       da, u, du, n / 4, drop_thresh(dropout_p), 1.0f / (1.0f - dropout_p), mix_seed(seed), ispk_seed_source());
    return ispk_launch_status();
}
extern "C" int32_t ispk_gelu_bwd_bf16(const uint16_t* da, const float* u, uint16_t* du, int64_t n, float dropout_p, uint64_t seed,
                                      ispk_stream_t stream) {
    ISPK_REQUIRE(da && u && du, -1, "ispk_gelu_bwd_bf16: null pointer");
    ISPK_REQUIRE(n >= 0 && n % 4 == 0 && ispk_aligned(da, 8) && ispk_aligned(u, 16) && ispk_aligned(du, 8), -2,
                 "ispk_gelu_bwd_bf16: n must be a multiple of 4, u 16-byte and da / du 8-byte aligned");
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, -3, "ispk_gelu_bwd_bf16: dropout_p must be in [0, 1)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gelu_bwd_kernel<true>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       da, u, du, n / 4, drop_thresh(dropout_p), 1.0f / (1.0f - dropout_p), mix_seed(seed), ispk_seed_source());
    return ispk_launch_status();
}

extern "C" int32_t ispk_gelu_f32(const float* u, float* a, int64_t n, float dropout_p, uint64_t seed, ispk_stream_t stream) {
    ISPK_REQUIRE(u && a, -1, "ispk_gelu_f32: null pointer");
    ISPK_REQUIRE(n >= 0 && n % 4 == 0 && ispk_aligned(u, 16) && ispk_aligned(a, 16), -2,
                 "ispk_gelu_f32: n must be a multiple of 4 and the arrays 16-byte aligned");
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, -3, "ispk_gelu_f32: dropout_p must be in [0, 1)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gelu_fwd_kernel<false>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       u, a, n / 4, drop_thresh(dropout_p), 1.0f / (1.0f - dropout_p), mix_seed(seed), ispk_seed_source());
    return ispk_launch_status();
}
extern "C" int32_t ispk_gelu_f32_bf16(const float* u, uint16_t* a, int64_t n, float dropout_p, uint64_t seed, ispk_stream_t stream) {
    ISPK_REQUIRE(u && a, -1, "ispk_gelu_f32_bf16: null pointer");
    ISPK_REQUIRE(n >= 0 && n % 4 == 0 && ispk_aligned(u, 16) && ispk_aligned(a, 8), -2,
                 "ispk_gelu_f32_bf16: n must be a multiple of 4, u 16-byte and a 8-byte aligned");
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, -3, "ispk_gelu_f32_bf16: dropout_p must be in [0, 1)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gelu_fwd_kernel<true>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       u, a, n / 4, drop_thresh(dropout_p), 1.0f / (1.0f - dropout_p), mix_seed(seed), ispk_seed_source());
    return ispk_launch_status();
}

// the pair with the pre-activation stored in bf16 as well (autocast: the first Linear's output is bf16)
extern "C" int32_t ispk_gelu_bf16(const uint16_t* u, uint16_t* a, int64_t n, float dropout_p, uint64_t seed, ispk_stream_t stream) {
    ISPK_REQUIRE(u && a, -1, "ispk_gelu_bf16: null pointer");
    ISPK_REQUIRE(n >= 0 && n % 4 == 0 && ispk_aligned(u, 8) && ispk_aligned(a, 8), -2, "ispk_gelu_bf16: n %% 4, 8-byte aligned arrays");
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, -3, "ispk_gelu_bf16: dropout_p must be in [0, 1)");
    if (n == 0) return 0;
    hipLaunchKernelGGL((gelu_fwd_kernel<true, true>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       u, a, n / 4, drop_thresh(dropout_p), 1.0f / (1.0f - dropout_p), mix_seed(seed), ispk_seed_source());
    return ispk_launch_status();
}

extern "C" int32_t ispk_gelu_bwd_b16(const uint16_t* da, const uint16_t* u, uint16_t* du, int64_t n, float dropout_p, uint64_t seed,
                                     ispk_stream_t stream) {
    ISPK_REQUIRE(da && u && du, -1, "ispk_gelu_bwd_b16: null pointer");
    ISPK_REQUIRE(n >= 0 && n % 4 == 0 && ispk_aligned(da, 8) && ispk_aligned(u, 8) && ispk_aligned(du, 8), -2,
                 "ispk_gelu_bwd_b16: n %% 4, 8-byte aligned arrays");
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, -3, "ispk_gelu_bwd_b16: dropout_p must be in [0, 1)");
    if (n == 0) return 0;
    hipLaunchKernelGGL((gelu_bwd_kernel<true, true>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       da, u, du, n / 4, drop_thresh(dropout_p), 1.0f / (1.0f - dropout_p), mix_seed(seed), ispk_seed_source());
    return ispk_launch_status();
}

extern "C" int32_t ispk_dropout_mask_u8(uint8_t* out, int64_t n, float dropout_p, uint64_t seed, ispk_stream_t stream) {
    ISPK_REQUIRE(out && n >= 0 && dropout_p >= 0.f && dropout_p < 1.f, -1, "ispk_dropout_mask_u8: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       out, n, drop_thresh(dropout_p), mix_seed(seed), ispk_seed_source());
    return ispk_launch_status();
}

static int32_t attn_train_launch(const float* qkv, int64_t ld_qkv, const float* slopes, const int64_t* key_len, float* o, int64_t ld_o,
                                 float* lse, int32_t B, int32_t N, int32_t H, float dropout_p, uint64_t seed, hipStream_t s,
                                 const char* who) {
    ISPK_REQUIRE(qkv && slopes && o && lse, -1, "%s: null pointer", who);
    ISPK_REQUIRE(B >= 1 && N >= 1 && H >= 1 && H <= 8 && ld_qkv >= H * 64 + 128 && ld_o >= H * 64 && ld_qkv % 4 == 0 &&
                     ld_o % 4 == 0 && B <= 65535, -2, "%s: bad shape B=%d N=%d H=%d", who, B, N, H);
    ISPK_REQUIRE(ispk_aligned(qkv, 16) && ispk_aligned(o, 16), -3, "%s: arrays must be 16-byte aligned", who);
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, -4, "%s: dropout_p must be in [0, 1)", who);
    const dim3 grid((N + 31) / 32, H, B);
    hipLaunchKernelGGL(attn_train_fwd_kernel, grid, dim3(64), 0, s, qkv, ld_qkv, slopes, key_len, o, ld_o, lse, N, H, 0.125f,
                       drop_thresh(dropout_p), 1.0f / (1.0f - dropout_p), mix_seed(seed), ispk_seed_source());
    return ispk_launch_status();
}

extern "C" int32_t ispk_alibi_mqa_attn_train_f32(const float* qkv, int64_t ld_qkv, const float* slopes, const int64_t* key_len,
                                                 float* o, int64_t ld_o, float* lse, int32_t B, int32_t N, int32_t H,
                                                 float dropout_p, uint64_t seed, ispk_stream_t stream) {
    return attn_train_launch(qkv, ld_qkv, slopes, key_len, o, ld_o, lse, B, N, H, dropout_p, seed,
                             reinterpret_cast<hipStream_t>(stream), "ispk_alibi_mqa_attn_train_f32");
}

template <bool kDrop>
static void attn_bwd_kernels(hipStream_t s, int tiles, const float* qkv, int64_t ld_qkv, const float* o, const float* d_o, int64_t ld_o,
                             const float* slopes, const int64_t* key_len, float* dqkv, float* lse, float* delta, float* spart, int B,
                             int N, int H, const float* lse_in, uint32_t thresh, float inv_keep, uint64_t seed) {
    const float scale = 0.125f;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<kDrop>), dim3(tiles, H, B), dim3(64), 0, s, qkv, ld_qkv, o, d_o, ld_o, slopes, key_len,
                       dqkv, lse, delta, spart, N, H, scale, lse_in, thresh, inv_keep, seed, ispk_seed_source());
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<kDrop>), dim3(tiles, B), dim3(64 * H), 0, s, qkv, ld_qkv, d_o, ld_o, slopes, key_len,
                       lse, delta, dqkv, N, H, scale, thresh, inv_keep, seed, ispk_seed_source());
}

extern "C" int32_t ispk_alibi_mqa_attn_bwd_f32(const float* qkv, int64_t ld_qkv, const float* o, const float* d_o, int64_t ld_o,
                                               const float* slopes, const int64_t* key_len, float* dqkv, float* dlogslopes,
                                               float* workspace, int64_t workspace_floats, int32_t B, int32_t N, int32_t H,
                                               const float* lse_in, float dropout_p, uint64_t seed, ispk_stream_t stream) {
    const char* who = "ispk_alibi_mqa_attn_bwd_f32";
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ISPK_REQUIRE(qkv && o && d_o && slopes && dqkv && workspace, -1, "%s: null pointer", who);
    ISPK_REQUIRE(B >= 1 && N >= 1 && H >= 1 && H <= 8 && ld_qkv >= H * 64 + 128 && ld_o >= H * 64 && ld_qkv % 4 == 0 &&
                     ld_o % 4 == 0, -2, "%s: bad shape B=%d N=%d H=%d", who, B, N, H);
    ISPK_REQUIRE(ispk_aligned(qkv, 16) && ispk_aligned(o, 16) && ispk_aligned(d_o, 16) && ispk_aligned(dqkv, 16), -3,
                 "%s: arrays must be 16-byte aligned", who);
    const int tiles = (N + 31) / 32;
    const int64_t stat = (int64_t)B * H * N, need = 2 * stat + (int64_t)H * B * tiles;
    ISPK_REQUIRE(workspace_floats >= need, -4, "%s: workspace needs %lld floats", who, (long long)need);
    float *lse = workspace, *delta = workspace + stat, *spart = workspace + 2 * stat;
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, -5, "%s: dropout_p must be in [0, 1)", who);
    const uint32_t thresh = drop_thresh(dropout_p);
    const float inv_keep = 1.0f / (1.0f - dropout_p);
    if (thresh)
        attn_bwd_kernels<true>(s, tiles, qkv, ld_qkv, o, d_o, ld_o, slopes, key_len, dqkv, lse, delta, spart, B, N, H, lse_in, thresh,
                               inv_keep, mix_seed(seed));
    else
        attn_bwd_kernels<false>(s, tiles, qkv, ld_qkv, o, d_o, ld_o, slopes, key_len, dqkv, lse, delta, spart, B, N, H, lse_in, thresh,
                                inv_keep, mix_seed(seed));
    if (dlogslopes)
        hipLaunchKernelGGL(slope_reduce_kernel, dim3(1), dim3(64 * H), 0, s, spart, B * tiles, slopes, dlogslopes, H);
    return ispk_launch_status();
}
