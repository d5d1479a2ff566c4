// ALiBi-biased multi-query attention for gfx950 — flash-style, nothing of size N x N ever touches HBM.
//
// Reference semantics: /root/reference/tts/modules/transformer/attend.py:49-122 (SDPA over a materialised
// [B,H,N,N] fp32 bias = slopes[h] * -|i-j| with masked keys filled with min/2), embeddings.py:51-72,
// attention.py:128-152.  One K/V head is shared by all H query heads (attend.py:63-67).
//
// Work decomposition: a workgroup owns one (batch item, 64-query tile) for ALL heads: 2*H waves, wave w handles head
// w % H and the 32-query half w / H.  The K/V tile (64 keys x (64 + 64) fp32) is staged ONCE in LDS and consumed by
// all 2*H waves — the MQA reuse the reference throws away by expanding K/V to H heads.
//
// fp32 path (v_mfma_f32_32x32x2_f32, exact fp32 products):
//   * Sᵀ = K·Qᵀ (keys on the MFMA row axis, queries on the lane axis).  A lane then holds, for ITS query, 16 keys of
//     the 32-key block in its accumulator registers, so the online-softmax row reductions are 16 in-register ops plus
//     one cross-half exchange, and the exponentiated Pᵀ accumulator is ALREADY the B operand of the next product
//     Oᵀ += Vᵀ·Pᵀ (register r of lane half h is key (r&3) + 8(r>>2) + 4h: the V row is simply read at that key).
//     P never goes through LDS and is never transposed.
//   * Q is held in registers (32 fp32 per lane: lane half h owns head dims 32h..32h+31), pre-scaled by 1/sqrt(64)
//     (a power of two, so bitwise equal to scaling the product).
//   * K rows are padded to 68 dwords so each ds_read_b128 lane group hits 16 distinct 4-bank slots.
//   * bias -slope*|i-j| and the key-length mask are computed in registers; fully masked key blocks are skipped.
#include "common.h"

namespace {

constexpr int kLdk = 68;   // padded K/V tile row (64 + 4 dwords)
constexpr int kTileKeys = 64;

__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

template <int MAXT>  // 768 (H <= 6: 3 waves per SIMD, 168 VGPRs) or 1024 (H = 7, 8)
__global__ __launch_bounds__(MAXT) void attn_f32_kernel(const float* __restrict__ q, int64_t ldq,
                                                        const float* __restrict__ k, const float* __restrict__ v,
                                                        int64_t ldkv, const float* __restrict__ slopes,
                                                        const int64_t* __restrict__ key_len, float* __restrict__ out,
                                                        int64_t ldo, int N, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);  // [2][64][kLdk]
    float* Vs = Ks + 2 * kTileKeys * kLdk;           // [2][64][kLdk]

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

    // ---- Q fragment: lane (query l31, half h) holds Q[q][32h .. 32h+31] / 8
    const int qi = q0 + l31;
    const int qrow = qi < N ? qi : N - 1;
    f32x4 qf[8];
    {
        const float* qp = q + ((int64_t)b * N + qrow) * ldq + head * 64 + h * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            qf[c] = *reinterpret_cast<const f32x4*>(qp + c * 4);
            qf[c] *= 0.125f;
        }
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = ninf, l_run = 0.f;

    const float* kb = k + (int64_t)b * N * ldkv;
    const float* vb = v + (int64_t)b * N * ldkv;
    const int ntiles = (klen + kTileKeys - 1) / kTileKeys;

    // K/V tile staging is split (guide T14): the global loads of tile t+1 are ISSUED before tile t is computed and their
    // LDS writes happen after it, so a tile's HBM/L2 latency hides under the previous tile's MFMAs instead of stalling
    // every wave at the top of each tile.  2048 float4 per tile over <= 1024 threads: 3 (at 768 threads) per thread.
    constexpr int kStageMax = 4;  // ceil(2048 / 512): enough down to 8 waves (H = 4)
    f32x4 sreg[kStageMax];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int i = 0; i < kStageMax; ++i) {
            const int idx = tid + i * nthreads;
            const int isv = idx >> 10, rem = idx & 1023;
            const int row = rem >> 4, c4 = (rem & 15) * 4;
            const int key = t * kTileKeys + row;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (idx < kTileKeys * 32 && key < N)
                val = *reinterpret_cast<const f32x4*>((isv ? vb : kb) + (int64_t)key * ldkv + c4);
            sreg[i] = val;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kStageMax; ++i) {
            const int idx = tid + i * nthreads;
            const int isv = idx >> 10, rem = idx & 1023;
            const int row = rem >> 4, c4 = (rem & 15) * 4;
            if (idx < kTileKeys * 32)
                *reinterpret_cast<f32x4*>((isv ? Vs : Ks) + (buf * kTileKeys + row) * kLdk + c4) = sreg[i];
        }
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) stage_load(t + 1);
#pragma unroll 1
        for (int kblk = 0; kblk < 2; ++kblk) {
            const int key0 = t * kTileKeys + kblk * 32;
            if (key0 >= klen) break;  // wave-uniform
            // ---- Sᵀ[key][query] = sum_d K[key][d] * Q[query][d]
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const float* kp = Ks + (buf * kTileKeys + kblk * 32 + l31) * kLdk + h * 32;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[c][e], s, 0, 0, 0);
            }
            // ---- bias, mask, online softmax (per query = per lane pair)
            float smax = ninf;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int dist = key > qi ? key - qi : qi - key;
                float val = s[r] - slope * (float)dist;
                val = key < klen ? val : ninf;
                s[r] = val;
                smax = fmaxf(smax, val);
            }
            smax = xhalf_max(smax);
            const float m_new = fmaxf(m_run, smax);
            const float alpha = expf(m_run - m_new);  // first block: exp(-inf) = 0
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pexp = expf(s[r] - m_new);
                s[r] = pexp;
                psum += pexp;
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0[r] *= alpha;
                o1[r] *= alpha;
            }
            // ---- Oᵀ[d][query] += sum_key V[key][d] * P[key][query]
            const float* vp = Vs + (buf * kTileKeys + kblk * 32 + 4 * h) * kLdk + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vr = vp + ((r & 3) + 8 * (r >> 2)) * kLdk;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], s[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], s[r], o1, 0, 0, 0);
            }
        }
        if (t + 1 < ntiles) stage_store(buf ^ 1);
        __syncthreads();
    }

    // ---- normalise and store: lane (query, half) holds d = dblk*32 + (r&3) + 8(r>>2) + 4h
    const float inv = 1.0f / xhalf_sum(l_run);
    if (qi < N) {
        float* op = out + ((int64_t)b * N + qi) * ldo + head * 64 + 4 * h;
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

}  // namespace

extern "C" int32_t ispk_alibi_mqa_attn_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                           const float* slopes, const int64_t* key_len, float* out, int64_t ldo,
                                           int32_t B, int32_t N, int32_t H, ispk_stream_t stream) {
    ISPK_REQUIRE(q && k && v && slopes && out, ISPK_E_NULL, "attn: null pointer");
    ISPK_REQUIRE(B >= 0 && N >= 1 && H >= 1 && H <= 8, ISPK_E_SHAPE, "attn: bad shape B=%d N=%d H=%d (H <= 8)", B, N, H);
    ISPK_REQUIRE(B <= 65535, ISPK_E_SHAPE, "attn: B=%d exceeds the grid limit 65535", B);
    ISPK_REQUIRE(ldq >= H * 64 && ldo >= H * 64 && ldkv >= 64, ISPK_E_SHAPE, "attn: leading strides too small");
    ISPK_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0, ISPK_E_ALIGN, "attn: strides must be multiples of 4");
    ISPK_REQUIRE(ispk_aligned(q, 16) && ispk_aligned(k, 16) && ispk_aligned(v, 16) && ispk_aligned(out, 16),
                 ISPK_E_ALIGN, "attn: pointers must be 16-byte aligned");
    if (B == 0) return 0;
    constexpr size_t lds = (size_t)4 * kTileKeys * kLdk * sizeof(float);  // 69,632 B
    static_assert(lds <= 160 * 1024, "LDS budget");
    dim3 grid((N + 63) / 64, B), block(2 * H * 64);
    if (H <= 6) {
        ISPK_RESERVE_LDS(&attn_f32_kernel<768>, lds, "attn");
        hipLaunchKernelGGL(attn_f32_kernel<768>, grid, block, lds, reinterpret_cast<hipStream_t>(stream), q, ldq, k, v,
                           ldkv, slopes, key_len, out, ldo, N, H);
    } else {
        ISPK_RESERVE_LDS(&attn_f32_kernel<1024>, lds, "attn");
        hipLaunchKernelGGL(attn_f32_kernel<1024>, grid, block, lds, reinterpret_cast<hipStream_t>(stream), q, ldq, k, v,
                           ldkv, slopes, key_len, out, ldo, N, H);
    }
    return ispk_launch_status();
}

namespace {

// ---------------------------------------------------------------------------------------------------------------
// bf16 path (v_mfma_f32_32x32x16_bf16, fp32 accumulate; softmax statistics, bias and mask in fp32).
// Same decomposition and the same Sᵀ = K·Qᵀ trick.  What changes with the 16-deep MFMA:
//   * Q: 4 fragments of 8 bf16 per lane (lane half h owns head dims 16ks + 8h .. +7);
//   * the Pᵀ accumulator (fp32, register r = key (r&3) + 8(r>>2) + 4h) is packed pairwise to bf16 and used directly as
//     the B operand of Oᵀ += Vᵀ·Pᵀ: registers 8s..8s+7 form k-step s, whose element j is key 16s + 8(j>>2) + 4h + (j&3)
//     (guide §3 "An accumulator tile as the next MFMA's operand").  The A operand must present V in that same key order:
//     V stays ROW-major in LDS ([key][d]) and is read with the transposing ds_read_b64_tr_b16 (guide T10): per 16-lane
//     group a 4-key x 16-dim block comes back with one dim per lane and its 4 consecutive keys packed in 8 bytes - two
//     such reads are one A fragment (layout and swizzle: see the staging comment above the kernel);
//   * exp() runs as v_exp_f32 on log2-domain scores; bias and reference maximum ride in the score MFMA's accumulator
//     init (see the softmax comment in the kernel).
// Per 32x32 (key x query) block a wave issues 8 MFMAs (256 cycles) and ~100 VALU instructions incl. 16 v_exp_f32: the
// kernel is VALU-issue-bound, not MFMA-bound; 3 waves per SIMD overlap one wave's softmax with another's MFMAs.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void lds_read_tr16_b64(u32x2& dst, uint32_t lds_byte_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF) : "memory");
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {   // ONE v_cvt_pk_bf16_f32
    f32x2 v;
    v.x = lo; v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// max over the two 32-lane halves without the LDS round trip of a bpermute: v_permlane32_swap exchanges the upper half
// of one register with the lower half of another (gfx950).  (Inline asm: the builtin's second result was miscompiled.)
// v_max3_f32 without the canonicalising v_max x, x that fmaxf() puts in front of every operand (scores are never sNaN)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// exchange the upper half (lanes 32-63) of x with the lower half (lanes 0-31) of y
__device__ __forceinline__ void half_swap(uint32_t& x, uint32_t& y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ float xhalf_max_swap(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// K/V staging.  The loop was latency-bound with register-staged 64-key tiles one tile ahead (a tile's 2,000 cycles of
// work cannot cover a 4,000-cycle miss; deeper register prefetch does not fit 168 VGPRs).  Now K and V stream into a
// 4-slot LDS ring of 128-key chunks by LDS-DMA (global_load_lds_dwordx4: no registers, no LDS store instructions), up
// to three chunks (96 KB) in flight; for N <= 512 the whole sequence is requested before the first score.  A DMA
// instruction writes 64 lanes x 16 B = 8 rows linearly, so rows are unpadded 128 B and bank conflicts are avoided by an
// XOR swizzle applied on the SOURCE side: LDS position (row r, 16-byte slot pc) holds logical chunk pc ^ sw(r),
//     K: sw = (r >> 1) & 7          -> the 16 rows of a ds_read_b128 lane group land on 16 distinct (parity, slot) pairs
//     V: sw = ((r >> 1) & 1) << 2   -> the 4-key x 16-dim blocks of a half's ds_read_b64_tr_b16 cover all 64 banks once
// The first 8 waves are the loaders (IPL DMA instructions per chunk each: K and V rows 16w .. 16w+15); every wave
// waits for its own DMAs with a counted vmcnt and one raw s_barrier per chunk publishes the chunk (and retires the
// slot the next DMA overwrites).  No other vector-memory instruction is issued inside the loop, so vmcnt counts DMAs.
constexpr int kChunkKeys = 128, kSlots = 4;
constexpr int kSlotBytes = kChunkKeys * 128;            // one operand of one slot
constexpr size_t kAttnLds = (size_t)2 * kSlots * kSlotBytes;   // 128 KB

template <int MAXT, int IPL, bool ST = false>   // IPL: DMA instructions per loader wave per chunk = 32 / (loader waves)
__global__ __launch_bounds__(MAXT) void attn_bf16_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                         const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                         int64_t ldkv, const float* __restrict__ slopes,
                                                         const int64_t* __restrict__ key_len,
                                                         uint16_t* __restrict__ out, int64_t ldo, int N, int H, int qpw,
                                                         uint64_t* __restrict__ stamps = nullptr) {
    [[maybe_unused]] uint64_t ts[6] = {0, 0, 0, 0, 0, 0}, tk0 = 0, ta = 0, tb = 0, tc = 0, td = 0;
    if constexpr (ST) tk0 = __builtin_readcyclecounter();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* const Kl = smem_raw;                          // [kSlots][128 rows][128 B]
    char* const Vl = smem_raw + kSlots * kSlotBytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: scalar branches and addresses)
    const int head = wave % H, qhalf = wave / H;
    const int l31 = lane & 31, h = lane >> 5;
    // XCD-aware (batch item, tile group) mapping: workgroups are dealt to the 8 XCDs round-robin in linear order, so the
    // gridDim.x workgroups of one batch item - which fetch the same K/V - would land on different L2s.  Within each run of
    // 8 * gridDim.x workgroups, item = 8 * run + (linear % 8), group = (linear / 8) % gridDim.x: one item, one XCD.
    int b = blockIdx.y, bx = blockIdx.x;
    {
        const int gx = gridDim.x, lin = blockIdx.y * gx + blockIdx.x, run = lin / (8 * gx);
        if ((run + 1) * 8 <= (int)gridDim.y) {            // (a ragged last run keeps the plain mapping)
            const int r = lin - run * 8 * gx;
            b = run * 8 + (r & 7);
            bx = r >> 3;
        }
    }
    int klen = key_len ? (int)key_len[b] : N;
    klen = klen < 1 ? 1 : (klen > N ? N : klen);
    constexpr float kLog2e = 1.4426950408889634f;
    const float scale2 = 0.125f * kLog2e;
    const float ninf = -__builtin_huge_valf();
    const int nchunks = (klen + kChunkKeys - 1) / kChunkKeys;

    // ---- loaders: DMA instruction i of a chunk (i < IPL) moves 8 rows of K (i even) or V (i odd)
    constexpr int NL = 32 / IPL;                        // loader waves
    const bool loader = wave < NL;
    const uint16_t* kb = k + (int64_t)b * N * ldkv;
    const uint16_t* vb = v + (int64_t)b * N * ldkv;
    const int lrow = lane >> 3, lpc = lane & 7;         // this lane's row within the 8-row group and its LDS slot
    auto issue_chunk = [&](int c) {
        const int slot = c & (kSlots - 1);
#pragma unroll
        for (int i = 0; i < IPL; ++i) {
            const int isv = i & 1, grp = wave * (IPL / 2) + (i >> 1);      // 8-row group 0..15 of the chunk
            const int r = grp * 8 + lrow;                                  // row within the chunk
            const int sw = isv ? ((r >> 1) & 1) << 2 : (r >> 1) & 7;
            int key = c * kChunkKeys + r;
            key = key < N ? key : N - 1;                                   // rows past the end: a valid row, masked later
            const uint16_t* src = (isv ? vb : kb) + (int64_t)key * ldkv + ((lpc ^ sw) * 8);
            char* dst = (isv ? Vl : Kl) + slot * kSlotBytes + grp * 1024;  // wave-uniform; lane L lands at + 16 L
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    // A workgroup serves qpw consecutive 64-query tiles of its batch element.  When the whole key range fits the ring
    // (resident: nchunks <= kSlots) K/V are fetched ONCE, all chunks requested up front, and every tile after the first
    // runs without a single wait or barrier; the launcher picks qpw > 1 only then.
    const bool wide_store = (ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;   // 16-byte row pieces
    const bool resident = nchunks <= kSlots;
    const int ahead = resident ? kSlots : kSlots - 1;      // chunks in flight
    // Q fragments are loaded by opaque asm (hipcc would drain every DMA with vmcnt(0) at their first use): they are
    // issued BEFORE the DMAs of the first tile, so the counted vmcnt that admits chunk 0 has retired them too.
    auto load_q = [&](bf16x8 (&qf)[4], int q0) {
        const int qi = q0 + l31;
        const int qrow = qi < N ? qi : N - 1;
        const uint16_t* qp = q + ((int64_t)b * N + qrow) * ldq + head * 64 + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[ks]) : "v"(qp + ks * 16) : "memory");
    };
    bf16x8 qf[4];
    load_q(qf, (bx * qpw) * 64 + qhalf * 32);
    if (loader) {
#pragma unroll 1
        for (int c = 0; c < nchunks && c < ahead; ++c) issue_chunk(c);
    }

    // Softmax bookkeeping, built to keep the per-block VALU work small (the kernel is VALU-issue-bound):
    //   * the ALiBi bias enters the score MFMA as its C operand.  Away from the diagonal block the distance key - query
    //     has a fixed sign, so the bias is "a per-lane base -/+ a per-register constant": the constants are two loop-
    //     invariant register vectors handed to the first MFMA as C (no instruction at all), and the base - together
    //     with the running reference maximum - rides in the one FMA that scales a score into the exp2 argument,
    //         p = exp2(fma(acc, scale2, base2)),   base2 = -m_ref -/+ slope2 * (key0 - query)   [exp2 units];
    //     only the key0 == q0 block pays for the |.|;
    //   * m_ref is a LAZY reference: it is raised (and O, l rescaled) only when a block's scores exceed it by more than
    //     2^kLazy, so most blocks skip the 32-register rescale of O; exp2 arguments stay <= kLazy and the row's true
    //     maximum contributes p >= 1, so nothing overflows or underflows (m_ref starts as block 0's exact maximum);
    //   * plain fp32 instructions throughout (packed fp32 issues slowly beside MFMAs; -fno-slp-vectorize for this file).
    constexpr float kLazy = 16.0f;
    const float nsl = -8.0f * slopes[head];              // = -slope2 / scale2, exact
    const float nsl2 = nsl * scale2;                      // = -slope2
    // +/- nsl * (key offset of accumulator register r): MFMA C operands.  Resident (opaque, or hipcc rebuilds them from
    // literals in front of every block) where the register budget allows: 168 VGPRs at 768 threads; the 1024-thread
    // instances (128 VGPRs) recompute the vector per block instead.
    constexpr bool kResidentC = MAXT <= 768;
    f32x16 cpos, cneg;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        cpos[r] = nsl * (float)((r & 3) + 8 * (r >> 2));
        cneg[r] = -cpos[r];
    }
    if constexpr (kResidentC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(cpos[r]), "+v"(cneg[r]));
    }
    const float scale2s = [&] { float c = scale2; asm volatile("" : "+s"(c)); return c; }();   // in an SGPR: no literal dwords

    // per-lane LDS byte offsets inside a slot.  K fragment of k-step ks: row l31 (+ 32 per block), logical chunk 2ks + h.
    const uint32_t kl_base = lds_addr(Kl), vl_base = lds_addr(Vl);
    uint32_t koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = kl_base + l31 * 128 + (((2 * ks + h) ^ ((l31 >> 1) & 7)) << 4);
    // V transposing read: 16-lane group = (h, dim half dh); lane 4qq + p of the group points at key row 4h + qq, dims
    // 4p .. 4p+3 of the group's 16-dim block, i.e. logical chunk 4dt + 2dh + (p >> 1), byte 8 (p & 1) inside it.
    const int qq = (lane & 15) >> 2, pp = lane & 3, dh = (lane >> 4) & 1;
    uint32_t voff[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
        voff[dt] = vl_base + (4 * h + qq) * 128 + (((4 * dt + 2 * dh + (pp >> 1)) ^ ((qq >> 1) << 2)) << 4) + 8 * (pp & 1);

    if constexpr (ST) { ta = __builtin_readcyclecounter(); ts[0] = ta - tk0; }
    const int nqt = (N + 63) / 64;
#pragma unroll 1
    for (int it = 0; it < qpw; ++it) {
    const int qt = bx * qpw + it;
    if (qt >= nqt) break;                                  // workgroup-uniform
    const int q0 = qt * 64 + qhalf * 32, qi = q0 + l31;
    const int q0w = q0;                                    // the wave's first query: the diagonal block has key0 == q0w
    if (it > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's Q fragments (no DMA is outstanding)
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float mref2 = 0.f;                                     // = -m_ref, exp2 units
    float l2a = 0.f, l2b = 0.f;                            // row sum, two chains
    for (int c = 0; c < nchunks; ++c) {
        if constexpr (ST) ta = __builtin_readcyclecounter();
        if (it == 0) {
            // chunk c has landed once only this wave's younger DMAs (chunks c+1 .. c+ahead-1) are outstanding; the Q
            // loads are older than every DMA.  Waves that load nothing wait for their Q fragments once.
            if (loader) {
                const int after = (nchunks - 1 - c) < (ahead - 1) ? (nchunks - 1 - c) : (ahead - 1);
                if (after >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * IPL) : "memory");
                else if (after == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPL) : "memory");
                else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPL) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (c == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();             // publishes chunk c; every wave is done with chunk c-1
            asm volatile("" ::: "memory");
            if (loader && !resident && c + ahead < nchunks) issue_chunk(c + ahead);   // into the slot chunk c-1 just left
        }
        const int slot = c & (kSlots - 1);
        if constexpr (ST) { tb = __builtin_readcyclecounter(); ts[1] += tb - ta; }
#pragma unroll 1
        for (int kblk = 0; kblk < kChunkKeys / 32; ++kblk) {
            const int key0 = c * kChunkKeys + kblk * 32;
            if (key0 >= klen) break;  // wave-uniform
            if constexpr (ST) { ta = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
            // the 4 K fragments are requested together (opaque asm reads: hipcc would sink each next to its MFMA and wait
            // for it there); the 8 transposed V reads follow the score MFMAs and land during the softmax
            const uint32_t blk = (uint32_t)(slot * kSlotBytes + kblk * 32 * 128);
            bf16x8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) lds_read_b128_asm<0>(kf[ks], koff[ks] + blk);
            f32x16 s;
            float base2;                                   // exp2 argument = fma(s, scale2, base2)
            const float d0 = (float)(key0 + 4 * h - qi);   // key - query of accumulator register 0
            if constexpr (kResidentC) {
                lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
                if (key0 == q0w) {                          // wave-uniform: the one block that straddles the diagonal
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = fabsf(d0 + (float)((r & 3) + 8 * (r >> 2))) * nsl;
                    base2 = mref2;
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], s, 0, 0, 0);
                } else if (key0 < q0w) {                    // keys before the queries: |d| = -(d0 + c_r)
                    base2 = fmaf(-nsl2, d0, mref2);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], cneg, 0, 0, 0);
                } else {
                    base2 = fmaf(nsl2, d0, mref2);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], cpos, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
            } else {
                if (key0 == q0w) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = fabsf(d0 + (float)((r & 3) + 8 * (r >> 2))) * nsl;
                    base2 = mref2;
                } else if (key0 < q0w) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = cneg[r];
                    base2 = fmaf(-nsl2, d0, mref2);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = cpos[r];
                    base2 = fmaf(nsl2, d0, mref2);
                }
                lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            u32x2 vr[2][2][2];   // [st][dim tile][run]: keys key0 + 16st + 4h + 0..3 (run 0) and + 8 (run 1) of this lane's dim
            static_for<0, 8>([&](auto ic) {
                constexpr int i8 = decltype(ic)::value, st = i8 >> 2, dt = (i8 >> 1) & 1, run = i8 & 1;
                lds_read_tr16_b64<(16 * st + 8 * run) * 128>(vr[st][dt][run], voff[dt] + blk);
            });
            if (key0 + 32 >= klen && it + 1 < qpw && qt + 1 < nqt) {
                // the tile's last score product: Q is dead, fetch the next tile's fragments behind the softmax / PV / store
                __builtin_amdgcn_sched_barrier(0);
                load_q(qf, (qt + 1) * 64 + qhalf * 32);
            }
            if constexpr (ST) { __builtin_amdgcn_sched_barrier(0); tb = __builtin_readcyclecounter(); ts[2] += tb - ta; }
            if (key0 + 32 > klen) {   // the one block that straddles key_len (wave-uniform test)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    s[r] = key < klen ? s[r] : ninf;
                }
            }
            float bmax = max3_raw(s[0], s[1], s[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) bmax = max3_raw(bmax, s[r], s[r + 1]);
            // this lane half's block maximum in exp2 units relative to m_ref (the base differs between the halves), then the row's
            bmax = xhalf_max_swap(fmaf(fmaxf(bmax, s[15]), scale2s, base2));
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
                s[2 * j] = __builtin_amdgcn_exp2f(fmaf(s[2 * j], scale2s, base2));
                s[2 * j + 1] = __builtin_amdgcn_exp2f(fmaf(s[2 * j + 1], scale2s, base2));
                l2a += s[2 * j];
                l2b += s[2 * j + 1];
            }
            // P -> bf16 B-operand fragments (k-step st = registers 8st .. 8st+7)
            union { uint32_t u[4]; bf16x8 f; } pf[2];
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 4; ++e) pf[st].u[e] = pack_bf16(s[8 * st + 2 * e], s[8 * st + 2 * e + 1]);
            if constexpr (ST) { __builtin_amdgcn_sched_barrier(0); tc = __builtin_readcyclecounter(); ts[3] += tc - tb; }
            lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                union { uint32_t u[4]; bf16x8 f; } a0, a1;
                a0.u[0] = vr[st][0][0][0]; a0.u[1] = vr[st][0][0][1]; a0.u[2] = vr[st][0][1][0]; a0.u[3] = vr[st][0][1][1];
                a1.u[0] = vr[st][1][0][0]; a1.u[1] = vr[st][1][0][1]; a1.u[2] = vr[st][1][1][0]; a1.u[3] = vr[st][1][1][1];
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.f, pf[st].f, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.f, pf[st].f, o1, 0, 0, 0);
            }
            if constexpr (ST) { __builtin_amdgcn_sched_barrier(0); td = __builtin_readcyclecounter(); ts[4] += td - tc; }
        }
    }
    const float inv = 1.0f / xhalf_sum(l2a + l2b);
    // A lane holds four 4-dim groups of each 32-dim tile (dims 8g + 4h ..): as they are, 8-byte stores - 16 per row and
    // store-issue-bound.  The two halves of a query trade groups (v_permlane32_swap: half 0 ends up with dims 8g .. 8g+7 of
    // g = 0 and 2, half 1 with those of g = 1 and 3), so a row leaves in 16-byte pieces, half as many instructions.
    uint2 pk[2][4];   // [tile][g]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        pk[0][g].x = pack_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv);
        pk[0][g].y = pack_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
        pk[1][g].x = pack_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv);
        pk[1][g].y = pack_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
    if (wide_store) {      // kernel-uniform
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                // x = group 2gp, y = group 2gp+1: afterwards half 0 holds (its x, the partner's x), half 1 (the partner's y, its y)
                half_swap(pk[t][2 * gp].x, pk[t][2 * gp + 1].x);
                half_swap(pk[t][2 * gp].y, pk[t][2 * gp + 1].y);
            }
        if (qi < N) {
            uint16_t* op = out + ((int64_t)b * N + qi) * ldo + head * 64 + 8 * h;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    uint4 v;
                    v.x = pk[t][2 * gp].x; v.y = pk[t][2 * gp].y; v.z = pk[t][2 * gp + 1].x; v.w = pk[t][2 * gp + 1].y;
                    *reinterpret_cast<uint4*>(op + 32 * t + 16 * gp) = v;
                }
        }
    } else if (qi < N) {
        uint16_t* op = out + ((int64_t)b * N + qi) * ldo + head * 64 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<uint2*>(op + 8 * g) = pk[0][g];
            *reinterpret_cast<uint2*>(op + 32 + 8 * g) = pk[1][g];
        }
    }
    }   // query tiles
    if constexpr (ST) {
        ts[5] = __builtin_readcyclecounter() - tk0;
        if (lane == 0) {
            uint64_t* o = stamps + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave) * 6;
            for (int i = 0; i < 6; ++i) o[i] = ts[i];
        }
    }

}

}  // namespace

extern "C" int32_t ispk_alibi_mqa_attn_bf16_tiles(const uint16_t* q, int64_t ldq, const uint16_t* k, const uint16_t* v,
                                                  int64_t ldkv, const float* slopes, const int64_t* key_len,
                                                  uint16_t* out, int64_t ldo, int32_t B, int32_t N, int32_t H,
                                                  int32_t q_tiles_per_workgroup, ispk_stream_t stream) {
    ISPK_REQUIRE(q && k && v && slopes && out, ISPK_E_NULL, "attn: null pointer");
    ISPK_REQUIRE(B >= 0 && N >= 1 && H >= 1 && H <= 8, ISPK_E_SHAPE, "attn: bad shape B=%d N=%d H=%d (H <= 8)", B, N, H);
    ISPK_REQUIRE(B <= 65535, ISPK_E_SHAPE, "attn: B=%d exceeds the grid limit 65535", B);
    ISPK_REQUIRE(ldq >= H * 64 && ldo >= H * 64 && ldkv >= 64, ISPK_E_SHAPE, "attn: leading strides too small");
    ISPK_REQUIRE(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 4 == 0, ISPK_E_ALIGN,
                 "attn: ldq/ldkv must be multiples of 8 and ldo of 4 (bf16)");
    ISPK_REQUIRE(ispk_aligned(q, 16) && ispk_aligned(k, 16) && ispk_aligned(v, 16) && ispk_aligned(out, 8),
                 ISPK_E_ALIGN, "attn: q/k/v must be 16-byte and out 8-byte aligned");
    if (B == 0) return 0;
    // query tiles per workgroup: with the whole key range resident in the ring, one workgroup serves several tiles of its
    // batch element off a single K/V fetch - as many as still leave >= 256 workgroups (one per CU)
    const int nqt = (N + 63) / 64;
    int qpw = 1;
    if (N <= kChunkKeys * kSlots)
        while (qpw < 4 && (int64_t)B * ((nqt + 2 * qpw - 1) / (2 * qpw)) >= 256) qpw *= 2;
    ISPK_REQUIRE(q_tiles_per_workgroup >= 0 && q_tiles_per_workgroup <= 8, ISPK_E_SHAPE,
                 "attn: q_tiles_per_workgroup=%d (0 = automatic, else 1..8)", q_tiles_per_workgroup);
    if (q_tiles_per_workgroup > 0) qpw = N <= kChunkKeys * kSlots ? q_tiles_per_workgroup : 1;   // explicit choice
    dim3 grid((nqt + qpw - 1) / qpw, B), block(2 * H * 64);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define ISPK_ATTN_GO(MAXT_, IPL_)                                                                                  \
    do {                                                                                                           \
        ISPK_RESERVE_LDS((&attn_bf16_kernel<MAXT_, IPL_>), kAttnLds, "attn");                                      \
        hipLaunchKernelGGL((attn_bf16_kernel<MAXT_, IPL_>), grid, block, kAttnLds, st, q, ldq, k, v, ldkv, slopes,  \
                           key_len, out, ldo, N, H, qpw, nullptr);                                                 \
    } while (0)
#ifdef ISPK_EXPERIMENTS
    if (const char* e = ispk_knob("ISPK_ATTN_STAMP")) {   // experiments only: per-wave phase cycle sums -> uint64[waves][6]
        ISPK_REQUIRE(H >= 4 && H <= 6, ISPK_E_UNSUPPORTED, "attn stamps: H = 4..6 only");
        uint64_t* stamps = reinterpret_cast<uint64_t*>(strtoull(e, nullptr, 16));
        ISPK_RESERVE_LDS((&attn_bf16_kernel<768, 4, true>), kAttnLds, "attn");
        hipLaunchKernelGGL((attn_bf16_kernel<768, 4, true>), grid, block, kAttnLds, st, q, ldq, k, v, ldkv, slopes, key_len,
                           out, ldo, N, H, qpw, stamps);
        return ispk_launch_status();
    }
#endif
    if (H >= 7) ISPK_ATTN_GO(1024, 4);       // 14 / 16 waves, 8 loaders
    else if (H >= 4) ISPK_ATTN_GO(768, 4);   // 8 .. 12 waves, 8 loaders
    else if (H >= 2) ISPK_ATTN_GO(768, 8);   // 4 / 6 waves, 4 loaders
    else ISPK_ATTN_GO(768, 16);              // 2 waves, both load
#undef ISPK_ATTN_GO
    return ispk_launch_status();
}

extern "C" int32_t ispk_alibi_mqa_attn_bf16(const uint16_t* q, int64_t ldq, const uint16_t* k, const uint16_t* v,
                                            int64_t ldkv, const float* slopes, const int64_t* key_len, uint16_t* out,
                                            int64_t ldo, int32_t B, int32_t N, int32_t H, ispk_stream_t stream) {
    return ispk_alibi_mqa_attn_bf16_tiles(q, ldq, k, v, ldkv, slopes, key_len, out, ldo, B, N, H, 0, stream);
}
