// Shared pieces of the Linear kernels (gemm.hip, split.hip): the launch parameter block and the epilogues.
#pragma once
#include "common.h"
#include "dropout.h"

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
    // LayerNorm of the INPUT rows computed by the kernel itself (fused feed-forward, ispk_ffn_bf16_prenorm): A is fp32
    const float* lx_gamma = nullptr;
    const float* lx_beta = nullptr;
    float lx_eps = 1e-5f;
    // ... and the attention output projection in front of it (ispk_attn_out_ffn_bf16): x1 = pj_x + mask * (pj_o · pj_wᵀ)
    const uint16_t* pj_o = nullptr;
    int64_t pj_ldo = 0;
    const uint16_t* pj_w = nullptr;
    const float* pj_x = nullptr;
    int64_t pj_ldx = 0;
    // split-f16 operands (split.hip): element offset of the lo plane from the hi plane for A, W and a split output C
    int64_t a_plane = 0, w_plane = 0, c_plane = 0;
    // batched fp32 product (ispk_gemm_f32_batched): element strides of A, W and C from one batch item (blockIdx.z) to the next
    int64_t za = 0, zw = 0, zc = 0;
    // training epilogues of the panel kernel (ISPK_EP_DUAL_GELU / ISPK_EP_GELU_BWD): a second bf16 tensor of C's shape and the
    // dropout of the feed-forward activation (dropout.h: mask = hash(seed, row * N + feature))
    void* aux = nullptr;
    int64_t ld_aux = 0;
    uint64_t drop_seed = 0;
    const uint64_t* drop_src = nullptr;
    uint32_t drop_thresh = 0;
    float drop_inv_keep = 1.f;
};

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

// Compile-time epilogue description.  EP < 0 (kEpDyn): every flag / optional pointer is tested at run time (generic);
// EP >= 0: the ISPK_EP_* flag word plus kEpBias / kEpResid, promised by the launcher to equal the run-time values — the
// flag tests then fold away and a hot instance carries no branches, no dead GELU/SiLU code and no bias round trip.
constexpr int kEpDyn = -1;
constexpr int kEpBias = 1 << 16, kEpResid = 1 << 17;
template <int EP> __device__ __forceinline__ bool ep_flag(const GemmParams& p, uint32_t f) {
    if constexpr (EP < 0) return (p.flags & f) != 0; else return ((uint32_t)EP & f) != 0;
}
template <int EP> __device__ __forceinline__ bool ep_bias(const GemmParams& p) {
    if constexpr (EP < 0) return p.bias != nullptr; else return (EP & kEpBias) != 0;
}
template <int EP> __device__ __forceinline__ bool ep_resid(const GemmParams& p) {
    if constexpr (EP < 0) return p.resid != nullptr; else return (EP & kEpResid) != 0;
}
inline int ep_key(const GemmParams& p) {
    return (int)(p.flags & 0xffffu) | (p.bias ? kEpBias : 0) | (p.resid ? kEpResid : 0);
}

// two fp32 -> packed bf16x2 in ONE v_cvt_pk_bf16_f32 (round to nearest even)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2 v;
    v.x = lo; v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <int EP = kEpDyn>
__device__ __forceinline__ void pre_stage(const GemmParams& p, int n, float (&v)[4], float mk) {
    if (ep_bias<EP>(p)) {
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
    }
    if (ep_flag<EP>(p, ISPK_EP_GELU)) {  // bf16-operand kernels only: the packed A&S erf (|err| <= 3e-7) is far below their noise
        f32x2 a, b;
        a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3];
        a = gelu_fast2(a);
        b = gelu_fast2(b);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
    if (ep_flag<EP>(p, ISPK_EP_SILU)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu(v[e]);
    }
    if (ep_flag<EP>(p, ISPK_EP_MASK_ACC)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= mk;
    }
}

// The residual values one store_rows_f32 call adds, in its store layout (lane -> rows 8i + (lane>>3), features
// n0 + 4(lane&7) .. +3).  Loaded by the caller ahead of time: inside store_rows_f32 each load would be followed at once
// by its use, one exposed memory round trip per 32-feature tile (12 of them made up a sixth of the fused FFN's time).
template <int EP = kEpDyn>
__device__ __forceinline__ void resid_prefetch(const GemmParams& p, int m0, int n0, int lane, float4 (&r4)[4]) {
    if (!ep_resid<EP>(p)) return;
    const int n = n0 + 4 * (lane & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 8 * i + (lane >> 3);
        r4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < p.M && n < p.N) {
            const int64_t ro = (int64_t)m * p.ldr + n;
            if (ep_flag<EP>(p, ISPK_EP_RESID_BF16)) {
                const uint2 rr = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p.resid) + ro);
                r4[i].x = bf16_to_f32((uint16_t)(rr.x & 0xffffu)); r4[i].y = bf16_to_f32((uint16_t)(rr.x >> 16));
                r4[i].z = bf16_to_f32((uint16_t)(rr.y & 0xffffu)); r4[i].w = bf16_to_f32((uint16_t)(rr.y >> 16));
            } else {
                r4[i] = *reinterpret_cast<const float4*>(static_cast<const float*>(p.resid) + ro);
            }
        }
    }
}

// The ISPK_EP_MASK_OUT multipliers of the 4 rows a lane stores (rows m0 + 8i + (lane>>3)): the same for every feature tile
// of a wave's 32 rows, so they are read once (read inside store_rows_f32 they are one more dependent load per tile).
template <int EP = kEpDyn>
__device__ __forceinline__ void mask_rows(const GemmParams& p, int m0, int lane, float (&mo4)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 8 * i + (lane >> 3);
        mo4[i] = (ep_flag<EP>(p, ISPK_EP_MASK_OUT) && p.mask && m < p.M) ? (p.mask[m] ? 1.0f : 0.0f) : 1.0f;
    }
}

// fp32 output: one 32-feature tile (features n0 .. n0+31) of the wave's 32 rows (m0 .. m0+31).  pre: the tile's residual
// values from resid_prefetch (nullptr: loaded here); mo4: mask_rows() of these rows (nullptr: read here).
template <int EP = kEpDyn>
__device__ __forceinline__ void store_rows_f32(const GemmParams& p, char* stage, int m0, int n0, const f32x16& acc,
                                               float mk, int lane, float4* keep = nullptr, const float4* pre = nullptr,
                                               const float* mo4 = nullptr) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e];
        const int n = n0 + 8 * g + 4 * h;
        pre_stage<EP>(p, n < p.N ? n : 0, v, mk);
        *reinterpret_cast<float4*>(stage + l31 * kStageRow + (8 * g + 4 * h) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    const int c = lane & 7, n = n0 + 4 * c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), m = m0 + r;
        float4 v = *reinterpret_cast<const float4*>(stage + r * kStageRow + c * 16);
        if (m < p.M && n < p.N) {
            if (ep_resid<EP>(p)) {
                if (pre) {
                    v.x += pre[i].x; v.y += pre[i].y; v.z += pre[i].z; v.w += pre[i].w;
                } else {
                    const int64_t ro = (int64_t)m * p.ldr + n;
                    if (ep_flag<EP>(p, ISPK_EP_RESID_BF16)) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p.resid) + ro);
                        v.x += bf16_to_f32((uint16_t)(rr.x & 0xffffu)); v.y += bf16_to_f32((uint16_t)(rr.x >> 16));
                        v.z += bf16_to_f32((uint16_t)(rr.y & 0xffffu)); v.w += bf16_to_f32((uint16_t)(rr.y >> 16));
                    } else {
                        const float4 rr = *reinterpret_cast<const float4*>(static_cast<const float*>(p.resid) + ro);
                        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                    }
                }
            }
            if (ep_flag<EP>(p, ISPK_EP_MASK_OUT)) {
                const float mo = mo4 ? mo4[i] : (p.mask[m] ? 1.0f : 0.0f);
                v.x *= mo; v.y *= mo; v.z *= mo; v.w *= mo;
            }
            *reinterpret_cast<float4*>(static_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = v;
        }
        if (keep) keep[i] = v;  // final values in row layout: rows 8i + (lane>>3), features n0 + 4(lane&7) .. +3
    }
}

// bf16 output: two adjacent 32-feature tiles (features n0 .. n0+63); no residual on this path
template <int EP = kEpDyn>
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
            pre_stage<EP>(p, n < p.N ? n : 0, v, mk);
            if (ep_flag<EP>(p, ISPK_EP_MASK_OUT)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= mk;
            }
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(stage + l31 * kStageRow + (t * 32 + 8 * g + 4 * h) * 2) = o;
        }
    const int c = lane & 7, n = n0 + 8 * c;
    if (ep_flag<EP>(p, ISPK_EP_GELU_BWD)) {
        // C = du = da gelu'(u) [keep / (1 - p)] with u (bf16, aux) read in the store layout: the da -> du pass of the
        // feed-forward backward (gelu_bwd_kernel<true, true>, same expression, same mask) without its own launch
        const uint64_t seed = run_seed(p.drop_seed, p.drop_src);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3), m = m0 + r;
            if (m >= p.M || n >= p.N) continue;
            const uint4 dv = *reinterpret_cast<const uint4*>(stage + r * kStageRow + c * 16);
            const uint4 uv = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.aux) + (int64_t)m * p.ld_aux + n);
            const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w}, uw[4] = {uv.x, uv.y, uv.z, uv.w};
            uint32_t ow[4];
            const uint32_t idx0 = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float da = bf16_to_f32((uint16_t)(k ? dw[e] >> 16 : dw[e] & 0xffffu));
                    const float u = bf16_to_f32((uint16_t)(k ? uw[e] >> 16 : uw[e] & 0xffffu));
                    float v = da * gelu_grad_fast(u);
                    if (p.drop_thresh) v = drop_keep(seed, idx0 + 2 * e + k, p.drop_thresh) ? v * p.drop_inv_keep : 0.f;
                    g[k] = v;
                }
                ow[e] = pack_bf16x2(g[0], g[1]);
            }
            *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p.C) + (int64_t)m * p.ldc + n) = uint4{ow[0], ow[1], ow[2], ow[3]};
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), m = m0 + r;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + r * kStageRow + c * 16);
        if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p.C) + (int64_t)m * p.ldc + n) = v;
    }
    if (ep_flag<EP>(p, ISPK_EP_DUAL_GELU)) {
        // second output aux = dropout(gelu(bf16(u))): what gelu_fwd_kernel<true, true> makes of the u just stored (same
        // expression on the ROUNDED pre-activation, same mask), without re-reading it
        const uint64_t seed = run_seed(p.drop_seed, p.drop_src);
        const uint32_t row_idx = (uint32_t)(m0 + l31) * (uint32_t)p.N;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nn = n0 + t * 32 + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = bf16_to_f32(f32_to_bf16(t ? acc1[4 * g + e] : acc0[4 * g + e]));
                    float a = gelu_as28(u);
                    if (p.drop_thresh) a = drop_keep(seed, row_idx + (uint32_t)(nn + e), p.drop_thresh) ? a * p.drop_inv_keep : 0.f;
                    v[e] = a;
                }
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(stage + l31 * kStageRow + (t * 32 + 8 * g + 4 * h) * 2) = o;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3), m = m0 + r;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + r * kStageRow + c * 16);
            if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p.aux) + (int64_t)m * p.ld_aux + n) = v;
        }
    }
}


// ISPK_EP_ROWS_T: the wave's 32 rows are frames t of batch item b = m / T; output feature n goes to C[b][n][t].  In the
// transposed-compute accumulator the frame sits on the lane, so each register is already a frame-contiguous 128-byte
// segment (x 2 halves) of one output channel: no LDS pass.  cb = &C[b][0][t] of this lane's row (nullptr: row >= M).
template <int EP = kEpDyn>
__device__ __forceinline__ void store_rows_t(const GemmParams& p, float* cb, int n0, const f32x16& acc, float mo, int h) {
    if (cb == nullptr) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + 8 * g + 4 * h;
        if (n >= p.N) continue;   // N % 4 == 0 (checked by the launcher)
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ep_bias<EP>(p)) bb = *reinterpret_cast<const float4*>(p.bias + n);
        float* c = cb + (int64_t)n * p.ldc;
        c[0] = (acc[4 * g] + bb.x) * mo;
        c[p.ldc] = (acc[4 * g + 1] + bb.y) * mo;
        c[2 * p.ldc] = (acc[4 * g + 2] + bb.z) * mo;
        c[3 * p.ldc] = (acc[4 * g + 3] + bb.w) * mo;
    }
}

// can the row-coalescing epilogue be used?  (else: epilogue_vec4)
inline bool rows_epilogue_ok(const GemmParams& p) {
    if (p.flags & ISPK_EP_OUT_BF16)
        return !p.resid && p.N % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0;
    return true;  // fp32 out: vec_epilogue_ok() already guarantees 16-byte alignment of C / resid rows
}

// 16 zero bytes: the source of LDS-DMA lanes whose k index lies beyond K (a DMA cannot zero-fill)
__device__ __attribute__((aligned(16))) const uint16_t g_zero16[8] = {0, 0, 0, 0, 0, 0, 0, 0};

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace
