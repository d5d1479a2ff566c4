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
    const int lane = tid & 63, wave = tid >> 6;
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
//     (guide §3 "An accumulator tile as the next MFMA's operand").  The A operand must present V in that same key order,
//     so the V tile is staged TRANSPOSED in LDS (Vt[d][key]): a lane reads two 8-byte runs of 4 consecutive keys;
//   * exp() runs as v_exp_f32 on log2-domain scores: s*log2(e)/8 - slope*log2(e)*|i-j| in one FMA.
// Per 32x32 (key x query) block a wave issues 8 MFMAs (256 cycles) but ~16 exp + ~130 VALU ops, so this kernel is
// VALU-bound, not MFMA-bound; 3 waves per SIMD overlap one wave's softmax with another's MFMAs.
constexpr int kLdh = 72;  // padded row of the bf16 K / Vt tiles (64 + 8 elements = 144 B)

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

template <int MAXT>
__global__ __launch_bounds__(MAXT) void attn_bf16_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                         const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                         int64_t ldkv, const float* __restrict__ slopes,
                                                         const int64_t* __restrict__ key_len,
                                                         uint16_t* __restrict__ out, int64_t ldo, int N, int H) {
    __shared__ __attribute__((aligned(16))) uint16_t Ks[2 * kTileKeys * kLdh];  // [2][key][d]
    __shared__ __attribute__((aligned(16))) uint16_t Vt[2 * 64 * kLdh];         // [2][d][key]

    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int head = wave % H, qhalf = wave / H;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 64 + qhalf * 32;
    int klen = key_len ? (int)key_len[b] : N;
    klen = klen < 1 ? 1 : (klen > N ? N : klen);
    constexpr float kLog2e = 1.4426950408889634f;
    const float slope2 = slopes[head] * kLog2e;
    const float scale2 = 0.125f * kLog2e;
    const float ninf = -__builtin_huge_valf();

    const int qi = q0 + l31;
    const int qrow = qi < N ? qi : N - 1;
    bf16x8 qf[4];
    {
        const uint16_t* qp = q + ((int64_t)b * N + qrow) * ldq + head * 64 + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = ninf, l_run = 0.f;

    const uint16_t* kb = k + (int64_t)b * N * ldkv;
    const uint16_t* vb = v + (int64_t)b * N * ldkv;
    const int ntiles = (klen + kTileKeys - 1) / kTileKeys;

    // split staging (see attn_f32_kernel): 1024 16-byte pieces per tile (K row-major; V scattered transposed into
    // Vt[d][key]), loads issued before the current tile's compute, LDS writes after it.
    constexpr int kStageMax = 2;  // ceil(1024 / 512)
    u32x4 sreg[kStageMax];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int i = 0; i < kStageMax; ++i) {
            const int idx = tid + i * nthreads;
            const int isv = idx >> 9, rem = idx & 511;
            const int row = rem >> 3, c8 = (rem & 7) * 8;
            const int key = t * kTileKeys + row;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (idx < kTileKeys * 16 && key < N)
                val = *reinterpret_cast<const u32x4*>((isv ? vb : kb) + (int64_t)key * ldkv + c8);
            sreg[i] = val;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kStageMax; ++i) {
            const int idx = tid + i * nthreads;
            if (idx >= kTileKeys * 16) continue;
            const int isv = idx >> 9, rem = idx & 511;
            const int row = rem >> 3, c8 = (rem & 7) * 8;
            if (!isv) {
                *reinterpret_cast<u32x4*>(Ks + (buf * kTileKeys + row) * kLdh + c8) = sreg[i];
            } else {
                uint16_t* dst = Vt + (buf * 64 + c8) * kLdh + row;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dst[(2 * e) * kLdh] = (uint16_t)(sreg[i][e] & 0xffffu);
                    dst[(2 * e + 1) * kLdh] = (uint16_t)(sreg[i][e] >> 16);
                }
            }
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
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const uint16_t* kp = Ks + (buf * kTileKeys + kblk * 32 + l31) * kLdh + h * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            }
            // log2-domain score: s*log2(e)/8 - slope*log2(e)*|key - query|.  The distance is fp32 from the start
            // (d0 + compile-time register offset), so an element costs one add and one FMA with an |.| source modifier;
            // the key-length mask is applied only in the one block that straddles key_len (wave-uniform test).
            const float d0 = (float)(key0 + 4 * h - qi);
            float smax = ninf;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dist = fabsf(d0 + (float)((r & 3) + 8 * (r >> 2)));
                s[r] = fmaf(s[r], scale2, -slope2 * dist);
            }
            if (key0 + 32 > klen) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    s[r] = key < klen ? s[r] : ninf;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) smax = fmaxf(smax, s[r]);
            smax = xhalf_max(smax);
            const float m_new = fmaxf(m_run, smax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // bare v_exp_f32; exp2(-inf) = 0 on first block
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pexp = __builtin_amdgcn_exp2f(s[r] - m_new);
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
            // P -> bf16 B-operand fragments (k-step st = registers 8st .. 8st+7)
            const uint16_t* vp = Vt + (buf * 64 + l31) * kLdh + kblk * 32 + 4 * h;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                union { uint32_t u[4]; bf16x8 f; } pf, va, vc;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16(s[8 * st + 2 * e], s[8 * st + 2 * e + 1]);
                // A operand: Vt[d][key0 + 16st + 4h + 0..3] and [.. + 8 + 0..3]
                const uint2 a_lo = *reinterpret_cast<const uint2*>(vp + 16 * st);
                const uint2 a_hi = *reinterpret_cast<const uint2*>(vp + 16 * st + 8);
                va.u[0] = a_lo.x; va.u[1] = a_lo.y; va.u[2] = a_hi.x; va.u[3] = a_hi.y;
                const uint2 c_lo = *reinterpret_cast<const uint2*>(vp + 32 * kLdh + 16 * st);
                const uint2 c_hi = *reinterpret_cast<const uint2*>(vp + 32 * kLdh + 16 * st + 8);
                vc.u[0] = c_lo.x; vc.u[1] = c_lo.y; vc.u[2] = c_hi.x; vc.u[3] = c_hi.y;
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va.f, pf.f, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc.f, pf.f, o1, 0, 0, 0);
            }
        }
        if (t + 1 < ntiles) stage_store(buf ^ 1);
        __syncthreads();
    }

    const float inv = 1.0f / xhalf_sum(l_run);
    if (qi < N) {
        uint16_t* op = out + ((int64_t)b * N + qi) * ldo + head * 64 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 a, c;
            a.x = pack_bf16(o0[4 * g] * inv, o0[4 * g + 1] * inv);
            a.y = pack_bf16(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            c.x = pack_bf16(o1[4 * g] * inv, o1[4 * g + 1] * inv);
            c.y = pack_bf16(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + 8 * g) = a;
            *reinterpret_cast<uint2*>(op + 32 + 8 * g) = c;
        }
    }
}

}  // namespace

extern "C" int32_t ispk_alibi_mqa_attn_bf16(const uint16_t* q, int64_t ldq, const uint16_t* k, const uint16_t* v,
                                            int64_t ldkv, const float* slopes, const int64_t* key_len, uint16_t* out,
                                            int64_t ldo, int32_t B, int32_t N, int32_t H, ispk_stream_t stream) {
    ISPK_REQUIRE(q && k && v && slopes && out, ISPK_E_NULL, "attn: null pointer");
    ISPK_REQUIRE(B >= 0 && N >= 1 && H >= 1 && H <= 8, ISPK_E_SHAPE, "attn: bad shape B=%d N=%d H=%d (H <= 8)", B, N, H);
    ISPK_REQUIRE(B <= 65535, ISPK_E_SHAPE, "attn: B=%d exceeds the grid limit 65535", B);
    ISPK_REQUIRE(ldq >= H * 64 && ldo >= H * 64 && ldkv >= 64, ISPK_E_SHAPE, "attn: leading strides too small");
    ISPK_REQUIRE(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 4 == 0, ISPK_E_ALIGN,
                 "attn: ldq/ldkv must be multiples of 8 and ldo of 4 (bf16)");
    ISPK_REQUIRE(ispk_aligned(q, 16) && ispk_aligned(k, 16) && ispk_aligned(v, 16) && ispk_aligned(out, 8),
                 ISPK_E_ALIGN, "attn: q/k/v must be 16-byte and out 8-byte aligned");
    if (B == 0) return 0;
    dim3 grid((N + 63) / 64, B), block(2 * H * 64);
    if (H <= 6)
        hipLaunchKernelGGL(attn_bf16_kernel<768>, grid, block, 0, reinterpret_cast<hipStream_t>(stream), q, ldq, k, v,
                           ldkv, slopes, key_len, out, ldo, N, H);
    else
        hipLaunchKernelGGL(attn_bf16_kernel<1024>, grid, block, 0, reinterpret_cast<hipStream_t>(stream), q, ldq, k, v,
                           ldkv, slopes, key_len, out, ldo, N, H);
    return ispk_launch_status();
}
