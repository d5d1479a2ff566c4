// Aligner front-end (the ConvAttention that produces the MAS input) for gfx950, fp32.
//
// Reference: /root/reference/tts/models/acoustic/modules/alignment.py:69-83 (ConvBlock1D: x*mask -> Conv1d -> GELU ->
// masked instance norm), :159-208 (ConvAttention.forward), :18-37 (batch_diagonal_prior), and
// modules/normalization.py:160-208 (_masked_norm, "instance").  The reference runs ~100 small PyTorch kernels here
// (masks, convs through a channel-first layout, a 16-op instance norm three times, matmul, clamp, prior, log-softmax,
// clone, masked-fill, softmax, mask).  Here:
//   * activations are channel-LAST and padded in time, [B][T+4][C] with two zero rows on each side of every utterance.
//     A Conv1d with kernel 5 / padding 2 is then ONE GEMM over OVERLAPPING rows: output row r is the dot of the weight
//     [O][5*C] with the 5*C contiguous floats starting at padded row r (leading stride C < K).  No im2col buffer exists;
//     the GEMM is the same MFMA kernel as every other Linear (ispk_gemm_f32, GELU in its epilogue);
//   * `ispk_pad_rows_f32` builds the first padded buffer (mask applied, optional channel-first -> channel-last transpose);
//   * `ispk_masked_instnorm_f32` does statistics over the valid frames, normalises, applies the affine, re-applies the
//     mask for the next convolution and writes straight into the next padded buffer;
//   * `ispk_aligner_scores_f32` fuses Q·Kᵀ/sqrt(d), the clamp, log-softmax over ALL key columns (padded ones included,
//     alignment.py:196), the analytic diagonal prior (never materialised), `attn_logits`, the key-masked softmax and the
//     final mask into one pass, computing Sᵀ = K·Qᵀ on the fp32 MFMA so a lane owns one mel row and the row reductions
//     are in-register.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ pad + mask (+T)
template <typename T_> __device__ __forceinline__ void put(T_* p, float v);
template <> __device__ __forceinline__ void put<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void put<uint16_t>(uint16_t* p, float v) { *p = f32_to_bf16(v); }
// split fp16 planes (csrc/split.hip): hi = fp16(v) at p, lo = fp16(v - hi) `plane` elements behind; OutT = _Float16
__device__ __forceinline__ void put_split(_Float16* p, float v, int64_t plane) {
    v = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
    const _Float16 hi = (_Float16)v;
    p[0] = hi;
    p[plane] = (_Float16)(v - (float)hi);
}
template <typename OutT> __device__ __forceinline__ void put_any(OutT* p, float v, int64_t plane) {
    if constexpr (std::is_same<OutT, _Float16>::value) put_split(p, v, plane);
    else put<OutT>(p, v);
}

template <typename OutT>
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ x, int64_t sb, int64_t st, int64_t sc,
                                                       const int64_t* __restrict__ len, OutT* __restrict__ out, int T,
                                                       int C, int64_t total, int64_t plane) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t row = idx / C;
    const int tp = (int)(row % (T + 4)), b = (int)(row / (T + 4));
    const int t = tp - 2;
    float v = 0.f;
    if (t >= 0 && t < T && t < (int)len[b]) v = x[b * sb + t * st + c * sc];
    put_any<OutT>(out + idx, v, plane);
}

// Channel-first input ([B][C][T], time contiguous: the collator's mel layout): a 32 x 32 tile transposed through LDS so
// that both the reads (along t) and the writes (along c) are coalesced; the element-wise kernel above reads such an
// input with a stride of T floats between neighbouring threads.  grid (ceil((T+4)/32), ceil(C/32), B), block (32, 8).
template <typename OutT>
__global__ __launch_bounds__(256) void pad_rows_cf_kernel(const float* __restrict__ x, int64_t sb, int64_t sc,
                                                          const int64_t* __restrict__ len, OutT* __restrict__ out, int T,
                                                          int C, int64_t plane) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y, b = blockIdx.z;
    const int tp0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    int n = (int)len[b];
    n = n < T ? n : T;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, t = tp0 + tx - 2;
        tile[ty + 8 * k][tx] = (c < C && t >= 0 && t < n) ? x[b * sb + c * sc + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tp = tp0 + ty + 8 * k, c = c0 + tx;
        if (tp < T + 4 && c < C) put_any<OutT>(out + ((int64_t)b * (T + 4) + tp) * C + c, tile[tx][ty + 8 * k], plane);
    }
}

// ------------------------------------------------------------------------------------------------ masked instance norm
// grid (ceil(C/64), B), 1024 threads = 16 time lanes x 64 channels (256-B coalesced rows).  y: [B][T+4][C] conv output
// (row r = b*(T+4) + t is frame t); out: next padded buffer, frame t at row t+2.  Two passes like the reference
// (mean, then centred variance), a third to write.
constexpr int kTL = 16;
template <typename OutT>
__global__ __launch_bounds__(1024) void masked_instnorm_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                               const float* __restrict__ bias,
                                                               const int64_t* __restrict__ len, OutT* __restrict__ out,
                                                               int T, int C, float eps, int64_t plane) {
    __shared__ float red[kTL][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + cl;
    const bool cok = c < C;
    int n = (int)len[b];
    n = n < 1 ? 1 : (n > T ? T : n);
    const float* yb = y + (int64_t)b * (T + 4) * C;
    OutT* ob = out + (int64_t)b * (T + 4) * C;
    auto total = [&]() {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < kTL; ++i) v += red[i][cl];
        return v;
    };
    // (the three time loops keep their summation order but issue 8 loads at a time: with one load per iteration each
    // of a thread's 32 steps waited out a memory round trip and the kernel ran at a tenth of the HBM rate)
    float s = 0.f;
    if (cok) {
        int t = tl;
        for (; t + 7 * kTL < n; t += 8 * kTL) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = yb[(int64_t)(t + u * kTL) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; t < n; t += kTL) s += yb[(int64_t)t * C + c];
    }
    red[tl][cl] = s;
    __syncthreads();
    const float mean = total() / (float)n;
    __syncthreads();
    float ss = 0.f;
    if (cok) {
        int t = tl;
        for (; t + 7 * kTL < n; t += 8 * kTL) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = yb[(int64_t)(t + u * kTL) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d = v[u] - mean;
                ss += d * d;
            }
        }
        for (; t < n; t += kTL) {
            const float d = yb[(int64_t)t * C + c] - mean;
            ss += d * d;
        }
    }
    red[tl][cl] = ss;
    __syncthreads();
    const float var = total() / (float)n;
    if (!cok) return;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float g = w[c], be = bias[c];
#pragma unroll 8
    for (int tp = tl; tp < T + 4; tp += kTL) {
        const int t = tp - 2;
        const int tc = t < 0 ? 0 : (t < n ? t : n - 1);          // clamped (always in-bounds) load, selected below
        const float raw = yb[(int64_t)tc * C + c];
        const float v = (t >= 0 && t < n) ? (raw - mean) * rstd * g + be : 0.f;
        put_any<OutT>(ob + (int64_t)tp * C + c, v, plane);
    }
}

// ------------------------------------------------------------------------------------------------ scores + softmaxes
constexpr int kAD = 128;        // attention_dim of the aligner
// Key tile in LDS: [keys][128] fp32, UNPADDED (320 keys x 512 B = exactly the 160 KiB of a CU) with the 16-byte chunk
// index XOR-ed with the key row (chunk' = chunk ^ (row & 31)), so the 16 lanes of a ds_read_b128 group, which read the
// same chunk of 16 different rows, land on 16 different 4-bank slots.
__device__ __forceinline__ int ks_off(int row, int chunk) { return row * kAD + ((chunk ^ (row & 31)) << 2); }

// grid (ceil(M/64), B), 128 threads: wave w owns mel rows blockIdx.x*64 + w*32 .. +31.  Two waves per workgroup, so that
// two workgroups share a CU at L_max <= 160 (64 KB of keys each) and one's load / MFMA / transcendental / store phases
// overlap the other's (one 4-wave workgroup per CU ran them strictly one after the other: 56 us at B=64 x 512 x 100).
// NB = ceil(L_max / 32) 32-key blocks held in registers (NB <= 10: L_max <= 320).
constexpr int kAsWaves = 2;
// kFast (ispk_aligner_scores_fast_f32, the bf16 compute path - its q / k already carry the convolutions' bf16 rounding): the
// score products as three bf16 MFMAs over hi / lo splits of the fp32 operands (~2^-16 relative; 24 MFMAs of 32 cycles per
// 32-key block instead of 64 of 64 cycles) and v_exp_f32 / v_log_f32 instead of libm's expf / logf.
typedef uint32_t as_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void as_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    union { as_u32x4 u; bf16x8 f; } h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f2 a;
        a.x = v[2 * e]; a.y = v[2 * e + 1];
        const uint32_t ph = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf2));
        f2 r;
        r.x = a.x - __builtin_bit_cast(float, ph << 16);
        r.y = a.y - __builtin_bit_cast(float, ph & 0xffff0000u);
        h.u[e] = ph;
        l.u[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf2));
    }
    hi = h.f;
    lo = l.f;
}
template <bool kFast>
__device__ __forceinline__ float as_exp(float x) {
    if constexpr (kFast) return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
    else return expf(x);
}
template <bool kFast>
__device__ __forceinline__ float as_log(float x) {
    if constexpr (kFast) return __builtin_amdgcn_logf(x) * 0.6931471805599453f;
    else return logf(x);
}

template <int NB, bool kFast = false>
__global__ __launch_bounds__(64 * kAsWaves) void aligner_scores_kernel(const float* __restrict__ qe, int64_t q_stride_b,
                                                             const float* __restrict__ ke, int64_t k_stride_b,
                                                             const int64_t* __restrict__ text_len,
                                                             const int64_t* __restrict__ mel_len,
                                                             float* __restrict__ logits, float* __restrict__ soft,
                                                             int M, int L, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);                        // [NB*32][128], swizzled (ks_off)
    char* stage = smem_raw + (threadIdx.x >> 6) * (32 * 144);  // per-wave transpose patch: aliases Ks once S is done

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int m0 = blockIdx.x * (32 * kAsWaves) + wave * 32;
    int tl = (int)text_len[b], ml = (int)mel_len[b];
    tl = tl < 1 ? 1 : (tl > L ? L : tl);
    ml = ml < 1 ? 1 : (ml > M ? M : ml);
    const float ninf = -__builtin_huge_valf();

    // stage the encoded keys of this utterance (rows >= L zero-filled)
    const float* kb = ke + (int64_t)b * k_stride_b;
    {   // 8 loads in flight per thread (a load -> store loop left as is runs one L2 round trip per 16 bytes)
        constexpr int kPieces = NB * 32 * (kAD / 4) / (64 * kAsWaves), kBatch = 8;
        static_assert(kPieces % kBatch == 0, "key tile pieces per thread");
        for (int i0 = 0; i0 < kPieces; i0 += kBatch) {
            float4 v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int idx = tid + (i0 + u) * (64 * kAsWaves);
                const int row = idx / (kAD / 4), c4 = (idx - row * (kAD / 4)) * 4;
                const int rr = row < L ? row : L - 1;                       // clamped: unconditional loads
                v[u] = *reinterpret_cast<const float4*>(kb + (int64_t)rr * kAD + c4);
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int idx = tid + (i0 + u) * (64 * kAsWaves);
                const int row = idx / (kAD / 4), c4 = (idx - row * (kAD / 4)) * 4;
                *reinterpret_cast<float4*>(Ks + ks_off(row, c4 >> 2)) = row < L ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // this lane's mel row as the MFMA B operand: half h owns dims 64h .. 64h+63
    const int m = m0 + l31;
    const int mrow = m < M ? m : M - 1;
    f32x4 qf[16];
    {
        const float* qp = qe + (int64_t)b * q_stride_b + (int64_t)mrow * kAD + h * 64;
#pragma unroll
        for (int c = 0; c < 16; ++c) qf[c] = *reinterpret_cast<const f32x4*>(qp + c * 4);
    }
    __syncthreads();

    // S^T[key][mel]: register r of block kb_ is key 32*kb_ + (r&3) + 8(r>>2) + 4h
    f32x16 s[NB];
    if constexpr (kFast) {
        // step st of lane half h covers dims 64 h + 8 st .. + 7 (chunks 2 st, 2 st + 1 of the half) in both operands
        bf16x8 qh[8], ql[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = qf[2 * st][e]; v[4 + e] = qf[2 * st + 1][e]; }
            as_split8(v, qh[st], ql[st]);
        }
#pragma unroll
        for (int kb_ = 0; kb_ < NB; ++kb_) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb_][r] = 0.f;
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const f32x4 k0 = *reinterpret_cast<const f32x4*>(Ks + ks_off(kb_ * 32 + l31, h * 16 + 2 * st));
                const f32x4 k1 = *reinterpret_cast<const f32x4*>(Ks + ks_off(kb_ * 32 + l31, h * 16 + 2 * st + 1));
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = k0[e]; v[4 + e] = k1[e]; }
                bf16x8 kh, kl;
                as_split8(v, kh, kl);
                s[kb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[st], s[kb_], 0, 0, 0);
                s[kb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[st], s[kb_], 0, 0, 0);
                s[kb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[st], s[kb_], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
    for (int kb_ = 0; kb_ < NB; ++kb_) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb_][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + ks_off(kb_ * 32 + l31, h * 16 + c));
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kb_] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[c][e], s[kb_], 0, 0, 0);
        }
    }
    }

    __syncthreads();  // every wave has its scores in registers: the key tile is dead, its space becomes the patches

    // ---- log_softmax over ALL L key columns of scale*S (clamped to the fp32 max like the reference, alignment.py:192)
    float rmax = ninf;
#pragma unroll
    for (int kb_ = 0; kb_ < NB; ++kb_)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = fminf(s[kb_][r] * scale, 3.4028234663852886e+38f);
            v = key < L ? v : ninf;
            s[kb_][r] = v;
            rmax = fmaxf(rmax, v);
        }
    rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
    float rsum = 0.f, psum = 0.f;
    const float mq = (float)m / (float)ml;
    float pe[NB][16];   // the un-normalised prior, kept for the second pass (an expf and a division per element)
#pragma unroll
    for (int kb_ = 0; kb_ < NB; ++kb_)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            rsum += as_exp<kFast>(s[kb_][r] - rmax);  // exp(-inf) = 0 for key >= L
            // un-normalised diagonal prior (alignment.py:22-32): exp(-(t/T - m/M)^2 / (2 * 0.1^2)), 0 outside the lengths
            pe[kb_][r] = 0.f;
            if (key < tl && m < ml) {
                const float g = (float)key / (float)tl - mq;
                pe[kb_][r] = as_exp<kFast>(-(g * g) / (2.0f * 0.1f * 0.1f));
                psum += pe[kb_][r];
            }
        }
    rsum += __shfl_xor(rsum, 32, 64);
    psum += __shfl_xor(psum, 32, 64);
    const float lse = rmax + as_log<kFast>(rsum);
    const float pinv = 1.0f / (psum + 1e-5f);

    // ---- logits = log_softmax + log(prior + 1e-6); soft = mask * softmax(logits with masked keys)
    float smax = ninf;
#pragma unroll
    for (int kb_ = 0; kb_ < NB; ++kb_)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float pr = 0.f;
            if (key < tl && m < ml) {
                pr = pe[kb_][r] * pinv;
                pr = pr < 1e-4f ? 0.f : pr;
            }
            const float lg = (s[kb_][r] - lse) + as_log<kFast>(pr + 1e-6f);
            s[kb_][r] = lg;  // keys >= L: -inf (never stored)
            if (key < tl) smax = fmaxf(smax, lg);
        }
    smax = fmaxf(smax, __shfl_xor(smax, 32, 64));
    float esum = 0.f;
#pragma unroll
    for (int kb_ = 0; kb_ < NB; ++kb_)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (key < tl) esum += as_exp<kFast>(s[kb_][r] - smax);
        }
    esum += __shfl_xor(esum, 32, 64);
    const float einv = (m < ml) ? 1.0f / esum : 0.f;

    // ---- transpose each 32-key block through the wave's LDS patch and write 128-B row segments of both outputs
    float* lb = logits + (int64_t)b * M * L;
    float* sbp = soft + (int64_t)b * M * L;
    const int c = lane & 7;
    const bool vec = (L % 4) == 0;
#pragma unroll
    for (int kb_ = 0; kb_ < NB; ++kb_) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                float* pv = &v.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kb_ * 32 + 8 * g + 4 * h + e;
                    const float lg = s[kb_][4 * g + e];
                    pv[e] = pass == 0 ? lg : (key < tl ? as_exp<kFast>(lg - smax) * einv : 0.f);
                }
                *reinterpret_cast<float4*>(stage + l31 * 144 + (8 * g + 4 * h) * 4) = v;
            }
            float* dst = pass == 0 ? lb : sbp;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 8 * i + (lane >> 3), mm = m0 + r, key = kb_ * 32 + 4 * c;
                const float4 v = *reinterpret_cast<const float4*>(stage + r * 144 + c * 16);
                if (mm < M && key < L) {
                    float* o = dst + (int64_t)mm * L + key;
                    if (vec) {
                        *reinterpret_cast<float4*>(o) = v;
                    } else {
                        const float* pv = &v.x;
                        for (int e = 0; e < 4 && key + e < L; ++e) o[e] = pv[e];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ soft averages
// TemporalAverager, soft branch (temporal_adaptor.py:446-449), for pitch and energy at once, plus log1p(duration):
// feats[b][l][0] = log1p(dur[b][l]); feats[b][l][1+f] = mask * sum_m x_f[b][m] A[b][m][l] / (sum_m A[b][m][l] + 1e-5)
// grid (ceil(L/64), B), 1024 threads = 16 frame lanes x 64 text columns (only 2 x B workgroups exist, so each carries as
// many loads in flight as a workgroup can: 8 independent frames per thread and trip).
constexpr int kSaLanes = 16;
__global__ __launch_bounds__(64 * kSaLanes) void soft_average_kernel(const float* __restrict__ attn,
                                                                     const float* __restrict__ pitch,
                                                                     const float* __restrict__ energy,
                                                                     const int64_t* __restrict__ dur,
                                                                     const int64_t* __restrict__ text_len,
                                                                     float* __restrict__ feats, int M, int L) {
    __shared__ float red[3][kSaLanes][64];
    const int b = blockIdx.y, j = threadIdx.x & 63, l = blockIdx.x * 64 + j, ml = threadIdx.x >> 6;
    const bool ok = l < L;
    const float* ab = attn + (int64_t)b * M * L;
    const float* pb = pitch + (int64_t)b * M;
    const float* eb = energy + (int64_t)b * M;
    float sa = 0.f, sp = 0.f, se = 0.f;
    if (ok) {
        int mm = ml;
        for (; mm + 7 * kSaLanes < M; mm += 8 * kSaLanes) {
            float a[8], pv[8], ev[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = ab[(int64_t)(mm + u * kSaLanes) * L + l];
                pv[u] = pb[mm + u * kSaLanes];
                ev[u] = eb[mm + u * kSaLanes];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sa += a[u];
                sp = fmaf(pv[u], a[u], sp);
                se = fmaf(ev[u], a[u], se);
            }
        }
        for (; mm < M; mm += kSaLanes) {
            const float a = ab[(int64_t)mm * L + l];
            sa += a;
            sp = fmaf(pb[mm], a, sp);
            se = fmaf(eb[mm], a, se);
        }
    }
    red[0][ml][j] = sa;
    red[1][ml][j] = sp;
    red[2][ml][j] = se;
    __syncthreads();
    if (ml == 0 && ok) {
        float a = 0.f, pp = 0.f, ee = 0.f;
#pragma unroll
        for (int u = 0; u < kSaLanes; ++u) {
            a += red[0][u][j];
            pp += red[1][u][j];
            ee += red[2][u][j];
        }
        const float mk = l < (int)text_len[b] ? 1.0f : 0.0f;
        float* f = feats + ((int64_t)b * L + l) * 3;
        f[0] = dur ? log1pf((float)dur[(int64_t)b * L + l]) : 0.0f;
        f[1] = pp / (a + 1e-5f) * mk;
        f[2] = ee / (a + 1e-5f) * mk;
    }
}


// ------------------------------------------------------------------------------------------------ flow-matching algebra
// (temporal_adaptor.py:120-147 of the reference; [B][L][C] tensors with C = 3: a dozen PyTorch element-wise launches each.)
// flow_mix:  x_t = (1 - (1 - sigma) t_b) x0 + t_b x1 ;  flow = x1 - (1 - sigma) x0      one rounding per operation, in the
//            reference's order (no FMA contraction), s = float(1 - sigma) as PyTorch wraps the Python scalar
__global__ __launch_bounds__(256) void flow_mix_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                       const float* __restrict__ t, float s, float* __restrict__ xt,
                                                       float* __restrict__ flow, int per_batch, int64_t total) {
#pragma clang fp contract(off)   // HIP's __fmul_rn / __fadd_rn are plain operators and hipcc contracts them into FMAs by default
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float tb = t[i / per_batch], a = x0[i], b = x1[i];
    const float w0 = 1.0f - s * tb;
    const float p0 = w0 * a, p1 = tb * b;
    xt[i] = p0 + p1;
    const float sa = s * a;
    flow[i] = b - sa;
}

// flow_finish (ONE workgroup for the whole batch: B * L * C is a few thousand values, and the loss needs the mean over the
//            batch):  pf = raw * m ;  pred = (x0 + pf) * m ;  dur = max(exp(pred[..., 0]) - 1, 0) ;
//            ratio[b] = sum_{valid l, c} (pf - flow)^2 / max(C * #valid, 1e-5)   (utils.masked_mean before its .mean()) ;
//            loss = mean_b ratio[b]                                               (its .mean(), summed in index order)
// ONE workgroup (the mean over the batch needs every ratio, and a fixed summation order): 16 lanes per utterance, 64
// utterances per pass of the 1024 threads; a lane's elements are independent loads (unrolled), the 16 partial sums of an
// utterance meet in a fixed shuffle tree.  (The first version walked an utterance with one wave, four utterances per wave in
// turn: 20 dependent load rounds = 36 us for 19,200 elements, all of it at the very end of the forward's side branch.)
__global__ __launch_bounds__(1024) void flow_finish_kernel(const float* __restrict__ raw, const float* __restrict__ flow,
                                                           const float* __restrict__ x0, const uint8_t* __restrict__ mask,
                                                           float* __restrict__ pred, float* __restrict__ dur,
                                                           float* __restrict__ ratio, float* __restrict__ loss, int B, int L,
                                                           int C) {
    __shared__ float sratio[1024];
    const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    const int n = L * C;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + grp;
        float num = 0.f, den = 0.f;
        if (b < B) {
            const float* rb = raw + (int64_t)b * n;
            const float* fb = flow + (int64_t)b * n;
            const float* xb = x0 + (int64_t)b * n;
            const uint8_t* mb = mask + (int64_t)b * L;
            float* pb = pred + (int64_t)b * n;
#pragma unroll 4
            for (int e = sub; e < n; e += 16) {
                const int l = e / C, c = e - l * C;
                const bool m = mb[l] != 0;
                const float r = rb[e], f = fb[e], x = xb[e];
                const float pf = m ? r : 0.f;
                const float pr = m ? x + pf : 0.f;
                pb[e] = pr;
                if (c == 0) dur[(int64_t)b * L + l] = fmaxf(expf(pr) - 1.0f, 0.f);
                const float d = pf - f;
                num += m ? d * d : 0.f;
                den += m ? 1.0f : 0.f;
            }
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            num += __shfl_xor(num, off, 64);
            den += __shfl_xor(den, off, 64);
        }
        if (b < B && sub == 0) {
            const float q = num / fmaxf(den, 1e-5f);
            ratio[b] = q;
            if (b < 1024) sratio[b] = q;
        }
    }
    __syncthreads();
    if (loss && tid == 0) {
        float s = 0.f;
        if (B <= 1024) {
            for (int b = 0; b < B; ++b) s += sratio[b];
        } else {                                  // (global writes of this workgroup are visible to it after the barrier)
            for (int b = 0; b < B; ++b) s += ratio[b];
        }
        loss[0] = s / (float)B;
    }
}


// ------------------------------------------------------------------------------------------------ the predictor's head in one pass
// What ends FlowTransformerTemporalModule.forward (temporal_adaptor.py:127-147) after the stack's last layer: the stack's final
// LayerNorm (transformer.py:205-206, with its row mask), the 256 -> 3 `linear_layer`, and the flow-matching algebra of
// flow_finish_kernel - three launches that trail the whole forward (they are the LAST kernels of its side branch, after the
// decoder has finished) as two: flow_head_kernel (one wave per 4 rows: everything per row) leaves per-workgroup partial sums of
// the masked squared error, flow_head_finalize_kernel adds them in block order per utterance and takes the batch mean in index
// order (fixed orders: bit-reproducible).  D = 256 (one float4 per lane), C = 3.
constexpr int kFhRows = 16;      // rows per workgroup: 4 waves x 4 rows
__device__ __forceinline__ float fh_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__global__ __launch_bounds__(256) void flow_head_kernel(const float* __restrict__ y, int64_t ldy, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const float* __restrict__ flow,
                                                        const float* __restrict__ x0, const uint8_t* __restrict__ mask,
                                                        float* __restrict__ pred, float* __restrict__ dur,
                                                        float* __restrict__ part, int L, int nblk) {
#pragma clang fp contract(off)
    __shared__ float sp[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, blk = blockIdx.x;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + 4 * lane), b4 = *reinterpret_cast<const f32x4*>(beta + 4 * lane);
    f32x4 w4[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) w4[c] = *reinterpret_cast<const f32x4*>(W + c * 256 + 4 * lane);
    const float bc[3] = {bias[0], bias[1], bias[2]};
    float num = 0.f, den = 0.f;
    const int l0 = blk * kFhRows + wave * 4;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int l = l0 + i < L ? l0 + i : L - 1;          // rows past the end: a valid row, never stored
        v[i] = *reinterpret_cast<const f32x4*>(y + ((int64_t)b * L + l) * ldy + 4 * lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int l = l0 + i;
        const bool live = l < L;
        const int64_t r = (int64_t)b * L + (live ? l : L - 1);
        const bool m = mask[r] != 0;
        const float mean = fh_wave_sum((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) * (1.0f / 256.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            q = fmaf(d, d, q);
        }
        const float rstd = 1.0f / sqrtf(fh_wave_sum(q) * (1.0f / 256.0f) + eps);
        float h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = m ? fmaf((v[i][e] - mean) * rstd, g4[e], b4[e]) : 0.f;      // (LayerNorm's row mask)
        float raw[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) a = fmaf(h[e], w4[c][e], a);
            raw[c] = fh_wave_sum(a) + bc[c];
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float f = flow[r * 3 + c], x = x0[r * 3 + c];
                const float pf = m ? raw[c] : 0.f;
                const float pr = m ? x + pf : 0.f;
                if (lane == c) pred[r * 3 + c] = pr;
                if (c == 0 && lane == 0) dur[r] = fmaxf(expf(pr) - 1.0f, 0.f);
                const float d = pf - f;
                num += m ? d * d : 0.f;
                den += m ? 1.0f : 0.f;
            }
        }
    }
    if (lane == 0) { sp[wave][0] = num; sp[wave][1] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t o = ((int64_t)b * nblk + blk) * 2;
        part[o] = (sp[0][0] + sp[1][0]) + (sp[2][0] + sp[3][0]);
        part[o + 1] = (sp[0][1] + sp[1][1]) + (sp[2][1] + sp[3][1]);
    }
}
__global__ __launch_bounds__(1024) void flow_head_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ ratio,
                                                                  float* __restrict__ loss, int B) {
    __shared__ float sratio[1024];
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int b = b0 + tid;
        if (b < B) {
            float num = 0.f, den = 0.f;
            for (int k = 0; k < nblk; ++k) {
                num += part[((int64_t)b * nblk + k) * 2];
                den += part[((int64_t)b * nblk + k) * 2 + 1];
            }
            const float q = num / fmaxf(den, 1e-5f);
            ratio[b] = q;
            sratio[tid] = q;
        }
        __syncthreads();
        if (tid == 0) {
            const int n = B - b0 < 1024 ? B - b0 : 1024;
            for (int i = 0; i < n; ++i) s += sratio[i];
        }
        __syncthreads();
    }
    if (loss && tid == 0) loss[0] = s / (float)B;
}

// ------------------------------------------------------------------------------------------------ infer: Euler update, features
// FlowTransformerTemporalModule.infer (temporal_adaptor.py:158-170): x_t <- x_t + velocity * dt per step, `* mask` after the
// last one.  dt comes from the host (the warped grid depends only on (steps, step_factor): :150-156).
__global__ __launch_bounds__(256) void flow_euler_kernel(const float* __restrict__ xt, const float* __restrict__ vel, float dt,
                                                         const uint8_t* __restrict__ mask, float* __restrict__ out, int C,
                                                         int64_t total) {
#pragma clang fp contract(off)   // the reference rounds the product, then the sum
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float step = vel[i] * dt;
    float v = xt[i] + step;
    if (mask) v = mask[i / C] ? v : v * 0.0f;
    out[i] = v;
}

// FlowTemporalAdaptor.infer (temporal_adaptor.py:351-384) between the predictor and the embedding stack, one thread per token:
//   duration = clamp(duration_factor * (exp(pred[..., 0]) - 1), min 0), replaced by the target where one is given (>= 0; the
//              reference fills only the negative entries of a target with predictions, :355-362)
//   features = [ (pitch_target | pred[..., 1]) * pitch_factor + pitch_delta, (energy_target | pred[..., 2]) * ef + ed ]
__global__ __launch_bounds__(256) void infer_features_kernel(const float* __restrict__ pred, const float* __restrict__ dur_f,
                                                             const int64_t* __restrict__ dur_i, const float* __restrict__ pitch_t,
                                                             const float* __restrict__ energy_t, float df, float pf, float pd,
                                                             float ef, float ed, float* __restrict__ duration,
                                                             float* __restrict__ feats, int64_t n) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float e = expf(pred[3 * i]) - 1.0f;
    float d = fmaxf(df * e, 0.0f);
    if (dur_f) { const float t = dur_f[i]; d = t < 0.0f ? d : t; }
    if (dur_i) { const int64_t t = dur_i[i]; d = t < 0 ? d : (float)t; }
    duration[i] = d;
    const float p = pitch_t ? pitch_t[i] : pred[3 * i + 1], q = energy_t ? energy_t[i] : pred[3 * i + 2];
    const float pm = p * pf, qm = q * ef;
    feats[2 * i] = pm + pd;
    feats[2 * i + 1] = qm + ed;
}

}  // namespace

extern "C" int32_t ispk_pad_rows_f32(const float* x, int64_t stride_b, int64_t stride_t, int64_t stride_c,
                                     const int64_t* len, void* out, int32_t out_bf16, int32_t B, int32_t T, int32_t C,
                                     ispk_stream_t stream) {
    ISPK_REQUIRE(x && len && out, ISPK_E_NULL, "pad_rows: null pointer");
    ISPK_REQUIRE(B >= 0 && T >= 1 && C >= 1, ISPK_E_SHAPE, "pad_rows: bad shape B=%d T=%d C=%d", B, T, C);
    if (B == 0) return 0;
    const int64_t total = (int64_t)B * (T + 4) * C;
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ISPK_REQUIRE(out_bf16 >= 0 && out_bf16 <= 2, ISPK_E_UNSUPPORTED, "pad_rows: output mode %d (0 fp32, 1 bf16, 2 split fp16 planes)", out_bf16);
    if (stride_t == 1 && stride_c > 1 && B <= 65535) {   // channel-first: transposing tile kernel
        dim3 gcf((T + 4 + 31) / 32, (C + 31) / 32, B), bcf(32, 8);
        if (out_bf16 == 2)
            hipLaunchKernelGGL(pad_rows_cf_kernel<_Float16>, gcf, bcf, 0, s, x, stride_b, stride_c, len,
                               static_cast<_Float16*>(out), T, C, total);
        else if (out_bf16)
            hipLaunchKernelGGL(pad_rows_cf_kernel<uint16_t>, gcf, bcf, 0, s, x, stride_b, stride_c, len,
                               static_cast<uint16_t*>(out), T, C, (int64_t)0);
        else
            hipLaunchKernelGGL(pad_rows_cf_kernel<float>, gcf, bcf, 0, s, x, stride_b, stride_c, len,
                               static_cast<float*>(out), T, C, (int64_t)0);
        return ispk_launch_status();
    }
    if (out_bf16 == 2)
        hipLaunchKernelGGL(pad_rows_kernel<_Float16>, grid, dim3(256), 0, s, x, stride_b, stride_t, stride_c, len,
                           static_cast<_Float16*>(out), T, C, total, total);
    else if (out_bf16)
        hipLaunchKernelGGL(pad_rows_kernel<uint16_t>, grid, dim3(256), 0, s, x, stride_b, stride_t, stride_c, len,
                           static_cast<uint16_t*>(out), T, C, total, (int64_t)0);
    else
        hipLaunchKernelGGL(pad_rows_kernel<float>, grid, dim3(256), 0, s, x, stride_b, stride_t, stride_c, len,
                           static_cast<float*>(out), T, C, total, (int64_t)0);
    return ispk_launch_status();
}

extern "C" int32_t ispk_masked_instnorm_f32(const float* y, const float* weight, const float* bias, const int64_t* len,
                                            void* out, int32_t out_bf16, int32_t B, int32_t T, int32_t C, float eps,
                                            ispk_stream_t stream) {
    ISPK_REQUIRE(y && weight && bias && len && out, ISPK_E_NULL, "masked_instnorm: null pointer");
    ISPK_REQUIRE(B >= 0 && T >= 1 && C >= 1, ISPK_E_SHAPE, "masked_instnorm: bad shape B=%d T=%d C=%d", B, T, C);
    ISPK_REQUIRE(B <= 65535, ISPK_E_SHAPE, "masked_instnorm: B=%d exceeds the grid limit", B);
    if (B == 0) return 0;
    dim3 grid((C + 63) / 64, B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ISPK_REQUIRE(out_bf16 >= 0 && out_bf16 <= 2, ISPK_E_UNSUPPORTED, "masked_instnorm: output mode %d (0 fp32, 1 bf16, 2 split fp16 planes)", out_bf16);
    if (out_bf16 == 2)
        hipLaunchKernelGGL(masked_instnorm_kernel<_Float16>, grid, dim3(1024), 0, s, y, weight, bias, len,
                           static_cast<_Float16*>(out), T, C, eps, (int64_t)B * (T + 4) * C);
    else if (out_bf16)
        hipLaunchKernelGGL(masked_instnorm_kernel<uint16_t>, grid, dim3(1024), 0, s, y, weight, bias, len,
                           static_cast<uint16_t*>(out), T, C, eps, (int64_t)0);
    else
        hipLaunchKernelGGL(masked_instnorm_kernel<float>, grid, dim3(1024), 0, s, y, weight, bias, len,
                           static_cast<float*>(out), T, C, eps, (int64_t)0);
    return ispk_launch_status();
}

static int32_t aligner_scores_launch(const float* q_enc, int64_t q_stride_b, const float* k_enc, int64_t k_stride_b,
                                     const int64_t* text_len, const int64_t* mel_len, float* attn_logits, float* attn_soft,
                                     int32_t B, int32_t M, int32_t L, int32_t D, ispk_stream_t stream, bool fast) {
    ISPK_REQUIRE(q_enc && k_enc && text_len && mel_len && attn_logits && attn_soft, ISPK_E_NULL,
                 "aligner_scores: null pointer");
    ISPK_REQUIRE(D == kAD, ISPK_E_UNSUPPORTED, "aligner_scores: attention_dim %d (built for 128)", D);
    ISPK_REQUIRE(B >= 0 && M >= 1 && L >= 1 && L <= 320, ISPK_E_SHAPE, "aligner_scores: bad shape B=%d M=%d L=%d (L <= 320)",
                 B, M, L);
    ISPK_REQUIRE(B <= 65535, ISPK_E_SHAPE, "aligner_scores: B=%d exceeds the grid limit", B);
    ISPK_REQUIRE(ispk_aligned(q_enc, 16) && ispk_aligned(k_enc, 16) && ispk_aligned(attn_logits, 16) &&
                     ispk_aligned(attn_soft, 16) && q_stride_b % 4 == 0 && k_stride_b % 4 == 0,
                 ISPK_E_ALIGN, "aligner_scores: 16-byte alignment required");
    if (B == 0) return 0;
    const int nb = (L + 31) / 32;
    size_t lds = (size_t)nb * 32 * kAD * 4;
    if (lds < kAsWaves * 32 * 144) lds = kAsWaves * 32 * 144;
    const float scale = 1.0f / sqrtf((float)D);
    dim3 grid((M + 32 * kAsWaves - 1) / (32 * kAsWaves), B), block(64 * kAsWaves);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define ISPK_AL_CASE(NB)                                                                                              \
    case NB:                                                                                                          \
        if (fast) {                                                                                                   \
            ISPK_RESERVE_LDS((&aligner_scores_kernel<NB, true>), lds, "aligner_scores");                              \
            hipLaunchKernelGGL((aligner_scores_kernel<NB, true>), grid, block, lds, s, q_enc, q_stride_b, k_enc,      \
                               k_stride_b, text_len, mel_len, attn_logits, attn_soft, M, L, scale);                   \
        } else {                                                                                                      \
            ISPK_RESERVE_LDS((&aligner_scores_kernel<NB, false>), lds, "aligner_scores");                             \
            hipLaunchKernelGGL((aligner_scores_kernel<NB, false>), grid, block, lds, s, q_enc, q_stride_b, k_enc,     \
                               k_stride_b, text_len, mel_len, attn_logits, attn_soft, M, L, scale);                   \
        }                                                                                                             \
        break;
    switch (nb) {
        ISPK_AL_CASE(1) ISPK_AL_CASE(2) ISPK_AL_CASE(3) ISPK_AL_CASE(4) ISPK_AL_CASE(5) ISPK_AL_CASE(6) ISPK_AL_CASE(7)
        ISPK_AL_CASE(8) ISPK_AL_CASE(9) ISPK_AL_CASE(10)
        default: ISPK_FAIL(ISPK_E_SHAPE, "aligner_scores: unsupported L=%d", L);
    }
#undef ISPK_AL_CASE
    return ispk_launch_status();
}

extern "C" int32_t ispk_aligner_scores_f32(const float* q_enc, int64_t q_stride_b, const float* k_enc,
                                           int64_t k_stride_b, const int64_t* text_len, const int64_t* mel_len,
                                           float* attn_logits, float* attn_soft, int32_t B, int32_t M, int32_t L,
                                           int32_t D, ispk_stream_t stream) {
    return aligner_scores_launch(q_enc, q_stride_b, k_enc, k_stride_b, text_len, mel_len, attn_logits, attn_soft, B, M, L, D, stream,
                                 false);
}

extern "C" int32_t ispk_aligner_scores_fast_f32(const float* q_enc, int64_t q_stride_b, const float* k_enc,
                                                int64_t k_stride_b, const int64_t* text_len, const int64_t* mel_len,
                                                float* attn_logits, float* attn_soft, int32_t B, int32_t M, int32_t L,
                                                int32_t D, ispk_stream_t stream) {
    return aligner_scores_launch(q_enc, q_stride_b, k_enc, k_stride_b, text_len, mel_len, attn_logits, attn_soft, B, M, L, D, stream,
                                 true);
}

extern "C" int32_t ispk_soft_average_f32(const float* attn_soft, const float* pitch, const float* energy,
                                         const int64_t* duration, const int64_t* text_len, float* feats, int32_t B,
                                         int32_t M, int32_t L, ispk_stream_t stream) {
    ISPK_REQUIRE(attn_soft && pitch && energy && text_len && feats, ISPK_E_NULL, "soft_average: null pointer");
    ISPK_REQUIRE(B >= 0 && M >= 1 && L >= 1 && B <= 65535, ISPK_E_SHAPE, "soft_average: bad shape B=%d M=%d L=%d", B, M, L);
    if (B == 0) return 0;
    hipLaunchKernelGGL(soft_average_kernel, dim3((L + 63) / 64, B), dim3(64 * kSaLanes), 0, reinterpret_cast<hipStream_t>(stream),
                       attn_soft, pitch, energy, duration, text_len, feats, M, L);
    return ispk_launch_status();
}

extern "C" int32_t ispk_flow_mix_f32(const float* x0, const float* x1, const float* t, float sigma, float* x_t, float* flow,
                                     int32_t B, int32_t L, int32_t C, ispk_stream_t stream) {
    ISPK_REQUIRE(x0 && x1 && t && x_t && flow, ISPK_E_NULL, "flow_mix: null pointer");
    ISPK_REQUIRE(B >= 0 && L >= 1 && C >= 1, ISPK_E_SHAPE, "flow_mix: bad shape B=%d L=%d C=%d", B, L, C);
    if (B == 0) return 0;
    const int64_t total = (int64_t)B * L * C;
    const float s = (float)(1.0 - (double)sigma);
    hipLaunchKernelGGL(flow_mix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x0, x1, t, s, x_t, flow, L * C, total);
    return ispk_launch_status();
}

extern "C" int32_t ispk_flow_finish_f32(const float* pred_raw, const float* flow, const float* x0, const uint8_t* mask,
                                        float* pred, float* duration, float* loss_ratio, float* loss_mean, int32_t B,
                                        int32_t L, int32_t C, ispk_stream_t stream) {
    ISPK_REQUIRE(pred_raw && flow && x0 && mask && pred && duration && loss_ratio, ISPK_E_NULL, "flow_finish: null pointer");
    ISPK_REQUIRE(B >= 0 && L >= 1 && C >= 1 && B <= 65535, ISPK_E_SHAPE, "flow_finish: bad shape B=%d L=%d C=%d", B, L, C);
    if (B == 0) return 0;
    hipLaunchKernelGGL(flow_finish_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), pred_raw, flow, x0,
                       mask, pred, duration, loss_ratio, loss_mean, B, L, C);
    return ispk_launch_status();
}

extern "C" int32_t ispk_flow_head_f32(const float* y, int64_t ldy, const float* norm_gamma, const float* norm_beta, float norm_eps,
                                      const float* W, const float* bias, const float* flow, const float* x0, const uint8_t* mask,
                                      float* pred, float* duration, float* loss_ratio, float* loss_mean, float* workspace,
                                      int32_t B, int32_t L, int32_t D, int32_t C, ispk_stream_t stream) {
    ISPK_REQUIRE(y && norm_gamma && norm_beta && W && bias && flow && x0 && mask && pred && duration && loss_ratio && workspace,
                 ISPK_E_NULL, "flow_head: null pointer");
    ISPK_REQUIRE(D == 256 && C == 3, ISPK_E_UNSUPPORTED, "flow_head: built for dim 256 -> 3 flow channels (got %d -> %d)", D, C);
    ISPK_REQUIRE(B >= 0 && L >= 1 && B <= 65535 && ldy >= D && ldy % 4 == 0, ISPK_E_SHAPE, "flow_head: bad shape B=%d L=%d", B, L);
    ISPK_REQUIRE(ispk_aligned(y, 16) && ispk_aligned(norm_gamma, 16) && ispk_aligned(norm_beta, 16) && ispk_aligned(W, 16), ISPK_E_ALIGN,
                 "flow_head: y / gamma / beta / W must be 16-byte aligned");
    if (B == 0) return 0;
    const int nblk = (L + kFhRows - 1) / kFhRows;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(flow_head_kernel, dim3(nblk, B), dim3(256), 0, s, y, ldy, norm_gamma, norm_beta, norm_eps, W, bias, flow, x0, mask,
                       pred, duration, workspace, L, nblk);
    if (int32_t rc = ispk_launch_status()) return rc;
    hipLaunchKernelGGL(flow_head_finalize_kernel, dim3(1), dim3(1024), 0, s, workspace, nblk, loss_ratio, loss_mean, B);
    return ispk_launch_status();
}

extern "C" int32_t ispk_flow_euler_f32(const float* x_t, const float* velocity, float dt, const uint8_t* mask, float* out,
                                       int32_t B, int32_t L, int32_t C, ispk_stream_t stream) {
    ISPK_REQUIRE(x_t && velocity && out, ISPK_E_NULL, "flow_euler: null pointer");
    ISPK_REQUIRE(B >= 0 && L >= 1 && C >= 1, ISPK_E_SHAPE, "flow_euler: bad shape B=%d L=%d C=%d", B, L, C);
    if (B == 0) return 0;
    const int64_t total = (int64_t)B * L * C;
    hipLaunchKernelGGL(flow_euler_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x_t, velocity, dt, mask, out, C, total);
    return ispk_launch_status();
}

extern "C" int32_t ispk_infer_features_f32(const float* pred, const float* duration_target_f32, const int64_t* duration_target_i64,
                                           const float* pitch_target, const float* energy_target, float duration_factor,
                                           float pitch_factor, float pitch_delta, float energy_factor, float energy_delta,
                                           float* duration, float* features, int32_t B, int32_t L, ispk_stream_t stream) {
    ISPK_REQUIRE(pred && duration && features, ISPK_E_NULL, "infer_features: null pointer");
    ISPK_REQUIRE(!(duration_target_f32 && duration_target_i64), ISPK_E_UNSUPPORTED, "infer_features: one duration target at most");
    ISPK_REQUIRE(B >= 0 && L >= 1, ISPK_E_SHAPE, "infer_features: bad shape B=%d L=%d", B, L);
    if (B == 0) return 0;
    const int64_t n = (int64_t)B * L;
    hipLaunchKernelGGL(infer_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), pred, duration_target_f32, duration_target_i64, pitch_target,
                       energy_target, duration_factor, pitch_factor, pitch_delta, energy_factor, energy_delta, duration, features, n);
    return ispk_launch_status();
}
