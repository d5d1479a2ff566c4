// The small steps between the transformer stacks of AcousticModel.forward / .infer, as gfx950 kernels (SURVEY rows a13,
// a15, f3): token embedding + key mask, the flow predictor's time embedding, soft length regulation (optionally from a
// soft path generated on the fly) with the decoder lengths and mask.  In the reference each is a handful of ATen
// element-wise launches or a library bmm; here every one is a single launch on the caller's stream.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ token embedding + mask
// models/acoustic/model.py:131-134: emb = Embedding(text) (a row gather; row 0 of the table is the padding row and is
// simply read), enc_mask = arange(L) < text_len[:, None] (utils/functions.py:61-65).
// One wave per token row: D/4 float4 loads of the table row, same stores.  Ids outside [0, vocab) cannot raise on the
// device like F.embedding does on the host; they read the padding row (row 0).
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* __restrict__ text, const float* __restrict__ table,
                                                           int64_t ld_table, int vocab, const int64_t* __restrict__ text_len,
                                                           float* __restrict__ emb, uint8_t* __restrict__ mask, int rows,
                                                           int L, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    int64_t id = text[row];
    id = (id < 0 || id >= vocab) ? 0 : id;
    const f32x4* src = reinterpret_cast<const f32x4*>(table + id * ld_table);
    f32x4* dst = reinterpret_cast<f32x4*>(emb + (int64_t)row * D);
    for (int c = lane; c < D / 4; c += 64) dst[c] = src[c];
    if (mask && lane == 0) {
        const int b = row / L, l = row - b * L;
        mask[row] = text_len ? (uint8_t)(l < text_len[b]) : (uint8_t)1;
    }
}

// ------------------------------------------------------------------------------------------------ speaker embedding
// models/acoustic/model.py:205-207 (`infer`): enc_out = enc_out + speaker_embedding(speaker) - nn.Embedding rows broadcast
// over the text axis.  In place, every row of the utterance (padded ones too: the reference adds before any re-masking).
// One wave per row; speaker ids at stride `id_stride` (0: one id for the whole batch).  Ids outside [0, speakers) are
// clamped (the host-side F.embedding would raise).
__global__ __launch_bounds__(256) void add_speaker_kernel(float* __restrict__ x, const float* __restrict__ table, int64_t ld_table,
                                                          int speakers, const int64_t* __restrict__ speaker, int id_stride,
                                                          int rows, int L, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    int64_t id = speaker[(int64_t)(row / L) * id_stride];
    id = id < 0 ? 0 : (id >= speakers ? speakers - 1 : id);
    const f32x4* src = reinterpret_cast<const f32x4*>(table + id * ld_table);
    f32x4* dst = reinterpret_cast<f32x4*>(x + (int64_t)row * D);
    for (int c = lane; c < D / 4; c += 64) {
        f32x4 v = dst[c];
        const f32x4 e = src[c];
        v[0] += e[0]; v[1] += e[1]; v[2] += e[2]; v[3] += e[3];
        dst[c] = v;
    }
}

// ------------------------------------------------------------------------------------------------ time embedding
// modules/transformer/embeddings.py:131-157 as built at temporal_adaptor.py:87-89 (freq_dim 64, with_steps):
//   f = [t, sin(t * freq_scale * inv_freq[0..H)), cos(...)]  (1 + 2H values; the reference multiplies in this order)
//   out = W1 silu(W0 f + b0) + b1
// One wave per time value; lane j < E owns hidden unit j, then output j (E <= 64).
__global__ __launch_bounds__(64) void time_embedding_kernel(const float* __restrict__ t, const float* __restrict__ inv_freq,
                                                            const float* __restrict__ freq_scale, int H,
                                                            const float* __restrict__ w0, const float* __restrict__ b0,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            int E, float* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ float f[1 + 2 * 64];
    __shared__ float h[64];
    const int n = blockIdx.x, j = threadIdx.x;
    const float pos = t[n], fs = freq_scale[0];
    if (j == 0) f[0] = pos;
    for (int i = j; i < H; i += 64) {
        const float a = pos * fs * inv_freq[i];
        f[1 + i] = sinf(a);
        f[1 + H + i] = cosf(a);
    }
    __syncthreads();
    const int K0 = 1 + 2 * H;
    if (j < E) {
        float acc = b0[j];
        for (int k = 0; k < K0; ++k) acc = fmaf(f[k], w0[(int64_t)j * K0 + k], acc);
        h[j] = acc / (1.0f + expf(-acc));   // SiLU
    }
    __syncthreads();
    if (j < E) {
        float acc = b1[j];
        for (int k = 0; k < E; ++k) acc = fmaf(h[k], w1[(int64_t)j * E + k], acc);
        out[(int64_t)n * E + j] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ length regulation
// models/acoustic/modules/temporal_adaptor.py:411-436 (LengthRegulator, soft branch) and :468-478 (generate_soft_path):
//   out[b][y][:] = sum_t A[b][y][t] * x[b][t][:]        A = the aligner's attn_soft (forward), or the soft path
//   dec_len[b]   = (sum_t dur[b][t] + 0.5).long()  [clamped to max_len in forward]
//   soft path (infer): cum = cumsum(dur);  P[t][y] = clamp(cum[t] - y, 0, 1) - clamp(cum[t-1] - y, 0, 1), cum[-1] -> 0 row,
//                      A[y][t] = P[t][y] * (t < enc_len[b]) * (y < dec_len[b])
// A workgroup owns 64 frames x all D features of one utterance; wave w the features [w*D/4, (w+1)*D/4).  Exact fp32
// products on v_mfma_f32_32x32x2_f32 (this feeds the fp32 parity path too).  The token axis streams through LDS in chunks
// of 16; the next chunk is fetched into registers while the current one multiplies.
constexpr int kLrRows = 64, kLrChunk = 16, kLrAld = kLrChunk + 1;

// kSplit (ispk_length_regulate_split_bf16, the bf16 compute path): every fp32 operand value v is split in registers into
// hi = bf16(v) and lo = bf16(v - hi), and a product is three v_mfma_f32_32x32x16_bf16 (hi hi + hi lo + lo hi; the lo lo term
// is below 2^-16 of the product) instead of eight v_mfma_f32_32x32x2_f32: 18 MFMAs of 32 cycles per 16-token chunk and wave
// against 48 of 64 cycles, at ~2^-16 relative error per product - three decimal digits finer than the bf16 GEMMs the result
// feeds.  The fp32 parity path keeps the exact fp32 MFMAs.
// kSplit == 2 (ispk_length_regulate_split_f16, the split-fp16 parity path): the same three products over fp16 terms - 22
// significant bits per operand, fp32-grade results (csrc/split.hip) - on v_mfma_f32_32x32x16_f16.
typedef uint32_t lr_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 lr_f16x8 __attribute__((ext_vector_type(8)));
template <int kSplit>
__device__ __forceinline__ void lr_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    union { lr_u32x4 u; bf16x8 f; } h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f2 a;
        a.x = v[2 * e]; a.y = v[2 * e + 1];
        if constexpr (kSplit == 2) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            a.x = __builtin_amdgcn_fmed3f(a.x, -65504.0f, 65504.0f);
            a.y = __builtin_amdgcn_fmed3f(a.y, -65504.0f, 65504.0f);
            h2 ph, pl;
            ph.x = (_Float16)a.x; ph.y = (_Float16)a.y;
            pl.x = (_Float16)(a.x - (float)ph.x); pl.y = (_Float16)(a.y - (float)ph.y);
            h.u[e] = __builtin_bit_cast(uint32_t, ph);
            l.u[e] = __builtin_bit_cast(uint32_t, pl);
        } else {
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            const uint32_t ph = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf2));
            f2 r;
            r.x = a.x - __builtin_bit_cast(float, ph << 16);
            r.y = a.y - __builtin_bit_cast(float, ph & 0xffff0000u);
            h.u[e] = ph;
            l.u[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf2));
        }
    }
    hi = h.f;
    lo = l.f;
}
template <int kSplit>
__device__ __forceinline__ f32x16 lr_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (kSplit == 2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(lr_f16x8, a), __builtin_bit_cast(lr_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int NT, int kSplit = 0>   // D = 128 * NT: a wave owns NT 32-feature tiles; kSplit: 0 exact fp32, 1 bf16 terms, 2 fp16 terms
__global__ __launch_bounds__(256) void length_regulate_kernel(const float* __restrict__ align, const float* __restrict__ dur_f,
                                                              const int64_t* __restrict__ dur_i,
                                                              const int64_t* __restrict__ enc_len,
                                                              const float* __restrict__ x, int64_t ldx,
                                                              float* __restrict__ out, int64_t* __restrict__ dec_len,
                                                              uint8_t* __restrict__ dec_mask, int M, int L, int max_len,
                                                              int dur_cols) {
    constexpr int D = 128 * NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t& s_dec = *reinterpret_cast<int64_t*>(smem);                      // (all LDS in the dynamic region: a static
    float* As = reinterpret_cast<float*>(smem + 16);                         //  object would shift its 16-byte alignment)
    float* Xs = As + kLrRows * kLrAld;                                       // [16][D]   (As: [64][17] = 4352 B)
    float* cum = Xs + kLrChunk * D;                                          // [L + 1]   (soft path only): cum[t] = sum_{u<t}
    const int b = blockIdx.y, y0 = blockIdx.x * kLrRows, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool soft = align == nullptr;

    // ---- decoder length of this utterance (every workgroup of the utterance computes the same value; tile 0 stores it)
    if (wave == 0) {
        int64_t dl;
        if (dur_i) {
            int64_t s = 0;
            for (int t = lane; t < dur_cols; t += 64) s += dur_i[(int64_t)b * dur_cols + t];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            dl = (int64_t)((float)s + 0.5f);
        } else {
            // sequential fp32 sum in index order (what a CPU reduction over <= a few hundred values does); lane 0 only
            float s = 0.f;
            if (lane == 0) {
                cum[0] = 0.f;
                for (int t = 0; t < L; ++t) {
                    s += dur_f[(int64_t)b * L + t];
                    cum[t + 1] = s;
                }
            }
            s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s)));
            dl = (int64_t)(s + 0.5f);
        }
        if (max_len >= 0 && dl > max_len) dl = max_len;
        if (lane == 0) {
            s_dec = dl;
            if (blockIdx.x == 0) dec_len[b] = dl;
        }
    }
    __syncthreads();
    const int64_t dl = s_dec;
    if (dec_mask)
        for (int r = tid; r < kLrRows; r += 256)
            if (y0 + r < M) dec_mask[(int64_t)b * M + y0 + r] = (uint8_t)(y0 + r < dl);
    const int el = enc_len ? (int)enc_len[b] : L;

    f32x16 acc[2][NT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;

    const float* xb = x + (int64_t)b * L * ldx;
    const float* ab = soft ? nullptr : align + (int64_t)b * M * L;
    // staging registers: A chunk 64 x 16 = 4 values per thread (thread -> row tid/4, k (tid%4)*4 ..+3); X chunk 16 x D:
    // D/64 float4 per thread (thread -> token tid/16, 4-feature groups (tid%16) + 16*i)
    float ar[4];
    f32x4 xr[D / 64];
    const int a_row = tid >> 2, a_k = (tid & 3) * 4, x_t = tid >> 4, x_c = tid & 15;
    auto fetch = [&](int t0) {
        const int y = y0 + a_row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + a_k + u;
            float v = 0.f;
            if (y < M && t < L) {
                if (!soft) {
                    v = ab[(int64_t)y * L + t];
                } else {
                    const float fy = (float)y;
                    const float hi = fminf(fmaxf(cum[t + 1] - fy, 0.f), 1.f), lo = fminf(fmaxf(cum[t] - fy, 0.f), 1.f);
                    v = (t < el && y < dl) ? hi - lo : 0.f;
                }
            }
            ar[u] = v;
        }
        const int t = t0 + x_t;
#pragma unroll
        for (int i = 0; i < D / 64; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < L) v = *reinterpret_cast<const f32x4*>(xb + (int64_t)t * ldx + (x_c + 16 * i) * 4);
            xr[i] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) As[a_row * kLrAld + a_k + u] = ar[u];
#pragma unroll
        for (int i = 0; i < D / 64; ++i) *reinterpret_cast<f32x4*>(Xs + x_t * D + (x_c + 16 * i) * 4) = xr[i];
    };

    fetch(0);
    for (int t0 = 0; t0 < L; t0 += kLrChunk) {
        __syncthreads();          // everyone is done reading the previous chunk
        stash();
        __syncthreads();
        if (t0 + kLrChunk < L) fetch(t0 + kLrChunk);
        const int r = lane & 31, kh = lane >> 5;
        if constexpr (kSplit != 0) {
            // lane half kh takes tokens 8 kh .. 8 kh + 7 of the chunk in both operands
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float av[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) av[e] = As[(32 * rt + r) * kLrAld + 8 * kh + e];
                lr_split8<kSplit>(av, ah[rt], al[rt]);
            }
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                float bv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) bv[e] = Xs[(8 * kh + e) * D + wave * (32 * NT) + ct * 32 + r];
                bf16x8 bh, bl;
                lr_split8<kSplit>(bv, bh, bl);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    acc[rt][ct] = lr_mfma<kSplit>(al[rt], bh, acc[rt][ct]);
                    acc[rt][ct] = lr_mfma<kSplit>(ah[rt], bl, acc[rt][ct]);
                    acc[rt][ct] = lr_mfma<kSplit>(ah[rt], bh, acc[rt][ct]);
                }
            }
            continue;
        }
#pragma unroll
        for (int kk = 0; kk < kLrChunk; kk += 2) {
            const float a0 = As[r * kLrAld + kk + kh], a1 = As[(32 + r) * kLrAld + kk + kh];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const float bv = Xs[(kk + kh) * D + wave * (32 * NT) + ct * 32 + r];
                acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][ct], 0, 0, 0);
            }
        }
    }
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* ob = out + (int64_t)b * M * D;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int y = y0 + rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (y < M) ob[(int64_t)y * D + wave * (32 * NT) + ct * 32 + (lane & 31)] = acc[rt][ct][i];
            }
}

}  // namespace

extern "C" int32_t ispk_embed_tokens_f32(const int64_t* text, const float* table, int64_t ld_table, int32_t vocab,
                                         const int64_t* text_len, float* emb, uint8_t* mask, int32_t B, int32_t L,
                                         int32_t D, ispk_stream_t stream) {
    ISPK_REQUIRE(text && table && emb, ISPK_E_NULL, "embed_tokens: null pointer");
    ISPK_REQUIRE(B >= 0 && L >= 1 && D >= 4 && vocab >= 1, ISPK_E_SHAPE, "embed_tokens: bad shape B=%d L=%d D=%d V=%d", B, L,
                 D, vocab);
    ISPK_REQUIRE(D % 4 == 0 && ld_table % 4 == 0 && ld_table >= D && ispk_aligned(table, 16) && ispk_aligned(emb, 16),
                 ISPK_E_ALIGN, "embed_tokens: D / ld_table must be multiples of 4 and table / emb 16-byte aligned");
    if (B == 0) return 0;
    const int rows = B * L;
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), text,
                       table, ld_table, vocab, text_len, emb, mask, rows, L, D);
    return ispk_launch_status();
}

extern "C" int32_t ispk_add_speaker_f32(float* x, const float* table, int64_t ld_table, int32_t speakers, const int64_t* speaker,
                                        int32_t id_stride, int32_t B, int32_t L, int32_t D, ispk_stream_t stream) {
    ISPK_REQUIRE(x && table && speaker, ISPK_E_NULL, "add_speaker: null pointer");
    ISPK_REQUIRE(B >= 0 && L >= 1 && D >= 4 && speakers >= 1 && (id_stride == 0 || id_stride == 1), ISPK_E_SHAPE,
                 "add_speaker: bad shape B=%d L=%d D=%d speakers=%d id_stride=%d", B, L, D, speakers, id_stride);
    ISPK_REQUIRE(D % 4 == 0 && ld_table % 4 == 0 && ld_table >= D && ispk_aligned(table, 16) && ispk_aligned(x, 16),
                 ISPK_E_ALIGN, "add_speaker: D / ld_table must be multiples of 4 and table / x 16-byte aligned");
    if (B == 0) return 0;
    const int rows = B * L;
    hipLaunchKernelGGL(add_speaker_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, table,
                       ld_table, speakers, speaker, id_stride, rows, L, D);
    return ispk_launch_status();
}

extern "C" int32_t ispk_time_embedding_f32(const float* t, int32_t n, const float* inv_freq, const float* freq_scale,
                                           int32_t half_dim, const float* w0, const float* b0, const float* w1,
                                           const float* b1, int32_t emb_dim, float* out, ispk_stream_t stream) {
    ISPK_REQUIRE(t && inv_freq && freq_scale && w0 && b0 && w1 && b1 && out, ISPK_E_NULL, "time_embedding: null pointer");
    ISPK_REQUIRE(n >= 0 && half_dim >= 1 && half_dim <= 64 && emb_dim >= 1 && emb_dim <= 64, ISPK_E_SHAPE,
                 "time_embedding: bad shape n=%d half_dim=%d emb_dim=%d (both <= 64)", n, half_dim, emb_dim);
    if (n == 0) return 0;
    hipLaunchKernelGGL(time_embedding_kernel, dim3(n), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), t, inv_freq,
                       freq_scale, half_dim, w0, b0, w1, b1, emb_dim, out);
    return ispk_launch_status();
}

static int32_t length_regulate_launch(const float* alignment, const float* dur_f32, const int64_t* dur_i64, const int64_t* enc_len,
                                      const float* x, int64_t ldx, float* out, int64_t* dec_len, uint8_t* dec_mask, int32_t B,
                                      int32_t M, int32_t L, int32_t D, int32_t max_len, int32_t dur_cols, ispk_stream_t stream,
                                      int split) {
    ISPK_REQUIRE(x && out && dec_len, ISPK_E_NULL, "length_regulate: null pointer");
    ISPK_REQUIRE((dur_f32 != nullptr) != (dur_i64 != nullptr), ISPK_E_NULL,
                 "length_regulate: exactly one of dur_f32 / dur_i64 must be given");
    ISPK_REQUIRE(alignment || dur_f32, ISPK_E_NULL, "length_regulate: the soft path (alignment NULL) needs fp32 durations");
    ISPK_REQUIRE(dur_cols == L || (dur_i64 && dur_cols >= 1), ISPK_E_SHAPE,
                 "length_regulate: dur_cols=%d (L, or any width >= 1 for int64 durations that are only summed)", dur_cols);
    ISPK_REQUIRE(B >= 0 && M >= 1 && L >= 1 && L <= 4096 && B <= 65535, ISPK_E_SHAPE,
                 "length_regulate: bad shape B=%d M=%d L=%d", B, M, L);
    ISPK_REQUIRE(D == 256 || D == 384, ISPK_E_UNSUPPORTED, "length_regulate: dim %d (built for 256 / 384)", D);
    ISPK_REQUIRE(ldx % 4 == 0 && ldx >= D && ispk_aligned(x, 16) && ispk_aligned(out, 16), ISPK_E_ALIGN,
                 "length_regulate: x / out must be 16-byte aligned, ldx a multiple of 4");
    if (B == 0) return 0;
    const size_t lds = 16 + (size_t)(kLrRows * kLrAld + kLrChunk * D + L + 1) * sizeof(float);
    const dim3 grid((M + kLrRows - 1) / kLrRows, B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define ISPK_LR(NT_, SP_)                                                                                              \
    do {                                                                                                               \
        ISPK_RESERVE_LDS((&length_regulate_kernel<NT_, SP_>), lds, "length_regulate");                                 \
        hipLaunchKernelGGL((length_regulate_kernel<NT_, SP_>), grid, dim3(256), lds, s, alignment, dur_f32, dur_i64, enc_len, x, \
                           ldx, out, dec_len, dec_mask, M, L, max_len, dur_cols);                                      \
    } while (0)
    if (D == 384) { if (split == 2) ISPK_LR(3, 2); else if (split) ISPK_LR(3, 1); else ISPK_LR(3, 0); }
    else { if (split == 2) ISPK_LR(2, 2); else if (split) ISPK_LR(2, 1); else ISPK_LR(2, 0); }
#undef ISPK_LR
    return ispk_launch_status();
}

extern "C" int32_t ispk_length_regulate_f32(const float* alignment, const float* dur_f32, const int64_t* dur_i64,
                                            const int64_t* enc_len, const float* x, int64_t ldx, float* out,
                                            int64_t* dec_len, uint8_t* dec_mask, int32_t B, int32_t M, int32_t L, int32_t D,
                                            int32_t max_len, int32_t dur_cols, ispk_stream_t stream) {
    return length_regulate_launch(alignment, dur_f32, dur_i64, enc_len, x, ldx, out, dec_len, dec_mask, B, M, L, D, max_len, dur_cols,
                                  stream, 0);
}

extern "C" int32_t ispk_length_regulate_split_bf16(const float* alignment, const float* dur_f32, const int64_t* dur_i64,
                                                   const int64_t* enc_len, const float* x, int64_t ldx, float* out,
                                                   int64_t* dec_len, uint8_t* dec_mask, int32_t B, int32_t M, int32_t L,
                                                   int32_t D, int32_t max_len, int32_t dur_cols, ispk_stream_t stream) {
    return length_regulate_launch(alignment, dur_f32, dur_i64, enc_len, x, ldx, out, dec_len, dec_mask, B, M, L, D, max_len, dur_cols,
                                  stream, 1);
}

extern "C" int32_t ispk_length_regulate_split_f16(const float* alignment, const float* dur_f32, const int64_t* dur_i64,
                                                  const int64_t* enc_len, const float* x, int64_t ldx, float* out,
                                                  int64_t* dec_len, uint8_t* dec_mask, int32_t B, int32_t M, int32_t L,
                                                  int32_t D, int32_t max_len, int32_t dur_cols, ispk_stream_t stream) {
    return length_regulate_launch(alignment, dur_f32, dur_i64, enc_len, x, ldx, out, dec_len, dec_mask, B, M, L, D, max_len, dur_cols,
                                  stream, 2);
}
