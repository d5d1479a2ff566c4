// Backward of the aligner front-end: the pieces that have no counterpart among the transformer kernels.
//   ispk_aligner_scores_bwd_f32    d (scaled q . k scores) from d attn_soft and d attn_logits (alignment.py:187-208)
//   ispk_masked_instnorm_bwd_f32   masked instance norm (modules/normalization.py:160-208) as ispk_masked_instnorm_f32 applies it
//   ispk_soft_average_bwd_f32      d attn_soft from the gradient of the soft-averaged pitch / energy (temporal_adaptor.py:446-449)
// The convolutions' own gradients reuse the GEMMs: dW = gemm_tn over the padded windows, dX = the forward's conv GEMM over
// the padded output gradient with flipped taps (train/aligner.py).  fp32, fixed summation orders.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ scores
// Forward (one row = one mel frame m of utterance b, j over the L key columns):
//   ls_j = log_softmax_j(scale q.k_j) over ALL L columns;  A_j = ls_j + log(prior_j + 1e-6)  (= attn_logits)
//   soft_j = [m < mel_len] softmax over the columns j < text_len of A_j  (0 elsewhere)
// Backward, with gs = d soft, gl = d logits:
//   dA_j = gl_j + soft_j (gs_j - sum_v soft_v gs_v);   d(q.k)_j = scale (dA_j - exp(ls_j) sum_j dA_j)
// exp(ls_j) = exp(A_j - log(prior_j + 1e-6)) with the analytic prior of alignment.py:18-37 recomputed.
// One wave per row; writes dS[b][m][j] (row stride ld_s) and its transpose dSt[b][j][m] (row stride ld_t); the padding
// columns of both (ld_s > L, ld_t > M) must arrive zeroed.
__global__ __launch_bounds__(256) void aligner_scores_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ soft,
                                                                 const float* __restrict__ d_soft, const float* __restrict__ d_logits,
                                                                 const int64_t* __restrict__ text_len,
                                                                 const int64_t* __restrict__ mel_len, float* __restrict__ dS,
                                                                 int64_t ld_s, float* __restrict__ dSt, int64_t ld_t, int B, int M,
                                                                 int L, float scale) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (int64_t)B * M) return;
    const int b = (int)(row / M), m = (int)(row - (int64_t)b * M);
    int tl = (int)text_len[b], ml = (int)mel_len[b];
    tl = tl < 1 ? 1 : (tl > L ? L : tl);
    ml = ml < 1 ? 1 : (ml > M ? M : ml);
    const float* A = logits + row * L;
    const float* S = soft + row * L;
    const float* gs = d_soft ? d_soft + row * L : nullptr;
    const float* gl = d_logits ? d_logits + row * L : nullptr;
    // the prior's row normaliser
    const float mq = (float)m / (float)ml;
    float psum = 0.f;
    if (m < ml)
        for (int j = lane; j < tl; j += 64) {
            const float g = (float)j / (float)tl - mq;
            psum += expf(-(g * g) / (2.0f * 0.1f * 0.1f));
        }
    for (int off = 32; off > 0; off >>= 1) psum += __shfl_xor(psum, off, 64);
    const float pinv = 1.0f / (psum + 1e-5f);
    // sum_v soft_v gs_v
    float sg = 0.f;
    if (gs)
        for (int j = lane; j < tl; j += 64) sg += S[j] * gs[j];
    for (int off = 32; off > 0; off >>= 1) sg += __shfl_xor(sg, off, 64);
    // dA and its row sum
    float tot = 0.f;
    for (int j = lane; j < L; j += 64) {
        float da = gl ? gl[j] : 0.f;
        if (gs && j < tl) da += S[j] * (gs[j] - sg);
        tot += da;
    }
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
    for (int j = lane; j < L; j += 64) {
        float da = gl ? gl[j] : 0.f;
        if (gs && j < tl) da += S[j] * (gs[j] - sg);
        float pr = 0.f;
        if (j < tl && m < ml) {
            const float g = (float)j / (float)tl - mq;
            pr = expf(-(g * g) / (2.0f * 0.1f * 0.1f)) * pinv;
            pr = pr < 1e-4f ? 0.f : pr;
        }
        const float p_all = expf(A[j] - logf(pr + 1e-6f));          // softmax over all L columns of the scaled scores
        const float v = scale * (da - p_all * tot);
        dS[((int64_t)b * M + m) * ld_s + j] = v;
        dSt[((int64_t)b * L + j) * ld_t + m] = v;
    }
}

// ------------------------------------------------------------------------------------------------ masked instance norm
// Forward (ispk_masked_instnorm_f32): per (utterance, channel) mean / biased variance over the n valid frames;
// out[t] = (y[t] - mean) rstd w + b for t < n, 0 for the frames past n (the next block's mask) - so only t < n matter:
//   g_t = d out[t] w;  d y[t] = rstd (g_t - mean_t(g) - yhat_t mean_t(g yhat)), t < n;  0 for t >= n (and the scratch rows)
//   d w += sum_t d out[t] yhat_t,  d b += sum_t d out[t]   (per utterance partials [B][2][C], summed over B by the caller's
//   second launch in utterance order).
// y, d_out, d_y: [B][T+4][C] with row t = frame t (the conv-output convention).  grid (ceil(C/64), B), 16 time lanes x 64 channels.
constexpr int kNbTL = 16;
__global__ __launch_bounds__(1024) void masked_instnorm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ d_out,
                                                                   const float* __restrict__ w, const int64_t* __restrict__ len,
                                                                   float* __restrict__ d_y, float* __restrict__ part, int T, int C,
                                                                   float eps) {
    __shared__ float red[kNbTL][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + cl;
    const bool cok = c < C;
    int n = (int)len[b];
    n = n < 1 ? 1 : (n > T ? T : n);
    const int64_t base = (int64_t)b * (T + 4) * C;
    auto total = [&](float v) {
        __syncthreads();
        red[tl][cl] = v;
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < kNbTL; ++i) s += red[i][cl];
        return s;
    };
    float s = 0.f;
    if (cok)
        for (int t = tl; t < n; t += kNbTL) s += y[base + (int64_t)t * C + c];
    const float mean = total(s) / (float)n;
    float q = 0.f;
    if (cok)
        for (int t = tl; t < n; t += kNbTL) {
            const float d = y[base + (int64_t)t * C + c] - mean;
            q += d * d;
        }
    const float rstd = 1.0f / sqrtf(total(q) / (float)n + eps);
    float sg = 0.f, sgx = 0.f;
    if (cok)
        for (int t = tl; t < n; t += kNbTL) {
            const float g = d_out[base + (int64_t)t * C + c];
            sg += g;
            sgx += g * (y[base + (int64_t)t * C + c] - mean) * rstd;
        }
    const float tg = total(sg), tgx = total(sgx);
    if (!cok) return;
    if (tl == 0) {
        part[((int64_t)b * 2 + 0) * C + c] = tgx;      // d w partial of this utterance
        part[((int64_t)b * 2 + 1) * C + c] = tg;       // d b partial
    }
    const float wv = w[c], c1 = tg * wv / (float)n, c2 = tgx * wv / (float)n;
    for (int t = tl; t < T + 4; t += kNbTL) {
        float v = 0.f;
        if (t < n) {
            const float yh = (y[base + (int64_t)t * C + c] - mean) * rstd;
            v = rstd * (d_out[base + (int64_t)t * C + c] * wv - c1 - yh * c2);
        }
        d_y[base + (int64_t)t * C + c] = v;
    }
}

__global__ __launch_bounds__(256) void instnorm_param_sum_kernel(const float* __restrict__ part, int B, int C, float* __restrict__ dw,
                                                                 float* __restrict__ db) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, bb = 0.f;
    for (int b = 0; b < B; ++b) {
        a += part[((int64_t)b * 2 + 0) * C + c];
        bb += part[((int64_t)b * 2 + 1) * C + c];
    }
    dw[c] = a;
    db[c] = bb;
}

// ------------------------------------------------------------------------------------------------ soft averages
// f_k[l] = mask_l N_k[l] / D[l],  N_k[l] = sum_m x_k[m] A[m][l],  D[l] = sum_m A[m][l] + 1e-5  (k = pitch, energy)
//   d A[m][l] (+)= sum_k g_k[l] mask_l (x_k[m] - N_k[l] / D[l]) / D[l]
// Pass 1 (one wave per (b, l) column): D and N_k / D.  Pass 2: element-wise over [B][M][L].
__global__ __launch_bounds__(256) void soft_average_cols_kernel(const float* __restrict__ A, const float* __restrict__ pitch,
                                                                const float* __restrict__ energy, float* __restrict__ cols, int B,
                                                                int M, int L) {
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (col >= (int64_t)B * L) return;
    const int b = (int)(col / L), l = (int)(col - (int64_t)b * L);
    float d = 0.f, np = 0.f, ne = 0.f;
    for (int m = lane; m < M; m += 64) {
        const float a = A[((int64_t)b * M + m) * L + l];
        d += a;
        np += a * pitch[(int64_t)b * M + m];
        ne += a * energy[(int64_t)b * M + m];
    }
    for (int off = 32; off > 0; off >>= 1) {
        d += __shfl_xor(d, off, 64);
        np += __shfl_xor(np, off, 64);
        ne += __shfl_xor(ne, off, 64);
    }
    if (lane == 0) {
        d += 1e-5f;
        cols[col * 3 + 0] = 1.0f / d;
        cols[col * 3 + 1] = np / d;
        cols[col * 3 + 2] = ne / d;
    }
}

__global__ __launch_bounds__(256) void soft_average_bwd_kernel(const float* __restrict__ pitch, const float* __restrict__ energy,
                                                               const float* __restrict__ d_feats, const int64_t* __restrict__ text_len,
                                                               const float* __restrict__ cols, float* __restrict__ dA, int accumulate,
                                                               int B, int M, int L) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * M * L) return;
    const int l = (int)(i % L);
    const int64_t bm = i / L;
    const int b = (int)(bm / M);
    const int64_t col = (int64_t)b * L + l;
    float v = 0.f;
    if (l < text_len[b]) {
        const float inv = cols[col * 3];
        const float gp = d_feats[col * 3 + 1], ge = d_feats[col * 3 + 2];   // feats [B][L][3]: (log1p duration, pitch, energy)
        v = inv * (gp * (pitch[bm] - cols[col * 3 + 1]) + ge * (energy[bm] - cols[col * 3 + 2]));
    }
    dA[i] = accumulate ? dA[i] + v : v;
}

}  // namespace

extern "C" int32_t ispk_aligner_scores_bwd_f32(const float* attn_logits, const float* attn_soft, const float* d_soft,
                                               const float* d_logits, const int64_t* text_len, const int64_t* mel_len, float* dS,
                                               int64_t ld_s, float* dSt, int64_t ld_t, int32_t B, int32_t M, int32_t L, float scale,
                                               ispk_stream_t stream) {
    ISPK_REQUIRE(attn_logits && attn_soft && text_len && mel_len && dS && dSt && (d_soft || d_logits), -1,
                 "ispk_aligner_scores_bwd_f32: null pointer");
    ISPK_REQUIRE(B >= 1 && M >= 1 && L >= 1 && ld_s >= L && ld_t >= M, -2, "ispk_aligner_scores_bwd_f32: bad shape B=%d M=%d L=%d", B,
                 M, L);
    const int64_t rows = (int64_t)B * M;
    hipLaunchKernelGGL(aligner_scores_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       attn_logits, attn_soft, d_soft, d_logits, text_len, mel_len, dS, ld_s, dSt, ld_t, B, M, L, scale);
    return ispk_launch_status();
}

extern "C" int32_t ispk_masked_instnorm_bwd_f32(const float* y, const float* d_out, const float* weight, const int64_t* lengths,
                                                float* d_y, float* d_weight, float* d_bias, float* workspace,
                                                int64_t workspace_floats, int32_t B, int32_t T, int32_t C, float eps,
                                                ispk_stream_t stream) {
    ISPK_REQUIRE(y && d_out && weight && lengths && d_y && d_weight && d_bias && workspace, -1,
                 "ispk_masked_instnorm_bwd_f32: null pointer");
    ISPK_REQUIRE(B >= 1 && T >= 1 && C >= 1 && B <= 65535 && workspace_floats >= (int64_t)B * 2 * C, -2,
                 "ispk_masked_instnorm_bwd_f32: bad shape B=%d T=%d C=%d (workspace: 2 B C floats)", B, T, C);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(masked_instnorm_bwd_kernel, dim3((C + 63) / 64, B), dim3(1024), 0, s, y, d_out, weight, lengths, d_y, workspace,
                       T, C, eps);
    hipLaunchKernelGGL(instnorm_param_sum_kernel, dim3((C + 255) / 256), dim3(256), 0, s, workspace, B, C, d_weight, d_bias);
    return ispk_launch_status();
}

extern "C" int32_t ispk_soft_average_bwd_f32(const float* attn_soft, const float* pitch, const float* energy, const float* d_feats,
                                             const int64_t* text_len, float* workspace, int64_t workspace_floats, float* d_attn,
                                             int32_t accumulate, int32_t B, int32_t M, int32_t L, ispk_stream_t stream) {
    ISPK_REQUIRE(attn_soft && pitch && energy && d_feats && text_len && workspace && d_attn, -1,
                 "ispk_soft_average_bwd_f32: null pointer");
    ISPK_REQUIRE(B >= 1 && M >= 1 && L >= 1 && workspace_floats >= (int64_t)B * L * 3, -2,
                 "ispk_soft_average_bwd_f32: bad shape B=%d M=%d L=%d (workspace: 3 B L floats)", B, M, L);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t cols = (int64_t)B * L, total = (int64_t)B * M * L;
    hipLaunchKernelGGL(soft_average_cols_kernel, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, s, attn_soft, pitch, energy, workspace,
                       B, M, L);
    hipLaunchKernelGGL(soft_average_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pitch, energy, d_feats,
                       text_len, workspace, d_attn, accumulate, B, M, L);
    return ispk_launch_status();
}
