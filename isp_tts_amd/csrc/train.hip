// The optimizer side of the training step, as gfx950 kernels over FLAT parameter / gradient / moment
// arenas: the squared gradient norm of the weight-decay group (what experiments/optimizers.py:236-237 clips), AdamW with
// the clip folded in (torch.optim.AdamW as built at optimizers.py:72-74 with the parameter grouping of :15-20, :34-40),
// and the mel loss with its gradient (models/acoustic/loss.py:22-35).
//
// All of it is HBM-bound streaming: AdamW reads p, g, m, v and writes p, m, v once - 28 B per parameter, 648 MB for the
// 23.1 M parameters of the recipe model, ~0.1 ms at 8 TB/s - where the reference's unfused torch.optim.AdamW makes ~12
// element-wise passes per tensor over ~290 tensors (~3,500 launches).
#include <math.h>

#include <string.h>

#include "common.h"

namespace {

constexpr int kNormBlocks = 1024;   // partial sums of stage 1 (fixed: the summation order never depends on the device)

// ------------------------------------------------------------------------------------------------ sum of squares
// Stage 1: block b sums elements b*256*4 + k*stride ... in a fixed order (per-thread running sum over its float4s, then a
// wave tree by DPP-free shuffles, then 4 wave partials in index order).  Stage 2: one block adds the 1024 partials in a
// fixed tree.  Deterministic run to run and independent of the CU count.
__global__ __launch_bounds__(256) void sqnorm_stage1_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ partial) {
    const int64_t n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    double acc = 0.0;      // fp64 running sums: the pass is HBM-bound, and the clip coefficient then carries no summation error
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)kNormBlocks * 256) {
        const f32x4 v = x4[i];
        acc += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {          // tail (n not a multiple of 4)
        const double t = x[(n4 << 2) + threadIdx.x];
        acc += t * t;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(1024) void sqnorm_stage2_kernel(const double* __restrict__ partial, float* __restrict__ out) {
    __shared__ double s[kNormBlocks];
    s[threadIdx.x] = partial[threadIdx.x];
    __syncthreads();
    for (int half = kNormBlocks / 2; half > 0; half >>= 1) {
        if ((int)threadIdx.x < half) s[threadIdx.x] += s[threadIdx.x + half];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)s[0];
}

// ------------------------------------------------------------------------------------------------ AdamW
// One element per lane-slot, float4 at a time.  The update is torch's _single_tensor_adamw in its order:
//   p *= 1 - lr*wd;  m += (g - m)*(1 - b1);  v = v*b2 + (1 - b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// Elements [0, n_decay) are the weight-decay group (tensors with >= 2 non-unit dimensions): they get the decay and the
// gradient-norm clip (clip_grad_norm_ on param_groups[0] only, optimizers.py:236-237); the rest get neither.
struct AdamArgs {
    float decay_mul;      // 1 - lr * weight_decay
    float one_m_b1, b2, one_m_b2;
    float step_size;      // lr / (1 - b1^t)
    float inv_bc2_sqrt;   // 1 / sqrt(1 - b2^t)   (torch divides; one rounding apart)
    float bc2_sqrt;
    float eps, max_norm, grad_scale;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float gmul, bool decay) {
#pragma clang fp contract(off)
    g *= gmul;
    if (decay) p *= a.decay_mul;
    m = m + a.one_m_b1 * (g - m);
    v = v * a.b2 + (a.one_m_b2 * g) * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, int64_t n_decay,
                                                    const float* __restrict__ sqnorm, AdamArgs a,
                                                    const AdamArgs* __restrict__ a_dev) {
    if (a_dev) a = *a_dev;         // ispk_adamw_f32_dev: the step's factors live in device memory (a captured step replays them)
    // clip coefficient of the decay group: clamp(max_norm / (norm + 1e-6), max = 1) on the SCALED gradients (torch
    // clip_grad_norm_).  torch.clamp PROPAGATES a NaN norm (every clipped gradient, and with it the group, turns NaN - the
    // reference's step() then reports grad_norm None, optimizers.py:238-239, but has already stepped); fminf(NaN, 1) would
    // return 1 and hide it, so the NaN is passed on explicitly.  An infinite norm gives 0, as in torch.
    float clip = 1.f;
    if (sqnorm) {
        const float norm = sqrtf(sqnorm[0]) * a.grad_scale;
        const float c = a.max_norm / (norm + 1e-6f);
        clip = c != c ? c : fminf(c, 1.f);
    }
    const float g_decay = a.grad_scale * clip, g_rest = a.grad_scale;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 p4 = reinterpret_cast<f32x4*>(p)[i], m4 = reinterpret_cast<f32x4*>(m)[i], v4 = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 g4 = reinterpret_cast<const f32x4*>(g)[i];
        float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
        const int64_t e = i << 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool d = e + k < n_decay;
            adam_one(pp[k], gg[k], mm[k], vv[k], a, d ? g_decay : g_rest, d);
        }
        reinterpret_cast<f32x4*>(p)[i] = f32x4{pp[0], pp[1], pp[2], pp[3]};
        reinterpret_cast<f32x4*>(m)[i] = f32x4{mm[0], mm[1], mm[2], mm[3]};
        reinterpret_cast<f32x4*>(v)[i] = f32x4{vv[0], vv[1], vv[2], vv[3]};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t e = (n4 << 2) + threadIdx.x;
        const bool d = e < n_decay;
        adam_one(p[e], g[e], m[e], v[e], a, d ? g_decay : g_rest, d);
    }
}

// ------------------------------------------------------------------------------------------------ mel loss
// loss.py:22-35 + utils/functions.py:44-58: MSE(mel_out, mel_target) per element, masked to t < mel_len[b], summed per
// utterance over (80, T), divided by max(80 * len_b, 1e-5), mean over the batch.  One workgroup per utterance computes the
// ratio (fixed-order reduction) and, when grad is asked for, writes d loss / d mel_out = 2 (out - tgt) / (den_b * B) * go
// (0 on padded frames); a second tiny launch averages the B ratios in index order.
__global__ __launch_bounds__(1024) void mel_loss_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                       const int64_t* __restrict__ mel_len, float* __restrict__ ratio,
                                                       float* __restrict__ grad, float grad_out, int B, int C, int T) {
    const int b = blockIdx.x;
    const int len = (int)min((int64_t)T, max((int64_t)0, mel_len[b]));
    const float den = fmaxf((float)((int64_t)C * len), 1e-5f);
    const float gmul = 2.f * grad_out / (den * (float)B);
    const float* o = out + (int64_t)b * C * T;
    const float* t = tgt + (int64_t)b * C * T;
    float* gr = grad ? grad + (int64_t)b * C * T : nullptr;
    float acc = 0.f;
    for (int i = threadIdx.x; i < C * T; i += 1024) {
        const int f = i % T;
        const float d = o[i] - t[i];
        const bool valid = f < len;
        acc += valid ? d * d : 0.f;
        if (gr) gr[i] = valid ? d * gmul : 0.f;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ float wsum[16];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < 16; ++w) tot += wsum[w];          // in wave order
        ratio[b] = tot / den;
    }
}

__global__ __launch_bounds__(64) void mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += x[i];
        out[0] = s / (float)n;
    }
}

// ------------------------------------------------------------------------------------------------ to_mel backward, step 1
// mel = mask * (dec W^T + b) stored [B][C][T] (model.py:167-168).  The GEMMs of the backward want frames as rows:
// g[(b, t)][c] = mask[b][t] * dmel[b][c][t].  32 x 32 tiles through LDS: reads run along t, writes along c.
__global__ __launch_bounds__(256) void mel_grad_rows_kernel(const float* __restrict__ dmel, const uint8_t* __restrict__ mask,
                                                            float* __restrict__ g, int C, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && t0 + tx < T) tile[k][tx] = dmel[((int64_t)b * C + c0 + k) * T + t0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int t = t0 + k, c = c0 + tx;
        if (t < T && c < C) {
            const float m = mask ? (mask[(int64_t)b * T + t] ? 1.f : 0.f) : 1.f;
            g[((int64_t)b * T + t) * C + c] = tile[tx][k] * m;
        }
    }
}

// column sums of a [rows][cols] matrix (bias gradients), two stages with a fixed order: block k adds rows k, k + P, ...
constexpr int kColParts = 256;
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int cols,
                                                            const uint8_t* __restrict__ mask, float* __restrict__ part) {
    for (int c = threadIdx.x; c < cols; c += 256) {
        float s = 0.f;
        for (int64_t r = blockIdx.x; r < rows; r += kColParts)
            if (!mask || mask[r]) s += x[r * ld + c];
        part[(int64_t)blockIdx.x * cols + c] = s;
    }
}

// Weight gradient of a Linear with a handful of input features (the adaptor's 2 -> 256 embedding projection):
// out[n][k] = sum_r g[r][n] x[r][k], k < K <= 8.  Same two stages as the column sums.
__global__ __launch_bounds__(256) void smallk_wgrad_stage1_kernel(const float* __restrict__ g, int64_t ldg,
                                                                  const float* __restrict__ x, int64_t ldx, int64_t rows, int N,
                                                                  int K, float* __restrict__ part) {
    for (int n = threadIdx.x; n < N; n += 256) {
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int64_t r = blockIdx.x; r < rows; r += kColParts) {
            const float gv = g[r * ldg + n];
            for (int k = 0; k < K; ++k) s[k] += gv * x[r * ldx + k];
        }
        for (int k = 0; k < K; ++k) part[((int64_t)blockIdx.x * N + n) * K + k] = s[k];
    }
}

// d table[v][:] = sum over the token positions r with ids[r] == v of d_emb[r][:], rows visited in index order (nn.Embedding's
// backward, model.py:131; the padding row gets no gradient).  One workgroup per vocabulary row, the token ids walked in
// segments of kEmbSeg: every thread tests its ids of the segment (independent loads), wave 0 compacts the hits into an
// ordered row list (ballot + prefix popcount), then thread c adds d_emb[row][c] over the list - the same order as a serial
// scan, without its 'rows' dependent loads per thread.
constexpr int kEmbSeg = 4096;
__global__ __launch_bounds__(128) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ d_emb,
                                                            int64_t rows, int D, int padding_idx, float* __restrict__ d_table,
                                                            int64_t ld_table) {
    __shared__ uint8_t hit[kEmbSeg];
    __shared__ int list[kEmbSeg];
    __shared__ int count_s;
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    float s[4] = {0.f, 0.f, 0.f, 0.f};                      // columns tid, tid + 128, ... (D <= 512)
    if (v != padding_idx) {
        for (int64_t seg = 0; seg < rows; seg += kEmbSeg) {
            const int n = (int)(rows - seg < kEmbSeg ? rows - seg : kEmbSeg);
            for (int i = tid; i < kEmbSeg; i += 128) hit[i] = (i < n && ids[seg + i] == v) ? 1 : 0;
            __syncthreads();
            if (tid < 64) {
                int count = 0;
                for (int base = 0; base < n; base += 64) {
                    const bool mine = hit[base + lane] != 0;
                    const unsigned long long m = __ballot(mine);
                    if (mine) list[count + __popcll(m & ((1ull << lane) - 1ull))] = base + lane;
                    count += __popcll(m);
                }
                if (lane == 0) count_s = count;
            }
            __syncthreads();
            const int count = count_s;
#pragma unroll 4
            for (int k = 0; k < count; ++k) {
                const float* g = d_emb + (seg + list[k]) * D;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (tid + 128 * j < D) s[j] += g[tid + 128 * j];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (tid + 128 * j < D) d_table[(int64_t)v * ld_table + tid + 128 * j] = s[j];
}

__global__ __launch_bounds__(256) void colsum_stage2_kernel(const float* __restrict__ part, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k = 0; k < kColParts; ++k) s += part[(int64_t)k * cols + c];
    out[c] = s;
}

// ------------------------------------------------------------------------------------------------ binarisation loss
// loss.py:90-107 (AttentionBinarizationLoss): -sum over the hard path of log(clamp(soft, eps)) / (number of path cells).
// One wave per (utterance, frame) row finds the row's hot column in the int16 one-hot MAS output (rows past mel_len are all
// zero and contribute nothing), stage 1 leaves one (sum of logs, count) pair per workgroup, stage 2 adds them in order.
// The gradient wrt soft is -go / (count * soft) at the hot cell where soft > eps (clamp: zero below), written by a second
// pass once the count is known (grad must be zero-initialised by the caller, or NULL).
constexpr int kBinParts = 1024;
__global__ __launch_bounds__(256) void bin_loss_stage1_kernel(const float* __restrict__ soft, const int16_t* __restrict__ hard,
                                                              int64_t rows, int L, float eps, float* __restrict__ part) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float lsum = 0.f, cnt = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)kBinParts * 4) {
        for (int j = lane; j < L; j += 64)
            if (hard[r * L + j] != 0) {
                lsum += logf(fmaxf(soft[r * L + j], eps));
                cnt += 1.f;
            }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lsum += __shfl_down(lsum, off, 64);
        cnt += __shfl_down(cnt, off, 64);
    }
    __shared__ float w[4][2];
    if (lane == 0) {
        w[wave][0] = lsum;
        w[wave][1] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (w[0][0] + w[1][0]) + (w[2][0] + w[3][0]);
        part[2 * blockIdx.x + 1] = (w[0][1] + w[1][1]) + (w[2][1] + w[3][1]);
    }
}

__global__ __launch_bounds__(1024) void bin_loss_stage2_kernel(const float* __restrict__ part, float* __restrict__ out) {
    __shared__ float s[kBinParts], c[kBinParts];
    s[threadIdx.x] = part[2 * threadIdx.x];
    c[threadIdx.x] = part[2 * threadIdx.x + 1];
    __syncthreads();
    for (int half = kBinParts / 2; half > 0; half >>= 1) {
        if ((int)threadIdx.x < half) {
            s[threadIdx.x] += s[threadIdx.x + half];
            c[threadIdx.x] += c[threadIdx.x + half];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = -s[0] / c[0];      // (no path cell at all: 0 / 0 = NaN, as the reference)
        out[1] = c[0];
    }
}

__global__ __launch_bounds__(256) void bin_loss_grad_kernel(const float* __restrict__ soft, const int16_t* __restrict__ hard,
                                                            int64_t rows, int L, float eps, const float* __restrict__ loss_cnt,
                                                            float grad_out, float* __restrict__ grad) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float scale = -grad_out / loss_cnt[1];
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4)
        for (int j = lane; j < L; j += 64)
            if (hard[r * L + j] != 0) {
                const float v = soft[r * L + j];
                grad[r * L + j] = v > eps ? scale / v : 0.f;
            }
}

// ------------------------------------------------------------------------------------------------ attention CTC loss
// loss.py:39-77 (AttentionCTCLoss): the aligner's logits [B][M][L] get a blank class in front (logit `blank`, -1 in the
// recipes), log-softmax over the L + 1 classes, then nn.CTCLoss(blank = 0, zero_infinity = True, reduction "mean") with
// the targets 1 .. text_len[b] (every text token once, in order) and mel_len[b] input frames.
//   extended targets l'_s, s = 0 .. 2U: blank for even s, token (s + 1) / 2 for odd s (all tokens distinct, so the skip
//   s - 2 -> s is always allowed for odd s >= 3);
//   alpha_t(s) = lp_t(l'_s) + lse(alpha_{t-1}(s), alpha_{t-1}(s-1), [odd s >= 3] alpha_{t-1}(s-2)); beta mirrored;
//   nll = -lse(alpha_{T-1}(2U), alpha_{T-1}(2U-1));  loss = mean_b nll_b / max(U_b, 1)  (inf -> 0: zero_infinity);
//   d loss / d logit_t(c) = w_b (softmax_t(c) - exp(lse_{s: l'_s = c}(alpha_t(s) + beta_t(s)) + nll - lp_t(c))),  t < T,
//   with w_b = go / (B max(U_b, 1)) - the gradient torch's ctc_loss_backward returns for log-softmax'ed inputs.
// Three launches: row normalisers (one wave per frame), the two recursions (one 512-thread workgroup per utterance: waves
// 0-3 walk alpha forwards while waves 4-7 walk beta backwards, one barrier per frame; tables in a workspace), then the
// gradient with one wave per frame.  fp32 log domain; a first correct cut, not tuned.
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    return m == -INFINITY ? -INFINITY : m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    return m == -INFINITY ? -INFINITY : m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

__global__ __launch_bounds__(256) void ctc_row_lse_kernel(const float* __restrict__ logits, float blank, int64_t rows, int L,
                                                          float* __restrict__ row_lse) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float m = blank;
    for (int j = lane; j < L; j += 64) m = fmaxf(m, logits[r * L + j]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float sum = lane == 0 ? expf(blank - m) : 0.f;
    for (int j = lane; j < L; j += 64) sum += expf(logits[r * L + j] - m);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) row_lse[r] = m + logf(sum);
}

__global__ __launch_bounds__(512) void ctc_alpha_beta_kernel(const float* __restrict__ logits, const float* __restrict__ row_lse,
                                                             float blank, const int64_t* __restrict__ text_len,
                                                             const int64_t* __restrict__ mel_len, float* __restrict__ alpha,
                                                             float* __restrict__ beta, float* __restrict__ nll, int M, int L,
                                                             int S_pad) {
    extern __shared__ float sm[];                 // [2 tables][2 buffers][S_pad + 2]: two -inf guard cells on either side
    const int b = blockIdx.x, tid = threadIdx.x, back = tid >> 8, t8 = tid & 255;
    const int U = (int)min((int64_t)L, max((int64_t)0, text_len[b])), T = (int)min((int64_t)M, max((int64_t)0, mel_len[b]));
    const int S = 2 * U + 1;
    float* tab = sm + back * 2 * (S_pad + 4);
    const float* lg = logits + (int64_t)b * M * L;
    const float* rl = row_lse + (int64_t)b * M;
    float* out = (back ? beta : alpha) + (int64_t)b * M * S_pad;
    for (int i = t8; i < 2 * (S_pad + 4); i += 256) tab[i] = -INFINITY;
    __syncthreads();
    if (T == 0) {
        if (tid == 0) nll[b] = INFINITY;
        return;
    }
    auto lp = [&](int t, int s) {                 // log-probability of the extended target's class at position s, frame t
        const float v = (s & 1) ? lg[(int64_t)t * L + (s >> 1)] : blank;
        return v - rl[t];
    };
    if (S <= 256) {
        // One state per thread.  A step's only global reads - this state's class log-probability and the frame's normaliser -
        // do not depend on the recursion, so they are fetched FOUR steps ahead into a register ring: the step itself is then
        // LDS reads, one lse3 and the barrier (the loads' latency used to sit inside every one of the T steps).
        const int sx = t8;
        const bool act = sx < S;
        auto fetch = [&](int step) -> float {
            if (step >= T || !act) return 0.f;
            const int t = back ? T - 1 - step : step;
            return ((sx & 1) ? lg[(int64_t)t * L + (sx >> 1)] : blank) - rl[t];
        };
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = fetch(k);
        for (int step0 = 0; step0 < T; step0 += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int step = step0 + k;
                if (step < T) {                                   // (workgroup-uniform)
                    const int t = back ? T - 1 - step : step;
                    float* cur = tab + (step & 1) * (S_pad + 4) + 2;
                    const float* prev = tab + ((step & 1) ^ 1) * (S_pad + 4) + 2;
                    if (act) {
                        float v;
                        if (step == 0) {
                            const bool start = back ? (sx >= S - 2) : (sx <= 1);
                            v = start ? q[k] : -INFINITY;
                        } else if (!back) {
                            const float skip = ((sx & 1) && sx >= 3) ? prev[sx - 2] : -INFINITY;
                            v = lse3(prev[sx], prev[sx - 1], skip);
                            v = v == -INFINITY ? v : v + q[k];
                        } else {
                            const float nxt = sx + 1 < S ? prev[sx + 1] : -INFINITY;
                            const float skip = ((sx & 1) && sx + 2 < S) ? prev[sx + 2] : -INFINITY;
                            v = lse3(prev[sx], nxt, skip);
                            v = v == -INFINITY ? v : v + q[k];
                        }
                        cur[sx] = v;
                        out[(int64_t)t * S_pad + sx] = v;
                    }
                    q[k] = fetch(step + 4);
                    __syncthreads();
                }
            }
        }
    } else {
    for (int step = 0; step < T; ++step) {
        const int t = back ? T - 1 - step : step;
        float* cur = tab + (step & 1) * (S_pad + 4) + 2;
        const float* prev = tab + ((step & 1) ^ 1) * (S_pad + 4) + 2;
        for (int s = t8; s < S; s += 256) {
            float v;
            if (step == 0) {
                const bool start = back ? (s >= S - 2) : (s <= 1);
                v = start ? lp(t, s) : -INFINITY;
            } else if (!back) {
                const float skip = ((s & 1) && s >= 3) ? prev[s - 2] : -INFINITY;
                v = lse3(prev[s], prev[s - 1], skip);
                v = v == -INFINITY ? v : v + lp(t, s);
            } else {
                const float nxt = s + 1 < S ? prev[s + 1] : -INFINITY;
                const float skip = ((s & 1) && s + 2 < S) ? prev[s + 2] : -INFINITY;
                v = lse3(prev[s], nxt, skip);
                v = v == -INFINITY ? v : v + lp(t, s);
            }
            cur[s] = v;
            out[(int64_t)t * S_pad + s] = v;
        }
        __syncthreads();
    }
    }
    if (tid == 0) {
        const float* last = sm + ((T - 1) & 1) * (S_pad + 4) + 2;     // the alpha table's final buffer
        nll[b] = -lse2(last[S - 1], S >= 2 ? last[S - 2] : -INFINITY);
    }
}

// The two recursions for S = 2 U + 1 <= 256 (every recipe: U <= 127 phonemes ... the launcher falls back above that): ONE WAVE
// per (utterance, direction), four consecutive states per lane in registers, the neighbours' edge states by DPP wave shifts -
// no LDS, no barrier (the workgroup version above spends a barrier and an LDS round trip per frame: 0.65 us x 512 frames).
// exp / log on v_exp_f32 / v_log_f32: the log-sum-exp of a step is m + log(sum) with sum in [1, 3], so a step's error is an
// absolute ~1e-7 however large the accumulated log-probability is.  The frame's three inputs (two class logits of the lane,
// the row normaliser) are fetched eight frames ahead.
__device__ __forceinline__ float ctc_shr1(float src) {      // lane l <- lane l - 1; lane 0 <- -inf
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -INFINITY), __builtin_bit_cast(int, src), 0x138,
                                                                 0xf, 0xf, false));
}
__device__ __forceinline__ float ctc_shl1(float src) {      // lane l <- lane l + 1; lane 63 <- -inf
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -INFINITY), __builtin_bit_cast(int, src), 0x130,
                                                                 0xf, 0xf, false));
}
__device__ __forceinline__ float ctc_lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    const float k = 1.4426950408889634f;
    const float sum = __builtin_amdgcn_exp2f((a - m) * k) + __builtin_amdgcn_exp2f((b - m) * k) + __builtin_amdgcn_exp2f((c - m) * k);
    return m == -INFINITY ? -INFINITY : fmaf(__builtin_amdgcn_logf(sum), 0.6931471805599453f, m);
}
constexpr int kCtcAhead = 8;
__global__ __launch_bounds__(64) void ctc_wave_kernel(const float* __restrict__ logits, const float* __restrict__ row_lse, float blank,
                                                      const int64_t* __restrict__ text_len, const int64_t* __restrict__ mel_len,
                                                      float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ nll,
                                                      int M, int L, int S_pad) {
    __shared__ float fin[256];
    const int b = blockIdx.x, back = blockIdx.y, lane = threadIdx.x, s0 = 4 * lane;
    const int U = (int)min((int64_t)L, max((int64_t)0, text_len[b])), T = (int)min((int64_t)M, max((int64_t)0, mel_len[b]));
    const int S = 2 * U + 1;
    if (T == 0) {
        if (!back && lane == 0) nll[b] = INFINITY;
        return;
    }
    const float* lg = logits + (int64_t)b * M * L;
    const float* rl = row_lse + (int64_t)b * M;
    float* out = (back ? beta : alpha) + (int64_t)b * M * S_pad;
    const bool c0ok = 2 * lane < L, c1ok = 2 * lane + 1 < L, st_ok = s0 < S_pad;
    bool act[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) act[k] = s0 + k < S;
    float x0q[kCtcAhead], x1q[kCtcAhead], rq[kCtcAhead];
    auto fetch = [&](int step, int slot) {
        x0q[slot] = x1q[slot] = rq[slot] = 0.f;
        if (step < T) {
            const int t = back ? T - 1 - step : step;
            if (c0ok) x0q[slot] = lg[(int64_t)t * L + 2 * lane];
            if (c1ok) x1q[slot] = lg[(int64_t)t * L + 2 * lane + 1];
            rq[slot] = rl[t];
        }
    };
#pragma unroll
    for (int k = 0; k < kCtcAhead; ++k) fetch(k, k);
    float a[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int step0 = 0; step0 < T; step0 += kCtcAhead) {
#pragma unroll
        for (int k = 0; k < kCtcAhead; ++k) {
            const int step = step0 + k;
            if (step < T) {                                   // (wave-uniform)
                const int t = back ? T - 1 - step : step;
                const float lpe = blank - rq[k], lp[4] = {lpe, x0q[k] - rq[k], lpe, x1q[k] - rq[k]};
                float nw[4];
                if (step == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) nw[q] = (back ? (s0 + q >= S - 2) : (s0 + q <= 1)) ? 0.f : -INFINITY;
                } else if (!back) {
                    const float u3 = ctc_shr1(a[3]);          // state s0 - 1 (its skip target s0 + 1 is odd)
                    nw[0] = ctc_lse3(a[0], u3, -INFINITY);
                    nw[1] = ctc_lse3(a[1], a[0], u3);         // odd state: skip from s - 2 (lane 0: -inf, as s = 1 < 3 asks)
                    nw[2] = ctc_lse3(a[2], a[1], -INFINITY);
                    nw[3] = ctc_lse3(a[3], a[2], a[1]);
                } else {
                    const float d0 = ctc_shl1(a[0]), d1 = ctc_shl1(a[1]);     // states s0 + 4, s0 + 5
                    nw[0] = ctc_lse3(a[0], a[1], -INFINITY);
                    nw[1] = ctc_lse3(a[1], a[2], a[3]);       // odd state: skip to s + 2
                    nw[2] = ctc_lse3(a[2], a[3], -INFINITY);
                    nw[3] = ctc_lse3(a[3], d0, d1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = (nw[q] == -INFINITY || !act[q]) ? -INFINITY : nw[q] + lp[q];
                if (st_ok) *reinterpret_cast<f32x4*>(out + (int64_t)t * S_pad + s0) = f32x4{a[0], a[1], a[2], a[3]};
                fetch(step + kCtcAhead, k);
            }
        }
    }
    if (!back) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fin[s0 + q] = a[q];
        __syncthreads();
        if (lane == 0) nll[b] = -lse2(fin[S - 1], S >= 2 ? fin[S - 2] : -INFINITY);
    }
}

__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logits, const float* __restrict__ row_lse,
                                                       float blank, const int64_t* __restrict__ text_len,
                                                       const int64_t* __restrict__ mel_len, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, const float* __restrict__ nll,
                                                       float grad_out, float* __restrict__ grad, int B, int M, int L, int S_pad) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= (int64_t)B * M) return;
    const int b = (int)(r / M), t = (int)(r - (int64_t)b * M);
    const int U = (int)min((int64_t)L, max((int64_t)0, text_len[b])), T = (int)min((int64_t)M, max((int64_t)0, mel_len[b]));
    const float n = nll[b];
    float* g = grad + r * L;
    if (t >= T || !(n < INFINITY)) {              // frames past the input length, or an impossible alignment (zero_infinity)
        for (int j = lane; j < L; j += 64) g[j] = 0.f;
        return;
    }
    const float w = grad_out / ((float)B * (float)max(U, 1));
    const float* a = alpha + r * S_pad;
    const float* be = beta + r * S_pad;
    const float lse_t = row_lse[r];
    for (int j = lane; j < L; j += 64) {          // class j + 1: one extended position (2 j + 1) when j < U, none otherwise
        const float lpv = logits[r * L + j] - lse_t;
        float v = expf(lpv);
        if (j < U) v -= expf(a[2 * j + 1] + be[2 * j + 1] + n - lpv);
        g[j] = w * v;
    }
}

__global__ __launch_bounds__(64) void ctc_mean_kernel(const float* __restrict__ nll, const int64_t* __restrict__ text_len, int B,
                                                      int L, float* __restrict__ loss) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            const int U = (int)min((int64_t)L, max((int64_t)0, text_len[b]));
            const float v = nll[b];
            s += (v < INFINITY) ? v / (float)max(U, 1) : 0.f;
        }
        loss[0] = s / (float)B;
    }
}

// ------------------------------------------------------------------------------------------------ flow predictor, backward pieces
// Flow loss (temporal_adaptor.py:145-146, utils/functions.py:44-58): loss = mean_b sum_{valid l, c} (raw m - flow)^2 / max(C n_b, 1e-5);
// d loss / d raw[b][l][c] = go 2 m (raw m - flow) / (max(C n_b, 1e-5) B).  One workgroup per utterance (it counts n_b first).
__global__ __launch_bounds__(256) void flow_loss_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ flow,
                                                            const uint8_t* __restrict__ mask, float grad_out, float* __restrict__ d_raw,
                                                            int B, int L, int C) {
    const int b = blockIdx.x;
    __shared__ int cnt[4];
    int n = 0;
    for (int l = threadIdx.x; l < L; l += 256) n += mask[(int64_t)b * L + l] ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = n;
    __syncthreads();
    const int valid = (cnt[0] + cnt[1]) + (cnt[2] + cnt[3]);
    const float w = 2.f * grad_out / (fmaxf((float)(C * valid), 1e-5f) * (float)B);
    for (int e = threadIdx.x; e < L * C; e += 256) {
        const int64_t i = (int64_t)b * L * C + e;
        const bool m = mask[(int64_t)b * L + e / C] != 0;
        d_raw[i] = m ? w * (raw[i] - flow[i]) : 0.f;
    }
}

// AdaptiveLayerNorm backward (normalization.py:37-61): y = xhat * scale_b + shift_b [* mask], xhat = (x - mean) rstd without
// affine; per utterance b (rows_per_batch rows): dx = rstd (g - mean(g) - xhat mean(g xhat)) with g = dy mask scale_b,
// d scale_b = sum_rows dy mask xhat, d shift_b = sum_rows dy mask.  One workgroup per utterance, rows over its 4 waves, a
// fixed-order cross-wave sum: [B][D] outputs with no second stage.  D = 64 * NPL.  NW = 16 waves: a wave walks its rows one
// after the other through three shuffle reductions each, so the 100 rows of an utterance want many waves (4 waves: 64 us).
template <int NPL, int NW = 16>
__global__ __launch_bounds__(NW * 64) void adaln_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                        int64_t lddy, const float* __restrict__ scale, int64_t ld_scale,
                                                        const uint8_t* __restrict__ mask, float* __restrict__ dx, int64_t lddx,
                                                        int add_to_dx, float* __restrict__ dscale, float* __restrict__ dshift,
                                                        int64_t ld_out, int rows_per_batch, float eps) {
    constexpr int D = NPL * 64;
    const int b = blockIdx.x, wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    float sc[NPL], ds[NPL], dt[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        sc[k] = scale[(int64_t)b * ld_scale + l + 64 * k];
        ds[k] = 0.f;
        dt[k] = 0.f;
    }
    for (int rr = wave; rr < rows_per_batch; rr += NW) {
        const int64_t row = (int64_t)b * rows_per_batch + rr;
        const float mk = mask ? (mask[row] ? 1.f : 0.f) : 1.f;
        float xv[NPL], gv[NPL];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            xv[k] = x[row * ldx + l + 64 * k];
            gv[k] = dy[row * lddy + l + 64 * k] * mk;
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
            xv[k] *= rstd;
            ds[k] += gv[k] * xv[k];
            dt[k] += gv[k];
            gv[k] *= sc[k];
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
            float* d = dx + row * lddx + l + 64 * k;
            *d = add_to_dx ? *d + v : v;
        }
    }
    __shared__ float red[NW][2][D];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        red[wave][0][l + 64 * k] = ds[k];
        red[wave][1][l + 64 * k] = dt[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += NW * 64) {
        const int w = i / D, col = i - w * D;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) v += red[k][w][col];        // in wave order
        (w ? dshift : dscale)[(int64_t)b * ld_out + col] = v;
    }
}

// Time embedding backward (embeddings.py:131-157 as ispk_time_embedding_f32 computes it: f = [t, sin, cos], h = silu(W0 f + b0),
// out = W1 h + b1; t itself gets no gradient).  One workgroup of 1024 threads; the n time values go through LDS in chunks of 64:
// per chunk the features, pre-activations and their gradients of all its values are computed in parallel, then every thread
// adds the chunk's terms of the outputs it owns (rows of dW0 / dW1, the bias sums) walking the time values IN ORDER - every
// sum has a fixed order.  E <= 64, 1 + 2H <= 160.
constexpr int kTeChunk = 64, kTeOwn = 16;      // outputs per thread: (64 * 160 + 64 * 64 + 128) / 1024 < 16
__global__ __launch_bounds__(1024) void time_embedding_bwd_kernel(const float* __restrict__ t, int n, const float* __restrict__ inv_freq,
                                                                  const float* __restrict__ freq_scale, int H,
                                                                  const float* __restrict__ w0, const float* __restrict__ b0,
                                                                  const float* __restrict__ w1, int E, const float* __restrict__ d_out,
                                                                  float* __restrict__ dw0, float* __restrict__ db0,
                                                                  float* __restrict__ dw1, float* __restrict__ db1) {
#pragma clang fp contract(off)
    extern __shared__ float te_lds[];
    const int K0 = 1 + 2 * H, tid = threadIdx.x;
    float* f = te_lds;                       // [chunk][K0]
    float* hh = f + kTeChunk * K0;           // [chunk][E]  silu(pre)
    float* go = hh + kTeChunk * E;           // [chunk][E]  d out
    float* dp = go + kTeChunk * E;           // [chunk][E]  d pre
    float* pre = dp + kTeChunk * E;          // [chunk][E]
    const float fs = freq_scale[0];
    const int n_w0 = E * K0, n_w1 = E * E, n_out = n_w0 + n_w1 + 2 * E;      // [dW0 | dW1 | db0 | db1]
    float acc[kTeOwn];
#pragma unroll
    for (int q = 0; q < kTeOwn; ++q) acc[q] = 0.f;
    for (int i0 = 0; i0 < n; i0 += kTeChunk) {
        const int cn = n - i0 < kTeChunk ? n - i0 : kTeChunk;
        __syncthreads();
        for (int e = tid; e < cn * K0; e += 1024) {
            const int i = e / K0, k = e - i * K0;
            const float pos = t[i0 + i];
            float v = pos;
            if (k >= 1) {
                const int kk = k - 1 < H ? k - 1 : k - 1 - H;
                const float a = pos * fs * inv_freq[kk];
                v = k - 1 < H ? sinf(a) : cosf(a);
            }
            f[e] = v;
        }
        for (int e = tid; e < cn * E; e += 1024) go[e] = d_out[(int64_t)(i0 + e / E) * E + e % E];
        __syncthreads();
        for (int e = tid; e < cn * E; e += 1024) {
            const int i = e / E, j = e - i * E;
            float pr = b0[j];
            for (int k = 0; k < K0; ++k) pr = fmaf(f[i * K0 + k], w0[(int64_t)j * K0 + k], pr);
            pre[e] = pr;
            hh[e] = pr / (1.0f + expf(-pr));
        }
        __syncthreads();
        for (int e = tid; e < cn * E; e += 1024) {
            const int i = e / E, j = e - i * E;
            float dh = 0.f;                                   // d loss / d h_j = sum_k go[k] W1[k][j]
            for (int k = 0; k < E; ++k) dh = fmaf(go[i * E + k], w1[(int64_t)k * E + j], dh);
            const float pr = pre[e], sg = 1.0f / (1.0f + expf(-pr));
            dp[e] = dh * (sg * (1.0f + pr * (1.0f - sg)));      // silu'(pre)
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kTeOwn; ++q) {
            const int o = tid + q * 1024;
            if (o >= n_out) break;
            float a = acc[q];
            if (o < n_w0) {                      // dW0[j][k] += d pre[i][j] f[i][k]
                const int j = o / K0, k = o - j * K0;
                for (int i = 0; i < cn; ++i) a += dp[i * E + j] * f[i * K0 + k];
            } else if (o < n_w0 + n_w1) {        // dW1[j][k] += d out[i][j] h[i][k]
                const int j = (o - n_w0) / E, k = (o - n_w0) - j * E;
                for (int i = 0; i < cn; ++i) a += go[i * E + j] * hh[i * E + k];
            } else {
                const int j = (o - n_w0 - n_w1) % E;
                const float* src = o < n_w0 + n_w1 + E ? dp : go;
                for (int i = 0; i < cn; ++i) a += src[i * E + j];
            }
            acc[q] = a;
        }
    }
#pragma unroll
    for (int q = 0; q < kTeOwn; ++q) {
        const int o = tid + q * 1024;
        if (o >= n_out) break;
        if (o < n_w0) dw0[o] = acc[q];
        else if (o < n_w0 + n_w1) dw1[o - n_w0] = acc[q];
        else if (o < n_w0 + n_w1 + E) db0[o - n_w0 - n_w1] = acc[q];
        else db1[o - n_w0 - n_w1 - E] = acc[q];
    }
}

}  // namespace

extern "C" int32_t ispk_flow_loss_bwd_f32(const float* pred_raw, const float* flow, const uint8_t* mask, float grad_out, float* d_raw,
                                          int32_t B, int32_t L, int32_t C, ispk_stream_t stream) {
    ISPK_REQUIRE(pred_raw && flow && mask && d_raw, -1, "ispk_flow_loss_bwd_f32: null pointer");
    ISPK_REQUIRE(B >= 1 && L >= 1 && C >= 1 && B <= 65535, -2, "ispk_flow_loss_bwd_f32: bad shape B=%d L=%d C=%d", B, L, C);
    hipLaunchKernelGGL(flow_loss_bwd_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pred_raw, flow, mask,
                       grad_out, d_raw, B, L, C);
    return ispk_launch_status();
}

extern "C" int32_t ispk_adaln_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* scale, int64_t ld_scale,
                                      const uint8_t* row_mask, float* dx, int64_t lddx, int32_t add_to_dx, float* dscale,
                                      float* dshift, int64_t ld_out, int32_t B, int32_t rows_per_batch, int32_t dim, float eps,
                                      ispk_stream_t stream) {
    ISPK_REQUIRE(x && dy && scale && dx && dscale && dshift, -1, "ispk_adaln_bwd_f32: null pointer");
    ISPK_REQUIRE(B >= 1 && rows_per_batch >= 1 && (dim == 256 || dim == 384) && ldx >= dim && lddy >= dim && lddx >= dim &&
                     ld_scale >= dim && ld_out >= dim && B <= 65535, -2,
                 "ispk_adaln_bwd_f32: bad shape B=%d rows_per_batch=%d dim=%d (dim 256 or 384)", B, rows_per_batch, dim);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dim == 256)
        hipLaunchKernelGGL((adaln_bwd_kernel<4, 16>), dim3(B), dim3(1024), 0, s, x, ldx, dy, lddy, scale, ld_scale, row_mask, dx, lddx,
                           add_to_dx, dscale, dshift, ld_out, rows_per_batch, eps);
    else
        hipLaunchKernelGGL((adaln_bwd_kernel<6, 16>), dim3(B), dim3(1024), 0, s, x, ldx, dy, lddy, scale, ld_scale, row_mask, dx, lddx,
                           add_to_dx, dscale, dshift, ld_out, rows_per_batch, eps);
    return ispk_launch_status();
}

extern "C" int32_t ispk_time_embedding_bwd_f32(const float* t, int32_t n, const float* inv_freq, const float* freq_scale,
                                               int32_t half_dim, const float* w0, const float* b0, const float* w1, int32_t emb_dim,
                                               const float* d_out, float* dw0, float* db0, float* dw1, float* db1,
                                               ispk_stream_t stream) {
    ISPK_REQUIRE(t && inv_freq && freq_scale && w0 && b0 && w1 && d_out && dw0 && db0 && dw1 && db1, -1,
                 "ispk_time_embedding_bwd_f32: null pointer");
    ISPK_REQUIRE(n >= 1 && half_dim >= 1 && 1 + 2 * half_dim <= 160 && emb_dim >= 1 && emb_dim <= 64, -2,
                 "ispk_time_embedding_bwd_f32: bad shape n=%d half_dim=%d emb_dim=%d", n, half_dim, emb_dim);
    const size_t lds = (size_t)kTeChunk * ((1 + 2 * half_dim) + 4 * emb_dim) * sizeof(float);        // <= 106 KB
    ISPK_RESERVE_LDS(time_embedding_bwd_kernel, lds, "ispk_time_embedding_bwd_f32");
    hipLaunchKernelGGL(time_embedding_bwd_kernel, dim3(1), dim3(1024), lds, reinterpret_cast<hipStream_t>(stream), t, n, inv_freq,
                       freq_scale, half_dim, w0, b0, w1, emb_dim, d_out, dw0, db0, dw1, db1);
    return ispk_launch_status();
}

extern "C" int32_t ispk_attn_ctc_loss_f32(const float* attn_logits, const int64_t* text_len, const int64_t* mel_len,
                                          float blank_logprob, float* workspace, int64_t workspace_floats, float* loss,
                                          float* grad, float grad_out, int32_t B, int32_t M, int32_t L, ispk_stream_t stream) {
    ISPK_REQUIRE(attn_logits && text_len && mel_len && workspace && loss, -1, "ispk_attn_ctc_loss_f32: null pointer");
    ISPK_REQUIRE(B >= 1 && M >= 1 && L >= 1 && L <= 2048, -2, "ispk_attn_ctc_loss_f32: bad shape B=%d M=%d L=%d", B, M, L);
    const int S_pad = (2 * L + 1 + 63) / 64 * 64;
    const int64_t rows = (int64_t)B * M, need = rows + B + 2 * rows * S_pad;
    ISPK_REQUIRE(workspace_floats >= need, -3, "ispk_attn_ctc_loss_f32: workspace needs %lld floats", (long long)need);
    float *row_lse = workspace, *nll = workspace + rows, *alpha = nll + B, *beta = alpha + rows * S_pad;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ctc_row_lse_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, attn_logits, blank_logprob, rows, L,
                       row_lse);
    if (2 * L + 1 <= 256) {        // (S <= 2 L + 1: every state of the extended target in one wave's registers)
        hipLaunchKernelGGL(ctc_wave_kernel, dim3(B, 2), dim3(64), 0, s, attn_logits, row_lse, blank_logprob, text_len, mel_len, alpha,
                           beta, nll, M, L, S_pad);
    } else {
        const size_t lds = (size_t)4 * (S_pad + 4) * sizeof(float);
        ISPK_RESERVE_LDS(ctc_alpha_beta_kernel, lds, "ispk_attn_ctc_loss_f32");
        hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(B), dim3(512), lds, s, attn_logits, row_lse, blank_logprob, text_len, mel_len,
                           alpha, beta, nll, M, L, S_pad);
    }
    hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, s, nll, text_len, B, L, loss);
    if (grad)
        hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, attn_logits, row_lse, blank_logprob,
                           text_len, mel_len, alpha, beta, nll, grad_out, grad, B, M, L, S_pad);
    return ispk_launch_status();
}

extern "C" int32_t ispk_attn_bin_loss_f32(const float* attn_soft, const int16_t* attn_hard, float eps, float* workspace,
                                          float* loss, float* grad, float grad_out, int32_t B, int32_t M, int32_t L,
                                          ispk_stream_t stream) {
    ISPK_REQUIRE(attn_soft && attn_hard && workspace && loss, -1, "ispk_attn_bin_loss_f32: null pointer");
    ISPK_REQUIRE(B >= 1 && M >= 1 && L >= 1, -2, "ispk_attn_bin_loss_f32: bad shape B=%d M=%d L=%d", B, M, L);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t rows = (int64_t)B * M;
    hipLaunchKernelGGL(bin_loss_stage1_kernel, dim3(kBinParts), dim3(256), 0, s, attn_soft, attn_hard, rows, L, eps, workspace);
    hipLaunchKernelGGL(bin_loss_stage2_kernel, dim3(1), dim3(kBinParts), 0, s, workspace, loss);
    if (grad)
        hipLaunchKernelGGL(bin_loss_grad_kernel, dim3(1024), dim3(256), 0, s, attn_soft, attn_hard, rows, L, eps, loss, grad_out,
                           grad);
    return ispk_launch_status();
}

extern "C" int32_t ispk_mel_grad_rows_f32(const float* dmel, const uint8_t* mask, float* g, int32_t B, int32_t C, int32_t T,
                                          ispk_stream_t stream) {
    ISPK_REQUIRE(dmel && g, -1, "ispk_mel_grad_rows_f32: null pointer");
    ISPK_REQUIRE(B >= 1 && C >= 1 && T >= 1 && B <= 65535, -2, "ispk_mel_grad_rows_f32: bad shape B=%d C=%d T=%d", B, C, T);
    hipLaunchKernelGGL(mel_grad_rows_kernel, dim3((T + 31) / 32, (C + 31) / 32, B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), dmel, mask, g, C, T);
    return ispk_launch_status();
}

extern "C" int32_t ispk_smallk_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t rows, int32_t N, int32_t K,
                                         float* workspace, int64_t workspace_floats, float* out, ispk_stream_t stream) {
    ISPK_REQUIRE(g && x && workspace && out, -1, "ispk_smallk_wgrad_f32: null pointer");
    ISPK_REQUIRE(rows >= 1 && N >= 1 && K >= 1 && K <= 8 && ldg >= N && ldx >= K && workspace_floats >= (int64_t)kColParts * N * K, -2,
                 "ispk_smallk_wgrad_f32: rows=%lld N=%d K=%d (K <= 8), workspace needs %lld floats", (long long)rows, N, K,
                 (long long)kColParts * N * K);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(smallk_wgrad_stage1_kernel, dim3(kColParts), dim3(256), 0, s, g, ldg, x, ldx, rows, N, K, workspace);
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((N * K + 255) / 256), dim3(256), 0, s, workspace, N * K, out);
    return ispk_launch_status();
}

extern "C" int32_t ispk_embedding_bwd_f32(const int64_t* ids, const float* d_emb, int64_t rows, int32_t dim, int32_t vocab,
                                          int32_t padding_idx, float* d_table, int64_t ld_table, ispk_stream_t stream) {
    ISPK_REQUIRE(ids && d_emb && d_table, -1, "ispk_embedding_bwd_f32: null pointer");
    ISPK_REQUIRE(rows >= 0 && dim >= 1 && dim <= 512 && vocab >= 1 && ld_table >= dim, -2, "ispk_embedding_bwd_f32: bad shape rows=%lld dim=%d (<= 512) vocab=%d",
                 (long long)rows, dim, vocab);
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(vocab), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), ids, d_emb, rows, dim,
                       padding_idx, d_table, ld_table);
    return ispk_launch_status();
}

extern "C" int32_t ispk_colsum_f32(const float* x, int64_t ldx, int64_t rows, int32_t cols, const uint8_t* row_mask, float* workspace,
                                   int64_t workspace_floats, float* out, ispk_stream_t stream) {
    ISPK_REQUIRE(x && workspace && out, -1, "ispk_colsum_f32: null pointer");
    ISPK_REQUIRE(rows >= 1 && cols >= 1 && ldx >= cols && workspace_floats >= (int64_t)kColParts * cols, -2,
                 "ispk_colsum_f32: rows=%lld cols=%d, workspace needs %lld floats", (long long)rows, cols,
                 (long long)kColParts * cols);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(colsum_stage1_kernel, dim3(kColParts), dim3(256), 0, s, x, ldx, rows, cols, row_mask, workspace);
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, workspace, cols, out);
    return ispk_launch_status();
}

extern "C" int32_t ispk_grad_sqnorm_f32(const float* g, int64_t n, float* partial, float* out, ispk_stream_t stream) {
    ISPK_REQUIRE(g && partial && out, -1, "ispk_grad_sqnorm_f32: null pointer");
    ISPK_REQUIRE(n >= 0 && ispk_aligned(g, 16) && ispk_aligned(partial, 8), -2,
                 "ispk_grad_sqnorm_f32: n < 0, g not 16-byte aligned or partial not 8-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* part = reinterpret_cast<double*>(partial);
    hipLaunchKernelGGL(sqnorm_stage1_kernel, dim3(kNormBlocks), dim3(256), 0, s, g, n, part);
    hipLaunchKernelGGL(sqnorm_stage2_kernel, dim3(1), dim3(kNormBlocks), 0, s, part, out);
    return ispk_launch_status();
}

extern "C" int32_t ispk_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                  const float* grad_sqnorm, float max_norm, float grad_scale, ispk_stream_t stream) {
    ISPK_REQUIRE(p && g && m && v, -1, "ispk_adamw_f32: null pointer");
    ISPK_REQUIRE(n >= 0 && n_decay >= 0 && n_decay <= n && step >= 1, -2,
                 "ispk_adamw_f32: need 0 <= n_decay <= n and step >= 1 (got n=%lld n_decay=%lld step=%d)", (long long)n,
                 (long long)n_decay, step);
    ISPK_REQUIRE(ispk_aligned(p, 16) && ispk_aligned(g, 16) && ispk_aligned(m, 16) && ispk_aligned(v, 16), -3,
                 "ispk_adamw_f32: arenas must be 16-byte aligned");
    ISPK_REQUIRE(!grad_sqnorm || max_norm > 0.f, -4, "ispk_adamw_f32: max_norm must be positive when clipping");
    if (n == 0) return 0;
    // scalar factors in double, as torch's python-side arithmetic computes them, then rounded once
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamArgs a;
    a.decay_mul = (float)(1.0 - (double)lr * (double)weight_decay);
    a.one_m_b1 = (float)(1.0 - (double)beta1);
    a.b2 = beta2;
    a.one_m_b2 = (float)(1.0 - (double)beta2);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    a.eps = eps;
    a.max_norm = max_norm;
    a.grad_scale = grad_scale;
    const int64_t n4 = (n + 3) >> 2;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n, n_decay,
                       grad_sqnorm, a, static_cast<const AdamArgs*>(nullptr));
    return ispk_launch_status();
}

static_assert(sizeof(AdamArgs) == sizeof(ispk_adam_args_t), "ispk_adam_args_t mirrors AdamArgs");

extern "C" int32_t ispk_adam_args_f32(float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                                      float grad_scale, ispk_adam_args_t* out) {
    ISPK_REQUIRE(out && step >= 1, -1, "ispk_adam_args_f32: null output or step < 1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamArgs a;
    a.decay_mul = (float)(1.0 - (double)lr * (double)weight_decay);
    a.one_m_b1 = (float)(1.0 - (double)beta1);
    a.b2 = beta2;
    a.one_m_b2 = (float)(1.0 - (double)beta2);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    a.eps = eps;
    a.max_norm = max_norm;
    a.grad_scale = grad_scale;
    memcpy(out, &a, sizeof(a));
    return 0;
}

extern "C" int32_t ispk_adamw_f32_dev(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_decay,
                                      const ispk_adam_args_t* args_dev, const float* grad_sqnorm, ispk_stream_t stream) {
    ISPK_REQUIRE(p && g && m && v && args_dev, -1, "ispk_adamw_f32_dev: null pointer");
    ISPK_REQUIRE(n >= 0 && n_decay >= 0 && n_decay <= n, -2, "ispk_adamw_f32_dev: need 0 <= n_decay <= n");
    ISPK_REQUIRE(ispk_aligned(p, 16) && ispk_aligned(g, 16) && ispk_aligned(m, 16) && ispk_aligned(v, 16) && ispk_aligned(args_dev, 4), -3,
                 "ispk_adamw_f32_dev: arenas must be 16-byte aligned");
    if (n == 0) return 0;
    const int64_t n4 = (n + 3) >> 2;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n, n_decay,
                       grad_sqnorm, AdamArgs{}, reinterpret_cast<const AdamArgs*>(args_dev));
    return ispk_launch_status();
}

extern "C" int32_t ispk_mel_loss_f32(const float* mel_out, const float* mel_target, const int64_t* mel_len, float* ratio,
                                     float* loss, float* grad, float grad_out, int32_t B, int32_t C, int32_t T,
                                     ispk_stream_t stream) {
    ISPK_REQUIRE(mel_out && mel_target && mel_len && ratio && loss, -1, "ispk_mel_loss_f32: null pointer");
    ISPK_REQUIRE(B >= 1 && C >= 1 && T >= 1, -2, "ispk_mel_loss_f32: empty batch (B=%d C=%d T=%d)", B, C, T);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mel_loss_kernel, dim3(B), dim3(1024), 0, s, mel_out, mel_target, mel_len, ratio, grad, grad_out, B, C, T);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, s, ratio, B, loss);
    return ispk_launch_status();
}
