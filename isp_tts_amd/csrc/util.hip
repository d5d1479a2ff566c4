// Small data-movement kernels of the training step, so that a step issues no PyTorch (ATen) kernel at all: gradient delivery
// into the optimizer arena, concatenation / casting of weights that change every step, zero fills, scalar algebra on the
// losses.  All launches are plain stream work (graph-capturable), none synchronises.
#include "common.h"

namespace {

constexpr int kSegMax = 32;
struct SegPack {
    const float* src[kSegMax];
    void* dst[kSegMax];
    int64_t n[kSegMax];
    int32_t mode[kSegMax];
};

// one segment per blockIdx.y: dst = src (mode 0), dst += src (1), dst = bf16(src) (2); fp32 sources, contiguous both sides
__global__ __launch_bounds__(256) void segments_kernel(SegPack p) {
    const int sg = blockIdx.y;
    const float* __restrict__ src = p.src[sg];
    const int64_t n = p.n[sg];
    const int mode = p.mode[sg];
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (mode == 2) {
        uint16_t* __restrict__ dst = static_cast<uint16_t*>(p.dst[sg]);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = f32_to_bf16(src[i]);
    } else {
        float* __restrict__ dst = static_cast<float*>(p.dst[sg]);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = mode == 1 ? dst[i] + src[i] : src[i];
    }
}

__global__ __launch_bounds__(256) void fill_zero_kernel(uint32_t* __restrict__ p, int64_t words, uint8_t* __restrict__ tail, int tail_bytes) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) p[i] = 0u;
    if (blockIdx.x == 0 && (int)threadIdx.x < tail_bytes) tail[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ s_dev, float s_host) {
    const float s = s_dev ? s_dev[0] * s_host : s_host;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] *= s;
}

struct ScalarPack {
    const float* p[8];
    float w[8];
    int n;
};
__global__ void sum_scalars_kernel(ScalarPack s, float* __restrict__ out) {
    float v = 0.f;
    for (int i = 0; i < s.n; ++i) v += s.w[i] * s.p[i][0];     // in index order
    out[0] = v;
}

// dst[i] = exp(src[i]) for i < n, 0 for n <= i < total (ALiBi slopes from their logarithms, embeddings.py:66-82); mode 1:
// dst[i] = sqrt(src[i]) * scale (the gradient norm from its square)
__global__ __launch_bounds__(64) void unary_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int total, int mode,
                                                   float scale) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= total) return;
    dst[i] = i < n ? (mode == 1 ? sqrtf(src[i]) * scale : expf(src[i])) : 0.f;
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd, int rows,
                                                     int cols) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds_ + c];
}

// dst[a][c][b] = src[a][b][flip ? B - 1 - b : b][c] ... i.e. the last two axes of [A][B][C] swapped (optionally with the B axis
// reversed first): Conv1d weights [O][C][k] <-> GEMM weights [O][k][C] and their gradients
__global__ __launch_bounds__(256) void permute021_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int B, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)A * B * C) return;
    const int c = (int)(i % C), b = (int)((i / C) % B), a = (int)(i / ((int64_t)B * C));
    dst[((int64_t)a * C + c) * B + b] = src[i];
}
// wf[c][j * O + o] = w[o][c][K - 1 - j]: the flipped-tap GEMM weight of a convolution's input gradient
__global__ __launch_bounds__(256) void conv_flip_kernel(const float* __restrict__ w, float* __restrict__ wf, int O, int C, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)O * C * K) return;
    const int k = (int)(i % K), c = (int)((i / K) % C), o = (int)(i / ((int64_t)K * C));
    wf[(int64_t)c * K * O + (int64_t)(K - 1 - k) * O + o] = w[i];
}

// Weight images for the kernels from fp32 parameters, many per launch (a training step re-stages every image after every
// update): 32 x 32 tiles through LDS; flags 1 = transposed image dst[c][r], 2 = bf16 output, 4 = exp() of the values (ALiBi
// slopes from their logarithms); dst rows ld_dst elements apart.
constexpr int kStageMax = 16;
struct StagePack {
    const float* src[kStageMax];
    void* dst[kStageMax];
    int64_t ld_dst[kStageMax];
    int32_t rows[kStageMax], cols[kStageMax], flags[kStageMax];
};
__global__ __launch_bounds__(256) void stage_kernel(StagePack p) {
    __shared__ float t[32][33];
    const int sg = blockIdx.y;
    const float* __restrict__ src = p.src[sg];
    const int rows = p.rows[sg], cols = p.cols[sg], flags = p.flags[sg];
    const int64_t ldd = p.ld_dst[sg];
    const bool tr = flags & 1, b16 = flags & 2, ex = flags & 4;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tcols = (cols + 31) / 32, ntiles = ((rows + 31) / 32) * tcols;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int r0 = (tile / tcols) * 32, c0 = (tile % tcols) * 32;
        __syncthreads();
        for (int k = ty; k < 32; k += 8)
            if (r0 + k < rows && c0 + tx < cols) {
                const float v = src[(int64_t)(r0 + k) * cols + c0 + tx];
                t[k][tx] = ex ? expf(v) : v;
            }
        __syncthreads();
        for (int k = ty; k < 32; k += 8) {
            // plain image: element (r0 + k, c0 + tx); transposed image: row c0 + k, column r0 + tx
            const int sr = tr ? tx : k, sc = tr ? k : tx;
            if (r0 + sr >= rows || c0 + sc >= cols) continue;
            const float v = t[sr][sc];
            const int64_t at = tr ? (int64_t)(c0 + k) * ldd + r0 + tx : (int64_t)(r0 + k) * ldd + c0 + tx;
            if (b16) static_cast<uint16_t*>(p.dst[sg])[at] = f32_to_bf16(v);
            else static_cast<float*>(p.dst[sg])[at] = v;
        }
    }
}

}  // namespace

extern "C" int32_t ispk_stage_weights(const ispk_stage_t* segs, int32_t nseg, ispk_stream_t stream) {
    ISPK_REQUIRE(nseg >= 0 && (nseg == 0 || segs), ISPK_E_NULL, "stage_weights: null table");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int base = 0; base < nseg; base += kStageMax) {
        StagePack p;
        const int cnt = nseg - base < kStageMax ? nseg - base : kStageMax;
        int most = 0;
        for (int i = 0; i < cnt; ++i) {
            const ispk_stage_t& g = segs[base + i];
            const bool tr = g.flags & 1;
            ISPK_REQUIRE(g.src && g.dst && g.rows >= 1 && g.cols >= 1 && g.ld_dst >= (tr ? g.rows : g.cols) && (g.flags & ~7) == 0, ISPK_E_SHAPE,
                         "stage_weights: bad segment %d", base + i);
            p.src[i] = g.src; p.dst[i] = g.dst; p.ld_dst[i] = g.ld_dst; p.rows[i] = g.rows; p.cols[i] = g.cols; p.flags[i] = g.flags;
            const int tiles = ((g.rows + 31) / 32) * ((g.cols + 31) / 32);
            most = tiles > most ? tiles : most;
        }
        most = most > 128 ? 128 : most;
        hipLaunchKernelGGL(stage_kernel, dim3(most, cnt), dim3(256), 0, s, p);
    }
    return ispk_launch_status();
}

extern "C" int32_t ispk_segments_f32(const ispk_segment_t* segs, int32_t nseg, ispk_stream_t stream) {
    ISPK_REQUIRE(nseg >= 0 && (nseg == 0 || segs), ISPK_E_NULL, "segments: null table");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int base = 0; base < nseg; base += kSegMax) {
        SegPack p;
        const int cnt = nseg - base < kSegMax ? nseg - base : kSegMax;
        int64_t longest = 0;
        for (int i = 0; i < cnt; ++i) {
            const ispk_segment_t& g = segs[base + i];
            ISPK_REQUIRE(g.src && g.dst && g.n >= 0 && g.mode >= 0 && g.mode <= 2, ISPK_E_SHAPE, "segments: bad segment %d", base + i);
            p.src[i] = g.src; p.dst[i] = g.dst; p.n[i] = g.n; p.mode[i] = g.mode;
            longest = g.n > longest ? g.n : longest;
        }
        if (longest == 0) continue;
        int64_t bx = (longest + 1023) / 1024;
        bx = bx > 256 ? 256 : bx;
        hipLaunchKernelGGL(segments_kernel, dim3((unsigned)bx, cnt), dim3(256), 0, s, p);
    }
    return ispk_launch_status();
}

extern "C" int32_t ispk_fill_zero(void* p, int64_t bytes, ispk_stream_t stream) {
    ISPK_REQUIRE(bytes >= 0 && (bytes == 0 || p) && ispk_aligned(p, 4), ISPK_E_ALIGN, "fill_zero: 4-byte aligned buffer");
    if (bytes == 0) return 0;
    const int64_t words = bytes / 4;
    int64_t bx = (words + 1023) / 1024;
    bx = bx < 1 ? 1 : (bx > 2048 ? 2048 : bx);
    hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)bx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<uint32_t*>(p),
                       words, static_cast<uint8_t*>(p) + 4 * words, (int)(bytes - 4 * words));
    return ispk_launch_status();
}

extern "C" int32_t ispk_scale_f32(float* x, int64_t n, const float* s_dev, float s_host, ispk_stream_t stream) {
    ISPK_REQUIRE(n >= 0 && (n == 0 || x), ISPK_E_NULL, "scale: null pointer");
    if (n == 0) return 0;
    int64_t bx = (n + 1023) / 1024;
    bx = bx > 1024 ? 1024 : bx;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)bx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n, s_dev, s_host);
    return ispk_launch_status();
}

extern "C" int32_t ispk_sum_scalars_f32(const float* const* terms, const float* weights, int32_t n, float* out, ispk_stream_t stream) {
    ISPK_REQUIRE(terms && out && n >= 1 && n <= 8, ISPK_E_SHAPE, "sum_scalars: 1..8 terms");
    ScalarPack s;
    s.n = n;
    for (int i = 0; i < n; ++i) {
        ISPK_REQUIRE(terms[i], ISPK_E_NULL, "sum_scalars: null term %d", i);
        s.p[i] = terms[i];
        s.w[i] = weights ? weights[i] : 1.0f;
    }
    hipLaunchKernelGGL(sum_scalars_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), s, out);
    return ispk_launch_status();
}

extern "C" int32_t ispk_exp_pad_f32(const float* src, float* dst, int32_t n, int32_t total, ispk_stream_t stream) {
    ISPK_REQUIRE(src && dst && n >= 0 && total >= n, ISPK_E_SHAPE, "exp_pad: bad arguments");
    if (total == 0) return 0;
    hipLaunchKernelGGL(unary_kernel, dim3((total + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), src, dst, n, total, 0, 1.0f);
    return ispk_launch_status();
}

extern "C" int32_t ispk_sqrt_scale_f32(const float* src, float* dst, int32_t n, float scale, ispk_stream_t stream) {
    ISPK_REQUIRE(src && dst && n >= 0, ISPK_E_SHAPE, "sqrt_scale: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(unary_kernel, dim3((n + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), src, dst, n, n, 1, scale);
    return ispk_launch_status();
}

extern "C" int32_t ispk_copy2d_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int32_t rows, int32_t cols,
                                   ispk_stream_t stream) {
    ISPK_REQUIRE(src && dst && rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols, ISPK_E_SHAPE, "copy2d: bad arguments");
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return 0;
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, ld_src, dst,
                       ld_dst, rows, cols);
    return ispk_launch_status();
}

extern "C" int32_t ispk_permute021_f32(const float* src, float* dst, int32_t A, int32_t B, int32_t C, ispk_stream_t stream) {
    ISPK_REQUIRE(src && dst && A >= 0 && B >= 0 && C >= 0, ISPK_E_SHAPE, "permute021: bad arguments");
    const int64_t n = (int64_t)A * B * C;
    if (n == 0) return 0;
    hipLaunchKernelGGL(permute021_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, A, B, C);
    return ispk_launch_status();
}

extern "C" int32_t ispk_conv_weight_flip_f32(const float* w, float* wf, int32_t O, int32_t C, int32_t K, ispk_stream_t stream) {
    ISPK_REQUIRE(w && wf && O >= 0 && C >= 0 && K >= 1, ISPK_E_SHAPE, "conv_weight_flip: bad arguments");
    const int64_t n = (int64_t)O * C * K;
    if (n == 0) return 0;
    hipLaunchKernelGGL(conv_flip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, wf, O, C, K);
    return ispk_launch_status();
}
