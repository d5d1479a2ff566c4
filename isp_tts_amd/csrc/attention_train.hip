// Training attention of an AMP step on bf16 operands: ALiBi-biased multi-query attention with attention dropout
// (attend.py:49-122, embeddings.py:51-82, attention.py:128-152; the reference trains under autocast, recipes/default.yaml:56,
// so SDPA and its backward take bf16 q / k / v / dO and return bf16).  Three kernels, all on v_mfma_f32_32x32x16_bf16 with
// fp32 statistics, every tile that more than one wave reads staged ONCE per workgroup in LDS:
//
//   attn_train_fwd_bf16_kernel   O = dropout(softmax(S)) V and the rows' log-sum-exp, one pass (running maximum);
//   attn_bwd_dq_bf16_kernel      delta = rowsum(O dO), dQ = scale dS K, the slope-gradient partials;
//   attn_bwd_dkv_bf16_kernel     dK = scale sum_h dS_h^T Q_h, dV = sum_h Pdrop_h^T dO_h (the heads share K / V: summed in the
//                                workgroup in head order - no atomics, a fixed summation order).
//
// Decomposition.  Forward and dQ: a workgroup = one (batch item, 64-query tile) for ALL heads, wave = (head, 32-query half),
// S^T = K Q^T so that a lane owns one query's row; the item's K and V rows (up to 512 at a time) are staged once and shared
// by the 2 H waves.  dK / dV: a workgroup = one (batch item, 64-key tile), wave = (head, 32-key half) with its K / V rows as
// register operands and S = Q K^T so that a lane owns one key; the heads' Q and dO rows stream through a double-buffered LDS
// image in 32-query steps.  Accumulator tiles (P, dS) feed the next product as its B operand without moving; the matching A
// operand - the transposed tile - comes from the SAME row-major LDS image through ds_read_b64_tr_b16.
//
// LDS image of a [rows][64] bf16 tile: plain 128-byte rows, 16-byte slot s of row r stored at slot s ^ sw(r),
// sw(r) = ((r >> 1) & 1) << 2 | (r >> 2) & 3.  Over the lane groups of a ds_read_b128 (rows 0-3, 12-15, 20-27 ...) the eight row
// pairs take eight different slots, and the four rows of a transposed read's lane group fall on disjoint banks (bit 2 of the
// XOR separates the row pairs that share a parity): both kinds of read are conflict-free on one image.
//
// Dropout is the counter-based mask of dropout.h on the element index ((b H + h) N + i) N + j: the three kernels evaluate the
// same function, nothing is stored.
#include "common.h"
#include "dropout.h"

namespace {

typedef uint32_t au32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));

constexpr float kAtLog2e = 1.4426950408889634f, kAtLn2 = 0.6931471805599453f;
constexpr int kAtResKeys = 512;        // keys staged per round of the forward / dQ kernels

__device__ __forceinline__ uint32_t at_pack(float lo, float hi) {   // one v_cvt_pk_bf16_f32
    f32x2 v;
    v.x = lo; v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, abf16x2));
}
__device__ __forceinline__ int at_sw(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

template <int OFF>
__device__ __forceinline__ void at_read_tr(au32x2& dst, uint32_t lds_byte_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF) : "memory");
}

// Per-lane byte offsets into a 32-row block of an image.
//   row[ks]: A / row operand of k-step ks - row l31, logical slot 2 ks + h (dims 16 ks + 8 h .. + 7);
//   tr[dt][run]: transposed operand of dim tile dt - the lane's 16-lane group is (h, dim half dh); lane 4 qq + pp of it points
//   at row 4 h + qq (+ 8 run; + 16 per k-step, an immediate), dims 4 pp .. 4 pp + 3 of the group's 16-dim block.
struct AtLane {
    uint32_t row[4];
    uint32_t tr[2][2];
};
__device__ __forceinline__ AtLane at_lane(int lane) {
    const int l31 = lane & 31, h = lane >> 5, qq = (lane & 15) >> 2, pp = lane & 3, dh = (lane >> 4) & 1;
    AtLane a;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a.row[ks] = (uint32_t)(l31 * 128 + (((2 * ks + h) ^ at_sw(l31)) << 4));
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int run = 0; run < 2; ++run) {
            const int r = 4 * h + qq + 8 * run;
            a.tr[dt][run] = (uint32_t)(r * 128 + (((4 * dt + 2 * dh + (pp >> 1)) ^ at_sw(r)) << 4) + 8 * (pp & 1));
        }
    return a;
}
__device__ __forceinline__ void at_read_rows(bf16x8 (&f)[4], uint32_t base, const AtLane& a) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) lds_read_b128_asm<0>(f[ks], base + a.row[ks]);
}
// t[st][dt][run]: rows 16 st + 8 run + 4 h .. + 3 of this lane's dim of tile dt
__device__ __forceinline__ void at_read_trs(au32x2 (&t)[2][2][2], uint32_t base, const AtLane& a) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int run = 0; run < 2; ++run) {
            at_read_tr<0>(t[0][dt][run], base + a.tr[dt][run]);
            at_read_tr<2048>(t[1][dt][run], base + a.tr[dt][run]);
        }
}
__device__ __forceinline__ bf16x8 at_frag(const au32x2& r0, const au32x2& r1) {
    union { uint32_t u[4]; bf16x8 f; } x;
    x.u[0] = r0[0]; x.u[1] = r0[1]; x.u[2] = r1[0]; x.u[3] = r1[1];
    return x.f;
}
// an accumulator tile as the next product's B operand: k-step st = registers 8 st .. 8 st + 7
__device__ __forceinline__ void at_pack_acc(const float (&w)[16], bf16x8 (&f)[2]) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        union { uint32_t u[4]; bf16x8 v; } x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x.u[e] = at_pack(w[8 * st + 2 * e], w[8 * st + 2 * e + 1]);
        f[st] = x.v;
    }
}
// acc[dt] += tile^T (dims 32 dt ..) x w over the tile's 32 rows
__device__ __forceinline__ void at_tr_mma(f32x16 (&acc)[2], const au32x2 (&t)[2][2][2], const bf16x8 (&w)[2]) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(t[st][0][0], t[st][0][1]), w[st], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_frag(t[st][1][0], t[st][1][1]), w[st], acc[1], 0, 0, 0);
    }
}
__device__ __forceinline__ f32x16 at_dot(const bf16x8 (&a)[4], const bf16x8 (&b)[4]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[ks], acc, 0, 0, 0);
    return acc;
}
// B / column operand of row `row` (clamped by the caller) straight from global memory: dims 16 ks + 8 h .. + 7
__device__ __forceinline__ void at_load_cols(bf16x8 (&f)[4], const uint16_t* rowp, int h) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const bf16x8*>(rowp + ks * 16 + h * 8);
}
// a transposed accumulator pair acc[mt][r] (dim 32 mt + (r & 3) + 8 (r >> 2) + 4 h of the lane's row) -> bf16 row pieces
__device__ __forceinline__ void at_store_row(uint16_t* rowp, const f32x16 (&acc)[2], float mul, int h) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 v;
            v.x = at_pack(acc[mt][4 * g] * mul, acc[mt][4 * g + 1] * mul);
            v.y = at_pack(acc[mt][4 * g + 2] * mul, acc[mt][4 * g + 3] * mul);
            *reinterpret_cast<uint2*>(rowp + 32 * mt + 8 * g + 4 * h) = v;
        }
}
// XCD-aware (batch item, tile) mapping: workgroups go to the 8 XCDs round-robin in linear order; within each run of
// 8 gridDim.x workgroups item = 8 run + linear % 8, so the tiles of one item - which stage the same rows - share an L2.
__device__ __forceinline__ void at_block(int& b, int& bx) {
    b = blockIdx.y;
    bx = blockIdx.x;
    const int gx = gridDim.x, lin = blockIdx.y * gx + blockIdx.x, run = lin / (8 * gx);
    if ((run + 1) * 8 <= (int)gridDim.y) {
        const int r = lin - run * 8 * gx;
        b = run * 8 + (r & 7);
        bx = r >> 3;
    }
}
// K and V rows key0 .. key0 + rows - 1 of one item (256 contiguous bytes per key in qkv) into the two images; rows past N: zeros
__device__ __forceinline__ void at_stage_kv(char* Kl, char* Vl, const uint16_t* kv, int64_t ld, int key0, int rows, int N, int tid,
                                            int nt) {
#pragma unroll 4
    for (int idx = tid; idx < rows * 16; idx += nt) {
        const int r = idx >> 4, s = idx & 15, key = key0 + r;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (key < N) v = *reinterpret_cast<const u32x4*>(kv + (int64_t)key * ld + s * 8);
        char* dst = (s >= 8 ? Vl : Kl) + r * 128 + (((s & 7) ^ at_sw(r)) << 4);
        *reinterpret_cast<u32x4*>(dst) = v;
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <int MAXT, bool kDrop>
__global__ __launch_bounds__(MAXT) void attn_train_fwd_bf16_kernel(const uint16_t* __restrict__ qkv, int64_t ld,
                                                                   const float* __restrict__ slopes,
                                                                   const int64_t* __restrict__ key_len, uint16_t* __restrict__ o,
                                                                   int64_t ldo, float* __restrict__ lse, int N, int H, int kvrows,
                                                                   uint32_t thresh, float inv_keep, uint64_t seed,
        const uint64_t* __restrict__ seed_src) {
    seed = run_seed(seed, seed_src);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* const Kl = smem_raw;
    char* const Vl = smem_raw + (size_t)kvrows * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = wave % H, qhalf = wave / H, l31 = lane & 31, h = lane >> 5;
    int b, bx;
    at_block(b, bx);
    int klen = key_len ? (int)key_len[b] : N;
    klen = klen < 1 ? 1 : (klen > N ? N : klen);
    const int q0 = bx * 64 + qhalf * 32, qi = q0 + l31;
    const uint16_t* qb = qkv + (int64_t)b * N * ld;
    bf16x8 qf[4];
    at_load_cols(qf, qb + (int64_t)(qi < N ? qi : N - 1) * ld + head * 64, h);
    const AtLane ln = at_lane(lane);
    const uint32_t kbase = lds_addr(Kl), vbase = lds_addr(Vl);
    const float scale2 = 0.125f * kAtLog2e, slope2 = slopes[head] * kAtLog2e, ninf = -__builtin_huge_valf();
    const uint32_t row_idx = (((uint32_t)b * H + head) * N + (uint32_t)(qi < N ? qi : 0)) * (uint32_t)N;
    f32x16 ot[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[0][r] = ot[1][r] = 0.f;
    float m = ninf, l = 0.f;
    for (int kc0 = 0; kc0 < klen; kc0 += kAtResKeys) {
        if (kc0) __syncthreads();
        const int span = klen - kc0 < kAtResKeys ? klen - kc0 : kAtResKeys, nblk = (span + 31) >> 5;
        at_stage_kv(Kl, Vl, qb + H * 64, ld, kc0, nblk * 32, N, tid, blockDim.x);
        __syncthreads();
#pragma unroll 1
        for (int blk = 0; blk < nblk; ++blk) {
            const int key0 = kc0 + blk * 32;
            bf16x8 kf[4];
            at_read_rows(kf, kbase + blk * 4096, ln);
            au32x2 vt[2][2][2];
            at_read_trs(vt, vbase + blk * 4096, ln);
            lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            const f32x16 s = at_dot(kf, qf);
            const float d0 = (float)(key0 + 4 * h - qi);
            const bool edge = key0 + 32 > klen;              // wave-uniform
            float p[16], bmax = ninf;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = (r & 3) + 8 * (r >> 2);
                float x = fmaf(s[r], scale2, -slope2 * fabsf(d0 + (float)kr));
                if (edge) x = key0 + kr + 4 * h < klen ? x : ninf;
                p[r] = x;
                bmax = fmaxf(bmax, x);
            }
            bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));     // both halves of a query: the row's block maximum (key0 < klen: finite)
            const float mnew = fmaxf(m, bmax), alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(p[r] - mnew);
                psum += p[r];
            }
            l = fmaf(l, alpha, psum);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ot[0][r] *= alpha;
                ot[1][r] *= alpha;
            }
            if constexpr (kDrop) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t key = (uint32_t)(key0 + (r & 3) + 8 * (r >> 2) + 4 * h);
                    p[r] = drop_keep(seed, row_idx + key, thresh) ? p[r] * inv_keep : 0.f;
                }
            }
            bf16x8 pw[2];
            at_pack_acc(p, pw);
            at_tr_mma(ot, vt, pw);
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (qi < N) {
        at_store_row(o + ((int64_t)b * N + qi) * ldo + head * 64, ot, 1.0f / l, h);
        if (h == 0) lse[((int64_t)b * H + head) * N + qi] = (m + __builtin_amdgcn_logf(l)) * kAtLn2;   // v_log_f32 = log2
    }
}

// ------------------------------------------------------------------------------------------------ dQ
template <int MAXT, bool kDrop>
__global__ __launch_bounds__(MAXT) void attn_bwd_dq_bf16_kernel(const uint16_t* __restrict__ qkv, int64_t ld,
                                                                const uint16_t* __restrict__ o, const uint16_t* __restrict__ dout,
                                                                int64_t ldo, const float* __restrict__ slopes,
                                                                const int64_t* __restrict__ key_len, const float* __restrict__ lse,
                                                                uint16_t* __restrict__ dqkv, float* __restrict__ delta,
                                                                float* __restrict__ slope_part, int N, int H, int kvrows,
                                                                uint32_t thresh, float inv_keep, uint64_t seed,
        const uint64_t* __restrict__ seed_src) {
    seed = run_seed(seed, seed_src);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* const Kl = smem_raw;
    char* const Vl = smem_raw + (size_t)kvrows * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = wave % H, qhalf = wave / H, l31 = lane & 31, h = lane >> 5;
    int b, bx;
    at_block(b, bx);
    int klen = key_len ? (int)key_len[b] : N;
    klen = klen < 1 ? 1 : (klen > N ? N : klen);
    const int q0 = bx * 64 + qhalf * 32, qi = q0 + l31, qrow = qi < N ? qi : N - 1;
    const uint16_t* qb = qkv + (int64_t)b * N * ld;
    bf16x8 qf[4], dof[4];
    at_load_cols(qf, qb + (int64_t)qrow * ld + head * 64, h);
    at_load_cols(dof, dout + ((int64_t)b * N + qrow) * ldo + head * 64, h);
    float dl = 0.f;
    {
        bf16x8 of[4];
        at_load_cols(of, o + ((int64_t)b * N + qrow) * ldo + head * 64, h);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dl = fmaf(bf16_to_f32((uint16_t)of[ks][e]), bf16_to_f32((uint16_t)dof[ks][e]), dl);
        dl += __shfl_xor(dl, 32, 64);
    }
    const int64_t stat = ((int64_t)b * H + head) * N + qi;
    const float L2 = qi < N ? lse[stat] * kAtLog2e : __builtin_huge_valf();     // rows past N: P = exp2(-inf) = 0
    if (h == 0 && qi < N) delta[stat] = dl;
    const AtLane ln = at_lane(lane);
    const uint32_t kbase = lds_addr(Kl), vbase = lds_addr(Vl);
    const float scale2 = 0.125f * kAtLog2e, slope2 = slopes[head] * kAtLog2e;
    const uint32_t row_idx = (((uint32_t)b * H + head) * N + (uint32_t)(qi < N ? qi : 0)) * (uint32_t)N;
    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[0][r] = dq[1][r] = 0.f;
    float gs = 0.f;
    for (int kc0 = 0; kc0 < klen; kc0 += kAtResKeys) {
        if (kc0) __syncthreads();
        const int span = klen - kc0 < kAtResKeys ? klen - kc0 : kAtResKeys, nblk = (span + 31) >> 5;
        at_stage_kv(Kl, Vl, qb + H * 64, ld, kc0, nblk * 32, N, tid, blockDim.x);
        __syncthreads();
#pragma unroll 1
        for (int blk = 0; blk < nblk; ++blk) {
            const int key0 = kc0 + blk * 32;
            f32x16 s, dp;
            {
                bf16x8 kf[4], vf[4];
                at_read_rows(kf, kbase + blk * 4096, ln);
                at_read_rows(vf, vbase + blk * 4096, ln);
                lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
                s = at_dot(kf, qf);
                dp = at_dot(vf, dof);
            }
            au32x2 kt[2][2][2];                  // lands under the softmax arithmetic below
            at_read_trs(kt, kbase + blk * 4096, ln);
            const float d0 = (float)(key0 + 4 * h - qi);
            const bool edge = key0 + 32 > klen;
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = (r & 3) + 8 * (r >> 2);
                const float dist = fabsf(d0 + (float)kr);
                float p = __builtin_amdgcn_exp2f(fmaf(s[r], scale2, -slope2 * dist) - L2);
                if (edge) p = key0 + kr + 4 * h < klen ? p : 0.f;
                float dpr = dp[r];
                if constexpr (kDrop) dpr = drop_keep(seed, row_idx + (uint32_t)(key0 + kr + 4 * h), thresh) ? dpr * inv_keep : 0.f;
                ds[r] = p * (dpr - dl);
                gs = fmaf(-ds[r], dist, gs);
            }
            bf16x8 dsw[2];
            at_pack_acc(ds, dsw);
            lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            at_tr_mma(dq, kt, dsw);          // dQ^T[d][i] += sum_j K[j][d] dS^T[j][i]
        }
    }
    if (qi < N) at_store_row(dqkv + ((int64_t)b * N + qi) * ld + head * 64, dq, 0.125f, h);
    for (int off = 32; off > 0; off >>= 1) gs += __shfl_xor(gs, off, 64);
    if (lane == 0) slope_part[((int64_t)head * gridDim.y + b) * (2 * gridDim.x) + 2 * bx + qhalf] = gs;
}

// ------------------------------------------------------------------------------------------------ dK / dV
// LDS: two step buffers, each H x (Q tile 4 KB | dO tile 4 KB) then H x (32 log2-domain LSE | 32 delta) floats; behind them
// the workgroup's own 64 K rows and 64 V rows as images (the waves' B operands: re-read per step, 32 registers freed).
// The step tiles arrive by LDS-DMA (global_load_lds_dwordx4, 8 rows x 128 B per wave instruction, the XOR applied on the source
// side; rows past N re-read row N - 1: their LSE is +inf, so P = dS = 0), four instructions per wave and step, one step ahead.
template <int MAXT, bool kDrop>
__global__ __launch_bounds__(MAXT) void attn_bwd_dkv_bf16_kernel(const uint16_t* __restrict__ qkv, int64_t ld,
                                                                 const uint16_t* __restrict__ dout, int64_t ldo,
                                                                 const float* __restrict__ slopes,
                                                                 const int64_t* __restrict__ key_len, const float* __restrict__ lse,
                                                                 const float* __restrict__ delta, uint16_t* __restrict__ dqkv, int N,
                                                                 int H, uint32_t thresh, float inv_keep, uint64_t seed,
        const uint64_t* __restrict__ seed_src) {
    seed = run_seed(seed, seed_src);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = wave % H, khalf = wave / H, l31 = lane & 31, h = lane >> 5;
    int b, kt;
    at_block(b, kt);
    int klen = key_len ? (int)key_len[b] : N;
    klen = klen < 1 ? 1 : (klen > N ? N : klen);
    const int key = kt * 64 + khalf * 32 + l31;
    const uint16_t* qb = qkv + (int64_t)b * N * ld;
    const uint16_t* dob = dout + (int64_t)b * N * ldo;
    const int buf_bytes = H * 8192 + H * 256;
    char* const Kl = smem_raw + 2 * buf_bytes;
    char* const Vl = Kl + 8192;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[0][r] = dk[1][r] = dv[0][r] = dv[1][r] = 0.f;
    if (kt * 64 < klen) {                                     // workgroup-uniform: some key of the tile is attended to
        const bool kvalid = key < klen;
        const AtLane ln = at_lane(lane);
        const float scale2 = 0.125f * kAtLog2e, slope2 = slopes[head] * kAtLog2e;
        const int nsteps = (N + 31) >> 5;
        const uint32_t kbase = lds_addr(Kl) + khalf * 4096, vbase = lds_addr(Vl) + khalf * 4096;
        float pst = 0.f;
        auto fetch = [&](int step, int bufi) {
            const int q0s = step * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int id = wave * 4 + i, hd = id >> 3, dten = (id >> 2) & 1, grp = id & 3;
                const int row = grp * 8 + (lane >> 3), q = q0s + row < N ? q0s + row : N - 1;
                const uint16_t* src = (dten ? dob + (int64_t)q * ldo : qb + (int64_t)q * ld) + hd * 64 + (((lane & 7) ^ at_sw(row)) << 3);
                char* dst = smem_raw + bufi * buf_bytes + hd * 8192 + dten * 4096 + grp * 1024;     // wave-uniform; lane L lands at + 16 L
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            if (tid < 64 * H) {
                const int hd = tid >> 6, which = (tid >> 5) & 1, q = q0s + (tid & 31);
                const int64_t at = ((int64_t)b * H + hd) * N + q;
                pst = q < N ? (which ? delta[at] : lse[at] * kAtLog2e) : (which ? 0.f : __builtin_huge_valf());
            }
        };
        auto put = [&](int bufi) {
            if (tid < 64 * H) reinterpret_cast<float*>(smem_raw + bufi * buf_bytes + H * 8192)[tid] = pst;
        };
        fetch(0, 0);
        at_stage_kv(Kl, Vl, qb + H * 64, ld, kt * 64, 64, N, tid, blockDim.x);
        put(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll 1
        for (int step = 0; step < nsteps; ++step) {
            if (step + 1 < nsteps) fetch(step + 1, (step + 1) & 1);
            const char* base = smem_raw + (step & 1) * buf_bytes;
            const uint32_t qbase = lds_addr(base + head * 8192), dbase = qbase + 4096;
            const float* st = reinterpret_cast<const float*>(base + H * 8192) + head * 64;
            const int q0s = step * 32;
            float p[16], ds[16];
            {
                bf16x8 qa[4], kf[4];
                at_read_rows(qa, qbase, ln);
                at_read_rows(kf, kbase, ln);
                lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 s = at_dot(qa, kf);              // S[q][key]
                const float d0 = (float)(q0s + 4 * h - key);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 Lg = *reinterpret_cast<const f32x4*>(st + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const float dist = fabsf(d0 + (float)(e + 8 * g));
                        const float x = __builtin_amdgcn_exp2f(fmaf(s[r], scale2, -slope2 * dist) - Lg[e]);
                        p[r] = kvalid ? x : 0.f;
                    }
                }
            }
            au32x2 t[2][2][2];
            {
                bf16x8 da[4], vf[4];
                at_read_rows(da, dbase, ln);
                at_read_rows(vf, vbase, ln);
                lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 dp = at_dot(da, vf);             // dP[q][key]
                at_read_trs(t, dbase, ln);                    // dO^T: lands under the arithmetic below
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 Dg = *reinterpret_cast<const f32x4*>(st + 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        if constexpr (kDrop) {
                            const uint32_t q = (uint32_t)(q0s + e + 8 * g + 4 * h);
                            const bool keep = drop_keep(seed, (((uint32_t)b * H + head) * N + (q < (uint32_t)N ? q : 0u)) * (uint32_t)N + (uint32_t)key, thresh);
                            ds[r] = p[r] * ((keep ? dp[r] * inv_keep : 0.f) - Dg[e]);
                            p[r] = keep ? p[r] * inv_keep : 0.f;           // dV takes the DROPPED probabilities
                        } else {
                            ds[r] = p[r] * (dp[r] - Dg[e]);
                        }
                    }
                }
            }
            bf16x8 pw[2], dsw[2];
            at_pack_acc(p, pw);
            at_pack_acc(ds, dsw);
            lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            at_tr_mma(dv, t, pw);                              // dV^T[d][key] += sum_q dO[q][d] Pdrop[q][key]
            __builtin_amdgcn_sched_barrier(0);
            at_read_trs(t, qbase, ln);
            lds_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            at_tr_mma(dk, t, dsw);                             // dK^T[d][key] += sum_q Q[q][d] dS[q][key]
            if (step + 1 < nsteps) put((step + 1) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // heads add into one image in index order (a fixed summation order); the waves of head 0 write the rows out
    float* red = reinterpret_cast<float*>(smem_raw) + khalf * (4 * 16 * 64);      // [dk mt0 | dk mt1 | dv mt0 | dv mt1][r][lane]
    for (int hh = 0; hh < H; ++hh) {
        if (head == hh) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* a = red + (mt * 16 + r) * 64 + lane;
                    float* c = red + ((2 + mt) * 16 + r) * 64 + lane;
                    *a = hh == 0 ? dk[mt][r] : *a + dk[mt][r];
                    *c = hh == 0 ? dv[mt][r] : *c + dv[mt][r];
                }
        }
        __syncthreads();
    }
    if (head == 0 && key < N) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dk[mt][r] = red[(mt * 16 + r) * 64 + lane];
                dv[mt][r] = red[((2 + mt) * 16 + r) * 64 + lane];
            }
        uint16_t* dst = dqkv + ((int64_t)b * N + key) * ld + H * 64;
        at_store_row(dst, dk, 0.125f, h);
        at_store_row(dst + 64, dv, 1.0f, h);
    }
}

// d log-slope_h = slope_h * sum of the (batch, tile) partials, in index order (the parameter is log-slope:
// slope = exp(learned_logslopes), embeddings.py:59-82)
__global__ __launch_bounds__(512) void at_slope_reduce_kernel(const float* __restrict__ part, int per_head, const float* __restrict__ slopes,
                                                              float* __restrict__ dlogslopes, int H) {
    const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    if (h >= H) return;
    float s = 0.f;
    for (int k = l; k < per_head; k += 64) s += part[(int64_t)h * per_head + k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (l == 0) dlogslopes[h] = s * slopes[h];
}

static int at_kvrows(int N) { return ((N < kAtResKeys ? N : kAtResKeys) + 31) / 32 * 32; }

}  // namespace

extern "C" int32_t ispk_alibi_mqa_attn_train_bf16(const uint16_t* qkv, int64_t ld_qkv, const float* slopes, const int64_t* key_len,
                                                  uint16_t* o, int64_t ld_o, float* lse, int32_t B, int32_t N, int32_t H,
                                                  float dropout_p, uint64_t seed, ispk_stream_t stream) {
    ISPK_REQUIRE(qkv && slopes && o && lse, ISPK_E_NULL, "attn_train_bf16: null pointer");
    ISPK_REQUIRE(B >= 0 && N >= 1 && H >= 1 && H <= 6 && B <= 65535 && (int64_t)N * N < ((int64_t)1 << 31), ISPK_E_SHAPE,
                 "attn_train_bf16: bad shape B=%d N=%d H=%d (H <= 6: 2 H waves of 168 registers)", B, N, H);
    ISPK_REQUIRE(ld_qkv >= H * 64 + 128 && ld_o >= H * 64 && ld_qkv % 8 == 0 && ld_o % 4 == 0, ISPK_E_ALIGN,
                 "attn_train_bf16: ld_qkv %% 8, ld_o %% 4, strides cover the rows");
    ISPK_REQUIRE(ispk_aligned(qkv, 16) && ispk_aligned(o, 8), ISPK_E_ALIGN, "attn_train_bf16: qkv 16-byte, o 8-byte aligned");
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, ISPK_E_SHAPE, "attn_train_bf16: dropout_p must be in [0, 1)");
    if (B == 0) return 0;
    const int kvrows = at_kvrows(N);
    const size_t lds = (size_t)2 * kvrows * 128;
    const dim3 grid((N + 63) / 64, B), block(2 * H * 64);
    const uint32_t thresh = drop_thresh(dropout_p);
    const float inv_keep = 1.0f / (1.0f - dropout_p);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define ISPK_AT_FWD(MAXT_, DROP_)                                                                                              \
    do {                                                                                                                        \
        ISPK_RESERVE_LDS((&attn_train_fwd_bf16_kernel<MAXT_, DROP_>), lds, "attn_train_bf16");                                   \
        hipLaunchKernelGGL((attn_train_fwd_bf16_kernel<MAXT_, DROP_>), grid, block, lds, s, qkv, ld_qkv, slopes, key_len, o, ld_o, \
                           lse, N, H, kvrows, thresh, inv_keep, mix_seed(seed), ispk_seed_source());                                                \
    } while (0)
    if (thresh) ISPK_AT_FWD(768, true); else ISPK_AT_FWD(768, false);
#undef ISPK_AT_FWD
    return ispk_launch_status();
}

extern "C" int32_t ispk_alibi_mqa_attn_bwd_bf16(const uint16_t* qkv, int64_t ld_qkv, const uint16_t* o, const uint16_t* d_o,
                                                int64_t ld_o, const float* slopes, const int64_t* key_len, const float* lse,
                                                uint16_t* dqkv, float* dlogslopes, float* workspace, int64_t workspace_floats,
                                                int32_t B, int32_t N, int32_t H, float dropout_p, uint64_t seed,
                                                ispk_stream_t stream) {
    ISPK_REQUIRE(qkv && o && d_o && slopes && lse && dqkv && workspace, ISPK_E_NULL, "attn_bwd_bf16: null pointer");
    ISPK_REQUIRE(B >= 0 && N >= 1 && H >= 1 && H <= 6 && B <= 65535 && (int64_t)N * N < ((int64_t)1 << 31), ISPK_E_SHAPE,
                 "attn_bwd_bf16: bad shape B=%d N=%d H=%d (H <= 6: 2 H waves of 168 registers)", B, N, H);
    ISPK_REQUIRE(ld_qkv >= H * 64 + 128 && ld_o >= H * 64 && ld_qkv % 8 == 0 && ld_o % 8 == 0, ISPK_E_ALIGN,
                 "attn_bwd_bf16: ld_qkv %% 8, ld_o %% 8, strides cover the rows");
    ISPK_REQUIRE(ispk_aligned(qkv, 16) && ispk_aligned(o, 16) && ispk_aligned(d_o, 16) && ispk_aligned(dqkv, 8), ISPK_E_ALIGN,
                 "attn_bwd_bf16: qkv / o / d_o 16-byte, dqkv 8-byte aligned");
    ISPK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, ISPK_E_SHAPE, "attn_bwd_bf16: dropout_p must be in [0, 1)");
    const int tiles = (N + 63) / 64;
    const int64_t stat = (int64_t)B * H * N, need = stat + (int64_t)H * B * 2 * tiles;
    ISPK_REQUIRE(workspace_floats >= need, ISPK_E_SHAPE, "attn_bwd_bf16: workspace needs %lld floats", (long long)need);
    if (B == 0) return 0;
    float *delta = workspace, *spart = workspace + stat;
    const int kvrows = at_kvrows(N);
    const size_t lds_q = (size_t)2 * kvrows * 128;
    const size_t step_bytes = (size_t)2 * (H * 8192 + H * 256) + 16384, red_bytes = (size_t)2 * 4 * 16 * 64 * sizeof(float);
    const size_t lds_kv = step_bytes > red_bytes ? step_bytes : red_bytes;
    const dim3 grid(tiles, B), block(2 * H * 64);
    const uint32_t thresh = drop_thresh(dropout_p);
    const float inv_keep = 1.0f / (1.0f - dropout_p);
    const uint64_t sd = mix_seed(seed);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define ISPK_AT_BWD(MAXT_, DROP_)                                                                                              \
    do {                                                                                                                        \
        ISPK_RESERVE_LDS((&attn_bwd_dq_bf16_kernel<MAXT_, DROP_>), lds_q, "attn_bwd_bf16");                                      \
        hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<MAXT_, DROP_>), grid, block, lds_q, s, qkv, ld_qkv, o, d_o, ld_o, slopes,     \
                           key_len, lse, dqkv, delta, spart, N, H, kvrows, thresh, inv_keep, sd, ispk_seed_source());                               \
        ISPK_RESERVE_LDS((&attn_bwd_dkv_bf16_kernel<MAXT_, DROP_>), lds_kv, "attn_bwd_bf16");                                    \
        hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<MAXT_, DROP_>), grid, block, lds_kv, s, qkv, ld_qkv, d_o, ld_o, slopes,      \
                           key_len, lse, delta, dqkv, N, H, thresh, inv_keep, sd, ispk_seed_source());                                              \
    } while (0)
    if (thresh) ISPK_AT_BWD(768, true); else ISPK_AT_BWD(768, false);
#undef ISPK_AT_BWD
    if (dlogslopes)
        hipLaunchKernelGGL(at_slope_reduce_kernel, dim3(1), dim3(64 * H), 0, s, spart, B * 2 * tiles, slopes, dlogslopes, H);
    return ispk_launch_status();
}
