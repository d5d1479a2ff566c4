// Monotonic Alignment Search for gfx950: one wavefront runs the DP of one utterance.
//
// Semantics: /root/reference/tts/modules/aligner/mas.py:7-35 (oracle: oracle/mas_oracle.c).
//   Q[0][0] = lp[0][0], Q[0][j>0] = -inf;  Q[i][j] = lp[i][j] + max(Q[i-1][j-1], Q[i-1][j])
//   predecessor of (i, j) is j-1 iff j > 0 and Q[i-1][j-1] >= Q[i-1][j]  (ties -> diagonal); backtrack from (n-1, m-1).
//
// Mapping (not the reference's CUDA scheme of 256 threads + 2 block barriers per row + 3 global scratch arrays):
//   * lane l of wave 0 owns text columns l, l+64, ... (NC = ceil(L/64) registers), so a logits row is read with one
//     coalesced 256-B load per 64 columns and never re-read: 4 B/cell in, and nothing but the result goes back out.
//   * Q[i-1][j-1] comes from the neighbouring lane through a DPP wave shift (v_mov_b32_dpp wave_shr:1); the carry
//     between 64-column chunks is a second DPP (wave_ror:1 of the previous chunk supplies lane 0's value).  No LDS
//     or barrier inside the row loop.
//   * the back-pointer of a cell is ONE bit: per row and chunk the wave's ballot (64 bits) is stored in LDS
//     (M*NC*8 bytes: 8 KB at M=512, L=100; 69 KB at M=1723, L=300).
//   * logits rows are prefetched R rows ahead into registers, so the dependent chain per row is
//     DPP -> compare -> select -> add only.
//   * backtrack: 64 rows at a time, lane r holds the ballot words of row (top - r); the serial walk reads them with
//     v_readlane (SGPR chain), no LDS round trip per row.
//   * the 4 waves of the block then write the one-hot int16 rows (16-B stores), the path and the durations.
#include <stdlib.h>

#include "common.h"

namespace {

// logits rows prefetched ahead of the DP, per chunk count (register budget: kRowsAhead * NC * 2 VGPRs).  One wave has
// only these loads in flight, so the depth sets the memory-level parallelism: with 8 rows the DP ran at 146 ns/row
// (latency-bound); with 32 it is bound by the dependent VALU chain instead.
template <int NC> constexpr int rows_ahead() { return NC <= 2 ? 32 : (NC <= 4 ? 16 : 8); }

__device__ __forceinline__ float dpp_shr1(float src, float lane0_value) {
    // lane l <- lane l-1 ; lane 0 keeps `lane0_value` (bound_ctrl off: invalid source lanes keep `old`)
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lane0_value), __builtin_bit_cast(int, src), 0x138,
                                           0xf, 0xf, false));
}

// lane `lane` of dst <- the wave-uniform value src (v_writelane_b32; this clang has no builtin for it)
// lane LANE of (lo, hi) <- the wave-uniform 64-bit value (src_lo, src_hi).  v_writelane_b32 has no builtin in this
// clang; inside inline asm the compiler's hazard recogniser does not see it, and a v_writelane issued right behind the
// v_cmp that produced its SGPR operand reads the STALE register (observed: every row stored the previous row's word) -
// hence the 4 wait states in front.
template <int LANE>
__device__ __forceinline__ void writelane64(uint32_t& lo, uint32_t& hi, uint32_t src_lo, uint32_t src_hi) {
    asm("s_nop 3\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(lo), "+v"(hi) : "s"(src_lo), "s"(src_hi), "n"(LANE));
}

__device__ __forceinline__ float dpp_ror1(float src) {
    // lane l <- lane l-1, lane 0 <- lane 63 (wave_ror:1): carries a chunk's last column to the next chunk's lane 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x13C, 0xf, 0xf, false));
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

template <int NC>
__global__ __launch_bounds__(256) void mas_kernel(const float* __restrict__ logits, const int64_t* __restrict__ text_len,
                                                  const int64_t* __restrict__ mel_len, int16_t* __restrict__ attn_hard,
                                                  int64_t* __restrict__ dur, int16_t* __restrict__ path_out, int M_max,
                                                  int L_max, int64_t stride_b, int64_t stride_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* bp = reinterpret_cast<uint64_t*>(smem);                        // [M_max][NC]
    int16_t* path = reinterpret_cast<int16_t*>(smem + (size_t)M_max * NC * 8);  // [M_max] (padded to 16 B)
    int* cnt = reinterpret_cast<int*>(smem + (size_t)M_max * NC * 8 + (((size_t)M_max * 2 + 15) & ~(size_t)15));

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    int n = (int)mel_len[b];
    int m = (int)text_len[b];
    n = n < 1 ? 1 : (n > M_max ? M_max : n);  // host validates shapes; lengths are device data, so clamp here
    m = m < 1 ? 1 : (m > L_max ? L_max : m);
    const float* lp = logits + (int64_t)b * stride_b;
    const float ninf = -__builtin_huge_valf();

    for (int j = tid; j < L_max; j += 256) cnt[j] = 0;
#ifdef ISPK_EXPERIMENTS
    const int abl = (int)(stride_m >> 40);   // phase ablation code (tools/bench_mas.py) in the high stride bits
    stride_m &= ((int64_t)1 << 40) - 1;
#else
    constexpr int abl = 0;
#endif

    if (wave == 0) {
        // ---------------------------------------------------------------- forward DP
        float q[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) q[c] = (lane + 64 * c == 0) ? lp[0] : ninf;

        constexpr int kRowsAhead = rows_ahead<NC>();
        float cur[kRowsAhead][NC], nxt[kRowsAhead][NC];
        auto load_rows = [&](float (&dst)[kRowsAhead][NC], int base) {
#pragma unroll
            for (int r = 0; r < kRowsAhead; ++r) {
                int i = base + r;
                i = i < n ? i : n - 1;
                const float* row = lp + (int64_t)i * stride_m;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int col = lane + 64 * c;           // (columns >= m never influence columns < m; any finite
                    dst[r][c] = row[col < m ? col : m - 1];  //  value will do, and a clamped load needs no exec branch)
                }
            }
        };
        load_rows(cur, 1);
        // Row step, per 64-column chunk: DPP shift, compare, select, add - and nothing else on the dependent chain:
        //   * the compare's lane mask IS the row's back-pointer word; it is parked in lane r of two VGPRs with
        //     v_writelane and the wave stores a whole block of kRowsAhead rows to LDS at once (a per-row `if (lane == 0)`
        //     LDS store costs an exec save / restore and an LDS instruction on every row);
        //   * column 0 needs no `j > 0` test: its "left neighbour" is NaN, and NaN >= x is false;
        //   * no per-row `i < n` branch: rows past the end recompute the last row (clamped loads) and are not stored.
        const float qnan = __builtin_nanf("");
        for (int base = 1; base < n; base += kRowsAhead) {
            load_rows(nxt, base + kRowsAhead);
            uint32_t wlo[NC], whi[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) wlo[c] = whi[c] = 0u;
            static_for<0, kRowsAhead>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                float left[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) left[c] = dpp_shr1(q[c], c > 0 ? dpp_ror1(q[c - 1]) : qnan);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const bool diag = left[c] >= q[c];
                    const uint64_t word = __ballot(diag);
                    q[c] = cur[r][c] + (diag ? left[c] : q[c]);
                    writelane64<r>(wlo[c], whi[c], (uint32_t)word, (uint32_t)(word >> 32));
                }
            });
            if (lane < kRowsAhead && base + lane < n) {
#pragma unroll
                for (int c = 0; c < NC; ++c) bp[(size_t)(base + lane) * NC + c] = ((uint64_t)whi[c] << 32) | wlo[c];
            }
#pragma unroll
            for (int r = 0; r < kRowsAhead; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) cur[r][c] = nxt[r][c];
        }
    }
    __syncthreads();

    if (wave == 0 && abl != 1) {
        // ---------------------------------------------------------------- backtrack, 64 rows per pass
        int j = m - 1;
        for (int top = n - 1; top >= 0; top -= 64) {
            const int myrow = top - lane;
            const int c_hi = j >> 6;
            uint64_t w_hi = 0, w_lo = 0;
            if (myrow >= 1) {
                w_hi = bp[(size_t)myrow * NC + c_hi];
                if (c_hi > 0) w_lo = bp[(size_t)myrow * NC + c_hi - 1];
            }
            // 16 rows at a time: their ballot words go to SGPRs first (32 independent v_readlane pairs), so the serial
            // walk itself is scalar arithmetic only - with a readlane inside every step each row waited for a
            // VALU -> SGPR round trip.  Rows above the top of the utterance hold zero words and leave j unchanged.
            int myj = -1;
#pragma unroll 1
            for (int r0 = 0; r0 < 64; r0 += 16) {
                uint64_t hi[16], lo[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    hi[u] = readlane_u64(w_hi, r0 + u);
                    lo[u] = readlane_u64(w_lo, r0 + u);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (lane == r0 + u) myj = j;
                    const uint64_t w = (j >> 6) == c_hi ? hi[u] : lo[u];
                    j -= (int)((w >> (j & 63)) & 1);
                }
            }
            if (myrow >= 0) path[myrow] = (int16_t)myj;
        }
    }
    __syncthreads();

    if (abl == 1 || abl == 2) return;
    // -------------------------------------------------------------------- outputs (all 4 waves)
    for (int i = tid; i < n; i += 256) atomicAdd(&cnt[path[i]], 1);
    if (path_out) {
        int16_t* po = path_out + (int64_t)b * M_max;
        for (int i = tid; i < M_max; i += 256) po[i] = i < n ? path[i] : (int16_t)-1;
    }
    const int64_t total = (int64_t)M_max * L_max;
    int16_t* out = attn_hard + (int64_t)b * total;
    if ((total & 7) == 0 && (((uintptr_t)attn_hard) & 15) == 0) {
        for (int64_t e0 = (int64_t)tid * 8; e0 < total; e0 += 256 * 8) {
            int i = (int)(e0 / L_max);
            int col = (int)(e0 - (int64_t)i * L_max);
            int hot = i < n ? path[i] : -1;
            uint16_t v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v[t] = col == hot ? 1 : 0;
                if (++col == L_max) {
                    col = 0;
                    ++i;
                    hot = i < n ? path[i] : -1;
                }
            }
            uint4 pk;
            pk.x = v[0] | ((uint32_t)v[1] << 16);
            pk.y = v[2] | ((uint32_t)v[3] << 16);
            pk.z = v[4] | ((uint32_t)v[5] << 16);
            pk.w = v[6] | ((uint32_t)v[7] << 16);
            *reinterpret_cast<uint4*>(out + e0) = pk;
        }
    } else {
        for (int64_t e = tid; e < total; e += 256) {
            int i = (int)(e / L_max);
            int col = (int)(e - (int64_t)i * L_max);
            out[e] = (i < n && col == path[i]) ? 1 : 0;
        }
    }
    __syncthreads();
    if (dur) {
        int64_t* d = dur + (int64_t)b * L_max;
        // column sums of the one-hot rows; they add up to n.  The reference's fix-up (alignment.py:278-282: when an item's
        // durations do not sum to mel_len, the difference goes to column 0) is applied here too - it is non-zero only for a
        // mel_len outside [1, M_max], which the kernel clamped above
        const int64_t fix = mel_len[b] - (int64_t)n;
        for (int jx = tid; jx < L_max; jx += 256) d[jx] = cnt[jx] + (jx == 0 ? fix : 0);
    }
}

template <int NC>
int32_t launch(const float* logits, const int64_t* text_len, const int64_t* mel_len, int16_t* attn_hard, int64_t* dur,
               int16_t* path, int B, int M_max, int L_max, int64_t sb, int64_t sm, size_t lds, hipStream_t stream) {
    ISPK_RESERVE_LDS((&mas_kernel<NC>), lds, "mas");
#ifdef ISPK_EXPERIMENTS
    if (const char* e = ispk_knob("ISPK_MAS_ABLATE")) sm |= (int64_t)atoi(e) << 40;  // experiments only
#endif
    hipLaunchKernelGGL(mas_kernel<NC>, dim3(B), dim3(256), lds, stream, logits, text_len, mel_len, attn_hard, dur, path,
                       M_max, L_max, sb, sm);
    return ispk_launch_status();
}

}  // namespace

extern "C" int32_t ispk_mas_f32(const float* logits, const int64_t* text_len, const int64_t* mel_len,
                                int16_t* attn_hard, int64_t* dur, int16_t* path, int32_t B, int32_t M_max,
                                int32_t L_max, int64_t stride_b, int64_t stride_m, ispk_stream_t stream) {
    ISPK_REQUIRE(logits && text_len && mel_len && attn_hard, ISPK_E_NULL, "mas: null pointer argument");
    ISPK_REQUIRE(B >= 0 && M_max >= 1 && L_max >= 1, ISPK_E_SHAPE, "mas: bad shape B=%d M=%d L=%d", B, M_max, L_max);
    ISPK_REQUIRE(L_max <= 512 && M_max <= 4096, ISPK_E_SHAPE, "mas: L_max %d > 512 or M_max %d > 4096", L_max, M_max);
    ISPK_REQUIRE(stride_m >= L_max && stride_b >= (int64_t)M_max * 1, ISPK_E_SHAPE, "mas: bad strides");
    if (B == 0) return 0;
    const int nc = (L_max + 63) / 64;
    const size_t lds = (size_t)M_max * nc * 8 + (((size_t)M_max * 2 + 15) & ~(size_t)15) + (size_t)L_max * 4;
    ISPK_REQUIRE(lds <= 160 * 1024, ISPK_E_SHAPE, "mas: M_max=%d x L_max=%d needs %zu B of LDS (> 160 KiB)", M_max,
                 L_max, lds);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define ISPK_MAS_CASE(NC) \
    case NC:              \
        return launch<NC>(logits, text_len, mel_len, attn_hard, dur, path, B, M_max, L_max, stride_b, stride_m, lds, s);
    switch (nc) {
        ISPK_MAS_CASE(1)
        ISPK_MAS_CASE(2)
        ISPK_MAS_CASE(3)
        ISPK_MAS_CASE(4)
        ISPK_MAS_CASE(5)
        ISPK_MAS_CASE(6)
        ISPK_MAS_CASE(7)
        ISPK_MAS_CASE(8)
    }
#undef ISPK_MAS_CASE
    ISPK_FAIL(ISPK_E_SHAPE, "mas: unsupported chunk count %d", nc);
}
