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
//   * the back-pointer of a cell is ONE bit, shifted into a per-lane history word (32 rows of the lane's column per word);
//     a word goes to LDS once per 32 rows: hist[row block][column], M*L/8 bytes (6.4 KB at M=512, L=100).
//   * logits rows are prefetched R rows ahead into registers through a buffer descriptor (per-lane column offset fixed,
//     row offset in an SGPR: no vector address arithmetic per row), two register sets swapping roles at compile time, so
//     the row step is DPP -> compare -> select -> add plus two instructions for the history bit: 19 instructions per row
//     at L <= 128 (round 1: 37, with a v_writelane pair + 4 wait states per row and chunk).
//   * backtrack: one history block (32 rows) per pass; lane c holds the word of column j0 - c, a row's decisions across
//     those columns are one ballot -> SGPR pair, and the serial walk is two scalar instructions per row (s_bitcmp1_b64 on
//     the current offset, s_addc_u32); the decisions taken are collected in a scalar mask from which every lane counts
//     its own row's column (popcount) - no per-row vector work.
//   * the 4 waves of the block then write the one-hot int16 rows (16-B stores), the path and the durations.
// 40 us at B=64, M=512, L=100 (DP 23, backtrack 8, outputs 9; round 1: 80 = 44 + 25 + 11).
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

__device__ __forceinline__ float dpp_ror1(float src) {
    // lane l <- lane l-1, lane 0 <- lane 63 (wave_ror:1): carries a chunk's last column to the next chunk's lane 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x13C, 0xf, 0xf, false));
}

// R DP rows from the prefetched logits `use`.  Per row and 64-column chunk the dependent chain is DPP shift -> compare ->
// select -> add; the decision bit is shifted into a per-lane history word (one word = kBlk = 32 rows of the lane's
// column, bit 31 - r <-> row r of the block), which goes to LDS as hist[block][column].
constexpr int kBlk = 32;

template <int NC, int R>
__device__ __forceinline__ void mas_rows(float (&q)[NC], const float (&use)[R][NC], uint32_t (&h)[NC], float qnan) {
    static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        float left[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) left[c] = dpp_shr1(q[c], c > 0 ? dpp_ror1(q[c - 1]) : qnan);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool diag = left[c] >= q[c];       // column 0: NaN >= x is false, no `j > 0` test
            h[c] = h[c] + h[c] + (diag ? 1u : 0u);   // shift the decision in (one add-with-carry)
            q[c] = use[r][c] + (diag ? left[c] : q[c]);
        }
    });
}

// One row of the backtrack: if bit d of the row's decision word is set the path moves one column left (d += 1); the
// decision is shifted into `taken` (rows are visited top down, so after 32 rows bit r belongs to row r of the block).
__device__ __forceinline__ void walk_step(int& d, uint32_t& taken, uint64_t word) {
    uint32_t t;
    asm volatile("s_bitcmp1_b64 %3, %0\n\ts_cselect_b32 %2, 1, 0\n\ts_addc_u32 %0, %0, 0\n\ts_lshl1_add_u32 %1, %1, %2"
                 : "+s"(d), "+s"(taken), "=&s"(t)
                 : "s"(word)
                 : "scc");
}

template <int NC>
__global__ __launch_bounds__(256) void mas_kernel(const float* __restrict__ logits, const int64_t* __restrict__ text_len,
                                                  const int64_t* __restrict__ mel_len, int16_t* __restrict__ attn_hard,
                                                  int64_t* __restrict__ dur, int16_t* __restrict__ path_out, int M_max,
                                                  int L_max, int64_t stride_b, int64_t stride_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nblk = (M_max + kBlk - 1) / kBlk;
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);                             // [nblk][NC * 64]
    int16_t* path = reinterpret_cast<int16_t*>(smem + (size_t)nblk * NC * 256);     // [M_max] (padded to 16 B)
    int* cnt = reinterpret_cast<int*>(smem + (size_t)nblk * NC * 256 + (((size_t)M_max * 2 + 15) & ~(size_t)15));

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
        // Logits rows through a buffer descriptor: a per-lane byte offset fixed for the whole kernel (the column, clamped
        // to m - 1: columns >= m never influence columns < m, any finite value will do) plus a wave-uniform row offset in
        // an SGPR - no vector address arithmetic per row.  Rows past the end re-read row n - 1 (clamped offset) and their
        // history bits are never looked at.
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lp), 0,
                                                            (int)((int64_t)M_max * stride_m * 4), 0x00020000);
        int voff[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) voff[c] = 4 * min(lane + 64 * c, m - 1);
        const int row_bytes = (int)(stride_m * 4), last_off = (n - 1) * row_bytes;
        // kRows rows are in registers ahead of the DP (32 / 16 / 8 by chunk count: 2 * kRows * NC VGPRs for the ping-pong
        // pair); a trip of the loop below covers 64 rows = two history blocks = an even number of sub-blocks, so the two
        // buffers swap roles at compile time and no register is ever copied.
        constexpr int kRows = rows_ahead<NC>(), kSub = kBlk / kRows;
        float bufa[kRows][NC], bufb[kRows][NC];
        auto load_rows = [&](float (&dst)[kRows][NC], int base) __attribute__((always_inline)) {
            int off = base * row_bytes;
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const int so = min(off, last_off);
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    dst[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[c], so, 0));
                off += row_bytes;
            }
        };
        const float qnan = __builtin_nanf("");
        load_rows(bufa, 1);
        uint32_t h[NC];
        for (int base = 1; base < n; base += 2 * kBlk) {
            static_for<0, 2 * kSub>([&](auto sc) {
                constexpr int sub = decltype(sc)::value, in_blk = sub % kSub;
                const int rows_base = base + sub * kRows;
                if (rows_base < n) {
                    if (sub % 2 == 0) load_rows(bufb, rows_base + kRows); else load_rows(bufa, rows_base + kRows);
                    if (in_blk == 0) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) h[c] = 0u;
                    }
                    if (sub % 2 == 0) mas_rows<NC, kRows>(q, bufa, h, qnan); else mas_rows<NC, kRows>(q, bufb, h, qnan);
                    // the block's word so far, already in its final position (rows not computed yet read as 0): a block
                    // that the utterance ends in is complete after whichever sub-block was its last
                    uint32_t* dst = hist + (size_t)((rows_base - 1) / kBlk) * NC * 64 + lane;
#pragma unroll
                    for (int c = 0; c < NC; ++c) dst[64 * c] = h[c] << (kRows * (kSub - 1 - in_blk));
                }
            });
        }
    }
    __syncthreads();

    if (wave == 0 && abl != 1) {
        // ---------------------------------------------------------------- backtrack, one history block (32 rows) per pass
        // Lane c holds the history word of column j0 - c (j0 = the path's column at the block's top row); within a block
        // the path moves left by at most 32, so lanes 0..32 cover it.  Row r's decisions across those columns are one
        // ballot -> an SGPR word; the serial walk itself is then scalar arithmetic only (d = how far left of j0).
        int j0 = m - 1;
        for (int top = n - 1; top >= 1;) {
            const int blk = (top - 1) / kBlk, base = 1 + blk * kBlk, rows = top - base + 1;   // rows base .. top
            const int col = j0 - lane;
            uint32_t w = (col >= 0 && lane <= kBlk) ? hist[(size_t)blk * NC * 64 + col] : 0u;
            if (rows < kBlk) w &= ~((1u << (kBlk - rows)) - 1u);      // rows above `top`: never computed / not on the path
            // serial walk, two scalar instructions on the dependent chain per row: test bit d of the row's word, add the
            // carry.  The decisions taken are collected in `taken` (bit r = row r moved left); lane r then counts the moves
            // of the rows above it to get its own column - no per-row vector work.
            int d = 0;
            uint32_t taken = 0u;
#pragma unroll
            for (int r0 = kBlk - 16; r0 >= 0; r0 -= 16) {                          // upper 16 rows of the block first
                uint64_t word[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) word[u] = __ballot(((w >> (kBlk - 1 - (r0 + u))) & 1u) != 0u);
#pragma unroll
                for (int u = 15; u >= 0; --u) walk_step(d, taken, word[u]);
            }
            const int myj = j0 - __builtin_popcountll((uint64_t)taken >> (lane + 1));   // lanes >= 32: shift clears it
            if (lane < rows) path[base + lane] = (int16_t)myj;
            j0 -= d;
            top = base - 1;
        }
        if (lane == 0) path[0] = (int16_t)j0;
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
        // (row, column) of a thread's 8-element group advance by a fixed step: one 32-bit division per thread up front
        // instead of a 64-bit one per group
        const uint32_t step_i = 2048u / (uint32_t)L_max, step_c = 2048u % (uint32_t)L_max;
        int gi = (int)((uint32_t)(tid * 8) / (uint32_t)L_max), gc = (int)((uint32_t)(tid * 8) % (uint32_t)L_max);
        for (int64_t e0 = (int64_t)tid * 8; e0 < total; e0 += 256 * 8) {
            int i = gi, col = gc;
            gi += (int)step_i;
            gc += (int)step_c;
            if (gc >= L_max) {
                gc -= L_max;
                ++gi;
            }
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
    const size_t lds = (size_t)((M_max + 31) / 32) * nc * 256 + (((size_t)M_max * 2 + 15) & ~(size_t)15) + (size_t)L_max * 4;
    ISPK_REQUIRE((int64_t)M_max * stride_m * 4 < ((int64_t)1 << 31), ISPK_E_SHAPE,
                 "mas: one utterance's logits (M_max * stride_m floats) must span less than 2 GiB");
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
