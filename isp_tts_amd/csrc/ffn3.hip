// Pre-norm feed-forward block of a transformer layer, decoder-sized batches, bf16 operands (third generation):
//     out = [mask] * ( x + gelu( LN(x) · W1ᵀ ) · W2ᵀ )        transformer.py:101-110, normalization.py:20-27, feedforward.py:33-40
// ONE kernel, the [rows, inner] hidden activations never leave the register file.
//
// What the first two generations measured: the four-wave kernel of round 1 (gemm.hip, one wave per
// SIMD) staged its weights through registers and ds_write (LDS-write-bound) and ran its GELU on packed fp32 (slow beside
// MFMAs): 114 us.  The eight-wave kernel of round 2 (ffn2.hip, two waves per SIMD, weights by LDS-DMA) splits both products
// between the two waves of a SIMD: the partial sums and the activations cross LDS, all eight waves meet at a barrier per
// 32-hidden chunk, and with everything removable removed one at a time (weight stream, GELU, operand reads, priorities, the
// distribution of the DMA instructions) it stays at 86 - 100 us: two in-order waves that share a matrix pipe spend their time
// waiting for each other.  This kernel goes back to ONE wave per SIMD - nothing to arbitrate, nothing to exchange - and keeps
// what the second generation learned:
//   * a workgroup = 128 rows = 4 waves x 32 rows; a wave keeps its 32 normalised rows as B-operand fragments (96 VGPRs) and
//     its 32 x 384 outputs as accumulators (192 registers), and walks the inner dimension in chunks of 32 hidden units:
//        phase A   S  = W1[chunk] · LN(x)ᵀ             24 MFMAs (32 hidden x 32 rows, K = 384)
//        GELU      P  = bf16(gelu(S)): the accumulator registers, packed pairwise, ARE the B operand of phase B
//                  (register 8 s + j of lane half h holds hidden 16 s + 8 (j >> 2) + 4 h + (j & 3); the W2 image carries
//                  its hidden columns in that order: ispk_ffn_pack_w2_bf16) - no LDS round trip, no exchange
//        phase B   Yᵀ += W2[:, chunk] · Pᵀ             24 MFMAs (12 feature tiles x 2 k-steps)
//   * weights stream through LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), one group
//     {W1 chunk it+1, W2 chunk it} per iteration into the buffer the previous iteration released, the 12 - 13 instructions of a
//     wave spread over the iteration's MFMA gaps (issued in one burst they queue on the CU's one address path);
//   * the GELU (Abramowitz-Stegun 7.1.27, 11 plain fp32 instructions per value, 16 values per lane and chunk) is cut into
//     LEVELS of 16 independent instructions and dealt over ALL 48 MFMA gaps of an iteration, about four per gap - the first
//     six levels under phase B of the iteration that produced S, the rest and the bf16 packing under phase A of the next one
//     (an MFMA holds the SIMD's issue port for 8 of its 32 cycles: four or five single-issue instructions fit beside it);
//   * operand fragments come through a ring of hand-counted asm reads (common.h), one read issued per gap, four in flight.
// Prologue (LayerNorm into a bf16 LDS tile, from which the fragments are taken) and epilogue (64 rows at a time through an
// fp32 LDS tile: + residual, mask, the next layer's row statistics, 16-byte stores) are the second generation's.
//
// STATUS (round 4, MI355X, 32,768 x 384 x 1536; tools/bench_ffn.py): correct (same bounds as ffn2 against float64, 2e-5 rms from
// ffn2's output), NOT faster - 111 - 114 us against ffn2's 102 - 104 - and therefore compiled into the EXPERIMENTS build only
// (libispk_exp.so; tools/bench_ffn.py binds the entry point itself; it is not part of include/ispk.h).  Stamps: main loop
// 2,900 cycles per 32-hidden chunk (ffn2: 3,340; the 48 MFMAs are 1,536), prologue + epilogue 81 k cycles (ffn2: 46 k).  What
// bounds it: ONE in-order wave issues an instruction every ~4 cycles, and a chunk is ~470 of them (184 GELU, 48 reads, 48
// waits, 13 DMA x 7, 48 MFMA that hold the port 8 cycles each): 2,000 cycles of issue before any stall - measured: without the
// GELU instructions 2,270, without the DMA instructions 2,700.  The second wave of a SIMD that ffn2 has is what issues beside
// a stalled or matrix-bound partner; what it costs ffn2 is the exchange.  The LayerNorm prologue and the row-wise epilogue (shuffle
// reductions, conversions) are vector-instruction work as well and take twice as long on four waves as on eight.
#include "common.h"

#ifdef ISPK_EXPERIMENTS

namespace {

constexpr int kD = 384, kHC = 32, kKS = kD / 16, kNT = kD / 32;
constexpr int kW1Row = kD * 2 + 16;              // W1 chunk rows in LDS: 768 B + one 16-byte pad: conflict-free fragment reads
constexpr int kW1Dma = 25;                       // DMA instructions (1 KB each) that cover the padded W1 image (25,088 B)
constexpr int kW1Bytes = kW1Dma * 1024;          // 25,600
constexpr int kW1Src = kHC * kD * 2;             // 24,576: W1 chunk [32 hidden][384] bf16 in memory
constexpr int kW2Bytes = kD * kHC * 2;           // 24,576: W2 chunk [384 features][32 hidden, fragment order] bf16
constexpr int kW2Dma = kW2Bytes / 1024;          // 24
constexpr int kGroup = kW1Dma + kW2Dma;          // 49 instructions per group
constexpr int kWbuf = kW1Bytes + kW2Bytes;       // 50,176 per buffer
constexpr int kXtOff = kWbuf;                    // prologue: LN(x) tile [128 rows][768 B] behind buffer 0
constexpr int kLdT = 388;                        // epilogue: fp32 tile [64 rows][388] at 0 (99,328 B)
constexpr int kLds = kXtOff + 128 * kD * 2;      // 148,480 B
static_assert(2 * kWbuf <= kLds && 64 * kLdT * 4 <= kLds, "LDS carve-up");
constexpr int kDmaSlots = 13;                    // instructions per wave and group: q = wave + 4 j (wave 0: j = 12 too)
constexpr int kRing = 8;                         // operand fragments in flight (4: 2 us slower)

struct Ffn3Params {
    const float* x;
    int64_t ldx;
    const float* gamma;
    const float* beta;
    float eps;
    const uint16_t* W1;    // [inner][384]
    const uint16_t* W2p;   // [inner / 32][384][32 in fragment order]  (ispk_ffn_pack_w2_bf16)
    const uint8_t* mask;
    float* out;
    int64_t ldo;
    int rows, inner;
    uint32_t flags;
    float* stats;
    float stats_eps;
    unsigned long long* stamps = nullptr;   // experiments build, ABL == 3: per wave [prologue, fill, main loop, epilogue, total]
};

typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef float f32x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {      // ONE v_cvt_pk_bf16_f32 for the pair
    f32x2_v v;
    v.x = lo; v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_v));
}

__device__ __forceinline__ float sgpr_const(float c) {     // coefficients in SGPRs: as literals every FMA is a two-dword instruction
    asm volatile("" : "+s"(c));
    return c;
}

// ABL (experiments build, tools/bench_ffn.py): 1 = no weight DMA inside the main loop (stale buffers: WRONG results, compute-
// bound timing); 2 = no GELU work in the gaps (WRONG results); 3 = s_memtime stamps.
template <int ABL>
__global__ __launch_bounds__(256, 1) void ffn3_bf16_kernel(Ffn3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    [[maybe_unused]] unsigned long long ts[5] = {0, 0, 0, 0, 0};
    [[maybe_unused]] unsigned long long t_prev = 0, t_first = 0;
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if constexpr (ABL == 3) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
            if (slot >= 0) ts[slot] += t - t_prev; else t_first = t;
            t_prev = t;
        }
    };
    stamp(-1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 128;
    const int nchunks = p.inner / kHC;
    const char* W1b = reinterpret_cast<const char*>(p.W1);
    const char* W2b = reinterpret_cast<const char*>(p.W2p);

    // ---- this lane's source offsets for the wave's DMA instructions q = wave + 4 j of a group's 49.  q < 25: the padded W1 image
    // - 16-byte slot t = 64 q + lane is (row t / 49, piece t % 49), piece 48 and rows past 31 are padding (they re-fetch a
    // neighbouring piece); q >= 25: the W2 image, slot -> (row, piece ^ ((row >> 2) & 3)).  The destination is lane-linear.
    // (W2: t = 64 (q - 25) + lane -> row 16 (q - 25) + (lane >> 2), whose swizzle (row >> 2) & 3 = (lane >> 4) & 3 does not depend on
    // q: ONE lane offset serves all of a wave's W2 instructions, the instruction adds 1024 (q - 25).)
    constexpr int kW1Slots = (kW1Dma + 3) / 4;      // 7: j with wave + 4 j < 25 for some wave
    uint32_t soff1[kW1Slots];
#pragma unroll
    for (int j = 0; j < kW1Slots; ++j) {
        const uint32_t q = wave + 4 * j;
        const uint32_t t = 64u * (q < (uint32_t)kW1Dma ? q : (uint32_t)kW1Dma - 1) + lane;
        uint32_t r = t / 49u, c = t - r * 49u;
        r = r < 32u ? r : 31u;
        c = c < 48u ? c : 47u;
        soff1[j] = r * 768u + 16u * c;
    }
    const uint32_t soff2 = (uint32_t)(lane >> 2) * 64u + 16u * ((uint32_t)(lane & 3) ^ ((uint32_t)(lane >> 4) & 3u));
    int64_t dma_o1 = 0, dma_o2 = 0;     // byte offsets of the chunks being fetched
    char* dma_base = smem;
    auto dma_begin = [&](int c1, int c2, int buf) __attribute__((always_inline)) {
        dma_o1 = (int64_t)c1 * kW1Src;
        dma_o2 = (int64_t)c2 * kW2Bytes;
        dma_base = smem + buf * kWbuf;
    };
    auto dma_one = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const int q = wave + 4 * j;               // wave-uniform
        if (j == kDmaSlots - 1 && q >= kGroup) return;            // only wave 0 has a thirteenth instruction
        const char* src;
        if constexpr (j < kW1Slots - 1) src = W1b + dma_o1 + soff1[j];                       // q <= 23 + 3: always W1?  (q < 25 for j <= 5)
        else if constexpr (j == kW1Slots - 1) src = q < kW1Dma ? W1b + dma_o1 + soff1[j] : W2b + dma_o2 + (q - kW1Dma) * 1024 + soff2;
        else src = W2b + dma_o2 + (q - kW1Dma) * 1024 + soff2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dma_base + q * 1024), 16, 0, 0);
    };
    // ---- prologue: LayerNorm of the wave's 32 rows (two rows per pass of 32 lanes x 3 float4), bf16 into the tile.  All 48 row
    // loads of a lane go out first, the weight group behind them (vector memory operations complete in order: the rows, which the
    // arithmetic waits for, must not queue behind 49 KB of weights), then the arithmetic.
    {
        f32x4 v[16][3];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int r = row0 + wave * 32 + 2 * i + h;
            r = r < p.rows ? r : p.rows - 1;            // rows past the end: a valid row, never stored
#pragma unroll
            for (int j = 0; j < 3; ++j) v[i][j] = *reinterpret_cast<const f32x4*>(p.x + (int64_t)r * p.ldx + 4 * (l31 + 32 * j));
        }
        f32x4 g4[3], b4[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            g4[j] = *reinterpret_cast<const f32x4*>(p.gamma + 4 * (l31 + 32 * j));
            b4[j] = *reinterpret_cast<const f32x4*>(p.beta + 4 * (l31 + 32 * j));
        }
        __builtin_amdgcn_sched_barrier(0);
        dma_begin(0, 0, 0);      // group 0 = {W1 chunk 0, (a W2 chunk nobody reads)} -> buffer 0, on its way during the LayerNorm
        static_for<0, kDmaSlots>([&](auto jc) { dma_one(jc); });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) s += (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            const float mean = s * (1.0f / kD);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[i][j][e] - mean;
                    q = fmaf(d, d, q);
                }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
            const float rstd = 1.0f / sqrtf(q * (1.0f / kD) + p.eps);
            const int rl = wave * 32 + 2 * i + h;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaf((v[i][j][e] - mean) * rstd, g4[j][e], b4[j][e]);
                uint2 pk;
                pk.x = pack2_bf16(y[0], y[1]);
                pk.y = pack2_bf16(y[2], y[3]);
                const int c16 = (l31 + 32 * j) >> 1;     // 16-byte chunk of the row; this lane owns its half (l31 & 1)
                *reinterpret_cast<uint2*>(smem + kXtOff + rl * 768 + 16 * (c16 ^ (rl & 15)) + 8 * (l31 & 1)) = pk;
            }
        }
    }
    __syncthreads();
    bf16x8 xf[kKS];   // B operands of phase A: LN(x)[row wave*32 + l31][all 384 features], k-step ks = 16 features
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks)
        xf[ks] = *reinterpret_cast<const bf16x8*>(smem + kXtOff + (wave * 32 + l31) * 768 + 16 * ((2 * ks + h) ^ (l31 & 15)));
    __syncthreads();   // the tile is dead: buffer 1 may be written from here on
    stamp(0);

    dma_begin(nchunks > 1 ? 1 : 0, 0, 1);      // group 1 = {W1 chunk 1, W2 chunk 0} -> buffer 1
    static_for<0, kDmaSlots>([&](auto jc) { dma_one(jc); });
    // group 0 has landed when at most group 1's instructions of this wave are outstanding (13 for wave 0, 12 for the others)
    if (wave == 0) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __syncthreads();

    const uint32_t lds0 = lds_addr(smem);
    const int psw = (l31 >> 2) & 3;   // swizzle of a 64-byte W2 row (row 32 nt + l31)
    const uint32_t w1a = lds0 + l31 * kW1Row + 16 * h;                                     // + buffer, + 32 ks
    const uint32_t w2a0 = lds0 + kW1Bytes + l31 * 64 + 16 * ((0 + h) ^ psw);               // + buffer, + 2048 nt  (k-step 0)
    const uint32_t w2a1 = lds0 + kW1Bytes + l31 * 64 + 16 * ((2 + h) ^ psw);               //                        (k-step 1)

    f32x16 acc1;                      // S of the chunk in flight
    f32x16 acc2[kNT];
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[nt][r] = 0.f;
    union Frag { uint32_t u[4]; bf16x8 f; };
    Frag pf[2];                       // P: the B operand of phase B (k-steps 0, 1)
    float gz[16], gq[16], ghv[16];      // GELU state of 16 values between its levels

    // GELU levels.  erf by Abramowitz-Stegun 7.1.27: erf(z) = 1 - (1 + a1 z + a2 z^2 + a3 z^3 + a4 z^4)^-4, z >= 0, |error| <= 5e-4,
    // so |gelu error| <= 2.5e-4 |x|: an eighth of the bf16 rounding step or less.  gelu(v) = 0.5 v + 0.5 |v| erf(|v| / sqrt 2).
    const float kRs2 = sgpr_const(0.70710678118654752440f), a4 = sgpr_const(0.078108f), a3 = sgpr_const(0.000972f),
                a2 = sgpr_const(0.230389f), a1 = sgpr_const(0.278393f);
    // part 1 (6 levels, reads S): n = 16 level + i
    auto gelu_a = [&](int n) __attribute__((always_inline)) {
        const int lv = n >> 4, i = n & 15;
        if (lv == 0) gz[i] = fabsf(acc1[i]) * kRs2;
        else if (lv == 1) ghv[i] = 0.5f * acc1[i];
        else if (lv == 2) gq[i] = fmaf(gz[i], a4, a3);
        else if (lv == 3) gq[i] = fmaf(gq[i], gz[i], a2);
        else if (lv == 4) gq[i] = fmaf(gq[i], gz[i], a1);
        else gq[i] = fmaf(gq[i], gz[i], 1.0f);
    };
    constexpr int kOpsA = 6 * 16;
    // part 2 (5 levels + 8 packs): n = 16 level + i; the packs are n = 80 + k
    auto gelu_b = [&](int n) __attribute__((always_inline)) {
        const int lv = n >> 4, i = n & 15;
        if (lv == 0) gq[i] = gq[i] * gq[i];
        else if (lv == 1) gq[i] = gq[i] * gq[i];
        else if (lv == 2) gq[i] = __builtin_amdgcn_rcpf(gq[i]);
        else if (lv == 3) gq[i] = fmaf(-fabsf(ghv[i]), gq[i], fabsf(ghv[i]));        // hx - hx r = 0.5 |v| erf   (hx = |0.5 v|: source modifiers)
        else if (lv == 4) gq[i] = ghv[i] + gq[i];
        else {
            const int k = n - 80;                                        // pair k: values 2k, 2k + 1 -> P fragment k >> 2, word k & 3
            pf[k >> 2].u[k & 3] = pack2_bf16(gq[2 * k], gq[2 * k + 1]);
        }
    };
    constexpr int kOpsB = 5 * 16 + 8;

    // ---- pipeline fill: S(0) = W1[chunk 0] · xfᵀ with plain reads, then the whole GELU of it (no overlap)
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(smem + (w1a - lds0) + 32 * ks);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[ks], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < kOpsA; ++n) gelu_a(n);
    }
    stamp(1);

    // ---- main loop, iteration it = 1 .. nchunks (group `it` = {W1 chunk it, W2 chunk it - 1} in buffer it & 1):
    //   phase A (stages 0 .. 23)   S(it) = W1[it] · xfᵀ   - at it == nchunks a re-run of the last chunk, never consumed -
    //                              with part 2 of GELU(S(it - 1)) in its gaps -> pf
    //   phase B (stages 24 .. 47)  acc2 += W2[it - 1] · pfᵀ, with part 1 of GELU(S(it)) in its gaps
    // and the DMA instructions of group it + 1 in every third gap.
#pragma unroll 1
    for (int it = 1; it <= nchunks; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of group `it`
        __syncthreads();                                   // all of group `it` has landed; every wave is done with buffer (it + 1) & 1
        if constexpr (ABL != 1) dma_begin(it + 1 < nchunks ? it + 1 : nchunks - 1, it < nchunks ? it : nchunks - 1, (it + 1) & 1);
        const uint32_t a1 = w1a + (it & 1) * kWbuf;
        const uint32_t b0 = w2a0 + (it & 1) * kWbuf, b1 = w2a1 + (it & 1) * kWbuf;
        bf16x8 ring[kRing];
        auto issue = [&](auto sc) __attribute__((always_inline)) {
            constexpr int st = decltype(sc)::value;
            if constexpr (st < kKS) {
                lds_read_b128_asm<32 * st>(ring[st % kRing], a1);
            } else {
                constexpr int j = st - kKS;
                if constexpr (j & 1) lds_read_b128_asm<2048 * (j >> 1)>(ring[st % kRing], b1);
                else lds_read_b128_asm<2048 * (j >> 1)>(ring[st % kRing], b0);
            }
        };
        static_for<0, kRing>(issue);
        static_for<0, 2 * kKS>([&](auto sc) {
            constexpr int st = decltype(sc)::value;
            constexpr int left = 2 * kKS - 1 - st;
            lds_wait<(left < kRing - 1 ? left : kRing - 1)>();      // fragment st is in; younger reads may be in flight
            __builtin_amdgcn_sched_barrier(0);
            // Phase A's chain accumulates in ARCH VGPRs - the GELU's vector instructions read S, and hipcc allocates every MFMA
            // result it selects itself to the accumulator half of the file (then copies S out in one block behind the last MFMA,
            // or, short of accumulator registers, evicts tiles of acc2 for it) - so these 24 are written out.  Hazards the
            // assembler does not see: MFMA -> MFMA on the SAME accumulator registers needs no wait state (the hardware
            // forwards); MFMA result -> vector-instruction read needs 11: the s_nop block in gap 24.
            if constexpr (st == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc1) : "v"(ring[st % kRing]), "v"(xf[st]));
            } else if constexpr (st < kKS) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(ring[st % kRing]), "v"(xf[st]));
            } else {
                constexpr int j = st - kKS;
                acc2[j >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[st % kRing], pf[j & 1].f, acc2[j >> 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap work
            if constexpr (st + kRing < 2 * kKS) issue(std::integral_constant<int, st + kRing>{});
            if constexpr (ABL != 1 && st % 3 == 2 && st / 3 < kDmaSlots) dma_one(std::integral_constant<int, st / 3>{});
            if constexpr (st == kKS) asm volatile("s_nop 7\n\ts_nop 4" ::: "memory");     // (S: MFMA write -> VALU read, see above)
            if constexpr (ABL != 2) {
                if constexpr (st < kKS) {                       // part 2 of the previous chunk's GELU: ops [st, st + 1) * kOpsB / 24
                    constexpr int lo = st * kOpsB / kKS, hi = (st + 1) * kOpsB / kKS;
                    static_for<lo, hi>([&](auto nc) { gelu_b(decltype(nc)::value); });
                } else if constexpr (st > kKS) {                // part 1 of this chunk's GELU over gaps 25 .. 47 (gap 24 is left to
                    constexpr int g = st - kKS - 1;             // the latency of phase A's last MFMA)
                    constexpr int lo = g * kOpsA / (kKS - 1), hi = (g + 1) * kOpsA / (kKS - 1);
                    static_for<lo, hi>([&](auto nc) { gelu_a(decltype(nc)::value); });
                }
            }
            // (the next stage's wait for its fragment stays BEHIND this gap's work: the wave issues in order, and a wait in
            // front of the GELU instructions would hold them back until the LDS has answered)
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    stamp(2);

    // ---- epilogue: 64 rows per pass through the fp32 tile; then whole rows: + x, mask, statistics, coalesced stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last iteration's group: fetched, never read)
    const bool mask_acc = p.flags & ISPK_EP_MASK_ACC, mask_out = p.flags & ISPK_EP_MASK_OUT;
    float* T = reinterpret_cast<float*>(smem);
    // the residual rows (and mask bytes) of BOTH passes - 16 rows per lane half, 192 registers: the operand fragments are dead -
    // are requested first: 196 KB in flight per workgroup, their latency passes while the tile is written and the barriers are crossed
    f32x4 xr[2][8][3];
    float mk[2][8];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = row0 + pass * 64 + wave * 16 + 2 * i + h;
            const int rc = r < p.rows ? r : p.rows - 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) xr[pass][i][j] = *reinterpret_cast<const f32x4*>(p.x + (int64_t)rc * p.ldx + 4 * (l31 + 32 * j));
            mk[pass][i] = (p.mask && (mask_acc || mask_out)) ? (p.mask[rc] ? 1.0f : 0.0f) : 1.0f;
        }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        if ((wave >> 1) == pass) {
            float* trow = T + ((wave & 1) * 32 + l31) * kLdT + 4 * h;
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
                for (int gq4 = 0; gq4 < 4; ++gq4) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc2[nt][4 * gq4 + e];
                    *reinterpret_cast<f32x4*>(trow + 32 * nt + 8 * gq4) = o;
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = wave * 16 + 2 * i + h;                // row of the tile
            const int r = row0 + pass * 64 + rl;
            const bool live = r < p.rows;
            f32x4 y[3];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = 4 * (l31 + 32 * j);
                f32x4 a = *reinterpret_cast<const f32x4*>(T + rl * kLdT + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = mask_acc ? a[e] * mk[pass][i] : a[e];
                    t += xr[pass][i][j][e];
                    a[e] = mask_out ? t * mk[pass][i] : t;
                }
                y[j] = a;
                s += (a[0] + a[1]) + (a[2] + a[3]);
            }
            if (live) {
#pragma unroll
                for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(p.out + (int64_t)r * p.ldo + 4 * (l31 + 32 * j)) = y[j];
            }
            if (p.stats) {
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                const float mean = s * (1.0f / kD);
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = y[j][e] - mean;
                        q = fmaf(d, d, q);
                    }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
                if (live && l31 == 0) {
                    p.stats[2 * (int64_t)r] = mean;
                    p.stats[2 * (int64_t)r + 1] = 1.0f / sqrtf(q * (1.0f / kD) + p.stats_eps);
                }
            }
        }
    }
    if constexpr (ABL == 3) {
        stamp(3);
        ts[4] = t_prev - t_first;
        if (lane == 0 && p.stamps) {
            unsigned long long* o = p.stamps + ((int64_t)blockIdx.x * 4 + wave) * 5;
            for (int i = 0; i < 5; ++i) o[i] = ts[i];
        }
    }
}

}  // namespace

extern "C" int32_t ispk_ffn_bf16_prenorm3(const float* x, int64_t ldx, const float* norm_gamma, const float* norm_beta,
                                          float norm_eps, const uint16_t* W1, const uint16_t* W2_packed, const uint8_t* mask,
                                          float* out, int64_t ldo, int32_t rows, int32_t dim, int32_t inner, uint32_t flags,
                                          float* row_stats, float stats_eps, ispk_stream_t stream) {
    ISPK_REQUIRE(x && norm_gamma && norm_beta && W1 && W2_packed && out, ISPK_E_NULL, "ffn_prenorm3: null pointer");
    ISPK_REQUIRE(dim == kD, ISPK_E_UNSUPPORTED, "ffn_prenorm3: dim %d (built for 384)", dim);
    ISPK_REQUIRE(rows >= 0 && inner >= 64 && inner % 32 == 0, ISPK_E_SHAPE, "ffn_prenorm3: bad shape rows=%d inner=%d", rows, inner);
    ISPK_REQUIRE((flags & ~(ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) == 0, ISPK_E_UNSUPPORTED, "ffn_prenorm3: unsupported flags");
    ISPK_REQUIRE(!((flags & (ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) && !mask), ISPK_E_NULL, "ffn_prenorm3: mask flag without mask");
    ISPK_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= dim && ldo >= dim && ispk_aligned(x, 16) && ispk_aligned(out, 16) &&
                     ispk_aligned(W1, 16) && ispk_aligned(W2_packed, 16) && ispk_aligned(norm_gamma, 16) &&
                     ispk_aligned(norm_beta, 16) && (!row_stats || ispk_aligned(row_stats, 8)),
                 ISPK_E_ALIGN, "ffn_prenorm3: 16-byte alignment / strides that are multiples of 4 required");
    if (rows == 0) return 0;
    Ffn3Params p{x, ldx, norm_gamma, norm_beta, norm_eps, W1, W2_packed, mask, out, ldo, rows, inner, flags, row_stats, stats_eps};
    const dim3 grid((rows + 127) / 128);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (const char* e = ispk_knob("ISPK_FFN3_ABLATE")) {
#define ISPK_FFN3_AB(N_)                                                               \
        if (atoi(e) == N_) {                                                           \
            ISPK_RESERVE_LDS((&ffn3_bf16_kernel<N_>), kLds, "ffn_prenorm3");          \
            hipLaunchKernelGGL(ffn3_bf16_kernel<N_>, grid, dim3(256), kLds, s, p);     \
            return ispk_launch_status();                                               \
        }
        ISPK_FFN3_AB(1) ISPK_FFN3_AB(2)
        if (atoi(e) == 3) {
            const char* sp = ispk_knob("ISPK_FFN3_STAMP");
            p.stamps = sp ? reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 16)) : nullptr;
        }
        ISPK_FFN3_AB(3)
#undef ISPK_FFN3_AB
    }
    ISPK_RESERVE_LDS((&ffn3_bf16_kernel<0>), kLds, "ffn_prenorm3");
    hipLaunchKernelGGL(ffn3_bf16_kernel<0>, grid, dim3(256), kLds, s, p);
    return ispk_launch_status();
}

#endif  // ISPK_EXPERIMENTS
