// Pre-norm feed-forward block of a transformer layer, decoder-sized batches, bf16 operands (second generation):
//     out = [mask] * ( x + gelu( LN(x) · W1ᵀ ) · W2ᵀ )        transformer.py:101-110, normalization.py:20-27, feedforward.py:33-40
// ONE kernel; the [rows, inner] hidden activations never leave the CU.
//
// Why a second kernel (gemm.hip's ffn_bf16_kernel is the first): that one gives a wave 32 rows x ALL 384 output features,
// 192 accumulator + 96 operand registers = one wave per SIMD, and a wave's own VALU work (the GELU) does not overlap its
// own MFMAs: the matrix pipe was busy 35 % of the time.  Here a workgroup of 128 rows runs EIGHT waves, two per SIMD, so
// one wave's GELU / LDS traffic / waits run beside the other's MFMAs - and every weight fragment read from LDS still
// feeds a 32-row MFMA (v_mfma_f32_32x32x16_bf16: 1 KB of LDS per 32 matrix cycles per SIMD = half the LDS bandwidth;
// 16-row waves would need all of it).  The two waves of a SIMD share the same 32 rows ("row group" rg = wave & 3) and
// split the work of both products between them (half = wave >> 2):
//   product 1  S = W1[chunk] · LN(x)ᵀ  (32 hidden x 32 rows): each half sums over ITS HALF OF K (192 features, 12 MFMAs;
//              only that half of the normalised rows stays in registers: 48) and hands the partner the 8 partial sums per
//              lane that the partner will finish (fp32, through LDS: 32 B per lane);
//   finish     own 8 partial sums + the partner's, GELU, round to bf16, into the row group's P tile in LDS;
//   product 2  Yᵀ[192 features of this half] += W2[those features][chunk] · Pᵀ   (12 MFMAs, 96 accumulator registers).
// The three stages of a 32-hidden chunk run in consecutive iterations (software pipeline of depth 3), so ONE workgroup
// barrier per iteration covers every hand-off, and the two halves run the stages in a different order - while one is in
// its VALU stage the other is in a matrix stage (two waves of a SIMD that run the same program in lock-step would do
// their VALU work at the same time and leave the matrix pipe idle).
// Weights stream through LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers): per iteration one 48-KB group
// {W1 chunk it+1, W2 chunk it-1} into the buffer the previous iteration released, 6 DMA instructions per wave, a whole
// iteration to land.  Both images are lane-linear as the DMA writes them; bank conflicts are removed by XOR-swizzling the
// LDS images are what the DMA's per-lane SOURCE addresses make them (the destination is lane-linear): the W1 chunk gets
// rows of 768 + 16 bytes (the pad slots re-fetch a neighbouring piece) - conflict-free fragment reads AND one address
// register for all twelve of them (base + immediate; an XOR swizzle would need an address per k-step); W2 and P rows
// (64 B) are XOR-swizzled, 16-byte chunk ^ ((row >> 2) & 3): two addresses.  tools/lds_conflicts.py checks every
// fragment read: 4 LDS cycles, conflict-free.
// Prologue: the LayerNorm (two-pass fp32 statistics, 16 rows per wave) writes bf16 rows into an LDS tile from which every
// wave takes its fragments.  Epilogue: 64 rows at a time through an fp32 LDS tile, so that whole rows come back out:
// residual add (the fp32 rows are re-read: the file has no room to keep them), mask, the (mean, rstd) of the finished
// rows for the next layer's q/kv GEMM, 16-byte coalesced stores.
//
// Instances (template argument): 0 the block above; 20 the split-inner form for small batches (a slice of the inner dimension per
// workgroup, raw partial products, ispk_ffn_combine_ln_f32 adds them); 50 with the attention block's OUTPUT PROJECTION as the
// prologue - x1 = x + mask * (o Woᵀ) is formed in product 2's accumulators and never reaches memory (transformer.py:91,
// attention.py:172); 51 = 50 + the NEXT layer's attention_norm and q/kv projection as the epilogue (transformer.py:79-80,
// attention.py:63-64); 21 = 20 with the projection prologue.  Everything else is an ablation / stamp variant of the experiments build.
#include "common.h"

namespace {

constexpr int kD = 384, kHC = 32;
constexpr int kW1Row = kD * 2 + 16;              // W1 chunk rows in LDS: 768 B + one 16-byte pad (see the header comment)
constexpr int kW1Dma = 25;                       // DMA instructions (1 KB each) that cover the padded W1 image (25,088 B)
constexpr int kW1Bytes = kW1Dma * 1024;          // 25,600
constexpr int kW1Src = kHC * kD * 2;             // 24,576: W1 chunk [32 hidden][384] bf16 in memory
constexpr int kW2Bytes = kD * kHC * 2;           // 24,576: W2 chunk [384 features][32 hidden] bf16 (24 DMA instructions)
constexpr int kWbuf = kW1Bytes + kW2Bytes;       // 50,176 per buffer, two buffers
constexpr int kDmaPerWave = 7;                   // 49 instructions per group over 8 waves: wave w issues q = w, w + 8, ...
constexpr int kDmaMax = 10;                      // (uneven distributions: up to this many per wave)
constexpr int kPOff = 2 * kWbuf;                 // P tiles  [2][4 row groups][32 rows][64 B]
constexpr int kXcOff = kPOff + 2 * 4 * 2048;     // exchange [2][8 waves][2 planes][64 lanes][16 B]
constexpr int kLds = kXcOff + 2 * 8 * 2048;      // 149,504 B
constexpr int kXtOff = kWbuf;                    // prologue: LN(x) tile [128 rows][768 B] over buffer 1 + P + exchange
constexpr int kLdT = 388;                        // epilogue: fp32 tile [64 rows][388] at 0 (99,328 B)
static_assert(kXtOff + 128 * kD * 2 <= kLds && 64 * kLdT * 4 <= kLds, "LDS carve-up");

constexpr int kSplitMode = 20;   // template argument of the split-inner instance (ispk_ffn_bf16_prenorm2_split)
constexpr int kProjMode = 50;    // ... of the instance whose prologue is the attention block's output projection (ispk_attn_out_ffn_bf16)
constexpr int kSplitProjMode = 21;   // ... of the split-inner instance with the projection prologue (ispk_attn_out_ffn_split_bf16)
constexpr int kProjQkvMode = 51; // ... and whose epilogue is also the NEXT layer's attention_norm + q/kv projection (ispk_attn_out_ffn_qkv_bf16)
constexpr int kNq = 512;         // q/kv features of that mode: 6 heads x 64 + 128
constexpr int kQChunk = kNq * 16 * 2;            // one k-step of the q/kv weight: [512 features][16] bf16 = 16 KB (ispk_chunk_k16_bf16)
constexpr int kX2Off = 32 * 388 * 4;             // that epilogue: fp32 tile of 32 rows at 0, then the bf16 tile of LN_next(out) [128][768 B]
constexpr int kOutRow = kNq * 2 + 16;            // ... and the q/kv staging tile [128 rows][1024 + 16 B]

struct Ffn2Params {
    const float* x;
    int64_t ldx;
    const float* gamma;
    const float* beta;
    float eps;
    const uint16_t* W1;    // [inner][384]
    const uint16_t* W2c;   // [inner / 32][384][32]  (ispk_ffn_chunk_w2_bf16)
    const uint8_t* mask;
    float* out;
    int64_t ldo;
    int rows, inner;
    uint32_t flags;
    float* stats;
    float stats_eps;
    int chunk_count = 0;          // split mode (blockIdx.y = split): chunks per split; 0 = the whole inner dimension
    int64_t part_stride = 0;      // split mode: floats between the splits' partial outputs
    unsigned long long* stamps = nullptr;   // experiments build, ABL == 3: per-wave phase cycle sums [grid * 8][8]
    const uint16_t* o = nullptr;  // projection mode: attention output rows [rows][384] bf16 ...
    int64_t ld_o = 0;
    const uint16_t* WoC = nullptr;   // ... and to_out's weight as twelve chunks [384 / 32][384][32] (ispk_ffn_chunk_w2_bf16)
    const float* gamma2 = nullptr;   // q/kv mode: the next layer's attention_norm ...
    const float* beta2 = nullptr;
    float eps2 = 0.f;
    const uint16_t* WqC = nullptr;   // ... its [to_q; to_kv] weight as 24 k-step chunks [384 / 16][512][16] ...
    uint16_t* qkv = nullptr;         // ... and the q/kv rows it produces, bf16 [rows][512]
    int64_t ld_qkv = 0;
    void* ln_out = nullptr;          // projection mode, last layer of a stack: LN_final(out) rows (gamma2 / beta2 / eps2), bf16 or fp32 ...
    int64_t ld_ln = 0;
    int ln_bf16 = 0, ln_mask = 0;    // ... row-masked if ln_mask; p.out may then be NULL (the raw rows are not stored)
};
static_assert(kX2Off + 128 * kD * 2 <= kLds && 3 * kQChunk <= kX2Off && 128 * kOutRow <= kLds, "LDS carve-up of the q/kv epilogue");

// GELU(erf) for values that are rounded to bf16 right away.  erf by Abramowitz-Stegun 7.1.27:
// erf(z) = 1 - (1 + a1 z + a2 z^2 + a3 z^3 + a4 z^4)^-4, z >= 0, |error| <= 5e-4, so |gelu error| <= 2.5e-4 |x| - an eighth of
// the bf16 rounding step or less over the whole range.  11 plain fp32 instructions per value (scalar on purpose: packed
// fp32 instructions are slow beside MFMAs).  Eight values at a time, LEVEL BY LEVEL: written value by value hipcc
// interleaves only two of the dependent chains, and every instruction then waits for its predecessor's result (measured:
// 8 cycles per instruction instead of 4).  The coefficients sit in SGPRs (`opaque`): as 32-bit literals every FMA is a
// two-dword instruction.
__device__ __forceinline__ float opaque(float c) {
    asm volatile("" : "+s"(c));
    return c;
}
__device__ __forceinline__ void gelu8_bf16_grade(float (&v)[8]) {
    const float kRs2 = opaque(0.70710678118654752440f), a4 = opaque(0.078108f), a3 = opaque(0.000972f),
                a2 = opaque(0.230389f), a1 = opaque(0.278393f);
    float z[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = fabsf(v[i]) * kRs2;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = fmaf(z[i], a4, a3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = fmaf(q[i], z[i], a2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = fmaf(q[i], z[i], a1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = fmaf(q[i], z[i], 1.0f);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = q[i] * q[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = q[i] * q[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = __builtin_amdgcn_rcpf(q[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = 0.5f * fabsf(v[i]);        // hx
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = fmaf(-z[i], q[i], z[i]);    // hx - hx r
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaf(0.5f, v[i], q[i]);
}

// The same, with a hook after every level (experiment ABL 40: the wave's weight-DMA instructions go out between the levels)
template <typename Hook>
__device__ __forceinline__ void gelu8_bf16_grade_hooked(float (&v)[8], Hook&& hook) {
    const float kRs2 = opaque(0.70710678118654752440f), a4 = opaque(0.078108f), a3 = opaque(0.000972f),
                a2 = opaque(0.230389f), a1 = opaque(0.278393f);
    float z[8], q[8];
#define ISPK_LVL(n_, expr_)                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) { expr_; } \
    __builtin_amdgcn_sched_barrier(0);               \
    hook(std::integral_constant<int, n_>{});         \
    __builtin_amdgcn_sched_barrier(0);
    ISPK_LVL(0, z[i] = fabsf(v[i]) * kRs2)
    ISPK_LVL(1, q[i] = fmaf(z[i], a4, a3))
    ISPK_LVL(2, q[i] = fmaf(q[i], z[i], a2))
    ISPK_LVL(3, q[i] = fmaf(q[i], z[i], a1))
    ISPK_LVL(4, q[i] = fmaf(q[i], z[i], 1.0f))
    ISPK_LVL(5, q[i] = q[i] * q[i])
    ISPK_LVL(6, q[i] = q[i] * q[i])
    ISPK_LVL(7, q[i] = __builtin_amdgcn_rcpf(q[i]))
    ISPK_LVL(8, z[i] = 0.5f * fabsf(v[i]))
    ISPK_LVL(9, q[i] = fmaf(-z[i], q[i], z[i]))
#undef ISPK_LVL
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaf(0.5f, v[i], q[i]);
}

// Variant: the tanh form, gelu(x) ~ x / (1 + exp(-2 u)), u = sqrt(2/pi) (x + 0.044715 x^3): 7 instructions per value, two
// of them transcendental; max |error| vs the erf form about 5e-4 (at |x| ~ 2).
__device__ __forceinline__ void gelu8_tanh_form(float (&v)[8]) {
    const float c1 = opaque(-2.0f * 0.7978845608028654f * 1.4426950408889634f), c3 = opaque(-2.0f * 0.7978845608028654f * 0.044715f * 1.4426950408889634f);
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = v[i] * v[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = fmaf(t[i], c3, c1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = t[i] * v[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_exp2f(t[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = t[i] + 1.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_rcpf(t[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] * t[i];
}

// A global load the compiler's wait insertion does not see, and the counted wait that covers it (the loaded registers are
// operands of the wait: no use of them can be scheduled in front of it)
template <int OFF>
__device__ __forceinline__ void gload_b128_asm(u32x4& dst, const void* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait_tied(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// ABL (experiments build only, tools/bench_ffn.py): 1 = weight DMA only for the first two groups (the products then run on
// stale buffers: WRONG results, compute-bound timing); 2 = DMA and barriers only, no products / finish (streaming-bound
// timing); 3 = s_memtime stamps: per wave, cycles in [prologue, barrier waits, DMA issue, finish, product 1, product 2,
// epilogue, total]
template <int ABL>
__global__ __launch_bounds__(512, 2) void ffn2_bf16_kernel(Ffn2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // experiment switches (ABL 6 .. 9, 12): MFMA stages WITHOUT raised priority; tanh-form GELU; the DMA group issued by half 0 only
    // (matrix stages at raised priority: adopted - 102.8 -> 101.7 us, same results; ABL 6 = the kernel without it)
    constexpr bool kPrio = ABL != 6 && ABL != 16 && ABL != 17, kStaticPrio = ABL == 16 || ABL == 17, kTanh = ABL == 7 || ABL == 9, kDmaHalf0 = ABL == 8 || ABL == 9 || ABL == 12;
    [[maybe_unused]] unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] unsigned long long t_prev = 0, t_first = 0;
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if constexpr (ABL == 3 || ABL == 34) {
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
    const int rg = wave & 3, half = wave >> 2;
    const int l31 = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 128;
    // split mode (small batches, ispk_ffn_bf16_prenorm2_split): workgroup (row block, split) walks only its `chunk_count`
    // chunks of the inner dimension and leaves a raw fp32 partial product; both weight images are chunk-contiguous
    // (kHC rows of W1 = kHC * kD * 2 bytes = one W2 chunk), so a split is a pointer offset
    constexpr bool split_mode = ABL == kSplitMode || ABL == kSplitProjMode;      // (own instances: distinct kernel names in profiles)
    const int nchunks = split_mode ? p.chunk_count : p.inner / kHC;
    const int64_t wskip = split_mode ? (int64_t)blockIdx.y * p.chunk_count * (kHC * kD * 2) : 0;
    const char* W1b = reinterpret_cast<const char*>(p.W1) + wskip;
    const char* W2b = reinterpret_cast<const char*>(p.W2c) + wskip;

    // ---- this lane's part of the wave's DMA instructions (q = wave + 8 j of the group's 49): byte offset inside the chunk
    // in memory.  q < 25: the padded W1 image - 16-byte slot t = 64 q + lane is (row t / 49, piece t % 49), piece 48 and rows
    // past 31 are padding; q >= 25: the W2 image, slot -> (row, piece ^ ((row >> 2) & 3)).
    // Distribution of the 49 instructions over the waves.  Even: q = wave + 8 j (6 each, wave 0 a seventh).  Uneven (kH0 per
    // half-0 wave, the rest to half 1): the two halves run the stages in different orders and half 1 - the younger wave of
    // each SIMD, which loses every issue arbitration - is the one the barrier waits for; a DMA instruction costs its issuer
    // 70 - 115 cycles, so half 0 takes more of them.
    constexpr int kH0 = (ABL == 13 || ABL == 17 || ABL == 30 || ABL == 34) ? 8 : (ABL == 14 || ABL == 32) ? 9 : (ABL == 15 || ABL == 33) ? 10 : 0;        // 0 = even
    constexpr int kH1 = kH0 ? (48 - 4 * kH0) / 4 : 0;                              // 8 -> 4, 9 -> 3, 10 -> 2  (+ q = 48: wave 0)
    constexpr int kPerWave = kH0 ? kH0 + 1 : kDmaPerWave;
    auto q_of = [&](int j) __attribute__((always_inline)) -> int {                 // instruction j of this wave; -1 = none
        if constexpr (kH0 == 0) {
            const int q = wave + 8 * j;
            return q < kW1Dma + 24 ? q : -1;
        } else {
            if (half == 0) return j < kH0 ? wave + 4 * j : (j == kH0 && wave == 0 ? 48 : -1);
            return j < kH1 ? 4 * kH0 + (wave - 4) + 4 * j : -1;
        }
    };
    uint32_t soff[kPerWave];
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) {
        const int qq = q_of(j);
        const uint32_t q = qq < 0 ? 0u : (uint32_t)qq;
        if (q < (uint32_t)kW1Dma) {
            const uint32_t t = 64u * q + lane;
            uint32_t r = t / 49u, c = t - r * 49u;
            r = r < 32u ? r : 31u;
            c = c < 48u ? c : 47u;
            soff[j] = r * 768u + 16u * c;
        } else {
            const uint32_t t = 64u * (q - kW1Dma) + lane;
            const uint32_t r = t >> 2, c = t & 3u;
            soff[j] = r * 64u + 16u * (c ^ ((r >> 2) & 3u));
        }
    }
    // Every instruction's source base / chunk stride is wave-uniform and fixed (W1 or W2 image): no branch per instruction.
    // The caller clamps the chunk indices, so a group is always issued whole - at the ends of the pipeline a few pieces
    // are fetched again into a buffer nobody reads (cheaper than a dozen scalar branches per iteration).
    // One DMA instruction occupies the CU's address path for 16 cycles; eight waves issuing their six or seven right
    // behind the barrier queue up for 49 x 16 cycles (stamped: 440 - 700 cycles per wave and iteration).  Issuing them one
    // at a time from inside the matrix stages (dma_one<J>, experiment ABL 5) moved that wait into the stages and changed
    // nothing in total, so the group goes out at the barrier.
    int64_t dma_o1 = 0, dma_o2 = 0;     // byte offsets of the chunks being fetched this iteration
    char* dma_base = smem;
    auto dma_begin = [&](int c1, int c2, int buf) __attribute__((always_inline)) {
        dma_o1 = (int64_t)c1 * kW1Src;
        dma_o2 = (int64_t)c2 * kW2Bytes;
        dma_base = smem + buf * kWbuf;
    };
    auto dma_one = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < kPerWave) {
            const int q = q_of(j);                    // wave-uniform
            if (q < 0) return;
            if constexpr (ABL == 36) { if (q >= kW1Dma && dma_o1 > kW1Src) return; }      // experiment: half the bytes
            const char* src = (q < kW1Dma ? W1b + dma_o1 : W2b + dma_o2) + soff[j];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dma_base + q * 1024), 16, 0, 0);
        }
    };
    auto issue = [&](int c1, int c2, int buf) __attribute__((always_inline)) {     // a whole group at once
        dma_begin(c1, c2, buf);
        static_for<0, kPerWave>([&](auto jc) { dma_one(jc); });
    };
    bf16x8 xf[12];   // B operands of product 1: LN(x)[row rg*32 + l31][this half's 192 features], k-step ks = 16 features
    f32x16 acc2[6];
    float keep[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) keep[i] = 0.f;

    // Operand fragments come through small register rings of hand-counted asm reads (common.h: lds_read_b128_asm /
    // lds_wait): hipcc sinks every plain ds_read next to its MFMA and waits for it there - at most two in flight, an LDS
    // round trip exposed per pair of MFMAs.  A stage's first fragments are requested BEFORE the stage in front of it (both
    // weight images and the P tile are complete at the iteration's barrier), so each matrix stage starts on landed data.
    const int psw = (l31 >> 2) & 3;   // swizzle of a 64-byte row (P tile row l31; W2 row 192*half + 32*nt + l31)
    const uint32_t lds0 = lds_addr(smem);
    const uint32_t w1a = lds0 + l31 * kW1Row + 16 * (24 * half + h);                        // + buffer, + 32 ks
    const uint32_t w2a0 = lds0 + kW1Bytes + (192 * half + l31) * 64 + 16 * ((0 + h) ^ psw);  // + buffer, + 2048 nt  (k-step 0)
    const uint32_t w2a1 = lds0 + kW1Bytes + (192 * half + l31) * 64 + 16 * ((2 + h) ^ psw);  //                        (k-step 1)
    const uint32_t pa0 = lds0 + kPOff + rg * 2048 + l31 * 64 + 16 * ((0 + h) ^ psw);         // + 8192 parity
    const uint32_t pa1 = lds0 + kPOff + rg * 2048 + l31 * 64 + 16 * ((2 + h) ^ psw);
    constexpr int kRing = 4;   // operand fragments in flight per matrix stage (6: 246 VGPRs, measured 2 % slower)
    bf16x8 r1[kRing];          // product 1 ring: W1 fragments
    bf16x8 r2[kRing], pb[2];   // product 2 ring: W2 fragments; the two P fragments

    auto prefetch1 = [&](int it) __attribute__((always_inline)) {     // first 4 W1 fragments of chunk `it`
        const uint32_t a = w1a + (it & 1) * kWbuf;
        if constexpr (ABL != 11)
            static_for<0, kRing>([&](auto kc) { lds_read_b128_asm<32 * decltype(kc)::value>(r1[decltype(kc)::value], a); });
    };
    auto product1 = [&](int it, bool dma) __attribute__((always_inline)) {   // S = W1[chunk it][:, this half of K] · xfᵀ ; send 8, keep 8
        const uint32_t a = w1a + (it & 1) * kWbuf;
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        if constexpr (kPrio) __builtin_amdgcn_s_setprio(1);
        static_for<0, 12>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            if constexpr (ABL != 11) lds_wait<(11 - ks) < kRing - 1 ? (11 - ks) : kRing - 1>();   // fragment ks is in; younger reads may be in flight
            __builtin_amdgcn_sched_barrier(0);
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1[ks % kRing], xf[ks], S, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ks + kRing < 12 && ABL != 11) lds_read_b128_asm<32 * (ks + kRing)>(r1[ks % kRing], a);
            if constexpr ((ks & 3) == 1) {
                if (dma) dma_one(std::integral_constant<int, ks / 4>{});           // DMA instructions 0, 1, 2
            }
        });
        if constexpr (kPrio) __builtin_amdgcn_s_setprio(0);
        // S[r] = partial H[hidden (r & 3) + 8 (r >> 2) + 4 h][row l31]; half 0 finishes r < 8, half 1 finishes r >= 8
        f32x4* xc = reinterpret_cast<f32x4*>(smem + kXcOff + ((it & 1) * 8 + wave) * 2048 + lane * 16);   // two 1-KB planes
        f32x4 s0, s1v;
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] = S[8 + i]; s1v[i] = S[12 + i]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) keep[i] = S[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] = S[i]; s1v[i] = S[4 + i]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) keep[i] = S[8 + i];
        }
        xc[0] = s0;
        xc[64] = s1v;
    };
    constexpr bool kDmaInFinish = ABL == 40;
    auto finish = [&](int c, bool dma_here = false) __attribute__((always_inline)) {         // chunk c: own + partner's partial sums -> GELU -> bf16 -> P tile
        const f32x4* pc = reinterpret_cast<const f32x4*>(smem + kXcOff + ((c & 1) * 8 + (wave ^ 4)) * 2048 + lane * 16);
        const f32x4 a0 = pc[0], a1 = pc[64];
        float g[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g[i] = keep[i] + a0[i];
            g[4 + i] = keep[4 + i] + a1[i];
        }
        if constexpr (kDmaInFinish) {
            // this wave's DMA instructions of the iteration's group between the GELU's levels: ~60 cycles apart instead of a burst
            // of 49 from eight waves at the barrier, which queues on the CU's one address path (stamped: 70 - 115 cycles each)
            gelu8_bf16_grade_hooked(g, [&](auto lc) {
                constexpr int lv = decltype(lc)::value;
                if constexpr (lv < kPerWave) {
                    if (dma_here) dma_one(std::integral_constant<int, lv>{});
                }
            });
        } else if constexpr (kTanh) gelu8_tanh_form(g); else gelu8_bf16_grade(g);
        char* pt = smem + kPOff + ((c & 1) * 4 + rg) * 2048 + l31 * 64 + 8 * h;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            uint2 pk;
            pk.x = pack_bf16(g[4 * gq], g[4 * gq + 1]);
            pk.y = pack_bf16(g[4 * gq + 2], g[4 * gq + 3]);
            *reinterpret_cast<uint2*>(pt + 16 * ((2 * half + gq) ^ psw)) = pk;
        }
    };
    // product 2 reads, in order: P k-step 0, P k-step 1, then W2 fragment j = 2 nt + ks for j = 0 .. 11
    auto w2read = [&](auto jc, uint32_t b0, uint32_t b1) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j & 1) lds_read_b128_asm<2048 * (j >> 1)>(r2[j % kRing], b1);
        else lds_read_b128_asm<2048 * (j >> 1)>(r2[j % kRing], b0);
    };
    auto prefetch2 = [&](int it) __attribute__((always_inline)) {     // the P fragments and the first 4 W2 fragments
        const uint32_t b0 = w2a0 + (it & 1) * kWbuf, b1 = w2a1 + (it & 1) * kWbuf;
        if constexpr (ABL != 11) {
            lds_read_b128_asm<0>(pb[0], pa0 + (it & 1) * 8192);
            lds_read_b128_asm<0>(pb[1], pa1 + (it & 1) * 8192);
            static_for<0, kRing>([&](auto jc) { w2read(jc, b0, b1); });
        }
    };
    auto product2_at = [&](uint32_t b0, uint32_t b1, bool dma) __attribute__((always_inline)) {   // (b0 / b1: the W2 image's k-step 0 / 1 addresses)
        if constexpr (kPrio) __builtin_amdgcn_s_setprio(1);
        static_for<0, 12>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (ABL != 11) lds_wait<(11 - j) < kRing - 1 ? (11 - j) : kRing - 1>();
            __builtin_amdgcn_sched_barrier(0);
            acc2[j >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r2[j % kRing], pb[j & 1], acc2[j >> 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j + kRing < 12 && ABL != 11) w2read(std::integral_constant<int, j + kRing>{}, b0, b1);
            if constexpr ((j & 3) == 1) {
                if (dma) dma_one(std::integral_constant<int, 3 + j / 4>{});        // DMA instructions 3, 4, 5
            }
            if constexpr (j == 10) {
                if (dma) dma_one(std::integral_constant<int, 6>{});                // (wave 0 only)
            }
        });
        if constexpr (kPrio) __builtin_amdgcn_s_setprio(0);
    };
    auto product2 = [&](int it, bool dma) __attribute__((always_inline)) {   // acc2 += W2[this half's 192 features][chunk it - 2] · Pᵀ
        product2_at(w2a0 + (it & 1) * kWbuf, w2a1 + (it & 1) * kWbuf, dma);
    };

    // ---- prologue (a lambda: projection mode runs it INSIDE the two branches that select the main loop's stage order - with the
    // accumulators live across that branch hipcc has to agree on one register assignment for both loop copies and spills 76 VGPRs)
    constexpr bool qkv_mode = ABL == kProjQkvMode;
    constexpr bool proj_mode = ABL == kProjMode || qkv_mode || ABL == kSplitProjMode;
    auto prologue = [&]() __attribute__((always_inline)) {
    if constexpr (proj_mode) {
        // Projection mode: this row block's residual rows start in the accumulators of product 2 and the attention block's
        // output projection is twelve more "product 2" steps on top of them (to_out's weight cut into [384][32] chunks like W2,
        // the attention output rows as B operands straight from global memory into registers):
        //     x1 = x + [mask] * (o · Woᵀ)                              attention.py:172, transformer.py:91
        // x1 never exists in memory: its LayerNorm is taken from the accumulators (a row lives in four lanes: l31 and l31 + 32 of
        // the two waves of a SIMD), and the feed-forward block then accumulates onto it - no residual read in the epilogue.
        const char* WoB = reinterpret_cast<const char*>(p.WoC);
        // Chunk buffers of the projection: the W2 areas of both weight buffers and buffer 1's W1 area - three, so that chunk c + 2
        // is on its way while chunk c is multiplied (a DMA's latency is longer than one 12-MFMA step)
        constexpr int kPjBuf[3] = {kW1Bytes, kWbuf + kW1Bytes, kWbuf};
        auto dma_w1_0 = [&](auto jc) __attribute__((always_inline)) {           // this wave's part of W1 chunk 0 -> buffer 0
            constexpr int j = decltype(jc)::value;
            const int q = q_of(j);                    // wave-uniform
            if (q < 0 || q >= kW1Dma) return;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W1b + soff[j]),
                                             (__attribute__((address_space(3))) void*)(smem + q * 1024), 16, 0, 0);
        };
        // ... of a Wo chunk (24 instructions of 1 KB): wave w issues pieces w, w + 8, w + 16 - three each, no branch (straight-line
        // code lets the compiler count the outstanding loads exactly; behind a wave-uniform branch it waits for all of them)
        uint32_t swo[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint32_t t = 64u * (wave + 8 * j) + lane, r = t >> 2, c = t & 3u;
            swo[j] = r * 64u + 16u * (c ^ ((r >> 2) & 3u));
        }
        auto dma_wo = [&](int c, int boff) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(WoB + c * kW2Bytes + swo[j]),
                                                 (__attribute__((address_space(3))) void*)(smem + boff + (wave + 8 * j) * 1024), 16, 0, 0);
        };
        int rr = row0 + rg * 32 + l31;
        rr = rr < p.rows ? rr : p.rows - 1;           // rows past the end: a valid row, never stored
        // attention output, row l31 of the row group, as B operands: the two k-steps of a chunk are requested right behind the
        // chunk's weight DMA (five vector-memory instructions per wave and chunk: the waits below count them)
        const uint16_t* orow = p.o + (int64_t)rr * p.ld_o + 8 * h;
        const uint32_t row_on = (!(p.flags & ISPK_EP_MASK_ACC) || p.mask[rr]) ? 0xffffffffu : 0u;
        u32x4 af[3][2];
        auto fetch = [&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            dma_wo(c, kPjBuf[c % 3]);
            // (asm: hipcc's own wait in front of the first use of a plain load is vmcnt(0) once LDS-DMA instructions are in
            // flight - it would drain the chunks requested ahead; these are covered by the counted waits below)
            gload_b128_asm<64 * c>(af[c % 3][0], orow);
            gload_b128_asm<64 * c + 32>(af[c % 3][1], orow);
        };
        static_for<0, kPerWave>([&](auto jc) { dma_w1_0(jc); });      // (its W1 area is not touched before the main loop)
        {
            const float* xr = p.x + (int64_t)rr * p.ldx + 192 * half + 4 * h;
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 32 * nt + 8 * gq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[nt][4 * gq + e] = v[e];
                }
        }
        fetch(std::integral_constant<int, 0>{});
        fetch(std::integral_constant<int, 1>{});
        static_for<0, 12>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            // chunk c (everything older than the five instructions of chunk c + 1) has landed for this wave ...
            // (the fragments are operands of the wait so that no use of them can be scheduled in front of it)
            vm_wait_tied<(c < 11 ? 5 : 0)>(af[c % 3][0], af[c % 3][1]);
            __builtin_amdgcn_s_barrier();                        // ... and for the others, who are also done with chunk c - 1
            asm volatile("" ::: "memory");
            if constexpr (c + 2 < 12) fetch(std::integral_constant<int, c + 2>{});      // into the buffer chunk c - 1 has left
            pb[0] = __builtin_bit_cast(bf16x8, af[c % 3][0] & row_on);      // (masked rows contribute nothing: attention.py:172)
            pb[1] = __builtin_bit_cast(bf16x8, af[c % 3][1] & row_on);
            const uint32_t b0 = w2a0 - kW1Bytes + kPjBuf[c % 3], b1 = w2a1 - kW1Bytes + kPjBuf[c % 3];
            static_for<0, kRing>([&](auto jc) { w2read(jc, b0, b1); });
            product2_at(b0, b1, false);
        });
        __syncthreads();     // (everyone is done with the last chunks: the sums below and the tile go over their buffers)
        // LayerNorm of x1 from the accumulators; the partner wave's sums come through buffer 0's W2 area
        float* sc = reinterpret_cast<float*>(smem + kW1Bytes);
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc2[nt][r];
        s += __shfl_xor(s, 32, 64);
        sc[wave * 64 + lane] = s;
        __syncthreads();
        const float mean = (s + sc[(wave ^ 4) * 64 + lane]) * (1.0f / kD);
        float q = 0.f;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc2[nt][r] - mean;
                q = fmaf(d, d, q);
            }
        q += __shfl_xor(q, 32, 64);
        sc[512 + wave * 64 + lane] = q;
        __syncthreads();
        const float rstd = 1.0f / sqrtf((q + sc[512 + (wave ^ 4) * 64 + lane]) * (1.0f / kD) + p.eps);
        // bf16 rows into the tile (over buffer 1, the P tiles and the exchange area - all idle; not over the sums above)
        const int rl = rg * 32 + l31;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int f0 = 192 * half + 32 * nt + 8 * gq + 4 * h;
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma + f0), b4 = *reinterpret_cast<const f32x4*>(p.beta + f0);
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaf((acc2[nt][4 * gq + e] - mean) * rstd, g4[e], b4[e]);
                uint2 pk;
                pk.x = pack_bf16(y[0], y[1]);
                pk.y = pack_bf16(y[2], y[3]);
                *reinterpret_cast<uint2*>(smem + kXtOff + rl * 768 + 16 * ((f0 >> 3) ^ (rl & 15)) + 8 * h) = pk;
            }
        if constexpr (split_mode) {
            // every split needs x1 for its LayerNorm, but only split 0's partial product carries it into the combine pass
            if (blockIdx.y != 0) {
#pragma unroll
                for (int nt = 0; nt < 6; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[nt][r] = 0.f;
            }
        }
    } else {
        issue(0, 0, 0);   // W1 chunk 0 -> buffer 0 (and a W2 chunk nobody reads), on its way during the LayerNorm (the tile below
                          // does not touch buffer 0)

        // ---- prologue: LayerNorm of 16 rows per wave (two rows at a time: 32 lanes x 3 float4 cover a row), bf16 into the tile
        {
            const int rbase = rg * 32 + half * 16;
            f32x4 g4[3], b4[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                g4[j] = *reinterpret_cast<const f32x4*>(p.gamma + 4 * (l31 + 32 * j));
                b4[j] = *reinterpret_cast<const f32x4*>(p.beta + 4 * (l31 + 32 * j));
            }
            f32x4 v[8][3];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int r = row0 + rbase + 2 * i + h;
                r = r < p.rows ? r : p.rows - 1;            // rows past the end: a valid row, never stored
#pragma unroll
                for (int j = 0; j < 3; ++j) v[i][j] = *reinterpret_cast<const f32x4*>(p.x + (int64_t)r * p.ldx + 4 * (l31 + 32 * j));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
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
                const int rl = rbase + 2 * i + h;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = fmaf((v[i][j][e] - mean) * rstd, g4[j][e], b4[j][e]);
                    uint2 pk;
                    pk.x = pack_bf16(y[0], y[1]);
                    pk.y = pack_bf16(y[2], y[3]);
                    const int c16 = (l31 + 32 * j) >> 1;     // 16-byte chunk of the row; this lane owns its half (l31 & 1)
                    *reinterpret_cast<uint2*>(smem + kXtOff + rl * 768 + 16 * (c16 ^ (rl & 15)) + 8 * (l31 & 1)) = pk;
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < 6; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[nt][r] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 12; ++ks)
        xf[ks] = *reinterpret_cast<const bf16x8*>(smem + kXtOff + (rg * 32 + l31) * 768 + 16 * ((24 * half + 2 * ks + h) ^ (l31 & 15)));
    __syncthreads();   // the tile is dead: buffer 1, the P tiles and the exchange area may be written from here on
    stamp(0);
    };
    if constexpr (!proj_mode) prologue();

    // ---- main loop.  Iteration `it`: product 1 of chunk it, finish of chunk it-1, product 2 of chunk it-2.
    // Stage order: finish, product 1, product 2 for half 0; product 2, finish, product 1 for half 1 (finish always
    // precedes product 1, which overwrites the partial sums it consumes): whenever one wave of a SIMD is in its VALU stage
    // the other is in a matrix stage.  The whole loop exists twice, once per order, selected ONCE: a per-iteration choice
    // (a branch or a rotation loop around the stages) turns the 96 accumulators into loop-carried phi copies - twice the
    // registers, spills, and 96 moves per iteration.
    auto main_loop = [&](auto order) __attribute__((always_inline)) {
        constexpr int kOrder = decltype(order)::value;
#pragma unroll 1
        for (int it = 0; it <= nchunks + 1; ++it) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA instructions of the previous iteration
            __syncthreads();   // group `it` has landed; every wave is done with iteration it-1
            stamp(1);
            // group it+1 = {W1 chunk it+1, W2 chunk it-1} (indices clamped at the pipeline's ends) into the buffer iteration
            // it-1 released; nothing in the last iteration (the epilogue reuses the buffers right after the loop)
            bool dma = it <= nchunks && (ABL != 1 || it < 2);
            dma_begin(it + 1 < nchunks ? it + 1 : nchunks - 1, it < 1 ? 0 : (it <= nchunks ? it - 1 : nchunks - 1), (it + 1) & 1);
            const bool p1 = ABL != 2 && it < nchunks, fi = ABL != 2 && ABL != 10 && it >= 1 && it <= nchunks, p2 = ABL != 2 && it >= 2;
            // a stage that does not run this iteration (pipeline fill / drain) cannot carry its share of the DMA group
            if constexpr (kDmaHalf0) {   // experiment: half 0 (which waits at the barrier anyway) issues the whole group
                if (dma && half == 0) {
                    static_for<0, 13>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const int q = wave + 4 * j;
                        if (q < kW1Dma + 24) {
                            const uint32_t t = 64u * (q < kW1Dma ? q : q - kW1Dma) + lane;
                            uint32_t so;
                            if (q < kW1Dma) {
                                uint32_t r = t / 49u, c = t - r * 49u;
                                r = r < 32u ? r : 31u;
                                c = c < 48u ? c : 47u;
                                so = r * 768u + 16u * c;
                            } else {
                                const uint32_t r = t >> 2, c = t & 3u;
                                so = r * 64u + 16u * (c ^ ((r >> 2) & 3u));
                            }
                            const char* src = (q < kW1Dma ? W1b + dma_o1 : W2b + dma_o2) + so;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                             (__attribute__((address_space(3))) void*)(smem + ((it + 1) & 1) * kWbuf + q * 1024), 16, 0, 0);
                        }
                    });
                }
                dma = false;
            }
            bool dma_fin = false;
            if constexpr (kDmaInFinish) {
                if (dma && fi) { dma_fin = true; dma = false; }      // (pipeline fill: no finish stage yet - the group goes out at the barrier)
            }
            if (ABL != 5) {        // the whole group right behind the barrier (ABL == 5, experiment: one instruction at a time
                                   // from inside the matrix stages - measured no faster, and hipcc then drops the vmcnt(0)
                                   // in front of the barrier: racy without the explicit wait below)
                if (dma) static_for<0, kPerWave>([&](auto jc) { dma_one(jc); });
                dma = false;
            }
            if (dma && !p1) static_for<0, 3>([&](auto jc) { dma_one(jc); });
            if (dma && !p2) static_for<3, kPerWave>([&](auto jc) { dma_one(jc); });
            stamp(2);
            if constexpr (kOrder == 0) {
                if (p1) prefetch1(it);
                if (fi) finish(it - 1, dma_fin);
                stamp(3);
                if (p1) product1(it, dma);
                stamp(4);
                if (p2) {
                    prefetch2(it);
                    product2(it, dma);
                }
                stamp(5);
            } else {
                if (p2) {
                    prefetch2(it);
                    product2(it, dma);
                }
                stamp(5);
                if (p1) prefetch1(it);
                if (fi) finish(it - 1, dma_fin);
                stamp(3);
                if (p1) product1(it, dma);
                stamp(4);
            }
        }
    };
    // ---- TWO-SLOT schedule (experiment ABL 30 - 33).  The loop above pairs three stages per wave (one vector, two matrix), so one
    // pairing per iteration is matrix beside matrix and two are a 900-cycle vector stage beside a 384-cycle matrix stage: the
    // matrix pipe idles under the vector stages.  Here an iteration is two slots, a workgroup barrier in front of each; in a
    // slot one wave of every SIMD runs BOTH products of its chunk back to back (24 MFMAs) while its partner runs its vector
    // work (finish of the previous chunk + its share of the DMA group), then they swap:
    //     slot A(k):  half 1: product 1 (chunk k), product 2 (chunk k-2)      half 0: DMA part of group k+1, finish (chunk k-1)
    //     slot B(k):  half 0: product 1 (chunk k), product 2 (chunk k-2)      half 1: DMA part of group k+1, finish (chunk k-1)
    // Hand-offs as before (exchange planes and P tiles by chunk parity, weight buffers by iteration parity), each one slot
    // boundary or more apart.  Half 1 finishes chunk k-1 AFTER it has started chunk k: its own eight partial sums of k-1 are
    // carried in a second register set.  Half 0's DMA share has two slots to land, half 1's one: half 1 gets the smaller share.
    constexpr bool kSlot2 = ABL >= 30 && ABL <= 35;     // (35: two-slot WITHOUT weight DMA after the first groups: compute-bound timing, wrong results)
    if constexpr (kSlot2) {
        auto dma_mine = [&]() __attribute__((always_inline)) { static_for<0, kPerWave>([&](auto jc) { dma_one(jc); }); };
        if (half == 0) {
#pragma unroll 1
            for (int k = 0; k <= nchunks + 1; ++k) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of group k (issued in slot A(k-1))
                __syncthreads();                                        // ---- slot A(k)
                stamp(1);
                dma_begin(k + 1 < nchunks ? k + 1 : nchunks - 1, k < 1 ? 0 : (k <= nchunks ? k - 1 : nchunks - 1), (k + 1) & 1);
                if (k <= nchunks && (ABL != 35 || k < 2)) dma_mine();
                stamp(2);
                if (k >= 1 && k <= nchunks) finish(k - 1);
                stamp(3);
                __syncthreads();                                        // ---- slot B(k)
                stamp(1);
                if (k < nchunks) { prefetch1(k); product1(k, false); }
                stamp(4);
                if (k >= 2) { prefetch2(k); product2(k, false); }
                stamp(5);
            }
        } else {
            float keep_prev[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) keep_prev[i] = 0.f;
#pragma unroll 1
            for (int k = 0; k <= nchunks + 1; ++k) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of group k (issued in slot B(k-1))
                __syncthreads();                                        // ---- slot A(k)
                stamp(1);
                if (k < nchunks) { prefetch1(k); product1(k, false); }     // (writes keep[])
                stamp(4);
                if (k >= 2) { prefetch2(k); product2(k, false); }
                stamp(5);
                __syncthreads();                                        // ---- slot B(k)
                stamp(1);
                dma_begin(k + 1 < nchunks ? k + 1 : nchunks - 1, k < 1 ? 0 : (k <= nchunks ? k - 1 : nchunks - 1), (k + 1) & 1);
                if (k <= nchunks && (ABL != 35 || k < 2)) dma_mine();
                stamp(2);
                // finish(k - 1) on the partial sums of chunk k-1: swap them in for the duration of the stage
                float keep_now[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { keep_now[i] = keep[i]; keep[i] = keep_prev[i]; }
                if (k >= 1 && k <= nchunks) finish(k - 1);
                stamp(3);
#pragma unroll
                for (int i = 0; i < 8; ++i) keep_prev[i] = keep_now[i];
            }
        }
    } else
    if constexpr (kStaticPrio) {     // experiment: the younger half of every SIMD at raised priority for the whole loop, no per-stage flips
        if (half == 1) __builtin_amdgcn_s_setprio(1);
    }
    if constexpr (!kSlot2) {
        if (half == 0) {
            if constexpr (proj_mode) prologue();
            main_loop(std::integral_constant<int, 0>{});
        } else {
            if constexpr (proj_mode) prologue();
            main_loop(std::integral_constant<int, 1>{});
        }
    }
    if constexpr (kStaticPrio) __builtin_amdgcn_s_setprio(0);

    // ---- epilogue of the q/kv mode: 32 rows per pass through the fp32 tile (mask, coalesced stores, row statistics as below) and
    // from the same registers the NEXT layer's attention_norm of the finished rows, bf16, into a second tile - then that
    // layer's q/kv projection of the 128 rows: wave (rg, half) owns rows 32 rg .. and features 256 half .., the weight streams
    // through a three-chunk ring in the fp32 tile's place (one k-step = 16 KB per chunk, requested two steps ahead), the rows
    // leave through a staging tile as whole 1-KB lines.
    if constexpr (qkv_mode) {
        const bool mask_out = p.flags & ISPK_EP_MASK_OUT;
        float* T = reinterpret_cast<float*>(smem);
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            __syncthreads();
            if (rg == pass) {
                float* trow = T + l31 * kLdT + 192 * half + 4 * h;
#pragma unroll
                for (int nt = 0; nt < 6; ++nt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = acc2[nt][4 * gq + e];
                        *reinterpret_cast<f32x4*>(trow + 32 * nt + 8 * gq) = o;
                    }
            }
            __syncthreads();
#pragma unroll 1
            for (int i = 0; i < 2; ++i) {
                const int rl = wave * 4 + 2 * i + h;                 // row of the tile
                const int rb = pass * 32 + rl;                       // row of the block
                const int r = row0 + rb;
                const bool live = r < p.rows;
                const int rc = live ? r : p.rows - 1;
                const float mk = (p.mask && mask_out) ? (p.mask[rc] ? 1.0f : 0.0f) : 1.0f;
                f32x4 y[3];
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    f32x4 a = *reinterpret_cast<const f32x4*>(T + rl * kLdT + 4 * (l31 + 32 * j));
                    if (mask_out) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] *= mk;
                    }
                    y[j] = a;
                    sum += (a[0] + a[1]) + (a[2] + a[3]);
                }
                if (live) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(p.out + (int64_t)r * p.ldo + 4 * (l31 + 32 * j)) = y[j];
                }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
                const float mean = sum * (1.0f / kD);
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
                const float rstd = 1.0f / sqrtf(q * (1.0f / kD) + p.eps2);
                if (p.stats && live && l31 == 0) {
                    p.stats[2 * (int64_t)r] = mean;
                    p.stats[2 * (int64_t)r + 1] = 1.0f / sqrtf(q * (1.0f / kD) + p.stats_eps);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int c = 4 * (l31 + 32 * j);
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma2 + c), b4 = *reinterpret_cast<const f32x4*>(p.beta2 + c);
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaf((y[j][e] - mean) * rstd, g4[e], b4[e]);
                    uint2 pk;
                    pk.x = pack_bf16(o[0], o[1]);
                    pk.y = pack_bf16(o[2], o[3]);
                    *reinterpret_cast<uint2*>(smem + kX2Off + rb * 768 + 16 * ((c >> 3) ^ (rb & 15)) + 8 * (l31 & 1)) = pk;
                }
            }
        }
        __syncthreads();     // the fp32 tile is dead (the ring takes its place), the bf16 tile is complete
        const char* WqB = reinterpret_cast<const char*>(p.WqC);
        auto dma_q = [&](int c, int buf) __attribute__((always_inline)) {     // chunk c: 16 pieces of 1 KB, wave w takes w and w + 8
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(WqB + (int64_t)c * kQChunk + (wave + 8 * j) * 1024 + lane * 16),
                    (__attribute__((address_space(3))) void*)(smem + buf * kQChunk + (wave + 8 * j) * 1024), 16, 0, 0);
        };
        dma_q(0, 0);
        dma_q(1, 1);
        f32x16 aq[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) aq[t][r] = 0.f;
        const uint32_t xb = lds0 + kX2Off + (rg * 32 + l31) * 768;        // B fragments: + 16 ((2 c + h) ^ (l31 & 15))
        const uint32_t wa = lds0 + (256 * half + l31) * 32 + 16 * h;       // A fragments: + chunk buffer + 1024 t (lane-linear: conflict-free)
        bf16x8 bq;
#pragma unroll 1
        for (int c = 0, buf = 0; c < 24; ++c) {
            // chunk c has landed for this wave (at most chunk c + 1's two instructions are younger) ...
            if (c < 23) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // ... and for the others, who are also done with chunk c - 1
            asm volatile("" ::: "memory");
            if (c + 2 < 24) dma_q(c + 2, buf >= 1 ? buf - 1 : 2);      // (buf + 2) % 3: the buffer chunk c - 1 has left
            const uint32_t a = wa + buf * kQChunk;
            lds_read_b128_asm<0>(bq, xb + 16 * ((2 * c + h) ^ (l31 & 15)));
            static_for<0, kRing>([&](auto tc) { lds_read_b128_asm<1024 * decltype(tc)::value>(r2[decltype(tc)::value], a); });
            if constexpr (kPrio) __builtin_amdgcn_s_setprio(1);
            static_for<0, 8>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                lds_wait<(7 - t) < kRing - 1 ? (7 - t) : kRing - 1>();
                __builtin_amdgcn_sched_barrier(0);
                aq[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r2[t % kRing], bq, aq[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (t + kRing < 8) lds_read_b128_asm<1024 * (t + kRing)>(r2[t % kRing], a);
            });
            if constexpr (kPrio) __builtin_amdgcn_s_setprio(0);
            buf = buf == 2 ? 0 : buf + 1;
        }
        __syncthreads();     // both tiles and the ring are dead: the staging tile goes over them
        {
            char* orow = smem + (rg * 32 + l31) * kOutRow + 2 * (256 * half + 4 * h);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    uint2 pk;
                    pk.x = pack_bf16(aq[t][4 * gq], aq[t][4 * gq + 1]);
                    pk.y = pack_bf16(aq[t][4 * gq + 2], aq[t][4 * gq + 3]);
                    *reinterpret_cast<uint2*>(orow + 2 * (32 * t + 8 * gq)) = pk;
                }
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int rb = wave * 16 + i, r = row0 + rb;
            if (r < p.rows)
                *reinterpret_cast<u32x4*>(p.qkv + (int64_t)r * p.ld_qkv + 8 * lane) =
                    *reinterpret_cast<const u32x4*>(smem + rb * kOutRow + 16 * lane);
        }
        return;
    }
    // ---- epilogue: 64 rows per pass through the fp32 tile; then whole rows: + x, mask, statistics, coalesced stores
    const bool mask_acc = p.flags & ISPK_EP_MASK_ACC, mask_out = p.flags & ISPK_EP_MASK_OUT;
    float* T = reinterpret_cast<float*>(smem);
    float* outp = p.out + (split_mode ? (int64_t)blockIdx.y * p.part_stride : 0);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        if ((rg >> 1) == pass) {
            float* trow = T + ((rg & 1) * 32 + l31) * kLdT + 192 * half + 4 * h;
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc2[nt][4 * gq + e];
                    *reinterpret_cast<f32x4*>(trow + 32 * nt + 8 * gq) = o;
                }
        }
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int rl = wave * 8 + 2 * i + h;                 // row of the tile
            const int r = row0 + pass * 64 + rl;
            const bool live = r < p.rows;
            const int rc = live ? r : p.rows - 1;
            const float mk = (p.mask && (mask_acc || mask_out)) ? (p.mask[rc] ? 1.0f : 0.0f) : 1.0f;   // (projection mode: MASK_ACC belongs to the prologue)
            f32x4 y[3];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = 4 * (l31 + 32 * j);
                f32x4 a = *reinterpret_cast<const f32x4*>(T + rl * kLdT + c);
                if constexpr (split_mode) {
                    // the raw partial product; residual, mask and sums happen in the combine pass (projection mode: split 0's
                    // partial product already contains x1)
                } else if constexpr (proj_mode) {      // (the residual has been in the accumulators since the prologue)
                    if (mask_out) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] *= mk;
                    }
                } else {
                    const f32x4 xr = *reinterpret_cast<const f32x4*>(p.x + (int64_t)rc * p.ldx + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = mask_acc ? a[e] * mk : a[e];
                        t += xr[e];
                        a[e] = mask_out ? t * mk : t;
                    }
                }
                y[j] = a;
                s += (a[0] + a[1]) + (a[2] + a[3]);
            }
            if (live && (ABL != kProjMode || p.out)) {
#pragma unroll
                for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(outp + (int64_t)r * p.ldo + 4 * (l31 + 32 * j)) = y[j];
            }
            if constexpr (ABL == kProjMode) {
                if (p.ln_out) {      // the stack's final LayerNorm (transformer.py:205-206) from the same registers
                    float sum = s;
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
                    const float mean = sum * (1.0f / kD);
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
                    const float rstd = 1.0f / sqrtf(q * (1.0f / kD) + p.eps2), om = p.ln_mask ? mk : 1.0f;
                    if (live) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int c = 4 * (l31 + 32 * j);
                            const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma2 + c), b4 = *reinterpret_cast<const f32x4*>(p.beta2 + c);
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = ((y[j][e] - mean) * rstd * g4[e] + b4[e]) * om;
                            if (p.ln_bf16) {
                                uint2 pk;
                                pk.x = pack_bf16(o[0], o[1]);
                                pk.y = pack_bf16(o[2], o[3]);
                                *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.ln_out) + (int64_t)r * p.ld_ln + c) = pk;
                            } else {
                                *reinterpret_cast<f32x4*>(static_cast<float*>(p.ln_out) + (int64_t)r * p.ld_ln + c) = f32x4{o[0], o[1], o[2], o[3]};
                            }
                        }
                    }
                }
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
    if constexpr (ABL == 3 || ABL == 34) {
        stamp(6);
        ts[7] = t_prev - t_first;
        if (lane == 0 && p.stamps) {
            unsigned long long* o = p.stamps + ((int64_t)blockIdx.x * 8 + wave) * 8;
            for (int i = 0; i < 8; ++i) o[i] = ts[i];
        }
    }
}

// W2 [dim][inner] (nn.Linear layout) -> chunk-contiguous [inner / 32][dim][32]: chunk c is one 24-KB block
__global__ __launch_bounds__(256) void ffn_chunk_w2_kernel(const uint16_t* __restrict__ W2, int64_t ldw2,
                                                           uint16_t* __restrict__ out, int dim, int inner) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one 16-byte piece (8 hidden units) per thread
    const int64_t total = (int64_t)dim * inner / 8;
    if (i >= total) return;
    const int piece = (int)(i & 3);
    const int64_t rest = i >> 2;
    const int d = (int)(rest % dim), c = (int)(rest / dim);
    *reinterpret_cast<u32x4*>(out + (((int64_t)c * dim + d) * 32 + piece * 8)) =
        *reinterpret_cast<const u32x4*>(W2 + (int64_t)d * ldw2 + c * 32 + piece * 8);
}

// W [N][K] (nn.Linear layout) -> k-step chunks [K / 16][N][16]: chunk c is N rows of 32 bytes - as an LDS image the MFMA A fragments
// of 32 consecutive rows are one contiguous kilobyte
__global__ __launch_bounds__(256) void chunk_k16_kernel(const uint16_t* __restrict__ W, int64_t ldw, uint16_t* __restrict__ out,
                                                        int N, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one 16-byte piece (8 k-values) per thread
    const int64_t total = (int64_t)N * K / 8;
    if (i >= total) return;
    const int piece = (int)(i & 1);
    const int64_t rest = i >> 1;
    const int n = (int)(rest % N), c = (int)(rest / N);
    *reinterpret_cast<u32x4*>(out + (((int64_t)c * N + n) * 16 + piece * 8)) =
        *reinterpret_cast<const u32x4*>(W + (int64_t)n * ldw + c * 16 + piece * 8);
}

}  // namespace

extern "C" int32_t ispk_chunk_k16_bf16(const uint16_t* W, int64_t ldw, int32_t N, int32_t K, uint16_t* out, ispk_stream_t stream) {
    ISPK_REQUIRE(W && out, ISPK_E_NULL, "chunk_k16: null pointer");
    ISPK_REQUIRE(N >= 1 && K >= 16 && K % 16 == 0 && ldw >= K, ISPK_E_SHAPE, "chunk_k16: bad shape %d x %d", N, K);
    ISPK_REQUIRE(ldw % 8 == 0 && ispk_aligned(W, 16) && ispk_aligned(out, 16), ISPK_E_ALIGN, "chunk_k16: 16-byte alignment required");
    const int64_t total = (int64_t)N * K / 8;
    hipLaunchKernelGGL(chunk_k16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       W, ldw, out, N, K);
    return ispk_launch_status();
}

extern "C" int32_t ispk_ffn_chunk_w2_bf16(const uint16_t* W2, int64_t ldw2, int32_t dim, int32_t inner, uint16_t* out,
                                          ispk_stream_t stream) {
    ISPK_REQUIRE(W2 && out, ISPK_E_NULL, "ffn_chunk_w2: null pointer");
    ISPK_REQUIRE(dim >= 1 && inner >= 32 && inner % 32 == 0 && ldw2 >= inner, ISPK_E_SHAPE, "ffn_chunk_w2: bad shape %d x %d",
                 dim, inner);
    ISPK_REQUIRE(ldw2 % 8 == 0 && ispk_aligned(W2, 16) && ispk_aligned(out, 16), ISPK_E_ALIGN,
                 "ffn_chunk_w2: 16-byte alignment required");
    const int64_t total = (int64_t)dim * inner / 8;
    hipLaunchKernelGGL(ffn_chunk_w2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), W2, ldw2, out, dim, inner);
    return ispk_launch_status();
}

// ------------------------------------------------------------------------------------------------ small batches: split + combine
// y = [mask] * (x + sum_s part[s]) in split order, then (optionally) LN(y) for the layer that consumes it.  One row per 32
// lanes, three float4 per lane (dim 384), two-pass statistics in registers.
template <typename TOut>
__global__ __launch_bounds__(256) void ffn_combine_ln_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ part,
                                                             int64_t part_stride, int splits, const uint8_t* __restrict__ mask,
                                                             float* __restrict__ y, int64_t ldy, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int ln_mask,
                                                             TOut* __restrict__ ln_out, int64_t ld_ln, int rows) {
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (row >= rows) return;
    const float mk = mask ? (mask[row] ? 1.f : 0.f) : 1.f;
    f32x4 v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)      // (x NULL: the residual is inside part 0 - ispk_attn_out_ffn_split_bf16)
        v[j] = x ? *reinterpret_cast<const f32x4*>(x + (int64_t)row * ldx + 4 * (l + 32 * j)) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) {
        const float* ps = part + (int64_t)s * part_stride + (int64_t)row * kD;
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] += *reinterpret_cast<const f32x4*>(ps + 4 * (l + 32 * j));
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        v[j] *= mk;
        *reinterpret_cast<f32x4*>(y + (int64_t)row * ldy + 4 * (l + 32 * j)) = v[j];
        sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    if (!ln_out) return;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum * (1.0f / kD);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[j][e] - mean;
            q = fmaf(d, d, q);
        }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / kD) + eps), om = ln_mask ? mk : 1.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = 4 * (l + 32 * j);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ((v[j][e] - mean) * rstd * g[e] + bt[e]) * om;
        if constexpr (std::is_same<TOut, float>::value) {
            *reinterpret_cast<f32x4*>(ln_out + (int64_t)row * ld_ln + c) = f32x4{o[0], o[1], o[2], o[3]};
        } else {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
            *reinterpret_cast<uint2*>(ln_out + (int64_t)row * ld_ln + c) = pk;
        }
    }
}

extern "C" int32_t ispk_ffn_bf16_prenorm2_split(const float* x, int64_t ldx, const float* norm_gamma, const float* norm_beta,
                                                float norm_eps, const uint16_t* W1, const uint16_t* W2_chunks, float* parts,
                                                int64_t part_stride, int32_t splits, int32_t rows, int32_t dim, int32_t inner,
                                                ispk_stream_t stream) {
    ISPK_REQUIRE(x && norm_gamma && norm_beta && W1 && W2_chunks && parts, ISPK_E_NULL, "ffn_prenorm2_split: null pointer");
    ISPK_REQUIRE(dim == kD, ISPK_E_UNSUPPORTED, "ffn_prenorm2_split: dim %d (built for 384)", dim);
    ISPK_REQUIRE(rows >= 0 && inner >= 32 && inner % 32 == 0 && splits >= 1 && (inner / 32) % splits == 0 &&
                     (inner / 32) / splits >= 2 && part_stride >= (int64_t)rows * kD,
                 ISPK_E_SHAPE, "ffn_prenorm2_split: bad shape rows=%d inner=%d splits=%d (chunks of 32 must divide evenly, >= 2 each)",
                 rows, inner, splits);
    ISPK_REQUIRE(ldx % 4 == 0 && ldx >= dim && part_stride % 4 == 0 && ispk_aligned(x, 16) && ispk_aligned(parts, 16) &&
                     ispk_aligned(W1, 16) && ispk_aligned(W2_chunks, 16) && ispk_aligned(norm_gamma, 16) &&
                     ispk_aligned(norm_beta, 16), ISPK_E_ALIGN, "ffn_prenorm2_split: 16-byte alignment required");
    if (rows == 0) return 0;
    Ffn2Params p{x, ldx, norm_gamma, norm_beta, norm_eps, W1, W2_chunks, nullptr, parts, kD, rows, inner, 0u, nullptr, 0.f};
    p.chunk_count = (inner / 32) / splits;
    p.part_stride = part_stride;
    ISPK_RESERVE_LDS((&ffn2_bf16_kernel<kSplitMode>), kLds, "ffn_prenorm2_split");
    hipLaunchKernelGGL(ffn2_bf16_kernel<kSplitMode>, dim3((rows + 127) / 128, splits), dim3(512), kLds,
                       reinterpret_cast<hipStream_t>(stream), p);
    return ispk_launch_status();
}

// Small batches, with the attention block's output projection as the prologue of EVERY split (each needs LN(x1); split 0's partial
// product carries x1 itself, so the combine pass runs with x = NULL):  parts[0] = x1 + ffn_0(LN(x1)), parts[s] = ffn_s(LN(x1))
extern "C" int32_t ispk_attn_out_ffn_split_bf16(const float* x, int64_t ldx, const uint16_t* attn_out, int64_t ld_attn,
                                                const uint16_t* Wo_chunks, const float* norm_gamma, const float* norm_beta,
                                                float norm_eps, const uint16_t* W1, const uint16_t* W2_chunks,
                                                const uint8_t* mask, uint32_t flags, float* parts, int64_t part_stride,
                                                int32_t splits, int32_t rows, int32_t dim, int32_t inner, ispk_stream_t stream) {
    ISPK_REQUIRE(x && attn_out && Wo_chunks && norm_gamma && norm_beta && W1 && W2_chunks && parts, ISPK_E_NULL,
                 "attn_out_ffn_split: null pointer");
    ISPK_REQUIRE(dim == kD, ISPK_E_UNSUPPORTED, "attn_out_ffn_split: dim %d (built for 384 = heads * 64)", dim);
    ISPK_REQUIRE(rows >= 0 && inner >= 32 && inner % 32 == 0 && splits >= 1 && (inner / 32) % splits == 0 &&
                     (inner / 32) / splits >= 2 && part_stride >= (int64_t)rows * kD,
                 ISPK_E_SHAPE, "attn_out_ffn_split: bad shape rows=%d inner=%d splits=%d (chunks of 32 must divide evenly, >= 2 each)",
                 rows, inner, splits);
    ISPK_REQUIRE((flags & ~ISPK_EP_MASK_ACC) == 0 && !((flags & ISPK_EP_MASK_ACC) && !mask), ISPK_E_UNSUPPORTED,
                 "attn_out_ffn_split: flags other than MASK_ACC (with a mask) are the combine pass's business");
    ISPK_REQUIRE(ldx % 4 == 0 && ldx >= dim && ld_attn % 8 == 0 && ld_attn >= dim && part_stride % 4 == 0 && ispk_aligned(x, 16) &&
                     ispk_aligned(attn_out, 16) && ispk_aligned(Wo_chunks, 16) && ispk_aligned(parts, 16) && ispk_aligned(W1, 16) &&
                     ispk_aligned(W2_chunks, 16) && ispk_aligned(norm_gamma, 16) && ispk_aligned(norm_beta, 16),
                 ISPK_E_ALIGN, "attn_out_ffn_split: 16-byte alignment required");
    if (rows == 0) return 0;
    Ffn2Params p{x, ldx, norm_gamma, norm_beta, norm_eps, W1, W2_chunks, mask, parts, kD, rows, inner, flags, nullptr, 0.f};
    p.chunk_count = (inner / 32) / splits;
    p.part_stride = part_stride;
    p.o = attn_out;
    p.ld_o = ld_attn;
    p.WoC = Wo_chunks;
    ISPK_RESERVE_LDS((&ffn2_bf16_kernel<kSplitProjMode>), kLds, "attn_out_ffn_split");
    hipLaunchKernelGGL(ffn2_bf16_kernel<kSplitProjMode>, dim3((rows + 127) / 128, splits), dim3(512), kLds,
                       reinterpret_cast<hipStream_t>(stream), p);
    return ispk_launch_status();
}

extern "C" int32_t ispk_ffn_combine_ln_f32(const float* x, int64_t ldx, const float* parts, int64_t part_stride, int32_t splits,
                                           const uint8_t* mask, float* y, int64_t ldy, const float* ln_gamma,
                                           const float* ln_beta, float ln_eps, int32_t ln_mask, void* ln_out, int64_t ld_ln,
                                           int32_t ln_bf16, int32_t rows, int32_t dim, ispk_stream_t stream) {
    ISPK_REQUIRE(parts && y, ISPK_E_NULL, "ffn_combine_ln: null pointer");
    if (!x) ldx = dim;
    ISPK_REQUIRE(dim == kD && rows >= 0 && splits >= 1 && ldx >= dim && ldy >= dim && ldx % 4 == 0 && ldy % 4 == 0 &&
                     part_stride % 4 == 0, ISPK_E_SHAPE, "ffn_combine_ln: bad shape rows=%d dim=%d splits=%d", rows, dim, splits);
    ISPK_REQUIRE(!ln_out || (ln_gamma && ln_beta && ld_ln >= dim && ld_ln % 4 == 0), ISPK_E_NULL,
                 "ffn_combine_ln: LayerNorm output without gamma / beta");
    ISPK_REQUIRE((!x || ispk_aligned(x, 16)) && ispk_aligned(parts, 16) && ispk_aligned(y, 16) && (!ln_out || ispk_aligned(ln_out, 8)),
                 ISPK_E_ALIGN, "ffn_combine_ln: 16-byte alignment required");
    if (rows == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((rows + 7) / 8);
    if (ln_out && ln_bf16)
        hipLaunchKernelGGL(ffn_combine_ln_kernel<uint16_t>, grid, dim3(256), 0, s, x, ldx, parts, part_stride, splits, mask, y, ldy,
                           ln_gamma, ln_beta, ln_eps, ln_mask, static_cast<uint16_t*>(ln_out), ld_ln, rows);
    else
        hipLaunchKernelGGL(ffn_combine_ln_kernel<float>, grid, dim3(256), 0, s, x, ldx, parts, part_stride, splits, mask, y, ldy,
                           ln_gamma, ln_beta, ln_eps, ln_mask, static_cast<float*>(ln_out), ld_ln, rows);
    return ispk_launch_status();
}

// The attention block's output projection + the feed-forward block of a pre-norm layer, one kernel (kProjMode above):
//     x1 = x + [mask] * (attn_out · Woᵀ);   out = [mask] * (x1 + gelu(LN(x1) · W1ᵀ) · W2ᵀ)     transformer.py:91-110
// and (kProjQkvMode) also the next layer's   qkv = LN_next(out) · [Wq; Wkv]ᵀ                    transformer.py:79-80, attention.py:63-64
static int32_t attn_out_ffn_launch(const float* x, int64_t ldx, const uint16_t* attn_out, int64_t ld_attn,
                                   const uint16_t* Wo_chunks, const float* norm_gamma, const float* norm_beta, float norm_eps,
                                   const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask, float* out, int64_t ldo,
                                   int32_t rows, int32_t dim, int32_t inner, uint32_t flags, float* row_stats, float stats_eps,
                                   const float* next_gamma, const float* next_beta, float next_eps, const uint16_t* Wqkv_chunks,
                                   uint16_t* qkv, int64_t ld_qkv, bool with_qkv, ispk_stream_t stream) {
    ISPK_REQUIRE(x && attn_out && Wo_chunks && norm_gamma && norm_beta && W1 && W2_chunks && out, ISPK_E_NULL,
                 "attn_out_ffn: null pointer");
    ISPK_REQUIRE(dim == kD, ISPK_E_UNSUPPORTED, "attn_out_ffn: dim %d (built for 384 = heads * 64)", dim);
    ISPK_REQUIRE(rows >= 0 && inner >= 64 && inner % 32 == 0, ISPK_E_SHAPE, "attn_out_ffn: bad shape rows=%d inner=%d", rows, inner);
    ISPK_REQUIRE((flags & ~(ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) == 0, ISPK_E_UNSUPPORTED, "attn_out_ffn: unsupported flags");
    ISPK_REQUIRE(!((flags & (ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) && !mask), ISPK_E_NULL, "attn_out_ffn: mask flag without mask");
    ISPK_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ld_attn % 8 == 0 && ldx >= dim && ldo >= dim && ld_attn >= dim &&
                     ispk_aligned(x, 16) && ispk_aligned(out, 16) && ispk_aligned(attn_out, 16) && ispk_aligned(Wo_chunks, 16) &&
                     ispk_aligned(W1, 16) && ispk_aligned(W2_chunks, 16) && ispk_aligned(norm_gamma, 16) &&
                     ispk_aligned(norm_beta, 16) && (!row_stats || ispk_aligned(row_stats, 8)),
                 ISPK_E_ALIGN, "attn_out_ffn: 16-byte alignment / strides that are multiples of 4 (fp32) and 8 (bf16) required");
    if (with_qkv) {
        ISPK_REQUIRE(next_gamma && next_beta && Wqkv_chunks && qkv, ISPK_E_NULL, "attn_out_ffn_qkv: null pointer");
        ISPK_REQUIRE(ld_qkv >= kNq && ld_qkv % 8 == 0 && ispk_aligned(qkv, 16) && ispk_aligned(Wqkv_chunks, 16) &&
                         ispk_aligned(next_gamma, 16) && ispk_aligned(next_beta, 16),
                     ISPK_E_ALIGN, "attn_out_ffn_qkv: q/kv rows of 512 bf16, 16-byte aligned, row stride a multiple of 8");
    }
    if (rows == 0) return 0;
    Ffn2Params p{x, ldx, norm_gamma, norm_beta, norm_eps, W1, W2_chunks, mask, out, ldo, rows, inner, flags, row_stats, stats_eps};
    p.o = attn_out;
    p.ld_o = ld_attn;
    p.WoC = Wo_chunks;
    const dim3 grid((rows + 127) / 128);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (with_qkv) {
        p.gamma2 = next_gamma;
        p.beta2 = next_beta;
        p.eps2 = next_eps;
        p.WqC = Wqkv_chunks;
        p.qkv = qkv;
        p.ld_qkv = ld_qkv;
        ISPK_RESERVE_LDS((&ffn2_bf16_kernel<kProjQkvMode>), kLds, "attn_out_ffn_qkv");
        hipLaunchKernelGGL(ffn2_bf16_kernel<kProjQkvMode>, grid, dim3(512), kLds, s, p);
    } else {
        ISPK_RESERVE_LDS((&ffn2_bf16_kernel<kProjMode>), kLds, "attn_out_ffn");
        hipLaunchKernelGGL(ffn2_bf16_kernel<kProjMode>, grid, dim3(512), kLds, s, p);
    }
    return ispk_launch_status();
}

extern "C" int32_t ispk_attn_out_ffn_bf16(const float* x, int64_t ldx, const uint16_t* attn_out, int64_t ld_attn,
                                          const uint16_t* Wo_chunks, const float* norm_gamma, const float* norm_beta,
                                          float norm_eps, const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask,
                                          float* out, int64_t ldo, int32_t rows, int32_t dim, int32_t inner, uint32_t flags,
                                          float* row_stats, float stats_eps, ispk_stream_t stream) {
    return attn_out_ffn_launch(x, ldx, attn_out, ld_attn, Wo_chunks, norm_gamma, norm_beta, norm_eps, W1, W2_chunks, mask, out, ldo,
                               rows, dim, inner, flags, row_stats, stats_eps, nullptr, nullptr, 0.f, nullptr, nullptr, 0, false, stream);
}

// ... with the stack's FINAL LayerNorm as the epilogue's second output (last layer): ln_out = [mask if ln_mask] * LN_final(out), bf16 or
// fp32; `out` may be NULL when only the normalised rows are consumed (the decoder feeding to_mel)
extern "C" int32_t ispk_attn_out_ffn_norm_bf16(const float* x, int64_t ldx, const uint16_t* attn_out, int64_t ld_attn,
                                               const uint16_t* Wo_chunks, const float* norm_gamma, const float* norm_beta,
                                               float norm_eps, const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask,
                                               float* out, int64_t ldo, int32_t rows, int32_t dim, int32_t inner, uint32_t flags,
                                               const float* final_gamma, const float* final_beta, float final_eps, int32_t ln_mask,
                                               void* ln_out, int64_t ld_ln, int32_t ln_bf16, ispk_stream_t stream) {
    ISPK_REQUIRE(x && attn_out && Wo_chunks && norm_gamma && norm_beta && W1 && W2_chunks && final_gamma && final_beta && ln_out,
                 ISPK_E_NULL, "attn_out_ffn_norm: null pointer");
    ISPK_REQUIRE(dim == kD, ISPK_E_UNSUPPORTED, "attn_out_ffn_norm: dim %d (built for 384 = heads * 64)", dim);
    ISPK_REQUIRE(rows >= 0 && inner >= 64 && inner % 32 == 0, ISPK_E_SHAPE, "attn_out_ffn_norm: bad shape rows=%d inner=%d", rows, inner);
    ISPK_REQUIRE((flags & ~(ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) == 0, ISPK_E_UNSUPPORTED, "attn_out_ffn_norm: unsupported flags");
    ISPK_REQUIRE(!(((flags & (ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) || ln_mask) && !mask), ISPK_E_NULL,
                 "attn_out_ffn_norm: mask flag without mask");
    ISPK_REQUIRE(ldx % 4 == 0 && ld_attn % 8 == 0 && ldx >= dim && ld_attn >= dim && ld_ln >= dim && ld_ln % 4 == 0 &&
                     (!out || (ldo % 4 == 0 && ldo >= dim && ispk_aligned(out, 16))) && ispk_aligned(x, 16) &&
                     ispk_aligned(attn_out, 16) && ispk_aligned(Wo_chunks, 16) && ispk_aligned(W1, 16) && ispk_aligned(W2_chunks, 16) &&
                     ispk_aligned(norm_gamma, 16) && ispk_aligned(norm_beta, 16) && ispk_aligned(final_gamma, 16) &&
                     ispk_aligned(final_beta, 16) && ispk_aligned(ln_out, ln_bf16 ? 8 : 16),
                 ISPK_E_ALIGN, "attn_out_ffn_norm: 16-byte alignment / strides that are multiples of 4 (fp32) and 8 (bf16) required");
    if (rows == 0) return 0;
    Ffn2Params p{x, ldx, norm_gamma, norm_beta, norm_eps, W1, W2_chunks, mask, out, ldo, rows, inner, flags, nullptr, 0.f};
    p.o = attn_out;
    p.ld_o = ld_attn;
    p.WoC = Wo_chunks;
    p.gamma2 = final_gamma;
    p.beta2 = final_beta;
    p.eps2 = final_eps;
    p.ln_out = ln_out;
    p.ld_ln = ld_ln;
    p.ln_bf16 = ln_bf16;
    p.ln_mask = ln_mask;
    ISPK_RESERVE_LDS((&ffn2_bf16_kernel<kProjMode>), kLds, "attn_out_ffn_norm");
    hipLaunchKernelGGL(ffn2_bf16_kernel<kProjMode>, dim3((rows + 127) / 128), dim3(512), kLds, reinterpret_cast<hipStream_t>(stream), p);
    return ispk_launch_status();
}

extern "C" int32_t ispk_attn_out_ffn_qkv_bf16(const float* x, int64_t ldx, const uint16_t* attn_out, int64_t ld_attn,
                                              const uint16_t* Wo_chunks, const float* norm_gamma, const float* norm_beta,
                                              float norm_eps, const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask,
                                              float* out, int64_t ldo, int32_t rows, int32_t dim, int32_t inner, uint32_t flags,
                                              const float* next_gamma, const float* next_beta, float next_eps,
                                              const uint16_t* Wqkv_chunks, uint16_t* qkv, int64_t ld_qkv, ispk_stream_t stream) {
    return attn_out_ffn_launch(x, ldx, attn_out, ld_attn, Wo_chunks, norm_gamma, norm_beta, norm_eps, W1, W2_chunks, mask, out, ldo,
                               rows, dim, inner, flags, nullptr, 1e-5f, next_gamma, next_beta, next_eps, Wqkv_chunks, qkv, ld_qkv,
                               true, stream);
}

extern "C" int32_t ispk_ffn_bf16_prenorm2(const float* x, int64_t ldx, const float* norm_gamma, const float* norm_beta,
                                          float norm_eps, const uint16_t* W1, const uint16_t* W2_chunks, const uint8_t* mask,
                                          float* out, int64_t ldo, int32_t rows, int32_t dim, int32_t inner, uint32_t flags,
                                          float* row_stats, float stats_eps, ispk_stream_t stream) {
    ISPK_REQUIRE(x && norm_gamma && norm_beta && W1 && W2_chunks && out, ISPK_E_NULL, "ffn_prenorm2: null pointer");
    ISPK_REQUIRE(dim == kD, ISPK_E_UNSUPPORTED, "ffn_prenorm2: dim %d (built for 384)", dim);
    ISPK_REQUIRE(rows >= 0 && inner >= 32 && inner % 32 == 0, ISPK_E_SHAPE, "ffn_prenorm2: bad shape rows=%d inner=%d", rows,
                 inner);
    ISPK_REQUIRE((flags & ~(ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) == 0, ISPK_E_UNSUPPORTED, "ffn_prenorm2: unsupported flags");
    ISPK_REQUIRE(!((flags & (ISPK_EP_MASK_OUT | ISPK_EP_MASK_ACC)) && !mask), ISPK_E_NULL, "ffn_prenorm2: mask flag without mask");
    ISPK_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= dim && ldo >= dim && ispk_aligned(x, 16) && ispk_aligned(out, 16) &&
                     ispk_aligned(W1, 16) && ispk_aligned(W2_chunks, 16) && ispk_aligned(norm_gamma, 16) &&
                     ispk_aligned(norm_beta, 16) && (!row_stats || ispk_aligned(row_stats, 8)),
                 ISPK_E_ALIGN, "ffn_prenorm2: 16-byte alignment / strides that are multiples of 4 required");
    if (rows == 0) return 0;
    Ffn2Params p{x, ldx, norm_gamma, norm_beta, norm_eps, W1, W2_chunks, mask, out, ldo, rows, inner, flags, row_stats, stats_eps};
    const dim3 grid((rows + 127) / 128);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#ifdef ISPK_EXPERIMENTS
    if (const char* e = ispk_knob("ISPK_FFN2_ABLATE")) {
        if (atoi(e) == 1) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<1>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<1>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 2) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<2>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<2>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 6) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<6>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<6>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 7) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<7>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<7>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 8) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<8>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<8>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 9) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<9>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<9>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 10) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<10>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<10>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 11) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<11>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<11>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 12) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<12>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<12>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
#define ISPK_FFN2_AB(N_)                                                                    \
        if (atoi(e) == N_) {                                                                \
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<N_>), kLds, "ffn_prenorm2");               \
            hipLaunchKernelGGL(ffn2_bf16_kernel<N_>, grid, dim3(512), kLds, s, p);          \
            return ispk_launch_status();                                                    \
        }
        ISPK_FFN2_AB(13) ISPK_FFN2_AB(14) ISPK_FFN2_AB(15) ISPK_FFN2_AB(16) ISPK_FFN2_AB(17) ISPK_FFN2_AB(30) ISPK_FFN2_AB(31) ISPK_FFN2_AB(32) ISPK_FFN2_AB(33) ISPK_FFN2_AB(35) ISPK_FFN2_AB(36) ISPK_FFN2_AB(40)
#undef ISPK_FFN2_AB
        if (atoi(e) == 5) {
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<5>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<5>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 3) {
            const char* sp = ispk_knob("ISPK_FFN2_STAMP");
            p.stamps = sp ? reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 16)) : nullptr;
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<3>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<3>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
        if (atoi(e) == 34) {
            const char* sp = ispk_knob("ISPK_FFN2_STAMP");
            p.stamps = sp ? reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 16)) : nullptr;
            ISPK_RESERVE_LDS((&ffn2_bf16_kernel<34>), kLds, "ffn_prenorm2");
            hipLaunchKernelGGL(ffn2_bf16_kernel<34>, grid, dim3(512), kLds, s, p);
            return ispk_launch_status();
        }
    }
#endif
    ISPK_RESERVE_LDS((&ffn2_bf16_kernel<0>), kLds, "ffn_prenorm2");
    hipLaunchKernelGGL(ffn2_bf16_kernel<0>, grid, dim3(512), kLds, s, p);
    return ispk_launch_status();
}
