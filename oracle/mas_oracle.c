/* ORACLE (test infrastructure, not product): Monotonic Alignment Search on the CPU.
 *
 * Plain-C restatement of the reference's `mas_width1` / `b_mas`
 * (/root/reference/tts/modules/aligner/mas.py:7-35; device twin cuda_mas.py:11-46):
 *
 *   Q[0][0]   = lp[0][0];          Q[0][j>0] = -inf                       (mas.py:11)
 *   Q[i][0]   = Q[i-1][0] + lp[i][0]          (np.cumsum, sequential fp32) (mas.py:12)
 *   Q[i][j>0] = lp[i][j] + max(Q[i-1][j-1], Q[i-1][j])                     (mas.py:13-14)
 *   prev(i,j) = j-1  iff  j>0 and Q[i-1][j-1] >= Q[i-1][j]   (ties -> diagonal, mas.py:17)
 *   backtrack from (n-1, m-1), one-hot int16 rows                          (mas.py:20-24)
 *
 * One fp32 add per cell, so results are bit-reproducible.  Unlike the reference's CPU branch
 * this does NOT mutate its input.  Rows >= out_len and columns >= in_len stay zero (mas.py:31-34).
 * n < m is allowed (leading tokens then receive no frames), exactly as the reference behaves.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC   (see oracle/build_oracle.py)
 * Used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void mas_one(const float* lp, int64_t row_stride, int n, int m, int16_t* out, int64_t out_row_stride,
                    int16_t* path /* [n] or NULL */) {
    if (n <= 0 || m <= 0) return;
    float* prev = (float*)malloc(sizeof(float) * (size_t)m * 2);
    float* cur = prev + m;
    /* one back-pointer bit per cell: 1 = came from the diagonal (j-1) */
    size_t words = ((size_t)m + 63) / 64;
    uint64_t* bp = (uint64_t*)calloc((size_t)n * words, sizeof(uint64_t));
    prev[0] = lp[0];
    for (int j = 1; j < m; ++j) prev[j] = -INFINITY;
    for (int i = 1; i < n; ++i) {
        const float* row = lp + (int64_t)i * row_stride;
        uint64_t* bits = bp + (size_t)i * words;
        cur[0] = prev[0] + row[0];
        for (int j = 1; j < m; ++j) {
            float d = prev[j - 1], s = prev[j];
            int diag = d >= s;
            float best = diag ? d : s;      /* == np.maximum for non-NaN input */
            cur[j] = row[j] + best;
            if (diag) bits[j >> 6] |= (uint64_t)1 << (j & 63);
        }
        float* t = prev; prev = cur; cur = t;
    }
    int j = m - 1;
    for (int i = n - 1; i >= 0; --i) {
        out[(int64_t)i * out_row_stride + j] = 1;
        if (path) path[i] = (int16_t)j;
        if (i > 0 && ((bp[(size_t)i * words + (j >> 6)] >> (j & 63)) & 1)) --j;
    }
    free(bp);
    free(prev < cur ? prev : cur);
}

/* logits [B][M][L] fp32 contiguous; in_lens = text lengths, out_lens = mel lengths (int64, as the collator
 * makes them: collator.py:36,45); attn_out [B][M][L] int16 (zero-filled here); path [B][M] int16 or NULL
 * (text index chosen per mel row, -1 beyond out_len). */
void oracle_b_mas(const float* logits, const int64_t* in_lens, const int64_t* out_lens, int16_t* attn_out,
                  int16_t* path, int64_t B, int64_t M, int64_t L) {
    memset(attn_out, 0, sizeof(int16_t) * (size_t)(B * M * L));
    if (path) memset(path, 0xff, sizeof(int16_t) * (size_t)(B * M));
#pragma omp parallel for schedule(dynamic)
    for (int64_t b = 0; b < B; ++b) {
        mas_one(logits + b * M * L, L, (int)out_lens[b], (int)in_lens[b], attn_out + b * M * L, L,
                path ? path + b * M : NULL);
    }
}
