// Counter-based dropout shared by the training kernels (backward.hip, attention_train.hip).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

// Dropout masks are a pure function of (seed, element index): forward and backward evaluate the same function instead of
// storing a mask.  One 32-bit multiply-xorshift round over (index ^ seed word), see drop_hash; an element is KEPT when the hash is
// >= p * 2^32.  (The reference draws torch's Philox stream: same distribution, another sequence - masks are tested for
// their rate and forward / backward consistency, gradients against autograd with the exported mask.)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // "lowbias32": two multiply-xorshift rounds, 32-bit arithmetic only
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint32_t idx) {
    // ONE round over (index ^ low seed word), the high word folded in after it: 7 VALU instructions per element (a second
    // round: 14; a 64-bit finaliser: ~30).  `seed` here is the launcher's splitmix64 of the caller's seed (mix_seed below),
    // so consecutive caller seeds - one per layer - give unrelated words: two layers' masks are the same hash values at
    // indices a random 32-bit XOR apart, not shifted copies of each other.  The element index enters modulo 2^32.
    return mix32(idx ^ (uint32_t)seed) ^ (uint32_t)(seed >> 32);
}
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint32_t idx, uint32_t thresh) { return drop_hash(seed, idx) >= thresh; }

// Seed source (ispk_set_dropout_seed_source): while the process has one, every dropout kernel launched from any thread folds the 64-bit
// word at that DEVICE address into its launch seed when it RUNS - a captured training step then draws fresh masks on every
// replay (the host rewrites the word between replays) although its launch arguments are frozen.
const uint64_t* ispk_seed_source();
__device__ __forceinline__ uint64_t run_seed(uint64_t seed, const uint64_t* __restrict__ src) {
    if (!src) return seed;
    uint64_t z = (seed ^ src[0]) + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static inline uint64_t mix_seed(uint64_t z) {   // splitmix64 finaliser: the kernels' two seed words from the caller's seed
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static inline uint32_t drop_thresh(float p) {   // keep when hash >= thresh; 0 = dropout off
    if (!(p > 0.f)) return 0u;
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 4294967295u : (t < 1.0 ? 1u : (uint32_t)t);
}
