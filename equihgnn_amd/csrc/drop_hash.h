// Dropout decisions as a hash of (seed, element index): no mask tensor, recomputed by every pass that needs it (forward,
// backward, and the fused Linear + dropout + frame-mean epilogue of gemm_x6.hip), graph-replay safe (the seed lives in
// device memory).
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

// Dropout decision of element i: 32-bit avalanche (the murmur3 finaliser) of the element index, keyed by the 64-bit
// seed.  (Round 1 used the 64-bit splitmix finaliser per ELEMENT: ~25 instructions with its 64-bit multiplies -- on the
// [E * 8, 256] frame tensors the hash, not the memory pass, bounded drop_mean and the fused hidden-layer kernels; this
// one is 8.)  The seed enters through a key that is derived ONCE per thread by the full 64-bit mix: an XOR offset of
// the index AND the first multiplier of the avalanche.  With the seed only XOR-ed in before / after a fixed avalanche
// (round 2) the masks of two steps, or of two dropout sites, were XOR-translates of one fixed pattern; a seed-dependent
// odd multiplier makes them different functions of the index at no cost per element.
struct DropKey {
    uint32_t x, m, a;   // index offset, odd multiplier, addend of the high index word
};
__device__ __forceinline__ DropKey drop_key(uint64_t seed) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;            // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    uint64_t y = z + 0x9E3779B97F4A7C15ull;
    y = (y ^ (y >> 30)) * 0xBF58476D1CE4E5B9ull;
    y = (y ^ (y >> 27)) * 0x94D049BB133111EBull;
    y ^= y >> 31;
    return DropKey{(uint32_t)z, (uint32_t)(z >> 32) | 1u, (uint32_t)y | 1u};
}
// One avalanche decides TWO elements: the hash of the element pair i >> 1 gives its low 16 bits to the even element and
// its high 16 bits to the odd one (the keep probability is quantised to 2^-16).  The kernels handle elements in aligned
// pairs / float4 groups, so the per-element cost of the hash halves (round 4: it was ~10 of the 60 / 106 VALU
// instructions per hidden unit of the frame hidden layer, forward / backward, and a third of the frame-mean epilogue).
__device__ __forceinline__ uint32_t drop_hash(const DropKey& key, uint64_t pair) {
    uint32_t h = ((uint32_t)pair ^ key.x) + (uint32_t)(pair >> 32) * key.a;
    h ^= h >> 16; h *= key.m;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
// Host side: the 32-bit threshold handed to the kernels.  Only its high 16 bits are compared (two decisions per hash), so
// the drop probability IS round(p * 2^16) / 2^16: rounded to nearest (truncation biased every site low by up to 2^-16) and
// never 0 for p > 0 (a tiny p used to drop nothing and still rescale by 1 / (1 - p)).  The kept elements are scaled by
// exactly 1 / (1 - p), as torch.nn.functional.dropout does (fa_former_layer.py:20-21 uses nn.Dropout); the expectation is
// off by at most 2^-17 / (1 - p) relative, below fp32 rounding of the sums it enters.
__host__ __device__ static inline uint32_t drop_threshold(float p) {
    if (!(p > 0.f)) return 0u;
    long t16 = (long)((double)p * 65536.0 + 0.5);
    t16 = t16 < 1 ? 1 : (t16 > 65535 ? 65535 : t16);
    return (uint32_t)t16 << 16;
}
// Host side: the scale of the kept elements.  1 / (1 - p) as torch.nn.functional.dropout -- except at the ends of the range
// the 16-bit decision cannot represent: p >= 1 drops everything (nn.Dropout(p=1) returns zeros; the 2^-16 of the elements the
// clamped threshold still keeps are multiplied by 0, not by 1 / 0), and a p whose threshold was clamped to 65535 / 65536
// rescales by the REALISED keep probability 2^-16, so the expectation stays that of the input.
__host__ __device__ static inline float drop_inv_keep(float p) {
    if (!(p > 0.f)) return 1.f;
    if (p >= 1.f) return 0.f;
    return ((double)p * 65536.0 + 0.5 >= 65536.0) ? 65536.f : 1.0f / (1.0f - p);
}
// keep-scale of element i: 0 (dropped) or 1 / (1 - p); threshold = drop_threshold(p)
__device__ __forceinline__ float keep_scale(const DropKey& key, uint64_t i, uint32_t threshold, float inv_keep) {
    const uint32_t h = drop_hash(key, i >> 1);
    return (((i & 1) ? (h >> 16) : (h & 0xffffu)) >= (threshold >> 16)) ? inv_keep : 0.f;
}
// the same for the aligned pair (i, i + 1).  CONTRACT: i is EVEN (the pair shares one hash: an odd i would give element i the
// decision of i - 1's partner); every caller indexes float2 / float4 groups of rows whose length is a multiple of 4.
__device__ __forceinline__ void keep_scale2(const DropKey& key, uint64_t i, uint32_t threshold, float inv_keep, float& k0,
                                            float& k1) {
    const uint32_t h = drop_hash(key, i >> 1), t16 = threshold >> 16;
    k0 = (h & 0xffffu) >= t16 ? inv_keep : 0.f;
    k1 = (h >> 16) >= t16 ? inv_keep : 0.f;
}
// ... and for an aligned float4.  CONTRACT: i is a MULTIPLE OF 4.  v *= keep, two hashes
__device__ __forceinline__ void keep_scale4(const DropKey& key, uint64_t i, uint32_t threshold, float inv_keep, float4& v) {
    float k0, k1, k2, k3;
    keep_scale2(key, i, threshold, inv_keep, k0, k1);
    keep_scale2(key, i + 2, threshold, inv_keep, k2, k3);
    v.x *= k0; v.y *= k1; v.z *= k2; v.w *= k3;
}
