// sampling_noise.h -- the sampler's counter-based noise (encodable_block/sampling/gumbel.rs:1-81): Philox4x32-10 keyed by the row's seed,
// one 24-bit uniform per logit index, Gumbel(0, 1) = -ln(-ln u).  Shared by UnifiedSampling (k_sampling.hip) and WeaverTopChildren
// (k_speculator.hip); bit-exact against the CPU kernels (uzu_math.h: glibc's logf).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uzu_math.h"

namespace uzu {
namespace k {

__device__ __forceinline__ void philox4x32_10(uint32_t (&ctr)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        if (round) k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
        const uint32_t hi0 = __umulhi(0xD2511F53u, ctr[0]), lo0 = 0xD2511F53u * ctr[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr[2]), lo1 = 0xCD9E8D57u * ctr[2];
        const uint32_t n0 = hi1 ^ ctr[1] ^ k0, n2 = hi0 ^ ctr[3] ^ k1;
        ctr[0] = n0, ctr[1] = lo1, ctr[2] = n2, ctr[3] = lo0;
    }
}
// gumbel.rs:34-64 + revidx (gumbel.rs:66-81): noise of logit `i`
__device__ __forceinline__ float gumbel_of(uint64_t seed, uint32_t i, uint32_t vocab_size) {
    const uint32_t thread_idx = i % 1024u, block_idx = i / 1024u;
    const uint32_t offset = ((vocab_size + 4095u) / 4096u) * thread_idx + block_idx / 4u, word = block_idx % 4u;
    uint32_t ctr[4] = {offset, 0u, 0u, 0u};
    philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w = word == 0 ? ctr[0] : word == 1 ? ctr[1] : word == 2 ? ctr[2] : ctr[3];
    const uint32_t top = w >> 8;
    const float u = (float)(top > 1u ? top : 1u) * (1.0f / 16777216.0f);
    return -logf_glibc(-logf_glibc(u));
}

} // namespace k
} // namespace uzu
