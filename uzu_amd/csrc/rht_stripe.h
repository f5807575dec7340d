// rht_stripe.h -- one 32-element stripe through a randomised Hadamard transform IN REGISTERS, exactly as activation_transform_kernel does it across
// 32 lanes (k_activation_transform.hip::hadamard32, mod.rs:27-47): strides 1, 2, 4, 8, 16, the lower element keeps a + b, the upper gets a - b,
// then 1/sqrt(32); the sign factors before the butterfly for InputRht, after it for OutputRht; the result rounded to bf16.  `bits`: bit i = the
// factor of element i is -1 (a multiplication by -1.0f and a negation are the same operation).  Used by the fused decode step's GEMV prologue
// (k_decode.hip, PRO == 3) and by the one-thread-per-stripe row kernels of k_elementwise.hip.
#pragma once
#include "device_utils.h"

namespace uzu {
namespace k {

// the same butterfly across 32 lanes (lane l = element l of the stripe; both halves of a wave run one stripe each): activation_transform_kernel's form
__device__ __forceinline__ float hadamard32(float v, int l) {
#pragma unroll
    for (int stride = 1; stride < 32; stride <<= 1) {
        const float other = __shfl_xor(v, stride, 64);
        v = (l & stride) ? other - v : v + other; // lower lane keeps a + b, upper lane gets a - b (a = the lower lane's value)
    }
    return v * (1.0f / sqrtf(32.0f));
}

template <bool INPUT>
__device__ __forceinline__ void rht_stripe_regs(float (&a)[32], uint32_t bits) {
    if (INPUT) {
#pragma unroll
        for (int i = 0; i < 32; ++i) a[i] = (bits >> i) & 1u ? -a[i] : a[i];
    }
#pragma unroll
    for (int stride = 1; stride < 32; stride <<= 1) {
#pragma unroll
        for (int l = 0; l < 32; ++l) {
            if (l & stride) continue;
            const float lo = a[l], hi = a[l | stride];
            a[l] = lo + hi, a[l | stride] = lo - hi;
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        float v = a[i] * (1.0f / sqrtf(32.0f));
        if (!INPUT) v = (bits >> i) & 1u ? -v : v;
        a[i] = round_bf16(v);
    }
}

// The same transform with a stripe SPREAD over 32 / E consecutive, aligned lanes whose threads own E consecutive elements each (E in {4, 8, 16, 32}:
// the Normalization prologue's element mapping, thread t <-> elements [t E, t E + E)): strides below E run in the thread's registers, the others with
// xor shuffles; the butterfly order (1, 2, 4, 8, 16; the lower element keeps a + b, the upper gets a - b), the 1 / sqrt(32) and the place of the sign
// factors are rht_stripe_regs': the same f32 operations in the same order per element => bit-identical values, without the LDS round trip, the two
// barriers and the 32 / E-fold under-occupation of the one-thread-per-stripe form (round 5: 2.5 us per RHT prologue).  `bits` = the stripe's sign word.
template <int E>
__device__ __forceinline__ void hadamard_spread(float (&v)[E], int lane) {
#pragma unroll
    for (int stride = 1; stride < E; stride <<= 1) {
#pragma unroll
        for (int l = 0; l < E; ++l) {
            if (l & stride) continue;
            const float lo = v[l], hi = v[l | stride];
            v[l] = lo + hi, v[l | stride] = lo - hi;
        }
    }
#pragma unroll
    for (int stride = E; stride < 32; stride <<= 1) {
        const int m = stride / E;
        const bool upper = (lane & m) != 0;
#pragma unroll
        for (int l = 0; l < E; ++l) {
            const float other = __shfl_xor(v[l], m, 64);
            v[l] = upper ? other - v[l] : v[l] + other;
        }
    }
#pragma unroll
    for (int l = 0; l < E; ++l) v[l] = v[l] * (1.0f / sqrtf(32.0f));
}
template <int E, bool INPUT>
__device__ __forceinline__ void rht_spread(float (&v)[E], uint32_t bits, uint32_t first_bit, int lane) { // element i of v carries sign bit first_bit + i
    if (INPUT) {
#pragma unroll
        for (int i = 0; i < E; ++i) v[i] = (bits >> (first_bit + i)) & 1u ? -v[i] : v[i];
    }
    hadamard_spread<E>(v, lane);
#pragma unroll
    for (int i = 0; i < E; ++i) {
        float x = v[i];
        if (!INPUT) x = (bits >> (first_bit + i)) & 1u ? -x : x;
        v[i] = round_bf16(x);
    }
}

} // namespace k
} // namespace uzu
