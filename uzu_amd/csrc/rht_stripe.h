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

} // namespace k
} // namespace uzu
