// decode_epilogue.h -- epilogue arithmetic shared by the decode GEMV kernels (k_decode.hip, k_stream.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uzu_math.h"

namespace uzu {
namespace k {

// ActivationType::activate for T = bf16 (activation_type.rs:16-42): evaluated in f32, rounded to bf16; `exp_tab` = kExp2fTab or a
// copy of it in LDS
__device__ __forceinline__ float act_bf16(uint32_t act, float x, const uint64_t* exp_tab) {
    switch (act) {
    case 0: return round_bf16(x / (1.0f + expf_glibc_tab(-1.0f * x, exp_tab)));
    case 1: return round_bf16(0.5f * x * (1.0f + tanhf(0.7978846f * (x + 0.044715f * x * x * x))));
    case 2: return round_bf16(0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)));
    case 3: return x;
    default: return x > 20.0f ? x : round_bf16(logf_glibc(1.0f + expf_glibc_tab(x, exp_tab)));
    }
}

} // namespace k
} // namespace uzu
