// gemm_convert.h -- exact integer-code -> bf16 conversions shared by the matrix-core GEMMs (k_gemm.hip, k_gemm128.hip).
// The centred codes are exact in bf16: int4 (q - 8) / 16 through v_cvt_off_f32_i4 with SDWA byte selects, int8 q - 128
// through v_cvt_f32_i32 sext byte selects; 14 VALU operations per 8 codes (1 shift / xor + 8 converts + 4 packs).
#pragma once
#include "device_utils.h"

namespace uzu {
namespace k {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) { // v_cvt_pk_bf16_f32 (round to nearest even)
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ u32x4_t dequant4(uint32_t w) { // nibbles = two's complement of q - 8 -> bf16 (q - 8) / 16
    uint32_t h = w >> 4;
    asm volatile("" : "+v"(h));
    u32x4_t r;
    r.x = pack_bf16(__builtin_amdgcn_cvt_off_f32_i4(w & 0xFF), __builtin_amdgcn_cvt_off_f32_i4(h & 0xFF));
    r.y = pack_bf16(__builtin_amdgcn_cvt_off_f32_i4((w >> 8) & 0xFF), __builtin_amdgcn_cvt_off_f32_i4((h >> 8) & 0xFF));
    r.z = pack_bf16(__builtin_amdgcn_cvt_off_f32_i4((w >> 16) & 0xFF), __builtin_amdgcn_cvt_off_f32_i4((h >> 16) & 0xFF));
    r.w = pack_bf16(__builtin_amdgcn_cvt_off_f32_i4(w >> 24), __builtin_amdgcn_cvt_off_f32_i4(h >> 24));
    return r;
}
__device__ __forceinline__ float sbyte(uint32_t w, int i) { return (float)(int)(int8_t)((w >> (8 * i)) & 0xFFu); }
__device__ __forceinline__ u32x4_t dequant8(uint32_t w0, uint32_t w1) { // bytes = two's complement of q - 128
    u32x4_t r;
    r.x = pack_bf16(sbyte(w0, 0), sbyte(w0, 1));
    r.y = pack_bf16(sbyte(w0, 2), sbyte(w0, 3));
    r.z = pack_bf16(sbyte(w1, 0), sbyte(w1, 1));
    r.w = pack_bf16(sbyte(w1, 2), sbyte(w1, 3));
    return r;
}

} // namespace k
} // namespace uzu
