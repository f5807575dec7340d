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
// Pair form (k_gemm128.hip): 0x4180 | u << 3 is bf16(16 + u) for an unsigned code u, so one v_and_or_b32 converts the two codes
// that sit 16 bits apart in the word: 8 codes = 4 shifts + 4 and-ors (8 VALU instead of 14), element order (u0, u4, u1, u5, u2, u6,
// u3, u7) -- the activation fragment is permuted the same way while it is staged (permute_pairs).  The constant 16 joins the
// group's offset term.
__device__ __forceinline__ u32x4_t dequant4_pairs(uint32_t w) {
    // v_and_or_b32 may read ONE scalar operand (gfx9 constant bus): the mask stays in an SGPR, the magic in a VGPR; opaque to
    // the optimiser, which would otherwise re-materialise both as literals or split the and-or
    uint32_t mask = 0x00780078u, magic = 0x41804180u;
    asm("" : "+s"(mask));
    asm("" : "+v"(magic));
    u32x4_t r;
    r.x = ((w << 3) & mask) | magic;
    r.y = ((w >> 1) & mask) | magic;
    r.z = ((w >> 5) & mask) | magic;
    r.w = ((w >> 9) & mask) | magic;
    return r;
}
// eight consecutive bf16 (a0 .. a7 in four words) -> (a0, a4, a1, a5, a2, a6, a3, a7): four v_perm_b32
__device__ __forceinline__ u32x4_v permute_pairs(u32x4_v v) {
    u32x4_v r;
    r.x = __builtin_amdgcn_perm(v.z, v.x, 0x05040100u);
    r.y = __builtin_amdgcn_perm(v.z, v.x, 0x07060302u);
    r.z = __builtin_amdgcn_perm(v.w, v.y, 0x05040100u);
    r.w = __builtin_amdgcn_perm(v.w, v.y, 0x07060302u);
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
