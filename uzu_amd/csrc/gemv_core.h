// gemv_core.h -- the per-lane building blocks shared by the stand-alone GEMV (k_matmul.hip) and the fused decode
// GEMV (k_decode.hip).  Both kernels use the SAME lane mapping and the SAME arithmetic, so that the fused
// decode path is bit-identical to the one-kernel-per-reference-kernel path:
//
//   * a lane step covers 32 consecutive K elements: 16 bytes of int4 codes or 32 bytes of int8 codes;
//   * `lpr` (power of two) lanes cooperate on a row, lane `sl` owns steps c = sl + lpr * j;
//   * step dot product: 4 independent f32 accumulator chains (element index & 3), combined (d0+d1)+(d2+d3)
//     -- four chains instead of one give the in-order SIMD something to overlap (a single wave per SIMD is
//     the normal case for the small decode matrices);
//   * group correction in the grouped form: acc = fma(scale, dot, fma(offset, sum(x), acc));
//   * row reduction: xor butterfly over the lpr lanes, offsets lpr/2 ... 1.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uzu_math.h"

namespace uzu {
namespace k {

struct Codes4 { uint4 a; };          // 32 int4 codes
struct Codes8 { uint4 a, b; };       // 32 int8 codes
template <int BITS> struct CodesT;
template <> struct CodesT<4> { using type = Codes4; };
template <> struct CodesT<8> { using type = Codes8; };

__device__ __forceinline__ void load_codes(Codes4& c, const uint8_t* p) { c.a = *(const uint4*)p; }
__device__ __forceinline__ void load_codes(Codes8& c, const uint8_t* p) { c.a = ((const uint4*)p)[0], c.b = ((const uint4*)p)[1]; }
// streamed weights (each byte is read once per token by one CU): non-temporal loads keep them from displacing the
// activations / KV / state lines in L2 and the memory-side cache (MI355X_MICROARCH.md, row nt-weights)
#ifndef UZU_GEMV_NT
#define UZU_GEMV_NT 1
#endif
typedef uint32_t gc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load16_stream(const uint8_t* p) {
#if UZU_GEMV_NT
    const gc_u32x4 v = __builtin_nontemporal_load((const gc_u32x4*)p);
#else
    const gc_u32x4 v = *(const gc_u32x4*)p;
#endif
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void load_codes_stream(Codes4& c, const uint8_t* p) { c.a = load16_stream(p); }
__device__ __forceinline__ void load_codes_stream(Codes8& c, const uint8_t* p) { c.a = load16_stream(p), c.b = load16_stream(p + 16); }
__device__ __forceinline__ void flip_codes(Codes4& c, uint32_t m) { c.a.x ^= m, c.a.y ^= m, c.a.z ^= m, c.a.w ^= m; }
__device__ __forceinline__ void flip_codes(Codes8& c, uint32_t m) { c.a.x ^= m, c.a.y ^= m, c.a.z ^= m, c.a.w ^= m, c.b.x ^= m, c.b.y ^= m, c.b.z ^= m, c.b.w ^= m; }

// byte n of a word as f32: v_cvt_f32_ubyteN (exact)
__device__ __forceinline__ float ub0(uint32_t w) { return (float)(w & 0xFFu); }
__device__ __forceinline__ float ub1(uint32_t w) { return (float)((w >> 8) & 0xFFu); }
__device__ __forceinline__ float ub2(uint32_t w) { return (float)((w >> 16) & 0xFFu); }
__device__ __forceinline__ float ub3(uint32_t w) { return (float)(w >> 24); }

// sum_{i<32} code_i * x_i ; element i = 8*word + nibble (low nibble first)
__device__ __forceinline__ float dot32(const Codes4& c, const float (&x)[32]) {
    const uint32_t ws[4] = {c.a.x, c.a.y, c.a.z, c.a.w};
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t lo = ws[i] & 0x0F0F0F0Fu;        // codes 0,2,4,6
        uint32_t hi = (ws[i] >> 4) & 0x0F0F0F0Fu; // codes 1,3,5,7
#if defined(__HIP_DEVICE_COMPILE__)
        // keep the masked words opaque: otherwise the byte extracts below are rewritten into v_bfe_u32 of the
        // original word + v_cvt_f32_ubyte0 (2 instructions per code) instead of v_cvt_f32_ubyte0..3 (1 per code)
        asm("" : "+v"(lo));
        asm("" : "+v"(hi));
#endif
        d0 = fmaf(ub0(lo), x[8 * i + 0], d0);
        d1 = fmaf(ub0(hi), x[8 * i + 1], d1);
        d2 = fmaf(ub1(lo), x[8 * i + 2], d2);
        d3 = fmaf(ub1(hi), x[8 * i + 3], d3);
        d0 = fmaf(ub2(lo), x[8 * i + 4], d0);
        d1 = fmaf(ub2(hi), x[8 * i + 5], d1);
        d2 = fmaf(ub3(lo), x[8 * i + 6], d2);
        d3 = fmaf(ub3(hi), x[8 * i + 7], d3);
    }
    return (d0 + d1) + (d2 + d3);
}
__device__ __forceinline__ float dot32(const Codes8& c, const float (&x)[32]) {
    const uint32_t ws[8] = {c.a.x, c.a.y, c.a.z, c.a.w, c.b.x, c.b.y, c.b.z, c.b.w};
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d0 = fmaf(ub0(ws[i]), x[4 * i + 0], d0);
        d1 = fmaf(ub1(ws[i]), x[4 * i + 1], d1);
        d2 = fmaf(ub2(ws[i]), x[4 * i + 2], d2);
        d3 = fmaf(ub3(ws[i]), x[4 * i + 3], d3);
    }
    return (d0 + d1) + (d2 + d3);
}
// ---- int4 x bf16 on the packed-dot unit -------------------------------------------------------------------------
// A nibble q placed in mantissa bits 6..3 of a bf16 whose exponent field says 2^4 reads as the value 16 + q:
//   0x4180 | (q << 3)  ==  bf16(16 + q)   (exact: 16 * (1 + 8q / 128)).
// One shift + one v_and_or_b32 therefore turns a code word into a PAIR of bf16 codes (nibble s of the low half-word
// and nibble s of the high half-word, i.e. elements 8w + s and 8w + 4 + s), which v_dot2c_f32_bf16 multiplies with
// the matching pair of bf16 activations and accumulates in f32: 1.5 VALU instructions per weight instead of 2.25
// (v_cvt_f32_ubyteN + v_fma_f32 + the nibble masks), and the activation step lives in 16 registers instead of 32.
// Every product (16 + q) * x is exact (5 x 8 significant bits); the surplus 16 * sum(x) leaves through the group
// offset: acc += scale * D + (offset - 16 * scale) * sum(x).  Offset 16 rather than 128 (mantissa bits 3..0, no
// shift for one nibble in four): the accumulated value is then at most ~2x the wanted one instead of ~17x, so the
// f32 rounding noise stays at the level of the plain q * x chain (tools/sim_offset_trick.py).
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
constexpr float kQ4Offset = 16.0f;
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_v, a), __builtin_bit_cast(bf16x2_v, b), c, false);
#else
    return c;
#endif
}
// activation step for the packed dot: word 4w + s = (x[8w + s], x[8w + 4 + s]) as a bf16 pair (low half = first)
struct XPack { uint32_t v[16]; };
__device__ __forceinline__ uint32_t pack_bf16_pair(float lo, float hi) { // both bf16-representable
    return __builtin_amdgcn_perm(f32_to_bits(hi), f32_to_bits(lo), 0x07060302u);
}
__device__ __forceinline__ void xpack_from_f32(XPack& o, const float (&x)[32]) {
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int s = 0; s < 4; ++s) o.v[4 * w + s] = pack_bf16_pair(x[8 * w + s], x[8 * w + 4 + s]);
}
// 32 consecutive bf16 activations (four 16-byte words as loaded) -> packed-dot order; also returns their sum (f32,
// through the same unit: x * 1.0 pairs)
typedef uint32_t gc_raw4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float xpack_from_raw(XPack& o, const gc_raw4 (&raw)[4]) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    uint32_t ones = 0x3F803F80u;
    asm("" : "+v"(ones));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const gc_raw4 u = raw[w]; // (x0,x1) (x2,x3) (x4,x5) (x6,x7)
        o.v[4 * w + 0] = __builtin_amdgcn_perm(u.z, u.x, 0x05040100u);
        o.v[4 * w + 1] = __builtin_amdgcn_perm(u.z, u.x, 0x07060302u);
        o.v[4 * w + 2] = __builtin_amdgcn_perm(u.w, u.y, 0x05040100u);
        o.v[4 * w + 3] = __builtin_amdgcn_perm(u.w, u.y, 0x07060302u);
        s0 = dot2_bf16(o.v[4 * w + 0], ones, s0);
        s1 = dot2_bf16(o.v[4 * w + 1], ones, s1);
        s2 = dot2_bf16(o.v[4 * w + 2], ones, s2);
        s3 = dot2_bf16(o.v[4 * w + 3], ones, s3);
    }
    return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ float xpack_load(XPack& o, const uint16_t* p) {
    const uint4* src = (const uint4*)p;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    uint32_t ones = 0x3F803F80u;
    asm("" : "+v"(ones));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint4 u = src[w]; // (x0,x1) (x2,x3) (x4,x5) (x6,x7)
        o.v[4 * w + 0] = __builtin_amdgcn_perm(u.z, u.x, 0x05040100u);
        o.v[4 * w + 1] = __builtin_amdgcn_perm(u.z, u.x, 0x07060302u);
        o.v[4 * w + 2] = __builtin_amdgcn_perm(u.w, u.y, 0x05040100u);
        o.v[4 * w + 3] = __builtin_amdgcn_perm(u.w, u.y, 0x07060302u);
        s0 = dot2_bf16(o.v[4 * w + 0], ones, s0);
        s1 = dot2_bf16(o.v[4 * w + 1], ones, s1);
        s2 = dot2_bf16(o.v[4 * w + 2], ones, s2);
        s3 = dot2_bf16(o.v[4 * w + 3], ones, s3);
    }
    return (s0 + s1) + (s2 + s3);
}
// sum_{i<32} (16 + code_i) * x_i
__device__ __forceinline__ float dot32p(const Codes4& c, const XPack& x) {
    const uint32_t ws[4] = {c.a.x, c.a.y, c.a.z, c.a.w};
    uint32_t mask = 0x00780078u, magic = 0x41804180u;
    // opaque register constants: with literals the compiler emits v_and_b32 + v_or_b32 (a VOP3 cannot carry a literal)
    asm("" : "+s"(mask));
    asm("" : "+v"(magic));
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t q0 = ((ws[w] << 3) & mask) | magic; // nibbles 0 and 4
        const uint32_t q1 = ((ws[w] >> 1) & mask) | magic; // nibbles 1 and 5
        const uint32_t q2 = ((ws[w] >> 5) & mask) | magic; // nibbles 2 and 6
        const uint32_t q3 = ((ws[w] >> 9) & mask) | magic; // nibbles 3 and 7
        d0 = dot2_bf16(q0, x.v[4 * w + 0], d0);
        d1 = dot2_bf16(q1, x.v[4 * w + 1], d1);
        d2 = dot2_bf16(q2, x.v[4 * w + 2], d2);
        d3 = dot2_bf16(q3, x.v[4 * w + 3], d3);
    }
    return (d0 + d1) + (d2 + d3);
}

__device__ __forceinline__ float sum32(const float (&x)[32]) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s0 += x[4 * i], s1 += x[4 * i + 1], s2 += x[4 * i + 2], s3 += x[4 * i + 3];
    return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ float sumsq32(const float (&x)[32]) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s0 = fmaf(x[4 * i], x[4 * i], s0);
        s1 = fmaf(x[4 * i + 1], x[4 * i + 1], s1);
        s2 = fmaf(x[4 * i + 2], x[4 * i + 2], s2);
        s3 = fmaf(x[4 * i + 3], x[4 * i + 3], s3);
    }
    return (s0 + s1) + (s2 + s3);
}

// 32 consecutive bf16 / f32 activations -> f32 registers (16-byte loads)
__device__ __forceinline__ void load32_bf16(const uint16_t* p, float (&x)[32]) {
    const uint4* src = (const uint4*)p;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const uint4 u = src[v];
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[v * 8 + 2 * j] = bits_to_f32(w[j] << 16);
            x[v * 8 + 2 * j + 1] = bits_to_f32(w[j] & 0xFFFF0000u);
        }
    }
}
__device__ __forceinline__ void load32_f32(const float* p, float (&x)[32]) {
    const float4* src = (const float4*)p;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const float4 u = src[v];
        x[v * 4] = u.x, x[v * 4 + 1] = u.y, x[v * 4 + 2] = u.z, x[v * 4 + 3] = u.w;
    }
}
__device__ __forceinline__ void store32_bf16(uint16_t* p, const float (&x)[32]) { // x holds bf16-representable values
    uint4* dst = (uint4*)p;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (f32_to_bits(x[v * 8 + 2 * j]) >> 16) | (f32_to_bits(x[v * 8 + 2 * j + 1]) & 0xFFFF0000u);
        dst[v] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// v + v[lane ^ off] with compile-time patterns: DPP for 1/2/4/8, DPP row_bcast + v_readlane for 16 and 32.
// (hipcc lowers __shfl_xor to ds_bpermute_b32, ~100+ cycles of dependent latency per step.)  For a SUM the
// mirror patterns are equivalent to the xor patterns once the lower levels have been reduced, and float
// addition is commutative, so the result equals the plain xor butterfly bit for bit.
__device__ __forceinline__ float xadd1(float v) { return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float xadd2(float v) { return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float xadd4(float v) { return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); } // row_half_mirror
__device__ __forceinline__ float xadd8(float v) { return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); } // row_mirror
// level 16 (every row of 16 lanes already holds its row total): DPP row_bcast:15 hands a row's total to the next row,
// so the odd rows hold row(2i) + row(2i+1); v_readlane of lanes 31 / 63 + a select by the wave half broadcast it
// (same two operands as the xor-16 butterfly, no LDS round trip -- ds_swizzle costs ~100 cycles of dependent latency).
__device__ __forceinline__ float xadd16(float v) {
    const float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
    const float t = prev + v; // valid in rows 1 and 3
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 31));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
    return (__lane_id() & 32) ? hi : lo;
}
// level 32 (both halves already hold their 32-lane totals): DPP row_bcast:31 hands the lower total to the upper
// half, v_readlane of lane 63 broadcasts lower + upper to the whole wave (same operands as the xor butterfly).
__device__ __forceinline__ float xadd32(float v) {
    const float lower = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xF, 0xF, true));
    const float t = lower + v; // valid in lanes 32..63
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
}
// sum over the lpr (power of two) consecutive lanes that contain this lane; levels 1, 2, 4, ... lpr/2
__device__ __forceinline__ float row_sum_rt(float v, int lpr) {
    if (lpr > 1) v = xadd1(v);
    if (lpr > 2) v = xadd2(v);
    if (lpr > 4) v = xadd4(v);
    if (lpr > 8) v = xadd8(v);
    if (lpr > 16) v = xadd16(v);
    if (lpr > 32) v = xadd32(v);
    return v;
}
__device__ __forceinline__ float wave_sum_fast(float v) { return row_sum_rt(v, 64); }

// lanes per row for K elements: smallest power of two >= K/32, capped at 64
inline int gemv_lpr_log2(uint32_t k) {
    const uint32_t steps = (k + 31) / 32;
    int l = 0;
    while ((1u << l) < steps && l < 6) ++l;
    return l;
}

} // namespace k
} // namespace uzu
