// uzu_math.h -- scalar math shared by every kernel: bf16 <-> f32 with the `half` crate's rounding,
// and expf / logf that reproduce glibc's results so that element-wise kernels can be compared
// BIT-EXACTLY with the reference's CPU backend (Rust `f32::exp` lowers to the platform `expf`).
//
// expf: the published Szabolcs-Nagy/ARM-optimized-routines algorithm that glibc >= 2.27 ships
// (sysdeps/ieee754/flt-32/e_expf.c): N = 32 table of 2^(i/32), degree-3 polynomial, everything in
// double, one final rounding to float.  All operations are IEEE double ops, so host and gfx950 give
// identical bits.  (glibc's x86-64 ifunc build contracts the polynomial into FMAs on FMA hardware;
// we use explicit fma() to match -- the two variants differ in < 1e-8 of inputs anyway.)
// This header also compiles as plain C++ on the host (tests/test_math_host.py validates it against
// the system expf / logf over dense input sweeps).
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define UZU_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define UZU_HD static inline
#endif

namespace uzu {

UZU_HD float bits_to_f32(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
UZU_HD uint32_t f32_to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
UZU_HD double bits_to_f64(uint64_t u) {
    double f;
    memcpy(&f, &u, 8);
    return f;
}
UZU_HD uint64_t f64_to_bits(double f) {
    uint64_t u;
    memcpy(&u, &f, 8);
    return u;
}

// half 2.7 bf16::to_f32 / bf16::from_f32 (round to nearest even, NaN quieted)
UZU_HD float bf16_to_f32(uint16_t v) { return bits_to_f32(((uint32_t)v) << 16); }
UZU_HD uint16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950 has a hardware round-to-nearest-even conversion (v_cvt_pk_bf16_f32); same result as the software
    // path below for every non-NaN input (NaNs stay NaNs; tests/test_gpu_kernels.py::test_bf16_rounding_matches_host)
    const __bf16 b = (__bf16)f;
    uint16_t u;
    __builtin_memcpy(&u, &b, 2);
    return u;
#else
    const uint32_t x = f32_to_bits(f);
    if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((x >> 16) | 0x0040u);
    return (uint16_t)((x + 0x7FFFu + ((x >> 16) & 1u)) >> 16);
#endif
}
// round an f32 through bf16 (the `T::from(x)` of a bf16 kernel, value kept in an f32 register)
UZU_HD float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

#if defined(__HIPCC__) || defined(__HIP__)
__device__ static const uint64_t kExp2fTab[32] = {
#else
static const uint64_t kExp2fTab[32] = {
#endif
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};

// `tab`: the 32-entry table -- kExp2fTab, or a copy of it in LDS (decode epilogues: a table read from global memory is
// a dependent memory round trip on the critical path of a 4 us kernel)
UZU_HD float expf_glibc_tab(float x, const uint64_t* tab) {
    const uint32_t ix = f32_to_bits(x);
    const uint32_t abstop = (ix >> 20) & 0x7ff;
    if (abstop >= 0x42b) { // |x| >= 88 or NaN/Inf
        if (ix == 0xff800000u) return 0.0f;                 // exp(-inf)
        if (abstop >= 0x7f8) return x + x;                  // NaN, +inf
        if (x > 88.72283172607421875f) return bits_to_f32(0x7f800000u);   // overflow (0x1.62e42ep6)
        if (x < -103.97207641601562500f) return 0.0f;       // underflow (-0x1.9fe368p6)
    }
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
    const double Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + Shift;
    const uint64_t ki = f64_to_bits(kd);
    kd -= Shift;
    const double r = z - kd;
    uint64_t t = tab[ki & 31];
    t += ki << (52 - 5);
    const double s = bits_to_f64(t);
    z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

UZU_HD float expf_glibc(float x) { return expf_glibc_tab(x, kExp2fTab); }

// logf: glibc sysdeps/ieee754/flt-32/e_logf.c (same family: 16-entry table, degree-3 polynomial in
// double).  T[i] = {invc, logc} with c near the centre of the i-th subinterval of [0x3f330000 .. *2).
#if defined(__HIPCC__) || defined(__HIP__)
__device__ static const double kLogfTab[16][2] = {
#else
static const double kLogfTab[16][2] = {
#endif
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
};

UZU_HD float logf_glibc(float x) {
    uint32_t ix = f32_to_bits(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -bits_to_f32(0x7f800000u);       // log(0) = -inf
        if (ix == 0x7f800000u) return x;                          // log(inf)
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return bits_to_f32(0x7fc00000u); // NaN
        ix = f32_to_bits(x * 0x1p23f);                            // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (tmp >> (23 - 4)) % 16;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = kLogfTab[i][0];
    const double logc = kLogfTab[i][1];
    const double z = (double)bits_to_f32(iz);
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = __builtin_fma((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = __builtin_fma(A1, r, A2);
    y = __builtin_fma(A0, r2, y);
    y = __builtin_fma(y, r2, y0 + r);
    return (float)y;
}

// activation_type.rs:44-65: x / (1 + exp(-alpha*x)) with alpha = 1, evaluated in f32
UZU_HD float silu_f32(float x) { return x / (1.0f + expf_glibc(-1.0f * x)); }
UZU_HD float silu_f32_tab(float x, const uint64_t* tab) { return x / (1.0f + expf_glibc_tab(-1.0f * x, tab)); }

// ActivationType::activate with T = bf16 (gpu_types/activation_type.rs:16-65): the arithmetic of k_elementwise.hip::activate<bf16_t>,
// shared with the GEMM's fused GatedActMul epilogue (k_gemm128.hip)
UZU_HD float activate_bf16(uint32_t act, float x) {
    switch (act) {
    case 0: return round_bf16(x / (1.0f + expf_glibc(-1.0f * x)));
    case 1: return round_bf16(0.5f * x * (1.0f + tanhf(0.7978846f * (x + 0.044715f * x * x * x))));
    case 2: return round_bf16(0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)));
    case 3: return x;
    case 4: return x > 20.0f ? x : round_bf16(logf_glibc(1.0f + expf_glibc(x)));
    default: return x;
    }
}
// same, with the exp table parked in LDS by the caller (kExp2fTab, 32 entries): no dependent global-memory load per element
UZU_HD float activate_bf16_tab(uint32_t act, float x, const uint64_t* tab) {
    switch (act) {
    case 0: return round_bf16(x / (1.0f + expf_glibc_tab(-1.0f * x, tab)));
    case 4: return x > 20.0f ? x : round_bf16(logf_glibc(1.0f + expf_glibc_tab(x, tab)));
    default: return activate_bf16(act, x);
    }
}
} // namespace uzu
