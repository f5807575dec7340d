// k_matmul.hip -- MatmulKernel for gfx950: D[m,n] = ab_scale * sum_k A[m,k] * deq(B)[n,k] (+D)(+bias)(soft-cap).
//
// Reference semantics: BU/cpu/kernel/matmul/kernel.rs:164-293 (B codes `[N,K]` row-major, low nibble
// first, per-group `scale*q + bias | -scale*zp | -scale*2^(bits-1)`), dispatch policy of the Metal
// backend BU/metal/kernel/matmul/mod.rs:67-102 (GEMV for small M).  Nothing else is shared with the
// Metal shaders: this is a wave64 design.
//
// gemv_q_kernel (decode, HBM-bound)
//   * one 16-byte chunk of packed codes (32 int4 / 16 int8 weights) per lane per load: a wave's 64
//     lanes fetch 1 KiB of contiguous codes -> fully coalesced global_load_dwordx4 straight to VGPRs
//     (no LDS round trip: every weight byte is used exactly once, cdna guide "GEMV / M<=16" row).
//   * `lpr` (power of two) lanes cooperate on one output row; 64/lpr rows per wave pass, R passes
//     unrolled so a lane has R independent 16-B loads in flight per k-step.
//   * fused dequant in the grouped form  acc += scale * sum(q*x) + offset * sum(x)  (SURVEY.md H6):
//     integer codes go through v_cvt_f32_ubyteN (exact), products are exact f32, f32 accumulation.
//   * butterfly reduction over the lpr lanes (fixed tree => deterministic), epilogue by lane 0.
// matmul_ref_kernel: one thread per output, the reference's loop order.  Used for shapes the fast path
//   does not cover (full-precision B, K not a multiple of the chunk) and, with UZU_HIP_EXACT=1, for every
//   matmul: it reproduces the CPU path bit for bit and is the tool that separates "reduction order"
//   differences from real bugs.
#include <stdlib.h>

#include "device_utils.h"
#include "gemv_core.h"
#include "kernels.h"
#include "kernels_decode.h"

namespace uzu {
namespace k {

// epilogue in the reference's order (kernel.rs:281-292)
__device__ __forceinline__ void epilogue_store(const MatmulParams& p, uint32_t row, uint32_t col, float accumulator) {
    const size_t output_index = (size_t)row * p.n + col;
    float value = p.ab_scale * accumulator;
    if (p.accumulate) value += ldt(p.d, p.d_dt, output_index);
    if (p.bias) value += ldt(p.bias, p.w_dt, col);
    if (p.has_soft_cap) value = p.soft_cap * tanhf(value / p.soft_cap);
    stt(p.d, p.d_dt, output_index, value);
}

// ------------------------------------------------------------------------------------------------
// Reference-order kernel: one thread per output element.
__global__ void __launch_bounds__(256) matmul_ref_kernel(MatmulParams p, uint32_t b_transpose, uint32_t ld) {
    const size_t total = (size_t)p.m * p.n;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const uint32_t row = idx / p.n, col = idx % p.n;
    const size_t k = p.k;
    const size_t b_col = p.gather ? p.gather[(size_t)row * p.n + col] : col;
    const bool quant = p.b_kind != UZU_MATMUL_B_FULL_PRECISION;
    const uint32_t bits = p.bits;
    const size_t num_groups_k = quant ? (k + p.group_size - 1) / p.group_size : 0;
    const size_t zero_point_stride = bits == 4 ? (num_groups_k + 1) / 2 : num_groups_k;
    const size_t pack_factor = bits == 4 ? 8 : 4;
    float accumulator = 0.0f;
    for (size_t inner = 0; inner < k; ++inner) {
        const float a_value = ldt(p.a, p.a_dt, (size_t)row * k + inner);
        float b_value;
        if (!quant) {
            const size_t index = b_transpose ? b_col * ld + inner : inner * ld + b_col;
            b_value = ldt(p.b, p.w_dt, index);
        } else {
            const size_t weight_linear_index = b_col * k + inner;
            const size_t word_index = weight_linear_index / pack_factor;
            const uint32_t bit_offset = (uint32_t)(weight_linear_index % pack_factor) * bits;
            const uint8_t* wb = (const uint8_t*)p.b + word_index * 4;
            const uint32_t word = (uint32_t)wb[0] | ((uint32_t)wb[1] << 8) | ((uint32_t)wb[2] << 16) | ((uint32_t)wb[3] << 24);
            uint32_t weight_code = (word >> bit_offset) & ((1u << bits) - 1u);
            if (p.signed_codes) weight_code ^= 1u << (bits - 1);
            const float quantized_value = (float)weight_code;
            const size_t group_index = inner / p.group_size;
            const float scale = ldt(p.scales, p.w_dt, b_col * num_groups_k + group_index);
            float bias_term;
            if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) {
                float zp;
                if (bits == 4) {
                    const uint8_t byte_value = p.zero_points[b_col * zero_point_stride + (group_index >> 1)];
                    zp = (group_index & 1) == 0 ? (float)(byte_value & 0x0F) : (float)((byte_value >> 4) & 0x0F);
                } else {
                    zp = (float)p.zero_points[b_col * zero_point_stride + group_index];
                }
                bias_term = -scale * zp;
            } else if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) {
                bias_term = ldt(p.biases, p.w_dt, b_col * num_groups_k + group_index);
            } else {
                bias_term = -scale * (float)(1u << (bits - 1));
            }
            b_value = scale * quantized_value + bias_term;
        }
        accumulator += a_value * b_value;
    }
    epilogue_store(p, row, col, accumulator);
}

// Reference-order kernel, vectorised: the SAME arithmetic per element in the SAME order as matmul_ref_kernel (one thread per output,
// `b_value = scale * code + bias_term; accumulator += a_value * b_value` for inner = 0 .. k - 1, no contraction) -- so bit-identical
// results -- but the memory side is a 16-byte vector of codes per 32 (int4) / 16 (int8) elements, the group's scale / bias once per
// vector, and a wave-uniform activation row (a wave = 64 consecutive columns of ONE row: the activation loads are broadcasts).  Round 5:
// matmul_ref_kernel issued ~6 dependent byte / halfword loads per element; a reference-order 4096-token Llama-3-8B prefill took 257 s
// and a decode step 0.8 s, which priced reference-order mode out of configuration-scale parity tests and of a parity census.
// Covers quantised B, bf16 / f32 activations and scales, no gather; everything else stays on matmul_ref_kernel.
template <int BITS, class TA, class TW, int MR>
__global__ void __launch_bounds__(256) matmul_ref_vec_kernel(MatmulParams p) {
    // a thread owns column `col` of MR consecutive rows (MR = 4 for prefill-sized M: the dequantised b_value of an element is computed once
    // and used by four accumulators; the rows' activation values are wave-uniform -- the row index comes from blockIdx alone)
    constexpr uint32_t EPV = 128 / BITS; // elements per 16-byte vector of codes
    const uint32_t col = blockIdx.x * 256 + threadIdx.x;
    const uint32_t row0 = blockIdx.y * MR;
    const bool live = col < p.n;
    const uint32_t b_col = live ? col : p.n - 1; // clamped: computed, never stored
    const uint32_t k = p.k, group_size = p.group_size;
    const size_t num_groups_k = (k + group_size - 1) / group_size;
    const size_t zero_point_stride = BITS == 4 ? (num_groups_k + 1) / 2 : num_groups_k;
    const uint4* codes = (const uint4*)((const uint8_t*)p.b + (size_t)b_col * k * BITS / 8);
    const TA* a[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) a[r] = (const TA*)p.a + (size_t)(row0 + r < p.m ? row0 + r : p.m - 1) * k; // clamped rows: computed, never stored
    const TW* scales = (const TW*)p.scales + (size_t)b_col * num_groups_k;
    const TW* biases = p.biases ? (const TW*)p.biases + (size_t)b_col * num_groups_k : nullptr;
    const uint32_t flip = p.signed_codes ? 1u << (BITS - 1) : 0u;
    float accumulator[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) accumulator[r] = 0.0f;
    for (uint32_t k0 = 0; k0 < k; k0 += EPV) {
        const uint4 cv = codes[k0 / EPV];
        const uint32_t words[4] = {cv.x, cv.y, cv.z, cv.w};
        const size_t group_index = k0 / group_size; // group_size % EPV == 0: one group per vector
        const float scale = ld<TW>(scales, group_index);
        float bias_term;
        if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) {
            float zp;
            if (BITS == 4) {
                const uint8_t byte_value = p.zero_points[(size_t)b_col * zero_point_stride + (group_index >> 1)];
                zp = (group_index & 1) == 0 ? (float)(byte_value & 0x0F) : (float)((byte_value >> 4) & 0x0F);
            } else {
                zp = (float)p.zero_points[(size_t)b_col * zero_point_stride + group_index];
            }
            bias_term = -scale * zp;
        } else if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) {
            bias_term = ld<TW>(biases, group_index);
        } else {
            bias_term = -scale * (float)(1u << (BITS - 1));
        }
#pragma unroll
        for (uint32_t j = 0; j < EPV; ++j) {
            const uint32_t weight_code = ((words[j * BITS / 32] >> ((j * BITS) % 32)) & ((1u << BITS) - 1u)) ^ flip;
            const float quantized_value = (float)weight_code;
            const float b_value = scale * quantized_value + bias_term;
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                const float a_value = ld<TA>(a[r], k0 + j);
                accumulator[r] += a_value * b_value;
            }
        }
    }
    if (live)
#pragma unroll
        for (int r = 0; r < MR; ++r)
            if (row0 + r < p.m) epilogue_store(p, row0 + r, col, accumulator[r]);
}
static bool matmul_ref_vec_supported(const MatmulParams& p) {
    if (p.b_kind == UZU_MATMUL_B_FULL_PRECISION || p.gather || (p.bits != 4 && p.bits != 8)) return false;
    const uint32_t epv = 128 / p.bits;
    if (p.k % epv || p.group_size % epv || (uintptr_t)p.b % 16 || ((size_t)p.k * p.bits / 8) % 16) return false;
    if (!(p.a_dt == UZU_BF16 || p.a_dt == UZU_F32) || !(p.w_dt == UZU_BF16 || p.w_dt == UZU_F32)) return false;
    return tune_env("exact_scalar") == nullptr; // UZU_EXACT_SCALAR=1: the element-by-element kernel (A/B of the two: tests)
}
template <int BITS, int MR> static uzu_status launch_ref_vec_r(hipStream_t s, const MatmulParams& p) {
    const dim3 grid((p.n + 255) / 256, (p.m + MR - 1) / MR);
    return launch_check([&] {
        if (p.a_dt == UZU_BF16 && p.w_dt == UZU_BF16) hipLaunchKernelGGL((matmul_ref_vec_kernel<BITS, bf16_t, bf16_t, MR>), grid, dim3(256), 0, s, p);
        else if (p.a_dt == UZU_BF16) hipLaunchKernelGGL((matmul_ref_vec_kernel<BITS, bf16_t, float, MR>), grid, dim3(256), 0, s, p);
        else if (p.w_dt == UZU_BF16) hipLaunchKernelGGL((matmul_ref_vec_kernel<BITS, float, bf16_t, MR>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((matmul_ref_vec_kernel<BITS, float, float, MR>), grid, dim3(256), 0, s, p);
    }, "matmul_ref_vec");
}
template <int BITS> static uzu_status launch_ref_vec(hipStream_t s, const MatmulParams& p) {
    return p.m >= 4 ? launch_ref_vec_r<BITS, 4>(s, p) : launch_ref_vec_r<BITS, 1>(s, p);
}

// ------------------------------------------------------------------------------------------------
// Fast quantised GEMV (lane mapping and arithmetic: gemv_core.h).
template <class TA> __device__ __forceinline__ void load_x32(const TA* a, size_t e, float (&xf)[32]);
template <> __device__ __forceinline__ void load_x32<bf16_t>(const bf16_t* a, size_t e, float (&xf)[32]) { load32_bf16((const uint16_t*)a + e, xf); }
template <> __device__ __forceinline__ void load_x32<float>(const float* a, size_t e, float (&xf)[32]) { load32_f32(a + e, xf); }

template <class TW, class TA, int BITS, int MT, int R>
__global__ void __launch_bounds__(256) gemv_q_kernel(MatmulParams p, int lpr_log2) {
    using Codes = typename CodesT<BITS>::type;
    constexpr int STEP_BYTES = 4 * BITS; // code bytes per 32-element step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpr = 1 << lpr_log2, rpw = 64 >> lpr_log2;
    const int sl = lane & (lpr - 1), rsub = lane >> lpr_log2;
    const uint32_t row_base = (blockIdx.x * 4 + wave) * (uint32_t)(R * rpw);
    const uint32_t m0 = blockIdx.y * MT;
    const uint32_t C = p.k / 32;
    const size_t row_bytes = (size_t)p.k * BITS / 8;
    const uint32_t G = (p.k + p.group_size - 1) / p.group_size;
    const uint32_t zp_stride = BITS == 4 ? (G + 1) / 2 : G;
    const uint32_t flip = p.signed_codes ? (BITS == 4 ? 0x88888888u : 0x80808080u) : 0u;

    uint32_t rows[R];
    size_t brow[R];
    bool valid[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rows[r] = row_base + r * rpw + rsub;
        valid[r] = rows[r] < p.n;
        const uint32_t rr = valid[r] ? rows[r] : 0;
        brow[r] = p.gather ? p.gather[(size_t)m0 * p.n + rr] : rr;
    }
    float acc[MT][R];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[mi][r] = 0.f;

    const TW* scales = (const TW*)p.scales;
    const TW* biases = (const TW*)p.biases;
    const TA* a = (const TA*)p.a;
    const uint8_t* bcodes = (const uint8_t*)p.b;

    for (uint32_t c = sl; c < C; c += lpr) {
        Codes w[R];
        float sc[R], of[R];
        const uint32_t grp = (c * 32) / p.group_size;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            load_codes(w[r], bcodes + brow[r] * row_bytes + (size_t)c * STEP_BYTES);
            sc[r] = ld(scales, brow[r] * G + grp);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) {
                of[r] = ld(biases, brow[r] * G + grp);
            } else if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) {
                uint32_t zp;
                if (BITS == 4) {
                    const uint8_t byte_value = p.zero_points[brow[r] * zp_stride + (grp >> 1)];
                    zp = (grp & 1) ? (byte_value >> 4) : (byte_value & 0x0F);
                } else {
                    zp = p.zero_points[brow[r] * zp_stride + grp];
                }
                of[r] = -sc[r] * (float)zp;
            } else {
                of[r] = -sc[r] * (float)(1u << (BITS - 1));
            }
            if (flip) flip_codes(w[r], flip);
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            if (m0 + mi < p.m) {
                float xf[32];
                load_x32<TA>(a, (size_t)(m0 + mi) * p.k + (size_t)c * 32, xf);
                const float xsum = sum32(xf);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float dq = dot32(w[r], xf);
                    acc[mi][r] = fmaf(sc[r], dq, fmaf(of[r], xsum, acc[mi][r]));
                }
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float v = row_sum_rt(acc[mi][r], lpr);
            if (sl == 0 && valid[r] && m0 + mi < p.m) epilogue_store(p, m0 + mi, rows[r], v);
        }
}

template <class TW, class TA, int BITS, int MT>
static uzu_status launch_gemv_r(hipStream_t s, const MatmulParams& p, int lpr_log2, int R, uint32_t m_tiles) {
    const int rpw = 64 >> lpr_log2;
    const uint32_t rows_per_wg = 4 * R * rpw;
    const dim3 grid((p.n + rows_per_wg - 1) / rows_per_wg, m_tiles);
    switch (R) {
    case 1: return launch_check([&] { hipLaunchKernelGGL((gemv_q_kernel<TW, TA, BITS, MT, 1>), grid, dim3(256), 0, s, p, lpr_log2); }, "gemv_q");
    case 2: return launch_check([&] { hipLaunchKernelGGL((gemv_q_kernel<TW, TA, BITS, MT, 2>), grid, dim3(256), 0, s, p, lpr_log2); }, "gemv_q");
    default: return launch_check([&] { hipLaunchKernelGGL((gemv_q_kernel<TW, TA, BITS, MT, 4>), grid, dim3(256), 0, s, p, lpr_log2); }, "gemv_q");
    }
}

static const char* gemv_label(int bits, int mt, int R) {
    static const char* names[2][3][3] = {
        {{"gemv_q4_m1_r1", "gemv_q4_m1_r2", "gemv_q4_m1_r4"}, {"gemv_q4_m2_r1", "gemv_q4_m2_r2", "gemv_q4_m2_r4"}, {"gemv_q4_m4_r1", "gemv_q4_m4_r2", "gemv_q4_m4_r4"}},
        {{"gemv_q8_m1_r1", "gemv_q8_m1_r2", "gemv_q8_m1_r4"}, {"gemv_q8_m2_r1", "gemv_q8_m2_r2", "gemv_q8_m2_r4"}, {"gemv_q8_m4_r1", "gemv_q8_m4_r2", "gemv_q8_m4_r4"}}};
    return names[bits == 8][mt == 1 ? 0 : mt == 2 ? 1 : 2][R == 1 ? 0 : R == 2 ? 1 : 2];
}

template <class TW, class TA, int BITS>
static uzu_status launch_gemv(hipStream_t s, const MatmulParams& p, int num_cus, const char** variant) {
    const int lpr_log2 = gemv_lpr_log2(p.k);
    const int rpw = 64 >> lpr_log2;
    // rows per wave: enough waves to fill the chip (>= 8 waves per CU) before unrolling rows per lane
    const uint32_t target_waves = (uint32_t)num_cus * 8;
    int R = 4;
    while (R > 1 && (p.n + (uint32_t)(R * rpw) - 1) / (uint32_t)(R * rpw) < target_waves) R >>= 1;
    const int mt = (p.gather || p.m == 1) ? 1 : (p.m == 2 ? 2 : 4);
    if (variant) *variant = gemv_label(BITS, mt, R);
    if (mt == 1) return launch_gemv_r<TW, TA, BITS, 1>(s, p, lpr_log2, R, p.m);
    if (mt == 2) return launch_gemv_r<TW, TA, BITS, 2>(s, p, lpr_log2, R, 1);
    return launch_gemv_r<TW, TA, BITS, 4>(s, p, lpr_log2, R, (p.m + 3) / 4);
}

static int g_exact = -1;
bool exact_mode() {
    if (g_exact < 0) {
        const char* e = getenv("UZU_HIP_EXACT");
        g_exact = (e && e[0] == '1') ? 1 : 0;
    }
    return g_exact == 1;
}
void set_exact_matmul(bool enabled) { g_exact = enabled ? 1 : 0; }

size_t matmul_algorithmic_bytes(const MatmulParams& p) {
    const size_t wsz = p.w_dt == UZU_F32 ? 4 : 2, asz = p.a_dt == UZU_F32 ? 4 : 2, dsz = p.d_dt == UZU_F32 ? 4 : 2;
    size_t b = (size_t)p.m * p.k * asz + (size_t)p.m * p.n * dsz;
    if (p.b_kind == UZU_MATMUL_B_FULL_PRECISION) return b + (size_t)p.n * p.k * wsz;
    const size_t groups = (p.k + p.group_size - 1) / p.group_size;
    b += (size_t)p.n * p.k * p.bits / 8 + (size_t)p.n * groups * wsz;
    if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) b += (size_t)p.n * groups * wsz;
    if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) b += (size_t)p.n * (p.bits == 4 ? (groups + 1) / 2 : groups);
    return b;
}

uzu_status matmul(hipStream_t s, const MatmulParams& p, int num_cus, const char** variant) {
    if (variant) *variant = "matmul_ref";
    if (p.m == 0 || p.n == 0) return UZU_OK;
    const bool quant = p.b_kind != UZU_MATMUL_B_FULL_PRECISION;
    if (quant && p.bits != 4 && p.bits != 8) {
        set_error("matmul: unsupported code width %u", p.bits);
        return UZU_ERR_UNSUPPORTED;
    }
    if (quant && p.group_size == 0) {
        set_error("matmul: group size must be non-zero");
        return UZU_ERR_UNSUPPORTED;
    }
    const uint32_t wpc = quant ? 32 : 1; // a lane step covers 32 K elements (gemv_core.h)
    const bool aligned = ((uintptr_t)p.b % 16 == 0) && ((uintptr_t)p.a % 16 == 0);
    const bool fast = quant && !exact_mode() && aligned && p.k % wpc == 0 && p.group_size % wpc == 0 &&
                      (p.w_dt == p.a_dt) && (p.w_dt == UZU_BF16 || p.w_dt == UZU_F32);
    if (!fast) {
        if (matmul_ref_vec_supported(p)) return p.bits == 4 ? launch_ref_vec<4>(s, p) : launch_ref_vec<8>(s, p);
        const size_t total = (size_t)p.m * p.n;
        return launch_check([&] {
            hipLaunchKernelGGL(matmul_ref_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, p, 1u, p.k);
        }, "matmul_ref");
    }
    // M == 1 with the plain epilogue: the decode GEMV of k_decode.hip (same lane mapping and arithmetic as gemv_q_kernel
    // below, i.e. bit-identical results; ping-pong prefetch, persistent grid for big matrices)
    if (p.m == 1 && p.w_dt == UZU_BF16 && p.d_dt == UZU_BF16 && !p.signed_codes && p.ab_scale == 1.0f && !p.accumulate && !p.has_soft_cap && !p.gather && !p.act_mul &&
        (p.group_size & (p.group_size - 1)) == 0 && (uint64_t)p.n * p.k * p.bits / 8 < (1ull << 32)) {
        DecGemvParams q{};
        q.w[0] = (const uint8_t*)p.b, q.scales[0] = (const uint16_t*)p.scales, q.biases[0] = (const uint16_t*)p.biases, q.zp[0] = p.zero_points;
        q.out_bias[0] = (const uint16_t*)p.bias, q.out[0] = (uint16_t*)p.d, q.n[0] = p.n;
        q.k = p.k, q.bits = p.bits, q.group_size = p.group_size, q.b_kind = p.b_kind, q.x = (const uint16_t*)p.a;
        if (variant) *variant = p.bits == 4 ? "gemv_dec<4>" : "gemv_dec<8>";
        return gemv_dec(s, q, num_cus, nullptr);
    }
    if (gemv_rows_mfma_supported(p)) { // a handful of rows (speculative verify passes, prefill tails): one pass over the weights on the matrix cores
        if (variant) *variant = "gemv_rows_mfma";
        return gemv_rows_mfma(s, p);
    }
    if (gemm_q_mfma_supported(p)) { // prefill-sized M: bf16 matrix cores (k_gemm.hip)
        if (variant) *variant = p.bits == 4 ? "gemm_q_mfma<4>" : "gemm_q_mfma<8>";
        return gemm_q_mfma(s, p, num_cus);
    }
    if (p.w_dt == UZU_BF16) {
        if (p.bits == 4) return launch_gemv<bf16_t, bf16_t, 4>(s, p, num_cus, variant);
        return launch_gemv<bf16_t, bf16_t, 8>(s, p, num_cus, variant);
    }
    if (p.bits == 4) return launch_gemv<float, float, 4>(s, p, num_cus, variant);
    return launch_gemv<float, float, 8>(s, p, num_cus, variant);
}

bool matmul_act_mul_supported(hipStream_t s, const MatmulParams& p, int num_cus) {
    MatmulParams q = p;
    q.act_mul = 1;
    if (exact_mode() || q.b_kind == UZU_MATMUL_B_FULL_PRECISION || (uintptr_t)q.b % 16 || (uintptr_t)q.a % 16) return false;
    if (gemv_rows_mfma_supported(q)) return true; // a handful of rows: the few-rows kernel carries the epilogue too (no workspace: fine under capture)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false; // no workspace during capture
    return gemm_q_mfma_supported(q) && gemm_q_mfma128_supported(q, num_cus);
}

// full-precision B with explicit layout (b_transpose / leading dimension) -- reference-order kernel
uzu_status matmul_full_precision(hipStream_t s, const MatmulParams& p, uint32_t b_transpose, uint32_t ld) {
    const size_t total = (size_t)p.m * p.n;
    if (!total) return UZU_OK;
    return launch_check([&] {
        hipLaunchKernelGGL(matmul_ref_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, p, b_transpose, ld);
    }, "matmul_ref");
}

} // namespace k
} // namespace uzu

extern "C" void uzu_hip_set_exact_matmul(int32_t enabled) { uzu::k::set_exact_matmul(enabled != 0); }
// the whole reference-order mode: every reduction kernel, not only the matmul (same switch; the older name is kept)
extern "C" void uzu_hip_set_exact(int32_t enabled) { uzu::k::set_exact_matmul(enabled != 0); }
