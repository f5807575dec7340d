// k_gemm.hip -- prefill MatmulKernel for gfx950: D[m,n] = ab_scale * sum_k A[m,k] * deq(B)[n,k] (+D)(+bias)(soft-cap)
// for M >= 16 rows of bf16 activations against int4 / int8 block-quantised weights, on the bf16 matrix cores.
//
// Reference semantics: BU/cpu/kernel/matmul/kernel.rs:164-293 (same as k_matmul.hip).  The design is MI355X-first:
//
//   * The centred integer codes are EXACT in bf16 (int4: (q - 8) / 16 through v_cvt_off_f32_i4 with SDWA byte selects,
//     int8: q - 128 through v_cvt_f32_i32 sext byte selects; 14 VALU ops per 8 codes), the activations are bf16
//     already, so every MFMA product is exact in f32.  Centring matters: with unsigned (or offset) codes the group
//     accumulator carries mean(code) * sum(a), which costs accuracy when it is cancelled against the bias term.  The quantisation scale never touches the
//     MFMA operands: one group of `group_size` k's accumulates into a per-group accumulator tile, and at the group
//     boundary   acc += scale[n,g] * acc_g + coef[n,g] * sum_k A[m,k in g]   runs on the VALU in f32 -- the same
//     grouped form the decode GEMV uses (gemv_core.h), so prefill and decode see the same arithmetic up to summation
//     order.  Nothing is ever rounded to a bf16 *weight*.
//   * v_mfma_f32_32x32x16_bf16; workgroup tile 64 x 64 x 64, four waves as 2 x 2, each wave one 32 x 32 block (three
//     workgroups per CU).  This is the kernel for 20 <= M < 128 and for shapes the large-tile kernel (k_gemm128.hip: 128 x 128
//     tiles, weights straight from global memory into the MFMA operand) does not cover.
//   * The k order inside an MFMA is irrelevant as long as A and B agree, so the kernel picks the order that makes the
//     weight fetch one vector per lane: lanes 0..31 take k = 8s..8s+7, lanes 32..63 take k = 32+8s..32+8s+7 at step s
//     of a 64-wide k-step.  The two waves that cover the same 32 columns convert half of the k16 steps each and share the
//     converted fragments through LDS in operand order; the activations are staged through LDS as well (double buffered,
//     144-byte row pitch => conflict-free ds_read_b128).
//   * Row sums of the activations per quant group come from the matrix cores as well (one extra MFMA per k16 step with
//     an all-ones B operand): the result lands in the accumulator layout the group fold needs.
#include <stdlib.h>

#include "device_utils.h"
#include "gemm_convert.h"
#include "kernels.h"

namespace uzu {
namespace k {


namespace {
constexpr int BK = 64;
constexpr int A_PITCH = 144; // bytes per staged activation row: 64 bf16 + 16 bytes of pad

__device__ __forceinline__ float chunk_sum(uint4 v) { // sum of 8 bf16
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s0 += bits_to_f32(w[i] << 16), s1 += bits_to_f32(w[i] & 0xFFFF0000u);
    return s0 + s1;
}
} // namespace

template <int BITS, int MB, int NB> // a wave owns MB x NB blocks of 32 x 32; four waves as 2 x 2 => workgroup tile (64 MB) x (64 NB)
__global__ void __launch_bounds__(256) gemm_q_mfma_kernel(MatmulParams p, uint32_t lds_groups) {
    constexpr int BM = 64 * MB, BN = 64 * NB;
    constexpr int PARTS = 256 / BM;        // staging threads per activation row (2 or 4)
    constexpr int CH = 8 / PARTS;          // 16-byte chunks (8 k) per staging thread per k-step
    constexpr int WV = BITS / 4; // 16-byte vectors of codes per lane per 32-column block per k-step
    __shared__ __attribute__((aligned(16))) uint8_t s_a[2][BM * A_PITCH];
    // dequantised weight fragments in MFMA operand order, shared by the two waves that cover the same 32 columns:
    // [buffer][column half wn][k16 step][lane] x 16 bytes.  Each of those two waves converts half of the steps.
    __shared__ __attribute__((aligned(16))) uint4 s_b[2][2][4][64];
    static_assert(MB == 1 && NB == 1, "the shared-fragment path assumes one 32 x 32 block per wave");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l32 = lane & 31;
    const uint32_t M = p.m, N = p.n, K = p.k;
    const uint32_t m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const uint32_t G = (K + p.group_size - 1) / p.group_size;
    const uint32_t gs = p.group_size / BK; // k-steps per quant group
    const uint32_t KT = K / BK;
    const uint32_t row_bytes = K * BITS / 8;
    const uint32_t zp_stride = BITS == 4 ? (G + 1) / 2 : G;

    // ---- staging role: thread -> (row, CH chunks of 8 k)
    const int ra = tid / PARTS, part = tid % PARTS;
    const bool a_valid = m0 + ra < M;
    const uint16_t* a_src = (const uint16_t*)p.a + (size_t)(m0 + ra) * K + 8 * CH * part;
    uint8_t* a_dst0 = &s_a[0][ra * A_PITCH + part * 16 * CH];
    uint8_t* a_dst1 = &s_a[1][ra * A_PITCH + part * 16 * CH];
    constexpr int D = 4; // register stages: the operands of k-step kt + D - 1 are requested while k-step kt is computed
    uint4 a_st[D][CH];
    auto load_a = [&](uint32_t kt, uint4 (&st)[CH]) {
        if (a_valid && kt < KT) {
            const uint4* src = (const uint4*)(a_src + (size_t)kt * BK);
#pragma unroll
            for (int j = 0; j < CH; ++j) st[j] = src[j];
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) st[j] = make_uint4(0, 0, 0, 0);
        }
    };
    auto stage_a = [&](uint32_t kt, const uint4 (&st)[CH]) { // st holds k-step kt
        uint8_t* dst = (kt & 1) ? a_dst1 : a_dst0;
#pragma unroll
        for (int j = 0; j < CH; ++j) *(uint4*)(dst + 16 * j) = st[j];
    };

    // ---- compute role
    uint32_t ncol[NB];
    const uint8_t* w_src[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t n = n0 + wn * (32 * NB) + nb * 32 + l32;
        ncol[nb] = n < N ? n : N - 1;
        w_src[nb] = (const uint8_t*)p.b + (size_t)ncol[nb] * row_bytes + (size_t)(32 * half) * BITS / 8;
    }
    // wave (wm, wn) converts k16 steps 2 wm and 2 wm + 1 of its 32 columns: 16 codes per lane per k-step
    uint2 w_st[D][WV];
    auto load_w = [&](uint32_t kt, uint2 (&w)[WV]) {
        if (kt >= KT) return;
        const uint8_t* src = w_src[0] + (size_t)kt * BK * BITS / 8 + (size_t)wm * 2 * BITS; // 16 codes = 2 * BITS bytes
#pragma unroll
        for (int v = 0; v < WV; ++v) w[v] = ((const uint2*)src)[v];
    };
    // unsigned code q (after the optional `signed_codes` flip of the top bit, kernel.rs:268-275) -> two's complement of q - 2^(bits-1)
    const uint32_t flip = p.signed_codes ? 0u : (BITS == 4 ? 0x88888888u : 0x80808080u);

    // acc_s: the group's activation row sums, produced by the matrix cores too (B = all ones): it comes out in the
    // accumulator layout the group fold needs (16 rows per lane) -- no VALU adds, no LDS exchange
    f32x16_t acc_g[MB][NB], acc_t[MB][NB], acc_s[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_g[mb][nb][r] = 0.f, acc_t[mb][nb][r] = 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_s[mb][r] = 0.f;
    const u32x4_t ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; // eight bf16 1.0

    // Group scales / offsets of the tile's BN columns: staged once in LDS ([group][column], scale | offset << 16) when
    // they fit (`lds_groups` != 0) -- a per-group global load issued one group ahead is consumed two k-steps later,
    // far inside the ~1 us load latency; otherwise (very long K) they are fetched per group as before.
    extern __shared__ __attribute__((aligned(16))) uint32_t s_sc[];
    const bool sc_lds = lds_groups != 0;
    if (sc_lds) {
        for (uint32_t idx = tid; idx < G * BN; idx += 256) {
            const uint32_t col = idx % BN, g = idx / BN;
            const uint32_t n = n0 + col < N ? n0 + col : N - 1;
            uint32_t v = ((const uint16_t*)p.scales)[(size_t)n * G + g];
            if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) v |= (uint32_t)((const uint16_t*)p.biases)[(size_t)n * G + g] << 16;
            else if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) {
                const uint8_t z = p.zero_points[(size_t)n * zp_stride + (BITS == 4 ? (g >> 1) : g)];
                v |= (uint32_t)(BITS == 4 ? ((g & 1) ? (z >> 4) : (z & 0xF)) : z) << 16;
            }
            s_sc[g * BN + col] = v;
        }
    }
    uint16_t sc_raw[NB] = {}, of_raw[NB] = {};
    auto load_group = [&](uint32_t g) {
        if (sc_lds) return;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            sc_raw[nb] = ((const uint16_t*)p.scales)[(size_t)ncol[nb] * G + g];
            if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) of_raw[nb] = ((const uint16_t*)p.biases)[(size_t)ncol[nb] * G + g];
            else if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) {
                const uint8_t z = p.zero_points[(size_t)ncol[nb] * zp_stride + (BITS == 4 ? (g >> 1) : g)];
                of_raw[nb] = BITS == 4 ? ((g & 1) ? (z >> 4) : (z & 0xF)) : z;
            }
        }
    };

    // conversion of this wave's two k16 steps of k-step kt into the shared fragment buffer
    auto stage_b = [&](uint32_t kt, const uint2 (&w)[WV]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4_t f;
            if (BITS == 4) f = dequant4((j ? w[0].y : w[0].x) ^ flip);
            else f = dequant8(w[j].x ^ flip, w[j].y ^ flip);
            s_b[kt & 1][wn][2 * wm + j][lane] = make_uint4(f.x, f.y, f.z, f.w);
        }
    };
    // one k-step: MFMAs from the LDS buffers kt & 1
    auto compute = [&](uint32_t kt) {
        const uint8_t* a_base = &s_a[kt & 1][(wm * (32 * MB) + l32) * A_PITCH + half * 64];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4_t bfrag[NB], afrag[MB];
            {
                const uint4 t = s_b[kt & 1][wn][s][lane];
                bfrag[0] = u32x4_t{t.x, t.y, t.z, t.w};
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const uint4 t = *(const uint4*)(a_base + mb * 32 * A_PITCH + s * 16);
                afrag[mb] = u32x4_t{t.x, t.y, t.z, t.w};
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc_g[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, afrag[mb]), __builtin_bit_cast(bf16x8_t, bfrag[nb]),
                                                                           acc_g[mb][nb], 0, 0, 0);
                acc_s[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, afrag[mb]), __builtin_bit_cast(bf16x8_t, ones), acc_s[mb], 0, 0, 0);
            }
        }
        if ((kt + 1) % gs == 0) { // group boundary: fold the group accumulator into the total with the f32 scale
            const uint32_t g = kt / gs;
            float sc[NB], coef[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                uint16_t sr = sc_raw[nb], orw = of_raw[nb];
                if (sc_lds) {
                    const uint32_t v = s_sc[g * BN + wn * (32 * NB) + nb * 32 + l32];
                    sr = (uint16_t)(v & 0xFFFFu), orw = (uint16_t)(v >> 16);
                }
                const float scale = bf16_to_f32(sr);
                const float mid = (float)(1u << (BITS - 1)); // the codes were fed centred: q - mid (int4: divided by 16)
                if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) coef[nb] = fmaf(mid, scale, bf16_to_f32(orw));
                else if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) coef[nb] = scale * (mid - (float)orw);
                else coef[nb] = 0.0f;
                sc[nb] = BITS == 4 ? 16.0f * scale : scale;
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc_t[mb][nb][r] = fmaf(sc[nb], acc_g[mb][nb][r], fmaf(coef[nb], acc_s[mb][r], acc_t[mb][nb][r]));
                        acc_g[mb][nb][r] = 0.f;
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_s[mb][r] = 0.f;
            }
            if (g + 1 < G) load_group(g + 1);
        }
    };

    // Software pipeline, D register stages deep (global/L2 latency is ~1 us, a k-step of MFMAs ~0.1 us): stage u holds
    // k-step kt0 + u.  Iteration kt: MFMAs of kt; then the activations of kt + 1 (requested D - 1 iterations ago) move
    // from registers to the other LDS buffer, and stage u is refilled with k-step kt + D.
#pragma unroll
    for (int u = 0; u < D; ++u) {
        load_a(u, a_st[u]);
        load_w(u, w_st[u]);
    }
    load_group(0);
    stage_a(0, a_st[0]);
    stage_b(0, w_st[0]);
    load_a(D, a_st[0]);
    load_w(D, w_st[0]);
    __syncthreads();
    for (uint32_t kt0 = 0; kt0 < KT; kt0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const uint32_t kt = kt0 + u;
            if (kt < KT) {
                compute(kt);
                if (kt + 1 < KT) { // operands of k-step kt + 1 (requested D iterations ago): registers -> the other LDS buffers
                    stage_a(kt + 1, a_st[(u + 1) % D]);
                    stage_b(kt + 1, w_st[(u + 1) % D]);
                }
                load_a(kt + 1 + D, a_st[(u + 1) % D]);
                load_w(kt + 1 + D, w_st[(u + 1) % D]);
            }
            __syncthreads();
        }
    }

    // ---- epilogue in the reference's order (kernel.rs:281-292): lane holds column n, rows (r&3) + 8*(r>>2) + 4*half
    uint16_t* d = (uint16_t*)p.d;
    float* d32 = (float*)p.d;
    const bool out_f32 = p.d_dt == UZU_F32; // tensor-parallel partial sums (engine.hip)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t n = n0 + wn * (32 * NB) + nb * 32 + l32;
        if (n >= N) continue;
        const float bias = p.bias ? bf16_to_f32(((const uint16_t*)p.bias)[n]) : 0.0f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = m0 + wm * (32 * MB) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= M) continue;
                const size_t idx = (size_t)m * N + n;
                float value = p.ab_scale * acc_t[mb][nb][r];
                if (p.accumulate) value += out_f32 ? d32[idx] : bf16_to_f32(d[idx]);
                if (p.bias) value += bias;
                if (p.has_soft_cap) value = p.soft_cap * tanhf(value / p.soft_cap);
                if (out_f32) d32[idx] = value;
                else d[idx] = f32_to_bf16(value);
            }
    }
}

// Shapes the matrix-core path covers; everything else stays on the GEMV-tiled / reference kernels of k_matmul.hip.
bool gemm_q_mfma_supported(const MatmulParams& p) {
    static const uint32_t min_m = [] {
        // below ~20 rows the GEMV in passes of four rows is faster than a 64-row tile a quarter full: a 16-node speculative tree pass of
        // Qwen3.5-0.8B takes 2.93 ms with the GEMVs against 3.64 ms with the tiles (tools/verify_cost.py, profiles/r4_verify_cost.json)
        const char* e = lab_env("UZU_GEMM_MIN_M");
        return e ? (uint32_t)atoi(e) : 20u;
    }();
    if (p.b_kind == UZU_MATMUL_B_FULL_PRECISION || (p.bits != 4 && p.bits != 8)) return false;
    if (p.w_dt != UZU_BF16 || p.a_dt != UZU_BF16 || (p.d_dt != UZU_BF16 && p.d_dt != UZU_F32)) return false;
    if (p.m < min_m || p.gather) return false; // act_mul: only the 128-tile kernel has the fused epilogue (checked in gemm_q_mfma)
    if (p.k % BK || p.group_size % BK || p.k % p.group_size) return false;
    if ((uintptr_t)p.a % 16 || (uintptr_t)p.b % 16 || ((size_t)p.k * p.bits / 8) % 16) return false;
    return true;
}

template <int MB, int NB> static uzu_status launch_gemm(hipStream_t s, const MatmulParams& p) {
    const dim3 grid((p.n + 64 * NB - 1) / (64 * NB), (p.m + 64 * MB - 1) / (64 * MB));
    const uint32_t groups = p.k / p.group_size;
    const size_t sc_bytes = (size_t)groups * 64 * NB * 4;
    const uint32_t lds_groups = sc_bytes <= 16 * 1024 ? groups : 0; // 18 KB + 16 KB static: three workgroups per CU stay resident
    const size_t dyn = lds_groups ? sc_bytes : 0;
    if (p.bits == 4) return launch_check([&] { hipLaunchKernelGGL((gemm_q_mfma_kernel<4, MB, NB>), grid, dim3(256), dyn, s, p, lds_groups); }, "gemm_q_mfma");
    return launch_check([&] { hipLaunchKernelGGL((gemm_q_mfma_kernel<8, MB, NB>), grid, dim3(256), dyn, s, p, lds_groups); }, "gemm_q_mfma");
}
// Tile choice: the 128 x 128 workgroup tile needs 304 VGPRs (one wave per SIMD: nothing hides the operand latency);
// 64 x 64 runs four waves per SIMD and is 1.4-1.7x faster from 1024 x 1024 x 2048 up to 4096 x 14336 x 4096.
uzu_status gemm_q_mfma(hipStream_t s, const MatmulParams& p, int num_cus) {
    static const int force = [] {
        const char* e = lab_env("UZU_GEMM_TILE");
        return e ? atoi(e) : 0;
    }();
    // M >= 128: the 128 x 128 tile kernel (k_gemm128.hip).  Its scratch (row-sum pieces of A, split-K partials) is the
    // stream's workspace block, which is not available while the stream is being captured into a graph.
    if ((force != 64 || p.act_mul) && gemm_q_mfma128_supported(p, num_cus)) {
        if (void* ws = stream_workspace(s, gemm_q_mfma128_workspace_bytes(p, num_cus))) return gemm_q_mfma128(s, p, num_cus, ws);
    }
    if (p.act_mul) {
        set_error("matmul: the fused GatedActMul epilogue is not available for this shape (callers ask matmul_act_mul_supported first)");
        return UZU_ERR_UNSUPPORTED;
    }
    switch (force) { // tools/kbench KB_GEMM sweep: 64 x 64 tiles (108 VGPRs, 4 waves / SIMD) win at every shape tried
    default: return launch_gemm<1, 1>(s, p);
    }
}

} // namespace k
} // namespace uzu
