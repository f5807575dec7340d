// k_activation_transform.hip -- the HybridSpec / "incoherence" path of the reference:
//
//   ActivationTransform   BU/cpu/kernel/activation_transform/activation_transform.rs:43-136 (+ mod.rs:9-47)
//       randomised Hadamard transform over stripes of 32 elements -- sign factors before the butterfly (InputRht, Quantize*) or
//       after it (OutputRht) -- optionally followed by symmetric int8 quantisation per activation group with i32 code sums;
//   MatmulA::Int8Symmetric BU/common/kernel/matmul/matmul_a.rs:9-14, CPU semantics cpu/kernel/matmul/kernel.rs:98-133,190-200
//       D = sum_k (q_a[m,k] * s_a[m,g]) * deq(B)[n,k]: int8 activations against int4 / int8 weight codes.
//
// gfx950 form.  The transform is one 32-lane half-wave per stripe: the butterfly is five lane exchanges, every addition is the
// reference's own (a + b, a - b in the same order), so the result is BIT-EXACT.  The matmul runs the integer part of every
// 32-element step on the packed int8 dot unit (v_dot4_i32_i8: exact) against centred weight codes and applies the f32 scales per
// step: acc += s_a * (s_w * D + (s_w * c + beta_w) * S), D = sum a (q - c), S = sum a, c = 8 / 128 -- the same value the
// reference reaches by sequential f32 multiply-adds of (q_a s_a) * (s_w q + beta_w), up to f32 summation order (tolerance class,
// like every other matmul here).  An int8-MFMA tile kernel (v_mfma_i32_32x32x32_i8, 2x the bf16 rate) is the planned next step
// for prefill-sized M; this kernel is the functional path and the parity anchor.
#include "device_utils.h"
#include "gemm_tile_map.h"
#include "kernels.h"
#include "rht_stripe.h"

namespace uzu {
namespace k {

namespace {

// (the 32-lane Hadamard butterfly, mod.rs:27-47: rht_stripe.h::hadamard32)

// grid (ceil(columns / 256), rows); 256 threads = 8 stripes of one row
template <class T>
__global__ void __launch_bounds__(256) activation_transform_kernel(const T* input, T* fp_out, int8_t* q_out, float* scales_out, int32_t* group_sums_out,
                                                                   const int32_t* rht_factors, uint32_t columns, uint32_t op, uint32_t scale_group,
                                                                   uint32_t sum_group) {
    __shared__ float s_mag[8];
    __shared__ int32_t s_sum[8];
    const uint32_t row = blockIdx.y, tid = threadIdx.x, l = tid & 31, stripe = tid >> 5;
    const uint32_t index = blockIdx.x * 256u + tid;
    const bool live = index < columns; // columns % 32 == 0: a stripe is live or dead as a whole
    const size_t row_offset = (size_t)row * columns;
    const bool input_rht = op != 1u, quantize = op >= 2u;
    float v = 0.f, factor = 1.f;
    if (live) {
        v = ld(input, row_offset + index);
        factor = (float)rht_factors[index];
    }
    if (input_rht) v = v * factor;
    v = hadamard32(v, (int)l);
    if (!input_rht) v = v * factor;
    if (!quantize) {
        if (live) st(fp_out, row_offset + index, v);
        return;
    }
    // symmetric int8 per `scale_group` (32 / 64 / 128 / 256 elements = 1 / 2 / 4 / 8 stripes of this workgroup)
    float mag = fabsf(v);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) mag = fmaxf(mag, __shfl_xor(mag, off, 64));
    if (l == 0) s_mag[stripe] = live ? mag : 0.f;
    __syncthreads();
    const uint32_t spg = scale_group / 32u, g0 = stripe / spg * spg;
    float magnitude = 0.f;
    for (uint32_t i = 0; i < spg; ++i) magnitude = fmaxf(magnitude, s_mag[g0 + i]);
    const float scale = (magnitude > 0.0f && magnitude < INFINITY) ? magnitude / 127.0f : 1.0f;
    float r = roundf(v / scale);
    r = fminf(fmaxf(r, -127.0f), 127.0f);
    const int32_t code = (int32_t)r;
    if (live) {
        q_out[row_offset + index] = (int8_t)code;
        if (index % scale_group == 0) scales_out[(size_t)row * (columns / scale_group) + index / scale_group] = scale;
    }
    if (op == 3u) {
        int32_t sum = live ? code : 0;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (l == 0) s_sum[stripe] = sum;
        __syncthreads();
        const uint32_t sps = sum_group / 32u;
        if (live && l == 0 && stripe % sps == 0) {
            int32_t t = 0;
            for (uint32_t i = 0; i < sps; ++i) t += s_sum[stripe + i];
            group_sums_out[(size_t)row * (columns / sum_group) + index / sum_group] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------- int8 activations x quantised weights
__device__ __forceinline__ int32_t dot4(uint32_t a, uint32_t b, int32_t c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }

// 64 x 64 output tile per workgroup, 4 x 4 outputs per thread, K in steps of 32 (every step lies in one activation group and one
// weight group: both sizes are multiples of 32).  LDS: the step's activation codes [64 rows][32] and centred weight codes [64 cols][32]
// as int8, padded to 36 bytes per row.
template <int BITS>
__global__ void __launch_bounds__(256) matmul_a8_kernel(MatmulParams p, const int8_t* a_q, const float* a_scales, uint32_t a_group) {
    constexpr int LD = 36; // bytes per LDS row (32 + 4 pad: 9 words, conflict-free column walks)
    __shared__ __attribute__((aligned(16))) uint8_t s_a[64 * LD];
    __shared__ __attribute__((aligned(16))) uint8_t s_w[64 * LD];
    const uint32_t tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const uint32_t n0 = blockIdx.x * 64u, m0 = blockIdx.y * 64u;
    const uint32_t K = p.k, steps = K / 32u;
    const uint32_t a_groups = K / a_group, w_groups = (K + p.group_size - 1) / p.group_size;
    const uint32_t zp_stride = BITS == 4 ? (w_groups + 1) / 2 : w_groups;
    const float centre = BITS == 4 ? 8.0f : 128.0f;
    const uint32_t flip = p.signed_codes ? (BITS == 4 ? 0x88888888u : 0x80808080u) : 0u;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // loader roles: thread t stages 8 bytes: row (t / 4) of the tile, bytes [8 (t % 4), +8) of the step
    const uint32_t lrow = tid >> 2, lpart = tid & 3;
    for (uint32_t step = 0; step < steps; ++step) {
        {   // activations: int8 codes as they are
            const uint32_t m = m0 + lrow;
            uint2 v = make_uint2(0u, 0u);
            if (m < p.m) v = *(const uint2*)(a_q + (size_t)m * K + step * 32u + lpart * 8u);
            *(uint32_t*)(s_a + lrow * LD + lpart * 8) = v.x;
            *(uint32_t*)(s_a + lrow * LD + lpart * 8 + 4) = v.y;
        }
        {   // weights: centred codes as int8
            const uint32_t n = n0 + lrow;
            uint32_t w0 = 0, w1 = 0;
            if (n < p.n) {
                if (BITS == 4) {
                    uint32_t w = *(const uint32_t*)((const uint8_t*)p.b + ((size_t)n * K + step * 32u) / 2 + lpart * 4u) ^ flip; // 8 nibbles
                    const uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;                                             // codes 0,2,4,6 / 1,3,5,7
                    const uint32_t e0 = __builtin_amdgcn_perm(hi, lo, 0x05010400u), e1 = __builtin_amdgcn_perm(hi, lo, 0x07030602u); // k order
                    w0 = ((e0 | 0x80808080u) - 0x08080808u) ^ 0x80808080u; // bytewise q - 8
                    w1 = ((e1 | 0x80808080u) - 0x08080808u) ^ 0x80808080u;
                } else {
                    const uint2 w = *(const uint2*)((const uint8_t*)p.b + (size_t)n * K + step * 32u + lpart * 8u);
                    w0 = (w.x ^ flip) ^ 0x80808080u, w1 = (w.y ^ flip) ^ 0x80808080u; // bytewise q - 128
                }
            }
            *(uint32_t*)(s_w + lrow * LD + lpart * 8) = w0;
            *(uint32_t*)(s_w + lrow * LD + lpart * 8 + 4) = w1;
        }
        __syncthreads();
        uint32_t av[4][8], wv[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 8; ++q) av[i][q] = *(const uint32_t*)(s_a + (ty * 4 + i) * LD + q * 4), wv[i][q] = *(const uint32_t*)(s_w + (tx * 4 + i) * LD + q * 4);
        const uint32_t ga = step * 32u / a_group, gw = step * 32u / p.group_size;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t m = m0 + ty * 4 + i;
            int32_t S = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) S = dot4(av[i][q], 0x01010101u, S);
            const float s_a_f = m < p.m ? a_scales[(size_t)m * a_groups + ga] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t n = n0 + tx * 4 + j;
                int32_t D = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) D = dot4(av[i][q], wv[j][q], D);
                float s_w_f = 0.f, beta = 0.f;
                if (n < p.n) {
                    s_w_f = ldt(p.scales, p.w_dt, (size_t)n * w_groups + gw);
                    if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) beta = ldt(p.biases, p.w_dt, (size_t)n * w_groups + gw);
                    else if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) {
                        const uint8_t zb = p.zero_points[(size_t)n * zp_stride + (BITS == 4 ? gw >> 1 : gw)];
                        const uint32_t zp = BITS == 4 ? ((gw & 1) ? (zb >> 4) : (zb & 0x0F)) : zb;
                        beta = -s_w_f * (float)zp;
                    } else beta = -s_w_f * centre;
                }
                const float t = fmaf(s_w_f, (float)D, fmaf(s_w_f, centre, beta) * (float)S);
                acc[i][j] = fmaf(s_a_f, t, acc[i][j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m >= p.m || n >= p.n) continue;
            const size_t output_index = (size_t)m * p.n + n;
            float value = p.ab_scale * acc[i][j]; // kernel.rs:281-292
            if (p.accumulate) value += ldt(p.d, p.d_dt, output_index);
            if (p.bias) value += ldt(p.bias, p.w_dt, n);
            if (p.has_soft_cap) value = p.soft_cap * tanhf(value / p.soft_cap);
            stt(p.d, p.d_dt, output_index, value);
        }
}

// ---------------------------------------------------------------------------------------------- the same on the int8 matrix cores
// Prefill-sized M: 128 x 128 output tile per workgroup, four waves as 2 x 2 (64 x 64 each = 2 x 2 blocks of v_mfma_i32_32x32x32_i8), K in
// STAGES of GK = min(activation group, weight group) in {64, 128} elements: inside a stage every (row, column) pair has ONE activation scale
// and ONE weight scale / offset, so the integer dot product of the stage is exact on the matrix core and the f32 work is one fold per stage:
//     acc += s_a[m] * (s_w[n] * D[m, n] + (s_w[n] * c + beta_w[n]) * S[m]),   D = sum a (u - c),  S = sum a  (i32, exact)
// -- the VALU kernel's formula with a 64 / 128-element step instead of 32.  int4 codes go in UNSIGNED (0 .. 15 fits an int8: c = 0, no
// centring arithmetic), int8 codes as u - 128 (one xor).  Both operands are staged through LDS as int8 rows (the unpack of the int4 codes to
// k order is the VALU kernel's two permutes per 8 codes); the stage's row sums S come from the staging threads (packed dot with ones + a
// DPP butterfly over the 8 threads of a row), the activation scales of the stage's 128 rows sit beside them.
// MFMA operand layout (32x32x32 i8): A lane l = row l % 32, k bytes 16 (l / 32) .. + 15; B lane l = column l % 32, same k bytes;
// C / D register r = row (r & 3) + 8 (r >> 2) + 4 (l / 32) of column l % 32.
typedef int a8_i32x4 __attribute__((ext_vector_type(4)));
typedef int a8_i32x16 __attribute__((ext_vector_type(16)));
typedef float a8_f32x2 __attribute__((ext_vector_type(2)));
template <int BITS, int GK>
__global__ void __launch_bounds__(256, 2) gemm_a8_mfma_kernel(MatmulParams p, const int8_t* a_q, const float* a_scales, uint32_t a_group) {
    constexpr int PITCH = GK + 16;  // bytes per LDS row: odd multiple of 16 (GK 64 -> 80, 128 -> 144)
    constexpr int KS = GK / 32;     // MFMA k-steps per stage
    constexpr int VPR = GK / 16;    // 16-byte vectors per row and stage
    constexpr int RPT = 128 * VPR / 256; // vectors a thread stages per operand and stage: 2 (GK 64) / 4 (GK 128): all of ONE row
    constexpr int TPR = VPR / RPT;       // threads per row: 2
    // two stages in LDS: the operands of stage st + 1 are requested from memory before the products of stage st and parked after them
    __shared__ __attribute__((aligned(16))) uint8_t s_a[2][128 * PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t s_w[2][128 * PITCH];
    __shared__ __attribute__((aligned(16))) float s_sa[2][128], s_rs[2][128]; // activation scale and scale * row sum of the stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, h = lane >> 5, c = lane & 31;
    // XCD-aware tile numbering (gemm_tile_map.h, as the bf16 kernel): the ~64 workgroups an XCD runs at a time share 8 x 8 panels in its L2
    // (plain x = column tile, y = row tile: every workgroup-stage went to the memory side, 3.4 TB/s of operand traffic at 4096 x 14336 x 4096)
    uint32_t m_t, n_t;
    if (!gemm_tile_of_block(blockIdx.x, (p.m + 127u) / 128u, (p.n + 127u) / 128u, &m_t, &n_t)) return;
    const uint32_t n0 = n_t * 128u, m0 = m_t * 128u, K = p.k, stages = K / GK;
    const uint32_t a_groups = K / a_group, w_groups = K / p.group_size;
    const uint32_t zp_stride = BITS == 4 ? (w_groups + 1) / 2 : w_groups;
    const float centre = BITS == 4 ? 0.0f : 128.0f;                        // what the staged weight bytes are short of the unsigned code
    const uint32_t flip = p.signed_codes ? (BITS == 4 ? 0x88888888u : 0x80808080u) : 0u;
    float acc[2][2][16];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
    uint32_t ncol[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) ncol[nb] = min(n0 + wn * 64 + nb * 32 + c, p.n - 1);
    // staging role: the RPT vectors [tid * RPT, + RPT) of the tile = a contiguous piece of row tid / TPR (clamped rows are computed, never stored)
    const uint32_t srow = (uint32_t)tid / TPR, sv0 = ((uint32_t)tid % TPR) * RPT;
    const int8_t* a_src = a_q + (size_t)min(m0 + srow, p.m - 1) * K + sv0 * 16;
    const uint8_t* w_src = (const uint8_t*)p.b + (size_t)min(n0 + srow, p.n - 1) * K * BITS / 8 + sv0 * 16 * BITS / 8;
    const float* sa_src = a_scales + (size_t)min(m0 + srow, p.m - 1) * a_groups;
    a8_i32x4 ra[RPT];
    uint2 rw4[BITS == 4 ? RPT : 1];
    a8_i32x4 rw8[BITS == 8 ? RPT : 1];
    float r_sa = 0.f;
    auto fetch = [&](uint32_t st) { // memory -> registers (stage st; past the end: the last stage again, never parked)
        const uint32_t k0 = min(st, stages - 1) * GK;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            ra[i] = *(const a8_i32x4*)(a_src + k0 + i * 16);
            if constexpr (BITS == 4) rw4[i] = *(const uint2*)(w_src + k0 / 2 + i * 8);
            else rw8[i] = *(const a8_i32x4*)(w_src + k0 + i * 16);
        }
        r_sa = sa_src[k0 / a_group];
    };
    auto park = [&](int buf) { // registers -> LDS buffer `buf`
        int32_t t = 0;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            *(a8_i32x4*)(&s_a[buf][srow * PITCH + (sv0 + i) * 16]) = ra[i];
#pragma unroll
            for (int w = 0; w < 4; ++w) t = dot4((uint32_t)ra[i][w], 0x01010101u, t);
            a8_i32x4 wv;
            if constexpr (BITS == 4) {
                const uint32_t w2[2] = {rw4[i].x ^ flip, rw4[i].y ^ flip};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const uint32_t lo = w2[q] & 0x0F0F0F0Fu, hi = (w2[q] >> 4) & 0x0F0F0F0Fu;   // codes 0,2,4,6 / 1,3,5,7
                    wv[2 * q] = (int)__builtin_amdgcn_perm(hi, lo, 0x05010400u);                 // k order, unsigned 0 .. 15
                    wv[2 * q + 1] = (int)__builtin_amdgcn_perm(hi, lo, 0x07030602u);
                }
            } else {
#pragma unroll
                for (int w = 0; w < 4; ++w) wv[w] = (int)(((uint32_t)rw8[i][w] ^ flip) ^ 0x80808080u); // bytewise u - 128
            }
            *(a8_i32x4*)(&s_w[buf][srow * PITCH + (sv0 + i) * 16]) = wv;
        }
        t += __shfl_xor(t, 1, 64); // the row's other thread (TPR = 2)
        if (tid % TPR == 0) s_sa[buf][srow] = r_sa, s_rs[buf][srow] = r_sa * (float)t;
    };
    // weight scale and offset coefficient of this lane's two columns for a stage: requested one stage ahead as well (read in the fold they
    // are four dependent memory round trips behind the MFMAs of every stage: 3.5 us per stage at 4096 x 14336 x 4096)
    float sw[2], cf[2], sw_n[2], cf_n[2];
    auto fetch_cols = [&](uint32_t st, float (&o_sw)[2], float (&o_cf)[2]) {
        const uint32_t gw = min(st, stages - 1) * GK / p.group_size;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const uint32_t n = ncol[nb];
            const float s_w_f = ldt(p.scales, p.w_dt, (size_t)n * w_groups + gw);
            float beta;
            if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) beta = ldt(p.biases, p.w_dt, (size_t)n * w_groups + gw);
            else if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) {
                const uint8_t zb = p.zero_points[(size_t)n * zp_stride + (BITS == 4 ? gw >> 1 : gw)];
                const uint32_t zp = BITS == 4 ? ((gw & 1) ? (zb >> 4) : (zb & 0x0F)) : zb;
                beta = -s_w_f * (float)zp;
            } else beta = -s_w_f * (BITS == 4 ? 8.0f : 128.0f);
            o_sw[nb] = s_w_f, o_cf[nb] = fmaf(s_w_f, centre, beta);
        }
    };
    fetch(0);
    fetch_cols(0, sw, cf);
    park(0);
    __syncthreads();
    for (uint32_t st = 0; st < stages; ++st) {
        const int buf = st & 1;
        fetch(st + 1); // unconditional: in flight during the products
        fetch_cols(st + 1, sw_n, cf_n);
        // ---- integer products of the stage
        a8_i32x16 d[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) d[mb][nb][r] = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            a8_i32x4 af[2], bf[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) af[mb] = *(const a8_i32x4*)(&s_a[buf][(wm * 64 + mb * 32 + c) * PITCH + ks * 32 + h * 16]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) bf[nb] = *(const a8_i32x4*)(&s_w[buf][(wn * 64 + nb * 32 + c) * PITCH + ks * 32 + h * 16]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) d[mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[mb], bf[nb], d[mb][nb], 0, 0, 0);
        }
        // ---- fold: this lane's two columns, rows (r & 3) + 8 (r >> 2) + 4 h of each block
        // packed f32 math (two rows per instruction): u = s_a * D, acc = fma(s_w, u, acc), acc = fma(coef, s_a * S, acc) -- 2 converts + 3 packed
        // operations per pair of outputs; the fold is what bounds this kernel (64 outputs per lane and stage against 16 MFMAs)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int row = wm * 64 + mb * 32 + 8 * r4 + 4 * h; // rows row .. row + 3 = registers 4 r4 .. 4 r4 + 3
                const float4 sa4 = *(const float4*)(&s_sa[buf][row]), rs4 = *(const float4*)(&s_rs[buf][row]);
                const a8_f32x2 sa2[2] = {{sa4.x, sa4.y}, {sa4.z, sa4.w}}, rs2[2] = {{rs4.x, rs4.y}, {rs4.z, rs4.w}};
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const a8_f32x2 sw2 = {sw[nb], sw[nb]}, cf2 = {cf[nb], cf[nb]};
#pragma unroll
                    for (int i2 = 0; i2 < 2; ++i2) {
                        const int r = 4 * r4 + 2 * i2;
                        const a8_f32x2 dv = {(float)d[mb][nb][r], (float)d[mb][nb][r + 1]};
                        a8_f32x2 av = {acc[mb][nb][r], acc[mb][nb][r + 1]};
                        av = __builtin_elementwise_fma(sw2, dv * sa2[i2], av);
                        av = __builtin_elementwise_fma(cf2, rs2[i2], av);
                        acc[mb][nb][r] = av.x, acc[mb][nb][r + 1] = av.y;
                    }
                }
            }
        if (st + 1 < stages) park(buf ^ 1); // the other buffer: its readers finished before the last barrier
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) sw[nb] = sw_n[nb], cf[nb] = cf_n[nb];
        __syncthreads();
    }
    // ---- epilogue (kernel.rs:281-292)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const uint32_t n = n0 + wn * 64 + nb * 32 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= p.m || n >= p.n) continue;
                const size_t output_index = (size_t)m * p.n + n;
                float value = p.ab_scale * acc[mb][nb][r];
                if (p.accumulate) value += ldt(p.d, p.d_dt, output_index);
                if (p.bias) value += ldt(p.bias, p.w_dt, n);
                if (p.has_soft_cap) value = p.soft_cap * tanhf(value / p.soft_cap);
                stt(p.d, p.d_dt, output_index, value);
            }
        }
}

// (An LDS-DMA ring form of this tile -- gemm_a8_dma_kernel, round 3 -- ran at the same rate and was removed in round 5: the int8-activation GEMM is
// never selected by the engine (select_activation_format answers Bf16), one matrix-core form is enough for the C ABI's MatmulA::Int8Symmetric.)
} // namespace

uzu_status activation_transform(hipStream_t s, const void* input, void* fp_out, int8_t* q_out, float* scales_out, int32_t* group_sums_out,
                                const int32_t* rht_factors, uint32_t dt, uint32_t batch_size, uint32_t element_count, uint32_t op,
                                uint32_t activation_scale_group_size, uint32_t sum_group_size) {
    if (!batch_size || !element_count) return UZU_OK;
    if (element_count % 32u) {
        set_error("activation_transform: element_count %u is not a multiple of the Hadamard block (32)", element_count);
        return UZU_ERR_INVALID_ARGUMENT;
    }
    if (op >= 2u) {
        const uint32_t g = activation_scale_group_size;
        if (!(g == 32 || g == 64 || g == 128 || g == 256) || element_count % g) {
            set_error("activation_transform: activation group %u must be 32 / 64 / 128 / 256 and divide %u", g, element_count);
            return UZU_ERR_UNSUPPORTED;
        }
        if (op == 3u && (!(sum_group_size == 32 || sum_group_size == 64 || sum_group_size == 128 || sum_group_size == 256) || element_count % sum_group_size)) {
            set_error("activation_transform: sum group %u must be 32 / 64 / 128 / 256 and divide %u", sum_group_size, element_count);
            return UZU_ERR_UNSUPPORTED;
        }
    }
    const void* in = input ? input : fp_out;
    const dim3 grid((element_count + 255u) / 256u, batch_size);
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] {
            hipLaunchKernelGGL((activation_transform_kernel<T>), grid, dim3(256), 0, s, (const T*)in, (T*)fp_out, q_out, scales_out, group_sums_out, rht_factors,
                               element_count, op, activation_scale_group_size, sum_group_size);
        }, "activation_transform");
    });
}

uzu_status matmul_a8(hipStream_t s, const MatmulParams& p, const int8_t* a_q, const float* a_scales, uint32_t a_group_size) {
    if (!p.m || !p.n) return UZU_OK;
    const bool groups_ok = (a_group_size == 32 || a_group_size == 64 || a_group_size == 128) && (p.group_size == 32 || p.group_size == 64 || p.group_size == 128);
    if (!groups_ok || p.k % a_group_size || p.k % 32u || p.b_kind == UZU_MATMUL_B_FULL_PRECISION || (p.bits != 4 && p.bits != 8)) {
        // MatmulError::IncompatibleA (cpu/kernel/matmul/kernel.rs:104-133)
        set_error("matmul: symmetric int8 activations require a supported 32/64/128 activation and weight group (a %u, w %u, k %u)", a_group_size, p.group_size, p.k);
        return UZU_ERR_UNSUPPORTED;
    }
    if (p.gather) {
        set_error("matmul: gather_indices with int8 activations is not supported");
        return UZU_ERR_UNSUPPORTED;
    }
    // prefill-sized M: the int8 matrix cores (stages of min(activation group, weight group) = 64 or 128 elements)
    static const bool mfma_on = [] { // UZU_A8_MFMA=0: the VALU kernel everywhere (A/B runs)
        const char* e = lab_env("UZU_A8_MFMA");
        return !e || atoi(e) != 0;
    }();
    const uint32_t gk = a_group_size < p.group_size ? a_group_size : p.group_size;
    if (mfma_on && p.m >= 128 && (gk == 64 || gk == 128) && p.k % gk == 0 && (p.bits == 8 || p.k % 32 == 0) && (uintptr_t)a_q % 16 == 0 && (uintptr_t)p.b % 16 == 0 &&
        p.k % 16 == 0) {
        const dim3 g2(gemm_grid_x((p.m + 127u) / 128u, (p.n + 127u) / 128u));
#define UZU_A8(B, G) return launch_check([&] { hipLaunchKernelGGL((gemm_a8_mfma_kernel<B, G>), g2, dim3(256), 0, s, p, a_q, a_scales, a_group_size); }, "gemm_a8_mfma")
        if (p.bits == 4) {
            if (gk == 64) UZU_A8(4, 64);
            UZU_A8(4, 128);
        }
        if (gk == 64) UZU_A8(8, 64);
        UZU_A8(8, 128);
#undef UZU_A8
    }
    const dim3 grid((p.n + 63u) / 64u, (p.m + 63u) / 64u);
    if (p.bits == 4) return launch_check([&] { hipLaunchKernelGGL(matmul_a8_kernel<4>, grid, dim3(256), 0, s, p, a_q, a_scales, a_group_size); }, "matmul_a8<4>");
    return launch_check([&] { hipLaunchKernelGGL(matmul_a8_kernel<8>, grid, dim3(256), 0, s, p, a_q, a_scales, a_group_size); }, "matmul_a8<8>");
}

} // namespace k
} // namespace uzu
