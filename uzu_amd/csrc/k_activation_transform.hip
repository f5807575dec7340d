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
#include "kernels.h"

namespace uzu {
namespace k {

namespace {

// 32-lane Hadamard butterfly (mod.rs:27-47); `half` lanes l = 0..31 hold element l of the stripe
__device__ __forceinline__ float hadamard32(float v, int l) {
#pragma unroll
    for (int stride = 1; stride < 32; stride <<= 1) {
        const float other = __shfl_xor(v, stride, 64);
        v = (l & stride) ? other - v : v + other; // lower lane keeps a + b, upper lane gets a - b (a = the lower lane's value)
    }
    return v * (1.0f / sqrtf(32.0f));
}

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
    const dim3 grid((p.n + 63u) / 64u, (p.m + 63u) / 64u);
    if (p.bits == 4) return launch_check([&] { hipLaunchKernelGGL(matmul_a8_kernel<4>, grid, dim3(256), 0, s, p, a_q, a_scales, a_group_size); }, "matmul_a8<4>");
    return launch_check([&] { hipLaunchKernelGGL(matmul_a8_kernel<8>, grid, dim3(256), 0, s, p, a_q, a_scales, a_group_size); }, "matmul_a8<8>");
}

} // namespace k
} // namespace uzu
