// k_elementwise.hip -- normalisation, rope/KV scatter, activations, embedding gather, sampling and the
// other bandwidth-trivial kernels of the forward path.  Each kernel states the reference kernel it
// re-implements (BU = crates/backend-uzu/src/backends).  Element-wise kernels reproduce the reference's
// rounding points exactly (compiled with -ffp-contract=off, glibc-exact expf), so they are BIT-EXACT
// against the CPU path; kernels with a reduction (norms, argmax is exact) use fixed-order trees.
#include "device_utils.h"
#include "gemv_core.h"
#include "kernels.h"
#include "rht_stripe.h"

namespace uzu {
namespace k {

// =============================================================== Normalization
// BU/cpu/kernel/normalization/normalization.rs:56-125.  One workgroup (256 threads) per row.  Reduction order
// (shared with the fused decode prologue in k_decode.hip): thread t owns the E = ceil(n/256) consecutive
// elements [t*E, t*E+E), accumulates them in order with one fma chain, then the wave butterfly 32..1, then the
// four wave results are added in wave order.
__device__ __forceinline__ float block_sum4(float v, float* red) { // 256 threads; red: 4 floats of LDS
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}
template <class T, class TA>
__global__ void __launch_bounds__(256) normalization_kernel(NormParams p) {
    __shared__ float red[8];
    const uint32_t n = p.element_count, E = (n + 255) / 256;
    const size_t off = (size_t)blockIdx.x * n;
    const T* input = p.input ? (const T*)p.input : (const T*)p.output;
    T* shortcut = (T*)p.shortcut;
    const uint32_t e0 = threadIdx.x * E, e1 = e0 + E < n ? e0 + E : n;
    float sum = 0.f, sum_sq = 0.f;
    for (uint32_t i = e0; i < e1; ++i) {
        float val = ld(input, off + i);
        if (p.copy_to_shortcut) {
            if (p.residual_add) {
                val = rnd<T>(val + ld(shortcut, off + i));
                if (p.scale_residual_sum) val = rnd<T>(val * p.post_layer_scalar);
            }
            st(shortcut, off + i, val);
        }
        if (p.subtract_mean) sum += val;
        sum_sq = fmaf(val, val, sum_sq);
    }
    const float cnt = (float)n;
    float mean = 0.f;
    if (p.subtract_mean) mean = block_sum4(sum, red) / cnt;
    const float variance = block_sum4(sum_sq, red + 4) / cnt - mean * mean;
    const float rms_inv = 1.0f / sqrtf(variance + p.epsilon);
    T* out = (T*)p.output;
    const TA* scales = (const TA*)p.scales;
    const TA* biases = (const TA*)p.biases;
    const T* src = p.residual_add ? (const T*)shortcut : input; // normalization.rs:95-99 (own elements: same thread wrote them)
    float gsum = 0.f; // rowsum_out: this thread's share of one group sum of the output row
    for (uint32_t i = e0; i < e1; ++i) {
        const float normalized = (ld(src, off + i) - mean) * rms_inv;
        float result;
        if (scales) {
            const float scale_val = ld(scales, i);
            if (p.full_layer)
                result = rnd<T>(normalized * (scale_val + p.scale_offset));
            else
                result = rnd<T>(rnd<T>(normalized) * rnd<T>(scale_val + p.scale_offset));
        } else {
            result = rnd<T>(normalized);
        }
        if (biases) result = rnd<T>(result + ld(biases, i));
        if (p.scale_output) result = rnd<T>(result * rnd<T>(p.post_layer_scalar));
        st(out, off + i, result);
        gsum += result;
    }
    if (p.rowsum_out) { // normalization_rowsum_supported: a thread's E elements lie inside one group, group / E neighbouring threads hold it
        const uint32_t lanes = p.rowsum_group / E, Mp = (p.batch_size + 3) & ~3u;
        for (uint32_t o = 1; o < lanes; o <<= 1) gsum += __shfl_xor(gsum, (int)o, 64);
        if (threadIdx.x % lanes == 0) p.rowsum_out[(size_t)(e0 / p.rowsum_group) * Mp + blockIdx.x] = gsum;
        if (blockIdx.x == p.batch_size - 1)
            for (uint32_t r = p.batch_size; r < Mp; ++r)
                for (uint32_t g = threadIdx.x; g < n / p.rowsum_group; g += 256) p.rowsum_out[(size_t)g * Mp + r] = 0.f;
    }
}
// bf16 rows of 1024 NV elements: the same arithmetic in the same order (thread t owns elements [4 NV t, 4 NV (t + 1)), one chain, block_sum4)
// with the row in registers between the two passes and 8-byte loads / stores -- the general kernel re-reads what it has just written to the
// shortcut (a store -> load round trip through the L2) with 2-byte accesses: 6.9 -> ~4.5 us per 1024 x 1024 prefill chunk.
// PART: the input rows are still the split-K partial tiles of the GEMM that produces them (k_gemm128.hip): the thread adds its elements' partials in
// split order, applies the GEMM's epilogue (bias), rounds and stores the GEMM's output -- gemm_split_reduce_kernel's arithmetic and result -- and goes
// on with the rounded values: one launch and one round trip of the rows instead of two.
template <int NV, class TA, bool PART>
__global__ void __launch_bounds__(256) normalization_rows_kernel(NormParams p, NormPartials sp) {
    __shared__ float red[8];
    constexpr uint32_t E = 4 * NV;
    const uint32_t n = p.element_count;
    const size_t off = (size_t)blockIdx.x * n;
    const uint16_t* input = p.input ? (const uint16_t*)p.input : (const uint16_t*)p.output;
    uint16_t* shortcut = (uint16_t*)p.shortcut;
    const uint32_t e0 = threadIdx.x * E;
    float v[E];
    u32x2_v sraw[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if constexpr (PART) {
            const size_t idx = off + e0 + 4 * j;
            f32x4_v acc = *(const f32x4_v*)(sp.partials + idx);
            for (uint32_t z = 1; z < sp.splits; ++z) acc += *(const f32x4_v*)(sp.partials + (size_t)z * sp.total + idx);
            float val[4] = {1.0f * acc.x, 1.0f * acc.y, 1.0f * acc.z, 1.0f * acc.w}; // ab_scale = 1 (checked by the caller)
            if (sp.bias) {
                const u32x2_v braw = *(const u32x2_v*)(sp.bias + e0 + 4 * j);
                val[0] += bits_to_f32(braw.x << 16), val[1] += bits_to_f32(braw.x & 0xFFFF0000u), val[2] += bits_to_f32(braw.y << 16), val[3] += bits_to_f32(braw.y & 0xFFFF0000u);
            }
            u32x2_v o;
            o.x = (uint32_t)f32_to_bf16(val[0]) | ((uint32_t)f32_to_bf16(val[1]) << 16);
            o.y = (uint32_t)f32_to_bf16(val[2]) | ((uint32_t)f32_to_bf16(val[3]) << 16);
            *(u32x2_v*)(sp.d + idx) = o; // the GEMM's output rows, as the separate reduction writes them
            v[4 * j] = bits_to_f32(o.x << 16), v[4 * j + 1] = bits_to_f32(o.x & 0xFFFF0000u);
            v[4 * j + 2] = bits_to_f32(o.y << 16), v[4 * j + 3] = bits_to_f32(o.y & 0xFFFF0000u);
        } else {
            const u32x2_v raw = *(const u32x2_v*)(input + off + e0 + 4 * j);
            v[4 * j] = bits_to_f32(raw.x << 16), v[4 * j + 1] = bits_to_f32(raw.x & 0xFFFF0000u);
            v[4 * j + 2] = bits_to_f32(raw.y << 16), v[4 * j + 3] = bits_to_f32(raw.y & 0xFFFF0000u);
        }
        if (p.copy_to_shortcut && p.residual_add) sraw[j] = *(const u32x2_v*)(shortcut + off + e0 + 4 * j);
    }
    float sum = 0.f, sum_sq = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (p.copy_to_shortcut) {
            if (p.residual_add) {
                const float sc[4] = {bits_to_f32(sraw[j].x << 16), bits_to_f32(sraw[j].x & 0xFFFF0000u), bits_to_f32(sraw[j].y << 16), bits_to_f32(sraw[j].y & 0xFFFF0000u)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float val = round_bf16(v[4 * j + i] + sc[i]);
                    if (p.scale_residual_sum) val = round_bf16(val * p.post_layer_scalar);
                    v[4 * j + i] = val;
                }
            }
            u32x2_v o;
            o.x = (f32_to_bits(v[4 * j]) >> 16) | (f32_to_bits(v[4 * j + 1]) & 0xFFFF0000u);
            o.y = (f32_to_bits(v[4 * j + 2]) >> 16) | (f32_to_bits(v[4 * j + 3]) & 0xFFFF0000u);
            *(u32x2_v*)(shortcut + off + e0 + 4 * j) = o; // (values are bf16-representable: truncation is exact)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (p.subtract_mean) sum += v[4 * j + i];
            sum_sq = fmaf(v[4 * j + i], v[4 * j + i], sum_sq);
        }
    }
    const float cnt = (float)n;
    float mean = 0.f;
    if (p.subtract_mean) mean = block_sum4(sum, red) / cnt;
    const float variance = block_sum4(sum_sq, red + 4) / cnt - mean * mean;
    const float rms_inv = 1.0f / sqrtf(variance + p.epsilon);
    uint16_t* out = (uint16_t*)p.output;
    const TA* scales = (const TA*)p.scales;
    const TA* biases = (const TA*)p.biases;
    float gsum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t e = e0 + 4 * j + i;
            const float normalized = (v[4 * j + i] - mean) * rms_inv;
            float result;
            if (scales) {
                const float scale_val = ld(scales, e);
                if (p.full_layer)
                    result = round_bf16(normalized * (scale_val + p.scale_offset));
                else
                    result = round_bf16(round_bf16(normalized) * round_bf16(scale_val + p.scale_offset));
            } else {
                result = round_bf16(normalized);
            }
            if (biases) result = round_bf16(result + ld(biases, e));
            if (p.scale_output) result = round_bf16(result * round_bf16(p.post_layer_scalar));
            r[i] = result;
            gsum += result;
        }
        u32x2_v o;
        o.x = (f32_to_bits(r[0]) >> 16) | (f32_to_bits(r[1]) & 0xFFFF0000u);
        o.y = (f32_to_bits(r[2]) >> 16) | (f32_to_bits(r[3]) & 0xFFFF0000u);
        *(u32x2_v*)(out + off + e0 + 4 * j) = o;
    }
    if (p.rowsum_out) {
        const uint32_t lanes = p.rowsum_group / E, Mp = (p.batch_size + 3) & ~3u;
        for (uint32_t o = 1; o < lanes; o <<= 1) gsum += __shfl_xor(gsum, (int)o, 64);
        if (threadIdx.x % lanes == 0) p.rowsum_out[(size_t)(e0 / p.rowsum_group) * Mp + blockIdx.x] = gsum;
        if (blockIdx.x == p.batch_size - 1)
            for (uint32_t r = p.batch_size; r < Mp; ++r)
                for (uint32_t g = threadIdx.x; g < n / p.rowsum_group; g += 256) p.rowsum_out[(size_t)g * Mp + r] = 0.f;
    }
}
bool normalization_rowsum_supported(uint32_t n, uint32_t group) {
    if (!group || n % 256 || n % group) return false;
    const uint32_t E = n / 256;
    if (group % E) return false;
    const uint32_t lanes = group / E;
    return lanes >= 1 && lanes <= 64 && (lanes & (lanes - 1)) == 0;
}

uzu_status normalization(hipStream_t s, const NormParams& p) {
    if (p.batch_size == 0) return UZU_OK;
    if (p.rowsum_out && (exact_mode() || !normalization_rowsum_supported(p.element_count, p.rowsum_group))) {
        set_error("normalization: rowsum_out is not available for %u elements in groups of %u (or in reference-order mode)", p.element_count, p.rowsum_group);
        return UZU_ERR_UNSUPPORTED;
    }
    if (exact_mode()) return normalization_exact(s, p);
    static const bool fast_rows = [] { // UZU_NORM_ROWS=0: the general kernel everywhere (A/B runs)
        const char* e = lab_env("UZU_NORM_ROWS");
        return !e || atoi(e) != 0;
    }();
    if (fast_rows && p.io_dt == UZU_BF16 && p.batch_size >= 16 && p.element_count % 1024 == 0 && (p.affine_dt == UZU_F32 || p.affine_dt == UZU_BF16) &&
        (((uintptr_t)p.input | (uintptr_t)p.output | (uintptr_t)p.shortcut) & 7) == 0) {
        const uint32_t nv = p.element_count / 1024;
        const NormPartials none{};
#define UZU_ROWS(NVV)                                                                                                                            \
    case NVV:                                                                                                                                    \
        if (p.affine_dt == UZU_F32)                                                                                                              \
            return launch_check([&] { hipLaunchKernelGGL((normalization_rows_kernel<NVV, float, false>), dim3(p.batch_size), dim3(256), 0, s, p, none); }, "normalization"); \
        return launch_check([&] { hipLaunchKernelGGL((normalization_rows_kernel<NVV, bf16_t, false>), dim3(p.batch_size), dim3(256), 0, s, p, none); }, "normalization");
        switch (nv) {
            UZU_ROWS(1)
            UZU_ROWS(2)
            UZU_ROWS(4)
            UZU_ROWS(5)
            UZU_ROWS(8)
        default: break;
        }
#undef UZU_ROWS
    }
    return UZU_DISPATCH_T(p.io_dt, [&]() -> uzu_status {
        if (p.affine_dt == UZU_F32)
            return launch_check([&] { hipLaunchKernelGGL((normalization_kernel<T, float>), dim3(p.batch_size), dim3(256), 0, s, p); }, "normalization");
        if (p.affine_dt == UZU_BF16)
            return launch_check([&] { hipLaunchKernelGGL((normalization_kernel<T, bf16_t>), dim3(p.batch_size), dim3(256), 0, s, p); }, "normalization");
        set_error("normalization: unsupported affine dtype %u", p.affine_dt);
        return UZU_ERR_UNSUPPORTED;
    });
}

// Split-K reduction + the GEMM's epilogue + the Normalization of the rows it produces, one launch (normalization_rows_kernel<.., true>).
bool normalization_from_partials_supported(const NormParams& p, const NormPartials& sp) {
    static const bool on = [] { // UZU_NORM_PARTIALS=0: the reduction and the normalisation as two launches (A/B runs)
        const char* e = tune_env("norm_partials");
        return !e || atoi(e) != 0;
    }();
    if (!on || exact_mode() || p.io_dt != UZU_BF16 || p.batch_size < 16 || p.element_count % 1024 || (p.affine_dt != UZU_F32 && p.affine_dt != UZU_BF16)) return false;
    const uint32_t nv = p.element_count / 1024;
    if (nv != 1 && nv != 2 && nv != 4 && nv != 5 && nv != 8) return false;
    if (!sp.partials || sp.splits < 2 || !sp.d || sp.total != (size_t)p.batch_size * p.element_count) return false;
    if (p.rowsum_out && !normalization_rowsum_supported(p.element_count, p.rowsum_group)) return false;
    return ((((uintptr_t)p.output | (uintptr_t)p.shortcut | (uintptr_t)sp.d | (uintptr_t)sp.bias) & 7) == 0) && ((uintptr_t)sp.partials & 15) == 0;
}
uzu_status normalization_from_partials(hipStream_t s, const NormParams& p, const NormPartials& sp) {
    const uint32_t nv = p.element_count / 1024;
#define UZU_ROWS(NVV)                                                                                                                            \
    case NVV:                                                                                                                                    \
        if (p.affine_dt == UZU_F32)                                                                                                              \
            return launch_check([&] { hipLaunchKernelGGL((normalization_rows_kernel<NVV, float, true>), dim3(p.batch_size), dim3(256), 0, s, p, sp); }, "normalization_partials"); \
        return launch_check([&] { hipLaunchKernelGGL((normalization_rows_kernel<NVV, bf16_t, true>), dim3(p.batch_size), dim3(256), 0, s, p, sp); }, "normalization_partials");
    switch (nv) {
        UZU_ROWS(1)
        UZU_ROWS(2)
        UZU_ROWS(4)
        UZU_ROWS(5)
        UZU_ROWS(8)
    default: break;
    }
#undef UZU_ROWS
    set_error("normalization_from_partials: %u elements per row", p.element_count);
    return UZU_ERR_UNSUPPORTED;
}

// =============================================================== QKVNorm
// BU/cpu/kernel/attention/qkv_norm.rs:32-77: one wave per (token, head), in place.
template <class T>
__global__ void __launch_bounds__(256) qkv_norm_kernel(T* qkv, const float* scales, uint32_t batch_size,
                                                       uint32_t total_heads, uint32_t head_dim, float epsilon,
                                                       float scale_offset, uint32_t head_offset, uint32_t head_count,
                                                       uint32_t full_layer) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= batch_size * head_count) return;
    const uint32_t batch = wave / head_count, head = wave % head_count;
    const size_t offset = (size_t)batch * total_heads * head_dim + (size_t)(head_offset + head) * head_dim;
    float total = 0.f;
    for (uint32_t i = lane; i < head_dim; i += 64) {
        const float v = ld(qkv, offset + i);
        total += v * v;
    }
    total = wave_sum(total);
    const float rms_norm = 1.0f / sqrtf(total / (float)head_dim + epsilon);
    for (uint32_t i = lane; i < head_dim; i += 64) {
        const float normalized = ld(qkv, offset + i) * rms_norm;
        float result;
        if (!scales)
            result = rnd<T>(normalized);
        else if (full_layer)
            result = rnd<T>(normalized * (scales[i] + scale_offset));
        else
            result = rnd<T>(rnd<T>(normalized) * rnd<T>(scales[i] + scale_offset));
        st(qkv, offset + i, result);
    }
}
uzu_status qkv_norm(hipStream_t s, void* qkv, uint32_t dt, const float* scales, uint32_t batch_size,
                    uint32_t total_heads, uint32_t head_dim, float epsilon, float scale_offset, uint32_t head_offset,
                    uint32_t head_count, uint32_t full_layer) {
    const uint32_t waves = batch_size * head_count;
    if (!waves) return UZU_OK;
    if (exact_mode()) return qkv_norm_exact(s, qkv, dt, scales, batch_size, total_heads, head_dim, epsilon, scale_offset, head_offset, head_count, full_layer);
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] {
            hipLaunchKernelGGL((qkv_norm_kernel<T>), dim3((waves + 3) / 4), dim3(256), 0, s, (T*)qkv, scales, batch_size,
                               total_heads, head_dim, epsilon, scale_offset, head_offset, head_count, full_layer);
        }, "qkv_norm");
    });
}

// =============================================================== AttentionPrepare
// BU/cpu/kernel/attention/attention_prepare.rs:7-126: split packed QKV, half-rotation RoPE on Q and K,
// Q -> [heads, batch, hd], K/V -> cache rows kv_token_offset + batch_idx, layout [tokens, kv_heads, hd].
__global__ void __launch_bounds__(256) attention_prepare_kernel(
    const uint16_t* __restrict__ qkv, uint16_t* __restrict__ queries, uint16_t* __restrict__ keys,
    uint16_t* __restrict__ values, const float* __restrict__ cosines, const float* __restrict__ sines,
    uint32_t num_q_heads, uint32_t num_kv_heads, uint32_t head_dim, uint32_t rope_dim, uint32_t kv_token_offset,
    uint32_t batch_dim, uint32_t has_kv, const uint32_t* dyn, uint32_t kv_rows_fixed, const uint32_t* __restrict__ trie) {
    const uint32_t total_heads = has_kv ? num_q_heads + 2 * num_kv_heads : num_q_heads;
    const size_t total = (size_t)batch_dim * total_heads * head_dim;
    const uint32_t pos0 = dyn ? *dyn : 0u;
    const uint32_t kv_row0 = kv_token_offset + (kv_rows_fixed ? 0u : pos0);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const uint32_t d = idx % head_dim;
        const uint32_t head_idx = (idx / head_dim) % total_heads;
        const uint32_t batch_idx = idx / ((size_t)head_dim * total_heads);
        const uint16_t* head = qkv + (size_t)batch_idx * total_heads * head_dim + (size_t)head_idx * head_dim;
        const bool is_query = !has_kv || head_idx < num_q_heads;
        const bool is_key = has_kv && head_idx >= num_q_heads && head_idx < num_q_heads + num_kv_heads;
        uint16_t element = head[d];
        if (rope_dim && d < rope_dim && (is_query || is_key)) {
            const uint32_t half = rope_dim / 2;
            const uint32_t paired_idx = d < half ? d + half : d - half;
            const float input = bf16_to_f32(head[d]);
            const float paired = bf16_to_f32(head[paired_idx]);
            const float signed_paired = d < half ? -paired : paired;
            const size_t r = (size_t)((trie ? trie[3 * (size_t)batch_idx + 2] : batch_idx) + pos0) * rope_dim + d; // tree: position = base + height
            element = f32_to_bf16(input * cosines[r] + signed_paired * sines[r]);
        }
        if (is_query) {
            queries[(size_t)head_idx * batch_dim * head_dim + (size_t)batch_idx * head_dim + d] = element;
        } else if (is_key) {
            keys[(size_t)(kv_row0 + batch_idx) * num_kv_heads * head_dim +
                 (size_t)(head_idx - num_q_heads) * head_dim + d] = element;
        } else {
            values[(size_t)(kv_row0 + batch_idx) * num_kv_heads * head_dim +
                   (size_t)(head_idx - num_q_heads - num_kv_heads) * head_dim + d] = element;
        }
    }
}
uzu_status attention_prepare(hipStream_t s, const uint16_t* qkv, uint16_t* queries, uint16_t* keys, uint16_t* values,
                             const float* cosines, const float* sines, uint32_t num_q_heads, uint32_t num_kv_heads,
                             uint32_t head_dim, uint32_t rope_dim, uint32_t kv_token_offset, uint32_t batch_dim,
                             uint32_t has_kv, const uint32_t* dyn, uint32_t kv_rows_fixed, const uint32_t* trie) {
    const uint32_t total_heads = has_kv ? num_q_heads + 2 * num_kv_heads : num_q_heads;
    const size_t total = (size_t)batch_dim * total_heads * head_dim;
    if (!total) return UZU_OK;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    return launch_check([&] {
        hipLaunchKernelGGL(attention_prepare_kernel, dim3(blocks), dim3(256), 0, s, qkv, queries, keys, values, cosines,
                           sines, num_q_heads, num_kv_heads, head_dim, rope_dim, kv_token_offset, batch_dim, has_kv, dyn, kv_rows_fixed, trie);
    }, "attention_prepare");
}

// QKVNorm (query heads, key heads, value heads) + AttentionPrepare as ONE launch (engine passes of more than one row, round 6): a wave per (row, head) keeps the head's
// elements in registers -- element i = lane + 64 j, qkv_norm_kernel's mapping, its sum order and its wave_sum, so the normalised values are the same bits -- parks them in
// a wave-private LDS row for the rotation's partner element and writes the row where attention_prepare_kernel would.  Three launches -> one per attention layer.
struct HeadNorm {
    const float* scales; // null: no scales (value normalisation)
    float epsilon, scale_offset;
    uint32_t full_layer, present;
};
constexpr uint32_t kPrepMaxHeadDim = 512;
__global__ void __launch_bounds__(256) attention_prepare_normed_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ queries, uint16_t* __restrict__ keys,
                                                                       uint16_t* __restrict__ values, const float* __restrict__ cosines, const float* __restrict__ sines,
                                                                       HeadNorm qn, HeadNorm kn, HeadNorm vn, uint32_t num_q_heads, uint32_t num_kv_heads, uint32_t head_dim,
                                                                       uint32_t rope_dim, uint32_t kv_token_offset, uint32_t batch_dim, uint32_t has_kv, const uint32_t* dyn,
                                                                       uint32_t kv_rows_fixed, const uint32_t* __restrict__ trie) {
    __shared__ uint16_t s_row[4][kPrepMaxHeadDim];
    using T = bf16_t;
    const uint32_t total_heads = has_kv ? num_q_heads + 2 * num_kv_heads : num_q_heads;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4 + wv;
    const bool live = wave < batch_dim * total_heads;
    const uint32_t batch_idx = live ? wave / total_heads : 0, head_idx = live ? wave % total_heads : 0;
    const uint32_t pos0 = dyn ? *dyn : 0u;
    const uint32_t kv_row0 = kv_token_offset + (kv_rows_fixed ? 0u : pos0);
    const bool is_query = !has_kv || head_idx < num_q_heads;
    const bool is_key = has_kv && head_idx >= num_q_heads && head_idx < num_q_heads + num_kv_heads;
    const HeadNorm N = is_query ? qn : is_key ? kn : vn;
    const size_t offset = (size_t)batch_idx * total_heads * head_dim + (size_t)head_idx * head_dim;
    const T* src = (const T*)qkv;
    if (N.present) {
        float total = 0.f;
        for (uint32_t i = lane; i < head_dim; i += 64) {
            const float v = ld(src, offset + i);
            total += v * v;
        }
        total = wave_sum(total);
        const float rms_norm = 1.0f / sqrtf(total / (float)head_dim + N.epsilon);
        for (uint32_t i = lane; i < head_dim; i += 64) {
            const float normalized = ld(src, offset + i) * rms_norm;
            float result;
            if (!N.scales)
                result = rnd<T>(normalized);
            else if (N.full_layer)
                result = rnd<T>(normalized * (N.scales[i] + N.scale_offset));
            else
                result = rnd<T>(rnd<T>(normalized) * rnd<T>(N.scales[i] + N.scale_offset));
            s_row[wv][i] = f32_to_bf16(result);
        }
    } else {
        for (uint32_t i = lane; i < head_dim; i += 64) s_row[wv][i] = qkv[offset + i];
    }
    __syncthreads();
    if (!live) return;
    const uint16_t* head = s_row[wv];
    for (uint32_t d = lane; d < head_dim; d += 64) {
        uint16_t element = head[d];
        if (rope_dim && d < rope_dim && (is_query || is_key)) {
            const uint32_t half = rope_dim / 2;
            const uint32_t paired_idx = d < half ? d + half : d - half;
            const float input = bf16_to_f32(head[d]);
            const float paired = bf16_to_f32(head[paired_idx]);
            const float signed_paired = d < half ? -paired : paired;
            const size_t r = (size_t)((trie ? trie[3 * (size_t)batch_idx + 2] : batch_idx) + pos0) * rope_dim + d; // tree: position = base + height
            element = f32_to_bf16(input * cosines[r] + signed_paired * sines[r]);
        }
        if (is_query)
            queries[(size_t)head_idx * batch_dim * head_dim + (size_t)batch_idx * head_dim + d] = element;
        else if (is_key)
            keys[(size_t)(kv_row0 + batch_idx) * num_kv_heads * head_dim + (size_t)(head_idx - num_q_heads) * head_dim + d] = element;
        else
            values[(size_t)(kv_row0 + batch_idx) * num_kv_heads * head_dim + (size_t)(head_idx - num_q_heads - num_kv_heads) * head_dim + d] = element;
    }
}
bool attention_prepare_normed_supported(uint32_t head_dim) { return !exact_mode() && head_dim <= kPrepMaxHeadDim; }
uzu_status attention_prepare_normed(hipStream_t s, const uint16_t* qkv, uint16_t* queries, uint16_t* keys, uint16_t* values, const float* cosines, const float* sines,
                                    const PrepNorm& qn, const PrepNorm& kn, const PrepNorm& vn, uint32_t num_q_heads, uint32_t num_kv_heads, uint32_t head_dim, uint32_t rope_dim,
                                    uint32_t kv_token_offset, uint32_t batch_dim, uint32_t has_kv, const uint32_t* dyn, uint32_t kv_rows_fixed, const uint32_t* trie) {
    const uint32_t total_heads = has_kv ? num_q_heads + 2 * num_kv_heads : num_q_heads;
    const uint32_t waves = batch_dim * total_heads;
    if (!waves) return UZU_OK;
    auto hn = [](const PrepNorm& p) { return HeadNorm{p.scales, p.epsilon, p.scale_offset, p.full_layer, p.present}; };
    return launch_check([&] {
        hipLaunchKernelGGL(attention_prepare_normed_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, qkv, queries, keys, values, cosines, sines, hn(qn), hn(kn), hn(vn), num_q_heads,
                           num_kv_heads, head_dim, rope_dim, kv_token_offset, batch_dim, has_kv, dyn, kv_rows_fixed, trie);
    }, "attention_prepare_normed");
}

// =============================================================== KVCacheUpdate
// BU/cpu/kernel/attention/kv_cache_update.rs:9-28.  The reference executes copies sequentially per
// element column; copy lists on this path never chain (ring insert / accept compaction read rows of the
// suffix region and write rows of the prefix region), so one thread per (copy, element) is equivalent.
// Copies arrive as inline constants (<= MAX_INLINE_BYTES) and travel in the kernel argument buffer.
struct CopyList {
    uzu_kv_copy c[UZU_HIP_MAX_INLINE_BYTES / sizeof(uzu_kv_copy) / 2];
};
template <class T>
__global__ void kv_cache_update_kernel(T* keys, T* values, CopyList list, uint32_t copy_count, uint32_t element_dim) {
    const size_t total = (size_t)copy_count * element_dim;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const uint32_t e = idx % element_dim, i = idx / element_dim;
        const size_t sidx = (size_t)list.c[i].source * element_dim + e, didx = (size_t)list.c[i].destination * element_dim + e;
        keys[didx] = keys[sidx];
        values[didx] = values[sidx];
    }
}
// the reference's own shape: one thread per element column walks the copies in order (kv_cache_update.rs:14-27) -- for lists whose
// copies depend on each other
template <class T>
__global__ void kv_cache_update_ordered_kernel(T* keys, T* values, CopyList list, uint32_t copy_count, uint32_t element_dim) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= element_dim) return;
    for (uint32_t i = 0; i < copy_count; ++i) {
        const size_t sidx = (size_t)list.c[i].source * element_dim + e, didx = (size_t)list.c[i].destination * element_dim + e;
        keys[didx] = keys[sidx];
        values[didx] = values[sidx];
    }
}
uzu_status kv_cache_update(hipStream_t s, void* keys, void* values, uint32_t dt, const uzu_kv_copy* copies,
                           uint32_t copy_count, uint32_t element_dim) {
    constexpr uint32_t kMax = sizeof(CopyList) / sizeof(uzu_kv_copy);
    for (uint32_t base = 0; base < copy_count; base += kMax) {
        const uint32_t n = copy_count - base < kMax ? copy_count - base : kMax;
        CopyList list;
        for (uint32_t i = 0; i < n; ++i) list.c[i] = copies[base + i];
        // A row that one copy writes and another copy reads or writes (the accept compaction of a speculated path does that: copy i
        // moves row a_i down to row i, and a later copy's destination j can be an earlier copy's source a_i) needs the reference's
        // sequential order per element column; independent lists run one thread per (copy, element).
        bool ordered = false;
        for (uint32_t i = 0; i < n && !ordered; ++i)
            for (uint32_t j = 0; j < n; ++j)
                if (i != j && (list.c[j].source == list.c[i].destination || list.c[j].destination == list.c[i].destination)) {
                    ordered = true;
                    break;
                }
        const size_t total = (size_t)n * element_dim;
        const uint32_t blocks = (uint32_t)(((ordered ? element_dim : total) + 255) / 256);
        uzu_status st = UZU_DISPATCH_T(dt, [&]() -> uzu_status {
            return launch_check([&] {
                if (ordered) hipLaunchKernelGGL((kv_cache_update_ordered_kernel<T>), dim3(blocks), dim3(256), 0, s, (T*)keys, (T*)values, list, n, element_dim);
                else hipLaunchKernelGGL((kv_cache_update_kernel<T>), dim3(blocks), dim3(256), 0, s, (T*)keys, (T*)values, list, n, element_dim);
            }, "kv_cache_update");
        });
        if (st != UZU_OK) return st;
    }
    return UZU_OK;
}

// Ring insert = AttentionState::encode_accept on a Ring (state.rs:200-219), full flat accept, with the accepted-token count read on the device
template <class T>
__global__ void kv_ring_insert_kernel(T* keys, T* values, const uint32_t* accepted, uint32_t batch, uint32_t W, uint32_t element_dim) {
    const uint32_t n = *accepted;
    const uint32_t first = batch > W ? batch - W : 0u; // earlier suffix rows would be overwritten by later ones (sequential copies)
    const size_t total = (size_t)(batch - first) * element_dim;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const uint32_t e = idx % element_dim, i = first + (uint32_t)(idx / element_dim);
        const size_t sidx = (size_t)(W + i) * element_dim + e, didx = (size_t)((n + i) % W) * element_dim + e;
        keys[didx] = keys[sidx];
        values[didx] = values[sidx];
    }
}
uzu_status kv_ring_insert(hipStream_t s, void* keys, void* values, uint32_t dt, const uint32_t* accepted, uint32_t batch, uint32_t ring_window, uint32_t element_dim) {
    if (!batch || !ring_window) return UZU_OK;
    const size_t total = (size_t)(batch > ring_window ? ring_window : batch) * element_dim;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] { hipLaunchKernelGGL((kv_ring_insert_kernel<T>), dim3(blocks), dim3(256), 0, s, (T*)keys, (T*)values, accepted, batch, ring_window, element_dim); },
                            "kv_ring_insert");
    });
}

// =============================================================== SigmoidGate (sigmoid_gate.rs:9-22)
template <class T> __global__ void sigmoid_gate_kernel(const T* gate, T* output, uint32_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float g = ld(gate, i);
        const float sigmoid = 1.0f / (1.0f + expf_glibc(-g));
        st(output, i, ld(output, i) * sigmoid);
    }
}
uzu_status sigmoid_gate(hipStream_t s, const void* gate, void* output, uint32_t dt, uint32_t total) {
    if (!total) return UZU_OK;
    const uint32_t blocks = (total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256;
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] { hipLaunchKernelGGL((sigmoid_gate_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)gate, (T*)output, total); }, "sigmoid_gate");
    });
}

// =============================================================== activations (gpu_types/activation_type.rs:16-65)
template <class T> __device__ __forceinline__ float activate(uint32_t act, float x) {
    switch (act) {
    case 0: return rnd<T>(x / (1.0f + expf_glibc(-1.0f * x)));                                         // SILU
    case 1: return rnd<T>(0.5f * x * (1.0f + tanhf(0.7978846f * (x + 0.044715f * x * x * x))));       // GELUApprox
    case 2: return rnd<T>(0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)));                      // GELUExact
    case 3: return x;                                                                                  // IDENTITY
    case 4: return x > 20.0f ? x : rnd<T>(logf_glibc(1.0f + expf_glibc(x)));                           // SOFTPLUS
    default: return x;
    }
}

// =============================================================== GatedActMul (gated_act_mul.rs:36-70, mod.rs:5-12)
template <class T>
__global__ void gated_act_mul_kernel(const T* act_operand, const T* value_operand, T* fp_out, uint32_t gated_dim,
                                     uint32_t batch_dim, uint32_t value_offset, uint32_t value_row_stride,
                                     uint32_t act_type, uint32_t interleaved) {
    const size_t total = (size_t)gated_dim * batch_dim;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t batch = idx / gated_dim, gated = idx % gated_dim;
        size_t act_index;
        float value;
        if (interleaved) {
            const size_t base = batch * 2 * gated_dim;
            act_index = base + gated_dim + gated;
            value = ld(act_operand, base + gated);
        } else {
            act_index = batch * gated_dim + gated;
            value = ld(value_operand, batch * value_row_stride + value_offset + gated);
        }
        const float gate = ld(act_operand, act_index);
        st(fp_out, idx, rnd<T>(value * activate<T>(act_type, gate)));
    }
}
uzu_status gated_act_mul(hipStream_t s, const void* act_operand, const void* value_operand, void* fp_out, uint32_t dt,
                         uint32_t gated_dim, uint32_t batch_dim, uint32_t value_offset, uint32_t value_row_stride,
                         uint32_t act_type, uint32_t interleaved) {
    const size_t total = (size_t)gated_dim * batch_dim;
    if (!total) return UZU_OK;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] {
            hipLaunchKernelGGL((gated_act_mul_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)act_operand, (const T*)value_operand,
                               (T*)fp_out, gated_dim, batch_dim, value_offset, value_row_stride, act_type, interleaved);
        }, "gated_act_mul");
    });
}

// =============================================================== RHT rows of the fused decode step (engine.hip::encode_decode_fused)
// A lane per element, 32 lanes per stripe (activation_transform_kernel's own mapping and butterfly: rht_stripe.h::hadamard32): the composition of
// the reference's kernels on ONE row, with their rounding points -- OutputRht of the up projection's two halves (+ its bias), GatedActMul,
// InputRht for the down projection (activation_transform OUTPUT_RHT -> tensor_add_bias -> gated_act_mul -> activation_transform INPUT_RHT;
// gated_act_mul.rs:36-70).  One launch instead of four: the activation (a glibc-exact expf) stays one per lane.
__global__ void __launch_bounds__(256) rht_mlp_join_kernel(const uint16_t* up_row, const uint32_t* up_out_bits, const uint16_t* up_bias, const uint32_t* down_in_bits,
                                                           uint16_t* out, uint32_t hidden, uint32_t act_type) {
    const uint32_t index = blockIdx.x * 256 + threadIdx.x, l = threadIdx.x & 31;
    const bool live = index < hidden; // hidden % 32 == 0: a stripe is live or dead as a whole
    const uint32_t at = live ? index : 0, stripes = hidden / 32;
    auto sign = [&](const uint32_t* bits, uint32_t word, float v) { return (bits[word] >> l) & 1u ? -v : v; };
    float up = bf16_to_f32(up_row[at]), gate = bf16_to_f32(up_row[hidden + at]);
    up = round_bf16(sign(up_out_bits, at / 32, hadamard32(up, (int)l)));
    gate = round_bf16(sign(up_out_bits, stripes + at / 32, hadamard32(gate, (int)l)));
    if (up_bias) up = round_bf16(up + bf16_to_f32(up_bias[at])), gate = round_bf16(gate + bf16_to_f32(up_bias[hidden + at]));
    float v = rnd<bf16_t>(up * activate<bf16_t>(act_type, gate));
    if (down_in_bits) v = round_bf16(hadamard32(sign(down_in_bits, at / 32, v), (int)l));
    if (live) out[index] = f32_to_bf16(v);
}
uzu_status rht_mlp_join(hipStream_t s, const uint16_t* up_row, const uint32_t* up_out_bits, const uint16_t* up_bias, const uint32_t* down_in_bits, uint16_t* out,
                        uint32_t hidden, uint32_t act_type) {
    if (!hidden || hidden % 32 || !up_out_bits) {
        set_error("rht_mlp_join: hidden %u is not a whole number of Hadamard blocks / no factors", hidden);
        return UZU_ERR_INVALID_ARGUMENT;
    }
    return launch_check([&] { hipLaunchKernelGGL(rht_mlp_join_kernel, dim3((hidden + 255) / 256), dim3(256), 0, s, up_row, up_out_bits, up_bias, down_in_bits, out, hidden, act_type); },
                        "rht_mlp_join");
}

// OutputRht (+ bias) of up to two raw output rows in place (query / key / value row and the gate row of a gated attention layer: one launch), and
// for row 0 the DeltaNetConvUpdate of its first conv_dim channels right behind it (conv_update.rs:17-55 on the value the transform has just
// produced; the channel's taps are read, shifted and written by that lane only) -- activation_transform OUTPUT_RHT -> tensor_add_bias
// [-> delta_net_conv_update] as one launch, same arithmetic per element.
__global__ void __launch_bounds__(256) rht_out_rows_kernel(uint16_t* row0, const uint32_t* bits0, const uint16_t* bias0, uint32_t n0, uint16_t* row1, const uint32_t* bits1,
                                                           const uint16_t* bias1, uint32_t n1, const float* conv_w, const float* conv_b, float* conv_state, uint32_t kernel_size,
                                                           uint32_t conv_dim) {
    const uint32_t blocks0 = (n0 + 255) / 256;
    const bool second = blockIdx.x >= blocks0;
    uint16_t* row = second ? row1 : row0;
    const uint32_t* bits = second ? bits1 : bits0;
    const uint16_t* bias = second ? bias1 : bias0;
    const uint32_t n = second ? n1 : n0, index = (second ? blockIdx.x - blocks0 : blockIdx.x) * 256 + threadIdx.x, l = threadIdx.x & 31;
    const bool live = index < n;
    const uint32_t at = live ? index : 0;
    float v = hadamard32(bf16_to_f32(row[at]), (int)l);
    v = round_bf16((bits[at / 32] >> l) & 1u ? -v : v);
    if (bias) v = round_bf16(v + bf16_to_f32(bias[at]));
    if (!live) return;
    if (!second && conv_w && index < conv_dim) {
        const uint32_t tap_count = kernel_size - 1;
        float* st_row = conv_state + (size_t)index * tap_count;
        const float* w = conv_w + (size_t)index * kernel_size;
        float acc = conv_b ? conv_b[index] : 0.0f;
        for (uint32_t tap = 0; tap < tap_count; ++tap) acc += st_row[tap] * w[tap];
        acc += v * w[tap_count];
        for (uint32_t tap = 1; tap < tap_count; ++tap) st_row[tap - 1] = st_row[tap];
        st_row[tap_count - 1] = v;
        v = silu_f32(acc);
    }
    row[index] = f32_to_bf16(v);
}
uzu_status rht_out_rows(hipStream_t s, uint16_t* row0, const uint32_t* bits0, const uint16_t* bias0, uint32_t n0, uint16_t* row1, const uint32_t* bits1, const uint16_t* bias1,
                        uint32_t n1, const float* conv_w, const float* conv_b, float* conv_state, uint32_t kernel_size, uint32_t conv_dim) {
    if (!n0 || n0 % 32 || n1 % 32 || !bits0 || (n1 && !bits1) || (conv_w && (!conv_state || kernel_size < 2 || conv_dim > n0))) {
        set_error("rht_out_rows: rows of %u / %u elements are not whole Hadamard blocks, or factors / conv operands are missing", n0, n1);
        return UZU_ERR_INVALID_ARGUMENT;
    }
    const uint32_t blocks = (n0 + 255) / 256 + (n1 + 255) / 256;
    return launch_check([&] { hipLaunchKernelGGL(rht_out_rows_kernel, dim3(blocks), dim3(256), 0, s, row0, bits0, bias0, n0, row1, bits1, bias1, n1, conv_w, conv_b, conv_state,
                                                  kernel_size, conv_dim); }, "rht_out_rows");
}

// =============================================================== embedding lookups (quant_embedding.rs:36-116)
template <class T>
__global__ void quantized_embedding_lookup_kernel(const uint32_t* token_ids, const uint8_t* weights, const T* scales,
                                                  const uint8_t* zero_points, const T* biases, T* output,
                                                  uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim,
                                                  float input_scale, uint32_t group_size, uint32_t bits, uint32_t method) {
    const uint32_t packing_divisor = 8 / bits;
    const size_t weights_stride = model_dim / packing_divisor;
    const size_t num_groups = (model_dim + group_size - 1) / group_size;
    const size_t zero_points_stride = bits == 4 ? (num_groups + 1) / 2 : num_groups;
    const size_t total = (size_t)batch_size * model_dim;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t b = idx / model_dim, dim_idx = idx % model_dim;
        const uint32_t token_id = token_ids[b];
        if (token_id >= vocab_size) {
            st(output, idx, 0.0f);
            continue;
        }
        const size_t group_idx = dim_idx / group_size;
        const float scale = ld(scales, (size_t)token_id * num_groups + group_idx);
        int quantized_value;
        if (bits == 4) {
            const uint8_t packed = weights[(size_t)token_id * weights_stride + dim_idx / 2];
            quantized_value = (dim_idx & 1) == 0 ? (packed & 0x0F) : ((packed >> 4) & 0x0F);
        } else {
            quantized_value = weights[(size_t)token_id * weights_stride + dim_idx];
        }
        float bias;
        if (method == 0) {
            bias = ld(biases, (size_t)token_id * num_groups + group_idx);
        } else if (method == 1) {
            uint8_t zero_point;
            if (bits == 4) {
                const uint8_t packed = zero_points[(size_t)token_id * zero_points_stride + group_idx / 2];
                zero_point = (group_idx & 1) == 0 ? (packed & 0x0F) : ((packed >> 4) & 0x0F);
            } else {
                zero_point = zero_points[(size_t)token_id * zero_points_stride + group_idx];
            }
            bias = -scale * (float)zero_point;
        } else {
            bias = -scale * (float)(1 << (bits - 1));
        }
        float out_f = scale * (float)quantized_value + bias;
        out_f = out_f * input_scale;
        st(output, idx, out_f);
    }
}
uzu_status quantized_embedding_lookup(hipStream_t s, const uint32_t* token_ids, const uint8_t* weights,
                                      const void* scales, const uint8_t* zero_points, const void* biases, void* output,
                                      uint32_t dt, uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim,
                                      float input_scale, uint32_t group_size, uint32_t bits, uint32_t method) {
    const size_t total = (size_t)batch_size * model_dim;
    if (!total) return UZU_OK;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] {
            hipLaunchKernelGGL((quantized_embedding_lookup_kernel<T>), dim3(blocks), dim3(256), 0, s, token_ids, weights, (const T*)scales,
                               zero_points, (const T*)biases, (T*)output, batch_size, vocab_size, model_dim, input_scale,
                               group_size, bits, method);
        }, "quantized_embedding_lookup");
    });
}

template <class T>
__global__ void full_precision_embedding_lookup_kernel(const uint32_t* token_ids, const T* weights, T* output,
                                                       uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim,
                                                       float input_scale) {
    const size_t total = (size_t)batch_size * model_dim;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t b = idx / model_dim, d = idx % model_dim;
        const uint32_t token_id = token_ids[b];
        if (token_id >= vocab_size)
            st(output, idx, 0.0f);
        else
            st(output, idx, ld(weights, (size_t)token_id * model_dim + d) * rnd<T>(input_scale));
    }
}
uzu_status full_precision_embedding_lookup(hipStream_t s, const uint32_t* token_ids, const void* weights, void* output,
                                           uint32_t dt, uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim,
                                           float input_scale) {
    const size_t total = (size_t)batch_size * model_dim;
    if (!total) return UZU_OK;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] {
            hipLaunchKernelGGL((full_precision_embedding_lookup_kernel<T>), dim3(blocks), dim3(256), 0, s, token_ids, (const T*)weights,
                               (T*)output, batch_size, vocab_size, model_dim, input_scale);
        }, "full_precision_embedding_lookup");
    });
}

// =============================================================== LogitTransform / Tensor*
template <class T>
__global__ void logit_transform_kernel(T* logits, uint32_t length, float scale, float soft_cap, uint32_t has_soft_cap) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < length; i += (size_t)gridDim.x * blockDim.x) {
        float value = ld(logits, i) * scale;
        if (has_soft_cap) value = tanhf(value / soft_cap) * soft_cap; // NB: ocml tanhf, <= 1 ulp from glibc
        st(logits, i, value);
    }
}
uzu_status logit_transform(hipStream_t s, void* logits, uint32_t dt, uint32_t length, float scale, float soft_cap,
                           uint32_t has_soft_cap) {
    if (!length) return UZU_OK;
    const uint32_t blocks = (length + 255) / 256 > 4096 ? 4096 : (length + 255) / 256;
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] { hipLaunchKernelGGL((logit_transform_kernel<T>), dim3(blocks), dim3(256), 0, s, (T*)logits, length, scale, soft_cap, has_soft_cap); }, "logit_transform");
    });
}

template <class T, class TB>
__global__ void tensor_add_bias_kernel(const T* input, const TB* bias, T* output, uint32_t num_cols, uint32_t length) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < length; i += (size_t)gridDim.x * blockDim.x)
        st(output, i, ld(input, i) + ld(bias, i % num_cols));
}
uzu_status tensor_add_bias(hipStream_t s, const void* input, const void* bias, void* output, uint32_t dt,
                           uint32_t bias_dt, uint32_t num_cols, uint32_t length) {
    if (!length) return UZU_OK;
    const void* in = input ? input : output;
    const uint32_t blocks = (length + 255) / 256 > 4096 ? 4096 : (length + 255) / 256;
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        if (bias_dt == UZU_F32)
            return launch_check([&] { hipLaunchKernelGGL((tensor_add_bias_kernel<T, float>), dim3(blocks), dim3(256), 0, s, (const T*)in, (const float*)bias, (T*)output, num_cols, length); }, "tensor_add_bias");
        return launch_check([&] { hipLaunchKernelGGL((tensor_add_bias_kernel<T, bf16_t>), dim3(blocks), dim3(256), 0, s, (const T*)in, (const bf16_t*)bias, (T*)output, num_cols, length); }, "tensor_add_bias");
    });
}
template <class T>
__global__ void tensor_add_scale_kernel(const T* input, const T* bias, T* output, uint32_t num_cols, uint32_t length, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < length; i += (size_t)gridDim.x * blockDim.x)
        st(output, i, (ld(input, i) + ld(bias, i % num_cols)) * scale);
}
uzu_status tensor_add_scale(hipStream_t s, const void* input, const void* bias, void* output, uint32_t dt,
                            uint32_t num_cols, uint32_t length, float scale) {
    if (!length) return UZU_OK;
    const void* in = input ? input : output;
    const uint32_t blocks = (length + 255) / 256 > 4096 ? 4096 : (length + 255) / 256;
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] { hipLaunchKernelGGL((tensor_add_scale_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)in, (const T*)bias, (T*)output, num_cols, length, scale); }, "tensor_add_scale");
    });
}
template <class T> __global__ void tensor_add_swap_kernel(T* skip, T* main_buf, uint32_t length) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < length; i += (size_t)gridDim.x * blockDim.x) {
        const float r = rnd<T>(ld(skip, i) + ld(main_buf, i));
        st(skip, i, r);
        st(main_buf, i, r);
    }
}
uzu_status tensor_add_swap(hipStream_t s, void* skip, void* main_buf, uint32_t dt, uint32_t length) {
    if (!length) return UZU_OK;
    const uint32_t blocks = (length + 255) / 256 > 4096 ? 4096 : (length + 255) / 256;
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] { hipLaunchKernelGGL((tensor_add_swap_kernel<T>), dim3(blocks), dim3(256), 0, s, (T*)skip, (T*)main_buf, length); }, "tensor_add_swap");
    });
}
uzu_status tensor_copy(hipStream_t s, const void* src, void* dst, uint32_t dt, uint32_t length) {
    const size_t bytes = (size_t)length * (dt == UZU_F32 ? 4 : 2);
    if (!bytes) return UZU_OK;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) {
        set_error("tensor_copy: %s", hipGetErrorString(e));
        return UZU_ERR_HIP;
    }
    return UZU_OK;
}

// =============================================================== greedy UnifiedSampling
// BU/cpu/kernel/sampling/unified_sampling.rs:90-98: arg-max, ties -> lowest index.  Two passes:
// up to 1024 workgroups reduce strided slices to (value, index) pairs, one workgroup finishes.
// (value, index) order: larger value wins; equal value -> smaller index wins; NaN never wins.
__device__ __forceinline__ void amax_combine(float& bv, uint32_t& bi, float v, uint32_t i) {
    if (v > bv || (v == bv && i < bi)) {
        bv = v;
        bi = i;
    }
}
template <class T>
__global__ void __launch_bounds__(256) argmax_pass1(const T* logits, uint32_t vocab_size, float* pv, uint32_t* pi) {
    __shared__ float sv[4];
    __shared__ uint32_t si[4];
    const uint32_t row = blockIdx.y;
    const T* l = logits + (size_t)row * vocab_size;
    // element 0 is the initial candidate exactly as in the reference's fold
    float bv = ld(l, 0);
    uint32_t bi = 0;
    if (!(bv == bv)) bv = INFINITY; // reference fold: a NaN first element is never replaced (partial_cmp -> Equal -> keep)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < vocab_size; i += gridDim.x * blockDim.x)
        amax_combine(bv, bi, ld(l, i), i);
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const uint32_t oi = __shfl_xor(bi, off, 64);
        amax_combine(bv, bi, ov, oi);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sv[wave] = bv, si[wave] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) amax_combine(bv, bi, sv[w], si[w]);
        pv[(size_t)row * gridDim.x + blockIdx.x] = bv;
        pi[(size_t)row * gridDim.x + blockIdx.x] = bi;
    }
}
__global__ void __launch_bounds__(256) argmax_pass2(const float* pv, const uint32_t* pi, uint32_t parts, uint32_t* output) {
    __shared__ float sv[4];
    __shared__ uint32_t si[4];
    const uint32_t row = blockIdx.x;
    float bv = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < parts; i += blockDim.x) amax_combine(bv, bi, pv[(size_t)row * parts + i], pi[(size_t)row * parts + i]);
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const uint32_t oi = __shfl_xor(bi, off, 64);
        amax_combine(bv, bi, ov, oi);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sv[wave] = bv, si[wave] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) amax_combine(bv, bi, sv[w], si[w]);
        output[row] = bi == 0xFFFFFFFFu ? 0u : bi;
    }
}
static constexpr uint32_t kArgmaxParts = 256;
size_t argmax_scratch_bytes(uint32_t batch_size) { return (size_t)batch_size * kArgmaxParts * 8; }
uzu_status argmax(hipStream_t s, const void* logits, uint32_t dt, uint32_t* output, uint32_t vocab_size,
                  uint32_t batch_size, void* scratch) {
    if (!batch_size) return UZU_OK;
    float* pv = (float*)scratch;
    uint32_t* pi = (uint32_t*)((char*)scratch + (size_t)batch_size * kArgmaxParts * 4);
    uint32_t parts = (vocab_size + 1023) / 1024;
    if (parts > kArgmaxParts) parts = kArgmaxParts;
    if (parts == 0) parts = 1;
    UZU_PROPAGATE(UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] { hipLaunchKernelGGL((argmax_pass1<T>), dim3(parts, batch_size), dim3(256), 0, s, (const T*)logits, vocab_size, pv, pi); }, "argmax_pass1");
    }));
    return launch_check([&] { hipLaunchKernelGGL(argmax_pass2, dim3(batch_size), dim3(256), 0, s, pv, pi, parts, output); }, "argmax_pass2");
}

// =============================================================== engine helpers
__global__ void advance_u32_kernel(uint32_t* counter, uint32_t amount) { *counter += amount; }
uzu_status advance_u32(hipStream_t s, uint32_t* counter, uint32_t amount) {
    return launch_check([&] { hipLaunchKernelGGL(advance_u32_kernel, dim3(1), dim3(1), 0, s, counter, amount); }, "advance_u32");
}
__global__ void fill_u32_kernel(uint32_t* dst, uint32_t value, uint32_t count) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) dst[i] = value;
}
uzu_status fill_u32(hipStream_t s, uint32_t* dst, uint32_t value, uint32_t count) {
    if (!count) return UZU_OK;
    return launch_check([&] { hipLaunchKernelGGL(fill_u32_kernel, dim3((count + 255) / 256 > 1024 ? 1024 : (count + 255) / 256), dim3(256), 0, s, dst, value, count); }, "fill_u32");
}

} // namespace k
} // namespace uzu
