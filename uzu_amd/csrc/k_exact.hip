// k_exact.hip -- reference-ORDER variants of every reduction kernel on the forward path (UZU_HIP_EXACT=1 / uzu_hip_set_exact).
//
// The production kernels sum in parallel trees (butterflies, split-KV merges, chunked scans): bit-identical to each other, but only
// tolerance-equal to the reference's scalar loops.  These kernels do the same reductions in the reference's own order -- one thread
// per reduction, sequential loops, the same rounding points, the glibc-exact expf / logf of uzu_math.h -- so that with the matmul's
// reference-order kernel (k_matmul.hip::matmul_ref_kernel) a whole forward pass reproduces the CPU backend BIT FOR BIT.  Slow by
// construction (a decode attention is `heads` threads); this is the proof mode that separates "reduction order" from real defects,
// not a product path.  Each kernel cites the reference loop it follows.
#include <stdlib.h>

#include "attention_mask.h"
#include "device_utils.h"
#include "kernels.h"

namespace uzu {
namespace k {

// ---------------------------------------------------------------------------------------------- Normalization
// cpu/kernel/normalization/normalization.rs:56-125 (AccumT = f32): one thread per row.
template <class T, class AffineT>
__global__ void __launch_bounds__(64) normalization_exact_kernel(NormParams p) {
    const size_t batch = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (batch >= p.batch_size) return;
    const T* input = (const T*)(p.input ? p.input : p.output);
    T* output = (T*)p.output;
    T* shortcut = (T*)p.shortcut;
    const AffineT* scales = (const AffineT*)p.scales;
    const AffineT* biases = (const AffineT*)p.biases;
    const size_t element_count = p.element_count, off = batch * element_count;
    const float element_count_accum = (float)element_count;
    float sum = 0.0f, sum_sq = 0.0f;
    for (size_t i = 0; i < element_count; ++i) {
        float val = ld(input, off + i);
        if (p.copy_to_shortcut) {
            if (p.residual_add) {
                val = rnd<T>(val + ld(shortcut, off + i));
                if (p.scale_residual_sum) val = rnd<T>(val * p.post_layer_scalar);
            }
            st(shortcut, off + i, val);
        }
        const float accum_val = val;
        if (p.subtract_mean) sum = sum + accum_val;
        sum_sq = sum_sq + accum_val * accum_val;
    }
    const float mean = p.subtract_mean ? sum / element_count_accum : 0.0f;
    const float variance = sum_sq / element_count_accum - mean * mean;
    const float rms_inv = 1.0f / sqrtf(variance + p.epsilon);
    for (size_t i = 0; i < element_count; ++i) {
        const float input_val = p.residual_add ? ld(shortcut, off + i) : ld(input, off + i);
        const float normalized = (input_val - mean) * rms_inv;
        float result;
        if (scales) {
            const float scale_val = ld(scales, i);
            if (p.full_layer) {
                result = rnd<T>(normalized * (scale_val + p.scale_offset));
            } else {
                const float normalized_out = rnd<T>(normalized);
                const float scale_out = rnd<T>(scale_val + p.scale_offset);
                result = rnd<T>(normalized_out * scale_out);
            }
        } else {
            result = rnd<T>(normalized);
        }
        if (biases) result = rnd<T>(result + ld(biases, i));
        if (p.scale_output) result = rnd<T>(result * rnd<T>(p.post_layer_scalar));
        st(output, off + i, result);
    }
}
// The same loops with a workgroup per row (round 5: one thread walking a 4096-element row through global memory took 1.9 ms per decode
// launch).  Everything that is element-wise runs on 256 threads -- the residual add, the rounding to T, the shortcut store, the output
// formula: each element's arithmetic is the scalar kernel's, so the values are bit-identical -- and only the two running sums, whose ORDER
// is the point of this mode, stay sequential: thread 0 adds the staged values in index order (normalization.rs:56-78).
template <class T, class AffineT>
__global__ void __launch_bounds__(256) normalization_exact_row_kernel(NormParams p) {
    extern __shared__ float s_val[]; // element_count values + {mean, rms_inv}
    const size_t batch = blockIdx.x;
    const T* input = (const T*)(p.input ? p.input : p.output);
    T* output = (T*)p.output;
    T* shortcut = (T*)p.shortcut;
    const AffineT* scales = (const AffineT*)p.scales;
    const AffineT* biases = (const AffineT*)p.biases;
    const size_t element_count = p.element_count, off = batch * element_count;
    const float element_count_accum = (float)element_count;
    for (size_t i = threadIdx.x; i < element_count; i += 256) {
        float val = ld(input, off + i);
        if (p.copy_to_shortcut) {
            if (p.residual_add) {
                val = rnd<T>(val + ld(shortcut, off + i));
                if (p.scale_residual_sum) val = rnd<T>(val * p.post_layer_scalar);
            }
            st(shortcut, off + i, val);
        }
        s_val[i] = val;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sum = 0.0f, sum_sq = 0.0f;
        for (size_t i = 0; i < element_count; ++i) {
            const float accum_val = s_val[i];
            if (p.subtract_mean) sum = sum + accum_val;
            sum_sq = sum_sq + accum_val * accum_val;
        }
        const float mean = p.subtract_mean ? sum / element_count_accum : 0.0f;
        const float variance = sum_sq / element_count_accum - mean * mean;
        s_val[element_count] = mean;
        s_val[element_count + 1] = 1.0f / sqrtf(variance + p.epsilon);
    }
    __syncthreads();
    const float mean = s_val[element_count], rms_inv = s_val[element_count + 1];
    for (size_t i = threadIdx.x; i < element_count; i += 256) {
        // (the scalar kernel re-reads the shortcut it has just stored / the input: the same value as the staged one, T-rounded where it was stored)
        const float input_val = p.residual_add ? (p.copy_to_shortcut ? s_val[i] : ld(shortcut, off + i)) : ld(input, off + i);
        const float normalized = (input_val - mean) * rms_inv;
        float result;
        if (scales) {
            const float scale_val = ld(scales, i);
            if (p.full_layer) {
                result = rnd<T>(normalized * (scale_val + p.scale_offset));
            } else {
                const float normalized_out = rnd<T>(normalized);
                const float scale_out = rnd<T>(scale_val + p.scale_offset);
                result = rnd<T>(normalized_out * scale_out);
            }
        } else {
            result = rnd<T>(normalized);
        }
        if (biases) result = rnd<T>(result + ld(biases, i));
        if (p.scale_output) result = rnd<T>(result * rnd<T>(p.post_layer_scalar));
        st(output, off + i, result);
    }
}

uzu_status normalization_exact(hipStream_t s, const NormParams& p) {
    if (p.batch_size == 0) return UZU_OK;
    // in-place rows whose input IS the output and is not restaged through the shortcut would read elements another thread has already
    // overwritten only if an element depended on its neighbours -- it does not: element i reads and writes index i alone
    if (p.element_count <= 12288 && tune_env("exact_scalar") == nullptr) {
        const size_t lds = ((size_t)p.element_count + 2) * 4;
        return UZU_DISPATCH_T(p.io_dt, [&]() -> uzu_status {
            if (p.affine_dt == UZU_F32) return launch_check([&] { hipLaunchKernelGGL((normalization_exact_row_kernel<T, float>), dim3(p.batch_size), dim3(256), lds, s, p); }, "normalization_exact");
            if (p.affine_dt == UZU_BF16) return launch_check([&] { hipLaunchKernelGGL((normalization_exact_row_kernel<T, bf16_t>), dim3(p.batch_size), dim3(256), lds, s, p); }, "normalization_exact");
            set_error("normalization: unsupported affine dtype %u", p.affine_dt);
            return UZU_ERR_UNSUPPORTED;
        });
    }
    const uint32_t grid = (p.batch_size + 63) / 64;
    return UZU_DISPATCH_T(p.io_dt, [&]() -> uzu_status {
        if (p.affine_dt == UZU_F32) return launch_check([&] { hipLaunchKernelGGL((normalization_exact_kernel<T, float>), dim3(grid), dim3(64), 0, s, p); }, "normalization_exact");
        if (p.affine_dt == UZU_BF16) return launch_check([&] { hipLaunchKernelGGL((normalization_exact_kernel<T, bf16_t>), dim3(grid), dim3(64), 0, s, p); }, "normalization_exact");
        set_error("normalization: unsupported affine dtype %u", p.affine_dt);
        return UZU_ERR_UNSUPPORTED;
    });
}

// ---------------------------------------------------------------------------------------------- QKVNorm
// cpu/kernel/attention/qkv_norm.rs:32-77: one thread per (row, head).
template <class T>
__global__ void __launch_bounds__(64) qkv_norm_exact_kernel(T* qkv, const float* scales, uint32_t batch_size, uint32_t total_heads, uint32_t head_dim, float epsilon,
                                                             float scale_offset, uint32_t head_offset, uint32_t head_count, uint32_t full_layer) {
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= (size_t)batch_size * head_count) return;
    const size_t batch = idx / head_count, head = idx % head_count;
    const size_t qkv_stride = (size_t)total_heads * head_dim;
    const size_t offset = batch * qkv_stride + (head_offset + head) * head_dim;
    float total_sum = 0.0f;
    for (size_t i = 0; i < head_dim; ++i) {
        const float v = ld(qkv, offset + i);
        total_sum = total_sum + v * v;
    }
    const float mean_square = total_sum / (float)head_dim;
    const float rms_norm = 1.0f / sqrtf(mean_square + epsilon);
    for (size_t i = 0; i < head_dim; ++i) {
        const float normalized = ld(qkv, offset + i) * rms_norm;
        float result;
        if (!scales) result = rnd<T>(normalized);
        else if (full_layer) result = rnd<T>(normalized * (scales[i] + scale_offset));
        else result = rnd<T>(rnd<T>(normalized) * rnd<T>(scales[i] + scale_offset));
        st(qkv, offset + i, result);
    }
}
uzu_status qkv_norm_exact(hipStream_t s, void* qkv, uint32_t dt, const float* scales, uint32_t batch_size, uint32_t total_heads, uint32_t head_dim, float epsilon,
                          float scale_offset, uint32_t head_offset, uint32_t head_count, uint32_t full_layer) {
    const size_t total = (size_t)batch_size * head_count;
    if (!total) return UZU_OK;
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] {
            hipLaunchKernelGGL((qkv_norm_exact_kernel<T>), dim3((uint32_t)((total + 63) / 64)), dim3(64), 0, s, (T*)qkv, scales, batch_size, total_heads, head_dim, epsilon,
                               scale_offset, head_offset, head_count, full_layer);
        }, "qkv_norm_exact");
    });
}

// ---------------------------------------------------------------------------------------------- attention
// cpu/kernel/attention/attention_single_pass.rs:37-127 (BLOCKS = 1, init max -inf, output T) and attention_two_pass.rs:41-141 (32
// stride-interleaved key blocks, init max -1e9, f32 partials): one thread per (head, query[, block]); q / o live in private memory.
constexpr uint32_t kExactMaxHeadDim = 512;
template <class T>
__global__ void __launch_bounds__(64) attention_exact_kernel(AttentionParams a_in, uint32_t num_blocks, float init_max, T* out, float* partials, float* sums, float* maxs) {
    AttentionParams a = a_in;
    attention_resolve_dyn(a);
    const uint32_t HD = a.head_dim;
    const uint32_t sequence_length = a.sequence_length;
    const uint32_t prefix_length = sequence_length - a.suffix_length;
    const uint32_t suffix_position = a.is_kv_cache_ring ? a.ring_length : prefix_length;
    const size_t total = (size_t)a.num_heads * a.suffix_length * num_blocks;
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= total) return;
    const uint32_t block_idx = (uint32_t)(idx % num_blocks);
    const size_t hq = idx / num_blocks;
    const uint32_t head_idx = (uint32_t)(hq / a.suffix_length), q_seq_idx = (uint32_t)(hq % a.suffix_length);
    const uint32_t kv_head_idx = head_idx / a.gqa_factor;
    const size_t o_offset = (size_t)q_seq_idx * a.num_heads + head_idx;
    const size_t q_offset = (size_t)head_idx * a.suffix_length + q_seq_idx;
    const uint32_t query_position = attention_query_position(a, suffix_position, q_seq_idx);
    const T* queries = (const T*)a.queries;
    const T* keys = (const T*)a.keys;
    const T* values = (const T*)a.values;
    float q[kExactMaxHeadDim], o[kExactMaxHeadDim];
    for (uint32_t j = 0; j < HD; ++j) {
        q[j] = a.scale * ld(queries, q_offset * HD + j);
        o[j] = 0.0f;
    }
    float max_score = init_max, sum_exp_score = 0.0f;
    if (a.sinks && block_idx == 0) {
        max_score = ld((const T*)a.sinks, num_blocks == 1 ? head_idx % a.num_heads : head_idx);
        sum_exp_score = 1.0f;
    }
    for (uint32_t i = block_idx; i < sequence_length; i += num_blocks) {
        if (!should_use_key(a, q_seq_idx, prefix_length, suffix_position, query_position, i)) continue;
        const size_t kb = (size_t)kv_head_idx * a.k_head_stride + (size_t)i * a.k_seq_stride;
        float score = 0.0f;
        for (uint32_t j = 0; j < HD; ++j) score += q[j] * ld(keys, kb + j);
        const float new_max = fmaxf(max_score, score);
        const float factor = expf_glibc(max_score - new_max);
        const float exp_score = expf_glibc(score - new_max);
        max_score = new_max;
        sum_exp_score = sum_exp_score * factor + exp_score;
        const size_t vb = (size_t)kv_head_idx * a.v_head_stride + (size_t)i * a.v_seq_stride;
        for (uint32_t j = 0; j < HD; ++j) o[j] = o[j] * factor + exp_score * ld(values, vb + j);
    }
    if (partials) {
        float* out_base = partials + (o_offset * num_blocks + block_idx) * HD;
        for (uint32_t j = 0; j < HD; ++j) out_base[j] = o[j];
        sums[o_offset * num_blocks + block_idx] = sum_exp_score;
        maxs[o_offset * num_blocks + block_idx] = max_score;
    } else {
        for (uint32_t j = 0; j < HD; ++j) st(out, o_offset * HD + j, o[j] / sum_exp_score);
    }
}
// The same reduction orders with a WORKGROUP per (head, query, block) instead of a thread (round 5: a decode step has heads x blocks = 8 .. 1280
// of them; one thread each, with q / o in private memory, took 3.5 ms per launch).  Per tile of 256 of the block's keys:
//   1. thread t owns key t: score_t = sum_j q_j k_tj, j ascending, one accumulator (the scalar kernel's inner loop);
//   2. the running maximum is a prefix maximum (max is exact and associative): every thread scans the tile's scores up to its own key, then
//      factor_t = exp(max_before_t - max_t) and exp_score_t = exp(score_t - max_t) -- the scalar kernel's values, computed in parallel;
//      thread 0 runs the one chain that is order-dependent, sum = sum * factor_t + exp_score_t, over the tile in key order;
//   3. thread j owns output element j (and j + 256): o_j = o_j * factor_t + exp_score_t * v_tj for the tile's keys in order.
// Masked keys are skipped exactly as the scalar kernel's `continue` does.  Bit-identical to attention_exact_kernel (GPU test).
template <class T>
__global__ void __launch_bounds__(256) attention_exact_coop_kernel(AttentionParams a_in, uint32_t num_blocks, float init_max, T* out, float* partials, float* sums, float* maxs) {
    __shared__ float q_s[kExactMaxHeadDim];
    __shared__ float score_s[256], fac_s[256], exp_s[256];
    __shared__ uint32_t use_s[256];
    __shared__ float s_state[2]; // running max, running sum
    AttentionParams a = a_in;
    attention_resolve_dyn(a);
    const uint32_t HD = a.head_dim, tid = threadIdx.x;
    const uint32_t sequence_length = a.sequence_length;
    const uint32_t prefix_length = sequence_length - a.suffix_length;
    const uint32_t suffix_position = a.is_kv_cache_ring ? a.ring_length : prefix_length;
    const size_t idx = blockIdx.x;
    const uint32_t block_idx = (uint32_t)(idx % num_blocks);
    const size_t hq = idx / num_blocks;
    const uint32_t head_idx = (uint32_t)(hq / a.suffix_length), q_seq_idx = (uint32_t)(hq % a.suffix_length);
    const uint32_t kv_head_idx = head_idx / a.gqa_factor;
    const size_t o_offset = (size_t)q_seq_idx * a.num_heads + head_idx;
    const size_t q_offset = (size_t)head_idx * a.suffix_length + q_seq_idx;
    const uint32_t query_position = attention_query_position(a, suffix_position, q_seq_idx);
    const T* queries = (const T*)a.queries;
    const T* keys = (const T*)a.keys;
    const T* values = (const T*)a.values;
    for (uint32_t j = tid; j < HD; j += 256) q_s[j] = a.scale * ld(queries, q_offset * HD + j);
    float o0 = 0.0f, o1 = 0.0f;
    if (tid == 0) {
        float max_score = init_max, sum_exp_score = 0.0f;
        if (a.sinks && block_idx == 0) {
            max_score = ld((const T*)a.sinks, num_blocks == 1 ? head_idx % a.num_heads : head_idx);
            sum_exp_score = 1.0f;
        }
        s_state[0] = max_score, s_state[1] = sum_exp_score;
    }
    __syncthreads();
    const uint32_t keys_of_block = sequence_length > block_idx ? (sequence_length - block_idx + num_blocks - 1) / num_blocks : 0;
    for (uint32_t base = 0; base < keys_of_block; base += 256) {
        // 1. scores
        const uint32_t n = base + tid;
        const uint32_t i = block_idx + n * num_blocks;
        const bool use = n < keys_of_block && should_use_key(a, q_seq_idx, prefix_length, suffix_position, query_position, i);
        float score = 0.0f;
        if (use) {
            const size_t kb = (size_t)kv_head_idx * a.k_head_stride + (size_t)i * a.k_seq_stride;
            for (uint32_t j = 0; j < HD; ++j) score += q_s[j] * ld(keys, kb + j);
        }
        score_s[tid] = score, use_s[tid] = use ? 1u : 0u;
        __syncthreads();
        // 2. prefix maxima -> factors, exponentials; the sum chain on thread 0
        {
            float before = s_state[0];
            for (uint32_t t = 0; t < tid; ++t)
                if (use_s[t]) before = fmaxf(before, score_s[t]);
            if (use) {
                const float new_max = fmaxf(before, score);
                fac_s[tid] = expf_glibc(before - new_max);
                exp_s[tid] = expf_glibc(score - new_max);
            }
        }
        __syncthreads();
        if (tid == 0) {
            float max_score = s_state[0], sum_exp_score = s_state[1];
            for (uint32_t t = 0; t < 256; ++t)
                if (use_s[t]) {
                    max_score = fmaxf(max_score, score_s[t]);
                    sum_exp_score = sum_exp_score * fac_s[t] + exp_s[t];
                }
            s_state[0] = max_score, s_state[1] = sum_exp_score;
        }
        // 3. outputs
        const uint32_t tile = keys_of_block - base < 256 ? keys_of_block - base : 256;
        for (uint32_t t = 0; t < tile; ++t) {
            if (!use_s[t]) continue;
            const uint32_t it = block_idx + (base + t) * num_blocks;
            const size_t vb = (size_t)kv_head_idx * a.v_head_stride + (size_t)it * a.v_seq_stride;
            const float factor = fac_s[t], exp_score = exp_s[t];
            if (tid < HD) o0 = o0 * factor + exp_score * ld(values, vb + tid);
            if (tid + 256 < HD) o1 = o1 * factor + exp_score * ld(values, vb + tid + 256);
        }
        __syncthreads();
    }
    const float max_score = s_state[0], sum_exp_score = s_state[1];
    if (partials) {
        float* out_base = partials + (o_offset * num_blocks + block_idx) * HD;
        if (tid < HD) out_base[tid] = o0;
        if (tid + 256 < HD) out_base[tid + 256] = o1;
        if (tid == 0) sums[o_offset * num_blocks + block_idx] = sum_exp_score, maxs[o_offset * num_blocks + block_idx] = max_score;
    } else {
        if (tid < HD) st(out, o_offset * HD + tid, o0 / sum_exp_score);
        if (tid + 256 < HD) st(out, o_offset * HD + tid + 256, o1 / sum_exp_score);
    }
}
static uzu_status attention_exact_launch(hipStream_t s, const AttentionParams& a, uint32_t num_blocks, float init_max, void* out, float* partials, float* sums, float* maxs) {
    if (!a.suffix_length || !a.num_heads) return UZU_OK;
    if (a.head_dim > kExactMaxHeadDim) {
        set_error("attention (reference order): head_dim %u > %u", a.head_dim, kExactMaxHeadDim);
        return UZU_ERR_UNSUPPORTED;
    }
    const size_t total = (size_t)a.num_heads * a.suffix_length * num_blocks;
    if (total <= 16384 && tune_env("exact_scalar") == nullptr) // decode steps, tree passes, short suffixes: a workgroup per reduction
        return UZU_DISPATCH_T(a.dt, [&]() -> uzu_status {
            return launch_check([&] {
                hipLaunchKernelGGL((attention_exact_coop_kernel<T>), dim3((uint32_t)total), dim3(256), 0, s, a, num_blocks, init_max, (T*)out, partials, sums, maxs);
            }, "attention_exact");
        });
    return UZU_DISPATCH_T(a.dt, [&]() -> uzu_status {
        return launch_check([&] {
            hipLaunchKernelGGL((attention_exact_kernel<T>), dim3((uint32_t)((total + 63) / 64)), dim3(64), 0, s, a, num_blocks, init_max, (T*)out, partials, sums, maxs);
        }, "attention_exact");
    });
}
uzu_status attention_single_pass_exact(hipStream_t s, const AttentionParams& a, void* out) { return attention_exact_launch(s, a, 1, -INFINITY, out, nullptr, nullptr, nullptr); }
uzu_status attention_two_pass1_exact(hipStream_t s, const AttentionParams& a, float* partials, float* sums, float* maxs) {
    return attention_exact_launch(s, a, 32, -1e9f, nullptr, partials, sums, maxs);
}
// attention_two_pass.rs:154-190: one thread per output row (the production pass 2 already follows this order; kept here so that the
// whole exact path lives in one file and uses one thread per reduction)
template <class T>
__global__ void __launch_bounds__(64) attention_two_pass2_exact_kernel(const float* partials, const float* sums, const float* maxs, T* out, uint32_t HD, uint32_t total_rows) {
    const size_t o_offset = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (o_offset >= total_rows) return;
    const float* mx = maxs + o_offset * 32;
    const float* sm = sums + o_offset * 32;
    float global_max = -INFINITY;
    for (uint32_t b = 0; b < 32; ++b) global_max = fmaxf(global_max, mx[b]);
    float global_sum = 0.0f;
    for (uint32_t b = 0; b < 32; ++b) global_sum += sm[b] * expf_glibc(mx[b] - global_max);
    for (uint32_t j = 0; j < HD; ++j) {
        float val = 0.0f;
        for (uint32_t b = 0; b < 32; ++b) val += partials[(o_offset * 32 + b) * HD + j] * expf_glibc(mx[b] - global_max);
        st(out, o_offset * HD + j, val / global_sum);
    }
}
// a workgroup per output row: thread j owns output element j (its 32-term sum in block order, as above); the weights exp(max_b - global_max)
// are the same values for every j and are computed once per block by thread b
template <class T>
__global__ void __launch_bounds__(256) attention_two_pass2_exact_row_kernel(const float* partials, const float* sums, const float* maxs, T* out, uint32_t HD) {
    __shared__ float w_s[32];
    __shared__ float s_sum;
    const size_t o_offset = blockIdx.x;
    const float* mx = maxs + o_offset * 32;
    const float* sm = sums + o_offset * 32;
    float global_max = -INFINITY;
    for (uint32_t b = 0; b < 32; ++b) global_max = fmaxf(global_max, mx[b]);
    if (threadIdx.x < 32) w_s[threadIdx.x] = expf_glibc(mx[threadIdx.x] - global_max);
    __syncthreads();
    if (threadIdx.x == 0) {
        float global_sum = 0.0f;
        for (uint32_t b = 0; b < 32; ++b) global_sum += sm[b] * w_s[b];
        s_sum = global_sum;
    }
    __syncthreads();
    const float global_sum = s_sum;
    for (uint32_t j = threadIdx.x; j < HD; j += 256) {
        float val = 0.0f;
        for (uint32_t b = 0; b < 32; ++b) val += partials[(o_offset * 32 + b) * HD + j] * w_s[b];
        st(out, o_offset * HD + j, val / global_sum);
    }
}
uzu_status attention_two_pass2_exact(hipStream_t s, const float* partials, const float* sums, const float* maxs, void* out, uint32_t dt, uint32_t head_dim, uint32_t num_heads,
                                     uint32_t suffix_length) {
    const uint32_t rows = num_heads * suffix_length;
    if (!rows) return UZU_OK;
    if (rows <= 16384 && tune_env("exact_scalar") == nullptr)
        return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
            return launch_check([&] { hipLaunchKernelGGL((attention_two_pass2_exact_row_kernel<T>), dim3(rows), dim3(256), 0, s, partials, sums, maxs, (T*)out, head_dim); },
                                "attention_two_pass2_exact");
        });
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] { hipLaunchKernelGGL((attention_two_pass2_exact_kernel<T>), dim3((rows + 63) / 64), dim3(64), 0, s, partials, sums, maxs, (T*)out, head_dim, rows); },
                            "attention_two_pass2_exact");
    });
}

// ---------------------------------------------------------------------------------------------- Gated DeltaNet, decode
// cpu/kernel/gdn/update.rs:30-143.  The reference loops over value heads; inside a head the Dv state rows are independent until the
// norm over o: kernel 1 = one thread per (head, row i) -- q / k normalisation, beta, decay, k.q recomputed per thread in the reference's
// order -- writes o[i] (f32 scratch) and the new state row; kernel 2 = one thread per head: sequential sum of o^2, norm * SiLU(z).
__device__ __forceinline__ void dn_head_scalars(const uint16_t* in_proj, const float* a_log, const float* dt_bias, size_t hv, size_t hk, uint32_t head_k_dim, uint32_t key_dim,
                                                uint32_t value_dim, uint32_t num_v_heads, float* q, float* k, float& beta, float& decay, float& kq_dot) {
    const size_t conv_dim = 2 * (size_t)key_dim + value_dim;
    const size_t q_offset = hk * head_k_dim, k_offset = key_dim + hk * head_k_dim;
    for (size_t j = 0; j < head_k_dim; ++j) {
        q[j] = bf16_to_f32(in_proj[q_offset + j]);
        k[j] = bf16_to_f32(in_proj[k_offset + j]);
    }
    float q_norm_sq = 0.0f, k_norm_sq = 0.0f;
    for (size_t j = 0; j < head_k_dim; ++j) q_norm_sq += q[j] * q[j];
    for (size_t j = 0; j < head_k_dim; ++j) k_norm_sq += k[j] * k[j];
    const float q_inv_norm = 1.0f / sqrtf(q_norm_sq + 1e-6f);
    const float k_inv_norm = 1.0f / sqrtf(k_norm_sq + 1e-6f);
    for (size_t j = 0; j < head_k_dim; ++j) {
        q[j] *= q_inv_norm;
        k[j] *= k_inv_norm;
    }
    const float q_scale = 1.0f / sqrtf((float)head_k_dim);
    for (size_t j = 0; j < head_k_dim; ++j) q[j] *= q_scale;
    const float beta_raw = bf16_to_f32(in_proj[conv_dim + value_dim + hv]);
    beta = 1.0f / (1.0f + expf_glibc(-beta_raw));
    const float a_raw = bf16_to_f32(in_proj[conv_dim + value_dim + num_v_heads + hv]);
    const float sp_input = a_raw + dt_bias[hv];
    const float sp = sp_input > 20.0f ? sp_input : logf_glibc(1.0f + expf_glibc(sp_input));
    const float g = -expf_glibc(a_log[hv]) * sp;
    decay = expf_glibc(g);
    kq_dot = 0.0f;
    for (size_t j = 0; j < head_k_dim; ++j) kq_dot += k[j] * q[j];
}
constexpr uint32_t kExactMaxHeadK = 256;
__global__ void __launch_bounds__(64) delta_net_update_exact_rows_kernel(const uint16_t* in_proj, const float* a_log, const float* dt_bias, float* state, float* o_out,
                                                                          uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim,
                                                                          uint32_t value_dim) {
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= (size_t)num_v_heads * head_v_dim) return;
    const size_t hv = idx / head_v_dim, i = idx % head_v_dim;
    const size_t hk = hv / (num_v_heads / num_k_heads);
    float q[kExactMaxHeadK], k[kExactMaxHeadK];
    float beta, decay, kq_dot;
    dn_head_scalars(in_proj, a_log, dt_bias, hv, hk, head_k_dim, key_dim, value_dim, num_v_heads, q, k, beta, decay, kq_dot);
    const float v_i = bf16_to_f32(in_proj[2 * (size_t)key_dim + hv * head_v_dim + i]);
    float* srow = state + hv * head_v_dim * head_k_dim + i * head_k_dim;
    float sq_acc = 0.0f, sk_acc = 0.0f;
    for (size_t j = 0; j < head_k_dim; ++j) {
        const float sv = srow[j];
        sq_acc += sv * q[j];
        sk_acc += sv * k[j];
    }
    const float retrieved_i = decay * sk_acc;
    const float delta_i = beta * (v_i - retrieved_i);
    o_out[idx] = decay * sq_acc + delta_i * kq_dot;
    for (size_t j = 0; j < head_k_dim; ++j) srow[j] = decay * srow[j] + k[j] * delta_i;
}
__global__ void __launch_bounds__(64) delta_net_update_exact_gate_kernel(const uint16_t* in_proj, const float* norm_weight, const float* o, uint16_t* out, uint32_t num_v_heads,
                                                                          uint32_t head_v_dim, uint32_t key_dim, uint32_t value_dim, float norm_epsilon) {
    const size_t hv = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (hv >= num_v_heads) return;
    const size_t conv_dim = 2 * (size_t)key_dim + value_dim;
    const float* oh = o + hv * head_v_dim;
    float sumsq = 0.0f;
    for (size_t i = 0; i < head_v_dim; ++i) sumsq += oh[i] * oh[i];
    const float inv_rms = 1.0f / sqrtf(sumsq / (float)head_v_dim + norm_epsilon);
    for (size_t i = 0; i < head_v_dim; ++i) {
        const float z_i = bf16_to_f32(in_proj[conv_dim + hv * head_v_dim + i]);
        const float z_silu = silu_f32(z_i);
        const float final_val = oh[i] * inv_rms * norm_weight[i] * z_silu;
        out[hv * head_v_dim + i] = f32_to_bf16(final_val);
    }
}
uzu_status delta_net_update_exact(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias, const float* norm_weight, float* state, uint16_t* out,
                                  uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim, uint32_t value_dim, float norm_epsilon) {
    if (!num_v_heads) return UZU_OK;
    if (head_k_dim > kExactMaxHeadK || num_k_heads == 0 || num_v_heads % num_k_heads) {
        set_error("delta_net_update (reference order): head_k_dim %u > %u or Hv %% Hk != 0", head_k_dim, kExactMaxHeadK);
        return UZU_ERR_UNSUPPORTED;
    }
    const size_t rows = (size_t)num_v_heads * head_v_dim;
    float* o = (float*)stream_workspace(s, rows * sizeof(float));
    if (!o) {
        set_error("delta_net_update (reference order): no workspace (the exact mode does not run under stream capture)");
        return UZU_ERR_UNSUPPORTED;
    }
    UZU_PROPAGATE(launch_check([&] {
        hipLaunchKernelGGL(delta_net_update_exact_rows_kernel, dim3((uint32_t)((rows + 63) / 64)), dim3(64), 0, s, in_proj, a_log, dt_bias, state, o, num_v_heads, num_k_heads,
                           head_k_dim, head_v_dim, key_dim, value_dim);
    }, "delta_net_update_exact_rows"));
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_update_exact_gate_kernel, dim3((num_v_heads + 63) / 64), dim3(64), 0, s, in_proj, norm_weight, o, out, num_v_heads, head_v_dim, key_dim, value_dim,
                           norm_epsilon);
    }, "delta_net_update_exact_gate");
}

// ---------------------------------------------------------------------------------------------- Gated DeltaNet, prefill
// cpu/kernel/gdn/prefill_prep.rs:30-113: one thread per (token, k head).
__global__ void __launch_bounds__(64) delta_net_prefill_prep_exact_kernel(const uint16_t* in_proj, const float* a_log, const float* dt_bias, float* q_norm_out, float* k_norm_out,
                                                                           float* beta_out, float* decay_out, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim,
                                                                           uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len) {
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= (size_t)suffix_len * num_k_heads) return;
    const size_t token = idx / num_k_heads, hk = idx % num_k_heads;
    const size_t conv_dim = 2 * (size_t)key_dim + value_dim;
    const size_t total_proj_dim = conv_dim + value_dim + 2 * (size_t)num_v_heads;
    const size_t groups_per_head = num_v_heads / num_k_heads;
    const size_t tok_offset = token * total_proj_dim;
    const size_t q_off = tok_offset + hk * head_k_dim;
    float q_sq = 0.0f;
    for (size_t j = 0; j < head_k_dim; ++j) {
        const float v = bf16_to_f32(in_proj[q_off + j]);
        q_sq += v * v;
    }
    const float q_inv = 1.0f / sqrtf(q_sq + 1e-6f);
    const float q_scale = 1.0f / sqrtf((float)head_k_dim);
    for (size_t j = 0; j < head_k_dim; ++j) q_norm_out[token * key_dim + hk * head_k_dim + j] = bf16_to_f32(in_proj[q_off + j]) * q_inv * q_scale;
    const size_t k_off = tok_offset + key_dim + hk * head_k_dim;
    float k_sq = 0.0f;
    for (size_t j = 0; j < head_k_dim; ++j) {
        const float v = bf16_to_f32(in_proj[k_off + j]);
        k_sq += v * v;
    }
    const float k_inv = 1.0f / sqrtf(k_sq + 1e-6f);
    for (size_t j = 0; j < head_k_dim; ++j) k_norm_out[token * key_dim + hk * head_k_dim + j] = bf16_to_f32(in_proj[k_off + j]) * k_inv;
    for (size_t group = 0; group < groups_per_head; ++group) {
        const size_t hv = hk * groups_per_head + group;
        const float beta_raw = bf16_to_f32(in_proj[tok_offset + conv_dim + value_dim + hv]);
        const float beta = 1.0f / (1.0f + expf_glibc(-beta_raw));
        const float a_raw = bf16_to_f32(in_proj[tok_offset + conv_dim + value_dim + num_v_heads + hv]);
        const float sp_in = a_raw + dt_bias[hv];
        const float sp = sp_in > 20.0f ? sp_in : logf_glibc(1.0f + expf_glibc(sp_in));
        const float log_decay = -expf_glibc(a_log[hv]) * sp;
        beta_out[token * num_v_heads + hv] = beta;
        decay_out[token * num_v_heads + hv] = expf_glibc(log_decay);
    }
}
uzu_status delta_net_prefill_prep_exact(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias, float* q_norm_out, float* k_norm_out, float* beta_out,
                                        float* decay_out, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len) {
    const size_t total = (size_t)suffix_len * num_k_heads;
    if (!total) return UZU_OK;
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_prefill_prep_exact_kernel, dim3((uint32_t)((total + 63) / 64)), dim3(64), 0, s, in_proj, a_log, dt_bias, q_norm_out, k_norm_out, beta_out, decay_out,
                           num_v_heads, num_k_heads, head_k_dim, key_dim, value_dim, suffix_len);
    }, "delta_net_prefill_prep_exact");
}
// cpu/kernel/gdn/prefill.rs:39-80: one thread per state row (head, i), tokens in order.
__global__ void __launch_bounds__(64) delta_net_prefill_exact_kernel(const float* q_norm, const float* k_norm, const float* beta_buf, const float* decay_buf, const uint16_t* in_proj,
                                                                      float* state, uint16_t* out, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim,
                                                                      uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len) {
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= (size_t)num_v_heads * head_v_dim) return;
    const size_t hv = idx / head_v_dim, i = idx % head_v_dim;
    const size_t conv_dim = 2 * (size_t)key_dim + value_dim;
    const size_t total_proj_dim = conv_dim + value_dim + 2 * (size_t)num_v_heads;
    const size_t hk = hv / (num_v_heads / num_k_heads);
    float* srow = state + (hv * head_v_dim + i) * head_k_dim;
    for (size_t token = 0; token < suffix_len; ++token) {
        const size_t qk_off = token * key_dim + hk * head_k_dim;
        const float decay = decay_buf[token * num_v_heads + hv];
        const float beta = beta_buf[token * num_v_heads + hv];
        float kv_mem = 0.0f;
        for (size_t j = 0; j < head_k_dim; ++j) kv_mem += (decay * srow[j]) * k_norm[qk_off + j];
        const float v_val = bf16_to_f32(in_proj[token * total_proj_dim + 2 * key_dim + hv * head_v_dim + i]);
        const float delta = beta * (v_val - kv_mem);
        float o_val = 0.0f;
        for (size_t j = 0; j < head_k_dim; ++j) {
            const float new_s = decay * srow[j] + k_norm[qk_off + j] * delta;
            srow[j] = new_s;
            o_val += new_s * q_norm[qk_off + j];
        }
        out[token * value_dim + hv * head_v_dim + i] = f32_to_bf16(o_val);
    }
}
uzu_status delta_net_prefill_exact(hipStream_t s, const float* q_norm, const float* k_norm, const float* beta, const float* decay, const uint16_t* in_proj, float* state, uint16_t* out,
                                   uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len) {
    const size_t rows = (size_t)num_v_heads * head_v_dim;
    if (!rows || !suffix_len) return UZU_OK;
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_prefill_exact_kernel, dim3((uint32_t)((rows + 63) / 64)), dim3(64), 0, s, q_norm, k_norm, beta, decay, in_proj, state, out, num_v_heads, num_k_heads,
                           head_k_dim, head_v_dim, key_dim, value_dim, suffix_len);
    }, "delta_net_prefill_exact");
}
// cpu/kernel/gdn/norm_gate.rs:32-66: one thread per (token, head).
__global__ void __launch_bounds__(64) delta_net_norm_gate_exact_kernel(uint16_t* in_out, const uint16_t* in_proj, const float* norm_weight, uint32_t num_v_heads, uint32_t head_v_dim,
                                                                        uint32_t value_dim, uint32_t conv_dim, uint32_t total_proj_dim, float norm_epsilon, uint32_t suffix_len) {
    const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= (size_t)suffix_len * num_v_heads) return;
    const size_t token = idx / num_v_heads, hv = idx % num_v_heads;
    const size_t base = token * value_dim + hv * head_v_dim;
    float sumsq = 0.0f;
    for (size_t i = 0; i < head_v_dim; ++i) {
        const float val = bf16_to_f32(in_out[base + i]);
        sumsq += val * val;
    }
    const float inv_rms = 1.0f / sqrtf(sumsq / (float)head_v_dim + norm_epsilon);
    for (size_t i = 0; i < head_v_dim; ++i) {
        const float o_i = bf16_to_f32(in_out[base + i]);
        const float z_i = bf16_to_f32(in_proj[token * (size_t)total_proj_dim + conv_dim + hv * head_v_dim + i]);
        const float final_val = o_i * inv_rms * norm_weight[i] * silu_f32(z_i);
        in_out[base + i] = f32_to_bf16(final_val);
    }
}
uzu_status delta_net_norm_gate_exact(hipStream_t s, uint16_t* in_out, const uint16_t* in_proj, const float* norm_weight, uint32_t num_v_heads, uint32_t head_v_dim, uint32_t value_dim,
                                     uint32_t conv_dim, uint32_t total_proj_dim, float norm_epsilon, uint32_t suffix_len) {
    const size_t total = (size_t)suffix_len * num_v_heads;
    if (!total) return UZU_OK;
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_norm_gate_exact_kernel, dim3((uint32_t)((total + 63) / 64)), dim3(64), 0, s, in_out, in_proj, norm_weight, num_v_heads, head_v_dim, value_dim, conv_dim,
                           total_proj_dim, norm_epsilon, suffix_len);
    }, "delta_net_norm_gate_exact");
}

} // namespace k
} // namespace uzu
