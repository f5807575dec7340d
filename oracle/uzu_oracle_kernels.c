/*
 * uzu_oracle_kernels.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).  See uzu_oracle.h.
 *
 * Each function restates one kernel of the reference's CPU backend
 * (crates/backend-uzu/src/backends/cpu/kernel/**, abbreviated BU/cpu/... below) loop for loop.
 * Values of element type T are carried as `float` holding an exactly representable T value;
 * `rnd(dtype, x)` is the `T::from(x)` / store conversion (bf16: round to nearest even).
 * OpenMP is applied only across independent output elements, never inside a reduction, so results
 * do not depend on the thread count.
 */
#include "uzu_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* half 2.7: bf16::from_f32 (round to nearest even, NaN quieted) / to_f32 (exact) */
uint16_t orc_f32_to_bf16(float v) {
    uint32_t x;
    memcpy(&x, &v, 4);
    if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((x >> 16) | 0x0040u);
    const uint32_t round_bit = 0x00008000u;
    if ((x & round_bit) != 0 && (x & (3u * round_bit - 1u)) != 0) return (uint16_t)((x >> 16) + 1u);
    return (uint16_t)(x >> 16);
}
float orc_bf16_to_f32(uint16_t v) {
    uint32_t x = ((uint32_t)v) << 16;
    float f;
    memcpy(&f, &x, 4);
    return f;
}
void orc_f32_to_bf16_array(const float* src, uint16_t* dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = orc_f32_to_bf16(src[i]);
}
void orc_bf16_to_f32_array(const uint16_t* src, float* dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = orc_bf16_to_f32(src[i]);
}

/* BU/cpu/kernel/matmul/reference.rs:113-143 */
static inline float rd(const void* base, uint32_t dt, size_t i) {
    return dt == ORC_F32 ? ((const float*)base)[i] : orc_bf16_to_f32(((const uint16_t*)base)[i]);
}
static inline void wr(void* base, uint32_t dt, size_t i, float v) {
    if (dt == ORC_F32)
        ((float*)base)[i] = v;
    else
        ((uint16_t*)base)[i] = orc_f32_to_bf16(v);
}
static inline float rnd(uint32_t dt, float v) { return dt == ORC_F32 ? v : orc_bf16_to_f32(orc_f32_to_bf16(v)); }

/* ------------------------------------------------------------------ MatmulKernel
 * BU/cpu/kernel/matmul/kernel.rs:164-293 (quantized B: element index n*K+k, little-nibble-first
 * u32 words :236-242; per-element dequant scale*q + bias_term :267-275; sequential f32
 * accumulation :278; epilogue ab_scale, accumulate, bias, soft-cap :281-292). */
void orc_matmul(const orc_matmul_args* g) {
    const size_t m = g->m, n = g->n, k = g->k;
    const int quant = g->method != UZU_QUANT_NONE;
    const uint32_t bits = g->bits;
    const size_t group_size = g->group_size;
    const size_t num_groups_k = quant ? (k + group_size - 1) / group_size : 0;
    const size_t zero_point_stride = quant ? (bits == 4 ? (num_groups_k + 1) / 2 : num_groups_k) : 0;
    const size_t pack_factor = bits == 4 ? 8 : 4;
    size_t ld = g->b_leading_dimension ? g->b_leading_dimension : (g->b_transpose ? k : n);

    for (size_t row = 0; row < m; ++row) {
#pragma omp parallel for schedule(static) if (n * k >= 262144)
        for (size_t col = 0; col < n; ++col) {
            const size_t b_col = g->gather_indices ? g->gather_indices[row * n + col] : col;
            float accumulator = 0.0f;
            for (size_t inner = 0; inner < k; ++inner) {
                float a_value;
                if (g->a) {
                    a_value = rd(g->a, g->a_dtype, row * k + inner);
                } else { /* AData::Int8 (kernel.rs:190-200): q * scale of the activation group */
                    const size_t groups = (k + g->a_group_size - 1) / g->a_group_size;
                    a_value = (float)g->a_q[row * k + inner] * g->a_scales[row * groups + inner / g->a_group_size];
                }
                float b_value;
                if (!quant) {
                    const size_t index = g->b_transpose ? b_col * ld + inner : inner * ld + b_col;
                    b_value = rd(g->b, g->w_dtype, index);
                } else {
                    const size_t weight_linear_index = b_col * k + inner;
                    const size_t word_index = weight_linear_index / pack_factor;
                    const uint32_t bit_offset = (uint32_t)(weight_linear_index % pack_factor) * bits;
                    uint32_t word;
                    memcpy(&word, (const uint8_t*)g->b + word_index * 4, 4); /* read_unaligned */
                    const uint32_t code_mask = (1u << bits) - 1u;
                    uint8_t weight_code = (uint8_t)((word >> bit_offset) & code_mask);
                    if (g->signed_codes) weight_code ^= (uint8_t)(1u << (bits - 1));
                    const float quantized_value = (float)weight_code;
                    const size_t group_index = inner / group_size;
                    const float scale = rd(g->scales, g->w_dtype, b_col * num_groups_k + group_index);
                    const float midpoint = (float)(1u << (bits - 1));
                    float bias_term;
                    if (g->zero_points) {
                        float zp;
                        if (bits == 4) {
                            const uint8_t byte_value = g->zero_points[b_col * zero_point_stride + (group_index >> 1)];
                            zp = (group_index & 1) == 0 ? (float)(byte_value & 0x0F) : (float)((byte_value >> 4) & 0x0F);
                        } else {
                            zp = (float)g->zero_points[b_col * zero_point_stride + group_index];
                        }
                        bias_term = -scale * zp;
                    } else if (g->biases) {
                        bias_term = rd(g->biases, g->w_dtype, b_col * num_groups_k + group_index);
                    } else {
                        bias_term = -scale * midpoint;
                    }
                    b_value = scale * quantized_value + bias_term;
                }
                accumulator += a_value * b_value;
            }
            const size_t output_index = row * n + col;
            float value = g->ab_scale * accumulator;
            if (g->accumulate) value += rd(g->d, g->d_dtype, output_index);
            if (g->bias && !g->rht_factors) value += rd(g->bias, g->w_dtype, col); /* bias_after_rht */
            if (g->has_soft_cap) value = g->soft_cap * tanhf(value / g->soft_cap);
            wr(g->d, g->d_dtype, output_index, value);
        }
    }
    if (g->rht_factors) { /* kernel.rs:296-303: OutputRht in place on D, then TensorAddBias */
        orc_activation_transform(NULL, g->d, NULL, NULL, NULL, g->rht_factors, g->d_dtype, g->m, g->n, 1u, 0u, 0u);
        if (g->bias)
            for (size_t i = 0; i < m * n; ++i) wr(g->d, g->d_dtype, i, rd(g->d, g->d_dtype, i) + rd(g->bias, g->w_dtype, i % n)); /* tensor_add_bias.rs */
    }
}

/* ------------------------------------------------------------------ ActivationTransform
 * BU/cpu/kernel/activation_transform/mod.rs:9-47 */
static float orc_min_max_symmetric_divisor(const float* values, size_t n) {
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = 0; i < n; ++i) {
        mn = fminf(mn, values[i]); /* f32::min / max: the non-NaN operand wins, as fminf / fmaxf */
        mx = fmaxf(mx, values[i]);
    }
    const float magnitude = fmaxf(fabsf(mn), fabsf(mx));
    return (isfinite(magnitude) && magnitude > 0.0f) ? magnitude / 127.0f : 1.0f;
}
static int8_t orc_quantize_symmetric_i8(float value, float divisor) {
    float r = roundf(value / divisor); /* f32::round: half away from zero */
    if (r < -127.0f) r = -127.0f;
    if (r > 127.0f) r = 127.0f;
    return (int8_t)r;
}
static void orc_hadamard32(float* values) {
    for (size_t stride = 1; stride < 32; stride <<= 1)
        for (size_t lane = 0; lane < 32; ++lane)
            if ((lane & stride) == 0) {
                const float a = values[lane], b = values[lane | stride];
                values[lane] = a + b;
                values[lane | stride] = a - b;
            }
    const float scale = 1.0f / sqrtf(32.0f);
    for (size_t i = 0; i < 32; ++i) values[i] *= scale;
}
/* quantize_transformed_row (activation_transform.rs:11-41): symmetric int8 per scale group, optional i32 code sums per sum group */
static void orc_quantize_transformed_row(const float* transformed, size_t columns, size_t activation_scale_group_size, size_t sum_group_size,
                                         int8_t* values, float* scales, int32_t* group_sums) {
    const size_t scales_per_row = columns / activation_scale_group_size;
    if (group_sums) memset(group_sums, 0, columns / sum_group_size * sizeof(int32_t));
    for (size_t gidx = 0; gidx < scales_per_row; ++gidx) {
        const float* source = transformed + gidx * activation_scale_group_size;
        const float scale = orc_min_max_symmetric_divisor(source, activation_scale_group_size);
        scales[gidx] = scale;
        for (size_t i = 0; i < activation_scale_group_size; ++i) {
            const int8_t code = orc_quantize_symmetric_i8(source[i], scale);
            const size_t absolute_index = gidx * activation_scale_group_size + i;
            values[absolute_index] = code;
            if (group_sums) group_sums[absolute_index / sum_group_size] += (int32_t)code;
        }
    }
}
/* activation_transform.rs:43-136 */
void orc_activation_transform(const void* input, void* fp_out, int8_t* q_out, float* scales_out, int32_t* group_sums_out,
                              const int32_t* rht_factors, uint32_t dtype, uint32_t batch_size, uint32_t element_count, uint32_t op,
                              uint32_t activation_scale_group_size, uint32_t sum_group_size) {
    const void* in = input ? input : fp_out;
    const size_t rows = batch_size, columns = element_count;
    const int input_rht = op != 1u;
    const int quantize = op == 2u || op == 3u;
    float* transformed = (float*)malloc(columns * sizeof(float));
    for (size_t row = 0; row < rows; ++row) {
        const size_t row_offset = row * columns;
        for (size_t stripe_start = 0; stripe_start < columns; stripe_start += 32) {
            float stripe[32];
            for (size_t lane = 0; lane < 32; ++lane) {
                const size_t index = stripe_start + lane;
                const float value = rd(in, dtype, row_offset + index);
                const float factor = (float)rht_factors[index];
                stripe[lane] = input_rht ? value * factor : value;
            }
            orc_hadamard32(stripe);
            for (size_t lane = 0; lane < 32; ++lane) {
                const size_t index = stripe_start + lane;
                const float factor = (float)rht_factors[index];
                transformed[index] = input_rht ? stripe[lane] : stripe[lane] * factor;
            }
        }
        if (quantize) {
            orc_quantize_transformed_row(transformed, columns, activation_scale_group_size, op == 3u ? sum_group_size : 0, q_out + row_offset,
                                         scales_out + row * (columns / activation_scale_group_size),
                                         op == 3u ? group_sums_out + row * (columns / sum_group_size) : NULL);
        } else {
            for (size_t index = 0; index < columns; ++index) wr(fp_out, dtype, row_offset + index, transformed[index]);
        }
    }
    free(transformed);
}

/* ------------------------------------------------------------------ Normalization
 * BU/cpu/kernel/normalization/normalization.rs:56-125 (AccumT = f32). */
void orc_normalization(const orc_norm_args* g) {
    const uint32_t dt = g->io_dtype;
    const void* input = g->input ? g->input : g->output;
    const size_t element_count = g->element_count;
    const float element_count_accum = (float)element_count;
#pragma omp parallel for schedule(static) if (g->batch_size >= 16)
    for (size_t batch = 0; batch < g->batch_size; ++batch) {
        const size_t off = batch * element_count;
        float sum = 0.0f, sum_sq = 0.0f;
        for (size_t i = 0; i < element_count; ++i) {
            float val = rd(input, dt, off + i);
            if (g->copy_to_shortcut) {
                if (g->residual_add) {
                    val = rnd(dt, val + rd(g->shortcut, dt, off + i)); /* InputT add */
                    if (g->scale_residual_sum) val = rnd(dt, val * g->post_layer_scalar);
                }
                wr(g->shortcut, dt, off + i, val);
            }
            const float accum_val = val;
            if (g->subtract_mean) sum = sum + accum_val;
            sum_sq = sum_sq + accum_val * accum_val;
        }
        const float mean = g->subtract_mean ? sum / element_count_accum : 0.0f;
        const float variance = sum_sq / element_count_accum - mean * mean;
        const float rms_inv = 1.0f / sqrtf(variance + g->epsilon);
        for (size_t i = 0; i < element_count; ++i) {
            const float input_val = g->residual_add ? rd(g->shortcut, dt, off + i) : rd(input, dt, off + i);
            const float normalized = (input_val - mean) * rms_inv;
            float result;
            if (g->scales) {
                const float scale_val = rd(g->scales, g->affine_dtype, i);
                if (g->full_layer) {
                    result = rnd(dt, normalized * (scale_val + g->scale_offset));
                } else {
                    const float normalized_out = rnd(dt, normalized);
                    const float scale_out = rnd(dt, scale_val + g->scale_offset);
                    result = rnd(dt, normalized_out * scale_out); /* OutputT mul */
                }
            } else {
                result = rnd(dt, normalized);
            }
            if (g->biases) result = rnd(dt, result + rd(g->biases, g->affine_dtype, i));
            if (g->scale_output) result = rnd(dt, result * rnd(dt, g->post_layer_scalar));
            wr(g->output, dt, off + i, result);
        }
    }
}

/* ------------------------------------------------------------------ QKVNorm
 * BU/cpu/kernel/attention/qkv_norm.rs:32-77 (in place, AccumT = f32, ScaleT = f32). */
void orc_qkv_norm(void* qkv, uint32_t dt, const float* scales, uint32_t batch_size, uint32_t total_heads,
                  uint32_t head_dim, float epsilon, float scale_offset, uint32_t head_offset, uint32_t head_count,
                  uint32_t full_layer) {
    const size_t qkv_stride = (size_t)total_heads * head_dim;
    for (size_t batch = 0; batch < batch_size; ++batch) {
        for (size_t head = 0; head < head_count; ++head) {
            const size_t offset = batch * qkv_stride + (head_offset + head) * head_dim;
            float total_sum = 0.0f;
            for (size_t i = 0; i < head_dim; ++i) {
                const float v = rd(qkv, dt, offset + i);
                total_sum = total_sum + v * v;
            }
            const float mean_square = total_sum / (float)head_dim;
            const float rms_norm = 1.0f / sqrtf(mean_square + epsilon);
            for (size_t i = 0; i < head_dim; ++i) {
                const float normalized = rd(qkv, dt, offset + i) * rms_norm;
                float result;
                if (!scales) {
                    result = rnd(dt, normalized);
                } else if (full_layer) {
                    result = rnd(dt, normalized * (scales[i] + scale_offset));
                } else {
                    result = rnd(dt, rnd(dt, normalized) * rnd(dt, scales[i] + scale_offset));
                }
                wr(qkv, dt, offset + i, result);
            }
        }
    }
}

/* ------------------------------------------------------------------ host RoPE tables
 * BU/encodable_block/mixer/attention/rope.rs:13-114 (Unscaled, Linear, Llama-3). */
void orc_rope_tables(const uzu_rope_desc* rope, const uint32_t* token_positions, uint32_t n_pos, float* cosines,
                     float* sines) {
    const uint32_t head_dim = rope->head_dim, half_dim = head_dim / 2;
    float attention_scaling_factor = 1.0f; /* rope.rs:21-27 */
    if (rope->kind == UZU_ROPE_YARN) attention_scaling_factor = 0.1f * logf(rope->scaling_factor) + 1.0f;
    else if (rope->kind == UZU_ROPE_LONGROPE && rope->scaling_factor > 1.0f)
        attention_scaling_factor = sqrtf(1.0f + logf(rope->scaling_factor) / logf((float)rope->original_context_length));
    for (uint32_t pair_index = 0; pair_index < half_dim; ++pair_index) {
        const uint32_t channel_index = pair_index * 2;
        float inverse_frequency = 1.0f / powf(rope->base, (float)channel_index / (float)head_dim);
        if (rope->kind == UZU_ROPE_LINEAR) {
            inverse_frequency = inverse_frequency / rope->scaling_factor;
        } else if (rope->kind == UZU_ROPE_LLAMA) {
            const float low_frequency_wavelength = (float)rope->original_context_length / rope->low_frequency_factor;
            const float high_frequency_wavelength = (float)rope->original_context_length / rope->high_frequency_factor;
            const float wavelength = 2.0f * 3.14159265358979323846f / inverse_frequency;
            const float scaled_frequency = inverse_frequency / rope->scaling_factor;
            if (wavelength < high_frequency_wavelength) {
                /* unchanged */
            } else if (wavelength > low_frequency_wavelength) {
                inverse_frequency = scaled_frequency;
            } else {
                float smoothing_factor = (float)rope->original_context_length / wavelength - rope->low_frequency_factor;
                smoothing_factor = smoothing_factor / (rope->high_frequency_factor - rope->low_frequency_factor);
                inverse_frequency = smoothing_factor * inverse_frequency + (1.0f - smoothing_factor) * scaled_frequency;
            }
        } else if (rope->kind == UZU_ROPE_YARN) { /* rope.rs:60-81 (double for the ramp bounds, as the reference) */
            const double dim = (double)rope->head_dim, base = (double)rope->base, original_context_length = (double)rope->original_context_length;
            double low = dim * log(original_context_length / ((double)rope->beta_fast * 2.0 * 3.14159265358979323846)) / (2.0 * log(base));
            double high = dim * log(original_context_length / ((double)rope->beta_slow * 2.0 * 3.14159265358979323846)) / (2.0 * log(base));
            if (rope->truncate) low = floor(low), high = ceil(high);
            const float low_f = (float)(low > 0.0 ? low : 0.0);
            float high_f = (float)(high < (double)(rope->head_dim - 1) ? high : (double)(rope->head_dim - 1));
            if (low_f == high_f) high_f += 0.001f;
            float ramp = ((float)pair_index - low_f) / (high_f - low_f);
            ramp = ramp < 0.0f ? 0.0f : (ramp > 1.0f ? 1.0f : ramp);
            const float smoothing_factor = 1.0f - ramp;
            const float scaled_frequency = inverse_frequency / rope->scaling_factor;
            inverse_frequency = scaled_frequency * (1.0f - smoothing_factor) + inverse_frequency * smoothing_factor;
        } else if (rope->kind == UZU_ROPE_LONGROPE) { /* rope.rs:82-89 */
            const float* factors = rope->max_sequence_length > rope->original_context_length ? rope->long_factor : rope->short_factor;
            inverse_frequency = inverse_frequency / factors[pair_index];
        }
        for (uint32_t token_index = 0; token_index < n_pos; ++token_index) {
            const float embedding = (float)token_positions[token_index] * inverse_frequency;
            const float sine = sinf(embedding) * attention_scaling_factor;
            const float cosine = cosf(embedding) * attention_scaling_factor;
            const size_t pair_offset = (size_t)token_index * head_dim + pair_index;
            sines[pair_offset] = sine;
            sines[pair_offset + half_dim] = sine;
            cosines[pair_offset] = cosine;
            cosines[pair_offset + half_dim] = cosine;
        }
    }
}

/* ------------------------------------------------------------------ AttentionPrepare
 * BU/cpu/kernel/attention/attention_prepare.rs:7-126. */
static inline uint16_t apply_rope(const uint16_t* head, const float* cosines, const float* sines, size_t batch_idx,
                                  size_t head_dim_idx, size_t rope_dim) {
    const size_t half = rope_dim / 2;
    const size_t paired_idx = head_dim_idx < half ? head_dim_idx + half : head_dim_idx - half;
    const float input = orc_bf16_to_f32(head[head_dim_idx]);
    const float paired = orc_bf16_to_f32(head[paired_idx]);
    const float signed_paired = head_dim_idx < half ? -paired : paired;
    const float cos_val = cosines[batch_idx * rope_dim + head_dim_idx];
    const float sin_val = sines[batch_idx * rope_dim + head_dim_idx];
    return orc_f32_to_bf16(input * cos_val + signed_paired * sin_val);
}
void orc_attention_prepare(const uint16_t* qkv, uint16_t* queries, uint16_t* keys, uint16_t* values,
                           const float* cosines, const float* sines, uint32_t num_q_heads, uint32_t num_kv_heads,
                           uint32_t head_dim, uint32_t rope_dim, uint32_t kv_token_offset, uint32_t batch_dim,
                           uint32_t has_kv) {
    const size_t total_heads = has_kv ? num_q_heads + 2 * (size_t)num_kv_heads : num_q_heads;
    for (size_t batch_idx = 0; batch_idx < batch_dim; ++batch_idx) {
        for (size_t head_idx = 0; head_idx < total_heads; ++head_idx) {
            const uint16_t* qkv_head = qkv + batch_idx * total_heads * head_dim + head_idx * head_dim;
            const int is_query = !has_kv || head_idx < num_q_heads;
            const int is_key = has_kv && head_idx >= num_q_heads && head_idx < (size_t)num_q_heads + num_kv_heads;
            for (size_t d = 0; d < head_dim; ++d) {
                uint16_t element = qkv_head[d];
                if (rope_dim && d < rope_dim && (is_query || is_key))
                    element = apply_rope(qkv_head, cosines, sines, batch_idx, d, rope_dim);
                if (is_query) {
                    queries[head_idx * batch_dim * head_dim + batch_idx * head_dim + d] = element;
                } else if (is_key) {
                    keys[(kv_token_offset + batch_idx) * num_kv_heads * head_dim + (head_idx - num_q_heads) * head_dim + d] =
                        element;
                } else {
                    values[(kv_token_offset + batch_idx) * num_kv_heads * head_dim +
                           (head_idx - num_q_heads - num_kv_heads) * head_dim + d] = element;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ attention mask
 * BU/cpu/kernel/attention/mask.rs:3-61. */
static inline int should_use_key(const orc_attention_args* a, uint32_t q_seq_idx, uint32_t prefix_length,
                                 uint32_t suffix_position, uint32_t query_position, uint32_t i) {
    int use_key = 1;
    uint32_t key_position;
    if (i >= prefix_length) {
        const uint32_t key_position_in_suffix = i - prefix_length;
        if (a->trie) {
            const uint32_t* trie_node = a->trie + 3 * (size_t)key_position_in_suffix; /* {trie_start, trie_end, height} */
            key_position = suffix_position + trie_node[2];
            if (a->is_causal) use_key &= q_seq_idx >= trie_node[0] && q_seq_idx <= trie_node[1];
        } else {
            key_position = suffix_position + key_position_in_suffix;
            if (a->is_causal) use_key &= key_position_in_suffix <= q_seq_idx;
        }
    } else {
        if (a->is_kv_cache_ring) {
            key_position = (prefix_length + i - a->ring_offset) % prefix_length;
            use_key &= key_position < a->ring_length;
        } else {
            key_position = i;
        }
    }
    if (a->is_sliding_window) {
        const uint32_t w = a->sliding_window_size;
        if (a->is_causal)
            use_key &= key_position <= query_position && (query_position - key_position) < w;
        else if (key_position <= query_position)
            use_key &= (query_position - key_position) <= w / 2;
        else
            use_key &= (key_position - query_position) <= w / 2;
    }
    return use_key;
}

/* ------------------------------------------------------------------ AttentionSinglePass
 * BU/cpu/kernel/attention/attention_single_pass.rs:37-127. */
void orc_attention_single_pass(const orc_attention_args* a, void* out) {
    const uint32_t HD = a->head_dim, dt = a->dtype;
    const uint32_t prefix_length = a->sequence_length - a->suffix_length;
    const uint32_t suffix_position = a->is_kv_cache_ring ? a->ring_length : prefix_length;
    const size_t total = (size_t)a->num_heads * a->suffix_length;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t hq = 0; hq < total; ++hq) {
        const uint32_t head_idx = (uint32_t)(hq / a->suffix_length), q_seq_idx = (uint32_t)(hq % a->suffix_length);
        const uint32_t kv_head_idx = head_idx / a->gqa_factor;
        const size_t o_offset = (size_t)q_seq_idx * a->num_heads + head_idx;
        const size_t q_offset = (size_t)head_idx * a->suffix_length + q_seq_idx;
        const uint32_t query_position = suffix_position + (a->trie ? a->trie[3 * (size_t)q_seq_idx + 2] : q_seq_idx); /* attention_single_pass.rs:55-61 */
        float* q = (float*)malloc(sizeof(float) * HD * 2);
        float* o = q + HD;
        for (uint32_t j = 0; j < HD; ++j) {
            q[j] = a->scale * rd(a->queries, dt, q_offset * HD + j);
            o[j] = 0.0f;
        }
        float max_score = -INFINITY, sum_exp_score = 0.0f;
        if (a->sinks) {
            max_score = rd(a->sinks, dt, head_idx % a->num_heads);
            sum_exp_score = 1.0f;
        }
        for (uint32_t i = 0; i < a->sequence_length; ++i) {
            if (!should_use_key(a, q_seq_idx, prefix_length, suffix_position, query_position, i)) continue;
            const size_t kb = (size_t)kv_head_idx * a->k_head_stride + (size_t)i * a->k_seq_stride;
            float score = 0.0f;
            for (uint32_t j = 0; j < HD; ++j) score += q[j] * rd(a->keys, dt, kb + j);
            const float new_max = fmaxf(max_score, score);
            const float factor = expf(max_score - new_max);
            const float exp_score = expf(score - new_max);
            max_score = new_max;
            sum_exp_score = sum_exp_score * factor + exp_score;
            const size_t vb = (size_t)kv_head_idx * a->v_head_stride + (size_t)i * a->v_seq_stride;
            for (uint32_t j = 0; j < HD; ++j) o[j] = o[j] * factor + exp_score * rd(a->values, dt, vb + j);
        }
        for (uint32_t j = 0; j < HD; ++j) wr(out, dt, o_offset * HD + j, o[j] / sum_exp_score);
        free(q);
    }
}

/* ------------------------------------------------------------------ AttentionTwoPass1/2
 * BU/cpu/kernel/attention/attention_two_pass.rs:41-190 (32 stride-interleaved key blocks). */
#define ORC_TOTAL_BLOCKS_COUNT 32u
void orc_attention_two_pass1(const orc_attention_args* a, float* partials, float* sums, float* maxs) {
    const uint32_t HD = a->head_dim, dt = a->dtype;
    const uint32_t prefix_length = a->sequence_length - a->suffix_length;
    const uint32_t suffix_position = a->is_kv_cache_ring ? a->ring_length : prefix_length;
    const size_t total = (size_t)a->num_heads * a->suffix_length * ORC_TOTAL_BLOCKS_COUNT;
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t idx = 0; idx < total; ++idx) {
        const uint32_t block_idx = (uint32_t)(idx % ORC_TOTAL_BLOCKS_COUNT);
        const size_t hq = idx / ORC_TOTAL_BLOCKS_COUNT;
        const uint32_t head_idx = (uint32_t)(hq / a->suffix_length), q_seq_idx = (uint32_t)(hq % a->suffix_length);
        const uint32_t query_position = suffix_position + (a->trie ? a->trie[3 * (size_t)q_seq_idx + 2] : q_seq_idx); /* attention_single_pass.rs:55-61 */
        const size_t o_offset = (size_t)q_seq_idx * a->num_heads + head_idx;
        const size_t q_offset = (size_t)head_idx * a->suffix_length + q_seq_idx;
        const uint32_t kv_head_idx = head_idx / a->gqa_factor;
        float* q = (float*)malloc(sizeof(float) * HD * 2);
        float* o = q + HD;
        for (uint32_t j = 0; j < HD; ++j) {
            q[j] = a->scale * rd(a->queries, dt, q_offset * HD + j);
            o[j] = 0.0f;
        }
        float max_score = -1e9f, sum_exp_score = 0.0f;
        if (a->sinks && block_idx == 0) {
            max_score = rd(a->sinks, dt, head_idx);
            sum_exp_score = 1.0f;
        }
        for (uint32_t i = block_idx; i < a->sequence_length; i += ORC_TOTAL_BLOCKS_COUNT) {
            if (!should_use_key(a, q_seq_idx, prefix_length, suffix_position, query_position, i)) continue;
            const size_t kb = (size_t)kv_head_idx * a->k_head_stride + (size_t)i * a->k_seq_stride;
            float score = 0.0f;
            for (uint32_t j = 0; j < HD; ++j) score += q[j] * rd(a->keys, dt, kb + j);
            const float new_max = fmaxf(max_score, score);
            const float factor = expf(max_score - new_max);
            const float exp_score = expf(score - new_max);
            max_score = new_max;
            sum_exp_score = sum_exp_score * factor + exp_score;
            const size_t vb = (size_t)kv_head_idx * a->v_head_stride + (size_t)i * a->v_seq_stride;
            for (uint32_t j = 0; j < HD; ++j) o[j] = o[j] * factor + exp_score * rd(a->values, dt, vb + j);
        }
        float* out_base = partials + (o_offset * ORC_TOTAL_BLOCKS_COUNT + block_idx) * HD;
        for (uint32_t j = 0; j < HD; ++j) out_base[j] = o[j];
        sums[o_offset * ORC_TOTAL_BLOCKS_COUNT + block_idx] = sum_exp_score;
        maxs[o_offset * ORC_TOTAL_BLOCKS_COUNT + block_idx] = max_score;
        free(q);
    }
}
void orc_attention_two_pass2(const float* partials, const float* sums, const float* maxs, void* out, uint32_t dt,
                             uint32_t HD, uint32_t num_heads, uint32_t suffix_length) {
    const size_t total = (size_t)num_heads * suffix_length;
#pragma omp parallel for schedule(static)
    for (size_t o_offset = 0; o_offset < total; ++o_offset) {
        const float* mx = maxs + o_offset * ORC_TOTAL_BLOCKS_COUNT;
        const float* sm = sums + o_offset * ORC_TOTAL_BLOCKS_COUNT;
        float global_max = -INFINITY;
        for (uint32_t b = 0; b < ORC_TOTAL_BLOCKS_COUNT; ++b) global_max = fmaxf(global_max, mx[b]);
        float global_sum = 0.0f;
        for (uint32_t b = 0; b < ORC_TOTAL_BLOCKS_COUNT; ++b) global_sum += sm[b] * expf(mx[b] - global_max);
        for (uint32_t j = 0; j < HD; ++j) {
            float val = 0.0f;
            for (uint32_t b = 0; b < ORC_TOTAL_BLOCKS_COUNT; ++b)
                val += partials[(o_offset * ORC_TOTAL_BLOCKS_COUNT + b) * HD + j] * expf(mx[b] - global_max);
            wr(out, dt, o_offset * HD + j, val / global_sum);
        }
    }
}

/* ------------------------------------------------------------------ small ops */
/* BU/cpu/kernel/attention/kv_cache_update.rs:9-28 */
void orc_kv_cache_update(void* keys, void* values, uint32_t dt, const orc_copy* copies, uint32_t copy_count,
                         uint32_t element_dim) {
    const size_t esz = dt == ORC_F32 ? 4 : 2;
    for (size_t e = 0; e < element_dim; ++e)
        for (size_t i = 0; i < copy_count; ++i) {
            const size_t s = ((size_t)copies[i].source * element_dim + e) * esz;
            const size_t d = ((size_t)copies[i].destination * element_dim + e) * esz;
            memcpy((uint8_t*)keys + d, (uint8_t*)keys + s, esz);
            memcpy((uint8_t*)values + d, (uint8_t*)values + s, esz);
        }
}
/* BU/cpu/kernel/attention/sigmoid_gate.rs:9-22 */
void orc_sigmoid_gate(const void* gate, void* output, uint32_t dt, uint32_t total) {
    for (size_t idx = 0; idx < total; ++idx) {
        const float g = rd(gate, dt, idx);
        const float sigmoid = 1.0f / (1.0f + expf(-g));
        wr(output, dt, idx, rd(output, dt, idx) * sigmoid);
    }
}
/* BU/backends/common/gpu_types/activation_type.rs:16-65: activate<T>(x: T) -> T */
float orc_activate(uint32_t act, float x, uint32_t dt) {
    switch (act) {
    case UZU_ACT_SILU: return rnd(dt, x / (1.0f + expf(-1.0f * x)));
    case UZU_ACT_GELU_APPROX: {
        const float tan_arg = 0.7978846f * (x + 0.044715f * x * x * x);
        return rnd(dt, 0.5f * x * (1.0f + tanhf(tan_arg)));
    }
    case UZU_ACT_GELU_EXACT: return rnd(dt, 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)));
    case UZU_ACT_IDENTITY: return x;
    case UZU_ACT_SOFTPLUS:
        if (x > 20.0f) return x;
        return rnd(dt, logf(1.0f + expf(x)));
    default: return x;
    }
}
/* BU/cpu/kernel/gated_act_mul/gated_act_mul.rs:36-70 + mod.rs:5-12 (FullPrecision op, no RHT):
 * result = (value * act(gate)) in T, -> f32 -> T. */
void orc_gated_act_mul(const void* act_operand, const void* value_operand, void* fp_out, uint32_t dt,
                       uint32_t gated_dim, uint32_t batch_dim, uint32_t value_offset, uint32_t value_row_stride,
                       uint32_t act_type, uint32_t interleaved) {
#pragma omp parallel for schedule(static) if (batch_dim >= 16)
    for (size_t batch = 0; batch < batch_dim; ++batch)
        for (size_t gated = 0; gated < gated_dim; ++gated) {
            size_t act_index;
            float value;
            if (interleaved) {
                const size_t base = batch * 2 * gated_dim;
                act_index = base + gated_dim + gated;
                value = rd(act_operand, dt, base + gated);
            } else {
                act_index = batch * gated_dim + gated;
                value = rd(value_operand, dt, batch * value_row_stride + value_offset + gated);
            }
            const float gate = rd(act_operand, dt, act_index);
            const float result = rnd(dt, value * orc_activate(act_type, gate, dt));
            wr(fp_out, dt, batch * gated_dim + gated, result);
        }
}
/* BU/cpu/kernel/gated_act_mul/gated_act_mul.rs:47-118 with use_hadamard: the gated products of a row (rounded to T, mod.rs:5-12)
 * times the sign factors, 32-point butterflies, then either stored as T (ops 0 FullPrecision) or quantised (1 Quantize,
 * 2 QuantizeWithGroupSums: gpu_types/gated_act_mul.rs:3-7). */
void orc_gated_act_mul_rht(const void* act_operand, const void* value_operand, void* fp_out, int8_t* q_out, float* scales_out, int32_t* group_sums_out,
                           const int32_t* hadamard_factors, uint32_t dt, uint32_t gated_dim, uint32_t batch_dim, uint32_t value_offset,
                           uint32_t value_row_stride, uint32_t act_type, uint32_t interleaved, uint32_t ops, uint32_t activation_scale_group_size,
                           uint32_t sum_group_size) {
    float* transformed = (float*)malloc((size_t)gated_dim * sizeof(float));
    for (size_t batch = 0; batch < batch_dim; ++batch) {
        for (size_t gated = 0; gated < gated_dim; ++gated) {
            size_t act_index;
            float value;
            if (interleaved) {
                const size_t base = batch * 2 * gated_dim;
                act_index = base + gated_dim + gated;
                value = rd(act_operand, dt, base + gated);
            } else {
                act_index = batch * gated_dim + gated;
                value = rd(value_operand, dt, batch * value_row_stride + value_offset + gated);
            }
            const float gate = rd(act_operand, dt, act_index);
            transformed[gated] = rnd(dt, value * orc_activate(act_type, gate, dt));
        }
        for (size_t stripe_start = 0; stripe_start < gated_dim; stripe_start += 32) {
            float stripe[32];
            for (size_t lane = 0; lane < 32; ++lane) stripe[lane] = transformed[stripe_start + lane] * (float)hadamard_factors[stripe_start + lane];
            orc_hadamard32(stripe);
            memcpy(transformed + stripe_start, stripe, sizeof stripe);
        }
        if (ops != 0u) {
            orc_quantize_transformed_row(transformed, gated_dim, activation_scale_group_size, ops == 2u ? sum_group_size : 0, q_out + batch * gated_dim,
                                         scales_out + batch * (gated_dim / activation_scale_group_size),
                                         ops == 2u ? group_sums_out + batch * (gated_dim / sum_group_size) : NULL);
        } else {
            for (size_t gated = 0; gated < gated_dim; ++gated) wr(fp_out, dt, batch * gated_dim + gated, transformed[gated]);
        }
    }
    free(transformed);
}
/* BU/cpu/kernel/embedding/quant_embedding.rs:36-116 (U4 / U8 codes) */
void orc_quantized_embedding_lookup(const uint32_t* token_ids, const uint8_t* weights, const void* scales,
                                    const uint8_t* zero_points, const void* biases, void* output, uint32_t dt,
                                    uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim, float input_scale,
                                    uint32_t group_size, uint32_t bits, uint32_t method) {
    const uint32_t packing_divisor = 8 / bits;
    const size_t weights_stride = model_dim / packing_divisor;
    const size_t num_groups = (model_dim + group_size - 1) / group_size;
    const size_t zero_points_stride = bits == 4 ? (num_groups + 1) / 2 : num_groups;
    for (size_t b = 0; b < batch_size; ++b) {
        const uint32_t token_id = token_ids[b];
        for (size_t dim_idx = 0; dim_idx < model_dim; ++dim_idx) {
            const size_t out_idx = b * model_dim + dim_idx;
            if (token_id >= vocab_size) {
                wr(output, dt, out_idx, 0.0f);
                continue;
            }
            const size_t group_idx = dim_idx / group_size;
            const float scale = rd(scales, dt, (size_t)token_id * num_groups + group_idx);
            int32_t quantized_value;
            if (bits == 4) {
                const uint8_t packed = weights[(size_t)token_id * weights_stride + dim_idx / 2];
                quantized_value = (dim_idx & 1) == 0 ? (packed & 0x0F) : ((packed >> 4) & 0x0F);
            } else {
                quantized_value = weights[(size_t)token_id * weights_stride + dim_idx];
            }
            float bias;
            if (method == UZU_QUANT_SCALE_BIAS) {
                bias = rd(biases, dt, (size_t)token_id * num_groups + group_idx);
            } else if (method == UZU_QUANT_SCALE_ZERO_POINT) {
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
            wr(output, dt, out_idx, out_f);
        }
    }
}
/* BU/cpu/kernel/embedding/full_precision_embedding.rs:17-31: weights[idx] * T::from(input_scale) in T */
void orc_full_precision_embedding_lookup(const uint32_t* token_ids, const void* weights, void* output, uint32_t dt,
                                         uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim,
                                         float input_scale) {
    for (size_t b = 0; b < batch_size; ++b) {
        const uint32_t token_id = token_ids[b];
        for (size_t d = 0; d < model_dim; ++d) {
            if (token_id >= vocab_size)
                wr(output, dt, b * model_dim + d, 0.0f);
            else
                wr(output, dt, b * model_dim + d, rd(weights, dt, (size_t)token_id * model_dim + d) * rnd(dt, input_scale));
        }
    }
}
/* BU/cpu/kernel/logit_transform/logit_transform.rs:16-25 */
void orc_logit_transform(void* logits, uint32_t dt, uint32_t length, float scale, float soft_cap,
                         uint32_t has_soft_cap) {
    for (size_t p = 0; p < length; ++p) {
        float value = rd(logits, dt, p) * scale;
        if (has_soft_cap) value = tanhf(value / soft_cap) * soft_cap;
        wr(logits, dt, p, value);
    }
}
/* BU/cpu/kernel/tensor_add_bias/tensor_add_bias.rs */
void orc_tensor_add_bias(const void* input, const void* bias, void* output, uint32_t dt, uint32_t bias_dt,
                         uint32_t num_cols, uint32_t length) {
    const void* in = input ? input : output;
    for (size_t i = 0; i < length; ++i) wr(output, dt, i, rd(in, dt, i) + rd(bias, bias_dt, i % num_cols));
}
/* BU/cpu/kernel/tensor_add_scale/tensor_add_scale.rs */
void orc_tensor_add_scale(const void* input, const void* bias, void* output, uint32_t dt, uint32_t num_cols,
                          uint32_t length, float scale) {
    const void* in = input ? input : output;
    for (size_t i = 0; i < length; ++i) wr(output, dt, i, (rd(in, dt, i) + rd(bias, dt, i % num_cols)) * scale);
}
/* BU/cpu/kernel/tensor_add_swap/tensor_add_swap.rs */
void orc_tensor_add_swap(void* skip, void* main_buf, uint32_t dt, uint32_t length) {
    for (size_t i = 0; i < length; ++i) {
        const float r = rnd(dt, rd(skip, dt, i) + rd(main_buf, dt, i));
        wr(skip, dt, i, r);
        wr(main_buf, dt, i, r);
    }
}
void orc_tensor_copy(const void* src, void* dst, uint32_t dt, uint32_t length) {
    memcpy(dst, src, (size_t)length * (dt == ORC_F32 ? 4 : 2));
}
/* BU/cpu/kernel/sampling/unified_sampling.rs:90-98, greedy: max_by(partial_cmp.then(b.0.cmp(&a.0)))
 * => largest logit, ties -> lowest index; NaN compares Equal => index tie-break applies. */
void orc_argmax(const void* logits, uint32_t dt, uint32_t* output, uint32_t vocab_size, uint32_t batch_size) {
    for (size_t b = 0; b < batch_size; ++b) {
        size_t best = 0;
        float best_v = rd(logits, dt, b * vocab_size);
        for (size_t i = 1; i < vocab_size; ++i) {
            const float v = rd(logits, dt, b * vocab_size + i);
            /* Iterator::max_by keeps the LAST maximal element; ordering (a vs b) = cmp(value) then
             * reverse index, so an equal value with a larger index compares Less -> not taken. */
            if (v > best_v) {
                best_v = v;
                best = i;
            }
        }
        output[b] = (uint32_t)best;
    }
}

/* ------------------------------------------------------------------ stochastic sampling
 * Counter-based RNG of the sampler: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11;
 * the reference's copy: BU/encodable_block/sampling/gumbel.rs:1-57).  Pinned in tests/test_oracle_sampling.py by the
 * published Random123 known-answer vectors and by the reference's own unit test of unit_interval (gumbel_test.rs). */
void orc_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t ctr[4] = {ctr_in[0], ctr_in[1], ctr_in[2], ctr_in[3]};
    uint32_t key[2] = {key_in[0], key_in[1]};
    for (int round = 0; round < 10; ++round) {
        if (round) { /* philox4x32_bumpkey */
            key[0] += 0x9E3779B9u;
            key[1] += 0xBB67AE85u;
        }
        const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0], p1 = (uint64_t)0xCD9E8D57u * ctr[2];
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ ctr[1] ^ key[0], n2 = hi0 ^ ctr[3] ^ key[1];
        ctr[0] = n0, ctr[1] = lo1, ctr[2] = n2, ctr[3] = lo0;
    }
    for (int i = 0; i < 4; ++i) out[i] = ctr[i];
}
/* gumbel.rs:52-57: uniform in [2^-24, 1 - 2^-24] from the top 24 bits */
float orc_unit_interval(uint32_t word) {
    const uint32_t top = word >> 8;
    return (float)(top > 1u ? top : 1u) * (1.0f / 16777216.0f);
}
/* gumbel.rs:34-50: counter (offset, 0, 0, 0), key = the 64-bit seed, word selects one of the four outputs */
float orc_uniform_float(uint64_t key, uint32_t offset, uint32_t word) {
    const uint32_t ctr[4] = {offset, 0u, 0u, 0u}, k[2] = {(uint32_t)key, (uint32_t)(key >> 32)};
    uint32_t out[4];
    orc_philox4x32_10(ctr, k, out);
    return orc_unit_interval(out[word & 3u]);
}
float orc_gumbel_float(uint64_t key, uint32_t offset, uint32_t word) { /* gumbel.rs:59-64 */
    return -logf(-logf(orc_uniform_float(key, offset, word)));
}
/* gumbel.rs:66-81: (counter offset, word) of logit `logit_idx` -- the numbering of the GPU kernel's 1024-thread groups */
void orc_revidx(uint32_t logit_idx, uint32_t vocab_size, uint32_t* offset, uint32_t* word) {
    const uint32_t tg = 1024u, words = 4u;
    const uint32_t thread_idx = logit_idx % tg;
    const uint32_t thread_offset = ((vocab_size + tg * words - 1u) / (tg * words)) * thread_idx;
    const uint32_t block_idx = logit_idx / tg;
    *offset = thread_offset + block_idx / words;
    *word = block_idx % words;
}

typedef struct {
    uint32_t index;
    float logit;
} orc_indexed_logit;
static int orc_sort_desc(const void* pa, const void* pb) { /* b.1.partial_cmp(&a.1).unwrap_or(Equal).then(a.0.cmp(&b.0)) */
    const orc_indexed_logit *a = (const orc_indexed_logit*)pa, *b = (const orc_indexed_logit*)pb;
    if (b->logit < a->logit) return -1;
    if (b->logit > a->logit) return 1;
    return a->index < b->index ? -1 : (a->index > b->index ? 1 : 0);
}
/* BU/cpu/kernel/sampling/unified_sampling.rs:13-99, every specialisation: grammar bitmask, temperature, top-k / top-p /
 * min-p (the three cuts act on ONE descending pass, "parallel", as the reference notes), Gumbel-max noise, arg-max with
 * ties -> lowest index.  seeds == NULL <=> !is_stochastic, bitmask == NULL <=> !has_bitmask; has_* select the filters. */
void orc_unified_sampling(const void* logits_in, uint32_t dt, uint32_t* output, const uint64_t* seeds, const uint32_t* bitmask,
                          uint32_t has_temperature, float temperature, uint32_t has_top_k, uint32_t top_k, uint32_t has_top_p,
                          float top_p, uint32_t has_min_p, float min_p, uint32_t vocab_size, uint32_t batch_size) {
    float* logits = (float*)malloc((size_t)vocab_size * sizeof(float));
    orc_indexed_logit* sorted = (orc_indexed_logit*)malloc((size_t)vocab_size * sizeof(orc_indexed_logit));
    const uint32_t mask_words = (vocab_size + 31u) / 32u;
    for (size_t b = 0; b < batch_size; ++b) {
        for (size_t i = 0; i < vocab_size; ++i) logits[i] = rd(logits_in, dt, b * vocab_size + i);
        if (bitmask) {
            const uint32_t* bm = bitmask + b * mask_words;
            for (size_t i = 0; i < vocab_size; ++i)
                if ((bm[i / 32] & (1u << (i % 32))) == 0) logits[i] = -INFINITY;
        }
        if (has_temperature) {
            const float recip_temperature = 1.0f / temperature;
            for (size_t i = 0; i < vocab_size; ++i) logits[i] *= recip_temperature;
        }
        if (has_top_k || has_top_p || has_min_p) {
            for (size_t i = 0; i < vocab_size; ++i) sorted[i].index = (uint32_t)i, sorted[i].logit = logits[i];
            qsort(sorted, vocab_size, sizeof(orc_indexed_logit), orc_sort_desc); /* total order (index tie-break): any sort gives the same result */
            const float logits_max = sorted[0].logit;
            float logits_norm = 0.0f;
            for (size_t i = 0; i < vocab_size; ++i) logits_norm += expf(sorted[i].logit - logits_max);
            for (size_t i = 0; i < vocab_size; ++i) logits[i] = -INFINITY;
            float top_p_mass = 0.0f;
            const float min_p_ln = has_min_p ? logf(min_p) : 0.0f;
            for (size_t rank = 0; rank < vocab_size; ++rank) {
                const float logit = sorted[rank].logit;
                if ((has_top_k && (uint32_t)rank >= top_k) || (has_top_p && top_p_mass >= top_p) || (has_min_p && logit < logits_max + min_p_ln)) break;
                logits[sorted[rank].index] = logit;
                top_p_mass += expf(logit - logits_max) / logits_norm;
            }
        }
        if (seeds) {
            const uint64_t seed = seeds[b];
            for (size_t i = 0; i < vocab_size; ++i) {
                uint32_t offset, word;
                orc_revidx((uint32_t)i, vocab_size, &offset, &word);
                logits[i] += orc_gumbel_float(seed, offset, word);
            }
        }
        size_t best = 0;
        for (size_t i = 1; i < vocab_size; ++i)
            if (logits[i] > logits[best]) best = i; /* max_by(partial_cmp .then(b.0.cmp(&a.0))): ties and NaN -> lowest index */
        output[b] = (uint32_t)best;
    }
    free(sorted);
    free(logits);
}

/* ------------------------------------------------------------------ Gated DeltaNet, decode
 * BU/cpu/kernel/gdn/conv_update.rs:17-55 */
void orc_delta_net_conv_update(const float* conv_weight, const float* bias, uint16_t* in_out, float* state,
                               uint32_t kernel_size, uint32_t conv_dim, uint32_t state_stride) {
    const size_t tap_count = kernel_size - 1;
    for (size_t channel = 0; channel < conv_dim; ++channel) {
        const size_t state_offset = channel * state_stride, weight_offset = channel * kernel_size;
        const float x = orc_bf16_to_f32(in_out[channel]);
        float acc = bias ? bias[channel] : 0.0f;
        for (size_t tap = 0; tap < tap_count; ++tap) acc += state[state_offset + tap] * conv_weight[weight_offset + tap];
        acc += x * conv_weight[weight_offset + tap_count];
        in_out[channel] = orc_f32_to_bf16(orc_activate(UZU_ACT_SILU, acc, ORC_F32));
        for (size_t tap = 1; tap < tap_count; ++tap) state[state_offset + tap - 1] = state[state_offset + tap];
        state[state_offset + tap_count - 1] = x;
    }
}
/* BU/cpu/kernel/gdn/update.rs:30-143 */
void orc_delta_net_update(const uint16_t* in_proj, const float* a_log, const float* dt_bias,
                          const float* norm_weight, float* state, uint16_t* out, uint32_t num_v_heads,
                          uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim,
                          uint32_t value_dim, float norm_epsilon) {
    const size_t conv_dim = 2 * (size_t)key_dim + value_dim;
#pragma omp parallel for schedule(static)
    for (size_t hv = 0; hv < num_v_heads; ++hv) {
        const size_t hk = hv / (num_v_heads / num_k_heads);
        const size_t q_offset = hk * head_k_dim, k_offset = key_dim + hk * head_k_dim;
        float* q = (float*)malloc(sizeof(float) * (2 * head_k_dim + head_v_dim));
        float* k = q + head_k_dim;
        float* o = k + head_k_dim;
        for (size_t j = 0; j < head_k_dim; ++j) {
            q[j] = orc_bf16_to_f32(in_proj[q_offset + j]);
            k[j] = orc_bf16_to_f32(in_proj[k_offset + j]);
        }
        float q_norm_sq = 0.0f, k_norm_sq = 0.0f; /* iter().map(|x| x*x).sum() = sequential from 0.0 */
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

        const float beta_raw = orc_bf16_to_f32(in_proj[conv_dim + value_dim + hv]);
        const float beta = 1.0f / (1.0f + expf(-beta_raw));
        const float a_raw = orc_bf16_to_f32(in_proj[conv_dim + value_dim + num_v_heads + hv]);
        const float sp_input = a_raw + dt_bias[hv];
        const float sp = sp_input > 20.0f ? sp_input : logf(1.0f + expf(sp_input));
        const float g = -expf(a_log[hv]) * sp;
        const float decay = expf(g);

        float kq_dot = 0.0f;
        for (size_t j = 0; j < head_k_dim; ++j) kq_dot += k[j] * q[j];

        for (size_t i = 0; i < head_v_dim; ++i) {
            const float v_i = orc_bf16_to_f32(in_proj[2 * key_dim + hv * head_v_dim + i]);
            float* srow = state + hv * head_v_dim * head_k_dim + i * head_k_dim;
            float sq_acc = 0.0f, sk_acc = 0.0f;
            for (size_t j = 0; j < head_k_dim; ++j) {
                const float s = srow[j];
                sq_acc += s * q[j];
                sk_acc += s * k[j];
            }
            const float retrieved_i = decay * sk_acc;
            const float delta_i = beta * (v_i - retrieved_i);
            o[i] = decay * sq_acc + delta_i * kq_dot;
            for (size_t j = 0; j < head_k_dim; ++j) srow[j] = decay * srow[j] + k[j] * delta_i;
        }
        float sumsq = 0.0f;
        for (size_t i = 0; i < head_v_dim; ++i) sumsq += o[i] * o[i];
        const float inv_rms = 1.0f / sqrtf(sumsq / (float)head_v_dim + norm_epsilon);
        for (size_t i = 0; i < head_v_dim; ++i) {
            const float z_i = orc_bf16_to_f32(in_proj[conv_dim + hv * head_v_dim + i]);
            const float z_silu = orc_activate(UZU_ACT_SILU, z_i, ORC_F32);
            const float final_val = o[i] * inv_rms * norm_weight[i] * z_silu;
            out[hv * head_v_dim + i] = orc_f32_to_bf16(final_val);
        }
        free(q);
    }
}

/* ------------------------------------------------------------------ Gated DeltaNet, prefill
 * BU/cpu/kernel/ssm/conv1d.rs:10-40 (Conv1dPack, StateT = f32, InputT = bf16) */
void orc_conv1d_pack(const float* state_in, const uint16_t* x, float* padded, uint32_t state_stride,
                     uint32_t row_stride, uint32_t suffix_len, uint32_t num_channels) {
    for (size_t c = 0; c < num_channels; ++c)
        for (size_t row = 0; row < (size_t)state_stride + suffix_len; ++row) {
            const size_t pi = row * row_stride + c;
            if (row < state_stride)
                padded[pi] = state_in[c * state_stride + row];
            else
                padded[pi] = orc_bf16_to_f32(x[(row - state_stride) * row_stride + c]);
        }
}
/* BU/cpu/kernel/gdn/conv_scan.rs:32-73 */
void orc_delta_net_conv_scan(const float* conv_padded, const float* conv_weight, const float* bias, uint16_t* in_proj,
                             float* state_out, uint32_t suffix_len, uint32_t kernel_size, uint32_t row_stride,
                             uint32_t state_stride, uint32_t conv_dim, uint32_t out_stride) {
    for (size_t token = 0; token < suffix_len; ++token)
        for (size_t channel = 0; channel < conv_dim; ++channel) {
            float acc = bias ? bias[channel] : 0.0f;
            for (size_t tap = 0; tap < kernel_size; ++tap)
                acc += conv_weight[channel * kernel_size + tap] * conv_padded[(token + tap) * row_stride + channel];
            in_proj[token * out_stride + channel] = orc_f32_to_bf16(orc_activate(UZU_ACT_SILU, acc, ORC_F32));
        }
    for (size_t channel = 0; channel < conv_dim; ++channel)
        for (size_t tap = 0; tap < state_stride; ++tap)
            state_out[channel * state_stride + tap] = conv_padded[((size_t)suffix_len + tap) * row_stride + channel];
}
/* BU/cpu/kernel/gdn/prefill_prep.rs:30-113 (QKT = f32, write_log_decay = false) */
void orc_delta_net_prefill_prep(const uint16_t* in_proj, const float* a_log, const float* dt_bias, float* q_norm_out,
                                float* k_norm_out, float* beta_out, float* decay_out, uint32_t num_v_heads,
                                uint32_t num_k_heads, uint32_t head_k_dim, uint32_t key_dim, uint32_t value_dim,
                                uint32_t suffix_len) {
    const size_t conv_dim = 2 * (size_t)key_dim + value_dim;
    const size_t total_proj_dim = conv_dim + value_dim + 2 * (size_t)num_v_heads;
    const size_t groups_per_head = num_v_heads / num_k_heads;
    for (size_t token = 0; token < suffix_len; ++token) {
        const size_t tok_offset = token * total_proj_dim;
        for (size_t hk = 0; hk < num_k_heads; ++hk) {
            const size_t q_off = tok_offset + hk * head_k_dim;
            float q_sq = 0.0f;
            for (size_t j = 0; j < head_k_dim; ++j) {
                const float v = orc_bf16_to_f32(in_proj[q_off + j]);
                q_sq += v * v;
            }
            const float q_inv = 1.0f / sqrtf(q_sq + 1e-6f);
            const float q_scale = 1.0f / sqrtf((float)head_k_dim);
            for (size_t j = 0; j < head_k_dim; ++j)
                q_norm_out[token * key_dim + hk * head_k_dim + j] = orc_bf16_to_f32(in_proj[q_off + j]) * q_inv * q_scale;
            const size_t k_off = tok_offset + key_dim + hk * head_k_dim;
            float k_sq = 0.0f;
            for (size_t j = 0; j < head_k_dim; ++j) {
                const float v = orc_bf16_to_f32(in_proj[k_off + j]);
                k_sq += v * v;
            }
            const float k_inv = 1.0f / sqrtf(k_sq + 1e-6f);
            for (size_t j = 0; j < head_k_dim; ++j)
                k_norm_out[token * key_dim + hk * head_k_dim + j] = orc_bf16_to_f32(in_proj[k_off + j]) * k_inv;
            for (size_t group = 0; group < groups_per_head; ++group) {
                const size_t hv = hk * groups_per_head + group;
                const float beta_raw = orc_bf16_to_f32(in_proj[tok_offset + conv_dim + value_dim + hv]);
                const float beta = 1.0f / (1.0f + expf(-beta_raw));
                const float a_raw = orc_bf16_to_f32(in_proj[tok_offset + conv_dim + value_dim + num_v_heads + hv]);
                const float sp_in = a_raw + dt_bias[hv];
                const float sp = sp_in > 20.0f ? sp_in : logf(1.0f + expf(sp_in));
                const float log_decay = -expf(a_log[hv]) * sp;
                beta_out[token * num_v_heads + hv] = beta;
                decay_out[token * num_v_heads + hv] = expf(log_decay);
            }
        }
    }
}
/* BU/cpu/kernel/gdn/prefill.rs:39-80 (state f32 per the Metal kernel / block allocation) */
void orc_delta_net_prefill(const float* q_norm, const float* k_norm, const float* beta_buf, const float* decay_buf,
                           const uint16_t* in_proj, float* state, uint16_t* out, uint32_t num_v_heads,
                           uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim,
                           uint32_t value_dim, uint32_t suffix_len) {
    const size_t conv_dim = 2 * (size_t)key_dim + value_dim;
    const size_t total_proj_dim = conv_dim + value_dim + 2 * (size_t)num_v_heads;
    const size_t groups_per_head = num_v_heads / num_k_heads;
#pragma omp parallel for schedule(static)
    for (size_t hv = 0; hv < num_v_heads; ++hv) {
        const size_t hk = hv / groups_per_head;
        for (size_t token = 0; token < suffix_len; ++token) {
            const size_t qk_off = token * key_dim + hk * head_k_dim;
            const float decay = decay_buf[token * num_v_heads + hv];
            const float beta = beta_buf[token * num_v_heads + hv];
            for (size_t i = 0; i < head_v_dim; ++i) {
                float* srow = state + (hv * head_v_dim + i) * head_k_dim;
                float kv_mem = 0.0f;
                for (size_t j = 0; j < head_k_dim; ++j) kv_mem += (decay * srow[j]) * k_norm[qk_off + j];
                const float v_val = orc_bf16_to_f32(in_proj[token * total_proj_dim + 2 * key_dim + hv * head_v_dim + i]);
                const float delta = beta * (v_val - kv_mem);
                float o_val = 0.0f;
                for (size_t j = 0; j < head_k_dim; ++j) {
                    const float new_s = decay * srow[j] + k_norm[qk_off + j] * delta;
                    srow[j] = new_s;
                    o_val += new_s * q_norm[qk_off + j];
                }
                out[token * value_dim + hv * head_v_dim + i] = orc_f32_to_bf16(o_val);
            }
        }
    }
}
/* BU/cpu/kernel/gdn/norm_gate.rs:32-66 (norm_weight f32 per the block's allocation) */
void orc_delta_net_norm_gate(uint16_t* in_out, const uint16_t* in_proj, const float* norm_weight,
                             uint32_t num_v_heads, uint32_t head_v_dim, uint32_t value_dim, uint32_t conv_dim,
                             uint32_t total_proj_dim, float norm_epsilon, uint32_t suffix_len) {
    for (size_t token = 0; token < suffix_len; ++token)
        for (size_t hv = 0; hv < num_v_heads; ++hv) {
            const size_t base = token * value_dim + hv * head_v_dim;
            float sumsq = 0.0f;
            for (size_t i = 0; i < head_v_dim; ++i) {
                const float val = orc_bf16_to_f32(in_out[base + i]);
                sumsq += val * val;
            }
            const float inv_rms = 1.0f / sqrtf(sumsq / (float)head_v_dim + norm_epsilon);
            for (size_t i = 0; i < head_v_dim; ++i) {
                const float o_i = orc_bf16_to_f32(in_out[base + i]);
                const float z_i = orc_bf16_to_f32(in_proj[token * total_proj_dim + conv_dim + hv * head_v_dim + i]);
                const float final_val = o_i * inv_rms * norm_weight[i] * orc_activate(UZU_ACT_SILU, z_i, ORC_F32);
                in_out[base + i] = orc_f32_to_bf16(final_val);
            }
        }
}
