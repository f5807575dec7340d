// engine_build.hip -- model driver above the kernel boundary (include/uzu_hip_engine.h): construction of models and sequence states.
//
// Restates, for one sequence, the op order of the reference's backend-generic graph code:
//   Decoder::encode            BU/../encodable_block/decoder.rs:138-203
//   Transformer::encode        BU/../encodable_block/transformer.rs:226-329
//   TransformerLayer::encode   BU/../encodable_block/transformer_layer.rs:194-238
//   Attention::attend          BU/../encodable_block/mixer/attention/mode.rs:45-144
//   AttentionCores::encode     BU/../encodable_block/mixer/attention/core/mod.rs:81-93
//   DeltaNet::encode           BU/../encodable_block/mixer/delta_net.rs:473-645
//   DenseMlp::encode           BU/../encodable_block/mlp/dense.rs:32-48
//   Embedding::encode_readout  BU/../encodable_block/embedding.rs:374-456
//   LanguageModelStream        BU/../engine/language_model/stream/stream.rs:190-345 (prefill), 593-751 (decode)
// MI355X execution strategy: weights, KV cache and DeltaNet state resident in HBM; the context length,
// the next input token and the sampled-token history live in device memory, so ONE captured hipGraph is
// replayed for every decode step and steps are chained without a host round trip.
#include "engine_types.h"

using namespace uzu;
using namespace uzu::eng;

namespace uzu {
namespace eng {

uzu_status dev_alloc(uzu_hip_model* m, size_t bytes, void** out, bool zero) {
    void* p = nullptr;
    const size_t alloc = bytes ? (bytes + 255) & ~(size_t)255 : 256;
    hipError_t e = hipMalloc(&p, alloc);
    if (e != hipSuccess) {
        set_error("engine: hipMalloc(%zu) failed: %s", alloc, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? UZU_ERR_OUT_OF_MEMORY : UZU_ERR_HIP;
    }
    m->allocations.push_back(p);
    m->allocation_bytes.push_back(alloc);
    m->ctx->current_bytes += alloc;
    if (m->ctx->current_bytes > m->ctx->peak_bytes) m->ctx->peak_bytes = m->ctx->current_bytes;
    // on the engine's OWN stream: hipMemset runs on the null stream and returns before the fill has happened, and the engine's stream is
    // hipStreamNonBlocking -- a pass started right after model creation raced the zeroing of its scratch (first model of a process, wide
    // layers, a short prompt: the tail of the fills -- `logits` -- landed after the pass had written them: token 0)
    if (zero) HIPCHK(hipMemsetAsync(p, 0, alloc, m->ctx->stream));
    else if (poison_level()) UZU_PROPAGATE(poison_fill(p, alloc, m->ctx->stream, true)); // CI mode (internal.h): what nothing wrote reads as NaN
    *out = p;
    return UZU_OK;
}

void dev_free(uzu_hip_model* m, void* p) {
    for (size_t i = 0; i < m->allocations.size(); ++i)
        if (m->allocations[i] == p) {
            (void)hipFree(p);
            m->ctx->current_bytes -= m->allocation_bytes[i] < m->ctx->current_bytes ? m->allocation_bytes[i] : m->ctx->current_bytes;
            m->allocations.erase(m->allocations.begin() + i);
            m->allocation_bytes.erase(m->allocation_bytes.begin() + i);
            return;
        }
}

uzu_status state_alloc(uzu_hip_state* st, size_t bytes, void** out, bool zero_by_contract) {
    void* p = nullptr;
    const size_t alloc = bytes ? (bytes + 255) & ~(size_t)255 : 256;
    hipError_t e = hipMalloc(&p, alloc);
    if (e != hipSuccess) {
        set_error("state: hipMalloc(%zu) failed: %s", alloc, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? UZU_ERR_OUT_OF_MEMORY : UZU_ERR_HIP;
    }
    st->allocations.push_back(p);
    st->allocation_bytes.push_back(alloc);
    st->bytes += alloc;
    uzu_hip_context* ctx = st->m->ctx;
    ctx->current_bytes += alloc;
    if (ctx->current_bytes > ctx->peak_bytes) ctx->peak_bytes = ctx->current_bytes;
    // stream-ordered with every pass that will use the state (see dev_alloc).  KV rows are undefined until AttentionPrepare writes them (the reference allocates
    // the caches without a fill): UZU_HIP_POISON=2 hands them out full of NaNs; conv / SSM states and the control block start at zero by contract
    if (!zero_by_contract && poison_level() >= 2) UZU_PROPAGATE(poison_fill(p, alloc, ctx->stream, false));
    else HIPCHK(hipMemsetAsync(p, 0, alloc, ctx->stream));
    *out = p;
    return UZU_OK;
}

// device side of a state (graphs, caches); the host struct stays
void state_release(uzu_hip_state* st) {
    if (!st || !st->m) return;
    if (st->graph_single) (void)hipGraphExecDestroy(st->graph_single);
    if (st->graph_two) (void)hipGraphExecDestroy(st->graph_two);
    st->graph_single = st->graph_two = nullptr;
    for (void* p : st->allocations) (void)hipFree(p);
    st->allocations.clear();
    st->allocation_bytes.clear();
    uzu_hip_context* ctx = st->m->ctx;
    ctx->current_bytes -= st->bytes < ctx->current_bytes ? st->bytes : ctx->current_bytes;
    st->bytes = 0;
}
void state_free(uzu_hip_state* st) {
    if (!st) return;
    state_release(st);
    delete st;
}

// KV caches for max_context_length + 1024 rows (mixer/attention/state.rs:14), DeltaNet conv / SSM states, control block
uzu_status state_build(uzu_hip_model* m, uzu_hip_state** out) {
    auto* st = new uzu_hip_state();
    st->m = m;
    st->layers.resize(m->layers.size());
    uzu_status r = UZU_OK;
    auto need = [&](size_t bytes, void** p) {
        if (r == UZU_OK) r = state_alloc(st, bytes, p);
    };
    for (size_t l = 0; l < m->layers.size(); ++l) {
        const uzu_layer_desc& h = m->layers[l].d;
        void* p = nullptr;
        if (h.mixer_kind == UZU_MIXER_ATTENTION && h.is_kv_sharing) {
            // TransformerLayerStateType::Shared(kv_source_layer_index) (transformer.rs:205-216): no state of its own
        } else if (h.mixer_kind == UZU_MIXER_ATTENTION) {
            // AttentionState::create_empty (state.rs:69-136): a causal sliding-window layer keeps a RING of `window` rows + the suffix region
            const size_t kv_rows = h.sliding_window_size ? (size_t)h.sliding_window_size + m->chunk : (size_t)m->max_positions;
            const size_t kv_bytes = kv_rows * h.num_groups * h.head_dim * 2;
            // a Full cache's rows are undefined until AttentionPrepare writes them; a RING's unfilled slots are read by the attention kernels and masked by weight
            // (ring_length), so they must hold finite values: zero by contract
            const bool zero_by_contract = h.sliding_window_size != 0;
            if (r == UZU_OK) r = state_alloc(st, kv_bytes, &p, zero_by_contract), st->layers[l].keys = (uint16_t*)p;
            if (r == UZU_OK) r = state_alloc(st, kv_bytes, &p, zero_by_contract), st->layers[l].values = (uint16_t*)p;
        } else {
            need(m->layers[l].conv_state_bytes, &p), st->layers[l].conv_state = (float*)p;
            need(m->layers[l].ssm_state_bytes, &p), st->layers[l].ssm_state = (float*)p;
        }
    }
    void* p = nullptr;
    need(4, &p), st->d_ctx_len = (uint32_t*)p;
    need((size_t)m->chunk * 4, &p), st->d_tokens = (uint32_t*)p;
    need(4, &p), st->d_out_token = (uint32_t*)p;
    need((size_t)m->max_positions * 4, &p), st->d_sampled = (uint32_t*)p;
    if (r != UZU_OK) {
        state_free(st);
        return r;
    }
    *out = st;
    return UZU_OK;
}

// Make `st` the state the encoders work on: its pointers go into the DLayer / model fields, the host-side mirrors
// (context length, graphs) of the previously bound state are written back first.
void bind_state(uzu_hip_model* m, uzu_hip_state* st) {
    if (m->bound == st) return;
    if (m->bound) {
        m->bound->context_length = m->context_length;
        m->bound->graph_single = m->graph_single, m->bound->graph_two = m->graph_two, m->bound->graph_epoch = m->graph_epoch;
    }
    for (size_t l = 0; l < m->layers.size(); ++l) {
        // a KV-sharing layer reads the rows of the layer that owns them (MaybeMut::Const(owned layer state), transformer.rs:264-275)
        const size_t src = m->layers[l].d.mixer_kind == UZU_MIXER_ATTENTION && m->layers[l].d.is_kv_sharing ? m->layers[l].d.kv_source_layer_index : l;
        m->layers[l].keys = st->layers[src].keys, m->layers[l].values = st->layers[src].values;
        m->layers[l].conv_state = st->layers[l].conv_state, m->layers[l].ssm_state = st->layers[l].ssm_state;
    }
    m->d_ctx_len = st->d_ctx_len, m->d_tokens = st->d_tokens, m->d_out_token = st->d_out_token, m->d_sampled = st->d_sampled;
    m->context_length = st->context_length;
    m->graph_single = st->graph_single, m->graph_two = st->graph_two, m->graph_epoch = st->graph_epoch;
    m->hidden_ready = false; // row 0 of the scratch `hidden` belongs to whoever ran last
    m->bound = st;
}

uzu_status upload_bytes(uzu_hip_model* m, const void* host, size_t bytes, void** out) {
    if (!host) {
        *out = nullptr;
        return UZU_OK;
    }
    void* p;
    UZU_PROPAGATE(dev_alloc(m, bytes, &p));
    HIPCHK(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    m->weight_bytes += bytes;
    *out = p;
    return UZU_OK;
}

uzu_status upload_linear(uzu_hip_model* m, const uzu_linear_desc& h, DLinear* o, bool is_embedding) {
    o->n = h.n, o->k = h.k, o->bits = h.bits, o->group = h.group_size, o->method = h.method;
    if (!h.weights) return UZU_OK;
    if (h.method == UZU_QUANT_NONE) {
        UZU_PROPAGATE(upload(m, h.weights, (size_t)h.n * h.k * 2, &o->w));
    } else {
        UZU_REQUIRE(h.bits == 4 || h.bits == 8, "engine: linear with %u-bit codes", h.bits);
        UZU_REQUIRE(h.group_size > 0, "engine: quantized linear with group_size 0");
        const size_t groups = (h.k + h.group_size - 1) / h.group_size;
        UZU_PROPAGATE(upload(m, h.weights, (size_t)h.n * h.k * h.bits / 8, &o->w));
        UZU_PROPAGATE(upload(m, h.scales, (size_t)h.n * groups * 2, &o->scales));
        if (h.method == UZU_QUANT_SCALE_BIAS) UZU_PROPAGATE(upload(m, h.biases, (size_t)h.n * groups * 2, &o->biases));
        if (h.method == UZU_QUANT_SCALE_ZERO_POINT)
            UZU_PROPAGATE(upload(m, h.zero_points, (size_t)h.n * (h.bits == 4 ? (groups + 1) / 2 : groups), &o->zp));
    }
    UZU_PROPAGATE(upload(m, h.out_biases, (size_t)h.n * 2, &o->out_biases));
    if (h.method != UZU_QUANT_NONE && !is_embedding) { // prefill GEMM: coefficient half of its pre-pass, once (k_gemm128.hip)
        k::MatmulParams cp{};
        cp.b = o->w, cp.scales = o->scales, cp.biases = o->biases, cp.zero_points = o->zp;
        cp.b_kind = h.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS : h.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
        cp.bits = h.bits, cp.group_size = h.group_size, cp.n = h.n, cp.k = h.k;
        if (k::gemm_coef_table_supported(cp)) {
            void* cptr;
            UZU_PROPAGATE(dev_alloc(m, (size_t)h.n * (h.k / h.group_size) * sizeof(float), &cptr));
            UZU_PROPAGATE(k::gemm_coef_table(m->ctx->stream, cp, (float*)cptr));
            o->coef = (float*)cptr;
        }
    }
    if (h.input_signs || h.output_signs) {
        UZU_REQUIRE(is_embedding || (h.input_signs && h.output_signs), "engine: an RHT linear needs both input_signs and output_signs (HybridSpec InputOutput)");
        UZU_REQUIRE(h.k % 32 == 0 && (is_embedding || h.n % 32 == 0), "engine: RHT linear %u x %u is not a whole number of 32-wide Hadamard blocks", h.n, h.k);
        // embedding tables: both vectors run over model_dim = k (the table's output side is the lookup's row, embedding.rs:161-188)
        if (h.input_signs) UZU_PROPAGATE(upload(m, h.input_signs, (size_t)h.k * 4, &o->in_signs));
        if (h.output_signs) UZU_PROPAGATE(upload(m, h.output_signs, (size_t)(is_embedding ? h.k : h.n) * 4, &o->out_signs));
        m->rht_max_k = m->rht_max_k > h.k ? m->rht_max_k : h.k;
        auto pack = [&](const int32_t* f, size_t count, uint32_t** out) -> uzu_status {
            std::vector<uint32_t> words(count / 32, 0u);
            for (size_t i = 0; i < count; ++i) {
                if (f[i] != 1 && f[i] != -1) return UZU_OK; // not a sign vector: no packed form (the fused step is then not taken)
                if (f[i] < 0) words[i / 32] |= 1u << (i % 32);
            }
            return upload(m, words.data(), words.size() * 4, out);
        };
        if (h.input_signs) {
            UZU_PROPAGATE(pack((const int32_t*)h.input_signs, h.k, &o->in_bits));
            if (o->in_bits) {
                o->in_words.assign(h.k / 32, 0u);
                for (size_t i = 0; i < h.k; ++i)
                    if (((const int32_t*)h.input_signs)[i] < 0) o->in_words[i / 32] |= 1u << (i % 32);
            }
        }
        if (h.output_signs) UZU_PROPAGATE(pack((const int32_t*)h.output_signs, is_embedding ? h.k : h.n, &o->out_bits));
    }
    if (h.lora_rank) { // QLoRALinearWrapper::new (qlora_wrapper.rs:61-175)
        UZU_REQUIRE(!is_embedding && h.method != UZU_QUANT_NONE, "engine: a QLoRA adapter needs a quantized base linear");
        UZU_REQUIRE(!h.out_biases, "engine: QLoRA linear with biases is not supported (the reference asserts the same)");
        UZU_REQUIRE(h.adapter_down && h.adapter_up, "engine: QLoRA linear of rank %u without adapter tensors", h.lora_rank);
        o->lora_rank = h.lora_rank;
        UZU_PROPAGATE(upload(m, h.adapter_down, (size_t)h.lora_rank * h.k * 2, &o->adapter_down));
        UZU_PROPAGATE(upload(m, h.adapter_up, (size_t)h.n * h.lora_rank * 2, &o->adapter_up));
        m->lora_max_rank = m->lora_max_rank > h.lora_rank ? m->lora_max_rank : h.lora_rank;
        m->rht_max_k = m->rht_max_k > h.k ? m->rht_max_k : h.k; // the base input is a transformed COPY (the adapter reads the original)
    }
    return UZU_OK;
}

uzu_status upload_norm(uzu_hip_model* m, const uzu_norm_desc& h, uint32_t dim, DNorm* o) {
    o->present = h.present != 0;
    o->full_layer = h.full_layer, o->subtract_mean = h.subtract_mean, o->eps = h.epsilon, o->offset = h.scale_offset;
    if (!o->present) return UZU_OK;
    UZU_PROPAGATE(upload(m, h.scales, (size_t)dim * 4, &o->scales));
    UZU_PROPAGATE(upload(m, h.biases, (size_t)dim * 4, &o->biases));
    return UZU_OK;
}

// host RoPE table: encodable_block/mixer/attention/rope.rs:13-114 (Unscaled / Linear / Llama-3 / YaRN / LongRoPE), computed
// with the platform libm exactly as the reference does per pass, but once for all positions.
void rope_tables(const uzu_rope_desc& r, uint32_t n_pos, std::vector<float>& cosines, std::vector<float>& sines) {
    const uint32_t head_dim = r.head_dim, half_dim = head_dim / 2;
    cosines.assign((size_t)n_pos * head_dim, 0.f);
    sines.assign((size_t)n_pos * head_dim, 0.f);
    float attention_scaling_factor = 1.0f; /* rope.rs:21-27 */
    if (r.kind == UZU_ROPE_YARN) attention_scaling_factor = 0.1f * logf(r.scaling_factor) + 1.0f;
    else if (r.kind == UZU_ROPE_LONGROPE && r.scaling_factor > 1.0f)
        attention_scaling_factor = sqrtf(1.0f + logf(r.scaling_factor) / logf((float)r.original_context_length));
    for (uint32_t pair_index = 0; pair_index < half_dim; ++pair_index) {
        const uint32_t channel_index = pair_index * 2;
        float inverse_frequency = 1.0f / powf(r.base, (float)channel_index / (float)head_dim);
        if (r.kind == UZU_ROPE_LINEAR) {
            inverse_frequency = inverse_frequency / r.scaling_factor;
        } else if (r.kind == UZU_ROPE_LLAMA) {
            const float low_frequency_wavelength = (float)r.original_context_length / r.low_frequency_factor;
            const float high_frequency_wavelength = (float)r.original_context_length / r.high_frequency_factor;
            const float wavelength = 2.0f * 3.14159265358979323846f / inverse_frequency;
            const float scaled_frequency = inverse_frequency / r.scaling_factor;
            if (wavelength < high_frequency_wavelength) {
            } else if (wavelength > low_frequency_wavelength) {
                inverse_frequency = scaled_frequency;
            } else {
                float smoothing_factor = (float)r.original_context_length / wavelength - r.low_frequency_factor;
                smoothing_factor = smoothing_factor / (r.high_frequency_factor - r.low_frequency_factor);
                inverse_frequency = smoothing_factor * inverse_frequency + (1.0f - smoothing_factor) * scaled_frequency;
            }
        } else if (r.kind == UZU_ROPE_YARN) { /* rope.rs:60-81 (double for the ramp bounds, as the reference) */
            const double dim = (double)r.head_dim, base = (double)r.base, original_context_length = (double)r.original_context_length;
            double low = dim * log(original_context_length / ((double)r.beta_fast * 2.0 * 3.14159265358979323846)) / (2.0 * log(base));
            double high = dim * log(original_context_length / ((double)r.beta_slow * 2.0 * 3.14159265358979323846)) / (2.0 * log(base));
            if (r.truncate) low = floor(low), high = ceil(high);
            const float low_f = (float)(low > 0.0 ? low : 0.0);
            float high_f = (float)(high < (double)(r.head_dim - 1) ? high : (double)(r.head_dim - 1));
            if (low_f == high_f) high_f += 0.001f;
            float ramp = ((float)pair_index - low_f) / (high_f - low_f);
            ramp = ramp < 0.0f ? 0.0f : (ramp > 1.0f ? 1.0f : ramp);
            const float smoothing_factor = 1.0f - ramp;
            const float scaled_frequency = inverse_frequency / r.scaling_factor;
            inverse_frequency = scaled_frequency * (1.0f - smoothing_factor) + inverse_frequency * smoothing_factor;
        } else if (r.kind == UZU_ROPE_LONGROPE) { /* rope.rs:82-89 */
            const float* factors = r.max_sequence_length > r.original_context_length ? r.long_factor : r.short_factor;
            inverse_frequency = inverse_frequency / factors[pair_index];
        }
        for (uint32_t pos = 0; pos < n_pos; ++pos) {
            const float embedding = (float)pos * inverse_frequency;
            const float sine = sinf(embedding) * attention_scaling_factor, cosine = cosf(embedding) * attention_scaling_factor;
            const size_t o = (size_t)pos * head_dim + pair_index;
            sines[o] = sine, sines[o + half_dim] = sine, cosines[o] = cosine, cosines[o + half_dim] = cosine;
        }
    }
}

} // namespace eng
} // namespace uzu

extern "C" {

// sizes of the description structs this library was built with (include/uzu_model_desc.h: they are part of the ABI -- uzu_layer_desc is an array element)
void uzu_hip_desc_abi(uint32_t out[6]) {
    if (!out) return;
    out[0] = (uint32_t)sizeof(uzu_linear_desc), out[1] = (uint32_t)sizeof(uzu_norm_desc), out[2] = (uint32_t)sizeof(uzu_rope_desc), out[3] = (uint32_t)sizeof(uzu_layer_desc);
    out[4] = (uint32_t)sizeof(uzu_model_desc), out[5] = (uint32_t)sizeof(uzu_dflash_desc);
}

uzu_status uzu_hip_model_create(uzu_hip_context* ctx, const uzu_model_desc* desc, uint32_t flags, uzu_hip_model** out) {
    return uzu_hip_model_create_tp(ctx, desc, flags, nullptr, 0, out);
}

uzu_status uzu_hip_model_create_tp(uzu_hip_context* ctx, const uzu_model_desc* desc, uint32_t flags, uzu_hip_tp_comm* comm, uint32_t vocab_offset,
                                   uzu_hip_model** out) {
    UZU_REQUIRE(ctx && desc && out, "model_create: null argument");
    UZU_REQUIRE(desc->num_layers > 0 && desc->layers, "model_create: no layers");
    (void)hipSetDevice(ctx->device);
    auto* m = new uzu_hip_model();
    m->ctx = ctx;
    m->flags = flags;
    m->d = *desc;
    m->d.layers = nullptr;
    m->tp = (uzu::tp::Comm*)comm;
    m->vocab_offset = vocab_offset;
    if (comm && desc->tied_embeddings) {
        delete m;
        set_error("model_create: a tensor-parallel shard describes its read-out rows as an untied output_embedding");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    const uint32_t d = desc->model_dim;
    uzu_status st = UZU_OK;
    auto fail = [&](uzu_status s) {
        uzu_hip_model_destroy(m);
        return s;
    };
    if (comm) { // the options a shard cannot carry, from the description alone: refused before anything is uploaded
        bool options = desc->embedding_norm.present || desc->has_ple;
        for (uint32_t l = 0; l < desc->num_layers; ++l) {
            const uzu_layer_desc& h = desc->layers[l];
            options = options || h.has_post_layer_scalar || h.has_ple || (h.mixer_kind == UZU_MIXER_ATTENTION && (h.is_kv_sharing || h.normalize_values));
        }
        if (options) {
            set_error("model_create: post-layer scalars, embedding norm, KV sharing, value normalisation and per-layer embeddings are not sharded (single GPU only)");
            return fail(UZU_ERR_UNSUPPORTED);
        }
    }
#define TRY(x) do { st = (x); if (st != UZU_OK) return fail(st); } while (0)
    TRY(upload_linear(m, desc->embedding, &m->embedding, true));
    if (!desc->tied_embeddings) TRY(upload_linear(m, desc->output_embedding, &m->output_embedding, true));
    TRY(upload_norm(m, desc->output_norm, d, &m->output_norm));
    TRY(upload_norm(m, desc->embedding_norm, d, &m->embedding_norm));
    m->gemma_options = desc->embedding_norm.present || desc->has_ple;
    if (desc->has_ple) { // PerLayerEmbedding::new (per_layer_embedding.rs:47-106)
        const uint32_t total = desc->num_layers * desc->ple_dim;
        if (!desc->ple_dim || desc->ple_token_embedding.k != total || desc->ple_token_embedding.n != desc->ple_vocab_size || desc->ple_model_projection.n != total ||
            desc->ple_model_projection.k != d || !desc->ple_projection_norm.present || desc->ple_model_projection_scale == 0.0f) {
            set_error("model_create: per-layer embedding shapes inconsistent (token table [%u, %u], projection [%u, %u], %u layers x ple_dim %u)", desc->ple_token_embedding.n,
                      desc->ple_token_embedding.k, desc->ple_model_projection.n, desc->ple_model_projection.k, desc->num_layers, desc->ple_dim);
            return fail(UZU_ERR_INVALID_ARGUMENT);
        }
        TRY(upload_linear(m, desc->ple_token_embedding, &m->ple_token_embedding, true));
        TRY(upload_linear(m, desc->ple_model_projection, &m->ple_model_projection));
        TRY(upload_norm(m, desc->ple_projection_norm, desc->ple_dim, &m->ple_projection_norm));
        // per_layer_embedding.rs:75-90: epsilon / model_projection_scale^2, PostLayerScalar::ScaleOutput(input_scale)
        m->ple_projection_norm.eps = desc->ple_projection_norm.epsilon / (desc->ple_model_projection_scale * desc->ple_model_projection_scale);
        m->ple_projection_norm.scalar_mode = 2, m->ple_projection_norm.scalar = desc->ple_input_scale;
    }
    if (desc->num_ropes && !desc->ropes) return fail((set_error("model_create: num_ropes without a ropes table"), UZU_ERR_INVALID_ARGUMENT));
    m->chunk = prefill_chunk_rows();
    m->max_positions = desc->max_context_length + m->chunk;
    m->layers.resize(desc->num_layers);
    uint32_t max_qkv = 0, max_qdim = 0, max_hidden = 0, max_proj = 0, max_value = 0, max_key = 0, max_hv = 0, max_hd = 0, max_heads = 0;
    uint32_t max_moe_k = 0, max_moe_ff = 0, max_moe_e = 0;
    for (uint32_t l = 0; l < desc->num_layers; ++l) {
        const uzu_layer_desc& h = desc->layers[l];
        DLayer& L = m->layers[l];
        L.d = h;
        TRY(upload_norm(m, h.pre_mixer_norm, d, &L.pre_mixer));
        TRY(upload_norm(m, h.post_mixer_norm, d, &L.post_mixer));
        TRY(upload_norm(m, h.pre_mlp_norm, d, &L.pre_mlp));
        TRY(upload_norm(m, h.post_mlp_norm, d, &L.post_mlp));
        if (!L.pre_mlp.present) {
            set_error("model_create: layer %u has no pre_mlp_norm", l);
            return fail(UZU_ERR_INVALID_ARGUMENT);
        }
        if (!L.pre_mixer.present && l != 0) { // TransformerLayerError::MissingPreMixerNormConfig (transformer_layer.rs:110-114)
            set_error("model_create: layer %u has no pre_mixer_norm (only the first layer may omit it)", l);
            return fail(UZU_ERR_INVALID_ARGUMENT);
        }
        L.last_reader = l;
        if (h.has_post_layer_scalar) {
            if (!L.post_mlp.present) { // TransformerLayerError::PostLayerScalarWithoutPostMlpNorm (transformer_layer.rs:61-66)
                set_error("model_create: layer %u has a post-layer scalar but no post_mlp_norm", l);
                return fail(UZU_ERR_INVALID_ARGUMENT);
            }
            if (!h.has_ple) // with a PLE projection the projection owns the scalar (transformer_layer.rs:78-84)
                L.pre_mlp.scalar_mode = 1, L.pre_mlp.scalar = h.post_layer_scalar, L.post_mlp.scalar_mode = 2, L.post_mlp.scalar = h.post_layer_scalar;
            m->gemma_options = true;
        }
        if (h.has_ple) { // PerLayerEmbeddingProjection::new (per_layer_embedding.rs:166-215)
            if (!desc->has_ple || h.ple_dim != desc->ple_dim || h.ple_gate.n != h.ple_dim || h.ple_gate.k != d || h.ple_projection.n != d || h.ple_projection.k != h.ple_dim ||
                !h.ple_norm.present) {
                set_error("model_create: layer %u per-layer embedding projection inconsistent with the model's per-layer embedding", l);
                return fail(UZU_ERR_INVALID_ARGUMENT);
            }
            TRY(upload_linear(m, h.ple_gate, &L.ple_gate));
            TRY(upload_linear(m, h.ple_projection, &L.ple_projection));
            TRY(upload_norm(m, h.ple_norm, d, &L.ple_norm));
        }
        if (h.mlp_kind == UZU_MLP_MOE) { // MoeBlock::new (mlp/moe/mod.rs:84-200): the refusals are the reference's own
            const uzu_moe_desc& M = h.moe;
            if (d % 8 || M.num_routed_experts == 0 || M.num_routed_experts > 512 || M.num_active_experts == 0 || M.num_active_experts > 128 || M.num_active_experts > M.num_routed_experts ||
                M.expert_hidden_dim == 0 || M.expert_hidden_dim % 8 || (M.gating_sel != 2 && M.gating_sel != 3) || !M.router_weights || !M.router_biases || !M.w13 || !M.w2 || !M.up_biases ||
                !M.down_biases) {
                set_error("model_create: layer %u MoE description refused (model_dim %% 8, 1..512 routed / 1..128 active experts, SiLU / GELUApprox experts, router + up + down biases required)", l);
                return fail(UZU_ERR_UNSUPPORTED);
            }
            if (m->tp) return fail((set_error("model_create: MoE layers are not sharded (single GPU only)"), UZU_ERR_UNSUPPORTED));
            const size_t E = M.num_routed_experts, F = M.expert_hidden_dim;
            TRY(upload(m, M.router_weights, E * d * 2, &L.moe.router_weights));
            TRY(upload(m, M.router_biases, E * 2, &L.moe.router_biases));
            TRY(upload(m, M.w13, E * 2 * F * d * 2, &L.moe.w13));
            TRY(upload(m, M.w2, E * d * F * 2, &L.moe.w2));
            TRY(upload(m, M.up_biases, E * 2 * F * 2, &L.moe.up_biases));
            TRY(upload(m, M.down_biases, E * d * 2, &L.moe.down_biases));
            max_moe_k = max_moe_k > M.num_active_experts ? max_moe_k : M.num_active_experts;
            max_moe_ff = max_moe_ff > M.expert_hidden_dim ? max_moe_ff : M.expert_hidden_dim;
            max_moe_e = max_moe_e > M.num_routed_experts ? max_moe_e : M.num_routed_experts;
        } else {
            TRY(upload_linear(m, h.up_projection, &L.up));
            TRY(upload_linear(m, h.down_projection, &L.down));
            if (h.up_projection.n != 2 * h.hidden_dim || h.down_projection.k != h.hidden_dim) {
                set_error("model_create: layer %u MLP shapes inconsistent", l);
                return fail(UZU_ERR_INVALID_ARGUMENT);
            }
            max_hidden = max_hidden > h.hidden_dim ? max_hidden : h.hidden_dim;
        }
        if (h.mixer_kind == UZU_MIXER_ATTENTION) {
            if (h.is_kv_sharing) { // TransformerLayerStateType::Shared (transformer.rs:205-216, 264-275)
                const uint32_t src = h.kv_source_layer_index;
                if (src >= l || desc->layers[src].mixer_kind != UZU_MIXER_ATTENTION || desc->layers[src].is_kv_sharing) {
                    set_error("model_create: layer %u shares the KV state of layer %u, which is not an earlier attention layer that owns its state", l, src);
                    return fail(UZU_ERR_INVALID_ARGUMENT);
                }
                // the core's ring / window specialisation comes from the layer's own config, the ring parameters from the state it reads
                // (mixer/attention/mod.rs:166-198, core/single_pass.rs:60-70): only equal geometry is a meaningful configuration
                const uzu_layer_desc& S = desc->layers[src];
                if (S.sliding_window_size != h.sliding_window_size || S.num_groups != h.num_groups || S.head_dim != h.head_dim) {
                    // (a narrowing of this library, not a construction error of the reference: UNSUPPORTED)
                    set_error("model_create: layer %u and its KV source %u differ in window / kv heads / head_dim", l, src);
                    return fail(UZU_ERR_UNSUPPORTED);
                }
                m->layers[src].last_reader = l;
                m->gemma_options = true;
            }
            if (h.normalize_values) m->gemma_options = true;
            if (h.is_kv_sharing && h.qkv_projection.n != h.num_heads * h.head_dim) { // queries only (mixer/attention/mod.rs:89-95)
                set_error("model_create: KV-sharing layer %u: the packed projection has %u rows, expected heads * head_dim = %u", l, h.qkv_projection.n, h.num_heads * h.head_dim);
                return fail(UZU_ERR_INVALID_ARGUMENT);
            }
            if (h.use_rope) { // (tables are uploaded below; here only the index check)
                if (desc->num_ropes ? h.rope_index >= desc->num_ropes : desc->rope.kind == UZU_ROPE_NONE) {
                    set_error("model_create: layer %u rotates with a RoPE configuration the model does not carry", l);
                    return fail(UZU_ERR_INVALID_ARGUMENT);
                }
            }
            TRY(upload_linear(m, h.qkv_projection, &L.qkv));
            if (h.has_gate) TRY(upload_linear(m, h.gate_projection, &L.gate));
            TRY(upload_linear(m, h.out_projection, &L.out));
            TRY(upload_norm(m, h.query_norm, h.head_dim, &L.qn));
            TRY(upload_norm(m, h.key_norm, h.head_dim, &L.kn));
            if (h.has_sinks) {
                if (!h.sinks) return fail((set_error("model_create: layer %u has_sinks without a sinks tensor", l), UZU_ERR_INVALID_ARGUMENT));
                TRY(upload(m, h.sinks, (size_t)h.num_heads * 2, &L.sinks));
            }
            const uint32_t qdim = h.num_heads * h.head_dim;
            max_qkv = max_qkv > h.qkv_projection.n ? max_qkv : h.qkv_projection.n;
            max_qdim = max_qdim > qdim ? max_qdim : qdim;
            max_hd = max_hd > h.head_dim ? max_hd : h.head_dim;
            max_heads = max_heads > h.num_heads ? max_heads : h.num_heads;
        } else {
            TRY(upload_linear(m, h.dn_in_proj, &L.in_proj));
            TRY(upload_linear(m, h.dn_out_proj, &L.out_proj));
            const uint32_t key_dim = h.dn_num_groups * h.dn_head_dim, value_dim = h.dn_num_heads * h.dn_value_head_dim;
            const uint32_t conv_dim = 2 * key_dim + value_dim;
            TRY(upload(m, h.dn_conv_weights, (size_t)conv_dim * h.dn_kernel_size * 4, &L.conv_w));
            TRY(upload(m, h.dn_conv_biases, (size_t)conv_dim * 4, &L.conv_b));
            TRY(upload(m, h.dn_a_log, (size_t)h.dn_num_heads * 4, &L.a_log));
            TRY(upload(m, h.dn_dt_bias, (size_t)h.dn_num_heads * 4, &L.dt_bias));
            TRY(upload(m, h.dn_norm_scales, (size_t)h.dn_value_head_dim * 4, &L.dn_norm));
            L.conv_state_bytes = (size_t)conv_dim * (h.dn_kernel_size - 1) * 4;
            L.ssm_state_bytes = (size_t)h.dn_num_heads * h.dn_value_head_dim * h.dn_head_dim * 4;
            const uint32_t proj = conv_dim + value_dim + 2 * h.dn_num_heads;
            max_proj = max_proj > proj ? max_proj : proj;
            max_value = max_value > value_dim ? max_value : value_dim;
            max_key = max_key > key_dim ? max_key : key_dim;
            max_hv = max_hv > h.dn_num_heads ? max_hv : h.dn_num_heads;
        }
    }
    {   // one table pair per distinct RoPE configuration (Transformer::new dedups them, transformer.rs:101-118); a layer points at its own
        const uint32_t n_ropes = desc->num_ropes ? desc->num_ropes : (desc->rope.kind != UZU_ROPE_NONE ? 1u : 0u);
        m->ropes.resize(n_ropes);
        const size_t saved = m->weight_bytes;
        for (uint32_t r = 0; r < n_ropes; ++r) {
            const uzu_rope_desc& R = desc->num_ropes ? desc->ropes[r] : desc->rope;
            if (R.kind == UZU_ROPE_NONE || !R.head_dim) return fail((set_error("model_create: RoPE configuration %u is empty", r), UZU_ERR_INVALID_ARGUMENT));
            std::vector<float> c, sn;
            rope_tables(R, m->max_positions, c, sn);
            TRY(upload(m, c.data(), c.size() * 4, &m->ropes[r].cos));
            TRY(upload(m, sn.data(), sn.size() * 4, &m->ropes[r].sin));
            m->ropes[r].dim = R.head_dim;
        }
        m->weight_bytes = saved;
        for (DLayer& L : m->layers)
            if (L.d.mixer_kind == UZU_MIXER_ATTENTION && L.d.use_rope) {
                const uzu_hip_model::RopeTable& T = m->ropes[desc->num_ropes ? L.d.rope_index : 0];
                L.rope_cos = T.cos, L.rope_sin = T.sin, L.rope_dim = T.dim;
            }
    }
    if (m->tp && m->gemma_options) {
        set_error("model_create: post-layer scalars, embedding norm, KV sharing, value normalisation and per-layer embeddings are not sharded (single GPU only)");
        return fail(UZU_ERR_UNSUPPORTED);
    }
    void* p;
    const size_t C = m->chunk;
    // scratch blocks are zero-filled; UZU_HIP_POISON=2 hands them out full of NaNs instead (nothing may depend on the zeros).  ZALLOC: zero by contract
    const bool zero_scratch = poison_level() < 2;
#define ALLOC(field, type, elems) do { TRY(dev_alloc(m, (size_t)(elems) * sizeof(type), &p, zero_scratch)); m->field = (type*)p; } while (0)
#define ZALLOC(field, type, elems) do { TRY(dev_alloc(m, (size_t)(elems) * sizeof(type), &p, true)); m->field = (type*)p; } while (0)
    TRY(state_build(m, &m->state0));
    bind_state(m, m->state0);
    m->max_seqs = (flags >> 8) & 0xFFu ? (flags >> 8) & 0xFFu : 1u;
    const size_t CB = C * m->max_seqs; // rows of a batched pass: row-major activation buffers are sized for it
    ALLOC(batch_tokens, uint32_t, CB);
    ALLOC(hidden, uint16_t, CB * d);
    ALLOC(normed, uint16_t, CB * d);
    m->rowsum_floats = (size_t)(d / 32) * (CB + 4); // groups of >= 32 elements
    ALLOC(rowsum, float, m->rowsum_floats);
    {   // row sums filed by the producers of `gated` (parts of 64 columns) and `delta_out` (parts of one value head) -- sized at their allocations below
        uint32_t max_hidden = 0, max_heads = 0;
        for (uint32_t l = 0; l < desc->num_layers; ++l) {
            const uzu_layer_desc& h = desc->layers[l];
            if (h.mlp_kind != UZU_MLP_MOE && h.hidden_dim > max_hidden) max_hidden = h.hidden_dim;
            if (h.mixer_kind == UZU_MIXER_DELTA_NET && h.dn_num_heads > max_heads) max_heads = h.dn_num_heads;
        }
        if (max_hidden >= 64) {
            m->rs_gated.floats = (size_t)(max_hidden / 64) * (CB + 4);
            TRY(dev_alloc(m, m->rs_gated.floats * 4, &p, zero_scratch)); m->rs_gated.buf = (float*)p;
        }
        if (max_heads) {
            m->rs_delta.floats = (size_t)max_heads * (CB + 4);
            TRY(dev_alloc(m, m->rs_delta.floats * 4, &p, zero_scratch)); m->rs_delta.buf = (float*)p;
        }
    }
    ALLOC(mixed, uint16_t, CB * d);
    ALLOC(shortcut, uint16_t, CB * d);
    if (max_qkv) {
        ALLOC(qkv, uint16_t, CB * max_qkv);
        ALLOC(gate, uint16_t, CB * max_qdim);
        ALLOC(queries, uint16_t, CB * max_qdim);
        ALLOC(attn_out, uint16_t, CB * max_qdim);
        TRY(ensure_partials(m, max_heads, max_hd)); // decode rows; prefill grows it on demand
    }
    ALLOC(up, uint16_t, CB * 2 * max_hidden);
    ALLOC(gated, uint16_t, CB * max_hidden);
    if (max_moe_k) {
        const size_t rows = CB * max_moe_k;
        ALLOC(moe.topk_ids, int32_t, rows);
        ALLOC(moe.bucketed_ids, int32_t, rows);
        ALLOC(moe.tok2row, int32_t, rows);
        ALLOC(moe.topk_probs, uint16_t, rows);
        ALLOC(moe.bucketed_probs, uint16_t, rows);
        ALLOC(moe.x_perm, uint16_t, rows * d);
        ALLOC(moe.y_partial, uint16_t, rows * d);
        ALLOC(moe.offsets, uint32_t, max_moe_e + 1);
        ALLOC(moe.sumk, uint32_t, 1);
        ALLOC(moe.row_expert_map, uint32_t, rows);
        ALLOC(moe.hidden, float, rows * max_moe_ff);
    }
    if (max_proj) {
        ALLOC(in_proj, uint16_t, CB * max_proj);
        ALLOC(delta_out, uint16_t, CB * max_value);
        ALLOC(dn_ws, float, k::delta_net_chunk_workspace_bytes(max_hv, max_value, (uint32_t)C) / sizeof(float));
        ALLOC(dn_o, float, max_value);
        ALLOC(dn_sz, float, max_value);
        ALLOC(padded, float, (C + 8) * max_proj);
        ALLOC(qn, float, C * max_key);
        ALLOC(kn, float, C * max_key);
        ALLOC(beta, float, C * max_hv);
        ALLOC(decay, float, C * max_hv);
    }
    ALLOC(shortcut_b, uint16_t, C * d);
    if (desc->has_ple) {
        const size_t total = (size_t)desc->num_layers * desc->ple_dim;
        ALLOC(ple_inputs, uint16_t, CB * total);
        ALLOC(ple_token, uint16_t, CB * total);
        ALLOC(ple_projected, uint16_t, CB * total);
        ALLOC(ple_gate_out, uint16_t, CB * desc->ple_dim);
        ALLOC(ple_activated, uint16_t, CB * desc->ple_dim);
    }
    if (m->rht_max_k) ALLOC(rht_scratch, uint16_t, CB * m->rht_max_k);
    if (m->lora_max_rank) ALLOC(lora_scratch, uint16_t, CB * m->lora_max_rank);
    ALLOC(amax_val, float, kArgmaxPartials);
    ALLOC(amax_idx, uint32_t, kArgmaxPartials);
    if (max_qkv) {
        uint32_t wgs = 1; // kv_heads * head-subgroups of the widest attention layer
        for (auto& L : m->layers)
            if (L.d.mixer_kind == UZU_MIXER_ATTENTION) {
                const uint32_t gqa = L.d.num_heads / L.d.num_groups;
                const uint32_t w = L.d.num_groups * (gqa / k::attn_dec_group_size(gqa));
                wgs = wgs > w ? wgs : w;
            }
        // ~256 workgroups (one per CU); more splits shorten attn_dec but lengthen attn_merge (measured: 64 best at 2k context).
        // Long contexts are latency bound on the chain of K / V batches a key group walks (4 keys each, ~1 us per batch):
        // twice the workgroups halve it (Qwen3-14B-class at 8k: 37 us per layer with 320 workgroups of 16 batches)
        uint32_t splits = (desc->max_context_length >= 4096 ? 512 : 256) / wgs;
        if (const char* ev = lab_env("UZU_DEC_SPLITS")) splits = (uint32_t)atoi(ev);
        m->dec_splits = splits < 8 ? 8 : (splits > 128 ? 128 : splits);
        ALLOC(dec_partials, float, (size_t)max_heads * m->dec_splits * max_hd);
        ALLOC(dec_sums, float, (size_t)max_heads * m->dec_splits);
        ALLOC(dec_maxs, float, (size_t)max_heads * m->dec_splits);
        ZALLOC(dec_tickets, uint32_t, (size_t)max_heads); // monotonic arrival counters start at 0 (groups <= heads)
    }
    ALLOC(last_normed, uint16_t, d);
    ALLOC(logits, uint16_t, desc->vocab_size);
    if (m->tp) {
        ALLOC(tp_buf, float, CB * d);
        ALLOC(tp_key, unsigned long long, k::kDnTreeMaxNodes);
    }
    TRY(dev_alloc(m, k::argmax_scratch_bytes(1), &m->argmax_scratch));
    TRY(dev_alloc(m, k::unified_sampling_scratch_bytes(1), &m->sampling_scratch));
    TRY(dev_alloc(m, 8, &p));
    m->d_seed = (uint64_t*)p;
    if (flags & UZU_MODEL_DEBUG_TAPS) ALLOC(taps, uint16_t, (size_t)desc->num_layers * C * d);
#undef ALLOC
#undef ZALLOC
#undef TRY
    m->fusable = model_fusable(m);
    if (hipEventCreate(&m->ev0) != hipSuccess || hipEventCreate(&m->ev1) != hipSuccess) {
        set_error("model_create: hipEventCreate failed");
        return fail(UZU_ERR_HIP);
    }
    // the fills of the state and scratch blocks and the load-time tables are done before the model is handed out (host-side readers
    // -- hipMemcpy on the null stream -- are not ordered behind the engine's stream)
    if (hipStreamSynchronize(m->ctx->stream) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        set_error("model_create: synchronisation after load failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(UZU_ERR_HIP);
    }
    *out = m;
    return UZU_OK;
}

} // extern "C"
namespace uzu {
namespace eng {
void drop_tree_graphs(uzu_hip_model* m, uzu_hip_state* st) {
    auto& gs = m->tree.graphs;
    for (size_t i = 0; i < gs.size();) {
        if (!st || gs[i].state == st) {
            (void)hipGraphExecDestroy(gs[i].exec);
            gs.erase(gs.begin() + i);
        } else {
            ++i;
        }
    }
}
} // namespace eng
} // namespace uzu
extern "C" {

void uzu_hip_model_destroy(uzu_hip_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    drop_tree_graphs(m, nullptr);
    if (m->bound) m->bound->graph_single = m->graph_single, m->bound->graph_two = m->graph_two;
    state_free(m->state0);
    // states created with uzu_hip_state_create belong to the caller; one that outlives its model loses its device memory here and
    // is neutralised (m = null), so that the caller's later uzu_hip_state_destroy only deletes the host struct
    for (uzu_hip_state* st : m->user_states) {
        state_release(st);
        st->m = nullptr;
    }
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    for (void* p : m->allocations) (void)hipFree(p);
    for (size_t b : m->allocation_bytes) m->ctx->current_bytes -= b < m->ctx->current_bytes ? b : m->ctx->current_bytes;
    delete m;
}

uzu_status uzu_hip_model_reset(uzu_hip_model* m) {
    UZU_REQUIRE(m, "model_reset: null model");
    hipStream_t s = m->ctx->stream;
    HIPCHK(hipMemsetAsync(m->d_ctx_len, 0, 4, s));
    for (auto& L : m->layers) {
        if (L.conv_state) HIPCHK(hipMemsetAsync(L.conv_state, 0, L.conv_state_bytes, s));
        if (L.ssm_state) HIPCHK(hipMemsetAsync(L.ssm_state, 0, L.ssm_state_bytes, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    m->context_length = 0;
    m->hidden_ready = false;
    if (m->tree.state == m->bound) m->tree.size = 0, m->tree.state = nullptr; // a pending tree of this sequence is dropped
    return UZU_OK;
}

// ---- sequence states (LanguageModelState, engine/language_model/state.rs:9-16) ----
uzu_status uzu_hip_state_create(uzu_hip_model* m, uzu_hip_state** out) {
    UZU_REQUIRE(m && out, "state_create: null argument");
    (void)hipSetDevice(m->ctx->device);
    UZU_PROPAGATE(state_build(m, out));
    m->user_states.push_back(*out);
    HIPCHK(hipStreamSynchronize(m->ctx->stream)); // the zero fills of the new caches (state_alloc)
    return UZU_OK;
}
void uzu_hip_state_destroy(uzu_hip_state* st) {
    if (!st) return;
    uzu_hip_model* m = st->m;
    if (!m) { // the model went first (uzu_hip_model_destroy released the device side)
        delete st;
        return;
    }
    for (size_t i = 0; i < m->user_states.size(); ++i)
        if (m->user_states[i] == st) {
            m->user_states.erase(m->user_states.begin() + i);
            break;
        }
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    drop_tree_graphs(m, st); // captured tree passes carry this state's cache pointers
    if (m->tree.state == st) m->tree.size = 0, m->tree.state = nullptr;
    if (m->bound == st) { // hand the model back to its own state first
        st->graph_single = m->graph_single, st->graph_two = m->graph_two;
        m->bound = nullptr;
        bind_state(m, m->state0);
    }
    if (st != m->state0) state_free(st);
}
uzu_status uzu_hip_model_bind_state(uzu_hip_model* m, uzu_hip_state* st) {
    UZU_REQUIRE(m, "model_bind_state: null model");
    UZU_REQUIRE(!st || st->m == m, "model_bind_state: the state belongs to another model");
    bind_state(m, st ? st : m->state0);
    return UZU_OK;
}
uzu_status uzu_hip_state_reset(uzu_hip_state* st) {
    UZU_REQUIRE(st && st->m, "state_reset: null state (or its model was destroyed)");
    uzu_hip_model* m = st->m;
    uzu_hip_state* prev = m->bound;
    bind_state(m, st);
    const uzu_status r = uzu_hip_model_reset(m);
    bind_state(m, prev);
    return r;
}
// dst <- src: KV caches, DeltaNet conv / SSM states, token history, context length (both states of ONE model: the same buffers in the same
// order).  A prompt prefix prefilled once can so be continued many times (tools/parity_census.py: prompts that share all but their tail).
uzu_status uzu_hip_state_copy(uzu_hip_state* dst, const uzu_hip_state* src) {
    UZU_REQUIRE(dst && src && dst->m && dst->m == src->m, "state_copy: null state, or states of different models");
    if (dst == src) return UZU_OK;
    uzu_hip_model* m = dst->m;
    UZU_REQUIRE(dst->allocations.size() == src->allocations.size() && dst->allocation_bytes == src->allocation_bytes, "state_copy: the states differ in layout");
    (void)hipSetDevice(m->ctx->device);
    hipStream_t s = m->ctx->stream;
    for (size_t i = 0; i < dst->allocations.size(); ++i) HIPCHK(hipMemcpyAsync(dst->allocations[i], src->allocations[i], dst->allocation_bytes[i], hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    const uint32_t len = m->bound == src ? m->context_length : src->context_length;
    dst->context_length = len;
    if (m->bound == dst) m->context_length = len, m->hidden_ready = false, m->tree.size = 0, m->tree.state = nullptr;
    return UZU_OK;
}
uint32_t uzu_hip_state_context_length(const uzu_hip_state* st) {
    if (!st || !st->m) return 0;
    return st->m->bound == st ? st->m->context_length : st->context_length;
}


} // extern "C"
