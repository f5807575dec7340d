// engine.hip -- model driver above the kernel boundary (include/uzu_hip_engine.h).
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
#include <math.h>
#include <string.h>

#include <vector>

#include "../../include/uzu_hip_engine.h"
#include "internal.h"
#include "kernels.h"
#include "kernels_decode.h"
#include "tp.h"

using namespace uzu;

namespace {

constexpr uint32_t kSuffixCapacity = 1024; // ATTENTION_SUFFIX_CAPACITY, mixer/attention/state.rs:14: the reference's rows per forward pass (and its single- / two-pass rule)
// Rows of one PREFILL pass here (uzu_hip_model::chunk; UZU_PREFILL_CHUNK = 1024 ... 8192, a multiple of 1024).  The reference feeds a prompt in
// passes of <= 1024 tokens; on this chip a 1024-row GEMM of a 0.8B model is 64-128 tiles for 256 CUs, so the default pass is 2048 rows (same
// results up to the order of a split-K sum: the DeltaNet chunks and the attention key tiles of a 1024-aligned boundary do not move).
static uint32_t prefill_chunk_rows() {
    const char* e = getenv("UZU_PREFILL_CHUNK");
    const long v = e ? atol(e) : 2048;
    return (v >= 1024 && v <= 8192 && v % 1024 == 0) ? (uint32_t)v : 2048u;
}
constexpr uint32_t kArgmaxPartials = 4096; // capacity of the read-out GEMV's per-workgroup arg-max partials (DecGemvParams::part_capacity)

struct DLinear {
    uint32_t n = 0, k = 0, bits = 0, group = 0, method = UZU_QUANT_NONE;
    void* w = nullptr;
    void* scales = nullptr;
    void* biases = nullptr;
    uint8_t* zp = nullptr;
    void* out_biases = nullptr;
    int32_t* in_signs = nullptr;  // HybridSpec InputOutput (RHTLinearWrapper): sign factors of the input / output Hadamard transforms
    int32_t* out_signs = nullptr;
    // the same factors as one bit per element (bit i of word s: element 32 s + i is -1) for the fused decode step's Hadamard prologue
    // (k_decode.hip, PRO == 3); null when a factor is not +-1 -- then the model decodes through the reference's kernel sequence
    uint32_t *in_bits = nullptr, *out_bits = nullptr;
    std::vector<uint32_t> in_words; // host copy of in_bits (two linears behind one prologue must share it)
    uint32_t lora_rank = 0;       // HybridSpec with a LowRankSpec adapter (QLoRALinearWrapper): bf16 [rank, k] and [n, rank]
    uint16_t *adapter_down = nullptr, *adapter_up = nullptr;
    float* coef = nullptr;        // [groups][n] f32: the prefill GEMM's offset coefficients (MatmulParams::pre_coef), tabulated once at load
};
struct DNorm {
    bool present = false;
    uint32_t full_layer = 0, subtract_mean = 0;
    float eps = 0.f, offset = 0.f;
    float* scales = nullptr;
    float* biases = nullptr;
    // PostLayerScalar (encodable_block/normalization.rs:17-21,76-80): 0 None, 1 ScaleResidualSum(scalar), 2 ScaleOutput(scalar)
    int scalar_mode = 0;
    float scalar = 1.0f;
};
struct DLayer {
    uzu_layer_desc d; // scalars only (pointers are host pointers: never dereferenced after create)
    DNorm pre_mixer, post_mixer, pre_mlp, post_mlp, qn, kn;
    DLinear qkv, gate, out, in_proj, out_proj, up, down;
    float *conv_w = nullptr, *conv_b = nullptr, *a_log = nullptr, *dt_bias = nullptr, *dn_norm = nullptr;
    uint16_t* sinks = nullptr; // bf16 [heads] (has_sinks)
    // the layer's RoPE configuration (uzu_model_desc::ropes[rope_index], or the model's single `rope`): tables [max positions][rope_dim]
    float *rope_cos = nullptr, *rope_sin = nullptr;
    uint32_t rope_dim = 0;
    // PerLayerEmbeddingProjection (per_layer_embedding.rs:150-271) at the end of the layer (d.has_ple)
    DLinear ple_gate, ple_projection;
    DNorm ple_norm;
    uint16_t *keys = nullptr, *values = nullptr; // a KV-sharing layer (d.is_kv_sharing): its source layer's rows (bind_state)
    // the last layer that reads this layer's KV state in a pass (itself, or the last layer sharing it): a ring takes the pass's suffix rows
    // (encode_accept, after the WHOLE pass in the reference: stream.rs:441-444) only once that layer has run
    uint32_t last_reader = 0;
    float *conv_state = nullptr, *ssm_state = nullptr;
    size_t conv_state_bytes = 0, ssm_state_bytes = 0;
};

} // namespace

// LanguageModelState (engine/language_model/state.rs:9-16): everything that belongs to ONE sequence -- KV caches,
// DeltaNet conv / SSM states, the device-resident decode control block and the captured decode graphs (their nodes
// carry this state's pointers).  Weights and scratch stay in the model; a model works on its currently BOUND state,
// whose pointers are mirrored in the DLayer / model fields the encoders read (bind_state).
struct uzu_hip_state {
    uzu_hip_model* m = nullptr;
    struct Layer {
        uint16_t *keys = nullptr, *values = nullptr;
        float *conv_state = nullptr, *ssm_state = nullptr;
    };
    std::vector<Layer> layers;
    uint32_t *d_ctx_len = nullptr, *d_tokens = nullptr, *d_out_token = nullptr, *d_sampled = nullptr;
    uint32_t context_length = 0;
    hipGraphExec_t graph_single = nullptr, graph_two = nullptr;
    uint32_t graph_epoch = 0; // sampling_epoch of the model when the graphs were captured (they bake the sampling kernels in)
    std::vector<void*> allocations;
    std::vector<size_t> allocation_bytes; // parallel to `allocations` (uzu_hip_state_copy)
    size_t bytes = 0;
};

struct uzu_hip_model {
    uzu_hip_context* ctx = nullptr;
    uzu_hip_state* state0 = nullptr; // the model's own sequence state (uzu_hip_model_create); owned
    uzu_hip_state* bound = nullptr;  // state whose pointers the fields below currently mirror
    std::vector<uzu_hip_state*> user_states; // live states of uzu_hip_state_create (neutralised if the model is destroyed first)
    uint32_t flags = 0;
    uzu_model_desc d; // scalars only
    std::vector<DLayer> layers;
    DLinear embedding, output_embedding;
    DNorm output_norm;
    std::vector<void*> allocations;
    std::vector<size_t> allocation_bytes; // parallel to `allocations` (context memory accounting)
    size_t weight_bytes = 0;

    struct RopeTable {
        float *cos = nullptr, *sin = nullptr;
        uint32_t dim = 0;
    };
    std::vector<RopeTable> ropes; // one per distinct RoPE configuration (transformer.rs:101-118)
    DNorm embedding_norm;         // decoder.rs:68-83
    // PerLayerEmbedding (per_layer_embedding.rs:36-148): token table + projection of the embedded rows -> per_layer_inputs [rows][layers][ple_dim]
    DLinear ple_token_embedding, ple_model_projection;
    DNorm ple_projection_norm;
    uint16_t *ple_inputs = nullptr, *ple_token = nullptr, *ple_projected = nullptr; // [rows][layers * ple_dim]
    uint16_t *ple_gate_out = nullptr, *ple_activated = nullptr;                      // [rows][ple_dim]
    bool gemma_options = false; // any option that only the one-kernel-per-reference-kernel pass implements (no fused decode step, no tensor parallelism)
    // device-resident sequence state
    uint32_t* d_ctx_len = nullptr;  // current context length
    uint32_t* d_tokens = nullptr;   // [1024] input token ids of the pass
    uint32_t* d_out_token = nullptr;
    uint32_t* d_sampled = nullptr;  // [max positions] token sampled from the row at absolute position p
    uint32_t context_length = 0;    // host mirror of *d_ctx_len
    uint32_t max_positions = 0;
    uint32_t chunk = kSuffixCapacity; // rows of one prefill pass (prefill_chunk_rows())
    uint32_t max_seqs = 1;          // sequences one batched prefill pass may carry (UZU_MODEL_BATCH(n) at creation)
    uint32_t* batch_tokens = nullptr; // [max_seqs * 1024] token ids of a batched pass

    // scratch (sized for one 1024-token chunk)
    uint16_t *hidden = nullptr, *normed = nullptr, *mixed = nullptr, *shortcut = nullptr;
    uint16_t *qkv = nullptr, *gate = nullptr, *queries = nullptr, *attn_out = nullptr;
    uint16_t *up = nullptr, *gated = nullptr;
    uint16_t *in_proj = nullptr, *delta_out = nullptr;
    float *padded = nullptr, *qn = nullptr, *kn = nullptr, *beta = nullptr, *decay = nullptr;
    float *partials = nullptr, *sums = nullptr, *maxs = nullptr;
    uint32_t partial_rows = 0;
    uint16_t *last_normed = nullptr, *logits = nullptr;
    void* argmax_scratch = nullptr;
    // SamplingMethod::Stochastic for the engine's own prefill / decode loop (uzu_hip_model_set_sampling); greedy when !on
    struct {
        bool on = false;
        uint64_t seed = 0;
        k::UnifiedSamplingParams p{};
    } sampling;
    uint64_t* d_seed = nullptr;
    void* sampling_scratch = nullptr;
    uint32_t sampling_epoch = 0;
    uint16_t* taps = nullptr; // [layers][chunk][d] (chunk = rows of one prefill pass)
    uint32_t tap_rows = 0;

    // fused decode path
    bool fusable = false;
    uint16_t* shortcut_b = nullptr; // ping-pong partner of `shortcut`
    uint16_t* rht_scratch = nullptr; // [rows][widest RHT input]: InputRht works on a copy of the rows
    // group row sums of `normed`, written by the normalisation that produced it (MatmulParams::pre_rowsum of the GEMMs that read it):
    // valid for (rs_rows rows, rs_k elements, groups of rs_group) until `normed` is written again
    float* rowsum = nullptr;
    size_t rowsum_floats = 0;
    uint32_t rs_rows = 0, rs_k = 0, rs_group = 0;
    uint32_t rht_max_k = 0;
    uint16_t* lora_scratch = nullptr; // [rows][widest adapter rank]: x down^T of a QLoRA linear
    uint32_t lora_max_rank = 0;
    float *dec_partials = nullptr, *dec_sums = nullptr, *dec_maxs = nullptr;
    uint32_t* dec_tickets = nullptr; // one monotonic arrival counter per (kv head, head sub-group): attn_dec's in-launch pass 2
    float* dn_ws = nullptr; // chunked DeltaNet prefill: T / P matrices of one 1024-token pass (k_deltanet_chunk.hip)
    float *dn_o = nullptr, *dn_sz = nullptr; // raw DeltaNet outputs and SiLU(z) of the decode token (f32 [value_dim])
    uint32_t dec_splits = 0;
    float* amax_val = nullptr;
    uint32_t* amax_idx = nullptr;

    // tensor parallel (tp.hip): this model holds one shard; row-parallel linears exchange f32 partial sums
    uzu::tp::Comm* tp = nullptr; // borrowed; null => single GPU
    uint32_t vocab_offset = 0;   // first vocabulary row of this rank's read-out shard
    float* tp_buf = nullptr;     // [1024 rows][model_dim] f32 partial sums
    unsigned long long* tp_key = nullptr; // [kDnTreeMaxNodes] packed (logit, index) keys: one per sampled row
    // stochastic sampling over the vocab-sharded read-out: the ranks' logit shards gathered into whole rows (tp::gather_logits)
    float* tp_gather_f32 = nullptr;   // [tp_gather_rows][vocab]
    uint16_t* tp_gather_bf16 = nullptr;
    uint32_t tp_gather_rows = 0;

    // A speculated tree between uzu_hip_model_verify_tree and uzu_hip_model_accept (stream.rs:556-628, 380-470): the attention layers
    // keep the suffix rows behind the caches' logical end, a DeltaNet layer keeps its DeltaNetSuffixStatus::Tree (delta_net.rs:39-46).
    struct TreeLayer {
        float* conv_states = nullptr;             // f32 [nodes, conv_dim, k-1]
        uint16_t *k = nullptr, *v = nullptr;      // bf16 [nodes, key_dim] / [nodes, value_dim]
        float *log_decay = nullptr, *beta = nullptr; // f32 [nodes, Hv]
    };
    struct {
        std::vector<TreeLayer> layers;
        uint32_t* d_trie = nullptr;     // [kDnTreeMaxNodes][3]
        int32_t* d_parents = nullptr;   // [kDnTreeMaxNodes]
        uint32_t* d_sampled = nullptr;  // [kDnTreeMaxNodes] token sampled at every node
        uint32_t* d_accepted = nullptr; // [kDnTreeMaxNodes] accepted node indices of the accept in flight
        uint16_t* q = nullptr;          // bf16 [nodes, widest key_dim] (scratch of one layer)
        uint16_t* normed = nullptr;     // bf16 [nodes, model_dim]: output norm of every node
        uint16_t* logits = nullptr;     // bf16 [nodes, vocab rows of this rank]
        void* argmax_scratch = nullptr;
        uint64_t* d_seeds = nullptr;    // [kDnTreeMaxNodes] per-node sampling seeds (stochastic sampling)
        bool host_seeds = false;        // d_seeds was uploaded by the caller (the trie's token_seeds, stream.rs:694) instead of derived on the device
        void* sampling_scratch = nullptr;
        bool allocated = false;
        bool active = false;            // the forward pass being encoded is a tree pass
        float last_gpu_ms = 0.f;        // device time of the last tree pass (events around the launches / the graph replay)
        uint32_t size = 0;              // nodes of the pending tree (0 = none)
        uzu_hip_state* state = nullptr; // the state it hangs off
        std::vector<int32_t> parents;
        std::vector<uint32_t> sampled;
        // the tree pass of (sequence state, node count, attention regime) as a hipGraph: everything that changes from pass to pass --
        // token ids, trie nodes, parents, context length -- sits in device buffers the kernels read, so the captured pass is replayable
        struct Graph {
            uzu_hip_state* state;
            uint32_t nodes;
            bool two_pass;
            bool host_seeds; // the pass takes the caller's per-node seeds (no derive_tree_seeds launch in it)
            hipGraphExec_t exec;
            uint32_t launches;
        };
        std::vector<Graph> graphs;
        uint32_t graph_epoch = 0; // sampling_epoch the graphs were captured under
    } tree;

    hipGraphExec_t graph_single = nullptr, graph_two = nullptr;
    uint32_t graph_epoch = 0; // sampling_epoch of the model when the graphs were captured (they bake the sampling kernels in)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool hidden_ready = false; // row 0 of `hidden` already holds the embedding of the next input token (written by the fused commit)
    uint32_t launches = 0; // kernel launches of the last encoded forward
    void* prof_sink = nullptr; // std::vector<ProfEntry>* while profiling one step
    int regime_override = -1; // graph capture: 0 = single-pass attention, 1 = two-pass (else decided by context_length)
};

namespace {

#define HIPCHK(expr) UZU_HIP_TRY(expr)

uzu_status dev_alloc(uzu_hip_model* m, size_t bytes, void** out, bool zero = false) {
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

uzu_status state_alloc(uzu_hip_state* st, size_t bytes, void** out) {
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
    HIPCHK(hipMemsetAsync(p, 0, alloc, ctx->stream)); // stream-ordered with every pass that will use the state (see dev_alloc)
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
            need(kv_bytes, &p), st->layers[l].keys = (uint16_t*)p;
            need(kv_bytes, &p), st->layers[l].values = (uint16_t*)p;
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

template <class T> uzu_status upload(uzu_hip_model* m, const void* host, size_t bytes, T** out) {
    if (!host) {
        *out = nullptr;
        return UZU_OK;
    }
    void* p;
    UZU_PROPAGATE(dev_alloc(m, bytes, &p));
    HIPCHK(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    m->weight_bytes += bytes;
    *out = (T*)p;
    return UZU_OK;
}

uzu_status upload_linear(uzu_hip_model* m, const uzu_linear_desc& h, DLinear* o, bool is_embedding = false) {
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

// ---------------------------------------------------------------------------------- encoding helpers
struct ProfEntry {
    const char* name;
    size_t bytes;
    hipEvent_t e0, e1; // recorded on the stream around the launch (fallback)
    hipEvent_t x0, x1; // stamped by the launch itself (hipExtLaunchKernel): the kernel's begin -> end
};
struct Enc {
    uzu_hip_model* m;
    hipStream_t s;
    uzu_status st = UZU_OK;
    std::vector<ProfEntry>* prof = nullptr;
    hipEvent_t pending = nullptr;
    LaunchTimer timer{nullptr, nullptr};
    // begin(): called before a launch when profiling; run(): after it
    void begin() {
        if (!prof) return;
        (void)hipEventCreate(&pending);
        (void)hipEventRecord(pending, s);
        (void)hipEventCreate(&timer.start);
        (void)hipEventCreate(&timer.stop);
        tl_launch_timer = &timer;
    }
    void run(uzu_status r, const char* name = "other", size_t bytes = 0) {
        if (st == UZU_OK) st = r;
        ++m->launches;
        if (prof && pending) {
            tl_launch_timer = nullptr;
            hipEvent_t e1;
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e1, s);
            prof->push_back({name, bytes, pending, e1, timer.start, timer.stop});
            pending = nullptr;
        }
    }
};

#define RUN(name, bytes, expr) do { e.begin(); e.run((expr), name, bytes); } while (0)

// `row_parallel`: under tensor parallelism this linear's K is split over the ranks (out-proj, down-proj): the matmul
// writes f32 partial sums, the ranks all-reduce them, and the sum is rounded to bf16 into `output`.
//
// RHT linears (RHTLinearWrapper::encode_input, linear/rht_wrapper.rs:215-298, full-precision activation format): InputRht on a
// copy of the rows (the reference transforms its own allocation in place), the inner matmul without its bias, OutputRht in place
// on the result, then the bias (MatmulDOps::rht_factors, kernel.rs:296-303).
// QLoRALinearWrapper::encode (linear/qlora_wrapper.rs:177-251): intermediate = x down^T; base input = InputRht of a copy of the rows (when the
// spec carries signs); output = base matmul (no bias); output += intermediate up^T (MatmulDOps::accumulate); OutputRht in place.
void linear_qlora(Enc& e, const DLinear& L, const uint16_t* input, uint16_t* output, uint32_t batch) {
    uzu_hip_model* m = e.m;
    auto fp = [&](const uint16_t* a, const uint16_t* b, uint16_t* d, uint32_t n, uint32_t k, bool accumulate) {
        k::MatmulParams p{};
        p.a = a, p.b = b, p.d = d, p.w_dt = p.a_dt = p.d_dt = UZU_BF16, p.b_kind = UZU_MATMUL_B_FULL_PRECISION, p.bits = 16, p.ab_scale = 1.0f;
        p.accumulate = accumulate ? 1u : 0u, p.m = batch, p.n = n, p.k = k;
        const char* variant = "matmul";
        e.begin();
        const uzu_status r = k::matmul(e.s, p, m->ctx->num_cus, &variant);
        e.run(r, "matmul_adapter", k::matmul_algorithmic_bytes(p));
    };
    fp(input, L.adapter_down, m->lora_scratch, L.lora_rank, L.k, false);
    const uint16_t* base_input = input;
    if (L.in_signs) {
        RUN("activation_transform", 0, k::activation_transform(e.s, input, m->rht_scratch, nullptr, nullptr, nullptr, L.in_signs, UZU_BF16, batch, L.k,
                                                                UZU_ACTIVATION_TRANSFORM_INPUT_RHT, 0, 0));
        base_input = m->rht_scratch;
    }
    k::MatmulParams p{};
    p.a = base_input, p.b = L.w, p.scales = L.scales, p.biases = L.biases, p.zero_points = L.zp, p.d = output;
    p.w_dt = p.a_dt = p.d_dt = UZU_BF16;
    p.b_kind = L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.bits = L.bits, p.group_size = L.group, p.ab_scale = 1.0f, p.m = batch, p.n = L.n, p.k = L.k;
    const char* variant = "matmul";
    e.begin();
    const uzu_status r = k::matmul(e.s, p, m->ctx->num_cus, &variant);
    e.run(r, variant, k::matmul_algorithmic_bytes(p));
    fp(m->lora_scratch, L.adapter_up, output, L.n, L.lora_rank, true);
    if (L.out_signs)
        RUN("activation_transform", 0, k::activation_transform(e.s, nullptr, output, nullptr, nullptr, nullptr, L.out_signs, UZU_BF16, batch, L.n,
                                                                UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
}

// the large-tile GEMM's offset tables where the engine has them: coefficients from load time, row sums of `normed` from its normalisation
static void offset_tables(uzu_hip_model* m, const DLinear& L, const uint16_t* input, uint32_t batch, k::MatmulParams* p) {
    static const bool enabled = [] { // UZU_GEMM_TABLES=0: the GEMM's own pre-pass launch every time (A/B runs)
        const char* v = getenv("UZU_GEMM_TABLES");
        return !v || atoi(v) != 0;
    }();
    if (!enabled || batch < 128) return;
    p->pre_coef = L.coef;
    if (input == m->normed && m->rs_rows == batch && m->rs_k == L.k && m->rs_group == L.group) p->pre_rowsum = m->rowsum;
}

// `post`: the Normalization that reads `output` next (PostNorm below): a split-K prefill GEMM then ends with one reduction + epilogue + normalisation
// launch and sets post->done; on every other path the caller runs the normalisation itself.
struct PostNorm {
    k::NormParams p{};
    uint32_t done = 0;
};
void linear(Enc& e, const DLinear& L, const uint16_t* input, uint16_t* output, uint32_t batch, bool row_parallel = false, PostNorm* post = nullptr) {
    if (L.lora_rank) return linear_qlora(e, L, input, output, batch); // (tensor-parallel shards of QLoRA linears are refused by the planner)
    const bool exchange = row_parallel && e.m->tp != nullptr;
    if (L.in_signs) {
        RUN("activation_transform", 0, k::activation_transform(e.s, input, e.m->rht_scratch, nullptr, nullptr, nullptr, L.in_signs, UZU_BF16, batch, L.k,
                                                                UZU_ACTIVATION_TRANSFORM_INPUT_RHT, 0, 0));
        input = e.m->rht_scratch;
    }
    k::MatmulParams p{};
    p.a = input, p.b = L.w, p.scales = L.scales, p.biases = L.biases, p.zero_points = L.zp, p.d = output, p.bias = L.out_biases;
    p.w_dt = p.a_dt = p.d_dt = UZU_BF16;
    if (L.out_signs) p.bias = nullptr; // bias_after_rht
    if (exchange) p.d = e.m->tp_buf, p.d_dt = UZU_F32;
    p.b_kind = L.method == UZU_QUANT_NONE ? UZU_MATMUL_B_FULL_PRECISION
             : L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS
             : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.bits = L.bits, p.group_size = L.group, p.ab_scale = 1.0f;
    p.m = batch, p.n = L.n, p.k = L.k;
    offset_tables(e.m, L, input, batch, &p);
    if (post && !exchange && !L.out_signs && batch >= 128) p.post_norm = &post->p, p.post_norm_done = &post->done;
    const char* variant = "matmul";
    e.begin();
    const uzu_status r = k::matmul(e.s, p, e.m->ctx->num_cus, &variant);
    e.run(r, post && post->done ? "gemm_q_mfma128+norm" : variant, k::matmul_algorithmic_bytes(p));
    if (exchange) {
        const size_t count = (size_t)batch * L.n;
        RUN("all_reduce", count * 4, tp::all_reduce_sum_f32(e.m->tp, e.s, e.m->tp_buf, count, output)); // sums rounded to bf16 into `output`
    }
    if (L.out_signs) {
        RUN("activation_transform", 0, k::activation_transform(e.s, nullptr, output, nullptr, nullptr, nullptr, L.out_signs, UZU_BF16, batch, L.n,
                                                                UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
        if (L.out_biases) RUN("tensor_add_bias", 0, k::tensor_add_bias(e.s, output, L.out_biases, output, UZU_BF16, UZU_BF16, L.n, (size_t)batch * L.n));
    }
}

// up projection + GatedActMul in one kernel (the matrix-core GEMM's epilogue pairs the up and gate columns of an output);
// false = not available for this shape / mode: the caller runs the two kernels
bool linear_gated(Enc& e, const DLinear& L, const uint16_t* input, uint16_t* gated_out, uint32_t batch, uint32_t act_type) {
    static const bool enabled = [] {
        const char* v = getenv("UZU_GEMM_ACT");
        return !v || atoi(v) != 0;
    }();
    // RHT / QLoRA linears run as their wrappers compose them: the fused GEMM knows nothing of the adapter term (x down^T) up^T
    if (!enabled || L.in_signs || L.out_signs || L.out_biases || L.lora_rank || L.method == UZU_QUANT_NONE || (e.m->flags & UZU_MODEL_NO_FUSION)) return false;
    k::MatmulParams p{};
    p.a = input, p.b = L.w, p.scales = L.scales, p.biases = L.biases, p.zero_points = L.zp, p.d = gated_out;
    p.w_dt = p.a_dt = p.d_dt = UZU_BF16;
    p.b_kind = L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.bits = L.bits, p.group_size = L.group, p.ab_scale = 1.0f;
    p.m = batch, p.n = L.n, p.k = L.k;
    if (!k::matmul_act_mul_supported(e.s, p, e.m->ctx->num_cus)) return false;
    p.act_mul = 1, p.act_type = act_type;
    offset_tables(e.m, L, input, batch, &p);
    const char* variant = "matmul";
    e.begin();
    const uzu_status r = k::matmul(e.s, p, e.m->ctx->num_cus, &variant);
    e.run(r, "gemm_q_mfma+act", k::matmul_algorithmic_bytes(p));
    return true;
}

bool norm_fusable(const DNorm& N);
// Two or three rows (small speculative verify passes, prefill tails): the Normalization as the PROLOGUE of the linear that reads its rows
// (k_gemv_rows.hip, RowsNorm) -- one launch instead of two, the rows bit-identical to the separate kernel's.  mode 1 copy / 2 add; the residual rows go
// from `sc_in` to `sc_out` (two buffers: the other workgroups still read sc_in); `normed_out` (optional) receives the normalised rows for a second
// linear.  gated: L is the fused up | gate matrix and `output` = GatedActMul of its halves.  false = not available: the caller runs the separate kernels.
bool linear_normed(Enc& e, const DNorm& N, int mode, const DLinear& L, const uint16_t* x, const uint16_t* sc_in, uint16_t* sc_out, uint16_t* normed_out, uint16_t* output,
                   uint32_t rows, bool gated, uint32_t act_type) {
    const char* env = getenv("UZU_ROWS_NORM"); // =0: the separate Normalization launch (A/B runs, tests; read when a pass is encoded or captured)
    const bool enabled = !env || atoi(env) != 0;
    uzu_hip_model* m = e.m;
    // rows <= 3 only: EVERY workgroup of the linear normalises all the rows it stages (hundreds of workgroups x rows x two passes through the L2), which
    // costs more than the launch it saves from 4 rows on -- measured on Qwen3.5-0.8B (profiles/r5_verify_cost.json, `rows_norm_ab`): 2 nodes 1560 -> 1492 us,
    // 4 nodes 1603 -> 1656, 8 nodes 1711 -> 1996, 16 nodes 1959 -> 2794 with the prologue at every size
    static const uint32_t max_rows = [] {
        const char* v = getenv("UZU_ROWS_NORM_MAX");
        return (uint32_t)(v && atoi(v) > 0 ? atoi(v) : 3);
    }();
    if (!enabled || rows < 2 || rows > max_rows || rows > 16 || k::exact_mode() || (m->flags & UZU_MODEL_NO_FUSION) || !norm_fusable(N) || mode == 0) return false;
    if (L.in_signs || L.out_signs || L.lora_rank || L.method == UZU_QUANT_NONE || L.bits != 4 || (gated && L.out_biases)) return false;
    k::MatmulParams p{};
    p.a = x, p.b = L.w, p.scales = L.scales, p.biases = L.biases, p.zero_points = L.zp, p.d = output, p.bias = L.out_biases;
    p.w_dt = p.a_dt = p.d_dt = UZU_BF16;
    p.b_kind = L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.bits = L.bits, p.group_size = L.group, p.ab_scale = 1.0f, p.m = rows, p.n = L.n, p.k = L.k;
    if (gated) p.act_mul = 1, p.act_type = act_type;
    if (!k::gemv_rows_norm_supported(p)) return false;
    k::RowsNorm rn{};
    rn.scales = N.scales, rn.eps = N.eps, rn.offset = N.offset, rn.full_layer = N.full_layer, rn.residual_add = mode == 2;
    rn.shortcut_in = mode == 2 ? sc_in : nullptr, rn.shortcut_out = sc_out, rn.normed_out = normed_out;
    e.begin();
    e.run(k::gemv_rows_mfma(e.s, p, &rn), gated ? "gemv_rows[norm+up+act]" : "gemv_rows[norm+linear]", k::matmul_algorithmic_bytes(p));
    return true;
}

// mode: 0 none, 1 copy, 2 add (ShortcutMode, encodable_block/normalization.rs:22-27)
// `consumer`: the quantised linear that reads `output` next as a prefill GEMM: the kernel then files the group row sums of the rows it writes
k::NormParams norm_params(Enc& e, const DNorm& N, const uint16_t* input, uint16_t* output, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim, const DLinear* consumer = nullptr) {
    k::NormParams p{};
    p.input = input, p.scales = N.scales, p.biases = N.biases, p.output = output, p.shortcut = mode ? shortcut : nullptr;
    p.io_dt = UZU_BF16, p.affine_dt = UZU_F32;
    p.batch_size = rows, p.element_count = dim;
    p.epsilon = N.eps, p.scale_offset = N.offset, p.post_layer_scalar = N.scalar_mode ? N.scalar : 1.0f;
    p.scale_residual_sum = N.scalar_mode == 1, p.scale_output = N.scalar_mode == 2;
    p.subtract_mean = N.subtract_mean, p.full_layer = N.full_layer;
    p.copy_to_shortcut = mode != 0, p.residual_add = mode == 2;
    uzu_hip_model* m = e.m;
    if (consumer && consumer->coef && !consumer->in_signs && !consumer->lora_rank && output == m->normed && rows >= 128 && consumer->k == dim && !k::exact_mode() &&
        k::normalization_rowsum_supported(dim, consumer->group) && (size_t)(dim / consumer->group) * ((rows + 3) & ~3u) <= m->rowsum_floats)
        p.rowsum_out = m->rowsum, p.rowsum_group = consumer->group;
    return p;
}
// book-keeping of the filed row sums, at the point where the normalisation `p` is ISSUED (its parameters may have been drawn up earlier: PostNorm)
void norm_issued(uzu_hip_model* m, const k::NormParams& p) {
    if (p.output != m->normed) return;
    m->rs_rows = 0; // whatever was filed for the old rows is stale
    if (p.rowsum_out) m->rs_rows = p.batch_size, m->rs_k = p.element_count, m->rs_group = p.rowsum_group;
}
void norm(Enc& e, const DNorm& N, const uint16_t* input, uint16_t* output, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim, const DLinear* consumer = nullptr) {
    const k::NormParams p = norm_params(e, N, input, output, shortcut, mode, rows, dim, consumer);
    norm_issued(e.m, p);
    RUN("normalization", 0, k::normalization(e.s, p));
}

// whole logit rows for stochastic sampling on a vocab shard (allocated on first use: set_sampling / a stochastic tree pass under TP)
uzu_status ensure_tp_gather(uzu_hip_model* m, uint32_t rows) {
    if (!m->tp || m->tp_gather_rows >= rows) return UZU_OK;
    void* p = nullptr;
    if (m->tp_gather_f32) dev_free(m, m->tp_gather_f32), dev_free(m, m->tp_gather_bf16);
    m->tp_gather_f32 = nullptr, m->tp_gather_bf16 = nullptr, m->tp_gather_rows = 0;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * m->d.vocab_size * 4, &p));
    m->tp_gather_f32 = (float*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * m->d.vocab_size * 2, &p));
    m->tp_gather_bf16 = (uint16_t*)p;
    m->tp_gather_rows = rows;
    return UZU_OK;
}

uzu_status ensure_partials(uzu_hip_model* m, uint32_t rows, uint32_t head_dim) {
    if (rows <= m->partial_rows) return UZU_OK;
    // regrow: the old blocks stay allocated until the model is destroyed -- captured decode graphs (this state's and every
    // other uzu_hip_state's graph_two on the unfused path) carry their addresses in attention_two_pass1/2 nodes, and nothing
    // re-captures them on a regrow.  They are small (decode rows) next to the prefill-sized blocks that replace them.
    void* p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * 32 * head_dim * 4, &p));
    m->partials = (float*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * 32 * 4, &p));
    m->sums = (float*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * 32 * 4, &p));
    m->maxs = (float*)p;
    m->partial_rows = rows;
    return UZU_OK;
}

// The sequences of one forward pass.  n == 0: the bound state, `count` rows (the plain single-sequence pass).  n >= 1:
// `count` rows per sequence, sequence q in rows [q * count, (q + 1) * count) of every activation buffer: the linear
// layers, norms and element-wise kernels run once over all n * count rows (one GEMM with M = n * count: the weights are
// streamed once for all sequences), attention / KV append / DeltaNet run per sequence on its own state.  The reference
// has no cross-sequence batching: per sequence the arithmetic is that of the single-sequence pass.
struct Seqs {
    uzu_hip_state** st = nullptr;
    uint32_t n = 0;
    uint32_t count = 0;
    uint32_t rows() const { return (n ? n : 1) * count; }
};

void attention_core(Enc& e, DLayer& L, uint32_t batch, size_t row0);

// `first_done`: the layer's first projection (qkv; the DeltaNet in-projection) has been run with its Normalization prologue already (linear_normed)
void attention_mixer(Enc& e, DLayer& L, const uint16_t* hidden, uint16_t* out, const Seqs& q, PostNorm* post = nullptr, bool first_done = false) {
    uzu_hip_model* m = e.m;
    // KV sharing: the packed projection yields queries only, key / value norms are dropped (mixer/attention/mod.rs:80-95,135-137)
    const uint32_t hd = L.d.head_dim, nq = L.d.num_heads, nkv = L.d.is_kv_sharing ? 0u : L.d.num_groups, total_heads = nq + 2 * nkv;
    const uint32_t rows = q.rows();
    if (L.d.has_gate) linear(e, L.gate, hidden, m->gate, rows);
    if (!first_done) linear(e, L.qkv, hidden, m->qkv, rows);
    if (L.qn.present)
        RUN("qkv_norm", 0, k::qkv_norm(e.s, m->qkv, UZU_BF16, L.qn.scales, rows, total_heads, hd, L.qn.eps, L.qn.offset, 0, nq, L.qn.full_layer));
    if (L.kn.present && nkv)
        RUN("qkv_norm", 0, k::qkv_norm(e.s, m->qkv, UZU_BF16, L.kn.scales, rows, total_heads, hd, L.kn.eps, L.kn.offset, nq, nkv, L.kn.full_layer));
    if (L.d.normalize_values && nkv) // AttentionConfig::value_norm_config (config/token_mixer/attention.rs:32-42): eps 1e-6, FullLayer, no scales
        RUN("qkv_norm", 0, k::qkv_norm(e.s, m->qkv, UZU_BF16, nullptr, rows, total_heads, hd, 1e-6f, 0.0f, nq + nkv, nkv, 1));
    if (q.n == 0) {
        attention_core(e, L, q.count, 0);
    } else {
        for (uint32_t i = 0; i < q.n; ++i) {
            bind_state(m, q.st[i]);
            attention_core(e, L, q.count, (size_t)i * q.count);
        }
    }
    if (L.d.has_gate) RUN("sigmoid_gate", 0, k::sigmoid_gate(e.s, m->gate, m->attn_out, UZU_BF16, rows * nq * hd));
    linear(e, L.out, m->attn_out, out, rows, true, post);
}

// AttentionPrepare + attention of `batch` rows of the bound sequence, which start at row `row0` of qkv / queries / attn_out
void attention_core(Enc& e, DLayer& L, uint32_t batch, size_t row0) {
    uzu_hip_model* m = e.m;
    const bool has_kv = !L.d.is_kv_sharing; // prepare_queries (mode.rs:234-259): no KV rows are written; the source layer wrote this pass's already
    const uint32_t hd = L.d.head_dim, nq = L.d.num_heads, nkv = L.d.num_groups, total_heads = nq + (has_kv ? 2 * nkv : 0);
    const uint16_t* qkv = m->qkv + row0 * total_heads * hd;
    uint16_t* queries = m->queries + row0 * nq * hd;
    uint16_t* attn_out = m->attn_out + row0 * nq * hd;
    const uint32_t rope_dim = L.d.use_rope ? L.rope_dim : 0;
    // Ring state (causal sliding window; state.rs:16-55, mode.rs:66-78): the new rows go to the suffix region behind the ring
    // (kv_token_offset = physical_prefix_length = window), the attention sees window + batch rows with ring parameters derived on the
    // device from the accepted-token count, and the rows enter the ring afterwards (encode_accept, state.rs:200-219).
    const uint32_t W = L.d.sliding_window_size;
    const uint32_t* trie = m->tree.active ? m->tree.d_trie : nullptr; // a speculated tree: RoPE positions = context + height, trie mask (mode.rs:178-192)
    RUN("attention_prepare", 0, k::attention_prepare(e.s, qkv, queries, L.keys, L.values, L.rope_cos, L.rope_sin, nq, nkv, hd, rope_dim, W, batch, has_kv ? 1u : 0u,
                               m->d_ctx_len, W ? 1u : 0u, trie));
    k::AttentionParams a{};
    a.queries = queries, a.keys = L.keys, a.values = L.values;
    a.dt = UZU_BF16, a.head_dim = hd, a.gqa_factor = nq / nkv;
    a.sequence_length = batch; // + *d_ctx_len on the device
    a.k_head_stride = hd, a.k_seq_stride = nkv * hd, a.v_head_stride = hd, a.v_seq_stride = nkv * hd;
    a.scale = L.d.attention_scale != 0.0f ? L.d.attention_scale : 1.0f / sqrtf((float)hd);
    a.num_heads = nq, a.suffix_length = batch, a.is_causal = 1;
    a.dyn = m->d_ctx_len;
    a.trie = trie;
    if (W) a.ring_window = W, a.is_kv_cache_ring = 1, a.is_sliding_window = 1, a.sliding_window_size = W;
    if (L.sinks) a.sinks = L.sinks;
    const uint32_t physical_prefix = W ? W : m->context_length; // AttentionStateType::physical_prefix_length (state.rs:26-37)
    const size_t kv_bytes = (size_t)2 * (physical_prefix + batch) * nkv * hd * 2; // K and V rows read once
    const bool two_pass = (m->regime_override >= 0 && !W) ? m->regime_override == 1 : physical_prefix + batch > 1024; // core/mod.rs:89-92
    if (k::attention_prefill_mfma_supported(a)) { // prefill chunk: flash-attention tiles on the matrix cores, any context length
        RUN("attention_prefill_mfma", kv_bytes, k::attention_prefill_mfma(e.s, a, attn_out));
    } else if (two_pass) { // core/mod.rs:89-92
        RUN("attention_two_pass1", kv_bytes, k::attention_two_pass1(e.s, a, m->partials, m->sums, m->maxs));
        RUN("attention_two_pass2", 0, k::attention_two_pass2(e.s, m->partials, m->sums, m->maxs, attn_out, UZU_BF16, hd, nq, batch));
    } else {
        RUN("attention_single_pass", kv_bytes, k::attention_single_pass(e.s, a, attn_out));
    }
    if (W && !trie && has_kv && L.last_reader == (uint32_t)(&L - m->layers.data()))
        RUN("kv_ring_insert", 0, k::kv_ring_insert(e.s, L.keys, L.values, UZU_BF16, m->d_ctx_len, batch, W, nkv * hd));
}

void delta_net_core(Enc& e, DLayer& L, uint32_t batch, size_t row0);

// DeltaNet::encode_tree_verify (delta_net.rs:334-437): conv tree scan + tree prep (one launch), the tree-verify composite, norm-gate;
// the layer's DeltaNetSuffixStatus::Tree stays in m->tree.layers[layer] for uzu_hip_model_accept
void delta_net_tree_core(Enc& e, DLayer& L, uint32_t layer, uint32_t n) {
    uzu_hip_model* m = e.m;
    const uint32_t Hv = L.d.dn_num_heads, Hk = L.d.dn_num_groups, Dk = L.d.dn_head_dim, Dv = L.d.dn_value_head_dim;
    const uint32_t key_dim = Hk * Dk, value_dim = Hv * Dv, conv_dim = 2 * key_dim + value_dim;
    const uint32_t total_proj_dim = conv_dim + value_dim + 2 * Hv, ks = L.d.dn_kernel_size;
    uzu_hip_model::TreeLayer& T = m->tree.layers[layer];
    RUN("dn_tree_prep", 0, k::delta_net_tree_prep(e.s, m->in_proj, L.conv_w, L.conv_b, L.conv_state, m->tree.d_parents, nullptr, T.conv_states, L.a_log, L.dt_bias, m->tree.q,
                                                   T.k, T.v, T.beta, T.log_decay, n, ks, Hk, Hv, Dk, Dv, true, true));
    RUN("dn_tree_verify", (size_t)Hv * Dv * Dk * 4, k::delta_net_tree_verify(e.s, m->tree.q, T.k, T.v, m->tree.d_trie, T.log_decay, T.beta, L.ssm_state, m->delta_out, n, Hk, Hv, Dk, Dv));
    // the norm-gate reads z from the rows' pass-through section: ConvTreeScan copies those channels unchanged, so the in-proj rows serve
    RUN("delta_net_norm_gate", 0, k::delta_net_norm_gate(e.s, m->delta_out, m->in_proj, L.dn_norm, Hv, Dv, value_dim, conv_dim, total_proj_dim, L.d.dn_norm_epsilon, n));
}

void delta_net_mixer(Enc& e, DLayer& L, const uint16_t* hidden, uint16_t* out, const Seqs& q, PostNorm* post = nullptr, bool first_done = false) {
    uzu_hip_model* m = e.m;
    const uint32_t rows = q.rows();
    if (!first_done) linear(e, L.in_proj, hidden, m->in_proj, rows);
    if (m->tree.active) { // !batch_dim.full_accept() (delta_net.rs:496-502)
        delta_net_tree_core(e, L, (uint32_t)(&L - m->layers.data()), q.count);
    } else if (q.n == 0) {
        delta_net_core(e, L, q.count, 0);
    } else {
        for (uint32_t i = 0; i < q.n; ++i) {
            bind_state(m, q.st[i]);
            delta_net_core(e, L, q.count, (size_t)i * q.count);
        }
    }
    linear(e, L.out_proj, m->delta_out, out, rows, true, post);
}

// conv + delta rule + norm-gate over `batch` rows of the bound sequence, starting at row `row0` of in_proj / delta_out
void delta_net_core(Enc& e, DLayer& L, uint32_t batch, size_t row0) {
    uzu_hip_model* m = e.m;
    const uint32_t Hv = L.d.dn_num_heads, Hk = L.d.dn_num_groups, Dk = L.d.dn_head_dim, Dv = L.d.dn_value_head_dim;
    const uint32_t key_dim = Hk * Dk, value_dim = Hv * Dv, conv_dim = 2 * key_dim + value_dim;
    const uint32_t total_proj_dim = conv_dim + value_dim + 2 * Hv, ks = L.d.dn_kernel_size;
    uint16_t* in_proj = m->in_proj + row0 * total_proj_dim;
    uint16_t* delta_out = m->delta_out + row0 * value_dim;
    if (batch == 1) {
        RUN("delta_net_conv_update", 0, k::delta_net_conv_update(e.s, L.conv_w, L.conv_b, in_proj, L.conv_state, ks, conv_dim, ks - 1));
        RUN("delta_net_update", (size_t)2 * Hv * Dv * Dk * 4, k::delta_net_update(e.s, in_proj, L.a_log, L.dt_bias, L.dn_norm, L.ssm_state, delta_out, Hv, Hk, Dk, Dv, key_dim, value_dim,
                                  L.d.dn_norm_epsilon));
    } else {
        if (ks <= 8 && k::delta_net_conv_fused_workspace_floats(batch, ks, conv_dim) <= (size_t)(m->chunk + 8) * total_proj_dim) {
            RUN("delta_net_conv_fused", 0, k::delta_net_conv_fused(e.s, in_proj, L.conv_w, L.conv_b, L.conv_state, m->padded, batch, ks, conv_dim, total_proj_dim));
        } else {
            RUN("conv1d_pack", 0, k::conv1d_pack(e.s, L.conv_state, in_proj, m->padded, ks - 1, total_proj_dim, batch, conv_dim));
            RUN("delta_net_conv_scan", 0, k::delta_net_conv_scan(e.s, m->padded, L.conv_w, L.conv_b, in_proj, L.conv_state, batch, ks, total_proj_dim, ks - 1, conv_dim,
                                         total_proj_dim));
        }
        RUN("delta_net_prefill_prep", 0, k::delta_net_prefill_prep(e.s, in_proj, L.a_log, L.dt_bias, m->qn, m->kn, m->beta, m->decay, Hv, Hk, Dk, key_dim, value_dim, batch));
        if (m->dn_ws && k::delta_net_prefill_chunked_supported(Hv, Hk, Dk, Dv, batch))
            RUN("delta_net_prefill_chunked", 0, k::delta_net_prefill_chunked(e.s, m->qn, m->kn, m->beta, m->decay, in_proj, L.ssm_state, delta_out, m->dn_ws, Hv, Hk, Dv,
                                                                          key_dim, value_dim, batch));
        else
            RUN("delta_net_prefill", 0, k::delta_net_prefill(e.s, m->qn, m->kn, m->beta, m->decay, in_proj, L.ssm_state, delta_out, Hv, Hk, Dk, Dv, key_dim, value_dim, batch));
        RUN("delta_net_norm_gate", 0, k::delta_net_norm_gate(e.s, delta_out, in_proj, L.dn_norm, Hv, Dv, value_dim, conv_dim, total_proj_dim, L.d.dn_norm_epsilon, batch));
    }
}

__global__ void commit_kernel(uint32_t* ctx_len, uint32_t* tokens, const uint32_t* out_token, uint32_t* sampled, uint32_t count, uint32_t has_token) {
    const uint32_t len = *ctx_len;
    if (has_token) {
        const uint32_t t = *out_token;
        sampled[len + count - 1] = t;
        tokens[0] = t;
    }
    *ctx_len = len + count;
}

// One forward pass.  `seqs` null: `count` tokens of the bound state, already in m->d_tokens.  `seqs` non-null: `count`
// tokens for each of the `nseq` states, token ids already in m->batch_tokens (row q * count + i); see struct Seqs.
// `sample` => output norm + readout + argmax on the last row (of every sequence).
uzu_status encode_forward(uzu_hip_model* m, hipStream_t s, uint32_t count, bool sample, uzu_hip_state** seqs = nullptr, uint32_t nseq = 0) {
    Enc e{m, s};
    e.prof = (std::vector<ProfEntry>*)m->prof_sink;
    m->launches = 0;
    const uint32_t d = m->d.model_dim;
    Seqs q;
    q.st = seqs, q.n = seqs ? nseq : 0, q.count = count;
    const uint32_t rows = q.rows();
    const uint32_t* token_ids = seqs ? m->batch_tokens : m->d_tokens;
    uint16_t* hidden = m->hidden;
    if (m->embedding.method == UZU_QUANT_NONE)
        RUN("full_precision_embedding_lookup", 0, k::full_precision_embedding_lookup(s, token_ids, m->embedding.w, hidden, UZU_BF16, rows, m->d.vocab_size, d, m->d.input_scale));
    else
        RUN("quantized_embedding_lookup", 0, k::quantized_embedding_lookup(s, token_ids, (const uint8_t*)m->embedding.w, m->embedding.scales, m->embedding.zp, m->embedding.biases,
                                            hidden, UZU_BF16, rows, m->d.vocab_size, d, m->d.input_scale, m->embedding.group, m->embedding.bits,
                                            m->embedding.method));
    // EmbeddingTable with output Hadamard factors (embedding_table.rs:34-125; quant_embedding.metal, use_hadamard): OutputRht of the rows
    if (m->embedding.out_signs)
        RUN("activation_transform", 0, k::activation_transform(s, nullptr, hidden, nullptr, nullptr, nullptr, m->embedding.out_signs, UZU_BF16, rows, d,
                                                                UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
    // Decoder::encode (decoder.rs:149-154): the embedding norm, no shortcut
    if (m->embedding_norm.present) {
        norm(e, m->embedding_norm, hidden, m->normed, nullptr, 0, rows, d);
        RUN("tensor_copy", 0, k::tensor_copy(s, m->normed, hidden, UZU_BF16, rows * d));
    }
    // PerLayerEmbedding::encode (per_layer_embedding.rs:108-147): per_layer_inputs [rows][layers][ple_dim] = token table row * (ple_embed_scale *
    // input_scale) + projection_norm(model_projection(embedded rows)) [ScaleOutput(input_scale); epsilon / model_projection_scale^2 at load]
    if (m->d.has_ple) {
        const uint32_t total = m->d.num_layers * m->d.ple_dim;
        const DLinear& T = m->ple_token_embedding;
        const float fused_token_scale = m->d.ple_embed_scale * m->d.ple_input_scale; // per_layer_embedding.rs:103
        if (T.method == UZU_QUANT_NONE)
            RUN("full_precision_embedding_lookup", 0, k::full_precision_embedding_lookup(s, token_ids, T.w, m->ple_token, UZU_BF16, rows, m->d.ple_vocab_size, total, fused_token_scale));
        else
            RUN("quantized_embedding_lookup", 0, k::quantized_embedding_lookup(s, token_ids, (const uint8_t*)T.w, T.scales, T.zp, T.biases, m->ple_token, UZU_BF16, rows,
                                                m->d.ple_vocab_size, total, fused_token_scale, T.group, T.bits, T.method));
        linear(e, m->ple_model_projection, hidden, m->ple_projected, rows);
        norm(e, m->ple_projection_norm, m->ple_projected, m->ple_inputs, nullptr, 0, rows * m->d.num_layers, m->d.ple_dim);
        RUN("tensor_add_scale", 0, k::tensor_add_scale(s, m->ple_token, m->ple_inputs, m->ple_inputs, UZU_BF16, rows * total, rows * total, 1.0f));
    }
    // Prefill-sized passes: a row-parallel projection whose rows go straight into the next Normalization hands that normalisation to its GEMM
    // (PostNorm: the split-K reduction, the epilogue and the normalisation of a row are one launch).  `hidden_normed`: the pre-mixer normalisation of
    // the layer about to run has been done that way by the previous layer's down projection.
    bool hidden_normed = false;
    // the residual rows: in place in m->shortcut, except where a Normalization rides in a few-rows linear's prologue (linear_normed: 2 .. 16 rows, one
    // sequence) -- those read sc_cur and write the other buffer of the pair
    uint16_t* sc_cur = m->shortcut;
    auto sc_other = [&]() { return sc_cur == m->shortcut ? m->shortcut_b : m->shortcut; };
    const bool few_rows = !seqs && m->shortcut_b != nullptr;
    // Transformer::prefill_cache_layer_count (transformer.rs:186-199,239-243): a pass that produces no output (a prefill chunk that is not the
    // prompt's last) only has to fill the caches -- it stops behind the last layer that owns a state; trailing KV-sharing layers write nothing
    uint32_t layer_count = m->d.num_layers;
    if (!sample && !m->tree.active && !m->taps)
        while (layer_count > 1 && m->layers[layer_count - 1].d.mixer_kind == UZU_MIXER_ATTENTION && m->layers[layer_count - 1].d.is_kv_sharing) --layer_count;
    for (uint32_t l = 0; l < layer_count; ++l) {
        DLayer& L = m->layers[l];
        const uint16_t* h = hidden;
        bool first_done = false;
        if (L.pre_mixer.present) {
            if (!hidden_normed) {
                const bool att = L.d.mixer_kind == UZU_MIXER_ATTENTION;
                // (a gated attention layer's gate projection reads the same normalised rows: the prologue files them in m->normed)
                if (few_rows && linear_normed(e, L.pre_mixer, l > 0 ? 2 : 1, att ? L.qkv : L.in_proj, hidden, sc_cur, sc_other(), att && L.d.has_gate ? m->normed : nullptr,
                                              att ? m->qkv : m->in_proj, rows, false, 0)) {
                    first_done = true, sc_cur = sc_other();
                } else {
                    norm(e, L.pre_mixer, hidden, m->normed, sc_cur, l > 0 ? 2 : 1, rows, d, att ? &L.qkv : &L.in_proj);
                }
            }
            h = m->normed;
        } else {
            RUN("tensor_copy", 0, k::tensor_copy(s, hidden, sc_cur, UZU_BF16, rows * d));
        }
        hidden_normed = false;
        PostNorm mlp_norm; // the pre-MLP normalisation, offered to the mixer's out projection when nothing sits between them
        const bool offer_mlp = !L.post_mixer.present && rows >= 128 && !L.pre_mlp.scalar_mode;
        if (offer_mlp) mlp_norm.p = norm_params(e, L.pre_mlp, m->mixed, m->normed, sc_cur, 2, rows, d, &L.up);
        if (L.d.mixer_kind == UZU_MIXER_ATTENTION)
            attention_mixer(e, L, h, m->mixed, q, offer_mlp ? &mlp_norm : nullptr, first_done);
        else
            delta_net_mixer(e, L, h, m->mixed, q, offer_mlp ? &mlp_norm : nullptr, first_done);
        const uint16_t* mixed = m->mixed;
        if (L.post_mixer.present) {
            norm(e, L.post_mixer, m->mixed, m->normed, nullptr, 0, rows, d);
            RUN("tensor_copy", 0, k::tensor_copy(s, m->normed, m->mixed, UZU_BF16, rows * d));
        }
        bool mlp_done = false; // pre-MLP Normalization + up | gate + GatedActMul as ONE few-rows launch
        if (offer_mlp) {
            norm_issued(m, mlp_norm.p);
            if (!mlp_norm.done) RUN("normalization", 0, k::normalization(e.s, mlp_norm.p));
        } else if (few_rows && !L.post_mixer.present && linear_normed(e, L.pre_mlp, 2, L.up, mixed, sc_cur, sc_other(), nullptr, m->gated, rows, true, L.d.activation)) {
            mlp_done = true, sc_cur = sc_other();
        } else {
            norm(e, L.pre_mlp, mixed, m->normed, sc_cur, 2, rows, d, &L.up);
        }
        if (!mlp_done && !linear_gated(e, L.up, m->normed, m->gated, rows, L.d.activation)) { // prefill-sized rows: GatedActMul in the GEMM's epilogue
            linear(e, L.up, m->normed, m->up, rows);
            RUN("gated_act_mul", 0, k::gated_act_mul(s, m->up, nullptr, m->gated, UZU_BF16, L.d.hidden_dim, rows, 0, 0, L.d.activation, 1));
        }
        // the next layer's pre-mixer normalisation rides on this layer's down projection (not past the last layer: the output norm takes one row)
        PostNorm next_norm;
        const bool offer_next = rows >= 128 && !L.post_mlp.present && !L.d.has_ple && l + 1 < layer_count && m->layers[l + 1].pre_mixer.present;
        if (offer_next) {
            const DLayer& Nx = m->layers[l + 1];
            next_norm.p = norm_params(e, Nx.pre_mixer, hidden, m->normed, sc_cur, 2, rows, d, Nx.d.mixer_kind == UZU_MIXER_ATTENTION ? &Nx.qkv : &Nx.in_proj);
        }
        linear(e, L.down, m->gated, hidden, rows, true, offer_next ? &next_norm : nullptr);
        if (offer_next && next_norm.done) {
            norm_issued(m, next_norm.p);
            hidden_normed = true;
        }
        if (L.post_mlp.present) {
            norm(e, L.post_mlp, hidden, m->normed, nullptr, 0, rows, d);
            RUN("tensor_copy", 0, k::tensor_copy(s, m->normed, hidden, UZU_BF16, rows * d));
        }
        if (L.d.has_ple) {
            // PerLayerEmbeddingProjection::encode (per_layer_embedding.rs:217-270): shortcut += hidden; gate(shortcut) -> act(gate) * this layer's slice
            // of per_layer_inputs -> projection -> norm; shortcut = (shortcut + normed) * post_layer_scalar; hidden = 0 (transformer_layer.rs:231)
            const uint32_t length = rows * d, pd = L.d.ple_dim;
            RUN("tensor_add_bias", 0, k::tensor_add_bias(s, nullptr, hidden, sc_cur, UZU_BF16, UZU_BF16, length, length));
            linear(e, L.ple_gate, sc_cur, m->ple_gate_out, rows);
            RUN("gated_act_mul", 0, k::gated_act_mul(s, m->ple_gate_out, m->ple_inputs, m->ple_activated, UZU_BF16, pd, rows, l * pd, m->d.num_layers * pd, L.d.ple_activation, 0));
            linear(e, L.ple_projection, m->ple_activated, m->mixed, rows);
            norm(e, L.ple_norm, m->mixed, m->normed, nullptr, 0, rows, d);
            RUN("tensor_add_scale", 0, k::tensor_add_scale(s, nullptr, m->normed, sc_cur, UZU_BF16, length, length, L.d.has_post_layer_scalar ? L.d.post_layer_scalar : 1.0f));
            HIPCHK(hipMemsetAsync(hidden, 0, (size_t)length * 2, s));
        }
        // a ring whose rows later layers of this pass still had to read takes the pass's suffix rows now (DLayer::last_reader; at the latest
        // behind the last layer this pass runs)
        if (!m->tree.active)
            for (uint32_t o = 0; o <= l; ++o) {
                DLayer& Lo = m->layers[o];
                if (Lo.d.mixer_kind != UZU_MIXER_ATTENTION || Lo.d.is_kv_sharing || !Lo.d.sliding_window_size || Lo.last_reader == o) continue; // (== o: inserted by its own launch sequence)
                if (!(Lo.last_reader == l || (l + 1 == layer_count && Lo.last_reader > l))) continue;
                for (uint32_t i = 0; i < (q.n ? q.n : 1u); ++i) {
                    if (q.n) bind_state(m, q.st[i]);
                    RUN("kv_ring_insert", 0, k::kv_ring_insert(s, Lo.keys, Lo.values, UZU_BF16, m->d_ctx_len, count, Lo.d.sliding_window_size, Lo.d.num_groups * Lo.d.head_dim));
                }
            }
        if (m->taps && !seqs) RUN("tensor_copy", 0, k::tensor_copy(s, hidden, m->taps + ((size_t)l * m->chunk) * d, UZU_BF16, count * d));
    }
    m->tap_rows = count;
    if (m->tree.active) {
        // a tree pass: output norm, read-out and greedy sampling of EVERY node (output_range 0..size, stream.rs:618-628), no commit
        norm(e, m->output_norm, hidden, m->tree.normed, sc_cur, 2, count, d);
        DLinear ro = m->d.tied_embeddings ? m->embedding : m->output_embedding;
        ro.in_signs = m->d.tied_embeddings ? m->embedding.out_signs : m->output_embedding.in_signs;
        ro.out_signs = nullptr;
        linear(e, ro, m->tree.normed, m->tree.logits, count);
        if (m->d.logit_scale != 1.0f || m->d.logit_soft_cap != 0.0f)
            RUN("logit_transform", 0, k::logit_transform(s, m->tree.logits, UZU_BF16, ro.n * count, m->d.logit_scale, m->d.logit_soft_cap, m->d.logit_soft_cap != 0.0f));
        if (m->sampling.on) {
            // every node draws with ITS seed: the trie's token_seeds when the caller passed them (stream.rs:694), else the seed of its position,
            // PRng::derive(context + height) (dflash_tfm.rs:267,304)
            if (!m->tree.host_seeds) RUN("derive_tree_seeds", 0, k::derive_tree_seeds(s, m->sampling.seed, m->d_ctx_len, m->tree.d_trie, count, m->tree.d_seeds));
            k::UnifiedSamplingParams sp = m->sampling.p;
            sp.logits = m->tree.logits, sp.dt = UZU_BF16, sp.output = m->tree.d_sampled, sp.seeds = m->tree.d_seeds, sp.vocab_size = ro.n, sp.batch_size = count;
            if (m->tp) { // the whole rows on every rank: same seeds, same distribution => the same token everywhere
                RUN("tp_gather_logits", (size_t)m->d.vocab_size * 4 * count, tp::gather_logits(m->tp, s, m->tree.logits, ro.n, m->vocab_offset, m->d.vocab_size, count, m->tp_gather_f32, m->tp_gather_bf16));
                sp.logits = m->tp_gather_bf16, sp.vocab_size = m->d.vocab_size;
            }
            RUN("unified_sampling", (size_t)sp.vocab_size * 2 * count, k::unified_sampling(s, sp, m->tree.sampling_scratch));
        } else {
            RUN("argmax", (size_t)ro.n * 2 * count, k::argmax(s, m->tree.logits, UZU_BF16, m->tree.d_sampled, ro.n, count, m->tree.argmax_scratch));
            if (m->tp) { // vocab-sharded read-out: one (logit, global index) key per node, reduced with max
                RUN("tp_keys", 0, tp::keys_from_tokens(s, m->tree.logits, ro.n, m->tree.d_sampled, m->vocab_offset, m->tp_key, count));
                RUN("all_reduce", 8 * count, tp::all_reduce_max_u64(m->tp, s, m->tp_key, count));
                RUN("tp_tokens", 0, tp::tokens_from_keys(s, m->tp_key, m->tree.d_sampled, count));
            }
        }
        if (e.st != UZU_OK) return e.st;
        hipError_t terr = hipGetLastError();
        if (terr != hipSuccess) {
            set_error("engine: tree pass launch failed: %s", hipGetErrorString(terr));
            return UZU_ERR_HIP;
        }
        return UZU_OK;
    }
    for (uint32_t i = 0; i < (seqs ? nseq : 1u); ++i) { // per sequence: sample from its last row, then commit
        if (seqs) bind_state(m, seqs[i]);
        if (sample) {
            const size_t last = ((size_t)i * count + count - 1) * d;
            norm(e, m->output_norm, hidden + last, m->last_normed, sc_cur + last, 2, 1, d);
            // Embedding::encode_readout (embedding.rs:374-456): the read-out's private InputRht -- a tied table's output signs
            // (embedding.rs:167-173) or the untied output embedding's input signs (embedding.rs:255-274) -- then the plain matmul
            DLinear ro = m->d.tied_embeddings ? m->embedding : m->output_embedding;
            ro.in_signs = m->d.tied_embeddings ? m->embedding.out_signs : m->output_embedding.in_signs;
            ro.out_signs = nullptr;
            linear(e, ro, m->last_normed, m->logits, 1);
            if (m->d.logit_scale != 1.0f || m->d.logit_soft_cap != 0.0f)
                RUN("logit_transform", 0, k::logit_transform(s, m->logits, UZU_BF16, ro.n, m->d.logit_scale, m->d.logit_soft_cap, m->d.logit_soft_cap != 0.0f));
            if (m->sampling.on) { // stream.rs:248-258: seed = PRng::derive(position of the sampled row), then UnifiedSampling
                RUN("derive_seed", 0, k::derive_seed(s, m->sampling.seed, m->d_ctx_len, count - 1, m->d_seed));
                k::UnifiedSamplingParams sp = m->sampling.p;
                sp.logits = m->logits, sp.dt = UZU_BF16, sp.output = m->d_out_token, sp.seeds = m->d_seed, sp.vocab_size = ro.n, sp.batch_size = 1;
                if (m->tp) { // the whole row on every rank (tp::gather_logits): same seed, same distribution => the same token everywhere
                    RUN("tp_gather_logits", (size_t)m->d.vocab_size * 4, tp::gather_logits(m->tp, s, m->logits, ro.n, m->vocab_offset, m->d.vocab_size, 1, m->tp_gather_f32, m->tp_gather_bf16));
                    sp.logits = m->tp_gather_bf16, sp.vocab_size = m->d.vocab_size;
                }
                RUN("unified_sampling", (size_t)sp.vocab_size * 2, k::unified_sampling(s, sp, m->sampling_scratch));
            } else {
                RUN("argmax", (size_t)ro.n * 2, k::argmax(s, m->logits, UZU_BF16, m->d_out_token, ro.n, 1, m->argmax_scratch));
            }
            if (m->tp && !m->sampling.on) { // vocab-sharded read-out: every rank contributes (logit, global index) of its local winner
                RUN("tp_key", 0, tp::key_from_token(s, m->logits, m->d_out_token, m->vocab_offset, m->tp_key));
                RUN("all_reduce", 8, tp::all_reduce_max_u64(m->tp, s, m->tp_key, 1));
                RUN("tp_token", 0, tp::token_from_key(s, m->tp_key, m->d_out_token));
            }
        }
        // an earlier launch failed: leave the device-side context length / next token untouched so that they keep agreeing
        // with the host mirror (m->context_length is only advanced by the callers on success)
        if (e.st != UZU_OK) return e.st;
        hipLaunchKernelGGL(commit_kernel, dim3(1), dim3(1), 0, s, m->d_ctx_len, m->d_tokens, m->d_out_token, m->d_sampled, count, sample ? 1u : 0u);
        ++m->launches;
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("engine: forward launch failed: %s", hipGetErrorString(err));
        return UZU_ERR_HIP;
    }
    return UZU_OK;
}

// ---------------------------------------------------------------------------------- fused decode step
k::DecGemvParams dec_gemv_base(const DLinear& L, const uint16_t* x, uint16_t* out) {
    k::DecGemvParams p{};
    p.w[0] = (const uint8_t*)L.w, p.scales[0] = (const uint16_t*)L.scales, p.biases[0] = (const uint16_t*)L.biases, p.zp[0] = L.zp;
    p.out_bias[0] = L.out_signs ? nullptr : (const uint16_t*)L.out_biases, p.out[0] = out, p.n[0] = L.n; // bias_after_rht: with the OutputRht, later
    p.k = L.k, p.bits = L.bits, p.group_size = L.group;
    p.b_kind = L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS
             : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.x = x;
    return p;
}
void dec_add_second(k::DecGemvParams& p, const DLinear& L, uint16_t* out) {
    p.w[1] = (const uint8_t*)L.w, p.scales[1] = (const uint16_t*)L.scales, p.biases[1] = (const uint16_t*)L.biases, p.zp[1] = L.zp;
    p.out_bias[1] = L.out_signs ? nullptr : (const uint16_t*)L.out_biases, p.out[1] = out, p.n[1] = L.n;
}
// mode: 1 copy, 2 add
void dec_add_norm(k::DecGemvParams& p, const DNorm& N, int mode, const uint16_t* sc_in, uint16_t* sc_out) {
    p.norm_scales = N.scales;
    p.norm_plain = N.scales == nullptr;
    p.norm_eps = N.eps, p.norm_offset = N.offset, p.norm_full_layer = N.full_layer;
    p.residual_add = mode == 2;
    p.shortcut_in = mode == 2 ? sc_in : nullptr;
    p.shortcut_out = sc_out;
}
size_t dec_gemv_bytes(const k::DecGemvParams& p) {
    size_t b = 0;
    for (int i = 0; i < 2; ++i) {
        if (!p.n[i]) continue;
        const size_t groups = (p.k + p.group_size - 1) / p.group_size;
        b += (size_t)p.n[i] * p.k * p.bits / 8 + (size_t)p.n[i] * groups * 2;
        if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) b += (size_t)p.n[i] * groups * 2;
        if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) b += (size_t)p.n[i] * (p.bits == 4 ? (groups + 1) / 2 : groups);
        b += (size_t)p.n[i] * 2;
    }
    return b + (size_t)p.k * 2;
}
void dec_gemv(Enc& e, const k::DecGemvParams& p, const char* name, uint32_t* grid_out = nullptr) {
    e.begin();
    e.run(k::gemv_dec(e.s, p, e.m->ctx->num_cus, grid_out), name, dec_gemv_bytes(p));
}

// out-proj / down-proj of the fused decode step; under tensor parallelism: f32 partials -> all-reduce -> bf16
void dec_gemv_row_parallel(Enc& e, k::DecGemvParams p, const char* name) {
    uzu_hip_model* m = e.m;
    if (!m->tp) return dec_gemv(e, p, name);
    uint16_t* out = p.out[0];
    p.out_f32 = m->tp_buf;
    dec_gemv(e, p, name);
    RUN("all_reduce", (size_t)p.n[0] * 4, tp::all_reduce_sum_f32(m->tp, e.s, m->tp_buf, p.n[0], out));
}

bool linear_rht(const DLinear& L) { return L.in_signs || L.out_signs; }
bool linear_fusable(const DLinear& L) {
    if (!L.w || L.lora_rank) return false; // QLoRA linears run as their wrapper composes them (unfused decode)
    // RHT linears (round 4): InputRht in the Normalization prologue of the GEMV or as a launch of its own in front of a plain-row GEMV, OutputRht
    // (+ bias) in the prologue of the next normalised GEMV or as a launch of its own (encode_decode_fused); sign vectors of +-1 only
    if (linear_rht(L) && !(L.in_bits && L.out_bits && L.n % 32 == 0)) return false;
    if (L.method == UZU_QUANT_NONE || (L.bits != 4 && L.bits != 8)) return false;
    return L.k % 32 == 0 && L.group % 32 == 0 && (L.group & (L.group - 1)) == 0 && L.k <= 32768;
}
bool norm_fusable(const DNorm& N) { return N.present && !N.subtract_mean && !N.biases && !N.scalar_mode; }
// the fused Normalization prologue needs model_dim % 1024 == 0 and <= 8192 (k_decode.hip)
bool dim_fusable(uint32_t d) { return d % 1024 == 0 && d <= 8192; }

bool model_fusable(const uzu_hip_model* m) {
    if (m->gemma_options) return false; // post-layer scalars, embedding norm, KV sharing, value normalisation, per-layer embeddings: the one-kernel-per-reference-kernel pass
    if (m->d.logit_scale != 1.0f || m->d.logit_soft_cap != 0.0f) return false;
    if (!dim_fusable(m->d.model_dim)) return false;
    if (!norm_fusable(m->output_norm)) return false;
    if (!linear_fusable(m->d.tied_embeddings ? m->embedding : m->output_embedding)) return false;
    if (m->embedding.in_signs || m->embedding.out_signs) return false; // RHT embedding rows: the commit kernel's lookup has no transform
    for (const DLayer& L : m->layers) {
        if (!norm_fusable(L.pre_mixer) || !norm_fusable(L.pre_mlp) || L.post_mixer.present || L.post_mlp.present) return false;
        if (!linear_fusable(L.up) || !linear_fusable(L.down)) return false;
        if (L.d.mixer_kind == UZU_MIXER_ATTENTION) {
            if (!linear_fusable(L.qkv) || !linear_fusable(L.out)) return false;
            if (L.d.has_gate && (!linear_fusable(L.gate) || L.gate.bits != L.qkv.bits || L.gate.group != L.qkv.group || L.gate.method != L.qkv.method)) return false;
            // (the two matrices of the fused launch share one prologue: with different input transforms the gate gets a launch of its own)
            if (!(L.d.head_dim == 64 || L.d.head_dim == 128 || L.d.head_dim == 256)) return false;
            if (L.d.sliding_window_size || L.d.has_sinks) return false; // ring KV state / sinks: the one-kernel-per-reference-kernel path (attn_dec has neither)
            if ((L.qn.present && (L.qn.subtract_mean || L.qn.biases)) || (L.kn.present && (L.kn.subtract_mean || L.kn.biases))) return false;
        } else {
            if (!linear_fusable(L.in_proj) || !linear_fusable(L.out_proj)) return false;
            if (linear_rht(L.in_proj) && m->tp) return false; // (the stand-alone conv / update kernels of the RHT route are not sharded here)
            if (L.d.dn_head_dim != 128 || L.d.dn_value_head_dim > 512 || L.d.dn_kernel_size != 4) return false; // conv epilogue of the in-proj GEMV
            {   // norm-gate prologue of the out-proj GEMV (k_decode.hip): chunks of 8 outputs, <= 4 chunks per thread
                const uint32_t dv = L.d.dn_value_head_dim, kk = L.d.dn_num_heads * dv, nchunks = kk / 8, per = nchunks > 256 ? nchunks / 256 : 1;
                if (dv < 8 || (dv & (dv - 1)) || kk > 8192 || (nchunks > 256 && (nchunks % 256 || per > 4)) || (dv / 8) % per) return false;
            }
        }
    }
    return true;
}

// The commit kernel of a single-GPU fused step also writes the sampled token's embedding row (the next step's input).
bool commit_embeds(const uzu_hip_model* m) { return m->tp == nullptr; }

void encode_embed_row0(Enc& e) {
    uzu_hip_model* m = e.m;
    const uint32_t d = m->d.model_dim;
    if (m->embedding.method == UZU_QUANT_NONE)
        RUN("full_precision_embedding_lookup", 0, k::full_precision_embedding_lookup(e.s, m->d_tokens, m->embedding.w, m->hidden, UZU_BF16, 1, m->d.vocab_size, d, m->d.input_scale));
    else
        RUN("quantized_embedding_lookup", 0, k::quantized_embedding_lookup(e.s, m->d_tokens, (const uint8_t*)m->embedding.w, m->embedding.scales, m->embedding.zp, m->embedding.biases,
                                            m->hidden, UZU_BF16, 1, m->d.vocab_size, d, m->d.input_scale, m->embedding.group, m->embedding.bits, m->embedding.method));
}

// One decode step (count == 1, sampling) with the fused kernels of k_decode.hip.  `with_embed`: look the input token's
// embedding row up first (the first step after a prefill / set_next_token; later steps find it written by the commit).
uzu_status encode_decode_fused(uzu_hip_model* m, hipStream_t s, bool with_embed) {
    Enc e{m, s};
    e.prof = (std::vector<ProfEntry>*)m->prof_sink;
    m->launches = 0;
    const uint32_t d = m->d.model_dim;
    uint16_t* hidden = m->hidden;
    if (with_embed) encode_embed_row0(e);
    uint16_t* sc[2] = {m->shortcut, m->shortcut_b};
    int cur = 1; // the first norm (copy mode) writes sc[0]
    // RHT linears (RHTLinearWrapper, linear/rht_wrapper.rs:215-298) inside the fused step.  The raw output row of an out-projection / down
    // projection with Hadamard factors stays `pending`: the next GEMV with a Normalization prologue applies its OutputRht + bias to the row it
    // loads anyway (PRO == 3 instance of gemv_dec_kernel); a consumer that cannot (act-mul / conv epilogue) gets it flushed by the reference's
    // own two kernels first.  Rounding points are the unfused path's, so the two stay bit-identical.
    const DLinear* pending = nullptr;
    uint16_t* pending_row = nullptr;
    auto out_transform = [&](const DLinear& L, uint16_t* row) {
        if (!L.out_signs) return;
        RUN("activation_transform", 0, k::activation_transform(s, nullptr, row, nullptr, nullptr, nullptr, L.out_signs, UZU_BF16, 1, L.n, UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
        if (L.out_biases) RUN("tensor_add_bias", 0, k::tensor_add_bias(s, row, L.out_biases, row, UZU_BF16, UZU_BF16, L.n, L.n));
    };
    auto flush_pending = [&]() {
        if (pending) out_transform(*pending, pending_row);
        pending = nullptr;
    };
    auto in_transform = [&](const DLinear& L, const uint16_t* row) -> const uint16_t* {
        if (!L.in_signs) return row;
        RUN("activation_transform", 0, k::activation_transform(s, row, m->rht_scratch, nullptr, nullptr, nullptr, L.in_signs, UZU_BF16, 1, L.k, UZU_ACTIVATION_TRANSFORM_INPUT_RHT, 0, 0));
        return m->rht_scratch;
    };
    // `own`: the linear(s) behind this prologue; `epilogue`: the launch has an act-mul / conv epilogue (no Hadamard instance exists for those)
    auto next_norm = [&](k::DecGemvParams& p, const DNorm& N, int mode, const DLinear* own = nullptr, bool epilogue = false) {
        if (pending && epilogue) flush_pending();
        dec_add_norm(p, N, mode, sc[cur], sc[cur ^ 1]);
        cur ^= 1;
        if (pending) p.x_rht_bits = pending->out_bits, p.x_rht_bias = (const uint16_t*)pending->out_biases, pending = nullptr;
        if (own && own->in_bits) p.in_rht_bits = own->in_bits;
    };
    for (uint32_t l = 0; l < m->d.num_layers; ++l) {
        DLayer& L = m->layers[l];
        if (L.d.mixer_kind == UZU_MIXER_ATTENTION) {
            const uint32_t hd = L.d.head_dim, nq = L.d.num_heads, nkv = L.d.num_groups;
            k::DecGemvParams p = dec_gemv_base(L.qkv, hidden, m->qkv);
            const bool own_gate = L.d.has_gate && (linear_rht(L.qkv) != linear_rht(L.gate) || L.qkv.in_words != L.gate.in_words);
            if (L.d.has_gate && !own_gate) dec_add_second(p, L.gate, m->gate);
            next_norm(p, L.pre_mixer, l > 0 ? 2 : 1, &L.qkv);
            dec_gemv(e, p, own_gate ? "gemv_dec[norm+qkv]" : "gemv_dec[norm+qkv+gate]");
            if (own_gate) { // its own InputRht: the same Normalization again, of the residual row the launch above has just written (copy mode, nothing stored)
                k::DecGemvParams g = dec_gemv_base(L.gate, sc[cur], m->gate);
                dec_add_norm(g, L.pre_mixer, 1, nullptr, nullptr);
                g.in_rht_bits = L.gate.in_bits;
                dec_gemv(e, g, "gemv_dec[norm+gate]");
            }
            if (L.qkv.out_bits && L.d.has_gate && L.gate.out_bits) // both rows in one launch
                RUN("rht_out_rows", 0, k::rht_out_rows(s, m->qkv, L.qkv.out_bits, (const uint16_t*)L.qkv.out_biases, L.qkv.n, m->gate, L.gate.out_bits,
                                                        (const uint16_t*)L.gate.out_biases, L.gate.n, nullptr, nullptr, nullptr, 0, 0));
            else {
                out_transform(L.qkv, m->qkv);
                if (L.d.has_gate) out_transform(L.gate, m->gate);
            }
            k::AttnDecParams a{};
            a.qkv = m->qkv, a.keys = L.keys, a.values = L.values, a.cosines = L.rope_cos, a.sines = L.rope_sin, a.ctx_len = m->d_ctx_len;
            a.q_norm = {L.qn.present, L.qn.full_layer, L.qn.eps, L.qn.offset, L.qn.scales};
            a.k_norm = {L.kn.present, L.kn.full_layer, L.kn.eps, L.kn.offset, L.kn.scales};
            a.num_heads = nq, a.gqa_factor = nq / nkv, a.head_dim = hd, a.rope_dim = L.d.use_rope ? L.rope_dim : 0;
            a.scale = L.d.attention_scale != 0.0f ? L.d.attention_scale : 1.0f / sqrtf((float)hd);
            a.partials = m->dec_partials, a.sums = m->dec_sums, a.maxs = m->dec_maxs, a.cache_rows = m->max_positions;
            const size_t kv_bytes = (size_t)2 * (m->context_length + 1) * nkv * hd * 2;
            if (k::attn_dec_fused_supported(nq, nq / nkv, hd, m->dec_splits, m->ctx->num_cus)) {
                // "a fused SDPA decode kernel": pass 2 + SigmoidGate inside the launch (every workgroup merges its slice of its KV-head group's rows)
                a.tickets = m->dec_tickets, a.gate = L.d.has_gate ? m->gate : nullptr, a.out = m->attn_out;
                RUN("attn_dec", kv_bytes, k::attn_dec(s, a, m->dec_splits));
            } else {
                RUN("attn_dec", kv_bytes, k::attn_dec(s, a, m->dec_splits));
                RUN("attn_merge", 0, k::attn_merge(s, m->dec_partials, m->dec_sums, m->dec_maxs, L.d.has_gate ? m->gate : nullptr, m->attn_out, nq, hd, m->dec_splits));
            }
            if (L.out.in_bits && k::gemv_dec_plain_in_rht_supported(L.out.k, L.out.bits)) { // the out-projection's InputRht in the GEMV's registers (round 5)
                k::DecGemvParams op = dec_gemv_base(L.out, m->attn_out, m->mixed);
                op.in_rht_bits = L.out.in_bits;
                dec_gemv_row_parallel(e, op, "gemv_dec[out_proj]");
            } else {
                dec_gemv_row_parallel(e, dec_gemv_base(L.out, in_transform(L.out, m->attn_out), m->mixed), "gemv_dec[out_proj]");
            }
            if (L.out.out_signs) pending = &L.out, pending_row = m->mixed;
        } else {
            const uint32_t Hv = L.d.dn_num_heads, Hk = L.d.dn_num_groups, Dk = L.d.dn_head_dim, Dv = L.d.dn_value_head_dim;
            k::DecGemvParams p = dec_gemv_base(L.in_proj, hidden, m->in_proj);
            const uint32_t conv_dim = 2 * Hk * Dk + Hv * Dv;
            if (linear_rht(L.in_proj)) {
                // the conv needs the OutputRht of the row it convolves
                next_norm(p, L.pre_mixer, l > 0 ? 2 : 1, &L.in_proj);
                k::DecGemvParams ps = p;
                ps.ep_out_bits = L.in_proj.out_bits, ps.ep_bias = (const uint16_t*)L.in_proj.out_biases;
                ps.conv_w = L.conv_w, ps.conv_b = L.conv_b, ps.conv_state = L.conv_state, ps.conv_dim = conv_dim, ps.conv_ks = L.d.dn_kernel_size;
                if (L.in_proj.out_bits && k::gemv_dec_stripe_supported(ps, m->ctx->num_cus)) {
                    // round 5: the projection's workgroups own whole 32-row Hadamard blocks and finish them themselves (k_decode.hip, PRO == 5)
                    dec_gemv(e, ps, "gemv_dec[norm+in_proj+rht+conv]");
                } else { // projection, the transform, then DeltaNetConvUpdate as a launch of its own
                    dec_gemv(e, p, "gemv_dec[norm+in_proj]");
                    RUN("rht_out_rows", 0, k::rht_out_rows(s, m->in_proj, L.in_proj.out_bits, (const uint16_t*)L.in_proj.out_biases, L.in_proj.n, nullptr, nullptr, nullptr, 0, L.conv_w,
                                                            L.conv_b, L.conv_state, L.d.dn_kernel_size, conv_dim));
                }
            } else {
                next_norm(p, L.pre_mixer, l > 0 ? 2 : 1, nullptr, true);
                // DeltaNetConvUpdate rides in the in-proj epilogue: the lane that finishes a conv channel's row convolves it
                p.conv_w = L.conv_w, p.conv_b = L.conv_b, p.conv_state = L.conv_state, p.conv_dim = conv_dim, p.conv_ks = L.d.dn_kernel_size;
                dec_gemv(e, p, "gemv_dec[norm+in_proj+conv]");
            }
            {
                k::DeltaDecParams q{};
                q.in_proj = m->in_proj, q.a_log = L.a_log, q.dt_bias = L.dt_bias, q.state = L.ssm_state, q.o = m->dn_o, q.sz = m->dn_sz;
                q.num_v_heads = Hv, q.num_k_heads = Hk, q.head_v_dim = Dv, q.key_dim = Hk * Dk, q.value_dim = Hv * Dv;
                RUN("delta_dec", (size_t)2 * Hv * Dv * Dk * 4, k::delta_dec(s, q));
                // ... and the RMSNorm * SiLU(z) gate in the out-proj prologue (it needs all Dv outputs of a head)
                k::DecGemvParams op = dec_gemv_base(L.out_proj, m->delta_out, m->mixed);
                op.dg_o = m->dn_o, op.dg_sz = m->dn_sz, op.dg_w = L.dn_norm, op.dg_dv = Dv, op.dg_eps = L.d.dn_norm_epsilon;
                // an RHT out-projection (round 5: until then delta_net_update on 16 workgroups + an InputRht launch + a plain-row GEMV): its InputRht is
                // applied to the gated row inside the norm-gate prologue; its OutputRht (+ bias) rides in the next normalised GEMV's prologue as before
                op.in_rht_bits = L.out_proj.in_bits;
                dec_gemv_row_parallel(e, op, "gemv_dec[gate+out_proj]");
                if (L.out_proj.out_signs) pending = &L.out_proj, pending_row = m->mixed;
            }
        }
        const uint16_t* down_in = nullptr; // the down projection's input row once its InputRht has been applied
        if (linear_rht(L.up)) { // GatedActMul needs the OutputRht of both halves: projection, then the transform (+ bias) and the product
            k::DecGemvParams up = dec_gemv_base(L.up, m->mixed, m->up);
            next_norm(up, L.pre_mlp, 2, &L.up);
            down_in = L.down.in_bits ? m->rht_scratch : m->gated;
            k::DecGemvParams us = up;
            us.act_mul = 1, us.act_type = L.d.activation, us.out[0] = (uint16_t*)down_in;
            us.ep_out_bits = L.up.out_bits, us.ep_bias = (const uint16_t*)L.up.out_biases, us.ep_next_in_bits = L.down.in_bits;
            if (L.up.out_bits && k::gemv_dec_stripe_supported(us, m->ctx->num_cus)) {
                // round 5: up | gate rows, their OutputRht (+ bias), GatedActMul and the down projection's InputRht in ONE launch (k_decode.hip, PRO == 5)
                dec_gemv(e, us, "gemv_dec[norm+up+rht+act]");
            } else {
                dec_gemv(e, up, "gemv_dec[norm+up]");
                // ... as ONE launch with the down projection's InputRht (a thread per stripe; the reference's four kernels on one row)
                RUN("rht_mlp_join", 0, k::rht_mlp_join(s, m->up, L.up.out_bits, (const uint16_t*)L.up.out_biases, L.down.in_bits, (uint16_t*)down_in, L.d.hidden_dim, L.d.activation));
            }
        } else {
            k::DecGemvParams up = dec_gemv_base(L.up, m->mixed, m->gated);
            next_norm(up, L.pre_mlp, 2, nullptr, true);
            up.act_mul = 1, up.act_type = L.d.activation;
            dec_gemv(e, up, "gemv_dec[norm+up+act]");
        }
        dec_gemv_row_parallel(e, dec_gemv_base(L.down, down_in ? down_in : in_transform(L.down, m->gated), hidden), "gemv_dec[down]");
        if (L.down.out_signs) pending = &L.down, pending_row = hidden;
        if (m->taps) flush_pending(); // (debug taps hold finished rows)
        if (m->taps) RUN("tensor_copy", 0, k::tensor_copy(s, hidden, m->taps + ((size_t)l * m->chunk) * d, UZU_BF16, d));
    }
    m->tap_rows = 1;
    const DLinear& ro = m->d.tied_embeddings ? m->embedding : m->output_embedding;
    k::DecGemvParams r = dec_gemv_base(ro, hidden, m->logits);
    next_norm(r, m->output_norm, 2); // (an RHT read-out is not fused: model_fusable)
    r.normed_out = m->last_normed;
    r.part_val = m->amax_val, r.part_idx = m->amax_idx, r.part_capacity = kArgmaxPartials;
    uint32_t grid = 0;
    dec_gemv(e, r, "gemv_dec[norm+readout+argmax]", &grid);
    if (m->tp && m->sampling.on) { // stochastic sampling over the gathered row (stream.rs:598-600 seed), then the plain commit with that token
        RUN("derive_seed", 0, k::derive_seed(s, m->sampling.seed, m->d_ctx_len, 0, m->d_seed));
        RUN("tp_gather_logits", (size_t)m->d.vocab_size * 4, tp::gather_logits(m->tp, s, m->logits, ro.n, m->vocab_offset, m->d.vocab_size, 1, m->tp_gather_f32, m->tp_gather_bf16));
        k::UnifiedSamplingParams sp = m->sampling.p;
        sp.logits = m->tp_gather_bf16, sp.dt = UZU_BF16, sp.output = m->d_out_token, sp.seeds = m->d_seed, sp.vocab_size = m->d.vocab_size, sp.batch_size = 1;
        RUN("unified_sampling", (size_t)m->d.vocab_size * 2, k::unified_sampling(s, sp, m->sampling_scratch));
        k::CommitEmbed eb{};
        eb.token_in = m->d_out_token;
        RUN("argmax_commit", 0, k::argmax_commit(s, m->amax_val, m->amax_idx, grid, m->d_ctx_len, m->d_tokens, m->d_out_token, m->d_sampled, &eb));
    } else if (m->tp) {
        RUN("tp_argmax_key", 0, tp::argmax_key(s, m->amax_val, m->amax_idx, grid, m->vocab_offset, m->tp_key));
        RUN("all_reduce", 8, tp::all_reduce_max_u64(m->tp, s, m->tp_key, 1));
        RUN("tp_commit_key", 0, tp::commit_key(s, m->tp_key, m->d_ctx_len, m->d_tokens, m->d_out_token, m->d_sampled));
    } else {
        k::CommitEmbed eb{};
        eb.weights = (const uint8_t*)m->embedding.w, eb.scales = (const uint16_t*)m->embedding.scales, eb.zero_points = m->embedding.zp;
        eb.biases = (const uint16_t*)m->embedding.biases, eb.output = hidden;
        eb.vocab_size = m->d.vocab_size, eb.model_dim = d, eb.group_size = m->embedding.group, eb.bits = m->embedding.bits, eb.method = m->embedding.method;
        eb.input_scale = m->d.input_scale;
        if (m->sampling.on) { // stream.rs:598-600: the seed of a decode step is derived from the context length before it
            RUN("derive_seed", 0, k::derive_seed(s, m->sampling.seed, m->d_ctx_len, 0, m->d_seed));
            k::UnifiedSamplingParams sp = m->sampling.p;
            sp.logits = m->logits, sp.dt = UZU_BF16, sp.output = m->d_out_token, sp.seeds = m->d_seed, sp.vocab_size = ro.n, sp.batch_size = 1;
            RUN("unified_sampling", (size_t)ro.n * 2, k::unified_sampling(s, sp, m->sampling_scratch));
            eb.token_in = m->d_out_token;
        }
        RUN("argmax_commit", 0, k::argmax_commit(s, m->amax_val, m->amax_idx, grid, m->d_ctx_len, m->d_tokens, m->d_out_token, m->d_sampled, &eb));
    }
    if (e.st != UZU_OK) return e.st;
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("engine: fused decode launch failed: %s", hipGetErrorString(err));
        return UZU_ERR_HIP;
    }
    return UZU_OK;
}

// (the reference-order mode runs the one-kernel-per-reference-kernel path, eagerly: its kernels take scratch from the stream workspace)
bool decode_is_fused(const uzu_hip_model* m) { return m->fusable && !(m->flags & UZU_MODEL_NO_FUSION) && !k::exact_mode(); }

// eager decode step; keeps `hidden_ready` in step with what the step left behind
uzu_status encode_decode(uzu_hip_model* m, hipStream_t s) {
    if (decode_is_fused(m)) {
        const uzu_status st = encode_decode_fused(m, s, !m->hidden_ready);
        m->hidden_ready = st == UZU_OK && commit_embeds(m);
        return st;
    }
    m->hidden_ready = false;
    return encode_forward(m, s, 1, true);
}

uzu_status build_decode_graph(uzu_hip_model* m, hipGraphExec_t* out, bool two_pass) {
    hipStream_t s = m->ctx->stream;
    HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    m->regime_override = two_pass ? 1 : 0;
    // the captured step starts at layer 0 when its own commit leaves the next embedding row behind (enqueue_decode looks
    // the first one up eagerly); otherwise the lookup is part of the graph
    uzu_status st = decode_is_fused(m) ? encode_decode_fused(m, s, !commit_embeds(m)) : encode_forward(m, s, 1, true);
    m->regime_override = -1;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (st != UZU_OK) {
        if (g) (void)hipGraphDestroy(g);
        return st;
    }
    if (e != hipSuccess) {
        set_error("engine: graph capture failed: %s", hipGetErrorString(e));
        return UZU_ERR_HIP;
    }
    HIPCHK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(g));
    return UZU_OK;
}

uzu_status enqueue_decode(uzu_hip_model* m, uint32_t steps) {
    for (uint32_t i = 0; i < steps; ++i) {
        UZU_REQUIRE(m->context_length + 1 <= m->d.max_context_length, "decode: context length %u exceeds max_context_length %u", m->context_length + 1,
                    m->d.max_context_length);
        if ((m->flags & UZU_MODEL_NO_GRAPH) || k::exact_mode()) {
            UZU_PROPAGATE(encode_decode(m, m->ctx->stream));
        } else {
            const bool two = m->context_length + 1 > 1024;
            hipGraphExec_t* g = two ? &m->graph_two : &m->graph_single;
            if (!*g) UZU_PROPAGATE(build_decode_graph(m, g, two));
            if (decode_is_fused(m) && commit_embeds(m) && !m->hidden_ready) { // first step after a prefill / set_next_token
                Enc e{m, m->ctx->stream};
                encode_embed_row0(e);
                UZU_PROPAGATE(e.st);
            }
            HIPCHK(hipGraphLaunch(*g, m->ctx->stream));
            m->hidden_ready = decode_is_fused(m) && commit_embeds(m);
        }
        m->context_length += 1;
    }
    return UZU_OK;
}

} // namespace

extern "C" {

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
        TRY(upload_linear(m, h.up_projection, &L.up));
        TRY(upload_linear(m, h.down_projection, &L.down));
        if (h.up_projection.n != 2 * h.hidden_dim || h.down_projection.k != h.hidden_dim) {
            set_error("model_create: layer %u MLP shapes inconsistent", l);
            return fail(UZU_ERR_INVALID_ARGUMENT);
        }
        max_hidden = max_hidden > h.hidden_dim ? max_hidden : h.hidden_dim;
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
                    set_error("model_create: layer %u and its KV source %u differ in window / kv heads / head_dim", l, src);
                    return fail(UZU_ERR_INVALID_ARGUMENT);
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
#define ALLOC(field, type, elems) do { TRY(dev_alloc(m, (size_t)(elems) * sizeof(type), &p, true)); m->field = (type*)p; } while (0)
    TRY(state_build(m, &m->state0));
    bind_state(m, m->state0);
    m->max_seqs = (flags >> 8) & 0xFFu ? (flags >> 8) & 0xFFu : 1u;
    const size_t CB = C * m->max_seqs; // rows of a batched pass: row-major activation buffers are sized for it
    ALLOC(batch_tokens, uint32_t, CB);
    ALLOC(hidden, uint16_t, CB * d);
    ALLOC(normed, uint16_t, CB * d);
    m->rowsum_floats = (size_t)(d / 32) * (CB + 4); // groups of >= 32 elements
    ALLOC(rowsum, float, m->rowsum_floats);
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
        if (const char* ev = getenv("UZU_DEC_SPLITS")) splits = (uint32_t)atoi(ev);
        m->dec_splits = splits < 8 ? 8 : (splits > 128 ? 128 : splits);
        ALLOC(dec_partials, float, (size_t)max_heads * m->dec_splits * max_hd);
        ALLOC(dec_sums, float, (size_t)max_heads * m->dec_splits);
        ALLOC(dec_maxs, float, (size_t)max_heads * m->dec_splits);
        ALLOC(dec_tickets, uint32_t, (size_t)max_heads); // (zeroed by dev_alloc; groups <= heads)
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

static void drop_tree_graphs(uzu_hip_model* m, uzu_hip_state* st) { // st == nullptr: all of them
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

// A speculated tree that was verified but never accepted is void once the sequence moves on by any other route (prefill / decode advance
// the context: a later accept would compact KV rows at the new offsets and advance the DeltaNet states from stale tree buffers).
static void drop_pending_tree(uzu_hip_model* m) { m->tree.size = 0, m->tree.state = nullptr; }

// LanguageModelStream::new for `nseq` independent sequences at once: `count` prompt tokens each (token_ids row-major
// [nseq, count]), chunks of <= 1024 tokens per sequence, every chunk pass carrying all sequences (struct Seqs).
uzu_status uzu_hip_model_prefill_batch(uzu_hip_model* m, uzu_hip_state** states, uint32_t nseq, const uint32_t* token_ids, uint32_t count,
                                       uint32_t* first_tokens) {
    UZU_REQUIRE(m && states && token_ids && nseq > 0 && count > 0, "model_prefill_batch: null / empty input");
    UZU_REQUIRE(nseq <= m->max_seqs, "model_prefill_batch: %u sequences, model created for at most %u (UZU_MODEL_BATCH)", nseq, m->max_seqs);
    for (uint32_t i = 0; i < nseq; ++i) {
        UZU_REQUIRE(states[i] && states[i]->m == m, "model_prefill_batch: state %u is null or belongs to another model", i);
        for (uint32_t j = 0; j < i; ++j) UZU_REQUIRE(states[i] != states[j], "model_prefill_batch: state %u listed twice", i);
        UZU_REQUIRE(uzu_hip_state_context_length(states[i]) + count <= m->d.max_context_length, "model_prefill_batch: sequence %u exceeds max_context_length", i);
    }
    hipStream_t s = m->ctx->stream;
    uzu_hip_state* prev = m->bound;
    m->hidden_ready = false;
    drop_pending_tree(m);
    uint32_t max_heads = 0, max_hd = 0;
    for (auto& L : m->layers)
        if (L.d.mixer_kind == UZU_MIXER_ATTENTION) {
            max_heads = max_heads > L.d.num_heads ? max_heads : L.d.num_heads;
            max_hd = max_hd > L.d.head_dim ? max_hd : L.d.head_dim;
        }
    std::vector<uint32_t> staging((size_t)nseq * m->chunk);
    const uint32_t pass_rows = k::exact_mode() ? kSuffixCapacity : m->chunk; // reference-order mode: the reference's own passes (bit-identical logits)
    for (uint32_t start = 0; start < count; start += pass_rows) {
        const uint32_t n = count - start < pass_rows ? count - start : pass_rows;
        const bool last = start + n == count;
        for (uint32_t i = 0; i < nseq; ++i) memcpy(&staging[(size_t)i * n], token_ids + (size_t)i * count + start, (size_t)n * 4);
        HIPCHK(hipMemcpyAsync(m->batch_tokens, staging.data(), (size_t)nseq * n * 4, hipMemcpyHostToDevice, s));
        if (max_heads) UZU_PROPAGATE(ensure_partials(m, n * max_heads, max_hd)); // any sequence may be past 1024 keys
        UZU_PROPAGATE(encode_forward(m, s, n, last, states, nseq));
        HIPCHK(hipStreamSynchronize(s)); // the staging buffer is reused; also surfaces kernel faults per chunk
        if (m->tp) UZU_PROPAGATE(tp::p2p_check(m->tp));
        UZU_PROPAGATE(k::gemv_stream_check());
    UZU_PROPAGATE(k::attn_dec_check());
        for (uint32_t i = 0; i < nseq; ++i) {
            bind_state(m, states[i]);
            m->context_length += n;
        }
    }
    if (first_tokens)
        for (uint32_t i = 0; i < nseq; ++i) HIPCHK(hipMemcpy(first_tokens + i, states[i]->d_out_token, 4, hipMemcpyDeviceToHost));
    bind_state(m, prev);
    return UZU_OK;
}

uint32_t uzu_hip_model_context_length(const uzu_hip_model* m) { return m ? m->context_length : 0; }
size_t uzu_hip_model_weight_bytes(const uzu_hip_model* m) { return m ? m->weight_bytes : 0; }
uint32_t uzu_hip_model_decode_launch_count(const uzu_hip_model* m) { return m ? m->launches : 0; }
uzu_status uzu_hip_prefill_gemm_plan(uint32_t m, uint32_t n, uint32_t k, uint32_t bits, uint32_t group_size, uint32_t gated_act, uint32_t num_cus,
                                     uzu_prefill_gemm_plan* out) {
    if (!out || !m || !n || !k || (bits != 4 && bits != 8) || !group_size || k % group_size || !num_cus || (gated_act && (n & 1))) {
        set_error("prefill_gemm_plan: bad arguments");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    k::MatmulParams p{};
    static __attribute__((aligned(16))) uint16_t dummy[8] = {0};
    p.a = p.b = p.scales = p.biases = dummy, p.d = dummy; // only tested for presence / alignment
    p.m = m, p.n = n, p.k = k, p.bits = bits, p.group_size = group_size, p.b_kind = UZU_MATMUL_B_SCALE_BIAS, p.ab_scale = 1.0f;
    p.w_dt = p.a_dt = p.d_dt = UZU_BF16, p.act_mul = gated_act ? 1 : 0;
    k::gemm_q_mfma128_plan_query(p, (int)num_cus, &out->large_tile, &out->form, &out->splits, &out->workgroups);
    return UZU_OK;
}
uzu_status uzu_hip_decode_gemv_plan(uint32_t n0, uint32_t n1, uint32_t k, uint32_t bits, uint32_t normed, uint32_t gated_act, uint32_t num_cus,
                                    uzu_decode_gemv_plan* out) {
    if (!out || !n0 || !k || k % 32 || (bits != 4 && bits != 8) || !num_cus || (gated_act && (n0 & 1))) {
        set_error("decode_gemv_plan: bad arguments");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    k::DecGemvParams p{};
    static const float one = 1.0f;
    p.n[0] = n0, p.n[1] = n1, p.k = k, p.bits = bits, p.group_size = 128, p.act_mul = gated_act ? 1 : 0;
    if (normed) p.norm_scales = &one; // only tested for presence
    k::DecGemvPlan pl{};
    k::gemv_dec_plan_query(p, (int)num_cus, &pl);
    out->lanes_per_row = 1u << pl.lpr_log2, out->rows_per_lane_group = (uint32_t)pl.rows_per_lane_group, out->steps_per_lane = pl.steps_per_lane;
    out->waves_per_workgroup = pl.waves, out->batches = pl.wave_batches, out->workgroup_batches = pl.wg_batches, out->workgroups = pl.workgroups;
    return UZU_OK;
}

uzu_status uzu_hip_model_prefill(uzu_hip_model* m, const uint32_t* token_ids, uint32_t count, uint32_t* first_token) {
    UZU_REQUIRE(m && token_ids && count > 0, "model_prefill: null / empty input");
    UZU_REQUIRE(m->context_length + count <= m->d.max_context_length, "model_prefill: %u + %u tokens exceed max_context_length %u",
                m->context_length, count, m->d.max_context_length);
    hipStream_t s = m->ctx->stream;
    m->hidden_ready = false; // the prefill pass uses `hidden` for its own rows
    drop_pending_tree(m);
    const uint32_t pass_rows = k::exact_mode() ? kSuffixCapacity : m->chunk; // reference-order mode: the reference's own passes (bit-identical logits)
    for (uint32_t start = 0; start < count; start += pass_rows) {
        const uint32_t n = count - start < pass_rows ? count - start : pass_rows;
        const bool last = start + n == count;
        HIPCHK(hipMemcpyAsync(m->d_tokens, token_ids + start, (size_t)n * 4, hipMemcpyHostToDevice, s));
        {   // two-pass attention over this chunk (core/mod.rs:89-92: physical prefix + suffix > 1024; a ring's prefix is its window)
            uint32_t max_heads = 0, max_hd = 0;
            bool two_pass = false;
            for (auto& L : m->layers)
                if (L.d.mixer_kind == UZU_MIXER_ATTENTION) {
                    max_heads = max_heads > L.d.num_heads ? max_heads : L.d.num_heads;
                    max_hd = max_hd > L.d.head_dim ? max_hd : L.d.head_dim;
                    two_pass = two_pass || (L.d.sliding_window_size ? L.d.sliding_window_size : m->context_length) + n > 1024;
                }
            if (max_heads && two_pass) UZU_PROPAGATE(ensure_partials(m, n * max_heads, max_hd));
        }
        UZU_PROPAGATE(encode_forward(m, s, n, last));
        HIPCHK(hipStreamSynchronize(s)); // token_ids is caller memory; also surfaces kernel faults per chunk
        if (m->tp) UZU_PROPAGATE(tp::p2p_check(m->tp));
        UZU_PROPAGATE(k::gemv_stream_check());
    UZU_PROPAGATE(k::attn_dec_check());
        m->context_length += n;
    }
    if (first_token) HIPCHK(hipMemcpy(first_token, m->d_out_token, 4, hipMemcpyDeviceToHost));
    return UZU_OK;
}

static void drop_stale_graphs(uzu_hip_model* m);
uzu_status uzu_hip_model_decode_enqueue(uzu_hip_model* m, uint32_t steps) {
    UZU_REQUIRE(m, "model_decode: null model");
    UZU_REQUIRE(m->context_length > 0, "model_decode: prefill first (no input token)");
    drop_stale_graphs(m);
    drop_pending_tree(m);
    return enqueue_decode(m, steps);
}

uzu_status uzu_hip_model_read_tokens(uzu_hip_model* m, uint32_t first_position, uint32_t count, uint32_t* out_tokens) {
    UZU_REQUIRE(m && out_tokens, "model_read_tokens: null argument");
    UZU_REQUIRE(first_position + count <= m->max_positions, "model_read_tokens: range out of bounds");
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    if (m->tp) UZU_PROPAGATE(tp::p2p_check(m->tp));
    UZU_PROPAGATE(k::gemv_stream_check());
    UZU_PROPAGATE(k::attn_dec_check());
    HIPCHK(hipMemcpy(out_tokens, m->d_sampled + first_position, (size_t)count * 4, hipMemcpyDeviceToHost));
    return UZU_OK;
}

uzu_status uzu_hip_model_decode(uzu_hip_model* m, uint32_t steps, uint32_t* out_tokens, float* gpu_ms) {
    UZU_REQUIRE(m, "model_decode: null model");
    if (!steps) return UZU_OK;
    // make sure graph construction is not inside the timed region
    drop_stale_graphs(m);
    if (!(m->flags & UZU_MODEL_NO_GRAPH) && !k::exact_mode()) {
        if (m->context_length + 1 <= 1024 && !m->graph_single) UZU_PROPAGATE(build_decode_graph(m, &m->graph_single, false));
        if (m->context_length + steps > 1024 && !m->graph_two) UZU_PROPAGATE(build_decode_graph(m, &m->graph_two, true));
    }
    const uint32_t first = m->context_length;
    HIPCHK(hipEventRecord(m->ev0, m->ctx->stream));
    UZU_PROPAGATE(uzu_hip_model_decode_enqueue(m, steps));
    HIPCHK(hipEventRecord(m->ev1, m->ctx->stream));
    HIPCHK(hipEventSynchronize(m->ev1));
    if (m->tp) UZU_PROPAGATE(tp::p2p_check(m->tp));
    UZU_PROPAGATE(k::gemv_stream_check());
    UZU_PROPAGATE(k::attn_dec_check());
    if (gpu_ms) HIPCHK(hipEventElapsedTime(gpu_ms, m->ev0, m->ev1));
    if (out_tokens) UZU_PROPAGATE(uzu_hip_model_read_tokens(m, first, steps, out_tokens));
    return UZU_OK;
}

uzu_status uzu_hip_model_profile_decode_step(uzu_hip_model* m, uint32_t capacity, const char** names, uint64_t* bytes, float* ms, uint32_t* count) {
    UZU_REQUIRE(m && names && bytes && ms && count, "model_profile_decode_step: null argument");
    UZU_REQUIRE(m->context_length > 0 && m->context_length + 1 <= m->d.max_context_length, "model_profile_decode_step: bad context length");
    std::vector<ProfEntry> prof;
    m->prof_sink = &prof;
    uzu_status st = encode_decode(m, m->ctx->stream);
    m->prof_sink = nullptr;
    hipError_t e = hipStreamSynchronize(m->ctx->stream);
    if (st == UZU_OK && e != hipSuccess) {
        set_error("model_profile_decode_step: %s", hipGetErrorString(e));
        st = UZU_ERR_HIP;
    }
    if (st == UZU_OK) m->context_length += 1;
    uint32_t n = 0;
    for (auto& p : prof) {
        float t = 0.f, tx = 0.f;
        (void)hipEventElapsedTime(&t, p.e0, p.e1);
        // the launch's own begin -> end where the launch went through the timed path (every kernel of this library does); a
        // launch that did not (a library call such as an RCCL collective) keeps the bracketed time
        if (hipEventElapsedTime(&tx, p.x0, p.x1) == hipSuccess && tx > 0.f && tx <= t) t = tx;
        else (void)hipGetLastError();
        if (n < capacity) names[n] = p.name, bytes[n] = p.bytes, ms[n] = t, ++n;
        for (hipEvent_t ev : {p.e0, p.e1, p.x0, p.x1}) (void)hipEventDestroy(ev);
    }
    *count = n;
    return st;
}

// graphs captured under another sampling configuration are stale: drop them (they are rebuilt on the next decode)
static void drop_stale_graphs(uzu_hip_model* m) {
    if (m->graph_epoch == m->sampling_epoch) return;
    if (m->graph_single) (void)hipGraphExecDestroy(m->graph_single);
    if (m->graph_two) (void)hipGraphExecDestroy(m->graph_two);
    m->graph_single = m->graph_two = nullptr;
    m->graph_epoch = m->sampling_epoch;
}

uzu_status uzu_hip_model_set_sampling(uzu_hip_model* m, const uzu_sampling_config* cfg) {
    UZU_REQUIRE(m, "model_set_sampling: null model");
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    if (!cfg) {
        if (m->sampling.on) ++m->sampling_epoch;
        m->sampling.on = false;
        return UZU_OK;
    }
    // a vocab-sharded read-out gathers the whole row on every rank first (tp::gather_logits): every rank then draws the same token
    UZU_PROPAGATE(ensure_tp_gather(m, k::kDnTreeMaxNodes)); // (tree passes sample every node; one size: captured graphs hold the pointers)
    UZU_REQUIRE(!cfg->has_temperature || cfg->temperature > 0.0f, "model_set_sampling: temperature must be positive");
    UZU_REQUIRE(!cfg->has_top_k || cfg->top_k > 0, "model_set_sampling: top_k must be positive");
    m->sampling.on = true;
    m->sampling.seed = cfg->seed;
    k::UnifiedSamplingParams& p = m->sampling.p;
    p = k::UnifiedSamplingParams{};
    p.has_temperature = cfg->has_temperature, p.temperature = cfg->temperature;
    p.has_top_k = cfg->has_top_k, p.top_k = cfg->top_k;
    p.has_top_p = cfg->has_top_p, p.top_p = cfg->top_p;
    p.has_min_p = cfg->has_min_p, p.min_p = cfg->min_p;
    ++m->sampling_epoch;
    return UZU_OK;
}

uzu_status uzu_hip_model_set_next_token(uzu_hip_model* m, uint32_t token) {
    UZU_REQUIRE(m, "model_set_next_token: null model");
    HIPCHK(hipMemcpyAsync(m->d_tokens, &token, 4, hipMemcpyHostToDevice, m->ctx->stream));
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    m->hidden_ready = false;
    return UZU_OK;
}

uzu_status uzu_hip_model_read_logits(uzu_hip_model* m, uint16_t* logits_out) {
    UZU_REQUIRE(m && logits_out, "model_read_logits: null argument");
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    HIPCHK(hipMemcpy(logits_out, m->logits, (size_t)(m->d.tied_embeddings ? m->embedding.n : m->output_embedding.n) * 2, hipMemcpyDeviceToHost));
    return UZU_OK;
}

// ---- speculative decoding: one pass over a speculated tree, then accept a root path (stream.rs:380-470, 556-628) ----
static uzu_status ensure_tree(uzu_hip_model* m) {
    if (m->tree.allocated) return UZU_OK;
    const uint32_t N = k::kDnTreeMaxNodes;
    void* p = nullptr;
    m->tree.layers.resize(m->layers.size());
    uint32_t max_key = 0;
    for (size_t l = 0; l < m->layers.size(); ++l) {
        const uzu_layer_desc& h = m->layers[l].d;
        if (h.mixer_kind != UZU_MIXER_DELTA_NET) continue;
        const uint32_t key_dim = h.dn_num_groups * h.dn_head_dim, value_dim = h.dn_num_heads * h.dn_value_head_dim, conv_dim = 2 * key_dim + value_dim;
        auto& T = m->tree.layers[l];
        UZU_PROPAGATE(dev_alloc(m, (size_t)N * conv_dim * (h.dn_kernel_size - 1) * 4, &p));
        T.conv_states = (float*)p;
        UZU_PROPAGATE(dev_alloc(m, (size_t)N * key_dim * 2, &p));
        T.k = (uint16_t*)p;
        UZU_PROPAGATE(dev_alloc(m, (size_t)N * value_dim * 2, &p));
        T.v = (uint16_t*)p;
        UZU_PROPAGATE(dev_alloc(m, (size_t)N * h.dn_num_heads * 4, &p));
        T.log_decay = (float*)p;
        UZU_PROPAGATE(dev_alloc(m, (size_t)N * h.dn_num_heads * 4, &p));
        T.beta = (float*)p;
        max_key = max_key > key_dim ? max_key : key_dim;
    }
    const uint32_t vocab_rows = m->d.tied_embeddings ? m->embedding.n : m->output_embedding.n;
    UZU_PROPAGATE(dev_alloc(m, (size_t)N * 3 * 4, &p));
    m->tree.d_trie = (uint32_t*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)N * 4, &p));
    m->tree.d_parents = (int32_t*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)N * 4, &p));
    m->tree.d_sampled = (uint32_t*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)N * 4, &p));
    m->tree.d_accepted = (uint32_t*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)N * (max_key ? max_key : 1) * 2, &p));
    m->tree.q = (uint16_t*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)N * m->d.model_dim * 2, &p));
    m->tree.normed = (uint16_t*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)N * vocab_rows * 2, &p));
    m->tree.logits = (uint16_t*)p;
    UZU_PROPAGATE(dev_alloc(m, k::argmax_scratch_bytes(N), &p));
    m->tree.argmax_scratch = p; // the arg-max partials of N rows
    UZU_PROPAGATE(dev_alloc(m, (size_t)N * 8, &p));
    m->tree.d_seeds = (uint64_t*)p;
    UZU_PROPAGATE(dev_alloc(m, k::unified_sampling_scratch_bytes(N), &p));
    m->tree.sampling_scratch = p;
    m->tree.allocated = true;
    return UZU_OK;
}

// One forward pass over `tree_size` speculated tokens in DFS order hanging off the bound sequence (trie_nodes: {trie_start, trie_end,
// height} per node, FlatTrie::token_subtrie_ranges; node 0 = the root = the last sampled token): token positions = context + height,
// attention under the trie mask, DeltaNet layers through tree-verify, greedy token of EVERY node into sampled_out.  Nothing is accepted:
// follow with uzu_hip_model_accept.
uzu_status uzu_hip_model_verify_tree(uzu_hip_model* m, const uint32_t* token_ids, const uint32_t* trie_nodes, uint32_t tree_size, uint32_t* sampled_out) {
    return uzu_hip_model_verify_tree_seeded(m, token_ids, trie_nodes, nullptr, tree_size, sampled_out);
}

// ... with the trie's own per-node sampling seeds (FlatTrie::token_seeds, stream.rs:694: the speculator sets them); null = every node draws with
// PRng::derive(context + height), the convention of the reference's own speculators (dflash_tfm.rs:267,304).  Ignored under greedy sampling.
uzu_status uzu_hip_model_verify_tree_seeded(uzu_hip_model* m, const uint32_t* token_ids, const uint32_t* trie_nodes, const uint64_t* seeds, uint32_t tree_size,
                                            uint32_t* sampled_out) {
    UZU_REQUIRE(m && token_ids && trie_nodes && tree_size > 0, "model_verify_tree: null / empty input");
    (void)hipSetDevice(m->ctx->device);
    UZU_UNSUPPORTED(tree_size > k::kDnTreeMaxNodes, "model_verify_tree: %u nodes (at most %u per pass)", tree_size, k::kDnTreeMaxNodes);
    UZU_REQUIRE(m->tree.size == 0, "model_verify_tree: a speculated tree is already pending (accept it first)");
    UZU_REQUIRE(m->context_length > 0, "model_verify_tree: prefill first");
    UZU_REQUIRE(m->context_length + tree_size <= m->d.max_context_length, "model_verify_tree: %u + %u tokens exceed max_context_length %u", m->context_length, tree_size,
                m->d.max_context_length);
    // BatchTopology::new (batch_topology.rs:11-37): parents from the heights of the DFS order; also validates the nodes
    std::vector<int32_t> parents(tree_size);
    {
        std::vector<uint32_t> stack;
        for (uint32_t i = 0; i < tree_size; ++i) {
            const uint32_t start = trie_nodes[3 * i], end = trie_nodes[3 * i + 1], height = trie_nodes[3 * i + 2];
            UZU_REQUIRE(start == i && end >= i && end < tree_size && height <= stack.size() && (i > 0 || height == 0), "model_verify_tree: node %u {%u, %u, %u} is not a DFS-ordered trie node", i, start,
                        end, height);
            stack.resize(height);
            parents[i] = stack.empty() ? -1 : (int32_t)stack.back();
            UZU_REQUIRE(i == 0 || parents[i] >= 0, "model_verify_tree: node %u is a second root", i);
            stack.push_back(i);
        }
    }
    UZU_PROPAGATE(ensure_tree(m));
    hipStream_t s = m->ctx->stream;
    m->hidden_ready = false;
    HIPCHK(hipMemcpyAsync(m->d_tokens, token_ids, (size_t)tree_size * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->tree.d_trie, trie_nodes, (size_t)tree_size * 12, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->tree.d_parents, parents.data(), (size_t)tree_size * 4, hipMemcpyHostToDevice, s));
    m->tree.host_seeds = seeds != nullptr && m->sampling.on;
    if (m->tree.host_seeds) HIPCHK(hipMemcpyAsync(m->tree.d_seeds, seeds, (size_t)tree_size * 8, hipMemcpyHostToDevice, s));
    {
        uint32_t max_heads = 0, max_hd = 0;
        bool two_pass = false;
        for (auto& L : m->layers)
            if (L.d.mixer_kind == UZU_MIXER_ATTENTION) {
                max_heads = max_heads > L.d.num_heads ? max_heads : L.d.num_heads;
                max_hd = max_hd > L.d.head_dim ? max_hd : L.d.head_dim;
                two_pass = two_pass || m->context_length + tree_size > 1024;
            }
        if (max_heads && two_pass) UZU_PROPAGATE(ensure_partials(m, tree_size * max_heads, max_hd));
    }
    const bool two_pass_regime = m->context_length + tree_size > 1024;
    HIPCHK(hipEventRecord(m->ev0, s));
    if ((m->flags & UZU_MODEL_NO_GRAPH) || k::exact_mode()) {
        m->tree.active = true;
        const uzu_status st = encode_forward(m, s, tree_size, true);
        m->tree.active = false;
        UZU_PROPAGATE(st);
    } else {
        hipGraphExec_t exec = nullptr;
        if (m->tree.graph_epoch != m->sampling_epoch) { // the captured passes bake the sampling kernels in
            drop_tree_graphs(m, nullptr);
            m->tree.graph_epoch = m->sampling_epoch;
        }
        for (auto& g : m->tree.graphs)
            if (g.state == m->bound && g.nodes == tree_size && g.two_pass == two_pass_regime && g.host_seeds == m->tree.host_seeds) exec = g.exec, m->launches = g.launches;
        if (!exec) {
            HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            m->regime_override = two_pass_regime ? 1 : 0;
            m->tree.active = true;
            const uzu_status st = encode_forward(m, s, tree_size, true);
            m->tree.active = false;
            m->regime_override = -1;
            hipGraph_t g = nullptr;
            const hipError_t ce = hipStreamEndCapture(s, &g);
            if (st != UZU_OK || ce != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);
                if (st == UZU_OK) set_error("model_verify_tree: graph capture failed: %s", hipGetErrorString(ce));
                return st != UZU_OK ? st : UZU_ERR_HIP;
            }
            HIPCHK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
            HIPCHK(hipGraphDestroy(g));
            m->tree.graphs.push_back({m->bound, tree_size, two_pass_regime, m->tree.host_seeds, exec, m->launches});
        }
        HIPCHK(hipGraphLaunch(exec, s));
    }
    HIPCHK(hipEventRecord(m->ev1, s));
    m->tree.sampled.resize(tree_size);
    HIPCHK(hipMemcpyAsync(m->tree.sampled.data(), m->tree.d_sampled, (size_t)tree_size * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    (void)hipEventElapsedTime(&m->tree.last_gpu_ms, m->ev0, m->ev1);
    if (m->tp) UZU_PROPAGATE(tp::p2p_check(m->tp));
    UZU_PROPAGATE(k::gemv_stream_check());
    UZU_PROPAGATE(k::attn_dec_check());
    if (sampled_out) memcpy(sampled_out, m->tree.sampled.data(), (size_t)tree_size * 4);
    m->tree.size = tree_size, m->tree.state = m->bound, m->tree.parents = parents;
    return UZU_OK;
}

// TransformerState::encode_accept (stream.rs:441-444) with the accepted root path of the pending tree (FlatTrie::accept, trie.rs:271-305):
// attention caches compact the accepted rows (mixer/attention/state.rs:174-198), DeltaNet layers take the last accepted node's conv state
// and advance the SSM state along the path (delta_net.rs:65-120).  The token sampled at the last accepted node becomes the next input.
uzu_status uzu_hip_model_accept(uzu_hip_model* m, const uint32_t* accepted_indices, uint32_t count) {
    UZU_REQUIRE(m && accepted_indices && count > 0, "model_accept: null / empty input");
    (void)hipSetDevice(m->ctx->device);
    UZU_REQUIRE(m->tree.size > 0 && m->tree.state == m->bound, "model_accept: no speculated tree is pending on the bound sequence");
    for (uint32_t i = 0; i < count; ++i) { // delta_net.rs:88-90, state.rs:179
        UZU_REQUIRE(accepted_indices[i] < m->tree.size, "model_accept: index %u out of the tree", accepted_indices[i]);
        UZU_REQUIRE(m->tree.parents[accepted_indices[i]] == (i ? (int32_t)accepted_indices[i - 1] : -1), "model_accept: the accepted indices are not a root path of the tree");
    }
    hipStream_t s = m->ctx->stream;
    Enc e{m, s};
    HIPCHK(hipMemcpyAsync(m->tree.d_accepted, accepted_indices, (size_t)count * 4, hipMemcpyHostToDevice, s));
    std::vector<uzu_kv_copy> copies;
    for (uint32_t i = 0; i < count; ++i)
        if (accepted_indices[i] != i) copies.push_back({m->context_length + accepted_indices[i], m->context_length + i});
    for (size_t l = 0; l < m->layers.size(); ++l) {
        DLayer& L = m->layers[l];
        if (L.d.mixer_kind == UZU_MIXER_ATTENTION && L.d.is_kv_sharing) continue; // TransformerLayerStateType::Shared: nothing of its own to accept (transformer.rs:63-69)
        if (L.d.mixer_kind == UZU_MIXER_ATTENTION && L.d.sliding_window_size) {
            // AttentionStateType::Ring (state.rs:200-219): the accepted suffix rows (behind the ring, at window + index) enter the ring one by
            // one; with n tokens accepted so far the next slot is n mod window (what the offset / length bookkeeping amounts to)
            const uint32_t W = L.d.sliding_window_size;
            std::vector<uzu_kv_copy> ring(count);
            for (uint32_t i = 0; i < count; ++i) ring[i] = {W + accepted_indices[i], (m->context_length + i) % W};
            RUN("kv_cache_update", 0, k::kv_cache_update(s, L.keys, L.values, UZU_BF16, ring.data(), count, L.d.num_groups * L.d.head_dim));
        } else if (L.d.mixer_kind == UZU_MIXER_ATTENTION) {
            if (!copies.empty()) RUN("kv_cache_update", 0, k::kv_cache_update(s, L.keys, L.values, UZU_BF16, copies.data(), (uint32_t)copies.size(), L.d.num_groups * L.d.head_dim));
        } else {
            const auto& T = m->tree.layers[l];
            HIPCHK(hipMemcpyAsync(L.conv_state, (const char*)T.conv_states + (size_t)accepted_indices[count - 1] * L.conv_state_bytes, L.conv_state_bytes, hipMemcpyDeviceToDevice, s));
            RUN("dn_state_advance", L.ssm_state_bytes * 2, k::delta_net_state_advance(s, T.k, T.v, T.log_decay, T.beta, m->tree.d_accepted, L.ssm_state, count, L.d.dn_num_heads,
                                                                                     L.d.dn_num_groups, L.d.dn_head_dim));
        }
    }
    UZU_PROPAGATE(e.st);
    // control block: context length, the sampled tokens of the accepted nodes at their positions, the next input token
    std::vector<uint32_t> toks(count);
    for (uint32_t i = 0; i < count; ++i) toks[i] = m->tree.sampled[accepted_indices[i]];
    const uint32_t new_len = m->context_length + count;
    HIPCHK(hipMemcpyAsync(m->d_sampled + m->context_length, toks.data(), (size_t)count * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->d_tokens, &toks[count - 1], 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->d_out_token, &toks[count - 1], 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->d_ctx_len, &new_len, 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    m->context_length = new_len;
    m->hidden_ready = false;
    m->tree.size = 0, m->tree.state = nullptr;
    return UZU_OK;
}

// device time of the last tree pass in milliseconds (HIP events on the engine's stream around the pass)
uzu_status uzu_hip_model_verify_gpu_ms(uzu_hip_model* m, float* out_ms) {
    UZU_REQUIRE(m && out_ms, "model_verify_gpu_ms: null argument");
    *out_ms = m->tree.last_gpu_ms;
    return UZU_OK;
}

// logits (bf16 [tree_size, vocab rows]) of the pending tree's nodes
uzu_status uzu_hip_model_read_tree_logits(uzu_hip_model* m, uint16_t* logits_out) {
    UZU_REQUIRE(m && logits_out, "model_read_tree_logits: null argument");
    UZU_REQUIRE(m->tree.size > 0, "model_read_tree_logits: no speculated tree is pending");
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    const uint32_t vocab_rows = m->d.tied_embeddings ? m->embedding.n : m->output_embedding.n;
    HIPCHK(hipMemcpy(logits_out, m->tree.logits, (size_t)m->tree.size * vocab_rows * 2, hipMemcpyDeviceToHost));
    return UZU_OK;
}

// rows the last pass left in the taps / rows one layer's tap can hold: size the buffer of read_layer_output from `capacity`
uzu_status uzu_hip_model_layer_output_rows(uzu_hip_model* m, uint32_t* rows, uint32_t* capacity) {
    UZU_REQUIRE(m, "model_layer_output_rows: null model");
    if (rows) *rows = m->tap_rows;
    if (capacity) *capacity = m->chunk;
    return UZU_OK;
}

uzu_status uzu_hip_model_read_layer_output(uzu_hip_model* m, uint32_t layer, uint16_t* out, uint32_t* rows) {
    UZU_REQUIRE(m && out && layer < m->d.num_layers, "model_read_layer_output: bad argument");
    UZU_REQUIRE(m->taps, "model_read_layer_output: model was not created with UZU_MODEL_DEBUG_TAPS");
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    HIPCHK(hipMemcpy(out, m->taps + (size_t)layer * m->chunk * m->d.model_dim, (size_t)m->tap_rows * m->d.model_dim * 2, hipMemcpyDeviceToHost));
    if (rows) *rows = m->tap_rows;
    return UZU_OK;
}

// ---- tensor-parallel group (tp.hip) ----
uzu_status uzu_hip_tp_unique_id(uint8_t out[128]) {
    UZU_REQUIRE(out, "tp_unique_id: null argument");
    return uzu::tp::unique_id(out);
}
uzu_status uzu_hip_tp_comm_create(uzu_hip_context* ctx, const uint8_t id[128], int32_t rank, int32_t size, uzu_hip_tp_comm** out) {
    UZU_REQUIRE(ctx && id && out, "tp_comm_create: null argument");
    (void)hipSetDevice(ctx->device);
    uzu::tp::Comm* c = nullptr;
    UZU_PROPAGATE(uzu::tp::comm_create(id, rank, size, &c));
    *out = (uzu_hip_tp_comm*)c;
    return UZU_OK;
}
void uzu_hip_tp_comm_destroy(uzu_hip_tp_comm* comm) { uzu::tp::comm_destroy((uzu::tp::Comm*)comm); }
uzu_status uzu_hip_tp_comm_create_local(uzu_hip_context* ctx, int32_t rank, int32_t size, uzu_hip_tp_comm** out) {
    UZU_REQUIRE(ctx && out, "tp_comm_create_local: null argument");
    (void)hipSetDevice(ctx->device);
    uzu::tp::Comm* c = nullptr;
    UZU_PROPAGATE(uzu::tp::comm_create_local(rank, size, &c));
    *out = (uzu_hip_tp_comm*)c;
    return UZU_OK;
}
uzu_status uzu_hip_tp_p2p_export(uzu_hip_context* ctx, uzu_hip_tp_comm* comm, uint8_t out_handle[64]) {
    UZU_REQUIRE(ctx && comm, "tp_p2p_export: null argument");
    (void)hipSetDevice(ctx->device);
    return uzu::tp::p2p_export((uzu::tp::Comm*)comm, out_handle);
}
uzu_status uzu_hip_tp_p2p_connect(uzu_hip_context* ctx, uzu_hip_tp_comm* comm, const uint8_t* handles) {
    UZU_REQUIRE(ctx && comm, "tp_p2p_connect: null argument");
    (void)hipSetDevice(ctx->device);
    return uzu::tp::p2p_connect((uzu::tp::Comm*)comm, handles);
}
void uzu_hip_tp_p2p_disable(uzu_hip_tp_comm* comm) { uzu::tp::p2p_disable((uzu::tp::Comm*)comm); }
uzu_status uzu_hip_tp_p2p_error(uzu_hip_tp_comm* comm, uint32_t* out) { return uzu::tp::p2p_error((uzu::tp::Comm*)comm, out); }
uzu_status uzu_hip_tp_comm_stats(uzu_hip_tp_comm* comm, uint32_t* rccl_ranks, uint64_t* rccl_collectives, uint64_t* p2p_exchanges) {
    unsigned long long r = 0, p = 0;
    UZU_PROPAGATE(uzu::tp::comm_stats((uzu::tp::Comm*)comm, rccl_ranks, &r, &p));
    if (rccl_collectives) *rccl_collectives = r;
    if (p2p_exchanges) *p2p_exchanges = p;
    return UZU_OK;
}
// stand-alone collective entry points (tests, tools): in place on device buffers of the context's stream
uzu_status uzu_hip_tp_all_reduce_sum_f32(uzu_hip_context* ctx, uzu_hip_tp_comm* comm, uzu_hip_buffer* buf, size_t offset_bytes, size_t count) {
    UZU_REQUIRE(ctx && comm && buf && offset_bytes + count * 4 <= buf->size, "tp_all_reduce_sum_f32: bad argument");
    (void)hipSetDevice(ctx->device);
    return uzu::tp::all_reduce_sum_f32((uzu::tp::Comm*)comm, ctx->stream, (float*)((char*)buf->dptr + offset_bytes), count);
}
uzu_status uzu_hip_tp_all_reduce_max_u64(uzu_hip_context* ctx, uzu_hip_tp_comm* comm, uzu_hip_buffer* buf, size_t offset_bytes, size_t count) {
    UZU_REQUIRE(ctx && comm && buf && offset_bytes + count * 8 <= buf->size, "tp_all_reduce_max_u64: bad argument");
    (void)hipSetDevice(ctx->device);
    return uzu::tp::all_reduce_max_u64((uzu::tp::Comm*)comm, ctx->stream, (unsigned long long*)((char*)buf->dptr + offset_bytes), count);
}
uint32_t uzu_hip_model_logit_count(const uzu_hip_model* m) { return m ? (m->d.tied_embeddings ? m->embedding.n : m->output_embedding.n) : 0; }

} // extern "C"
