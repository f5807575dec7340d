// engine_types.h -- the model driver's shared types and the functions its translation units call across each other (not part of the C ABI).
//   engine_build.hip    model / state construction: device memory, weight upload, RoPE tables, uzu_hip_model_create / destroy, sequence states
//   engine_forward.hip  the forward encoders: one-kernel-per-reference-kernel pass (encode_forward), fused decode step, decode graphs
//   engine_api.hip      the C ABI on top: prefill, decode, sampling, speculative verify / accept, taps, tensor-parallel group
#pragma once

#include <math.h>
#include <string.h>

#include <vector>

#include "../../include/uzu_hip_engine.h"
#include "internal.h"
#include "kernels.h"
#include "kernels_decode.h"
#include "tp.h"

struct uzu_hip_model;
struct uzu_hip_state;

namespace uzu {
namespace eng {

constexpr uint32_t kSuffixCapacity = 1024; // ATTENTION_SUFFIX_CAPACITY, mixer/attention/state.rs:14: the reference's rows per forward pass (and its single- / two-pass rule)
// Rows of one PREFILL pass here (uzu_hip_model::chunk; UZU_PREFILL_CHUNK = 1024 ... 8192, a multiple of 1024).  The reference feeds a prompt in
// passes of <= 1024 tokens; on this chip a 1024-row GEMM of a 0.8B model is 64-128 tiles for 256 CUs, so the default pass is 2048 rows (same
// results up to the order of a split-K sum: the DeltaNet chunks and the attention key tiles of a 1024-aligned boundary do not move).
inline uint32_t prefill_chunk_rows() {
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
    // MoeBlock (mlp/moe/mod.rs:30-60; d.mlp_kind == UZU_MLP_MOE): bf16 tensors in HBM, the scalars stay in d.moe
    struct {
        uint16_t *router_weights = nullptr, *router_biases = nullptr, *w13 = nullptr, *w2 = nullptr, *up_biases = nullptr, *down_biases = nullptr;
    } moe;
};

} // namespace eng
} // namespace uzu

using uzu::eng::DLayer;
using uzu::eng::DLinear;
using uzu::eng::DNorm;
using uzu::eng::kSuffixCapacity;
namespace k = uzu::k;

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
    // scratch of the MoE layers (sized for chunk * max_seqs tokens x the widest active-expert count): MoeBlock::encode's scratch allocations (mod.rs:214-270)
    struct {
        int32_t *topk_ids = nullptr, *bucketed_ids = nullptr, *tok2row = nullptr;
        uint16_t *topk_probs = nullptr, *bucketed_probs = nullptr, *x_perm = nullptr, *y_partial = nullptr;
        uint32_t *offsets = nullptr, *sumk = nullptr, *row_expert_map = nullptr;
        float* hidden = nullptr;
    } moe;
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
    // The hidden-feature taps of a speculator (stream.rs:213-214,632-633; Transformer::capture_residual, transformer.rs:160-171,285-293): after each tapped
    // layer the residual-stream row shortcut + hidden of every row of the pass.  Production outputs (not UZU_MODEL_DEBUG_TAPS): uzu_hip_model_set_feature_layers
    std::vector<uint32_t> feature_layers; // tapped layer indices, in the caller's order
    uint16_t* features = nullptr;         // [feature_layers.size()][chunk][d]
    uint32_t feature_rows = 0;            // rows of the last pass
    // A DFlash draft model's layer stack runs on a uzu_hip_model of its own (engine_drafter.hip) without an embedding: the caller fills `hidden` with the rows,
    // the pass neither samples nor commits (its block is never accepted: dflash.rs:273-345), and leaves where the residual rows ended up
    bool headless = false;
    uint16_t* last_shortcut = nullptr;

    // fused decode path
    bool fusable = false;
    uint16_t* shortcut_b = nullptr; // ping-pong partner of `shortcut`
    uint16_t* rht_scratch = nullptr; // [rows][widest RHT input]: InputRht works on a copy of the rows
    // group row sums of `normed`, written by the normalisation that produced it (MatmulParams::pre_rowsum of the GEMMs that read it):
    // valid for (rs_rows rows, rs_k elements, groups of rs_group) until `normed` is written again
    float* rowsum = nullptr;
    size_t rowsum_floats = 0;
    uint32_t rs_rows = 0, rs_k = 0, rs_group = 0;
    // the same for the two other buffers a prefill GEMM reads: `gated` (filed by the up projection's GatedActMul epilogue in parts of 64 columns) and `delta_out`
    // (filed by the DeltaNet norm-gate in parts of one value head): MatmulParams::pre_rowsum + rowsum_parts_log2 of the down / out projection (round 6)
    struct FiledRowSums {
        float* buf = nullptr;
        size_t floats = 0;
        uint32_t rows = 0, k = 0, part = 0; // valid for `rows` rows of `k` elements in parts of `part` columns; rows == 0: nothing filed
    } rs_gated, rs_delta;
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
    // DecoderEncodeOutput::final_hidden of the last prefill / tree pass: the output-norm rows it sampled from (uzu_hip_model_read_final_hidden)
    const uint16_t* final_hidden = nullptr;
    uint32_t final_hidden_rows = 0;
    void* prof_sink = nullptr; // std::vector<ProfEntry>* while profiling one step
    int regime_override = -1; // graph capture: 0 = single-pass attention, 1 = two-pass (else decided by context_length)
};

namespace uzu {
namespace eng {

#define HIPCHK(expr) UZU_HIP_TRY(expr)

// ---- engine_build.hip: device memory of a model / a sequence state, weight upload, RoPE tables
uzu_status dev_alloc(uzu_hip_model* m, size_t bytes, void** out, bool zero = false);
void dev_free(uzu_hip_model* m, void* p);
uzu_status state_alloc(uzu_hip_state* st, size_t bytes, void** out, bool zero_by_contract = true);
void state_release(uzu_hip_state* st);
void state_free(uzu_hip_state* st);
uzu_status state_build(uzu_hip_model* m, uzu_hip_state** out);
void bind_state(uzu_hip_model* m, uzu_hip_state* st);
uzu_status upload_bytes(uzu_hip_model* m, const void* host, size_t bytes, void** out);
template <class T> uzu_status upload(uzu_hip_model* m, const void* host, size_t bytes, T** out) {
    void* p = nullptr;
    const uzu_status st = upload_bytes(m, host, bytes, &p);
    *out = (T*)p;
    return st;
}
uzu_status upload_linear(uzu_hip_model* m, const uzu_linear_desc& h, DLinear* o, bool is_embedding = false);
uzu_status upload_norm(uzu_hip_model* m, const uzu_norm_desc& h, uint32_t dim, DNorm* o);
void rope_tables(const uzu_rope_desc& r, uint32_t n_pos, std::vector<float>& cosines, std::vector<float>& sines);
void drop_tree_graphs(uzu_hip_model* m, uzu_hip_state* st); // st == nullptr: all of them

// ---- engine_forward.hip: the forward encoders
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

// `post`: the Normalization that reads `output` next: a split-K prefill GEMM then ends with one reduction + epilogue + normalisation launch and sets
// post->done; on every other path the caller runs the normalisation itself
struct PostNorm {
    k::NormParams p{};
    uint32_t done = 0;
};
// The sequences of one forward pass (engine_forward.hip)
struct Seqs {
    uzu_hip_state** st = nullptr;
    uint32_t n = 0;
    uint32_t count = 0;
    uint32_t rows() const { return (n ? n : 1) * count; }
};
void linear(Enc& e, const DLinear& L, const uint16_t* input, uint16_t* output, uint32_t batch, bool row_parallel = false, PostNorm* post = nullptr);
bool linear_gated(Enc& e, const DLinear& L, const uint16_t* input, uint16_t* gated_out, uint32_t batch, uint32_t act_type, const DLinear* consumer = nullptr);
bool rowsum_wanted(const uzu_hip_model* m, const DLinear& consumer, uint32_t rows, uint32_t part, const uzu_hip_model::FiledRowSums& f);
k::NormParams norm_params(Enc& e, const DNorm& N, const uint16_t* input, uint16_t* output, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim, const DLinear* consumer = nullptr);
void norm_issued(uzu_hip_model* m, const k::NormParams& p);
void norm(Enc& e, const DNorm& N, const uint16_t* input, uint16_t* output, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim, const DLinear* consumer = nullptr);
void attention_mixer(Enc& e, DLayer& L, const uint16_t* hidden, uint16_t* out, const Seqs& q, PostNorm* post = nullptr, bool first_done = false);
uzu_status ensure_tp_gather(uzu_hip_model* m, uint32_t rows);
uzu_status ensure_partials(uzu_hip_model* m, uint32_t rows, uint32_t head_dim);
uzu_status encode_forward(uzu_hip_model* m, hipStream_t s, uint32_t count, bool sample, uzu_hip_state** seqs = nullptr, uint32_t nseq = 0);
bool model_fusable(const uzu_hip_model* m);
bool decode_is_fused(const uzu_hip_model* m);
bool commit_embeds(const uzu_hip_model* m);
void encode_embed_row0(Enc& e);
uzu_status encode_decode(uzu_hip_model* m, hipStream_t s);
uzu_status build_decode_graph(uzu_hip_model* m, hipGraphExec_t* out, bool two_pass);
uzu_status enqueue_decode(uzu_hip_model* m, uint32_t steps);

} // namespace eng
} // namespace uzu
