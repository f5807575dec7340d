// engine_api.hip -- model driver above the kernel boundary (include/uzu_hip_engine.h): the C ABI of a running model.
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

extern "C" {

// A speculated tree that was verified but never accepted is void once the sequence moves on by any other route (prefill / decode advance
// the context: a later accept would compact KV rows at the new offsets and advance the DeltaNet states from stale tree buffers).
// Only the tree of the state that moves: another state's pending tree (verify on A, prefill B, bind A, accept) keeps its buffers and its suffix rows.
static void drop_pending_tree(uzu_hip_model* m, const uzu_hip_state* moving) {
    if (m->tree.state == moving) m->tree.size = 0, m->tree.state = nullptr;
}

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
    for (uint32_t i = 0; i < nseq; ++i) drop_pending_tree(m, states[i]);
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
    drop_pending_tree(m, m->bound);
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
    drop_pending_tree(m, m->bound);
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
