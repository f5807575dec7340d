/*
 * uzu_hip_engine.h -- C ABI of the model driver that sits ABOVE the kernel boundary (include/uzu_hip.h).
 *
 * In the reference this layer is Rust and backend-generic: `Decoder::encode`
 * (crates/backend-uzu/src/encodable_block/decoder.rs:138-203), `Transformer::encode` (transformer.rs:226-329),
 * `TransformerLayer::encode` (transformer_layer.rs:194-238) and the prefill / chained-decode loop of
 * `LanguageModelStream` (engine/language_model/stream/stream.rs:131-361, 363-782).  Once the Rust
 * `backends/hip` shim exists, the reference's own engine drives the kernels and this driver is not
 * needed for production; it exists so that parity tests and bench.py can run the full forward path
 * today (no Rust toolchain in this environment), and it is where the MI355X-specific execution
 * strategy lives: weights resident in HBM, one captured hipGraph per decode step, the sampled token
 * fed to the next step on the device (the reference's `encode_copy(prev.output_tokens -> token_ids)`
 * chaining, stream.rs:598-615), position-dependent scalars read from device memory.
 */
#ifndef UZU_HIP_ENGINE_H
#define UZU_HIP_ENGINE_H

#include "uzu_hip.h"
#include "uzu_model_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uzu_hip_model uzu_hip_model;

enum {
    UZU_MODEL_DEFAULT = 0,
    UZU_MODEL_NO_GRAPH = 1,   /* decode with plain stream launches instead of hipGraph replay */
    UZU_MODEL_NO_FUSION = 2,  /* one kernel per reference kernel (no fused prologues / epilogues) */
    UZU_MODEL_DEBUG_TAPS = 4  /* keep every layer's output of the last forward pass for inspection */
};
/* bits 8..15 of `flags`: sequences one batched prefill pass may carry (scratch is sized for it); 0 = 1 */
#define UZU_MODEL_BATCH(n) (((uint32_t)(n) & 0xFFu) << 8)

/* sizeof of {uzu_linear_desc, uzu_norm_desc, uzu_rope_desc, uzu_layer_desc, uzu_model_desc, uzu_dflash_desc} in THIS library build: the description structs have grown
 * in place over the rounds and uzu_layer_desc is an array element, so a caller built against another header must refuse to go on (uzu_amd/desc.py checks at import of the
 * engine; there is no version negotiation: one header, one library). */
void uzu_hip_desc_abi(uint32_t out[6]);

/* Uploads every tensor of `desc` into HBM (the desc's host pointers are not retained). */
uzu_status uzu_hip_model_create(uzu_hip_context* ctx, const uzu_model_desc* desc, uint32_t flags, uzu_hip_model** out);
void uzu_hip_model_destroy(uzu_hip_model* m);

/* ---- tensor parallelism (not in the reference): one process per GPU, RCCL over xGMI ----
 * Rank 0 draws an id (ncclGetUniqueId) and ships the 128 bytes to every rank (torch.distributed broadcast, a file,
 * MPI ...); every rank then calls uzu_hip_tp_comm_create collectively.  A tensor-parallel model is created from a
 * SHARD description (uzu_amd/tp.py: column-parallel qkv / in-proj / up, row-parallel out-proj / down with K split at
 * quant-group boundaries, the full embedding table for the lookup and `output_embedding` = this rank's rows of the
 * read-out starting at `vocab_offset`).  The engine all-reduces (sum, f32) after every row-parallel linear and
 * all-reduces (max) one packed (logit, index) key per sampled token; every rank returns the same token ids.
 * uzu_hip_model_read_logits then returns uzu_hip_model_logit_count() values: this rank's shard. */
typedef struct uzu_hip_tp_comm uzu_hip_tp_comm;
uzu_status uzu_hip_tp_unique_id(uint8_t out[128]);
uzu_status uzu_hip_tp_comm_create(uzu_hip_context* ctx, const uint8_t id[128], int32_t rank, int32_t size, uzu_hip_tp_comm** out);
void uzu_hip_tp_comm_destroy(uzu_hip_tp_comm* comm);
/* One-shot peer-to-peer exchange for the decode-sized all-reduces (<= 32 KB: the 4-20 KB rows after out-proj / down-proj and
 * the 8-byte arg-max key), csrc/tp.hip: every rank exports a mailbox (hipIpc handle, 64 bytes), the handles travel over any
 * host channel, every rank opens the others'; an all-reduce is then ONE kernel per rank -- push to all mailboxes, publish a
 * sequence number, poll the local mailbox, add in rank order (bit-identical on all ranks) -- and can be captured in a
 * hipGraph.  Larger messages (prefill) keep using RCCL.  uzu_hip_tp_comm_create_local makes a group without an RCCL
 * communicator (P2P exchanges only).  uzu_hip_tp_p2p_error returns the sequence number of an exchange whose bounded wait
 * (2 s) for a peer gave up, 0 if none. */
uzu_status uzu_hip_tp_comm_create_local(uzu_hip_context* ctx, int32_t rank, int32_t size, uzu_hip_tp_comm** out);
uzu_status uzu_hip_tp_p2p_export(uzu_hip_context* ctx, uzu_hip_tp_comm* comm, uint8_t out_handle[64]);
uzu_status uzu_hip_tp_p2p_connect(uzu_hip_context* ctx, uzu_hip_tp_comm* comm, const uint8_t* handles /* [size][64] */);
void uzu_hip_tp_p2p_disable(uzu_hip_tp_comm* comm); /* back to RCCL for every size (all ranks must agree on the path) */
uzu_status uzu_hip_tp_p2p_error(uzu_hip_tp_comm* comm, uint32_t* out);
/* What the group has actually used: ranks the RCCL communicator itself reports (ncclCommCount; 0 for a local group), collectives enqueued
 * through RCCL, exchanges enqueued through the mailboxes (host-side counts: a captured graph counts once, not per replay).  bench.py puts
 * them into the line's `tp` object.  Test hook: UZU_TP_INJECT_TIMEOUT_AT=<n> makes mailbox exchange number n fail as a timed-out wait does. */
uzu_status uzu_hip_tp_comm_stats(uzu_hip_tp_comm* comm, uint32_t* rccl_ranks, uint64_t* rccl_collectives, uint64_t* p2p_exchanges);
uzu_status uzu_hip_tp_all_reduce_sum_f32(uzu_hip_context* ctx, uzu_hip_tp_comm* comm, uzu_hip_buffer* buf, size_t offset_bytes, size_t count);
uzu_status uzu_hip_tp_all_reduce_max_u64(uzu_hip_context* ctx, uzu_hip_tp_comm* comm, uzu_hip_buffer* buf, size_t offset_bytes, size_t count);
/* `comm` may be NULL (single GPU; identical to uzu_hip_model_create).  The communicator must outlive the model. */
uzu_status uzu_hip_model_create_tp(uzu_hip_context* ctx, const uzu_model_desc* desc, uint32_t flags, uzu_hip_tp_comm* comm,
                                   uint32_t vocab_offset, uzu_hip_model** out);
uint32_t uzu_hip_model_logit_count(const uzu_hip_model* m);
/* ---- sequence state (LanguageModelState, engine/language_model/state.rs:9-16) ----
 * The reference keeps everything that belongs to one sequence (KV caches, DeltaNet conv / SSM states, token history)
 * apart from the model, so that several sequences share one set of weights (BASELINE configs 3 and 5).  A model is
 * created with a state of its own; further ones come from uzu_hip_state_create.  prefill / decode / reset / read_* act on
 * the state BOUND to the model (uzu_hip_model_bind_state; NULL re-binds the model's own).  Each state owns its captured
 * decode graphs.  States must be destroyed before their model. */
typedef struct uzu_hip_state uzu_hip_state;
uzu_status uzu_hip_state_create(uzu_hip_model* m, uzu_hip_state** out);
void uzu_hip_state_destroy(uzu_hip_state* st);
uzu_status uzu_hip_state_reset(uzu_hip_state* st);
uint32_t uzu_hip_state_context_length(const uzu_hip_state* st);
/* dst <- src (two states of one model): caches, recurrent states, token history, context length -- a prefilled prompt prefix can be
 * continued any number of times.  Synchronises the context's stream. */
uzu_status uzu_hip_state_copy(uzu_hip_state* dst, const uzu_hip_state* src);
uzu_status uzu_hip_model_bind_state(uzu_hip_model* m, uzu_hip_state* st);
/* Prefill `nseq` (<= UZU_MODEL_BATCH at creation) independent sequences with `count` prompt tokens each (token_ids is
 * row-major [nseq, count]): the linear layers see one matrix of nseq * chunk rows -- weights are streamed once for all
 * sequences -- while attention and DeltaNet run per sequence on its own state.  Per sequence the results are those of
 * uzu_hip_model_prefill on that state (the reference has no cross-sequence batching: SURVEY.md F10).  first_tokens[i]
 * receives the token sampled from sequence i's last row.  The binding in force before the call is restored. */
uzu_status uzu_hip_model_prefill_batch(uzu_hip_model* m, uzu_hip_state** states, uint32_t nseq, const uint32_t* token_ids, uint32_t count,
                                       uint32_t* first_tokens);
/* LanguageModelState reset: context length 0, DeltaNet conv / SSM state zeroed. */
uzu_status uzu_hip_model_reset(uzu_hip_model* m);
uint32_t uzu_hip_model_context_length(const uzu_hip_model* m);
size_t uzu_hip_model_weight_bytes(const uzu_hip_model* m);

/* LanguageModelStream::new: append `count` prompt tokens in chunks of <= 1024 (stream.rs:194-195),
 * sample greedily from the last row; the sampled token becomes the input of the next decode step. */
uzu_status uzu_hip_model_prefill(uzu_hip_model* m, const uint32_t* token_ids, uint32_t count, uint32_t* first_token);
/* `steps` chained greedy decode steps (Iterator::next x steps).  out_tokens[i] is the token sampled by step i
 * (the input of step 0 is the token sampled by the previous prefill / decode call).  Blocks until done;
 * gpu_ms (optional) receives the GPU time between the first and the last step measured with HIP events. */
uzu_status uzu_hip_model_decode(uzu_hip_model* m, uint32_t steps, uint32_t* out_tokens, float* gpu_ms);
/* Asynchronous form used by bench.py: enqueue `steps` decode steps, do not wait. */
uzu_status uzu_hip_model_decode_enqueue(uzu_hip_model* m, uint32_t steps);
uzu_status uzu_hip_model_read_tokens(uzu_hip_model* m, uint32_t first_position, uint32_t count, uint32_t* out_tokens);
/* SamplingMethod::Stochastic for the engine's own loop (the default is greedy).  As in LanguageModelStream, the seed of the row
 * at absolute position p is PRng::new(seed).derive(p) (encodable_block/sampling/prng.rs:12-24; stream.rs:248-258 for the last
 * row of a prefill, :598-600 for a decode step), derived on the device from the resident context length so that a replayed graph
 * draws a fresh seed every step; the token then comes from UnifiedSampling (unified_sampling.rs:33-98: temperature, top-k, top-p,
 * min-p, Gumbel-max).  NULL returns to greedy.  Captured decode graphs of every state are rebuilt at their next decode.  Not
 * available for a vocab-sharded (tensor-parallel) read-out. */
typedef struct {
    uint64_t seed;
    uint32_t has_temperature;
    float temperature;
    uint32_t has_top_k, top_k;
    uint32_t has_top_p;
    float top_p;
    uint32_t has_min_p;
    float min_p;
} uzu_sampling_config;
uzu_status uzu_hip_model_set_sampling(uzu_hip_model* m, const uzu_sampling_config* cfg);
/* Teacher forcing for parity tests: overwrite the next input token. */
uzu_status uzu_hip_model_set_next_token(uzu_hip_model* m, uint32_t token);

/* ---- speculative decoding: verify a speculated token tree in one pass, accept a root path ----
 * LanguageModelStream with a speculator (engine/language_model/stream/stream.rs:556-628 propose / verify, :380-470 accept).  The tree
 * arrives linearised in DFS order as the reference hands it to the decoder (trie.rs:154-172, 200-212): token_ids[i] and trie_nodes[3 i ..]
 * = {trie_start, trie_end, height} of node i (gpu_types/trie.rs: node j is an ancestor-or-self of i iff trie_start_j <= i <= trie_end_j);
 * node 0 is the root = the last sampled token of the sequence.  One forward pass over all nodes: token positions = context + height
 * (transformer.rs:247), attention under the trie mask (mask.rs:21-29; the nodes' K / V rows go behind the cache's logical end),
 * Gated DeltaNet layers through ConvTreeScan + tree prep + DeltaNetTreeVerify (encodable_block/mixer/delta_net.rs:334-437,
 * cpu/kernel/gdn/tree_verify/*.rs) with their Tree suffix status kept for the accept, output norm + read-out + greedy sampling of EVERY
 * node (sampled_out[tree_size]; greedy, or with uzu_hip_model_set_sampling the node's own seed PRng::derive(context + height),
 * speculators/dflash_tfm.rs:267,304).  At most 32 nodes per pass (the reference speculates <= 16).  Full and ring (sliding-window) KV
 * states.  On a tensor-parallel shard (uzu_hip_model_create_tp) the nodes' tokens come out of one all-reduce(max) of per-node (logit,
 * index) keys; stochastic sampling there draws from whole logit rows gathered on every rank (same seeds => the same tokens everywhere).
 *
 * uzu_hip_model_accept: TransformerState::encode_accept with the accepted root path (FlatTrie::accept, trie.rs:271-305; the host mirror
 * is uzu_amd/trie.py): KV rows of the accepted nodes compacted to context .. context + count - 1 (mixer/attention/state.rs:174-198),
 * DeltaNet conv state of the last accepted node + StateAdvance along the path (delta_net.rs:65-120); the context grows by `count` and
 * the token sampled at the last accepted node is the next input token (decode / the next verify_tree's root). */
uzu_status uzu_hip_model_verify_tree(uzu_hip_model* m, const uint32_t* token_ids, const uint32_t* trie_nodes, uint32_t tree_size, uint32_t* sampled_out);
/* ... with the trie's own per-node sampling seeds (FlatTrie::token_seeds, stream.rs:694); `seeds` NULL = PRng::derive(context + height) per
 * node on the device (the reference speculators' convention).  Ignored under greedy sampling. */
uzu_status uzu_hip_model_verify_tree_seeded(uzu_hip_model* m, const uint32_t* token_ids, const uint32_t* trie_nodes, const uint64_t* seeds, uint32_t tree_size,
                                            uint32_t* sampled_out);
uzu_status uzu_hip_model_accept(uzu_hip_model* m, const uint32_t* accepted_indices, uint32_t count);
/* Device time of the last verify_tree pass, ms (HIP events around the pass; the call itself adds three small uploads, one download and a sync). */
uzu_status uzu_hip_model_verify_gpu_ms(uzu_hip_model* m, float* out_ms);
/* bf16 logits [tree_size, vocab] of the pending tree's nodes (between verify_tree and accept). */
uzu_status uzu_hip_model_read_tree_logits(uzu_hip_model* m, uint16_t* logits_out);

/* ---- the tree speculator's draft model: DFlash (encodable_block/dflash.rs:41-346, speculators/dflash_tfm.rs) ----
 * The reference's LanguageModelStream asks its speculator for a tree in front of every verify pass (stream.rs:551-576): the DFlash draft model -- a few
 * TransformerLayers with block attention whose KV state is fed with projections of the TARGET's hidden features -- drafts a block of tokens in one pass;
 * tree shaping (the Argmax chain, trie, pruning) is host code (uzu_amd/speculator.py <- dflash_tfm.rs:133-343, uzu_amd/trie.py <- trie.rs).
 *
 * uzu_hip_model_set_feature_layers: the target's hidden-feature taps (DFlashTfmSpeculator::hidden_feature_layer_indices, stream.rs:213-214,632-633):
 *   every later pass (prefill chunk, decode step, tree pass) files Transformer::capture_residual(shortcut, hidden) = bf16(shortcut + hidden) of every row after
 *   each listed layer (transformer.rs:160-171,285-293).  Production outputs, independent of UZU_MODEL_DEBUG_TAPS; count 0 removes the taps.  A model with taps
 *   decodes through the one-kernel-per-reference-kernel pass.  Not on tensor-parallel shards (UZU_ERR_UNSUPPORTED).
 * uzu_hip_model_read_features: bf16 [rows, model_dim] of tap `index` (position in the list) for the rows of the last pass; `out` NULL = only `rows`.
 * uzu_hip_drafter_create: uploads the draft model (DFlash::new); it keeps a borrowed pointer to `target` (embedding lookup / read-out / taps), which must
 *   outlive it, and owns an AttentionState per layer for desc->context_capacity rows (DFlash::empty_state).
 * uzu_hip_drafter_accept: DFlash::encode_accept over rows `accepted_indices` of the target's LAST pass -- call it after every prefill chunk (indices 0..n) and,
 *   after a verify pass, BEFORE or AFTER uzu_hip_model_accept with the same indices (the taps are not touched by accept); the target's taps must be exactly
 *   desc->target_layer_ids in order.
 * uzu_hip_drafter_draft: DFlash::encode_draft for [target_output_token, mask, ...] (batch_size rows, 2..block_size) + the greedy token of every lookahead row
 *   (the Argmax construction, dflash_tfm.rs:167-217): tokens_out [batch_size - 1].  Nothing is accepted; the drafter's context does not move.
 * uzu_hip_drafter_read_draft: DFlashOutput of the last draft: draft_hidden bf16 [rows, model_dim], logits f32 [rows - 1, vocab]. */
typedef struct uzu_hip_drafter uzu_hip_drafter;
uzu_status uzu_hip_model_set_feature_layers(uzu_hip_model* m, const uint32_t* layer_ids, uint32_t count);
uzu_status uzu_hip_model_read_features(uzu_hip_model* m, uint32_t index, uint16_t* out, uint32_t* rows);
uzu_status uzu_hip_drafter_create(uzu_hip_context* ctx, uzu_hip_model* target, const uzu_dflash_desc* desc, uzu_hip_drafter** out);
void uzu_hip_drafter_destroy(uzu_hip_drafter* f);
uzu_status uzu_hip_drafter_reset(uzu_hip_drafter* f);
uint32_t uzu_hip_drafter_context_length(const uzu_hip_drafter* f);
uzu_status uzu_hip_drafter_accept(uzu_hip_drafter* f, const uint32_t* accepted_indices, uint32_t count);
uzu_status uzu_hip_drafter_draft(uzu_hip_drafter* f, uint32_t target_output_token, uint32_t batch_size, uint32_t* tokens_out);
uzu_status uzu_hip_drafter_read_draft(uzu_hip_drafter* f, uint16_t* draft_hidden_out, float* logits_out, uint32_t* rows);
/* device time of the last accept / draft, ms (HIP events on the engine's stream) */
uzu_status uzu_hip_drafter_gpu_ms(uzu_hip_drafter* f, float* accept_ms, float* draft_ms);

/* DecoderEncodeOutput::final_hidden (BU/../encodable_block/decoder.rs; ForwardPassChaining's output_norm, BU/../language_model/stream.rs:466-476) of the last prefill
 * (1 row: the sampled one) or tree pass (one row per node): the output-norm rows, bf16 [rows][model_dim].  `out` may be NULL (rows only). */
uzu_status uzu_hip_model_read_final_hidden(uzu_hip_model* m, uint16_t* out, uint32_t capacity_rows, uint32_t* rows);

/* The Weaver tree constructor of a DFlash speculator (BU/../encodable_block/weaver.rs:166-676, weaver_layer.rs:150-200; DFlashTfmTreeConstructionMethod::Weaver,
 * BU/../speculators/dflash_tfm.rs:224-292).  Built on a drafter (its last draft's hidden rows and f32 logits are the tree's inputs and stay in HBM) and that drafter's
 * target (embedding lookup, sparse read-out).  Destroy it BEFORE the drafter.  The launches of a shape (~25 per round on <= 32 rows) are captured once into a
 * hipGraph and replayed; in reference-order mode they run eagerly. */
typedef struct uzu_hip_weaver uzu_hip_weaver;
uzu_status uzu_hip_weaver_create(uzu_hip_context* ctx, uzu_hip_drafter* drafter, const uzu_weaver_desc* desc, uzu_hip_weaver** out);
void uzu_hip_weaver_destroy(uzu_hip_weaver* w);
uint32_t uzu_hip_weaver_max_depth(const uzu_hip_weaver* w);
/* Weaver::encode_tree over the drafter's LAST draft (uzu_hip_drafter_draft with batch_size == shape->dflash_depth).  target_hidden_row: bf16 [target_model_dim], the
 * target's output-norm row of the position that sampled root_token_id.  depth_seeds [depth_seed_count == desc.max_depth] = PRng::derive(root position + depth).
 * Outputs (host): packed_tree u32 [6][slots], frontier u32 [7][slots * expand_width], slots = 1 + (rounds - 1) * expand_per_round (EncodedWeaverTree's two
 * structure-of-arrays buffers; EncodedWeaverTree::read_nodes is host code: uzu_amd/speculator.py).  UZU_ERR_INVALID_ARGUMENT "invalid Weaver tree input" =
 * WeaverEncodeError::InvalidTreeInput. */
uzu_status uzu_hip_weaver_encode_tree(uzu_hip_weaver* w, const uint16_t* target_hidden_row, const uint64_t* depth_seeds, uint32_t depth_seed_count, uint32_t root_token_id,
                                      const uzu_weaver_tree_shape* shape, uint32_t* packed_tree_out, uint32_t* frontier_out);
/* device time (ms) and kernel launches of the last tree */
uzu_status uzu_hip_weaver_stats(uzu_hip_weaver* w, float* gpu_ms, uint32_t* launches);

/* bf16 logits [vocab] of the last sampled row. */
uzu_status uzu_hip_model_read_logits(uzu_hip_model* m, uint16_t* logits_out);
/* Debug taps (UZU_MODEL_DEBUG_TAPS): bf16 [rows, model_dim] output of `layer` in the last forward pass.
 * rows the last pass left in the taps, and the rows one layer's tap can hold (= rows of one prefill pass): `out` of read_layer_output must
 * hold rows * model_dim bf16 values -- size it from `capacity`. */
uzu_status uzu_hip_model_layer_output_rows(uzu_hip_model* m, uint32_t* rows, uint32_t* capacity);
/* bf16 [rows, model_dim] output of `layer` in the last forward pass (rows <= capacity above). */
uzu_status uzu_hip_model_read_layer_output(uzu_hip_model* m, uint32_t layer, uint16_t* out, uint32_t* rows);
/* Runs ONE decode step with plain launches and a HIP event pair around every kernel on the context stream;
 * returns per launch: kernel label (static string), algorithmic bytes (0 where not meaningful), duration in ms.
 * Used by bench.py for the per-kernel roofline of the dominant kernel. */
uzu_status uzu_hip_model_profile_decode_step(uzu_hip_model* m, uint32_t capacity, const char** names, uint64_t* bytes, float* ms,
                                             uint32_t* count);
/* Reference-ORDER mode (diagnostic; also UZU_HIP_EXACT=1): every reduction kernel of the forward path -- matmul, Normalization,
 * QKVNorm, attention single / two pass, DeltaNet update / prefill / norm-gate -- runs one thread per reduction in the reference CPU
 * kernel's loop order (csrc/k_exact.hip, k_matmul.hip::matmul_ref_kernel); the fused decode kernels, the matrix-core GEMMs /
 * attention, the chunked DeltaNet scan and the LDS weight stream are bypassed and decode runs eagerly.  A forward pass then
 * reproduces the CPU backend BIT FOR BIT (tests/test_gpu_model.py::test_exact_mode_*): what separates "reduction order" from real
 * defects.  Slow by construction; never used for measurements.  uzu_hip_set_exact_matmul is the older name of the same switch. */
void uzu_hip_set_exact(int32_t enabled);
void uzu_hip_set_exact_matmul(int32_t enabled);
/* Number of kernel launches / graph nodes of one decode step (reported by bench.py). */
uint32_t uzu_hip_model_decode_launch_count(const uzu_hip_model* m);
/* The launch plan of the decode GEMV (csrc/k_decode.hip: gemv_dec_plan) for a quantized linear of n0 (+ n1: a second matrix in the same
 * launch) rows over k columns on a device with `num_cus` compute units -- host arithmetic only, no GPU needed; the CPU tests pin the
 * decisions DESIGN.md section 3 quotes measurements for.  workgroup_batches / workgroups are non-zero when a matrix smaller than one
 * round of the resident waves is spread over every CU; otherwise the grid is the persistent one (occupancy x CUs, known at launch). */
typedef struct {
    uint32_t lanes_per_row, rows_per_lane_group, steps_per_lane, waves_per_workgroup;
    uint32_t batches; /* rounded up to a multiple of four */
    uint32_t workgroup_batches, workgroups;
} uzu_decode_gemv_plan;
uzu_status uzu_hip_decode_gemv_plan(uint32_t n0, uint32_t n1, uint32_t k, uint32_t bits, uint32_t normed, uint32_t gated_act, uint32_t num_cus,
                                    uzu_decode_gemv_plan* out);

/* The plan of the prefill GEMM (csrc/k_gemm128.hip) for a quantized linear of n rows over k columns at m activation rows on a device with
 * `num_cus` compute units -- host arithmetic only, no GPU needed.  large_tile = 0: the shape goes to the 64 x 64-tile kernel / the few-rows
 * kernel / the GEMV.  form: 0 = 256-thread workgroups (128 x 128 tile, weights straight into the MFMA operand), 1 = ping-pong (128 x 256 tile
 * per 512-thread workgroup, weight fragments converted once per half into LDS), 2 = wave-specialised (consumer waves fed from LDS + producer
 * waves); the three are bit-identical, the choice follows the same-box measurements in profiles/r5_gemm_pp_ab.txt.  workgroups = tiles with
 * work x splits. */
typedef struct {
    uint32_t large_tile, form, splits, workgroups;
} uzu_prefill_gemm_plan;
uzu_status uzu_hip_prefill_gemm_plan(uint32_t m, uint32_t n, uint32_t k, uint32_t bits, uint32_t group_size, uint32_t gated_act, uint32_t num_cus,
                                     uzu_prefill_gemm_plan* out);

#ifdef __cplusplus
}
#endif
#endif
