// tp.h -- tensor-parallel exchange used by the engine (tp.hip): RCCL communicator + the small kernels either side of it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uzu_hip.h"

namespace uzu {
namespace tp {

struct Comm;

uzu_status unique_id(uint8_t out[128]);                                  // rank 0: ncclGetUniqueId
uzu_status comm_create(const uint8_t id[128], int rank, int size, Comm** out); // collective over the group
void comm_destroy(Comm* c);
uzu_status comm_stats(Comm* c, uint32_t* rccl_ranks, unsigned long long* rccl_calls, unsigned long long* p2p_calls);
int comm_rank(const Comm* c);
int comm_size(const Comm* c);

// one-shot peer-to-peer exchange for messages <= 32 KB (tp.hip): export this rank's mailbox, open the peers'
uzu_status p2p_export(Comm* c, uint8_t out_handle[64]);
uzu_status p2p_connect(Comm* c, const uint8_t* handles);
bool p2p_connected(const Comm* c);
void p2p_disable(Comm* c);
uzu_status p2p_error(Comm* c, uint32_t* out);
uzu_status p2p_check(Comm* c); // UZU_ERR_HIP (sticky) once a bounded wait gave up; called at the engine's host sync points
uzu_status comm_create_local(int rank, int size, Comm** out); // no RCCL communicator: P2P exchanges only (tests; small groups)

uzu_status all_reduce_sum_f32(Comm* c, hipStream_t s, float* buf, size_t count, uint16_t* bf16_out = nullptr); // in place; optional bf16 copy of the sums
uzu_status all_reduce_max_u64(Comm* c, hipStream_t s, unsigned long long* buf, size_t count);

uzu_status cast_f32_bf16(hipStream_t s, const float* in, uint16_t* out, size_t n);
// greedy arg-max across vocab shards: key = (orderable(logit) << 32) | ~global_index, reduced with max
uzu_status argmax_key(hipStream_t s, const float* part_val, const uint32_t* part_idx, uint32_t parts, uint32_t vocab_offset, unsigned long long* key);
uzu_status key_from_token(hipStream_t s, const uint16_t* logits, const uint32_t* local_token, uint32_t vocab_offset, unsigned long long* key);
uzu_status token_from_key(hipStream_t s, const unsigned long long* key, uint32_t* out_token);
// `rows` sampled rows at once (the nodes of a speculated tree): keys[r] from the local arg-max of row r, and back
uzu_status keys_from_tokens(hipStream_t s, const uint16_t* logits, size_t row_stride, const uint32_t* local_tokens, uint32_t vocab_offset, unsigned long long* keys, uint32_t rows);
uzu_status tokens_from_keys(hipStream_t s, const unsigned long long* keys, uint32_t* out_tokens, uint32_t rows);
// full bf16 logit rows [rows, vocab] on every rank from the ranks' shards [rows, local_n] (stochastic sampling needs the whole distribution):
// scatter into a zeroed f32 row at vocab_offset, all-reduce(sum) -- exact, one non-zero term per element
uzu_status gather_logits(Comm* c, hipStream_t s, const uint16_t* local, uint32_t local_n, uint32_t vocab_offset, uint32_t vocab, uint32_t rows, float* full_f32, uint16_t* full_bf16);
uzu_status commit_key(hipStream_t s, const unsigned long long* key, uint32_t* ctx_len, uint32_t* tokens, uint32_t* out_token, uint32_t* sampled);

} // namespace tp
} // namespace uzu
