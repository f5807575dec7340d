// k_speculator.hip -- the kernels the reference's tree speculators run beside the forward path (BU = crates/backend-uzu/src):
//   AncestorAttention             BU/backends/cpu/kernel/attention/ancestor_attention.rs:8-139   (Metal: attention/ancestor_attention.metal:55-75)
//   WeaverFrontierSelect          BU/backends/cpu/kernel/weaver/weaver_frontier_select.rs:7-144
//   WeaverFrontierInsertChildren  BU/backends/cpu/kernel/weaver/weaver_frontier_insert_children.rs:16-65
//   WeaverTopChildren             BU/backends/cpu/kernel/weaver/weaver_top_children.rs:9-53
// Structure-of-arrays layouts of BU/backends/common/gpu_types/weaver.rs (field f of slot s at [f * capacity + s]).
//
// These are tiny index kernels (a frontier of <= 2048 slots, <= 32 nodes per round, <= 512 candidates per node): one launch each, sized
// for latency, results BIT-IDENTICAL to the CPU kernels -- integer selection rules as total orders (the sequential scan's "first wins" is
// the lower slot / index), the two float pieces (a log-sum-exp, Gumbel noise) in the reference's order with glibc-exact exp / log.
// AncestorAttention is tolerance-class like every attention kernel (keys split over waves, f32 online softmax).
#include "device_utils.h"
#include "kernels.h"
#include "sampling_noise.h"

namespace uzu {
namespace k {

namespace {
enum { FR_TOKEN = 0, FR_PARENT, FR_DEPTH, FR_PATH_LOGPROB, FR_EDGE_LOGPROB, FR_SCORE_KEY, FR_ACTIVE, FR_COUNT };
enum { TR_TOKEN = 0, TR_PARENT, TR_DEPTH, TR_PATH_LOGPROB, TR_EDGE_LOGPROB, TR_VALID, TR_COUNT };
enum { MD_DEPTH = 0, MD_ANCESTOR_COUNT, MD_TREE_SLOT, MD_COUNT };
constexpr uint32_t kNoWinner = 0xFFFFFFFFu, kMaxSlots = 2048u, kMaxWidth = 32u, kCandidatesMax = 512u;

// ------------------------------------------------------------------------------------------------ AncestorAttention
// half-rotation RoPE of head `head` of component `component` (0 = q, 1 = k) of a node's qkv row at position depth + 1 -> dst[0 .. HD) (f32
// values of the bf16 results); threads 0 .. HD / 2 - 1 each take a pair
template <int HD> __device__ __forceinline__ void rotate_head(const uint16_t* cur, const float* cosines, const float* sines, uint32_t model_dim, uint32_t head,
                                                              uint32_t component, uint32_t position, float* dst, uint32_t tid) {
    constexpr uint32_t half = HD / 2;
    if (tid < half) {
        const size_t base = (size_t)component * model_dim + (size_t)head * HD;
        const float low = bf16_to_f32(cur[base + tid]), high = bf16_to_f32(cur[base + half + tid]);
        const size_t index = (size_t)position * HD + tid;
        dst[tid] = round_bf16(low * cosines[index] - high * sines[index]);
        dst[half + tid] = round_bf16(high * cosines[index + half] + low * sines[index + half]);
    }
}
// step 1 of 2: every row's node slot receives the rotated key and the value (ancestor_attention.rs:127-136).  Written BEFORE the attention
// launch: a row may attend to the slot of an earlier row of the same call (its parent selected in the same round) exactly as the reference's
// row-by-row loop lets it; a row never attends to a later row's slot (an ancestor is selected before its descendants).
template <int HD>
__global__ void __launch_bounds__(HD) ancestor_store_kernel(uint16_t* node_kv, const uint16_t* current_qkv, const float* cosines, const float* sines,
                                                           const uint32_t* node_metadata, const uint32_t* node_indices, uint32_t rows, uint32_t node_capacity,
                                                           uint32_t num_heads) {
    __shared__ float s_k[HD];
    const uint32_t row = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, model_dim = num_heads * HD;
    const uint16_t* cur = current_qkv + (size_t)row * 3 * model_dim;
    const uint32_t position = node_metadata[(size_t)MD_DEPTH * rows + row] + 1;
    rotate_head<HD>(cur, cosines, sines, model_dim, head, 1, position, s_k, tid);
    __syncthreads();
    const uint32_t node = node_indices[row];
    if (node >= node_capacity) return;
    node_kv[(size_t)node * model_dim + head * HD + tid] = f32_to_bf16(s_k[tid]);
    node_kv[(size_t)node_capacity * model_dim + (size_t)node * model_dim + head * HD + tid] = cur[2 * (size_t)model_dim + head * HD + tid];
}
// step 2: grid (rows, heads), 4 waves; keys = prefix rows, the ancestors' node slots, the node's own rotated key; wave w takes keys w, w + 4, ...
// with the head dimension across its lanes (HD / 64 elements each); the four online-softmax states merge through LDS in wave order.
template <int HD>
__global__ void __launch_bounds__(256) ancestor_attention_kernel(const uint16_t* prefix_kv, const uint16_t* node_kv, const uint16_t* current_qkv, const float* cosines,
                                                                 const float* sines, const uint32_t* node_metadata, const uint32_t* ancestor_indices,
                                                                 const uint32_t* ancestor_counts, uint16_t* output, uint32_t rows, uint32_t prefix_length,
                                                                 uint32_t ancestor_stride, uint32_t node_capacity, float scale, uint32_t num_heads) {
    constexpr int EPL = HD / 64;
    __shared__ float s_q[HD], s_k[HD];
    __shared__ float s_o[4][HD], s_m[4], s_l[4];
    const uint32_t row = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, model_dim = num_heads * HD;
    const uint16_t* cur = current_qkv + (size_t)row * 3 * model_dim;
    const uint32_t position = node_metadata[(size_t)MD_DEPTH * rows + row] + 1;
    rotate_head<HD>(cur, cosines, sines, model_dim, head, 0, position, s_q, tid);
    rotate_head<HD>(cur, cosines, sines, model_dim, head, 1, position, s_k, tid >= 64 ? tid - 64 : HD); // wave 1 rotates the key (HD / 2 <= 64 pairs)
    __syncthreads();
    const uint32_t count = ancestor_counts[row], length = prefix_length + count + 1;
    float q[EPL], o[EPL], m = -INFINITY, l = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[e] = scale * s_q[lane * EPL + e], o[e] = 0.f;
    for (uint32_t i = wave; i < length; i += 4) {
        float kf[EPL], vf[EPL];
        if (i + 1 == length) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) kf[e] = s_k[lane * EPL + e], vf[e] = bf16_to_f32(cur[2 * (size_t)model_dim + head * HD + lane * EPL + e]);
        } else {
            const uint16_t *kp, *vp;
            if (i < prefix_length) {
                kp = prefix_kv + (size_t)i * model_dim, vp = prefix_kv + (size_t)prefix_length * model_dim + (size_t)i * model_dim;
            } else {
                const uint32_t anc = min(ancestor_indices[(size_t)row * ancestor_stride + (i - prefix_length)], node_capacity - 1);
                kp = node_kv + (size_t)anc * model_dim, vp = node_kv + (size_t)node_capacity * model_dim + (size_t)anc * model_dim;
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) kf[e] = bf16_to_f32(kp[head * HD + lane * EPL + e]), vf[e] = bf16_to_f32(vp[head * HD + lane * EPL + e]);
        }
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) part = fmaf(q[e], kf[e], part);
        const float score = wave_sum(part);
        const float m_new = fmaxf(m, score), factor = fast_exp(m - m_new), p = fast_exp(score - m_new);
        l = l * factor + p, m = m_new;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = o[e] * factor + p * vf[e];
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) s_o[wave][lane * EPL + e] = o[e];
    if (lane == 0) s_m[wave] = m, s_l[wave] = l;
    __syncthreads();
    if (tid < HD) {
        const float mm = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float lt = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = s_m[w] == -INFINITY ? 0.f : fast_exp(s_m[w] - mm);
            lt += s_l[w] * f, acc += s_o[w][tid] * f;
        }
        output[(size_t)row * model_dim + head * HD + tid] = f32_to_bf16(acc / lt);
    }
}

// Reference-order form (UZU_HIP_EXACT): the CPU kernel gathers the row's keys / values and runs attention_single_pass.rs:37-127 on them -- per (row, head) one
// sequential dot product per key (j ascending, multiply then add), one online-softmax chain over the keys in order with glibc-exact exp, o_j updated per key.
// Workgroup per (row, head), HD threads: thread t computes the scores of keys t, t + HD, ... (each sequential over j), then every thread runs the same chain
// and owns output element j.  Bit-identical to the CPU kernel.
constexpr uint32_t kExactMaxKeys = 15360; // the scores of one (row, head) live in dynamic LDS (checked by the launcher)
template <int HD>
__global__ void __launch_bounds__(HD) ancestor_attention_exact_kernel(const uint16_t* prefix_kv, const uint16_t* node_kv, const uint16_t* current_qkv, const float* cosines,
                                                                      const float* sines, const uint32_t* node_metadata, const uint32_t* ancestor_indices,
                                                                      const uint32_t* ancestor_counts, uint16_t* output, uint32_t rows, uint32_t prefix_length,
                                                                      uint32_t ancestor_stride, uint32_t node_capacity, float scale, uint32_t num_heads) {
    __shared__ float s_q[HD], s_k[HD];
    extern __shared__ float s_score[];
    const uint32_t row = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, model_dim = num_heads * HD;
    const uint16_t* cur = current_qkv + (size_t)row * 3 * model_dim;
    const uint32_t position = node_metadata[(size_t)MD_DEPTH * rows + row] + 1;
    rotate_head<HD>(cur, cosines, sines, model_dim, head, 0, position, s_q, tid);
    rotate_head<HD>(cur, cosines, sines, model_dim, head, 1, position, s_k, tid >= HD / 2 ? tid - HD / 2 : HD);
    __syncthreads();
    s_q[tid] = scale * s_q[tid];
    __syncthreads();
    const uint32_t count = ancestor_counts[row], length = prefix_length + min(count, ancestor_stride) + 1;
    auto key_row = [&](uint32_t i) -> const uint16_t* {
        if (i < prefix_length) return prefix_kv + (size_t)i * model_dim + head * HD;
        const uint32_t anc = min(ancestor_indices[(size_t)row * ancestor_stride + (i - prefix_length)], node_capacity - 1);
        return node_kv + (size_t)anc * model_dim + head * HD;
    };
    for (uint32_t i = tid; i < length; i += HD) {
        float score = 0.0f;
        if (i + 1 == length) {
            for (uint32_t j = 0; j < HD; ++j) score += s_q[j] * s_k[j];
        } else {
            const uint16_t* kp = key_row(i);
            for (uint32_t j = 0; j < HD; ++j) score += s_q[j] * bf16_to_f32(kp[j]);
        }
        s_score[i] = score;
    }
    __syncthreads();
    float max_score = -INFINITY, sum_exp_score = 0.0f, o = 0.0f;
    for (uint32_t i = 0; i < length; ++i) {
        const float score = s_score[i];
        const float new_max = fmaxf(max_score, score);
        const float factor = expf_glibc(max_score - new_max);
        const float exp_score = expf_glibc(score - new_max);
        max_score = new_max;
        sum_exp_score = sum_exp_score * factor + exp_score;
        float v;
        if (i + 1 == length) v = bf16_to_f32(cur[2 * (size_t)model_dim + head * HD + tid]);
        else if (i < prefix_length) v = bf16_to_f32(prefix_kv[(size_t)prefix_length * model_dim + (size_t)i * model_dim + head * HD + tid]);
        else {
            const uint32_t anc = min(ancestor_indices[(size_t)row * ancestor_stride + (i - prefix_length)], node_capacity - 1);
            v = bf16_to_f32(node_kv[(size_t)node_capacity * model_dim + (size_t)anc * model_dim + head * HD + tid]);
        }
        o = o * factor + exp_score * v;
    }
    output[(size_t)row * model_dim + head * HD + tid] = f32_to_bf16(o / sum_exp_score);
}

// ------------------------------------------------------------------------------------------------ WeaverFrontierSelect
// One workgroup.  Per node: every thread scans its slots (tid, tid + 256, ...) for the best (key, parent, token, slot) under the reference's
// rule -- key descending, parent ascending, token ascending; the sequential scan keeps the FIRST of fully equal slots, i.e. the lower slot --
// an LDS tree reduction under the same total order picks the winner, thread 0 .. do the node's bookkeeping, the winner leaves the frontier.
struct Pick {
    uint32_t key, parent, token, slot;
};
__device__ __forceinline__ bool better(const Pick& a, const Pick& b) { // a beats b
    if (a.key != b.key) return a.key > b.key;
    if (a.parent != b.parent) return a.parent < b.parent;
    if (a.token != b.token) return a.token < b.token;
    return a.slot < b.slot;
}
__global__ void __launch_bounds__(256) weaver_frontier_select_kernel(uint32_t* frontier, uint32_t* packed_tree, uint32_t* slot_ancestors, uint32_t* node_token_ids,
                                                                     uint32_t* node_metadata, uint32_t* node_ancestor_indices, uint32_t* node_valid,
                                                                     const uint32_t* candidate_pool_ids, const float* candidate_pool_logits,
                                                                     uint32_t* node_candidate_ids, float* node_candidate_logits, uint32_t fc, uint32_t ts, uint32_t nc,
                                                                     uint32_t batch_start_slot, uint32_t as, uint32_t lookahead_count, uint32_t cdc, uint32_t cpd) {
    __shared__ Pick s_pick[256];
    __shared__ uint32_t s_active[kMaxSlots]; // the Active column: updated in LDS between picks, written back at the end
    const uint32_t tid = threadIdx.x;
    for (uint32_t s = tid; s < fc; s += 256) s_active[s] = frontier[FR_ACTIVE * fc + s];
    __syncthreads();
    for (uint32_t node = 0; node < nc; ++node) {
        // the initial state of the reference's scan: (0, NO_WINNER, NO_WINNER) with no winner -- a slot must BEAT it to be picked
        Pick best{0u, kNoWinner, kNoWinner, kNoWinner};
        for (uint32_t s = tid; s < fc; s += 256) {
            if (s_active[s] == 0) continue;
            const uint32_t expandable = frontier[FR_DEPTH * fc + s] < lookahead_count ? 1u : 0u;
            const Pick next{(expandable << 31) | (frontier[FR_SCORE_KEY * fc + s] >> 1), frontier[FR_PARENT * fc + s], frontier[FR_TOKEN * fc + s], s};
            if (better(next, best)) best = next;
        }
        s_pick[tid] = best;
        __syncthreads();
        for (uint32_t stride = 128; stride > 0; stride >>= 1) {
            if (tid < stride && better(s_pick[tid + stride], s_pick[tid])) s_pick[tid] = s_pick[tid + stride];
            __syncthreads();
        }
        const Pick win = s_pick[0];
        // (a slot equal to the initial state in key / parent / token cannot win in the reference either: the comparison is strict)
        const bool real = win.slot != kNoWinner && (win.key != 0u || win.parent != kNoWinner || win.token != kNoWinner);
        const uint32_t w = real ? win.slot : 0u, tree_slot = batch_start_slot + node;
        const uint32_t tok = real ? frontier[FR_TOKEN * fc + w] : 0u, depth = real ? frontier[FR_DEPTH * fc + w] : 0u;
        const uint32_t parent_slot = (real && win.parent < ts) ? win.parent : 0u;
        if (tid == 0) {
            packed_tree[TR_TOKEN * ts + tree_slot] = tok;
            packed_tree[TR_PARENT * ts + tree_slot] = real ? win.parent : kNoWinner;
            packed_tree[TR_DEPTH * ts + tree_slot] = depth;
            packed_tree[TR_PATH_LOGPROB * ts + tree_slot] = real ? frontier[FR_PATH_LOGPROB * fc + w] : 0u;
            packed_tree[TR_EDGE_LOGPROB * ts + tree_slot] = real ? frontier[FR_EDGE_LOGPROB * fc + w] : 0u;
            packed_tree[TR_VALID * ts + tree_slot] = real ? 1u : 0u;
            if (real) s_active[w] = 0u;
            node_token_ids[node] = tok;
            const bool expandable = depth < lookahead_count;
            node_metadata[MD_DEPTH * nc + node] = expandable ? depth : 0u; // PADDING_DEPTH
            node_metadata[MD_ANCESTOR_COUNT * nc + node] = depth;
            node_metadata[MD_TREE_SLOT * nc + node] = tree_slot;
            node_valid[node] = (real && expandable) ? 1u : 0u;
        }
        // ancestors = the parent's ancestors + the parent (a later node of this call may read this slot's row: written before the barrier)
        for (uint32_t index = tid; index < as; index += 256) {
            uint32_t ancestor = 0u;
            if (real && index + 1 < depth) ancestor = slot_ancestors[(size_t)parent_slot * as + index];
            else if (real && index + 1 == depth) ancestor = parent_slot;
            slot_ancestors[(size_t)tree_slot * as + index] = ancestor;
            node_ancestor_indices[(size_t)node * as + index] = ancestor;
        }
        if (depth < cdc)
            for (uint32_t c = tid; c < cpd; c += 256) {
                node_candidate_ids[(size_t)node * cpd + c] = candidate_pool_ids[(size_t)depth * cpd + c];
                node_candidate_logits[(size_t)node * cpd + c] = candidate_pool_logits[(size_t)depth * cpd + c];
            }
        __threadfence_block();
        __syncthreads();
    }
    for (uint32_t s = tid; s < fc; s += 256) frontier[FR_ACTIVE * fc + s] = s_active[s];
}

// ------------------------------------------------------------------------------------------------ WeaverFrontierInsertChildren
__global__ void __launch_bounds__(256) weaver_frontier_insert_children_kernel(const uint32_t* packed_tree, const uint32_t* node_metadata, const uint32_t* node_valid,
                                                                              const uint32_t* child_ids, const float* child_logprobs, uint32_t* frontier, uint32_t fc,
                                                                              uint32_t ts, uint32_t nc, uint32_t ew) {
    const uint32_t index = blockIdx.x * 256 + threadIdx.x;
    if (index >= nc * ew) return;
    const uint32_t row = index / ew;
    if (node_valid[row] == 0) return;
    const uint32_t parent = node_metadata[MD_TREE_SLOT * nc + row];
    const uint64_t slot64 = (uint64_t)parent * ew + index % ew;
    if (parent >= ts || slot64 >= fc) return;
    const uint32_t slot = (uint32_t)slot64;
    const float logprob = child_logprobs[index];
    const float cumulative = bits_to_f32(packed_tree[TR_PATH_LOGPROB * ts + parent]) + logprob;
    const uint32_t cb = f32_to_bits(cumulative);
    frontier[FR_TOKEN * fc + slot] = child_ids[index];
    frontier[FR_PARENT * fc + slot] = parent;
    frontier[FR_DEPTH * fc + slot] = packed_tree[TR_DEPTH * ts + parent] + 1;
    frontier[FR_PATH_LOGPROB * fc + slot] = cb;
    frontier[FR_EDGE_LOGPROB * fc + slot] = f32_to_bits(logprob);
    frontier[FR_SCORE_KEY * fc + slot] = (cb & 0x80000000u) == 0 ? (cb ^ 0x80000000u) : ~cb; // top_k_score_key
    frontier[FR_ACTIVE * fc + slot] = 1u;
}

// ------------------------------------------------------------------------------------------------ WeaverTopChildren
// One workgroup per node.  logits = candidate + residual; thread 0 takes the log-sum-exp IN INDEX ORDER with the glibc exp / log (the
// reference's f32 sum; bit-identical log-probabilities), every thread adds the Gumbel noise of its candidates' token ids, and a candidate's
// rank in the order (perturbed descending under f32::total_cmp, token ascending, index ascending = the stable sort) is counted directly.
__device__ __forceinline__ int32_t total_key(float v) {
    const int32_t bits = (int32_t)f32_to_bits(v);
    return bits ^ (int32_t)((uint32_t)(bits >> 31) >> 1);
}
__global__ void __launch_bounds__(256) weaver_top_children_kernel(const uint16_t* residual_logits, const float* candidate_logits, const uint32_t* candidate_ids,
                                                                  const uint64_t* depth_seeds, const uint32_t* node_metadata, uint32_t* output_token_ids,
                                                                  float* output_model_logprobs, uint32_t rows, uint32_t candidates, uint32_t expand_width,
                                                                  uint32_t vocab_size) {
    __shared__ float s_logit[kCandidatesMax];
    __shared__ int32_t s_key[kCandidatesMax];
    __shared__ uint32_t s_tok[kCandidatesMax];
    __shared__ float s_log_sum;
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)row * candidates;
    const uint64_t seed = depth_seeds[node_metadata[(size_t)MD_DEPTH * rows + row]];
    for (uint32_t i = tid; i < candidates; i += 256) {
        const float lg = candidate_logits[base + i] + bf16_to_f32(residual_logits[base + i]);
        const uint32_t tok = candidate_ids[base + i];
        s_logit[i] = lg, s_tok[i] = tok;
        s_key[i] = total_key(lg + gumbel_of(seed, tok, vocab_size));
    }
    __syncthreads();
    if (tid == 0) {
        float mx = -INFINITY;
        for (uint32_t i = 0; i < candidates; ++i) mx = fmaxf(mx, s_logit[i]);
        float sum = 0.0f;
        for (uint32_t i = 0; i < candidates; ++i) sum += expf_glibc(s_logit[i] - mx);
        s_log_sum = logf_glibc(sum) + mx;
    }
    __syncthreads();
    for (uint32_t i = tid; i < candidates; i += 256) {
        const int32_t ki = s_key[i];
        const uint32_t ti = s_tok[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < candidates; ++j) {
            const int32_t kj = s_key[j];
            const uint32_t tj = s_tok[j];
            rank += (kj > ki || (kj == ki && (tj < ti || (tj == ti && j < i)))) ? 1u : 0u;
        }
        if (rank < expand_width) {
            output_token_ids[(size_t)row * expand_width + rank] = ti;
            output_model_logprobs[(size_t)row * expand_width + rank] = s_logit[i] - s_log_sum;
        }
    }
}
} // namespace

uzu_status ancestor_attention(hipStream_t s, const uint16_t* prefix_kv, uint16_t* node_kv, const uint16_t* current_qkv, const float* cosines, const float* sines,
                              const uint32_t* node_metadata, const uint32_t* ancestor_indices, const uint32_t* ancestor_counts, const uint32_t* node_indices,
                              uint16_t* output, uint32_t rows, uint32_t prefix_length, uint32_t ancestor_stride, uint32_t node_capacity, uint32_t max_depth, float scale,
                              uint32_t num_heads, uint32_t head_dim) {
    (void)max_depth; // the reference asserts depth < max_depth (the RoPE tables hold max_depth + 1 rows); the caller sizes them
    if (!rows || !num_heads) return UZU_OK;
    if (head_dim != 128) {
        set_error("ancestor_attention: variants are HEAD_DIM = 128 (got %u)", head_dim);
        return UZU_ERR_UNSUPPORTED;
    }
    if (node_capacity > 0)
        UZU_PROPAGATE(launch_check([&] {
            hipLaunchKernelGGL(ancestor_store_kernel<128>, dim3(rows, num_heads), dim3(128), 0, s, node_kv, current_qkv, cosines, sines, node_metadata, node_indices, rows,
                               node_capacity, num_heads);
        }, "ancestor_store"));
    if (exact_mode()) {
        if (prefix_length + ancestor_stride + 1 > kExactMaxKeys) {
            set_error("ancestor_attention: reference-order form holds %u keys (prefix %u + ancestors %u + 1)", kExactMaxKeys, prefix_length, ancestor_stride);
            return UZU_ERR_UNSUPPORTED;
        }
        return launch_check([&] {
            hipLaunchKernelGGL(ancestor_attention_exact_kernel<128>, dim3(rows, num_heads), dim3(128), (prefix_length + ancestor_stride + 1) * sizeof(float), s, prefix_kv, node_kv, current_qkv, cosines, sines, node_metadata,
                               ancestor_indices, ancestor_counts, output, rows, prefix_length, ancestor_stride, node_capacity ? node_capacity : 1u, scale, num_heads);
        }, "ancestor_attention_exact");
    }
    return launch_check([&] {
        hipLaunchKernelGGL(ancestor_attention_kernel<128>, dim3(rows, num_heads), dim3(256), 0, s, prefix_kv, node_kv, current_qkv, cosines, sines, node_metadata,
                           ancestor_indices, ancestor_counts, output, rows, prefix_length, ancestor_stride, node_capacity ? node_capacity : 1u, scale, num_heads);
    }, "ancestor_attention");
}

uzu_status weaver_frontier_select(hipStream_t s, uint32_t* frontier, uint32_t* packed_tree, uint32_t* slot_ancestors, uint32_t* node_token_ids, uint32_t* node_metadata,
                                  uint32_t* node_ancestor_indices, uint32_t* node_valid, const uint32_t* candidate_pool_ids, const float* candidate_pool_logits,
                                  uint32_t* node_candidate_ids, float* node_candidate_logits, uint32_t frontier_capacity, uint32_t tree_slot_count, uint32_t node_count,
                                  uint32_t batch_start_slot, uint32_t ancestor_stride, uint32_t max_depth, uint32_t lookahead_count, uint32_t candidate_depth_count,
                                  uint32_t candidates_per_depth) {
    // the reference kernel's own guard (weaver_frontier_select.rs:30-42): outside it the call is a no-op
    if (frontier_capacity == 0 || frontier_capacity > kMaxSlots || node_count == 0 || node_count > kMaxWidth || ancestor_stride == 0 || max_depth == 0 ||
        tree_slot_count == 0 || batch_start_slot + node_count > tree_slot_count || candidate_depth_count == 0 || candidates_per_depth == 0)
        return UZU_OK;
    return launch_check([&] {
        hipLaunchKernelGGL(weaver_frontier_select_kernel, dim3(1), dim3(256), 0, s, frontier, packed_tree, slot_ancestors, node_token_ids, node_metadata, node_ancestor_indices,
                           node_valid, candidate_pool_ids, candidate_pool_logits, node_candidate_ids, node_candidate_logits, frontier_capacity, tree_slot_count, node_count,
                           batch_start_slot, ancestor_stride, lookahead_count, candidate_depth_count, candidates_per_depth);
    }, "weaver_frontier_select");
}

uzu_status weaver_frontier_insert_children(hipStream_t s, const uint32_t* packed_tree, const uint32_t* node_metadata, const uint32_t* node_valid, const uint32_t* child_ids,
                                           const float* child_logprobs, uint32_t* frontier, uint32_t frontier_capacity, uint32_t tree_slot_count, uint32_t node_count,
                                           uint32_t expand_width) {
    if (frontier_capacity == 0 || tree_slot_count == 0 || expand_width == 0 || node_count == 0) return UZU_OK;
    const uint32_t total = node_count * expand_width;
    return launch_check([&] {
        hipLaunchKernelGGL(weaver_frontier_insert_children_kernel, dim3((total + 255) / 256), dim3(256), 0, s, packed_tree, node_metadata, node_valid, child_ids, child_logprobs,
                           frontier, frontier_capacity, tree_slot_count, node_count, expand_width);
    }, "weaver_frontier_insert_children");
}

uzu_status weaver_top_children(hipStream_t s, const uint16_t* residual_logits, const float* candidate_logits, const uint32_t* candidate_ids, const uint64_t* depth_seeds,
                               const uint32_t* node_metadata, uint32_t* output_token_ids, float* output_model_logprobs, uint32_t rows, uint32_t candidates,
                               uint32_t expand_width, uint32_t vocab_size) {
    if (candidates == 0 || candidates > kCandidatesMax || expand_width == 0 || expand_width > candidates || rows == 0) return UZU_OK; // weaver_top_children.rs:27-29
    return launch_check([&] {
        hipLaunchKernelGGL(weaver_top_children_kernel, dim3(rows), dim3(256), 0, s, residual_logits, candidate_logits, candidate_ids, depth_seeds, node_metadata,
                           output_token_ids, output_model_logprobs, rows, candidates, expand_width, vocab_size);
    }, "weaver_top_children");
}


// ---------------------------------------------------------------------------------------------- RadixTopKSmall
// cpu/kernel/radix_top_k_small.rs:25-79: per row the k (<= 512) best of `columns` f32 values, ordered by (value descending under f32::total_cmp, column ascending).
// One 1024-thread workgroup per row: the key of a value = its total-order integer; four 8-bit radix passes find the k-th largest key T and how many keys are larger;
// the winners are every key > T plus the LOWEST-column keys == T (ordered compaction, chunk by chunk), then a bitonic sort of <= 512 (key, ~column) pairs in LDS.
// Integer work throughout: exact.
namespace {
__device__ __forceinline__ uint32_t topk_key(float v) { // larger float (total order) <=> larger unsigned key
    const uint32_t b = f32_to_bits(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ void __launch_bounds__(1024) radix_top_k_small_kernel(const float* input, uint32_t* output_ids, float* output_scores, uint32_t columns, uint32_t k) {
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_prefix, s_need, s_count, s_wave[16], s_taken;
    __shared__ unsigned long long s_items[512];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = input + (size_t)blockIdx.x * columns;
    if (tid == 0) s_prefix = 0, s_need = k, s_count = 0, s_taken = 0;
    for (uint32_t i = tid; i < 512; i += 1024) s_items[i] = 0ull;
    __syncthreads();
    // radix select: after pass p the top 8 (p + 1) bits of T are known; s_need = winners still to be found among the keys that match the prefix
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t c = tid; c < columns; c += 1024) {
            const uint32_t key = topk_key(row[c]);
            if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t need = s_need, bin = 255;
            for (;; --bin) { // from the largest digit down: the bin in which the need-th winner lies
                const uint32_t h = s_hist[bin];
                if (h >= need || bin == 0) break;
                need -= h;
            }
            s_need = need;
            s_prefix = prefix | (bin << shift);
        }
        __syncthreads();
    }
    const uint32_t T = s_prefix, need_eq = s_need; // keys > T are all winners; need_eq of the keys == T (lowest columns first)
    // winners above the threshold: any order (they are sorted below)
    for (uint32_t c = tid; c < columns; c += 1024) {
        const uint32_t key = topk_key(row[c]);
        if (key > T) {
            const uint32_t slot = atomicAdd(&s_count, 1u);
            if (slot < 512) s_items[slot] = ((unsigned long long)key << 32) | (uint32_t)~c;
        }
    }
    __syncthreads();
    const uint32_t above = s_count;
    // the threshold's own keys in column order: ranks inside a 1024-column chunk from ballots + wave totals
    for (uint32_t start = 0; start < columns && s_taken < need_eq; start += 1024) {
        const uint32_t c = start + tid;
        const bool mine = c < columns && topk_key(row[c]) == T;
        const unsigned long long ballot = __ballot(mine);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(ballot);
        __syncthreads();
        uint32_t base = s_taken, total = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            if (w < wave) base += s_wave[w];
            total += s_wave[w];
        }
        const uint32_t rank = base + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
        if (mine && rank < need_eq && above + rank < 512) s_items[above + rank] = ((unsigned long long)T << 32) | (uint32_t)~c;
        __syncthreads();
        if (tid == 0) s_taken += total;
        __syncthreads();
    }
    // bitonic sort, descending, of 512 (key, ~column) pairs (unused slots are 0: they sink to the end)
    for (uint32_t size = 2; size <= 512; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < 256) {
                const uint32_t lo = 2 * tid - (tid & (stride - 1)), hi = lo + stride;
                const bool descending = (lo & size) == 0;
                const unsigned long long a = s_items[lo], b = s_items[hi];
                if ((a < b) == descending) s_items[lo] = b, s_items[hi] = a;
            }
            __syncthreads();
        }
    for (uint32_t r = tid; r < k; r += 1024) {
        const uint32_t c = ~(uint32_t)(s_items[r] & 0xFFFFFFFFull);
        output_ids[(size_t)blockIdx.x * k + r] = c;
        output_scores[(size_t)blockIdx.x * k + r] = row[c];
    }
}
} // namespace
uzu_status radix_top_k_small(hipStream_t s, const float* input, uint32_t* output_ids, float* output_scores, uint32_t rows, uint32_t columns, uint32_t k) {
    if (!rows) return UZU_OK;
    if (!k || k > 512 || k > columns) {
        set_error("radix_top_k_small: k %u outside 1..min(512, columns %u)", k, columns);
        return UZU_ERR_INVALID_ARGUMENT;
    }
    return launch_check([&] { hipLaunchKernelGGL(radix_top_k_small_kernel, dim3(rows), dim3(1024), 0, s, input, output_ids, output_scores, columns, k); }, "radix_top_k_small");
}

} // namespace k
} // namespace uzu
