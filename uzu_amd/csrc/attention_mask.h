// attention_mask.h -- should_use_key (cpu/kernel/attention/mask.rs:3-61): shared by the attention kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace uzu {
namespace k {

// attention_single_pass.rs:55-61: a query of a speculated tree sits at its node's height
__device__ __forceinline__ uint32_t attention_query_position(const AttentionParams& a, uint32_t suffix_position, uint32_t q_seq_idx) {
    return suffix_position + (a.trie ? a.trie[3 * (size_t)q_seq_idx + 2] : q_seq_idx);
}

__device__ __forceinline__ bool should_use_key(const AttentionParams& a, uint32_t q_seq_idx, uint32_t prefix_length,
                                               uint32_t suffix_position, uint32_t query_position, uint32_t i) {
    bool use_key = true;
    uint32_t key_position;
    if (i >= prefix_length) {
        const uint32_t key_position_in_suffix = i - prefix_length;
        if (a.trie) { // the key's place in the speculated tree
            const uint32_t* node = a.trie + 3 * (size_t)key_position_in_suffix;
            key_position = suffix_position + node[2];
            if (a.is_causal) use_key &= q_seq_idx >= node[0] && q_seq_idx <= node[1];
        } else {
            key_position = suffix_position + key_position_in_suffix;
            if (a.is_causal) use_key &= key_position_in_suffix <= q_seq_idx;
        }
    } else {
        if (a.is_kv_cache_ring) {
            key_position = (prefix_length + i - a.ring_offset) % prefix_length;
            use_key &= key_position < a.ring_length;
        } else {
            key_position = i;
        }
    }
    if (a.is_sliding_window) {
        const uint32_t w = a.sliding_window_size;
        if (a.is_causal)
            use_key &= key_position <= query_position && (query_position - key_position) < w;
        else if (key_position <= query_position)
            use_key &= (query_position - key_position) <= w / 2;
        else
            use_key &= (key_position - query_position) <= w / 2;
    }
    return use_key;
}

} // namespace k
} // namespace uzu
