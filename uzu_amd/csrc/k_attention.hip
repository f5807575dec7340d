// k_attention.hip -- scaled-dot-product attention over the KV cache for gfx950.
//
// Reference semantics: BU/cpu/kernel/attention/attention_single_pass.rs:37-127,
// attention_two_pass.rs:41-190, mask.rs:3-61 (non-trie).  Dispatch (single vs two pass) is the
// caller's: BU/../encodable_block/mixer/attention/core/mod.rs:81-93.
//
// Design: the KV cache is read ONCE per kv-head -- one workgroup handles all `GS`
// query heads that share the kv-head (GQA), unlike the reference kernels which re-read K/V per query
// head.  A workgroup = 4 waves; every wave is split into key groups of hd/8 lanes, each lane owning
// 8 contiguous head-dim elements (one 16-byte load per K row and per V row => coalesced 128..512 B
// row reads).  Each key group runs an online softmax over its own keys (f32 running max / sum /
// output in registers); groups are merged by shuffles inside a wave and through LDS across waves with
// a fixed order (deterministic).  `num_blocks` = 1 gives the single-pass kernel, 32 the first pass of
// the reference's split-KV scheme: block b owns keys b, b+32, ... exactly as the reference does, so
// the (partials, sums, maxs) buffers have the reference's meaning and pass 2 can be bit-exact.
#include "device_utils.h"
#include "kernels.h"

#include "attention_mask.h"

namespace uzu {
namespace k {

template <class T> __device__ __forceinline__ void load8(const T* p, float (&f)[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&f)[8]) {
    const uint4 u = *(const uint4*)p;
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = bits_to_f32(w[j] << 16);
        f[2 * j + 1] = bits_to_f32(w[j] & 0xFFFF0000u);
    }
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&f)[8]) {
    const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
    f[0] = a.x, f[1] = a.y, f[2] = a.z, f[3] = a.w, f[4] = b.x, f[5] = b.y, f[6] = b.z, f[7] = b.w;
}

// merge online-softmax state (m2,l2,o2) into (m,l,o)
template <int N> __device__ __forceinline__ void merge_state(float& m, float& l, float (&o)[N], float m2, float l2, const float (&o2)[N]) {
    const float nm = fmaxf(m, m2);
    const float f1 = (m == -INFINITY) ? 0.f : fast_exp(m - nm);
    const float f2 = (m2 == -INFINITY) ? 0.f : fast_exp(m2 - nm);
    l = l * f1 + l2 * f2;
#pragma unroll
    for (int e = 0; e < N; ++e) o[e] = o[e] * f1 + o2[e] * f2;
    m = nm;
}

// grid: (kv_head * head_subgroups + sub, block, q_seq_idx); 256 threads
template <class T, int HD, int GS>
__global__ void __launch_bounds__(256) attention_block_kernel(AttentionParams a_in, uint32_t num_blocks, float init_max,
                                                              T* out, float* partials, float* sums, float* maxs) {
    AttentionParams a = a_in;
    attention_resolve_dyn(a); // context length / ring parameters from device memory (graph replay)
    constexpr int LPK = HD / 8;  // lanes per key
    constexpr int KG = 64 / LPK; // key groups per wave
    constexpr int NGRP = 4 * KG; // key groups per workgroup
    __shared__ float s_o[4][GS][HD];
    __shared__ float s_m[4][GS], s_l[4][GS];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kgrp = lane / LPK, sl = lane % LPK;
    const uint32_t subs = a.gqa_factor / GS;
    const uint32_t kv_head_idx = blockIdx.x / subs, sub = blockIdx.x % subs;
    const uint32_t head0 = kv_head_idx * a.gqa_factor + sub * GS;
    const uint32_t block_idx = blockIdx.y, q_seq_idx = blockIdx.z;
    const uint32_t sequence_length = a.sequence_length;
    const uint32_t prefix_length = sequence_length - a.suffix_length;
    const uint32_t suffix_position = a.is_kv_cache_ring ? a.ring_length : prefix_length;
    const uint32_t query_position = attention_query_position(a, suffix_position, q_seq_idx);

    const T* queries = (const T*)a.queries;
    const T* keys = (const T*)a.keys + (size_t)kv_head_idx * a.k_head_stride + sl * 8;
    const T* values = (const T*)a.values + (size_t)kv_head_idx * a.v_head_stride + sl * 8;

    float q[GS][8], o[GS][8], mx[GS], sm[GS];
#pragma unroll
    for (int g = 0; g < GS; ++g) {
        const size_t q_offset = (size_t)(head0 + g) * a.suffix_length + q_seq_idx;
        float t[8];
        load8<T>(queries + q_offset * HD + sl * 8, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[g][e] = a.scale * t[e], o[g][e] = 0.f;
        mx[g] = init_max;
        sm[g] = 0.f;
    }
    const uint32_t my_group = wave * KG + kgrp;
    if (a.sinks && block_idx == 0 && my_group == 0) {
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            mx[g] = ld((const T*)a.sinks, head0 + g);
            sm[g] = 1.0f;
        }
    }
    for (uint32_t i = block_idx + num_blocks * my_group; i < sequence_length; i += num_blocks * NGRP) {
        if (!should_use_key(a, q_seq_idx, prefix_length, suffix_position, query_position, i)) continue;
        float kf[8], vf[8];
        load8<T>(keys + (size_t)i * a.k_seq_stride, kf);
        load8<T>(values + (size_t)i * a.v_seq_stride, vf);
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) part = fmaf(q[g][e], kf[e], part);
            const float score = group_sum<LPK>(part);
            const float new_max = fmaxf(mx[g], score);
            const float factor = fast_exp(mx[g] - new_max);
            const float exp_score = fast_exp(score - new_max);
            mx[g] = new_max;
            sm[g] = sm[g] * factor + exp_score;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[g][e] = o[g][e] * factor + exp_score * vf[e];
        }
    }
    // merge the KG key groups of this wave (butterfly over the group index)
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            const float m2 = __shfl_xor(mx[g], off, 64), l2 = __shfl_xor(sm[g], off, 64);
            float o2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o2[e] = __shfl_xor(o[g][e], off, 64);
            // symmetric merge: both partners compute the same (max, sum, out)
            const bool lower = (lane & off) == 0;
            float ma = lower ? mx[g] : m2, la = lower ? sm[g] : l2, mb = lower ? m2 : mx[g], lb = lower ? l2 : sm[g];
            float oa[8], ob[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) oa[e] = lower ? o[g][e] : o2[e], ob[e] = lower ? o2[e] : o[g][e];
            merge_state<8>(ma, la, oa, mb, lb, ob);
            mx[g] = ma, sm[g] = la;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[g][e] = oa[e];
        }
    }
    if (kgrp == 0) {
#pragma unroll
        for (int g = 0; g < GS; ++g) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s_o[wave][g][sl * 8 + e] = o[g][e];
            if (sl == 0) s_m[wave][g] = mx[g], s_l[wave][g] = sm[g];
        }
    }
    __syncthreads();
    // final merge over the 4 waves, in wave order, one thread per (head, element)
    for (int idx = threadIdx.x; idx < GS * HD; idx += 256) {
        const int g = idx / HD, e = idx % HD;
        float m = s_m[0][g];
#pragma unroll
        for (int w = 1; w < 4; ++w) m = fmaxf(m, s_m[w][g]);
        float l = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = (s_m[w][g] == -INFINITY) ? 0.f : fast_exp(s_m[w][g] - m);
            l += s_l[w][g] * f;
            acc += s_o[w][g][e] * f;
        }
        const size_t o_offset = (size_t)q_seq_idx * a.num_heads + head0 + g;
        if (partials) {
            partials[(o_offset * num_blocks + block_idx) * HD + e] = acc;
            if (e == 0) {
                sums[o_offset * num_blocks + block_idx] = l;
                maxs[o_offset * num_blocks + block_idx] = m;
            }
        } else {
            st(out, o_offset * HD + e, acc / l);
        }
    }
}

template <class T, int HD>
static uzu_status launch_block(hipStream_t s, const AttentionParams& a, uint32_t num_blocks, float init_max, void* out,
                               float* partials, float* sums, float* maxs) {
    const uint32_t kv_heads = a.num_heads / a.gqa_factor;
    const uint32_t cap = HD >= 256 ? 4 : 8;
    uint32_t gs = 1;
    for (uint32_t c = cap; c >= 1; c >>= 1)
        if (a.gqa_factor % c == 0) {
            gs = c;
            break;
        }
    const dim3 grid(kv_heads * (a.gqa_factor / gs), num_blocks, a.suffix_length);
#define UZU_LAUNCH_GS(G)                                                                                              \
    return launch_check([&] {                                                                                         \
        hipLaunchKernelGGL((attention_block_kernel<T, HD, G>), grid, dim3(256), 0, s, a, num_blocks, init_max, (T*)out, \
                           partials, sums, maxs);                                                                     \
    }, "attention_block")
    switch (gs) {
    case 8: if constexpr (HD < 256) { UZU_LAUNCH_GS(8); } [[fallthrough]];
    case 4: UZU_LAUNCH_GS(4);
    case 2: UZU_LAUNCH_GS(2);
    default: UZU_LAUNCH_GS(1);
    }
#undef UZU_LAUNCH_GS
}

template <class T>
static uzu_status dispatch_hd(hipStream_t s, const AttentionParams& a, uint32_t num_blocks, float init_max, void* out,
                              float* partials, float* sums, float* maxs) {
    switch (a.head_dim) {
    case 64: return launch_block<T, 64>(s, a, num_blocks, init_max, out, partials, sums, maxs);
    case 128: return launch_block<T, 128>(s, a, num_blocks, init_max, out, partials, sums, maxs);
    case 256: return launch_block<T, 256>(s, a, num_blocks, init_max, out, partials, sums, maxs);
    case 512: return launch_block<T, 512>(s, a, num_blocks, init_max, out, partials, sums, maxs);
    default:
        set_error("attention: unsupported head_dim %u (64/128/256/512)", a.head_dim);
        return UZU_ERR_UNSUPPORTED;
    }
}

static uzu_status check_attention(const AttentionParams& a) {
    if (a.gqa_factor == 0 || a.num_heads % a.gqa_factor != 0) {
        set_error("attention: num_heads %u not divisible by gqa_factor %u", a.num_heads, a.gqa_factor);
        return UZU_ERR_INVALID_ARGUMENT;
    }
    if (a.k_head_stride % 8 || a.k_seq_stride % 8 || a.v_head_stride % 8 || a.v_seq_stride % 8) {
        set_error("attention: K/V strides must be multiples of 8 elements");
        return UZU_ERR_UNSUPPORTED;
    }
    return UZU_OK;
}

uzu_status attention_single_pass(hipStream_t s, const AttentionParams& a, void* out) {
    if (!a.suffix_length || !a.num_heads) return UZU_OK;
    UZU_PROPAGATE(check_attention(a));
    if (exact_mode()) return attention_single_pass_exact(s, a, out);
    if (attention_prefill_mfma_supported(a)) return attention_prefill_mfma(s, a, out); // prefill-sized causal tiles
    return UZU_DISPATCH_T(a.dt, [&]() -> uzu_status { return dispatch_hd<T>(s, a, 1, -INFINITY, out, nullptr, nullptr, nullptr); });
}

uzu_status attention_two_pass1(hipStream_t s, const AttentionParams& a, float* partials, float* sums, float* maxs) {
    if (!a.suffix_length || !a.num_heads) return UZU_OK;
    UZU_PROPAGATE(check_attention(a));
    if (exact_mode()) return attention_two_pass1_exact(s, a, partials, sums, maxs);
    return UZU_DISPATCH_T(a.dt, [&]() -> uzu_status { return dispatch_hd<T>(s, a, 32, -1e9f, nullptr, partials, sums, maxs); });
}

// BU/cpu/kernel/attention/attention_two_pass.rs:154-190: same block order as the reference => bit-exact
template <class T>
__global__ void __launch_bounds__(256) attention_two_pass2_kernel(const float* partials, const float* sums,
                                                                  const float* maxs, T* out, uint32_t HD,
                                                                  uint32_t total_rows) {
    const uint32_t o_offset = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (o_offset >= total_rows) return;
    const float* mx = maxs + (size_t)o_offset * 32;
    const float* sm = sums + (size_t)o_offset * 32;
    float global_max = -INFINITY;
    for (int b = 0; b < 32; ++b) global_max = fmaxf(global_max, mx[b]);
    float global_sum = 0.f;
    float w[32];
    // lane b evaluates the weight of block b ONCE (the libm-exact exp is ~150 double-precision operations; every lane used to evaluate all 32) and the
    // wave reads it from there: same values, same summation order
    const float w_mine = expf_glibc(mx[lane & 31] - global_max);
#pragma unroll
    for (int b = 0; b < 32; ++b) {
        w[b] = __shfl(w_mine, b, 64);
        global_sum += sm[b] * w[b];
    }
    for (uint32_t j = lane; j < HD; j += 64) {
        float val = 0.f;
#pragma unroll
        for (int b = 0; b < 32; ++b) val += partials[((size_t)o_offset * 32 + b) * HD + j] * w[b];
        st(out, (size_t)o_offset * HD + j, val / global_sum);
    }
}
uzu_status attention_two_pass2(hipStream_t s, const float* partials, const float* sums, const float* maxs, void* out,
                               uint32_t dt, uint32_t head_dim, uint32_t num_heads, uint32_t suffix_length) {
    const uint32_t rows = num_heads * suffix_length;
    if (!rows) return UZU_OK;
    if (exact_mode()) return attention_two_pass2_exact(s, partials, sums, maxs, out, dt, head_dim, num_heads, suffix_length);
    return UZU_DISPATCH_T(dt, [&]() -> uzu_status {
        return launch_check([&] {
            hipLaunchKernelGGL((attention_two_pass2_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, partials, sums, maxs, (T*)out, head_dim, rows);
        }, "attention_two_pass2");
    });
}

} // namespace k
} // namespace uzu
