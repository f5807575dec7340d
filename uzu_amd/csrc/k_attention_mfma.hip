// k_attention_mfma.hip -- prefill attention on the bf16 matrix cores (flash-attention form) for gfx950.
//
// Reference semantics: AttentionSinglePass / TwoPass (BU/cpu/kernel/attention/attention_single_pass.rs:37-127,
// mask.rs:3-61): per (head, query) softmax(scale * q.k) over the visible keys (all cached keys + the causal part of
// the suffix), output [M, heads, hd] bf16.  The stand-alone VALU kernels (k_attention.hip) spend ~45 instructions per
// (query, key, head); at a 1024-token chunk that is the third-largest item of the prefill.  Here:
//
//   * a workgroup = one kv head x four wave tasks; a task is (32-query tile, q head of the kv head's GQA group), tasks are
//     numbered tile-major, so the four waves of a workgroup are four heads of one tile (GQA factor 4, 8, ...), two heads of two
//     tiles (factor 2), four tiles (factor 1) or whatever four consecutive tasks are (factor 3, 5, 6, 7: Qwen3-14B has 40 q /
//     8 kv heads) -- they share every K / V tile, staged once in LDS, and a wave skips the tiles past its own causal range;
//   * everything is computed TRANSPOSED so that a lane owns one query: S^T = K Q^T (A = K rows from LDS, B = Q^T from
//     registers), the C layout of v_mfma_f32_32x32x16_bf16 then gives a lane 16 key scores of ITS query (col = lane &
//     31), so the online-softmax statistics are per-lane scalars (one exchange with lane ^ 32 per tile for the
//     running maximum);  O^T = V^T P^T: the P^T operand is exactly those 16 registers (converted to bf16) -- no
//     shuffle, no LDS round trip -- and the O^T accumulator columns are again "this lane's query", so the rescale by
//     exp(m_old - m_new) and the final 1 / l are per-lane multiplies;
//   * V is stored transposed in LDS ([hd][key]): the contraction runs over keys, so a lane needs 8 keys of one hd
//     column; K stays [key][hd];
//   * q, k, v are bf16 already => every q.k product is exact in f32.  The probabilities are NOT bf16: they are fed
//     as hi + lo (two bf16 MFMAs, 16 significant bits), so the only arithmetic difference to the f32 reference is
//     the summation order.
#include <stdlib.h>

#include "device_utils.h"
#include "kernels.h"

namespace uzu {
namespace k {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

namespace {
constexpr int TK = 32; // keys per tile (one MFMA block)
constexpr int TQ = 32; // queries per wave

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
// value of lane ^ 32 (the other half of the wave): the one cross-half exchange per tile
__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32, 64); }
} // namespace

// grid: (kv heads, ceil(gqa * query tiles / tpw)); 256 threads.  Workgroups with the longest key ranges are numbered first.
// tpw = wave tasks per workgroup (4, 2 or 1): models with few heads (Qwen3.5-0.8B: 8 q heads -> 256 tasks per 1024-token chunk)
// would fill 64 CUs at four tasks per workgroup; with fewer tasks the idle waves still help staging the K / V tiles.
// Key split (few-head models: Qwen3.5-0.8B has 2 kv heads -> 64 workgroups of four tasks per 1024-token chunk): gridDim.z workgroups share a
// task group, each takes an equal share of the key tiles ITS queries can see and leaves the unnormalised O^T, the running maximum and the
// sum as f32 partials ([split][query][head]); attention_prefill_merge_kernel folds them (the AttentionTwoPass2 step of the decode path).
// GEN: the general mask of mask.rs:3-61 without the trie -- sliding window, ring KV prefix (AttentionStateType::Ring: the prefix rows are a
// ring of `prefix` slots, key position = (prefix + i - ring_offset) mod prefix, live iff < ring_length) and attention sinks (the running
// maximum starts at the head's sink logit with a unit of mass: attention_single_pass.rs:70-74) -- evaluated per key on its POSITION; the
// plain causal instantiation keeps its one comparison per key.
template <int HD, bool GEN>
__global__ void __launch_bounds__(256) attention_prefill_mfma_kernel(AttentionParams a, uint16_t* out, uint32_t tpw, float* part_o, float* part_ml) {
    constexpr int KP = HD * 2 + 16;  // K tile row pitch in bytes (conflict-free ds_read_b128)
    constexpr int VP = TK * 2 + 8;   // V^T tile row pitch in bytes: 32 keys + 8 bytes of pad (conflict-free ds_read_b64)
    constexpr int NS = HD / 16;      // k16 steps of the QK^T contraction
    constexpr int NB = HD / 32;      // 32-row blocks of O^T
    __shared__ __attribute__((aligned(16))) uint8_t s_k[2][TK * KP];
    __shared__ __attribute__((aligned(16))) uint8_t s_vt[2][HD * VP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
    const uint32_t M = a.suffix_length;
    const uint32_t kv_head = blockIdx.x;
    const uint32_t n_tasks = a.gqa_factor * ((M + TQ - 1) / TQ);
    const uint32_t task0 = (gridDim.y - 1 - blockIdx.y) * tpw; // heaviest (last query tiles) first
    const uint32_t task = task0 + wave;
    const bool wave_live = (uint32_t)wave < tpw && task < n_tasks;
    const uint32_t task_c = wave_live ? task : n_tasks - 1;
    const uint32_t head = kv_head * a.gqa_factor + task_c % a.gqa_factor;
    if constexpr (GEN) attention_resolve_dyn(a); // ring parameters / sequence length from the device-resident accepted-token count
    const uint32_t sequence_length = a.sequence_length + (a.dyn ? *a.dyn : 0u);
    const uint32_t prefix = sequence_length - M;
    const uint32_t suffix_position = (GEN && a.is_kv_cache_ring) ? a.ring_length : prefix; // attention_single_pass.rs:52-61
    const uint32_t q0 = (task_c / a.gqa_factor) * TQ; // first query of this wave
    const uint32_t qi = q0 + l32;                           // this lane's query (suffix index)
    const bool q_live = wave_live && qi < M;

    // ---- Q^T operand: lane = (query l32, hd half): 8 consecutive hd elements per k16 step, kept in registers
    u32x4_t qf[NS];
    {
        const uint16_t* qrow = (const uint16_t*)a.queries + ((size_t)head * M + (q_live ? qi : 0)) * HD + 8 * half;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint4 t = q_live ? *(const uint4*)(qrow + 16 * s) : make_uint4(0, 0, 0, 0);
            qf[s] = u32x4_t{t.x, t.y, t.z, t.w};
        }
    }

    // ---- staging role: thread -> (key of the tile, 64-byte slice of its hd row)
    const uint16_t* kbase = (const uint16_t*)a.keys + (size_t)kv_head * a.k_head_stride;
    const uint16_t* vbase = (const uint16_t*)a.values + (size_t)kv_head * a.v_head_stride;
    constexpr int SLICES = HD / 32;               // 32 hd elements (64 bytes) per slice
    constexpr int ROUNDS = (TK * SLICES + 255) / 256;
    u32x4_v kst[ROUNDS][4], vst[ROUNDS][4]; // native vectors + unconditional loads: stay in registers and in flight across lds_barrier()
    // keys visible to ANY query of the workgroup: 0 .. prefix + last query of its last live task (tasks are tile-major)
    const uint32_t last_tile = ((task0 + tpw - 1 < n_tasks ? task0 + tpw - 1 : n_tasks - 1) / a.gqa_factor + 1) * TQ;
    const uint32_t wg_last_q = (last_tile < M ? last_tile : M) - 1;
    const uint32_t key_end = prefix + wg_last_q + 1;
    const uint32_t n_tiles_all = (key_end + TK - 1) / TK;
    const uint32_t per_split = (n_tiles_all + gridDim.z - 1) / gridDim.z;
    const uint32_t t_lo = min(blockIdx.z * per_split, n_tiles_all), n_tiles = min(t_lo + per_split, n_tiles_all); // this workgroup: tiles [t_lo, n_tiles)
    auto load_tile = [&](uint32_t t) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t idx = tid + 256 * r, key = idx / SLICES, sl = idx % SLICES;
            const uint32_t i = t * TK + key;
            const bool live = idx < TK * SLICES && i < key_end;
            const uint32_t ic = i < key_end ? i : key_end - 1; // clamped row: masked keys get probability 0 anyway, zeroed for hygiene
            const u32x4_v* ks = (const u32x4_v*)(kbase + (size_t)ic * a.k_seq_stride + (sl % SLICES) * 32);
            const u32x4_v* vs = (const u32x4_v*)(vbase + (size_t)ic * a.v_seq_stride + (sl % SLICES) * 32);
            const u32x4_v zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4_v kk = ks[j], vv = vs[j];
                kst[r][j] = live ? kk : zero, vst[r][j] = live ? vv : zero;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t idx = tid + 256 * r, key = idx / SLICES, sl = idx % SLICES;
            if (idx >= TK * SLICES) continue;
            uint8_t* kd = &s_k[buf][key * KP + sl * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) *(u32x4_v*)(kd + 16 * j) = kst[r][j];
            // V transposed: element (key, hd) -> s_vt[hd][key]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w[4] = {vst[r][j][0], vst[r][j][1], vst[r][j][2], vst[r][j][3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t hd0 = sl * 32 + j * 8 + 2 * e;
                    *(uint16_t*)&s_vt[buf][hd0 * VP + key * 2] = (uint16_t)(w[e] & 0xFFFFu);
                    *(uint16_t*)&s_vt[buf][(hd0 + 1) * VP + key * 2] = (uint16_t)(w[e] >> 16);
                }
            }
        }
    };

    f32x16_t o[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f; // l_run: this lane's half of the keys only (merged at the end)
    const uint32_t my_last_key = prefix + qi; // causal: keys 0 .. prefix + qi
    if constexpr (GEN) {
        if (a.sinks && blockIdx.z == 0) { // the sink's unit of mass is counted once: by half 0 of the first key split
            m_run = bf16_to_f32(((const uint16_t*)a.sinks)[head]);
            l_run = half == 0 ? 1.0f : 0.0f;
        }
    }
    const uint32_t query_position = suffix_position + qi;

    if (t_lo < n_tiles) {
        load_tile(t_lo);
        store_tile(t_lo & 1);
    }
    __syncthreads();
    for (uint32_t t = t_lo; t < n_tiles; ++t) {
        const int buf = t & 1;
        load_tile(t + 1 < n_tiles ? t + 1 : t); // unconditional: the compiler can count the loads in flight
        // a wave whose queries all end before this tile has nothing to add (causal), but must keep the barriers
        const uint32_t wave_last_key = prefix + (q0 + TQ - 1 < M ? q0 + TQ - 1 : M - 1);
        if (wave_live && t * TK <= wave_last_key) {
            // ---- S^T = K Q^T
            f32x16_t sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const uint8_t* kt = &s_k[buf][l32 * KP + half * 16];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint4 kv = *(const uint4*)(kt + s * 32);
                const u32x4_t kfrag = {kv.x, kv.y, kv.z, kv.w};
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kfrag), __builtin_bit_cast(bf16x8_t, qf[s]), sacc, 0, 0, 0);
            }
            // ---- scale, causal mask, online softmax (lane = one query; its 16 keys: (r & 3) + 8 (r >> 2) + 4 half)
            float sc[16];
            float m_loc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t key = t * TK + (r & 3) + 8 * (r >> 2) + 4 * half;
                bool vis = q_live && key <= my_last_key;
                if constexpr (GEN) { // mask.rs:31-60 on positions
                    uint32_t key_position;
                    if (key >= prefix) {
                        key_position = suffix_position + (key - prefix);
                    } else if (a.is_kv_cache_ring) {
                        key_position = prefix + key - a.ring_offset;
                        if (key_position >= prefix) key_position -= prefix;
                        vis = vis && key_position < a.ring_length;
                    } else {
                        key_position = key;
                    }
                    if (a.is_sliding_window) vis = vis && key_position <= query_position && query_position - key_position < a.sliding_window_size;
                }
                sc[r] = vis ? a.scale * sacc[r] : -INFINITY;
                m_loc = fmaxf(m_loc, sc[r]);
            }
            const float m_new = fmaxf(m_run, fmaxf(m_loc, other_half(m_loc)));
            const float alpha = m_new == -INFINITY ? 1.0f : fast_exp(m_run - m_new); // m_run = -inf => exp(-inf) = 0
            float psum = 0.f;
            uint32_t p_hi[8], p_lo[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = m_new == -INFINITY ? 0.f : fast_exp(sc[r] - m_new), p1 = m_new == -INFINITY ? 0.f : fast_exp(sc[r + 1] - m_new);
                psum += p0 + p1;
                const float h0 = round_bf16(p0), h1 = round_bf16(p1);
                p_hi[r / 2] = pack2(h0, h1);
                p_lo[r / 2] = pack2(p0 - h0, p1 - h1);
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            // ---- O^T += V^T P^T : two k16 steps (registers 0..7 and 8..15 of the score tile), P as hi + lo
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const u32x4_t ph = {p_hi[4 * st], p_hi[4 * st + 1], p_hi[4 * st + 2], p_hi[4 * st + 3]};
                const u32x4_t pl = {p_lo[4 * st], p_lo[4 * st + 1], p_lo[4 * st + 2], p_lo[4 * st + 3]};
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    // A = V^T: lane = (hd row b*32 + l32, half): keys 16 st + 4 half + {0..3} and 16 st + 8 + 4 half + {0..3}
                    const uint8_t* vr = &s_vt[buf][(b * 32 + l32) * VP + (16 * st + 4 * half) * 2];
                    const uint2 v0 = *(const uint2*)vr, v1 = *(const uint2*)(vr + 16);
                    const u32x4_t vfrag = {v0.x, v0.y, v1.x, v1.y};
                    o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfrag), __builtin_bit_cast(bf16x8_t, ph), o[b], 0, 0, 0);
                    o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfrag), __builtin_bit_cast(bf16x8_t, pl), o[b], 0, 0, 0);
                }
            }
        }
        if (t + 1 < n_tiles) store_tile(buf ^ 1);
        lds_barrier();
    }
    // ---- finish: l = both halves' sums (same maximum), out[q][head][hd] = O / l
    const float l_tot = l_run + other_half(l_run);
    if (part_o) { // key split: f32 partials, row = (split, query, head)
        if (q_live) {
            const size_t row = ((size_t)blockIdx.z * M + qi) * a.num_heads + head;
            float* po = part_o + row * HD;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(float4*)(po + b * 32 + 8 * j + 4 * half) = make_float4(o[b][4 * j], o[b][4 * j + 1], o[b][4 * j + 2], o[b][4 * j + 3]);
            if (half == 0) part_ml[2 * row] = m_run, part_ml[2 * row + 1] = l_tot;
        }
        return;
    }
    if (q_live) {
        const float inv = 1.0f / l_tot;
        uint16_t* orow = out + ((size_t)qi * a.num_heads + head) * HD;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) { // rows (hd) 8 j + 4 half + {0..3} of block b
                uint2 w;
                w.x = pack2(o[b][4 * j] * inv, o[b][4 * j + 1] * inv);
                w.y = pack2(o[b][4 * j + 2] * inv, o[b][4 * j + 3] * inv);
                *(uint2*)(orow + b * 32 + 8 * j + 4 * half) = w;
            }
    }
}

bool attention_prefill_mfma_supported(const AttentionParams& a) {
    if (exact_mode()) return false; // reference-order mode: the scalar-order kernels of k_exact.hip
    static const uint32_t min_m = [] {
        const char* e = lab_env("UZU_ATTN_MFMA_MIN_M");
        return e ? (uint32_t)atoi(e) : 16u;
    }();
    static const bool general_on = [] { // UZU_ATTN_MFMA_GENERAL=0: sliding-window / ring / sink layers back on the VALU kernels (A/B runs)
        const char* e = lab_env("UZU_ATTN_MFMA_GENERAL");
        return !e || atoi(e) != 0;
    }();
    if (a.dt != UZU_BF16 || !a.is_causal || a.trie) return false;
    if (!general_on && (a.sinks || a.is_sliding_window || a.is_kv_cache_ring)) return false;
    if (a.is_kv_cache_ring && !a.is_sliding_window) return false; // a ring without a window does not occur (state.rs:69-136)
    if (a.suffix_length < min_m || !(a.head_dim == 64 || a.head_dim == 128 || a.head_dim == 256)) return false;
    if (a.gqa_factor == 0 || a.num_heads % a.gqa_factor) return false;
    if (a.k_head_stride % 8 || a.k_seq_stride % 8 || a.v_head_stride % 8 || a.v_seq_stride % 8) return false;
    return true;
}

// out[q][head][:] = sum_s w_s O_s / sum_s w_s l_s, w_s = exp(m_s - max m) (a split that saw no key has m = -inf, l = 0: weight 0).
// grid = rows / 4, 256 threads: 64 threads per (query, head) row, HD / 64 elements each.
template <int HD>
// gate (optional): SigmoidGate (sigmoid_gate.rs:9-22) on the merged rows in the same launch -- out = bf16(bf16(o) * sigmoid(gate)), the separate kernel's two roundings
__global__ void __launch_bounds__(256) attention_prefill_merge_kernel(const float* part_o, const float* part_ml, uint16_t* out, uint32_t rows, uint32_t splits, const uint16_t* gate) {
    constexpr int EPT = HD / 64;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6), e0 = (threadIdx.x & 63) * EPT;
    if (row >= rows) return;
    float m = -INFINITY;
    for (uint32_t sp = 0; sp < splits; ++sp) m = fmaxf(m, part_ml[2 * ((size_t)sp * rows + row)]);
    float acc[EPT], l = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc[e] = 0.f;
    for (uint32_t sp = 0; sp < splits; ++sp) {
        const size_t r = (size_t)sp * rows + row;
        const float ms = part_ml[2 * r];
        if (ms == -INFINITY) continue;
        const float w = fast_exp(ms - m);
        l = fmaf(part_ml[2 * r + 1], w, l);
        const float* po = part_o + r * HD + e0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc[e] = fmaf(po[e], w, acc[e]);
    }
    const float inv = 1.0f / l;
    uint16_t* orow = out + (size_t)row * HD + e0;
    if (gate) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const float o = bf16_to_f32((uint16_t)pack2(acc[e] * inv, 0.f)), g = bf16_to_f32(gate[(size_t)row * HD + e0 + e]);
            acc[e] = o * (1.0f / (1.0f + expf_glibc(-g)));
        }
        if constexpr (EPT == 1) {
            orow[0] = f32_to_bf16(acc[0]);
        } else {
#pragma unroll
            for (int e = 0; e < EPT; e += 2) *(uint32_t*)(orow + e) = (uint32_t)f32_to_bf16(acc[e]) | ((uint32_t)f32_to_bf16(acc[e + 1]) << 16);
        }
    } else if constexpr (EPT == 1) {
        orow[0] = (uint16_t)pack2(acc[0] * inv, 0.f);
    } else {
#pragma unroll
        for (int e = 0; e < EPT; e += 2) *(uint32_t*)(orow + e) = pack2(acc[e] * inv, acc[e + 1] * inv);
    }
}

uzu_status attention_prefill_mfma(hipStream_t s, const AttentionParams& a, void* out, const void* gate, uint32_t* gate_done) {
    if (gate_done) *gate_done = 0;
    const uint32_t kv_heads = a.num_heads / a.gqa_factor;
    const uint32_t n_tasks = a.gqa_factor * ((a.suffix_length + TQ - 1) / TQ);
    static const uint32_t force_tpw = [] {
        const char* e = lab_env("UZU_ATTN_TPW");
        return e ? (uint32_t)atoi(e) : 0u;
    }();
    static const int force_split = [] { // UZU_ATTN_KSPLIT: 1 = never split the keys (A/B runs), n = that many splits
        const char* e = lab_env("UZU_ATTN_KSPLIT");
        return e ? atoi(e) : 0;
    }();
    // Few task groups (few kv heads x short chunks): keep four tasks per workgroup -- they share every staged K / V tile -- and split the KEYS
    // over workgroups until the chip is covered, instead of thinning the workgroups to one MFMA wave each (r2: 3.4 % MFMA busy at the 0.8B shape).
    uint32_t tpw = 4, splits = 1;
    float* part = nullptr;
    const uint32_t groups4 = kv_heads * ((n_tasks + 3) / 4);
    const uint32_t max_tiles = (a.sequence_length + TK - 1) / TK; // (host view: without the device-side dyn term -- only bounds the split count)
    if (force_split != 1 && groups4 < 192) { // (a.dyn: the context length joins on the device -- the tile count here is a lower bound, a split may come up empty)
        splits = (256 + groups4 - 1) / groups4;
        if (splits > 8) splits = 8;
        if (force_split > 1) splits = (uint32_t)force_split;
        while (splits > 1 && max_tiles / splits < 2) --splits;
        if (splits > 1) {
            const size_t rows = (size_t)a.suffix_length * a.num_heads;
            part = (float*)stream_workspace(s, (size_t)splits * rows * (a.head_dim + 2) * sizeof(float)); // null while a graph is being captured
            if (!part) splits = 1;
        }
    }
    if (splits == 1) { // fewer tasks per workgroup until the grid covers most of the chip
        while (tpw > 1 && kv_heads * ((n_tasks + tpw - 1) / tpw) < 192) tpw >>= 1;
    }
    if (force_tpw == 1 || force_tpw == 2 || force_tpw == 4) tpw = force_tpw;
    const dim3 grid(kv_heads, (n_tasks + tpw - 1) / tpw, splits);
    const size_t rows = (size_t)a.suffix_length * a.num_heads;
    float* part_o = part;
    float* part_ml = part ? part + (size_t)splits * rows * a.head_dim : nullptr;
    const bool general = a.sinks || a.is_sliding_window || a.is_kv_cache_ring;
#define UZU_LAUNCH(H)                                                                                                                                              \
    {                                                                                                                                                              \
        UZU_PROPAGATE(launch_check([&] {                                                                                                                           \
            if (general) hipLaunchKernelGGL((attention_prefill_mfma_kernel<H, true>), grid, dim3(256), 0, s, a, (uint16_t*)out, tpw, part_o, part_ml);             \
            else hipLaunchKernelGGL((attention_prefill_mfma_kernel<H, false>), grid, dim3(256), 0, s, a, (uint16_t*)out, tpw, part_o, part_ml);                    \
        }, "attention_prefill_mfma"));                                                                                                                             \
        if (!part) return UZU_OK;                                                                                                                                  \
        if (gate && gate_done) *gate_done = 1;                                                                                                                     \
        return launch_check([&] { hipLaunchKernelGGL(attention_prefill_merge_kernel<H>, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, s, part_o, part_ml, (uint16_t*)out, (uint32_t)rows, splits, gate_done ? (const uint16_t*)gate : nullptr); }, \
                            "attention_prefill_merge");                                                                                                            \
    }
    switch (a.head_dim) {
    case 64: UZU_LAUNCH(64);
    case 128: UZU_LAUNCH(128);
    default: UZU_LAUNCH(256);
    }
#undef UZU_LAUNCH
}

} // namespace k
} // namespace uzu
