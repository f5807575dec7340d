// k_moe.hip -- the Mixture-of-Experts MLP for gfx950: MoeBlock::encode (BU/../encodable_block/mlp/moe/mod.rs:204-350).
//
// Reference kernels and where their statement of record is:
//   MoeRouterTopK                  cpu/kernel/moe/router_topk.rs:9-141           (body)   -> moe_router_topk: BIT-EXACT (the CPU kernel's own four-accumulator order)
//   MoeCountsOffsetsFused          cpu/kernel/moe/counts_offsets_fused.rs:4-55   (body)   -> moe_counts_offsets: integer, exact
//   MoeBlockBasesFromPartials + MoeScatterBucketsMap   scatter_buckets.rs = todo!(), metal/kernel/moe/scatter_buckets.metal -> moe_scatter_buckets: rows of an expert in
//                                  (token, slot) order (what moe_experts_test.rs:92-133 expects); also writes the row -> expert map (MoePassABuildRowMap)
//   MoeGatherXPerm                 cpu/kernel/moe/gather.rs:7-40                 (body)   -> moe_gather: exact
//   MoeExperts{Decode,Prefill}PassA  todo!() on the CPU; metal/kernel/moe/experts_two_pass_decode.metal:12-118 -> moe_experts_pass_a (tolerance-class: wave reductions;
//                                  reference-order form: one thread per output in the order of the repository's CPU statement)
//   MoeExpertsDecodeDownFused2D / PrefillPassB   cpu/kernel/moe/experts_two_pass_decode.rs:33-77 (body) -> moe_experts_down (tolerance-class; reference-order form = the body's fma chain)
//   MoeFinalize                    cpu/kernel/moe/finalize.rs:7-50               (body)   -> moe_finalize: BIT-EXACT (k <= 128 terms summed in slot order)
// The tile-map / dispatch-argument kernels of the Metal path (tiles_map.rs, tiles_pass_a.rs) shape Metal's indirect dispatches; here the grids are launched for the
// row CAPACITY (tokens x active experts) and rows past the routed count (read from device memory) return at once.
//
// MI355X notes: the experts' weights are bf16 and full precision (not the quantised path): pass A / pass B are bandwidth-bound row-times-matrix products.  A wave
// owns one output column of an expert's matrix for up to four rows of that expert's segment -- the weight row is streamed once (16-byte loads, 512 elements per
// wave-step) and used for every row; decode (8 rows of 8 different experts) degenerates to one row per wave, i.e. a GEMV per active expert over 256 CUs.
#include "device_utils.h"
#include "kernels.h"

namespace uzu {
namespace k {

namespace {

constexpr int kMoeRowsPerWave = 4;

// ---- router: one workgroup per token; thread x computes the logit of expert x (x + 256, ...) exactly as router_topk.rs:57-84; thread 0 selects
__global__ void __launch_bounds__(256) moe_router_topk_kernel(const uint16_t* input, const uint16_t* weight, const uint16_t* bias, int32_t* topk_ids, uint16_t* topk_probs, uint32_t d_model,
                                                              uint32_t e, uint32_t k, uint32_t renorm) {
    __shared__ float s_logits[512];
    __shared__ float s_best[128];
    __shared__ int32_t s_ids[128];
    const uint32_t token = blockIdx.x;
    const uint16_t* x_row = input + (size_t)token * d_model;
    for (uint32_t expert = threadIdx.x; expert < e; expert += blockDim.x) {
        const uint16_t* w_row = weight + (size_t)expert * d_model;
        float accum[4] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t chunk = 0; chunk < d_model; chunk += 4) {
            const u32x2_v xv = *(const u32x2_v*)(x_row + chunk), wv = *(const u32x2_v*)(w_row + chunk);
            const float x[4] = {bits_to_f32(xv.x << 16), bits_to_f32(xv.x & 0xFFFF0000u), bits_to_f32(xv.y << 16), bits_to_f32(xv.y & 0xFFFF0000u)};
            const float w[4] = {bits_to_f32(wv.x << 16), bits_to_f32(wv.x & 0xFFFF0000u), bits_to_f32(wv.y << 16), bits_to_f32(wv.y & 0xFFFF0000u)};
#pragma unroll
            for (int i = 0; i < 4; ++i) accum[i] = accum[i] + w[i] * x[i]; // (no contraction: -ffp-contract=off)
        }
        const float sum = (accum[0] + accum[1]) + (accum[2] + accum[3]);
        s_logits[expert] = sum + (bias ? bf16_to_f32(bias[expert]) : 0.0f);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (uint32_t j = 0; j < k; ++j) s_best[j] = -INFINITY, s_ids[j] = -1;
    for (uint32_t expert = 0; expert < e; ++expert) { // insertion into the sorted best list: ties keep the lower expert id (router_topk.rs:90-107)
        const float v = s_logits[expert];
        int insert_pos = -1;
        for (int j = (int)k - 1; j >= 0; --j)
            if (v > s_best[j] || (v == s_best[j] && (s_ids[j] < 0 || (int32_t)expert < s_ids[j]))) insert_pos = j;
        if (insert_pos >= 0) {
            for (int s = (int)k - 1; s > insert_pos; --s) s_best[s] = s_best[s - 1], s_ids[s] = s_ids[s - 1];
            s_best[insert_pos] = v;
            s_ids[insert_pos] = (int32_t)expert;
        }
    }
    const size_t base = (size_t)token * k;
    for (uint32_t kk = 0; kk < k; ++kk) topk_ids[base + kk] = s_ids[kk];
    if (renorm) {
        float max_v = -INFINITY;
        for (uint32_t kk = 0; kk < k; ++kk) max_v = fmaxf(max_v, s_best[kk]);
        float sum = 0.0f;
        for (uint32_t kk = 0; kk < k; ++kk) {
            s_logits[kk] = expf_glibc(s_best[kk] - max_v); // (k <= 128 <= e: the logits are no longer needed)
            sum += s_logits[kk];
        }
        for (uint32_t kk = 0; kk < k; ++kk) topk_probs[base + kk] = f32_to_bf16(sum > 0.0f ? s_logits[kk] / sum : 1.0f / (float)k);
    } else {
        for (uint32_t kk = 0; kk < k; ++kk) topk_probs[base + kk] = f32_to_bf16(s_best[kk]);
    }
}

// ---- counts / offsets: one workgroup; integer atomics in LDS (exact), exclusive scan by one thread (e <= 512)
__global__ void __launch_bounds__(256) moe_counts_offsets_kernel(const int32_t* topk_ids, uint32_t* offsets, uint32_t* sum_k_out, uint32_t* partials, uint32_t total, uint32_t e) {
    __shared__ uint32_t s_counts[512];
    for (uint32_t i = threadIdx.x; i < e; i += blockDim.x) s_counts[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
        const int32_t eid = topk_ids[i];
        if (eid >= 0 && (uint32_t)eid < e) atomicAdd(&s_counts[eid], 1u);
    }
    __syncthreads();
    if (partials)
        for (uint32_t i = threadIdx.x; i < e; i += blockDim.x) partials[i] = s_counts[i];
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (uint32_t i = 0; i < e; ++i) {
            offsets[i] = sum;
            sum += s_counts[i];
        }
        offsets[e] = sum;
        *sum_k_out = sum;
    }
}

// ---- scatter: workgroup x places the (token, slot) entries routed to expert x, in entry order, behind offsets[x]: a stable counting sort.  Chunks of 256 entries,
// rank inside a chunk from wave ballots + the waves' totals.  Block e (one past the experts) marks the entries with an id outside [0, e) in tok2row.
__global__ void __launch_bounds__(256) moe_scatter_buckets_kernel(const int32_t* topk_ids, const uint16_t* topk_probs, const uint32_t* offsets, int32_t* bucketed_ids, uint16_t* bucketed_probs,
                                                                  int32_t* tok2row, uint32_t* row_expert_map, uint32_t total, uint32_t e, uint32_t k) {
    const uint32_t expert = blockIdx.x;
    if (expert == e) {
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
            const int32_t eid = topk_ids[i];
            if (eid < 0 || (uint32_t)eid >= e) tok2row[i] = -1;
        }
        return;
    }
    __shared__ uint32_t s_wave[4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t base = offsets[expert];
    for (uint32_t start = 0; start < total; start += 256) {
        const uint32_t i = start + threadIdx.x;
        const bool mine = i < total && topk_ids[i] == (int32_t)expert;
        const unsigned long long ballot = __ballot(mine);
        const uint32_t before = (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(ballot);
        __syncthreads();
        uint32_t wave_base = 0, chunk_total = 0;
        for (uint32_t w = 0; w < 4; ++w) {
            if (w < wave) wave_base += s_wave[w];
            chunk_total += s_wave[w];
        }
        if (mine) {
            const uint32_t row = base + wave_base + before;
            bucketed_ids[row] = (int32_t)(i / k);
            bucketed_probs[row] = topk_probs[i];
            tok2row[i] = (int32_t)row;
            row_expert_map[row] = expert;
        }
        base += chunk_total;
        __syncthreads();
    }
}

// ---- gather: x_perm[row] = x[bucketed_ids[row]] in 16-byte pieces
__global__ void __launch_bounds__(256) moe_gather_kernel(const uint16_t* x, const int32_t* bucketed_ids, uint16_t* x_perm, const uint32_t* sumk, uint32_t d_model, uint32_t capacity) {
    const uint32_t rows = min(*sumk, capacity), per_row = d_model / 8;
    const size_t total = (size_t)rows * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / per_row), c = (uint32_t)(i % per_row);
        const int32_t token = bucketed_ids[row];
        if (token < 0) continue;
        ((u32x4_v*)(x_perm + (size_t)row * d_model))[c] = ((const u32x4_v*)(x + (size_t)token * d_model))[c];
    }
}

__device__ __forceinline__ float moe_silu_alpha(float x, float alpha) { return x / (1.0f + expf_glibc(-alpha * x)); }
__device__ __forceinline__ float moe_gelu_approx(float x) { // activation_type.rs:44-49
    const float tan_arg = 0.7978846f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(tan_arg));
}
__device__ __forceinline__ float moe_activate(float up_raw, float gate_raw, float up_bias, float gate_bias, const MoeExpertParams& q) {
    float up_val = up_raw + up_bias;
    up_val = fminf(fmaxf(up_val, q.up_clip_min), q.up_clip_max);
    if (q.gating_sel <= 1) return q.gating_sel == 0 ? moe_gelu_approx(up_val) : moe_silu_alpha(up_val, q.silu_alpha);
    float gate_val = gate_raw + gate_bias;
    gate_val = fminf(fmaxf(gate_val, q.gate_clip_min), q.gate_clip_max);
    return (q.gating_sel == 2 ? moe_silu_alpha(gate_val, q.silu_alpha) : moe_gelu_approx(gate_val)) * up_val;
}

__device__ __forceinline__ void unpack8(const u32x4_v v, float (&f)[8]) {
    f[0] = bits_to_f32(v.x << 16), f[1] = bits_to_f32(v.x & 0xFFFF0000u), f[2] = bits_to_f32(v.y << 16), f[3] = bits_to_f32(v.y & 0xFFFF0000u);
    f[4] = bits_to_f32(v.z << 16), f[5] = bits_to_f32(v.z & 0xFFFF0000u), f[6] = bits_to_f32(v.w << 16), f[7] = bits_to_f32(v.w & 0xFFFF0000u);
}

// ---- pass A: hidden[row][h] = act(gate) * up.  Grid (ceil(d_ff / 4), ceil(capacity / 4)): a wave owns hidden column h for the (<= 4) rows of its row group that lie
// in ONE expert's segment; a row group that straddles a segment boundary walks its experts one after the other (wave-uniform loop).
__global__ void __launch_bounds__(256) moe_experts_pass_a_kernel(const uint16_t* x_perm, const uint32_t* row_expert_map, const uint32_t* sumk, const uint16_t* w13_all, const uint16_t* up_biases,
                                                                 float* hidden_out, MoeExpertParams q, uint32_t capacity) {
    const uint32_t rows = min(*sumk, capacity);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t h = blockIdx.x * 4 + wave, row0 = blockIdx.y * kMoeRowsPerWave;
    if (h >= q.d_ff || row0 >= rows) return;
    const uint32_t row_end = min(row0 + kMoeRowsPerWave, rows), D = q.d_model;
    uint32_t r = row0;
    while (r < row_end) {
        const uint32_t expert = row_expert_map[r];
        uint32_t n = 1;
        while (r + n < row_end && row_expert_map[r + n] == expert) ++n;
        const uint16_t* w_up = w13_all + ((size_t)expert * 2 * q.d_ff + h) * D;
        const uint16_t* w_gate = w_up + (size_t)q.d_ff * D;
        float acc_up[kMoeRowsPerWave] = {0.f, 0.f, 0.f, 0.f}, acc_gate[kMoeRowsPerWave] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t c = lane * 8; c < D; c += 512) {
            float wu[8], wg[8];
            unpack8(*(const u32x4_v*)(w_up + c), wu);
            if (q.gating_sel > 1) unpack8(*(const u32x4_v*)(w_gate + c), wg);
#pragma unroll
            for (int j = 0; j < kMoeRowsPerWave; ++j) {
                if ((uint32_t)j >= n) break;
                float xv[8];
                unpack8(*(const u32x4_v*)(x_perm + (size_t)(r + j) * D + c), xv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc_up[j] = fmaf(xv[i], wu[i], acc_up[j]);
                    if (q.gating_sel > 1) acc_gate[j] = fmaf(xv[i], wg[i], acc_gate[j]);
                }
            }
        }
        const float up_bias = bf16_to_f32(up_biases[(size_t)expert * 2 * q.d_ff + h]);
        const float gate_bias = q.gating_sel > 1 ? bf16_to_f32(up_biases[(size_t)expert * 2 * q.d_ff + q.d_ff + h]) : 0.0f;
#pragma unroll
        for (int j = 0; j < kMoeRowsPerWave; ++j) {
            if ((uint32_t)j >= n) break;
            const float up = wave_sum(acc_up[j]), gate = q.gating_sel > 1 ? wave_sum(acc_gate[j]) : 0.0f;
            if (lane == 0) hidden_out[(size_t)(r + j) * q.d_ff + h] = moe_activate(up, gate, up_bias, gate_bias, q);
        }
        r += n;
    }
}
// reference-order form: one thread per (row, h), the dot products from zero in index order, bias behind them (the repository's CPU statement of the shader)
__global__ void __launch_bounds__(256) moe_experts_pass_a_exact_kernel(const uint16_t* x_perm, const uint32_t* row_expert_map, const uint32_t* sumk, const uint16_t* w13_all,
                                                                       const uint16_t* up_biases, float* hidden_out, MoeExpertParams q, uint32_t capacity) {
    const uint32_t rows = min(*sumk, capacity);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * q.d_ff) return;
    const uint32_t row = (uint32_t)(idx / q.d_ff), h = (uint32_t)(idx % q.d_ff), D = q.d_model, expert = row_expert_map[row];
    const uint16_t* x = x_perm + (size_t)row * D;
    const uint16_t* w_up = w13_all + ((size_t)expert * 2 * q.d_ff + h) * D;
    const uint16_t* w_gate = w_up + (size_t)q.d_ff * D;
    float acc_up = 0.0f, acc_gate = 0.0f;
    for (uint32_t d = 0; d < D; ++d) {
        const float xv = bf16_to_f32(x[d]);
        acc_up = acc_up + xv * bf16_to_f32(w_up[d]);
        if (q.gating_sel > 1) acc_gate = acc_gate + xv * bf16_to_f32(w_gate[d]);
    }
    const float up_bias = bf16_to_f32(up_biases[(size_t)expert * 2 * q.d_ff + h]);
    const float gate_bias = q.gating_sel > 1 ? bf16_to_f32(up_biases[(size_t)expert * 2 * q.d_ff + q.d_ff + h]) : 0.0f;
    hidden_out[idx] = moe_activate(acc_up, acc_gate, up_bias, gate_bias, q);
}

// ---- pass B: y[row][col] = hidden[row] . w2[expert][col] + bias.  Same decomposition as pass A over (col, row group).
__global__ void __launch_bounds__(256) moe_experts_down_kernel(const float* hidden, const uint32_t* row_expert_map, const uint32_t* sumk, const uint16_t* w2_all, const uint16_t* down_biases,
                                                               uint16_t* y_out, uint32_t d_model, uint32_t d_ff, uint32_t capacity) {
    const uint32_t rows = min(*sumk, capacity);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = blockIdx.x * 4 + wave, row0 = blockIdx.y * kMoeRowsPerWave;
    if (col >= d_model || row0 >= rows) return;
    const uint32_t row_end = min(row0 + kMoeRowsPerWave, rows);
    uint32_t r = row0;
    while (r < row_end) {
        const uint32_t expert = row_expert_map[r];
        uint32_t n = 1;
        while (r + n < row_end && row_expert_map[r + n] == expert) ++n;
        const uint16_t* w = w2_all + ((size_t)expert * d_model + col) * d_ff;
        float acc[kMoeRowsPerWave] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t c = lane * 8; c < d_ff; c += 512) {
            float wv[8];
            unpack8(*(const u32x4_v*)(w + c), wv);
#pragma unroll
            for (int j = 0; j < kMoeRowsPerWave; ++j) {
                if ((uint32_t)j >= n) break;
                const float* hrow = hidden + (size_t)(r + j) * d_ff + c;
                const f32x4_v h0 = *(const f32x4_v*)hrow, h1 = *(const f32x4_v*)(hrow + 4);
                const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[j] = fmaf(hv[i], wv[i], acc[j]);
            }
        }
        const float bias = bf16_to_f32(down_biases[(size_t)expert * d_model + col]);
#pragma unroll
        for (int j = 0; j < kMoeRowsPerWave; ++j) {
            if ((uint32_t)j >= n) break;
            const float s = wave_sum(acc[j]);
            if (lane == 0) y_out[(size_t)(r + j) * d_model + col] = f32_to_bf16(s + bias);
        }
        r += n;
    }
}
// reference-order form: experts_two_pass_decode.rs:52-73 -- an fma chain over h, then the bias, one rounding
__global__ void __launch_bounds__(256) moe_experts_down_exact_kernel(const float* hidden, const uint32_t* row_expert_map, const uint32_t* sumk, const uint16_t* w2_all,
                                                                     const uint16_t* down_biases, uint16_t* y_out, uint32_t d_model, uint32_t d_ff, uint32_t capacity) {
    const uint32_t rows = min(*sumk, capacity);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * d_model) return;
    const uint32_t row = (uint32_t)(idx / d_model), col = (uint32_t)(idx % d_model), expert = row_expert_map[row];
    const uint16_t* w = w2_all + ((size_t)expert * d_model + col) * d_ff;
    const float* hrow = hidden + (size_t)row * d_ff;
    float acc = 0.0f;
    for (uint32_t h = 0; h < d_ff; ++h) acc = fmaf(hrow[h], bf16_to_f32(w[h]), acc);
    acc = acc + bf16_to_f32(down_biases[(size_t)expert * d_model + col]);
    y_out[idx] = f32_to_bf16(acc);
}

// ---- finalize: y[t][f] = sum over the k slots of prob * y_partial[row][f], slot order, non-finite terms dropped (finalize.rs:22-47)
__global__ void __launch_bounds__(256) moe_finalize_kernel(const int32_t* tok2row, const uint16_t* probs, const uint16_t* y_partial, uint16_t* y, uint32_t t_count, uint32_t d_model, uint32_t k) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)t_count * d_model) return;
    const uint32_t ti = (uint32_t)(idx / d_model), f = (uint32_t)(idx % d_model);
    float acc = 0.0f;
    for (uint32_t kk = 0; kk < k; ++kk) {
        const int32_t row = tok2row[(size_t)ti * k + kk];
        if (row < 0) continue;
        float prob = bf16_to_f32(probs[(size_t)ti * k + kk]);
        if (!isfinite(prob)) prob = 0.0f;
        float val = bf16_to_f32(y_partial[(size_t)row * d_model + f]);
        if (!isfinite(val)) val = 0.0f;
        acc = acc + prob * val;
    }
    if (!isfinite(acc)) acc = 0.0f;
    y[idx] = f32_to_bf16(acc);
}

} // namespace

uzu_status moe_router_topk(hipStream_t s, const uint16_t* input, const uint16_t* weight, const uint16_t* bias, int32_t* topk_ids, uint16_t* topk_probs, uint32_t t, uint32_t d_model, uint32_t e,
                           uint32_t k, uint32_t renorm) {
    if (!t) return UZU_OK;
    if (d_model % 4 || e == 0 || e > 512 || k == 0 || k > 128 || k > e) {
        set_error("moe_router_topk: needs d_model %% 4 == 0, 1 <= k <= min(e, 128), e <= 512 (got d_model %u, e %u, k %u)", d_model, e, k);
        return UZU_ERR_INVALID_ARGUMENT;
    }
    return launch_check([&] { hipLaunchKernelGGL(moe_router_topk_kernel, dim3(t), dim3(256), 0, s, input, weight, bias, topk_ids, topk_probs, d_model, e, k, renorm); }, "moe_router_topk");
}
uzu_status moe_counts_offsets(hipStream_t s, const int32_t* topk_ids, uint32_t* offsets, uint32_t* sum_k_out, uint32_t* partials, uint32_t t, uint32_t e, uint32_t k) {
    if (e > 512) {
        set_error("moe_counts_offsets: %u experts (at most 512)", e);
        return UZU_ERR_INVALID_ARGUMENT;
    }
    return launch_check([&] { hipLaunchKernelGGL(moe_counts_offsets_kernel, dim3(1), dim3(256), 0, s, topk_ids, offsets, sum_k_out, partials, t * k, e); }, "moe_counts_offsets");
}
uzu_status moe_scatter_buckets(hipStream_t s, const int32_t* topk_ids, const uint16_t* topk_probs, const uint32_t* offsets, int32_t* bucketed_ids, uint16_t* bucketed_probs, int32_t* tok2row,
                               uint32_t* row_expert_map, uint32_t t, uint32_t e, uint32_t k) {
    if (!t || !k) return UZU_OK;
    return launch_check([&] {
        hipLaunchKernelGGL(moe_scatter_buckets_kernel, dim3(e + 1), dim3(256), 0, s, topk_ids, topk_probs, offsets, bucketed_ids, bucketed_probs, tok2row, row_expert_map, t * k, e, k);
    }, "moe_scatter_buckets");
}
uzu_status moe_gather(hipStream_t s, const uint16_t* x, const int32_t* bucketed_ids, uint16_t* x_perm, const uint32_t* sumk, uint32_t d_model, uint32_t t, uint32_t k) {
    if (!t || !k) return UZU_OK;
    if (d_model % 8) {
        set_error("moe_gather: model_dim %u is not a multiple of 8", d_model);
        return UZU_ERR_UNSUPPORTED;
    }
    const size_t total = (size_t)t * k * (d_model / 8);
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    return launch_check([&] { hipLaunchKernelGGL(moe_gather_kernel, dim3(blocks), dim3(256), 0, s, x, bucketed_ids, x_perm, sumk, d_model, t * k); }, "moe_gather");
}
uzu_status moe_experts_pass_a(hipStream_t s, const uint16_t* x_perm, const uint32_t* row_expert_map, const uint32_t* sumk, const uint16_t* w13_all, const uint16_t* up_biases, float* hidden_out,
                              const MoeExpertParams& q, uint32_t capacity) {
    if (!capacity) return UZU_OK;
    if (q.d_model % 8 || q.gating_sel > 3) {
        set_error("moe_experts_pass_a: model_dim %u %% 8 != 0 or gating %u", q.d_model, q.gating_sel);
        return UZU_ERR_UNSUPPORTED;
    }
    if (exact_mode()) {
        const size_t total = (size_t)capacity * q.d_ff;
        return launch_check([&] { hipLaunchKernelGGL(moe_experts_pass_a_exact_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, x_perm, row_expert_map, sumk, w13_all, up_biases, hidden_out, q, capacity); },
                            "moe_experts_pass_a_exact");
    }
    const dim3 grid((q.d_ff + 3) / 4, (capacity + kMoeRowsPerWave - 1) / kMoeRowsPerWave);
    return launch_check([&] { hipLaunchKernelGGL(moe_experts_pass_a_kernel, grid, dim3(256), 0, s, x_perm, row_expert_map, sumk, w13_all, up_biases, hidden_out, q, capacity); }, "moe_experts_pass_a");
}
uzu_status moe_experts_down(hipStream_t s, const float* hidden, const uint32_t* row_expert_map, const uint32_t* sumk, const uint16_t* w2_all, const uint16_t* down_biases, uint16_t* y_out,
                            uint32_t d_model, uint32_t d_ff, uint32_t capacity) {
    if (!capacity) return UZU_OK;
    if (d_ff % 8) {
        set_error("moe_experts_down: expert hidden dim %u is not a multiple of 8", d_ff);
        return UZU_ERR_UNSUPPORTED;
    }
    if (exact_mode()) {
        const size_t total = (size_t)capacity * d_model;
        return launch_check([&] { hipLaunchKernelGGL(moe_experts_down_exact_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, hidden, row_expert_map, sumk, w2_all, down_biases, y_out, d_model, d_ff, capacity); },
                            "moe_experts_down_exact");
    }
    const dim3 grid((d_model + 3) / 4, (capacity + kMoeRowsPerWave - 1) / kMoeRowsPerWave);
    return launch_check([&] { hipLaunchKernelGGL(moe_experts_down_kernel, grid, dim3(256), 0, s, hidden, row_expert_map, sumk, w2_all, down_biases, y_out, d_model, d_ff, capacity); }, "moe_experts_down");
}
uzu_status moe_finalize(hipStream_t s, const int32_t* tok2row, const uint16_t* probs, const uint16_t* y_partial, uint16_t* y, uint32_t t_count, uint32_t d_model, uint32_t k) {
    const size_t total = (size_t)t_count * d_model;
    if (!total) return UZU_OK;
    return launch_check([&] { hipLaunchKernelGGL(moe_finalize_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, tok2row, probs, y_partial, y, t_count, d_model, k); }, "moe_finalize");
}

} // namespace k
} // namespace uzu
