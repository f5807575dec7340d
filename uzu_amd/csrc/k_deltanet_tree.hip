// k_deltanet_tree.hip -- Gated DeltaNet over a speculated token TREE (BU = crates/backend-uzu/src).
//
// Reference: ConvTreeScan (BU/backends/cpu/kernel/gdn/tree_verify/conv_scan.rs), DeltaNetPrefillPrep in its tree instantiation
// (gdn/prefill_prep.rs with QKT = T, write_log_decay, write_compact_v: encodable_block/mixer/delta_net.rs:257-265), the
// DeltaNetTreeVerify composite (prefix -> Gram -> solve -> out: backends/metal/kernel/gdn/tree_verify.rs:92-187 over
// cpu/kernel/gdn/tree_verify/{prefix,tree_gram,tree_update_solve,out}.rs) and StateAdvance (tree_verify/state_advance.rs).
//
// Shape of the problem on MI355X: a verify pass carries n <= 32 tree nodes (the reference's stream speculates <= 16), a DeltaNet layer
// has Hv x Dv = 2048 independent output columns, and the only sizeable operand is the f32 state (64 KB per head, streamed ONCE for all
// nodes).  The whole chunked solve of one (head, 16-column tile) is ~n^2 dot products of 128 -- nowhere near worth matrix cores -- so it
// runs as ONE launch of Hv * Dv / 16 workgroups (128 for Qwen3.5) that each stage q / k / their state tile in LDS and walk the
// reference's loops with one thread per output element, sequentially in the reference's order.  That makes the results BIT-IDENTICAL to
// the CPU kernels (no second reference-order variant is needed) at a cost of a few microseconds of dependent latency per launch; three
// launches per layer (prep, solve, norm-gate) against the reference's six.  StateAdvance is the exception: 2048 state rows x <= 16
// sequential delta-rule steps in reference order would be one thread per row (~100 us); the production form gives a row to a half-wave
// (tolerance-class like the decode update), the reference-order form exists for uzu_hip_set_exact(1).
#include "device_utils.h"
#include "kernels.h"

namespace uzu {
namespace k {

namespace {
constexpr uint32_t kBlock = 16; // token block of the packed-A / inverse tiles (tree_gram.rs:7)

// ------------------------------------------------------------------------------------------------ conv tree scan + tree prep
// grid (slabs, n), 128 threads = one slab of 128 channels of one node.  Slabs [0, Hk): q heads, [Hk, 2Hk): k heads, [2Hk, 2Hk + Hv): value
// heads, slab 2Hk + Hv: the node's per-head scalars (beta, log decay).  head_k_dim == head_v_dim == 128 (delta_net.rs:176-187).
// do_conv: ConvTreeScan on the slab's channels (history walked through `parents`, taps newest first, SiLU); otherwise the rows are read as
// already convolved.  do_prep: l2-normalised q / k ROUNDED to bf16, compact v, beta, log decay.  out_proj (optional, do_conv): the full
// [n, total_proj_dim] rows ConvTreeScan writes (the non-conv channels pass through).
__global__ void __launch_bounds__(128) dn_tree_prep_kernel(const uint16_t* __restrict__ in_proj, const float* __restrict__ conv_w, const float* __restrict__ conv_b,
                                                           const float* __restrict__ base_state, const int32_t* __restrict__ parents, uint16_t* __restrict__ out_proj,
                                                           float* __restrict__ suffix_state, const float* __restrict__ a_log, const float* __restrict__ dt_bias,
                                                           uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_out, uint16_t* __restrict__ v_out, float* __restrict__ beta_out,
                                                           float* __restrict__ log_decay_out, uint32_t n, uint32_t ks, uint32_t Hk, uint32_t Hv, uint32_t do_conv, uint32_t do_prep) {
    constexpr uint32_t D = 128;
    const uint32_t key_dim = Hk * D, value_dim = Hv * D, conv_dim = 2 * key_dim + value_dim, total = conv_dim + value_dim + 2 * Hv;
    const uint32_t slab = blockIdx.x, node = blockIdx.y, c = threadIdx.x, state_stride = ks - 1;
    __shared__ float s_x[D];
    __shared__ float s_inv;
    if (slab == 2 * Hk + Hv) { // per-head scalars + the pass-through channels of ConvTreeScan's output rows
        if (do_conv && out_proj)
            for (uint32_t ch = conv_dim + c; ch < total; ch += D) out_proj[(size_t)node * total + ch] = in_proj[(size_t)node * total + ch];
        if (do_prep && c < Hv) {
            const uint32_t hv = c;
            const float beta_raw = bf16_to_f32(in_proj[(size_t)node * total + conv_dim + value_dim + hv]);
            const float beta = 1.0f / (1.0f + expf_glibc(-beta_raw));
            const float a_raw = bf16_to_f32(in_proj[(size_t)node * total + conv_dim + value_dim + Hv + hv]);
            const float sp_in = a_raw + dt_bias[hv];
            const float sp = sp_in > 20.0f ? sp_in : logf_glibc(1.0f + expf_glibc(sp_in));
            beta_out[(size_t)node * Hv + hv] = beta;
            log_decay_out[(size_t)node * Hv + hv] = -expf_glibc(a_log[hv]) * sp;
        }
        return;
    }
    const uint32_t ch = slab * D + c; // q heads, k heads and value heads are consecutive 128-channel slabs of the conv block
    float xb; // the channel's value as the activation type holds it
    if (do_conv) {
        float acc = conv_b ? conv_b[ch] : 0.0f;
        int32_t source_row = (int32_t)node;
        for (uint32_t h = 0; h < ks; ++h) { // conv_scan.rs:43-65
            float sample;
            if (source_row >= 0) sample = bf16_to_f32(in_proj[(size_t)source_row * total + ch]);
            else sample = base_state[(size_t)ch * state_stride + (state_stride - (uint32_t)(-source_row))];
            acc += sample * conv_w[(size_t)ch * ks + (ks - 1 - h)];
            if (h < state_stride) suffix_state[((size_t)node * conv_dim + ch) * state_stride + (state_stride - 1 - h)] = sample;
            source_row = source_row >= 0 ? parents[source_row] : source_row - 1;
        }
        const uint16_t bits = f32_to_bf16(silu_f32(acc));
        if (out_proj) out_proj[(size_t)node * total + ch] = bits;
        xb = bf16_to_f32(bits);
    } else {
        xb = bf16_to_f32(in_proj[(size_t)node * total + ch]);
    }
    if (!do_prep) return;
    if (slab >= 2 * Hk) { // compact v: the row's value section as it is (prefill_prep.rs:45-52)
        v_out[(size_t)node * value_dim + (slab - 2 * Hk) * D + c] = f32_to_bf16(xb);
        return;
    }
    s_x[c] = xb;
    __syncthreads();
    if (c == 0) { // the reference's sum: sequential from 0 (prefill_prep.rs:57-62, 72-77)
        float sq = 0.0f;
        for (uint32_t j = 0; j < D; ++j) sq += s_x[j] * s_x[j];
        s_inv = 1.0f / sqrtf(sq + 1e-6f);
    }
    __syncthreads();
    const float inv = s_inv;
    if (slab < Hk) {
        const float q_scale = 1.0f / sqrtf((float)D);
        q_out[(size_t)node * key_dim + slab * D + c] = f32_to_bf16(xb * inv * q_scale);
    } else {
        k_out[(size_t)node * key_dim + (slab - Hk) * D + c] = f32_to_bf16(xb * inv);
    }
}

// ------------------------------------------------------------------------------------------------ prefix + Gram + solve + out
// One workgroup = (value head hv, 16 value columns c0 .. c0 + 15), 256 threads, all n nodes.  Every output element is computed by one
// thread walking the reference's loop in the reference's order (see the file header); LDS rows are padded to 129 floats (threads of a
// wave read the same element index of different rows).
constexpr uint32_t kPad = 129;
struct TreeSolveLds {
    float q[kDnTreeMaxNodes][kPad];
    float k[kDnTreeMaxNodes][kPad];
    float h0[16][kPad];
    float a[kDnTreeMaxNodes][kDnTreeMaxNodes + 1];   // A[row][col] (strict ancestors), tree_gram.rs:63-96
    float qkd[kDnTreeMaxNodes][kDnTreeMaxNodes + 1]; // tree_gram.rs:44-60
    float a_inv[kDnTreeMaxNodes / kBlock][kBlock][kBlock + 1];
    float kh0[kDnTreeMaxNodes][17], qh0[kDnTreeMaxNodes][17], u[kDnTreeMaxNodes][17];
    float acc[kBlock][17];
    float prefix[kDnTreeMaxNodes], beta[kDnTreeMaxNodes], log_decay[kDnTreeMaxNodes];
    uint32_t t_start[kDnTreeMaxNodes], t_end[kDnTreeMaxNodes];
};
__global__ void __launch_bounds__(256) dn_tree_solve_kernel(const uint16_t* __restrict__ q_in, const uint16_t* __restrict__ k_in, const uint16_t* __restrict__ v_in,
                                                            const uint32_t* __restrict__ trie, const float* __restrict__ log_decay, const float* __restrict__ beta_in,
                                                            const float* __restrict__ h0, uint16_t* __restrict__ o, float scale, uint32_t n, uint32_t Hk, uint32_t Hv) {
    constexpr uint32_t D = 128;
    __shared__ TreeSolveLds L;
    const uint32_t tid = threadIdx.x;
    const uint32_t tiles = D / 16, hv = blockIdx.x / tiles, c0 = (blockIdx.x % tiles) * 16, hk = hv / (Hv / Hk);
    const uint32_t key_dim = Hk * D, value_dim = Hv * D;
    // every global operand is requested up front (round 4: the phases below used to start with a dependent global load each -- log decays inside
    // the prefix loop, v inside the solve -- one exposed L2 / HBM round trip per phase of a 20 us kernel)
    if (tid < n) {
        L.t_start[tid] = trie[3 * tid], L.t_end[tid] = trie[3 * tid + 1];
        L.beta[tid] = beta_in[(size_t)tid * Hv + hv];
        L.log_decay[tid] = log_decay[(size_t)tid * Hv + hv];
    }
    static_assert(kDnTreeMaxNodes / kBlock == 2, "v_pre is selected by hand below");
    float v_pre[kDnTreeMaxNodes / kBlock]; // v of (token blk * kBlock + tid / 16, value column tid % 16), one per block of the solve
#pragma unroll
    for (uint32_t blk = 0; blk < kDnTreeMaxNodes / kBlock; ++blk) {
        const uint32_t token = min(blk * kBlock + tid / 16, n - 1); // clamped: never consumed past n
        v_pre[blk] = bf16_to_f32(v_in[(size_t)token * value_dim + hv * D + c0 + tid % 16]);
    }
    for (uint32_t idx = tid; idx < n * D; idx += 256) {
        const uint32_t row = idx / D, d = idx % D;
        L.q[row][d] = bf16_to_f32(q_in[(size_t)row * key_dim + hk * D + d]);
        L.k[row][d] = bf16_to_f32(k_in[(size_t)row * key_dim + hk * D + d]);
    }
    for (uint32_t idx = tid; idx < 16 * D; idx += 256) L.h0[idx / D][idx % D] = h0[((size_t)hv * D + c0) * D + idx]; // 16 consecutive state rows: 8 KB contiguous
    __syncthreads();
    if (tid < n) { // prefix.rs:20-36: log decays of the ancestors-or-self, in column order
        float sum = 0.0f;
        for (uint32_t col = 0; col < n; ++col)
            if (tid >= L.t_start[col] && tid <= L.t_end[col]) sum += L.log_decay[col];
        L.prefix[tid] = sum;
    }
    __syncthreads();
    for (uint32_t idx = tid; idx < n * n; idx += 256) { // tree_gram.rs:44-96
        const uint32_t row = idx / n, col = idx % n;
        float qkd = 0.0f, a = 0.0f;
        if (row >= L.t_start[col] && row <= L.t_end[col]) {
            float qk = 0.0f, kk = 0.0f;
            for (uint32_t d = 0; d < D; ++d) qk += L.q[row][d] * L.k[col][d];
            const float e = expf_glibc(L.prefix[row] - L.prefix[col]);
            qkd = e * scale * qk;
            if (row != col) {
                for (uint32_t d = 0; d < D; ++d) kk += L.k[row][d] * L.k[col][d];
                a = L.beta[row] * e * kk;
            }
        }
        L.qkd[row][col] = qkd, L.a[row][col] = a;
    }
    for (uint32_t idx = tid; idx < n * 16; idx += 256) { // tree_gram.rs:139-158 (kh0) and the q . h0 term of out.rs:62-73
        const uint32_t token = idx / 16, j = idx % 16;
        float sk = 0.0f, sq = 0.0f;
        for (uint32_t d = 0; d < D; ++d) sk += L.k[token][d] * L.h0[j][d];
        for (uint32_t d = 0; d < D; ++d) sq += L.q[token][d] * L.h0[j][d];
        L.kh0[token][j] = sk, L.qh0[token][j] = sq;
    }
    __syncthreads();
    const uint32_t nb = (n + kBlock - 1) / kBlock;
    if (tid < nb * kBlock) { // tree_gram.rs:98-128: (I + A_diag)^-1 by forward substitution; a thread owns one column of one block
        const uint32_t blk = tid / kBlock, col = tid % kBlock, bs = min(kBlock, n - blk * kBlock);
        float inv[kBlock]; // the thread's column stays in registers (the LDS copy made every row wait for the previous row's store)
#pragma unroll
        for (uint32_t row = 0; row < kBlock; ++row) inv[row] = row == col ? 1.0f : 0.0f;
#pragma unroll
        for (uint32_t row = 1; row < kBlock; ++row) {
            if (row > col && row < bs) {
                float sum = 0.0f;
#pragma unroll
                for (uint32_t prev = 0; prev < row; ++prev)
                    if (prev >= col) sum += L.a[blk * kBlock + row][blk * kBlock + prev] * inv[prev];
                inv[row] = -sum;
            }
        }
#pragma unroll
        for (uint32_t row = 0; row < kBlock; ++row) L.a_inv[blk][row][col] = inv[row];
    }
    __syncthreads();
    const uint32_t lt = tid / 16, j = tid % 16; // (token of the block, value column)
    for (uint32_t blk = 0; blk < nb; ++blk) { // tree_update_solve.rs:58-127
        const uint32_t token = blk * kBlock + lt;
        if (token < n) {
            const float v_val = blk == 0 ? v_pre[0] : v_pre[1]; // (kDnTreeMaxNodes / kBlock == 2 blocks; no dynamic register index)
            float acc = L.beta[token] * (v_val - expf_glibc(L.prefix[token]) * L.kh0[token][j]);
            for (uint32_t prev = 0; prev < blk * kBlock; ++prev) acc -= L.a[token][prev] * L.u[prev][j];
            L.acc[lt][j] = acc;
        }
        __syncthreads();
        if (token < n) {
            float sum = 0.0f;
            for (uint32_t lp = 0; lp < kBlock && blk * kBlock + lp < n; ++lp) sum += L.a_inv[blk][lt][lp] * L.acc[lp][j];
            L.u[token][j] = sum;
        }
        __syncthreads();
    }
    for (uint32_t idx = tid; idx < n * 16; idx += 256) { // out.rs:55-84
        const uint32_t row = idx / 16, jj = idx % 16;
        float acc = 0.0f;
        acc += expf_glibc(L.prefix[row]) * scale * L.qh0[row][jj];
        for (uint32_t col = 0; col < n; ++col) acc += L.qkd[row][col] * L.u[col][jj];
        o[(size_t)row * value_dim + hv * D + c0 + jj] = f32_to_bf16(acc);
    }
}

// ------------------------------------------------------------------------------------------------ state advance
// state_advance.rs:25-58.  Production: a state row (hv, dv) lives in the registers of a half-wave (4 consecutive dk per lane) across the
// accepted path; per accepted node: decay, k . row (butterfly over the 32 lanes), delta, rank-one update.
__global__ void __launch_bounds__(256) dn_state_advance_kernel(const uint16_t* __restrict__ k_norm, const uint16_t* __restrict__ v, const float* __restrict__ log_decay,
                                                               const float* __restrict__ beta, const uint32_t* __restrict__ accepted, float* __restrict__ state,
                                                               uint32_t accepted_len, uint32_t Hv, uint32_t Hk) {
    constexpr uint32_t D = 128;
    const uint32_t row = blockIdx.x * 8 + threadIdx.x / 32, sub = threadIdx.x % 32; // 8 state rows per workgroup
    const uint32_t hv = row / D, dv = row % D, hk = hv / (Hv / Hk);
    float* srow = state + (size_t)row * D + sub * 4;
    f32x4_v st = *(const f32x4_v*)srow;
    for (uint32_t ai = 0; ai < accepted_len; ++ai) {
        const uint32_t node = accepted[ai];
        const float decay = expf_glibc(log_decay[(size_t)node * Hv + hv]);
        const float b = beta[(size_t)node * Hv + hv];
        const u32x2_v kb = *(const u32x2_v*)(k_norm + (size_t)node * Hk * D + hk * D + sub * 4);
        const float k0 = bits_to_f32(kb.x << 16), k1 = bits_to_f32(kb.x & 0xFFFF0000u), k2 = bits_to_f32(kb.y << 16), k3 = bits_to_f32(kb.y & 0xFFFF0000u);
        st.x *= decay, st.y *= decay, st.z *= decay, st.w *= decay;
        float part = ((st.x * k0 + st.y * k1) + st.z * k2) + st.w * k3;
        part = xadd1(part), part = xadd2(part), part = xadd4(part), part = xadd8(part);
        part += __shfl_xor(part, 16, 64); // the two rows of 16 lanes of this half-wave
        const float v_val = bf16_to_f32(v[(size_t)node * Hv * D + hv * D + dv]);
        const float delta = b * (v_val - part);
        st.x += k0 * delta, st.y += k1 * delta, st.z += k2 * delta, st.w += k3 * delta;
    }
    *(f32x4_v*)srow = st;
}
// reference order: one thread per state row, everything sequential (uzu_hip_set_exact)
__global__ void __launch_bounds__(64) dn_state_advance_exact_kernel(const uint16_t* k_norm, const uint16_t* v, const float* log_decay, const float* beta, const uint32_t* accepted,
                                                                    float* state, uint32_t accepted_len, uint32_t Hv, uint32_t Hk) {
    constexpr uint32_t D = 128;
    const uint32_t row = blockIdx.x * 64 + threadIdx.x;
    if (row >= Hv * D) return;
    const uint32_t hv = row / D, dv = row % D, hk = hv / (Hv / Hk);
    float* srow = state + (size_t)row * D;
    for (uint32_t ai = 0; ai < accepted_len; ++ai) {
        const uint32_t node = accepted[ai];
        const float decay = expf_glibc(log_decay[(size_t)node * Hv + hv]);
        const float b = beta[(size_t)node * Hv + hv];
        const uint16_t* kr = k_norm + (size_t)node * Hk * D + hk * D;
        float kv_mem = 0.0f;
        for (uint32_t d = 0; d < D; ++d) {
            srow[d] *= decay;
            kv_mem += srow[d] * bf16_to_f32(kr[d]);
        }
        const float delta = b * (bf16_to_f32(v[(size_t)node * Hv * D + hv * D + dv]) - kv_mem);
        for (uint32_t d = 0; d < D; ++d) srow[d] += bf16_to_f32(kr[d]) * delta;
    }
}
} // namespace

static uzu_status check_dims(const char* what, uint32_t n, uint32_t Hk, uint32_t Hv, uint32_t Dk, uint32_t Dv) {
    if (Dk != 128 || Dv != 128 || !Hk || !Hv || Hv % Hk) { // delta_net.rs:176-187
        set_error("%s: head_dim and value_head_dim must be 128 and value heads a multiple of key heads (got %u / %u, %u / %u heads)", what, Dk, Dv, Hk, Hv);
        return UZU_ERR_UNSUPPORTED;
    }
    if (n > kDnTreeMaxNodes) {
        set_error("%s: a verify pass carries at most %u tree nodes (got %u)", what, kDnTreeMaxNodes, n);
        return UZU_ERR_UNSUPPORTED;
    }
    return UZU_OK;
}

uzu_status delta_net_tree_prep(hipStream_t s, const uint16_t* in_proj, const float* conv_w, const float* conv_b, const float* base_state, const int32_t* parents,
                               uint16_t* out_proj, float* suffix_state, const float* a_log, const float* dt_bias, uint16_t* q_out, uint16_t* k_out, uint16_t* v_out,
                               float* beta_out, float* log_decay_out, uint32_t n, uint32_t kernel_size, uint32_t Hk, uint32_t Hv, uint32_t Dk, uint32_t Dv, bool do_conv,
                               bool do_prep) {
    if (!n) return UZU_OK;
    UZU_PROPAGATE(check_dims("delta_net_tree_prep", n, Hk, Hv, Dk, Dv));
    if (do_conv && kernel_size < 2) {
        set_error("conv_tree_scan: kernel_size must be >= 2");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    return launch_check([&] {
        hipLaunchKernelGGL(dn_tree_prep_kernel, dim3(2 * Hk + Hv + 1, n), dim3(128), 0, s, in_proj, conv_w, conv_b, base_state, parents, out_proj, suffix_state, a_log, dt_bias,
                           q_out, k_out, v_out, beta_out, log_decay_out, n, kernel_size, Hk, Hv, do_conv ? 1u : 0u, do_prep ? 1u : 0u);
    }, "dn_tree_prep");
}

uzu_status delta_net_tree_verify(hipStream_t s, const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint32_t* trie, const float* log_decay, const float* beta,
                                 const float* h0, uint16_t* out, uint32_t n, uint32_t Hk, uint32_t Hv, uint32_t Dk, uint32_t Dv) {
    if (!n) return UZU_OK;
    UZU_PROPAGATE(check_dims("delta_net_tree_verify", n, Hk, Hv, Dk, Dv));
    return launch_check([&] { hipLaunchKernelGGL(dn_tree_solve_kernel, dim3(Hv * (Dv / 16)), dim3(256), 0, s, q, k, v, trie, log_decay, beta, h0, out, 1.0f, n, Hk, Hv); },
                        "dn_tree_solve");
}

uzu_status delta_net_state_advance(hipStream_t s, const uint16_t* k_norm, const uint16_t* v, const float* log_decay, const float* beta, const uint32_t* accepted_indices,
                                   float* state, uint32_t accepted_len, uint32_t Hv, uint32_t Hk, uint32_t Dk) {
    if (!accepted_len) return UZU_OK;
    UZU_PROPAGATE(check_dims("state_advance", 0, Hk, Hv, Dk, Dk));
    if (exact_mode())
        return launch_check([&] {
            hipLaunchKernelGGL(dn_state_advance_exact_kernel, dim3((Hv * Dk + 63) / 64), dim3(64), 0, s, k_norm, v, log_decay, beta, accepted_indices, state, accepted_len, Hv, Hk);
        }, "dn_state_advance_exact");
    return launch_check([&] {
        hipLaunchKernelGGL(dn_state_advance_kernel, dim3(Hv * Dk / 8), dim3(256), 0, s, k_norm, v, log_decay, beta, accepted_indices, state, accepted_len, Hv, Hk);
    }, "dn_state_advance");
}

} // namespace k
} // namespace uzu
