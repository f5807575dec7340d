// k_deltanet.hip -- Gated DeltaNet (linear attention) kernels for gfx950: decode step (conv update +
// delta-rule state update) and the sequential prefill path.
//
// Reference semantics: BU/cpu/kernel/gdn/{conv_update,update,conv_scan,prefill_prep,prefill,norm_gate}.rs and
// BU/cpu/kernel/ssm/conv1d.rs (Conv1dPack), with the buffer types of the Metal kernels
// (BU/metal/kernel/gdn/update.metal:19-31): bf16 activations, f32 a_log / dt_bias / norm weights, f32
// conv and SSM state.
//
// State layout [Hv, Dv, Dk] f32 (1 MiB per layer for Qwen3.5): a row (one dv) is 512 contiguous
// bytes; half a wave (32 lanes x 16 B) reads/writes one row with dwordx4 accesses, so the 2 MiB of
// state traffic per layer per token is fully coalesced.
#include "device_utils.h"
#include "kernels.h"

namespace uzu {
namespace k {

// ---------------------------------------------------------------- DeltaNetConvUpdate (conv_update.rs:17-55)
__global__ void __launch_bounds__(256) delta_net_conv_update_kernel(const float* conv_weight, const float* bias,
                                                                    uint16_t* in_out, float* state,
                                                                    uint32_t kernel_size, uint32_t conv_dim,
                                                                    uint32_t state_stride) {
    const uint32_t channel = blockIdx.x * blockDim.x + threadIdx.x;
    if (channel >= conv_dim) return;
    const uint32_t tap_count = kernel_size - 1;
    float* st_row = state + (size_t)channel * state_stride;
    const float* w = conv_weight + (size_t)channel * kernel_size;
    const float x = bf16_to_f32(in_out[channel]);
    float acc = bias ? bias[channel] : 0.0f;
    for (uint32_t tap = 0; tap < tap_count; ++tap) acc += st_row[tap] * w[tap];
    acc += x * w[tap_count];
    in_out[channel] = f32_to_bf16(silu_f32(acc));
    for (uint32_t tap = 1; tap < tap_count; ++tap) st_row[tap - 1] = st_row[tap];
    st_row[tap_count - 1] = x;
}
uzu_status delta_net_conv_update(hipStream_t s, const float* conv_weight, const float* bias, uint16_t* in_out,
                                 float* state, uint32_t kernel_size, uint32_t conv_dim, uint32_t state_stride) {
    if (!conv_dim) return UZU_OK;
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_conv_update_kernel, dim3((conv_dim + 255) / 256), dim3(256), 0, s, conv_weight, bias, in_out,
                           state, kernel_size, conv_dim, state_stride);
    }, "delta_net_conv_update");
}

// ---------------------------------------------------------------- DeltaNetUpdate (update.rs:30-143), Dk = 128
// One workgroup (512 threads = 8 waves) per value head; a half-wave owns one state row at a time.
__global__ void __launch_bounds__(512) delta_net_update_kernel(const uint16_t* in_proj, const float* a_log,
                                                               const float* dt_bias, const float* norm_weight,
                                                               float* state, uint16_t* out, uint32_t num_v_heads,
                                                               uint32_t num_k_heads, uint32_t head_v_dim,
                                                               uint32_t key_dim, uint32_t value_dim, float norm_epsilon) {
    constexpr int DK = 128;
    __shared__ float s_o[512];
    __shared__ float s_red[16];
    const uint32_t hv = blockIdx.x;
    const uint32_t hk = hv / (num_v_heads / num_k_heads);
    const uint32_t conv_dim = 2 * key_dim + value_dim;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane & 31, half = lane >> 5;

    float q[4], kk[4];
    float q_sq = 0.f, k_sq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        q[e] = bf16_to_f32(in_proj[hk * DK + sl * 4 + e]);
        kk[e] = bf16_to_f32(in_proj[key_dim + hk * DK + sl * 4 + e]);
        q_sq += q[e] * q[e];
        k_sq += kk[e] * kk[e];
    }
    q_sq = group_sum<32>(q_sq);
    k_sq = group_sum<32>(k_sq);
    const float q_inv_norm = 1.0f / sqrtf(q_sq + 1e-6f);
    const float k_inv_norm = 1.0f / sqrtf(k_sq + 1e-6f);
    const float q_scale = 1.0f / sqrtf((float)DK);
    float kq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        q[e] = (q[e] * q_inv_norm) * q_scale;
        kk[e] = kk[e] * k_inv_norm;
        kq += kk[e] * q[e];
    }
    const float kq_dot = group_sum<32>(kq);

    const float beta_raw = bf16_to_f32(in_proj[conv_dim + value_dim + hv]);
    const float beta = delta_beta_fast(beta_raw);
    const float a_raw = bf16_to_f32(in_proj[conv_dim + value_dim + num_v_heads + hv]);
    const float decay = delta_decay_fast(a_raw, dt_bias[hv], a_log[hv]);

    const uint32_t rows_per_pass = (blockDim.x >> 6) * 2;
    for (uint32_t i0 = 0; i0 < head_v_dim; i0 += rows_per_pass) {
        const uint32_t i = i0 + wave * 2 + half;
        if (i < head_v_dim) {
            const float v_i = bf16_to_f32(in_proj[2 * key_dim + hv * head_v_dim + i]);
            float4* srow = (float4*)(state + ((size_t)hv * head_v_dim + i) * DK) + sl;
            const float4 sv = *srow;
            const float s4[4] = {sv.x, sv.y, sv.z, sv.w};
            float sq = 0.f, sk = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sq = fmaf(s4[e], q[e], sq);
                sk = fmaf(s4[e], kk[e], sk);
            }
            sq = group_sum<32>(sq);
            sk = group_sum<32>(sk);
            const float retrieved_i = decay * sk;
            const float delta_i = beta * (v_i - retrieved_i);
            const float o_i = decay * sq + delta_i * kq_dot;
            float4 ns;
            ns.x = decay * s4[0] + kk[0] * delta_i;
            ns.y = decay * s4[1] + kk[1] * delta_i;
            ns.z = decay * s4[2] + kk[2] * delta_i;
            ns.w = decay * s4[3] + kk[3] * delta_i;
            *srow = ns;
            if (sl == 0) s_o[i] = o_i;
        }
    }
    __syncthreads();
    if (wave == 0) { // chunk8_sumsq order (device_utils.h): lane t < Dv/8 owns chunk t, butterfly over the chunks
        const uint32_t chunks = head_v_dim / 8;
        const float c = (uint32_t)lane < chunks ? chunk8_sumsq(&s_o[8 * lane]) : 0.f;
        const float total = group_sum_rt(c, (int)chunks);
        if (lane == 0) s_red[0] = total;
    }
    __syncthreads();
    const float sumsq = s_red[0];
    const float inv_rms = 1.0f / sqrtf(sumsq / (float)head_v_dim + norm_epsilon);
    for (uint32_t i = threadIdx.x; i < head_v_dim; i += blockDim.x) {
        const float z_i = bf16_to_f32(in_proj[conv_dim + hv * head_v_dim + i]);
        const float final_val = s_o[i] * inv_rms * norm_weight[i] * silu_f32(z_i);
        out[hv * head_v_dim + i] = f32_to_bf16(final_val);
    }
}
uzu_status delta_net_update(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias,
                            const float* norm_weight, float* state, uint16_t* out, uint32_t num_v_heads,
                            uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim,
                            uint32_t value_dim, float norm_epsilon) {
    if (exact_mode()) return delta_net_update_exact(s, in_proj, a_log, dt_bias, norm_weight, state, out, num_v_heads, num_k_heads, head_k_dim, head_v_dim, key_dim, value_dim, norm_epsilon);
    if (head_k_dim != 128 || head_v_dim > 512 || head_v_dim < 8 || (head_v_dim & (head_v_dim - 1)) || num_k_heads == 0 || num_v_heads % num_k_heads) {
        set_error("delta_net_update: needs head_k_dim == 128, head_v_dim a power of two in [8, 512], Hv %% Hk == 0");
        return UZU_ERR_UNSUPPORTED;
    }
    if (!num_v_heads) return UZU_OK;
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_update_kernel, dim3(num_v_heads), dim3(512), 0, s, in_proj, a_log, dt_bias, norm_weight, state,
                           out, num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim, norm_epsilon);
    }, "delta_net_update");
}

// ---------------------------------------------------------------- Conv1dPack (ssm/conv1d.rs:10-40)
__global__ void __launch_bounds__(256) conv1d_pack_kernel(const float* state_in, const uint16_t* x, float* padded,
                                                          uint32_t state_stride, uint32_t row_stride,
                                                          uint32_t suffix_len, uint32_t num_channels) {
    const size_t total = (size_t)(state_stride + suffix_len) * num_channels;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const uint32_t c = idx % num_channels, row = idx / num_channels;
        const size_t pi = (size_t)row * row_stride + c;
        if (row < state_stride)
            padded[pi] = state_in[(size_t)c * state_stride + row];
        else
            padded[pi] = bf16_to_f32(x[(size_t)(row - state_stride) * row_stride + c]);
    }
}
uzu_status conv1d_pack(hipStream_t s, const float* state_in, const uint16_t* x, float* padded, uint32_t state_stride,
                       uint32_t row_stride, uint32_t suffix_len, uint32_t num_channels) {
    const size_t total = (size_t)(state_stride + suffix_len) * num_channels;
    if (!total) return UZU_OK;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    return launch_check([&] {
        hipLaunchKernelGGL(conv1d_pack_kernel, dim3(blocks), dim3(256), 0, s, state_in, x, padded, state_stride, row_stride, suffix_len, num_channels);
    }, "conv1d_pack");
}

// ---------------------------------------------------------------- DeltaNetConvScan (conv_scan.rs:32-73)
__global__ void __launch_bounds__(256) delta_net_conv_scan_kernel(const float* conv_padded, const float* conv_weight,
                                                                  const float* bias, uint16_t* in_proj, float* state_out,
                                                                  uint32_t suffix_len, uint32_t kernel_size,
                                                                  uint32_t row_stride, uint32_t state_stride,
                                                                  uint32_t conv_dim, uint32_t out_stride) {
    const size_t total = (size_t)suffix_len * conv_dim;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const uint32_t channel = idx % conv_dim, token = idx / conv_dim;
        float acc = bias ? bias[channel] : 0.0f;
        for (uint32_t tap = 0; tap < kernel_size; ++tap)
            acc += conv_weight[(size_t)channel * kernel_size + tap] * conv_padded[(size_t)(token + tap) * row_stride + channel];
        in_proj[(size_t)token * out_stride + channel] = f32_to_bf16(silu_f32(acc));
    }
    const size_t total_state = (size_t)conv_dim * state_stride;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_state; idx += (size_t)gridDim.x * blockDim.x) {
        const uint32_t tap = idx % state_stride, channel = idx / state_stride;
        state_out[idx] = conv_padded[(size_t)(suffix_len + tap) * row_stride + channel];
    }
}
uzu_status delta_net_conv_scan(hipStream_t s, const float* conv_padded, const float* conv_weight, const float* bias,
                               uint16_t* in_proj, float* state_out, uint32_t suffix_len, uint32_t kernel_size,
                               uint32_t row_stride, uint32_t state_stride, uint32_t conv_dim, uint32_t out_stride) {
    const size_t total = (size_t)suffix_len * conv_dim;
    if (!total) return UZU_OK;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_conv_scan_kernel, dim3(blocks), dim3(256), 0, s, conv_padded, conv_weight, bias, in_proj, state_out,
                           suffix_len, kernel_size, row_stride, state_stride, conv_dim, out_stride);
    }, "delta_net_conv_scan");
}

// ---------------------------------------------------------------- fused Conv1dPack + DeltaNetConvScan (engine prefill)
// The reference packs [state | suffix] into an f32 buffer (conv1d.rs:10-40) and convolves out of it (conv_scan.rs:32-73)
// because the scan writes its result over its own input.  At a 1024-token chunk that buffer is 34 MB written and read
// four times.  Same arithmetic without it: `conv_halo_kernel` saves the (kernel_size - 1) pre-conv rows in front of
// every block of TBLK tokens (block 0: the carried state; plus the rows that become the next state), then
// `conv_apply_kernel` gives every (block, channel) one thread that slides a register window down its block and writes
// in place -- no thread reads a row another thread rewrites.
namespace {
constexpr int CONV_TBLK = 16;
constexpr int CONV_MAXK = 8;
// X[i] of the virtual sequence [state | suffix]: i in [-(ks-1), T)
__device__ __forceinline__ float conv_x(const uint16_t* in_proj, const float* state, int i, uint32_t c, uint32_t ks, uint32_t out_stride) {
    return i < 0 ? state[(size_t)c * (ks - 1) + (ks - 1) + i] : bf16_to_f32(in_proj[(size_t)i * out_stride + c]);
}
}
// halo[b][tap][c], b in [0, nblocks]: rows X[b*TBLK - (ks-1) + tap]; slot nblocks holds the next state X[T - (ks-1) + tap]
__global__ void __launch_bounds__(256) conv_halo_kernel(const uint16_t* in_proj, const float* state, float* halo, uint32_t suffix_len, uint32_t ks,
                                                        uint32_t conv_dim, uint32_t out_stride, uint32_t nblocks) {
    const uint32_t taps = ks - 1;
    const size_t total = (size_t)(nblocks + 1) * taps * conv_dim;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t c = idx % conv_dim, tap = (idx / conv_dim) % taps, b = idx / ((size_t)conv_dim * taps);
        const int base = b < nblocks ? (int)(b * CONV_TBLK) : (int)suffix_len;
        halo[idx] = conv_x(in_proj, state, base - (int)taps + (int)tap, c, ks, out_stride);
    }
}
__global__ void __launch_bounds__(256) conv_apply_kernel(uint16_t* in_proj, const float* conv_weight, const float* bias, const float* halo, float* state,
                                                         uint32_t suffix_len, uint32_t ks, uint32_t conv_dim, uint32_t out_stride, uint32_t nblocks) {
    // exp table of the SiLU in LDS: read from memory it is a dependent round trip per element, 16 of them in a row per thread
    __shared__ uint64_t s_exp_tab[32];
    if (threadIdx.x < 32) s_exp_tab[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();
    const uint32_t taps = ks - 1;
    const size_t total = (size_t)nblocks * conv_dim;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t c = idx % conv_dim, b = idx / conv_dim;
        float w[CONV_MAXK], win[CONV_MAXK];
#pragma unroll
        for (uint32_t tap = 0; tap < CONV_MAXK; ++tap) {
            w[tap] = tap < ks ? conv_weight[(size_t)c * ks + tap] : 0.0f;
            win[tap] = tap < taps ? halo[((size_t)b * taps + tap) * conv_dim + c] : 0.0f;
        }
        const float b0 = bias ? bias[c] : 0.0f;
        const uint32_t t_end = (b + 1) * CONV_TBLK < suffix_len ? (b + 1) * CONV_TBLK : suffix_len;
        for (uint32_t t = b * CONV_TBLK; t < t_end; ++t) {
            uint16_t* px = in_proj + (size_t)t * out_stride + c;
            const float x = bf16_to_f32(*px);
            float acc = b0; // the reference's order: taps oldest first, the new token last (conv_scan.rs:44-52)
#pragma unroll
            for (uint32_t tap = 0; tap < CONV_MAXK; ++tap)
                if (tap < taps) acc += w[tap] * win[tap];
            acc += w[taps] * x;
            *px = f32_to_bf16(silu_f32_tab(acc, s_exp_tab));
#pragma unroll
            for (uint32_t tap = 0; tap + 1 < CONV_MAXK; ++tap)
                if (tap + 1 < taps) win[tap] = win[tap + 1];
            win[taps - 1] = x;
        }
        if (b + 1 == nblocks) // this channel's carried state for the next pass
            for (uint32_t tap = 0; tap < taps; ++tap) state[(size_t)c * taps + tap] = halo[((size_t)nblocks * taps + tap) * conv_dim + c];
    }
}
// The same per-channel arithmetic for four neighbouring channels per thread (kernel size 4): the block's 16 rows are requested up front with 8-byte
// accesses (the scalar kernel's 2-byte load -> SiLU -> 2-byte store chain in place cannot be reordered by the compiler: one exposed round trip per token),
// weights / halo / bias as 16-byte vectors.
__global__ void __launch_bounds__(256) conv_apply4_kernel(uint16_t* in_proj, const float* conv_weight, const float* bias, const float* halo, float* state,
                                                          uint32_t suffix_len, uint32_t conv_dim, uint32_t out_stride, uint32_t nblocks) {
    __shared__ uint64_t s_exp_tab[32];
    if (threadIdx.x < 32) s_exp_tab[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();
    constexpr uint32_t KS = 4, taps = KS - 1;
    const uint32_t cq = conv_dim / 4;
    const size_t total = (size_t)nblocks * cq;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t c = (uint32_t)(idx % cq) * 4, b = (uint32_t)(idx / cq);
        const uint32_t t_begin = b * CONV_TBLK, t_end = (b + 1) * CONV_TBLK < suffix_len ? (b + 1) * CONV_TBLK : suffix_len;
        u32x2_v raw[CONV_TBLK];
#pragma unroll
        for (int i = 0; i < CONV_TBLK; ++i) {
            const uint32_t t = t_begin + i < suffix_len ? t_begin + i : suffix_len - 1; // clamped rows are loaded and never consumed
            raw[i] = *(const u32x2_v*)(in_proj + (size_t)t * out_stride + c);
        }
        float w[4][KS], win[4][taps], b0[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const f32x4_v wv = *(const f32x4_v*)(conv_weight + (size_t)(c + ch) * KS);
            w[ch][0] = wv.x, w[ch][1] = wv.y, w[ch][2] = wv.z, w[ch][3] = wv.w;
        }
#pragma unroll
        for (uint32_t tap = 0; tap < taps; ++tap) {
            const f32x4_v hv = *(const f32x4_v*)(halo + ((size_t)b * taps + tap) * conv_dim + c);
            win[0][tap] = hv.x, win[1][tap] = hv.y, win[2][tap] = hv.z, win[3][tap] = hv.w;
        }
        {
            const f32x4_v zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4_v bv = bias ? *(const f32x4_v*)(bias + c) : zero;
            b0[0] = bv.x, b0[1] = bv.y, b0[2] = bv.z, b0[3] = bv.w;
        }
#pragma unroll
        for (int i = 0; i < CONV_TBLK; ++i) {
            if (t_begin + i >= t_end) break;
            const float x[4] = {bits_to_f32(raw[i].x << 16), bits_to_f32(raw[i].x & 0xFFFF0000u), bits_to_f32(raw[i].y << 16), bits_to_f32(raw[i].y & 0xFFFF0000u)};
            uint32_t y[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float acc = b0[ch]; // the reference's order: taps oldest first, the new token last (conv_scan.rs:44-52)
#pragma unroll
                for (uint32_t tap = 0; tap < taps; ++tap) acc += w[ch][tap] * win[ch][tap];
                acc += w[ch][taps] * x[ch];
                y[ch] = f32_to_bf16(silu_f32_tab(acc, s_exp_tab));
#pragma unroll
                for (uint32_t tap = 0; tap + 1 < taps; ++tap) win[ch][tap] = win[ch][tap + 1];
                win[ch][taps - 1] = x[ch];
            }
            u32x2_v o;
            o.x = y[0] | (y[1] << 16), o.y = y[2] | (y[3] << 16);
            *(u32x2_v*)(in_proj + (size_t)(t_begin + i) * out_stride + c) = o;
        }
        if (b + 1 == nblocks) // these channels' carried state for the next pass
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
                for (uint32_t tap = 0; tap < taps; ++tap) state[(size_t)(c + ch) * taps + tap] = halo[((size_t)nblocks * taps + tap) * conv_dim + c + ch];
    }
}
// Out of place (round 6): the conv'd channels go to `conv_out` [suffix_len][conv_dim] bf16 and the in-projection rows stay raw, so a block's window is read from the
// rows themselves -- no halo launch in front.  Kernel size 4, four channels per thread, the arithmetic and its order are conv_apply4_kernel's (bit-identical rows).
// The consumers of a chunked prefill (dn_chunk_prep_kernel<true>: q / k; the chunk scans: v) take the rows from conv_out.
template <int TB>
__global__ void __launch_bounds__(256) conv_apply4_oop_kernel(const uint16_t* in_proj, const float* conv_weight, const float* bias, const float* state, uint16_t* conv_out,
                                                              uint32_t suffix_len, uint32_t conv_dim, uint32_t in_stride, uint32_t nblocks) {
    __shared__ uint64_t s_exp_tab[32];
    if (threadIdx.x < 32) s_exp_tab[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();
    constexpr uint32_t KS = 4, taps = KS - 1;
    const uint32_t cq = conv_dim / 4;
    const size_t total = (size_t)nblocks * cq;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const uint32_t c = (uint32_t)(idx % cq) * 4, b = (uint32_t)(idx / cq);
        const uint32_t t_begin = b * TB, t_end = (b + 1) * TB < suffix_len ? (b + 1) * TB : suffix_len;
        u32x2_v raw[TB + taps]; // rows t_begin - 3 .. t_begin + TB - 1 (clamped rows are loaded and never consumed)
#pragma unroll
        for (int i = 0; i < TB + (int)taps; ++i) {
            const int ti = (int)t_begin - (int)taps + i;
            const uint32_t t = (uint32_t)(ti < 0 ? 0 : ti < (int)suffix_len ? ti : (int)suffix_len - 1);
            raw[i] = *(const u32x2_v*)(in_proj + (size_t)t * in_stride + c);
        }
        float w[4][KS], win[4][taps], b0[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const f32x4_v wv = *(const f32x4_v*)(conv_weight + (size_t)(c + ch) * KS);
            w[ch][0] = wv.x, w[ch][1] = wv.y, w[ch][2] = wv.z, w[ch][3] = wv.w;
        }
#pragma unroll
        for (uint32_t tap = 0; tap < taps; ++tap) { // X[t_begin - 3 + tap]: the carried state in front of the pass, raw rows otherwise
            const int ti = (int)t_begin - (int)taps + (int)tap;
            const float x[4] = {bits_to_f32(raw[tap].x << 16), bits_to_f32(raw[tap].x & 0xFFFF0000u), bits_to_f32(raw[tap].y << 16), bits_to_f32(raw[tap].y & 0xFFFF0000u)};
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) win[ch][tap] = ti < 0 ? state[(size_t)(c + ch) * taps + (uint32_t)(ti + (int)taps)] : x[ch];
        }
        {
            const f32x4_v zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4_v bv = bias ? *(const f32x4_v*)(bias + c) : zero;
            b0[0] = bv.x, b0[1] = bv.y, b0[2] = bv.z, b0[3] = bv.w;
        }
        // (the carried state is READ here by the pass's first blocks and must not be written by this launch: the chunk preparation behind it writes the next state --
        // dn_chunk_prep_kernel<true>, conv_state argument -- from the raw rows, which stay intact)
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            if (t_begin + i >= t_end) break;
            const u32x2_v r = raw[i + taps];
            const float x[4] = {bits_to_f32(r.x << 16), bits_to_f32(r.x & 0xFFFF0000u), bits_to_f32(r.y << 16), bits_to_f32(r.y & 0xFFFF0000u)};
            uint32_t y[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float acc = b0[ch]; // the reference's order: taps oldest first, the new token last (conv_scan.rs:44-52)
#pragma unroll
                for (uint32_t tap = 0; tap < taps; ++tap) acc += w[ch][tap] * win[ch][tap];
                acc += w[ch][taps] * x[ch];
                y[ch] = f32_to_bf16(silu_f32_tab(acc, s_exp_tab));
#pragma unroll
                for (uint32_t tap = 0; tap + 1 < taps; ++tap) win[ch][tap] = win[ch][tap + 1];
                win[ch][taps - 1] = x[ch];
            }
            u32x2_v o;
            o.x = y[0] | (y[1] << 16), o.y = y[2] | (y[3] << 16);
            *(u32x2_v*)(conv_out + (size_t)(t_begin + i) * conv_dim + c) = o;
        }
    }
}
size_t delta_net_conv_fused_workspace_floats(uint32_t suffix_len, uint32_t kernel_size, uint32_t conv_dim) {
    return (size_t)((suffix_len + CONV_TBLK - 1) / CONV_TBLK + 1) * (kernel_size - 1) * conv_dim;
}
uzu_status delta_net_conv_fused(hipStream_t s, uint16_t* in_proj, const float* conv_weight, const float* bias, float* state, float* halo, uint32_t suffix_len,
                                uint32_t kernel_size, uint32_t conv_dim, uint32_t out_stride) {
    if (kernel_size < 2 || kernel_size > CONV_MAXK) {
        set_error("delta_net_conv_fused: kernel size %u", kernel_size);
        return UZU_ERR_UNSUPPORTED;
    }
    if (!suffix_len || !conv_dim) return UZU_OK;
    const uint32_t nblocks = (suffix_len + CONV_TBLK - 1) / CONV_TBLK;
    const size_t h_total = (size_t)(nblocks + 1) * (kernel_size - 1) * conv_dim, a_total = (size_t)nblocks * conv_dim;
    const uint32_t gh = (uint32_t)((h_total + 255) / 256 > 4096 ? 4096 : (h_total + 255) / 256), ga = (uint32_t)((a_total + 255) / 256 > 8192 ? 8192 : (a_total + 255) / 256);
    UZU_PROPAGATE(launch_check([&] { hipLaunchKernelGGL(conv_halo_kernel, dim3(gh), dim3(256), 0, s, in_proj, state, halo, suffix_len, kernel_size, conv_dim, out_stride, nblocks); },
                               "conv_halo"));
    static const bool wide = [] { // UZU_CONV_APPLY4=0: one channel per thread everywhere (A/B runs)
        const char* e = tune_env("conv_apply4");
        return !e || atoi(e) != 0;
    }();
    if (wide && kernel_size == 4 && conv_dim % 4 == 0 && out_stride % 4 == 0 && (((uintptr_t)in_proj | (uintptr_t)halo) & 15) == 0 &&
        (((uintptr_t)conv_weight | (uintptr_t)bias) & 15) == 0) {
        const size_t a4 = (size_t)nblocks * (conv_dim / 4);
        const uint32_t g4 = (uint32_t)((a4 + 255) / 256 > 8192 ? 8192 : (a4 + 255) / 256);
        return launch_check([&] { hipLaunchKernelGGL(conv_apply4_kernel, dim3(g4), dim3(256), 0, s, in_proj, conv_weight, bias, halo, state, suffix_len, conv_dim, out_stride,
                                                     nblocks); }, "conv_apply4");
    }
    return launch_check([&] { hipLaunchKernelGGL(conv_apply_kernel, dim3(ga), dim3(256), 0, s, in_proj, conv_weight, bias, halo, state, suffix_len, kernel_size, conv_dim,
                                                 out_stride, nblocks); }, "conv_apply");
}

bool delta_net_conv_out_of_place_supported(const uint16_t* in_proj, const float* conv_weight, const float* bias, const uint16_t* conv_out, uint32_t kernel_size, uint32_t conv_dim,
                                           uint32_t in_stride) {
    static const bool on = [] { // UZU_HIP_TUNE=conv_oop=0: the in-place conv behind its halo launch (tests/test_gpu_prefill_switches.py: bit-identical)
        const char* e = tune_env("conv_oop");
        return !e || atoi(e) != 0;
    }();
    return on && !exact_mode() && kernel_size == 4 && conv_dim % 4 == 0 && in_stride % 4 == 0 && (((uintptr_t)in_proj | (uintptr_t)conv_out) & 7) == 0 &&
           (((uintptr_t)conv_weight | (uintptr_t)bias) & 15) == 0;
}
// DeltaNetConvScan of a prefill pass, out of place: conv_out [suffix_len][conv_dim] bf16; `state` is only read (the caller's next kernel writes the next state)
uzu_status delta_net_conv_out_of_place(hipStream_t s, const uint16_t* in_proj, const float* conv_weight, const float* bias, const float* state, uint16_t* conv_out,
                                       uint32_t suffix_len, uint32_t conv_dim, uint32_t in_stride) {
    if (!suffix_len || !conv_dim) return UZU_OK;
    // tokens per thread: 8 (rows 11 / 8 per output from the cache instead of 19 / 16, twice the threads) -- UZU_CONV_OOP_TB (lab builds) = 16 for the A/B
    static const int tb = [] {
        const char* e = lab_env("UZU_CONV_OOP_TB");
        return e && atoi(e) == 16 ? 16 : 8;
    }();
    const uint32_t nblocks = (suffix_len + tb - 1) / tb;
    const size_t a4 = (size_t)nblocks * (conv_dim / 4);
    const uint32_t g4 = (uint32_t)((a4 + 255) / 256 > 16384 ? 16384 : (a4 + 255) / 256);
    if (tb == 16)
        return launch_check([&] { hipLaunchKernelGGL(conv_apply4_oop_kernel<16>, dim3(g4), dim3(256), 0, s, in_proj, conv_weight, bias, state, conv_out, suffix_len, conv_dim, in_stride,
                                                     nblocks); }, "conv_apply4_oop");
    return launch_check([&] { hipLaunchKernelGGL(conv_apply4_oop_kernel<8>, dim3(g4), dim3(256), 0, s, in_proj, conv_weight, bias, state, conv_out, suffix_len, conv_dim, in_stride,
                                                 nblocks); }, "conv_apply4_oop");
}

// ---------------------------------------------------------------- DeltaNetPrefillPrep (prefill_prep.rs:30-113)
// one wave per (token, k-head): 128 elements = 2 per lane
__global__ void __launch_bounds__(256) delta_net_prefill_prep_kernel(const uint16_t* in_proj, const float* a_log,
                                                                     const float* dt_bias, float* q_norm_out,
                                                                     float* k_norm_out, float* beta_out, float* decay_out,
                                                                     uint32_t num_v_heads, uint32_t num_k_heads,
                                                                     uint32_t key_dim, uint32_t value_dim,
                                                                     uint32_t suffix_len) {
    constexpr int DK = 128;
    __shared__ uint64_t s_exp_tab[32]; // three chained exponentials per (token, value head): table reads from LDS, not dependent global loads
    if (threadIdx.x < 32) s_exp_tab[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= suffix_len * num_k_heads) return;
    const uint32_t token = wave / num_k_heads, hk = wave % num_k_heads;
    const uint32_t conv_dim = 2 * key_dim + value_dim;
    const size_t total_proj_dim = (size_t)conv_dim + value_dim + 2 * num_v_heads;
    const size_t tok_offset = (size_t)token * total_proj_dim;
    const uint32_t groups_per_head = num_v_heads / num_k_heads;
    float qv[2], kv[2];
    float q_sq = 0.f, k_sq = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        qv[e] = bf16_to_f32(in_proj[tok_offset + hk * DK + lane * 2 + e]);
        kv[e] = bf16_to_f32(in_proj[tok_offset + key_dim + hk * DK + lane * 2 + e]);
        q_sq += qv[e] * qv[e];
        k_sq += kv[e] * kv[e];
    }
    q_sq = wave_sum(q_sq);
    k_sq = wave_sum(k_sq);
    const float q_inv = 1.0f / sqrtf(q_sq + 1e-6f);
    const float q_scale = 1.0f / sqrtf((float)DK);
    const float k_inv = 1.0f / sqrtf(k_sq + 1e-6f);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        q_norm_out[(size_t)token * key_dim + hk * DK + lane * 2 + e] = qv[e] * q_inv * q_scale;
        k_norm_out[(size_t)token * key_dim + hk * DK + lane * 2 + e] = kv[e] * k_inv;
    }
    if (lane < groups_per_head) {
        const uint32_t hv = hk * groups_per_head + lane;
        const float beta_raw = bf16_to_f32(in_proj[tok_offset + conv_dim + value_dim + hv]);
        const float beta = 1.0f / (1.0f + expf_glibc_tab(-beta_raw, s_exp_tab));
        const float a_raw = bf16_to_f32(in_proj[tok_offset + conv_dim + value_dim + num_v_heads + hv]);
        const float sp_in = a_raw + dt_bias[hv];
        const float sp = sp_in > 20.0f ? sp_in : logf_glibc(1.0f + expf_glibc_tab(sp_in, s_exp_tab));
        const float log_decay = -expf_glibc_tab(a_log[hv], s_exp_tab) * sp;
        beta_out[(size_t)token * num_v_heads + hv] = beta;
        decay_out[(size_t)token * num_v_heads + hv] = expf_glibc_tab(log_decay, s_exp_tab);
    }
}
uzu_status delta_net_prefill_prep(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias,
                                  float* q_norm_out, float* k_norm_out, float* beta_out, float* decay_out,
                                  uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t key_dim,
                                  uint32_t value_dim, uint32_t suffix_len) {
    if (exact_mode()) return delta_net_prefill_prep_exact(s, in_proj, a_log, dt_bias, q_norm_out, k_norm_out, beta_out, decay_out, num_v_heads, num_k_heads, head_k_dim, key_dim, value_dim, suffix_len);
    if (head_k_dim != 128 || num_k_heads == 0 || num_v_heads % num_k_heads || num_v_heads / num_k_heads > 64) {
        set_error("delta_net_prefill_prep: needs head_k_dim == 128 and Hv/Hk <= 64");
        return UZU_ERR_UNSUPPORTED;
    }
    const uint32_t waves = suffix_len * num_k_heads;
    if (!waves) return UZU_OK;
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_prefill_prep_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, in_proj, a_log, dt_bias, q_norm_out,
                           k_norm_out, beta_out, decay_out, num_v_heads, num_k_heads, key_dim, value_dim, suffix_len);
    }, "delta_net_prefill_prep");
}

// ---------------------------------------------------------------- DeltaNetPrefill (prefill.rs:39-80)
// Scan over tokens with the state row held in registers: half a wave per (hv, dv) row, 8 rows of ONE value head per
// workgroup (all 256 CUs busy at Hv * Dv = 2048 rows).  Every step depends on the previous one and there is exactly
// one wave per SIMD, so the cost is the length of the dependent instruction chain -- not memory, not throughput:
//   * TWO tokens per step.  With s the row before the pair, (a, b, v, k, q)_{1,2} the pair's operands:
//         d1 = b1 (v1 - a1 s.k1)                          s1 = a1 s + d1 k1
//         d2 = b2 (v2 - a2 (a1 s.k2 + d1 k1.k2))          s2 = a2 s1 + d2 k2
//         o1 = a1 s.q1 + d1 k1.q1                         o2 = a2 (a1 s.q2 + d1 k1.q2) + d2 k2.q2
//     the four row dots s.k1, s.k2, s.q1, s.q2 are independent (their reductions interleave), the four cross dots
//     k1.k2, k1.q1, k1.q2, k2.q2 do not involve the state: they are computed once per staged tile.  Same algebra as
//     the reference's one-token recurrence, different rounding order (tolerance-class like every reduction kernel);
//   * the q / k rows, decay, beta and the 8 v values of TILE tokens are staged through double-buffered LDS, the loads
//     of tile n+1 are in flight while tile n is scanned, the LDS reads of pair p+1 are issued before pair p's math.
namespace {
constexpr int DNP_TILE = 32;      // tokens per staged tile
constexpr int DNP_ROWS = 8;       // state rows (one value head) per workgroup
constexpr int DNP_QK = 2 * 128;   // floats of k and q per token
constexpr int DNP_SLOT = DNP_QK + 16; // + decay, beta, v[8], cross dots[4] (even tokens), pad -> 272 floats
constexpr int DNP_X = DNP_QK + 10;    // offset of the pair's cross dots inside the even token's slot
}
__global__ void __launch_bounds__(256) delta_net_prefill_kernel(const float* q_norm, const float* k_norm,
                                                                const float* beta_buf, const float* decay_buf,
                                                                const uint16_t* in_proj, float* state, uint16_t* out,
                                                                uint32_t num_v_heads, uint32_t num_k_heads,
                                                                uint32_t head_v_dim, uint32_t key_dim,
                                                                uint32_t value_dim, uint32_t suffix_len) {
    constexpr int DK = 128;
    __shared__ __attribute__((aligned(16))) float s_tile[2][DNP_TILE * DNP_SLOT];
    const int tid = threadIdx.x, sl = tid & 31, hw = tid >> 5; // hw = half-wave = row inside the workgroup
    const uint32_t blocks_per_head = head_v_dim / DNP_ROWS;
    const uint32_t hv = blockIdx.x / blocks_per_head, i0 = (blockIdx.x % blocks_per_head) * DNP_ROWS, i = i0 + hw;
    const uint32_t hk = hv / (num_v_heads / num_k_heads);
    const uint32_t conv_dim = 2 * key_dim + value_dim;
    const size_t total_proj_dim = (size_t)conv_dim + value_dim + 2 * num_v_heads;
    const uint32_t row = hv * head_v_dim + i;
    float4* srow = (float4*)(state + (size_t)row * DK) + sl;
    float4 sv = *srow;
    float s4[4] = {sv.x, sv.y, sv.z, sv.w};
    uint16_t* orow = out + (size_t)hv * head_v_dim + i;

    // staging role: thread -> (token of the tile, 32-float slice of the k|q row); v: one value per thread
    const int st_tok = tid >> 3, st_part = tid & 7; // 32 tokens x 8 slices of 32 floats (k: slices 0-3, q: 4-7)
    float4 stage[8];
    float stage_x = 0.f, stage_d = 0.f, stage_b = 0.f;
    auto load_tile = [&](uint32_t t0) {
        const uint32_t token = t0 + st_tok;
        if (token < suffix_len) {
            const float* src = (st_part < 4 ? k_norm : q_norm) + (size_t)token * key_dim + hk * DK + (st_part & 3) * 32;
#pragma unroll
            for (int v = 0; v < 8; ++v) stage[v] = ((const float4*)src)[v];
            stage_x = bf16_to_f32(in_proj[(size_t)token * total_proj_dim + 2 * key_dim + hv * head_v_dim + i0 + st_part]);
        }
        if (tid < DNP_TILE && t0 + tid < suffix_len) {
            stage_d = decay_buf[(size_t)(t0 + tid) * num_v_heads + hv];
            stage_b = beta_buf[(size_t)(t0 + tid) * num_v_heads + hv];
        }
    };
    auto store_tile = [&](int buf) {
        float* slot = &s_tile[buf][st_tok * DNP_SLOT];
#pragma unroll
        for (int v = 0; v < 8; ++v) *(float4*)(slot + st_part * 32 + v * 4) = stage[v];
        slot[DNP_QK + 2 + st_part] = stage_x;
        if (tid < DNP_TILE) {
            float* sc = &s_tile[buf][tid * DNP_SLOT + DNP_QK];
            sc[0] = stage_d, sc[1] = stage_b;
        }
    };
    // cross dots of the 16 token pairs of a staged tile: 16 threads per pair = 4 dots x 4 partial threads
    auto cross_dots = [&](int buf) {
        const int pair = tid >> 4, dot = (tid >> 2) & 3, part = tid & 3;
        const float* t1 = &s_tile[buf][(2 * pair) * DNP_SLOT];
        const float* t2 = t1 + DNP_SLOT;
        const float* x = dot == 3 ? t2 : t1;                                   // k2 | k1
        const float* y = dot == 0 ? t2 : (dot == 1 ? t1 + 128 : t2 + 128);     // k2 | q1 | q2 | q2
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const float4 xa = *(const float4*)(x + part * 32 + v * 4), ya = *(const float4*)(y + part * 32 + v * 4);
            acc = fmaf(xa.x, ya.x, acc), acc = fmaf(xa.y, ya.y, acc), acc = fmaf(xa.z, ya.z, acc), acc = fmaf(xa.w, ya.w, acc);
        }
        acc = xadd2(xadd1(acc));
        if (part == 0) s_tile[buf][(2 * pair) * DNP_SLOT + DNP_X + dot] = acc;
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    cross_dots(0);
    __syncthreads();
    for (uint32_t t0 = 0; t0 < suffix_len; t0 += DNP_TILE) {
        const int buf = (t0 / DNP_TILE) & 1;
        const bool more = t0 + DNP_TILE < suffix_len;
        if (more) load_tile(t0 + DNP_TILE);
        const uint32_t steps = suffix_len - t0 < (uint32_t)DNP_TILE ? suffix_len - t0 : (uint32_t)DNP_TILE;
        const float* base = s_tile[buf];
        uint32_t t = 0;
        for (; t + 1 < steps; t += 2) { // ---- two tokens per step
            const float* p1 = base + (size_t)t * DNP_SLOT;
            const float* p2 = p1 + DNP_SLOT;
            const float4 k1 = *(const float4*)(p1 + sl * 4), q1 = *(const float4*)(p1 + 128 + sl * 4);
            const float4 k2 = *(const float4*)(p2 + sl * 4), q2 = *(const float4*)(p2 + 128 + sl * 4);
            const float a1 = p1[DNP_QK], b1 = p1[DNP_QK + 1], v1 = p1[DNP_QK + 2 + hw];
            const float a2 = p2[DNP_QK], b2 = p2[DNP_QK + 1], v2 = p2[DNP_QK + 2 + hw];
            const float ckk = p1[DNP_X], ck1q1 = p1[DNP_X + 1], ck1q2 = p1[DNP_X + 2], ck2q2 = p1[DNP_X + 3];
            const float k1f[4] = {k1.x, k1.y, k1.z, k1.w}, k2f[4] = {k2.x, k2.y, k2.z, k2.w};
            const float q1f[4] = {q1.x, q1.y, q1.z, q1.w}, q2f[4] = {q2.x, q2.y, q2.z, q2.w};
            float sk1 = 0.f, sk2 = 0.f, sq1 = 0.f, sq2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sk1 = fmaf(s4[e], k1f[e], sk1);
                sk2 = fmaf(s4[e], k2f[e], sk2);
                sq1 = fmaf(s4[e], q1f[e], sq1);
                sq2 = fmaf(s4[e], q2f[e], sq2);
            }
            sk1 = group_sum<32>(sk1), sk2 = group_sum<32>(sk2), sq1 = group_sum<32>(sq1), sq2 = group_sum<32>(sq2);
            const float d1 = b1 * (v1 - a1 * sk1);
            const float d2 = b2 * (v2 - a2 * (a1 * sk2 + d1 * ckk));
            const float o1 = a1 * sq1 + d1 * ck1q1;
            const float o2 = a2 * (a1 * sq2 + d1 * ck1q2) + d2 * ck2q2;
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] = a2 * (a1 * s4[e] + d1 * k1f[e]) + d2 * k2f[e];
            if (sl == 0) {
                orow[(size_t)(t0 + t) * value_dim] = f32_to_bf16(o1);
                orow[(size_t)(t0 + t + 1) * value_dim] = f32_to_bf16(o2);
            }
        }
        if (t < steps) { // ---- odd tail of the tile: the one-token recurrence
            const float* p1 = base + (size_t)t * DNP_SLOT;
            const float4 k1 = *(const float4*)(p1 + sl * 4), q1 = *(const float4*)(p1 + 128 + sl * 4);
            const float a1 = p1[DNP_QK], b1 = p1[DNP_QK + 1], v1 = p1[DNP_QK + 2 + hw];
            const float k1f[4] = {k1.x, k1.y, k1.z, k1.w}, q1f[4] = {q1.x, q1.y, q1.z, q1.w};
            float sk1 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) sk1 = fmaf(s4[e], k1f[e], sk1);
            sk1 = group_sum<32>(sk1);
            const float d1 = b1 * (v1 - a1 * sk1);
            float o1 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s4[e] = a1 * s4[e] + d1 * k1f[e];
                o1 = fmaf(s4[e], q1f[e], o1);
            }
            o1 = group_sum<32>(o1);
            if (sl == 0) orow[(size_t)(t0 + t) * value_dim] = f32_to_bf16(o1);
        }
        __syncthreads();
        if (more) {
            store_tile(buf ^ 1);
            __syncthreads();
            cross_dots(buf ^ 1);
            __syncthreads();
        }
    }
    sv.x = s4[0], sv.y = s4[1], sv.z = s4[2], sv.w = s4[3];
    *srow = sv;
}
uzu_status delta_net_prefill(hipStream_t s, const float* q_norm, const float* k_norm, const float* beta,
                             const float* decay, const uint16_t* in_proj, float* state, uint16_t* out,
                             uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim,
                             uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len) {
    if (exact_mode()) return delta_net_prefill_exact(s, q_norm, k_norm, beta, decay, in_proj, state, out, num_v_heads, num_k_heads, head_k_dim, head_v_dim, key_dim, value_dim, suffix_len);
    if (head_k_dim != 128 || num_k_heads == 0 || num_v_heads % num_k_heads || head_v_dim % DNP_ROWS) {
        set_error("delta_net_prefill: needs head_k_dim == 128 and head_v_dim %% 8 == 0");
        return UZU_ERR_UNSUPPORTED;
    }
    const uint32_t rows = num_v_heads * head_v_dim;
    if (!rows || !suffix_len) return UZU_OK;
    if (delta_net_prefill_chunked_supported(num_v_heads, num_k_heads, head_k_dim, head_v_dim, suffix_len)) {
        // long suffix: chunked form; scratch = the stream's workspace block (not while the stream is being captured)
        if (void* ws = stream_workspace(s, delta_net_chunk_workspace_bytes(num_v_heads, value_dim, suffix_len)))
            return delta_net_prefill_chunked(s, q_norm, k_norm, beta, decay, in_proj, state, out, (float*)ws, num_v_heads, num_k_heads, head_v_dim, key_dim,
                                             value_dim, suffix_len);
    }
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_prefill_kernel, dim3(rows / DNP_ROWS), dim3(256), 0, s, q_norm, k_norm, beta, decay, in_proj,
                           state, out, num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim, suffix_len);
    }, "delta_net_prefill");
}

// ---------------------------------------------------------------- DeltaNetNormGate (norm_gate.rs:32-66)
__global__ void __launch_bounds__(256) delta_net_norm_gate_kernel(uint16_t* in_out, const uint16_t* in_proj,
                                                                  const float* norm_weight, uint32_t num_v_heads,
                                                                  uint32_t head_v_dim, uint32_t value_dim,
                                                                  uint32_t conv_dim, uint32_t total_proj_dim,
                                                                  float norm_epsilon, uint32_t suffix_len, float* rowsum_out, uint32_t rowsum_stride,
                                                                  uint32_t rowsum_row0, uint32_t rowsum_pad_to) {
    __shared__ uint64_t s_exp_tab[32]; // the SiLU's exp table: an LDS read instead of a dependent global load behind the reduction
    if (threadIdx.x < 32) s_exp_tab[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= suffix_len * num_v_heads) return;
    const uint32_t token = wave / num_v_heads, hv = wave % num_v_heads;
    const size_t base = (size_t)token * value_dim + (size_t)hv * head_v_dim;
    float sumsq = 0.f;
    for (uint32_t i = lane; i < head_v_dim; i += 64) {
        const float v = bf16_to_f32(in_out[base + i]);
        sumsq += v * v;
    }
    sumsq = wave_sum(sumsq);
    const float inv_rms = 1.0f / sqrtf(sumsq / (float)head_v_dim + norm_epsilon);
    float row_sum = 0.f;
    for (uint32_t i = lane; i < head_v_dim; i += 64) {
        const float o_i = bf16_to_f32(in_out[base + i]);
        const float z_i = bf16_to_f32(in_proj[(size_t)token * total_proj_dim + conv_dim + hv * head_v_dim + i]);
        const uint16_t r = f32_to_bf16(o_i * inv_rms * norm_weight[i] * silu_f32_tab(z_i, s_exp_tab));
        in_out[base + i] = r;
        row_sum += bf16_to_f32(r);
    }
    if (rowsum_out) { // the out-projection's offset term wants the sums of the rows it will read (k_gemm128.hip): what it reads is the ROUNDED row
        row_sum = wave_sum(row_sum);
        if (lane == 0) {
            float* dst = rowsum_out + (size_t)hv * rowsum_stride + rowsum_row0;
            dst[token] = row_sum;
            if (token + 1 == suffix_len)
                for (uint32_t t = rowsum_row0 + suffix_len; t < rowsum_pad_to; ++t) rowsum_out[(size_t)hv * rowsum_stride + t] = 0.f;
        }
    }
}
uzu_status delta_net_norm_gate(hipStream_t s, uint16_t* in_out, const uint16_t* in_proj, const float* norm_weight,
                               uint32_t num_v_heads, uint32_t head_v_dim, uint32_t value_dim, uint32_t conv_dim,
                               uint32_t total_proj_dim, float norm_epsilon, uint32_t suffix_len, float* rowsum_out, uint32_t rowsum_stride, uint32_t rowsum_row0,
                               uint32_t rowsum_pad_to) {
    const uint32_t waves = suffix_len * num_v_heads;
    if (!waves) return UZU_OK;
    if (exact_mode()) return delta_net_norm_gate_exact(s, in_out, in_proj, norm_weight, num_v_heads, head_v_dim, value_dim, conv_dim, total_proj_dim, norm_epsilon, suffix_len);
    return launch_check([&] {
        hipLaunchKernelGGL(delta_net_norm_gate_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, in_out, in_proj, norm_weight, num_v_heads,
                           head_v_dim, value_dim, conv_dim, total_proj_dim, norm_epsilon, suffix_len, rowsum_out, rowsum_stride, rowsum_row0, rowsum_pad_to);
    }, "delta_net_norm_gate");
}

} // namespace k
} // namespace uzu
