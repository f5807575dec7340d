// k_deltanet_chunk.hip -- DeltaNetPrefill (BU/cpu/kernel/gdn/prefill.rs:39-80) in chunked form for gfx950.
//
// The reference recurrence per value head (state S [Dv, Dk], decay a_t, write strength b_t, unit k_t, scaled q_t):
//     d_t = b_t (v_t - a_t S_{t-1} k_t)        S_t = a_t S_{t-1} + d_t k_t^T        o_t = S_t q_t
// is sequential in t, and with only Hv * Dv = 2048 independent rows a row-per-half-wave scan
// (k_deltanet.hip::delta_net_prefill_kernel) runs one wave per SIMD at ~650 cycles per token: latency, not work.
// Over a chunk of C = 32 tokens starting at S_0 the same recurrence has a closed form (A_t = prod_{i<=t} a_i):
//     D = T (V - diag(A) K S_0^T)            T = (I + diag(b) L)^-1 diag(b),  L_ij = (A_i / A_j) k_i.k_j   (j < i)
//     O = diag(A) Q S_0^T + P D              P_ti = (A_t / A_i) q_t.k_i                                   (i <= t)
//     S_C = A_C S_0 + D^T diag(A_C / A_i) K
// T and P do not involve the state: `dn_chunk_prep_kernel` builds them for all chunks and heads in parallel (two
// 32 x 32 Gram matrices + one 32 x 32 forward substitution each).  `dn_chunk_scan_kernel` then walks the chunks with
// four small dense products per chunk instead of 32 dependent steps; a workgroup owns 16 of a head's Dv columns, so
// 128 workgroups are busy and every product has hundreds of independent FMAs per thread.
// Same algebra, different rounding order: results agree with the one-token recurrence to ~1e-6 relative in f32
// (tools/ prototype in the commit message), far below the bf16 output rounding.  Decays are carried in log space
// (log a clamped at -80) so that products over a chunk cannot underflow into 0 / 0.
#include "device_utils.h"
#include "kernels.h"

namespace uzu {
namespace k {

namespace {
constexpr int CC = 32;        // tokens per chunk
constexpr int DKC = 128;      // head_k_dim
constexpr int KP = DKC + 4;   // LDS row pitch of K / Q / S rows in floats (conflict-free b128 reads across rows)
constexpr int TP = CC + 1;    // LDS row pitch of the 32 x 32 matrices
constexpr int DVS = 16;       // value columns per scan workgroup
constexpr int WS_FLOATS = 2 * CC * CC + 2 * CC; // per (chunk, value head): T, P, A, W
} // namespace

size_t delta_net_chunk_workspace_bytes(uint32_t num_v_heads, uint32_t suffix_len) {
    return (size_t)((suffix_len + CC - 1) / CC) * num_v_heads * WS_FLOATS * sizeof(float);
}

// grid (chunks, Hv), 256 threads
__global__ void __launch_bounds__(256) dn_chunk_prep_kernel(const float* q_norm, const float* k_norm, const float* beta_buf, const float* decay_buf,
                                                            float* ws, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t key_dim, uint32_t suffix_len) {
    __shared__ __attribute__((aligned(16))) float sK[CC * KP], sQ[CC * KP];
    __shared__ float sKK[CC * TP], sQK[CC * TP], sM[CC * TP];
    __shared__ float s_lg[CC], s_b[CC];
    const int tid = threadIdx.x;
    const uint32_t chunk = blockIdx.x, hv = blockIdx.y, hk = hv / (num_v_heads / num_k_heads);
    const uint32_t t0 = chunk * CC;
    // K, Q rows of the chunk (zero rows past the end: b = 0, a = 1 there, so they change nothing)
    for (int idx = tid; idx < CC * (DKC / 4); idx += 256) {
        const int t = idx / (DKC / 4), c4 = idx % (DKC / 4);
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), qv = kv;
        if (t0 + t < suffix_len) {
            kv = *(const float4*)(k_norm + (size_t)(t0 + t) * key_dim + hk * DKC + c4 * 4);
            qv = *(const float4*)(q_norm + (size_t)(t0 + t) * key_dim + hk * DKC + c4 * 4);
        }
        *(float4*)(sK + t * KP + c4 * 4) = kv;
        *(float4*)(sQ + t * KP + c4 * 4) = qv;
    }
    if (tid < CC) {
        const bool live = t0 + tid < suffix_len;
        const float d = live ? decay_buf[(size_t)(t0 + tid) * num_v_heads + hv] : 1.0f;
        s_lg[tid] = fmaxf(logf_glibc(d), -80.0f); // log a_t (decay 0 => clamped: the state is wiped either way)
        s_b[tid] = live ? beta_buf[(size_t)(t0 + tid) * num_v_heads + hv] : 0.0f;
    }
    __syncthreads();
    if (tid == 0) { // inclusive prefix sum of 32 logs
        float run = 0.f;
        for (int t = 0; t < CC; ++t) {
            run += s_lg[t];
            s_lg[t] = run;
        }
    }
    // Gram matrices: thread -> row i, four columns j0..j0+3
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        float kk[4] = {0.f, 0.f, 0.f, 0.f}, qk[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d4 = 0; d4 < DKC; d4 += 4) {
            const float4 ki = *(const float4*)(sK + i * KP + d4), qi = *(const float4*)(sQ + i * KP + d4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 kj = *(const float4*)(sK + (j0 + j) * KP + d4);
                kk[j] = fmaf(ki.x, kj.x, kk[j]), kk[j] = fmaf(ki.y, kj.y, kk[j]), kk[j] = fmaf(ki.z, kj.z, kk[j]), kk[j] = fmaf(ki.w, kj.w, kk[j]);
                qk[j] = fmaf(qi.x, kj.x, qk[j]), qk[j] = fmaf(qi.y, kj.y, qk[j]), qk[j] = fmaf(qi.z, kj.z, qk[j]), qk[j] = fmaf(qi.w, kj.w, qk[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sKK[i * TP + j0 + j] = kk[j], sQK[i * TP + j0 + j] = qk[j];
    }
    __syncthreads();
    float* w_t = ws + ((size_t)chunk * num_v_heads + hv) * WS_FLOATS;
    float* w_p = w_t + CC * CC;
    for (int idx = tid; idx < CC * CC; idx += 256) {
        const int i = idx / CC, j = idx % CC;
        const float ratio = expf_glibc(fminf(s_lg[i] - s_lg[j], 0.0f)); // A_i / A_j for j <= i
        sM[i * TP + j] = j < i ? s_b[i] * ratio * sKK[i * TP + j] : 0.0f;
        w_p[idx] = j <= i ? ratio * sQK[i * TP + j] : 0.0f;
    }
    if (tid < CC) {
        w_t[2 * CC * CC + tid] = expf_glibc(s_lg[tid]);                      // A_t
        w_t[2 * CC * CC + CC + tid] = expf_glibc(s_lg[CC - 1] - s_lg[tid]);  // A_C / A_t
    }
    __syncthreads();
    // T = (I + M)^-1 diag(b): column c by forward substitution, one thread per column (M_ij is a broadcast LDS read)
    if (tid < CC) {
        float x[CC];
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            float acc = i == tid ? s_b[i] : 0.0f;
#pragma unroll
            for (int j = 0; j < i; ++j) acc = fmaf(-sM[i * TP + j], x[j], acc);
            x[i] = acc;
            w_t[i * CC + tid] = acc;
        }
    }
}

// grid (Dv / DVS, Hv), 256 threads; walks the chunks sequentially
__global__ void __launch_bounds__(256) dn_chunk_scan_kernel(const float* q_norm, const float* k_norm, const uint16_t* in_proj, const float* ws, float* state,
                                                            uint16_t* out, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim, uint32_t key_dim,
                                                            uint32_t value_dim, uint32_t suffix_len) {
    __shared__ __attribute__((aligned(16))) float sK[CC * KP], sQ[CC * KP], sS[DVS * KP];
    __shared__ float sT[CC * TP], sP[CC * TP];
    __shared__ float sR[CC * (DVS + 1)], sD[CC * (DVS + 1)];
    __shared__ float sA[CC], sW[CC];
    const int tid = threadIdx.x;
    const uint32_t hv = blockIdx.y, dv_base = blockIdx.x * DVS, hk = hv / (num_v_heads / num_k_heads);
    const uint32_t conv_dim = 2 * key_dim + value_dim;
    const size_t total_proj_dim = (size_t)conv_dim + value_dim + 2 * num_v_heads;
    const uint32_t n_chunks = (suffix_len + CC - 1) / CC;

    // state slice: thread -> (dv = tid / 16, 8 consecutive dk): registers for the whole scan, mirrored in LDS per chunk
    const int s_dv = tid >> 4, s_dk = (tid & 15) * 8;
    float* srow = state + ((size_t)hv * head_v_dim + dv_base + s_dv) * DKC + s_dk;
    float sreg[8];
    {
        const float4 a = *(const float4*)srow, b = *(const float4*)(srow + 4);
        sreg[0] = a.x, sreg[1] = a.y, sreg[2] = a.z, sreg[3] = a.w, sreg[4] = b.x, sreg[5] = b.y, sreg[6] = b.z, sreg[7] = b.w;
    }
    // product mapping of stages 1-3: thread -> (token t = tid / 8, two value columns dvc, dvc + 1)
    const int p_t = tid >> 3, p_dv = (tid & 7) * 2;

    for (uint32_t c = 0; c < n_chunks; ++c) {
        const uint32_t t0 = c * CC;
        const float* w_t = ws + ((size_t)c * num_v_heads + hv) * WS_FLOATS;
        // ---- stage 0: operands of the chunk into LDS
        for (int idx = tid; idx < CC * (DKC / 4); idx += 256) {
            const int t = idx / (DKC / 4), c4 = idx % (DKC / 4);
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), qv = kv;
            if (t0 + t < suffix_len) {
                kv = *(const float4*)(k_norm + (size_t)(t0 + t) * key_dim + hk * DKC + c4 * 4);
                qv = *(const float4*)(q_norm + (size_t)(t0 + t) * key_dim + hk * DKC + c4 * 4);
            }
            *(float4*)(sK + t * KP + c4 * 4) = kv;
            *(float4*)(sQ + t * KP + c4 * 4) = qv;
        }
        for (int idx = tid; idx < CC * CC; idx += 256) {
            sT[(idx / CC) * TP + idx % CC] = w_t[idx];
            sP[(idx / CC) * TP + idx % CC] = w_t[CC * CC + idx];
        }
        if (tid < CC) sA[tid] = w_t[2 * CC * CC + tid], sW[tid] = w_t[2 * CC * CC + CC + tid];
        *(float4*)(sS + s_dv * KP + s_dk) = make_float4(sreg[0], sreg[1], sreg[2], sreg[3]);
        *(float4*)(sS + s_dv * KP + s_dk + 4) = make_float4(sreg[4], sreg[5], sreg[6], sreg[7]);
        float v0 = 0.f, v1 = 0.f;
        if (t0 + p_t < suffix_len) {
            const uint16_t* vp = in_proj + (size_t)(t0 + p_t) * total_proj_dim + 2 * key_dim + hv * head_v_dim + dv_base + p_dv;
            v0 = bf16_to_f32(vp[0]), v1 = bf16_to_f32(vp[1]);
        }
        __syncthreads();
        // ---- stage 1: K S^T and Q S^T for (t, dv..dv+1); R = V - A K S^T
        float ks0 = 0.f, ks1 = 0.f, qs0 = 0.f, qs1 = 0.f;
        for (int d4 = 0; d4 < DKC; d4 += 4) {
            const float4 kt = *(const float4*)(sK + p_t * KP + d4), qt = *(const float4*)(sQ + p_t * KP + d4);
            const float4 sa = *(const float4*)(sS + p_dv * KP + d4), sb = *(const float4*)(sS + (p_dv + 1) * KP + d4);
            ks0 = fmaf(kt.x, sa.x, ks0), ks0 = fmaf(kt.y, sa.y, ks0), ks0 = fmaf(kt.z, sa.z, ks0), ks0 = fmaf(kt.w, sa.w, ks0);
            ks1 = fmaf(kt.x, sb.x, ks1), ks1 = fmaf(kt.y, sb.y, ks1), ks1 = fmaf(kt.z, sb.z, ks1), ks1 = fmaf(kt.w, sb.w, ks1);
            qs0 = fmaf(qt.x, sa.x, qs0), qs0 = fmaf(qt.y, sa.y, qs0), qs0 = fmaf(qt.z, sa.z, qs0), qs0 = fmaf(qt.w, sa.w, qs0);
            qs1 = fmaf(qt.x, sb.x, qs1), qs1 = fmaf(qt.y, sb.y, qs1), qs1 = fmaf(qt.z, sb.z, qs1), qs1 = fmaf(qt.w, sb.w, qs1);
        }
        const float a_t = sA[p_t];
        sR[p_t * (DVS + 1) + p_dv] = v0 - a_t * ks0;
        sR[p_t * (DVS + 1) + p_dv + 1] = v1 - a_t * ks1;
        __syncthreads();
        // ---- stage 2: D = T R (T is lower triangular; the stored zeros keep the trip count fixed)
        float d0 = 0.f, d1 = 0.f;
#pragma unroll 8
        for (int i = 0; i < CC; ++i) {
            const float tv = sT[p_t * TP + i];
            d0 = fmaf(tv, sR[i * (DVS + 1) + p_dv], d0);
            d1 = fmaf(tv, sR[i * (DVS + 1) + p_dv + 1], d1);
        }
        sD[p_t * (DVS + 1) + p_dv] = d0;
        sD[p_t * (DVS + 1) + p_dv + 1] = d1;
        __syncthreads();
        // ---- stage 3: O = A Q S^T + P D
        float o0 = a_t * qs0, o1 = a_t * qs1;
#pragma unroll 8
        for (int i = 0; i < CC; ++i) {
            const float pv = sP[p_t * TP + i];
            o0 = fmaf(pv, sD[i * (DVS + 1) + p_dv], o0);
            o1 = fmaf(pv, sD[i * (DVS + 1) + p_dv + 1], o1);
        }
        if (t0 + p_t < suffix_len) {
            const uint32_t packed = (uint32_t)f32_to_bf16(o0) | ((uint32_t)f32_to_bf16(o1) << 16);
            *(uint32_t*)(out + (size_t)(t0 + p_t) * value_dim + hv * head_v_dim + dv_base + p_dv) = packed;
        }
        // ---- stage 4: S = A_C S + D^T diag(W) K for (dv, 8 dk)
        {
            const float a_c = sA[CC - 1];
#pragma unroll
            for (int e = 0; e < 8; ++e) sreg[e] *= a_c;
#pragma unroll 4
            for (int i = 0; i < CC; ++i) {
                const float dw = sD[i * (DVS + 1) + s_dv] * sW[i];
                const float4 ka = *(const float4*)(sK + i * KP + s_dk), kb = *(const float4*)(sK + i * KP + s_dk + 4);
                sreg[0] = fmaf(dw, ka.x, sreg[0]), sreg[1] = fmaf(dw, ka.y, sreg[1]), sreg[2] = fmaf(dw, ka.z, sreg[2]), sreg[3] = fmaf(dw, ka.w, sreg[3]);
                sreg[4] = fmaf(dw, kb.x, sreg[4]), sreg[5] = fmaf(dw, kb.y, sreg[5]), sreg[6] = fmaf(dw, kb.z, sreg[6]), sreg[7] = fmaf(dw, kb.w, sreg[7]);
            }
        }
        __syncthreads(); // every LDS array is rewritten by the next chunk's stage 0
    }
    *(float4*)srow = make_float4(sreg[0], sreg[1], sreg[2], sreg[3]);
    *(float4*)(srow + 4) = make_float4(sreg[4], sreg[5], sreg[6], sreg[7]);
}

bool delta_net_prefill_chunked_supported(uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t suffix_len) {
    static const uint32_t min_t = [] {
        const char* e = getenv("UZU_DN_CHUNK_MIN_T");
        return e ? (uint32_t)atoi(e) : 64u;
    }();
    return head_k_dim == DKC && num_k_heads && num_v_heads % num_k_heads == 0 && head_v_dim % DVS == 0 && suffix_len >= min_t;
}

uzu_status delta_net_prefill_chunked(hipStream_t s, const float* q_norm, const float* k_norm, const float* beta, const float* decay, const uint16_t* in_proj,
                                     float* state, uint16_t* out, float* workspace, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim,
                                     uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len) {
    const uint32_t n_chunks = (suffix_len + CC - 1) / CC;
    UZU_PROPAGATE(launch_check([&] {
        hipLaunchKernelGGL(dn_chunk_prep_kernel, dim3(n_chunks, num_v_heads), dim3(256), 0, s, q_norm, k_norm, beta, decay, workspace, num_v_heads, num_k_heads,
                           key_dim, suffix_len);
    }, "delta_net_chunk_prep"));
    return launch_check([&] {
        hipLaunchKernelGGL(dn_chunk_scan_kernel, dim3(head_v_dim / DVS, num_v_heads), dim3(256), 0, s, q_norm, k_norm, in_proj, workspace, state, out, num_v_heads,
                           num_k_heads, head_v_dim, key_dim, value_dim, suffix_len);
    }, "delta_net_chunk_scan");
}

} // namespace k
} // namespace uzu
