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
// 32 x 32 Gram matrices + one 32 x 32 forward substitution each).  `dn_chunk_scan_mfma_kernel` then walks the chunks with
// four small dense products per chunk instead of 32 dependent steps; a workgroup owns 16 of a head's Dv columns, so
// 128 workgroups are busy and every product has hundreds of independent FMAs per thread.
// Same algebra, different rounding order: results agree with the one-token recurrence to ~1e-6 relative in f32
// (tools/ prototype in the commit message), far below the bf16 output rounding.  Decays are carried in log space
// (log a clamped at -80) so that products over a chunk cannot underflow into 0 / 0.
#include <stdlib.h>

#include "device_utils.h"
#include "kernels.h"

namespace uzu {
namespace k {

namespace {
constexpr int CC = 32;        // tokens per chunk
constexpr int DKC = 128;      // head_k_dim
constexpr int KP = DKC + 4;   // LDS row pitch of K / Q / S rows in floats (conflict-free b128 reads across rows)
constexpr int TP = CC + 1;    // LDS row pitch of the 32 x 32 matrices
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
#pragma unroll 4
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

// Matrix-core scan.  (The VALU scans of rounds 1-2 read two LDS operands per four FMAs: 780 KB of LDS reads per chunk and workgroup,
// 158 us per 1024 tokens at the 0.8B shape; removed in round 4, git history has them.)  The four products of a chunk run on v_mfma_f32_16x16x4_f32 -- f32
// operands, exact f32 products, f32 accumulation: the arithmetic of the VALU scan in another summation order -- with operands
// fetched once per chunk into registers.  A workgroup owns 16 value columns of a head (grid (Dv / 16, Hv), 4 waves):
//   stage 1  [K; Q] S^T     wave w = one 16-token tile of K (w = 0, 1) or Q (w = 2, 3): 16 x 16 x 128, 32 MFMAs
//   stage 2  D = T R        waves 0, 1: one 16-token tile each, 8 MFMAs; D and D diag(W) parked in LDS
//   stage 3  O = A Q S^T + P D   waves 2, 3: their stage-1 accumulator is the C operand, 8 MFMAs
//   stage 4  S = A_C S + (D W)^T K   every wave owns two 16-wide dk tiles of the state in its accumulators, 2 x 8 MFMAs
// MFMA operand layout (16x16x4): A lane l = row l % 16, k-quarter l / 16; B lane l = column l % 16, k-quarter l / 16; C / D four
// registers = rows 4 (l / 16) + r of column l % 16.  The contraction index of step s in quarter q is (K / 4) q + s for both
// operands, so a lane's operand values are consecutive in memory.
__global__ void __launch_bounds__(256) dn_chunk_scan_mfma_kernel(const float* q_norm, const float* k_norm, const uint16_t* in_proj, const float* ws, float* state,
                                                                 uint16_t* out, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim, uint32_t key_dim,
                                                                 uint32_t value_dim, uint32_t suffix_len) {
    constexpr int DVT = 16, TPP = 36, RPP = 18; // value columns per workgroup; LDS pitches of T / P (16-byte rows) and R / D (conflict-free column walks)
    __shared__ __attribute__((aligned(16))) float sK[CC * KP], sQ[CC * KP], sS[DVT * KP];
    __shared__ __attribute__((aligned(16))) float sT[CC * TPP], sP[CC * TPP];
    __shared__ float sR[CC * RPP], sD[CC * RPP], sDw[CC * RPP];
    __shared__ float sA[CC], sW[CC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    // XCD-aware (head, value tile) <- workgroup: consecutive workgroup ids go round the 8 XCDs, and the Dv / 16 workgroups of a head all walk
    // the head's K / Q rows: with the plain (x = tile, y = head) numbering every one of them sits on another XCD and the rows are fetched
    // 8 times (r2_pmc_fetch_size.csv: 186 MB against ~28 MB).  Here XCD x takes the pairs [x * total / 8, (x + 1) * total / 8): a head's tiles
    // share one L2.
    uint32_t hv = blockIdx.y, tile = blockIdx.x;
    {
        const uint32_t nx = gridDim.x, total = nx * gridDim.y, lin = blockIdx.x + nx * blockIdx.y;
        if (total % 8 == 0) {
            const uint32_t pair = (lin % 8) * (total / 8) + lin / 8;
            hv = pair / nx, tile = pair % nx;
        }
    }
    const uint32_t dv_base = tile * DVT, hk = hv / (num_v_heads / num_k_heads);
    const uint32_t conv_dim = 2 * key_dim + value_dim;
    const size_t total_proj_dim = (size_t)conv_dim + value_dim + 2 * num_v_heads;
    const uint32_t n_chunks = (suffix_len + CC - 1) / CC;

    // state: wave w owns dk tiles 2 w, 2 w + 1 in accumulator layout: creg[tile][r] = S[dv = 4 kq + r][dk = 16 (2 w + tile) + i16]
    f32x4_v creg[2];
    float* sbase = state + ((size_t)hv * head_v_dim + dv_base + 4 * kq) * DKC + 32 * wave + i16;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) creg[t][r] = sbase[(size_t)r * DKC + 16 * t];

    // ---- operand staging: registers <- memory (chunk c + 1) while chunk c is computed, LDS <- registers at the chunk boundary
    f32x4_v st_k[4], st_q[4];
    float st_t[4], st_p[4], st_a = 0.f, st_w = 0.f, st_v[4];
    const int v_row = 16 * (wave & 1) + 4 * kq; // stage-1 accumulator rows of waves 0, 1 (tokens of the chunk)
    auto fetch = [&](uint32_t c) {
        const uint32_t t0 = c * CC;
        const float* w_t = ws + ((size_t)c * num_v_heads + hv) * WS_FLOATS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = tid + 256 * r, t = idx / (DKC / 4), c4 = idx % (DKC / 4);
            const bool live = t0 + t < suffix_len;
            const size_t tok = live ? t0 + t : suffix_len - 1; // clamped + zeroed: unconditional loads stay countable
            const f32x4_v zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4_v kv = *(const f32x4_v*)(k_norm + tok * key_dim + hk * DKC + c4 * 4);
            const f32x4_v qv = *(const f32x4_v*)(q_norm + tok * key_dim + hk * DKC + c4 * 4);
            st_k[r] = live ? kv : zero, st_q[r] = live ? qv : zero;
            st_t[r] = w_t[idx], st_p[r] = w_t[CC * CC + idx];
        }
        st_a = w_t[2 * CC * CC + (tid & (CC - 1))], st_w = w_t[2 * CC * CC + CC + (tid & (CC - 1))];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = t0 + v_row + r < suffix_len;
            const float vv = bf16_to_f32(in_proj[(size_t)(live ? t0 + v_row + r : suffix_len - 1) * total_proj_dim + 2 * key_dim + hv * head_v_dim + dv_base + i16]);
            st_v[r] = live ? vv : 0.f;
        }
    };
    float vreg[4] = {0.f, 0.f, 0.f, 0.f};
    auto publish = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = tid + 256 * r, t = idx / (DKC / 4), c4 = idx % (DKC / 4);
            *(f32x4_v*)(sK + t * KP + c4 * 4) = st_k[r];
            *(f32x4_v*)(sQ + t * KP + c4 * 4) = st_q[r];
            sT[(idx / CC) * TPP + idx % CC] = st_t[r];
            sP[(idx / CC) * TPP + idx % CC] = st_p[r];
        }
        if (tid < CC) sA[tid] = st_a, sW[tid] = st_w;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sS[(4 * kq + r) * KP + 32 * wave + 16 * t + i16] = creg[t][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) vreg[r] = st_v[r];
    };
    auto mfma4 = [](float a, float b, f32x4_v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); };

    fetch(0);
    publish();
    __syncthreads();
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const uint32_t t0 = c * CC;
        fetch(c + 1 < n_chunks ? c + 1 : c); // unconditional (the last chunk refetches itself)
        // ---- stage 1: rows 16 (wave & 1) .. + 16 of K (waves 0, 1) or Q (waves 2, 3) against S^T, contraction over dk = 32 kq + s
        f32x4_v acc1 = {0.f, 0.f, 0.f, 0.f};
        {
            const float* xr = (wave < 2 ? sK : sQ) + (16 * (wave & 1) + i16) * KP + 32 * kq;
            const float* sr = sS + i16 * KP + 32 * kq;
            f32x4_v xa[8], sb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xa[q] = *(const f32x4_v*)(xr + 4 * q), sb[q] = *(const f32x4_v*)(sr + 4 * q);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1 = mfma4(xa[q][e], sb[q][e], acc1);
        }
        if (wave < 2) { // R = V - A (K S^T) for tokens v_row + r, value column i16
#pragma unroll
            for (int r = 0; r < 4; ++r) sR[(v_row + r) * RPP + i16] = vreg[r] - sA[v_row + r] * acc1[r];
        }
        lds_barrier();
        // ---- stage 2 (waves 0, 1): D = T R for tokens 16 wave + 4 kq + r; contraction over i = 8 kq + s
        if (wave < 2) {
            const float* tr = sT + (16 * wave + i16) * TPP + 8 * kq;
            const f32x4_v ta0 = *(const f32x4_v*)tr, ta1 = *(const f32x4_v*)(tr + 4);
            f32x4_v d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) d = mfma4(s2 < 4 ? ta0[s2 & 3] : ta1[s2 & 3], sR[(8 * kq + s2) * RPP + i16], d);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * wave + 4 * kq + r;
                sD[t * RPP + i16] = d[r];
                sDw[t * RPP + i16] = d[r] * sW[t];
            }
        }
        lds_barrier();
        // ---- stage 3 (waves 2, 3): O = A (Q S^T) + P D for tokens 16 (wave - 2) + 4 kq + r
        if (wave >= 2) {
            const int tb = 16 * (wave - 2);
            const float* pr = sP + (tb + i16) * TPP + 8 * kq;
            const f32x4_v pa0 = *(const f32x4_v*)pr, pa1 = *(const f32x4_v*)(pr + 4);
            f32x4_v o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = sA[tb + 4 * kq + r] * acc1[r];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) o = mfma4(s2 < 4 ? pa0[s2 & 3] : pa1[s2 & 3], sD[(8 * kq + s2) * RPP + i16], o);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t t = t0 + tb + 4 * kq + r;
                if (t < suffix_len) out[(size_t)t * value_dim + hv * head_v_dim + dv_base + i16] = f32_to_bf16(o[r]);
            }
        }
        // ---- stage 4 (all waves): S = A_C S + (D W)^T K for dk tiles 2 wave, 2 wave + 1; contraction over tokens 8 kq + s
        {
            const float a_c = sA[CC - 1];
            float dw[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) dw[s2] = sDw[(8 * kq + s2) * RPP + i16]; // A operand: row = value column i16
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) creg[t][r] *= a_c;
                const float* kc = sK + (8 * kq) * KP + 32 * wave + 16 * t + i16;
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) creg[t] = mfma4(dw[s2], kc[s2 * KP], creg[t]);
            }
        }
        lds_barrier(); // all reads of this chunk's LDS operands are done
        if (c + 1 < n_chunks) publish();
        lds_barrier();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sbase[(size_t)r * DKC + 16 * t] = creg[t][r];
}

bool delta_net_prefill_chunked_supported(uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t suffix_len) {
    if (exact_mode()) return false; // reference-order mode: the token-by-token recurrence in the reference's own loop order (k_exact.hip)
    static const uint32_t min_t = [] {
        const char* e = getenv("UZU_DN_CHUNK_MIN_T");
        return e ? (uint32_t)atoi(e) : 64u;
    }();
    return head_k_dim == DKC && num_k_heads && num_v_heads % num_k_heads == 0 && head_v_dim % 16 == 0 && suffix_len >= min_t;
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
        hipLaunchKernelGGL(dn_chunk_scan_mfma_kernel, dim3(head_v_dim / 16, num_v_heads), dim3(256), 0, s, q_norm, k_norm, in_proj, workspace, state, out,
                           num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim, suffix_len);
    }, "delta_net_chunk_scan");
}

} // namespace k
} // namespace uzu
