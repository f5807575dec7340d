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
#include "internal.h"
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

// Sequence split (round 4).  The chain of n chunk steps is the scan's whole cost (2.5 us per chunk with half of the CUs idle: 159 us per 2043-token pass
// of the 0.8B model), and the recurrence is AFFINE in the state, row by row -- S_t[i,:] = S_{t-1}[i,:] a_t (I - b_t k_t k_t^T) + b_t v_t[i] k_t^T -- with the
// same linear part for every row.  So the chunks [mid, n) do not have to wait for the chunks [0, mid): they run at the same time from a ZERO state with the
// real values (Z_t, outputs Z_t q_t) and, as Dk more "value columns" per head, from the IDENTITY with zero values (H_t, outputs u_t = H_t q_t); then
//     S_t = S_mid H_t + Z_t        o_t = Z_t q_t + S_mid u_t        (S_mid = the state after chunk mid - 1)
// and `dn_chunk_fixup_kernel` adds the S_mid terms (one [tokens, Dk] x [Dk, Dv] product per head on the f32 matrix cores) before the outputs are rounded.
// Twice the scan work for segment 1, on the CUs the single chain leaves idle.  Placement decides what that buys: as three 4-wave workgroups per tile
// (first form, 384 workgroups) the pairs that happen to share a CU share its matrix pipes and run 3.6 us per chunk -- 116 us for 32 + 32 chunks.  Here a
// segment-1 workgroup is EIGHT waves: group 0 the real tile, group 1 the homogeneous tile of the same head (same K / Q / T / P operands, staged once, each
// group fetching half of them), one workgroup per CU, and segment 0 -- alone on its CUs at 2.5 us per chunk -- takes the larger share of the chunks.
struct ScanSplit {
    uint32_t mid_chunk; // 0: one segment (the plain scan)
    float* mid_state;   // [Hv][Dv][Dk]  final state of segment 0
    float* z_end;       // [Hv][Dv][Dk]  final state of segment 1 started from zero
    float* h_end;       // [Hv][Dk][Dk]  final homogeneous rows of segment 1 (started from the identity)
    float* o32;         // [T][Hv Dv] f32: the real tiles' own outputs (rows of segment 1: Z_t q_t; segment 0's rows are not read again)
    float* u;           // [T - 32 mid][Hv Dk] f32: H_t q_t
};
static uint32_t split_mid_chunk(uint32_t n_chunks, uint32_t head_v_dim) {
    static const uint32_t min_chunks = [] { // UZU_DN_SPLIT: chunks from which the scan is split (0 = never; A/B runs).  Below ~12 chunks the fix-up launch costs more than the shorter chain saves
        const char* e = tune_env("dn_split");
        return e ? (uint32_t)atoi(e) : 16u;
    }();
    static const uint32_t pct = [] { // UZU_DN_SPLIT_PCT: segment 0's share of the chunks (its workgroups run ~2.5 us per chunk, segment 1's double groups ~3.4)
        const char* e = lab_env("UZU_DN_SPLIT_PCT");
        const int v = e ? atoi(e) : 58;
        return (uint32_t)(v < 10 ? 10 : v > 90 ? 90 : v);
    }();
    if (!min_chunks || n_chunks < min_chunks || head_v_dim != (uint32_t)DKC) return 0u; // (a segment-1 workgroup pairs value tile i with homogeneous tile i: Dv = Dk)
    const uint32_t mid = (n_chunks * pct + 50) / 100;
    return mid < 1 ? 1u : mid > n_chunks - 1 ? n_chunks - 1 : mid;
}
static size_t split_floats(uint32_t num_v_heads, uint32_t value_dim, uint32_t suffix_len) {
    return (size_t)2 * value_dim * DKC + (size_t)num_v_heads * DKC * DKC + (size_t)suffix_len * value_dim + (size_t)suffix_len * num_v_heads * DKC;
}
size_t delta_net_chunk_workspace_bytes(uint32_t num_v_heads, uint32_t value_dim, uint32_t suffix_len) {
    return ((size_t)((suffix_len + CC - 1) / CC) * num_v_heads * WS_FLOATS + split_floats(num_v_heads, value_dim, suffix_len)) * sizeof(float);
}

// grid (chunks, Hv), 256 threads.
// FUSED (round 6): DeltaNetPrefillPrep (prefill_prep.rs:30-113; k_deltanet.hip::delta_net_prefill_prep_kernel) runs inside this kernel -- the chunk's q / k rows come
// straight from the conv'd in-projection rows (bf16), are L2-normalised with that kernel's element mapping and reduction order (a wave per token, two elements per
// lane, wave_sum: the same bits), land in LDS for the Gram matrices AND in q_norm / k_norm for the scan (written by the first value head of a key head); beta and the
// decay of the chunk's tokens are computed where they are used.  One launch and one 33 MB read of the f32 rows less per DeltaNet layer (9.85 -> 9.57 ms per 2043-token
// pass of the 0.8B).  Also convolving the q / k channels here (the conv kernel then rewriting the v channels only) was built, bit-identical, and bought nothing
// (9.66-9.73 vs 9.60-9.72 ms, profiles/r6_prep_fused_ab.txt): the window loads and 4 SiLUs per token cost this kernel what the conv kernel saved; not kept.
template <bool FUSED>
__global__ void __launch_bounds__(256) dn_chunk_prep_kernel(const float* q_norm, const float* k_norm, const float* beta_buf, const float* decay_buf,
                                                            float* ws, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t key_dim, uint32_t suffix_len,
                                                            const uint16_t* in_proj, const float* a_log, const float* dt_bias, float* q_norm_out, float* k_norm_out,
                                                            uint32_t value_dim, const uint16_t* conv_rows, float* conv_state) {
    __shared__ __attribute__((aligned(16))) float sK[CC * KP], sQ[CC * KP];
    // the three 32 x 32 matrices live in the Q rows once the Gram products have read them (a barrier in between): 34 KB of LDS per workgroup instead of 47 -- four
    // workgroups per CU, the whole grid of a 2048-token pass (1024 workgroups) resident at once
    float* const sKK = sQ;
    float* const sQK = sQ + CC * TP;
    float* const sM = sQ + 2 * CC * TP;
    static_assert(3 * CC * TP <= CC * KP, "the Gram / M matrices fit in the Q rows");
    __shared__ float s_lg[CC], s_b[CC];
    __shared__ uint64_t s_exp_tab[32];
    const int tid = threadIdx.x;
    const uint32_t groups_per_head = num_v_heads / num_k_heads;
    const uint32_t chunk = blockIdx.x, hv = blockIdx.y, hk = hv / groups_per_head;
    const uint32_t t0 = chunk * CC;
    if constexpr (FUSED) {
        if (tid < 32) s_exp_tab[tid] = kExp2fTab[tid];
        const uint32_t lane = tid & 63, wave = tid >> 6;
        const uint32_t conv_dim = 2 * key_dim + value_dim;
        const size_t total_proj_dim = (size_t)conv_dim + value_dim + 2 * num_v_heads;
        const bool writer = hv % groups_per_head == 0; // one value head per key head files the rows for the scan
        constexpr int TW = CC / 4; // this wave's 8 tokens: every load requested before the first reduction
        const uint32_t tl0 = wave * TW;
        uint32_t qw[TW], kw[TW];
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            const uint32_t t = min(t0 + tl0 + i, suffix_len - 1);
            // (conv_rows: the conv'd channels were written out of place -- k_deltanet.hip::conv_apply4_oop_kernel -- and the in-projection rows are still raw)
            const uint16_t* row = (conv_rows ? conv_rows + (size_t)t * conv_dim : in_proj + (size_t)t * total_proj_dim) + hk * DKC + lane * 2;
            qw[i] = *(const uint32_t*)row;
            kw[i] = *(const uint32_t*)(row + key_dim);
        }
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            const uint32_t tl = tl0 + i, t = t0 + tl;
            const float qv[2] = {bits_to_f32(qw[i] << 16), bits_to_f32(qw[i] & 0xFFFF0000u)}, kv[2] = {bits_to_f32(kw[i] << 16), bits_to_f32(kw[i] & 0xFFFF0000u)};
            float q_sq = 0.f, k_sq = 0.f;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                q_sq += qv[e] * qv[e];
                k_sq += kv[e] * kv[e];
            }
            q_sq = wave_sum(q_sq);
            k_sq = wave_sum(k_sq);
            const float q_inv = 1.0f / sqrtf(q_sq + 1e-6f);
            const float q_scale = 1.0f / sqrtf((float)DKC);
            const float k_inv = 1.0f / sqrtf(k_sq + 1e-6f);
            const bool live = t < suffix_len;
            float2 qn = make_float2(0.f, 0.f), kn = qn; // zero rows past the end: b = 0, a = 1 there, so they change nothing
            if (live) qn = make_float2(qv[0] * q_inv * q_scale, qv[1] * q_inv * q_scale), kn = make_float2(kv[0] * k_inv, kv[1] * k_inv);
            *(float2*)(sQ + tl * KP + lane * 2) = qn;
            *(float2*)(sK + tl * KP + lane * 2) = kn;
            if (live && writer) {
                *(float2*)(q_norm_out + (size_t)t * key_dim + hk * DKC + lane * 2) = qn;
                *(float2*)(k_norm_out + (size_t)t * key_dim + hk * DKC + lane * 2) = kn;
            }
        }
        if (conv_state && chunk == 0) { // the next pass's carried conv state X[T - 3 + tap] (X = [state | raw rows]) -- the out-of-place conv kernel only reads the state;
            // the first chunk's workgroups share the channels; a channel's three values are read before any is written (one thread)
            for (uint32_t c = hv * 256 + tid; c < conv_dim; c += num_v_heads * 256) {
                float nx[3];
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) {
                    const int i = (int)suffix_len - 3 + tap;
                    nx[tap] = i < 0 ? conv_state[(size_t)c * 3 + 3 + i] : bf16_to_f32(in_proj[(size_t)i * total_proj_dim + c]);
                }
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) conv_state[(size_t)c * 3 + tap] = nx[tap];
            }
        }
        __syncthreads(); // (the exp table)
        if (tid < CC) {
            const bool live = t0 + tid < suffix_len;
            float d = 1.0f, b = 0.0f;
            if (live) {
                const uint16_t* row = in_proj + (size_t)(t0 + tid) * total_proj_dim + conv_dim + value_dim;
                const float beta_raw = bf16_to_f32(row[hv]);
                b = 1.0f / (1.0f + expf_glibc_tab(-beta_raw, s_exp_tab));
                const float a_raw = bf16_to_f32(row[num_v_heads + hv]);
                const float sp_in = a_raw + dt_bias[hv];
                const float sp = sp_in > 20.0f ? sp_in : logf_glibc(1.0f + expf_glibc_tab(sp_in, s_exp_tab));
                const float log_decay = -expf_glibc_tab(a_log[hv], s_exp_tab) * sp;
                d = expf_glibc_tab(log_decay, s_exp_tab);
            }
            s_lg[tid] = fmaxf(logf_glibc(d), -80.0f);
            s_b[tid] = b;
        }
    } else {
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
        __syncthreads(); // every thread has read its K / Q rows: the Q rows become the matrices
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
                                                                 uint32_t value_dim, uint32_t suffix_len, uint32_t v_row_stride) {
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
    // (v_row_stride != 0: `in_proj` is the out-of-place conv buffer, rows of conv_dim channels in the same q | k | v order)
    const size_t total_proj_dim = v_row_stride ? (size_t)v_row_stride : (size_t)conv_dim + value_dim + 2 * num_v_heads;
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

// The split scan (ScanSplit).  grid (2 Dv / 16, Hv), 512 threads = two groups of the plain kernel's four waves; the stage code is the plain kernel's.
//   workgroups [0, Dv / 16) of a head: segment 0, chunks [0, mid): group 0 = the value tile from the carried state, group 1 only stages operands;
//   the others:                        segment 1, chunks [mid, n): group 0 = the value tile from a zero state, group 1 = the homogeneous tile (identity, v = 0).
// Shared LDS operands K, Q, T, P, A, W: group 0's threads fetch and publish K and T, group 1's Q and P (pointer selects, no branch: a role-dependent branch
// around global loads or stores makes the compiler wait for every outstanding load at the join -- the next chunk's operands fetched ahead: 2.5 -> 3.8 us per
// chunk in the first form).  Every output goes to an f32 row of its role's buffer AND, rounded, to the output row, through one store path.
__global__ void __launch_bounds__(512) dn_chunk_scan_dual_kernel(const float* q_norm, const float* k_norm, const uint16_t* in_proj, const float* ws, const float* state,
                                                                 uint16_t* out, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim, uint32_t key_dim, uint32_t value_dim,
                                                                 uint32_t suffix_len, ScanSplit sp, uint32_t v_row_stride) {
    constexpr int DVT = 16, TPP = 36, RPP = 18;
    extern __shared__ __attribute__((aligned(16))) float dual_smem[];
    float* sK = dual_smem;                 // [CC][KP]
    float* sQ = sK + CC * KP;              // [CC][KP]
    float* sT = sQ + CC * KP;              // [CC][TPP]
    float* sP = sT + CC * TPP;             // [CC][TPP]
    float* sA = sP + CC * TPP;             // [CC]
    float* sW = sA + CC;                   // [CC]
    float* grp_base = sW + CC;             // per group: sS [DVT][KP], sR / sD / sDw [CC][RPP]
    constexpr int GRP_FLOATS = DVT * KP + 3 * CC * RPP;
    const int tid = threadIdx.x, grp = tid >> 8, ltid = tid & 255, lane = ltid & 63, wave = ltid >> 6, i16 = lane & 15, kq = lane >> 4;
    float* sS = grp_base + grp * GRP_FLOATS;
    float* sR = sS + DVT * KP;
    float* sD = sR + CC * RPP;
    float* sDw = sD + CC * RPP;
    uint32_t hv = blockIdx.y, tile = blockIdx.x;
    { // XCD-aware numbering as in the plain kernel: a head's workgroups share one L2
        const uint32_t nx = gridDim.x, total = nx * gridDim.y, lin = blockIdx.x + nx * blockIdx.y;
        if (total % 8 == 0) {
            const uint32_t pair = (lin % 8) * (total / 8) + lin / 8;
            hv = pair / nx, tile = pair % nx;
        }
    }
    const uint32_t tiles_v = head_v_dim / DVT;
    const bool seg1 = tile >= tiles_v;
    if (seg1) tile -= tiles_v;
    const bool active = seg1 || grp == 0;   // group 1 of a segment-0 workgroup only stages operands
    const bool homog = seg1 && grp == 1;
    const uint32_t dv_base = tile * DVT, hk = hv / (num_v_heads / num_k_heads);
    const uint32_t conv_dim = 2 * key_dim + value_dim;
    // (v_row_stride != 0: `in_proj` is the out-of-place conv buffer, rows of conv_dim channels in the same q | k | v order)
    const size_t total_proj_dim = v_row_stride ? (size_t)v_row_stride : (size_t)conv_dim + value_dim + 2 * num_v_heads;
    const uint32_t n_chunks = (suffix_len + CC - 1) / CC;
    const uint32_t c_begin = seg1 ? sp.mid_chunk : 0u, c_end = seg1 ? n_chunks : sp.mid_chunk;
    // f32 output row of token t = o32[t * o_stride] (real tiles: the shared [T][Hv Dv] buffer; homogeneous tile: u, whose row 0 is token 32 mid)
    const size_t o_stride = homog ? (size_t)num_v_heads * DKC : (size_t)value_dim;
    float* o32 = homog ? sp.u + (size_t)hv * DKC + dv_base + i16 - (size_t)sp.mid_chunk * CC * o_stride : sp.o32 + (size_t)hv * head_v_dim + dv_base + i16;
    // ... and, rounded, to the bf16 output row as well: final for segment 0 (its outputs need nothing more), overwritten by the fix-up for segment 1 (whose two
    // groups write the same cells: Dv = Dk) -- one store path for every role, and the fix-up has no conversion pass over segment 0
    uint16_t* out16 = out + (size_t)hv * head_v_dim + dv_base + i16;
    const uint32_t store_end = active ? suffix_len : 0u;
    const uint32_t v_keep = (active && !homog) ? 0xFFFFFFFFu : 0u; // the homogeneous rows have zero values

    // state in accumulator layout (plain kernel): creg[t][r] = S[dv = 4 kq + r][dk = 16 (2 wave + t) + i16]
    f32x4_v creg[2];
    {
        const float* sinit = state + ((size_t)hv * head_v_dim + dv_base + 4 * kq) * DKC + 32 * wave + i16;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                creg[t][r] = (!seg1 && grp == 0) ? sinit[(size_t)r * DKC + 16 * t] : (homog && dv_base + 4 * kq + r == (uint32_t)(32 * wave + 16 * t + i16)) ? 1.0f : 0.0f;
    }

    // ---- operand staging: this group's half of the shared operands (K + T | Q + P), A / W, its value rows
    const float* kq_src = grp ? q_norm : k_norm;
    float* kq_dst = grp ? sQ : sK;
    float* tp_dst = grp ? sP : sT;
    const uint32_t tp_off = grp ? CC * CC : 0;
    f32x4_v st_kq[4];
    float st_tp[4], st_a = 0.f, st_w = 0.f, st_v[4];
    const int v_row = 16 * (wave & 1) + 4 * kq;
    auto fetch = [&](uint32_t c) {
        const uint32_t t0 = c * CC;
        const float* w_t = ws + ((size_t)c * num_v_heads + hv) * WS_FLOATS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = ltid + 256 * r, t = idx / (DKC / 4), c4 = idx % (DKC / 4);
            const bool live = t0 + t < suffix_len;
            const size_t tok = live ? t0 + t : suffix_len - 1; // clamped + zeroed: unconditional loads stay countable
            const f32x4_v zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4_v xv = *(const f32x4_v*)(kq_src + tok * key_dim + hk * DKC + c4 * 4);
            st_kq[r] = live ? xv : zero;
            st_tp[r] = w_t[tp_off + idx];
        }
        st_a = w_t[2 * CC * CC + (ltid & (CC - 1))], st_w = w_t[2 * CC * CC + CC + (ltid & (CC - 1))];
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
            const int idx = ltid + 256 * r, t = idx / (DKC / 4), c4 = idx % (DKC / 4);
            *(f32x4_v*)(kq_dst + t * KP + c4 * 4) = st_kq[r];
            tp_dst[(idx / CC) * TPP + idx % CC] = st_tp[r];
        }
        if (tid < CC) sA[tid] = st_a, sW[tid] = st_w;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sS[(4 * kq + r) * KP + 32 * wave + 16 * t + i16] = creg[t][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) vreg[r] = bits_to_f32(f32_to_bits(st_v[r]) & v_keep); // (masked here, not at the load: a role condition in `fetch` puts the loads behind branches)
    };
    auto mfma4 = [](float a, float b, f32x4_v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); };

    fetch(c_begin);
    publish();
    __syncthreads();
    for (uint32_t c = c_begin; c < c_end; ++c) {
        const uint32_t t0 = c * CC;
        fetch(c + 1 < c_end ? c + 1 : c); // unconditional (the last chunk refetches itself)
        // ---- stage 1: rows 16 (wave & 1) .. + 16 of K (waves 0, 1) or Q (waves 2, 3) against S^T
        f32x4_v acc1 = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            const float* xr = (wave < 2 ? sK : sQ) + (16 * (wave & 1) + i16) * KP + 32 * kq;
            const float* sr = sS + i16 * KP + 32 * kq;
            f32x4_v xa[8], sb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xa[q] = *(const f32x4_v*)(xr + 4 * q), sb[q] = *(const f32x4_v*)(sr + 4 * q);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1 = mfma4(xa[q][e], sb[q][e], acc1);
            if (wave < 2) { // R = V - A (K S^T)
#pragma unroll
                for (int r = 0; r < 4; ++r) sR[(v_row + r) * RPP + i16] = vreg[r] - sA[v_row + r] * acc1[r];
            }
        }
        lds_barrier();
        // ---- stage 2 (waves 0, 1): D = T R
        if (active && wave < 2) {
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
        // ---- stage 3 (waves 2, 3): O = A (Q S^T) + P D
        if (active && wave >= 2) {
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
                if (t < store_end) {
                    o32[(size_t)t * o_stride] = o[r];
                    out16[(size_t)t * value_dim] = f32_to_bf16(o[r]);
                }
            }
        }
        // ---- stage 4: S = A_C S + (D W)^T K
        if (active) {
            const float a_c = sA[CC - 1];
            float dw[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) dw[s2] = sDw[(8 * kq + s2) * RPP + i16];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) creg[t][r] *= a_c;
                const float* kc = sK + (8 * kq) * KP + 32 * wave + 16 * t + i16;
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) creg[t] = mfma4(dw[s2], kc[s2 * KP], creg[t]);
            }
        }
        lds_barrier();
        if (c + 1 < c_end) publish();
        lds_barrier();
    }
    if (active) {
        float* sfin = homog  ? sp.h_end + ((size_t)hv * DKC + dv_base + 4 * kq) * DKC + 32 * wave + i16
                      : seg1 ? sp.z_end + ((size_t)hv * head_v_dim + dv_base + 4 * kq) * DKC + 32 * wave + i16
                             : sp.mid_state + ((size_t)hv * head_v_dim + dv_base + 4 * kq) * DKC + 32 * wave + i16;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sfin[(size_t)r * DKC + 16 * t] = creg[t][r];
    }
}
constexpr size_t kDualLdsBytes = (size_t)(2 * CC * KP + 2 * CC * 36 + 2 * CC + 2 * (16 * KP + 3 * CC * 18)) * sizeof(float);

// The S_mid terms of a split scan (see ScanSplit).  grid (tok_blocks + Dk / 16, Hv), 4 waves;
// v_mfma_f32_16x16x4_f32 with the operand layout of the scan (lane l: row / column l % 16, contraction indices 32 (l / 16) + s over 32 steps: a lane's
// 32 operand values are consecutive in memory).  Dv = Dk = 128 (split_mid_chunk).
//   the first tok_blocks:    64 tokens of segment 1:   out[t, dv] = bf16(o32[t, dv] + sum_dk u[t, dk] S_mid[dv, dk])    wave w = tokens 16 w ..: A = its u rows,
//                                                      B = S_mid rows from LDS (the head's state staged once per workgroup: one global round trip in all)
//   the last Dk / 16 blocks: 16 state columns:         S[dv, dk] = z_end[dv, dk] + sum_j S_mid[dv, j] h_end[j, dk]       A = S_mid rows, B = h_end columns
constexpr int FIX_TOK = 64;
constexpr size_t kFixLdsBytes = (size_t)DKC * KP * sizeof(float); // S_mid of one head, rows 4 banks apart
__global__ void __launch_bounds__(256) dn_chunk_fixup_kernel(ScanSplit sp, float* state, uint16_t* out, uint32_t num_v_heads, uint32_t head_v_dim, uint32_t value_dim,
                                                             uint32_t suffix_len) {
    constexpr uint32_t DV_TILES = DKC / 16;
    extern __shared__ __attribute__((aligned(16))) float fix_smem[]; // [Dv][KP]: S_mid of the head (token blocks)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    const uint32_t hv = blockIdx.y, t_mid = sp.mid_chunk * CC, T2 = suffix_len - t_mid;
    const uint32_t tok_blocks = (T2 + FIX_TOK - 1) / FIX_TOK;
    const float* smid = sp.mid_state + (size_t)hv * head_v_dim * DKC;
    auto mfma4 = [](float a, float b, f32x4_v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); };
    // the token blocks come first in the grid: they are the long ones (one global round trip + 256 matrix instructions per wave)
    if (blockIdx.x < tok_blocks) {
        // every global operand is requested before anything is consumed: S_mid (16 x 16 bytes per thread, through LDS), this wave's u rows, its o32 values
        f32x4_v sm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sm[r] = *(const f32x4_v*)(smid + (size_t)(tid + 256 * r) * 4);
        const uint32_t tok = blockIdx.x * FIX_TOK + 16 * wave; // this wave's 16 tokens, relative to segment 1 (past the end: computed on clamped rows, never stored)
        const size_t u_stride = (size_t)num_v_heads * DKC;
        const uint32_t row = min(tok + (uint32_t)i16, T2 - 1); // A: row = token
        const float* ur = sp.u + (size_t)row * u_stride + hv * DKC + 32 * kq;
        f32x4_v ua[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ua[q] = *(const f32x4_v*)(ur + 4 * q);
        float prev[DV_TILES][4]; // this lane's o32 values: rows 4 kq + r, column 16 vt + i16
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t t = min(tok + 4 * kq + r, T2 - 1);
            const float* pr = sp.o32 + (size_t)(t_mid + t) * value_dim + (size_t)hv * head_v_dim + i16;
#pragma unroll
            for (uint32_t vt = 0; vt < DV_TILES; ++vt) prev[vt][r] = pr[vt * 16];
        }
        __builtin_amdgcn_sched_barrier(0); // (every request is out before the first wait)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t e = (tid + 256 * r) * 4; // element of the [Dv][Dk] state
            *(f32x4_v*)(fix_smem + (e / DKC) * KP + e % DKC) = sm[r];
        }
        lds_barrier();
#pragma unroll
        for (uint32_t vt = 0; vt < DV_TILES; ++vt) {
            const float* br = fix_smem + (vt * 16 + i16) * KP + 32 * kq; // B: column dv = 16 vt + i16
            f32x4_v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4_v b = *(const f32x4_v*)(br + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma4(ua[q][e], b[e], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { // rows = tokens 4 kq + r of the tile, column = dv
                const uint32_t t = tok + 4 * kq + r;
                if (t < T2) out[(size_t)(t_mid + t) * value_dim + (size_t)hv * head_v_dim + vt * 16 + i16] = f32_to_bf16(prev[vt][r] + acc[r]);
            }
        }
    } else {
        const uint32_t dk0 = (blockIdx.x - tok_blocks) * 16;
        const float* hend = sp.h_end + (size_t)hv * DKC * DKC;
        float hb[32]; // B: column dk = dk0 + i16, contraction index j = 32 kq + s
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) hb[s2] = hend[(size_t)(32 * kq + s2) * DKC + dk0 + i16];
        // this wave's two value tiles (wave, wave + 4): all of their S_mid rows and z_end values are requested before the first product (read in place, every
        // 16-byte operand was a round trip of its own in front of four matrix instructions)
        f32x4_v sa[2][8];
        float zv[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t vt = wave + 4 * h;
            const float* ar = smid + (size_t)(vt * 16 + i16) * DKC + 32 * kq; // A: row dv = 16 vt + i16
#pragma unroll
            for (int q = 0; q < 8; ++q) sa[h][q] = *(const f32x4_v*)(ar + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) zv[h][r] = sp.z_end[((size_t)hv * head_v_dim + vt * 16 + 4 * kq + r) * DKC + dk0 + i16];
        }
        __builtin_amdgcn_sched_barrier(0); // (the scheduler would sink the loads back to their uses)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t vt = wave + 4 * h;
            f32x4_v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma4(sa[h][q][e], hb[4 * q + e], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) // rows dv = 16 vt + 4 kq + r, column dk
                state[((size_t)hv * head_v_dim + vt * 16 + 4 * kq + r) * DKC + dk0 + i16] = zv[h][r] + acc[r];
        }
    }
}

bool delta_net_prefill_chunked_supported(uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t suffix_len) {
    if (exact_mode()) return false; // reference-order mode: the token-by-token recurrence in the reference's own loop order (k_exact.hip)
    static const uint32_t min_t = [] {
        const char* e = lab_env("UZU_DN_CHUNK_MIN_T");
        return e ? (uint32_t)atoi(e) : 64u;
    }();
    return head_k_dim == DKC && num_k_heads && num_v_heads % num_k_heads == 0 && head_v_dim % 16 == 0 && suffix_len >= min_t;
}

// prep != null: DeltaNetPrefillPrep fused into the chunk preparation (q_norm / k_norm are then OUTPUTS of this call; beta / decay are not materialised)
struct FusedPrep {
    const float *a_log, *dt_bias;
    const uint16_t* conv_rows; // non-null: the conv'd q | k | v channels, [suffix_len][conv_dim] (out-of-place conv)
    float* conv_state;
};
static uzu_status prefill_chunked(hipStream_t s, float* q_norm, float* k_norm, const float* beta, const float* decay, const uint16_t* in_proj, float* state, uint16_t* out,
                                  float* workspace, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim, uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len,
                                  const FusedPrep* prep) {
    const uint32_t n_chunks = (suffix_len + CC - 1) / CC;
    const uint16_t* v_rows = prep && prep->conv_rows ? prep->conv_rows : in_proj; // where the scans read the conv'd v channels
    const uint32_t v_stride = prep && prep->conv_rows ? 2 * key_dim + value_dim : 0u;
    if (prep)
        UZU_PROPAGATE(launch_check([&] {
            hipLaunchKernelGGL(dn_chunk_prep_kernel<true>, dim3(n_chunks, num_v_heads), dim3(256), 0, s, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, workspace, num_v_heads, num_k_heads, key_dim, suffix_len, in_proj, prep->a_log, prep->dt_bias, q_norm, k_norm, value_dim, prep->conv_rows,
                               prep->conv_state);
        }, "delta_net_chunk_prep_fused"));
    else
        UZU_PROPAGATE(launch_check([&] {
            hipLaunchKernelGGL(dn_chunk_prep_kernel<false>, dim3(n_chunks, num_v_heads), dim3(256), 0, s, (const float*)q_norm, (const float*)k_norm, beta, decay, workspace,
                               num_v_heads, num_k_heads, key_dim, suffix_len, (const uint16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr,
                               (float*)nullptr, value_dim, (const uint16_t*)nullptr, (float*)nullptr);
        }, "delta_net_chunk_prep"));
    ScanSplit sp{};
    sp.mid_chunk = value_dim == num_v_heads * head_v_dim ? split_mid_chunk(n_chunks, head_v_dim) : 0u;
    static LdsLimit dual_lds;
    static LdsLimit fix_lds;
    if (sp.mid_chunk && !(raise_lds_limit(dual_lds, (const void*)dn_chunk_scan_dual_kernel, kDualLdsBytes) && raise_lds_limit(fix_lds, (const void*)dn_chunk_fixup_kernel, kFixLdsBytes)))
        sp.mid_chunk = 0; // (74 KB / 68 KB of LDS per workgroup)
    if (!sp.mid_chunk)
        return launch_check([&] {
            hipLaunchKernelGGL(dn_chunk_scan_mfma_kernel, dim3(head_v_dim / 16, num_v_heads), dim3(256), 0, s, q_norm, k_norm, v_rows, workspace, state, out,
                               num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim, suffix_len, v_stride);
        }, "delta_net_chunk_scan");
    { // the split's pieces live behind the T / P matrices (delta_net_chunk_workspace_bytes)
        float* w = workspace + (size_t)n_chunks * num_v_heads * WS_FLOATS;
        sp.mid_state = w, w += (size_t)value_dim * DKC;
        sp.z_end = w, w += (size_t)value_dim * DKC;
        sp.h_end = w, w += (size_t)num_v_heads * DKC * DKC;
        sp.o32 = w, w += (size_t)suffix_len * value_dim;
        sp.u = w;
    }
    UZU_PROPAGATE(launch_check([&] {
        hipLaunchKernelGGL(dn_chunk_scan_dual_kernel, dim3(2 * (head_v_dim / 16), num_v_heads), dim3(512), kDualLdsBytes, s, q_norm, k_norm, v_rows, workspace,
                           (const float*)state, out, num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim, suffix_len, sp, v_stride);
    }, "delta_net_chunk_scan_dual"));
    const uint32_t t_mid = sp.mid_chunk * CC, t2 = suffix_len - t_mid;
    return launch_check([&] {
        hipLaunchKernelGGL(dn_chunk_fixup_kernel, dim3((t2 + FIX_TOK - 1) / FIX_TOK + DKC / 16, num_v_heads), dim3(256), kFixLdsBytes, s, sp, state, out,
                           num_v_heads, head_v_dim, value_dim, suffix_len);
    }, "delta_net_chunk_fixup");
}

uzu_status delta_net_prefill_chunked(hipStream_t s, const float* q_norm, const float* k_norm, const float* beta, const float* decay, const uint16_t* in_proj,
                                     float* state, uint16_t* out, float* workspace, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim,
                                     uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len) {
    return prefill_chunked(s, (float*)q_norm, (float*)k_norm, beta, decay, in_proj, state, out, workspace, num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim, suffix_len, nullptr);
}
// UZU_HIP_TUNE=prep_fused=0: DeltaNetPrefillPrep as its own launch in front (tests/test_gpu_prefill_switches.py holds the fused kernel to it, bit for bit)
// UZU_HIP_TUNE=prep_fused=0: DeltaNetPrefillPrep as its own launch in front (tests/test_gpu_prefill_switches.py holds the fused kernel to it, bit for bit)
bool delta_net_prefill_prep_fused_enabled() {
    const char* e = tune_env("prep_fused");
    return !e || atoi(e) != 0;
}
uzu_status delta_net_prefill_chunked_fused(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias, float* q_norm_out, float* k_norm_out, float* state,
                                           uint16_t* out, float* workspace, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim, uint32_t key_dim, uint32_t value_dim,
                                           uint32_t suffix_len, const uint16_t* conv_rows, float* conv_state) {
    const FusedPrep prep{a_log, dt_bias, conv_rows, conv_rows ? conv_state : nullptr};
    return prefill_chunked(s, q_norm_out, k_norm_out, nullptr, nullptr, in_proj, state, out, workspace, num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim, suffix_len, &prep);
}

} // namespace k
} // namespace uzu
