// k_gemm128.hip -- the large-tile prefill MatmulKernel for gfx950 (M >= 128 rows): 128 x 128 x 64 workgroup tile,
// weights never touch LDS.  Same reference semantics as k_gemm.hip (BU/cpu/kernel/matmul/kernel.rs:164-293).
//
//   * Four waves as 2 x 2, each wave 64 x 64 = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16.  The k order inside a k-step
//     is free as long as A and B agree: lanes 0..31 take k = 8s..8s+7, lanes 32..63 take k = 32+8s..32+8s+7 at MFMA
//     s of a 64-wide k-step, so the B operand of a lane is one 16-byte vector of its column's packed codes: global ->
//     VGPR -> exact centred bf16 codes (v_cvt_off_f32_i4 / sext byte converts, as in k_gemm.hip) -> MFMA.  Only the
//     activations are staged through LDS (double buffered, 144-byte pitch), one LDS-only barrier per k-step.
//   * A quant group accumulates into acc_g (first MFMA of the group takes C = 0); at the group boundary
//     acc_t += scale[n,g] * acc_g on the VALU (packed f32 FMAs).
//   * The offset term  sum_g coef[n,g] * rowsum_g(A)[m]  is added once, in the epilogue: a pre-pass writes the group
//     row sums of A (f32, [group][row]), and every lane runs G packed FMAs per pair of its 64 outputs -- no row-sum
//     MFMAs and no per-group offset work in the main loop.
//   * Every global load is unconditional (rows / columns / steps clamped into range) and prefetched into a register
//     ring; the loop is unrolled by the ring depth so that all indices are static (vmcnt waits stay counted).
//   * Tiles are numbered in 8 x 8 super-tiles per XCD (tile_map) so that the ~64 workgroups an XCD runs at a time share
//     their activation and weight panels in its L2; few-tile shapes split K over gridDim.y with f32 partial tiles and
//     a fixed-order reduction.  bf16 results leave through LDS as 16-byte row segments.
#include <stdlib.h>

#include <type_traits>

#include "device_utils.h"
#include "gemm_convert.h"
#include "gemm_tile_map.h"
#include "kernels.h"
#include "uzu_math.h"

namespace uzu {
namespace k {


#ifndef UZU_GEMM_PAIR_DEQUANT
#define UZU_GEMM_PAIR_DEQUANT 1 // int4: bf16(16 + u) pairs (gemm_convert.h::dequant4_pairs); 0 = centred (u - 8) / 16 per code (A/B builds)
#endif
namespace {
constexpr int BK = 64, BM = 128, BN = 128;
constexpr int A_PITCH = 144;
constexpr bool kPairs = UZU_GEMM_PAIR_DEQUANT != 0;
#ifndef UZU_GEMM_PP_DB
#define UZU_GEMM_PP_DB 4 // ping-pong form: weight ring depth
#endif
#ifndef UZU_GEMM_PP_DA
#define UZU_GEMM_PP_DA 2 // ping-pong form: activation register stages
#endif
#ifndef UZU_GEMM_FOLD_PK
#define UZU_GEMM_FOLD_PK 0 // group fold with v_pk_fma_f32 (1) or pairs of v_fma_f32 (0)
#endif
#ifndef UZU_GEMM_PP_MSCHED
#define UZU_GEMM_PP_MSCHED 2 // ping-pong form, MFMA phase: 0 = operands | 16 MFMAs | staging; 1 = staging interleaved with the MFMAs; 2 = staging first
#endif
#ifndef UZU_GEMM_PIPE
#define UZU_GEMM_PIPE 1 // software-pipelined k16 steps (operands one step ahead of the MFMAs); 0 = operands right in front of them
#endif

} // namespace

// ---------------------------------------------------------------------------------------------- pre-pass
// Blocks [0, rowsum_blocks): rowsum[g][m] = sum of A[m, k in group g] (f32), row stride Mp = M rounded up to 4 (pad rows are
// zero); the remaining blocks: coef[g][n].  Row sums: one wave per row; lane l owns the 8-element chunks l, l + 64, ...;
// the lanes of a group (group_size / 8, a power of two <= 64) are reduced with xor shuffles in a fixed order.
__global__ void __launch_bounds__(256) gemm_prepass_kernel(MatmulParams p, float* rowsum, float* coef, uint32_t rowsum_blocks) {
    const uint32_t M = p.m, K = p.k, N = p.n, group_size = p.group_size, G = K / group_size;
    if (blockIdx.x >= rowsum_blocks) {
        // coef[g][n] (f32, the layout the epilogue's f32 MFMA operand wants: lanes = consecutive columns):
        // ScaleBias: bias + mid * scale, ZeroPoint: scale * (mid - zp)   (codes were fed centred on mid = 2^(bits-1))
        const uint32_t idx = (blockIdx.x - rowsum_blocks) * 256 + threadIdx.x;
        if (idx >= N * G) return;
        const uint32_t n = idx / G, g = idx % G; // consecutive threads read consecutive scales
        // `mid` = what the main loop subtracted from the unsigned code: 2^(bits-1) (centred codes) or -16 (int4 pair form: 16 + u)
        const float mid = (kPairs && p.bits == 4) ? -16.0f : (float)(1u << (p.bits - 1));
        const float scale = bf16_to_f32(((const uint16_t*)p.scales)[(size_t)n * G + g]);
        float c;
        if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) {
            const uint32_t zp_stride = p.bits == 4 ? (G + 1) / 2 : G;
            const uint32_t zb = p.zero_points[(size_t)n * zp_stride + (p.bits == 4 ? (g >> 1) : g)];
            c = scale * (mid - (float)(p.bits == 4 ? ((g & 1) ? (zb >> 4) : (zb & 0xF)) : zb));
        } else if (p.b_kind == UZU_MATMUL_B_SCALE_SYMMETRIC) {
            c = scale * (mid - (float)(1u << (p.bits - 1))); // w = scale * (u - 2^(bits-1)); zero for centred codes
        } else {
            c = fmaf(mid, scale, bf16_to_f32(((const uint16_t*)p.biases)[(size_t)n * G + g]));
        }
        coef[(size_t)g * N + n] = c;
        return;
    }
    const uint16_t* a = (const uint16_t*)p.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t m = blockIdx.x * 4 + wave, Mp = (M + 3) & ~3u;
    if (m >= Mp) return;
    const uint32_t lpg = group_size / 8;
    if (m >= M) { // pad rows are read (never stored) by the edge tiles
        for (uint32_t g = lane; g < G; g += 64) rowsum[(size_t)g * Mp + m] = 0.f;
        return;
    }
    const uint16_t* row = a + (size_t)m * K;
    for (uint32_t base = 0; base < K / 8; base += 64) {
        const uint32_t chunk = base + lane;
        float s = 0.f;
        if (chunk < K / 8) {
            const u32x4_v v = *(const u32x4_v*)(row + (size_t)chunk * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) s0 += bits_to_f32(w[i] << 16), s1 += bits_to_f32(w[i] & 0xFFFF0000u);
            s = s0 + s1;
        }
        for (uint32_t off = 1; off < lpg; off <<= 1) s += __shfl_xor(s, (int)off, 64);
        if (chunk < K / 8 && lane % lpg == 0) rowsum[(size_t)(chunk / lpg) * Mp + m] = s;
    }
}

// ---------------------------------------------------------------------------------------------- main kernel
// grid (8 S ceil(Q / 8), splits), see tile_map().  GS = k-steps per quant group (1, 2, 4 <=> group 64, 128, 256).
//
// PP ("ping-pong", round 5): ONE 512-thread workgroup per CU works on a 128 x 256 tile: two halves of four waves, each half the 128 x 128
// tile of the 256-thread form (same k order, same fold -> bit-identical results) over ONE activation tile in LDS that all eight waves
// stage (half the L2 -> L1 activation traffic of two independent workgroups).  Wave w and wave w + 4 share a SIMD.  A k-step is cut into
// a CONVERT phase (weight codes of the step -> bf16 fragments, ONCE per half: the two waves of a column pair split the work and share the
// fragments through LDS; the group fold; the weight prefetch: VALU / VMEM / 4 ds_write only) and an
// MFMA phase (16 ds_read_b128 + 16 back-to-back v_mfma_f32_32x32x16_bf16, with the staging of the next activation tile -- 2 global loads,
// 8 v_perm, 2 ds_write per thread -- in their shadow); an s_barrier over all eight waves after every phase keeps the halves half a step
// apart, so a SIMD always holds one wave that wants the matrix pipe and one that wants the vector ALU.
//
// WS ("wave-specialised", MODE 2): ONE 512-thread workgroup per CU on a 128 x 128 tile.  Waves 0-3 are CONSUMERS (the 2 x 2 wave tiles of the
// 256-thread form: nothing but ds_read_b128 + MFMA + the group fold; every operand, weights included, comes out of LDS), waves 4-7 -- the
// second wave of each SIMD -- are PRODUCERS: they stage the activation tile of the NEXT k-step and convert its weight codes (one 32-column
// block per wave, once per workgroup) into the other LDS stage while the consumers work on this one; one s_barrier per k-step swaps the
// stages.  The matrix pipe of a SIMD belongs to one wave that never converts; the vector ALU work runs beside it in a wave that never
// multiplies.  Same k order and fold as the other forms -> bit-identical results.
template <int BITS, int GS, int MODE>
__global__ void __launch_bounds__(MODE ? 512 : 256, MODE ? 1 : 2) gemm_q_mfma128_kernel(MatmulParams p, const float* rowsum, const float* coef, float* partials, unsigned long long* dbg) {
    constexpr bool PP = MODE == 1, WS = MODE == 2;
    constexpr int WV = BITS / 4;           // 16-byte code vectors per lane per 32-column block per k-step
    constexpr int DB = BITS == 8 ? 2 : PP ? UZU_GEMM_PP_DB : 4;  // weight ring depth (k-steps in flight + 1)
    constexpr int DA = PP ? UZU_GEMM_PP_DA : 2;                  // activation register stages
    constexpr int U = 4;                   // unroll: a multiple of DB, DA, 2 (LDS buffers) and GS
    constexpr int NH = MODE ? 2 : 1;
    __shared__ __attribute__((aligned(16))) uint8_t s_a_all[NH][2][BM * A_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t s_b[MODE ? 2 : 1][MODE ? 16384 : 16]; // PP: a half's converted weight fragments of one k-step; WS: the two stages
    __shared__ uint64_t s_exp_tab[32]; // gated epilogue only

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = PP ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
    uint8_t(*const s_a)[BM * A_PITCH] = s_a_all[0];                       // activation tiles of the main loop (PP: shared by the halves)
    const bool producer = WS && threadIdx.x >= 256;
    uint8_t* const s_ep = &s_a_all[PP ? half ^ 1 : WS ? 1 : 0][0][0];                // epilogue staging (PP: half 0 finishes first and must not touch the tiles half 1 still reads)
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, c = lane & 31;
    const uint32_t M = p.m, N = p.n, K = p.k;
    // GatedActMul fused into the epilogue (p.act_mul; the up projection's rows [0, N/2) = up, [N/2, N) = gate, gated_act_mul.rs:52-58):
    // a workgroup takes 64 up columns (its wn = 0 waves) and the 64 gate columns of the same outputs (wn = 1), so the two operands
    // of an output sit in the two wave tiles of one workgroup and D is [M, N/2].
    const bool gated = p.act_mul != 0;
    const uint32_t H = N / 2;
    const uint32_t m_tiles = (M + BM - 1) / BM, n_tiles = gated ? (H + 63) / 64 : (N + BN - 1) / BN;
    // Workgroups are dealt to the 8 XCDs round-robin and an XCD runs ~64 of them at a time, so tile numbering decides
    // what each 4 MB L2 sees: XCD x works through super-tiles x, x + 8, ... of TM x TN tiles (8 x 8 when the matrix is
    // large enough) -- the 64 resident workgroups then share 8 activation row blocks and 8 weight column blocks.
    uint32_t m_t, n_t;
    bool live;
    if (PP) { // the tile map runs over PAIRS of column tiles; half h takes column tile 2 pair + h of the same row block
        uint32_t pr;
        if (!gemm_tile_of_block(blockIdx.x, m_tiles, (n_tiles + 1) / 2, &m_t, &pr)) return;
        n_t = 2 * pr + half;
        live = n_t < n_tiles;
        if (!live) n_t = n_tiles - 1; // odd tile count: the idle half walks its neighbour's tile for the barriers' sake and stores nothing
    } else {
        live = gemm_tile_of_block(blockIdx.x, m_tiles, n_tiles, &m_t, &n_t);
        if (!live) return;
    }
    const uint32_t vblock = PP ? blockIdx.x * 2 + half : blockIdx.x;
    const uint32_t Mst = live ? M : 0u; // row bound of every store
    const uint32_t xcd = blockIdx.x & 7;
    unsigned long long ts[4];
    ts[0] = wall_clock64();
    const uint32_t m0 = m_t * BM, n0 = n_t * BN;
    const uint32_t G = K / p.group_size;
    const uint32_t splits = gridDim.y, z = blockIdx.y;
    const uint32_t Gz = G / splits, g_lo = z * Gz;
    const uint32_t kt_lo = g_lo * GS, KTz = Gz * GS; // KTz % U == 0 (host-checked)
    const uint32_t row_bytes = K * BITS / 8;
    // codes as the conversion wants them: two's complement of u - mid (centred form) or the unsigned u (int4 pair form)
    const uint32_t flip = (kPairs && BITS == 4) ? (p.signed_codes ? 0x88888888u : 0u) : p.signed_codes ? 0u : (BITS == 4 ? 0x88888888u : 0x80808080u);

    // ---- activation staging role: 8 lanes fetch one row's 128 bytes (one cache line per row per instruction), four
    // passes of 32 rows.  (One thread per (row, 64-byte half) costs 45 L1 accesses per wave instruction -- rocprofv3
    // TCP_TOTAL_CACHE_ACCESSES / TA_FLAT_READ_WAVEFRONTS -- and saturates the texture addresser at 4096^2-sized shapes.)
    // PP: all 512 threads stage the one tile, two passes of 64 rows.
    constexpr int NR = PP ? 2 : 4, RSTEP = PP ? 64 : 32;
    const int stid = PP ? (int)threadIdx.x : tid;
    const int chunk = stid & 7, rpass = stid >> 3;
    uint32_t a_off[NR]; // element offsets of the thread's rows
#pragma unroll
    for (int j = 0; j < NR; ++j) a_off[j] = min(m0 + RSTEP * j + (uint32_t)rpass, M - 1) * K + kt_lo * BK + 8 * chunk;
    const uint16_t* a_base = (const uint16_t*)p.a;
    u32x4_v a_st[DA][NR];
    auto load_a = [&](uint32_t kt, u32x4_v (&st)[NR]) {
        kt = min(kt, KTz - 1); // past the end: re-read the last tile (never staged into a buffer that is read)
#pragma unroll
        for (int j = 0; j < NR; ++j) st[j] = *(const u32x4_v*)((a_base + (size_t)kt * BK) + a_off[j]); // uniform base + 32-bit lane offset: no 64-bit vector address arithmetic
    };
    auto stage_a = [&](uint32_t kt, const u32x4_v (&st)[NR]) {
        uint8_t* dst = &s_a[kt & 1][rpass * A_PITCH + chunk * 16];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            u32x4_v v = (kPairs && BITS == 4) ? permute_pairs(st[j]) : st[j];
#ifdef UZU_GEMM_LAB_SWAP // LAB: the wave-specialised form with two k slots of every operand exchanged -- is the MFMA's sum independent of the slot order?
            if (WS) { const uint32_t t = v.x; v.x = v.w, v.w = t; }
#endif
            *(u32x4_v*)(dst + j * RSTEP * A_PITCH) = v;
        }
    };

    // ---- weight role: lane -> column c of each of the wave's two 32-column blocks, k half h
    uint32_t ncol[2];
    uint32_t w_off[2]; // byte offsets (< 2^32, host-checked) from a uniform base
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        ncol[nb] = gated ? (wn ? H : 0u) + min(n_t * 64 + nb * 32 + c, H - 1) : min(n0 + wn * 64 + nb * 32 + c, N - 1);
        w_off[nb] = ncol[nb] * row_bytes + (32 * h) * BITS / 8;
    }
    u32x4_v ring[DB][2][WV];
    auto load_w = [&](uint32_t kt, u32x4_v (&r)[2][WV]) {
        kt = min(kt, KTz - 1);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const u32x4_v* src = (const u32x4_v*)(((const uint8_t*)p.b + (size_t)(kt_lo + kt) * BK * BITS / 8) + w_off[nb]);
#pragma unroll
            for (int v = 0; v < WV; ++v) r[nb][v] = src[v];
        }
    };
    const uint16_t* scales = (const uint16_t*)p.scales;

    f32x16_t acc_g[2][2], acc_t[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_g[mb][nb][r] = 0.f, acc_t[mb][nb][r] = 0.f;

    // group scales: sc_cur = group being accumulated, sc_nxt = the next one (requested one group ahead)
    // (32-bit registers: as uint16_t pairs the compiler packs the two halves into one VGPR with a v_perm right behind the loads, i.e. an
    // s_waitcnt vmcnt(0) -- a full memory round trip, ~1000 cycles -- in every group fold: the ping-pong form's phase stamps, round 5)
    uint32_t sc_cur[2], sc_nxt[2];
    auto load_scale = [&](uint32_t g, uint32_t (&dst)[2]) {
        g = min(g_lo + g, G - 1);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) dst[nb] = (scales + g)[ncol[nb] * G];
    };

    const uint8_t* a_frag_base = &s_a[0][(wm * 64 + c) * A_PITCH + h * 64];
    // one k-step of code MFMAs out of LDS buffer (kt & 1) and a ring slot; `first` = first k-step of a quant group
    auto mfma_codes = [&](uint32_t kt, const u32x4_v (&raw)[2][WV], bool first) {
        const uint8_t* ab = a_frag_base + (kt & 1) * (BM * A_PITCH);
        auto operands = [&](int s, u32x4_t (&bf)[2], u32x4_t (&af)[2]) { // B fragments converted from the ring slot, A fragments from LDS
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) af[mb] = *(const u32x4_t*)(ab + mb * 32 * A_PITCH + s * 16);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
#ifdef UZU_GEMM_LAB_NOCONV // LAB (wrong results): one and-or per fragment instead of nine instructions -- the loop without its conversion work (finite operands: arbitrary bit patterns cost clock)
                { const uint32_t one = (raw[nb][0][s] & 0x00010001u) | 0x3f803f80u; bf[nb] = u32x4_t{one, one, one, one}; } // bf16 1.0 / 1.0078: finite, benign
#else
                if (BITS == 4) bf[nb] = kPairs ? dequant4_pairs(raw[nb][0][s] ^ flip) : dequant4(raw[nb][0][s] ^ flip);
                else bf[nb] = dequant8(raw[nb][s >> 1][(s & 1) * 2] ^ flip, raw[nb][s >> 1][(s & 1) * 2 + 1] ^ flip);
#endif
            }
        };
#if UZU_GEMM_PIPE
        // operands of k16 step s + 1 are requested / converted before the MFMAs of step s are issued: the LDS latency and the
        // conversion VALU run under the 4 x 32 cycles of the matrix pipe instead of in front of it
        u32x4_t bf[2][2], af[2][2];
        operands(0, bf[0], af[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) operands(s + 1, bf[(s + 1) & 1], af[(s + 1) & 1]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const f32x16_t zero = {};
                    acc_g[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[s & 1][mb]), __builtin_bit_cast(bf16x8_t, bf[s & 1][nb]),
                                                                           (first && s == 0) ? zero : acc_g[mb][nb], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4_t bf[2], af[2];
            operands(s, bf, af);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const f32x16_t zero = {};
                    acc_g[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[mb]), __builtin_bit_cast(bf16x8_t, bf[nb]),
                                                                           (first && s == 0) ? zero : acc_g[mb][nb], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0); // keep the conversions of MFMA s next to their use (register pressure)
        }
#endif
    };
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#if defined(UZU_GEMM_PP_TIMING)
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmark = 0; // shader cycles per phase (ping-pong form)
#define UZU_PP_MARK(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn = clock64(); tsum[i] += tn - tmark; tmark = tn; __builtin_amdgcn_sched_barrier(0); }
#if UZU_GEMM_PP_TIMING >= 2
#define UZU_FOLD_MARK(i) UZU_PP_MARK(i)
#define UZU_PP_MARK2(i) UZU_PP_MARK(i)
#else
#define UZU_FOLD_MARK(i)
#define UZU_PP_MARK2(i)
#endif
#else
#define UZU_PP_MARK2(i)
#define UZU_PP_MARK(i)
#define UZU_FOLD_MARK(i)
#endif
    auto fold = [&](bool hazard_wait = true) { // acc_t += scale * acc_g at a group boundary
        f32x2_t sc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) sc[nb].x = sc[nb].y = ((BITS == 4 && !kPairs) ? 16.0f : 1.0f) * bits_to_f32(sc_cur[nb] << 16);
        // In place on acc_t through inline asm: left to itself the register allocator writes the result over acc_g and
        // permutes the 16-register accumulator tuples around the loop (60-180 VGPRs of spills at the 256 budget).  The
        // hazard recogniser cannot see an MFMA -> VALU read through inline asm, so the wait for the last MFMA of the
        // group (at most 16 passes: 18 wait states) is spelled out.
        if (hazard_wait) asm volatile("s_nop 15\n\ts_nop 3");
        UZU_FOLD_MARK(6)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
#ifdef UZU_GEMM_LAB_NOFOLD // LAB (wrong results): two of a block's sixteen registers folded -- the MFMAs stay live, 7/8 of the fold's FMAs are gone
                for (int r = 0; r < 2; r += 2) {
#else
                for (int r = 0; r < 16; r += 2) {
#endif
#if UZU_GEMM_FOLD_PK == 1
                    f32x2_t t = {acc_t[mb][nb][r], acc_t[mb][nb][r + 1]};
                    const f32x2_t g = {acc_g[mb][nb][r], acc_g[mb][nb][r + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(sc[nb]), "v"(g));
                    acc_t[mb][nb][r] = t.x, acc_t[mb][nb][r + 1] = t.y;
#else
                    float t0 = acc_t[mb][nb][r], t1 = acc_t[mb][nb][r + 1];
#if UZU_GEMM_FOLD_PK == 2
                    asm volatile("v_fmac_f32_e32 %0, %2, %3\n\tv_fmac_f32_e32 %1, %2, %4" : "+v"(t0), "+v"(t1) : "v"(sc[nb].x), "v"(acc_g[mb][nb][r]), "v"(acc_g[mb][nb][r + 1]));
#else
                    asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %4, %1" : "+v"(t0), "+v"(t1) : "v"(sc[nb].x), "v"(acc_g[mb][nb][r]), "v"(acc_g[mb][nb][r + 1]));
#endif
                    acc_t[mb][nb][r] = t0, acc_t[mb][nb][r + 1] = t1;
#endif
                }
        __builtin_amdgcn_sched_barrier(0); // the next group's first MFMA must not be hoisted above the fold (it would need a second acc_g)
        UZU_FOLD_MARK(7)
    };

    if constexpr (WS) {
        auto step_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        if (producer) {
            // ---- producers: wave pw owns the 32-column block pw of the tile (block = wn * 2 + nb of the consumers' numbering)
            const int pw = wave, pwn = pw >> 1, pnb = pw & 1;
            const uint32_t pcol = gated ? (pwn ? H : 0u) + min(n_t * 64 + pnb * 32 + c, H - 1) : min(n0 + pwn * 64 + pnb * 32 + c, N - 1);
            const uint32_t w_own = pcol * row_bytes + (32 * h) * BITS / 8;
            u32x4_v ringp[DB][WV];
            auto load_w1 = [&](uint32_t kt, u32x4_v (&r)[WV]) {
                kt = min(kt, KTz - 1);
                const u32x4_v* src = (const u32x4_v*)(((const uint8_t*)p.b + (size_t)(kt_lo + kt) * BK * BITS / 8) + w_own);
#pragma unroll
                for (int v = 0; v < WV; ++v) r[v] = src[v];
            };
            auto convert_to = [&](uint32_t kt, const u32x4_v (&raw)[WV]) { // fragments of k-step kt -> stage kt & 1, [block][k16 step][lane]
                uint8_t* dst = &s_b[kt & 1][(pw * 4) * 1024 + lane * 16];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    u32x4_t f;
                    if (BITS == 4) f = kPairs ? dequant4_pairs(raw[0][s] ^ flip) : dequant4(raw[0][s] ^ flip);
                    else f = dequant8(raw[s >> 1][(s & 1) * 2] ^ flip, raw[s >> 1][(s & 1) * 2 + 1] ^ flip);
#ifdef UZU_GEMM_LAB_SWAP
                    { const uint32_t t = f.x; f.x = f.w, f.w = t; }
#endif
                    *(u32x4_t*)(dst + s * 1024) = f;
                }
            };
            {
                u32x4_v first[NR];
                load_a(0, first);
#pragma unroll
                for (int u = 1; u <= DA; ++u) load_a(u, a_st[u % DA]); // slot (kt + 1) % DA holds tile kt + 1
#pragma unroll
                for (int u = 0; u < DB; ++u) load_w1(u, ringp[u]);
                stage_a(0, first);
                convert_to(0, ringp[0]);
                load_w1(DB, ringp[0]);
            }
            step_barrier(); // stage 0 is complete
#ifdef UZU_GEMM_PP_TIMING
            tmark = clock64();
#endif
            for (uint32_t kt0 = 0; kt0 < KTz; kt0 += U) {
#pragma unroll
                for (int u = 0; u < U; ++u) { // while the consumers run k-step kt out of stage kt & 1: k-step kt + 1 -> the other stage
                    const uint32_t kt = kt0 + u;
                    stage_a(kt + 1, a_st[(u + 1) % DA]);
                    load_a(kt + 1 + DA, a_st[(u + 1) % DA]);
                    convert_to(kt + 1, ringp[(u + 1) % DB]);
                    load_w1(kt + 1 + DB, ringp[(u + 1) % DB]);
                    UZU_PP_MARK(0)
                    step_barrier();
                    UZU_PP_MARK(1)
                }
            }
#ifdef UZU_GEMM_PP_TIMING
            if (dbg && lane == 0 && live) {
                unsigned long long* o = dbg + ((size_t)65536 + (size_t)vblock * 8 + 4 + wave) * 8;
                for (int i = 0; i < 4; ++i) o[i] = tsum[i];
                o[4] = KTz, o[5] = 1, o[6] = 0, o[7] = 0;
            }
#endif
            if (gated) __syncthreads(); // the consumers' act-mul epilogue has one workgroup barrier
            return;
        }
        // ---- consumers
        const uint8_t* const s_br = &s_b[0][(wn * 2 * 4) * 1024 + lane * 16];
        load_scale(0, sc_cur);
        load_scale(1, sc_nxt);
        ts[1] = wall_clock64();
        step_barrier();
#ifdef UZU_GEMM_PP_TIMING
        tmark = clock64();
#endif
        for (uint32_t kt0 = 0; kt0 < KTz; kt0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t kt = kt0 + u;
                const uint8_t* ab = a_frag_base + (kt & 1) * (BM * A_PITCH);
                const uint8_t* bb = s_br + (kt & 1) * 16384;
                u32x4_t af[4][2], bf[4][2];
                auto reads = [&](int s) {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) af[s][mb] = *(const u32x4_t*)(ab + mb * 32 * A_PITCH + s * 16);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) bf[s][nb] = *(const u32x4_t*)(bb + (nb * 4 + s) * 1024);
                };
                auto mfmas = [&](int s) {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const f32x16_t zero = {};
                            acc_g[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[s][mb]), __builtin_bit_cast(bf16x8_t, bf[s][nb]),
                                                                                   (u % GS == 0 && s == 0) ? zero : acc_g[mb][nb], 0, 0, 0);
                        }
                };
                reads(0), reads(1), reads(2), reads(3);
                mfmas(0), mfmas(1), mfmas(2), mfmas(3);
                __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_barrier(0); // the fold's inline-asm FMAs must not be scheduled between the MFMAs (the hazard wait sits at its head)
                if ((u + 1) % GS == 0) {
                    fold();
                    sc_cur[0] = sc_nxt[0], sc_cur[1] = sc_nxt[1];
                    load_scale((kt + 1) / GS + 1, sc_nxt);
                }
                UZU_PP_MARK(2)
                step_barrier();
                UZU_PP_MARK(3)
            }
        }
#ifdef UZU_GEMM_PP_TIMING
        if (dbg && lane == 0 && live) {
            unsigned long long* o = dbg + ((size_t)65536 + (size_t)vblock * 8 + wave) * 8;
            for (int i = 0; i < 4; ++i) o[i] = tsum[i];
            o[4] = KTz, o[5] = 0, o[6] = 0, o[7] = 0;
        }
#endif
    } else if constexpr (PP) {
        // ---- ping-pong main loop.  Per half: C(kt) | barrier | M(kt) | barrier | C(kt + 1) ...; half 1 starts one phase late.
        // Weight fragments through LDS, converted ONCE per half: the two waves that share 64 columns (wm = 0 / 1) each convert one of the two
        // 32-column blocks (block wm: one 16-byte code vector per lane per k-step, 36 vector instructions instead of 72) and write the four
        // bf16 fragments lane-aligned -- [wn][block][k16 step][lane] x 16 bytes, reader lane = writer lane, conflict-free both ways; the
        // barrier that ends the phase publishes them, and every wave of the half reads the eight fragments of its columns in M(kt).
        uint8_t* const s_bw = &s_b[half][((wn * 2 + wm) * 4) * 1024 + lane * 16];
        const uint8_t* const s_br = &s_b[half][(wn * 2 * 4) * 1024 + lane * 16];
        const uint32_t w_own = wm ? w_off[1] : w_off[0];
        u32x4_v ringp[DB][WV];
        auto load_w1 = [&](uint32_t kt, u32x4_v (&r)[WV]) {
            kt = min(kt, KTz - 1);
            const u32x4_v* src = (const u32x4_v*)(((const uint8_t*)p.b + (size_t)(kt_lo + kt) * BK * BITS / 8) + w_own);
#pragma unroll
            for (int v = 0; v < WV; ++v) r[v] = src[v];
        };
        auto convert_own = [&](const u32x4_v (&raw)[WV]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4_t f;
                if (BITS == 4) f = kPairs ? dequant4_pairs(raw[0][s] ^ flip) : dequant4(raw[0][s] ^ flip);
                else f = dequant8(raw[s >> 1][(s & 1) * 2] ^ flip, raw[s >> 1][(s & 1) * 2 + 1] ^ flip);
                *(u32x4_t*)(s_bw + s * 1024) = f;
            }
        };
        auto mfma_phase = [&](uint32_t kt, bool first, u32x4_v (&ast)[NR]) {
            const uint8_t* ab = a_frag_base + (kt & 1) * (BM * A_PITCH);
            u32x4_t af[4][2], bf[4][2];
            auto reads = [&](int s) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) af[s][mb] = *(const u32x4_t*)(ab + mb * 32 * A_PITCH + s * 16);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) bf[s][nb] = *(const u32x4_t*)(s_br + (nb * 4 + s) * 1024);
            };
            auto mfmas = [&](int s) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const f32x16_t zero = {};
                        acc_g[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[s][mb]), __builtin_bit_cast(bf16x8_t, bf[s][nb]),
                                                                               (first && s == 0) ? zero : acc_g[mb][nb], 0, 0, 0);
                    }
            };
            // every operand request up front, sixteen MFMAs back to back (an instruction issued between two MFMAs delays the second by more than
            // its own slot), then this thread's part of tile kt + 1 -> the other LDS buffer and the request for tile kt + 1 + DA while the pipe drains
#if UZU_GEMM_PP_MSCHED == 2
            // staging FIRST (its registers arrived k-steps ago): the LDS stores and the new requests are under way while the MFMAs run, and the
            // lgkmcnt(0) that ends the phase finds them done -- behind the last MFMA their ~130-cycle drain is exposed
            stage_a(kt + 1, ast);
            load_a(kt + 1 + DA, ast);
            reads(0), reads(1);
            mfmas(0);
            reads(2);
            mfmas(1);
            reads(3);
            mfmas(2);
            mfmas(3);
            __builtin_amdgcn_sched_group_barrier(0x002, 4 * NR, 0); // v_perm
            __builtin_amdgcn_sched_group_barrier(0x200, NR, 0);     // ds_write
            __builtin_amdgcn_sched_group_barrier(0x020, NR, 0);     // global loads
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#elif UZU_GEMM_PP_MSCHED == 0
            reads(0), reads(1), reads(2), reads(3);
            __builtin_amdgcn_s_setprio(1);
            mfmas(0), mfmas(1), mfmas(2), mfmas(3);
            __builtin_amdgcn_s_setprio(0);
            stage_a(kt + 1, ast);
            load_a(kt + 1 + DA, ast);
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4 * NR, 0); // v_perm
            __builtin_amdgcn_sched_group_barrier(0x200, NR, 0);     // ds_write
            __builtin_amdgcn_sched_group_barrier(0x020, NR, 0);     // global loads
#else
            reads(0), reads(1);
            mfmas(0);
            reads(2);
            mfmas(1);
            reads(3);
            mfmas(2);
            stage_a(kt + 1, ast);
            mfmas(3);
            load_a(kt + 1 + DA, ast);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NR, 0); // v_perm
            }
            __builtin_amdgcn_sched_group_barrier(0x200, NR, 0); // ds_write
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, NR, 0); // global loads
#endif
        };
        auto phase_barrier = [&]() { // nothing of a phase moves into the other half's turn on the pipe
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        {
            u32x4_v first[NR];
            load_a(0, first);
#pragma unroll
            for (int u = 1; u <= DA; ++u) load_a(u, a_st[u % DA]); // slot (kt + 1) % DA holds tile kt + 1
#pragma unroll
            for (int u = 0; u < DB - 1; ++u) load_w1(u, ringp[u]);
            load_scale(0, sc_cur);
            load_scale(1, sc_nxt);
            stage_a(0, first); // both halves, before the first barrier either of them passes
        }
        ts[1] = wall_clock64();
        if (half) phase_barrier();
#ifdef UZU_GEMM_PP_TIMING
        tmark = clock64();
#endif
        for (uint32_t kt0 = 0; kt0 < KTz; kt0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t kt = kt0 + u;
                // ---- C(kt)
                // fragments first, fold last: the four LDS stores drain under the fold's FMAs instead of in front of the barrier, and the
                // conversion's ~40 instructions are the MFMA -> VALU hazard distance the fold would otherwise idle through
                load_w1(kt + DB - 1, ringp[(u + DB - 1) % DB]);
                UZU_PP_MARK2(5)
                convert_own(ringp[u % DB]);
                __builtin_amdgcn_sched_barrier(0);
                if (u % GS == 0 && kt) { // the group that M(kt - 1) completed
                    fold(false);
                    sc_cur[0] = sc_nxt[0], sc_cur[1] = sc_nxt[1];
                    load_scale(kt / GS + 1, sc_nxt);
                }
                UZU_PP_MARK2(4)
                __builtin_amdgcn_sched_barrier(0);
                UZU_PP_MARK(0)
                phase_barrier();
                UZU_PP_MARK(1)
                // ---- M(kt)
                mfma_phase(kt, u % GS == 0, a_st[(u + 1) % DA]);
                __builtin_amdgcn_sched_barrier(0);
                UZU_PP_MARK(2)
                phase_barrier();
                UZU_PP_MARK(3)
            }
        }
#ifdef UZU_GEMM_PP_TIMING
        if (dbg && lane == 0 && live) {
            unsigned long long* o = dbg + ((size_t)65536 + (size_t)vblock * 4 + wave) * 8;
            for (int i = 0; i < 4; ++i) o[i] = tsum[i];
            o[4] = KTz, o[5] = half | tsum[4] << 8, o[6] = tsum[5], o[7] = tsum[6] | tsum[7] << 32;
        }
#endif
        fold();
    } else {
        // ---- prologue: activation tile 0 -> LDS, tiles 1 .. DA -> registers, ring slots 0 .. DB-2, scales of groups 0 and 1
        {
            u32x4_v first[NR];
            load_a(0, first);
    #pragma unroll
            for (int u = 1; u <= DA; ++u) load_a(u, a_st[u % DA]); // slot (kt + 1) % DA holds tile kt + 1
    #pragma unroll
            for (int u = 0; u < DB - 1; ++u) load_w(u, ring[u]);
            load_scale(0, sc_cur);
            load_scale(1, sc_nxt);
            stage_a(0, first);
        }
        lds_barrier();
        ts[1] = wall_clock64();

        for (uint32_t kt0 = 0; kt0 < KTz; kt0 += U) {
    #pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t kt = kt0 + u;
                // (UZU_GEMM_LAB_*: ablation builds for tools/kbench -- wrong results, one ingredient of the k-step removed at a time)
                load_w(kt + DB - 1, ring[(u + DB - 1) % DB]);
                mfma_codes(kt, ring[u % DB], u % GS == 0);
#ifndef UZU_GEMM_LAB_NOSTAGE
                stage_a(kt + 1, a_st[(u + 1) % DA]); // tile kt + 1 (requested DA k-steps ago) -> the other LDS buffer
                load_a(kt + 1 + DA, a_st[(u + 1) % DA]);
#endif
                if ((u + 1) % GS == 0) {
                    fold();
                    sc_cur[0] = sc_nxt[0], sc_cur[1] = sc_nxt[1];
                    load_scale((kt + 1) / GS + 1, sc_nxt);
                }
#ifndef UZU_GEMM_LAB_NOBAR
                lds_barrier();
#endif
            }
        }
    }
    ts[2] = wall_clock64();

    // ---- offset term: acc_t[m, n] += sum_g rowsum[g][m] * coef[g][n] over this split's groups, on the matrix cores in
    // f32 (v_mfma_f32_32x32x2_f32: products and sums exact to f32): lane (c, h) supplies row / column c of group 2 i + h,
    // every load is one coalesced dword per lane.
    if (p.b_kind != UZU_MATMUL_B_SCALE_SYMMETRIC || (kPairs && BITS == 4)) {
        const uint32_t Mp = (M + 3) & ~3u;
        uint32_t arow[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) arow[mb] = min(m0 + wm * 64 + mb * 32 + c, Mp - 1);
        // (a producer may have filed 2^PL partial sums per group, rows [g << PL, (g + 1) << PL) of `rowsum`: added here in part order -- more loads in flight per
        // iteration, the same number of iterations and MFMAs; walking the parts as extra iterations cost the long-K down projections more than the pre-pass saved)
        const uint32_t PL = p.rowsum_parts_log2, g_end = g_lo + Gz;
#pragma unroll 4
        for (uint32_t g2 = g_lo; g2 < g_end; g2 += 2) {
            const uint32_t g = min(g2 + h, g_end - 1);
            const bool live = g2 + h < g_end;
            float av[2], bv[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const float* rp = rowsum + (size_t)(g << PL) * Mp + arow[mb];
                av[mb] = rp[0];
                if (PL >= 1) av[mb] += rp[Mp]; // (wave-uniform; PL <= 2: host-checked)
                if (PL >= 2) av[mb] += rp[2 * (size_t)Mp], av[mb] += rp[3 * (size_t)Mp];
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) bv[nb] = live ? coef[(size_t)g * N + ncol[nb]] : 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc_t[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb], bv[nb], acc_t[mb][nb], 0, 0, 0);
        }
    }

    // ---- epilogue (kernel.rs:281-292): lane holds column n, rows (r & 3) + 8 (r >> 2) + 4 h of each block
    uint16_t* d = (uint16_t*)p.d;
    float* d32 = (float*)p.d;
    const bool out_f32 = p.d_dt == UZU_F32;
    const bool via_lds = !partials && !out_f32 && !p.accumulate && !p.has_soft_cap && (gated ? H : N) % 8 == 0 && (uintptr_t)p.d % 16 == 0; // gated: host-checked
    if (via_lds) {
        // Common case, branch-free: bf16(ab_scale * acc + bias) goes to a wave-private LDS tile (64 rows x 64 columns,
        // 144-byte pitch; the activation buffers are free after the last barrier) and leaves as 16-byte row segments,
        // 8 lanes per 128-byte row -- instead of 64 two-byte stores per lane.
        uint16_t* s_d = (uint16_t*)s_ep + wave * (64 * 72);
        const float ab = p.ab_scale;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float bias = p.bias ? bf16_to_f32(((const uint16_t*)p.bias)[ncol[nb]]) : 0.0f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) { // rows r and r + 1 are consecutive
                    const uint32_t lr = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const uint32_t pk = pack_bf16(__fadd_rn(__fmul_rn(ab, acc_t[mb][nb][r]), bias), __fadd_rn(__fmul_rn(ab, acc_t[mb][nb][r + 1]), bias)); // two roundings, as the reference
                    s_d[lr * 72 + nb * 32 + c] = (uint16_t)pk;
                    s_d[(lr + 1) * 72 + nb * 32 + c] = (uint16_t)(pk >> 16);
                }
        }
        const int seg = lane & 7, rsub = lane >> 3;
        if (gated) {
            // the up tile (wave wm * 2) and the gate tile (wave wm * 2 + 1) hold the same (row, column) positions:
            // out = bf16(up * act(gate)), both operands already rounded to bf16 as the separate kernels would see them; each of the
            // pair's waves combines 32 of the 64 rows; the exp table of SiLU / softplus sits in LDS (a table read from memory per
            // element is a dependent round trip with two waves to hide it)
            if (tid < 32) s_exp_tab[tid] = kExp2fTab[tid];
            __syncthreads();
            const uint16_t* s_u = (const uint16_t*)s_ep + (wm * 2) * (64 * 72);
            const uint16_t* s_g = s_u + 64 * 72;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const uint32_t lr = wn * 32 + pass * 8 + rsub, m = m0 + wm * 64 + lr, n = n_t * 64 + seg * 8;
                const u32x4_v uv = *(const u32x4_v*)(s_u + lr * 72 + seg * 8), gv = *(const u32x4_v*)(s_g + lr * 72 + seg * 8);
                u32x4_v ov;
                float part = 0.f; // the row's sum over this lane's 8 ROUNDED outputs (columns past H count as zero)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float u0 = bits_to_f32(uv[w] << 16), u1 = bits_to_f32(uv[w] & 0xFFFF0000u);
                    const float g0 = bits_to_f32(gv[w] << 16), g1 = bits_to_f32(gv[w] & 0xFFFF0000u);
                    ov[w] = pack_bf16(u0 * activate_bf16_tab(p.act_type, g0, s_exp_tab), u1 * activate_bf16_tab(p.act_type, g1, s_exp_tab));
                    part += bits_to_f32(ov[w] << 16);
                    part += bits_to_f32(ov[w] & 0xFFFF0000u);
                }
                if (m < Mst && n < H) *(u32x4_v*)(d + (size_t)m * H + n) = ov;
                if (p.gated_rowsum_out) { // the 8 lanes of a row segment group are neighbours: three butterfly steps, lane seg 0 files the 64-column sum
                    if (n >= H) part = 0.f;
                    part += __shfl_xor(part, 1);
                    part += __shfl_xor(part, 2);
                    part += __shfl_xor(part, 4);
                    const uint32_t Mp = (M + 3) & ~3u;
                    if (seg == 0 && live && m < Mp) p.gated_rowsum_out[(size_t)n_t * Mp + m] = m < M ? part : 0.f;
                }
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // wave-private region: no workgroup barrier needed
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const uint32_t lr = pass * 8 + rsub, m = m0 + wm * 64 + lr, n = n0 + wn * 64 + seg * 8;
                const u32x4_v v = *(const u32x4_v*)(s_d + lr * 72 + seg * 8);
                if (m < Mst && n < N) *(u32x4_v*)(d + (size_t)m * N + n) = v;
            }
        }
    } else if (partials) { // split-K: raw f32 partial tile (32 lanes = one 128-byte row segment per store)
        float* pz = partials + (size_t)z * M * N;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const uint32_t n = n0 + wn * 64 + nb * 32 + c;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t m = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < Mst && n < N) pz[(size_t)m * N + n] = acc_t[mb][nb][r];
                }
        }
    } else {
#pragma unroll 1
        for (int nb = 0; nb < 2; ++nb) {
            const uint32_t n = n0 + wn * 64 + nb * 32 + c;
            const float bias = (p.bias && n < N) ? bf16_to_f32(((const uint16_t*)p.bias)[n]) : 0.0f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t m = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m >= Mst || n >= N) continue;
                    const size_t idx = (size_t)m * N + n;
                    const float acc = nb ? acc_t[mb][1][r] : acc_t[mb][0][r];
                    float value = p.ab_scale * acc;
                    if (p.accumulate) value += out_f32 ? d32[idx] : bf16_to_f32(d[idx]);
                    if (p.bias) value += bias;
                    if (p.has_soft_cap) value = p.soft_cap * tanhf(value / p.soft_cap);
                    if (out_f32) d32[idx] = value;
                    else d[idx] = f32_to_bf16(value);
                }
        }
    }
    if (PP && !half) lds_barrier(); // the partner half's last phase barrier (it runs one phase behind)
    if (dbg && tid == 0 && live) {
        ts[3] = wall_clock64();
        unsigned long long* o = dbg + (size_t)(blockIdx.y * gridDim.x * NH + vblock) * 8;
        for (int i = 0; i < 4; ++i) o[i] = ts[i];
        o[6] = (unsigned long long)m_t << 32 | n_t;
        o[7] = xcd;
    }
}

// split-K reduction in split order + the epilogue
__global__ void __launch_bounds__(256) gemm_split_reduce_kernel(MatmulParams p, const float* partials, uint32_t splits) {
    const size_t total = (size_t)p.m * p.n;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float acc = partials[idx];
    for (uint32_t z = 1; z < splits; ++z) acc += partials[(size_t)z * total + idx];
    const uint32_t n = (uint32_t)(idx % p.n);
    const bool out_f32 = p.d_dt == UZU_F32;
    float value = p.ab_scale * acc;
    if (p.accumulate) value += out_f32 ? ((const float*)p.d)[idx] : bf16_to_f32(((const uint16_t*)p.d)[idx]);
    if (p.bias) value += bf16_to_f32(((const uint16_t*)p.bias)[n]);
    if (p.has_soft_cap) value = p.soft_cap * tanhf(value / p.soft_cap);
    if (out_f32) ((float*)p.d)[idx] = value;
    else ((uint16_t*)p.d)[idx] = f32_to_bf16(value);
}

// ---------------------------------------------------------------------------------------------- host side
bool gemm_q_mfma128_supported(const MatmulParams& p, int num_cus);
unsigned long long* g_gemm128_dbg = nullptr; // tools/kbench KB_GEMM_DBG: per-workgroup phase timestamps (100 MHz wall clock)
static uint32_t gemm128_splits(const MatmulParams& p, int num_cus) {
    const char* e = tune_env("gemm_splits"); // read per call: tests/test_gpu_kernels.py pins it to cover both paths
    const int force = e ? atoi(e) : 0;
    const uint32_t G = p.k / p.group_size, gs = p.group_size / BK;
    const uint32_t tiles = ((p.m + BM - 1) / BM) * ((p.n + BN - 1) / BN);
    auto ok = [&](uint32_t s) { return G % s == 0 && ((G / s) * gs) % 4 == 0; };
    if (force > 0 && !p.act_mul) return ok((uint32_t)force) ? (uint32_t)force : 0u;
    if (!ok(1)) return 0;
    if (p.act_mul) return 1; // the gated epilogue pairs the two wave tiles of one workgroup: no partial tiles
    uint32_t best = 1;
    for (uint32_t s = 2; s <= 4; ++s) // each split adds an f32 tile round trip: stop once the chip has a workgroup per CU
        if (ok(s) && tiles * best < (uint32_t)num_cus * 3 / 4 && tiles * s <= (uint32_t)num_cus * 2) best = s;
    return best;
}
// Which form runs (0 = 256-thread, 1 = ping-pong, 2 = wave-specialised).  UZU_GEMM_FORM forces one (read per call: A/B runs and the bit-identity
// test flip it); otherwise the plan, from same-box A/B runs of all three (tools/kbench KB_GEMM_AB, profiles/r5_gemm_pp_ab.txt):
//   * ping-pong (x1.05-1.16): long reductions (>= 48 k-steps: its one workgroup per CU has no second workgroup whose main loop would cover its
//     prologue / epilogue), at least one 128 x 256 tile per CU, no split-K, no gated epilogue;
//   * wave-specialised (x1.02-1.05): the few-tile split-K shapes (N = 1024 projections of the 0.8B model);
//   * the 256-thread form everywhere else (gated epilogues, short reductions with many tiles: x0.6-0.9 for the other two).
static int gemm128_form(const MatmulParams& p, int num_cus, uint32_t splits) {
    if (const char* e = tune_env("gemm_form")) return atoi(e) == 2 ? 2 : atoi(e) == 1 ? 1 : 0;
    if (p.act_mul) return 0;
    if (splits > 1) return 2;
    const uint32_t m_tiles = (p.m + BM - 1) / BM, n_tiles = (p.n + BN - 1) / BN;
    return (p.k / BK >= 48 && m_tiles * ((n_tiles + 1) / 2) >= (uint32_t)num_cus) ? 1 : 0;
}
// host arithmetic only (uzu_hip_prefill_gemm_plan): which kernel / form / split-K a shape gets; UZU_GEMM_FORM / UZU_GEMM_SPLITS apply as at launch
void gemm_q_mfma128_plan_query(const MatmulParams& p, int num_cus, uint32_t* large_tile, uint32_t* form, uint32_t* splits, uint32_t* workgroups) {
    *large_tile = gemm_q_mfma128_supported(p, num_cus) ? 1u : 0u;
    *form = *splits = *workgroups = 0;
    if (!*large_tile) return;
    *splits = gemm128_splits(p, num_cus);
    *form = (uint32_t)gemm128_form(p, num_cus, *splits);
    const uint32_t m_tiles = (p.m + BM - 1) / BM, n_tiles = p.act_mul ? (p.n / 2 + 63) / 64 : (p.n + BN - 1) / BN;
    *workgroups = (*form == 1 ? m_tiles * ((n_tiles + 1) / 2) : m_tiles * n_tiles) * *splits; // tiles with work (the launch grid pads them to whole super-tiles)
}
bool gemm_q_mfma128_supported(const MatmulParams& p, int num_cus) {
    if (p.m < 128 || p.n < 64) return false;
    if (p.act_mul && (p.n % 16 || p.d_dt != UZU_BF16 || p.accumulate || p.has_soft_cap || (uintptr_t)p.d % 16 || gemm128_splits(p, num_cus) != 1)) return false;
    if (p.group_size != 64 && p.group_size != 128 && p.group_size != 256) return false;
    if ((size_t)p.k * p.bits / 8 % 16 || p.k % BK) return false;
    if ((uint64_t)p.m * p.k >= (1ull << 32) || (uint64_t)p.n * p.k * p.bits / 8 >= (1ull << 32)) return false; // 32-bit element offsets
    return gemm128_splits(p, num_cus) != 0;
}
bool gemm_coef_table_supported(const MatmulParams& p) {
    if (p.b_kind == UZU_MATMUL_B_FULL_PRECISION || !(p.bits == 4 || p.bits == 8)) return false;
    if (p.group_size != 64 && p.group_size != 128 && p.group_size != 256) return false;
    if (p.k % p.group_size) return false;
    return p.b_kind != UZU_MATMUL_B_SCALE_SYMMETRIC || (kPairs && p.bits == 4);
}
uzu_status gemm_coef_table(hipStream_t s, const MatmulParams& p, float* coef) {
    const uint32_t G = p.k / p.group_size;
    return launch_check([&] { hipLaunchKernelGGL(gemm_prepass_kernel, dim3((p.n * G + 255) / 256), dim3(256), 0, s, p, (float*)nullptr, coef, 0u); }, "gemm_coef_table");
}
size_t gemm_q_mfma128_workspace_bytes(const MatmulParams& p, int num_cus) {
    const uint32_t G = p.k / p.group_size;
    const uint32_t splits = gemm128_splits(p, num_cus);
    size_t bytes = ((size_t)((p.m + 3) & ~3u) * G * 4 + 255) / 256 * 256 + ((size_t)p.n * G * 4 + 255) / 256 * 256;
    if (splits > 1) bytes += (size_t)splits * p.m * p.n * 4;
    return bytes;
}
uzu_status gemm_q_mfma128(hipStream_t s, const MatmulParams& p, int num_cus, void* workspace) {
    const uint32_t G = p.k / p.group_size;
    const uint32_t splits = gemm128_splits(p, num_cus);
    const size_t rowsum_bytes = ((size_t)((p.m + 3) & ~3u) * G * 4 + 255) / 256 * 256, coef_bytes = ((size_t)p.n * G * 4 + 255) / 256 * 256;
    float* rowsum = (float*)workspace;
    float* coef = (float*)((uint8_t*)workspace + rowsum_bytes);
    float* partials = splits > 1 ? (float*)((uint8_t*)workspace + rowsum_bytes + coef_bytes) : nullptr;
    uzu_status st = UZU_OK;
    if (p.b_kind != UZU_MATMUL_B_SCALE_SYMMETRIC || (kPairs && p.bits == 4)) {
        // the caller may bring either table (engine: coefficients tabulated at load, row sums written by the normalisation that produced A)
        const uint32_t rowsum_blocks = p.pre_rowsum ? 0u : (((p.m + 3) & ~3u) + 3) / 4, coef_blocks = p.pre_coef ? 0u : (p.n * G + 255) / 256;
        if (rowsum_blocks + coef_blocks) {
            st = launch_check([&] { hipLaunchKernelGGL(gemm_prepass_kernel, dim3(rowsum_blocks + coef_blocks), dim3(256), 0, s, p, rowsum, coef, rowsum_blocks); }, "gemm_prepass");
            if (st != UZU_OK) return st;
        }
    }
    const float* rowsum_in = p.pre_rowsum ? p.pre_rowsum : rowsum;
    const float* coef_in = p.pre_coef ? p.pre_coef : coef;
    const uint32_t m_tiles = (p.m + BM - 1) / BM, n_tiles = p.act_mul ? (p.n / 2 + 63) / 64 : (p.n + BN - 1) / BN;
    const int form = gemm128_form(p, num_cus, splits);
    const dim3 grid(form == 1 ? gemm_grid_x(m_tiles, (n_tiles + 1) / 2) : gemm_grid_x(m_tiles, n_tiles), splits), block(form ? 512 : 256);
#define UZU_LAUNCH(B, GSV)                                                                                                                                            \
    st = form == 2 ? launch_check([&] { hipLaunchKernelGGL((gemm_q_mfma128_kernel<B, GSV, 2>), grid, block, 0, s, p, rowsum_in, coef_in, partials, g_gemm128_dbg); }, "gemm_q_mfma128ws") \
       : form == 1 ? launch_check([&] { hipLaunchKernelGGL((gemm_q_mfma128_kernel<B, GSV, 1>), grid, block, 0, s, p, rowsum_in, coef_in, partials, g_gemm128_dbg); }, "gemm_q_mfma128pp") \
                   : launch_check([&] { hipLaunchKernelGGL((gemm_q_mfma128_kernel<B, GSV, 0>), grid, block, 0, s, p, rowsum_in, coef_in, partials, g_gemm128_dbg); }, "gemm_q_mfma128")
    const uint32_t gs = p.group_size / BK;
    if (p.bits == 4) {
        if (gs == 1) UZU_LAUNCH(4, 1);
        else if (gs == 2) UZU_LAUNCH(4, 2);
        else UZU_LAUNCH(4, 4);
    } else {
        if (gs == 1) UZU_LAUNCH(8, 1);
        else if (gs == 2) UZU_LAUNCH(8, 2);
        else UZU_LAUNCH(8, 4);
    }
#undef UZU_LAUNCH
    if (st != UZU_OK || splits == 1) return st;
    const size_t total = (size_t)p.m * p.n;
    if (p.post_norm && p.post_norm_done) { // the rows go straight into a Normalization: reduction + epilogue + normalisation as one launch
        const NormPartials sp{partials, splits, total, (const uint16_t*)p.bias, (uint16_t*)p.d};
        if (p.ab_scale == 1.0f && !p.accumulate && !p.has_soft_cap && p.d_dt == UZU_BF16 && p.w_dt == UZU_BF16 && p.post_norm->batch_size == p.m &&
            p.post_norm->element_count == p.n && normalization_from_partials_supported(*p.post_norm, sp)) {
            *p.post_norm_done = 1;
            return normalization_from_partials(s, *p.post_norm, sp);
        }
    }
    return launch_check([&] { hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, p, partials, splits); }, "gemm_split_reduce");
}

} // namespace k
} // namespace uzu
