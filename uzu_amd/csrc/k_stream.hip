// k_stream.hip -- the weight-streaming engine of the decode GEMVs (bandwidth regime): LDS-staged weight tiles.
//
// gemv_dec (k_decode.hip) keeps its weights in registers: a wave has two 16-byte items per lane in flight and the loads of a
// workgroup start only when its waves do, so the HBM pipe idles through the 3-4 us Normalization prologue and through every
// ragged last round (Llama-3-8B up-projection: 62.5 MB in 17.9 us although its dot phase streams at 6.2 TB/s).  Here the
// stream is decoupled from the arithmetic:
//
//   * one workgroup per CU = 1 LOADER wave + 15 CONSUMER waves (7 where the activation row needs more than 128 registers);
//   * the loader walks the CU's share of the weight matrix in SLOTS of <= 16 KiB (a few whole rows of packed codes + their
//     bf16 scales / biases, contiguous in HBM because the reference layout is [N, K/2] row-major) and moves them with LDS-DMA
//     (global_load_lds_dwordx4, 64 lanes x 16 B = 1 KiB per instruction, non-temporal) into a ring of up to 8 slots -- no
//     registers, no waits on behalf of the arithmetic: it starts at kernel entry, runs through the prologue and stays <= 2-3
//     slots (40 VMEM operations) ahead of what has landed.  vmcnt is the only completion signal LDS-DMA has, so the loader
//     publishes "slots < f have landed" in an LDS word after a counted s_waitcnt (loads return in issue order);
//   * consumers draw work items (a few rows of a slot) from an LDS counter, wait for the slot's publication, read codes /
//     scales / biases with ds_read, run the SAME per-lane arithmetic as gemv_dec (gemv_core.h: lane sl of a row's lpr lanes owns
//     the 32-element steps sl + lpr j, packed bf16 dot, xor-butterfly row sum, same rounding points => bit-identical results
//     whichever kernel, wave or slot computes a row), and count the item done; the loader reuses a ring position when all items
//     of its previous slot are done;
//   * the activation row lives in registers (CPL x 16 packed-bf16 registers per lane: with 8 waves per CU every wave has 256);
//     the Normalization prologue (normalization.rs:56-125) runs on consumer waves 0-3 with the element mapping / reduction
//     order of normalization_kernel and synchronises through LDS counters, never through s_barrier -- the loader does not stop.
//
// Every spin is bounded (a stuck protocol sets the error word and lets the kernel end with garbage instead of hanging the GPU).
// Instantiated for int4 ScaleBias (the MLX layout of the BASELINE configs), one or two matrices, plain / Normalization
// prologue, plain / GatedActMul / arg-max epilogues.  gemv_dec routes here (gemv_stream_wanted) and stays the fallback.
#include <stdlib.h>

#include "decode_epilogue.h"
#include "device_utils.h"
#include "gemv_core.h"
#include "kernels.h"
#include "kernels_decode.h"

namespace uzu {
namespace k {

namespace {

constexpr uint32_t kMaxRing = 8;
constexpr uint32_t kSpinLimit = 1u << 22; // x s_sleep(1) ~ 64 cycles: ~0.1-0.3 s, then give up (error word, garbage, no hang)

struct StreamGeo {
    uint32_t lpr_log2;
    uint32_t rows_per_slot;  // logical rows (ACT: up / gate pairs) per slot; multiple of the item's rows
    uint32_t items_per_slot;
    uint32_t chunk_stride;   // LDS bytes of one chunk of codes (rows_per_slot * row_bytes rounded up to 1 KiB)
    uint32_t sb_stride;      // LDS bytes of one chunk of scales (or biases): rows_per_slot * G * 2 rounded up to 256 B
    uint32_t off_scales, off_biases;
    uint32_t slot_bytes;
    uint32_t ring_slots;
    uint32_t slots0, slots1; // slots of matrix 0 / 1
    uint32_t ops_per_slot;   // VMEM operations per slot (constant: partial slots issue clamped re-reads)
    uint32_t depth;          // landed-before-published window of the loader, in slots (1 or 2)
    uint32_t xs_off;         // dynamic-LDS offset of the f32 activation staging (Normalization prologue)
    uint32_t dump_off;       // dynamic-LDS offset of 256 bytes the padding operations land in
};

// ---- LDS-DMA (cdna_hip_programming.md 5.7: M0 is written in the statement that reads it and restored)
__device__ __forceinline__ void glds16_nt(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// scalar-base forms: address = SGPR pair + 32-bit per-lane offset + immediate; the immediate moves source AND destination, so four
// consecutive 1 KiB pieces share one M0 / base set-up (a loader wave is bound by its own instruction latencies, not by issue slots:
// r3 kbench: ~0.65 us of fixed cost per slot with per-lane 64-bit address arithmetic and an M0 round trip per piece)
__device__ __forceinline__ void glds16x4_nt(uint32_t voff, const void* sbase, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__device__ __forceinline__ void glds16s_nt(uint32_t voff, const void* sbase, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__device__ __forceinline__ void glds4s(uint32_t voff, const void* sbase, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
constexpr uint32_t kOpsPerSlot = 20; // VMEM operations per slot, padded with dummy reads: the counted waits below need immediates
__device__ __forceinline__ uint32_t lds_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// bounded wait for *flag >= target; false = gave up
__device__ __forceinline__ bool wait_ge(const uint32_t* flag, uint32_t target, uint32_t* err, uint32_t code) {
    uint32_t spins = 0;
    while (lds_load(flag) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) {
            if (err && (threadIdx.x & 63) == 0) atomicOr(err, code);
            return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; } // LDS aperture: the low 32 bits are the LDS byte address

} // namespace

// bit mask of bounded spins that gave up (host: uzu_hip_debug_decode_stream_error)
__device__ uint32_t g_stream_err_dev;

template <int CPL, bool ACT, int PRO, int NW>
__global__ void __launch_bounds__(64 * NW) gemv_stream_kernel(DecGemvParams p, StreamGeo g) {
    constexpr int NPHYS = ACT ? 2 : 1;
    constexpr int RR = ACT ? (CPL >= 2 ? 1 : 2) : (CPL >= 3 ? 1 : (CPL == 2 ? 2 : 4)); // row iterations of a work item: ~4 steps per lane
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_filled, s_next_item, s_sync, s_done[kMaxRing];
    __shared__ float s_red[4];
    __shared__ uint64_t s_exp_tab[32];
    __shared__ float s_bv[NW];
    __shared__ uint32_t s_bi[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t* const err = &g_stream_err_dev;
    UZU_TL_DECL;
    UZU_TL_STAMP(0);
    const uint32_t K = p.k;
    const uint32_t C = K / 32, row_bytes = K / 2;
    const uint32_t G = (K + p.group_size - 1) / p.group_size;
    const uint32_t n_log0 = ACT ? p.n[0] / 2 : p.n[0];
    const uint32_t R = g.rows_per_slot, S = g.ring_slots;
    const uint32_t T = g.slots0 + g.slots1;
    const uint32_t my_T = blockIdx.x < T ? (T - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t smem_base = lds_addr(smem);
    const uint32_t code_pieces = g.chunk_stride / 1024, sb_pieces = g.sb_stride / 256;

    // one slot of the loader: kOpsPerSlot VMEM operations, always (the counted waits need a constant)
    auto issue_slot = [&](uint32_t t, uint32_t pos) {
        const uint32_t gs = blockIdx.x + t * gridDim.x;
        const int mat = __builtin_amdgcn_readfirstlane(gs >= g.slots0 ? 1 : 0);
        const uint32_t ls = mat ? gs - g.slots0 : gs;
        const uint32_t nl = mat ? p.n[1] : n_log0;
        const uint32_t r0 = ls * R;
        const uint32_t rows_valid = nl - r0 < R ? nl - r0 : R;
        const uint32_t slot_lds = __builtin_amdgcn_readfirstlane(smem_base + pos * g.slot_bytes);
        const uint32_t voff16 = (uint32_t)lane * 16, voff4 = (uint32_t)lane * 4;
        const uint8_t* wbase = p.w[mat];
        const uint8_t* sbase = (const uint8_t*)p.scales[mat];
        const uint8_t* bbase = (const uint8_t*)p.biases[mat];
#pragma unroll
        for (int h = 0; h < NPHYS; ++h) {
            const uint32_t prow0 = ACT ? r0 + (h ? p.n[0] / 2 : 0) : r0;
            const uint8_t* src = wbase + (size_t)prow0 * row_bytes;           // whole rows, contiguous (reference layout [N, K/2])
            const uint8_t* ssrc = sbase + (size_t)prow0 * G * 2;
            const uint8_t* bsrc = bbase + (size_t)prow0 * G * 2;
            const uint32_t cdst = slot_lds + h * g.chunk_stride, sdst = slot_lds + g.off_scales + h * g.sb_stride, bdst = slot_lds + g.off_biases + h * g.sb_stride;
            if (rows_valid == R) { // full slot: scalar bases, immediates
                uint32_t q = 0;
                for (; q + 4 <= code_pieces; q += 4) glds16x4_nt(voff16, src + q * 1024, cdst + q * 1024);
                for (; q < code_pieces; ++q) glds16s_nt(voff16, src + q * 1024, cdst + q * 1024);
                for (uint32_t piece = 0; piece < sb_pieces; ++piece) {
                    glds4s(voff4, ssrc + piece * 256, sdst + piece * 256);
                    glds4s(voff4, bsrc + piece * 256, bdst + piece * 256);
                }
            } else { // the last slot of a matrix: per-lane clamped addresses (the bytes past the valid rows are re-reads, never consumed)
                const uint32_t valid = rows_valid * row_bytes, svalid = rows_valid * G * 2;
                for (uint32_t piece = 0; piece < code_pieces; ++piece) {
                    const uint32_t off = piece * 1024 + voff16;
                    glds16_nt(src + (off < valid ? off : valid - 16), __builtin_amdgcn_readfirstlane(cdst + piece * 1024));
                }
                for (uint32_t piece = 0; piece < sb_pieces; ++piece) {
                    const uint32_t off = piece * 256 + voff4;
                    const uint32_t o = off < svalid ? off : svalid - 4;
                    glds4(ssrc + o, __builtin_amdgcn_readfirstlane(sdst + piece * 256));
                    glds4(bsrc + o, __builtin_amdgcn_readfirstlane(bdst + piece * 256));
                }
            }
        }
        for (uint32_t i = g.ops_per_slot; i < kOpsPerSlot; ++i) glds4(sbase, __builtin_amdgcn_readfirstlane(smem_base + g.dump_off)); // padding
    };

    // Start-up.  No wave waits for another one's launch before its first memory request: the loader requests its first slot (ring
    // position 0 needs no flag), the prologue waves request the activation row / shortcut / norm scales, THEN the workgroup meets at
    // an LDS-only barrier (no vmcnt drain) behind which the flags are initialised.
    constexpr int NPRE = PRO == 1 ? 2 * CPL : 1; // E / 4 <= 2 CPL vectors of four elements per prologue thread (K = 32 lpr CPL, lpr <= 64)
    u32x2_v x_pre[NPRE], s_pre[NPRE];
    f32x4_v n_pre[NPRE];
    uint64_t exp_entry = 0;
    if (wave == 0) {
        if (tid == 0) {
            s_filled = 0, s_next_item = 0, s_sync = 0;
            for (uint32_t i = 0; i < kMaxRing; ++i) s_done[i] = 0;
        }
        if (my_T) issue_slot(0, 0);
        UZU_TL_STAMP(1); // first slot issued
    } else if (PRO == 1 && wave <= 4) {
        const uint32_t E = K / 256, pt = (uint32_t)tid - 64u;
#pragma unroll
        for (int qi = 0; qi < NPRE; ++qi) { // clamped: a vector past E is a re-read that is never consumed
            const uint32_t q = (uint32_t)qi * 4 < E ? (uint32_t)qi * 4 : E - 4;
            const uint32_t e = pt * E + q;
            x_pre[qi] = *(const u32x2_v*)(p.x + e);
            s_pre[qi] = *(const u32x2_v*)((p.residual_add ? p.shortcut_in : p.x) + e);
            n_pre[qi] = *(const f32x4_v*)(p.norm_scales ? p.norm_scales + e : (const float*)p.x);
        }
        if (ACT && wave == 1) exp_entry = kExp2fTab[lane & 31];
    }
    lds_barrier();

    if (wave == 0) {
        // ================================================================================================ loader
        const uint32_t D = g.depth;
        UZU_TL_STAMP(7);
        uint32_t pos = 0, round = 0; // ring position of slot t, times the ring has wrapped
        for (uint32_t t = 0; t < my_T; ++t) {
            if (t) {
                if (round && !wait_ge(&s_done[pos], round * g.items_per_slot, err, 1u)) break; // the position's previous tenant is consumed
                issue_slot(t, pos);
            }
            // loads return in issue order: <= D slots' operations outstanding  =>  slots <= t - D have landed
            if (D > 1) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            if (t >= D) lds_store(&s_filled, t - D + 1);
            if (++pos == S) pos = 0, ++round;
        }
        // drain: publish the last slots as they land
        if (my_T) {
            if (D > 1) {
                asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                lds_store(&s_filled, my_T - 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_store(&s_filled, my_T);
        }
        UZU_TL_STAMP(3); // the whole share has landed
    } else {
        // ============================================================================================== consumers
        const int cw = wave - 1; // consumer index 0 .. NW - 2
        const int lpr = 1 << g.lpr_log2, rpw = 64 >> g.lpr_log2;
        const int sl = lane & (lpr - 1), rsub = lane >> g.lpr_log2;
        const uint32_t gshift = 31 - __builtin_clz(p.group_size);
        XPack xq[CPL];
        float xsm[CPL];
        // ---- the activation row -> registers -----------------------------------------------------------------------
        if constexpr (PRO == 0) {
            if (ACT && cw == 0) {
                if (lane < 32) s_exp_tab[lane] = kExp2fTab[lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&s_sync, 1u);
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const uint32_t c = sl + lpr * j;
                if (c < C) xsm[j] = xpack_load(xq[j], p.x + (size_t)c * 32);
                else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) xq[j].v[i] = 0u;
                    xsm[j] = 0.f;
                }
            }
            if (ACT) wait_ge(&s_sync, 1u, err, 2u);
        } else {
            // Normalization (normalization.rs:56-125), element / thread mapping and reduction order of normalization_kernel and of
            // gemv_dec's prologue: thread t of 256 owns elements [t E, t E + E), E = K / 256; sum of squares: sequential fma per
            // thread, xor butterfly per wave, ((w0 + w1) + w2) + w3.
            float* xs = (float*)(smem + g.xs_off); // C slots of 36 floats
            const uint32_t E = K / 256;
            const bool pro = cw < 4;
            const uint32_t pt = (uint32_t)tid - 64u; // thread index inside the 256-thread prologue
            if (pro) {
                if (ACT && cw == 0 && lane < 32) s_exp_tab[lane] = exp_entry;
                float ss = 0.f;
#pragma unroll
                for (int qi = 0; qi < NPRE; ++qi) {
                    const uint32_t q = (uint32_t)qi * 4;
                    if (q < E) {
                        const uint32_t e = pt * E + q;
                        const u32x2_v xr = x_pre[qi];
                        float v[4] = {bits_to_f32(xr.x << 16), bits_to_f32(xr.x & 0xFFFF0000u), bits_to_f32(xr.y << 16), bits_to_f32(xr.y & 0xFFFF0000u)};
                        if (p.residual_add) {
                            const u32x2_v sr = s_pre[qi];
                            const float sc[4] = {bits_to_f32(sr.x << 16), bits_to_f32(sr.x & 0xFFFF0000u), bits_to_f32(sr.y << 16), bits_to_f32(sr.y & 0xFFFF0000u)};
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = round_bf16(v[i] + sc[i]);
                        }
                        if (p.shortcut_out && blockIdx.x == 0) {
                            uint2 o;
                            o.x = (f32_to_bits(v[0]) >> 16) | (f32_to_bits(v[1]) & 0xFFFF0000u);
                            o.y = (f32_to_bits(v[2]) >> 16) | (f32_to_bits(v[3]) & 0xFFFF0000u);
                            *(uint2*)(p.shortcut_out + e) = o;
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) ss = fmaf(v[i], v[i], ss);
                        *(float4*)(xs + (size_t)(e / 32) * 36 + e % 32) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
                ss = wave_sum(ss);
                if (lane == 0) s_red[cw] = ss;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&s_sync, 1u);
                wait_ge(&s_sync, 4u, err, 2u);
                const float total = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
                const float variance = total / (float)K - 0.0f * 0.0f;
                const float rms_inv = 1.0f / sqrtf(variance + p.norm_eps);
#pragma unroll
                for (int qi = 0; qi < NPRE; ++qi) {
                    const uint32_t q = (uint32_t)qi * 4;
                    if (q < E) {
                        const uint32_t e = pt * E + q;
                        float* slot = xs + (size_t)(e / 32) * 36 + e % 32;
                        const float4 vv = *(const float4*)slot; // own elements
                        float v[4] = {vv.x, vv.y, vv.z, vv.w};
                        const f32x4_v t4 = n_pre[qi];
                        const float scl[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float normalized = (v[i] - 0.0f) * rms_inv;
                            if (!p.norm_scales) v[i] = round_bf16(normalized);
                            else if (p.norm_full_layer) v[i] = round_bf16(normalized * (scl[i] + p.norm_offset));
                            else v[i] = round_bf16(round_bf16(normalized) * round_bf16(scl[i] + p.norm_offset));
                        }
                        *(float4*)slot = make_float4(v[0], v[1], v[2], v[3]);
                        if (p.normed_out && blockIdx.x == 0) {
                            uint2 o;
                            o.x = (f32_to_bits(v[0]) >> 16) | (f32_to_bits(v[1]) & 0xFFFF0000u);
                            o.y = (f32_to_bits(v[2]) >> 16) | (f32_to_bits(v[3]) & 0xFFFF0000u);
                            *(uint2*)(p.normed_out + e) = o;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&s_sync, 1u);
            }
            wait_ge(&s_sync, 8u, err, 2u); // the normalised row is complete in LDS
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const uint32_t c = sl + lpr * j;
                float xf[32];
                if (c < C) {
                    const float4* xv = (const float4*)(xs + (size_t)c * 36);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 t = xv[i];
                        xf[4 * i] = t.x, xf[4 * i + 1] = t.y, xf[4 * i + 2] = t.z, xf[4 * i + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 32; ++i) xf[i] = 0.f;
                }
                xsm[j] = sum32(xf);
                xpack_from_f32(xq[j], xf);
            }
        }
        if (cw == 0) UZU_TL_STAMP(2); // the activation row is in registers

        // ---- work items, software pipelined: the next item's bytes travel LDS -> registers while the current one is computed ----
        float best_v = -INFINITY;
        uint32_t best_i = 0xFFFFFFFFu;
        const uint32_t item_rows = (uint32_t)(RR * rpw);
        const uint32_t total_items = my_T * g.items_per_slot;
        struct Buf {
            Codes4 w[RR][NPHYS][CPL];
            uint16_t s[RR][NPHYS][CPL], b[RR][NPHYS][CPL];
        };
        auto fetch = [&]() -> uint32_t {
            uint32_t item = 0;
            if (lane == 0) item = atomicAdd(&s_next_item, 1u);
            return __builtin_amdgcn_readfirstlane(item);
        };
        auto load_item = [&](uint32_t item, Buf& bf) {
            const uint32_t t = item / g.items_per_slot, it_in_slot = item % g.items_per_slot;
            const uint8_t* slot = smem + (size_t)(t % S) * g.slot_bytes;
#pragma unroll
            for (int rr = 0; rr < RR; ++rr) {
                const uint32_t row_local = it_in_slot * item_rows + rr * rpw + rsub;
#pragma unroll
                for (int h = 0; h < NPHYS; ++h)
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const uint32_t c_raw = sl + lpr * j;
                        const uint32_t c = c_raw < C ? c_raw : C - 1; // clamped: read, never consumed
                        bf.w[rr][h][j].a = *(const uint4*)(slot + h * g.chunk_stride + row_local * row_bytes + c * 16);
                        const uint32_t gi = (row_local * G + ((c * 32) >> gshift)) * 2;
                        bf.s[rr][h][j] = *(const uint16_t*)(slot + g.off_scales + h * g.sb_stride + gi);
                        bf.b[rr][h][j] = *(const uint16_t*)(slot + g.off_biases + h * g.sb_stride + gi);
                    }
            }
        };
        float keep_up = 0.f, keep_gate = 0.f;
        uint32_t keep_row = 0xFFFFFFFFu, act_cnt = 0;
        auto act_flush = [&]() {
            if constexpr (ACT) {
                if (keep_row != 0xFFFFFFFFu) p.out[0][keep_row] = f32_to_bf16(round_bf16(keep_up * act_bf16(p.act_type, keep_gate, s_exp_tab))); // gated_act_mul/mod.rs:5-12
                keep_row = 0xFFFFFFFFu, act_cnt = 0;
            }
        };
        auto compute_item = [&](uint32_t item, const Buf& bf) {
            const uint32_t t = item / g.items_per_slot, it_in_slot = item % g.items_per_slot;
            const uint32_t gs = blockIdx.x + t * gridDim.x;
            const int mat = __builtin_amdgcn_readfirstlane(gs >= g.slots0 ? 1 : 0);
            const uint32_t ls = mat ? gs - g.slots0 : gs;
            const uint32_t nl = mat ? p.n[1] : n_log0;
#pragma unroll
            for (int rr = 0; rr < RR; ++rr) {
                float acc[NPHYS];
#pragma unroll
                for (int h = 0; h < NPHYS; ++h) {
                    acc[h] = 0.f;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const uint32_t c = sl + lpr * j;
                        if (c < C) {
                            const float sc = bf16_to_f32(bf.s[rr][h][j]);
                            float of = bf16_to_f32(bf.b[rr][h][j]);
                            const float dq = dot32p(bf.w[rr][h][j], xq[j]); // sum (16 + q) x
                            of = fmaf(-kQ4Offset, sc, of);
                            acc[h] = fmaf(sc, dq, fmaf(of, xsm[j], acc[h]));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0); // one row at a time (register pressure)
                }
                const float v0 = row_sum_rt(acc[0], lpr);
                const float v1 = ACT ? row_sum_rt(acc[NPHYS - 1], lpr) : 0.f;
                const uint32_t lrow = ls * R + it_in_slot * item_rows + rr * rpw + rsub;
                if constexpr (ACT) {
                    // batched GatedActMul (see gemv_dec): the pair parks in lane act_cnt, the activation runs once per 64 pairs
                    float value = 1.0f * v0, gate = 1.0f * v1;
                    if (p.out_bias[0] && lrow < nl) value += bf16_to_f32(p.out_bias[0][lrow]), gate += bf16_to_f32(p.out_bias[0][lrow + p.n[0] / 2]);
                    const float up_b = round_bf16(value), gate_b = round_bf16(gate);
                    for (int gsub = 0; gsub < rpw; ++gsub) {
                        const float u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, up_b), gsub << g.lpr_log2));
                        const float g2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gate_b), gsub << g.lpr_log2));
                        const uint32_t row_g = ls * R + it_in_slot * item_rows + rr * rpw + (uint32_t)gsub;
                        if ((uint32_t)lane == act_cnt + (uint32_t)gsub && row_g < nl) keep_up = u, keep_gate = g2, keep_row = row_g;
                    }
                    act_cnt += (uint32_t)rpw;
                    if (act_cnt + (uint32_t)rpw > 64u) act_flush();
                } else if (sl == 0 && lrow < nl) {
                    // MatmulKernel epilogue with ab_scale = 1, no accumulate / soft-cap (kernel.rs:281-292)
                    float value = 1.0f * v0;
                    if (p.out_bias[mat]) value += bf16_to_f32(p.out_bias[mat][lrow]);
                    {
                        const uint16_t ob = f32_to_bf16(value);
                        if (p.out_f32) p.out_f32[lrow] = value;
                        else p.out[mat][lrow] = ob;
                        if (p.part_val) {
                            const float lv = bf16_to_f32(ob);
                            if (lv > best_v || (lv == best_v && lrow < best_i)) best_v = lv, best_i = lrow;
                        }
                    }
                }
            }
        };
        // one pipeline stage: `cur` sits in `mine`; fetch the next item, hand `cur`'s ring position back as soon as its bytes are in
        // registers, start the next item's LDS reads into `other` if its slot has been published (otherwise after the arithmetic),
        // compute `cur`.  Returns false when the wave has run out of items.
        auto stage = [&](uint32_t& cur, Buf& mine, Buf& other) -> bool {
            const uint32_t nxt = fetch();
            const bool have_next = nxt < total_items;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // `mine` is in registers
            if (lane == 0) atomicAdd(&s_done[(cur / g.items_per_slot) % S], 1u);
            bool loaded = false;
            if (have_next && lds_load(&s_filled) >= nxt / g.items_per_slot + 1) {
                asm volatile("" ::: "memory");
                load_item(nxt, other);
                loaded = true;
            }
            compute_item(cur, mine);
            if (!have_next) return false;
            if (!loaded) {
                if (!wait_ge(&s_filled, nxt / g.items_per_slot + 1, err, 4u)) return false;
                load_item(nxt, other);
            }
            cur = nxt;
            return true;
        };
        if constexpr (CPL <= 4) {
            Buf bufA, bufB;
            uint32_t cur = fetch();
            if (cur < total_items && wait_ge(&s_filled, cur / g.items_per_slot + 1, err, 4u)) {
                load_item(cur, bufA);
                while (stage(cur, bufA, bufB) && stage(cur, bufB, bufA)) {
                }
            }
        } else { // long rows (7-9 steps per lane, 112-144 registers of activations): one item at a time, the steps of a row overlap each other
            Buf buf;
            for (;;) {
                const uint32_t cur = fetch();
                if (cur >= total_items || !wait_ge(&s_filled, cur / g.items_per_slot + 1, err, 4u)) break;
                load_item(cur, buf);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&s_done[(cur / g.items_per_slot) % S], 1u);
                compute_item(cur, buf);
            }
        }
        act_flush();
        if (cw == 0) UZU_TL_STAMP(4); // this consumer has run out of items
        if (p.part_val) { // UnifiedSampling (greedy) pass 1: one (value, index) partial per workgroup
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(best_v, off, 64);
                const uint32_t oi = __shfl_xor(best_i, off, 64);
                if (ov > best_v || (ov == best_v && oi < best_i)) best_v = ov, best_i = oi;
            }
            if (lane == 0) s_bv[wave] = best_v, s_bi[wave] = best_i;
        }
    }
    if (p.part_val) {
        __syncthreads(); // every wave, the loader included, comes through here exactly once
        if (tid == 64) {
            float bv = s_bv[1];
            uint32_t bi = s_bi[1];
            for (int w2 = 2; w2 < NW; ++w2)
                if (s_bv[w2] > bv || (s_bv[w2] == bv && s_bi[w2] < bi)) bv = s_bv[w2], bi = s_bi[w2];
            p.part_val[blockIdx.x] = bv;
            p.part_idx[blockIdx.x] = bi;
        }
    }
#ifdef UZU_TIMELINE
    // loader (thread 0): stamps 0 (entry), 7 (past the start barrier), 1 (first slot issued), 3 (whole share landed); first consumer
    // (thread 64): 2 (activation row in registers), 4 (out of items) -- merged through LDS into thread 0's record in
    // tools/timeline.py's slot order
    __shared__ unsigned long long s_tl[2];
    if (tid == 64) s_tl[0] = tl_t[2], s_tl[1] = tl_t[4];
    __syncthreads();
    if (tid == 0) tl_t[2] = s_tl[0], tl_t[5] = tl_t[6] = s_tl[1], tl_t[4] = __builtin_amdgcn_s_memrealtime(); // tools/timeline.py: 4 = exit, 5 = dots done
#endif
    UZU_TL_FLUSH(p);
}

// ================================================================================================================ MFMA consumers
// gemv_stream_mfma_kernel -- the same loader / ring, with the dot products on the matrix cores ("MFMA-tiled GEMV").
//
// Why: once the bytes are on chip the int4 x bf16 dot is VALU-bound -- gemv_core.h::dot32p costs ~223 SIMD cycles per KiB of codes
// (16 pair conversions + 16 v_dot2c at ~5 cycles each; tools/valu_rate.hip, profiles/r3_valu_rate.txt) = 44 GB/s per CU at 100 % issue,
// ~24 in a real kernel, i.e. no faster than the 25 GB/s per CU the loader delivers: an LDS-staged stream with dot2 consumers can only tie
// with the register GEMV.  v_mfma_f32_16x16x32_bf16 takes the dot2 half of that work off the VALU: B = 16 weight rows x 32 k (the SAME
// pair conversion (16 + q) as bf16, one 32-bit word of codes -> one lane's 8-element B operand), A = the activation row (all 16 A rows
// carry it, so D[., j] is row j's dot product in every lane with lane & 15 == j): 166 cycles per KiB (profiles/r3_valu_rate.txt), and the
// MFMA pipe runs beside the VALU of the other waves.
//
// Work decomposition (per CU):
//   * a UNIT is a group of 16 (RGS = 1) or 32 (RGS = 2) weight rows -- for the fused up / gate matrix 8 (16) up rows + the 8 (16) gate rows of
//     the same outputs --, streamed in SLICES of ks = 2048 (RGS 1) or 1024 (RGS 2) k: a slot = one slice of one unit = 16 KiB of codes in 16
//     one-KiB pieces (a piece = one row's 1 KiB / two rows' 512 B; pieces are padded by 16 / 32 B in LDS so that the 16 rows of a B-operand read
//     fall on 16 different bank quads) + 512 B of scales + 512 B of biases = kOpsPerSlot operations;
//   * the 15 consumer waves all help with the prologue; 12 of them stream, as 3 GANGS of 4: slot q belongs to gang q % 3, member m of the gang
//     takes the items (16-row group, 128-k super-step) m, m + 4, m + 8, m + 12 of the slot's 16 -- four independent chains per wave and visit;
//   * per item: 1 ds_read_b128 of codes, 4 ds_read_b128 of the activation row in packed-dot order (XPack, shared through LDS: 64 B per 32-k
//     step; written directly by the Normalization prologue threads where a thread owns whole 8-blocks of the row), 16 conversions, 4 MFMAs in
//     two chains, then acc = fma(scale, D, fma(bias - 16 scale, sum(x_group), acc)) -- the grouped form of gemv_core.h with the group's dot
//     product from the matrix core and sum(x_group) from a table the prologue fills;
//   * a unit's partial sums (<= 12 waves) meet in LDS; one member of the gang of the unit's last slice adds them in wave order (fixed:
//     deterministic) and runs the epilogue.  Summation order differs from the register GEMV (MFMA tree inside a group, slices in k order):
//     tolerance class, <= 1 bf16 ulp against the reference like every other production kernel; NOT bit-identical to gemv_dec;
//   * the loader issues a full unit's 16 pieces as 4 runs of 4 (one M0 round trip per run, scalar row bases) and runs at the LDS-DMA rate
//     (0.68 us per 16 KiB slot = 24 GB/s per CU = 6.2 TB/s: tools/timeline.py on the `down` shape).
// Measured (r3, tools/kbench, us per launch, register GEMV in brackets): qkv 9.6 (7.9), out 5.3 (5.2), up+act 19.1 (16.5), down 9.0 (8.7),
// Llama read-out 59.2 (53.3), Qwen3.5 read-out 37.6 (36.4): the matrix cores take the dot products off the VALU (the consumers keep up with the
// loader), but every hand-off of a slot between loader and consumers costs >= 1 us of LDS-flag latency per visit, and nothing is left of the
// run-ahead: NOT the default (UZU_STREAM_MFMA=1 / debug mode 3 select it; DESIGN.md section 3 has the other three decompositions tried).
struct MfmaGeo {
    uint32_t ks;          // k per slice: 2048 (RGS 1) / 1024 (RGS 2)
    uint32_t n_slices;    // K / ks
    uint32_t ring_slots, depth;
    uint32_t off_scales, off_biases, slot_bytes;
    uint32_t units;       // units of the matrix (logical rows / rows per unit, rounded up)
    uint32_t direct;      // PRO 1: the prologue threads own whole 8-blocks of the row (K / 256 in {8, 16, 32}) and write it packed themselves
    uint32_t pro_delay;   // PRO 1: s_sleep units the prologue waves wait before their first loads (the loader's first slot goes ahead of them)
    uint32_t xp_off, sg_off, xs_off, part_off; // dynamic-LDS offsets: packed activation row, its sums per quant group, f32 staging (PRO 1), unit partials
};
typedef float mf_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 mf_bf16x8 __attribute__((ext_vector_type(8)));

// four LDS-DMA pieces with one M0 round trip: scalar bases b0 .. b3, LDS destinations lds_dst + {0, 1, 2, 3} * STEP
template <uint32_t STEP>
__device__ __forceinline__ void glds16p4_nt(uint32_t voff, const void* b0, const void* b1, const void* b2, const void* b3, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 nt\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %4 nt\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5 nt\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %6 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_dst), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "n"(STEP)
                 : "memory", "scc");
}

template <bool ACT, int PRO, int RGS>
__global__ void __launch_bounds__(1024) gemv_stream_mfma_kernel(DecGemvParams p, MfmaGeo g) {
    constexpr int NW = 16;
    constexpr uint32_t KS = RGS == 1 ? 2048u : 1024u;
    constexpr uint32_t ROWS = 16u * RGS;              // physical rows per unit
    constexpr uint32_t SUPER = KS / 128;              // super-steps (quant groups) per slice and row: 16 / 8
    constexpr uint32_t ROW_BYTES_SLICE = KS / 2;      // 1024 / 512
    constexpr uint32_t PIECE = RGS == 1 ? 1040u : 1056u; // LDS bytes from piece to piece: 1 KiB + a pad that spreads the 16 rows of a B read over the banks
    constexpr uint32_t kConsumers = 15;               // consumer waves: every one helps with the prologue ...
    constexpr uint32_t NG = 3, GW = 4;                // ... and 3 gangs of 4 stream: slot q belongs to gang q % 3, each member takes 4 of its 16 items
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_filled, s_sync, s_done[kMaxRing], s_unit_cnt[2], s_unit_rd;
    __shared__ float s_red[4];
    __shared__ uint64_t s_exp_tab[32];
    __shared__ float s_bv[NW];
    __shared__ uint32_t s_bi[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t* const err = &g_stream_err_dev;
    UZU_TL_DECL;
    UZU_TL_STAMP(0);
    const uint32_t K = p.k, C = K / 32, row_bytes = K / 2, G = K / 128;
    const uint32_t half = p.n[0] / 2;                        // ACT: gate rows start here
    const uint32_t S = g.ring_slots, n_slices = g.n_slices;
    const uint32_t my_units = blockIdx.x < g.units ? (g.units - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t smem_base = lds_addr(smem);
    // physical row of unit-local row r (0 .. ROWS - 1) of global unit u; clamped into the matrix (a clamped row is streamed and never stored)
    auto phys_row = [&](uint32_t u, uint32_t r) -> uint32_t {
        if (ACT) {
            const uint32_t q = r / 16, j = r % 16;
            const uint32_t o = (u * RGS + q) * 8 + (j & 7);
            const uint32_t oc = o < half ? o : half - 1;
            return j < 8 ? oc : half + oc;
        }
        const uint32_t row = u * ROWS + r;
        return row < p.n[0] ? row : p.n[0] - 1;
    };
    // ---- loader side: a unit's constants (scalar: the unit's four runs of consecutive rows; per lane: where its scale / bias words live)
    struct UnitCtx {
        uint32_t u, full;
        uint32_t run_row[4]; // first physical row of pieces 4i .. 4i + 3
        uint32_t sb_off[2];  // per lane: byte offset of its dword of slice 0 in the scale / bias tables, for the two 256-byte operations
    };
    constexpr uint32_t DW_PER_ROW = SUPER / 2; // dwords of scales per row and slice: 8 / 4
    auto unit_ctx = [&](uint32_t ui) -> UnitCtx {
        UnitCtx c;
        c.u = blockIdx.x + ui * gridDim.x;
        if (ACT) {
            const uint32_t o0 = c.u * RGS * 8;
            c.full = o0 + RGS * 8 <= half;
            if (RGS == 1) c.run_row[0] = o0, c.run_row[1] = o0 + 4, c.run_row[2] = half + o0, c.run_row[3] = half + o0 + 4;
            else c.run_row[0] = o0, c.run_row[1] = half + o0, c.run_row[2] = o0 + 8, c.run_row[3] = half + o0 + 8;
        } else {
            const uint32_t r0 = c.u * ROWS;
            c.full = r0 + ROWS <= p.n[0];
            for (int i = 0; i < 4; ++i) c.run_row[i] = r0 + (uint32_t)i * (ROWS / 4);
        }
#pragma unroll
        for (uint32_t op = 0; op < 2; ++op) {
            const uint32_t flat = op * 64 + (uint32_t)lane, r = flat / DW_PER_ROW, dw = flat % DW_PER_ROW;
            c.sb_off[op] = phys_row(c.u, r) * G * 2 + dw * 4; // < 2^32: rows * groups * 2 bytes of one matrix
        }
        return c;
    };
    auto issue_slot = [&](const UnitCtx& c, uint32_t sl, uint32_t pos) {
        const uint32_t slot_lds = __builtin_amdgcn_readfirstlane(smem_base + pos * g.slot_bytes);
        const uint8_t* wsl = p.w[0] + (size_t)sl * ROW_BYTES_SLICE;
        if (c.full) {
            // RGS 1: a piece = 1 KiB of one row; RGS 2: a piece = 512 B of two consecutive rows (lanes 32.. take the second)
            const uint32_t voff = RGS == 1 ? (uint32_t)lane * 16 : ((uint32_t)lane >> 5) * row_bytes + ((uint32_t)lane & 31) * 16;
            const size_t step = (size_t)row_bytes * RGS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint8_t* b = wsl + (size_t)c.run_row[i] * row_bytes;
                glds16p4_nt<PIECE>(voff, b, b + step, b + 2 * step, b + 3 * step, slot_lds + (uint32_t)i * 4 * PIECE);
            }
        } else if constexpr (RGS == 1) {
            for (uint32_t piece = 0; piece < 16; ++piece)
                glds16s_nt((uint32_t)lane * 16, wsl + (size_t)phys_row(c.u, piece) * row_bytes, slot_lds + piece * PIECE);
        } else {
            for (uint32_t piece = 0; piece < 16; ++piece) {
                const uint32_t r = piece * 2 + ((uint32_t)lane >> 5);
                glds16_nt(wsl + (size_t)phys_row(c.u, r) * row_bytes + ((uint32_t)lane & 31) * 16, __builtin_amdgcn_readfirstlane(slot_lds + piece * PIECE));
            }
        }
        // scales / biases of the slice: SUPER bf16 per row, rows side by side: [row][SUPER] -- 512 B each, two 256-byte operations
        const uint8_t* sc = (const uint8_t*)p.scales[0] + (size_t)sl * SUPER * 2;
        const uint8_t* bi = (const uint8_t*)p.biases[0] + (size_t)sl * SUPER * 2;
#pragma unroll
        for (uint32_t op = 0; op < 2; ++op) {
            glds4s(c.sb_off[op], sc, slot_lds + g.off_scales + op * 256);
            glds4s(c.sb_off[op], bi, slot_lds + g.off_biases + op * 256);
        }
    };

    // start-up as in gemv_stream_kernel: first memory requests before any wave waits for another
    constexpr int NPRE = PRO == 1 ? 8 : 1; // K <= 8192 for the Normalization prologue: E / 4 <= 8 vectors per thread
    u32x2_v x_pre[NPRE], s_pre[NPRE];
    f32x4_v n_pre[NPRE];
    uint64_t exp_entry = 0;
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        if (tid == 0) {
            s_filled = 0, s_sync = 0, s_unit_cnt[0] = 0, s_unit_cnt[1] = 0, s_unit_rd = 0;
            for (uint32_t i = 0; i < kMaxRing; ++i) s_done[i] = 0;
        }
        if (my_units) issue_slot(unit_ctx(0), 0, 0);
        UZU_TL_STAMP(1);
    } else if (PRO == 1 && wave <= 4) {
        // the vector memory path of a CU returns in order: let the loader's first slot go in front of the activation row (which has just been
        // written by the previous launch and is slow to arrive), not behind it
        for (uint32_t i = 0; i < g.pro_delay; ++i) __builtin_amdgcn_s_sleep(1);
        const uint32_t E = K / 256, pt = (uint32_t)tid - 64u;
#pragma unroll
        for (int qi = 0; qi < NPRE; ++qi) {
            const uint32_t q = (uint32_t)qi * 4 < E ? (uint32_t)qi * 4 : E - 4;
            const uint32_t e = pt * E + q;
            x_pre[qi] = *(const u32x2_v*)(p.x + e);
            s_pre[qi] = *(const u32x2_v*)((p.residual_add ? p.shortcut_in : p.x) + e);
            n_pre[qi] = *(const f32x4_v*)(p.norm_scales ? p.norm_scales + e : (const float*)p.x);
        }
        if (ACT && wave == 1) exp_entry = kExp2fTab[lane & 31];
    }
    lds_barrier();

    if (wave == 0) {
        // ================================================================================================ loader
        const uint32_t D = g.depth;
        UZU_TL_STAMP(7);
        uint32_t pos = 0, round = 0, q = 0;
        bool ok = true;
        for (uint32_t ui = 0; ui < my_units && ok; ++ui) {
            const UnitCtx c = unit_ctx(ui);
            for (uint32_t sl = 0; sl < n_slices; ++sl, ++q) {
                if (q) {
                    if (round && !wait_ge(&s_done[pos], round * GW, err, 1u)) { // one gang per slot
                        ok = false;
                        break;
                    }
                    issue_slot(c, sl, pos);
                }
                if (D > 1) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                if (q >= D) lds_store(&s_filled, q - D + 1);
                if (++pos == S) pos = 0, ++round;
            }
        }
        if (q && ok) {
            if (D > 1) {
                asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                lds_store(&s_filled, q - 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_store(&s_filled, q);
        }
        UZU_TL_STAMP(3);
    } else {
        // ============================================================================================== consumers
        const int cw = wave - 1;                      // 0 .. 14
        const uint32_t ci = (uint32_t)cw * 64 + lane; // consumer-thread index, 960 of them
        uint32_t* xp = (uint32_t*)(smem + g.xp_off);  // [C][16] packed-dot words
        float* sg = (float*)(smem + g.sg_off);        // [K / 128] sum of the activation row per quant group
        // ---- the activation row -> LDS in packed-dot order + its partial sums ------------------------------------------------
        if constexpr (PRO == 0) {
            if (ACT && cw == 0 && lane < 32) s_exp_tab[lane] = kExp2fTab[lane];
            for (uint32_t c = ci; c < C; c += kConsumers * 64) {
                XPack x;
                const float sx = xpack_load(x, p.x + (size_t)c * 32);
                u32x4_v* dst = (u32x4_v*)(xp + (size_t)c * 16);
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    u32x4_v v;
                    v.x = x.v[4 * w4], v.y = x.v[4 * w4 + 1], v.z = x.v[4 * w4 + 2], v.w = x.v[4 * w4 + 3];
                    dst[w4] = v;
                }
                // a quant group = four steps = four neighbouring lanes (all of them in the loop together: C % 4 == 0)
                float gs = sx + __shfl_xor(sx, 1, 64);
                gs += __shfl_xor(gs, 2, 64);
                if ((lane & 3) == 0) sg[c / 4] = gs;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(&s_sync, 1u);
            wait_ge(&s_sync, kConsumers, err, 2u);
        } else {
            float* xs = (float*)(smem + g.xs_off); // C slots of 36 floats (normalization staging)
            const uint32_t E = K / 256;
            const uint32_t pt = (uint32_t)tid - 64u;
            const bool direct = g.direct != 0; // the prologue threads write the packed row themselves (E in {8, 16, 32})
            if (cw < 4) {
                if (ACT && cw == 0 && lane < 32) s_exp_tab[lane] = exp_entry;
                float ss = 0.f;
#pragma unroll
                for (int qi = 0; qi < NPRE; ++qi) {
                    const uint32_t q = (uint32_t)qi * 4;
                    if (q < E) {
                        const uint32_t e = pt * E + q;
                        const u32x2_v xr = x_pre[qi];
                        float v[4] = {bits_to_f32(xr.x << 16), bits_to_f32(xr.x & 0xFFFF0000u), bits_to_f32(xr.y << 16), bits_to_f32(xr.y & 0xFFFF0000u)};
                        if (p.residual_add) {
                            const u32x2_v sr = s_pre[qi];
                            const float sc[4] = {bits_to_f32(sr.x << 16), bits_to_f32(sr.x & 0xFFFF0000u), bits_to_f32(sr.y << 16), bits_to_f32(sr.y & 0xFFFF0000u)};
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = round_bf16(v[i] + sc[i]);
                        }
                        if (p.shortcut_out && blockIdx.x == 0) {
                            uint2 o;
                            o.x = (f32_to_bits(v[0]) >> 16) | (f32_to_bits(v[1]) & 0xFFFF0000u);
                            o.y = (f32_to_bits(v[2]) >> 16) | (f32_to_bits(v[3]) & 0xFFFF0000u);
                            *(uint2*)(p.shortcut_out + e) = o;
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) ss = fmaf(v[i], v[i], ss);
                        *(float4*)(xs + (size_t)(e / 32) * 36 + e % 32) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
                ss = wave_sum(ss);
                if (lane == 0) s_red[cw] = ss;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&s_sync, 1u);
                wait_ge(&s_sync, 4u, err, 2u);
                const float total = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
                const float variance = total / (float)K - 0.0f * 0.0f;
                const float rms_inv = 1.0f / sqrtf(variance + p.norm_eps);
                float psum = 0.f, lo[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int qi = 0; qi < NPRE; ++qi) {
                    const uint32_t q = (uint32_t)qi * 4;
                    if (q < E) {
                        const uint32_t e = pt * E + q;
                        float* slot = xs + (size_t)(e / 32) * 36 + e % 32;
                        const float4 vv = *(const float4*)slot;
                        float v[4] = {vv.x, vv.y, vv.z, vv.w};
                        const f32x4_v t4 = n_pre[qi];
                        const float scl[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float normalized = (v[i] - 0.0f) * rms_inv;
                            if (!p.norm_scales) v[i] = round_bf16(normalized);
                            else if (p.norm_full_layer) v[i] = round_bf16(normalized * (scl[i] + p.norm_offset));
                            else v[i] = round_bf16(round_bf16(normalized) * round_bf16(scl[i] + p.norm_offset));
                        }
                        if (direct) {
                            // packed-dot order: word s of an 8-block = (x[s], x[4 + s]); this thread owns whole 8-blocks (E % 8 == 0)
                            if (qi & 1) {
                                u32x4_v wv;
                                wv.x = pack_bf16_pair(lo[0], v[0]), wv.y = pack_bf16_pair(lo[1], v[1]);
                                wv.z = pack_bf16_pair(lo[2], v[2]), wv.w = pack_bf16_pair(lo[3], v[3]);
                                *(u32x4_v*)(xp + (size_t)((e - 4) / 8) * 4) = wv;
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i) lo[i] = v[i];
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) psum += v[i];
                        } else {
                            *(float4*)slot = make_float4(v[0], v[1], v[2], v[3]);
                        }
                        if (p.normed_out && blockIdx.x == 0) {
                            uint2 o;
                            o.x = (f32_to_bits(v[0]) >> 16) | (f32_to_bits(v[1]) & 0xFFFF0000u);
                            o.y = (f32_to_bits(v[2]) >> 16) | (f32_to_bits(v[3]) & 0xFFFF0000u);
                            *(uint2*)(p.normed_out + e) = o;
                        }
                    }
                }
                if (direct) { // a quant group = 128 / E neighbouring prologue threads (4, 8 or 16 lanes)
                    const uint32_t nsum = 128u / E;
                    psum += __shfl_xor(psum, 1, 64);
                    psum += __shfl_xor(psum, 2, 64);
                    if (nsum >= 8) psum += __shfl_xor(psum, 4, 64);
                    if (nsum >= 16) psum += __shfl_xor(psum, 8, 64);
                    if ((pt & (nsum - 1)) == 0) sg[pt / nsum] = psum;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&s_sync, 1u);
            }
            wait_ge(&s_sync, 8u, err, 2u); // the normalised row is complete (packed, or in the f32 staging)
            if (!direct) {
                for (uint32_t c = ci; c < C; c += kConsumers * 64) {
                    float xf[32];
                    const float4* xv = (const float4*)(xs + (size_t)c * 36);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 t = xv[i];
                        xf[4 * i] = t.x, xf[4 * i + 1] = t.y, xf[4 * i + 2] = t.z, xf[4 * i + 3] = t.w;
                    }
                    XPack x;
                    xpack_from_f32(x, xf);
                    u32x4_v* dst = (u32x4_v*)(xp + (size_t)c * 16);
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) {
                        u32x4_v v;
                        v.x = x.v[4 * w4], v.y = x.v[4 * w4 + 1], v.z = x.v[4 * w4 + 2], v.w = x.v[4 * w4 + 3];
                        dst[w4] = v;
                    }
                    const float sx = sum32(xf);
                    float gs = sx + __shfl_xor(sx, 1, 64);
                    gs += __shfl_xor(gs, 2, 64);
                    if ((lane & 3) == 0) sg[c / 4] = gs;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&s_sync, 1u);
                wait_ge(&s_sync, 8u + kConsumers, err, 2u);
            }
        }
        if (cw == 0) UZU_TL_STAMP(2);

        // ---- streaming: slot q belongs to gang q % 3; member m of the gang takes items m, m + 4, m + 8, m + 12 of its 16 --------------
        float best_v = -INFINITY;
        uint32_t best_i = 0xFFFFFFFFu;
        const uint32_t jcol = (uint32_t)lane & 15, kb = (uint32_t)lane >> 4;
        const uint32_t gang = (uint32_t)cw / GW, member = (uint32_t)cw % GW; // consumers 12 .. 14 only helped with the prologue
        const uint32_t inv_gangs = n_slices < NG ? n_slices : NG;           // gangs that meet in a unit
        const uint32_t others = inv_gangs * GW - 1;                          // partial sums a unit's combiner waits for
        uint32_t mask = 0x00780078u, magic = 0x41804180u;
        asm("" : "+s"(mask));
        asm("" : "+v"(magic));
        uint32_t pos = 0, q = 0, qg = 0; // qg = q % NG
        bool ok = gang < NG;
        for (uint32_t ui = 0; ui < my_units && ok; ++ui) {
            const uint32_t u = blockIdx.x + ui * gridDim.x;
            const uint32_t g_first = qg;                                     // gang of the unit's first slice
            float acc[RGS];
#pragma unroll
            for (int rq = 0; rq < RGS; ++rq) acc[rq] = 0.f;
            for (uint32_t sl = 0; sl < n_slices; ++sl, ++q) {
                if (qg == gang) {
                    if (!wait_ge(&s_filled, q + 1, err, 4u)) {
                        ok = false;
                        break;
                    }
                    const uint8_t* slot = smem + (size_t)pos * g.slot_bytes;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        constexpr int kHalf = RGS == 1 ? 4 : 2;                  // items per row group: 16 / 8 super-steps over 4 members
                        const int rq = it / kHalf;
                        const uint32_t ss = (uint32_t)(it % kHalf) * 4 + member;
                        const uint32_t r = (uint32_t)rq * 16 + jcol; // this lane's B column = unit row r
                        const uint8_t* rowp = RGS == 1 ? slot + r * PIECE : slot + (r >> 1) * PIECE + (r & 1) * 512;
                        const u32x4_v wl = *(const u32x4_v*)(rowp + ss * 64 + kb * 16);
                        const uint32_t cstep = (sl * KS) / 32 + ss * 4 + kb; // this lane's 32-k step of the activation row
                        const u32x4_v* xa = (const u32x4_v*)(xp + (size_t)cstep * 16);
                        const float sxg = sg[sl * SUPER + ss];               // quant group = super-step
                        const float sc = bf16_to_f32(((const uint16_t*)(slot + g.off_scales))[r * SUPER + ss]);
                        float of = bf16_to_f32(((const uint16_t*)(slot + g.off_biases))[r * SUPER + ss]);
                        const uint32_t ws[4] = {wl.x, wl.y, wl.z, wl.w};
                        mf_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            u32x4_v bv;
                            bv.x = ((ws[w] << 3) & mask) | magic, bv.y = ((ws[w] >> 1) & mask) | magic;
                            bv.z = ((ws[w] >> 5) & mask) | magic, bv.w = ((ws[w] >> 9) & mask) | magic;
                            const u32x4_v av = xa[w];
                            if (w & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, av), __builtin_bit_cast(mf_bf16x8, bv), d1, 0, 0, 0);
                            else d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, av), __builtin_bit_cast(mf_bf16x8, bv), d0, 0, 0, 0);
                        }
                        of = fmaf(-kQ4Offset, sc, of);
                        acc[rq] = fmaf(sc, d0.x + d1.x, fmaf(of, sxg, acc[rq]));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) atomicAdd(&s_done[pos], 1u);
                }
                if (++pos == S) pos = 0;
                if (++qg == NG) qg = 0;
            }
            if (!ok) break;
            // does this gang hold a part of the unit?  (slices went to gangs g_first, g_first + 1, ... mod 3)
            const uint32_t dist = gang >= g_first ? gang - g_first : gang + NG - g_first;
            if (dist >= inv_gangs) continue;
            // the partial sums of the unit meet in LDS; one member of the gang of the LAST slice adds them in consumer order (fixed:
            // deterministic) and writes the unit's outputs.  Two parities: unit ui's partials may be written once unit ui - 2's have been read
            // (the read acknowledgements are handed on in unit order).
            float* part = (float*)(smem + g.part_off) + (size_t)(ui & 1) * 16 * 32; // [parity][consumer][32]
            const uint32_t g_last = (g_first + n_slices - 1) % NG;
            const uint32_t comb = g_last * GW + (ui & (GW - 1));
            if ((uint32_t)cw != comb) {
                if (ui >= 2 && !wait_ge(&s_unit_rd, ui - 1, err, 16u)) break;
                if (kb == 0) {
#pragma unroll
                    for (int rq = 0; rq < RGS; ++rq) part[(size_t)cw * 32 + rq * 16 + jcol] = acc[rq];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&s_unit_cnt[ui & 1], 1u);
                continue;
            }
            if (!wait_ge(&s_unit_cnt[ui & 1], ((ui >> 1) + 1) * others, err, 16u)) break;
            float tot[RGS];
#pragma unroll
            for (int rq = 0; rq < RGS; ++rq) tot[rq] = 0.f;
            for (uint32_t m2 = 0; m2 < NG * GW; ++m2) {
                const uint32_t g2 = m2 / GW, d2 = g2 >= g_first ? g2 - g_first : g2 + NG - g_first;
                if (d2 >= inv_gangs) continue;
#pragma unroll
                for (int rq = 0; rq < RGS; ++rq) tot[rq] += m2 == (uint32_t)cw ? acc[rq] : part[(size_t)m2 * 32 + rq * 16 + jcol];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ui && !wait_ge(&s_unit_rd, ui, err, 16u)) break; // acknowledge in unit order
            if (lane == 0) atomicAdd(&s_unit_rd, 1u);
            // epilogue: lane j (< 16) holds row j of every 16-row group (all lanes with lane & 15 == j hold the same value)
#pragma unroll
            for (int rq = 0; rq < RGS; ++rq) {
                if constexpr (ACT) {
                    const uint32_t o = (u * RGS + (uint32_t)rq) * 8 + (jcol & 7);
                    float value = 1.0f * tot[rq];
                    const bool valid = o < half;
                    if (p.out_bias[0] && valid) value += bf16_to_f32(p.out_bias[0][jcol < 8 ? o : half + o]);
                    const float vb = round_bf16(value);
                    const float gate_b = __shfl(vb, (lane & 7) + 8, 64); // the gate row of output (lane & 7) sits 8 lanes up
                    if (lane < 8 && valid) p.out[0][o] = f32_to_bf16(round_bf16(vb * act_bf16(p.act_type, gate_b, s_exp_tab))); // gated_act_mul/mod.rs:5-12
                } else {
                    const uint32_t row = u * ROWS + (uint32_t)rq * 16 + jcol;
                    if (lane < 16 && row < p.n[0]) {
                        float value = 1.0f * tot[rq];
                        if (p.out_bias[0]) value += bf16_to_f32(p.out_bias[0][row]);
                        const uint16_t ob = f32_to_bf16(value);
                        if (p.out_f32) p.out_f32[row] = value;
                        else p.out[0][row] = ob;
                        if (p.part_val) {
                            const float lv = bf16_to_f32(ob);
                            if (lv > best_v || (lv == best_v && row < best_i)) best_v = lv, best_i = row;
                        }
                    }
                }
            }
        }
        if (cw == 0) UZU_TL_STAMP(4);
        if (p.part_val) {
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(best_v, off, 64);
                const uint32_t oi = __shfl_xor(best_i, off, 64);
                if (ov > best_v || (ov == best_v && oi < best_i)) best_v = ov, best_i = oi;
            }
            if (lane == 0) s_bv[wave] = best_v, s_bi[wave] = best_i;
        }
    }
    if (p.part_val) {
        __syncthreads();
        if (tid == 64) {
            float bv = s_bv[1];
            uint32_t bi = s_bi[1];
            for (int w2 = 2; w2 < NW; ++w2)
                if (s_bv[w2] > bv || (s_bv[w2] == bv && s_bi[w2] < bi)) bv = s_bv[w2], bi = s_bi[w2];
            p.part_val[blockIdx.x] = bv;
            p.part_idx[blockIdx.x] = bi;
        }
    }
#ifdef UZU_TIMELINE
    __shared__ unsigned long long s_tl[2];
    if (tid == 64) s_tl[0] = tl_t[2], s_tl[1] = tl_t[4];
    __syncthreads();
    if (tid == 0) tl_t[2] = s_tl[0], tl_t[5] = tl_t[6] = s_tl[1], tl_t[4] = __builtin_amdgcn_s_memrealtime();
#endif
    UZU_TL_FLUSH(p);
}

// ---------------------------------------------------------------------------------------------------------------- host side
static int stream_mode() { // UZU_DEC_STREAM: 0 = never, 1 (default) = the bandwidth regime, 2 = every supported shape (tests / A-B runs)
    static const int env = [] {
        const char* e = getenv("UZU_DEC_STREAM");
        return e ? atoi(e) : 1;
    }();
    return env;
}
static int g_stream_override = -1;
extern "C" void uzu_hip_debug_set_decode_stream(int mode) { g_stream_override = mode; } // -1 = environment / default
extern "C" uint32_t uzu_hip_debug_decode_stream_error(void) { // bit mask of bounded spins that gave up since the last call (0 = none)
    uint32_t v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_stream_err_dev), 4) != hipSuccess) {
        (void)hipGetLastError();
        return 0xFFFFFFFFu;
    }
    if (v) {
        const uint32_t zero = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stream_err_dev), &zero, 4);
    }
    return v;
}

// per device: which devices have run a weight-stream kernel (bit d); the error word is a __device__ symbol, i.e. one per device, and
// the symbol copy acts on the CURRENT device -- callers make their context's device current first (engine.hip does)
static uint32_t g_stream_launched = 0;
uzu_status gemv_stream_check() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(g_stream_launched & (1u << (dev & 31)))) return UZU_OK;
    const uint32_t v = uzu_hip_debug_decode_stream_error();
    if (v) {
        set_error("gemv_stream: a bounded wait inside a weight-stream kernel gave up (mask 0x%x): its outputs are invalid", v);
        return UZU_ERR_HIP;
    }
    return UZU_OK;
}

static int stream_cpl(const DecGemvParams& p) {
    const int lpr_log2 = gemv_lpr_log2(p.k);
    const uint32_t C = p.k / 32, lpr = 1u << lpr_log2;
    return (int)((C + lpr - 1) / lpr);
}

bool gemv_stream_supported(const DecGemvParams& p) {
    if (p.bits != 4 || p.b_kind != UZU_MATMUL_B_SCALE_BIAS || p.conv_w || p.dg_o || p.x_rht_bits || p.in_rht_bits) return false;
    if (p.k % 32 || p.group_size % 32 || (p.group_size & (p.group_size - 1))) return false;
    const uint32_t G = (p.k + p.group_size - 1) / p.group_size;
    if (p.k % p.group_size || (G & 1)) return false; // scale rows are fetched in 4-byte units
    const bool normed = p.norm_scales || p.norm_plain;
    if (normed && (p.k % 1024 || p.k > 8192)) return false;
    if (p.act_mul && (p.n[1] || (p.n[0] & 1))) return false;
    const int cpl = stream_cpl(p);
    if (!(cpl == 1 || cpl == 2 || cpl == 3 || cpl == 4 || cpl == 7 || cpl == 9)) return false;
    if ((normed || p.act_mul) && cpl > 4) return false; // K <= 8192 for the prologue / the fused up-gate rows (two double-buffered rows of codes)
    for (int i = 0; i < 2; ++i) {
        if (!p.n[i]) continue;
        if ((uintptr_t)p.w[i] % 16 || (uintptr_t)p.scales[i] % 4 || (uintptr_t)p.biases[i] % 4) return false;
    }
    if ((uintptr_t)p.x % 16) return false;
    return true;
}

static bool stream_mfma_on() { // UZU_STREAM_MFMA=1 / debug mode 3: matrix-core consumers where supported (measured slower: A/B runs and tests)
    static const int env = [] {
        const char* e = getenv("UZU_STREAM_MFMA");
        return e ? atoi(e) : 0;
    }();
    return g_stream_override == 3 || (g_stream_override < 0 && env != 0);
}
bool gemv_stream_mfma_supported(const DecGemvParams& p) {
    if (p.bits != 4 || p.b_kind != UZU_MATMUL_B_SCALE_BIAS || p.conv_w || p.dg_o || p.n[1]) return false;
    if (p.group_size != 128 || p.k % 1024 || p.k < 1024 || p.k > 32768) return false;
    const bool normed = p.norm_scales || p.norm_plain;
    if (normed && p.k > 8192) return false;
    if (p.act_mul && (p.n[0] & 1)) return false;
    if ((uintptr_t)p.w[0] % 16 || (uintptr_t)p.scales[0] % 4 || (uintptr_t)p.biases[0] % 4 || (uintptr_t)p.x % 16) return false;
    return true;
}

bool gemv_stream_wanted(const DecGemvParams& p) {
    const int mode = g_stream_override >= 0 ? g_stream_override : stream_mode();
    if (mode == 0 || exact_mode() || !(gemv_stream_supported(p) || (stream_mfma_on() && gemv_stream_mfma_supported(p)))) return false;
    if (mode >= 2) return true;
    // default: only where the LDS stream measured faster than the register GEMV -- short rows in the bandwidth regime (the Qwen3.5 read-out,
    // 248320 x 1024: 33.0 against 36.4 us); at K >= 4096 it ties or loses (profiles/r3_kbench_stream_ab.txt)
    const uint64_t weight_bytes = ((uint64_t)p.n[0] + p.n[1]) * p.k * p.bits / 8;
    return weight_bytes >= (64ull << 20) && p.k <= 2048;
}

template <int CPL, bool ACT, int PRO, int NW>
static uzu_status launch_stream(hipStream_t s, const DecGemvParams& p, const StreamGeo& g, uint32_t grid, size_t lds) {
    static LdsLimit lim;
    if (!raise_lds_limit(lim, (const void*)gemv_stream_kernel<CPL, ACT, PRO, NW>, lds)) {
        set_error("gemv_stream: %zu bytes of LDS are not available", lds);
        return UZU_ERR_UNSUPPORTED;
    }
    return launch_check([&] { hipLaunchKernelGGL((gemv_stream_kernel<CPL, ACT, PRO, NW>), dim3(grid), dim3(64 * NW), lds, s, p, g); }, "gemv_stream");
}
template <int CPL, int NW> static uzu_status launch_stream_c(hipStream_t s, const DecGemvParams& p, const StreamGeo& g, uint32_t grid, size_t lds) {
    const bool normed = p.norm_scales || p.norm_plain;
    if constexpr (CPL <= 4) {
        if (normed) return p.act_mul ? launch_stream<CPL, true, 1, NW>(s, p, g, grid, lds) : launch_stream<CPL, false, 1, NW>(s, p, g, grid, lds);
    }
    if constexpr (CPL <= 4) {
        if (p.act_mul) return launch_stream<CPL, true, 0, NW>(s, p, g, grid, lds);
    }
    return launch_stream<CPL, false, 0, NW>(s, p, g, grid, lds);
}
// waves per workgroup: 16 (15 consumers; 128 registers per wave) where the activation row is short (K <= 4096), 8 (256 registers) beyond
static int stream_waves(int cpl) {
    static const int env = [] { // UZU_STREAM_WAVES=8: 8-wave workgroups everywhere (A/B runs)
        const char* e = getenv("UZU_STREAM_WAVES");
        return e ? atoi(e) : 16;
    }();
    return (cpl <= 2 && env >= 16) ? 16 : 8;
}

template <bool ACT, int PRO, int RGS>
static uzu_status launch_stream_mfma(hipStream_t s, const DecGemvParams& p, const MfmaGeo& g, uint32_t grid, size_t lds) {
    static LdsLimit lim;
    if (!raise_lds_limit(lim, (const void*)gemv_stream_mfma_kernel<ACT, PRO, RGS>, lds)) {
        set_error("gemv_stream_mfma: %zu bytes of LDS are not available", lds);
        return UZU_ERR_UNSUPPORTED;
    }
    return launch_check([&] { hipLaunchKernelGGL((gemv_stream_mfma_kernel<ACT, PRO, RGS>), dim3(grid), dim3(1024), lds, s, p, g); }, "gemv_stream_mfma");
}
uzu_status gemv_stream_mfma(hipStream_t s, const DecGemvParams& p_in, int num_cus, uint32_t* grid_out) {
    DecGemvParams p = p_in;
#ifdef UZU_TIMELINE
    p.tl = timeline_next_slot();
#endif
    const bool act = p.act_mul != 0, normed = p.norm_scales || p.norm_plain;
    MfmaGeo g{};
    const int rgs = p.k % 2048 == 0 ? 1 : 2;
    g.ks = rgs == 1 ? 2048u : 1024u;
    g.n_slices = p.k / g.ks;
    const uint32_t rows = 16u * (uint32_t)rgs, out_per_unit = act ? rows / 2 : rows;
    const uint32_t n_log = act ? p.n[0] / 2 : p.n[0];
    g.units = (n_log + out_per_unit - 1) / out_per_unit;
    uint32_t grid = g.units < (uint32_t)num_cus ? g.units : (uint32_t)num_cus;
    if (p.part_val && p.part_capacity && grid > p.part_capacity) grid = p.part_capacity;
    const uint32_t piece_stride = rgs == 1 ? 1040u : 1056u;
    g.off_scales = 16 * piece_stride, g.off_biases = g.off_scales + 512, g.slot_bytes = g.off_biases + 512;
    static const int depth_env = [] {
        const char* e = getenv("UZU_STREAM_DEPTH");
        return e ? atoi(e) : 2;
    }();
    g.depth = depth_env >= 2 ? 2 : 1;
    const size_t C = p.k / 32;
    const uint32_t E = p.k / 256;
    g.direct = normed && (E == 8 || E == 16 || E == 32); // the prologue threads own whole 8-blocks of the row: they pack it themselves
    static const int delay_env = [] { // UZU_STREAM_PRO_DELAY: s_sleep units before the prologue waves' first loads (A/B runs)
        const char* e = getenv("UZU_STREAM_PRO_DELAY");
        return e ? atoi(e) : 8;
    }();
    g.pro_delay = delay_env > 0 ? (uint32_t)delay_env : 0;
    const size_t xp_bytes = (size_t)p.k * 2, st_bytes = ((size_t)p.k / 128 * 4 + 15) / 16 * 16, xs_bytes = normed ? (C * 36 + 16) * sizeof(float) : 0, part_bytes = 2 * 16 * 32 * 4;
    const size_t budget = 160u * 1024 - 2048;
    uint32_t ring = (uint32_t)((budget - xp_bytes - st_bytes - xs_bytes - part_bytes) / g.slot_bytes);
    if (ring > kMaxRing) ring = kMaxRing;
    if (ring < g.depth + 2) {
        set_error("gemv_stream_mfma: a ring of %u slots is too short (k %u)", ring, p.k);
        return UZU_ERR_UNSUPPORTED;
    }
    g.ring_slots = ring;
    g.xp_off = ring * g.slot_bytes;
    g.sg_off = g.xp_off + (uint32_t)xp_bytes;
    g.xs_off = g.sg_off + (uint32_t)st_bytes;
    g.part_off = g.xs_off + (uint32_t)xs_bytes;
    const size_t lds = (size_t)g.part_off + part_bytes;
    if (grid_out) *grid_out = grid;
    if (rgs == 1) {
        if (normed) return act ? launch_stream_mfma<true, 1, 1>(s, p, g, grid, lds) : launch_stream_mfma<false, 1, 1>(s, p, g, grid, lds);
        return act ? launch_stream_mfma<true, 0, 1>(s, p, g, grid, lds) : launch_stream_mfma<false, 0, 1>(s, p, g, grid, lds);
    }
    if (normed) return act ? launch_stream_mfma<true, 1, 2>(s, p, g, grid, lds) : launch_stream_mfma<false, 1, 2>(s, p, g, grid, lds);
    return act ? launch_stream_mfma<true, 0, 2>(s, p, g, grid, lds) : launch_stream_mfma<false, 0, 2>(s, p, g, grid, lds);
}

uzu_status gemv_stream(hipStream_t s, const DecGemvParams& p_in, int num_cus, uint32_t* grid_out) {
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        g_stream_launched |= 1u << (dev & 31);
    }
    if (stream_mfma_on() && gemv_stream_mfma_supported(p_in)) return gemv_stream_mfma(s, p_in, num_cus, grid_out); // matrix-core consumers
    DecGemvParams p = p_in;
#ifdef UZU_TIMELINE
    p.tl = timeline_next_slot();
#endif
    if (!gemv_stream_supported(p)) {
        set_error("gemv_stream: unsupported shape (bits %u, k %u, group %u)", p.bits, p.k, p.group_size);
        return UZU_ERR_UNSUPPORTED;
    }
    const bool act = p.act_mul != 0, normed = p.norm_scales || p.norm_plain;
    const int cpl = stream_cpl(p);
    const int nphys = act ? 2 : 1;
    StreamGeo g{};
    g.lpr_log2 = (uint32_t)gemv_lpr_log2(p.k);
    const uint32_t rpw = 64u >> g.lpr_log2;
    const uint32_t rr = act ? (cpl >= 2 ? 1u : 2u) : (cpl >= 3 ? 1u : (cpl == 2 ? 2u : 4u));
    const uint32_t item_rows = rr * rpw;
    const uint32_t row_bytes = p.k / 2, G = p.k / p.group_size;
    static const uint32_t slot_target = [] { // UZU_STREAM_SLOT_KB: bytes of codes per slot (A/B runs)
        const char* e = getenv("UZU_STREAM_SLOT_KB");
        return (uint32_t)(e && atoi(e) > 0 ? atoi(e) : 16) << 10;
    }();
    uint32_t items = slot_target / (nphys * item_rows * row_bytes);
    if (items < 1) items = 1;
    for (;; --items) { // the largest slot whose VMEM operation count fits the constant the loader's counted waits are written for
        g.items_per_slot = items;
        g.rows_per_slot = items * item_rows;
        g.chunk_stride = (g.rows_per_slot * row_bytes + 1023) / 1024 * 1024;
        g.sb_stride = (g.rows_per_slot * G * 2 + 255) / 256 * 256;
        g.ops_per_slot = nphys * (g.chunk_stride / 1024 + 2 * (g.sb_stride / 256));
        if (g.ops_per_slot <= kOpsPerSlot || items == 1) break;
    }
    if (g.ops_per_slot > kOpsPerSlot) {
        set_error("gemv_stream: %u VMEM operations per slot (k %u) exceed the %u the loader is written for", g.ops_per_slot, p.k, kOpsPerSlot);
        return UZU_ERR_UNSUPPORTED;
    }
    g.off_scales = nphys * g.chunk_stride;
    g.off_biases = g.off_scales + nphys * g.sb_stride;
    g.slot_bytes = g.off_biases + nphys * g.sb_stride;
    static const int depth_env = [] { // UZU_STREAM_DEPTH=1: one slot (instead of two) may be in flight behind the one being issued (A/B runs)
        const char* e = getenv("UZU_STREAM_DEPTH");
        return e ? atoi(e) : 2;
    }();
    g.depth = depth_env >= 2 ? 2 : 1;
    const size_t xs_bytes = normed ? ((size_t)(p.k / 32) * 36 + 16) * sizeof(float) : 0;
    const size_t lds_budget = 160u * 1024 - 2048 - 256; // static LDS of the kernel (flags, tables), the padding dump, margin
    uint32_t ring = (uint32_t)((lds_budget - xs_bytes) / g.slot_bytes);
    static const uint32_t ring_cap = [] {
        const char* e = getenv("UZU_STREAM_RING");
        return (uint32_t)(e && atoi(e) > 0 ? atoi(e) : (int)kMaxRing);
    }();
    if (ring > ring_cap) ring = ring_cap;
    if (ring > kMaxRing) ring = kMaxRing;
    if (ring < g.depth + 2) {
        set_error("gemv_stream: a ring of %u slots of %u bytes is too short", ring, g.slot_bytes);
        return UZU_ERR_UNSUPPORTED;
    }
    g.ring_slots = ring;
    const uint32_t n_log0 = act ? p.n[0] / 2 : p.n[0];
    g.slots0 = (n_log0 + g.rows_per_slot - 1) / g.rows_per_slot;
    g.slots1 = (p.n[1] + g.rows_per_slot - 1) / g.rows_per_slot;
    g.xs_off = ring * g.slot_bytes;
    g.dump_off = g.xs_off + (uint32_t)xs_bytes;
    const uint32_t T = g.slots0 + g.slots1;
    uint32_t grid = T < (uint32_t)num_cus ? T : (uint32_t)num_cus;
    if (p.part_val && p.part_capacity && grid > p.part_capacity) grid = p.part_capacity;
    if (grid_out) *grid_out = grid;
    const size_t lds = (size_t)ring * g.slot_bytes + xs_bytes + 256;
    const bool wide = stream_waves(cpl) == 16;
    switch (cpl) {
    case 1: return wide ? launch_stream_c<1, 16>(s, p, g, grid, lds) : launch_stream_c<1, 8>(s, p, g, grid, lds);
    case 2: return wide ? launch_stream_c<2, 16>(s, p, g, grid, lds) : launch_stream_c<2, 8>(s, p, g, grid, lds);
    case 3: return launch_stream_c<3, 8>(s, p, g, grid, lds);
    case 4: return launch_stream_c<4, 8>(s, p, g, grid, lds);
    case 7: return launch_stream_c<7, 8>(s, p, g, grid, lds);
    default: return launch_stream_c<9, 8>(s, p, g, grid, lds);
    }
}

} // namespace k
} // namespace uzu
