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

// (The matrix-core consumer variants of this engine -- gemv_stream_mfma_kernel, four decompositions, rounds 3-4 -- measured slower than the
// dot2 consumers on every shape (profiles/r3_kbench_stream_ab.txt) and were removed from the library in round 5; git history holds them.)

// ---------------------------------------------------------------------------------------------------------------- host side
static int stream_mode() { // UZU_DEC_STREAM: 0 = never, 1 (default) = the bandwidth regime, 2 = every supported shape (tests / A-B runs)
    static const int env = [] {
        const char* e = lab_env("UZU_DEC_STREAM");
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

bool gemv_stream_wanted(const DecGemvParams& p) {
    const int mode = g_stream_override >= 0 ? g_stream_override : stream_mode();
    if (mode == 0 || exact_mode() || !gemv_stream_supported(p)) return false;
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
        const char* e = lab_env("UZU_STREAM_WAVES");
        return e ? atoi(e) : 16;
    }();
    return (cpl <= 2 && env >= 16) ? 16 : 8;
}

uzu_status gemv_stream(hipStream_t s, const DecGemvParams& p_in, int num_cus, uint32_t* grid_out) {
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        g_stream_launched |= 1u << (dev & 31);
    }
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
        const char* e = lab_env("UZU_STREAM_SLOT_KB");
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
        const char* e = lab_env("UZU_STREAM_DEPTH");
        return e ? atoi(e) : 2;
    }();
    g.depth = depth_env >= 2 ? 2 : 1;
    const size_t xs_bytes = normed ? ((size_t)(p.k / 32) * 36 + 16) * sizeof(float) : 0;
    const size_t lds_budget = 160u * 1024 - 2048 - 256; // static LDS of the kernel (flags, tables), the padding dump, margin
    uint32_t ring = (uint32_t)((lds_budget - xs_bytes) / g.slot_bytes);
    static const uint32_t ring_cap = [] {
        const char* e = lab_env("UZU_STREAM_RING");
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
