// k_decode.hip -- fused batch-1 decode kernels for gfx950.
//
// Batch-1 decode of a small model is launch/latency bound on MI355X: a dependent kernel boundary costs
// ~1.5 us and every kernel pays ~1 us of dependent memory latency, while a whole Qwen3.5-0.8B layer streams
// only ~20 MB (3 us at HBM speed).  The engine therefore runs a layer as 5-6 kernels instead of the
// reference's 9-14 encodes (transformer_layer.rs:194-238), by moving the reference's small kernels into
// prologues/epilogues of the weight-streaming kernels:
//
//   gemv_dec       [residual add + RMSNorm prologue] -> int4/int8 GEMV over one or two weight matrices
//                  -> [SiLU(gate)*up | arg-max partial] epilogue
//                  = Normalization + MatmulKernel (+ second MatmulKernel) (+ GatedActMul | UnifiedSampling pass 1)
//   delta_dec      conv update + SiLU, delta-rule state update, RMSNorm * SiLU(z) gate
//                  = DeltaNetConvUpdate + DeltaNetUpdate
//   attn_dec       per-head q/k RMSNorm + RoPE + KV-cache append + split-KV attention pass 1
//                  = QKVNorm x2 + AttentionPrepare + AttentionTwoPass1 (own split count)
//   attn_merge     split merge + sigmoid gate = AttentionTwoPass2 + SigmoidGate
//   argmax_commit  arg-max pass 2 + token commit (next input token, history, context length)
//
// Arithmetic is the SAME as in the stand-alone kernels (same per-lane element assignment, same reduction
// trees, same rounding points): gemv_dec / delta_dec results are bit-identical to the unfused chain
// (tests/test_gpu_model.py::test_fused_decode_matches_unfused).  attn_dec uses its own KV split (more
// workgroups than the reference's 32 blocks) and is tolerance-equal.
#include <stdlib.h>

#include <type_traits>

#include "decode_epilogue.h"
#include "device_utils.h"
#include "gemv_core.h"
#include "kernels.h"
#include "kernels_decode.h"
#include "rht_stripe.h"

#ifndef UZU_GEMV_PRELOAD2
#ifndef UZU_GEMV_RAWX
#define UZU_GEMV_RAWX 1 // 0 = a plain bf16 activation row goes through f32 registers before it is packed for the dot unit (A/B builds)
#endif
#ifndef UZU_ATTN_SPEC
#define UZU_ATTN_SPEC 1 // 0 = attn_dec waits for the context length before its first K / V loads (A/B builds)
#endif
#define UZU_GEMV_PRELOAD2 1 // 0 = the second step of a wave's first batch is requested after the prologue (A/B builds)
#endif

namespace uzu {
namespace k {

// ---------------------------------------------------------------------------------------------- gemv_dec
// No LDS, no workgroup barrier: a wave is self-sufficient.
//   * lane mapping / arithmetic of gemv_core.h: lane `sl` of the row's lpr lanes owns the 32-element steps
//     sl + lpr*j, j < CPL, of the activation row and keeps them in registers for the whole kernel (CPL <= 4,
//     i.e. K <= 8192; larger K re-reads x per step, CPL == 0);
//   * optional Normalization prologue: each lane loads ITS steps of x / shortcut / norm scales, the sum of
//     squares is reduced over the lpr lanes (same tree as the stand-alone kernel), the lane normalises its own
//     steps in registers -- no redistribution, no second memory round trip;
//   * all global loads of the first row batch (codes, scales, offsets) are issued BEFORE the prologue, the
//     next batch is prefetched while the current one is computed (tools/microbench2: a kernel boundary costs
//     1.6 us, a dependent HBM round trip 0.3-0.5 us: the kernel should pay exactly one of the latter);
//   * epilogues: plain store | SiLU(gate)*up | arg-max partial per workgroup.
//   * PRO selects the prologue at compile time (0 plain row, 1 Normalization, 2 DeltaNet norm-gate): a run-time branch in
//     front of the first weight loads -- even a wave-uniform one -- makes the compiler fall back to vmcnt(0) waits.
//   * NW = waves per workgroup.  4 everywhere except the bandwidth regime's prologue kernels (K >= 4096, >= 16 MB of weights,
//     persistent grid): there every resident workgroup pulls the same 32 KB (x, shortcut, f32 norm scales at K = 4096) through
//     its CU's L1 and out of the same few L2 lines -- 1280 workgroups x 32 KB: the activation vector arrives after 3-4 us and
//     the prologue ends after 6-8 us of a 19 us kernel (Llama-3-8B up-projection, tools/timeline.py).  With 12-16 waves per
//     workgroup (one workgroup per CU) waves 0-3 run the same 256-thread prologue (same element mapping, same reduction
//     order: bit-identical) and the other waves only keep their weight loads in flight: a quarter of the traffic.
// One 32-element stripe of the prologue's LDS row through a randomised Hadamard transform (rht_stripe.h); OutputRht is followed by the linear's
// bias, rounded again (MatmulDOps::rht_factors, kernel.rs:296-303).
template <bool INPUT>
__device__ __forceinline__ void rht_stripe(float* slot, uint32_t bits, const uint16_t* bias) {
    float a[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 t = ((const float4*)slot)[i];
        a[4 * i] = t.x, a[4 * i + 1] = t.y, a[4 * i + 2] = t.z, a[4 * i + 3] = t.w;
    }
    rht_stripe_regs<INPUT>(a, bits);
    if (!INPUT && bias) {
#pragma unroll
        for (int i = 0; i < 32; ++i) a[i] = round_bf16(a[i] + bf16_to_f32(bias[i]));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) ((float4*)slot)[i] = make_float4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
}

template <int BITS, int CPLT, int R, bool ACT, int KIND, int PRO, bool CONV, int NW>
__global__ void __launch_bounds__(64 * NW) gemv_dec_kernel(const void* a0, const void* a1, const void* a2, uint32_t k_arg, int lpr_log2, const uint8_t* w0,
                                                       const uint16_t* s0, const uint16_t* o0, DecGemvParams p) {
    // The six pointers every address computation of the first loads starts from are separate leading kernel arguments:
    // with -mllvm -amdgpu-kernarg-preload-count=12 (csrc/Makefile, this file only) the command processor places them in
    // SGPRs at wave launch, so the first loads do not wait for a scalar load of the 300-byte parameter block
    // (tools/lat_lab: -0.1..0.3 us per launch).  They duplicate p.x / p.shortcut_in / p.norm_scales (or p.dg_o / p.dg_sz /
    // p.dg_w for the norm-gate prologue), p.k (the activation loads' addresses depend on it) and p.w[0] / p.scales[0] /
    // p.biases[0]: 14 dwords, the most the user-SGPR budget admits.
    p.k = k_arg;
    if (PRO == 2) {
        p.dg_o = (const float*)a0, p.dg_sz = (const float*)a1, p.dg_w = (const float*)a2;
    } else {
        p.x = (const uint16_t*)a0, p.shortcut_in = (const uint16_t*)a1, p.norm_scales = (const float*)a2;
    }
    p.w[0] = w0, p.scales[0] = s0, p.biases[0] = o0;
    using Codes = typename CodesT<BITS>::type;
    constexpr int STEP_BYTES = 4 * BITS;
    constexpr int NPHYS = ACT ? 2 : 1;
    constexpr int CPL = CPLT == 0 ? 1 : CPLT; // register-resident steps (CPLT == 0: streaming over j)
    constexpr int NT = 64 * NW;
    static_assert(NW == 4 || (!CONV && PRO != 2 && (PRO == 1 || PRO == 3 || PRO == 5 || (CPLT == 0 && BITS == 4))), "wide workgroups: prologue kernels of the bandwidth regime only");
    static_assert(PRO != 3 || (!ACT && !CONV), "an RHT linear's outputs go through OutputRht before any epilogue could use them");
    // PRO == 5 (round 5): PRO == 3's prologue + the STRIPE epilogue -- a workgroup owns whole 32-row Hadamard blocks of the output, parks the raw rows of a
    // block in LDS and one wave runs the linear's OutputRht (+ bias) on it and what follows (the GatedActMul + the next linear's InputRht of
    // rht_mlp_join, or the DeltaNet conv of rht_out_rows) before anything is stored: the join launches behind the up / in-projection disappear
    static_assert(PRO != 5 || ((NW == 4 || NW == 8) && R == 1 && !CONV), "the stripe epilogue: 4- or 8-wave workgroups, one row per lane group");
    constexpr bool RHTP = PRO == 3 || PRO == 5;
    constexpr bool STRIPE = PRO == 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    UZU_TL_DECL;
    UZU_TL_STAMP(0);
    const uint32_t K = p.k;
    const int lpr = 1 << lpr_log2, rpw = 64 >> lpr_log2;
    const int sl = lane & (lpr - 1), rsub = lane >> lpr_log2;
    const uint32_t C = K / 32;
    const uint32_t row_bytes = K * BITS / 8;
    const uint32_t G = (K + p.group_size - 1) / p.group_size;
    const uint32_t zp_stride = BITS == 4 ? (G + 1) / 2 : G;
    const uint32_t rows_per_batch = R * rpw;
    const uint32_t n_log0 = ACT ? p.n[0] / 2 : p.n[0];
    const uint32_t batches0 = (n_log0 + rows_per_batch - 1) / rows_per_batch;
    const uint32_t batches1 = ACT ? 0 : (p.n[1] + rows_per_batch - 1) / rows_per_batch;
    const uint32_t num_batches = batches0 + batches1;
    // wide workgroups: `bpw` batches per workgroup and round -- fewer than waves when the whole matrix is less than one round of the
    // resident waves, so that every CU streams (2560 two-row batches: 256 workgroups x 10 busy waves instead of 160 x 16)
    const uint32_t bpw = (NW > 4 && p.wg_batches) ? p.wg_batches : NW;
    const uint32_t total_waves = gridDim.x * bpw;
    const uint32_t steps_per_lane = CPLT == 0 ? (C + lpr - 1) / lpr : CPL;
    const uint32_t gshift = 31 - __builtin_clz(p.group_size); // group_size is a power of two (checked at launch)

    struct Item {
        Codes w[R][NPHYS];
        uint16_t s[R][NPHYS], o[R][NPHYS];
    };
    // Loads of one (batch, step) item.  Batches [0, batches0) belong to matrix 0, the rest to matrix 1 (wave-uniform:
    // the base pointers stay in SGPRs and the per-lane part of every address is a 32-bit byte offset -- host-checked).
    // STRIPE: batches per 32-row block (a multiple of the NW waves: rows_per_batch = rpw is 1, 2 or 4 -- host-checked), blocks of the matrix; workgroup w owns
    // blocks w, w + grid, ...: wave v takes batches v, v + NW, ... of a block
    const uint32_t nbs = STRIPE ? 32u / rows_per_batch : 1u;
    const uint32_t b_own = STRIPE ? min(blockIdx.x * nbs + (uint32_t)wave, num_batches - 1) : min((uint32_t)(blockIdx.x * bpw + min((uint32_t)wave, bpw - 1)), num_batches - 1);
    auto load_item = [&](uint32_t b, uint32_t j, Item& it) {
        // Unconditional: batch and step are clamped into range and a clamped reload is never consumed.  VMEM returns in
        // issue order and the compiler only emits counted waits (vmcnt(N)) across loads that are always issued; even a
        // wave-uniform branch around them costs the normed kernels 0.4 us and the readout 15 % (tools/kbench A/B).
        const uint32_t c_raw = sl + lpr * j;
        const uint32_t c = c_raw < C ? c_raw : C - 1;
        b = b < num_batches ? b : b_own; // past the end: re-read the wave's first batch (cache hit, spread over channels)
        const int mat = __builtin_amdgcn_readfirstlane(b >= batches0 ? 1 : 0);
        const uint32_t lb = mat ? b - batches0 : b;
        const uint32_t nl = mat ? p.n[1] : n_log0;
        const uint8_t* wp = p.w[mat];
        const uint16_t* sp = p.scales[mat];
        const uint16_t* bp = p.biases[mat];
        const uint8_t* zp = p.zp[mat];
        const uint32_t grp = (c * 32) >> gshift;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t lr = lb * rows_per_batch + r * rpw + rsub;
            if (lr >= nl) lr = 0;
#pragma unroll
            for (int h = 0; h < NPHYS; ++h) {
                const uint32_t prow = ACT ? lr + (h ? p.n[0] / 2 : 0) : lr;
                load_codes_stream(it.w[r][h], wp + (prow * row_bytes + c * STEP_BYTES));
                const uint32_t gi = prow * G + grp;
                it.s[r][h] = sp[gi];
                if (KIND == UZU_MATMUL_B_SCALE_BIAS) it.o[r][h] = bp[gi];
                else if (KIND == UZU_MATMUL_B_SCALE_ZERO_POINT) it.o[r][h] = BITS == 4 ? zp[prow * zp_stride + (grp >> 1)] : zp[prow * zp_stride + grp];
            }
        }
    };

    struct ConvPre { // taps / weights of the conv channels this lane finishes (kernel size 4: the common case)
        float tap[3];
        f32x4_v w;
        float bias;
    };
    auto conv_prefetch = [&](uint32_t b, ConvPre (&cp)[CONV ? R : 1]) { // issued one batch ahead: lands while the rows are computed
        // every lane of the row loads (same address: one request), rows outside the conv block read channel 0 and are
        // not consumed -- unconditional for the same reason as load_item
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t lrow_raw = b * rows_per_batch + r * rpw + rsub;
            const uint32_t lrow = (b < batches0 && lrow_raw < p.conv_dim && lrow_raw < n_log0) ? lrow_raw : 0;
            ConvPre& c4 = cp[CONV ? r : 0];
            const float* st_row = p.conv_state + lrow * 3;
            c4.tap[0] = st_row[0], c4.tap[1] = st_row[1], c4.tap[2] = st_row[2];
            c4.w = *(const f32x4_v*)(p.conv_w + lrow * 4);
            const float bias = (p.conv_b ? p.conv_b : p.conv_w)[lrow];
            c4.bias = p.conv_b ? bias : 0.0f;
        }
    };
    const uint32_t b0 = STRIPE ? blockIdx.x * nbs + (uint32_t)wave : ((uint32_t)wave < bpw ? blockIdx.x * bpw + wave : num_batches);
    // Wide workgroups hand their batches out dynamically.  On a CU the oldest waves win the issue arbitration: with a static
    // assignment the first-dispatched quarter of a 1024-workgroup grid finishes after 10.4 us, the last after 17.2 us
    // (Llama-3-8B up-projection, tools/timeline.py --detail; no difference between XCDs), and the CU idles towards the end with
    // ever fewer waves.  The workgroup's batches -- slot s of round r is batch blockIdx * NW + s + r * total_waves, so that its
    // waves keep streaming neighbouring rows -- are drawn from an LDS counter (round 0 is the static b0): every wave stays busy
    // until the workgroup's share is done.  Which wave computes a row does not change the row's arithmetic.
    __shared__ uint32_t s_next_slot;
    if (NW > 4 && tid == 0) s_next_slot = bpw; // published by the prologue's barrier, first drawn after it
    auto next_batch = [&](uint32_t b) -> uint32_t {
        if constexpr (STRIPE) {
            const uint32_t pos = b % nbs;
            return pos + NW < nbs ? b + NW : b - pos + gridDim.x * nbs + (uint32_t)wave; // the next batch of this block, or this wave's first of the workgroup's next block
        } else if constexpr (NW > 4) {
            uint32_t n = 0;
            if (lane == 0) n = atomicAdd(&s_next_slot, 1u);
            n = __builtin_amdgcn_readfirstlane(n);
            return blockIdx.x * bpw + n % bpw + (n / bpw) * total_waves;
        } else {
            return b + total_waves;
        }
    };
    Item itA, itB;
    // VMEM loads return in issue order.  The activation row (written by the previous kernel: L2 / memory-side cache,
    // ~0.8 us) is needed first and the weights (HBM, ~1.3 us) only after the prologue, so the activation loads are
    // issued FIRST and the first weight item right behind them; with LDS-only barriers in the prologue nothing waits
    // for the weights before the row loop.
    // staged 4-element vectors per thread: K / 1024 <= 2 CPL always (K = 32 lpr CPL, lpr <= 64).  ALL of them are requested
    // ahead of the weights: a vector fetched in place inside the prologue queues behind the first weight batch of every
    // resident workgroup (VMEM returns in order) -- 5-7 us of prologue at K = 4096 (tools/timeline.py, round 2)
    constexpr int NPRE = 2 * CPL;
    u32x2_v x_pre[NPRE], s_pre[NPRE];
    f32x4_v n_pre[NPRE];
    const bool pro_wave = NW == 4 || wave < 4; // the prologue is a 256-thread affair (wave-uniform)
    if ((PRO == 1 || RHTP) && pro_wave) {
        const uint32_t E = K / 256;
#pragma unroll
        for (int qi = 0; qi < NPRE; ++qi) {
            const uint32_t q = (uint32_t)qi * 4 < E ? (uint32_t)qi * 4 : E - 4; // clamped: never consumed past E
            const uint32_t e = tid * E + q;
            x_pre[qi] = *(const u32x2_v*)(p.x + e);
            s_pre[qi] = *(const u32x2_v*)((p.residual_add ? p.shortcut_in : p.x) + e);
            n_pre[qi] = *(const f32x4_v*)(p.norm_scales ? p.norm_scales + e : (const float*)p.x);
        }
    }
    uint32_t x_bits = 0, in_bits = 0; // PRO == 3: the sign words of the stripe this thread transforms (unconditional loads: an absent table reads the row)
    uint32_t x_bits_s = 0, in_bits_s = 0; // ... and of the stripe this thread's OWN elements [tid E, tid E + E) lie in (the spread form below)
    if constexpr (RHTP) {
        const uint32_t st = min((uint32_t)tid, C - 1);
        x_bits = (p.x_rht_bits ? p.x_rht_bits : (const uint32_t*)p.x)[st];
        in_bits = (p.in_rht_bits ? p.in_rht_bits : (const uint32_t*)p.x)[st];
        const uint32_t sw = min((uint32_t)tid * (K / 256) / 32, C - 1);
        x_bits_s = (p.x_rht_bits ? p.x_rht_bits : (const uint32_t*)p.x)[sw];
        in_bits_s = (p.in_rht_bits ? p.in_rht_bits : (const uint32_t*)p.x)[sw];
    }
    if constexpr (PRO == 2) { // the norm-gate prologue of an RHT out-projection: its InputRht on the gated row (round 5)
        if (p.in_rht_bits) in_bits = p.in_rht_bits[min((uint32_t)tid, C - 1)];
    }
    f32x4_v dg_pre[6]; // norm-gate prologue: chunk 0 of this thread (o, z, w: two vectors each); further chunks load in place
    if (PRO == 2) {
        const uint32_t nchunks = K / 8, per = nchunks > 256 ? nchunks / 256 : 1;
        const uint32_t chunk = min((uint32_t)tid * per, nchunks - 1), e0 = chunk * 8;
        dg_pre[0] = *(const f32x4_v*)(p.dg_o + e0), dg_pre[1] = *(const f32x4_v*)(p.dg_o + e0 + 4);
        dg_pre[2] = *(const f32x4_v*)(p.dg_sz + e0), dg_pre[3] = *(const f32x4_v*)(p.dg_sz + e0 + 4);
        dg_pre[4] = *(const f32x4_v*)(p.dg_w + e0 % p.dg_dv), dg_pre[5] = *(const f32x4_v*)(p.dg_w + e0 % p.dg_dv + 4);
    }
    constexpr int XST = (BITS == 4 && CPLT == 0) ? (NT >= 1024 ? 1 : 1024 / NT) : 1; // steps of the LDS-resident row staged per thread (K <= 32768)
    gc_raw4 xrow_raw[XST][4];
    if (BITS == 4 && CPLT == 0) {
#pragma unroll
        for (int q = 0; q < XST; ++q) {
            const uint32_t c = min((uint32_t)tid + (uint32_t)NT * q, C - 1); // clamped: a re-read, never stored
            const gc_raw4* src = (const gc_raw4*)(p.x + (size_t)c * 32);
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) xrow_raw[q][w4] = src[w4];
        }
    }
    // exp() table of the SiLU / softplus epilogues (uzu_math.h): fetched with the first loads, parked in LDS before the
    // first barrier -- the epilogue's table read is then an LDS access instead of a dependent global-memory round trip
    __shared__ uint64_t s_exp_tab[(ACT || CONV) ? 32 : 1];
    uint64_t exp_entry = 0;
    if (ACT || CONV) exp_entry = kExp2fTab[tid & 31];
    __builtin_amdgcn_sched_barrier(0); // keep the issue order: the scheduler would hoist the weight loads above the staging
    load_item(b0, 0, itA); // in flight during the whole prologue
    // ... and so is the batch's second step where there is one (K > 2048, or the streaming path): issued after the prologue it is a
    // second exposed round trip for a wave that owns a single batch (latency regime: Qwen3.5-0.8B decode +1.1 %, same-box A/B of
    // two builds).  Not in the wide workgroups of the bandwidth regime: there the extra requests in front of the shared prologue
    // cost more than they hide (Llama-3-8B 572 -> 545 tok/s), and not with two rows per wave (its 8 MB out-projection: 5.2 ->
    // 5.5 us): one row per wave is what the plan gives the small matrices.
    constexpr bool PRELOAD2 = UZU_GEMV_PRELOAD2 && NW == 4 && R == 1 && (CPLT == 0 || CPL >= 2);
    if constexpr (PRELOAD2) load_item(b0, 1, itB);
    __builtin_amdgcn_sched_barrier(0);
    ConvPre cp_cur[CONV ? R : 1]; // conv operands of the wave's first batch: in flight during the prologue as well
    if (CONV) conv_prefetch(b0, cp_cur);
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue ---------------------------------------------------------------------------------------
    // int4, plain bf16 activation row: bf16 pairs go straight into packed-dot order and the step sums through the dot unit (as for
    // K > 8192) -- 32 VALU per step instead of 79 through f32 registers (Llama-3-8B out-projection 5.18 -> 4.94 us, same-box A/B).
    // (Measured and dropped: converting the NORMED row once per workgroup -- one thread per step packs it in LDS, the lanes fetch
    // 4 x b128 + the sum -- instead of once per wave: Llama-3-8B up-projection 17.5 -> 17.1 us, but the extra serial phase and
    // barrier cost the K = 1024 kernels, where a lane converts a single step, 0.1-0.5 us each: Qwen3.5-0.8B 1664 -> 1605 tok/s.)
    constexpr bool RAWX = UZU_GEMV_RAWX && BITS == 4 && CPLT != 0 && PRO == 0;
    float xf[RAWX ? 1 : CPL][32];
    float xsm[CPL];
    XPack xq[BITS == 4 ? CPL : 1];
    if (CPLT != 0) {
        if (PRO == 2) {
            // DeltaNet norm-gate (tail of update.rs:30-143) as the out-proj prologue: thread t owns `per` chunks of 8
            // consecutive outputs; sum of squares per head in chunk8_sumsq order; x = bf16(o * inv_rms * w * SiLU(z)).
            extern __shared__ __attribute__((aligned(16))) float smem[];
            float* xs = smem;
            const uint32_t nchunks = K / 8, per = nchunks > 256 ? nchunks / 256 : 1;
            const uint32_t dv = p.dg_dv, lanes_per_head = (dv / 8) / per;
            float ov[4][8], zv[4][8];
            float local = 0.f;
            const bool active = (uint32_t)tid * per < nchunks;
            if (active) {
                float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    if (q >= per) break;
                    const uint32_t e0 = ((uint32_t)tid * per + q) * 8;
                    const f32x4_v a0 = q == 0 ? dg_pre[0] : *(const f32x4_v*)(p.dg_o + e0), a1 = q == 0 ? dg_pre[1] : *(const f32x4_v*)(p.dg_o + e0 + 4);
                    const f32x4_v z0 = q == 0 ? dg_pre[2] : *(const f32x4_v*)(p.dg_sz + e0), z1 = q == 0 ? dg_pre[3] : *(const f32x4_v*)(p.dg_sz + e0 + 4);
                    ov[q][0] = a0.x, ov[q][1] = a0.y, ov[q][2] = a0.z, ov[q][3] = a0.w, ov[q][4] = a1.x, ov[q][5] = a1.y, ov[q][6] = a1.z, ov[q][7] = a1.w;
                    zv[q][0] = z0.x, zv[q][1] = z0.y, zv[q][2] = z0.z, zv[q][3] = z0.w, zv[q][4] = z1.x, zv[q][5] = z1.y, zv[q][6] = z1.z, zv[q][7] = z1.w;
                    cs[q] = chunk8_sumsq(ov[q]);
                }
                local = per == 1 ? cs[0] : per == 2 ? cs[0] + cs[1] : (cs[0] + cs[1]) + (cs[2] + cs[3]);
            }
            const float sumsq = row_sum_rt(local, (int)lanes_per_head); // lanes of one head are consecutive and aligned
            const float inv_rms = 1.0f / sqrtf(sumsq / (float)dv + p.dg_eps);
            if (active) {
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    if (q >= per) break;
                    const uint32_t e0 = ((uint32_t)tid * per + q) * 8;
                    const f32x4_v w0 = q == 0 ? dg_pre[4] : *(const f32x4_v*)(p.dg_w + e0 % dv), w1 = q == 0 ? dg_pre[5] : *(const f32x4_v*)(p.dg_w + e0 % dv + 4);
                    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = round_bf16(ov[q][i] * inv_rms * wv[i] * zv[q][i]);
                    float* slot = xs + (size_t)(e0 / 32) * 36 + e0 % 32;
                    *(float4*)slot = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(slot + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
            if (p.in_rht_bits) { // RHTLinearWrapper (rht_wrapper.rs:215-298): this linear's InputRht on the gated row, stripe by stripe through the slots --
                lds_barrier();   // activation_transform INPUT_RHT's arithmetic on the bf16 values delta_net_update would have stored (bit-identical)
                if ((uint32_t)tid < C) rht_stripe<true>(xs + (size_t)tid * 36, in_bits, nullptr);
            }
            if ((ACT || CONV) && tid < 32) s_exp_tab[tid] = exp_entry; // rides on the barrier below
            lds_barrier();
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const uint32_t c = sl + lpr * j;
                if (c < C) {
                    const float4* xv = (const float4*)(xs + (size_t)c * 36);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 t = xv[i];
                        xf[RAWX ? 0 : j][4 * i] = t.x, xf[RAWX ? 0 : j][4 * i + 1] = t.y, xf[RAWX ? 0 : j][4 * i + 2] = t.z, xf[RAWX ? 0 : j][4 * i + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 32; ++i) xf[RAWX ? 0 : j][i] = 0.f;
                }
            }
        } else if (PRO == 0 || PRO == 4) { // plain activation row: every lane fetches its own steps
            if constexpr (RAWX) { // int4: bf16 pairs straight into packed-dot order, step sums through the dot unit (as for K > 8192)
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const uint32_t c = sl + lpr * j;
                    XPack& x = xq[BITS == 4 ? j : 0];
                    if (c < C) xsm[j] = xpack_load(x, p.x + (size_t)c * 32);
                    else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) x.v[i] = 0u;
                        xsm[j] = 0.f;
                    }
                }
            } else {
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const uint32_t c = sl + lpr * j;
                if (c < C) {
                    load32_bf16(p.x + (size_t)c * 32, xf[RAWX ? 0 : j]);
                    // PRO == 4: the InputRht of an RHT linear on its plain input row (attention rows in front of the out-projection): a lane's step IS
                    // one 32-element stripe, so the transform runs in its registers -- activation_transform INPUT_RHT's arithmetic (rht_stripe.h),
                    // redone by every lane group instead of a launch of its own in front of the GEMV (round 5)
                    if constexpr (PRO == 4) rht_stripe_regs<true>(xf[RAWX ? 0 : j], p.in_rht_bits[c]);
                } else {
#pragma unroll
                    for (int i = 0; i < 32; ++i) xf[RAWX ? 0 : j][i] = 0.f;
                }
            }
            }
        } else {
            // Normalization (normalization.rs:56-125) once per workgroup through LDS; element/thread mapping and
            // reduction order of normalization_kernel: thread t owns elements [t*E, t*E+E), E = K/256 (K % 1024 == 0).
            extern __shared__ __attribute__((aligned(16))) float smem[];
            float* xs = smem;                       // C slots of 36 floats (16-byte pad => conflict-free b128 reads)
            float* red = smem + (size_t)C * 36;     // [4]
            const uint32_t E = K / 256;
            float ss = 0.f;
            bool x_in_lds = false;
            // PRO == 3, spread form (rht_stripe.h::rht_spread): with E a power of two in [4, 32] a stripe is 32 / E neighbouring threads of the
            // prologue, and both transforms run on the elements where they already are -- in the owning threads' registers
            constexpr int EMAX = RHTP ? 8 * CPL : 1;
            float er[EMAX];
            const bool spread = RHTP && (E == 4 || E == 8 || E == 16 || E == 32) && E <= (uint32_t)EMAX;
            bool x_in_regs = false;
            auto spread_transform = [&](bool input, uint32_t bits) {
                const uint32_t fb = ((uint32_t)tid * E) % 32;
                if constexpr (RHTP) {
                    if (E == 4) input ? rht_spread<4, true>(*(float(*)[4])er, bits, fb, lane) : rht_spread<4, false>(*(float(*)[4])er, bits, fb, lane);
                    if constexpr (EMAX >= 8) {
                        if (E == 8) input ? rht_spread<8, true>(*(float(*)[8])er, bits, fb, lane) : rht_spread<8, false>(*(float(*)[8])er, bits, fb, lane);
                    }
                    if constexpr (EMAX >= 16) {
                        if (E == 16) input ? rht_spread<16, true>(*(float(*)[16])er, bits, fb, lane) : rht_spread<16, false>(*(float(*)[16])er, bits, fb, lane);
                    }
                    if constexpr (EMAX >= 32) {
                        if (E == 32) input ? rht_spread<32, true>(*(float(*)[32])er, bits, fb, lane) : rht_spread<32, false>(*(float(*)[32])er, bits, fb, lane);
                    }
                }
            };
            if constexpr (RHTP) {
                if (p.x_rht_bits && spread) { // the row is the raw output of an RHT linear: its OutputRht (+ bias) first
                    if (pro_wave) {
#pragma unroll
                        for (int qi = 0; qi < 2 * CPL; ++qi) {
                            const uint32_t q = (uint32_t)qi * 4;
                            if (q >= E) break;
                            const u32x2_v xr = qi < NPRE ? x_pre[qi < NPRE ? qi : 0] : *(const u32x2_v*)(p.x + tid * E + q);
                            er[4 * qi] = bits_to_f32(xr.x << 16), er[4 * qi + 1] = bits_to_f32(xr.x & 0xFFFF0000u);
                            er[4 * qi + 2] = bits_to_f32(xr.y << 16), er[4 * qi + 3] = bits_to_f32(xr.y & 0xFFFF0000u);
                        }
                        spread_transform(false, x_bits_s);
                        if (p.x_rht_bias) {
#pragma unroll
                            for (int qi = 0; qi < 2 * CPL; ++qi) {
                                const uint32_t q = (uint32_t)qi * 4;
                                if (q >= E) break;
                                const u32x2_v br = *(const u32x2_v*)(p.x_rht_bias + tid * E + q);
                                er[4 * qi] = round_bf16(er[4 * qi] + bits_to_f32(br.x << 16)), er[4 * qi + 1] = round_bf16(er[4 * qi + 1] + bits_to_f32(br.x & 0xFFFF0000u));
                                er[4 * qi + 2] = round_bf16(er[4 * qi + 2] + bits_to_f32(br.y << 16)), er[4 * qi + 3] = round_bf16(er[4 * qi + 3] + bits_to_f32(br.y & 0xFFFF0000u));
                            }
                        }
                    }
                    x_in_regs = true;
                } else if (p.x_rht_bits) { // (E not a power of two: stripe by stripe through the slots, one thread per stripe)
                    if (pro_wave) {
#pragma unroll
                        for (int qi = 0; qi < 2 * CPL; ++qi) {
                            const uint32_t q = (uint32_t)qi * 4;
                            if (q >= E) break;
                            const uint32_t e = tid * E + q;
                            const u32x2_v xr = qi < NPRE ? x_pre[qi < NPRE ? qi : 0] : *(const u32x2_v*)(p.x + e);
                            *(float4*)(xs + (size_t)(e / 32) * 36 + e % 32) =
                                make_float4(bits_to_f32(xr.x << 16), bits_to_f32(xr.x & 0xFFFF0000u), bits_to_f32(xr.y << 16), bits_to_f32(xr.y & 0xFFFF0000u));
                        }
                    }
                    lds_barrier();
                    if ((uint32_t)tid < C) rht_stripe<false>(xs + (size_t)tid * 36, x_bits, p.x_rht_bias ? p.x_rht_bias + (size_t)tid * 32 : nullptr);
                    lds_barrier();
                    x_in_lds = true;
                }
            }
            if (pro_wave) {
#pragma unroll
            for (int qi = 0; qi < 2 * CPL; ++qi) {
                const uint32_t q = (uint32_t)qi * 4;
                if (q >= E) break;
                const uint32_t e = tid * E + q;
                const u32x2_v xr = qi < NPRE ? x_pre[qi < NPRE ? qi : 0] : *(const u32x2_v*)(p.x + e);
                float v[4] = {bits_to_f32(xr.x << 16), bits_to_f32(xr.x & 0xFFFF0000u), bits_to_f32(xr.y << 16), bits_to_f32(xr.y & 0xFFFF0000u)};
                if (RHTP && x_in_lds) {
                    const float4 t = *(const float4*)(xs + (size_t)(e / 32) * 36 + e % 32);
                    v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
                }
                if constexpr (RHTP) {
                    if (x_in_regs) v[0] = er[4 * qi], v[1] = er[4 * qi + 1], v[2] = er[4 * qi + 2], v[3] = er[4 * qi + 3];
                }
                if (p.residual_add) {
                    const u32x2_v sr = qi < NPRE ? s_pre[qi < NPRE ? qi : 0] : *(const u32x2_v*)(p.shortcut_in + e);
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
            UZU_TL_STAMP(1); // the activation vector has arrived and gone through the residual add / sum of squares
            ss = wave_sum(ss);
            if (lane == 0) red[wave] = ss;
            }
            lds_barrier();
            const float total = ((red[0] + red[1]) + red[2]) + red[3];
            const float variance = total / (float)K - 0.0f * 0.0f;
            const float rms_inv = 1.0f / sqrtf(variance + p.norm_eps);
            if (pro_wave) {
#pragma unroll
            for (int qi = 0; qi < 2 * CPL; ++qi) {
                const uint32_t q = (uint32_t)qi * 4;
                if (q >= E) break;
                const uint32_t e = tid * E + q;
                float* slot = xs + (size_t)(e / 32) * 36 + e % 32;
                const float4 vv = *(const float4*)slot; // own elements: no barrier needed
                float v[4] = {vv.x, vv.y, vv.z, vv.w};
                const f32x4_v t4 = qi < NPRE ? n_pre[qi < NPRE ? qi : 0] : *(const f32x4_v*)(p.norm_scales ? p.norm_scales + e : (const float*)p.x);
                const float scl[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float normalized = (v[i] - 0.0f) * rms_inv;
                    if (!p.norm_scales) v[i] = round_bf16(normalized);
                    else if (p.norm_full_layer) v[i] = round_bf16(normalized * (scl[i] + p.norm_offset));
                    else v[i] = round_bf16(round_bf16(normalized) * round_bf16(scl[i] + p.norm_offset));
                }
                if constexpr (RHTP) {
                    if (spread && p.in_rht_bits) er[4 * qi] = v[0], er[4 * qi + 1] = v[1], er[4 * qi + 2] = v[2], er[4 * qi + 3] = v[3]; // (transformed below, then stored)
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
            if constexpr (RHTP) {
                if (p.in_rht_bits && spread) { // this linear's InputRht on the normalised row, in the owning threads' registers
                    if (pro_wave) {
                        spread_transform(true, in_bits_s);
#pragma unroll
                        for (int qi = 0; qi < 2 * CPL; ++qi) {
                            const uint32_t q = (uint32_t)qi * 4;
                            if (q >= E) break;
                            const uint32_t e = tid * E + q;
                            *(float4*)(xs + (size_t)(e / 32) * 36 + e % 32) = make_float4(er[4 * qi], er[4 * qi + 1], er[4 * qi + 2], er[4 * qi + 3]);
                        }
                    }
                } else if (p.in_rht_bits) { // this linear's InputRht on the normalised row
                    lds_barrier();
                    if ((uint32_t)tid < C) rht_stripe<true>(xs + (size_t)tid * 36, in_bits, nullptr);
                }
            }
            if ((ACT || CONV) && tid < 32) s_exp_tab[tid] = exp_entry; // rides on the barrier below (requested with the first loads)
            lds_barrier();
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const uint32_t c = sl + lpr * j;
                if (c < C) {
                    const float4* xv = (const float4*)(xs + (size_t)c * 36);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 t = xv[i];
                        xf[RAWX ? 0 : j][4 * i] = t.x, xf[RAWX ? 0 : j][4 * i + 1] = t.y, xf[RAWX ? 0 : j][4 * i + 2] = t.z, xf[RAWX ? 0 : j][4 * i + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 32; ++i) xf[RAWX ? 0 : j][i] = 0.f;
                }
            }
        }
        if constexpr (!RAWX) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) xsm[j] = sum32(xf[j]);
        }
    }
    // K > 8192 (CPLT == 0), int4: the whole activation row is parked once per workgroup in LDS, already in packed-dot order
    // (step c = 16 words at a stride of 20 words: conflict-free ds_read_b128 for 16 consecutive steps) together with the
    // per-step sums.  The per-step re-read from global memory it replaces cost four more VMEM instructions, 16 v_perm and 16
    // dot2 per 16-byte weight load (Llama-3-8B down-projection: 2.9 TB/s).  Its loads were issued ahead of the weights.
    if (BITS == 4 && CPLT == 0) {
        extern __shared__ __attribute__((aligned(16))) float smem[];
        uint32_t* xw = (uint32_t*)smem;
        float* xsum = smem + (size_t)C * 20;
#pragma unroll
        for (int q = 0; q < XST; ++q) {
            const uint32_t c = (uint32_t)tid + (uint32_t)NT * q;
            if (c < C) {
                XPack xp;
                const float sx = xpack_from_raw(xp, xrow_raw[q]);
                u32x4_v* dst = (u32x4_v*)(xw + (size_t)c * 20);
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    u32x4_v v;
                    v.x = xp.v[4 * w4], v.y = xp.v[4 * w4 + 1], v.z = xp.v[4 * w4 + 2], v.w = xp.v[4 * w4 + 3];
                    dst[w4] = v;
                }
                xsum[c] = sx;
            }
        }
        lds_barrier();
    }
    // int4: the row goes through the packed-dot unit (gemv_core.h: dot32p) -- bf16 pairs, 16 registers per step; the
    // f32 copy is dead from here on
    if constexpr (BITS == 4 && CPLT != 0 && !RAWX) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) xpack_from_f32(xq[BITS == 4 ? j : 0], xf[j]);
    }

    if ((ACT || CONV) && (PRO == 0 || PRO == 4 || CPLT == 0)) { // no prologue barrier to ride on
        if (tid < 32) s_exp_tab[tid] = exp_entry;
        lds_barrier();
    }
    UZU_TL_STAMP(2);
    // ---- row loop, software pipelined over (batch, step) items ---------------------------------------------
    // Two item buffers alternate roles (no register copies): while one is consumed the loads of the next (batch,
    // step) land in the other.
    float best_v = -INFINITY; // arg-max epilogue state (lane-local)
    uint32_t best_i = 0xFFFFFFFFu;
    auto compute = [&](const Item& it, uint32_t c, const auto& x, float xs, float (&acc)[R][NPHYS]) {
        const uint32_t grp = (c * 32) >> gshift;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int h = 0; h < NPHYS; ++h) {
                const float sc = bf16_to_f32(it.s[r][h]);
                float of;
                if (KIND == UZU_MATMUL_B_SCALE_BIAS) of = bf16_to_f32(it.o[r][h]);
                else if (KIND == UZU_MATMUL_B_SCALE_ZERO_POINT) {
                    const uint32_t zpv = BITS == 4 ? ((grp & 1) ? (it.o[r][h] >> 4) : (it.o[r][h] & 0x0F)) : it.o[r][h];
                    of = -sc * (float)zpv;
                } else of = -sc * (float)(1u << (BITS - 1));
                float dq;
                if constexpr (BITS == 4) {
                    dq = dot32p(it.w[r][h], x);       // sum (16 + q) x
                    of = fmaf(-kQ4Offset, sc, of);     // exact: both bf16-derived, exponents a few bits apart
                } else {
                    dq = dot32(it.w[r][h], x);
                }
                acc[r][h] = fmaf(sc, dq, fmaf(of, xs, acc[r][h]));
                // one row at a time: without the fence the scheduler interleaves the R dot products and keeps
                // R x 32 converted codes live (350+ VGPRs at R = 4 => one wave per SIMD)
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto stream_step = [&](const Item& it, uint32_t c, float (&acc)[R][NPHYS]) { // K > 8192: the row is re-read per step (L1 / L2)
        if constexpr (BITS == 4) {
            extern __shared__ __attribute__((aligned(16))) float smem[];
            const u32x4_v* src = (const u32x4_v*)((const uint32_t*)smem + (size_t)c * 20);
            XPack x;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4_v v = src[q];
                x.v[4 * q] = v.x, x.v[4 * q + 1] = v.y, x.v[4 * q + 2] = v.z, x.v[4 * q + 3] = v.w;
            }
            compute(it, c, x, (smem + (size_t)C * 20)[c], acc);
        } else {
            float x[32];
            load32_bf16(p.x + (size_t)c * 32, x);
            compute(it, c, x, sum32(x), acc);
        }
    };
    // GatedActMul epilogue, batched: a finished (up, gate) pair parks in the registers of lane `act_cnt` and the activation (a glibc-exact
    // expf in double precision: ~150 instructions) runs once per 64 pairs on all lanes instead of once per pair on one lane with 63 idle
    // -- the VALU is what bounds these kernels once the bytes are on chip (tools/valu_rate.hip).  Same arithmetic per value.
    float keep_up = 0.f, keep_gate = 0.f;
    uint32_t keep_row = 0xFFFFFFFFu, act_cnt = 0;
    auto act_flush = [&]() {
        if constexpr (ACT) {
            if (keep_row != 0xFFFFFFFFu) p.out[0][keep_row] = f32_to_bf16(round_bf16(keep_up * act_bf16(p.act_type, keep_gate, s_exp_tab))); // gated_act_mul/mod.rs:5-12
            keep_row = 0xFFFFFFFFu, act_cnt = 0;
        }
    };
    __shared__ float s_stripe[STRIPE ? 2 : 1][STRIPE ? 32 : 1];
    // One complete 32-row block: wave 0 runs rht_mlp_join_kernel's (ACT) / rht_out_rows_kernel's arithmetic on it -- hadamard32 across the lanes of a
    // half-wave, the sign factors, the roundings, the bias, then GatedActMul + the next linear's InputRht, or the DeltaNet conv of a channel --
    // and stores the finished rows.  ACT: lanes 0-31 carry the block's up rows, lanes 32-63 its gate rows.
    auto stripe_epilogue = [&](uint32_t st) {
        if constexpr (STRIPE) {
            lds_barrier();
            if (wave == 0) {
                const int l = lane & 31, half = lane >> 5;
                const uint32_t stripes = n_log0 / 32, row = st * 32 + (uint32_t)l;
                if constexpr (ACT) {
                    float v = hadamard32(s_stripe[half][l], l);
                    const uint32_t word = p.ep_out_bits[half ? stripes + st : st];
                    v = round_bf16((word >> l) & 1u ? -v : v);
                    if (p.ep_bias) v = round_bf16(v + bf16_to_f32(p.ep_bias[(half ? n_log0 : 0u) + row]));
                    const float gate = __shfl(v, l + 32, 64);
                    float r_ = round_bf16(v * act_bf16(p.act_type, gate, s_exp_tab)); // gated_act_mul/mod.rs:5-12 (lanes 32-63 compute on, unused)
                    if (p.ep_next_in_bits) {
                        const uint32_t iw = p.ep_next_in_bits[st];
                        r_ = round_bf16(hadamard32((iw >> l) & 1u ? -r_ : r_, l));
                    }
                    if (half == 0) p.out[0][row] = f32_to_bf16(r_);
                } else {
                    float v = hadamard32(s_stripe[0][l], l);
                    const uint32_t word = p.ep_out_bits[st];
                    v = round_bf16((word >> l) & 1u ? -v : v);
                    if (p.ep_bias) v = round_bf16(v + bf16_to_f32(p.ep_bias[row]));
                    if (half == 0) {
                        if (p.conv_w && row < p.conv_dim) { // DeltaNetConvUpdate of this channel (conv_update.rs:17-55), as rht_out_rows_kernel does it
                            const uint32_t tap_count = p.conv_ks - 1;
                            float* st_row = p.conv_state + (size_t)row * tap_count;
                            const float* w = p.conv_w + (size_t)row * p.conv_ks;
                            float cacc = p.conv_b ? p.conv_b[row] : 0.0f;
                            for (uint32_t tap = 0; tap < tap_count; ++tap) cacc += st_row[tap] * w[tap];
                            cacc += v * w[tap_count];
                            for (uint32_t tap = 1; tap < tap_count; ++tap) st_row[tap - 1] = st_row[tap];
                            st_row[tap_count - 1] = v;
                            v = silu_f32(cacc);
                        }
                        p.out[0][row] = f32_to_bf16(v);
                    }
                }
            }
            lds_barrier(); // the block's slots are free again
        }
    };
    auto finish = [&](uint32_t b, float (&acc)[R][NPHYS], const ConvPre (&cp)[CONV ? R : 1]) {
        const int mat = __builtin_amdgcn_readfirstlane(b >= batches0 ? 1 : 0);
        const uint32_t lb = mat ? b - batches0 : b;
        const uint32_t nl = mat ? p.n[1] : n_log0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float v0 = row_sum_rt(acc[r][0], lpr);
            const float v1 = ACT ? row_sum_rt(acc[r][NPHYS - 1], lpr) : 0.f;
            const uint32_t lrow = lb * rows_per_batch + r * rpw + rsub;
            if constexpr (STRIPE) {
                // the raw rows of the workgroup's current 32-row block, rounded to bf16 as the separate launch would have stored them (the linear's
                // own bias comes behind its OutputRht: bias_after_rht) -- stripe_epilogue takes over when the block is complete
                if (sl == 0) {
                    s_stripe[0][lrow & 31u] = round_bf16(1.0f * v0);
                    if (ACT) s_stripe[1][lrow & 31u] = round_bf16(1.0f * v1);
                }
            } else if constexpr (ACT) {
                // every lane of a row's group holds the row sums; MatmulKernel epilogue with ab_scale = 1 (kernel.rs:281-292), rounded to bf16
                float value = 1.0f * v0, gate = 1.0f * v1;
                if (p.out_bias[0] && lrow < nl) value += bf16_to_f32(p.out_bias[0][lrow]), gate += bf16_to_f32(p.out_bias[0][lrow + p.n[0] / 2]);
                const float up_b = round_bf16(value), gate_b = round_bf16(gate);
                for (int gsub = 0; gsub < rpw; ++gsub) { // the rpw rows of this pass go to lanes act_cnt .. act_cnt + rpw - 1
                    const float u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, up_b), gsub << lpr_log2));
                    const float g2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gate_b), gsub << lpr_log2));
                    const uint32_t row_g = lb * rows_per_batch + r * rpw + (uint32_t)gsub;
                    if ((uint32_t)lane == act_cnt + (uint32_t)gsub && row_g < nl) keep_up = u, keep_gate = g2, keep_row = row_g;
                }
                act_cnt += (uint32_t)rpw;
                if (act_cnt + (uint32_t)rpw > 64u) act_flush();
            } else if (sl == 0 && lrow < nl) {
                // MatmulKernel epilogue with ab_scale = 1, no accumulate / soft-cap (kernel.rs:281-292)
                float value = 1.0f * v0;
                if (p.out_bias[mat]) value += bf16_to_f32(p.out_bias[mat][lrow]);
                if (CONV && mat == 0 && lrow < p.conv_dim) {
                    // DeltaNetConvUpdate (conv_update.rs:17-55), kernel size 4, operands prefetched at batch start
                    float* st_row = p.conv_state + (size_t)lrow * 3;
                    const float xin = round_bf16(value);
                    const ConvPre& c4 = cp[CONV ? r : 0];
                    float cacc = c4.bias;
                    cacc += c4.tap[0] * c4.w.x;
                    cacc += c4.tap[1] * c4.w.y;
                    cacc += c4.tap[2] * c4.w.z;
                    cacc += xin * c4.w.w;
                    p.out[0][lrow] = f32_to_bf16(silu_f32_tab(cacc, s_exp_tab));
                    st_row[0] = c4.tap[1], st_row[1] = c4.tap[2], st_row[2] = xin;
                } else {
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
    // one batch whose first item sits in `first`; returns the wave's next batch, whose first item then sits in `first` (even
    // number of steps) or in `second` (odd number of steps)
    auto batch = [&](uint32_t b, Item& first, Item& second, auto have_second) -> uint32_t { // have_second: step 1 is loaded already
        constexpr bool HAVE2 = decltype(have_second)::value;
        const uint32_t bn = next_batch(b);
        float acc[R][NPHYS];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int h = 0; h < NPHYS; ++h) acc[r][h] = 0.f;
        if (CPLT != 0) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                Item& cur = (j & 1) ? second : first;
                Item& nxt = (j & 1) ? first : second;
                if (j + 1 < CPL) {
                    if (!(HAVE2 && j == 0)) load_item(b, j + 1, nxt);
                } else {
                    load_item(bn, 0, nxt);
                }
                const uint32_t c = sl + lpr * j;
                if (c < C) {
                    if constexpr (BITS == 4) compute(cur, c, xq[j], xsm[j], acc);
                    else compute(cur, c, xf[j], xsm[j], acc);
                }
            }
        } else {
            for (uint32_t j = 0; j < steps_per_lane; j += 2) {
                {
                    const bool last = j + 1 == steps_per_lane;
                    if (!(HAVE2 && j == 0)) load_item(last ? bn : b, last ? 0 : j + 1, second); // (the streaming path has >= 3 steps)
                    const uint32_t c = sl + lpr * j;
                    if (c < C) stream_step(first, c, acc);
                    if (last) { // odd step count: hand the prefetched item over (one copy per batch)
                        first = second;
                        break;
                    }
                }
                {
                    const bool last = j + 2 == steps_per_lane;
                    load_item(last ? bn : b, last ? 0 : j + 2, first);
                    const uint32_t c = sl + lpr * (j + 1);
                    if (c < C) stream_step(second, c, acc);
                }
            }
        }
        UZU_TL_STAMP(5); // dot products of the (last) batch done
        finish(b, acc, cp_cur);
        if constexpr (STRIPE) {
            if (b % nbs + NW >= nbs) stripe_epilogue(b / nbs); // this wave's last batch of the block: all waves meet here
        }
        UZU_TL_STAMP(6);
        if (CONV) conv_prefetch(bn, cp_cur); // operands of this wave's next batch
        return bn;
    };
    if (CPLT == 0 || (CPL & 1) == 0) {
        uint32_t b = b0;
        if (PRELOAD2 && b < num_batches) b = batch(b, itA, itB, std::true_type{});
        while (b < num_batches) b = batch(b, itA, itB, std::false_type{});
    } else {
        for (uint32_t b = b0; b < num_batches;) {
            b = batch(b, itA, itB, std::false_type{});
            if (b < num_batches) b = batch(b, itB, itA, std::false_type{});
        }
    }
    act_flush();
    UZU_TL_STAMP(3);
    if (p.part_val) { // UnifiedSampling (greedy) pass 1: one (value, index) partial per workgroup
        __shared__ float sv[NW];
        __shared__ uint32_t si[NW];
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best_v, off, 64);
            const uint32_t oi = __shfl_xor(best_i, off, 64);
            if (ov > best_v || (ov == best_v && oi < best_i)) best_v = ov, best_i = oi;
        }
        if (lane == 0) sv[wave] = best_v, si[wave] = best_i;
        __syncthreads();
        if (tid == 0) {
            for (int w2 = 1; w2 < NW; ++w2)
                if (sv[w2] > best_v || (sv[w2] == best_v && si[w2] < best_i)) best_v = sv[w2], best_i = si[w2];
            p.part_val[blockIdx.x] = best_v;
            p.part_idx[blockIdx.x] = best_i;
        }
    }
    UZU_TL_STAMP(4);
    UZU_TL_FLUSH(p);
}

// Launch geometry.  Two regimes (tools/kbench.cpp sweeps):
//   * small matrices (a few MB, latency-bound): as many waves as there are row batches, R chosen so that every
//     SIMD of the chip gets a wave;
//   * big matrices (>= 16 MB, bandwidth-bound): a PERSISTENT grid of exactly the resident workgroups
//     (occupancy x CUs, from the instance's register count) so that the prologue is paid once per resident
//     workgroup, and R = 1 (R = 2 for the 2-step register path): more waves beat more rows per wave.
static uint32_t gemv_dec_plan(const DecGemvParams& p, int num_cus, int* lpr_log2_out, int* R_out, bool* wide_out) {
    const int lpr_log2 = gemv_lpr_log2(p.k);
    const int rpw = 64 >> lpr_log2;
    const uint32_t n_log0 = p.act_mul ? p.n[0] / 2 : p.n[0];
    static int force_r = -1, tw = -1;
    if (force_r < 0) {
        const char* e = lab_env("UZU_DEC_R");
        force_r = e ? atoi(e) : 0;
        const char* t = lab_env("UZU_DEC_TW");
        tw = t ? atoi(t) : 0; // 0 = the rule below
    }
    const uint32_t C = p.k / 32, lpr = 1u << lpr_log2, cpl = (C + lpr - 1) / lpr;
    const uint64_t weight_bytes = ((uint64_t)p.n[0] + p.n[1]) * p.k * p.bits / 8;
    int R;
    auto nb = [&](int rr) { return (n_log0 + (uint32_t)(rr * rpw) - 1) / (uint32_t)(rr * rpw) + (p.n[1] + (uint32_t)(rr * rpw) - 1) / (uint32_t)(rr * rpw); };
    static const int wide_on = [] { // UZU_DEC_WIDE=0: 4-wave workgroups everywhere; 2: every bandwidth-regime kernel (A/B runs)
        const char* e = lab_env("UZU_DEC_WIDE");
        return e ? atoi(e) : 1;
    }();
    static const uint64_t big_bytes = [] { // UZU_DEC_BIG_MB: where the bandwidth regime (persistent grid) starts
        const char* e = lab_env("UZU_DEC_BIG_MB");
        return (uint64_t)(e && atoi(e) > 0 ? atoi(e) : 16) << 20;
    }();
    *wide_out = false;
    // a normalisation prologue at K >= 4096 is worth sharing from ~10 MB on (Llama-3-8B qkv, 12.6 MB: 9.4 -> 8.1 us); a plain
    // 13 MB out-projection is not (Qwen3-14B-class: 7.5 -> 8.2 us)
    const bool normed_mid = (p.norm_scales || p.norm_plain) && p.bits == 4 && cpl >= 2 && weight_bytes >= (10u << 20) && big_bytes == (16u << 20);
    if (weight_bytes >= big_bytes || normed_mid) {
        // int8 rows (4-wave workgroups, below) take two rows per lane group: 32 bytes of codes per lane and step leave a wave with 4 KB
        // in flight (Llama-3-8B int8, same-box sweep of UZU_DEC_R: down 19.7 -> 16.8 us, qkv 12.0 -> 11.7, read-out 101.5 -> 98.8;
        // the fused up / gate kernel already streams two rows: 26.4 -> 28.5 with four)
        R = ((cpl == 2 || p.bits == 8) && !p.act_mul) ? 2 : 1;
        // measured (Llama-3-8B, Qwen3-14B-class, same box A/B, tools/ab_decode_env.sh): int4 kernels with K >= 4096 5-25 % faster
        // (up 22.3 -> 20.2 us, down 13.1 -> 11.1, read-out 62 -> 55, 14B read-out 113 -> 100); K = 1024 (Qwen3.5 read-out: dozens of
        // batches per wave, a 2 KB activation row) 5 % slower; int8 10 % slower at 16 waves (128-register cap) and 1-2 % slower at 8
        // waves (up 27.0 -> 26.3 us but qkv 11.8 -> 13.5): int8 stays on 4-wave workgroups
        *wide_out = wide_on && force_r <= 0 && (wide_on == 2 || (p.bits == 4 && cpl >= 2));
        static const int wide_r = [] { // UZU_DEC_WIDE_R: rows per lane group of the wide workgroups (A/B runs; 0 = the rule above)
            const char* e = lab_env("UZU_DEC_WIDE_R");
            return e ? atoi(e) : 0;
        }();
        if (*wide_out && wide_r > 0 && !(p.act_mul && cpl > 2)) R = wide_r > 2 ? 2 : wide_r;
        else if (*wide_out && R == 1 && !p.act_mul) { // (the fused up / gate rows already are two rows in flight per lane group: Llama-3-8B's up-projection, 56 row pairs per 16 waves = 4 rounds either way, 17.4 -> 19.2 us with two pairs)
            // Rows per lane group by the round count.  A wave streams at a latency-bound rate (two items in flight), so a workgroup
            // of NW waves with B batches takes ceil(B / NW) rounds, the last one with whatever is left; two rows per lane group halve
            // the batches and keep twice the bytes in flight per wave.  Taken when that does not add row-rounds: Qwen3-14B-class
            // down-projection (20 rows per workgroup of 16 waves: 16 + 4 -> 10 x 2) 17.0 -> 14.5 us, its read-out (594 rows per
            // 12 waves) 100 -> 89; not Llama-3-8B's down-projection (16 rows = one full round, 8.5 -> 11.7 us with half the waves
            // idle) nor the 14B-class qkv (28 rows per 12 waves: 3 rounds against 2 x 2, 13.1 -> 14.2) -- same-box A/B, tools/ab_wide_r.sh.
            const uint32_t nw = (cpl > 2 && (p.norm_scales || p.norm_plain)) ? 12u : 16u; // launch_gemv_dec_c: NWV
            const uint32_t per_wg = (nb(1) + (uint32_t)num_cus - 1) / (uint32_t)num_cus;
            const uint32_t rounds1 = (per_wg + nw - 1) / nw, rounds2 = (per_wg + 2 * nw - 1) / (2 * nw) * 2;
            if (rounds2 <= rounds1) R = 2;
        }
    } else {
        R = p.act_mul ? 2 : 4;
        // Waves per CU a small matrix is cut into before rows per lane group are given up: 8, and 16 where a lane owns ONE step of the row (K <= 2048).  Round 4:
        // the one kernel of the headline model this changes is the DeltaNet in-projection (2056 two-row batches sat just above 8 x 256): one row per lane
        // group, 1028 workgroups, 1672 -> 1689-1692 tok/s in three same-box A/B pairs, token streams identical (which wave computes a row does not change the
        // row's arithmetic).  Not at K = 4096: Llama-3-8B's 8 MB out-projection with one row per lane group instead of two 160 -> 170 us per step (572 -> 566 tok/s).
        const uint32_t target_waves = (uint32_t)num_cus * (uint32_t)(tw > 0 ? tw : (cpl == 1 ? 16 : 8));
        while (R > 1 && nb(R) < target_waves) R >>= 1;
        if (cpl > 2 && R > 2 && (p.norm_scales || p.norm_plain)) R = 2; // the 4-step register path is instantiated for R <= 2
    }
    if (force_r > 0) R = p.act_mul && force_r > 2 ? 2 : force_r;
    *lpr_log2_out = lpr_log2;
    *R_out = R;
    return (nb(R) + 3) / 4;
}

template <int BITS, int CPLT, bool ACT, int KIND, int PRO, bool CONV>
static uzu_status launch_gemv_dec_c(hipStream_t s, const DecGemvParams& p, uint32_t want, int lpr_log2, int R, int num_cus, uint32_t* grid_out, bool wide) {
    const size_t lds = (p.norm_scales || p.norm_plain || p.dg_o) ? ((size_t)(p.k / 32) * 36 + 16) * sizeof(float)
                       : (CPLT == 0 && BITS == 4)                  ? ((size_t)(p.k / 32) * 21 + 16) * sizeof(float) // packed row + per-step sums
                                                                   : 0;
    static const int spread_on = [] { // UZU_DEC_SPREAD=0: a partial round keeps 16 busy waves per workgroup on fewer CUs (A/B runs)
        const char* c = lab_env("UZU_DEC_SPREAD");
        return c ? atoi(c) : 1;
    }();
    static const int cap_override = [] {
        const char* c = lab_env("UZU_DEC_CAP");
        return c ? atoi(c) : 0;
    }();
#define UZU_LAUNCH(RR) UZU_LAUNCH_N(RR, 4)
#define UZU_LAUNCH_N(RR, NWV)                                                                                                       \
    do {                                                                                                                            \
        static int occ = 0; /* resident workgroups per CU of this instance (LDS use only lowers it for K > 8192: ignored) */        \
        if (!occ) {                                                                                                                 \
            int n = 0;                                                                                                              \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemv_dec_kernel<BITS, CPLT, RR, ACT, KIND, PRO, CONV, NWV>, 64 * NWV, lds) != hipSuccess || n < 1) n = 2; \
            occ = n > 8 ? 8 : n;                                                                                                    \
        }                                                                                                                           \
        uint32_t cap = (uint32_t)num_cus * (uint32_t)(cap_override > 0 ? cap_override : occ);                                       \
        if (p.part_val && p.part_capacity && cap > p.part_capacity) cap = p.part_capacity; /* one arg-max partial per workgroup */  \
        const uint32_t want_n = (want * 4 + NWV - 1) / NWV;                                                                         \
        uint32_t grid = want_n > cap ? cap : want_n;                                                                                \
        DecGemvParams pl = p;                                                                                                       \
        if (NWV > 4 && spread_on && grid < (uint32_t)num_cus && !(p.part_val && p.part_capacity && (uint32_t)num_cus > p.part_capacity)) { /* less than one round for the chip: spread it over every CU */     \
            pl.wg_batches = (want * 4 + (uint32_t)num_cus - 1) / (uint32_t)num_cus;                                                 \
            grid = (want * 4 + pl.wg_batches - 1) / pl.wg_batches;                                                                  \
        }                                                                                                                           \
        if (lds > 65536) { /* K > ~24k on the LDS-resident-row path: raise the instance's dynamic-LDS limit to what this call needs */ \
            static LdsLimit lim;                                                                                                    \
            if (!raise_lds_limit(lim, (const void*)gemv_dec_kernel<BITS, CPLT, RR, ACT, KIND, PRO, CONV, NWV>, lds)) {             \
                set_error("gemv_dec: %zu bytes of LDS for the activation row of K = %u are not available", lds, p.k);               \
                return UZU_ERR_UNSUPPORTED;                                                                                         \
            }                                                                                                                       \
        }                                                                                                                           \
        if (grid_out) *grid_out = grid;                                                                                             \
        const void* a0 = PRO == 2 ? (const void*)p.dg_o : (const void*)p.x;                                                         \
        const void* a1 = PRO == 2 ? (const void*)p.dg_sz : (const void*)p.shortcut_in;                                              \
        const void* a2 = PRO == 2 ? (const void*)p.dg_w : (const void*)p.norm_scales;                                               \
        return launch_check([&] { hipLaunchKernelGGL((gemv_dec_kernel<BITS, CPLT, RR, ACT, KIND, PRO, CONV, NWV>), dim3(grid), dim3(64 * NWV), lds, s, a0, a1, a2, p.k, lpr_log2, p.w[0], p.scales[0], p.biases[0], pl); }, "gemv_dec"); \
    } while (0)
    if constexpr (!CONV && PRO != 2 && (PRO == 1 || PRO == 3 || (CPLT == 0 && BITS == 4))) {
        if (wide) { // bandwidth regime (gemv_dec_plan: R <= 2): one wide workgroup per CU shares the prologue
            // 150-190 registers on the 4-step path: 3 waves per SIMD; int8 rows need 100-170 registers: 8 waves (two workgroups per CU)
            constexpr int NWV = BITS == 8 ? 8 : (CPLT == 4 ? 12 : 16);
            if (R >= 2) UZU_LAUNCH_N(2, NWV);
            UZU_LAUNCH_N(1, NWV);
        }
    }
    if (!ACT && !CONV && R == 4) UZU_LAUNCH((ACT || CONV) ? 2 : 4);
    if (R >= 2) UZU_LAUNCH(2);
    UZU_LAUNCH(1);
#undef UZU_LAUNCH
#undef UZU_LAUNCH_N
}
// an RHT linear whose epilogue (GatedActMul / DeltaNet conv) needs its OutputRht first: whole 32-row blocks per workgroup (PRO == 5)
bool gemv_dec_stripe_supported(const DecGemvParams& p, int num_cus) {
    static const bool on = [] { // UZU_DEC_STRIPE=0: the join launches behind the GEMV (A/B runs)
        const char* e = lab_env("UZU_DEC_STRIPE");
        return !e || atoi(e) != 0;
    }();
    if (!on || p.bits != 4 || p.n[1] || !(p.norm_scales || p.norm_plain) || p.k % 1024 || p.out_f32 || p.part_val || p.dg_o) return false;
    const uint32_t n_log0 = p.act_mul ? p.n[0] / 2 : p.n[0];
    if (!n_log0 || n_log0 % 32) return false;
    const uint32_t C = p.k / 32, lpr = 1u << gemv_lpr_log2(p.k), cpl = (C + lpr - 1) / lpr;
    if (cpl > 2 || (32u / (64u / lpr)) % 4) return false; // (a block's batches split evenly over four waves)
    const uint64_t weight_bytes = (uint64_t)p.n[0] * p.k / 2;
    static const uint64_t max_mb = [] { // UZU_DEC_STRIPE_MAX_MB (lab builds): the largest matrix that takes the stripe epilogue (A/B runs on the bandwidth-regime shapes)
        const char* e = lab_env("UZU_DEC_STRIPE_MAX_MB");
        return (uint64_t)(e && atoi(e) > 0 ? atoi(e) : 10);
    }();
    if (weight_bytes >= (max_mb << 20)) return false; // the bandwidth regime keeps its wide workgroups (one block per workgroup would idle most of their waves)
    (void)num_cus;
    return true;
}
template <int CPLT, bool ACT, int KIND>
static uzu_status launch_gemv_dec_stripe(hipStream_t s, const DecGemvParams& p, int lpr_log2, int num_cus, uint32_t* grid_out) {
    if (!gemv_dec_stripe_supported(p, num_cus)) {
        set_error("gemv_dec: the stripe epilogue does not cover this shape (n %u, k %u)", p.n[0], p.k);
        return UZU_ERR_UNSUPPORTED;
    }
    const size_t lds = ((size_t)(p.k / 32) * 36 + 16) * sizeof(float);
    const uint32_t stripes = (p.act_mul ? p.n[0] / 2 : p.n[0]) / 32, cap = (uint32_t)num_cus * 4;
    const uint32_t grid = stripes < cap ? stripes : cap;
    if (grid_out) *grid_out = grid;
    return launch_check([&] {
        // eight waves where a block has at least eight batches (one or two rows per wave pass): two batches per wave instead of four
        const uint32_t nbs = 32u / (64u >> lpr_log2);
        static const int nw_env = [] { // UZU_DEC_STRIPE_NW=4: four-wave workgroups (A/B runs)
            const char* e = lab_env("UZU_DEC_STRIPE_NW");
            return e ? atoi(e) : 8;
        }();
        if (nbs >= 8 && nbs % 8 == 0 && nw_env == 8)
            hipLaunchKernelGGL((gemv_dec_kernel<4, CPLT, 1, ACT, KIND, 5, false, 8>), dim3(grid), dim3(512), lds, s, (const void*)p.x, (const void*)p.shortcut_in, (const void*)p.norm_scales, p.k,
                               lpr_log2, p.w[0], p.scales[0], p.biases[0], p);
        else
            hipLaunchKernelGGL((gemv_dec_kernel<4, CPLT, 1, ACT, KIND, 5, false, 4>), dim3(grid), dim3(256), lds, s, (const void*)p.x, (const void*)p.shortcut_in, (const void*)p.norm_scales, p.k,
                               lpr_log2, p.w[0], p.scales[0], p.biases[0], p);
    }, "gemv_dec");
}
template <int BITS, int CPLT, bool ACT, int KIND, int PRO>
static uzu_status launch_gemv_dec_p(hipStream_t s, const DecGemvParams& p, uint32_t want, int lpr_log2, int R, int num_cus, uint32_t* grid_out, bool wide) {
    if constexpr (!ACT && PRO != 2 && PRO != 3 && PRO != 4) {
        if (p.conv_w) { // conv epilogue with prefetched operands (kernel size 4, instantiated for R <= 2)
            if (R > 2) {
                want *= (uint32_t)(R / 2);
                R = 2;
            }
            return launch_gemv_dec_c<BITS, CPLT, false, KIND, PRO, true>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
        }
    }
    return launch_gemv_dec_c<BITS, CPLT, ACT, KIND, PRO, false>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
}
template <int BITS, int CPLT, bool ACT, int KIND>
static uzu_status launch_gemv_dec_k(hipStream_t s, const DecGemvParams& p, uint32_t want, int lpr_log2, int R, int num_cus, uint32_t* grid_out, bool wide) {
    const bool normed = p.norm_scales || p.norm_plain;
    if constexpr (CPLT != 0) { // prologues keep the row in registers (checked by the caller)
        if constexpr (!ACT)
            if (p.dg_o) return launch_gemv_dec_p<BITS, CPLT, false, KIND, 2>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
        if (normed && p.ep_out_bits) { // RHT linear with the stripe epilogue (gemv_dec_stripe_supported)
            if constexpr (BITS == 4 && CPLT != 4) return launch_gemv_dec_stripe<CPLT, ACT, KIND>(s, p, lpr_log2, num_cus, grid_out);
            set_error("gemv_dec: the stripe epilogue is instantiated for int4 rows of one or two steps per lane");
            return UZU_ERR_UNSUPPORTED;
        }
        if (normed && (p.x_rht_bits || p.in_rht_bits)) { // Hadamard transforms around the normalisation (RHT linears)
            if constexpr (!ACT) {
                if (!p.conv_w) return launch_gemv_dec_p<BITS, CPLT, false, KIND, 3>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
            }
            set_error("gemv_dec: the Hadamard prologue is instantiated without the act-mul / conv epilogues (an RHT linear's outputs need their OutputRht first)");
            return UZU_ERR_UNSUPPORTED;
        }
        if (normed) return launch_gemv_dec_p<BITS, CPLT, ACT, KIND, 1>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
        if constexpr (!ACT && CPLT != 4) {
            if (p.in_rht_bits && !p.conv_w) return launch_gemv_dec_p<BITS, CPLT, false, KIND, 4>(s, p, want, lpr_log2, R, num_cus, grid_out, wide); // plain row + InputRht
        }
    }
    if constexpr (CPLT == 4) {
        set_error("gemv_dec: the 4-step register path is instantiated for the prologue variants only");
        return UZU_ERR_UNSUPPORTED;
    } else {
        return launch_gemv_dec_p<BITS, CPLT, ACT, KIND, 0>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
    }
}
template <int BITS, int CPLT, bool ACT>
static uzu_status launch_gemv_dec_r(hipStream_t s, const DecGemvParams& p, uint32_t want, int lpr_log2, int R, int num_cus, uint32_t* grid_out, bool wide) {
    switch (p.b_kind) {
    case UZU_MATMUL_B_SCALE_BIAS: return launch_gemv_dec_k<BITS, CPLT, ACT, UZU_MATMUL_B_SCALE_BIAS>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
    case UZU_MATMUL_B_SCALE_ZERO_POINT: return launch_gemv_dec_k<BITS, CPLT, ACT, UZU_MATMUL_B_SCALE_ZERO_POINT>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
    default: return launch_gemv_dec_k<BITS, CPLT, ACT, UZU_MATMUL_B_SCALE_SYMMETRIC>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
    }
}
template <int BITS, bool ACT>
static uzu_status launch_gemv_dec_cpl(hipStream_t s, const DecGemvParams& p, uint32_t want, int lpr_log2, int R, int num_cus, uint32_t* grid_out, bool wide) {
    const uint32_t C = p.k / 32, lpr = 1u << lpr_log2, cpl = (C + lpr - 1) / lpr;
    if (cpl == 1) return launch_gemv_dec_r<BITS, 1, ACT>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
    if (cpl == 2) return launch_gemv_dec_r<BITS, 2, ACT>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
    if (cpl <= 4 && (p.norm_scales || p.norm_plain || p.dg_o)) return launch_gemv_dec_r<BITS, 4, ACT>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
    if (p.norm_scales || p.norm_plain || p.dg_o) {
        set_error("gemv_dec: Normalization / norm-gate prologue supports K <= 8192, K %% 1024 == 0 (got %u)", p.k);
        return UZU_ERR_UNSUPPORTED;
    }
    return launch_gemv_dec_r<BITS, 0, ACT>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
}

#ifdef UZU_TIMELINE
static unsigned long long* g_tl_base = nullptr;
static uint32_t g_tl_max = 0, g_tl_next = 0;
unsigned long long* timeline_next_slot() {
    if (!g_tl_base || g_tl_next >= g_tl_max) return nullptr;
    return g_tl_base + (size_t)(g_tl_next++) * 1024 * UZU_TL_SLOTS;
}
extern "C" void uzu_hip_debug_set_timeline(unsigned long long* base, uint32_t max_launches) { g_tl_base = base, g_tl_max = max_launches, g_tl_next = 0; }
#endif

// a plain-row GEMV takes its linear's InputRht in registers (PRO == 4) when a lane holds its steps of the row there: <= 2 steps per lane
bool gemv_dec_plain_in_rht_supported(uint32_t k, uint32_t bits) {
    if (k % 32 || (bits != 4 && bits != 8)) return false;
    const uint32_t C = k / 32, lpr = 1u << gemv_lpr_log2(k);
    return (C + lpr - 1) / lpr <= 2;
}
uzu_status gemv_dec(hipStream_t s, const DecGemvParams& p_in, int num_cus, uint32_t* grid_out) {
    DecGemvParams p = p_in;
#ifdef UZU_TIMELINE
    p.tl = timeline_next_slot();
#endif
    if ((p.bits != 4 && p.bits != 8) || p.k % 32 || p.group_size % 32 || (p.group_size & (p.group_size - 1)) || ((p.norm_scales || p.norm_plain) && p.k % 1024)) {
        set_error("gemv_dec: unsupported shape (bits %u, k %u, group %u)", p.bits, p.k, p.group_size);
        return UZU_ERR_UNSUPPORTED;
    }
    if (p.dg_o) {
        const uint32_t nchunks = p.k / 8, per = nchunks > 256 ? nchunks / 256 : 1, dv = p.dg_dv;
        const bool ok = p.k % 8 == 0 && dv >= 8 && (dv & (dv - 1)) == 0 && dv <= 512 && p.k % dv == 0 && (nchunks <= 256 || (nchunks % 256 == 0 && per <= 4)) &&
                        (dv / 8) % per == 0 && !p.norm_scales && !p.norm_plain && !p.act_mul;
        if (!ok) {
            set_error("gemv_dec: unsupported norm-gate prologue (k %u, dv %u)", p.k, dv);
            return UZU_ERR_UNSUPPORTED;
        }
    }
    if (p.in_rht_bits && !p.norm_scales && !p.norm_plain && !p.dg_o && !gemv_dec_plain_in_rht_supported(p.k, p.bits)) {
        set_error("gemv_dec: the InputRht of a plain input row runs on register-resident rows only (k %u: more than two steps per lane)", p.k);
        return UZU_ERR_UNSUPPORTED;
    }
    if (p.conv_w && (p.conv_ks != 4 || p.act_mul || p.out_f32)) {
        set_error("gemv_dec: the conv epilogue is instantiated for kernel size 4 (got %u)", p.conv_ks);
        return UZU_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < 2; ++i)
        if ((uint64_t)p.n[i] * p.k * p.bits / 8 >= (1ull << 32)) { // per-lane byte offsets are 32-bit
            set_error("gemv_dec: weight matrix of %u x %u exceeds 4 GiB", p.n[i], p.k);
            return UZU_ERR_UNSUPPORTED;
        }
    if (gemv_stream_wanted(p)) { // bandwidth regime: LDS-staged weight stream (k_stream.hip); a geometry it does not cover falls through
        const uzu_status st = gemv_stream(s, p_in, num_cus, grid_out);
        if (st != UZU_ERR_UNSUPPORTED) return st;
    }
    int lpr_log2, R;
    bool wide = false;
    const uint32_t want = gemv_dec_plan(p, num_cus, &lpr_log2, &R, &wide);
    if (p.act_mul) {
        if (p.bits == 4) return launch_gemv_dec_cpl<4, true>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
        return launch_gemv_dec_cpl<8, true>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
    }
    if (p.bits == 4) return launch_gemv_dec_cpl<4, false>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
    return launch_gemv_dec_cpl<8, false>(s, p, want, lpr_log2, R, num_cus, grid_out, wide);
}

// The launch plan gemv_dec takes for a shape, without launching (host arithmetic only: callable without a GPU).  `wide_waves` is the
// launcher's workgroup width (launch_gemv_dec_c: 16 waves, 12 on the 4-step register path, 4 otherwise); `wg_batches` / `workgroups`
// are the spread of a partial round (0 / 0 when the grid is the persistent one, whose size depends on the instance's occupancy).
void gemv_dec_plan_query(const DecGemvParams& p, int num_cus, DecGemvPlan* out) {
    int lpr_log2 = 0, R = 1;
    bool wide = false;
    const uint32_t want = gemv_dec_plan(p, num_cus, &lpr_log2, &R, &wide);
    const uint32_t C = p.k / 32, lpr = 1u << lpr_log2, cpl = (C + lpr - 1) / lpr;
    const bool normed = p.norm_scales || p.norm_plain;
    const bool four_step = cpl > 2 && cpl <= 4 && (normed || p.dg_o);
    const bool wide_instance = wide && !p.conv_w && !p.dg_o && (normed || (p.bits == 4 && !four_step && cpl != 1 && cpl != 2));
    out->lpr_log2 = lpr_log2, out->rows_per_lane_group = R, out->steps_per_lane = cpl;
    out->waves = wide_instance ? (p.bits == 8 ? 8u : (four_step ? 12u : 16u)) : 4u;
    out->wave_batches = want * 4; // batches, rounded up to a multiple of four
    out->wg_batches = 0, out->workgroups = 0;
    if (out->waves > 4) {
        const uint32_t want_n = (want * 4 + out->waves - 1) / out->waves;
        if (want_n < (uint32_t)num_cus) {
            out->wg_batches = (want * 4 + (uint32_t)num_cus - 1) / (uint32_t)num_cus;
            out->workgroups = (want * 4 + out->wg_batches - 1) / out->wg_batches;
        }
    }
}

// ---------------------------------------------------------------------------------------------- argmax_commit
__global__ void __launch_bounds__(256) argmax_commit_kernel(const float* pv, const uint32_t* pi, uint32_t parts, uint32_t* ctx_len,
                                                            uint32_t* tokens, uint32_t* out_token, uint32_t* sampled, CommitEmbed eb) {
    __shared__ float sv[4];
    __shared__ uint32_t si[4];
    __shared__ uint32_t s_token;
    float bv = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < parts; i += 256) {
        const float v = pv[i];
        const uint32_t ix = pi[i];
        if (v > bv || (v == bv && ix < bi)) bv = v, bi = ix;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const uint32_t oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sv[wave] = bv, si[wave] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) bv = sv[w], bi = si[w];
        const uint32_t t = eb.token_in ? *eb.token_in : (bi == 0xFFFFFFFFu ? 0u : bi);
        const uint32_t len = *ctx_len;
        *out_token = t;
        sampled[len] = t;
        tokens[0] = t;
        *ctx_len = len + 1;
        s_token = t;
    }
    if (!eb.output) return;
    __syncthreads();
    // the sampled token's embedding row = input of the next decode step (same arithmetic as k_elementwise.hip's lookups)
    const uint32_t token_id = s_token;
    for (uint32_t dim_idx = threadIdx.x; dim_idx < eb.model_dim; dim_idx += 256) {
        float out_f;
        if (token_id >= eb.vocab_size) {
            out_f = 0.0f;
        } else if (eb.method == UZU_QUANT_NONE) {
            out_f = bf16_to_f32(((const uint16_t*)eb.weights)[(size_t)token_id * eb.model_dim + dim_idx]) * round_bf16(eb.input_scale);
        } else {
            const uint32_t packing_divisor = 8 / eb.bits;
            const size_t weights_stride = eb.model_dim / packing_divisor;
            const size_t num_groups = (eb.model_dim + eb.group_size - 1) / eb.group_size;
            const size_t zero_points_stride = eb.bits == 4 ? (num_groups + 1) / 2 : num_groups;
            const size_t group_idx = dim_idx / eb.group_size;
            const float scale = bf16_to_f32(eb.scales[(size_t)token_id * num_groups + group_idx]);
            int quantized_value;
            if (eb.bits == 4) {
                const uint8_t packed = eb.weights[(size_t)token_id * weights_stride + dim_idx / 2];
                quantized_value = (dim_idx & 1) == 0 ? (packed & 0x0F) : ((packed >> 4) & 0x0F);
            } else {
                quantized_value = eb.weights[(size_t)token_id * weights_stride + dim_idx];
            }
            float bias;
            if (eb.method == UZU_QUANT_SCALE_BIAS) {
                bias = bf16_to_f32(eb.biases[(size_t)token_id * num_groups + group_idx]);
            } else if (eb.method == UZU_QUANT_SCALE_ZERO_POINT) {
                uint8_t zero_point;
                if (eb.bits == 4) {
                    const uint8_t packed = eb.zero_points[(size_t)token_id * zero_points_stride + group_idx / 2];
                    zero_point = (group_idx & 1) == 0 ? (packed & 0x0F) : ((packed >> 4) & 0x0F);
                } else {
                    zero_point = eb.zero_points[(size_t)token_id * zero_points_stride + group_idx];
                }
                bias = -scale * (float)zero_point;
            } else {
                bias = -scale * (float)(1 << (eb.bits - 1));
            }
            out_f = scale * (float)quantized_value + bias;
            out_f = out_f * eb.input_scale;
        }
        eb.output[dim_idx] = f32_to_bf16(out_f);
    }
}
uzu_status argmax_commit(hipStream_t s, const float* pv, const uint32_t* pi, uint32_t parts, uint32_t* ctx_len, uint32_t* tokens,
                         uint32_t* out_token, uint32_t* sampled, const CommitEmbed* embed) {
    CommitEmbed eb{};
    if (embed) eb = *embed;
    return launch_check([&] { hipLaunchKernelGGL(argmax_commit_kernel, dim3(1), dim3(256), 0, s, pv, pi, parts, ctx_len, tokens, out_token, sampled, eb); },
                        "argmax_commit");
}

// ---------------------------------------------------------------------------------------------- delta_dec
// DeltaNetUpdate (update.rs:30-143) for one token, spread over the whole chip: grid (value head, Dv / 8), 256
// threads, one state row (Dk = 128 f32) per half-wave.  The conv + SiLU already ran in the in-proj epilogue, the
// RMSNorm * SiLU(z) gate runs in the out-proj prologue (it needs all Dv outputs of a head), so a workgroup only
// streams its 8 state rows: one CU sustains ~25-50 GB/s, a 64 KB head on one CU was the bottleneck of the fused
// single-workgroup version.  Every load is issued before the scalar math (sigmoid / softplus / exp, ~250
// instructions) so that the math hides under the state-row latency.
__global__ void __launch_bounds__(256) delta_dec_kernel(DeltaDecParams p) {
    constexpr int DK = 128;
    UZU_TL_DECL;
    UZU_TL_STAMP(0);
    const uint32_t hv = blockIdx.x;
    const uint32_t gph = p.num_v_heads / p.num_k_heads;
    const uint32_t hk = hv / gph;
    const uint32_t key_dim = p.key_dim, value_dim = p.value_dim, conv_dim = 2 * key_dim + value_dim;
    const uint32_t Dv = p.head_v_dim;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sl = lane & 31, half = lane >> 5;
    const uint32_t i = blockIdx.y * 8 + wave * 2 + half; // state row = output index inside the head
    const bool live = i < Dv;
    const uint32_t row = live ? i : 0;

    float4* srow = (float4*)(p.state + ((size_t)hv * Dv + row) * DK) + sl;
    const float4 sv = *srow;
    const uint2 qr = *(const uint2*)(p.in_proj + hk * DK + sl * 4);
    const uint2 kr = *(const uint2*)(p.in_proj + key_dim + hk * DK + sl * 4);
    const float v_i = bf16_to_f32(p.in_proj[2 * key_dim + hv * Dv + row]);
    const float z_i = bf16_to_f32(p.in_proj[conv_dim + hv * Dv + row]);
    const float beta_raw = bf16_to_f32(p.in_proj[conv_dim + value_dim + hv]);
    const float a_raw = bf16_to_f32(p.in_proj[conv_dim + value_dim + p.num_v_heads + hv]);
    const float dt_b = p.dt_bias[hv], a_l = p.a_log[hv];

    float q[4] = {bits_to_f32(qr.x << 16), bits_to_f32(qr.x & 0xFFFF0000u), bits_to_f32(qr.y << 16), bits_to_f32(qr.y & 0xFFFF0000u)};
    float kk[4] = {bits_to_f32(kr.x << 16), bits_to_f32(kr.x & 0xFFFF0000u), bits_to_f32(kr.y << 16), bits_to_f32(kr.y & 0xFFFF0000u)};
    float q_sq = 0.f, k_sq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
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
    const float beta = delta_beta_fast(beta_raw);
    const float decay = delta_decay_fast(a_raw, dt_b, a_l);
    const float sz_i = silu_f32(z_i);

    UZU_TL_STAMP(2); // scalar math done (state row may still be in flight)
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
    if (live) {
        float4 ns;
        ns.x = decay * s4[0] + kk[0] * delta_i;
        ns.y = decay * s4[1] + kk[1] * delta_i;
        ns.z = decay * s4[2] + kk[2] * delta_i;
        ns.w = decay * s4[3] + kk[3] * delta_i;
        *srow = ns;
        if (sl == 0) {
            p.o[hv * Dv + i] = o_i;
            p.sz[hv * Dv + i] = sz_i;
        }
    }
    UZU_TL_STAMP(4);
    UZU_TL_FLUSH(p);
}
uzu_status delta_dec(hipStream_t s, const DeltaDecParams& p_in) {
    DeltaDecParams p = p_in;
#ifdef UZU_TIMELINE
    p.tl = timeline_next_slot();
#endif
    if (p.head_v_dim > 512 || p.num_k_heads == 0 || p.num_v_heads % p.num_k_heads) {
        set_error("delta_dec: unsupported configuration");
        return UZU_ERR_UNSUPPORTED;
    }
    return launch_check([&] { hipLaunchKernelGGL(delta_dec_kernel, dim3(p.num_v_heads, (p.head_v_dim + 7) / 8), dim3(256), 0, s, p); }, "delta_dec");
}

// ---------------------------------------------------------------------------------------------- attn_dec
// One decode token: QKVNorm (q, k) + AttentionPrepare (RoPE, KV append) + split-KV attention pass 1.
// grid (kv_head * subs + sub, split); 256 threads.  Keys of split s: i = s, s + S, s + 2S, ...
// Latency design: the context length is the only thing the kernel has to wait for before it can issue every
// other load (qkv row, norm scales, RoPE row, and the first batch of K/V rows of each key group).
__device__ uint32_t g_attn_err_dev; // set when a fused attn_dec's bounded wait for its group gave up (host: attn_dec_check)

template <int HD, int GS, bool FUSED>
__global__ void __launch_bounds__(256) attn_dec_kernel(AttnDecParams p) {
    constexpr int LPK = HD / 8, KG = 64 / LPK, NGRP = 4 * KG;
    constexpr int TB = 4; // keys per group per batch (all K/V loads of a batch are in flight together)
    __shared__ float s_q[GS][HD];
    __shared__ float s_knew[HD], s_vnew[HD];
    __shared__ float s_o[NGRP][GS][HD];
    __shared__ float s_m[NGRP][GS], s_l[NGRP][GS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    UZU_TL_DECL;
    UZU_TL_STAMP(0);
    const uint32_t subs = p.gqa_factor / GS;
    const uint32_t kvh = blockIdx.x / subs, sub = blockIdx.x % subs;
    const uint32_t head0 = kvh * p.gqa_factor + sub * GS;
    const uint32_t split = blockIdx.y, S = gridDim.y;
#if !UZU_ATTN_SPEC
    const uint32_t L = __builtin_amdgcn_readfirstlane(*p.ctx_len); // position of the new token = number of cached keys
#else
    // The context length lives in device memory (a replayed graph cannot carry it) and everything position-dependent hangs on it.
    // It is requested FIRST and through the vector pipe: vector loads return in issue order, so nothing issued behind it -- the new
    // token's rows, the first K / V batch -- delays it, and waiting for it does not wait for them (a scalar load would tie it to
    // every kernel-argument fetch: scalar loads return out of order and can only be waited for all at once; requested behind the
    // K / V batch it arrived after 1.9-2.4 us instead of ~1, tools/timeline.py column x).
    uint32_t lane_zero = 0;
    asm volatile("" : "+v"(lane_zero)); // opaque per-lane zero: keeps the load on the vector pipe
    const uint32_t L_v = p.ctx_len[lane_zero];
    __builtin_amdgcn_sched_barrier(0);
#endif
    const uint32_t nq = p.num_heads, nkv = p.num_heads / p.gqa_factor;
    const uint32_t rope_dim = p.rope_dim, half_rope = rope_dim / 2;

    const int kgrp = lane / LPK, sl = lane % LPK;
    const uint32_t my_group = wave * KG + kgrp;
    const uint16_t* kbase = p.keys + (size_t)kvh * HD + sl * 8;
    const uint16_t* vbase = p.values + (size_t)kvh * HD + sl * 8;
    const size_t seq_stride = (size_t)nkv * HD;
    const uint32_t key0 = split + S * my_group, key_step = S * NGRP;
    // Load order (VMEM returns in issue order, a load the compiler cannot count forces vmcnt(0)): the new token's q/k/v
    // rows and norm scales (L2) first, then the first K/V batch (HBM), then the RoPE row; everything unconditional with
    // clamped addresses.  The prologue between the loads and the key loop only uses LDS barriers.
    constexpr int NJ = (GS + 2 + 3) / 4;           // prologue jobs per wave
    constexpr int NROPE = ((GS + 1) * (HD / 2) + 255) / 256;
    float vals[NJ][HD / 64], nsc[NJ][HD / 64];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int job = min(wave + 4 * jj, GS + 1);
        const bool is_q = job < GS, is_k = job == GS;
        const uint32_t head_idx = is_q ? head0 + job : (is_k ? nq + kvh : nq + nkv + kvh);
        const uint16_t* src = p.qkv + (size_t)head_idx * HD;
        // (field by field: a reference picked between the two argument structs is a pointer into the kernel-argument segment, and
        // its `scales` a dependent vector load at the head of the kernel)
        const float* nscales = is_q ? p.q_norm.scales : p.k_norm.scales;
        const float* nsrc = nscales ? nscales : (const float*)p.qkv; // dummy source (>= 2 head rows long), never consumed
#pragma unroll
        for (int j = 0; j < HD / 64; ++j) {
            vals[jj][j] = bf16_to_f32(src[lane + 64 * j]);
            nsc[jj][j] = nsrc[lane + 64 * j];
        }
    }
    u32x4_v kq[TB], vq[TB];
    auto fetch_upto = [&](uint32_t first, uint32_t row_limit) {
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const uint32_t i = min(first + t * key_step, row_limit); // clamped rows are loaded (cache hit) and never consumed
            kq[t] = *(const u32x4_v*)(kbase + (size_t)i * seq_stride);
            vq[t] = *(const u32x4_v*)(vbase + (size_t)i * seq_stride);
        }
    };
#if UZU_ATTN_SPEC
    // The first K / V batch does not wait for the context length (a dependent round trip to the memory-side cache, ~1 us, at the
    // head of a kernel that is one latency chain): its rows are clamped to the cache's capacity instead -- rows past the context
    // are allocated, loaded and, like the clamped ones, never consumed.
    fetch_upto(key0, p.cache_rows - 1);
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t L = __builtin_amdgcn_readfirstlane(L_v); // position of the new token = number of cached keys
#else
    fetch_upto(key0, L ? L - 1 : 0);
#endif
    const uint32_t last_row = L ? L - 1 : 0;
    auto fetch = [&](uint32_t first) { fetch_upto(first, last_row); };
    float rc[NROPE][4];
    if (rope_dim) {
        const float* cosr = p.cosines + (size_t)L * rope_dim;
        const float* sinr = p.sines + (size_t)L * rope_dim;
#pragma unroll
        for (int r = 0; r < NROPE; ++r) {
            const uint32_t t = min((uint32_t)tid + 256u * r, (GS + 1) * half_rope - 1), d = t % half_rope;
            rc[r][0] = cosr[d], rc[r][1] = sinr[d], rc[r][2] = cosr[d + half_rope], rc[r][3] = sinr[d + half_rope];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    UZU_TL_STAMP(1); // the context length has arrived (the RoPE row's loads are out)

    // ---- prologue: normalised + roped q heads (one wave per head, elements lane + 64 j), new k and v rows ----
    // (QKVNorm: qkv_norm.rs:45-76 ; AttentionPrepare: attention_prepare.rs:7-31,104-121)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int job = wave + 4 * jj;
        if (job >= GS + 2) break;
        const bool is_q = job < GS, is_k = job == GS;
        float total = 0.f;
#pragma unroll
        for (int j = 0; j < HD / 64; ++j) total += vals[jj][j] * vals[jj][j];
        DecNorm nm;
        nm.present = is_q ? p.q_norm.present : p.k_norm.present, nm.full_layer = is_q ? p.q_norm.full_layer : p.k_norm.full_layer;
        nm.eps = is_q ? p.q_norm.eps : p.k_norm.eps, nm.offset = is_q ? p.q_norm.offset : p.k_norm.offset;
        nm.scales = is_q ? p.q_norm.scales : p.k_norm.scales;
        if ((is_q || is_k) && nm.present) {
            total = wave_sum(total);
            const float rms_norm = 1.0f / sqrtf(total / (float)HD + nm.eps);
#pragma unroll
            for (int j = 0; j < HD / 64; ++j) {
                const float normalized = vals[jj][j] * rms_norm;
                if (!nm.scales)
                    vals[jj][j] = round_bf16(normalized);
                else if (nm.full_layer)
                    vals[jj][j] = round_bf16(normalized * (nsc[jj][j] + nm.offset));
                else
                    vals[jj][j] = round_bf16(round_bf16(normalized) * round_bf16(nsc[jj][j] + nm.offset));
            }
        }
        float* dst = is_q ? s_q[job] : (is_k ? s_knew : s_vnew);
#pragma unroll
        for (int j = 0; j < HD / 64; ++j) dst[lane + 64 * j] = vals[jj][j];
    }
    lds_barrier();
    if (rope_dim) { // half-rotation RoPE on q heads and the new key (table row = absolute position L); one thread per pair
#pragma unroll
        for (int r = 0; r < NROPE; ++r) {
            const uint32_t t = (uint32_t)tid + 256u * r;
            if (t < (GS + 1) * half_rope) {
                const uint32_t v = t / half_rope, d = t % half_rope;
                float* vec = v < GS ? s_q[v] : s_knew;
                const float a = vec[d], b = vec[d + half_rope];
                const float lo = round_bf16(a * rc[r][0] + (-b) * rc[r][1]);
                const float hi = round_bf16(b * rc[r][2] + a * rc[r][3]);
                vec[d] = lo;
                vec[d + half_rope] = hi;
            }
        }
        lds_barrier();
    }
    if (split == 0 && sub == 0) { // append the new K / V rows to the cache (layout [tokens, kv_heads, hd])
        for (uint32_t d = tid; d < HD; d += 256) {
            p.keys[((size_t)L * nkv + kvh) * HD + d] = f32_to_bf16(s_knew[d]);
            p.values[((size_t)L * nkv + kvh) * HD + d] = f32_to_bf16(s_vnew[d]);
        }
    }

    UZU_TL_STAMP(2);
    // ---- split-KV online softmax over keys i = key0 + key_step * t, i <= L (causal, suffix length 1) ----
    float q[GS][8], o[GS][8], mx[GS], sm[GS];
#pragma unroll
    for (int g = 0; g < GS; ++g) {
#pragma unroll
        for (int e = 0; e < 8; ++e) q[g][e] = p.scale * s_q[g][sl * 8 + e], o[g][e] = 0.f;
        mx[g] = -1e9f;
        sm[g] = 0.f;
    }
    for (uint32_t base = split; base <= L; base += TB * key_step) { // wave-uniform trip count
        const uint32_t first = base + S * my_group;
        u32x4_v kc[TB], vc[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) kc[t] = kq[t], vc[t] = vq[t];
        fetch(first + TB * key_step); // next batch in flight while this one is consumed (clamped past the end)
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const uint32_t i = first + t * key_step;
            if (i > L) break;
            float kf[8], vf[8];
            if (i == L) {
#pragma unroll
                for (int e = 0; e < 8; ++e) kf[e] = s_knew[sl * 8 + e], vf[e] = s_vnew[sl * 8 + e];
            } else {
                const uint32_t kw[4] = {kc[t].x, kc[t].y, kc[t].z, kc[t].w}, vw[4] = {vc[t].x, vc[t].y, vc[t].z, vc[t].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kf[2 * j] = bits_to_f32(kw[j] << 16), kf[2 * j + 1] = bits_to_f32(kw[j] & 0xFFFF0000u);
                    vf[2 * j] = bits_to_f32(vw[j] << 16), vf[2 * j + 1] = bits_to_f32(vw[j] & 0xFFFF0000u);
                }
            }
#pragma unroll
            for (int g = 0; g < GS; ++g) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) part = fmaf(q[g][e], kf[e], part);
                const float score = group_sum<LPK>(part);
                const float new_max = fmaxf(mx[g], score);
                const float factor = fast_exp(mx[g] - new_max);
                const float exp_score = fast_exp(score - new_max);
                mx[g] = new_max;
                sm[g] = sm[g] * factor + exp_score;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[g][e] = o[g][e] * factor + exp_score * vf[e];
            }
        }
    }
    UZU_TL_STAMP(5); // K / V rows have arrived and are folded in
    // every key group parks its state in LDS; the merge below runs over the NGRP groups in group order
#pragma unroll
    for (int g = 0; g < GS; ++g) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[my_group][g][sl * 8 + e] = o[g][e];
        if (sl == 0) s_m[my_group][g] = mx[g], s_l[my_group][g] = sm[g];
    }
    lds_barrier();
    for (int idx = tid; idx < GS * HD; idx += 256) {
        const int g = idx / HD, e = idx % HD;
        float m = s_m[0][g];
#pragma unroll
        for (int w = 1; w < NGRP; ++w) m = fmaxf(m, s_m[w][g]);
        float l = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < NGRP; ++w) {
            const float f = fast_exp(s_m[w][g] - m);
            l += s_l[w][g] * f;
            acc += s_o[w][g][e] * f;
        }
        const size_t row = (size_t)(head0 + g) * S + split;
        if constexpr (FUSED) {
            // write-through (sc1) stores: the merging workgroups read them with sc1 loads, no cache maintenance on either side
            // (MI355X_MICROARCH.md, "Valid forms": sc1 payload -> vmcnt(0) -> flag; sc1 loads may replace the acquire when the producer stored sc1);
            // two elements per store: even lanes take their odd neighbour's value (8-byte sc1 stores cost half the fabric writes of 4-byte ones)
            const float hi = __shfl_down(acc, 1, 64);
            if ((e & 1) == 0) {
                const unsigned long long v = (unsigned long long)f32_to_bits(acc) | ((unsigned long long)f32_to_bits(hi) << 32);
                __hip_atomic_store((unsigned long long*)(p.partials + row * HD + e), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (e == 0) {
                __hip_atomic_store(p.sums + row, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.maxs + row, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            p.partials[row * HD + e] = acc;
            if (e == 0) p.sums[row] = l, p.maxs[row] = m;
        }
    }
    UZU_TL_STAMP(6);
    if constexpr (FUSED) {
        // ---- pass 2 inside the launch: AttentionTwoPass2 (attention_two_pass.rs:143-190) over the S splits + SigmoidGate, in
        // attn_merge_kernel's arithmetic order (same weights, same 16 x 16 fma chains, same sums): bit-identical rows.
        // Publish: every store of the workgroup acknowledged, then ONE ticket per workgroup.  The ticket counter is monotonic: launches on
        // one stream are serialised and every launch adds exactly S per group, so ticket / S numbers the launch and (ticket / S + 1) S is the
        // value the counter reaches when the group of THIS launch is complete -- nothing to reset between launches or graph replays.
        __shared__ float s_gsum[GS];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (the key groups' merge state is dead from here on: its LDS holds the merge weights [GS][256] and the 16 x 16 slice sums)
        float (*s_w)[256] = (float (*)[256]) & s_o[0][0][0];
        float (*s_acc2)[16] = (float (*)[16])(&s_o[0][0][0] + GS * 256);
        static_assert((size_t)NGRP * GS * HD >= (size_t)GS * 256 + 256, "merge scratch does not fit the key-group state");
        uint32_t* ticket = p.tickets + blockIdx.x;
        if (tid == 0) {
            const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = (t / S + 1u) * S;
            uint32_t spins = 0;
            while ((int32_t)(__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { // ~0.3 s: a workgroup of the group never arrived (it cannot be resident?) -- give up loudly, never hang
                    atomicOr(&g_attn_err_dev, 1u);
                    break;
                }
            }
        }
        __syncthreads();
        UZU_TL_STAMP(3); // the group is complete
        // ONE memory round trip: the partials of this workgroup's slice (first 16 outputs: thread = (element e of 16, split-slice of 16)) are
        // requested before the merge weights are derived from the maxima / sums
        constexpr uint32_t kOut = GS * HD;
        const uint32_t el = kOut / S, first = split * el; // el * S == kOut (host-checked)
        const uint32_t e16 = tid & 15, slice = tid >> 4;
        auto load_slice = [&](uint32_t base, float (&pv)[16], uint32_t& g, uint32_t& j, bool& live, float& gt) {
            const uint32_t oi = first + base + e16;
            live = base + e16 < el;
            g = live ? oi / HD : 0, j = live ? oi % HD : 0;
            const float* pp = p.partials + (size_t)(head0 + g) * S * HD + j;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint32_t b = slice + 16 * t;
                pv[t] = (live && b < S) ? __hip_atomic_load(pp + (size_t)b * HD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            }
            gt = (p.gate && live && slice == 0) ? bf16_to_f32(p.gate[(size_t)(head0 + g) * HD + j]) : 0.f;
        };
        float pv0[16], gt0;
        uint32_t g0, j0;
        bool live0;
        load_slice(0, pv0, g0, j0, live0, gt0);
        // merge weights of the heads this workgroup's slice touches (el <= HD: at most two): wave w takes head g_lo + w; lane L owns
        // splits L + 64 t (attn_merge_kernel, wave 0)
        const uint32_t g_lo = first / HD, g_hi = (first + el - 1) / HD;
        for (uint32_t g = g_lo + wave; g <= g_hi; g += 4) {
            const size_t hrow = (size_t)(head0 + g) * S;
            float mv[4], sv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t b = lane + 64 * t;
                mv[t] = b < S ? __hip_atomic_load(p.maxs + hrow + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
                sv[t] = b < S ? __hip_atomic_load(p.sums + hrow + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            }
            const float gmax = wave_max(fmaxf(fmaxf(mv[0], mv[1]), fmaxf(mv[2], mv[3])));
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t b = lane + 64 * t;
                if (b < S) {
                    const float w = fast_exp(mv[t] - gmax);
                    s_w[g][b] = w;
                    part += sv[t] * w;
                }
            }
            part = wave_sum(part);
            if (lane == 0) s_gsum[g] = part;
        }
        __syncthreads();
        auto finish = [&](const float (&pv)[16], uint32_t g, uint32_t j, bool live, float gt) {
            float val = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint32_t b = slice + 16 * t;
                if (b < S) val = fmaf(pv[t], s_w[g][b], val);
            }
            s_acc2[slice][e16] = val;
            __syncthreads();
            if (slice == 0 && live) {
                float total = 0.f;
#pragma unroll
                for (int q2 = 0; q2 < 16; ++q2) total += s_acc2[q2][e16];
                float r = round_bf16(total / s_gsum[g]);
                if (p.gate) r = round_bf16(r * (1.0f / (1.0f + expf_glibc(-gt)))); // SigmoidGate (sigmoid_gate.rs:9-22)
                p.out[(size_t)(head0 + g) * HD + j] = f32_to_bf16(r);
            }
            __syncthreads();
        };
        finish(pv0, g0, j0, live0, gt0);
        for (uint32_t base = 16; base < el; base += 16) { // slices of more than 16 outputs (few splits): further round trips
            float pv[16], gt;
            uint32_t g, j;
            bool live;
            load_slice(base, pv, g, j, live, gt);
            finish(pv, g, j, live, gt);
        }
    }
    UZU_TL_STAMP(4);
    UZU_TL_FLUSH(p);
}

template <int HD> static uzu_status launch_attn_dec(hipStream_t s, const AttnDecParams& p, uint32_t splits) {
    const uint32_t kv_heads = p.num_heads / p.gqa_factor;
    const uint32_t gs = attn_dec_group_size(p.gqa_factor);
    const dim3 grid(kv_heads * (p.gqa_factor / gs), splits);
// The in-launch pass 2 (FUSED) is compiled only with -DUZU_ATTN_FUSED_BUILD (make FUSED_ATTN=1): measured SLOWER than attn_dec + attn_merge on
// this chip (profiles/r5_sdpa_fused_ab.txt: 15.2 us against 6.9 + 4.4 us per layer on Qwen3.5-0.8B, 15.3 against 7.3 + 4.3 on Llama-3-8B,
// bit-identical rows): store acknowledgement -> ticket -> poll -> sc1 reads are four dependent fabric round trips where a kernel boundary
// costs 1.5 us.  The default library carries no instance of it.
#ifdef UZU_ATTN_FUSED_BUILD
#define UZU_LAUNCH(G)                                                                                                                           \
    return p.out ? launch_check([&] { hipLaunchKernelGGL((attn_dec_kernel<HD, G, true>), grid, dim3(256), 0, s, p); }, "attn_dec")              \
                 : launch_check([&] { hipLaunchKernelGGL((attn_dec_kernel<HD, G, false>), grid, dim3(256), 0, s, p); }, "attn_dec")
#else
#define UZU_LAUNCH(G) return launch_check([&] { hipLaunchKernelGGL((attn_dec_kernel<HD, G, false>), grid, dim3(256), 0, s, p); }, "attn_dec")
#endif
    switch (gs) {
    case 6: UZU_LAUNCH(6);
    case 5: UZU_LAUNCH(5);
    case 4: UZU_LAUNCH(4);
    case 3: UZU_LAUNCH(3);
    case 2: UZU_LAUNCH(2);
    default: UZU_LAUNCH(1);
    }
#undef UZU_LAUNCH
}
int g_attn_fused_override = -1; // uzu_hip_debug_set_attn_fused: -1 = environment / default, 0 = two launches, 1 = fused
bool attn_dec_fused_supported(uint32_t num_heads, uint32_t gqa_factor, uint32_t head_dim, uint32_t splits, int num_cus) {
#ifndef UZU_ATTN_FUSED_BUILD
    return false; // not compiled in (see UZU_LAUNCH above)
#endif
    static const bool env_on = [] { // UZU_ATTN_FUSED=1 selects it (a FUSED_ATTN=1 build; A/B runs; same arithmetic): off by default -- measured slower
        const char* e = lab_env("UZU_ATTN_FUSED");
        return e && e[0] == '1';
    }();
    const bool off = g_attn_fused_override >= 0 ? g_attn_fused_override == 0 : !env_on;
    if (off || !gqa_factor || num_heads % gqa_factor || !splits || splits > 256) return false;
    const uint32_t gs = attn_dec_group_size(gqa_factor), groups = (num_heads / gqa_factor) * (gqa_factor / gs);
    if ((gs * head_dim) % splits) return false;              // every workgroup merges gs * hd / splits outputs of its group
    return (uint64_t)groups * splits <= (uint64_t)num_cus * 2; // the group's workgroups wait for each other: all must be resident (<= 4 fit a CU)
}
static uint32_t g_attn_fused_launched = 0; // devices (bit d) that have run a fused attn_dec
uzu_status attn_dec_check() {
    if (!g_attn_fused_launched) return UZU_OK;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(g_attn_fused_launched >> (dev & 31) & 1u)) return UZU_OK;
    uint32_t v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_attn_err_dev), 4) != hipSuccess) {
        (void)hipGetLastError();
        return UZU_OK;
    }
    if (!v) return UZU_OK;
    const uint32_t zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_err_dev), &zero, 4);
    set_error("attn_dec: a workgroup gave up waiting for its KV-head group (fused pass 2): the attention rows since the last check are garbage");
    return UZU_ERR_HIP;
}
uzu_status attn_dec(hipStream_t s, const AttnDecParams& p_in, uint32_t splits) {
    AttnDecParams p = p_in;
    if (p.out) {
        if (!p.tickets) {
            set_error("attn_dec: the fused pass 2 needs the ticket words");
            return UZU_ERR_INVALID_ARGUMENT;
        }
        int dev = 0;
        (void)hipGetDevice(&dev);
        g_attn_fused_launched |= 1u << (dev & 31);
    }
#ifdef UZU_TIMELINE
    p.tl = timeline_next_slot();
#endif
    if (p.rope_dim > p.head_dim || (p.rope_dim & 1)) {
        set_error("attn_dec: bad rope_dim %u", p.rope_dim);
        return UZU_ERR_INVALID_ARGUMENT;
    }
    if (!p.cache_rows) {
        set_error("attn_dec: cache_rows (allocated rows of the K / V caches) must be given");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    switch (p.head_dim) {
    case 64: return launch_attn_dec<64>(s, p, splits);
    case 128: return launch_attn_dec<128>(s, p, splits);
    case 256: return launch_attn_dec<256>(s, p, splits);
    default:
        set_error("attn_dec: unsupported head_dim %u", p.head_dim);
        return UZU_ERR_UNSUPPORTED;
    }
}

// AttentionTwoPass2 over S splits + SigmoidGate.  grid (heads, hd / 16); thread = (element e of 16, key-slice of 16).
// Every load (partials of the thread's splits, the S maxima / sums, the gate) is issued before anything is
// consumed, so the kernel pays one memory round trip; wave 0 derives the S merge weights meanwhile.
__global__ void __launch_bounds__(256) attn_merge_kernel(const float* partials, const float* sums, const float* maxs, const uint16_t* gate,
                                                         uint16_t* out, uint32_t HD, uint32_t S) {
    __shared__ float sw[256];
    __shared__ float s_acc[16][16];
    __shared__ float s_gsum;
    const uint32_t head = blockIdx.x, e = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const uint32_t j = blockIdx.y * 16 + e;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = j < HD;
    float pv[16];
    const float* pp = partials + (size_t)head * S * HD + j;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t b = slice + 16 * t;
        pv[t] = (live && b < S) ? pp[(size_t)b * HD] : 0.f;
    }
    const float g = (gate && live && slice == 0) ? bf16_to_f32(gate[(size_t)head * HD + j]) : 0.f;
    if (wave == 0) {
        float mv[4], sv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t b = lane + 64 * t;
            mv[t] = b < S ? maxs[(size_t)head * S + b] : -INFINITY;
            sv[t] = b < S ? sums[(size_t)head * S + b] : 0.f;
        }
        const float gmax = wave_max(fmaxf(fmaxf(mv[0], mv[1]), fmaxf(mv[2], mv[3])));
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t b = lane + 64 * t;
            if (b < S) {
                const float w = fast_exp(mv[t] - gmax);
                sw[b] = w;
                part += sv[t] * w;
            }
        }
        part = wave_sum(part);
        if (lane == 0) s_gsum = part;
    }
    __syncthreads();
    float val = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t b = slice + 16 * t;
        if (b < S) val = fmaf(pv[t], sw[b], val);
    }
    s_acc[slice][e] = val;
    __syncthreads();
    if (slice == 0 && live) {
        float total = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) total += s_acc[q][e];
        float r = round_bf16(total / s_gsum);
        if (gate) r = round_bf16(r * (1.0f / (1.0f + expf_glibc(-g)))); // SigmoidGate (sigmoid_gate.rs:9-22)
        out[(size_t)head * HD + j] = f32_to_bf16(r);
    }
}
uzu_status attn_merge(hipStream_t s, const float* partials, const float* sums, const float* maxs, const uint16_t* gate, uint16_t* out,
                      uint32_t num_heads, uint32_t head_dim, uint32_t splits) {
    if (splits > 256) {
        set_error("attn_merge: at most 256 splits");
        return UZU_ERR_UNSUPPORTED;
    }
    return launch_check([&] {
        hipLaunchKernelGGL(attn_merge_kernel, dim3(num_heads, (head_dim + 15) / 16), dim3(256), 0, s, partials, sums, maxs, gate, out, head_dim, splits);
    }, "attn_merge");
}

} // namespace k
} // namespace uzu

// tests / A-B runs: -1 = environment / default, 0 = attn_dec + attn_merge, 1 = fused (decided when a step is encoded or captured)
extern "C" void uzu_hip_debug_set_attn_fused(int mode) { uzu::k::g_attn_fused_override = mode; }
#ifdef UZU_ATTN_FUSED_BUILD
extern "C" int uzu_hip_debug_attn_fused_built(void) { return 1; }
#else
extern "C" int uzu_hip_debug_attn_fused_built(void) { return 0; }
#endif
