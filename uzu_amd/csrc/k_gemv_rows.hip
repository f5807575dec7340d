// k_gemv_rows.hip -- quantised matmul for a HANDFUL of activation rows (2 <= M <= 16) on the matrix cores.
//
// Where it sits: between the decode GEMV (M = 1, k_decode.hip) and the tiled GEMMs (k_gemm.hip from ~20 rows).  A speculative
// verify pass (uzu_hip_model_verify_tree: <= 16 tree nodes) and short prefill tails run every linear at such an M; the GEMV in passes
// of four rows re-reads the weights per pass and pays 16 packed dots per row and step on the vector unit, a 64-row GEMM tile is a
// quarter full and pays its LDS pipeline for nothing (profiles/r4_verify_cost.json: 2.9 / 3.6 ms for a 16-node pass of Qwen3.5-0.8B
// against 1.3 ms at M = 1).
//
// Shape: 16 activation rows are exactly the N side of v_mfma_f32_16x16x32_bf16, so a wave computes a [16 weight rows] x [16 tokens] tile
// per instruction with the WEIGHTS as the A operand straight from global memory (no LDS, coalesced 64-byte row pieces, `nt` like the
// GEMV) and the activations as the B operand from LDS (staged once per workgroup, already in the k order of the code conversion):
//   * lane (r = l % 16, g = l / 16) loads 16 bytes = 32 int4 codes of weight row r at k0 + 32 g; word i of them converts to the 8 bf16
//     (16 + q) of MFMA i (gemv_core.h: two codes per v_and_or_b32) -- k order [n0 n4 n1 n5 n2 n6 n3 n7], which the staged activations share;
//   * the four MFMAs of a step contract over k0 .. k0 + 127 = ONE quant group (group_size % 128 == 0), so the group's scale / offset
//     fold is four packed multiply-adds per lane on the 16 x 16 result (rows 4 (l / 16) + v of token l % 16), the scalars broadcast inside
//     a 16-lane row with DPP row_share;
//   * sum_k x of the group (the offset term, and the 16 of the code trick) comes from the matrix core as well: the same B against an
//     all-ones A -- no vector work;
//   * the workgroup's waves split K (steps w, w + NW, ...) and add their tiles through LDS in wave order.
// Per 1 KiB of codes: 32 conversions + ~24 fold / broadcast VALU + 8 MFMAs for ALL 16 rows -- the M = 1 GEMV's instruction budget.
// Results are tolerance-class against the scalar reference like every reduction kernel (f32 accumulation in another order).
#include "decode_epilogue.h"
#include "device_utils.h"
#include "kernels.h"

namespace uzu {
namespace k {

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 gr_bf16x8;
typedef __attribute__((ext_vector_type(4))) float gr_f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t gr_u32x4;

// value of lane `src` (0..15) of this lane's 16-lane row (DPP row_share: one VALU, no LDS)
template <int SRC> __device__ __forceinline__ float row_share(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + SRC, 0xF, 0xF, false));
}

struct RowsParams {
    const uint16_t* a; // [m, k] bf16
    const uint8_t* w;  // [n, k / 2]
    const uint16_t* scales;
    const uint16_t* biases;
    const uint8_t* zp;
    const uint16_t* out_bias; // [n] or null
    void* d;                  // [m, n] bf16 or f32
    uint32_t m, n, k, group_size, kind, d_f32;
    uint32_t act_type; // ACT instances: n = 2 h weight rows [up | gate]; d = [m, h] = up * act(gate) (GatedActMul, gated_act_mul.rs:36-70)
    RowsNorm norm;     // NORM instances: `a` holds the raw rows
};

// ACT: the workgroup owns the 16 up rows row0 .. and the 16 gate rows h + row0 .. (two accumulators, one activation stream), and its epilogue is
// GatedActMul on the rounded pair -- the decode GEMV's act-mul epilogue (k_decode.hip) for a handful of rows: no [m, 2 h] round trip, no launch.
template <int NW, bool ACT, bool NORM>
__global__ void __launch_bounds__(64 * NW) gemv_rows_mfma_kernel(RowsParams p) {
    constexpr int NB = ACT ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t gr_smem[];
    const uint32_t K = p.k, pitch = K * 2 + 16; // bytes per activation row in LDS (+16: rows start 4 banks apart)
    uint8_t* xs = gr_smem;                      // [16][pitch]
    float* s_part = (float*)(gr_smem + (NORM ? 16 * pitch : 0)); // [NW][NB][16 rows][16 tokens]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t row0 = blockIdx.x * 16;
    const uint32_t steps = K / 128, G = K / p.group_size, gshift = 31 - __builtin_clz(p.group_size);
    const uint32_t zp_stride = (G + 1) / 2;
    // ---- weight row of this lane: clamped into range (a clamped row is computed and never stored); block 1 (ACT) = the gate rows, h further down
    const uint32_t out_n = ACT ? p.n / 2 : p.n;
    const uint32_t wrow = min(row0 + r, out_n - 1);
    const uint8_t* wp = p.w + (size_t)wrow * (K / 2) + 16 * g;
    // rows whose scale / offset this lane-group folds: 4 g + (lane % 4) (lanes 0..3 of the row carry them, the rest duplicate)
    const uint32_t frow = min(row0 + 4 * g + (r & 3), out_n - 1);
    // !NORM: the activation chunks go straight from the L2 into the lanes that feed them to the matrix core (lane (r, g): token r, k0 + 32 g + 8 i,
    // four 16-byte loads per step riding in the same prefetch slot as the codes).  The waves of a workgroup split K, so a staged copy in LDS has no
    // second reader inside the workgroup -- staging it first cost every one of the hundreds of workgroups a 32-112 KB copy and a barrier in front of
    // its first MFMA.  Same chunks, same k order, same MFMAs: bit-identical to the staged form (which the Normalization prologue keeps).
    struct Step {
        uint4 codes[NB];
        uint16_t sc[NB], of[NB];
        gr_u32x4 xv[NORM ? 1 : 4];
    };
    const uint16_t* xrow = p.a + (size_t)min(r, p.m - 1) * K + 32 * g; // rows >= m: a clamped row, zeroed before use
    auto load_step = [&](uint32_t s, Step& st) {
        const uint32_t sc_ = min(s, steps - 1); // past the end: a reload that is never consumed (keeps the loads countable)
        const uint32_t grp = (sc_ * 128) >> gshift;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const size_t wr = (size_t)wrow + (b ? out_n : 0), fr = (size_t)frow + (b ? out_n : 0);
            st.codes[b] = load16_stream(p.w + wr * (K / 2) + 16 * g + (size_t)sc_ * 64);
            st.sc[b] = p.scales[fr * G + grp];
            if (p.kind == UZU_MATMUL_B_SCALE_BIAS) st.of[b] = p.biases[fr * G + grp];
            else if (p.kind == UZU_MATMUL_B_SCALE_ZERO_POINT) {
                const uint8_t z = p.zp[fr * zp_stride + (grp >> 1)];
                st.of[b] = (grp & 1) ? (z >> 4) : (z & 0x0F);
            } else st.of[b] = 0;
        }
        if constexpr (!NORM) {
#pragma unroll
            for (int i = 0; i < 4; ++i) st.xv[i] = *(const gr_u32x4*)(xrow + (size_t)sc_ * 128 + 8 * i);
        }
    };
    (void)wp;
    Step stA, stB;
    load_step(wave, stA); // in flight while the activations are staged
    // ---- stage the activation rows: 16-byte chunks (8 consecutive k) in the conversion's k order; rows >= m are zero
    if constexpr (NORM) {
        // Normalization prologue (normalization.rs:56-125 with AccumT = f32, no mean / biases): normalization_kernel's arithmetic per row -- thread t of
        // 256 owns the E = K / 256 elements [t E, t E + E), one fma chain, the wave butterfly, ((w0 + w1) + w2) + w3 -- so the staged rows are the rows
        // the separate kernel would have written.  256 threads per row (a 512-thread workgroup takes two rows at a time); two passes over the rows
        // (sums, then values: the second read comes from the L2), element e of a row lands at chunk e / 8, pair e % 4, half e / 4 % 2.
        const uint32_t E = K / 256, t = tid & 255u, rg = tid >> 8;
        constexpr uint32_t RG = NW / 4;
        float* red = s_part; // [16 rows][4 waves]: free until the tiles are parked
        auto load4 = [&](uint32_t row, uint32_t e, float (&v)[4]) {
            const uint2 xr = *(const uint2*)(p.a + (size_t)row * K + e);
            v[0] = __builtin_bit_cast(float, xr.x << 16), v[1] = __builtin_bit_cast(float, xr.x & 0xFFFF0000u);
            v[2] = __builtin_bit_cast(float, xr.y << 16), v[3] = __builtin_bit_cast(float, xr.y & 0xFFFF0000u);
            if (p.norm.residual_add) {
                const uint2 sr = *(const uint2*)(p.norm.shortcut_in + (size_t)row * K + e);
                const float sc[4] = {__builtin_bit_cast(float, sr.x << 16), __builtin_bit_cast(float, sr.x & 0xFFFF0000u), __builtin_bit_cast(float, sr.y << 16),
                                     __builtin_bit_cast(float, sr.y & 0xFFFF0000u)};
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = round_bf16(v[i] + sc[i]);
            }
        };
        for (uint32_t row = rg; row < p.m; row += RG) {
            float ss = 0.f;
            for (uint32_t q = 0; q < E; q += 4) {
                float v[4];
                load4(row, t * E + q, v);
                if (p.norm.shortcut_out && blockIdx.x == 0) {
                    uint2 o;
                    o.x = (__builtin_bit_cast(uint32_t, v[0]) >> 16) | (__builtin_bit_cast(uint32_t, v[1]) & 0xFFFF0000u);
                    o.y = (__builtin_bit_cast(uint32_t, v[2]) >> 16) | (__builtin_bit_cast(uint32_t, v[3]) & 0xFFFF0000u);
                    *(uint2*)(p.norm.shortcut_out + (size_t)row * K + t * E + q) = o;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) ss = fmaf(v[i], v[i], ss);
            }
            ss = wave_sum(ss);
            if (lane == 0) red[row * 4 + (t >> 6)] = ss;
        }
        lds_barrier();
        for (uint32_t row = rg; row < p.m; row += RG) {
            const float total = ((red[row * 4] + red[row * 4 + 1]) + red[row * 4 + 2]) + red[row * 4 + 3];
            const float variance = total / (float)K - 0.0f * 0.0f;
            const float rms_inv = 1.0f / sqrtf(variance + p.norm.eps);
            for (uint32_t q = 0; q < E; q += 4) {
                const uint32_t e = t * E + q;
                float v[4];
                load4(row, e, v);
                float scl[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.norm.scales) {
                    const float4 s4 = *(const float4*)(p.norm.scales + e);
                    scl[0] = s4.x, scl[1] = s4.y, scl[2] = s4.z, scl[3] = s4.w;
                }
                uint16_t* dst = (uint16_t*)(xs + (size_t)row * pitch + (size_t)(e / 8) * 16) + (e % 8) / 4; // pair i of the chunk at +2 i halfwords; half e / 4 % 2
                uint16_t ob[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float normalized = (v[i] - 0.0f) * rms_inv;
                    float r_;
                    if (!p.norm.scales) r_ = round_bf16(normalized);
                    else if (p.norm.full_layer) r_ = round_bf16(normalized * (scl[i] + p.norm.offset));
                    else r_ = round_bf16(round_bf16(normalized) * round_bf16(scl[i] + p.norm.offset));
                    ob[i] = f32_to_bf16(r_);
                    dst[2 * i] = ob[i];
                }
                if (p.norm.normed_out && blockIdx.x == 0) {
                    uint2 o;
                    o.x = (uint32_t)ob[0] | ((uint32_t)ob[1] << 16), o.y = (uint32_t)ob[2] | ((uint32_t)ob[3] << 16);
                    *(uint2*)(p.norm.normed_out + (size_t)row * K + e) = o;
                }
            }
        }
        // rows >= m: zero
        const uint32_t chunks_per_row = K / 8, total = (16 - p.m) * chunks_per_row;
        for (uint32_t idx = tid; idx < total; idx += 64 * NW) {
            const uint32_t j = p.m + idx / chunks_per_row, c = idx % chunks_per_row;
            *(gr_u32x4*)(xs + (size_t)j * pitch + (size_t)c * 16) = gr_u32x4{0u, 0u, 0u, 0u};
        }
        lds_barrier();
    }
    uint32_t mask = 0x00780078u, magic = 0x41804180u, ones = 0x3F803F80u;
    asm("" : "+s"(mask));
    asm("" : "+v"(magic));
    asm("" : "+v"(ones));
    const gr_u32x4 ones4 = {ones, ones, ones, ones};
    gr_f32x4 acc[NB]; // rows 4 g + v of token r
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = gr_f32x4{0.f, 0.f, 0.f, 0.f};
    const uint8_t* xlane = xs + (size_t)r * pitch + 64 * g; // this lane's B chunks: token r, k0 + 32 g + 8 i
    auto compute = [&](uint32_t s, const Step& st) {
        gr_f32x4 dq[NB], dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NB; ++b) dq[b] = gr_f32x4{0.f, 0.f, 0.f, 0.f};
        const uint8_t* xb = xlane + (size_t)s * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gr_u32x4 bv;
            if constexpr (NORM) {
                bv = *(const gr_u32x4*)(xb + 16 * i);
            } else { // (x0,x1) (x2,x3) (x4,x5) (x6,x7) -> the conversion's k order (x0,x4) (x1,x5) (x2,x6) (x3,x7); rows >= m are zero
                const gr_u32x4 u = st.xv[i];
                bv.x = __builtin_amdgcn_perm(u.z, u.x, 0x05040100u), bv.y = __builtin_amdgcn_perm(u.z, u.x, 0x07060302u);
                bv.z = __builtin_amdgcn_perm(u.w, u.y, 0x05040100u), bv.w = __builtin_amdgcn_perm(u.w, u.y, 0x07060302u);
                if (r >= p.m) bv = gr_u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const uint32_t w = i == 0 ? st.codes[b].x : i == 1 ? st.codes[b].y : i == 2 ? st.codes[b].z : st.codes[b].w;
                gr_u32x4 av;
                av.x = ((w << 3) & mask) | magic, av.y = ((w >> 1) & mask) | magic, av.z = ((w >> 5) & mask) | magic, av.w = ((w >> 9) & mask) | magic;
                dq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gr_bf16x8, av), __builtin_bit_cast(gr_bf16x8, bv), dq[b], 0, 0, 0);
            }
            dx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gr_bf16x8, ones4), __builtin_bit_cast(gr_bf16x8, bv), dx, 0, 0, 0);
        }
        // group fold: acc[v] += scale_v * dq[v] + (offset_v - 16 scale_v) * sum_x   (dx[v] = sum_k x of token r, the same in every v)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float sc = bf16_to_f32(st.sc[b]);
            float of;
            if (p.kind == UZU_MATMUL_B_SCALE_BIAS) of = bf16_to_f32(st.of[b]);
            else if (p.kind == UZU_MATMUL_B_SCALE_ZERO_POINT) of = -sc * (float)st.of[b];
            else of = -sc * 8.0f;
            of = fmaf(-kQ4Offset, sc, of);
            const float s0 = row_share<0>(sc), s1 = row_share<1>(sc), s2 = row_share<2>(sc), s3 = row_share<3>(sc);
            const float o0 = row_share<0>(of), o1 = row_share<1>(of), o2 = row_share<2>(of), o3 = row_share<3>(of);
            acc[b].x = fmaf(s0, dq[b].x, fmaf(o0, dx.x, acc[b].x));
            acc[b].y = fmaf(s1, dq[b].y, fmaf(o1, dx.x, acc[b].y));
            acc[b].z = fmaf(s2, dq[b].z, fmaf(o2, dx.x, acc[b].z));
            acc[b].w = fmaf(s3, dq[b].w, fmaf(o3, dx.x, acc[b].w));
        }
    };
    for (uint32_t s = wave; s < steps; s += 2 * NW) {
        load_step(s + NW, stB);
        compute(s, stA);
        load_step(s + 2 * NW, stA);
        if (s + NW < steps) compute(s + NW, stB);
    }
    // ---- add the waves' tiles in wave order, bias, store: thread t = (row t / 16, token t % 16)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float* mine = s_part + ((size_t)wave * NB + b) * 256;
        mine[(4 * g + 0) * 16 + r] = acc[b].x, mine[(4 * g + 1) * 16 + r] = acc[b].y, mine[(4 * g + 2) * 16 + r] = acc[b].z, mine[(4 * g + 3) * 16 + r] = acc[b].w;
    }
    lds_barrier();
    if (tid < 256) {
        const uint32_t row = tid >> 4, j = tid & 15;
        float v[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            v[b] = s_part[b * 256 + tid];
#pragma unroll
            for (int w = 1; w < NW; ++w) v[b] += s_part[((size_t)w * NB + b) * 256 + tid];
        }
        if (j < p.m && row0 + row < out_n) {
            float value = 1.0f * v[0]; // MatmulKernel epilogue with ab_scale = 1 (kernel.rs:281-292)
            if (p.out_bias) value += bf16_to_f32(p.out_bias[row0 + row]);
            if constexpr (ACT) {
                float gate = 1.0f * v[1];
                if (p.out_bias) gate += bf16_to_f32(p.out_bias[out_n + row0 + row]);
                const float up_b = round_bf16(value), gate_b = round_bf16(gate); // the rounded outputs the unfused pair of kernels would have stored
                ((uint16_t*)p.d)[(size_t)j * out_n + row0 + row] = f32_to_bf16(round_bf16(up_b * act_bf16(p.act_type, gate_b, kExp2fTab))); // gated_act_mul/mod.rs:5-12
            } else if (p.d_f32) ((float*)p.d)[(size_t)j * p.n + row0 + row] = value;
            else ((uint16_t*)p.d)[(size_t)j * p.n + row0 + row] = f32_to_bf16(value);
        }
    }
}
} // namespace

bool gemv_rows_mfma_supported(const MatmulParams& p) {
    static const bool on = [] { // UZU_GEMV_ROWS=0: small M back on the GEMV passes / GEMM tiles (A/B runs)
        const char* e = lab_env("UZU_GEMV_ROWS");
        return !e || atoi(e) != 0;
    }();
    if (!on || exact_mode()) return false;
    if (p.m < 2 || p.m > 16 || p.bits != 4 || p.b_kind == UZU_MATMUL_B_FULL_PRECISION) return false;
    if (p.w_dt != UZU_BF16 || p.a_dt != UZU_BF16 || (p.d_dt != UZU_BF16 && p.d_dt != UZU_F32)) return false;
    if (p.signed_codes || p.ab_scale != 1.0f || p.accumulate || p.has_soft_cap || p.gather) return false;
    if (p.act_mul && ((p.n & 1) || p.d_dt != UZU_BF16)) return false; // GatedActMul epilogue: [up | gate] halves, bf16 out
    if (p.k % 128 || p.group_size % 128 || (p.group_size & (p.group_size - 1)) || p.k % p.group_size) return false;
    if ((uintptr_t)p.a % 16 || (uintptr_t)p.b % 16) return false;
    return true;
}

// the Normalization prologue needs whole 4-element vectors per thread of the 256-thread mapping (k % 1024 == 0) on top of the kernel's own conditions
// and the rows it stages must fit the CU's LDS
bool gemv_rows_norm_supported(const MatmulParams& p) {
    return gemv_rows_mfma_supported(p) && p.k % 1024 == 0 && (size_t)16 * (p.k * 2 + 16) + 8 * 256 * 4 <= 150 * 1024;
}

template <int NW, bool ACT, bool NORM> static uzu_status launch_rows(hipStream_t s, const RowsParams& q, uint32_t blocks, size_t lds) {
    static LdsLimit lim;
    if (!raise_lds_limit(lim, (const void*)gemv_rows_mfma_kernel<NW, ACT, NORM>, lds)) {
        set_error("gemv_rows: %zu bytes of LDS are not available", lds);
        return UZU_ERR_UNSUPPORTED;
    }
    return launch_check([&] { hipLaunchKernelGGL((gemv_rows_mfma_kernel<NW, ACT, NORM>), dim3(blocks), dim3(64 * NW), lds, s, q); }, "gemv_rows_mfma");
}

uzu_status gemv_rows_mfma(hipStream_t s, const MatmulParams& p, const RowsNorm* norm) {
    RowsParams q{};
    q.a = (const uint16_t*)p.a, q.w = (const uint8_t*)p.b, q.scales = (const uint16_t*)p.scales, q.biases = (const uint16_t*)p.biases, q.zp = p.zero_points;
    q.out_bias = (const uint16_t*)p.bias, q.d = p.d, q.m = p.m, q.n = p.n, q.k = p.k, q.group_size = p.group_size, q.kind = p.b_kind, q.d_f32 = p.d_dt == UZU_F32;
    q.act_type = p.act_type;
    if (norm) {
        if (!gemv_rows_norm_supported(p) || (norm->residual_add && !norm->shortcut_in) || (norm->shortcut_out && norm->shortcut_out == norm->shortcut_in)) {
            set_error("gemv_rows: the Normalization prologue does not cover this shape (k %u), or its residual rows alias", p.k);
            return UZU_ERR_UNSUPPORTED;
        }
        q.norm = *norm;
    }
    const bool act = p.act_mul != 0;
    const uint32_t blocks = ((act ? p.n / 2 : p.n) + 15) / 16, steps = p.k / 128;
    // eight waves split K where the row blocks alone leave most of the chip idle and there are steps to share
    const bool wide = !act && blocks < 256 && steps >= 16;
    const int nw = wide ? 8 : 4;
    const size_t lds = (norm ? (size_t)16 * (p.k * 2 + 16) : 0) + (size_t)nw * (act ? 2 : 1) * 256 * 4; // the staged rows only where the prologue writes them
    if (norm) return act ? launch_rows<4, true, true>(s, q, blocks, lds) : wide ? launch_rows<8, false, true>(s, q, blocks, lds) : launch_rows<4, false, true>(s, q, blocks, lds);
    return act ? launch_rows<4, true, false>(s, q, blocks, lds) : wide ? launch_rows<8, false, false>(s, q, blocks, lds) : launch_rows<4, false, false>(s, q, blocks, lds);
}

} // namespace k
} // namespace uzu
