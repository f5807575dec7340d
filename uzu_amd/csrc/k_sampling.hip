// k_sampling.hip -- UnifiedSampling, every specialisation (BU/cpu/kernel/sampling/unified_sampling.rs:13-99, Gumbel noise:
// BU/../encodable_block/sampling/gumbel.rs:1-81).
//
// Reference order of operations per row: grammar bitmask (masked -> -inf), temperature (multiply by 1 / T), the three
// filters top-k / top-p / min-p as ONE descending pass that stops at the first element failing any of them (survivors =
// a prefix of the order "value descending, index ascending"), Gumbel-max noise on every logit, arg-max (ties -> lowest
// index).
//
// gfx950 form.  Two regimes:
//   * no filter (greedy with a bitmask / temperature, or plain stochastic sampling): embarrassingly parallel -- 256
//     workgroups per row compute transform(+ Philox4x32-10 Gumbel noise) and a partial arg-max, a second kernel merges
//     (the noise is ~150 integer + 2 logf per logit: 248k logits want the whole chip);
//   * any filter: ONE 1024-thread workgroup per row finds the cut of the descending order WITHOUT sorting --
//     a 4 x 8-bit radix descent over order-preserving keys whose per-bucket histograms carry the element COUNT (top-k)
//     and the probability MASS (top-p) at once; min-p is a plain value threshold.  Mass is accumulated in 2^-40 fixed
//     point with integer atomics, so the result does not depend on the order in which lanes arrive (deterministic).  The
//     tie group at the cut (equal logits; bf16 logits tie a lot) is split by index rank, as the reference's sort does.
//     Noise is then evaluated for the survivors only.
// Tolerance class: the reference sums exp(l - max) and the running top-p mass sequentially in f32 in sorted order; here both
// are exact fixed-point sums.  The survivor set can therefore differ only when the running mass lands within ~1e-6 of top_p
// (tests/golden/sampling.json and the CPU-restatement comparisons in tests/test_gpu_kernels.py pass bit for bit).  Everything else -- mask, temperature,
// expf / logf (uzu_math.h: glibc's algorithms), Philox, the 24-bit uniform, the arg-max rule -- is exact.
#include "device_utils.h"
#include "kernels.h"
#include "sampling_noise.h"

namespace uzu {
namespace k {

namespace {

// PRng::derive (encodable_block/sampling/prng.rs:12-24): the seed of the row at absolute position `index` -- a 64-bit finaliser of
// seed + index.  The engine's chained decode reads the position from the device-resident context length (+ `offset` for a prefill
// chunk's last row), so a replayed hipGraph derives a fresh seed every step.
__global__ void derive_seed_kernel(uint64_t base, const uint32_t* position, uint32_t offset, uint64_t* out) {
    uint64_t hash = base + (uint64_t)(*position + offset);
    hash ^= hash >> 33;
    hash *= 0xff51afd7ed558ccdull;
    hash ^= hash >> 33;
    hash *= 0xc4ceb9fe1a85ec53ull;
    hash ^= hash >> 33;
    *out = hash;
}

struct SampleArgs {
    const void* logits;
    uint32_t* output;
    const uint64_t* seeds;   // null <=> greedy
    const uint32_t* bitmask; // null <=> no grammar mask
    float recip_temperature; // 1 / T, or 0 <=> no temperature
    uint32_t has_temperature, has_top_k, has_top_p, has_min_p;
    uint32_t top_k;
    float top_p, min_p;
    uint32_t vocab_size, batch_size;
};

template <class T> __device__ __forceinline__ float transformed(const SampleArgs& a, const T* row, const uint32_t* mask_row, uint32_t i) {
    float v = ld(row, i);
    if (mask_row && (mask_row[i / 32u] & (1u << (i % 32u))) == 0) v = -INFINITY;
    if (a.has_temperature) v *= a.recip_temperature;
    return v;
}

// ---- regime 1: no filter ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) sample_plain_pass1(SampleArgs a, float* pv, uint32_t* pi) {
    __shared__ float sv[4];
    __shared__ uint32_t si[4];
    const uint32_t row = blockIdx.y, V = a.vocab_size;
    const T* l = (const T*)a.logits + (size_t)row * V;
    const uint32_t* mask_row = a.bitmask ? a.bitmask + (size_t)row * ((V + 31u) / 32u) : nullptr;
    const uint64_t seed = a.seeds ? a.seeds[row] : 0ull;
    float bv = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < V; i += gridDim.x * 256u) {
        float v = transformed(a, l, mask_row, i);
        if (a.seeds) v += gumbel_of(seed, i, V);
        if (i == 0 && !(v == v)) v = INFINITY; // the reference's fold never replaces a NaN first element
        if (v > bv || (v == bv && i < bi)) bv = v, bi = i;
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
        pv[(size_t)row * gridDim.x + blockIdx.x] = bv;
        pi[(size_t)row * gridDim.x + blockIdx.x] = bi;
    }
}
__global__ void __launch_bounds__(256) sample_plain_pass2(const float* pv, const uint32_t* pi, uint32_t parts, uint32_t* output) {
    __shared__ float sv[4];
    __shared__ uint32_t si[4];
    const uint32_t row = blockIdx.x;
    float bv = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < parts; i += 256u) {
        const float v = pv[(size_t)row * parts + i];
        const uint32_t ix = pi[(size_t)row * parts + i];
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
        output[row] = bi == 0xFFFFFFFFu ? 0u : bi; // everything -inf (fully masked row): index 0, as the reference's fold
    }
}

// ---- regime 2: top-k / top-p / min-p ------------------------------------------------------------------------------------
constexpr int kFixShift = 40; // probability mass in units of 2^-40
__device__ __forceinline__ uint32_t orderable(float v) { // larger float <=> larger key; -0.0 == +0.0
    const uint32_t raw = f32_to_bits(v);
    const uint32_t b = (raw << 1) == 0u ? 0u : raw;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ unsigned long long to_fixed(float p) { return (unsigned long long)((double)p * (double)(1ull << kFixShift) + 0.5); }

template <class T>
__global__ void __launch_bounds__(1024) sample_filtered_kernel(SampleArgs a) {
    constexpr int NT = 1024;
    __shared__ float s_red_f[16];
    __shared__ unsigned long long s_red_u[16];
    __shared__ uint32_t s_cnt[256];
    __shared__ unsigned long long s_mass[256];
    __shared__ uint32_t s_scan[NT];
    __shared__ float s_max, s_norm;
    __shared__ uint32_t s_prefix, s_cb, s_found, s_keep, s_best_i[16];
    __shared__ unsigned long long s_mb;
    __shared__ float s_best_v[16];
    const uint32_t row = blockIdx.x, V = a.vocab_size, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const T* l = (const T*)a.logits + (size_t)row * V;
    const uint32_t* mask_row = a.bitmask ? a.bitmask + (size_t)row * ((V + 31u) / 32u) : nullptr;

    // (1) maximum of the transformed logits
    float mx = -INFINITY;
    for (uint32_t i = tid; i < V; i += NT) mx = fmaxf(mx, transformed(a, l, mask_row, i));
    mx = wave_max(mx);
    if (lane == 0) s_red_f[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        float m = s_red_f[0];
        for (int w = 1; w < 16; ++w) m = fmaxf(m, s_red_f[w]);
        s_max = m;
    }
    __syncthreads();
    const float logits_max = s_max;

    // (2) normaliser sum exp(l - max): fixed point, integer reduction
    unsigned long long acc = 0;
    for (uint32_t i = tid; i < V; i += NT) acc += to_fixed(expf_glibc(transformed(a, l, mask_row, i) - logits_max));
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) s_red_u[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += s_red_u[w];
        s_norm = (float)((double)t / (double)(1ull << kFixShift));
        s_prefix = 0, s_cb = 0, s_mb = 0, s_found = 1;
    }
    __syncthreads();
    const float logits_norm = s_norm;
    const unsigned long long top_p_fix = a.has_top_p ? to_fixed(a.top_p) : ~0ull;

    // (3) radix descent to the key of the first element that fails top-k / top-p (s_found == 0: nothing fails)
    const bool need_cut = a.has_top_k || a.has_top_p;
    if (need_cut) {
        for (int level = 0; level < 4; ++level) {
            const int shift = 24 - 8 * level;
            if (tid < 256) s_cnt[tid] = 0, s_mass[tid] = 0;
            __syncthreads();
            if (s_found) {
                const uint32_t prefix = s_prefix;
                const uint32_t hi_mask = level == 0 ? 0u : 0xFFFFFFFFu << (shift + 8);
                for (uint32_t i = tid; i < V; i += NT) {
                    const float v = transformed(a, l, mask_row, i);
                    const uint32_t key = orderable(v);
                    if ((key & hi_mask) != prefix) continue;
                    const uint32_t d = (key >> shift) & 0xFFu;
                    atomicAdd(&s_cnt[d], 1u);
                    atomicAdd(&s_mass[d], to_fixed(expf_glibc(v - logits_max) / logits_norm));
                }
            }
            __syncthreads();
            if (tid == 0 && s_found) {
                uint32_t cb = s_cb;
                unsigned long long mb = s_mb;
                int hit = -1;
                for (int d = 255; d >= 0; --d) {
                    const uint32_t c = s_cnt[d];
                    if (!c) continue;
                    const bool cond_k = a.has_top_k && cb + c > a.top_k;
                    const bool cond_p = a.has_top_p && mb + s_mass[d] >= top_p_fix;
                    if (cond_k || cond_p) {
                        hit = d;
                        break;
                    }
                    cb += c, mb += s_mass[d];
                }
                if (hit < 0) {
                    s_found = 0;
                } else {
                    s_prefix |= (uint32_t)hit << shift, s_cb = cb, s_mb = mb;
                    if (level == 3) { // tie group at the cut: `c` equal logits of per-element mass m / c, ordered by index
                        const uint32_t c = s_cnt[hit];
                        const unsigned long long each = s_mass[hit] / c;
                        uint32_t keep = c;
                        if (a.has_top_k) keep = min(keep, a.top_k > cb ? a.top_k - cb : 0u);
                        if (a.has_top_p) {
                            uint32_t np;
                            if (mb >= top_p_fix) np = 0;
                            else if (each == 0) np = c;
                            else np = (uint32_t)min((unsigned long long)c, (top_p_fix - mb + each - 1) / each); // elements with mass_before < top_p
                            keep = min(keep, np);
                        }
                        s_keep = keep;
                    }
                }
            }
            __syncthreads();
        }
    }
    const bool cut = need_cut && s_found;
    const uint32_t cut_key = s_prefix, cut_keep = s_keep;
    const float min_p_thr = a.has_min_p ? logits_max + logf_glibc(a.min_p) : -INFINITY;

    // (4) survivors, noise, arg-max.  Contiguous index chunks per thread so that the tie group can be ranked by index.
    const uint32_t chunk = (V + NT - 1) / NT, i0 = min(tid * chunk, V), i1 = min(i0 + chunk, V);
    uint32_t ties = 0;
    if (cut)
        for (uint32_t i = i0; i < i1; ++i) ties += orderable(transformed(a, l, mask_row, i)) == cut_key;
    s_scan[tid] = ties;
    __syncthreads();
    // exclusive scan over the 1024 per-thread tie counts (Hillis-Steele in LDS)
    for (uint32_t off = 1; off < NT; off <<= 1) {
        const uint32_t add = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += add;
        __syncthreads();
    }
    uint32_t tie_rank = s_scan[tid] - ties;
    const uint64_t seed = a.seeds ? a.seeds[row] : 0ull;
    float bv = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = i0; i < i1; ++i) {
        float v = transformed(a, l, mask_row, i);
        bool alive = true;
        if (cut) {
            const uint32_t key = orderable(v);
            if (key == cut_key) alive = tie_rank++ < cut_keep;
            else alive = key > cut_key;
        }
        if (a.has_min_p && v < min_p_thr) alive = false;
        if (!alive) v = -INFINITY;
        else if (a.seeds) v += gumbel_of(seed, i, V);
        if (v > bv || (v == bv && i < bi)) bv = v, bi = i;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const uint32_t oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
    }
    if (lane == 0) s_best_v[wave] = bv, s_best_i[wave] = bi;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (s_best_v[w] > bv || (s_best_v[w] == bv && s_best_i[w] < bi)) bv = s_best_v[w], bi = s_best_i[w];
        a.output[row] = bi == 0xFFFFFFFFu ? 0u : bi;
    }
}

} // namespace

static constexpr uint32_t kSampleParts = 256;
size_t unified_sampling_scratch_bytes(uint32_t batch_size) { return (size_t)batch_size * kSampleParts * 8; }

uzu_status unified_sampling(hipStream_t s, const UnifiedSamplingParams& p, void* scratch) {
    if (!p.batch_size || !p.vocab_size) return UZU_OK;
    SampleArgs a{};
    a.logits = p.logits, a.output = p.output, a.seeds = p.seeds, a.bitmask = p.bitmask;
    a.has_temperature = p.has_temperature, a.recip_temperature = p.has_temperature ? 1.0f / p.temperature : 0.0f;
    a.has_top_k = p.has_top_k, a.has_top_p = p.has_top_p, a.has_min_p = p.has_min_p;
    a.top_k = p.top_k, a.top_p = p.top_p, a.min_p = p.min_p;
    a.vocab_size = p.vocab_size, a.batch_size = p.batch_size;
    if (p.has_top_k || p.has_top_p || p.has_min_p) {
        return UZU_DISPATCH_T(p.dt, [&]() -> uzu_status {
            return launch_check([&] { hipLaunchKernelGGL((sample_filtered_kernel<T>), dim3(p.batch_size), dim3(1024), 0, s, a); }, "unified_sampling[filtered]");
        });
    }
    if (!scratch) {
        set_error("unified_sampling: no scratch for the two-level arg-max");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    float* pv = (float*)scratch;
    uint32_t* pi = (uint32_t*)((char*)scratch + (size_t)p.batch_size * kSampleParts * 4);
    uint32_t parts = (p.vocab_size + 255u) / 256u;
    if (parts > kSampleParts) parts = kSampleParts;
    UZU_PROPAGATE(UZU_DISPATCH_T(p.dt, [&]() -> uzu_status {
        return launch_check([&] { hipLaunchKernelGGL((sample_plain_pass1<T>), dim3(parts, p.batch_size), dim3(256), 0, s, a, pv, pi); }, "unified_sampling[pass1]");
    }));
    return launch_check([&] { hipLaunchKernelGGL(sample_plain_pass2, dim3(p.batch_size), dim3(256), 0, s, pv, pi, parts, p.output); }, "unified_sampling[pass2]");
}

// a speculated tree: node i samples with PRng::derive(root position + height_i) (speculators/dflash_tfm.rs:267,304; trie.rs:147-151)
__global__ void derive_tree_seeds_kernel(uint64_t base, const uint32_t* position, const uint32_t* trie, uint32_t nodes, uint64_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes) return;
    uint64_t hash = base + (uint64_t)(*position + trie[3 * i + 2]);
    hash ^= hash >> 33;
    hash *= 0xff51afd7ed558ccdull;
    hash ^= hash >> 33;
    hash *= 0xc4ceb9fe1a85ec53ull;
    hash ^= hash >> 33;
    out[i] = hash;
}
uzu_status derive_tree_seeds(hipStream_t s, uint64_t base, const uint32_t* position, const uint32_t* trie, uint32_t nodes, uint64_t* out) {
    if (!nodes) return UZU_OK;
    return launch_check([&] { hipLaunchKernelGGL(derive_tree_seeds_kernel, dim3((nodes + 63) / 64), dim3(64), 0, s, base, position, trie, nodes, out); }, "derive_tree_seeds");
}

uzu_status derive_seed(hipStream_t s, uint64_t base, const uint32_t* position, uint32_t offset, uint64_t* out) {
    return launch_check([&] { hipLaunchKernelGGL(derive_seed_kernel, dim3(1), dim3(1), 0, s, base, position, offset, out); }, "derive_seed");
}

} // namespace k
} // namespace uzu
