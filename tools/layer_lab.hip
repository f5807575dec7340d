// layer_lab.hip -- what can a PERSISTENT decode layer buy in the bandwidth regime?  An upper bound, measured.
//
// The dependency structure and byte volumes of a Llama-3-8B int4 (or 14B-class) decode layer's MLP-centred span
//     out-proj -> [all-to-all edge: residual row] -> norm + up|gate + act -> [edge: hidden row] -> down -> [edge: residual row] -> next qkv
// with the arithmetic replaced by a trivial fold (every loaded 16-byte vector is xor-ed into an accumulator), so that NOTHING but memory
// traffic, launch boundaries and the edges costs time: whatever the persistent form gains here is the most the real kernels could gain;
// the real consumers (int4 dequant + packed dot: VALU-bound at about the HBM rate) can only do worse.
//
//   launches   : one kernel per op (what the engine does today): every workgroup reads the whole input row the previous kernel wrote
//                (plain loads), streams its share of the op's weights (512 threads, non-temporal 16-byte loads, 8 in flight per lane), writes its share
//                of the output row; the ops of L layers are captured in one hipGraph and replayed.
//   persistent : ONE launch for all ops of all layers, one workgroup of 8 waves per CU.  Edges follow MI355X_MICROARCH.md's `allgather`
//                recipe: outputs are published as 8-byte {two bf16, tag} granules with relaxed agent-scope (sc1) stores -- the data is the
//                flag, no fence --; ONE wave per CU sweeps the row, 16 sc1 dwordx2 loads in flight per lane (8 KB per pass), parks valid
//                granules in LDS and re-polls only the chunks that still hold a stale tag; every spin is bounded.  The other seven waves
//                stream: they may run up to RING bytes (the LDS-DMA ring of k_stream.hip: 128 KB per CU) AHEAD of the gather -- the
//                "prefetch credit" a run-ahead loader buys -- and stall there until the row has arrived.
// Prints us per layer for both forms, per model shape, with the run-ahead window on and off, and the in-launch edge alone.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/layer_lab.hip -o /tmp/layer_lab && /tmp/layer_lab
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Op {
    const u32x4* w;      // this op's weights (16-byte vectors), vecs of them
    uint32_t vecs;       // total 16-byte vectors
    uint32_t in_gran;    // granules (= bf16 pairs) of the input row (0: no input edge -- the span's first op reads a row a previous launch wrote)
    uint32_t out_gran;   // granules of the output row
    uint32_t in_buf, out_buf; // granule buffer indices
};
constexpr int kMaxOps = 8;
struct Span {
    Op op[kMaxOps];
    uint32_t nops, layers;
    u64* gran[4];        // granule buffers (max row each)
    uint32_t* rows[4];   // plain rows for the launch form
    uint32_t ring_vecs;  // run-ahead window per wave, in 16-byte vectors per lane-iteration units (0 = no run-ahead: stream only after the gather)
};

__device__ __forceinline__ void st_granule(u64* g, uint32_t tag, uint32_t value) { __hip_atomic_store(g, ((u64)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_granule(const u64* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// stream vectors [first, first + count) of w with stride (all lanes of the streaming waves interleave), 8 loads in flight per lane
__device__ __forceinline__ uint32_t stream_fold(const u32x4* w, uint32_t begin, uint32_t end, uint32_t stride, uint32_t acc) {
    uint32_t i = begin;
    for (; i + 7 * stride < end; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(w + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < end; i += stride) {
        const u32x4 v = __builtin_nontemporal_load(w + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    return acc;
}

__global__ void __launch_bounds__(512) persistent_kernel(Span sp, uint32_t epoch0, uint32_t* out, uint32_t* err) {
    extern __shared__ uint32_t row[]; // the gathered input row (granule data words)
    __shared__ uint32_t s_ready;      // number of gathers completed so far (monotonic over the whole launch)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, wgs = gridDim.x;
    if (tid == 0) s_ready = 0;
    __syncthreads();
    uint32_t acc = b * 2654435761u + tid, edge_no = 0;
    for (uint32_t layer = 0; layer < sp.layers; ++layer) {
        for (uint32_t oi = 0; oi < sp.nops; ++oi) {
            const Op op = sp.op[oi];
            const uint32_t tag = epoch0 + layer * kMaxOps + oi; // tag of the OUTPUT of this op; its input carries the previous op's tag
            const bool has_in = op.in_gran != 0;
            const uint32_t my_edge = edge_no + 1;
            if (wave == 0) {
                // ---- gather (allgather recipe): 16 sc1 loads in flight per lane, chunks of 1024 granules; a chunk with a stale tag is re-polled
                if (has_in) {
                    const u64* buf = sp.gran[op.in_buf];
                    const uint32_t want = tag - 1, chunks = (op.in_gran + 1023) / 1024;
                    uint32_t spins = 0;
                    for (uint32_t c = 0; c < chunks; ++c) {
                        for (;;) {
                            u64 g[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const uint32_t idx = min(c * 1024 + u * 64 + lane, op.in_gran - 1);
                                g[u] = ld_granule(buf + idx);
                            }
                            bool ok = true;
#pragma unroll
                            for (int u = 0; u < 16; ++u) ok &= (uint32_t)(g[u] >> 32) == want;
                            if (__all(ok)) {
#pragma unroll
                                for (int u = 0; u < 16; ++u) {
                                    const uint32_t idx = c * 1024 + u * 64 + lane;
                                    if (idx < op.in_gran) row[idx] = (uint32_t)g[u];
                                }
                                break;
                            }
                            if (++spins > (1u << 22)) {
                                if (lane == 0) atomicOr(err, 1u);
                                break;
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&s_ready, my_edge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                // ---- stream this op's share: workgroup b takes vectors b, b + wgs, ... in units of 448 lanes (seven streaming waves)
                const uint32_t per = (op.vecs + wgs - 1) / wgs, first = b * per, last = min(first + per, op.vecs);
                const uint32_t sl = (wave - 1) * 64 + lane;
                uint32_t begin = first + sl;
                if (has_in) {
                    // run ahead of the gather by at most ring_vecs vectors per workgroup (the LDS ring of a run-ahead loader), then wait for the row
                    const uint32_t ahead_end = min(first + sp.ring_vecs, last);
                    if (sp.ring_vecs) {
                        acc = stream_fold(op.w, begin, ahead_end, 448, acc);
                        begin += ((ahead_end > begin ? ahead_end - begin + 447 : 0) / 448) * 448;
                    }
                    uint32_t spins = 0;
                    while (__hip_atomic_load(&s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < my_edge) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) {
                            if (lane == 0) atomicOr(err, 2u);
                            break;
                        }
                    }
                    acc ^= row[(sl * 7 + b) % op.in_gran]; // the "compute" depends on the gathered row
                }
                acc = stream_fold(op.w, begin, last, 448, acc);
            }
            if (has_in) edge_no = my_edge;
            __syncthreads(); // the op is complete in this workgroup: publish its slice of the output row
            {
                u64* obuf = sp.gran[op.out_buf];
                const uint32_t per_o = (op.out_gran + wgs - 1) / wgs;
                for (uint32_t i = tid; i < per_o; i += 512) {
                    const uint32_t g = b * per_o + i;
                    if (g < op.out_gran) st_granule(obuf + g, tag, acc + g);
                }
            }
        }
    }
    if (tid == 0) out[b] = acc;
}

__global__ void __launch_bounds__(512) op_kernel(Op op, const uint32_t* row_in, uint32_t* row_out, uint32_t* accs) {
    extern __shared__ uint32_t row[];
    const uint32_t tid = threadIdx.x, b = blockIdx.x, wgs = gridDim.x;
    for (uint32_t g = tid; g < op.in_gran; g += 512) row[g] = row_in[g];
    __syncthreads();
    uint32_t acc = accs[b * 512 + tid];
    if (op.in_gran) acc ^= row[(tid * 7 + b) % op.in_gran];
    const uint32_t per = (op.vecs + wgs - 1) / wgs, first = b * per, last = min(first + per, op.vecs);
    acc = stream_fold(op.w, first + tid, last, 512, acc);
    const uint32_t per_o = (op.out_gran + wgs - 1) / wgs;
    for (uint32_t i = tid; i < per_o; i += 512) {
        const uint32_t g = b * per_o + i;
        if (g < op.out_gran) row_out[g] = acc + g;
    }
    accs[b * 512 + tid] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Shape {
        const char* name;
        uint32_t d, hidden, qkv_n; // model_dim, ffn hidden, qkv rows
    } shapes[] = {{"llama-3-8b int4", 4096, 14336, 6144}, {"qwen3-14b-class int4", 5120, 17408, 7168}, {"qwen3.5-0.8b int4 (DeltaNet layer shapes)", 1024, 3584, 8224}};
    const uint32_t layers = 8;
    for (const Shape& sh : shapes) {
        // weights at 0.53125 bytes per weight (int4 codes + bf16 scale and bias per group of 128), a fresh region per layer so that nothing is served from a cache
        const double bpw = 0.53125;
        const size_t wb[4] = {(size_t)(sh.d * (double)sh.d * bpw), (size_t)(2.0 * sh.hidden * sh.d * bpw), (size_t)((double)sh.d * sh.hidden * bpw), (size_t)((double)sh.qkv_n * sh.d * bpw)};
        const uint32_t out_elems[4] = {sh.d, sh.hidden, sh.d, sh.qkv_n};
        size_t layer_bytes = 0;
        for (int i = 0; i < 4; ++i) layer_bytes += (wb[i] + 15) / 16 * 16;
        uint8_t* weights;
        CK(hipMalloc(&weights, layer_bytes * layers));
        CK(hipMemset(weights, 1, layer_bytes * layers));
        Span sp{};
        sp.nops = 4, sp.layers = 1;
        const uint32_t max_gran = (sh.hidden > sh.qkv_n ? sh.hidden : sh.qkv_n) / 2;
        for (int i = 0; i < 4; ++i) {
            CK(hipMalloc(&sp.gran[i], (size_t)max_gran * 8));
            CK(hipMemset(sp.gran[i], 0, (size_t)max_gran * 8));
            CK(hipMalloc(&sp.rows[i], (size_t)max_gran * 4));
            CK(hipMemset(sp.rows[i], 0, (size_t)max_gran * 4));
        }
        uint32_t *out, *err, *accs;
        CK(hipMalloc(&out, cus * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&accs, (size_t)cus * 512 * 4));
        CK(hipMemset(err, 0, 4)); CK(hipMemset(accs, 0, (size_t)cus * 512 * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto build_ops = [&](uint32_t layer, Span& spn) {
            size_t off = layer_bytes * layer;
            for (int i = 0; i < 4; ++i) {
                spn.op[i].w = (const u32x4*)(weights + off);
                spn.op[i].vecs = (uint32_t)(wb[i] / 16);
                off += (wb[i] + 15) / 16 * 16;
                // op 0 (out-proj) reads the attention row a previous launch wrote: no in-launch edge in front of it
                spn.op[i].in_gran = i == 0 ? 0 : out_elems[i - 1] / 2;
                spn.op[i].out_gran = out_elems[i] / 2;
                spn.op[i].in_buf = (i + 3) % 4, spn.op[i].out_buf = i;
            }
        };
        uint32_t epoch = 16;
        auto run_persistent = [&](uint32_t ring_bytes) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                // one launch per layer-span (attention sits between spans as its own launches in both forms): `layers` launches in a graph-free loop
                CK(hipEventRecord(e0, s));
                for (uint32_t l = 0; l < layers; ++l) {
                    Span spl = sp;
                    build_ops(l, spl);
                    spl.ring_vecs = ring_bytes / 16;
                    hipLaunchKernelGGL(persistent_kernel, dim3(cus), dim3(512), max_gran * 4, s, spl, epoch, out, err);
                    epoch += 64;
                }
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            return best * 1e3f / layers;
        };
        // launches: 4 kernels per layer in one graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (uint32_t l = 0; l < layers; ++l) {
            Span spl = sp;
            build_ops(l, spl);
            for (int i = 0; i < 4; ++i)
                hipLaunchKernelGGL(op_kernel, dim3(cus), dim3(512), max_gran * 4, s, spl.op[i], sp.rows[(i + 3) % 4], sp.rows[i], accs);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        float best_l = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_l) best_l = ms;
        }
        best_l = best_l * 1e3f / layers;
        // the same graph form for the persistent spans (one launch per layer, replayed): launch overhead out of the picture
        auto run_persistent_graph = [&](uint32_t ring_bytes) {
            hipGraph_t pg;
            hipGraphExec_t pge;
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) { // tags must be fresh per replay: re-capture with a new epoch base (capture cost is outside the timed region)
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (uint32_t l = 0; l < layers; ++l) {
                    Span spl = sp;
                    build_ops(l, spl);
                    spl.ring_vecs = ring_bytes / 16;
                    hipLaunchKernelGGL(persistent_kernel, dim3(cus), dim3(512), max_gran * 4, s, spl, epoch, out, err);
                    epoch += 64;
                }
                CK(hipStreamEndCapture(s, &pg));
                CK(hipGraphInstantiate(&pge, pg, nullptr, nullptr, 0));
                CK(hipEventRecord(e0, s));
                CK(hipGraphLaunch(pge, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipGraphExecDestroy(pge)); CK(hipGraphDestroy(pg));
            }
            return best * 1e3f / layers;
        };
        const float p_eager_ring = run_persistent(128 * 1024), p_graph_ring = run_persistent_graph(128 * 1024), p_graph_noring = run_persistent_graph(0);
        uint32_t herr = 0;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        const double mb = (double)layer_bytes / 1e6;
        printf("%-44s span of %6.1f MB (out-proj, up|gate, down, next qkv): 4 launches %6.2f us (%4.2f TB/s)   persistent, 128 KB run-ahead %6.2f us (%4.2f TB/s; eager %6.2f)   "
               "persistent, no run-ahead %6.2f us   ratio %.3f%s\n",
               sh.name, mb, best_l, mb / best_l, p_graph_ring, mb / p_graph_ring, p_eager_ring, p_graph_noring, p_graph_ring / best_l,
               herr ? "   A BOUNDED SPIN GAVE UP" : "");
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        for (int i = 0; i < 4; ++i) { CK(hipFree(sp.gran[i])); CK(hipFree(sp.rows[i])); }
        CK(hipFree(weights)); CK(hipFree(out)); CK(hipFree(err)); CK(hipFree(accs));
    }
    return 0;
}
