// edge_lab.hip -- what would a persistent decode layer buy for THIS model?  The deciding quantity, measured: the cost of one
// all-to-all dependency edge (every workgroup produces a slice of an activation row, every workgroup needs the whole row before its next
// op) kept INSIDE one launch, against the same edge as a kernel boundary in a replayed hipGraph.
//
//   in-launch : 256 workgroups (one per CU), persistent over E ops.  Op e: a workgroup "computes" its slice (a few dependent FMAs on the
//               gathered row, so the edge cannot be hoisted), publishes it as 8-byte {epoch, value} granules (one relaxed agent-scope
//               store each: cdna_hip_programming.md G16 recipe R2 -- the data is the flag, no fence), then one wave sweeps all granules
//               of the row with relaxed agent-scope loads until every tag carries the epoch, parks the row in LDS, barrier, next op.
//               Two row buffers alternate (an op's row is read while the next one is written); every spin is bounded.
//   launches  : the same op as its own kernel, E of them captured in one graph: plain loads of the row written by the previous kernel,
//               same dummy compute, plain stores.
// Rows: 1024 bf16 (Qwen3.5-0.8B residual / gated-attention edges: 512 granules), 3584 (the MLP's gated row), 8224 (DeltaNet in-proj).
// Prints us per edge for both forms.  A persistent layer replaces 5-6 boundaries per layer by 5-6 of these edges; what it can gain on top
// is the weight prefetch across the edge, which for this model is 1-4 MB per op = 0.2-0.7 us of streaming.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/edge_lab.hip -o /tmp/edge_lab && /tmp/edge_lab
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void st_granule(u64* g, uint32_t epoch, uint32_t value) { __hip_atomic_store(g, ((u64)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_granule(const u64* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// granules: [2][n_gran] u64 (zeroed before the launch); n_gran = row elements / 2; workgroup b owns granules [b * per, (b + 1) * per)
__global__ void __launch_bounds__(256) persistent_kernel(u64* granules, uint32_t n_gran, uint32_t ops, float* out, uint32_t* err) {
    extern __shared__ uint32_t row[]; // n_gran words
    const uint32_t tid = threadIdx.x, wgs = gridDim.x, b = blockIdx.x;
    const uint32_t per = (n_gran + wgs - 1) / wgs;
    float acc = (float)b;
    for (uint32_t op = 0; op < ops; ++op) {
        const uint32_t epoch = op + 1;
        u64* buf = granules + (size_t)(op & 1) * n_gran;
        // "compute": depends on the whole previous row (a few words of it per thread) -- cannot start before the gather
        if (op) acc = fmaf(acc, 0.999f, __uint_as_float(row[(tid * 7 + b) % n_gran] & 0x3FFFFFFFu) * 1e-30f);
        // publish this workgroup's slice
        for (uint32_t i = tid; i < per; i += 256) {
            const uint32_t g = b * per + i;
            if (g < n_gran) st_granule(buf + g, epoch, __float_as_uint(acc) + g);
        }
        // gather: wave 0 sweeps until every tag carries the epoch
        if (tid < 64) {
            uint32_t spins = 0;
            for (;;) {
                bool ok = true;
                for (uint32_t g = tid; g < n_gran; g += 64) {
                    const u64 x = ld_granule(buf + g);
                    ok &= (uint32_t)(x >> 32) == epoch;
                    row[g] = (uint32_t)x;
                }
                if (__all(ok)) break;
                if (++spins > (1u << 20)) {
                    if (tid == 0) atomicOr(err, 1u);
                    break;
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) out[b] = acc;
}

__global__ void __launch_bounds__(256) launch_kernel(const uint32_t* row_in, uint32_t* row_out, uint32_t n_gran, float* accs, uint32_t op) {
    extern __shared__ uint32_t row[];
    const uint32_t tid = threadIdx.x, wgs = gridDim.x, b = blockIdx.x;
    const uint32_t per = (n_gran + wgs - 1) / wgs;
    for (uint32_t g = tid; g < n_gran; g += 256) row[g] = row_in[g];
    __syncthreads();
    float acc = accs[b];
    if (op) acc = fmaf(acc, 0.999f, __uint_as_float(row[(tid * 7 + b) % n_gran] & 0x3FFFFFFFu) * 1e-30f);
    for (uint32_t i = tid; i < per; i += 256) {
        const uint32_t g = b * per + i;
        if (g < n_gran) row_out[g] = __float_as_uint(acc) + g;
    }
    if (tid == 0) accs[b] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const uint32_t ops = 120;
    for (uint32_t elems : {1024u, 3584u, 8224u}) {
        const uint32_t n_gran = elems / 2;
        u64* gran;
        float *out, *accs;
        uint32_t *err, *rows;
        CK(hipMalloc(&gran, (size_t)2 * n_gran * 8));
        CK(hipMalloc(&out, cus * 4)); CK(hipMalloc(&accs, cus * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&rows, (size_t)2 * n_gran * 4));
        CK(hipMemset(err, 0, 4)); CK(hipMemset(accs, 0, cus * 4)); CK(hipMemset(rows, 0, (size_t)2 * n_gran * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best_p = 1e9f, best_l = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(gran, 0, (size_t)2 * n_gran * 8, s)); // tags must start below every epoch
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(persistent_kernel, dim3(cus), dim3(256), n_gran * 4, s, gran, n_gran, ops, out, err);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_p) best_p = ms;
        }
        uint32_t herr = 0;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        // the same chain as `ops` launches in one graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (uint32_t op = 0; op < ops; ++op)
            hipLaunchKernelGGL(launch_kernel, dim3(cus), dim3(256), n_gran * 4, s, rows + (size_t)((op + 1) & 1) * n_gran, rows + (size_t)(op & 1) * n_gran, n_gran, accs, op);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_l) best_l = ms;
        }
        printf("row of %5u bf16 (%4u granules, %5.1f KB swept per workgroup): in-launch edge %6.2f us   kernel boundary edge %6.2f us   (%u ops, %d workgroups%s)\n", elems, n_gran,
               n_gran * 8 / 1024.0, best_p * 1e3 / ops, best_l * 1e3 / ops, ops, cus, herr ? "; A BOUNDED SPIN GAVE UP" : "");
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        CK(hipFree(gran)); CK(hipFree(out)); CK(hipFree(accs)); CK(hipFree(err)); CK(hipFree(rows));
    }
    return 0;
}
