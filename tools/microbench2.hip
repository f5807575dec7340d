// microbench2.hip -- latency anatomy of a small weight-streaming kernel inside a hipGraph on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// every lane loads `chain` dependent 16-byte values (address of load j+1 depends on load j), sums, writes 4 B per wave
template <int CHAIN>
__global__ void __launch_bounds__(256) chase(const uint4* __restrict__ buf, unsigned* out, size_t n16) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < CHAIN; ++j) {
        const uint4 v = buf[idx % n16];
        acc += v.x;
        idx = idx + (size_t)gridDim.x * 256 + (v.y & 1); // data-dependent (v.y is even in our fill)
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}
static hipStream_t s;
static double run(const char* name, int reps, const std::function<void(int)>& launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-60s %7.2f us/launch\n", name, ms * 1e3 / (5.0 * reps));
    return ms;
}
int main() {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned* out; CK(hipMalloc(&out, 256));
    const size_t MB = 1 << 20;
    // (A) 16 separate 4 MB allocations vs (B) one 1 GB arena carved into 4 MB pieces
    std::vector<uint4*> sep(16);
    for (auto& p : sep) { CK(hipMalloc(&p, 4 * MB)); CK(hipMemset(p, 0, 4 * MB)); }
    uint4* arena; CK(hipMalloc(&arena, 1024 * MB)); CK(hipMemset(arena, 0, 1024 * MB));
    const size_t n16 = 4 * MB / 16;
    for (int blocks : {256, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "chain1 4MB hot (same buffer), %d blocks", blocks);
        run(nm, 64, [&](int) { hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), 0, s, sep[0], out, n16); });
        snprintf(nm, sizeof nm, "chain1 4MB rotating 16 separate allocations, %d blocks", blocks);
        run(nm, 64, [&](int i) { hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), 0, s, sep[i % 16], out, n16); });
        snprintf(nm, sizeof nm, "chain1 4MB rotating inside one 1GB arena (stride 64MB), %d blocks", blocks);
        run(nm, 64, [&](int i) { hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), 0, s, arena + (size_t)(i % 16) * (64 * MB / 16), out, n16); });
        snprintf(nm, sizeof nm, "chain2 (2 dependent loads) arena, %d blocks", blocks);
        run(nm, 64, [&](int i) { hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(256), 0, s, arena + (size_t)(i % 16) * (64 * MB / 16), out, n16); });
        snprintf(nm, sizeof nm, "chain4 (4 dependent loads) arena, %d blocks", blocks);
        run(nm, 64, [&](int i) { hipLaunchKernelGGL(chase<4>, dim3(blocks), dim3(256), 0, s, arena + (size_t)(i % 16) * (64 * MB / 16), out, n16); });
        snprintf(nm, sizeof nm, "chain4 hot buffer, %d blocks", blocks);
        run(nm, 64, [&](int) { hipLaunchKernelGGL(chase<4>, dim3(blocks), dim3(256), 0, s, sep[0], out, n16); });
    }
    return 0;
}
