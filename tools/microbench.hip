// microbench.hip -- calibrates the box: kernel-boundary cost (eager vs hipGraph), streaming-copy bandwidth.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void tiny_kernel(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read_kernel(const uint4* __restrict__ a, unsigned* out, size_t n) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *out = acc;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* d; CK(hipMalloc(&d, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000; float ms;
    for (int variant = 0; variant < 3; ++variant) {
        auto launch = [&](hipStream_t st) {
            if (variant == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr);
            else if (variant == 1) hipLaunchKernelGGL(tiny_kernel, dim3(4), dim3(256), 0, st, d, 1024);
            else hipLaunchKernelGGL(tiny_kernel, dim3(1024), dim3(256), 0, st, d, 262144);
        };
        for (int i = 0; i < 100; ++i) launch(s);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); for (int i = 0; i < N; ++i) launch(s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d eager: %.3f us/kernel\n", variant, ms * 1e3 / N);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); for (int i = 0; i < 250; ++i) launch(s); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 8; ++i) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d graph(250 nodes): %.3f us/kernel\n", variant, ms * 1e3 / (8 * 250));
    }
    // bandwidth
    const size_t bytes = (size_t)1 << 30; float4 *a, *b; unsigned* o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4)); CK(hipMemset(a, 1, bytes));
    for (int blocks : {2048, 8192, 32768}) {
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, a, b, bytes / 16); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, a, b, bytes / 16);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy  %6d blocks: %.1f GB/s (read+write)\n", blocks, 2.0 * bytes * 5 / (ms * 1e-3) / 1e9);
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)a, o, bytes / 16);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("read  %6d blocks: %.1f GB/s\n", blocks, 1.0 * bytes * 5 / (ms * 1e-3) / 1e9);
    }
    // small read: 4 MB and 128 MB single launches (cold-ish: alternate two buffers)
    for (size_t sz : {(size_t)1 << 20, (size_t)4 << 20, (size_t)128 << 20}) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(read_kernel, dim3(sz / 16 / 256 > 4096 ? 4096 : sz / 16 / 256), dim3(256), 0, s, (const uint4*)((char*)a + (size_t)(i % 4) * (256 << 20)), o, sz / 16);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("read %4zu MB per launch: %.2f us/launch, %.1f GB/s\n", sz >> 20, ms * 1e3 / 20, sz * 20.0 / (ms * 1e-3) / 1e9);
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s CUs %d clock %d MHz memclk %d MHz l2 %d\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.l2CacheSize);
    return 0;
}
