// xadd_test.hip -- checks the DPP / readlane reduction primitives of gemv_core.h against plain __shfl_xor.
// build: hipcc --offload-arch=gfx950 -O3 -I uzu_amd/csrc tools/xadd_test.hip -o tools/xadd_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include "gemv_core.h"
using namespace uzu::k;
__global__ void kern(const float* in, float* out) {
    const int l = threadIdx.x;
    const float v = in[l];
    out[0 * 64 + l] = xadd1(v);
    out[1 * 64 + l] = xadd2(v);
    out[2 * 64 + l] = xadd4(v);
    out[3 * 64 + l] = xadd8(v);
    out[4 * 64 + l] = xadd16(v);
    out[5 * 64 + l] = row_sum_rt(v, 32);
    out[6 * 64 + l] = row_sum_rt(v, 64);
    out[7 * 64 + l] = row_sum_rt(v, 16);
}
int main() {
    float h[64], *d, *o, r[8 * 64];
    for (int i = 0; i < 64; ++i) h[i] = (float)(i * i % 37) + 0.25f * i;
    hipMalloc(&d, 256); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    const int offs[5] = {1, 2, 4, 8, 16};
    // xaddN after the lower levels are NOT applied here: the mirror patterns equal xor only for pre-reduced data, so
    // check the single-level primitives on data that is constant inside each group of N lanes
    float g[64];
    for (int t = 0; t < 5; ++t) {
        for (int i = 0; i < 64; ++i) g[i] = h[(i / offs[t]) * offs[t]];
        hipMemcpy(d, g, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, o);
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; ++i) {
            const float want = g[i] + g[i ^ offs[t]];
            if (r[t * 64 + i] != want) { if (bad < 8) printf("xadd%d lane %d: got %g want %g\n", offs[t], i, r[t * 64 + i], want); ++bad; }
        }
    }
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    const int widths[3] = {32, 64, 16};
    for (int t = 0; t < 3; ++t)
        for (int i = 0; i < 64; ++i) {
            float v[64];
            for (int j = 0; j < 64; ++j) v[j] = h[j];
            for (int off = 1; off < widths[t]; off <<= 1) { float n[64]; for (int j = 0; j < 64; ++j) n[j] = v[j] + v[j ^ off]; for (int j = 0; j < 64; ++j) v[j] = n[j]; }
            if (r[(5 + t) * 64 + i] != v[i]) { if (bad < 16) printf("row_sum %d lane %d: got %g want %g\n", widths[t], i, r[(5 + t) * 64 + i], v[i]); ++bad; }
        }
    printf(bad ? "FAILED %d\n" : "all reductions match the xor butterfly\n", bad);
    return bad != 0;
}
