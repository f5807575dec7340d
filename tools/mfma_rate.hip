// MFMA issue-rate probe (gfx950): independent v_mfma_f32_32x32x16_bf16 on 4 accumulators, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int KIND> __global__ void __launch_bounds__(256) probe(float* out, int iters) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x + i), b[i] = (__bf16)(float)(i + 1);
    f32x16_t c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4_t d0 = {}, d1 = {}, d2 = {}, d3 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            } else {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d3, 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, float* out, int cus, int wgs_per_cu, double flop_per_inst) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(probe<KIND>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0); hipLaunchKernelGGL(probe<KIND>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = 16.0 * iters * wgs_per_cu; // per SIMD
    printf("%-34s %d wave(s)/SIMD: %6.1f ns per MFMA per SIMD  -> %7.1f TFLOP/s chip\n", name, wgs_per_cu, ms * 1e6 / insts, flop_per_inst * insts * cus * 4 / (ms * 1e-3) / 1e12);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float* out; hipMalloc(&out, (size_t)p.multiProcessorCount * 4 * 256 * 4);
    printf("clock %d MHz, %d CUs\n", p.clockRate / 1000, p.multiProcessorCount);
    for (int w = 1; w <= 2; ++w) run<0>("v_mfma_f32_32x32x16_bf16", out, p.multiProcessorCount, w, 32768.0);
    for (int w = 1; w <= 2; ++w) run<1>("v_mfma_f32_16x16x32_bf16", out, p.multiProcessorCount, w, 16384.0);
    return 0;
}
