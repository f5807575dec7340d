// VALU issue-rate probe for the conversion instructions the dequant paths use (gfx950).
// One workgroup of 256 threads per CU (one wave per SIMD); each test issues 64 independent instructions per loop trip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int T> __global__ void __launch_bounds__(256) probe(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = 0x3f800000u + threadIdx.x;
    uint32_t r0 = a, r1 = b, r2 = c, r3 = a ^ b;
    uint64_t p0 = ((uint64_t)a << 32) | b, p1 = ((uint64_t)b << 32) | c;
    for (int i = 0; i < iters; ++i) {
        if (T == 0) { REP64(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(r1), "v"(r2));) }
        if (T == 1) { REP64(asm volatile("v_cvt_off_f32_i4_e32 %0, %1" : "=v"(r0) : "v"(r1));) }
        if (T == 2) { REP64(asm volatile("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r0) : "v"(r1));) }
        if (T == 3) { REP64(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r0) : "v"(r1), "v"(r2));) }
        if (T == 4) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p0) : "v"(p1));) }
        if (T == 5) { REP64(asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r0) : "v"(r1));) }
        if (T == 6) { REP64(asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r0) : "v"(r1));) }
        if (T == 7) { REP64(asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r0) : "v"(r1), "v"(r2), "v"(r3));) }
        if (T == 8) { REP64(asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r0) : "v"(r1), "v"(r2), "v"(r3));) }
        if (T == 9) { REP64(asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r0) : "v"(r1), "v"(r2));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + (uint32_t)p0;
}
template <int T> void run(const char* name, uint32_t* out, int cus) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(probe<T>, dim3(cus), dim3(256), 0, 0, out, 100, 1u);
    hipEventRecord(e0); hipLaunchKernelGGL(probe<T>, dim3(cus), dim3(256), 0, 0, out, iters, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %6.2f cycles / wave instruction (2.4 GHz assumed)\n", name, ms * 1e-3 * 2.4e9 / (64.0 * iters));
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    uint32_t* out; hipMalloc(&out, (size_t)p.multiProcessorCount * 256 * 4);
    run<0>("v_fma_f32 (dependent)", out, p.multiProcessorCount);
    run<1>("v_cvt_off_f32_i4", out, p.multiProcessorCount);
    run<2>("v_cvt_off_f32_i4 sdwa", out, p.multiProcessorCount);
    run<3>("v_cvt_pk_bf16_f32", out, p.multiProcessorCount);
    run<4>("v_pk_fma_f32 (dependent)", out, p.multiProcessorCount);
    run<5>("v_cvt_f32_ubyte1", out, p.multiProcessorCount);
    run<6>("v_cvt_f32_i32 sext sdwa", out, p.multiProcessorCount);
    run<7>("v_and_or_b32", out, p.multiProcessorCount);
    run<8>("v_perm_b32", out, p.multiProcessorCount);
    run<9>("v_xor_b32", out, p.multiProcessorCount);
    return 0;
}
