// valu_rate.hip -- what bounds the int4 x bf16 dot of the decode GEMV once the bytes are on chip: issue rates, per SIMD, of the
// instructions gemv_core.h::dot32p is made of, and of candidate replacements (round 3: the LDS-staged stream showed the register
// GEMV's dot phase and the LDS consumers both stop at ~24 GB/s per CU, whatever feeds them).
//   mode 0  v_fma_f32, 4 independent chains
//   mode 1  v_dot2c_f32_bf16, 4 independent chains
//   mode 2  the int4 -> bf16 pair conversion alone: (w << s) & mask | magic
//   mode 3  dot32p: one 16-byte step (32 weights per lane) = 16 conversions + 16 dot2
//   mode 4  the same 32 weights per lane through v_mfma_f32_16x16x32_bf16: 16 conversions (8 per word pair) + 4 MFMAs per 16 bytes
//   mode 5  mode 4 without the conversions (MFMA issue alone)
// For W = 1, 2, 4 waves per SIMD (blocks of 256 threads, W per CU).  Prints ns per 32-weight step per wave and the bytes of packed
// codes per second and CU that rate corresponds to.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -I uzu_amd/csrc tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "gemv_core.h"
using namespace uzu::k;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(const uint32_t* in, float* out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    Codes4 c;
    c.a = *(const uint4*)(in + (tid & 1023) * 4);
    XPack x;
    for (int i = 0; i < 16; ++i) x.v[i] = in[4096 + ((tid + i) & 1023)];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    uint32_t mask = 0x00780078u, magic = 0x41804180u;
    asm("" : "+s"(mask));
    asm("" : "+v"(magic));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { // 4 steps per iteration
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    a0 = fmaf(a0, 1.0001f, 0.5f), a1 = fmaf(a1, 1.0001f, 0.5f), a2 = fmaf(a2, 1.0001f, 0.5f), a3 = fmaf(a3, 1.0001f, 0.5f);
                }
            } else if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a0 = dot2_bf16(c.a.x, x.v[4 * k], a0), a1 = dot2_bf16(c.a.y, x.v[4 * k + 1], a1), a2 = dot2_bf16(c.a.z, x.v[4 * k + 2], a2), a3 = dot2_bf16(c.a.w, x.v[4 * k + 3], a3);
                }
            } else if (MODE == 2) {
                uint32_t acc = 0;
                const uint32_t ws[4] = {c.a.x, c.a.y, c.a.z, c.a.w};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    acc ^= ((ws[w] << 3) & mask) | magic;
                    acc ^= ((ws[w] >> 1) & mask) | magic;
                    acc ^= ((ws[w] >> 5) & mask) | magic;
                    acc ^= ((ws[w] >> 9) & mask) | magic;
                }
                c.a.x += acc; // 16 conversions + 16 xor + 1: the xors stand in for the consumer
                asm volatile("" : "+v"(c.a.x));
            } else if (MODE == 3) {
                a0 += dot32p(c, x);
                asm volatile("" : "+v"(c.a.x), "+v"(c.a.y), "+v"(c.a.z), "+v"(c.a.w));
            } else {
                const uint32_t ws[4] = {c.a.x, c.a.y, c.a.z, c.a.w};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    u32x4 av;
                    if (MODE == 4) {
                        av.x = ((ws[w] << 3) & mask) | magic, av.y = ((ws[w] >> 1) & mask) | magic, av.z = ((ws[w] >> 5) & mask) | magic, av.w = ((ws[w] >> 9) & mask) | magic;
                    } else {
                        av.x = ws[w], av.y = ws[w], av.z = ws[w], av.w = ws[w];
                    }
                    u32x4 bv = {x.v[4 * w], x.v[4 * w + 1], x.v[4 * w + 2], x.v[4 * w + 3]};
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), d, 0, 0, 0);
                }
                asm volatile("" : "+v"(c.a.x), "+v"(c.a.y), "+v"(c.a.z), "+v"(c.a.w));
            }
        }
    }
    out[tid] = a0 + a1 + a2 + a3 + d.x + d.y + d.z + d.w + (float)c.a.x;
}

template <int MODE> static void run(const char* name, const uint32_t* in, float* out, int cus) {
    for (int w : {1, 2, 4}) {
        const int iters = 4096, grid = cus * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, 64);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double steps = (double)iters * 4;                   // per wave
        const double ns_per_step = ms * 1e6 / steps;              // wall ns per 32-weight step of one wave (all waves run concurrently)
        const double bytes_per_s_cu = 4.0 * w * 64 * 16 / (ns_per_step * 1e-9); // 4 SIMDs x w waves x 64 lanes x 16 B per step
        printf("%-44s W=%d  %7.1f ns/step/wave  = %6.1f cycles @2.4GHz   -> %6.1f GB/s of int4 codes per CU\n", name, w, ns_per_step, ns_per_step * 2.4, bytes_per_s_cu / 1e9);
    }
}
int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::vector<uint32_t> h(8192);
    for (size_t i = 0; i < h.size(); ++i) h[i] = i < 4096 ? (uint32_t)(i * 2654435761u) : 0x3F803F80u;
    uint32_t* in;
    float* out;
    hipMalloc(&in, h.size() * 4), hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("v_fma_f32 x32 (4 chains)", in, out, cus);
    run<1>("v_dot2c_f32_bf16 x16 (4 chains)", in, out, cus);
    run<2>("int4->bf16 pair conversion x16 (+16 xor)", in, out, cus);
    run<3>("dot32p (16 conv + 16 dot2)", in, out, cus);
    run<4>("16 conv + 4 mfma_16x16x32_bf16", in, out, cus);
    run<5>("4 mfma_16x16x32_bf16 alone", in, out, cus);
    return 0;
}
