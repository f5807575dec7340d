// dot2_probe.hip -- what v_dot2c_f32_bf16 computes, bit for bit (the decode GEMV's int4 path runs on it: gemv_core.h).
// Compares the instruction against three candidate semantics on random operands:
//   fused   : round_f32(a0*b0 + a1*b1 + c)              (one rounding, exact inner sum)
//   seq_lo  : round_f32(round_f32(c + a0*b0) + a1*b1)   (two fmas, low pair first)
//   seq_hi  : round_f32(round_f32(c + a1*b1) + a0*b0)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/dot2_probe.hip -o /tmp/dot2_probe && /tmp/dot2_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
__global__ void probe(const uint32_t* a, const uint32_t* b, const float* c, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_v, a[i]), __builtin_bit_cast(bf16x2_v, b[i]), c[i], false);
}
static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int n = 1 << 20;
    std::mt19937 rng(7);
    std::vector<uint32_t> a(n), b(n);
    std::vector<float> c(n), out(n);
    auto rnd_bf16 = [&](int mode) -> uint16_t {
        if (mode == 0) { // code-like: 16 + q
            return (uint16_t)(0x4180 | ((rng() & 15) << 3));
        }
        float f = std::ldexp((float)((int)(rng() % 511) - 255) / 128.0f, (int)(rng() % 24) - 16); // 8 significant bits, wide exponent range
        uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16);
    };
    for (int i = 0; i < n; ++i) {
        a[i] = rnd_bf16(0) | ((uint32_t)rnd_bf16(0) << 16);
        b[i] = rnd_bf16(1) | ((uint32_t)rnd_bf16(1) << 16);
        c[i] = std::ldexp((float)((int)(rng() % 2000001) - 1000000) / 1000000.0f, (int)(rng() % 20) - 8);
    }
    // a few denormal accumulators / products
    for (int i = 0; i < 64; ++i) { c[i] = std::ldexp(1.0f, -140 + (i % 8)); b[i] = (b[i] & 0xFFFF0000u) | 0x0040u; /* bf16 denormal low half */ }
    uint32_t *da, *db; float *dc, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
    hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
    long fused = 0, seq_lo = 0, seq_hi = 0, none = 0, denorm_zero = 0;
    double max_rel = 0;
    for (int i = 0; i < n; ++i) {
        const double p0 = (double)bf(a[i] & 0xFFFF) * bf(b[i] & 0xFFFF), p1 = (double)bf(a[i] >> 16) * bf(b[i] >> 16);
        const float f = (float)(p0 + p1 + (double)c[i]);
        const float s0 = (float)((double)(float)((double)c[i] + p0) + p1);
        const float s1 = (float)((double)(float)((double)c[i] + p1) + p0);
        const bool mf = out[i] == f, m0 = out[i] == s0, m1 = out[i] == s1;
        fused += mf, seq_lo += m0, seq_hi += m1;
        if (!mf && !m0 && !m1) {
            ++none;
            if (i < 64 && out[i] == 0.0f) ++denorm_zero;
            if (f != 0) max_rel = fmax(max_rel, fabs(((double)out[i] - f) / f));
            if (none <= 5) printf("  no match at %d: got %a fused %a seq_lo %a seq_hi %a (c %a p0 %a p1 %a)\n", i, out[i], f, s0, s1, c[i], p0, p1);
        }
    }
    printf("v_dot2c_f32_bf16 over %d random operand sets: == fused %ld, == seq_lo %ld, == seq_hi %ld, none %ld (of which denormal inputs flushed to 0: %ld), max rel deviation from fused %.3g\n",
           n, fused, seq_lo, seq_hi, none, denorm_zero, max_rel);
    return 0;
}
