// xcd_lab.hip -- does it matter WHERE the activation vector of a decode launch lives?  (tools/timeline.py --detail: the workgroups of XCD 1 / 2 / 4 get
// the row 0.4-0.6 us later than those of XCD 7 / 0, launch after launch.)
// A chain of dependent launches inside a hipGraph, as in lat_lab: every workgroup reads the 2 KB vector the previous launch wrote (8 bytes per
// workgroup) and writes its 8 bytes of the next one.  Lane 0 stamps entry / vector arrived / exit and its XCC id.
//   mode "scan":  the vector pair sits at offset o of a big allocation, o over a list of candidates: median entry -> vector per XCD and offset.
//   mode "repl":  the vector exists in 8 copies (one per XCD, at the offsets the scan found best for that XCD); producers write all copies,
//                 consumers read their XCD's.
//   build: hipcc --offload-arch=gfx950 -O3 tools/xcd_lab.hip -o tools/xcd_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct Copies {
    const u32x2* in[8];
    u32x2* out[8];
    int n_out; // copies the producers write (1 or 8)
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

__global__ void __launch_bounds__(256) chain_kernel(Copies c, unsigned long long* __restrict__ tl) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const unsigned xcd = xcc_id();
    const u32x2* xin = c.in[c.n_out == 1 ? 0 : (xcd & 7)];
    u32x2 xv = xin[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned acc = xv.x + xv.y;
    asm volatile("" : "+v"(acc));
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (tid == 0) {
        u32x2 o;
        o.x = acc | 1u, o.y = blockIdx.x;
        for (int k = 0; k < c.n_out; ++k) c.out[k][blockIdx.x] = o;
        const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* slot = tl + (size_t)blockIdx.x * 4;
        slot[0] = t0, slot[1] = t1, slot[2] = t3, slot[3] = xcd;
    }
}

static hipStream_t s;
struct Result {
    double us_per_launch, xlat[8], xlat_all, span, gap;
};
static Result run(const Copies& even, const Copies& odd, int wgs, unsigned long long* tl) {
    const int reps = 48;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(chain_kernel, dim3(wgs), dim3(256), 0, s, (i & 1) ? odd : even, tl + (size_t)i * 4096);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 4; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)reps * 4096);
    CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
    Result r{};
    r.us_per_launch = ms * 1e3 / (4.0 * reps);
    std::vector<double> per[8], all;
    unsigned long long prev_end = 0;
    int ng = 0;
    for (int i = 1; i < reps; ++i) {
        unsigned long long first = ~0ull, end = 0;
        for (int b = 0; b < wgs; ++b) {
            const unsigned long long* o = &h[(size_t)i * 4096 + (size_t)b * 4];
            first = std::min(first, o[0]), end = std::max(end, o[2]);
            per[o[3] & 7].push_back((double)(o[1] - o[0]) * 0.01), all.push_back((double)(o[1] - o[0]) * 0.01);
        }
        if (i > 1) r.gap += (double)(first - prev_end) * 0.01, ++ng;
        r.span += (double)(end - first) * 0.01;
        prev_end = end;
    }
    r.gap /= ng, r.span /= (reps - 1);
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    for (int x = 0; x < 8; ++x) r.xlat[x] = med(per[x]);
    r.xlat_all = med(all);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return r;
}

int main(int argc, char** argv) {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* tl;
    CK(hipMalloc(&tl, (size_t)48 * 4096 * 8));
    const size_t MB = 1 << 20, span_bytes = 64 * MB;
    char* big;
    CK(hipMalloc(&big, span_bytes));
    CK(hipMemset(big, 1, span_bytes));
    const int wgs = 256;
    // candidate offsets: 16 consecutive 4 KB pages, then steps of 64 KB, 1 MB, 2 MB (each candidate holds the even and the odd vector 2 KB apart)
    std::vector<size_t> offs;
    for (int i = 0; i < 16; ++i) offs.push_back((size_t)i * 4096);
    for (int i = 1; i < 8; ++i) offs.push_back((size_t)i * 65536);
    for (int i = 1; i < 8; ++i) offs.push_back((size_t)i * MB);
    for (int i = 4; i < 16; ++i) offs.push_back((size_t)i * 2 * MB + 8192);
    std::vector<Result> res;
    printf("# scan: %d workgroups, vector at offset o (even launch) / o + 2 KB (odd); entry -> vector arrived, median per XCD, us\n", wgs);
    printf("# %10s  %6s %6s %6s |", "offset", "launch", "span", "gap");
    for (int x = 0; x < 8; ++x) printf("  xcd%d", x);
    printf("   all\n");
    for (size_t o : offs) {
        Copies ev{}, od{};
        ev.n_out = od.n_out = 1;
        ev.in[0] = (const u32x2*)(big + o), ev.out[0] = (u32x2*)(big + o + 2048);
        od.in[0] = (const u32x2*)(big + o + 2048), od.out[0] = (u32x2*)(big + o);
        Result r = run(ev, od, wgs, tl);
        res.push_back(r);
        printf("  %10zu  %6.2f %6.2f %6.2f |", o, r.us_per_launch, r.span, r.gap);
        for (int x = 0; x < 8; ++x) printf(" %5.2f", r.xlat[x]);
        printf("  %5.2f\n", r.xlat_all);
    }
    // best offset per XCD
    size_t best[8];
    printf("# best offset per XCD:");
    for (int x = 0; x < 8; ++x) {
        int bi = 0;
        for (size_t i = 0; i < offs.size(); ++i)
            if (res[i].xlat[x] < res[bi].xlat[x]) bi = (int)i;
        best[x] = offs[bi];
        printf(" xcd%d -> %zu (%.2f)", x, best[x], res[bi].xlat[x]);
    }
    printf("\n");
    // the best single offset by launch time
    int bl = 0;
    for (size_t i = 0; i < offs.size(); ++i)
        if (res[i].us_per_launch < res[bl].us_per_launch) bl = (int)i;
    printf("# best single offset by launch time: %zu (%.2f us/launch); worst %.2f\n", offs[bl], res[bl].us_per_launch,
           std::max_element(res.begin(), res.end(), [](const Result& a, const Result& b) { return a.us_per_launch < b.us_per_launch; })->us_per_launch);
    // replicated: copy x lives at best[x] (distinct offsets are not required: XCDs may share a copy)
    {
        Copies ev{}, od{};
        ev.n_out = od.n_out = 8;
        for (int x = 0; x < 8; ++x) {
            ev.in[x] = (const u32x2*)(big + best[x]), ev.out[x] = (u32x2*)(big + best[x] + 2048);
            od.in[x] = (const u32x2*)(big + best[x] + 2048), od.out[x] = (u32x2*)(big + best[x]);
        }
        Result r = run(ev, od, wgs, tl);
        printf("# replicated (8 copies at the per-XCD best offsets): %6.2f us/launch, span %5.2f, gap %5.2f |", r.us_per_launch, r.span, r.gap);
        for (int x = 0; x < 8; ++x) printf(" %5.2f", r.xlat[x]);
        printf("  %5.2f\n", r.xlat_all);
    }
    return 0;
}
