// lat_lab.hip -- anatomy of a dependent chain of small weight-streaming kernels inside a hipGraph (the decode step's
// structure): where do the ~4.5 us per launch go?  Every kernel reads a 2 KB activation vector that the PREVIOUS launch
// wrote (all workgroups read all of it), optionally streams its own 4 MB of weights (16 B per lane x 4), and writes 8
// bytes per workgroup of the next vector.  Lane 0 of every workgroup keeps 100 MHz wall-clock stamps in registers
// (entry / activation vector arrived / weights arrived / exit) and stores them at the very end.
//   build: hipcc --offload-arch=gfx950 -O3 tools/lat_lab.hip -o tools/lat_lab
//          hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=12 tools/lat_lab.hip -o tools/lat_lab_preload
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// XMODE: 0 plain loads of the vector, 1 = nontemporal.  WMODE: 0 none, 1 plain weight loads, 2 nontemporal weight loads.
template <int XMODE, int WMODE>
__global__ void __launch_bounds__(256) chain_kernel(const u32x2* __restrict__ xin, u32x2* __restrict__ xout, const u32x4* __restrict__ w,
                                                    unsigned long long* __restrict__ tl) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    u32x2 xv = XMODE ? __builtin_nontemporal_load(xin + tid) : xin[tid];
    u32x4 wv[4];
    if (WMODE) {
        const u32x4* wp = w + ((size_t)blockIdx.x * 256 + tid);
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = WMODE == 2 ? __builtin_nontemporal_load(wp + (size_t)i * gridDim.x * 256) : wp[(size_t)i * gridDim.x * 256];
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WMODE ? 4 : 0) : "memory");
    unsigned acc = xv.x + xv.y;
    asm volatile("" : "+v"(acc));
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (WMODE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += wv[i].x ^ wv[i].y ^ wv[i].z ^ wv[i].w;
        asm volatile("" : "+v"(acc));
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    // wave reduction stand-in + one 8-byte store per workgroup (the next launch's input)
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (tid == 0) {
        u32x2 o;
        o.x = acc | 1u, o.y = blockIdx.x;
        xout[blockIdx.x] = o;
        const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* slot = tl + (size_t)blockIdx.x * 4;
        slot[0] = t0, slot[1] = t1, slot[2] = t2, slot[3] = t3;
    }
}

static hipStream_t s;
template <int XMODE, int WMODE>
static void run(const char* name, int wgs, const std::vector<u32x4*>& wbufs, u32x2* xa, u32x2* xb, unsigned long long* tl) {
    const int reps = 64;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((chain_kernel<XMODE, WMODE>), dim3(wgs), dim3(256), 0, s, (i & 1) ? xb : xa, (i & 1) ? xa : xb, wbufs.empty() ? nullptr : wbufs[i % wbufs.size()],
                           tl + (size_t)i * 4096);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)reps * 4096);
    CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
    double gap = 0, ramp = 0, xlat = 0, wlat = 0, tail = 0, span = 0;
    unsigned long long prev_end = 0;
    int ng = 0;
    for (int i = 0; i < reps; ++i) {
        unsigned long long first = ~0ull, last_entry = 0, end = 0;
        std::vector<double> x, wl, tq;
        for (int b = 0; b < wgs && b < 1024; ++b) {
            const unsigned long long* o = &h[(size_t)i * 4096 + (size_t)b * 4];
            first = std::min(first, o[0]), last_entry = std::max(last_entry, o[0]), end = std::max(end, o[3]);
            x.push_back((double)(o[1] - o[0])), wl.push_back((double)(o[2] - o[0])), tq.push_back((double)(o[3] - o[2]));
        }
        std::sort(x.begin(), x.end()), std::sort(wl.begin(), wl.end()), std::sort(tq.begin(), tq.end());
        if (i > 0) gap += (double)(first - prev_end), ++ng;
        ramp += (double)(last_entry - first), xlat += x[x.size() / 2], wlat += wl[wl.size() / 2], tail += tq[tq.size() / 2], span += (double)(end - first);
        prev_end = end;
    }
    printf("%-58s %6.2f us/launch | gap %5.2f ramp %5.2f  entry->vector %5.2f  entry->weights %5.2f  tail %5.2f  span %5.2f\n", name, ms * 1e3 / (5.0 * reps), gap / ng * 0.01,
           ramp / reps * 0.01, xlat / reps * 0.01, wlat / reps * 0.01, tail / reps * 0.01, span / reps * 0.01);
}

int main() {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* tl;
    CK(hipMalloc(&tl, (size_t)64 * 4096 * 8));
    u32x2 *xa, *xb;
    CK(hipMalloc(&xa, 8192));
    CK(hipMalloc(&xb, 8192));
    CK(hipMemset(xa, 1, 8192));
    CK(hipMemset(xb, 1, 8192));
    const size_t MB = 1 << 20;
    std::vector<u32x4*> sep(96), arena(96), none;
    for (auto& p : sep) {
        CK(hipMalloc(&p, 4 * MB));
        CK(hipMemset(p, 0x11, 4 * MB));
    }
    u32x4* big;
    CK(hipMalloc(&big, 96 * 4 * MB));
    CK(hipMemset(big, 0x11, 96 * 4 * MB));
    for (int i = 0; i < 96; ++i) arena[i] = big + (size_t)i * (4 * MB / 16);
    std::vector<u32x4*> hot(1, sep[0]);
    for (int wgs : {256, 512, 1024}) {
        char nm[96];
        snprintf(nm, sizeof nm, "%4d WGs: vector only", wgs);
        run<0, 0>(nm, wgs, none, xa, xb, tl);
        snprintf(nm, sizeof nm, "%4d WGs: vector only, nt loads", wgs);
        run<1, 0>(nm, wgs, none, xa, xb, tl);
        if (wgs * 256 * 64 > 4 * (int)MB) continue; // 16 B x 4 per lane
        snprintf(nm, sizeof nm, "%4d WGs: + weights, same 4 MB every launch (cache-hot)", wgs);
        run<0, 1>(nm, wgs, hot, xa, xb, tl);
        snprintf(nm, sizeof nm, "%4d WGs: + weights, 96 separate 4 MB allocations", wgs);
        run<0, 1>(nm, wgs, sep, xa, xb, tl);
        snprintf(nm, sizeof nm, "%4d WGs: + weights, 96 x 4 MB inside one allocation", wgs);
        run<0, 1>(nm, wgs, arena, xa, xb, tl);
        snprintf(nm, sizeof nm, "%4d WGs: + nt weights, 96 separate 4 MB allocations", wgs);
        run<0, 2>(nm, wgs, sep, xa, xb, tl);
        snprintf(nm, sizeof nm, "%4d WGs: + nt weights, 96 x 4 MB inside one allocation", wgs);
        run<0, 2>(nm, wgs, arena, xa, xb, tl);
    }
    return 0;
}
