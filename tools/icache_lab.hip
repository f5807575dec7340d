// icache_lab.hip -- does a kernel that follows a big weight stream start cold?  The Llama-3-8B decode timeline shows kernel boundaries of
// 2.2-3.5 us and prologues of 3-4 us where the 0.8B model (whose layer fits the L2s) shows 1.0-1.6 and 1.0-2.4: the hypothesis is that
// the streamed weights evict the NEXT kernel's code (22-29 KB per decode kernel) from the XCDs' L2s, so its first waves fetch
// instructions from HBM.  Here: S(X MB, nt loads) -> B<i>, alternating two instances of a kernel with 8 / 24 KB of straight-line code;
// reported per X: boundary (first entry of B - last exit of S) and B's body (entry -> exit, median over workgroups), without and with S's
// workgroups 0-7 (one per XCD) touching B's code with plain loads three quarters of the way through their stream.
//   build: hipcc --offload-arch=gfx950 -O3 tools/icache_lab.hip -o tools/icache_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// stamps: [launch][wg][2] = entry, exit
__global__ void __launch_bounds__(256) stream_kernel(const u32x4* __restrict__ w, unsigned iters, unsigned* __restrict__ sink, u64* __restrict__ tl, const u64* __restrict__ next_pc,
                                                     unsigned prefetch_bytes) {
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    __shared__ u32x4 dump[256];
    const unsigned tid = threadIdx.x;
    const unsigned shift = (prefetch_bytes & 0x80000000u) ? (prefetch_bytes & 0xFFu) : 0u;
    const bool plain = (prefetch_bytes & 0x40000000u) != 0;
    if (prefetch_bytes & 0xC0000000u) prefetch_bytes = 0;
    const u32x4* wp = w + ((size_t)((blockIdx.x + shift) % gridDim.x) * 256 + tid);
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    const unsigned when = iters - (iters >> 2) - 1;
    for (unsigned i = 0; i < iters; ++i) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = plain ? wp[((size_t)i * 4 + u) * stride] : __builtin_nontemporal_load(wp + ((size_t)i * 4 + u) * stride);
        if (prefetch_bytes && blockIdx.x < 8 && i == when) {
            const char* code = (const char*)(*(volatile const u64*)next_pc & ~(u64)255);
            if (code)
                for (unsigned off = tid * 16; off < prefetch_bytes; off += 4096) { // into a dump area of the LDS: no destination registers, nothing waits; only the L2 fill matters
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(code + off), "s"(__builtin_amdgcn_readfirstlane((unsigned)(size_t)dump + (tid >> 6) * 1024u)) : "memory");
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
    if (tid == 0) {
        u64* slot = tl + (size_t)blockIdx.x * 2;
        slot[0] = t0, slot[1] = __builtin_amdgcn_s_memrealtime();
    }
}

// N VOP3 instructions (8 bytes each) in four independent chains; ID only makes two distinct functions
template <int ID, int N>
__global__ void __launch_bounds__(256) code_kernel(unsigned* __restrict__ out, u64* __restrict__ tl, u64* __restrict__ self_pc, unsigned seed) {
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u64 pc;
        asm volatile("s_getpc_b64 %0" : "=s"(pc));
        *self_pc = pc;
    }
    unsigned a = threadIdx.x + seed, b = a * 3 + ID, c = a ^ 0x55, d = a + 7;
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a) : "v"(seed));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(b) : "v"(seed));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(c) : "v"(seed));
        asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(d) : "v"(seed));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0) {
        u64* slot = tl + (size_t)blockIdx.x * 2;
        slot[0] = t0, slot[1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int N>
static void run(hipStream_t s, size_t mb, bool prefetch, bool hot, const u32x4* w, size_t wbytes, unsigned* sink, unsigned* out, u64* tl, u64* pcs) {
    const int reps = 24, wgs = 256;
    const unsigned iters = (unsigned)(mb * 1024 * 1024 / ((size_t)wgs * 256 * 16 * 4));
    const size_t slab = (size_t)iters * wgs * 256 * 16 * 4;
    const size_t nslabs = std::max<size_t>(1, wbytes / std::max<size_t>(slab, 1));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int launch = 0;
    for (int i = 0; i < reps; ++i) {
        if (!hot) {
            const u32x4* wp = (const u32x4*)((const char*)w + ((size_t)i % nslabs) * slab);
            hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(256), 0, s, wp, iters, sink, tl + (size_t)launch * wgs * 2, pcs + ((i & 1) ? 1 : 0), prefetch ? (unsigned)(N * 8 + 512) : 0u);
            ++launch;
        }
        if (i & 1)
            hipLaunchKernelGGL((code_kernel<1, N>), dim3(wgs), dim3(256), 0, s, out, tl + (size_t)launch * wgs * 2, pcs + 1, 3u);
        else
            hipLaunchKernelGGL((code_kernel<0, N>), dim3(wgs), dim3(256), 0, s, out, tl + (size_t)launch * wgs * 2, pcs + (hot ? 0 : 0), 3u);
        ++launch;
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int k = 0; k < 3; ++k) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    std::vector<u64> h((size_t)launch * wgs * 2);
    CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> gaps, bodies, sspans;
    for (int l = 1; l < launch; ++l) {
        const bool is_code = hot || (l & 1);
        u64 first = ~0ull, last = 0, plast = 0;
        std::vector<double> body;
        for (int wgi = 0; wgi < wgs; ++wgi) {
            const u64 e = h[((size_t)l * wgs + wgi) * 2], x = h[((size_t)l * wgs + wgi) * 2 + 1];
            first = std::min(first, e), last = std::max(last, x);
            plast = std::max(plast, h[((size_t)(l - 1) * wgs + wgi) * 2 + 1]);
            body.push_back((double)(x - e) * 0.01);
        }
        std::sort(body.begin(), body.end());
        if (is_code) {
            gaps.push_back((double)((long long)first - (long long)plast) * 0.01);
            bodies.push_back(body[body.size() / 2]);
        } else {
            sspans.push_back((double)(last - first) * 0.01);
        }
    }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("code %5d B  stream %4zu MB  %-9s  boundary before the code kernel %5.2f us   code kernel body (median wg) %5.2f us   stream span %6.2f us\n", N * 8, hot ? 0 : mb,
           hot ? "hot" : (prefetch ? "prefetch" : "plain"), med(gaps), med(bodies), med(sspans));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
}

// Does a weight stream run faster out of the Infinity Cache?  P reads a slab with the block order REVERSED (so that every line passes through another
// XCD's L2 than the one S will use), then S streams the same slab; slabs rotate over 1 GB, so without P the slab comes from HBM.
static void run_mall(hipStream_t s, size_t mb, int mode, const u32x4* w, size_t wbytes, unsigned* sink, u64* tl, u64* pcs) {
    const int reps = 16, wgs = 256;
    const unsigned iters = (unsigned)(mb * 1024 * 1024 / ((size_t)wgs * 256 * 16 * 4));
    const size_t slab = (size_t)iters * wgs * 256 * 16 * 4;
    const size_t nslabs = std::max<size_t>(1, wbytes / slab);
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int launch = 0;
    for (int i = 0; i < reps; ++i) {
        const char* base = (const char*)w + ((size_t)i % nslabs) * slab;
        if (mode == 1 || mode == 3) { // first pass: same bytes, shifted by 3 blocks => other XCDs; plain (allocating) loads
            hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(256), 0, s, (const u32x4*)base, iters, sink, tl + (size_t)launch * wgs * 2, pcs, 0xC0000003u);
            ++launch;
        }
        if (mode == 2 || mode == 4) { // same mapping twice, plain loads: own L2 + Infinity Cache
            hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(256), 0, s, (const u32x4*)base, iters, sink, tl + (size_t)launch * wgs * 2, pcs, 0x40000000u);
            ++launch;
        }
        hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(256), 0, s, (const u32x4*)base, iters, sink, tl + (size_t)launch * wgs * 2, pcs, mode >= 3 || mode == 5 ? 0x40000000u : 0u);
        ++launch;
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int k = 0; k < 3; ++k) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    std::vector<u64> h((size_t)launch * wgs * 2);
    CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> first_pass, second_pass;
    for (int l = 0; l < launch; ++l) {
        u64 first = ~0ull, last = 0;
        for (int wgi = 0; wgi < wgs; ++wgi) first = std::min(first, h[((size_t)l * wgs + wgi) * 2]), last = std::max(last, h[((size_t)l * wgs + wgi) * 2 + 1]);
        const double span = (double)(last - first) * 0.01;
        if (mode == 0 || mode == 5 || (l & 1)) second_pass.push_back(span);
        else first_pass.push_back(span);
    }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double t2 = med(second_pass);
    printf("stream %4zu MB  %-44s first pass %6.2f us   measured pass %6.2f us = %5.2f TB/s\n", mb,
           mode == 0 ? "nt, from HBM (rotating over 1 GB)" : mode == 5 ? "plain, from HBM" : mode == 1 ? "nt, after a plain pass via OTHER XCDs" : mode == 2 ? "nt, after the same plain pass" :
           mode == 3 ? "plain, after a plain pass via OTHER XCDs" : "plain, after the same plain pass", med(first_pass), t2,
           (double)slab / t2 * 1e-6);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t wbytes = (size_t)1 << 30;
    u32x4* w;
    unsigned *sink, *out;
    u64 *tl, *pcs;
    CK(hipMalloc(&w, wbytes));
    CK(hipMemset(w, 1, wbytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&out, 256 * 256 * 4));
    CK(hipMalloc(&tl, (size_t)64 * 256 * 2 * 8));
    CK(hipMalloc(&pcs, 64));
    CK(hipMemset(pcs, 0, 64));
    for (size_t mb : {(size_t)16, (size_t)64, (size_t)128})
        for (int mode = 0; mode < 6; ++mode) run_mall(s, mb, mode, w, wbytes, sink, tl, pcs);
    if (getenv("MALL_ONLY")) return 0;
    for (size_t mb : {(size_t)0}) {
        run<1024>(s, mb, false, true, w, wbytes, sink, out, tl, pcs);
        run<3072>(s, mb, false, true, w, wbytes, sink, out, tl, pcs);
    }
    for (size_t mb : {(size_t)4, (size_t)16, (size_t)64, (size_t)128}) {
        for (int pf = 0; pf < 2; ++pf) {
            run<1024>(s, mb, pf, false, w, wbytes, sink, out, tl, pcs);
            run<3072>(s, mb, pf, false, w, wbytes, sink, out, tl, pcs);
        }
    }
    return 0;
}
