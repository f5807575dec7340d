// k_probe.hip -- measurement probes bench.py runs beside the timed region (not part of the forward path).
//
// uzu_hip_probe_edge_floor: the price of ONE all-to-all dependency edge expressed as a kernel boundary inside a replayed hipGraph -- the
// structure of the batch-1 decode step: every workgroup of launch i reads the WHOLE activation row launch i - 1 wrote
// (`row_bytes`, all workgroups contributed to it) and writes its 8-byte share of the next row; nothing else.  A chain of `launches` such
// kernels over `workgroups` workgroups is captured once and replayed; the figure is wall time per launch between two events on the stream.
// What it prices: boundary (~1.0-1.6 us) + dispatch ramp + the row's trip from the memory side + one wave reduction + the store -- the part
// of a decode launch no kernel can stream behind (tools/lat_lab.hip is the stand-alone form with per-workgroup stamps).  bench.py reports
// `roofline.latency_floor` from it: edges per token x this + the read-out's stream time.
#include "device_utils.h"
#include "internal.h"

namespace uzu {
namespace {

typedef unsigned int u32x2_p __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) edge_chain_kernel(const u32x2_p* __restrict__ xin, u32x2_p* __restrict__ xout, uint32_t row_pairs) {
    unsigned acc = 0;
    for (uint32_t i = threadIdx.x; i < row_pairs; i += 256) {
        const u32x2_p v = xin[i];
        acc += v.x + v.y;
    }
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ unsigned s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32x2_p o;
        o.x = (s_part[0] + s_part[1] + s_part[2] + s_part[3]) | 1u, o.y = blockIdx.x;
        // every workgroup owns row_pairs / gridDim.x pairs of the next row (at least one): the row is rewritten completely by every launch
        const uint32_t per = row_pairs / gridDim.x ? row_pairs / gridDim.x : 1;
        for (uint32_t j = 0; j < per; ++j) {
            const uint32_t at = blockIdx.x * per + j;
            if (at < row_pairs) xout[at] = o;
        }
    }
}

} // namespace
} // namespace uzu

extern "C" uzu_status uzu_hip_probe_edge_floor(uzu_hip_context* ctx, uint32_t workgroups, uint32_t row_bytes, uint32_t launches, uint32_t replays, float* us_per_launch) {
    using namespace uzu;
    UZU_REQUIRE(ctx && us_per_launch && workgroups > 0 && workgroups <= 4096 && row_bytes >= 8 && row_bytes % 8 == 0 && row_bytes <= (1u << 20) && launches >= 2 && launches <= 1024 && replays > 0,
                "probe_edge_floor: bad arguments");
    UZU_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    void *a = nullptr, *b = nullptr;
    UZU_HIP_TRY(hipMalloc(&a, row_bytes));
    UZU_HIP_TRY(hipMalloc(&b, row_bytes));
    UZU_HIP_TRY(hipMemsetAsync(a, 1, row_bytes, s));
    UZU_HIP_TRY(hipMemsetAsync(b, 1, row_bytes, s));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uzu_status st = UZU_OK;
    auto fail = [&](const char* what, hipError_t e) {
        set_error("probe_edge_floor: %s: %s", what, hipGetErrorString(e));
        st = UZU_ERR_HIP;
    };
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) fail("begin capture", e);
    if (st == UZU_OK) {
        for (uint32_t i = 0; i < launches; ++i)
            hipLaunchKernelGGL(edge_chain_kernel, dim3(workgroups), dim3(256), 0, s, (const u32x2_p*)((i & 1) ? b : a), (u32x2_p*)((i & 1) ? a : b), row_bytes / 8);
        e = hipStreamEndCapture(s, &g);
        if (e != hipSuccess) fail("end capture", e);
    }
    if (st == UZU_OK && (e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)) != hipSuccess) fail("instantiate", e);
    if (st == UZU_OK && ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess)) fail("event", e);
    if (st == UZU_OK) {
        (void)hipGraphLaunch(ge, s); // warm-up replay
        (void)hipEventRecord(e0, s);
        for (uint32_t r = 0; r < replays; ++r) (void)hipGraphLaunch(ge, s);
        (void)hipEventRecord(e1, s);
        e = hipEventSynchronize(e1);
        if (e != hipSuccess) fail("replay", e);
        else {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            *us_per_launch = ms * 1e3f / ((float)launches * (float)replays);
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    (void)hipFree(a);
    (void)hipFree(b);
    return st;
}
