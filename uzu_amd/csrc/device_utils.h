// device_utils.h -- gfx950 device-side helpers: typed load/store, wave64 and workgroup reductions.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "gemv_core.h"
#include "internal.h"
#include "uzu_math.h"

namespace uzu {

struct bf16_t {
    uint16_t v;
};

template <class T> __device__ __forceinline__ float ld(const T* p, size_t i);
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p, size_t i) { return bf16_to_f32(p[i].v); }
template <> __device__ __forceinline__ float ld<float>(const float* p, size_t i) { return p[i]; }

template <class T> __device__ __forceinline__ void st(T* p, size_t i, float v);
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, size_t i, float v) { p[i].v = f32_to_bf16(v); }
template <> __device__ __forceinline__ void st<float>(float* p, size_t i, float v) { p[i] = v; }

// value of `T::from(x)` kept in an f32 register
template <class T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return round_bf16(v); }
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }

// run-time typed element access (bf16 / f32 selected by a DataType value)
__device__ __forceinline__ float ldt(const void* p, uint32_t dt, size_t i) {
    return dt == UZU_F32 ? ((const float*)p)[i] : bf16_to_f32(((const uint16_t*)p)[i]);
}
__device__ __forceinline__ void stt(void* p, uint32_t dt, size_t i, float v) {
    if (dt == UZU_F32)
        ((float*)p)[i] = v;
    else
        ((uint16_t*)p)[i] = f32_to_bf16(v);
}

constexpr int kWave = 64; // CDNA wavefront

// Native vector types for values that are HELD IN REGISTERS across other code (software-pipeline stages): HIP's
// uint4 / float4 are classes, an assignment from memory becomes llvm.memcpy into a private-memory temporary that SROA
// does not always dissolve -- the "registers" then live in scratch and every prefetch is waited for immediately.
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_v __attribute__((ext_vector_type(2)));
typedef float f32x4_v __attribute__((ext_vector_type(4)));

// exp for the softmax weights of the attention kernels: v_exp_f32 (2^x, <= 1 ulp) on x * log2(e).  The attention
// kernels are tolerance-class (parallel reductions), so the glibc-exact expf (a dozen f64 ops + a table load per
// call, hundreds of calls per lane) buys nothing there; the element-wise kernels keep expf_glibc.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; } // v_log_f32 (log2) * ln 2
// DeltaNet decode scalars (write strength, decay): tolerance-class inputs of the state update, evaluated with the
// hardware exp2 / log2 (<= 1 ulp each) instead of the ~150 double-precision operations of the libm-exact forms --
// a quarter of the 256-workgroup decode kernel's critical path.  Both decode kernels use these, so they agree bit for bit.
__device__ __forceinline__ float delta_beta_fast(float beta_raw) { return 1.0f / (1.0f + fast_exp(-beta_raw)); }
__device__ __forceinline__ float delta_decay_fast(float a_raw, float dt_bias, float a_log) {
    const float sp_input = a_raw + dt_bias;
    const float sp = sp_input > 20.0f ? sp_input : fast_log(1.0f + fast_exp(sp_input));
    return fast_exp(-fast_exp(a_log) * sp);
}

// butterfly sum over the `width` (power of two <= 64) consecutive lanes that contain this lane
template <int WIDTH> __device__ __forceinline__ float group_sum(float v) { return k::row_sum_rt(v, WIDTH); }
__device__ __forceinline__ float wave_sum(float v) { return k::row_sum_rt(v, 64); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
    return v;
}
__device__ __forceinline__ float group_sum_rt(float v, int width) { return k::row_sum_rt(v, width); }

// DeltaNet norm-gate, canonical sum of squares over a head's Dv outputs (shared by the stand-alone update kernel and
// the fused out-proj prologue so that both give the same bits): chunk sums of 8 consecutive elements (sequential
// fma), then the xor butterfly over the Dv / 8 chunks.
__device__ __forceinline__ float chunk8_sumsq(const float* v) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(v[e], v[e], s);
    return s;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + s_barrier, and the release fence makes
// hipcc drain vmcnt(0) -- EVERY outstanding global load, including the ones a software pipeline issued precisely so
// that they would still be in flight across the barrier (the counter is shared and in-order).  Kernels that exchange
// data through LDS only and prefetch global operands across iterations use this instead: LDS writes are complete
// (lgkmcnt(0)) before the barrier, the compiler still waits for a global load where its registers are consumed.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// workgroup sum (blockDim.x multiple of 64, <= 1024); `red` = 16 floats of LDS; result broadcast
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads(); // protect `red` from a previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i]; // fixed order => deterministic
    return t;
}

// Per-launch timing for uzu_hip_model_profile_decode_step: while `tl_launch_timer` is set, a launch goes through hipExtLaunchKernel,
// which stamps the kernel's own begin / end into the two events (the dispatch packet's timestamps -- what rocprofv3 reports),
// instead of events recorded around it on the stream (those add ~2 us of barrier packets to a 3-5 us kernel).
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                                   \
    do {                                                                                                                                    \
        if (::uzu::tl_launch_timer)                                                                                                         \
            hipExtLaunchKernelGGL((kernelName), (numBlocks), (numThreads), (memPerBlock), (streamId), ::uzu::tl_launch_timer->start,        \
                                  ::uzu::tl_launch_timer->stop, 0, __VA_ARGS__);                                                            \
        else                                                                                                                                \
            (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                                            \
    } while (0)

template <class F> uzu_status launch_check(F&& f, const char* what) {
    f();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return UZU_ERR_HIP;
    }
    return UZU_OK;
}

#define UZU_DISPATCH_T(dt, ...)                                   \
    [&]() -> uzu_status {                                          \
        if ((dt) == UZU_BF16) {                                    \
            using T = ::uzu::bf16_t;                               \
            return __VA_ARGS__();                                  \
        } else if ((dt) == UZU_F32) {                              \
            using T = float;                                       \
            return __VA_ARGS__();                                  \
        }                                                          \
        ::uzu::set_error("unsupported data type %u", (unsigned)(dt)); \
        return UZU_ERR_UNSUPPORTED;                                \
    }()

} // namespace uzu
