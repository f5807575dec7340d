// internal.h -- shared host-side declarations of the HIP backend library (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/uzu_hip.h"

namespace uzu {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// while set, kernel launches of this library stamp their own begin / end into the two events (device_utils.h: hipLaunchKernelGGL)
struct LaunchTimer {
    hipEvent_t start, stop;
};
extern thread_local LaunchTimer* tl_launch_timer; // runtime.hip

#define UZU_HIP_TRY(expr)                                                                                  \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            ::uzu::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
            return _e == hipErrorOutOfMemory ? UZU_ERR_OUT_OF_MEMORY : UZU_ERR_HIP;                        \
        }                                                                                                  \
    } while (0)

#define UZU_REQUIRE(cond, ...)               \
    do {                                     \
        if (!(cond)) {                       \
            ::uzu::set_error(__VA_ARGS__);   \
            return UZU_ERR_INVALID_ARGUMENT; \
        }                                    \
    } while (0)

#define UZU_UNSUPPORTED(cond, ...)           \
    do {                                     \
        if (cond) {                          \
            ::uzu::set_error(__VA_ARGS__);   \
            return UZU_ERR_UNSUPPORTED;      \
        }                                    \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: the "already raised" cache of a launch site is
// keyed by the current device (a process may hold contexts on several GPUs)
struct LdsLimit {
    size_t raised_to[16] = {0};
};
inline bool raise_lds_limit(LdsLimit& st, const void* fn, size_t lds) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& have = st.raised_to[dev & 15];
    if (lds <= have) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    have = lds;
    return true;
}

// Environment switches.  The shipped library reads SIX variables: UZU_HIP_EXACT, UZU_HIP_POISON, UZU_PREFILL_CHUNK, UZU_TP_TIMEOUT_MS,
// UZU_TP_INJECT_TIMEOUT_AT and UZU_HIP_TUNE -- one string "key=value,key=value" holding the A/B switches that tests cross in the product build
// (tune_env below; keys: gemm_form, gemm_splits, exact_scalar, rows_norm, conv_apply4, conv_oop, norm_partials, dn_split, prep_fused, attn_fused -- each documented where it is read;
// tests/test_gpu_prefill_switches.py holds the fused prefill paths to the paths they replace).  Every other knob is a LAB switch of the A/B scripts under
// tools/ and exists only in a library built with `make LAB=1` (-DUZU_LAB): in the product it reads as unset.
inline const char* tune_env(const char* key) {
    static thread_local char value[32];
    const char* e = getenv("UZU_HIP_TUNE");
    if (!e) return nullptr;
    const size_t kl = strlen(key);
    for (const char* q = e; *q;) {
        const char* end = strchr(q, ',');
        const size_t len = end ? (size_t)(end - q) : strlen(q);
        if (len > kl && !strncmp(q, key, kl) && q[kl] == '=') {
            size_t vl = len - kl - 1;
            if (vl >= sizeof value) vl = sizeof value - 1;
            memcpy(value, q + kl + 1, vl);
            value[vl] = 0;
            return value;
        }
        q += len;
        if (*q == ',') ++q;
    }
    return nullptr;
}
inline const char* lab_env(const char* name) {
#ifdef UZU_LAB
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// UZU_HIP_POISON (CI mode, runtime.hip): 1 = every fresh device allocation that is not zero-filled by contract, every user buffer and every hand-out of a
// recycled per-stream workspace block is filled with 0xFF bytes first (bf16 / f32 NaN, out-of-range indices) on the stream that will use it -- a kernel that
// reads what nothing wrote, or a pass that starts before its fills have landed, shows up as NaN logits instead of passing on zeros; 2 = the engine's scratch
// blocks as well (they are zero-filled otherwise).  No extra synchronisation of kernels; allocation-time fills are followed by a stream synchronisation.
int poison_level();
uzu_status poison_fill(void* p, size_t bytes, hipStream_t s, bool synchronise);

#define UZU_PROPAGATE(expr)            \
    do {                               \
        uzu_status _s = (expr);        \
        if (_s != UZU_OK) return _s;   \
    } while (0)

} // namespace uzu

struct uzu_hip_context {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t current_bytes = 0;
    size_t peak_bytes = 0;
    void* staging = nullptr; // pinned bounce buffer for uploads
    size_t staging_size = 0;
    int num_cus = 0;
    char name[128] = {0};
    // Context::start_capture / stop_capture: while active, every completed command buffer leaves a record (name, debug
    // groups, GPU time) that stop_capture writes to the trace file
    bool capture_active = false;
    std::string capture_path;
    struct CaptureRecord {
        std::string name;
        std::vector<std::string> groups;
        double gpu_ms;
    };
    std::vector<CaptureRecord> capture_records;
    bool vmm_supported = false; // hipMem* virtual memory management => DeviceCapabilities::SPARSE_BUFFERS
};

struct uzu_hip_buffer {
    uzu_hip_context* ctx = nullptr;
    void* dptr = nullptr;
    size_t size = 0;
    void* mirror = nullptr; // pinned host mirror (cpu_ptr)
    // SparseBuffer (buffer/sparse.rs:5-19): `size` bytes of reserved virtual address space, physical pages mapped on demand
    bool sparse = false;
    size_t page_bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> pages; // one handle per mapped page (null = unmapped)
};

enum class CmdbufState { Initial, Encoding, Executable, Pending, Completed };

struct uzu_hip_cmdbuf {
    uzu_hip_context* ctx = nullptr;
    std::string name;
    uint32_t flags = 0;
    CmdbufState state = CmdbufState::Initial;
    hipEvent_t ev_start = nullptr, ev_end = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    std::vector<std::string> debug_groups;
    std::vector<std::string> all_groups; // every group pushed while encoding (capture record)
    float last_ms = 0.f;
};

struct uzu_hip_kernel {
    uzu_hip_context* ctx = nullptr;
    uint32_t kind = 0;   // KernelKind
    uint32_t t[4] = {0}; // data types
    uint32_t f[12] = {0}; // specialization flags / consts
    // kernel-owned device scratch (UnifiedSampling's arg-max partials): hipMalloc'd outside any stream capture, grow-only
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
};

namespace uzu {

inline void* bptr(const uzu_buf& b) { return b.buffer ? (void*)((char*)b.buffer->dptr + b.offset) : nullptr; }
inline hipStream_t cb_stream(uzu_hip_cmdbuf* cb) { return cb->ctx->stream; }
uzu_status cmdbuf_check_encoding(uzu_hip_cmdbuf* cb);
uzu_status check_launch(const char* what);

} // namespace uzu
