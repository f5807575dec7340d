// runtime.hip -- Context / Buffer / CommandBuffer of the HIP backend (C ABI, include/uzu_hip.h).
//
// Mirrors BU/backends/common/{context.rs,buffer/*.rs,command_buffer.rs}.  MI355X-native choices:
//   * device memory is plain hipMalloc (HBM); host visibility is an explicit pinned mirror + async copies
//     on the context stream (never host-mapped device memory on the kernel path);
//   * one in-order HIP stream per context = the reference's "command buffers execute in submission order";
//   * a command buffer is either eager (kernels are enqueued as they are encoded) or a captured hipGraph
//     that submit() launches -- the launch-latency answer for decode loops.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "internal.h"

namespace uzu {

static thread_local char g_error[1024] = "";
thread_local LaunchTimer* tl_launch_timer = nullptr;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

uzu_status cmdbuf_check_encoding(uzu_hip_cmdbuf* cb) {
    if (!cb) {
        set_error("null command buffer");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    if (cb->state != CmdbufState::Encoding) {
        set_error("command buffer '%s' is not in the Encoding state", cb->name.c_str());
        return UZU_ERR_STATE;
    }
    return UZU_OK;
}

} // namespace uzu

namespace uzu {
static int g_poison = -1;
int poison_level() {
    if (g_poison < 0) {
        const char* e = getenv("UZU_HIP_POISON");
        g_poison = e ? atoi(e) : 0;
        if (g_poison < 0 || g_poison > 2) g_poison = 0;
    }
    return g_poison;
}
uzu_status poison_fill(void* p, size_t bytes, hipStream_t s, bool synchronise) {
    if (!p || !bytes) return UZU_OK;
    UZU_HIP_TRY(hipMemsetAsync(p, 0xFF, bytes, s));
    if (synchronise) UZU_HIP_TRY(hipStreamSynchronize(s));
    return UZU_OK;
}
} // namespace uzu

namespace uzu {
namespace k {
// Per-stream scratch for kernels that need a transient device buffer (split-K partials, DeltaNet chunk tables, arg-max
// partials).  One grow-only hipMalloc block per stream: consecutive users on a stream are ordered by the stream itself.
// (The stream-ordered pool, hipMallocAsync / hipFreeAsync, handed out blocks that were overwritten while the kernels of
// the call were still in flight on this ROCm build -- tools/dbg_gemm.py -- so it is not used.)
namespace {
struct WsEntry {
    void* ptr = nullptr;
    size_t bytes = 0;
};
std::mutex g_ws_mutex;
std::unordered_map<hipStream_t, WsEntry> g_ws;
} // namespace
void* stream_workspace(hipStream_t s, size_t bytes) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr; // a graph would pin a block that may be regrown
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    WsEntry& e = g_ws[s];
    if (e.bytes >= bytes) {
        // poison mode: a recycled block comes back full of NaNs, stream-ordered behind its previous user and in front of the next
        if (poison_level() && bytes) (void)poison_fill(e.ptr, bytes, s, false);
        return e.ptr;
    }
    if (e.ptr) { // earlier users may still be running: drain the stream before the block goes away
        (void)hipStreamSynchronize(s);
        (void)hipFree(e.ptr);
        e.ptr = nullptr, e.bytes = 0;
    }
    const size_t want = ((bytes > 2 * e.bytes ? bytes : 2 * e.bytes) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    e.ptr = p, e.bytes = want;
    if (poison_level()) (void)poison_fill(p, want, s, false);
    return p;
}
void stream_workspace_release(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws.find(s);
    if (it == g_ws.end()) return;
    if (it->second.ptr) (void)hipFree(it->second.ptr);
    g_ws.erase(it);
}
} // namespace k
} // namespace uzu

using namespace uzu;

// ---- roctx (optional): debug groups and capture brackets become marker ranges for an attached rocprofv3 --marker-trace.
// Loaded on demand with dlopen (Context::enable_capture or the first range); absent library => silent no-ops.
namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
typedef int (*roctx_ctl_fn)(uint64_t);
struct Roctx {
    bool tried = false;
    roctx_push_fn push = nullptr;
    roctx_pop_fn pop = nullptr;
    roctx_ctl_fn pause = nullptr, resume = nullptr;
} g_roctx;
void roctx_load() {
    if (g_roctx.tried) return;
    g_roctx.tried = true;
    void* h = nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    g_roctx.push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
    g_roctx.pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
    g_roctx.pause = (roctx_ctl_fn)dlsym(h, "roctxProfilerPause");
    g_roctx.resume = (roctx_ctl_fn)dlsym(h, "roctxProfilerResume");
}
void roctx_push(const char* name) {
    if (g_roctx.tried && g_roctx.push) (void)g_roctx.push(name);
}
void roctx_pop() {
    if (g_roctx.tried && g_roctx.pop) (void)g_roctx.pop();
}
void roctx_pause() {
    if (g_roctx.tried && g_roctx.pause) (void)g_roctx.pause(0);
}
void roctx_resume() {
    if (g_roctx.tried && g_roctx.resume) (void)g_roctx.resume(0);
}
} // namespace

extern "C" {

const char* uzu_hip_last_error(void) { return g_error; }

// ---------------------------------------------------------------------------------- Context
uzu_status uzu_hip_context_create(int32_t device_ordinal, uzu_hip_context** out) {
    UZU_REQUIRE(out != nullptr, "context_create: out is null");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        set_error("no HIP device available (%s): the uzu HIP backend needs an AMD GPU, there is no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "0 devices");
        return UZU_ERR_HIP;
    }
    UZU_REQUIRE(device_ordinal >= 0 && device_ordinal < count, "context_create: device %d out of range (0..%d)", device_ordinal, count - 1);
    UZU_HIP_TRY(hipSetDevice(device_ordinal));
    auto* ctx = new uzu_hip_context();
    ctx->device = device_ordinal;
    hipDeviceProp_t prop;
    UZU_HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    ctx->num_cus = prop.multiProcessorCount;
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
    int vmm = 0;
    if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, device_ordinal) != hipSuccess) {
        (void)hipGetLastError();
        vmm = 0;
    }
    ctx->vmm_supported = vmm != 0;
    UZU_HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->staging_size = 64u << 20;
    UZU_HIP_TRY(hipHostMalloc(&ctx->staging, ctx->staging_size, hipHostMallocDefault));
    *out = ctx;
    return UZU_OK;
}

void uzu_hip_context_destroy(uzu_hip_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->staging) (void)hipHostFree(ctx->staging);
    k::stream_workspace_release(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

uzu_status uzu_hip_context_peak_memory_usage(uzu_hip_context* ctx, size_t* out) {
    UZU_REQUIRE(ctx && out, "peak_memory_usage: null argument");
    *out = ctx->peak_bytes;
    return UZU_OK;
}

uzu_status uzu_hip_context_device_name(uzu_hip_context* ctx, char* out, size_t cap) {
    UZU_REQUIRE(ctx && out && cap, "device_name: null argument");
    snprintf(out, cap, "%s", ctx->name);
    return UZU_OK;
}

uzu_status uzu_hip_context_synchronize(uzu_hip_context* ctx) {
    UZU_REQUIRE(ctx, "synchronize: null context");
    UZU_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return UZU_OK;
}

void* uzu_hip_context_stream(uzu_hip_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ---------------------------------------------------------------------------------- Buffer
uzu_status uzu_hip_buffer_create(uzu_hip_context* ctx, size_t size, uzu_hip_buffer** out) {
    UZU_REQUIRE(ctx && out, "buffer_create: null argument");
    (void)hipSetDevice(ctx->device);
    auto* b = new uzu_hip_buffer();
    b->ctx = ctx;
    b->size = size;
    const size_t alloc = size ? (size + 255) & ~(size_t)255 : 256;
    hipError_t e = hipMalloc(&b->dptr, alloc);
    if (e != hipSuccess) {
        delete b;
        set_error("hipMalloc(%zu) failed: %s", alloc, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? UZU_ERR_OUT_OF_MEMORY : UZU_ERR_HIP;
    }
    ctx->current_bytes += alloc;
    if (ctx->current_bytes > ctx->peak_bytes) ctx->peak_bytes = ctx->current_bytes;
    if (poison_level()) (void)poison_fill(b->dptr, alloc, ctx->stream, true); // Buffer contents are undefined until written (as in the reference): make that visible
    *out = b;
    return UZU_OK;
}

void uzu_hip_buffer_destroy(uzu_hip_buffer* b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream); // work encoded against this buffer may still be in flight
    if (b->mirror) (void)hipHostFree(b->mirror);
    if (b->sparse) {
        for (size_t i = 0; i < b->pages.size(); ++i)
            if (b->pages[i]) {
                (void)hipMemUnmap((char*)b->dptr + i * b->page_bytes, b->page_bytes);
                (void)hipMemRelease(b->pages[i]);
                b->ctx->current_bytes -= b->page_bytes;
            }
        if (b->dptr) (void)hipMemAddressFree(b->dptr, b->size);
        delete b;
        return;
    }
    if (b->dptr) (void)hipFree(b->dptr);
    const size_t alloc = b->size ? (b->size + 255) & ~(size_t)255 : 256;
    b->ctx->current_bytes -= alloc;
    delete b;
}

// ---------------------------------------------------------------------------------- SparseBuffer (buffer/sparse.rs:5-19)
// Context::create_sparse_buffer: `capacity` bytes of virtual address space (rounded up to whole pages), no memory behind
// it.  map / unmap attach and detach physical pages (hipMemCreate + hipMemMap at the allocation granularity, 2 MiB on
// MI355X): the reference grows its KV caches this way (mixer/attention/state.rs:144-170).  gpu_ptr() is stable for the
// lifetime of the buffer, so kernels encoded against it see the pages mapped by the time they run.
static hipMemAllocationProp sparse_prop(int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    return prop;
}
uzu_status uzu_hip_sparse_buffer_create(uzu_hip_context* ctx, size_t capacity, uzu_hip_buffer** out) {
    UZU_REQUIRE(ctx && out, "sparse_buffer_create: null argument");
    (void)hipSetDevice(ctx->device);
    UZU_UNSUPPORTED(!ctx->vmm_supported, "sparse_buffer_create: this device / driver has no virtual memory management (device_capabilities lacks SPARSE_BUFFERS)");
    const hipMemAllocationProp prop = sparse_prop(ctx->device);
    size_t gran = 0;
    UZU_HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    UZU_REQUIRE(gran > 0, "sparse_buffer_create: zero allocation granularity");
    auto* b = new uzu_hip_buffer();
    b->ctx = ctx, b->sparse = true, b->page_bytes = gran;
    b->size = ((capacity ? capacity : 1) + gran - 1) / gran * gran;
    hipError_t e = hipMemAddressReserve(&b->dptr, b->size, gran, nullptr, 0);
    if (e != hipSuccess) {
        set_error("sparse_buffer_create: hipMemAddressReserve(%zu) failed: %s", b->size, hipGetErrorString(e));
        delete b;
        return UZU_ERR_HIP;
    }
    b->pages.assign(b->size / gran, (hipMemGenericAllocationHandle_t) nullptr);
    *out = b;
    return UZU_OK;
}
size_t uzu_hip_sparse_buffer_page_size(const uzu_hip_buffer* b) { return b && b->sparse ? b->page_bytes : 0; }
uzu_status uzu_hip_sparse_buffer_map(uzu_hip_buffer* b, size_t first_page, size_t end_page) {
    UZU_REQUIRE(b && b->sparse, "sparse_buffer_map: not a sparse buffer");
    UZU_REQUIRE(first_page <= end_page && end_page <= b->pages.size(), "sparse_buffer_map: pages %zu..%zu out of range (%zu pages)", first_page, end_page, b->pages.size());
    (void)hipSetDevice(b->ctx->device);
    const hipMemAllocationProp prop = sparse_prop(b->ctx->device);
    hipMemAccessDesc access = {};
    access.location.type = hipMemLocationTypeDevice, access.location.id = b->ctx->device, access.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t i = first_page; i < end_page; ++i) {
        if (b->pages[i]) continue; // already mapped: Metal's updateTextureMappings is idempotent as well
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, b->page_bytes, &prop, 0);
        if (e != hipSuccess) {
            set_error("sparse_buffer_map: hipMemCreate(%zu) failed: %s", b->page_bytes, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? UZU_ERR_OUT_OF_MEMORY : UZU_ERR_HIP;
        }
        void* va = (char*)b->dptr + i * b->page_bytes;
        if ((e = hipMemMap(va, b->page_bytes, 0, h, 0)) != hipSuccess || (e = hipMemSetAccess(va, b->page_bytes, &access, 1)) != hipSuccess) {
            (void)hipMemRelease(h);
            set_error("sparse_buffer_map: mapping page %zu failed: %s", i, hipGetErrorString(e));
            return UZU_ERR_HIP;
        }
        b->pages[i] = h;
        b->ctx->current_bytes += b->page_bytes;
        if (b->ctx->current_bytes > b->ctx->peak_bytes) b->ctx->peak_bytes = b->ctx->current_bytes;
    }
    return UZU_OK;
}
uzu_status uzu_hip_sparse_buffer_unmap(uzu_hip_buffer* b, size_t first_page, size_t end_page) {
    UZU_REQUIRE(b && b->sparse, "sparse_buffer_unmap: not a sparse buffer");
    UZU_REQUIRE(first_page <= end_page && end_page <= b->pages.size(), "sparse_buffer_unmap: pages %zu..%zu out of range (%zu pages)", first_page, end_page, b->pages.size());
    (void)hipSetDevice(b->ctx->device);
    UZU_HIP_TRY(hipStreamSynchronize(b->ctx->stream)); // kernels encoded against these pages may still be running
    for (size_t i = first_page; i < end_page; ++i) {
        if (!b->pages[i]) continue;
        UZU_HIP_TRY(hipMemUnmap((char*)b->dptr + i * b->page_bytes, b->page_bytes));
        UZU_HIP_TRY(hipMemRelease(b->pages[i]));
        b->pages[i] = nullptr;
        b->ctx->current_bytes -= b->page_bytes;
    }
    return UZU_OK;
}
// transfers touch mapped pages only
static uzu_status sparse_range_mapped(const uzu_hip_buffer* b, size_t offset, size_t size, const char* what) {
    if (!b->sparse || !size) return UZU_OK;
    for (size_t i = offset / b->page_bytes; i <= (offset + size - 1) / b->page_bytes; ++i)
        if (!b->pages[i]) {
            set_error("%s: page %zu of the sparse buffer is not mapped", what, i);
            return UZU_ERR_INVALID_ARGUMENT;
        }
    return UZU_OK;
}

// ---------------------------------------------------------------------------------- capture, capabilities (context.rs:36-47)
uzu_status uzu_hip_context_device_capabilities(uzu_hip_context* ctx, uint32_t* out) {
    UZU_REQUIRE(ctx && out, "device_capabilities: null argument");
    *out = ctx->vmm_supported ? UZU_DEVICE_CAP_SPARSE_BUFFERS : 0u;
    return UZU_OK;
}
void uzu_hip_context_enable_capture(void) { roctx_load(); }
uzu_status uzu_hip_context_start_capture(uzu_hip_context* ctx, const char* trace_path) {
    UZU_REQUIRE(ctx && trace_path && trace_path[0], "start_capture: null / empty trace path");
    if (ctx->capture_active) {
        set_error("start_capture: a capture is already running (%s)", ctx->capture_path.c_str());
        return UZU_ERR_STATE;
    }
    ctx->capture_active = true, ctx->capture_path = trace_path;
    ctx->capture_records.clear();
    roctx_resume(); // a profiler attached with its collection paused (rocprofv3 --collection-period / roctx control) starts here
    roctx_push(trace_path);
    return UZU_OK;
}
uzu_status uzu_hip_context_stop_capture(uzu_hip_context* ctx) {
    UZU_REQUIRE(ctx, "stop_capture: null context");
    if (!ctx->capture_active) {
        set_error("stop_capture: no capture is running");
        return UZU_ERR_STATE;
    }
    (void)hipSetDevice(ctx->device);
    UZU_HIP_TRY(hipStreamSynchronize(ctx->stream));
    roctx_pop();
    roctx_pause();
    ctx->capture_active = false;
    FILE* f = fopen(ctx->capture_path.c_str(), "w");
    if (!f) {
        set_error("stop_capture: cannot write %s", ctx->capture_path.c_str());
        return UZU_ERR_INVALID_ARGUMENT;
    }
    auto esc = [](const std::string& in) {
        std::string o;
        for (char c : in) {
            if (c == '"' || c == '\\') o += '\\';
            o += (unsigned char)c < 0x20 ? ' ' : c;
        }
        return o;
    };
    fprintf(f, "{\"device\": \"%s\", \"command_buffers\": [", esc(ctx->name).c_str());
    for (size_t i = 0; i < ctx->capture_records.size(); ++i) {
        const auto& r = ctx->capture_records[i];
        fprintf(f, "%s\n  {\"name\": \"%s\", \"gpu_time_ns\": %.0f, \"debug_groups\": [", i ? "," : "", esc(r.name).c_str(), r.gpu_ms * 1e6);
        for (size_t g = 0; g < r.groups.size(); ++g) fprintf(f, "%s\"%s\"", g ? ", " : "", esc(r.groups[g]).c_str());
        fprintf(f, "]}");
    }
    fprintf(f, "\n]}\n");
    fclose(f);
    ctx->capture_records.clear();
    return UZU_OK;
}

uint64_t uzu_hip_buffer_gpu_ptr(const uzu_hip_buffer* b) { return b ? (uint64_t)(uintptr_t)b->dptr : 0; }
size_t uzu_hip_buffer_size(const uzu_hip_buffer* b) { return b ? b->size : 0; }

uzu_status uzu_hip_buffer_cpu_ptr(uzu_hip_buffer* b, void** out) {
    UZU_REQUIRE(b && out, "buffer_cpu_ptr: null argument");
    UZU_UNSUPPORTED(b->sparse, "buffer_cpu_ptr: a sparse buffer has no host view (the reference's SparseBuffer is a Buffer, not a DenseBuffer)");
    if (!b->mirror) {
        UZU_HIP_TRY(hipHostMalloc(&b->mirror, b->size ? b->size : 1, hipHostMallocDefault));
        memset(b->mirror, 0, b->size);
    }
    *out = b->mirror;
    return UZU_OK;
}

uzu_status uzu_hip_buffer_flush_to_device(uzu_hip_buffer* b, size_t offset, size_t size) {
    UZU_REQUIRE(b && b->mirror, "flush_to_device: buffer has no host mirror");
    UZU_REQUIRE(offset + size <= b->size, "flush_to_device: range out of bounds");
    UZU_HIP_TRY(hipMemcpyAsync((char*)b->dptr + offset, (char*)b->mirror + offset, size, hipMemcpyHostToDevice, b->ctx->stream));
    return UZU_OK;
}

uzu_status uzu_hip_buffer_fetch_from_device(uzu_hip_buffer* b, size_t offset, size_t size) {
    UZU_REQUIRE(b, "fetch_from_device: null buffer");
    void* p;
    UZU_PROPAGATE(uzu_hip_buffer_cpu_ptr(b, &p));
    UZU_REQUIRE(offset + size <= b->size, "fetch_from_device: range out of bounds");
    UZU_HIP_TRY(hipMemcpyAsync((char*)b->mirror + offset, (char*)b->dptr + offset, size, hipMemcpyDeviceToHost, b->ctx->stream));
    UZU_HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return UZU_OK;
}

uzu_status uzu_hip_buffer_upload(uzu_hip_buffer* b, size_t offset, const void* src, size_t size) {
    UZU_REQUIRE(b && (src || !size), "buffer_upload: null argument");
    UZU_REQUIRE(offset + size <= b->size, "buffer_upload: range [%zu,+%zu) exceeds buffer size %zu", offset, size, b->size);
    UZU_PROPAGATE(sparse_range_mapped(b, offset, size, "buffer_upload"));
    uzu_hip_context* ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    // pageable source: bounce through the pinned staging buffer in stream order
    size_t done = 0;
    while (done < size) {
        const size_t n = size - done < ctx->staging_size ? size - done : ctx->staging_size;
        UZU_HIP_TRY(hipStreamSynchronize(ctx->stream)); // staging buffer is reused
        memcpy(ctx->staging, (const char*)src + done, n);
        UZU_HIP_TRY(hipMemcpyAsync((char*)b->dptr + offset + done, ctx->staging, n, hipMemcpyHostToDevice, ctx->stream));
        done += n;
    }
    UZU_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return UZU_OK;
}

uzu_status uzu_hip_buffer_download(uzu_hip_buffer* b, size_t offset, void* dst, size_t size) {
    UZU_REQUIRE(b && (dst || !size), "buffer_download: null argument");
    UZU_REQUIRE(offset + size <= b->size, "buffer_download: range [%zu,+%zu) exceeds buffer size %zu", offset, size, b->size);
    UZU_PROPAGATE(sparse_range_mapped(b, offset, size, "buffer_download"));
    uzu_hip_context* ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    size_t done = 0;
    while (done < size) {
        const size_t n = size - done < ctx->staging_size ? size - done : ctx->staging_size;
        UZU_HIP_TRY(hipMemcpyAsync(ctx->staging, (char*)b->dptr + offset + done, n, hipMemcpyDeviceToHost, ctx->stream));
        UZU_HIP_TRY(hipStreamSynchronize(ctx->stream));
        memcpy((char*)dst + done, ctx->staging, n);
        done += n;
    }
    return UZU_OK;
}

// ---------------------------------------------------------------------------------- CommandBuffer
uzu_status uzu_hip_cmdbuf_create(uzu_hip_context* ctx, const char* name, uint32_t flags, uzu_hip_cmdbuf** out) {
    UZU_REQUIRE(ctx && out, "cmdbuf_create: null argument");
    (void)hipSetDevice(ctx->device);
    auto* cb = new uzu_hip_cmdbuf();
    cb->ctx = ctx;
    cb->name = name ? name : "";
    cb->flags = flags;
    if (hipEventCreate(&cb->ev_start) != hipSuccess || hipEventCreate(&cb->ev_end) != hipSuccess) {
        if (cb->ev_start) (void)hipEventDestroy(cb->ev_start);
        delete cb;
        set_error("cmdbuf_create: hipEventCreate failed");
        return UZU_ERR_HIP;
    }
    *out = cb;
    return UZU_OK;
}

uzu_status uzu_hip_cmdbuf_start_encoding(uzu_hip_cmdbuf* cb) {
    UZU_REQUIRE(cb, "start_encoding: null command buffer");
    if (cb->state != CmdbufState::Initial) {
        set_error("start_encoding: command buffer '%s' is not Initial", cb->name.c_str());
        return UZU_ERR_STATE;
    }
    if (cb->flags & UZU_CMDBUF_GRAPH) {
        UZU_HIP_TRY(hipStreamBeginCapture(cb->ctx->stream, hipStreamCaptureModeThreadLocal));
    } else {
        UZU_HIP_TRY(hipEventRecord(cb->ev_start, cb->ctx->stream));
    }
    cb->state = CmdbufState::Encoding;
    return UZU_OK;
}

uzu_status uzu_hip_cmdbuf_encode_copy(uzu_hip_cmdbuf* cb, uzu_buf src, uzu_buf dst, size_t size) {
    UZU_PROPAGATE(cmdbuf_check_encoding(cb));
    UZU_REQUIRE(src.buffer && dst.buffer, "encode_copy: null buffer");
    UZU_REQUIRE(src.offset + size <= src.buffer->size && dst.offset + size <= dst.buffer->size, "encode_copy: range out of bounds");
    if (!size) return UZU_OK;
    UZU_HIP_TRY(hipMemcpyAsync(bptr(dst), bptr(src), size, hipMemcpyDeviceToDevice, cb->ctx->stream));
    return UZU_OK;
}

uzu_status uzu_hip_cmdbuf_encode_fill(uzu_hip_cmdbuf* cb, uzu_buf dst, size_t size, uint8_t value) {
    UZU_PROPAGATE(cmdbuf_check_encoding(cb));
    UZU_REQUIRE(dst.buffer, "encode_fill: null buffer");
    UZU_REQUIRE(dst.offset + size <= dst.buffer->size, "encode_fill: range out of bounds");
    if (!size) return UZU_OK;
    UZU_HIP_TRY(hipMemsetAsync(bptr(dst), value, size, cb->ctx->stream));
    return UZU_OK;
}

uzu_status uzu_hip_cmdbuf_encode_barrier(uzu_hip_cmdbuf* cb) { return cmdbuf_check_encoding(cb); }

uzu_status uzu_hip_cmdbuf_push_debug_group(uzu_hip_cmdbuf* cb, const char* name) {
    UZU_PROPAGATE(cmdbuf_check_encoding(cb));
    cb->debug_groups.push_back(name ? name : "");
    cb->all_groups.push_back(name ? name : "");
    roctx_push(name ? name : "");
    return UZU_OK;
}

uzu_status uzu_hip_cmdbuf_pop_debug_group(uzu_hip_cmdbuf* cb) {
    UZU_PROPAGATE(cmdbuf_check_encoding(cb));
    UZU_REQUIRE(!cb->debug_groups.empty(), "pop_debug_group: no open group");
    cb->debug_groups.pop_back();
    roctx_pop();
    return UZU_OK;
}

uzu_status uzu_hip_cmdbuf_end_encoding(uzu_hip_cmdbuf* cb) {
    UZU_PROPAGATE(cmdbuf_check_encoding(cb));
    if (cb->flags & UZU_CMDBUF_GRAPH) {
        UZU_HIP_TRY(hipStreamEndCapture(cb->ctx->stream, &cb->graph));
        UZU_HIP_TRY(hipGraphInstantiate(&cb->graph_exec, cb->graph, nullptr, nullptr, 0));
    }
    cb->state = CmdbufState::Executable;
    return UZU_OK;
}

uzu_status uzu_hip_cmdbuf_submit(uzu_hip_cmdbuf* cb) {
    UZU_REQUIRE(cb, "submit: null command buffer");
    const bool graph = (cb->flags & UZU_CMDBUF_GRAPH) != 0;
    // a graph command buffer may be re-submitted after completion (replay)
    if (!(cb->state == CmdbufState::Executable || (graph && cb->state == CmdbufState::Completed))) {
        set_error("submit: command buffer '%s' is not Executable", cb->name.c_str());
        return UZU_ERR_STATE;
    }
    if (graph) {
        UZU_HIP_TRY(hipEventRecord(cb->ev_start, cb->ctx->stream));
        UZU_HIP_TRY(hipGraphLaunch(cb->graph_exec, cb->ctx->stream));
    }
    UZU_HIP_TRY(hipEventRecord(cb->ev_end, cb->ctx->stream));
    cb->state = CmdbufState::Pending;
    return UZU_OK;
}

uzu_status uzu_hip_cmdbuf_wait_until_completed(uzu_hip_cmdbuf* cb) {
    UZU_REQUIRE(cb, "wait_until_completed: null command buffer");
    if (cb->state != CmdbufState::Pending) {
        set_error("wait_until_completed: command buffer '%s' is not Pending", cb->name.c_str());
        return UZU_ERR_STATE;
    }
    UZU_HIP_TRY(hipEventSynchronize(cb->ev_end));
    UZU_HIP_TRY(hipEventElapsedTime(&cb->last_ms, cb->ev_start, cb->ev_end));
    cb->state = CmdbufState::Completed;
    if (cb->ctx->capture_active) cb->ctx->capture_records.push_back({cb->name, cb->all_groups, (double)cb->last_ms});
    return UZU_OK;
}

uzu_status uzu_hip_cmdbuf_gpu_execution_time_ns(uzu_hip_cmdbuf* cb, uint64_t* out) {
    UZU_REQUIRE(cb && out, "gpu_execution_time: null argument");
    if (cb->state != CmdbufState::Completed) {
        set_error("gpu_execution_time: command buffer '%s' is not Completed", cb->name.c_str());
        return UZU_ERR_STATE;
    }
    *out = (uint64_t)((double)cb->last_ms * 1e6);
    return UZU_OK;
}

void uzu_hip_cmdbuf_destroy(uzu_hip_cmdbuf* cb) {
    if (!cb) return;
    (void)hipSetDevice(cb->ctx->device);
    if (cb->state == CmdbufState::Encoding && (cb->flags & UZU_CMDBUF_GRAPH)) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(cb->ctx->stream, &g);
        if (g) (void)hipGraphDestroy(g);
    }
    if (cb->state == CmdbufState::Pending) (void)hipEventSynchronize(cb->ev_end);
    if (cb->graph_exec) (void)hipGraphExecDestroy(cb->graph_exec);
    if (cb->graph) (void)hipGraphDestroy(cb->graph);
    (void)hipEventDestroy(cb->ev_start);
    (void)hipEventDestroy(cb->ev_end);
    delete cb;
}

void uzu_hip_kernel_destroy(uzu_hip_kernel* k) {
    if (!k) return;
    if (k->scratch) { // encodes that reference the block may still be in flight
        (void)hipStreamSynchronize(k->ctx->stream);
        (void)hipFree(k->scratch);
    }
    delete k;
}

} // extern "C"
