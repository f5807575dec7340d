// tp.hip -- tensor-parallel exchange for the engine: one process per GPU, RCCL all-reduce over xGMI.
//
// The reference has no multi-device path; this is the MI355X design for it:
//   * row-parallel linears (out-proj, down-proj) produce an f32 partial row per token on every rank; the ranks
//     exchange them with ONE in-place ncclAllReduce(sum, f32) and then round to bf16 -- the bf16 rounding happens
//     once, after the sum, exactly where the single-GPU path rounds the full dot product;
//   * the vocab-sharded read-out exchanges 8 bytes per rank: an order-preserving key (logit, ~index) reduced with
//     ncclMax, so every rank commits the same greedy token (ties -> lowest index, unified_sampling.rs:90-95);
//   * librccl is loaded with dlopen only when a communicator is created: the single-GPU library has no RCCL
//     dependency and fails loudly (UZU_ERR_UNSUPPORTED) if a TP group is requested where RCCL is absent.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "device_utils.h"
#include "internal.h"
#include "tp.h"

namespace uzu {
namespace tp {

namespace {
struct Api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Api g_api;

uzu_status load_api() {
    if (g_api.handle) return UZU_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
        set_error("tp: cannot load librccl (%s)", dlerror());
        return UZU_ERR_UNSUPPORTED;
    }
    Api a;
    a.handle = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
        set_error("tp: librccl lacks a required entry point");
        return UZU_ERR_UNSUPPORTED;
    }
    g_api = a;
    return UZU_OK;
}

uzu_status check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return UZU_OK;
    set_error("tp: %s failed: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
    return UZU_ERR_HIP;
}

__global__ void __launch_bounds__(256) cast_f32_bf16_kernel(const float* in, uint16_t* out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const float4 v = *(const float4*)(in + i);
        uint2 o;
        o.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
        o.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
        *(uint2*)(out + i) = o;
    } else {
        for (size_t j = i; j < n; ++j) out[j] = f32_to_bf16(in[j]);
    }
}

// order-preserving map f32 -> u32 (larger float <=> larger unsigned); NaN never wins a greedy arg-max upstream
__device__ __forceinline__ uint32_t orderable(float v) {
    // -0.0 and +0.0 compare equal in every per-rank reduction (ties -> lowest index, unified_sampling.rs:90-95), so they
    // must map to the same key: otherwise the winner between ranks would depend on how the vocabulary is sharded
    const uint32_t raw = f32_to_bits(v);
    const uint32_t b = (raw << 1) == 0u ? 0u : raw;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t u) { return bits_to_f32((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// (value, local index) partials -> one packed key per rank
__global__ void __launch_bounds__(256) argmax_key_kernel(const float* pv, const uint32_t* pi, uint32_t parts, uint32_t vocab_offset, unsigned long long* key) {
    __shared__ float sv[4];
    __shared__ uint32_t si[4];
    float bv = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < parts; i += 256) {
        const float v = pv[i];
        const uint32_t ix = pi[i];
        if (v > bv || (v == bv && ix < bi)) bv = v, bi = ix;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const uint32_t oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sv[wave] = bv, si[wave] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) bv = sv[w], bi = si[w];
        const uint32_t global = bi == 0xFFFFFFFFu ? 0xFFFFFFFFu : bi + vocab_offset;
        *key = ((unsigned long long)orderable(bv) << 32) | (unsigned long long)(0xFFFFFFFFu - global);
    }
}
__global__ void key_from_token_kernel(const uint16_t* logits, const uint32_t* local_token, uint32_t vocab_offset, unsigned long long* key) {
    const uint32_t t = *local_token;
    *key = ((unsigned long long)orderable(bf16_to_f32(logits[t])) << 32) | (unsigned long long)(0xFFFFFFFFu - (t + vocab_offset));
}
__global__ void token_from_key_kernel(const unsigned long long* key, uint32_t* out_token) {
    const uint32_t inv = (uint32_t)(*key & 0xFFFFFFFFull);
    const uint32_t t = 0xFFFFFFFFu - inv;
    *out_token = t == 0xFFFFFFFFu ? 0u : t;
}
// the fused decode step's commit (k_decode.hip::argmax_commit_kernel) with the token taken from the reduced key
__global__ void commit_key_kernel(const unsigned long long* key, uint32_t* ctx_len, uint32_t* tokens, uint32_t* out_token, uint32_t* sampled) {
    const uint32_t inv = (uint32_t)(*key & 0xFFFFFFFFull);
    uint32_t t = 0xFFFFFFFFu - inv;
    if (t == 0xFFFFFFFFu) t = 0u;
    const uint32_t len = *ctx_len;
    *out_token = t;
    sampled[len] = t;
    tokens[0] = t;
    *ctx_len = len + 1;
}
} // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
};

uzu_status unique_id(uint8_t out[128]) {
    UZU_PROPAGATE(load_api());
    ncclUniqueId id;
    UZU_PROPAGATE(check(g_api.GetUniqueId(&id), "ncclGetUniqueId"));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out, &id, 128);
    return UZU_OK;
}

uzu_status comm_create(const uint8_t id_bytes[128], int rank, int size, Comm** out) {
    UZU_REQUIRE(size >= 1 && rank >= 0 && rank < size, "tp: bad rank %d of %d", rank, size);
    UZU_PROPAGATE(load_api());
    ncclUniqueId id;
    memcpy(&id, id_bytes, 128);
    Comm* c = new Comm();
    c->rank = rank, c->size = size;
    const uzu_status st = check(g_api.CommInitRank(&c->comm, size, id, rank), "ncclCommInitRank");
    if (st != UZU_OK) {
        delete c;
        return st;
    }
    *out = c;
    return UZU_OK;
}

void comm_destroy(Comm* c) {
    if (!c) return;
    if (c->comm && g_api.CommDestroy) g_api.CommDestroy(c->comm);
    delete c;
}
int comm_rank(const Comm* c) { return c->rank; }
int comm_size(const Comm* c) { return c->size; }

uzu_status all_reduce_sum_f32(Comm* c, hipStream_t s, float* buf, size_t count) {
    return check(g_api.AllReduce(buf, buf, count, ncclFloat32, ncclSum, c->comm, s), "ncclAllReduce(sum,f32)");
}
uzu_status all_reduce_max_u64(Comm* c, hipStream_t s, unsigned long long* buf, size_t count) {
    return check(g_api.AllReduce(buf, buf, count, ncclUint64, ncclMax, c->comm, s), "ncclAllReduce(max,u64)");
}

uzu_status cast_f32_bf16(hipStream_t s, const float* in, uint16_t* out, size_t n) {
    const uint32_t grid = (uint32_t)((n + 1023) / 1024);
    return launch_check([&] { hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, s, in, out, n); }, "tp_cast");
}
uzu_status argmax_key(hipStream_t s, const float* pv, const uint32_t* pi, uint32_t parts, uint32_t vocab_offset, unsigned long long* key) {
    return launch_check([&] { hipLaunchKernelGGL(argmax_key_kernel, dim3(1), dim3(256), 0, s, pv, pi, parts, vocab_offset, key); }, "tp_argmax_key");
}
uzu_status key_from_token(hipStream_t s, const uint16_t* logits, const uint32_t* local_token, uint32_t vocab_offset, unsigned long long* key) {
    return launch_check([&] { hipLaunchKernelGGL(key_from_token_kernel, dim3(1), dim3(1), 0, s, logits, local_token, vocab_offset, key); }, "tp_key_from_token");
}
uzu_status token_from_key(hipStream_t s, const unsigned long long* key, uint32_t* out_token) {
    return launch_check([&] { hipLaunchKernelGGL(token_from_key_kernel, dim3(1), dim3(1), 0, s, key, out_token); }, "tp_token_from_key");
}
uzu_status commit_key(hipStream_t s, const unsigned long long* key, uint32_t* ctx_len, uint32_t* tokens, uint32_t* out_token, uint32_t* sampled) {
    return launch_check([&] { hipLaunchKernelGGL(commit_key_kernel, dim3(1), dim3(1), 0, s, key, ctx_len, tokens, out_token, sampled); }, "tp_commit_key");
}

} // namespace tp
} // namespace uzu
