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
#include <stdlib.h>
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
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr; // optional
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
    a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
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
// the same for `rows` sampled rows at once (a speculated tree's nodes): logits [rows, row_stride] of this rank's shard
__global__ void keys_from_tokens_kernel(const uint16_t* logits, size_t row_stride, const uint32_t* local_tokens, uint32_t vocab_offset, unsigned long long* keys, uint32_t rows) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint32_t t = local_tokens[r];
    keys[r] = ((unsigned long long)orderable(bf16_to_f32(logits[(size_t)r * row_stride + t])) << 32) | (unsigned long long)(0xFFFFFFFFu - (t + vocab_offset));
}
__global__ void tokens_from_keys_kernel(const unsigned long long* keys, uint32_t* out_tokens, uint32_t rows) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint32_t t = 0xFFFFFFFFu - (uint32_t)(keys[r] & 0xFFFFFFFFull);
    out_tokens[r] = t == 0xFFFFFFFFu ? 0u : t;
}
// Stochastic sampling needs the WHOLE distribution of a row (top-k / top-p / min-p, unified_sampling.rs:13-99): every rank files its shard of
// the bf16 logits at its vocabulary offset of a zeroed f32 row; the sum over the ranks is then the full row, exactly (one non-zero term per
// element), and its bf16 rounding gives back the shard's own bits.
__global__ void __launch_bounds__(256) scatter_shard_kernel(const uint16_t* local, uint32_t local_n, uint32_t vocab_offset, uint32_t vocab, float* full) {
    const uint32_t row = blockIdx.y;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < vocab; i += gridDim.x * 256) {
        const uint32_t j = i - vocab_offset; // wraps for i < vocab_offset
        full[(size_t)row * vocab + i] = j < local_n ? bf16_to_f32(local[(size_t)row * local_n + j]) : 0.0f;
    }
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

// ---- one-shot peer-to-peer all-reduce for the decode-sized messages (4-20 KB) ------------------------------------------------
// A ring all-reduce of a 4 KB row is pure latency: 2 (N - 1) dependent hops plus RCCL's launch protocol, ~48 times per token.
// On one node every GPU can write every other GPU's memory over xGMI, so the exchange collapses to ONE hop: every rank owns a
// MAILBOX (device memory exported with hipIpcGetMemHandle, opened by the peers with hipIpcOpenMemHandle); an all-reduce is one
// kernel per rank that
//   1. pushes its vector into slot [parity][rank] of EVERY rank's mailbox (its own included), 16-byte stores,
//   2. publishes: system-scope fence, then the sequence number into flag [parity][rank] of every mailbox,
//   3. polls the flags of its OWN mailbox (local memory) until all ranks' sequence numbers have arrived -- bounded spin,
//   4. adds the size vectors in RANK ORDER (the same order on every rank => bit-identical results everywhere).
// Two parities: a rank can run at most one exchange ahead of the slowest peer (it needs that peer's contribution to finish
// the current one), so slot p of exchange n is free again by exchange n + 2.  The sequence number lives in device memory and is
// advanced by the kernel itself, so the launch is a plain kernel node: the TP decode step can be captured and replayed as a
// hipGraph.  Mailboxes are allocated uncached (hipDeviceMallocUncached): remote writes must be seen by the owner's polls.
constexpr uint32_t kMailboxFloats = 8192;   // capacity per (parity, rank) slot: 32 KB (d_model <= 8192; keys use 2 words)
constexpr uint32_t kMaxRanks = 8;
struct Mailbox {
    float data[2][kMaxRanks][kMailboxFloats];
    uint32_t flags[2][kMaxRanks][16]; // one 64-byte line per flag
    uint32_t seq[16];                 // this rank's exchange counter (device-resident)
    uint32_t error[16];               // set when a bounded spin gave up
};
struct P2P {
    Mailbox* local = nullptr;
    Mailbox* peer[kMaxRanks] = {nullptr};
    bool connected = false;
};

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    P2P p2p;
    unsigned long long rccl_calls = 0, p2p_calls = 0; // collectives enqueued on either route (host counters: a captured graph counts once)
};

namespace {
struct P2PArgs {
    Mailbox* box[kMaxRanks];
    int rank, size;
    unsigned long long timeout_ticks; // bounded wait, in ticks of the 100 MHz s_memrealtime clock
    uint32_t inject_seq;              // test hook (UZU_TP_INJECT_TIMEOUT_AT): the exchange with this sequence number behaves as if its wait had given up
};
// OP 0: f32 sum over `count` floats (in place on buf; `bf16_out` (optional) also receives the sums rounded to bf16: the cast the
// engine would otherwise launch as its own kernel); OP 1: u64 max over `count` keys (buf = unsigned long long*).
//
// Mailbox traffic uses relaxed system-scope 8-byte atomics (sc0 sc1 accesses: they bypass the caches on both sides whatever
// memory type the peer mapping got) and explicit ordering instead of release / acquire fences -- a system-scope fence writes back
// or invalidates the whole L2 (1.7 us each on a clean cache): payload stores -> every wave s_waitcnt vmcnt(0) (a write-through
// store is acknowledged by the destination) -> workgroup barrier -> flag stores; on the other side the payload loads are issued
// after the polling loop has seen the flag and the barrier has released the workgroup (no speculation across it).
typedef unsigned long long u64;
__device__ __forceinline__ void st_sys(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 ld_sys(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <int OP>
__global__ void __launch_bounds__(256) p2p_all_reduce_kernel(P2PArgs a, void* buf, uint32_t count, uint16_t* bf16_out) {
    __shared__ uint32_t s_seq, s_ok;
    Mailbox* mine = a.box[a.rank];
    const uint32_t tid = threadIdx.x, words = OP == 0 ? count : 2 * count, pairs = (words + 1) / 2;
    // A failed exchange is STICKY and GROUP-WIDE: the rank whose wait gave up writes the failing sequence number into the error word
    // of EVERY mailbox (its own and its peers', system scope).  A slow peer -- the realistic skew case: it finds every flag it waits
    // for, because the rank that gave up had already pushed its row -- therefore still sees the failure: its next exchange starts
    // poisoned (NaN sums / a zero key, no wait) and its host finds the error word at its next sync point (p2p_check), so that every
    // rank raises instead of one rank sitting in a host barrier while the others carry on with garbage sums.
    if (tid == 0) s_seq = mine->seq[0] + 1u, s_ok = __hip_atomic_load(&mine->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u ? 1u : 0u;
    __syncthreads();
    const uint32_t seq = s_seq, par = seq & 1u;
    const bool poisoned = s_ok == 0u; // uniform
    __syncthreads();
    // 1. push: this rank's row into slot [parity][rank] of every mailbox (an odd tail word travels with a zero partner)
    const uint32_t* src = (const uint32_t*)buf;
    for (uint32_t i = tid; i < pairs; i += 256) {
        const u64 v = (u64)src[2 * i] | (2 * i + 1 < words ? (u64)src[2 * i + 1] << 32 : 0ull);
        for (int r = 0; r < a.size; ++r) st_sys((u64*)a.box[r]->data[par][a.rank] + i, v);
    }
    // 2. publish once every payload store of the workgroup has been acknowledged
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < (uint32_t)a.size) __hip_atomic_store(&a.box[tid]->flags[par][a.rank][0], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // 3. wait for every rank's contribution in the local mailbox (bounded: ~2 s of the 100 MHz clock)
    if (!poisoned && tid < (uint32_t)a.size) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(&mine->flags[par][tid][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                s_ok = 0u;
                break;
            }
        }
    }
    __syncthreads();
    if (tid == 0 && a.inject_seq && seq == a.inject_seq) s_ok = 0u; // injected failure: same path as a real time-out from here on
    __syncthreads();
    if (!s_ok) {
        // gave up (or an earlier exchange did): record the first failing sequence number, poison the result, and still advance the
        // sequence number so that this rank keeps its parity / slot discipline
        for (uint32_t i = tid; i < pairs; i += 256) {
            if (OP == 0) {
                float* out = (float*)buf;
                out[2 * i] = __builtin_nanf("");
                if (bf16_out) bf16_out[2 * i] = 0x7FC0u;
                if (2 * i + 1 < words) {
                    out[2 * i + 1] = __builtin_nanf("");
                    if (bf16_out) bf16_out[2 * i + 1] = 0x7FC0u;
                }
            } else {
                ((u64*)buf)[i] = 0ull;
            }
        }
        __syncthreads();
        if (!poisoned && tid < (uint32_t)a.size) // first failure on this rank: tell the whole group (own word included)
            __hip_atomic_store(&a.box[tid]->error[0], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tid == 0) mine->seq[0] = seq;
        return;
    }
    // 4. reduce in rank order (the same order on every rank: bit-identical results)
    for (uint32_t i = tid; i < pairs; i += 256) {
        u64 v = ld_sys((const u64*)mine->data[par][0] + i);
        if (OP == 0) {
            float lo = bits_to_f32((uint32_t)v), hi = bits_to_f32((uint32_t)(v >> 32));
            for (int r = 1; r < a.size; ++r) {
                const u64 w = ld_sys((const u64*)mine->data[par][r] + i);
                lo += bits_to_f32((uint32_t)w), hi += bits_to_f32((uint32_t)(w >> 32));
            }
            float* out = (float*)buf;
            out[2 * i] = lo;
            if (bf16_out) bf16_out[2 * i] = f32_to_bf16(lo);
            if (2 * i + 1 < words) {
                out[2 * i + 1] = hi;
                if (bf16_out) bf16_out[2 * i + 1] = f32_to_bf16(hi);
            }
        } else {
            for (int r = 1; r < a.size; ++r) {
                const u64 w = ld_sys((const u64*)mine->data[par][r] + i);
                v = w > v ? w : v;
            }
            ((u64*)buf)[i] = v;
        }
    }
    __syncthreads();
    if (tid == 0) mine->seq[0] = seq;
}
} // namespace

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

uzu_status comm_create_local(int rank, int size, Comm** out) {
    UZU_REQUIRE(size >= 1 && size <= (int)kMaxRanks && rank >= 0 && rank < size, "tp: bad rank %d of %d", rank, size);
    Comm* c = new Comm();
    c->rank = rank, c->size = size;
    *out = c;
    return UZU_OK;
}

void comm_destroy(Comm* c) {
    if (!c) return;
    for (int r = 0; r < c->size; ++r)
        if (r != c->rank && c->p2p.peer[r]) (void)hipIpcCloseMemHandle(c->p2p.peer[r]);
    if (c->p2p.local) (void)hipFree(c->p2p.local);
    if (c->comm && g_api.CommDestroy) g_api.CommDestroy(c->comm);
    delete c;
}

// ---- P2P set-up: (1) every rank creates its mailbox and exports it, (2) the 64-byte handles travel over any host channel
// (torch.distributed all_gather, a pipe ...), (3) every rank opens the others'.
uzu_status p2p_export(Comm* c, uint8_t out_handle[64]) {
    UZU_REQUIRE(c && out_handle, "tp_p2p_export: null argument");
    UZU_REQUIRE(c->size <= (int)kMaxRanks, "tp_p2p: at most %u ranks", kMaxRanks);
    if (!c->p2p.local) {
        void* p = nullptr;
        hipError_t e = hipExtMallocWithFlags(&p, sizeof(Mailbox), hipDeviceMallocUncached);
        if (e != hipSuccess) { // older runtimes: fine-grained device memory has the same visibility guarantees
            (void)hipGetLastError();
            e = hipExtMallocWithFlags(&p, sizeof(Mailbox), hipDeviceMallocFinegrained);
        }
        if (e != hipSuccess) {
            set_error("tp_p2p: mailbox allocation failed: %s", hipGetErrorString(e));
            return UZU_ERR_HIP;
        }
        UZU_HIP_TRY(hipMemset(p, 0, sizeof(Mailbox)));
        UZU_HIP_TRY(hipDeviceSynchronize());
        c->p2p.local = (Mailbox*)p;
    }
    hipIpcMemHandle_t h;
    UZU_HIP_TRY(hipIpcGetMemHandle(&h, c->p2p.local));
    static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
    memcpy(out_handle, &h, 64);
    return UZU_OK;
}
uzu_status p2p_connect(Comm* c, const uint8_t* handles /* [size][64], own entry ignored */) {
    UZU_REQUIRE(c && handles && c->p2p.local, "tp_p2p_connect: export first");
    for (int r = 0; r < c->size; ++r) {
        if (r == c->rank) {
            c->p2p.peer[r] = c->p2p.local;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * 64, 64);
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            set_error("tp_p2p: cannot open rank %d's mailbox: %s", r, hipGetErrorString(e));
            return UZU_ERR_HIP;
        }
        c->p2p.peer[r] = (Mailbox*)p;
    }
    c->p2p.connected = true;
    return UZU_OK;
}
bool p2p_connected(const Comm* c) { return c && c->p2p.connected; }
void p2p_disable(Comm* c) { // every rank of a group must take the same path: disable everywhere when one rank could not connect
    if (c) c->p2p.connected = false;
}
uzu_status p2p_error(Comm* c, uint32_t* out) { // sequence number of the exchange whose bounded wait gave up (0 = none)
    UZU_REQUIRE(c && out && c->p2p.local, "tp_p2p_error: no mailbox");
    UZU_HIP_TRY(hipMemcpy(out, c->p2p.local->error, 4, hipMemcpyDeviceToHost));
    return UZU_OK;
}
// Bounded wait of one exchange: UZU_TP_TIMEOUT_MS (default 20 s).  Ranks are independent host processes and skew by seconds around
// graph builds and first launches; a host barrier in front of the first exchange (bench.py, the tests) keeps the skew below this.
static unsigned long long p2p_timeout_ticks() {
    static const unsigned long long ticks = [] {
        const char* e = getenv("UZU_TP_TIMEOUT_MS");
        const long ms = e && atol(e) > 0 ? atol(e) : 20000;
        return (unsigned long long)ms * 100000ull; // 100 MHz
    }();
    return ticks;
}
// Host sync points of the TP engine (prefill chunk, decode, read_tokens) call this: a bounded wait that gave up anywhere since the
// last check turns into an error status here instead of silently wrong tokens.
uzu_status p2p_check(Comm* c) {
    if (!c || !c->p2p.local || !c->p2p.connected) return UZU_OK; // (after p2p_disable the mailboxes are out of use: an old timeout there does not concern the RCCL path)
    uint32_t e = 0;
    UZU_HIP_TRY(hipMemcpy(&e, c->p2p.local->error, 4, hipMemcpyDeviceToHost));
    if (e) {
        set_error("tp: peer-to-peer exchange %u of rank %d timed out waiting for a peer (UZU_TP_TIMEOUT_MS); results since then are poisoned", e, c->rank);
        return UZU_ERR_HIP;
    }
    return UZU_OK;
}
template <int OP> static uzu_status p2p_launch(Comm* c, hipStream_t s, void* buf, uint32_t count, uint16_t* bf16_out = nullptr) {
    P2PArgs a{};
    for (int r = 0; r < c->size; ++r) a.box[r] = c->p2p.peer[r];
    a.rank = c->rank, a.size = c->size;
    a.timeout_ticks = p2p_timeout_ticks();
    static const uint32_t inject = [] {
        const char* e = getenv("UZU_TP_INJECT_TIMEOUT_AT");
        return e && atol(e) > 0 ? (uint32_t)atol(e) : 0u;
    }();
    a.inject_seq = inject;
    ++c->p2p_calls;
    return launch_check([&] { hipLaunchKernelGGL(p2p_all_reduce_kernel<OP>, dim3(1), dim3(256), 0, s, a, buf, count, bf16_out); }, "tp_p2p_all_reduce");
}
// ranks the RCCL communicator itself reports (ncclCommCount; 0 = no communicator), collectives enqueued through RCCL / the mailboxes
uzu_status comm_stats(Comm* c, uint32_t* rccl_ranks, unsigned long long* rccl_calls, unsigned long long* p2p_calls) {
    UZU_REQUIRE(c, "tp_comm_stats: null communicator");
    int n = 0;
    if (c->comm && g_api.CommCount) UZU_PROPAGATE(check(g_api.CommCount(c->comm, &n), "ncclCommCount"));
    if (rccl_ranks) *rccl_ranks = (uint32_t)n;
    if (rccl_calls) *rccl_calls = c->rccl_calls;
    if (p2p_calls) *p2p_calls = c->p2p_calls;
    return UZU_OK;
}
int comm_rank(const Comm* c) { return c->rank; }
int comm_size(const Comm* c) { return c->size; }

uzu_status all_reduce_sum_f32(Comm* c, hipStream_t s, float* buf, size_t count, uint16_t* bf16_out) {
    if (c->p2p.connected && count <= kMailboxFloats) return p2p_launch<0>(c, s, buf, (uint32_t)count, bf16_out); // decode rows: one hop
    if (c->p2p.connected && !c->comm) {
        // a group without an RCCL communicator (comm_create_local: ranks that share one device, which RCCL refuses): prefill-sized
        // rows go through the mailboxes in slices of one slot.  Slow (one exchange kernel per 32 KB) and only meant for small groups
        // and the single-GPU multi-rank tests; same rank-order sums as the one-hop exchange.
        for (size_t off = 0; off < count; off += kMailboxFloats) {
            const size_t n = count - off < kMailboxFloats ? count - off : kMailboxFloats;
            UZU_PROPAGATE(p2p_launch<0>(c, s, buf + off, (uint32_t)n, bf16_out ? bf16_out + off : nullptr));
        }
        return UZU_OK;
    }
    UZU_REQUIRE(c->comm, "tp: no RCCL communicator for a %zu-float all-reduce", count);
    ++c->rccl_calls;
    UZU_PROPAGATE(check(g_api.AllReduce(buf, buf, count, ncclFloat32, ncclSum, c->comm, s), "ncclAllReduce(sum,f32)"));
    return bf16_out ? cast_f32_bf16(s, buf, bf16_out, count) : UZU_OK;
}
uzu_status all_reduce_max_u64(Comm* c, hipStream_t s, unsigned long long* buf, size_t count) {
    if (c->p2p.connected && count * 2 <= kMailboxFloats) return p2p_launch<1>(c, s, buf, (uint32_t)count);
    UZU_REQUIRE(c->comm, "tp: no RCCL communicator");
    ++c->rccl_calls;
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
uzu_status keys_from_tokens(hipStream_t s, const uint16_t* logits, size_t row_stride, const uint32_t* local_tokens, uint32_t vocab_offset, unsigned long long* keys, uint32_t rows) {
    return launch_check([&] { hipLaunchKernelGGL(keys_from_tokens_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, logits, row_stride, local_tokens, vocab_offset, keys, rows); }, "tp_keys_from_tokens");
}
uzu_status tokens_from_keys(hipStream_t s, const unsigned long long* keys, uint32_t* out_tokens, uint32_t rows) {
    return launch_check([&] { hipLaunchKernelGGL(tokens_from_keys_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, keys, out_tokens, rows); }, "tp_tokens_from_keys");
}
uzu_status gather_logits(Comm* c, hipStream_t s, const uint16_t* local, uint32_t local_n, uint32_t vocab_offset, uint32_t vocab, uint32_t rows, float* full_f32, uint16_t* full_bf16) {
    UZU_PROPAGATE(launch_check([&] { hipLaunchKernelGGL(scatter_shard_kernel, dim3(64, rows), dim3(256), 0, s, local, local_n, vocab_offset, vocab, full_f32); }, "tp_scatter_shard"));
    return all_reduce_sum_f32(c, s, full_f32, (size_t)rows * vocab, full_bf16);
}
uzu_status commit_key(hipStream_t s, const unsigned long long* key, uint32_t* ctx_len, uint32_t* tokens, uint32_t* out_token, uint32_t* sampled) {
    return launch_check([&] { hipLaunchKernelGGL(commit_key_kernel, dim3(1), dim3(1), 0, s, key, ctx_len, tokens, out_token, sampled); }, "tp_commit_key");
}

} // namespace tp
} // namespace uzu
