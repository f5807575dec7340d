/*
 * uzu_hip.h -- C ABI of the MI355X (gfx950) backend for uzu's transformer forward path.
 *
 * This is the drop-in boundary: every entry point replaces one item of the Rust trait surface in
 * /root/reference/crates/backend-uzu/src/backends/common (cited per declaration as BU/...).  A Rust
 * `backends/hip` shim implements `Backend / Context / CommandBuffer / DenseBuffer / Kernels` by
 * forwarding to these functions (see INTEGRATION.md for the binding).  Plain pointers and sizes only.
 *
 * Conventions
 *  - every function returns uzu_status (0 = ok); uzu_hip_last_error() gives the thread-local message
 *    (the reference's `Result<_, B::Error>`; kernel-argument asserts become UZU_ERR_INVALID_ARGUMENT).
 *  - a buffer argument is the reference's `BufferArg = (buffer, byte_offset)`; `buffer == NULL`
 *    means the `#[optional]` argument is absent.
 *  - encode order = execution order: one in-order HIP stream per context (the reference CPU backend
 *    runs command buffers on one worker thread, BU/../cpu/context.rs:20-33).
 *  - element types are DataType values (uzu_dtype); the LM path uses BF16 activations, F32 norm
 *    scales / rope tables / DeltaNet state.
 */
#ifndef UZU_HIP_H
#define UZU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t uzu_status;
enum {
    UZU_OK = 0,
    UZU_ERR_INVALID_ARGUMENT = 1,
    UZU_ERR_UNSUPPORTED = 2,   /* MatmulError::Unsupported* (BU/kernel/matmul/error.rs:9-30) */
    UZU_ERR_HIP = 3,           /* a HIP runtime call failed */
    UZU_ERR_OUT_OF_MEMORY = 4,
    UZU_ERR_STATE = 5          /* command-buffer typestate violated */
};
const char* uzu_hip_last_error(void);

/* crates/backend-uzu/src/data_type.rs:5-35 (declaration order) */
typedef enum {
    UZU_BF16 = 0, UZU_F16 = 1, UZU_F32 = 2, UZU_F64 = 3, UZU_I4 = 4, UZU_U4 = 5, UZU_I8 = 6, UZU_U8 = 7,
    UZU_I16 = 8, UZU_U16 = 9, UZU_I32 = 10, UZU_U32 = 11, UZU_I64 = 12, UZU_U64 = 13
} uzu_dtype;
/* BU/gpu_types/quantization.rs:8-15 */
typedef enum { UZU_QMODE_U4 = 0, UZU_QMODE_I8 = 1, UZU_QMODE_U8 = 2 } uzu_quant_mode;

/* Backend associated consts (BU/backend.rs:13-17) */
#define UZU_HIP_BACKEND_NAME "hip"
#define UZU_HIP_MIN_ALLOCATION_ALIGNMENT 256
#define UZU_HIP_MAX_ALLOCATION_ALIGNMENT 256
#define UZU_HIP_ALLOCATION_GRANULARITY (8u << 20)
#define UZU_HIP_MAX_INLINE_BYTES 4096

typedef struct uzu_hip_context uzu_hip_context;
typedef struct uzu_hip_buffer uzu_hip_buffer;
typedef struct uzu_hip_cmdbuf uzu_hip_cmdbuf;
typedef struct uzu_hip_kernel uzu_hip_kernel;

typedef struct { uzu_hip_buffer* buffer; size_t offset; } uzu_buf;

/* ---------------------------------------------------------------- Context (BU/context.rs:5-48) */
uzu_status uzu_hip_context_create(int32_t device_ordinal, uzu_hip_context** out);   /* Context::new */
void uzu_hip_context_destroy(uzu_hip_context* ctx);
uzu_status uzu_hip_context_peak_memory_usage(uzu_hip_context* ctx, size_t* out);   /* peak_memory_usage */
uzu_status uzu_hip_context_device_name(uzu_hip_context* ctx, char* out, size_t cap);
uzu_status uzu_hip_context_synchronize(uzu_hip_context* ctx);
void* uzu_hip_context_stream(uzu_hip_context* ctx); /* hipStream_t, for interop (rocprof / RCCL) */

/* ------------------------------------------- Buffer / DenseBuffer (BU/buffer/mod.rs:11-17, dense.rs:5-7)
 * Device memory is hipMalloc'ed (never host-mapped: SURVEY.md H3).  `cpu_ptr` semantics are provided
 * by a lazily allocated pinned host mirror plus explicit upload/download (what `Allocation::copyin /
 * as_slice` and `ParameterLeaf::read_allocation` need, BU/allocator/allocator.rs:29-56). */
/* Context::device_capabilities (BU/backends/common/device_capabilities.rs:6-8) */
enum { UZU_DEVICE_CAP_SPARSE_BUFFERS = 1u << 0 };
uzu_status uzu_hip_context_device_capabilities(uzu_hip_context* ctx, uint32_t* out);
/* Context::{enable_capture, start_capture, stop_capture} (context.rs:38-45; used by engine/capture.rs:33,71 and the matmul
 * benches).  Metal writes a .gputrace; here a capture (a) brackets the interval with roctx ranges and resumes / pauses an
 * attached profiler (rocprofv3 --marker-trace; librocprofiler-sdk-roctx is loaded with dlopen, absent library = no-op),
 * (b) turns every push_debug_group into a roctx range, and (c) writes to `trace_path`, at stop, one JSON record per command
 * buffer completed in between: name, debug groups, GPU time.  Calling stop without start, or start twice: UZU_ERR_STATE. */
void uzu_hip_context_enable_capture(void);
uzu_status uzu_hip_context_start_capture(uzu_hip_context* ctx, const char* trace_path);
uzu_status uzu_hip_context_stop_capture(uzu_hip_context* ctx);

uzu_status uzu_hip_buffer_create(uzu_hip_context* ctx, size_t size, uzu_hip_buffer** out); /* Context::create_buffer */
/* Context::create_sparse_buffer + SparseBuffer::{map, unmap, page_size_bytes} (context.rs:31-34, buffer/sparse.rs:5-19):
 * `capacity` bytes of reserved virtual address space (rounded up to whole pages; uzu_hip_buffer_size returns the rounded
 * size, total_pages = size / page_size), physical pages attached on demand with the HIP virtual-memory API.  gpu_ptr is
 * stable, so kernels encoded against the buffer see whatever is mapped when they run -- the reference grows its KV caches
 * this way (mixer/attention/state.rs:144-170).  The handle is an ordinary uzu_hip_buffer for uzu_buf / destroy / upload /
 * download (mapped ranges only); cpu_ptr is UZU_ERR_UNSUPPORTED.  Pages are [first_page, end_page); mapping a mapped page or
 * unmapping an unmapped one is a no-op.  UZU_ERR_UNSUPPORTED when device_capabilities lacks UZU_DEVICE_CAP_SPARSE_BUFFERS. */
uzu_status uzu_hip_sparse_buffer_create(uzu_hip_context* ctx, size_t capacity, uzu_hip_buffer** out);
uzu_status uzu_hip_sparse_buffer_map(uzu_hip_buffer* buf, size_t first_page, size_t end_page);
uzu_status uzu_hip_sparse_buffer_unmap(uzu_hip_buffer* buf, size_t first_page, size_t end_page);
size_t uzu_hip_sparse_buffer_page_size(const uzu_hip_buffer* buf);
void uzu_hip_buffer_destroy(uzu_hip_buffer* buf);
uint64_t uzu_hip_buffer_gpu_ptr(const uzu_hip_buffer* buf);   /* Buffer::gpu_ptr */
size_t uzu_hip_buffer_size(const uzu_hip_buffer* buf);        /* Buffer::size */
uzu_status uzu_hip_buffer_cpu_ptr(uzu_hip_buffer* buf, void** out);  /* DenseBuffer::cpu_ptr (pinned mirror) */
uzu_status uzu_hip_buffer_flush_to_device(uzu_hip_buffer* buf, size_t offset, size_t size);  /* mirror -> device */
uzu_status uzu_hip_buffer_fetch_from_device(uzu_hip_buffer* buf, size_t offset, size_t size); /* device -> mirror (syncs) */
uzu_status uzu_hip_buffer_upload(uzu_hip_buffer* buf, size_t offset, const void* src, size_t size);   /* stream-ordered, staged */
uzu_status uzu_hip_buffer_download(uzu_hip_buffer* buf, size_t offset, void* dst, size_t size);       /* waits for the stream */

/* ------------------------------------- CommandBuffer typestate (BU/command_buffer.rs:5-125)
 * Initial -> Encoding -> Executable -> Pending -> Completed, checked at run time.
 * UZU_CMDBUF_GRAPH: encoded work is captured into a hipGraph at end_encoding and launched by submit
 * (and may be re-submitted: an extension used for launch-bound decode loops); otherwise kernels are
 * enqueued on the context stream as they are encoded and submit only records the end event. */
enum { UZU_CMDBUF_EAGER = 0, UZU_CMDBUF_GRAPH = 1 };
uzu_status uzu_hip_cmdbuf_create(uzu_hip_context* ctx, const char* name, uint32_t flags, uzu_hip_cmdbuf** out);
uzu_status uzu_hip_cmdbuf_start_encoding(uzu_hip_cmdbuf* cb);
uzu_status uzu_hip_cmdbuf_encode_copy(uzu_hip_cmdbuf* cb, uzu_buf src, uzu_buf dst, size_t size);
uzu_status uzu_hip_cmdbuf_encode_fill(uzu_hip_cmdbuf* cb, uzu_buf dst, size_t size, uint8_t value);
uzu_status uzu_hip_cmdbuf_encode_barrier(uzu_hip_cmdbuf* cb); /* no-op: in-order stream (as cpu/metal do) */
uzu_status uzu_hip_cmdbuf_push_debug_group(uzu_hip_cmdbuf* cb, const char* name);
uzu_status uzu_hip_cmdbuf_pop_debug_group(uzu_hip_cmdbuf* cb);
uzu_status uzu_hip_cmdbuf_end_encoding(uzu_hip_cmdbuf* cb);
uzu_status uzu_hip_cmdbuf_submit(uzu_hip_cmdbuf* cb);
uzu_status uzu_hip_cmdbuf_wait_until_completed(uzu_hip_cmdbuf* cb);
uzu_status uzu_hip_cmdbuf_gpu_execution_time_ns(uzu_hip_cmdbuf* cb, uint64_t* out); /* CommandBufferCompleted */
void uzu_hip_cmdbuf_destroy(uzu_hip_cmdbuf* cb);

void uzu_hip_kernel_destroy(uzu_hip_kernel* k);

/* ================================================================= Kernels (BU/kernel/mod.rs:16-25)
 * `_create` = `XxxKernel::new(context, <type params as DataType>, <#[specialize] params>)`
 * `_encode` = `XxxKernel::encode(<args in declaration order>, encoder)`  (SURVEY.md Appendix A). */

/* ---- MatmulKernel (BU/kernel/matmul/kernel.rs:12-43; arguments.rs:4-15; matmul_a.rs; matmul_b.rs; d_ops.rs) */
typedef enum { UZU_MATMUL_B_FULL_PRECISION = 0, UZU_MATMUL_B_SCALE_BIAS = 1, UZU_MATMUL_B_SCALE_ZERO_POINT = 2,
               UZU_MATMUL_B_SCALE_SYMMETRIC = 3 } uzu_matmul_b_kind;
typedef struct {
    /* MatmulA::FullPrecision { values, offset(elements) }; for MatmulA::Int8Symmetric see a_kind at the end of the struct */
    uzu_buf a;
    size_t a_offset_elements;
    /* MatmulB */
    uint32_t b_kind;          /* uzu_matmul_b_kind */
    uzu_buf b;                /* codes [n, k/pack] or full precision */
    uzu_buf scales;           /* weights dtype [n, ceil(k/g)] */
    uzu_buf biases;           /* ScaleBias */
    uzu_buf zero_points;      /* ScaleZeroPoint */
    uint32_t mode;            /* uzu_quant_mode */
    uint32_t group_size;
    uint32_t signed_codes;
    uint32_t has_b_leading_dimension, b_leading_dimension;
    uint32_t b_transpose;
    uzu_buf d;
    /* MatmulDOps */
    float ab_scale;
    uint32_t accumulate;
    uzu_buf bias;             /* weights dtype [n] */
    uzu_buf rht_factors;      /* i32 [n] sign factors: output RHT in place on D after the store, THEN the bias (kernel.rs:296-303) */
    uint32_t has_soft_cap;
    float soft_cap;
    uzu_buf gather_indices;   /* u32 [m,n] */
    uint32_t m, n, k;
    /* MatmulA (matmul_a.rs:3-14): a_kind 0 = FullPrecision (a, a_offset_elements above); 1 = Int8Symmetric { values = a (i8
     * [m,k]), scales = a_scales (f32 [m, k / a_group_size]), group_sums = a_group_sums (i32, optional: accepted, the kernel sums on
     * the fly), group_size = a_group_size (32 / 64 / 128; weight groups 32 / 64 / 128) } -- else MatmulError::IncompatibleA. */
    uint32_t a_kind;
    uzu_buf a_scales;
    uzu_buf a_group_sums;
    uint32_t a_group_size;
} uzu_matmul_arguments;
uzu_status uzu_hip_matmul_create(uzu_hip_context* ctx, uint32_t weights_dt, uint32_t input_dt, uint32_t output_dt,
                                 uzu_hip_kernel** out);
uzu_status uzu_hip_matmul_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, const uzu_matmul_arguments* args);
/* MatmulKernel::{a8_activation_plan, select_activation_format} (backends/common/kernel/matmul/kernel.rs:28-42): the backend's say on the
 * activation format of a linear layer, asked once per shape by the HybridSpec linear wrappers (MatmulShape: routing.rs:20-33).
 * a8_activation_plan: *has_plan = 0 (None) when the backend cannot run the shape with symmetric int8 activations; otherwise the activation
 * scale group and (prologues with an offset term) the group-sum group ActivationTransform must produce.  select_activation_format:
 * UZU_ACTIVATION_FORMAT_BF16 / _INT8 for the shape as it would run with bf16 activations.  The HIP backend plans A8 for every quantised B it
 * supports and SELECTS Bf16 throughout: measured on MI355X the int8-activation GEMM runs at the bf16-activation GEMM's rate. */
typedef struct {
    uint32_t m, n, k;
    uint32_t b_transpose, has_b_leading_dimension, b_leading_dimension;
    uint32_t b_kind;       /* uzu_matmul_b_kind: the B prologue (GemmBPrologueKind) */
    uint32_t b_bits, b_group_size, signed_codes;
    uint32_t a_full_precision, gathered;
} uzu_matmul_shape;
typedef struct {
    uint32_t activation_group_size;
    uint32_t has_sum_group_size, sum_group_size;
} uzu_a8_activation_plan;
typedef enum { UZU_ACTIVATION_FORMAT_BF16 = 0, UZU_ACTIVATION_FORMAT_INT8 = 1 } uzu_activation_format;
uzu_status uzu_hip_matmul_a8_activation_plan(uzu_hip_kernel* k, const uzu_matmul_shape* shape, uint32_t* has_plan, uzu_a8_activation_plan* out);
uzu_status uzu_hip_matmul_select_activation_format(uzu_hip_kernel* k, const uzu_matmul_shape* bf16_shape, uint32_t* format_out);

/* ---- ActivationTransform (cpu/kernel/activation_transform/activation_transform.rs:43-60): randomised Hadamard transform over
 * stripes of 32 elements with +-1 factors (i32 [element_count]), optionally followed by symmetric int8 quantisation.
 * new(context, T, ops, in_place, activation_scale_group_size, sum_group_size); encode(input?, fp_out?, q_out?, scales_out?,
 * group_sums_out?, rht_factors, batch_size, element_count).  Optional buffers follow the #[optional] conditions of the
 * reference: input iff !in_place; fp_out iff ops is an RHT op; q_out / scales_out iff a Quantize op; group_sums_out iff
 * QuantizeWithGroupSums. */
typedef enum { UZU_ACTIVATION_TRANSFORM_INPUT_RHT = 0, UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT = 1, UZU_ACTIVATION_TRANSFORM_QUANTIZE = 2,
               UZU_ACTIVATION_TRANSFORM_QUANTIZE_WITH_GROUP_SUMS = 3 } uzu_activation_transform_op; /* gpu_types/activation_transform.rs */
uzu_status uzu_hip_activation_transform_create(uzu_hip_context* ctx, uint32_t t, uint32_t ops, uint32_t in_place,
                                               uint32_t activation_scale_group_size, uint32_t sum_group_size, uzu_hip_kernel** out);
uzu_status uzu_hip_activation_transform_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf fp_out, uzu_buf q_out,
                                               uzu_buf scales_out, uzu_buf group_sums_out, uzu_buf rht_factors, uint32_t batch_size,
                                               uint32_t element_count);

/* ---- Normalization (BU/../cpu/kernel/normalization/normalization.rs:7-39) */
uzu_status uzu_hip_normalization_create(uzu_hip_context* ctx, uint32_t input_t, uint32_t affine_t, uint32_t output_t,
                                        uint32_t accum_t, uint32_t in_place, uint32_t subtract_mean,
                                        uint32_t full_layer, uint32_t copy_to_shortcut, uint32_t residual_add,
                                        uint32_t use_hadamard, uint32_t scale_residual_sum, uint32_t scale_output,
                                        uint32_t has_biases, uint32_t has_scales, uzu_hip_kernel** out);
uzu_status uzu_hip_normalization_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf scales,
                                        uzu_buf biases, uzu_buf output, uzu_buf shortcut, uzu_buf hadamard_factors,
                                        uint32_t batch_size, uint32_t element_count, float epsilon,
                                        float scale_offset, float post_layer_scalar);

/* ---- QKVNorm (cpu/kernel/attention/qkv_norm.rs:7-31) */
uzu_status uzu_hip_qkv_norm_create(uzu_hip_context* ctx, uint32_t input_t, uint32_t scale_t, uint32_t output_t,
                                   uint32_t accum_t, uint32_t in_place, uint32_t has_scales, uzu_hip_kernel** out);
uzu_status uzu_hip_qkv_norm_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf qkv_input, uzu_buf scales,
                                   uzu_buf qkv_output, uint32_t batch_size, uint32_t total_heads, uint32_t head_dim,
                                   float epsilon, float scale_offset, uint32_t head_offset, uint32_t head_count,
                                   uint32_t full_layer);

/* ---- AttentionPrepare (cpu/kernel/attention/attention_prepare.rs:34-52) */
uzu_status uzu_hip_attention_prepare_create(uzu_hip_context* ctx, uint32_t element_t, uint32_t rope_t,
                                            uint32_t has_kv, uint32_t has_rope, uzu_hip_kernel** out);
uzu_status uzu_hip_attention_prepare_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf qkv, uzu_buf queries,
                                            uzu_buf keys, uzu_buf values, uzu_buf cosines, uzu_buf sines,
                                            uint32_t num_q_heads, uint32_t num_kv_heads, uint32_t head_dim,
                                            uint32_t rope_dim, uint32_t kv_token_offset, uint32_t batch_dim);

/* ---- AttentionSinglePass / TwoPass1 / TwoPass2 (cpu/kernel/attention/attention_{single,two}_pass.rs)
 * is_trie: `trie` holds one TrieNode {uint32 trie_start, trie_end, height} (BU/gpu_types/trie.rs) per suffix token -- a speculated tree in
 * depth-first order; mask.rs:21-29.  `trie` must be given iff the kernel was created with is_trie. */
typedef struct { uint32_t ring_offset, ring_length; } uzu_ring_params; /* BU/gpu_types/ring.rs */
uzu_status uzu_hip_attention_single_pass_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_dim,
                                                uint32_t has_sinks, uint32_t is_kv_cache_ring, uint32_t is_causal,
                                                uint32_t is_trie, uint32_t is_sliding_window, uzu_hip_kernel** out);
uzu_status uzu_hip_attention_single_pass_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf queries, uzu_buf keys,
                                                uzu_buf values, uzu_buf out, uint32_t gqa_factor,
                                                uint32_t sequence_length, uint32_t k_head_stride,
                                                uint32_t k_seq_stride, uint32_t v_head_stride, uint32_t v_seq_stride,
                                                uzu_ring_params ring_params, float scale, uzu_buf trie,
                                                uint32_t sliding_window_size, uzu_buf sinks, uint32_t num_heads,
                                                uint32_t suffix_length);
uzu_status uzu_hip_attention_two_pass1_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_dim, uint32_t has_sinks,
                                              uint32_t is_kv_cache_ring, uint32_t is_causal, uint32_t is_trie,
                                              uint32_t is_sliding_window, uzu_hip_kernel** out);
uzu_status uzu_hip_attention_two_pass1_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf queries, uzu_buf keys,
                                              uzu_buf values, uzu_buf out_partials, uzu_buf sums, uzu_buf maxs,
                                              uint32_t gqa_factor, uint32_t sequence_length, uint32_t k_head_stride,
                                              uint32_t k_seq_stride, uint32_t v_head_stride, uint32_t v_seq_stride,
                                              uzu_ring_params ring_params, float scale, uint32_t num_heads,
                                              uint32_t suffix_length, uzu_buf trie, uint32_t sliding_window_size,
                                              uzu_buf sinks);
uzu_status uzu_hip_attention_two_pass2_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_dim,
                                              uzu_hip_kernel** out);
uzu_status uzu_hip_attention_two_pass2_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf partials, uzu_buf sums,
                                              uzu_buf maxs, uzu_buf out, uint32_t num_heads, uint32_t suffix_length);

/* ---- AttentionGemmCore, a manual trait (BU/backends/common/kernel/attention_gemm/kernel.rs:8-24): the prefill attention
 * core on the matrix cores.  `uzu_attention_core_arguments` = AttentionCoreNewArguments (encodable_block/mixer/attention/
 * core/mod.rs:17-28; Option<T> as has_* + value).  is_supported is the trait's static query (causal bf16, head_dim 64 / 128 /
 * 256, with or without attention sinks, a sliding window, a ring KV prefix; any GQA factor; NOT the speculated-tree mask, which stays on
 * the single- / two-pass cores).  encode = AttentionCoreEncodeArguments
 * (core/mod.rs:30-38) for AttentionStateType::Full { length = prefix_length }: queries [q_heads, suffix, hd] as written by
 * AttentionPrepare, keys / values = the KV cache [tokens, kv_heads, hd] already holding the suffix rows, out [suffix, q_heads,
 * hd] (the Rust trait returns a fresh Allocation; in C the caller passes it). */
typedef struct {
    uint32_t head_dim, num_groups, num_q_heads;
    uint32_t has_sinks, is_kv_cache_ring, is_causal, is_trie;
    uint32_t has_sliding_window, sliding_window_size;
    uint32_t has_scale;
    float scale;
    uint32_t data_type;
} uzu_attention_core_arguments;
uzu_status uzu_hip_attention_gemm_is_supported(uzu_hip_context* ctx, const uzu_attention_core_arguments* arguments, uint32_t* out);
uzu_status uzu_hip_attention_gemm_create(uzu_hip_context* ctx, const uzu_attention_core_arguments* arguments, uzu_hip_kernel** out);
uzu_status uzu_hip_attention_gemm_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf queries, uzu_buf keys, uzu_buf values, uzu_buf out,
                                         uint32_t prefix_length, uint32_t suffix_length);
/* The full AttentionCoreEncodeArguments: `sinks` (bf16 [q_heads], iff has_sinks) and the state type -- Full { length } (is_ring = 0, `length`
 * = the prefix length) or Ring { offset, length, max_length } (mixer/attention/state.rs:16-55: keys / values hold `ring_max_length` ring
 * slots followed by the suffix rows). */
uzu_status uzu_hip_attention_gemm_encode_state(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf queries, uzu_buf keys, uzu_buf values, uzu_buf sinks,
                                               uzu_buf out, uint32_t is_ring, uint32_t length, uint32_t ring_offset, uint32_t ring_max_length,
                                               uint32_t suffix_length);

/* ---- KVCacheUpdate (cpu/kernel/attention/kv_cache_update.rs:7-15); copies are inline constants */
typedef struct { uint32_t source, destination; } uzu_kv_copy; /* BU/gpu_types/kv_cache_update.rs */
uzu_status uzu_hip_kv_cache_update_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_kv_cache_update_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_place_keys,
                                          uzu_buf in_place_values, const uzu_kv_copy* copies, uint32_t copy_count,
                                          uint32_t element_dim);

/* ---- SigmoidGate (cpu/kernel/attention/sigmoid_gate.rs:7-12) */
uzu_status uzu_hip_sigmoid_gate_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_sigmoid_gate_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf gate, uzu_buf output,
                                       uint32_t total_elements);

/* ---- GatedActMul (cpu/kernel/gated_act_mul/gated_act_mul.rs:13-35).  ops: 0 FullPrecision, 1 Quantize,
 *      2 QuantizeWithGroupSums; the quantized ops require use_hadamard (gated_act_mul.rs:36-42) and
 *      activation_scale_group_size (and sum_group_size for op 2) in {32, 64, 128, 256}.  RHT variants: the rounded
 *      gated product goes through ActivationTransform{InputRht | Quantize | QuantizeWithGroupSums}; bit-exact. */
uzu_status uzu_hip_gated_act_mul_create(uzu_hip_context* ctx, uint32_t t, uint32_t ops, uint32_t interleaved,
                                        uint32_t use_hadamard, uint32_t activation_scale_group_size,
                                        uint32_t sum_group_size, uzu_hip_kernel** out);
uzu_status uzu_hip_gated_act_mul_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf act_operand,
                                        uzu_buf value_operand, uzu_buf fp_out, uzu_buf q_out, uzu_buf scales_out,
                                        uzu_buf group_sums_out, uzu_buf hadamard_factors, uint32_t gated_dim,
                                        uint32_t batch_dim, uint32_t value_offset, uint32_t value_row_stride,
                                        uint32_t act_type);

/* ---- embeddings (cpu/kernel/embedding/{quant_embedding,full_precision_embedding}.rs) */
uzu_status uzu_hip_quantized_embedding_lookup_create(uzu_hip_context* ctx, uint32_t t, uint32_t group_size,
                                                     uint32_t quantization_mode, uint32_t quantization_method,
                                                     uint32_t use_hadamard, uzu_hip_kernel** out);
uzu_status uzu_hip_quantized_embedding_lookup_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf token_ids,
                                                     uzu_buf weights, uzu_buf scales, uzu_buf zero_points,
                                                     uzu_buf biases, uzu_buf output, uzu_buf output_hadamard_factors,
                                                     uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim,
                                                     float input_scale);
uzu_status uzu_hip_full_precision_embedding_lookup_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_full_precision_embedding_lookup_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf token_ids,
                                                          uzu_buf weights, uzu_buf output, uint32_t batch_size,
                                                          uint32_t vocab_size, uint32_t model_dim, float input_scale);

/* ---- LogitTransform / TensorAddBias / TensorAddScale / TensorAddSwap / TensorCopy */
uzu_status uzu_hip_logit_transform_create(uzu_hip_context* ctx, uint32_t t, uint32_t has_soft_cap,
                                          uzu_hip_kernel** out);
uzu_status uzu_hip_logit_transform_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf logits, uint32_t length,
                                          float scale, float soft_cap);
uzu_status uzu_hip_tensor_add_bias_create(uzu_hip_context* ctx, uint32_t t, uint32_t bias_t, uint32_t in_place,
                                          uzu_hip_kernel** out);
uzu_status uzu_hip_tensor_add_bias_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf bias,
                                          uzu_buf output, uint32_t num_cols, uint32_t length);
uzu_status uzu_hip_tensor_add_scale_create(uzu_hip_context* ctx, uint32_t t, uint32_t in_place, uzu_hip_kernel** out);
uzu_status uzu_hip_tensor_add_scale_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf bias,
                                           uzu_buf output, uint32_t num_cols, uint32_t length, float scale);
uzu_status uzu_hip_tensor_add_swap_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_tensor_add_swap_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf skip, uzu_buf main_buffer,
                                          uint32_t length);
uzu_status uzu_hip_tensor_copy_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_tensor_copy_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf src, uzu_buf dst,
                                      uint32_t length);

/* ---- UnifiedSampling (cpu/kernel/sampling/unified_sampling.rs:13-32): greedy (argmax, ties -> lowest id) and
 *      every stochastic specialisation -- bitmask, temperature, top-k, top-p, min-p, Gumbel-max with the
 *      Philox4x32-10 noise of (seed, index); token-identical to the CPU kernel (tests/golden/sampling.json). */
uzu_status uzu_hip_unified_sampling_create(uzu_hip_context* ctx, uint32_t t, uint32_t is_stochastic,
                                           uint32_t has_bitmask, uint32_t has_temperature, uint32_t has_top_k,
                                           uint32_t has_top_p, uint32_t has_min_p, uzu_hip_kernel** out);
uzu_status uzu_hip_unified_sampling_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf logits, uzu_buf output,
                                           uzu_buf seeds, uzu_buf bitmask, float temperature, uint32_t top_k,
                                           float top_p, float min_p, uint32_t vocab_size, uint32_t batch_size);

/* ---- Gated DeltaNet (cpu/kernel/gdn/ *.rs with the Metal buffer types, metal/kernel/gdn/update.metal:19-31) */
uzu_status uzu_hip_delta_net_conv_update_create(uzu_hip_context* ctx, uint32_t t, uint32_t has_bias,
                                                uzu_hip_kernel** out);
uzu_status uzu_hip_delta_net_conv_update_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf conv_weight,
                                                uzu_buf bias, uzu_buf in_out, uzu_buf state, uint32_t kernel_size,
                                                uint32_t conv_dim, uint32_t state_stride);
uzu_status uzu_hip_delta_net_update_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_k_dim,
                                           uzu_hip_kernel** out);
uzu_status uzu_hip_delta_net_update_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_proj, uzu_buf a_log,
                                           uzu_buf dt_bias, uzu_buf norm_weight, uzu_buf state, uzu_buf out,
                                           uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim,
                                           uint32_t key_dim, uint32_t value_dim, float norm_epsilon);
uzu_status uzu_hip_conv1d_pack_create(uzu_hip_context* ctx, uint32_t state_t, uint32_t input_t, uzu_hip_kernel** out);
uzu_status uzu_hip_conv1d_pack_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf state_in, uzu_buf x,
                                      uzu_buf padded, uint32_t state_stride, uint32_t row_stride, uint32_t suffix_len,
                                      uint32_t num_channels);
uzu_status uzu_hip_delta_net_conv_scan_create(uzu_hip_context* ctx, uint32_t t, uint32_t has_bias,
                                              uzu_hip_kernel** out);
uzu_status uzu_hip_delta_net_conv_scan_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf conv_padded,
                                              uzu_buf conv_weight, uzu_buf bias, uzu_buf in_proj, uzu_buf state_out,
                                              uint32_t suffix_len, uint32_t kernel_size, uint32_t row_stride,
                                              uint32_t state_stride, uint32_t conv_dim, uint32_t out_stride);
uzu_status uzu_hip_delta_net_prefill_prep_create(uzu_hip_context* ctx, uint32_t t, uint32_t qk_t, uint32_t head_k_dim,
                                                 uint32_t write_log_decay, uint32_t write_compact_v,
                                                 uzu_hip_kernel** out);
uzu_status uzu_hip_delta_net_prefill_prep_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_proj, uzu_buf a_log,
                                                 uzu_buf dt_bias, uzu_buf q_norm_out, uzu_buf k_norm_out,
                                                 uzu_buf compact_v_out, uzu_buf beta_out, uzu_buf decay_out,
                                                 uint32_t num_v_heads, uint32_t num_k_heads, uint32_t key_dim,
                                                 uint32_t value_dim, uint32_t suffix_len);
uzu_status uzu_hip_delta_net_prefill_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_k_dim,
                                            uzu_hip_kernel** out);
uzu_status uzu_hip_delta_net_prefill_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf q_norm, uzu_buf k_norm,
                                            uzu_buf beta, uzu_buf decay, uzu_buf in_proj, uzu_buf state, uzu_buf out,
                                            uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim,
                                            uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len,
                                            uint32_t num_dv_groups);
uzu_status uzu_hip_delta_net_norm_gate_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_delta_net_norm_gate_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_out, uzu_buf in_proj,
                                              uzu_buf norm_weight, uint32_t num_v_heads, uint32_t head_v_dim,
                                              uint32_t value_dim, uint32_t conv_dim, uint32_t total_proj_dim,
                                              float norm_epsilon, uint32_t suffix_len);

/* ---- Gated DeltaNet over a speculated token tree (speculative decoding; BU/src/backends/cpu/kernel/gdn/tree_verify/) ----
 * ConvTreeScanKernel::{new(context, data_type, kernel_size, has_bias), encode}  (tree_verify/conv_scan.rs:13-72): the causal conv walked
 * through `parents` (i32 [suffix_len], -1 = child of the accepted context) instead of the previous row; suffix_state f32 [suffix_len,
 * conv_dim, kernel_size - 1] = the conv state after the root path that ends in each node.  Rows are DeltaNet in-projection rows with
 * 128-wide heads (the only shape the block instantiates).
 * DeltaNetPrefillPrepKernel's tree instantiation (T = QKT = BF16, write_log_decay, write_compact_v: delta_net.rs:257-265) is
 * uzu_hip_delta_net_prefill_prep_create(.., UZU_BF16, UZU_BF16, 128, 1, 1, ..) above.
 * DeltaNetTreeVerify::{new(TreeVerifyNewArguments), encode(TreeVerifyEncodeArguments)}  (backends/common/kernel/delta_net_tree_verify.rs,
 * encodable_block/mixer/delta_net/tree_verify.rs:7-26; the Metal backend composes it from BuildTreePrefix / BuildTreeGram / TreeUpdateSolve /
 * BuildTreeOut, metal/kernel/gdn/tree_verify.rs:92-187 -- here it is ONE kernel whose results are bit-identical to the CPU kernels'
 * composition): q / k bf16 [tree, k_heads, 128], v bf16 [tree, v_heads, 128], trie u32 [tree, 3], log_decay / beta f32 [tree, v_heads],
 * h0 f32 [v_heads, 128, 128] -> output bf16 [tree, v_heads, 128].  tree_size <= 32.
 * StateAdvanceKernel::{new(context, data_type, head_k_dim, num_v_heads, num_k_heads), encode}  (tree_verify/state_advance.rs:9-59). */
uzu_status uzu_hip_conv_tree_scan_create(uzu_hip_context* ctx, uint32_t t, uint32_t kernel_size, uint32_t has_bias, uzu_hip_kernel** out);
uzu_status uzu_hip_conv_tree_scan_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_proj, uzu_buf conv_weight, uzu_buf bias /* optional */,
                                         uzu_buf base_state, uzu_buf parents, uzu_buf out_proj, uzu_buf suffix_state, uint32_t suffix_len,
                                         uint32_t total_proj_dim, uint32_t conv_dim);
uzu_status uzu_hip_delta_net_tree_verify_create(uzu_hip_context* ctx, uint32_t t, uint32_t num_k_heads, uint32_t num_v_heads, uint32_t head_k_dim,
                                                uint32_t head_v_dim, uzu_hip_kernel** out);
uzu_status uzu_hip_delta_net_tree_verify_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf q, uzu_buf k_norm, uzu_buf v, uzu_buf trie, uzu_buf log_decay,
                                                uzu_buf beta, uzu_buf h0, uzu_buf output, uint32_t tree_size);
uzu_status uzu_hip_state_advance_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_k_dim, uint32_t num_v_heads, uint32_t num_k_heads,
                                        uzu_hip_kernel** out);
uzu_status uzu_hip_state_advance_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf k_norm, uzu_buf v, uzu_buf log_decay, uzu_buf beta,
                                        uzu_buf accepted_indices, uzu_buf state, uint32_t accepted_len);

/* ---- the tree speculators' kernels (speculators/dflash_tfm.rs, encodable_block/weaver_*.rs drive them; the drafter MODELS are out of scope) ----
 * AncestorAttentionKernel::{new(context, HEAD_DIM = 128, num_heads), encode}  (cpu/kernel/attention/ancestor_attention.rs:8-139): per row (a node
 * of the drafter's tree) half-rotation RoPE of q / k at position depth + 1, attention over the prefix rows + the node's ancestors' slots + the
 * node itself, then the node's rotated key / value go to its node_kv slot.  prefix_kv bf16: keys [prefix, model_dim] then values; node_kv bf16:
 * keys [capacity, model_dim] then values; current_qkv bf16 [rows, 3 model_dim]; cosines / sines f32 [max_depth + 1, head_dim]; node_metadata u32
 * [MetadataIdx::COUNT, rows].  A row may attend to the slot of an EARLIER row of the same call (the reference's row-by-row order), never a later one.
 * WeaverFrontierSelect / WeaverFrontierInsertChildren / WeaverTopChildren  (cpu/kernel/weaver/ *.rs; layouts: gpu_types/weaver.rs -- field f of
 * slot s at [f * capacity + s]): bit-identical to the CPU kernels; shapes outside the reference kernels' own guards are no-ops, as there. */
uzu_status uzu_hip_ancestor_attention_create(uzu_hip_context* ctx, uint32_t head_dim, uint32_t num_heads, uzu_hip_kernel** out);
uzu_status uzu_hip_ancestor_attention_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf prefix_kv, uzu_buf node_kv, uzu_buf current_qkv, uzu_buf cosines, uzu_buf sines,
                                             uzu_buf node_metadata, uzu_buf ancestor_indices, uzu_buf ancestor_counts, uzu_buf node_indices, uzu_buf output, uint32_t rows,
                                             uint32_t prefix_length, uint32_t ancestor_stride, uint32_t node_capacity, uint32_t max_depth, float scale);
uzu_status uzu_hip_weaver_frontier_select_create(uzu_hip_context* ctx, uzu_hip_kernel** out);
uzu_status uzu_hip_weaver_frontier_select_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf frontier, uzu_buf packed_tree, uzu_buf slot_ancestors, uzu_buf node_token_ids,
                                                 uzu_buf node_metadata, uzu_buf node_ancestor_indices, uzu_buf node_valid, uzu_buf candidate_pool_ids,
                                                 uzu_buf candidate_pool_logits, uzu_buf node_candidate_ids, uzu_buf node_candidate_logits, uint32_t frontier_capacity,
                                                 uint32_t tree_slot_count, uint32_t node_count, uint32_t batch_start_slot, uint32_t ancestor_stride, uint32_t max_depth,
                                                 uint32_t lookahead_count, uint32_t candidate_depth_count, uint32_t candidates_per_depth);
uzu_status uzu_hip_weaver_frontier_insert_children_create(uzu_hip_context* ctx, uzu_hip_kernel** out);
uzu_status uzu_hip_weaver_frontier_insert_children_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf packed_tree, uzu_buf node_metadata, uzu_buf node_valid,
                                                          uzu_buf child_ids, uzu_buf child_logprobs, uzu_buf frontier, uint32_t frontier_capacity, uint32_t tree_slot_count,
                                                          uint32_t node_count, uint32_t expand_width);
uzu_status uzu_hip_weaver_top_children_create(uzu_hip_context* ctx, uzu_hip_kernel** out);
uzu_status uzu_hip_weaver_top_children_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf residual_logits, uzu_buf candidate_logits, uzu_buf candidate_ids,
                                              uzu_buf depth_seeds, uzu_buf node_metadata, uzu_buf output_token_ids, uzu_buf output_model_logprobs, uint32_t rows,
                                              uint32_t candidates, uint32_t expand_width, uint32_t vocab_size);

/* ---- measurement probe (bench.py; not part of the forward path) ----
 * The price of one all-to-all dependency edge kept as a kernel boundary inside a replayed hipGraph -- the structure of the batch-1 decode
 * step: `launches` dependent launches of `workgroups` workgroups, each reading the whole `row_bytes` row the previous launch wrote and
 * writing its share of the next one, captured once and replayed `replays` times; *us_per_launch = wall time per launch between two events
 * on the context's stream.  bench.py's roofline.latency_floor = edges per token x this + the read-out's stream time (csrc/k_probe.hip). */
uzu_status uzu_hip_probe_edge_floor(uzu_hip_context* ctx, uint32_t workgroups, uint32_t row_bytes, uint32_t launches, uint32_t replays, float* us_per_launch);

/* ---- Mixture of experts (MoeBlock::encode, encodable_block/mlp/moe/mod.rs:204-350).  BF16 tensors.  Where the reference's CPU kernel has a
 * body the entry follows it (router_topk.rs, counts_offsets_fused.rs, gather.rs, the down pass of experts_two_pass_decode.rs:33-77, finalize.rs); where it is todo!()
 * the Metal shader is the statement of record (scatter_buckets.metal, experts_two_pass_decode.metal pass A).  Metal's tile-map / dispatch-argument kernels
 * (tiles_map.rs, tiles_pass_a.rs) and MoeBlockBasesFromPartials only shape its indirect dispatches and have no counterpart: the expert passes are launched for the row
 * capacity (tokens x active experts), find a row's expert through `row_expert_map` (MoePassABuildRowMap) and stop at the routed count in `sumk_buf`. */
uzu_status uzu_hip_moe_router_top_k_create(uzu_hip_context* ctx, uint32_t scalar_t, uint32_t has_biases, uint32_t has_router_scales, uint32_t has_per_expert_scales,
                                           uint32_t has_router_input_scale, uint32_t normalize_router_input, uzu_hip_kernel** out);
uzu_status uzu_hip_moe_router_top_k_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf weight, uzu_buf bias, uzu_buf topk_ids, uzu_buf topk_probs, uint32_t t, uint32_t d_model,
                                           uint32_t e, uint32_t top_k, uint32_t renorm);
uzu_status uzu_hip_moe_counts_offsets_fused_create(uzu_hip_context* ctx, uzu_hip_kernel** out);
uzu_status uzu_hip_moe_counts_offsets_fused_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf topk_ids, uzu_buf offsets, uzu_buf sum_k_out, uzu_buf partials /* optional */, uint32_t t,
                                                   uint32_t e, uint32_t top_k);
uzu_status uzu_hip_moe_scatter_buckets_map_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_moe_scatter_buckets_map_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf topk_ids, uzu_buf topk_probs, uzu_buf offsets, uzu_buf out_ids, uzu_buf out_probs, uint32_t t,
                                                  uint32_t e, uint32_t top_k, uzu_buf tok2row, uzu_buf row_expert_map);
uzu_status uzu_hip_moe_gather_x_perm_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_moe_gather_x_perm_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf x, uzu_buf bucketed_ids, uzu_buf x_perm, uzu_buf sumk_buf, uint32_t d_model, uint32_t t, uint32_t top_k);
uzu_status uzu_hip_moe_experts_pass_a_create(uzu_hip_context* ctx, uint32_t t, uint32_t gating_sel, uzu_hip_kernel** out);
uzu_status uzu_hip_moe_experts_pass_a_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf x_perm, uzu_buf row_expert_map, uzu_buf sumk_buf, uzu_buf w13_all, uzu_buf up_biases, uzu_buf hidden_out,
                                             uint32_t d_model, uint32_t d_ff, float gate_clip_min, float gate_clip_max, float up_clip_min, float up_clip_max, float silu_alpha, uint32_t capacity);
uzu_status uzu_hip_moe_experts_down_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_moe_experts_down_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf hidden, uzu_buf row_expert_map, uzu_buf sumk_buf, uzu_buf w2_all, uzu_buf down_biases, uzu_buf y_out,
                                           uint32_t d_model, uint32_t d_ff, uint32_t capacity);
uzu_status uzu_hip_moe_finalize_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out);
uzu_status uzu_hip_moe_finalize_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf tok2row, uzu_buf probs, uzu_buf y_partial, uzu_buf y, uint32_t t_count, uint32_t d_model, uint32_t top_k);

#ifdef __cplusplus
}
#endif
#endif /* UZU_HIP_H */
