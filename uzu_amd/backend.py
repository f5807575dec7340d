"""Host-side mirror of the reference's backend trait surface over the HIP C ABI.

Names and argument order follow crates/backend-uzu/src/backends/common (Context, DenseBuffer,
CommandBuffer*/Encoder) and the generated ``XxxKernel::{new, encode}`` signatures
(SURVEY.md Appendix A), so that the parity tests read like the reference's own kernel tests
(crates/backend-uzu/tests/unit/backends/common/kernel/**).  Everything here executes on the GPU
through ``libuzu_hip.so``; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple, Union

import numpy as np

from . import _ffi
from ._ffi import Buf, KVCopy, MatmulArguments, RingParams, UzuHipError, call  # noqa: F401

# DataType (data_type.rs:5-35)
BF16, F16, F32, F64, I4, U4, I8, U8, I16, U16, I32, U32, I64, U64 = range(14)
# QuantizationMode (gpu_types/quantization.rs)
QMODE_U4, QMODE_I8, QMODE_U8 = 0, 1, 2
# MatmulB kinds
B_FULL_PRECISION, B_SCALE_BIAS, B_SCALE_ZERO_POINT, B_SCALE_SYMMETRIC = 0, 1, 2, 3
# ActivationType
SILU, GELU_APPROX, GELU_EXACT, IDENTITY, SOFTPLUS = 0, 1, 2, 3, 4

CMDBUF_EAGER, CMDBUF_GRAPH = 0, 1


class Context:
    """backends/common/context.rs:5-48"""

    NAME = "hip"

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        call("uzu_hip_context_create", C.c_int32(device), C.byref(self._h))

    @classmethod
    def new(cls, device: int = 0) -> "Context":
        return cls(device)

    def close(self):
        if self._h:
            _ffi.lib().uzu_hip_context_destroy(self._h)
            self._h = C.c_void_p()

    def create_buffer(self, size: int) -> "Buffer":
        return Buffer(self, size)

    def create_command_buffer(self, name: Optional[str] = None, graph: bool = False) -> "CommandBuffer":
        return CommandBuffer(self, name, graph)

    def peak_memory_usage(self) -> int:
        out = C.c_size_t()
        call("uzu_hip_context_peak_memory_usage", self._h, C.byref(out))
        return out.value

    def device_name(self) -> str:
        buf = C.create_string_buffer(128)
        call("uzu_hip_context_device_name", self._h, buf, C.c_size_t(128))
        return buf.value.decode()

    def synchronize(self):
        call("uzu_hip_context_synchronize", self._h)

    def device_capabilities(self) -> int:
        out = C.c_uint32()
        call("uzu_hip_context_device_capabilities", self._h, C.byref(out))
        return out.value

    def create_sparse_buffer(self, capacity: int) -> "SparseBuffer":
        return SparseBuffer(self, capacity)

    @staticmethod
    def enable_capture():
        fn = _ffi.lib().uzu_hip_context_enable_capture
        fn.restype, fn.argtypes = None, []
        fn()

    def start_capture(self, trace_path: str):
        call("uzu_hip_context_start_capture", self._h, str(trace_path).encode())

    def stop_capture(self):
        call("uzu_hip_context_stop_capture", self._h)

    @property
    def stream(self) -> int:
        return _ffi.lib().uzu_hip_context_stream(self._h) or 0

    # convenience used by the tests: upload a numpy array into a fresh buffer
    def buffer_from(self, array: np.ndarray) -> "Buffer":
        array = np.ascontiguousarray(array)
        b = self.create_buffer(max(array.nbytes, 1))
        if array.nbytes:
            b.upload(array)
        return b


DEVICE_CAP_SPARSE_BUFFERS = 1


class AttentionCoreArguments(C.Structure):
    """AttentionCoreNewArguments (encodable_block/mixer/attention/core/mod.rs:17-28)"""
    _fields_ = [("head_dim", C.c_uint32), ("num_groups", C.c_uint32), ("num_q_heads", C.c_uint32), ("has_sinks", C.c_uint32),
                ("is_kv_cache_ring", C.c_uint32), ("is_causal", C.c_uint32), ("is_trie", C.c_uint32), ("has_sliding_window", C.c_uint32),
                ("sliding_window_size", C.c_uint32), ("has_scale", C.c_uint32), ("scale", C.c_float), ("data_type", C.c_uint32)]


class Buffer:
    """buffer/mod.rs:11-17 + dense.rs:5-7"""

    def __init__(self, ctx: Context, size: int):
        self.ctx = ctx
        self._h = C.c_void_p()
        call("uzu_hip_buffer_create", ctx._h, C.c_size_t(size), C.byref(self._h))
        self._size = size

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                _ffi.lib().uzu_hip_buffer_destroy(self._h)
        except Exception:
            pass

    def gpu_ptr(self) -> int:
        return _ffi.lib().uzu_hip_buffer_gpu_ptr(self._h)

    def size(self) -> int:
        return self._size

    def upload(self, array: np.ndarray, offset: int = 0):
        array = np.ascontiguousarray(array)
        call("uzu_hip_buffer_upload", self._h, C.c_size_t(offset), C.c_void_p(array.ctypes.data), C.c_size_t(array.nbytes))

    def download(self, dtype, count: Optional[int] = None, offset: int = 0) -> np.ndarray:
        dtype = np.dtype(dtype)
        if count is None:
            count = (self._size - offset) // dtype.itemsize
        out = np.empty(count, dtype=dtype)
        call("uzu_hip_buffer_download", self._h, C.c_size_t(offset), C.c_void_p(out.ctypes.data), C.c_size_t(out.nbytes))
        return out


class SparseBuffer(Buffer):
    """buffer/sparse.rs:5-19: reserved address space, pages mapped on demand."""

    def __init__(self, ctx: Context, capacity: int):
        self.ctx = ctx
        self._h = C.c_void_p()
        call("uzu_hip_sparse_buffer_create", ctx._h, C.c_size_t(capacity), C.byref(self._h))
        self._size = _ffi.lib().uzu_hip_buffer_size(self._h)

    def page_size_bytes(self) -> int:
        fn = _ffi.lib().uzu_hip_sparse_buffer_page_size
        fn.restype, fn.argtypes = C.c_size_t, [C.c_void_p]
        return fn(self._h)

    def total_pages(self) -> int:
        return self._size // self.page_size_bytes()

    def map(self, first_page: int, end_page: int):
        call("uzu_hip_sparse_buffer_map", self._h, C.c_size_t(first_page), C.c_size_t(end_page))

    def unmap(self, first_page: int, end_page: int):
        call("uzu_hip_sparse_buffer_unmap", self._h, C.c_size_t(first_page), C.c_size_t(end_page))


BufArg = Union[None, Buffer, Tuple[Buffer, int]]


def _buf(x: BufArg) -> Buf:
    if x is None:
        return Buf(None, 0)
    if isinstance(x, tuple):
        return Buf(x[0]._h, x[1])
    return Buf(x._h, 0)


class CommandBuffer:
    """command_buffer.rs:5-125 (typestate checked at run time by the library); also plays Encoder."""

    def __init__(self, ctx: Context, name: Optional[str] = None, graph: bool = False):
        self.ctx = ctx
        self._h = C.c_void_p()
        call("uzu_hip_cmdbuf_create", ctx._h, (name or "").encode(), C.c_uint32(CMDBUF_GRAPH if graph else CMDBUF_EAGER), C.byref(self._h))

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                _ffi.lib().uzu_hip_cmdbuf_destroy(self._h)
        except Exception:
            pass

    def start_encoding(self) -> "CommandBuffer":
        call("uzu_hip_cmdbuf_start_encoding", self._h)
        return self

    def encode_copy(self, src: BufArg, dst: BufArg, size: int):
        call("uzu_hip_cmdbuf_encode_copy", self._h, _buf(src), _buf(dst), C.c_size_t(size))

    def encode_fill(self, dst: BufArg, size: int, value: int):
        call("uzu_hip_cmdbuf_encode_fill", self._h, _buf(dst), C.c_size_t(size), C.c_uint8(value))

    def encode_barrier(self):
        call("uzu_hip_cmdbuf_encode_barrier", self._h)

    def push_debug_group(self, name: str):
        call("uzu_hip_cmdbuf_push_debug_group", self._h, name.encode())

    def pop_debug_group(self):
        call("uzu_hip_cmdbuf_pop_debug_group", self._h)

    def end_encoding(self) -> "CommandBuffer":
        call("uzu_hip_cmdbuf_end_encoding", self._h)
        return self

    def submit(self) -> "CommandBuffer":
        call("uzu_hip_cmdbuf_submit", self._h)
        return self

    def wait_until_completed(self) -> "CommandBuffer":
        call("uzu_hip_cmdbuf_wait_until_completed", self._h)
        return self

    def gpu_execution_time(self) -> float:
        ns = C.c_uint64()
        call("uzu_hip_cmdbuf_gpu_execution_time_ns", self._h, C.byref(ns))
        return ns.value * 1e-9


Encoder = CommandBuffer


def _u(x) -> C.c_uint32:
    return C.c_uint32(int(x))


def _f(x) -> C.c_float:
    return C.c_float(float(x))


class _Kernel:
    _create = ""
    _encode = ""

    def __init__(self, ctx: Context, *params):
        self.ctx = ctx
        self._h = C.c_void_p()
        call(self._create, ctx._h, *[_u(p) for p in params], C.byref(self._h))

    @classmethod
    def new(cls, ctx: Context, *params):
        return cls(ctx, *params)

    def __del__(self):
        try:
            if self._h and self.ctx._h:  # a kernel dies before its context (it may own device scratch on the context's stream)
                _ffi.lib().uzu_hip_kernel_destroy(self._h)
        except Exception:
            pass

    def _enc(self, encoder: CommandBuffer, *args):
        call(self._encode, self._h, encoder._h, *args)


class MatmulShape(C.Structure):
    """MatmulShape (backends/common/kernel/matmul/routing.rs:20-33) as include/uzu_hip.h::uzu_matmul_shape lays it out"""
    _fields_ = [(n, C.c_uint32) for n in ("m", "n", "k", "b_transpose", "has_b_leading_dimension", "b_leading_dimension", "b_kind", "b_bits", "b_group_size",
                                           "signed_codes", "a_full_precision", "gathered")]


class A8ActivationPlan(C.Structure):
    _fields_ = [("activation_group_size", C.c_uint32), ("has_sum_group_size", C.c_uint32), ("sum_group_size", C.c_uint32)]


ACTIVATION_FORMAT_BF16, ACTIVATION_FORMAT_INT8 = 0, 1


class MatmulKernel(_Kernel):
    """kernel/matmul/kernel.rs:12-43: new(context, weights_dt, input_dt, output_dt); encode(MatmulArguments, encoder);
    a8_activation_plan(shape) -> Option<A8ActivationPlan>; select_activation_format(bf16_shape) -> ActivationFormat"""
    _create, _encode = "uzu_hip_matmul_create", "uzu_hip_matmul_encode"

    def a8_activation_plan(self, shape: "MatmulShape"):
        """-> (activation_group_size, sum_group_size | None), or None when the backend cannot run the shape with int8 activations"""
        has, plan = C.c_uint32(), A8ActivationPlan()
        call("uzu_hip_matmul_a8_activation_plan", self._h, C.byref(shape), C.byref(has), C.byref(plan))
        if not has.value:
            return None
        return plan.activation_group_size, (plan.sum_group_size if plan.has_sum_group_size else None)

    def select_activation_format(self, bf16_shape: "MatmulShape") -> int:
        fmt = C.c_uint32()
        call("uzu_hip_matmul_select_activation_format", self._h, C.byref(bf16_shape), C.byref(fmt))
        return fmt.value

    def encode(self, encoder: CommandBuffer, *, a: BufArg, b: BufArg, d: BufArg, m: int, n: int, k: int, a_offset: int = 0,
               b_kind: int = B_FULL_PRECISION, scales: BufArg = None, biases: BufArg = None, zero_points: BufArg = None,
               mode: int = QMODE_U4, group_size: int = 0, signed_codes: bool = False, b_leading_dimension: Optional[int] = None,
               b_transpose: bool = True, ab_scale: float = 1.0, accumulate: bool = False, bias: BufArg = None,
               rht_factors: BufArg = None, soft_cap: Optional[float] = None, gather_indices: BufArg = None,
               a_int8_scales: BufArg = None, a_group_sums: BufArg = None, a_group_size: int = 0):
        """`a_int8_scales` given => MatmulA::Int8Symmetric { values = a, scales, group_sums, group_size } (matmul_a.rs:9-14)."""
        args = MatmulArguments(
            _buf(a), a_offset, b_kind, _buf(b), _buf(scales), _buf(biases), _buf(zero_points), mode, group_size, int(signed_codes),
            int(b_leading_dimension is not None), b_leading_dimension or 0, int(b_transpose), _buf(d), ab_scale, int(accumulate),
            _buf(bias), _buf(rht_factors), int(soft_cap is not None), soft_cap or 0.0, _buf(gather_indices), m, n, k,
            int(a_int8_scales is not None), _buf(a_int8_scales), _buf(a_group_sums), a_group_size)
        self._enc(encoder, C.byref(args))


ATX_INPUT_RHT, ATX_OUTPUT_RHT, ATX_QUANTIZE, ATX_QUANTIZE_WITH_GROUP_SUMS = 0, 1, 2, 3


class ActivationTransformKernel(_Kernel):
    """new(ctx, T, ops, in_place, activation_scale_group_size, sum_group_size) (activation_transform.rs:43-60)"""
    _create, _encode = "uzu_hip_activation_transform_create", "uzu_hip_activation_transform_encode"

    def encode(self, input, fp_out, q_out, scales_out, group_sums_out, rht_factors, batch_size, element_count, encoder):
        self._enc(encoder, _buf(input), _buf(fp_out), _buf(q_out), _buf(scales_out), _buf(group_sums_out), _buf(rht_factors), _u(batch_size),
                  _u(element_count))


class NormalizationKernel(_Kernel):
    """new(ctx, InputT, AffineT, OutputT, AccumT, in_place, subtract_mean, full_layer, copy_to_shortcut, residual_add,
    use_hadamard, scale_residual_sum, scale_output, has_biases, has_scales)"""
    _create, _encode = "uzu_hip_normalization_create", "uzu_hip_normalization_encode"

    def encode(self, input, scales, biases, output, shortcut, hadamard_factors, batch_size, element_count, epsilon, scale_offset,
               post_layer_scalar, encoder):
        self._enc(encoder, _buf(input), _buf(scales), _buf(biases), _buf(output), _buf(shortcut), _buf(hadamard_factors),
                  _u(batch_size), _u(element_count), _f(epsilon), _f(scale_offset), _f(post_layer_scalar))


class QKVNormKernel(_Kernel):
    _create, _encode = "uzu_hip_qkv_norm_create", "uzu_hip_qkv_norm_encode"

    def encode(self, qkv_input, scales, qkv_output, batch_size, total_heads, head_dim, epsilon, scale_offset, head_offset, head_count,
               full_layer, encoder):
        self._enc(encoder, _buf(qkv_input), _buf(scales), _buf(qkv_output), _u(batch_size), _u(total_heads), _u(head_dim), _f(epsilon),
                  _f(scale_offset), _u(head_offset), _u(head_count), _u(full_layer))


class AttentionPrepareKernel(_Kernel):
    _create, _encode = "uzu_hip_attention_prepare_create", "uzu_hip_attention_prepare_encode"

    def encode(self, qkv, queries, keys, values, cosines, sines, num_q_heads, num_kv_heads, head_dim, rope_dim, kv_token_offset,
               batch_dim, encoder):
        self._enc(encoder, _buf(qkv), _buf(queries), _buf(keys), _buf(values), _buf(cosines), _buf(sines), _u(num_q_heads),
                  _u(num_kv_heads or 0), _u(head_dim), _u(rope_dim or 0), _u(kv_token_offset or 0), _u(batch_dim))


class AttentionSinglePassKernel(_Kernel):
    """new(ctx, T, HEAD_DIM, has_sinks, is_kv_cache_ring, is_causal, is_trie, is_sliding_window)"""
    _create, _encode = "uzu_hip_attention_single_pass_create", "uzu_hip_attention_single_pass_encode"

    def encode(self, queries, keys, values, out, gqa_factor, sequence_length, k_head_stride, k_seq_stride, v_head_stride, v_seq_stride,
               ring_params, scale, trie, sliding_window_size, sinks, num_heads, suffix_length, encoder):
        rp = RingParams(*(ring_params or (0, 0)))
        self._enc(encoder, _buf(queries), _buf(keys), _buf(values), _buf(out), _u(gqa_factor), _u(sequence_length), _u(k_head_stride),
                  _u(k_seq_stride), _u(v_head_stride), _u(v_seq_stride), rp, _f(scale), _buf(trie), _u(sliding_window_size or 0),
                  _buf(sinks), _u(num_heads), _u(suffix_length))


class AttentionTwoPass1Kernel(_Kernel):
    _create, _encode = "uzu_hip_attention_two_pass1_create", "uzu_hip_attention_two_pass1_encode"

    def encode(self, queries, keys, values, out, sums, maxs, gqa_factor, sequence_length, k_head_stride, k_seq_stride, v_head_stride,
               v_seq_stride, ring_params, scale, num_heads, suffix_length, trie, sliding_window_size, sinks, encoder):
        rp = RingParams(*(ring_params or (0, 0)))
        self._enc(encoder, _buf(queries), _buf(keys), _buf(values), _buf(out), _buf(sums), _buf(maxs), _u(gqa_factor),
                  _u(sequence_length), _u(k_head_stride), _u(k_seq_stride), _u(v_head_stride), _u(v_seq_stride), rp, _f(scale),
                  _u(num_heads), _u(suffix_length), _buf(trie), _u(sliding_window_size or 0), _buf(sinks))


class AttentionGemmCore:
    """AttentionGemmCore (attention_gemm/kernel.rs:8-24): is_supported / new / encode"""

    @staticmethod
    def is_supported(ctx: Context, arguments: AttentionCoreArguments) -> bool:
        out = C.c_uint32()
        call("uzu_hip_attention_gemm_is_supported", ctx._h, C.byref(arguments), C.byref(out))
        return bool(out.value)

    def __init__(self, ctx: Context, arguments: AttentionCoreArguments):
        self.ctx = ctx
        self._h = C.c_void_p()
        call("uzu_hip_attention_gemm_create", ctx._h, C.byref(arguments), C.byref(self._h))

    new = classmethod(lambda cls, ctx, arguments: cls(ctx, arguments))

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                _ffi.lib().uzu_hip_kernel_destroy(self._h)
        except Exception:
            pass

    def encode(self, queries, keys, values, out, prefix_length, suffix_length, encoder):
        call("uzu_hip_attention_gemm_encode", self._h, encoder._h, _buf(queries), _buf(keys), _buf(values), _buf(out), _u(prefix_length), _u(suffix_length))

    def encode_state(self, queries, keys, values, sinks, out, state, suffix_length, encoder):
        """AttentionCoreEncodeArguments with the state type: state = ("full", length) | ("ring", offset, length, max_length) (state.rs:16-24)"""
        if state[0] == "ring":
            _, offset, length, max_length = state
            ring = (1, length, offset, max_length)
        else:
            ring = (0, state[1], 0, 0)
        call("uzu_hip_attention_gemm_encode_state", self._h, encoder._h, _buf(queries), _buf(keys), _buf(values), _buf(sinks), _buf(out), *[_u(x) for x in ring],
             _u(suffix_length))


class AttentionTwoPass2Kernel(_Kernel):
    _create, _encode = "uzu_hip_attention_two_pass2_create", "uzu_hip_attention_two_pass2_encode"

    def encode(self, partials, sums, maxs, out, num_heads, suffix_length, encoder):
        self._enc(encoder, _buf(partials), _buf(sums), _buf(maxs), _buf(out), _u(num_heads), _u(suffix_length))


class KVCacheUpdateKernel(_Kernel):
    _create, _encode = "uzu_hip_kv_cache_update_create", "uzu_hip_kv_cache_update_encode"

    def encode(self, in_place_keys, in_place_values, copies: Sequence[Tuple[int, int]], copy_count, element_dim, encoder):
        arr = (KVCopy * max(len(copies), 1))(*[KVCopy(s, d) for s, d in copies])
        self._enc(encoder, _buf(in_place_keys), _buf(in_place_values), arr, _u(copy_count), _u(element_dim))


class SigmoidGateKernel(_Kernel):
    _create, _encode = "uzu_hip_sigmoid_gate_create", "uzu_hip_sigmoid_gate_encode"

    def encode(self, gate, output, total_elements, encoder):
        self._enc(encoder, _buf(gate), _buf(output), _u(total_elements))


class GatedActMulKernel(_Kernel):
    """new(ctx, T, ops, interleaved, use_hadamard, activation_scale_group_size, sum_group_size)"""
    _create, _encode = "uzu_hip_gated_act_mul_create", "uzu_hip_gated_act_mul_encode"

    def encode(self, act_operand, value_operand, fp_out, q_out, scales_out, group_sums_out, hadamard_factors, gated_dim, batch_dim,
               value_offset, value_row_stride, act_type, encoder):
        self._enc(encoder, _buf(act_operand), _buf(value_operand), _buf(fp_out), _buf(q_out), _buf(scales_out), _buf(group_sums_out),
                  _buf(hadamard_factors), _u(gated_dim), _u(batch_dim), _u(value_offset), _u(value_row_stride), _u(act_type))


class QuantizedEmbeddingLookupKernel(_Kernel):
    _create, _encode = "uzu_hip_quantized_embedding_lookup_create", "uzu_hip_quantized_embedding_lookup_encode"

    def encode(self, token_ids, weights, scales, zero_points, biases, output, output_hadamard_factors, batch_size, vocab_size, model_dim,
               input_scale, encoder):
        self._enc(encoder, _buf(token_ids), _buf(weights), _buf(scales), _buf(zero_points), _buf(biases), _buf(output),
                  _buf(output_hadamard_factors), _u(batch_size), _u(vocab_size), _u(model_dim), _f(input_scale))


class FullPrecisionEmbeddingLookupKernel(_Kernel):
    _create, _encode = "uzu_hip_full_precision_embedding_lookup_create", "uzu_hip_full_precision_embedding_lookup_encode"

    def encode(self, token_ids, weights, output, batch_size, vocab_size, model_dim, input_scale, encoder):
        self._enc(encoder, _buf(token_ids), _buf(weights), _buf(output), _u(batch_size), _u(vocab_size), _u(model_dim), _f(input_scale))


class LogitTransformKernel(_Kernel):
    _create, _encode = "uzu_hip_logit_transform_create", "uzu_hip_logit_transform_encode"

    def encode(self, logits, length, scale, soft_cap, encoder):
        self._enc(encoder, _buf(logits), _u(length), _f(scale), _f(soft_cap))


class TensorAddBiasKernel(_Kernel):
    _create, _encode = "uzu_hip_tensor_add_bias_create", "uzu_hip_tensor_add_bias_encode"

    def encode(self, input, bias, output, num_cols, length, encoder):
        self._enc(encoder, _buf(input), _buf(bias), _buf(output), _u(num_cols), _u(length))


class TensorAddScaleKernel(_Kernel):
    _create, _encode = "uzu_hip_tensor_add_scale_create", "uzu_hip_tensor_add_scale_encode"

    def encode(self, input, bias, output, num_cols, length, scale, encoder):
        self._enc(encoder, _buf(input), _buf(bias), _buf(output), _u(num_cols), _u(length), _f(scale))


class TensorAddSwapKernel(_Kernel):
    _create, _encode = "uzu_hip_tensor_add_swap_create", "uzu_hip_tensor_add_swap_encode"

    def encode(self, skip_buffer, main_buffer, length, encoder):
        self._enc(encoder, _buf(skip_buffer), _buf(main_buffer), _u(length))


class TensorCopyKernel(_Kernel):
    _create, _encode = "uzu_hip_tensor_copy_create", "uzu_hip_tensor_copy_encode"

    def encode(self, src_buffer, dst_buffer, length, encoder):
        self._enc(encoder, _buf(src_buffer), _buf(dst_buffer), _u(length))


class UnifiedSamplingKernel(_Kernel):
    """new(ctx, T, is_stochastic, has_bitmask, has_temperature, has_top_k, has_top_p, has_min_p)"""
    _create, _encode = "uzu_hip_unified_sampling_create", "uzu_hip_unified_sampling_encode"

    def encode(self, logits, output, seeds, bitmask, temperature, top_k, top_p, min_p, vocab_size, batch_size, encoder):
        self._enc(encoder, _buf(logits), _buf(output), _buf(seeds), _buf(bitmask), _f(temperature or 0.0), _u(top_k or 0), _f(top_p or 0.0),
                  _f(min_p or 0.0), _u(vocab_size), _u(batch_size))


class DeltaNetConvUpdateKernel(_Kernel):
    _create, _encode = "uzu_hip_delta_net_conv_update_create", "uzu_hip_delta_net_conv_update_encode"

    def encode(self, conv_weight, bias, in_out, state, kernel_size, conv_dim, state_stride, encoder):
        self._enc(encoder, _buf(conv_weight), _buf(bias), _buf(in_out), _buf(state), _u(kernel_size), _u(conv_dim), _u(state_stride))


class DeltaNetUpdateKernel(_Kernel):
    _create, _encode = "uzu_hip_delta_net_update_create", "uzu_hip_delta_net_update_encode"

    def encode(self, in_proj, a_log, dt_bias, norm_weight, state, out, num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim,
               norm_epsilon, encoder):
        self._enc(encoder, _buf(in_proj), _buf(a_log), _buf(dt_bias), _buf(norm_weight), _buf(state), _buf(out), _u(num_v_heads),
                  _u(num_k_heads), _u(head_v_dim), _u(key_dim), _u(value_dim), _f(norm_epsilon))


class Conv1dPackKernel(_Kernel):
    _create, _encode = "uzu_hip_conv1d_pack_create", "uzu_hip_conv1d_pack_encode"

    def encode(self, state_in, x, padded, state_stride, row_stride, suffix_len, num_channels, encoder):
        self._enc(encoder, _buf(state_in), _buf(x), _buf(padded), _u(state_stride), _u(row_stride), _u(suffix_len), _u(num_channels))


class DeltaNetConvScanKernel(_Kernel):
    _create, _encode = "uzu_hip_delta_net_conv_scan_create", "uzu_hip_delta_net_conv_scan_encode"

    def encode(self, conv_padded, conv_weight, bias, in_proj, state_out, suffix_len, kernel_size, row_stride, state_stride, conv_dim,
               out_stride, encoder):
        self._enc(encoder, _buf(conv_padded), _buf(conv_weight), _buf(bias), _buf(in_proj), _buf(state_out), _u(suffix_len),
                  _u(kernel_size), _u(row_stride), _u(state_stride), _u(conv_dim), _u(out_stride))


class DeltaNetPrefillPrepKernel(_Kernel):
    _create, _encode = "uzu_hip_delta_net_prefill_prep_create", "uzu_hip_delta_net_prefill_prep_encode"

    def encode(self, in_proj, a_log, dt_bias, q_norm_out, k_norm_out, compact_v_out, beta_out, decay_out, num_v_heads, num_k_heads,
               key_dim, value_dim, suffix_len, encoder):
        self._enc(encoder, _buf(in_proj), _buf(a_log), _buf(dt_bias), _buf(q_norm_out), _buf(k_norm_out), _buf(compact_v_out),
                  _buf(beta_out), _buf(decay_out), _u(num_v_heads), _u(num_k_heads), _u(key_dim), _u(value_dim), _u(suffix_len))


class DeltaNetPrefillKernel(_Kernel):
    _create, _encode = "uzu_hip_delta_net_prefill_create", "uzu_hip_delta_net_prefill_encode"

    def encode(self, q_norm, k_norm, beta, decay, in_proj, state, out, num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim,
               suffix_len, num_dv_groups, encoder):
        self._enc(encoder, _buf(q_norm), _buf(k_norm), _buf(beta), _buf(decay), _buf(in_proj), _buf(state), _buf(out), _u(num_v_heads),
                  _u(num_k_heads), _u(head_v_dim), _u(key_dim), _u(value_dim), _u(suffix_len), _u(num_dv_groups))


class DeltaNetNormGateKernel(_Kernel):
    _create, _encode = "uzu_hip_delta_net_norm_gate_create", "uzu_hip_delta_net_norm_gate_encode"

    def encode(self, in_out, in_proj, norm_weight, num_v_heads, head_v_dim, value_dim, conv_dim, total_proj_dim, norm_epsilon,
               suffix_len, encoder):
        self._enc(encoder, _buf(in_out), _buf(in_proj), _buf(norm_weight), _u(num_v_heads), _u(head_v_dim), _u(value_dim), _u(conv_dim),
                  _u(total_proj_dim), _f(norm_epsilon), _u(suffix_len))


# ---- Gated DeltaNet over a speculated token tree (cpu/kernel/gdn/tree_verify/*.rs) ----
class ConvTreeScanKernel(_Kernel):
    _create, _encode = "uzu_hip_conv_tree_scan_create", "uzu_hip_conv_tree_scan_encode"

    def encode(self, in_proj, conv_weight, bias, base_state, parents, out_proj, suffix_state, suffix_len, total_proj_dim, conv_dim, encoder):
        self._enc(encoder, _buf(in_proj), _buf(conv_weight), _buf(bias), _buf(base_state), _buf(parents), _buf(out_proj), _buf(suffix_state), _u(suffix_len),
                  _u(total_proj_dim), _u(conv_dim))


class DeltaNetTreeVerify(_Kernel):
    """DeltaNetTreeVerify::{new, encode} (backends/common/kernel/delta_net_tree_verify.rs)"""
    _create, _encode = "uzu_hip_delta_net_tree_verify_create", "uzu_hip_delta_net_tree_verify_encode"

    def encode(self, q, k, v, trie, log_decay, beta, h0, output, tree_size, encoder):
        self._enc(encoder, _buf(q), _buf(k), _buf(v), _buf(trie), _buf(log_decay), _buf(beta), _buf(h0), _buf(output), _u(tree_size))


class StateAdvanceKernel(_Kernel):
    _create, _encode = "uzu_hip_state_advance_create", "uzu_hip_state_advance_encode"

    def encode(self, k_norm, v, log_decay, beta, accepted_indices, state, accepted_len, encoder):
        self._enc(encoder, _buf(k_norm), _buf(v), _buf(log_decay), _buf(beta), _buf(accepted_indices), _buf(state), _u(accepted_len))


# ---- the tree speculators' kernels (cpu/kernel/attention/ancestor_attention.rs, cpu/kernel/weaver/*.rs) ----
class AncestorAttentionKernel(_Kernel):
    """new(context, HEAD_DIM, num_heads)"""
    _create, _encode = "uzu_hip_ancestor_attention_create", "uzu_hip_ancestor_attention_encode"

    def encode(self, prefix_kv, node_kv, current_qkv, cosines, sines, node_metadata, ancestor_indices, ancestor_counts, node_indices, output, rows, prefix_length,
               ancestor_stride, node_capacity, max_depth, scale, encoder):
        self._enc(encoder, _buf(prefix_kv), _buf(node_kv), _buf(current_qkv), _buf(cosines), _buf(sines), _buf(node_metadata), _buf(ancestor_indices), _buf(ancestor_counts),
                  _buf(node_indices), _buf(output), _u(rows), _u(prefix_length), _u(ancestor_stride), _u(node_capacity), _u(max_depth), _f(scale))


class WeaverFrontierSelectKernel(_Kernel):
    _create, _encode = "uzu_hip_weaver_frontier_select_create", "uzu_hip_weaver_frontier_select_encode"

    def encode(self, frontier, packed_tree, slot_ancestors, node_token_ids, node_metadata, node_ancestor_indices, node_valid, candidate_pool_ids, candidate_pool_logits,
               node_candidate_ids, node_candidate_logits, frontier_capacity, tree_slot_count, node_count, batch_start_slot, ancestor_stride, max_depth, lookahead_count,
               candidate_depth_count, candidates_per_depth, encoder):
        self._enc(encoder, *[_buf(b) for b in (frontier, packed_tree, slot_ancestors, node_token_ids, node_metadata, node_ancestor_indices, node_valid, candidate_pool_ids,
                                               candidate_pool_logits, node_candidate_ids, node_candidate_logits)],
                  *[_u(v) for v in (frontier_capacity, tree_slot_count, node_count, batch_start_slot, ancestor_stride, max_depth, lookahead_count, candidate_depth_count,
                                    candidates_per_depth)])


class WeaverFrontierInsertChildrenKernel(_Kernel):
    _create, _encode = "uzu_hip_weaver_frontier_insert_children_create", "uzu_hip_weaver_frontier_insert_children_encode"

    def encode(self, packed_tree, node_metadata, node_valid, child_ids, child_logprobs, frontier, frontier_capacity, tree_slot_count, node_count, expand_width, encoder):
        self._enc(encoder, *[_buf(b) for b in (packed_tree, node_metadata, node_valid, child_ids, child_logprobs, frontier)],
                  *[_u(v) for v in (frontier_capacity, tree_slot_count, node_count, expand_width)])


class WeaverTopChildrenKernel(_Kernel):
    _create, _encode = "uzu_hip_weaver_top_children_create", "uzu_hip_weaver_top_children_encode"

    def encode(self, residual_logits, candidate_logits, candidate_ids, depth_seeds, node_metadata, output_token_ids, output_model_logprobs, rows, candidates, expand_width,
               vocab_size, encoder):
        self._enc(encoder, *[_buf(b) for b in (residual_logits, candidate_logits, candidate_ids, depth_seeds, node_metadata, output_token_ids, output_model_logprobs)],
                  *[_u(v) for v in (rows, candidates, expand_width, vocab_size)])


# ---- Mixture of experts (include/uzu_hip.h "Mixture of experts"; MoeBlock::encode, encodable_block/mlp/moe/mod.rs:204-350); BF16 tensors
class MoeRouterTopKKernel(_Kernel):
    """new(context, ScalarT, has_biases, has_router_scales, has_per_expert_scales, has_router_input_scale, normalize_router_input)   (router_topk.rs:9-32)"""
    _create, _encode = "uzu_hip_moe_router_top_k_create", "uzu_hip_moe_router_top_k_encode"

    def encode(self, input, weight, bias, topk_ids, topk_probs, t, d_model, e, k, renorm, encoder):
        self._enc(encoder, _buf(input), _buf(weight), _buf(bias), _buf(topk_ids), _buf(topk_probs), _u(t), _u(d_model), _u(e), _u(k), _u(int(renorm)))


class MoeCountsOffsetsFusedKernel(_Kernel):
    _create, _encode = "uzu_hip_moe_counts_offsets_fused_create", "uzu_hip_moe_counts_offsets_fused_encode"

    def encode(self, topk_ids, offsets, sum_k_out, partials, t, e, k, encoder):
        self._enc(encoder, _buf(topk_ids), _buf(offsets), _buf(sum_k_out), _buf(partials), _u(t), _u(e), _u(k))


class MoeScatterBucketsMapKernel(_Kernel):
    """new(context, T): MoeBlockBasesFromPartials + MoeScatterBucketsMap + MoePassABuildRowMap as one launch (rows of an expert in (token, slot) order)"""
    _create, _encode = "uzu_hip_moe_scatter_buckets_map_create", "uzu_hip_moe_scatter_buckets_map_encode"

    def encode(self, topk_ids, topk_probs, offsets, out_ids, out_probs, t, e, k, tok2row, row_expert_map, encoder):
        self._enc(encoder, _buf(topk_ids), _buf(topk_probs), _buf(offsets), _buf(out_ids), _buf(out_probs), _u(t), _u(e), _u(k), _buf(tok2row), _buf(row_expert_map))


class MoeGatherXPermKernel(_Kernel):
    _create, _encode = "uzu_hip_moe_gather_x_perm_create", "uzu_hip_moe_gather_x_perm_encode"

    def encode(self, x, bucketed_ids, x_perm, sumk_buf, d_model, t, k, encoder):
        self._enc(encoder, _buf(x), _buf(bucketed_ids), _buf(x_perm), _buf(sumk_buf), _u(d_model), _u(t), _u(k))


class MoeExpertsPassAKernel(_Kernel):
    """new(context, T, gating_sel): MoeExperts{Decode,Prefill}PassA; rows are found through row_expert_map, `capacity` rows launched, *sumk_buf rows live"""
    _create, _encode = "uzu_hip_moe_experts_pass_a_create", "uzu_hip_moe_experts_pass_a_encode"

    def encode(self, x_perm, row_expert_map, sumk_buf, w13_all, up_biases, hidden_out, d_model, d_ff, gate_clip_min, gate_clip_max, up_clip_min, up_clip_max, silu_alpha,
               capacity, encoder):
        self._enc(encoder, _buf(x_perm), _buf(row_expert_map), _buf(sumk_buf), _buf(w13_all), _buf(up_biases), _buf(hidden_out), _u(d_model), _u(d_ff), _f(gate_clip_min),
                  _f(gate_clip_max), _f(up_clip_min), _f(up_clip_max), _f(silu_alpha), _u(capacity))


class MoeExpertsDownKernel(_Kernel):
    """new(context, T): MoeExpertsDecodeDownFused2D / MoeExpertsPrefillPassB"""
    _create, _encode = "uzu_hip_moe_experts_down_create", "uzu_hip_moe_experts_down_encode"

    def encode(self, hidden, row_expert_map, sumk_buf, w2_all, down_biases, y_out, d_model, d_ff, capacity, encoder):
        self._enc(encoder, _buf(hidden), _buf(row_expert_map), _buf(sumk_buf), _buf(w2_all), _buf(down_biases), _buf(y_out), _u(d_model), _u(d_ff), _u(capacity))


class MoeFinalizeKernel(_Kernel):
    _create, _encode = "uzu_hip_moe_finalize_create", "uzu_hip_moe_finalize_encode"

    def encode(self, tok2row, probs, y_partial, y, t_count, d_model, k, encoder):
        self._enc(encoder, _buf(tok2row), _buf(probs), _buf(y_partial), _buf(y), _u(t_count), _u(d_model), _u(k))
