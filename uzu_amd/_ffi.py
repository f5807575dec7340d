"""ctypes loader for ``uzu_amd/lib/libuzu_hip.so`` (the C ABI of include/uzu_hip.h + uzu_hip_engine.h).

There is NO CPU fallback anywhere in this package: if the library is missing or no AMD GPU is
visible, the calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# UZU_HIP_LIB: development switch for A/B runs against another in-tree build of the same sources (tools/ab_*.sh)
LIB_PATH = os.environ.get("UZU_HIP_LIB") or os.path.join(_HERE, "lib", "libuzu_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class UzuHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"uzu_hip status {status}: {message}")
        self.status = status
        self.message = message


STATUS_NAMES = {0: "OK", 1: "INVALID_ARGUMENT", 2: "UNSUPPORTED", 3: "HIP", 4: "OUT_OF_MEMORY", 5: "STATE"}


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


class Buf(C.Structure):
    """uzu_buf = (buffer handle, byte offset)"""
    _fields_ = [("buffer", C.c_void_p), ("offset", C.c_size_t)]


class RingParams(C.Structure):
    _fields_ = [("ring_offset", C.c_uint32), ("ring_length", C.c_uint32)]


class KVCopy(C.Structure):
    _fields_ = [("source", C.c_uint32), ("destination", C.c_uint32)]


class MatmulArguments(C.Structure):
    _fields_ = [
        ("a", Buf), ("a_offset_elements", C.c_size_t),
        ("b_kind", C.c_uint32), ("b", Buf), ("scales", Buf), ("biases", Buf), ("zero_points", Buf),
        ("mode", C.c_uint32), ("group_size", C.c_uint32), ("signed_codes", C.c_uint32),
        ("has_b_leading_dimension", C.c_uint32), ("b_leading_dimension", C.c_uint32), ("b_transpose", C.c_uint32),
        ("d", Buf),
        ("ab_scale", C.c_float), ("accumulate", C.c_uint32), ("bias", Buf), ("rht_factors", Buf),
        ("has_soft_cap", C.c_uint32), ("soft_cap", C.c_float), ("gather_indices", Buf),
        ("m", C.c_uint32), ("n", C.c_uint32), ("k", C.c_uint32),
        ("a_kind", C.c_uint32), ("a_scales", Buf), ("a_group_sums", Buf), ("a_group_size", C.c_uint32),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the uzu HIP backend has no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.uzu_hip_last_error.restype = C.c_char_p
        _lib.uzu_hip_buffer_gpu_ptr.restype = C.c_uint64
        _lib.uzu_hip_buffer_gpu_ptr.argtypes = [C.c_void_p]
        _lib.uzu_hip_buffer_size.restype = C.c_size_t
        _lib.uzu_hip_buffer_size.argtypes = [C.c_void_p]
        _lib.uzu_hip_context_stream.restype = C.c_void_p
        _lib.uzu_hip_context_stream.argtypes = [C.c_void_p]
        _lib.uzu_hip_model_context_length.restype = C.c_uint32
        _lib.uzu_hip_model_logit_count.restype = C.c_uint32
        _lib.uzu_hip_model_logit_count.argtypes = [C.c_void_p]
        _lib.uzu_hip_tp_comm_destroy.restype = None
        _lib.uzu_hip_tp_comm_destroy.argtypes = [C.c_void_p]
        _lib.uzu_hip_model_context_length.argtypes = [C.c_void_p]
        _lib.uzu_hip_model_weight_bytes.restype = C.c_size_t
        _lib.uzu_hip_model_weight_bytes.argtypes = [C.c_void_p]
        _lib.uzu_hip_model_decode_launch_count.restype = C.c_uint32
        _lib.uzu_hip_model_decode_launch_count.argtypes = [C.c_void_p]
        for name in ("uzu_hip_context_destroy", "uzu_hip_buffer_destroy", "uzu_hip_cmdbuf_destroy", "uzu_hip_kernel_destroy",
                     "uzu_hip_model_destroy"):
            getattr(_lib, name).restype = None
            getattr(_lib, name).argtypes = [C.c_void_p]
    return _lib


def check(status: int) -> None:
    if status != 0:
        msg = lib().uzu_hip_last_error().decode(errors="replace")
        raise UzuHipError(status, f"{STATUS_NAMES.get(status, '?')}: {msg}")


def call(name: str, *args) -> None:
    """Call a status-returning entry point, raising UzuHipError on failure."""
    fn = getattr(lib(), name)
    fn.restype = C.c_int32
    check(fn(*args))
