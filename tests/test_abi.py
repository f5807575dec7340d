"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol the
headers declare, and fails loudly (never falls back) without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for header in ("uzu_hip.h", "uzu_hip_engine.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(uzu_hip_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_builds_and_exports_every_declared_symbol():
    from uzu_amd import _ffi
    _ffi.build()
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) > 90
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_every_kernel_has_create_and_encode():
    syms = declared_symbols()
    creates = {s[: -len("_create")] for s in syms if s.endswith("_create")} - {"uzu_hip_context", "uzu_hip_buffer", "uzu_hip_cmdbuf", "uzu_hip_model", "uzu_hip_tp_comm", "uzu_hip_state", "uzu_hip_sparse_buffer", "uzu_hip_drafter", "uzu_hip_weaver"}
    encodes = {s[: -len("_encode")] for s in syms if s.endswith("_encode")}
    assert creates == encodes and len(creates) >= 25


def test_no_gpu_means_loud_failure_not_fallback():
    """On a box without an AMD GPU Context::new must raise; with a GPU this test is vacuous."""
    from uzu_amd import _ffi, backend
    lib = _ffi.lib()
    n = ctypes.c_int(0)
    hip = ctypes.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(backend.UzuHipError) as e:
        backend.Context.new(0)
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/ (task rule)."""
    pkg = os.path.join(ROOT, "uzu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower() or f in (), f"{os.path.join(dirpath, f)} mentions the oracle"


def test_desc_struct_layout_matches_c():
    """ctypes mirrors of include/uzu_model_desc.h have the C sizes (compiled probe)."""
    import subprocess
    import tempfile
    from uzu_amd import desc as D
    src = '#include <stdio.h>\n#include "uzu_model_desc.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(uzu_linear_desc), sizeof(uzu_norm_desc), sizeof(uzu_rope_desc), sizeof(uzu_layer_desc), sizeof(uzu_model_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "p.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "p")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(D.LinearDesc), ctypes.sizeof(D.NormDesc), ctypes.sizeof(D.RopeDesc), ctypes.sizeof(D.LayerDesc), ctypes.sizeof(D.ModelDesc)]


def _build_abi_smoke(tmp_path):
    """tests/host/abi_smoke.c: plain C99 against include/uzu_hip.h + libuzu_hip.so -- the header as an FFI generator sees it."""
    import subprocess
    from uzu_amd import _ffi
    _ffi.build()
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(_ffi.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "abi_smoke.c"),
                    "-L", libdir, "-luzu_hip", f"-Wl,-rpath,{libdir}", "-lm", "-o", exe], check=True)
    return exe


def test_c_program_compiles_against_the_header_and_fails_loudly_without_a_gpu(tmp_path):
    """No Python between the caller and the C ABI: the header is valid C99 (-Wall -Werror), the program links, and on a box
    without a GPU Context::new is refused with a status AND a message (with a GPU this leg is vacuous)."""
    import subprocess
    exe = _build_abi_smoke(tmp_path)
    out = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "context_create refused" in out.stdout or "a GPU is present" in out.stdout


@pytest.mark.gpu
def test_c_program_runs_matmul_and_normalization_bit_exact(tmp_path):
    """create -> encode -> end_encoding -> submit -> wait -> download from C for MatmulKernel (int4 ScaleBias) and Normalization
    (residual protocol, graph-captured command buffer) against constants with exactly representable expectations."""
    import subprocess
    exe = _build_abi_smoke(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "abi_smoke ok" in out.stdout, out.stdout + out.stderr
