"""uzu_amd/csrc/uzu_math.h (the expf / logf the kernels use) reproduces the system libm BIT FOR BIT on the host.
The same header compiles for gfx950, where every operation is an IEEE double op => identical bits on the GPU
(checked on the GPU by the bit-exact element-wise kernel tests)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_expf_logf_match_glibc_bitwise():
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "math_check")
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "host", "math_check.cpp"), "-lm"], check=True)
        out = subprocess.check_output([exe, "1009"]).decode()
    lines = dict(l.split(":") for l in out.strip().splitlines() if ":" in l)
    assert lines["expf"].strip().endswith(" 0 mismatches") and lines["logf"].strip().endswith(" 0 mismatches"), out
    assert int(lines["expf"].split()[0]) > 3_000_000


def test_gemm_tile_map_visits_every_tile_once():
    """uzu_amd/csrc/gemm_tile_map.h (workgroup id -> output tile of the large-tile prefill GEMM, XCD super-tiles):
    exhaustive host check over 48 x 160 tile counts."""
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "tile_map_check")
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "tile_map_check.cpp")], check=True)
        out = subprocess.check_output([exe]).decode()
    assert out.strip().endswith(" 0 bad"), out
