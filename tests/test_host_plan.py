"""CPU-side checks of the decode GEMV launch plan (csrc/k_decode.hip: gemv_dec_plan through uzu_hip_decode_gemv_plan -- host
arithmetic, no GPU).  Every decision pinned here is one DESIGN.md section 3 quotes a same-box measurement for; a change of the rule
has to come with a new measurement."""
import ctypes as C

import pytest

from uzu_amd import _ffi

CUS = 256  # MI355X


class Plan(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("lanes_per_row", "rows_per_lane_group", "steps_per_lane", "waves_per_workgroup", "batches",
                                          "workgroup_batches", "workgroups")]


def plan(n0, k, bits=4, normed=False, gated=False, n1=0, cus=CUS):
    lib = _ffi.lib()
    lib.uzu_hip_decode_gemv_plan.argtypes = [C.c_uint32] * 7 + [C.POINTER(Plan)]
    lib.uzu_hip_decode_gemv_plan.restype = C.c_int32
    p = Plan()
    status = lib.uzu_hip_decode_gemv_plan(n0, n1, k, bits, int(normed), int(gated), cus, C.byref(p))
    assert status == 0, _ffi.last_error() if hasattr(_ffi, "last_error") else status
    return p


@pytest.mark.parametrize("n,k,normed,gated", [(8224, 1024, True, False), (7168, 1024, True, True), (1024, 3584, False, False), (1024, 2048, False, False),
                                              (248320, 1024, True, False)])
def test_latency_regime_and_k1024_readout_stay_on_four_wave_workgroups(n, k, normed, gated):
    """Qwen3.5-0.8B: a few MB per matrix, and the 135 MB read-out at K = 1024 (wide workgroups measured 5 % slower there)."""
    p = plan(n, k, normed=normed, gated=gated)
    assert p.waves_per_workgroup == 4 and p.workgroup_batches == 0


def test_llama3_8b_int4_plans():
    qkv = plan(6144, 4096, normed=True)
    assert (qkv.waves_per_workgroup, qkv.rows_per_lane_group) == (16, 2)
    # 3072 batches are 192 full workgroups: spread over every CU instead (8.2 -> 7.8 us)
    assert (qkv.batches, qkv.workgroup_batches, qkv.workgroups) == (3072, 12, 256)
    up = plan(28672, 4096, normed=True, gated=True)
    assert (up.waves_per_workgroup, up.rows_per_lane_group, up.workgroup_batches) == (16, 1, 0)  # fused up / gate: never two row pairs
    down = plan(4096, 14336)
    assert (down.waves_per_workgroup, down.rows_per_lane_group, down.steps_per_lane) == (16, 1, 7)
    assert down.batches == CUS * 16 and down.workgroup_batches == 0  # exactly one round: two rows per lane group would idle half the waves
    out = plan(4096, 4096)
    assert out.waves_per_workgroup == 4  # 8 MB: latency regime
    readout = plan(128256, 4096, normed=True)
    assert readout.waves_per_workgroup == 16


def test_llama3_8b_int8_rows_take_two_rows_per_lane_group():
    down, qkv, up = plan(4096, 14336, bits=8), plan(6144, 4096, bits=8, normed=True), plan(28672, 4096, bits=8, normed=True, gated=True)
    assert all(p.waves_per_workgroup == 4 for p in (down, qkv, up))  # int8 stays on 4-wave workgroups
    assert (down.rows_per_lane_group, qkv.rows_per_lane_group, up.rows_per_lane_group) == (2, 2, 1)


def test_qwen3_14b_class_plans():
    down = plan(5120, 17408)
    # 20 rows per 16-wave workgroup = 16 + 4 with one row per lane group; two rows: 2560 batches, spread as 256 x 10
    assert (down.waves_per_workgroup, down.rows_per_lane_group) == (16, 2)
    assert (down.batches, down.workgroup_batches, down.workgroups) == (2560, 10, 256)
    readout = plan(151936, 5120, normed=True)
    assert (readout.waves_per_workgroup, readout.rows_per_lane_group) == (12, 2)  # 594 rows per workgroup: 50 rounds either way
    qkv = plan(7168, 5120, normed=True)
    assert (qkv.waves_per_workgroup, qkv.rows_per_lane_group) == (12, 1)  # 28 rows per 12 waves: 3 rounds against 2 x 2
    up = plan(34816, 5120, normed=True, gated=True)
    assert (up.waves_per_workgroup, up.rows_per_lane_group) == (12, 1)
    out = plan(5120, 5120)
    assert out.waves_per_workgroup == 4  # 13 MB without a prologue to share: not worth a wide workgroup (7.5 -> 8.2 us)


def test_plan_scales_with_the_device():
    """A smaller device has fewer resident waves: the same matrix is more than one round there and is not spread."""
    small = plan(6144, 4096, normed=True, cus=64)
    assert small.workgroup_batches == 0
    assert plan(6144, 4096, normed=True, cus=304).workgroup_batches == 11


def test_bad_arguments_are_rejected():
    lib = _ffi.lib()
    lib.uzu_hip_decode_gemv_plan.argtypes = [C.c_uint32] * 7 + [C.POINTER(Plan)]
    lib.uzu_hip_decode_gemv_plan.restype = C.c_int32
    p = Plan()
    assert lib.uzu_hip_decode_gemv_plan(0, 0, 4096, 4, 0, 0, CUS, C.byref(p)) != 0
    assert lib.uzu_hip_decode_gemv_plan(4096, 0, 4010, 4, 0, 0, CUS, C.byref(p)) != 0
    assert lib.uzu_hip_decode_gemv_plan(4096, 0, 4096, 5, 0, 0, CUS, C.byref(p)) != 0
    assert lib.uzu_hip_decode_gemv_plan(4097, 0, 4096, 4, 0, 1, CUS, C.byref(p)) != 0
    assert lib.uzu_hip_decode_gemv_plan(4096, 0, 4096, 4, 0, 0, CUS, None) != 0


# ---------------------------------------------------------------------------------------------- prefill GEMM plan
class GemmPlan(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("large_tile", "form", "splits", "workgroups")]


def gemm_plan(m, n, k, bits=4, group=128, gated=False, cus=CUS):
    lib = _ffi.lib()
    lib.uzu_hip_prefill_gemm_plan.argtypes = [C.c_uint32] * 7 + [C.POINTER(GemmPlan)]
    lib.uzu_hip_prefill_gemm_plan.restype = C.c_int32
    p = GemmPlan()
    assert lib.uzu_hip_prefill_gemm_plan(m, n, k, bits, group, int(gated), cus, C.byref(p)) == 0
    return p


def test_prefill_gemm_plan_follows_the_measured_ab(monkeypatch):
    """csrc/k_gemm128.hip::gemm128_form through uzu_hip_prefill_gemm_plan (host arithmetic).  The three forms are bit-identical
    (tests/test_gpu_kernels.py); which one runs is the same-box A/B of profiles/r5_gemm_pp_ab.txt: ping-pong on long reductions with at
    least one 128 x 256 tile per CU and a plain epilogue, wave-specialised on the few-tile split-K shapes, the 256-thread form elsewhere."""
    monkeypatch.delenv("UZU_HIP_TUNE", raising=False)
    # Llama-3-8B at 4096 rows: qkv / out / down on the ping-pong form (x1.05-1.16), the gated up projection on the 256-thread form (x0.95)
    for n, k in ((6144, 4096), (4096, 4096), (4096, 14336)):
        p = gemm_plan(4096, n, k)
        assert (p.large_tile, p.form, p.splits) == (1, 1, 1), (n, k)
    assert gemm_plan(4096, 28672, 4096, gated=True).form == 0
    # a 1024-row pass of the same model leaves CUs without a tile pair: 256-thread form
    assert gemm_plan(1024, 4096, 4096).form == 0
    # Qwen3.5-0.8B at 2048 rows: short reductions with many tiles stay on the 256-thread form, the N = 1024 projections split K and take
    # the wave-specialised form (x1.02-1.05)
    assert gemm_plan(2048, 8224, 1024).form == 0 and gemm_plan(2048, 8224, 1024).workgroups == 16 * 65
    assert gemm_plan(2048, 7168, 1024, gated=True).form == 0
    for n, k in ((1024, 3584), (1024, 2048)):
        p = gemm_plan(2048, n, k)
        assert p.form == 2 and p.splits > 1, (n, k)
    # below 128 rows the large-tile kernel is not used at all
    assert gemm_plan(64, 4096, 4096).large_tile == 0
    # the switch forces a form
    monkeypatch.setenv("UZU_HIP_TUNE", "gemm_form=0")
    assert gemm_plan(4096, 4096, 4096).form == 0
    monkeypatch.setenv("UZU_HIP_TUNE", "rows_norm=1,gemm_form=2")  # (a key anywhere in the list)
    assert gemm_plan(4096, 4096, 4096).form == 2


def test_prefill_gemm_plan_bad_arguments():
    lib = _ffi.lib()
    lib.uzu_hip_prefill_gemm_plan.argtypes = [C.c_uint32] * 7 + [C.POINTER(GemmPlan)]
    lib.uzu_hip_prefill_gemm_plan.restype = C.c_int32
    p = GemmPlan()
    assert lib.uzu_hip_prefill_gemm_plan(0, 4096, 4096, 4, 128, 0, CUS, C.byref(p)) != 0
    assert lib.uzu_hip_prefill_gemm_plan(4096, 4096, 4096, 5, 128, 0, CUS, C.byref(p)) != 0
    assert lib.uzu_hip_prefill_gemm_plan(4096, 4096, 4000, 4, 128, 0, CUS, C.byref(p)) != 0
    assert lib.uzu_hip_prefill_gemm_plan(4096, 4097, 4096, 4, 128, 1, CUS, C.byref(p)) != 0
