"""The mode-decision full-pel refinement search: svt_pme_sad_loop_kernel = SAD + MV-rate cost over a (sparse) search area
(Codec/product_coding_loop.c:1905-1950).  CPU: the oracle against the reference's own svt_pme_sad_loop_kernel_c (oracle/_ref), grid after
test/SadTest.cc:1580-1640.  GPU: svt_hip_pme_sad_batch and the pointer-level svt_pme_sad_loop_kernel_hip against the oracle."""
import ctypes as C

import numpy as np
import pytest

from pme_cases import MV_CENTRE, cost_tables, random_jobs, run_hip, run_oracle, run_ref
from svt_av1_psyex_amd import abi, api

W, H = 448, 320


def planes(rng, kind):
    if kind == "extremes":
        return np.zeros((H, W), np.uint8), np.full((H, W), 255, np.uint8)
    a = rng.integers(0, 256, (H, W), dtype=np.uint8)
    b = np.clip(np.roll(a, (3, -5), (0, 1)).astype(np.int32) + rng.integers(-12, 13, (H, W)), 0, 255).astype(np.uint8)
    return a, b


@pytest.mark.parametrize("cost_type", [0, 1, 2, 3, 4, 5])
def test_oracle_matches_reference(oracle, ref, cost_type):
    rng = np.random.default_rng(100 + cost_type)
    tables = cost_tables(rng)
    for kind in ("noise", "extremes"):
        src, rp = planes(rng, kind)
        for wild in (False, True):
            jobs = random_jobs(rng, W, H, 60, wild=wild)
            epb = 20542 if wild else int(rng.integers(1, 60000))
            a = run_ref(ref, src, rp, jobs, cost_type, epb, tables)
            b = run_oracle(oracle, src, rp, jobs, cost_type, epb, tables)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (kind, wild)


def test_struct_layouts(oracle):
    oracle.orc_sizeof_pme.restype = C.c_size_t
    assert oracle.orc_sizeof_pme(0) == abi.PME_JOB_DTYPE.itemsize
    assert oracle.orc_sizeof_pme(1) == C.sizeof(abi.PmeBatchDesc)
    assert oracle.orc_sizeof_pme(2) == C.sizeof(abi.MvCostParam)


def test_mv_cost_param_layout_is_the_references(oracle, ref):
    """SvtHipMvCostParam == MV_COST_PARAMS field by field: offsets, size, and the ONE-byte mv_cost_type (UENUM1BYTE); ctypes mirror too"""
    a, b = (C.c_size_t * 9)(), (C.c_size_t * 9)()
    ref.ref_mv_cost_param_layout(a)
    oracle.orc_mv_cost_param_layout(b)
    assert list(a) == list(b)
    assert a[3] >> 16 == 1
    m = abi.MvCostParam
    assert [C.sizeof(m), m.ref_mv.offset, m.full_ref_mv.offset, m.mv_cost_type.offset | (m.mv_cost_type.size << 16), m.mvjcost.offset, m.mvcost.offset,
            m.error_per_bit.offset, m.early_exit_th.offset, m.sad_per_bit.offset] == list(a)


@pytest.mark.parametrize("cost_type", [0, 1, 3, 4, 5])
def test_garbage_padding_after_mv_cost_type(oracle, ref, cost_type):
    """md_full_pel_search builds MV_COST_PARAMS on its stack field by field (product_coding_loop.c:2030-2049): the three padding bytes behind
    the one-byte mv_cost_type are garbage.  The oracle must read the type exactly as the reference does."""
    rng = np.random.default_rng(300 + cost_type)
    tables = cost_tables(rng)
    src, rp = planes(rng, "noise")
    jobs = random_jobs(rng, W, H, 40)
    a = run_ref(ref, src, rp, jobs, cost_type, 20542, tables)
    b = run_ref(ref, src, rp, jobs, cost_type, 20542, tables, garbage=0xFF)
    c = run_ref(oracle, src, rp, jobs, cost_type, 20542, tables, garbage=0x5A, fn="orc_pme_sad_loop_kernel")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


@pytest.mark.gpu
@pytest.mark.parametrize("cost_type", [0, 3, 4, 5])
def test_hip_batch_matches_oracle(hip_ctx, oracle, cost_type):
    rng = np.random.default_rng(200 + cost_type)
    tables = cost_tables(rng)
    for kind in ("noise", "extremes"):
        src, rp = planes(rng, kind)
        for wild in (False, True):
            jobs = random_jobs(rng, W, H, 400, wild=wild)
            epb = int(rng.integers(1, 60000))
            a = run_oracle(oracle, src, rp, jobs, cost_type, epb, tables)
            b = run_hip(hip_ctx, src, rp, jobs, cost_type, epb, tables)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (kind, wild)


@pytest.mark.gpu
def test_leaf_pme_sad_loop_kernel(hip_ctx, oracle):
    """svt_pme_sad_loop_kernel_hip with the reference's prototype and its MV_COST_PARAMS struct"""
    L = api.lib()
    assert L.svt_hip_leaf_bind(hip_ctx._h) == 0
    try:
        rng = np.random.default_rng(7)
        jc, tr, tc = cost_tables(rng)
        src, rp = planes(rng, "noise")
        jobs = random_jobs(rng, W, H, 24)
        want = run_oracle(oracle, src, rp, jobs, 0, 20542, (jc, tr, tc))
        for i, j in enumerate(jobs):
            rmv = abi.Mv(int(j["ref_mv"][0]), int(j["ref_mv"][1]))
            p = abi.MvCostParam()
            C.memset(C.byref(p), 0xA5, C.sizeof(p))  # the caller's stack garbage, padding bytes included
            p.ref_mv, p.mv_cost_type, p.mvjcost, p.error_per_bit = C.pointer(rmv), 0, jc.ctypes.data, 20542
            p.mvcost[0], p.mvcost[1] = tr.ctypes.data + 4 * MV_CENTRE, tc.ctypes.data + 4 * MV_CENTRE
            bc, bx, by = C.c_uint32(int(j["best_cost"])), C.c_int16(int(j["best_mvx"])), C.c_int16(int(j["best_mvy"]))
            L.svt_pme_sad_loop_kernel_hip(C.byref(p), C.c_void_p(src.ctypes.data + int(j["src_offset"])), C.c_uint32(W), C.c_void_p(rp.ctypes.data + int(j["ref_offset"])),
                                          C.c_uint32(W), C.c_uint32(int(j["height"])), C.c_uint32(int(j["width"])), C.byref(bc), C.byref(bx), C.byref(by),
                                          C.c_int16(int(j["start_x"])), C.c_int16(int(j["start_y"])), C.c_int16(int(j["sa_w"])), C.c_int16(int(j["sa_h"])),
                                          C.c_int16(int(j["step"])), C.c_int16(int(j["mvx"])), C.c_int16(int(j["mvy"])))
            assert (bc.value, bx.value, by.value) == (int(want[0][i]), int(want[1][i][0]), int(want[1][i][1])), i
    finally:
        L.svt_hip_leaf_bind(None)
