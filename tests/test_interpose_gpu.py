"""GPU: the drop-in boundary as an installed surface (BASELINE config 1's plumbing): svt_hip_install_rtcd stores this
library's `_hip` entries into the REFERENCE's own rtcd function pointers (oracle/_ref/libsvtref.so: the reference's sources compiled by
oracle/Makefile; the pointers svt_aom_setup_rtcd_internal fills, Codec/aom_dsp_rtcd.c:188), then the reference's own
svt_aom_motion_estimation_b64 (Codec/motion_estimation.c:3076) runs the committed CIF fixtures through them: the whole SAD family of
the open-loop search executes on the GPU behind the reference's unchanged driver, and MeSbResults must come out as the fixture holds them."""
import ctypes as C

import numpy as np
import pytest

import pyoracle
from golden_io import GoldenMeCase
from me_cases import compare
from svt_av1_psyex_amd import abi, api

pytestmark = pytest.mark.gpu

SAD_POINTERS = ["svt_sad_loop_kernel", "svt_nxm_sad_kernel", "svt_ext_all_sad_calculation_8x8_16x16", "svt_ext_eight_sad_calculation_32x32_64x64",
                "svt_ext_sad_calculation_8x8_16x16", "svt_ext_sad_calculation_32x32_64x64", "svt_initialize_buffer_32bits"]


class RtcdSlot(C.Structure):
    _fields_ = [("name", C.c_char_p), ("slot", C.c_void_p)]


@pytest.mark.parametrize("name", ["me_cif_m12_b", "me_cif_m6_p_base"])
def test_reference_driver_runs_on_installed_hip_leaves(hip_ctx, ref, name):
    L = api.lib()
    ref.ref_rtcd_slot.restype = C.c_void_p
    L.svt_hip_rtcd_lookup.restype = C.c_void_p
    slots = (RtcdSlot * (len(SAD_POINTERS) + 1))()
    for s, n in zip(slots, SAD_POINTERS):
        s.name, s.slot = n.encode(), ref.ref_rtcd_slot(n.encode())
        assert s.slot, n
    dummy = C.c_void_p(0x1234)  # a pointer this backend has no entry for: must be left alone and counted
    slots[len(SAD_POINTERS)].name, slots[len(SAD_POINTERS)].slot = b"svt_av1_build_compound_diffwtd_mask", C.addressof(dummy)
    skipped = C.c_uint32(99)
    case = GoldenMeCase(name)
    try:
        assert L.svt_hip_install_rtcd(hip_ctx._h, slots, len(slots), C.byref(skipped)) == 0
        assert skipped.value == 1 and dummy.value == 0x1234
        for s, n in zip(slots, SAD_POINTERS):  # the reference's pointer variables now hold this library's entries
            assert C.c_void_p.from_address(s.slot).value == L.svt_hip_rtcd_lookup(n.encode()), n
        ref.ref_set_simd(2)  # the harness keeps its hands off the pointers
        got = pyoracle.me_picture("ref", case.cfg, case.desc, case.cur, case.refs)
    finally:
        ref.ref_set_simd(0)  # back to the reference's `_c` kernels for every other test
        L.svt_hip_leaf_bind(None)
    assert not compare(case.expected, got, names=list(case.expected)), name


def test_device_failure_falls_back_to_the_encoders_kernels(hip_ctx, ref):
    """Fail closed on a live installation (csrc/leaf_guard.h): with the reference's `_c` kernels in the slots first, the installed `_hip`
    entries run on the GPU; when the device fails (svt_hip_leaf_inject_failure) the same calls are served by the reference's own kernels --
    the driver still produces the fixture's results, nothing aborts -- and the library says how many calls took that way."""
    L = api.lib()
    ref.ref_rtcd_slot.restype = C.c_void_p
    slots = (RtcdSlot * len(SAD_POINTERS))()
    ref.ref_set_simd(0)  # the reference's `_c` kernels are what the installer finds in the slots
    for s, n in zip(slots, SAD_POINTERS):
        s.name, s.slot = n.encode(), ref.ref_rtcd_slot(n.encode())
    before = [C.c_void_p.from_address(s.slot).value for s in slots]
    case = GoldenMeCase("me_cif_m12_b")
    fb, un = C.c_ulonglong(0), C.c_ulonglong(0)
    try:
        assert L.svt_hip_install_rtcd(hip_ctx._h, slots, len(slots), None) == 0
        ref.ref_set_simd(2)
        L.svt_hip_leaf_status(None, None, None, C.c_size_t(0))
        got_gpu = pyoracle.me_picture("ref", case.cfg, case.desc, case.cur, case.refs)
        assert L.svt_hip_leaf_status(C.byref(fb), C.byref(un), None, C.c_size_t(0)) == 0  # every call ran on the device
        L.svt_hip_leaf_inject_failure(1)
        got_fallback = pyoracle.me_picture("ref", case.cfg, case.desc, case.cur, case.refs)
        L.svt_hip_leaf_inject_failure(0)
        assert L.svt_hip_leaf_status(C.byref(fb), C.byref(un), None, C.c_size_t(0)) > 0 and fb.value > 0 and un.value == 0
        got_again = pyoracle.me_picture("ref", case.cfg, case.desc, case.cur, case.refs)
        assert L.svt_hip_leaf_status(C.byref(fb), C.byref(un), None, C.c_size_t(0)) == 0
    finally:
        L.svt_hip_leaf_inject_failure(0)
        L.svt_hip_uninstall_rtcd(slots, len(slots))
        ref.ref_set_simd(0)
        L.svt_hip_leaf_bind(None)
    assert [C.c_void_p.from_address(s.slot).value for s in slots] == before  # uninstall put the reference's kernels back
    for got in (got_gpu, got_fallback, got_again):
        assert not compare(case.expected, got, names=list(case.expected))
