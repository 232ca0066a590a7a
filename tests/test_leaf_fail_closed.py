"""The pointer-level entries fail CLOSED (csrc/leaf_guard.h; reference behaviour: Codec/aom_dsp_rtcd.c:38-48,73-99 -- every function pointer
always holds a working kernel).  No GPU here: nothing can be bound, so every `_hip` entry must hand its call, arguments intact, to the kernel
that sat in the encoder's slot before the installer -- stand-ins made with ctypes below -- and must never abort the process."""
import ctypes as C

import numpy as np
import pytest

from svt_av1_psyex_amd import api


class Slot(C.Structure):
    _fields_ = [("name", C.c_char_p), ("slot", C.POINTER(C.c_void_p))]


NXM = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32)
VAR = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint))
FWD = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint8, C.c_uint8)
SADLOOP = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_int16), C.POINTER(C.c_int16),
                      C.c_uint32, C.c_uint8, C.c_int16, C.c_int16)
HT = C.CFUNCTYPE(C.c_uint64, C.c_void_p)


@pytest.fixture()
def installed():
    L = api.lib()
    L.svt_hip_leaf_bind(None)
    L.svt_hip_leaf_status(None, None, None, C.c_size_t(0))
    calls = []

    def nxm(src, ss, ref, rs, h, w):
        calls.append(("nxm", ss, rs, h, w))
        return 4242

    def var(src, ss, ref, rs, sse):
        calls.append(("var", ss, rs))
        sse[0] = 777
        return 55

    def fwd(inp, out, stride, tx_type, bd):
        calls.append(("fwd", stride, tx_type, bd))
        C.cast(out, C.POINTER(C.c_int32))[0] = -9

    def sadloop(src, ss, ref, rs, bh, bw, best, x, y, raw, skip, saw, sah):
        calls.append(("sadloop", bh, bw, saw, sah))
        best[0], x[0], y[0] = 31, 2, -3

    def ht(out):
        calls.append(("ht",))
        return 1 << 40

    keep = [NXM(nxm), VAR(var), FWD(fwd), SADLOOP(sadloop), HT(ht)]
    names = [b"svt_nxm_sad_kernel", b"svt_aom_variance16x16", b"svt_av1_fwd_txfm2d_8x8", b"svt_sad_loop_kernel", b"svt_handle_transform64x64", b"svt_av1_optimize_b"]
    prev = [C.cast(k, C.c_void_p).value for k in keep] + [0x1234]
    vals = [C.c_void_p(p) for p in prev]
    slots = (Slot * len(names))(*[Slot(n, C.pointer(v)) for n, v in zip(names, vals)])
    yield L, slots, vals, prev, calls, keep
    L.svt_hip_uninstall_rtcd(slots, len(names))
    L.svt_hip_leaf_inject_failure(0)


def test_installer_needs_a_context_and_touches_nothing_without_one(installed):
    L, slots, vals, prev, calls, _ = installed
    skipped = C.c_uint32(99)
    assert L.svt_hip_install_rtcd(None, slots, len(prev), C.byref(skipped)) == 2  # SVT_HIP_ERR_BAD_PARAM: the encoder keeps its dispatch
    assert [v.value for v in vals] == prev


def test_entries_without_a_device_call_the_previous_kernels(installed):
    L, slots, vals, prev, calls, _ = installed
    skipped = C.c_uint32(0)
    assert L.svt_hip_rtcd_store(slots, len(prev), C.byref(skipped)) == 0
    assert skipped.value == 1 and vals[-1].value == 0x1234  # a name this library has no entry for: left alone
    for i, name in enumerate(["svt_nxm_sad_kernel_helper_hip", "svt_aom_variance16x16_hip", "svt_av1_fwd_txfm2d_8x8_hip", "svt_sad_loop_kernel_hip", "svt_handle_transform64x64_hip"]):
        assert vals[i].value == C.cast(getattr(L, name), C.c_void_p).value, name  # the slot holds this library's entry now
    a = np.arange(64, dtype=np.uint8)
    # through the slots, as the encoder would call them: nothing is bound, every call lands in the stand-ins with its arguments
    assert NXM(vals[0].value)(a.ctypes.data, 8, a.ctypes.data, 8, 8, 8) == 4242
    sse = C.c_uint(0)
    assert VAR(vals[1].value)(a.ctypes.data, 16, a.ctypes.data, 16, C.byref(sse)) == 55 and sse.value == 777  # nested: the per-size entry -> the generic one
    out = np.zeros(64, np.int32)
    res = np.zeros(64, np.int16)
    FWD(vals[2].value)(res.ctypes.data, out.ctypes.data, 8, 3, 10)
    assert out[0] == -9
    best, x, y = C.c_uint64(0), C.c_int16(0), C.c_int16(0)
    SADLOOP(vals[3].value)(a.ctypes.data, 8, a.ctypes.data, 8, 8, 8, C.byref(best), C.byref(x), C.byref(y), 8, 0, 8, 3)
    assert (best.value, x.value, y.value) == (31, 2, -3)
    assert HT(vals[4].value)(out.ctypes.data) == 1 << 40
    assert calls == [("nxm", 8, 8, 8, 8), ("var", 16, 16), ("fwd", 8, 3, 10), ("sadloop", 8, 8, 8, 3), ("ht",)]
    fb, un = C.c_ulonglong(0), C.c_ulonglong(0)
    msg = C.create_string_buffer(512)
    assert L.svt_hip_leaf_status(C.byref(fb), C.byref(un), msg, C.c_size_t(512)) == 5
    assert (fb.value, un.value) == (5, 0) and b"svt_handle_transform64x64_hip" in msg.value and b"no context bound" in msg.value
    assert b"svt_handle_transform64x64_hip" in L.svt_hip_last_error(None)  # the calling thread's error text as well
    # uninstall puts the encoder's kernels back
    assert L.svt_hip_uninstall_rtcd(slots, len(prev)) == 0
    assert [v.value for v in vals] == prev


def test_direct_call_without_previous_kernel_reports_and_returns(installed):
    L, *_ = installed
    a = np.arange(64, dtype=np.uint8)
    L.svt_aom_sad8x8_hip.restype = C.c_uint32
    assert L.svt_aom_sad8x8_hip(C.c_void_p(a.ctypes.data), 8, C.c_void_p(a.ctypes.data), 8) == 0  # nothing computed, no abort
    fb, un = C.c_ulonglong(0), C.c_ulonglong(0)
    L.svt_hip_leaf_status(C.byref(fb), C.byref(un), None, C.c_size_t(0))
    assert (fb.value, un.value) == (0, 1)
