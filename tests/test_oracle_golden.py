"""CPU: the oracle (oracle/*.c) against fixtures produced by the reference's own compiled C code
(tests/golden/, generator oracle/gen_golden.py).  This is what pins the oracle on machines without /root/reference."""
import ctypes as C
import os

import numpy as np
import pytest

from golden_io import GOLDEN, GoldenMeCase, me_fixture_names
from me_cases import compare


@pytest.mark.parametrize("name", me_fixture_names())
def test_me_picture_matches_reference_fixture(name):
    case = GoldenMeCase(name)
    got = case.run_cpu("oracle")
    assert not compare(case.expected, got), name


def test_sad_loop_kernel_known_answers(oracle):
    z = np.load(os.path.join(GOLDEN, "sad_loop_kat.npz"))
    so = ro = 0
    for (bw, bh, sw, sh, skip, best, x, y), stride, rows in zip(z["meta"], z["ref_stride"], z["ref_rows"]):
        src = np.ascontiguousarray(z["src"][so:so + bw * bh]); so += bw * bh
        refp = np.ascontiguousarray(z["ref"][ro:ro + stride * rows]); ro += stride * rows
        b, xs, ys = C.c_uint64(0), C.c_int16(-7), C.c_int16(-7)
        oracle.orc_sad_loop_kernel(src.ctypes.data_as(C.c_void_p), C.c_uint32(int(bw)), refp.ctypes.data_as(C.c_void_p), C.c_uint32(int(stride)),
                                   C.c_uint32(int(bh)), C.c_uint32(int(bw)), C.byref(b), C.byref(xs), C.byref(ys), C.c_uint32(int(stride)),
                                   C.c_uint8(int(skip)), C.c_int16(int(sw)), C.c_int16(int(sh)))
        assert (b.value, xs.value, ys.value) == (best, x, y), (bw, bh, sw, sh, skip)
    assert so == len(z["src"]) and ro == len(z["ref"])


def test_search_centre_outside_the_padded_plane_is_defined(oracle):
    """The one place where the oracle deliberately leaves the reference (whose read is out of bounds there, see me_cases.probe_outside_case):
    the result must not depend on what lies behind the plane in memory."""
    from me_cases import probe_outside_case
    import numpy as np
    a = probe_outside_case().run_cpu("oracle")
    junk = [np.full(1 << 20, v, np.uint8) for v in (0, 255)]  # disturb the heap between the two runs
    b = probe_outside_case().run_cpu("oracle")
    assert junk and all(np.array_equal(a[k], b[k]) for k in a)
