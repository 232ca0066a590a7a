"""GPU parity: svt_hip_dg_detector_hme_level0 (through the C-ABI) against the oracle and the committed reference fixture."""
import os

import numpy as np
import pytest

import pyoracle
from dg_cases import GRID, METRICS, DgCase

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dg_detector.npz")


def run_hip(ctx, c, device_pyramid=True):
    src, ref = ctx.upload(c.src, device_pyramid), ctx.upload(c.ref, device_pyramid)
    try:
        return ctx.dg_detector_hme_level0(src, ref, *c.args())
    finally:
        src.free()
        ref.free()


@pytest.mark.parametrize("w,h,kind", GRID)
def test_dg_detector_vs_oracle(hip_ctx, w, h, kind):
    c = DgCase(w, h, kind)
    want = pyoracle.dg_detector("oracle", c.src, c.ref, *c.args())
    got = run_hip(hip_ctx, c)
    np.testing.assert_array_equal(got["b64_sad"], want["b64_sad"])
    np.testing.assert_array_equal(got["b64_mv"], want["b64_mv"])
    assert {k: got[k] for k in METRICS} == {k: want[k] for k in METRICS}


def test_dg_detector_vs_reference_fixture(hip_ctx):
    z = np.load(GOLDEN)
    for i, (w, h, kind) in enumerate(zip(z["width"], z["height"], z["kind"])):
        c = DgCase(int(w), int(h), str(kind))
        got = run_hip(hip_ctx, c, device_pyramid=(i % 2 == 0))
        assert [got[k] for k in METRICS] == [int(v) for v in z["metrics"][i]], c


def test_dg_detector_2160p_properties(hip_ctx):
    """Full size (BASELINE's 3840x2160): sums agree with the per-block results, a picture against itself is all zero."""
    c = DgCase(3840, 2160, "pan", distance=8, seed=11)
    got = run_hip(hip_ctx, c)
    assert got["b64_sad"].shape == (60 * 34,)
    assert int(got["b64_sad"].astype(np.uint64).sum()) == got["tot_dist"]
    assert int((got["b64_mv"] != 0).any(axis=1).sum()) == got["tot_active"]
    assert int((got["b64_sad"] > 16 * 16 * 30).sum()) == got["tot_cplx"]
    src = hip_ctx.upload(c.src)
    same = hip_ctx.dg_detector_hme_level0(src, src, *c.args())
    src.free()
    # a zero SAD at the zero vector can still lose to an EARLIER zero-SAD position (flat areas); the distortion cannot
    assert same["tot_dist"] == 0 and same["tot_cplx"] == 0


def test_dg_detector_rejects_bad_arguments(hip_ctx):
    from svt_av1_psyex_amd import api
    c = DgCase(352, 288, "pan")
    src = hip_ctx.upload(c.src)
    with pytest.raises(api.SvtHipError):
        hip_ctx.dg_detector_hme_level0(src, src, 3840, 2160, 5)  # planes of a CIF picture cannot hold a 2160p grid
    src.free()
