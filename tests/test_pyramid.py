"""The luma pyramid pinned on the reference: tests/golden/pyramid.npz holds the 1/4 and 1/16 planes, padding
included, written by the reference's svt_aom_downsample_filtering_input_picture (pic_analysis_process.c:2139-2196:
svt_aom_downsample_2d_c :130-158 + svt_aom_generate_padding pic_operators.c:397-443).  Checked against it:
  * svt_av1_psyex_amd.synth.HostPyramid -- the numpy pyramid every ME fixture and test picture is built with (CPU test),
  * svt_hip_pa_picture_create's device-built planes (GPU test),
  * and, where oracle/_ref exists, the reference itself on more sizes (the fixture is not stale)."""
import os

import numpy as np
import pytest

from pyramid_cases import CASES, PAD_Q, PAD_S, luma
from svt_av1_psyex_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pyramid.npz")


def host_planes(img):
    p = synth.HostPyramid(img)
    out = []
    for level, pad in ((1, PAD_Q), (0, PAD_S)):
        buf, stride, pp, w, h = p.planes[level]
        assert pp == pad
        out.append(buf[:, :w + 2 * pad])
    return out


@pytest.mark.parametrize("name", list(CASES))
def test_host_pyramid_matches_reference_fixture(name):
    z = np.load(GOLDEN)
    q, s = host_planes(luma(*CASES[name]))
    assert np.array_equal(q, z[name + "_q"]) and np.array_equal(s, z[name + "_s"])


@pytest.mark.parametrize("size", [(64, 64), (8, 8), (136, 72), (640, 360), (72, 80), (1280, 720)])
def test_host_pyramid_matches_reference_build(ref, size):
    from pyramid_cases import ref_pyramid
    img = luma(size[0], size[1], "noise", size[0] + size[1])
    q, s = host_planes(img)
    rq, rs = ref_pyramid(img)
    assert np.array_equal(q, rq) and np.array_equal(s, rs)
    assert not (rq == 0xA5).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_device_pyramid_matches_reference_fixture(hip_ctx, name):
    z = np.load(GOLDEN)
    pic = hip_ctx.upload(synth.HostPyramid(luma(*CASES[name])), device_pyramid=True)
    try:
        assert np.array_equal(pic.download(1), z[name + "_q"]), "quarter"
        assert np.array_equal(pic.download(0), z[name + "_s"]), "sixteenth"
    finally:
        pic.free()
