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


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["svt_hip_pa_picture_update", "svt_hip_pa_picture_update_ahead"])
def test_picture_refill_rebuilds_the_pyramid(hip_ctx, entry):
    """A pooled picture object refilled with another picture (on the context stream, or ahead on the transfer stream): every level equals the
    fixture of the new content, and a download / ME launch issued right behind the refill sees it (the picture's event orders them)."""
    import ctypes as C

    from svt_av1_psyex_amd import api
    z = np.load(GOLDEN)
    names = list(CASES)[:1]
    first = synth.HostPyramid(luma(*CASES[names[0]]))
    pic = hip_ctx.upload(first, device_pyramid=True)
    try:
        other = synth.HostPyramid(np.ascontiguousarray(luma(*CASES[names[0]])[::-1, ::-1]))  # same geometry, other content
        full = other.desc(2)
        hip_ctx.check(getattr(api.lib(), entry)(hip_ctx._h, pic._h, C.byref(full), 0), entry)
        want = hip_ctx.upload(other, device_pyramid=True)
        try:
            for level in (0, 1, 2):  # the download waits for the picture's event
                assert np.array_equal(pic.download(level), want.download(level)), level
        finally:
            want.free()
        back = first.desc(2)
        hip_ctx.check(getattr(api.lib(), entry)(hip_ctx._h, pic._h, C.byref(back), 0), entry)
        assert np.array_equal(pic.download(1), z[names[0] + "_q"]) and np.array_equal(pic.download(0), z[names[0] + "_s"])
    finally:
        pic.free()
