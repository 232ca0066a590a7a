"""GPU: the batched block statistics (svt_hip_block_stats_batch) and the pointer-level `_hip` leaf entries
(include/svt_hip_leaf.h), through the C-ABI, against the reference fixtures and the oracle."""
import ctypes as C
import os

import numpy as np
import pyoracle
import pytest

from stats_cases import load_var10_fixture, FACADE_SETTINGS, PSY_RD, facade_arg, load_facade_fixture, load_fixture, load_subpel_fixture, mismatches
from svt_av1_psyex_amd import abi, api, stats

pytestmark = pytest.mark.gpu
P = C.c_void_p
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def p(a):
    return a.ctypes.data_as(P)


@pytest.mark.parametrize("bd", [8, 10])
def test_batch_matches_reference_fixture(hip_ctx, bd):
    src, ref, jobs, exp = load_fixture(bd)
    got = stats.run_hip(hip_ctx, src, ref, jobs, bd, satd=(bd == 8), psy_rd=PSY_RD)
    assert not mismatches(exp, got, bd)


@pytest.mark.parametrize("bd", [8, 10])
def test_batch_matches_oracle_on_random_jobs(hip_ctx, oracle, bd):
    rng = np.random.default_rng(40 + bd)
    W, H = 640, 360
    dt = np.uint8 if bd == 8 else np.uint16
    src = rng.integers(0, 1 << bd, (H, W)).astype(dt)
    ref = rng.integers(0, 1 << bd, (H, W)).astype(dt)
    jobs = stats.random_jobs(rng, W, H, 3000)
    odd = stats.random_jobs(rng, W, H, 64, sizes=[(12, 20), (1, 1), (3, 128), (128, 5), (24, 24), (100, 7)])  # not AV1 shapes: still defined
    jobs = np.concatenate([jobs, odd])
    a = pyoracle.block_stats(oracle, src, ref, jobs, bd, satd=(bd == 8))
    b = stats.run_hip(hip_ctx, src, ref, jobs, bd, satd=(bd == 8))
    assert not [m for m in mismatches(a, b, bd) if "psy" not in m]
    # psy-RD terms: AV1 shapes only (multiples of 4), three strengths incl. 0
    for psy_rd in (0.0, 0.5, 4.0):
        jp = jobs[:1500]
        a = pyoracle.block_stats(oracle, src, ref, jp, bd, satd=False, psy_rd=psy_rd)
        b = stats.run_hip(hip_ctx, src, ref, jp, bd, satd=False, psy_rd=psy_rd)
        assert not [m for m in mismatches(a, b, bd) if "satd" not in m], psy_rd


def test_batch_rejects_bad_descriptors(hip_ctx):
    L = api.lib()
    d = abi.BlockStatsDesc(bit_depth=9, n_jobs=1)
    assert L.svt_hip_block_stats_batch(hip_ctx._h, C.byref(d)) == 2
    d = abi.BlockStatsDesc(bit_depth=8, n_jobs=1)  # null planes
    assert L.svt_hip_block_stats_batch(hip_ctx._h, C.byref(d)) == 2
    d = abi.BlockStatsDesc(bit_depth=8, n_jobs=0)
    assert L.svt_hip_block_stats_batch(hip_ctx._h, C.byref(d)) == 0


@pytest.mark.parametrize("bd", [8, 10])
def test_batch_facade_matches_reference_fixture(hip_ctx, bd):
    """svt_spatial_full_distortion_kernel_facade (picture_operators_c.c:115-174) as the batch's facade_dist output."""
    src, ref, jobs, _ = load_fixture(bd)
    modes, comps, exp = load_facade_fixture(bd)
    for k, (_, _, psy_rd) in FACADE_SETTINGS.items():
        got = stats.run_hip(hip_ctx, src, ref, jobs, bd, satd=False, psy_rd=psy_rd, facade=facade_arg(modes, comps, k))
        assert np.array_equal(got["facade_dist"], exp[k]), k
    with pytest.raises(api.SvtHipError):  # temporal_layer_index beyond the reference's weight table
        stats.run_hip(hip_ctx, src, ref, jobs, bd, satd=False, facade=dict(pred_mode=modes, compound_type=comps, temporal_layer_index=6, spy_rd=1))


def test_batch_highbd_10_variance(hip_ctx, oracle):
    """svt_aom_highbd_10_variance{W}x{H} (svt_psnr.c:139-177): reference fixture on the AV1 shapes, oracle on every job."""
    src, ref, jobs, _ = load_fixture(10)
    v10, s10, ok = load_var10_fixture()
    got = stats.run_hip(hip_ctx, src, ref, jobs, 10, satd=False)
    want = pyoracle.block_stats(oracle, src, ref, jobs, 10, satd=False)
    assert np.array_equal(got["variance10"][ok], v10[ok]) and np.array_equal(got["var_sse10"][ok], s10[ok])
    assert np.array_equal(got["variance10"], want["variance10"]) and np.array_equal(got["var_sse10"], want["var_sse10"])
    with pytest.raises(api.SvtHipError):  # 8-bit planes have no highbd_10 variance
        d = abi.BlockStatsDesc(bit_depth=8, n_jobs=1, src=1, ref=1, jobs=1, variance10=1)
        hip_ctx.check(api.lib().svt_hip_block_stats_batch(hip_ctx._h, C.byref(d)), "svt_hip_block_stats_batch")


def test_leaf_highbd_10_variance(leaf, oracle):
    rng = np.random.default_rng(77)
    for (w, h) in abi.VARIANCE_SIZES:
        a = rng.integers(0, 1024, (h, w + 6)).astype(np.uint16)
        b = np.clip(a[:, :w + 2].astype(np.int32) + rng.integers(-300, 300, (h, w + 2)), 0, 1023).astype(np.uint16) if (w + h) % 3 else rng.integers(0, 1024, (h, w + 2)).astype(np.uint16)
        b = np.ascontiguousarray(b)
        s1, s2 = C.c_uint32(), C.c_uint32()
        want = oracle.orc_highbd_10_variance(p(a), w + 6, p(b), w + 2, w, h, C.byref(s1))
        got = getattr(leaf, f"svt_aom_highbd_10_variance{w}x{h}_hip")(C.c_void_p(a.ctypes.data >> 1), w + 6, C.c_void_p(b.ctypes.data >> 1), w + 2, C.byref(s2))
        assert (got & 0xFFFFFFFF, s2.value) == (want & 0xFFFFFFFF, s1.value), (w, h)


@pytest.fixture()
def leaf(hip_ctx):
    L = api.lib()
    assert L.svt_hip_leaf_bind(hip_ctx._h) == 0
    yield L
    L.svt_hip_leaf_bind(None)


def test_leaf_sad_loop_kernel_against_reference_kat(leaf):
    """svt_sad_loop_kernel_hip on the reference's known answers (tests/golden/sad_loop_kat.npz)."""
    z = np.load(os.path.join(GOLDEN, "sad_loop_kat.npz"))
    so = ro = 0
    for (bw, bh, sw, sh, skip, best, x, y), stride, rows in zip(z["meta"], z["ref_stride"], z["ref_rows"]):
        src = np.ascontiguousarray(z["src"][so:so + bw * bh]); so += bw * bh
        refp = np.ascontiguousarray(z["ref"][ro:ro + stride * rows]); ro += stride * rows
        b, xs, ys = C.c_uint64(0), C.c_int16(-7), C.c_int16(-7)
        leaf.svt_sad_loop_kernel_hip(p(src), C.c_uint32(int(bw)), p(refp), C.c_uint32(int(stride)), C.c_uint32(int(bh)), C.c_uint32(int(bw)),
                                     C.byref(b), C.byref(xs), C.byref(ys), C.c_uint32(int(stride)), C.c_uint8(int(skip)), C.c_int16(int(sw)), C.c_int16(int(sh)))
        assert (b.value, xs.value, ys.value) == (best, x, y), (bw, bh, sw, sh, skip)


def test_leaf_sad_variance_sse(leaf, oracle):
    rng = np.random.default_rng(9)
    leaf.svt_aom_sse_hip.restype = C.c_int64
    leaf.svt_spatial_full_distortion_kernel_hip.restype = C.c_uint64
    leaf.svt_full_distortion_kernel16_bits_hip.restype = C.c_uint64
    oracle.orc_spatial_sse8.restype = C.c_uint64
    oracle.orc_spatial_sse16.restype = C.c_uint64
    for (w, h) in abi.VARIANCE_SIZES:
        a = rng.integers(0, 256, (h, w + 3)).astype(np.uint8); b = rng.integers(0, 256, (h, w + 8)).astype(np.uint8)
        if (w + h) % 24 == 0:
            a[:] = 255; b[:] = 0
        s1, s2 = C.c_uint32(), C.c_uint32()
        va = getattr(leaf, f"svt_aom_variance{w}x{h}_hip")(p(a), w + 3, p(b), w + 8, C.byref(s1))
        vb = oracle.orc_variance8(p(a), w + 3, p(b), w + 8, w, h, C.byref(s2))
        assert (va & 0xFFFFFFFF, s1.value) == (vb & 0xFFFFFFFF, s2.value), (w, h)
        assert leaf.svt_nxm_sad_kernel_helper_hip(p(a), C.c_uint32(w + 3), p(b), C.c_uint32(w + 8), C.c_uint32(h), C.c_uint32(w)) == \
            oracle.orc_nxm_sad(p(a), C.c_uint32(w + 3), p(b), C.c_uint32(w + 8), C.c_uint32(h), C.c_uint32(w))
        assert leaf.svt_aom_sse_hip(p(a), w + 3, p(b), w + 8, w, h) == oracle.orc_spatial_sse8(p(a), C.c_uint32(0), C.c_uint32(w + 3), p(b), C.c_int32(0), C.c_uint32(w + 8), C.c_uint32(w), C.c_uint32(h))
        assert leaf.svt_spatial_full_distortion_kernel_hip(p(a), C.c_uint32(1), C.c_uint32(w + 3), p(b), C.c_int32(2), C.c_uint32(w + 8), C.c_uint32(w - 2), C.c_uint32(h)) == \
            oracle.orc_spatial_sse8(p(a), C.c_uint32(1), C.c_uint32(w + 3), p(b), C.c_int32(2), C.c_uint32(w + 8), C.c_uint32(w - 2), C.c_uint32(h))
        a16 = rng.integers(0, 1024, (h, w + 3)).astype(np.uint16); b16 = rng.integers(0, 1024, (h, w + 8)).astype(np.uint16)
        assert leaf.svt_full_distortion_kernel16_bits_hip(p(a16), C.c_uint32(1), C.c_uint32(w + 3), p(b16), C.c_int32(2), C.c_uint32(w + 8), C.c_uint32(w - 2), C.c_uint32(h)) == \
            oracle.orc_spatial_sse16(p(a16), C.c_uint32(1), C.c_uint32(w + 3), p(b16), C.c_int32(2), C.c_uint32(w + 8), C.c_uint32(w - 2), C.c_uint32(h))
        assert leaf.svt_aom_sad_16b_kernel_hip(p(a16), C.c_uint32(w + 3), p(b16), C.c_uint32(w + 8), C.c_uint32(h), C.c_uint32(w)) == \
            oracle.orc_sad_16b(p(a16), C.c_uint32(w + 3), p(b16), C.c_uint32(w + 8), C.c_uint32(h), C.c_uint32(w))
        leaf.svt_aom_highbd_sse_hip.restype = C.c_int64  # svt_aom_highbd_sse_c (enc_inter_prediction.c:559): the sum the 16-bit distortion kernel forms
        assert leaf.svt_aom_highbd_sse_hip(p(a16), w + 3, p(b16), w + 8, w, h) == \
            oracle.orc_spatial_sse16(p(a16), C.c_uint32(0), C.c_uint32(w + 3), p(b16), C.c_int32(0), C.c_uint32(w + 8), C.c_uint32(w), C.c_uint32(h))


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_leaf_hadamard_and_satd(leaf, oracle, n):
    """hadamard_test.cc patterns: random 9-bit residuals, all-max, all-min; compared coefficient by coefficient."""
    rng = np.random.default_rng(n)
    for pat in ("random", "max", "min"):
        res = rng.integers(-255, 256, (n, n + 3)).astype(np.int16)
        if pat == "max":
            res[:] = 255
        if pat == "min":
            res[:] = -255
        a, b = np.zeros(n * n, np.int32), np.zeros(n * n, np.int32)
        getattr(leaf, f"svt_aom_hadamard_{n}x{n}_hip")(p(res), C.c_ssize_t(n + 3), p(a))
        getattr(oracle, f"orc_hadamard_{n}x{n}")(p(res), C.c_ssize_t(n + 3), p(b))
        assert np.array_equal(a, b), (n, pat)
        assert leaf.svt_aom_satd_hip(p(a), n * n) == oracle.orc_satd(p(b), n * n)


def test_leaf_hadamard_path(leaf, oracle):
    rng = np.random.default_rng(77)
    leaf.svt_hip_hadamard_path.restype = C.c_uint32
    oracle.orc_hadamard_path.restype = C.c_uint32
    for n in (4, 8, 16, 32, 64, 128):
        a = rng.integers(0, 256, (n, n + 5)).astype(np.uint8); b = rng.integers(0, 256, (n, n + 9)).astype(np.uint8)
        assert leaf.svt_hip_hadamard_path(p(a), C.c_uint32(n + 5), p(b), C.c_uint32(n + 9), C.c_uint32(n)) == \
            oracle.orc_hadamard_path(p(a), C.c_uint32(n + 5), p(b), C.c_uint32(n + 9), C.c_uint32(n))


def test_leaf_psy_distortion(leaf, oracle):
    rng = np.random.default_rng(21)
    for name in ("svt_psy_distortion_hip", "svt_psy_distortion_hbd_hip", "get_svt_psy_full_dist_hip"):
        getattr(leaf, name).restype = C.c_uint64
    oracle.orc_psy_distortion.restype = C.c_uint64
    for (w, h) in [(4, 4), (16, 4), (8, 8), (32, 16), (64, 64), (128, 128)]:
        a = rng.integers(0, 256, (h, w + 3)).astype(np.uint8); b = rng.integers(0, 256, (h, w + 5)).astype(np.uint8)
        e = oracle.orc_psy_distortion(p(a), C.c_uint32(w + 3), p(b), C.c_uint32(w + 5), C.c_uint32(w), C.c_uint32(h), C.c_int(0))
        assert leaf.svt_psy_distortion_hip(p(a), C.c_uint32(w + 3), p(b), C.c_uint32(w + 5), C.c_uint32(w), C.c_uint32(h)) == e
        a16 = rng.integers(0, 1024, (h, w + 3)).astype(np.uint16); b16 = rng.integers(0, 1024, (h, w + 5)).astype(np.uint16)
        e16 = oracle.orc_psy_distortion(p(a16), C.c_uint32(w + 3), p(b16), C.c_uint32(w + 5), C.c_uint32(w), C.c_uint32(h), C.c_int(1))
        assert leaf.svt_psy_distortion_hbd_hip(p(a16), C.c_uint32(w + 3), p(b16), C.c_uint32(w + 5), C.c_uint32(w), C.c_uint32(h)) == e16
        got = leaf.get_svt_psy_full_dist_hip(p(a16), C.c_uint32(1), C.c_uint32(w + 3), p(b16), C.c_uint32(2), C.c_uint32(w + 5), C.c_uint32(w), C.c_uint32(h - 4 if h > 8 else h),
                                             C.c_uint8(1), C.c_double(0.75))
        hh = h - 4 if h > 8 else h
        if hh % 8 == 0 or hh < 8 or w < 8:  # keep to whole tiles
            a2 = a16.reshape(-1)[1:]; b2 = b16.reshape(-1)[2:]
            want = int(float(oracle.orc_psy_distortion(p(a2), C.c_uint32(w + 3), p(b2), C.c_uint32(w + 5), C.c_uint32(w), C.c_uint32(hh), C.c_int(1))) * 0.75)
            assert got == want, (w, h)


def test_leaf_psyex_facades(leaf, oracle):
    """svt_spatial_full_distortion_kernel_facade_hip / svt_spatial_psy_distortion_kernel_hip with the reference's prototypes."""
    leaf.svt_spatial_full_distortion_kernel_facade_hip.restype = leaf.svt_spatial_psy_distortion_kernel_hip.restype = C.c_uint64
    oracle.orc_spy_rd_facade.restype = C.c_int64
    oracle.orc_psy_distortion.restype = C.c_uint64
    rng = np.random.default_rng(33)
    for (w, h, mode, comp, tli) in [(64, 64, 0, 0, 3), (32, 32, 12, 1, 5), (16, 8, 9, 2, 2), (32, 64, 20, 3, 4), (8, 8, 16, 0, 1), (64, 64, 2, 0, 0)]:
        a = rng.integers(0, 256, (h + 1, w + 7)).astype(np.uint8); b = rng.integers(0, 256, (h + 1, w + 2)).astype(np.uint8)
        a2, b2 = a.reshape(-1)[3:].reshape(-1), b.reshape(-1)[1:].reshape(-1)
        sse = int(((a.reshape(-1)[3:3 + h * (w + 7)].reshape(h, w + 7)[:, :w].astype(np.int64) - b.reshape(-1)[1:1 + h * (w + 2)].reshape(h, w + 2)[:, :w]) ** 2).sum())
        for spy, psy in ((1, 0.0), (1, 1.0), (2, 0.0)):
            want = oracle.orc_spy_rd_facade(C.c_int64(sse), C.c_uint32(w), C.c_uint32(h), C.c_uint8(mode), C.c_uint8(comp), C.c_uint8(tli), C.c_double(psy), C.c_uint8(spy))
            got = leaf.svt_spatial_full_distortion_kernel_facade_hip(p(a), C.c_uint32(3), C.c_uint32(w + 7), p(b), C.c_int32(1), C.c_uint32(w + 2), C.c_uint32(w), C.c_uint32(h),
                                                                     C.c_bool(False), C.c_uint8(mode), C.c_uint8(comp), C.c_uint8(tli), C.c_double(psy), C.c_uint8(spy))
            assert got == want, (w, h, mode, comp, tli, spy, psy)
        a16 = (a.astype(np.uint16) << 2) | 1; b16 = (b.astype(np.uint16) << 2) | 2
        sse16 = int(((a16.reshape(-1)[3:3 + h * (w + 7)].reshape(h, w + 7)[:, :w].astype(np.int64) - b16.reshape(-1)[1:1 + h * (w + 2)].reshape(h, w + 2)[:, :w]) ** 2).sum())
        want = oracle.orc_spy_rd_facade(C.c_int64(sse16), C.c_uint32(w), C.c_uint32(h), C.c_uint8(mode), C.c_uint8(comp), C.c_uint8(tli), C.c_double(0.0), C.c_uint8(1))
        got = leaf.svt_spatial_full_distortion_kernel_facade_hip(p(a16), C.c_uint32(3), C.c_uint32(w + 7), p(b16), C.c_int32(1), C.c_uint32(w + 2), C.c_uint32(w), C.c_uint32(h),
                                                                 C.c_bool(True), C.c_uint8(mode), C.c_uint8(comp), C.c_uint8(tli), C.c_double(0.0), C.c_uint8(1))
        assert got == want, ("hbd", w, h, mode)
        e = oracle.orc_psy_distortion(p(a2), C.c_uint32(w + 7), p(b2), C.c_uint32(w + 2), C.c_uint32(w), C.c_uint32(h), C.c_int(0))
        for psy in (0.0, 0.6):
            got = leaf.svt_spatial_psy_distortion_kernel_hip(p(a), C.c_uint32(3), C.c_uint32(w + 7), p(b), C.c_int32(1), C.c_uint32(w + 2), C.c_uint32(w), C.c_uint32(h), C.c_double(psy))
            assert got == sse + (int(float(e) * psy) if psy > 0 else 0), (w, h, psy)


def test_sub_pixel_variance_fixture_and_oracle(hip_ctx, oracle):
    src, ref, jobs, exp = load_subpel_fixture()
    got = stats.run_hip(hip_ctx, src, ref, jobs, 8, satd=False)
    for k in exp:
        assert np.array_equal(exp[k], got[k]), k
    # every statistic of interpolated blocks, both bit depths, against the oracle
    for bd in (8, 10):
        rng = np.random.default_rng(60 + bd)
        W, H = 320, 200
        dt = np.uint8 if bd == 8 else np.uint16
        s = rng.integers(0, 1 << bd, (H, W)).astype(dt); r = rng.integers(0, 1 << bd, (H, W)).astype(dt)
        jb = stats.random_jobs(rng, W, H, 1200, subpel=True)
        a = pyoracle.block_stats(oracle, s, r, jb, bd, satd=(bd == 8), psy_rd=1.0)
        b = stats.run_hip(hip_ctx, s, r, jb, bd, satd=(bd == 8), psy_rd=1.0)
        assert not mismatches(a, b, bd), bd


def test_leaf_sub_pixel_variance(leaf, oracle):
    rng = np.random.default_rng(33)
    for (w, h) in [(4, 4), (8, 16), (16, 16), (64, 32), (128, 128)]:
        for (xo, yo) in [(0, 0), (3, 0), (0, 5), (7, 7), (4, 4), (1, 6)]:
            a = rng.integers(0, 256, (h + 1, w + 4)).astype(np.uint8); b = rng.integers(0, 256, (h, w + 2)).astype(np.uint8)
            s1, s2 = C.c_uint32(), C.c_uint32()
            x = getattr(leaf, f"svt_aom_sub_pixel_variance{w}x{h}_hip")(p(a), w + 4, xo, yo, p(b), w + 2, C.byref(s1)) & 0xFFFFFFFF
            y = oracle.orc_sub_pixel_variance8(p(a), w + 4, xo, yo, p(b), w + 2, w, h, C.byref(s2)) & 0xFFFFFFFF
            assert (x, s1.value) == (y, s2.value), (w, h, xo, yo)


def test_leaf_fixed_size_sad_and_highbd_variance(leaf, oracle):
    rng = np.random.default_rng(51)
    for (w, h) in abi.VARIANCE_SIZES:
        a = rng.integers(0, 256, (h, w + 3)).astype(np.uint8)
        refs = [rng.integers(0, 256, (h, w + 8)).astype(np.uint8) for _ in range(4)]
        want = [oracle.orc_nxm_sad(p(a), C.c_uint32(w + 3), p(r), C.c_uint32(w + 8), C.c_uint32(h), C.c_uint32(w)) for r in refs]
        assert getattr(leaf, f"svt_aom_sad{w}x{h}_hip")(p(a), w + 3, p(refs[0]), w + 8) == want[0]
        ptrs = (C.c_void_p * 4)(*[r.ctypes.data for r in refs])
        out = (C.c_uint32 * 4)()
        getattr(leaf, f"svt_aom_sad{w}x{h}x4d_hip")(p(a), w + 3, ptrs, w + 8, out)
        assert list(out) == want
    a16 = rng.integers(0, 1024, (32, 40)).astype(np.uint16); b16 = rng.integers(0, 1024, (32, 36)).astype(np.uint16)
    s1, s2 = C.c_uint32(), C.c_uint32()
    assert leaf.svt_aom_variance_highbd_hip(p(a16), 40, p(b16), 36, 32, 32, C.byref(s1)) & 0xFFFFFFFF == oracle.orc_variance16(p(a16), 40, p(b16), 36, 32, 32, C.byref(s2)) & 0xFFFFFFFF
    assert s1.value == s2.value


def test_leaf_coefficient_distortion_residual_and_estimate_transform(leaf, oracle):
    rng = np.random.default_rng(52)
    leaf.svt_av1_block_error_hip.restype = C.c_int64
    oracle.orc_block_error.restype = C.c_int64
    for (w, h) in [(4, 4), (16, 8), (32, 32), (64, 64)]:
        co = rng.integers(-(1 << 20), 1 << 20, (h, w + 2)).astype(np.int32); rc = (co + rng.integers(-300, 300, co.shape)).astype(np.int32)
        a, b = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        leaf.svt_full_distortion_kernel32_bits_hip(p(co), C.c_uint32(w + 2), p(rc), C.c_uint32(w + 2), a, C.c_uint32(w), C.c_uint32(h))
        oracle.orc_full_distortion32(p(co), C.c_uint32(w + 2), p(rc), C.c_uint32(w + 2), b, C.c_uint32(w), C.c_uint32(h))
        assert list(a) == list(b)
        leaf.svt_full_distortion_kernel_cbf_zero32_bits_hip(p(co), C.c_uint32(w + 2), a, C.c_uint32(w), C.c_uint32(h))
        oracle.orc_full_distortion32_cbf_zero(p(co), C.c_uint32(w + 2), b, C.c_uint32(w), C.c_uint32(h))
        assert list(a) == list(b)
        flat, flat2 = np.ascontiguousarray(co[:, :w]).reshape(-1), np.ascontiguousarray(rc[:, :w]).reshape(-1)
        z1, z2 = C.c_int64(), C.c_int64()
        assert leaf.svt_av1_block_error_hip(p(flat), p(flat2), C.c_ssize_t(flat.size), C.byref(z1)) == oracle.orc_block_error(p(flat), p(flat2), C.c_ssize_t(flat.size), C.byref(z2))
        assert z1.value == z2.value
        for bd, dt, fn, fo in ((8, np.uint8, "svt_residual_kernel8bit_hip", "orc_residual8"), (10, np.uint16, "svt_residual_kernel16bit_hip", "orc_residual16")):
            x = rng.integers(0, 1 << bd, (h, w + 3)).astype(dt); y = rng.integers(0, 1 << bd, (h, w + 5)).astype(dt)
            r1 = np.full((h, w + 1), 77, np.int16); r2 = r1.copy()
            getattr(leaf, fn)(p(x), C.c_uint32(w + 3), p(y), C.c_uint32(w + 5), p(r1), C.c_uint32(w + 1), C.c_uint32(w), C.c_uint32(h))
            getattr(oracle, fo)(p(x), C.c_uint32(w + 3), p(y), C.c_uint32(w + 5), p(r2), C.c_uint32(w + 1), C.c_uint32(w), C.c_uint32(h))
            assert np.array_equal(r1, r2)
    # svt_hip_estimate_transform == forward transform + 64-point repack / energy of the oracle, any int16 residual, every pf_shape
    from txfm_cases import TX_H, TX_W, valid_types
    oracle.orc_handle_transform.restype = C.c_uint64
    for ts in (0, 2, 4, 9, 12, 17):
        W, H = TX_W[ts], TX_H[ts]
        for tt in valid_types(ts)[:3]:
            for pf in range(4):
                res = rng.integers(-32768, 32768, (H, W + 3)).astype(np.int16)
                got = np.zeros(min(W, 32) * min(H, 32), np.int32); tq = C.c_uint64()
                assert leaf.svt_hip_estimate_transform(p(res), C.c_uint32(W + 3), p(got), ts, C.byref(tq), tt, pf) == 0
                full = np.zeros(W * H, np.int32)
                oracle.orc_fwd_txfm2d(p(res), p(full), C.c_uint32(W + 3), C.c_int(tt), C.c_int(ts))
                if pf:
                    f2 = full.reshape(H, W)
                    keep = np.zeros((H, W), bool)
                    if pf == 3:
                        keep[0, 0] = True
                    else:
                        keep[:H >> pf, :W >> pf] = True
                    f2[~keep] = 0
                e = oracle.orc_handle_transform(p(full), C.c_int(ts))
                assert np.array_equal(got, full[:got.size]), (ts, tt, pf)
                assert tq.value == (0 if pf else e), (ts, tt, pf)


@pytest.mark.parametrize("sub_sad", [0, 1])
def test_leaf_ext_sad_family(leaf, oracle, sub_sad):
    """svt_ext_{all,eight}_sad_calculation_* and the single-point forms (motion_estimation.c:98-425) + svt_initialize_buffer_32bits:
    same random grid as the oracle-vs-reference test (tests/test_oracle_vs_ref.py::test_ext_sad_family)."""
    rng = np.random.default_rng(11)
    for it in range(12):
        stride_s, stride_r = 64 + int(rng.integers(0, 9)), 80 + int(rng.integers(0, 9))
        src = rng.integers(0, 256, (64, stride_s), dtype=np.uint8)
        refp = rng.integers(0, 256, (64, stride_r), dtype=np.uint8)
        if it % 5 == 0:
            refp[:, :64] = src[:, :64]  # ties / zero SADs
        mv = int(rng.integers(0, 1 << 32))
        if it % 2:
            init = rng.integers(0, 20000, 85).astype(np.uint32)
        else:
            init = np.zeros(85, np.uint32)
            leaf.svt_initialize_buffer_32bits_hip(p(init), C.c_uint32(21), C.c_uint32(1), C.c_uint32(128 * 128 * 255))  # 21 * 4 + 1 = 85 (motion_estimation.c:1366)
            assert (init == 128 * 128 * 255).all()
        res = []
        for lib, pre, suf in ((leaf, "svt_ext_", "_hip"), (oracle, "orc_ext_", "")):
            bs, bm = init.copy(), np.zeros(85, np.uint32)
            e16, e8, e32 = np.zeros((16, 8), np.uint32), np.zeros((64, 8), np.uint32), np.zeros((4, 8), np.uint32)
            getattr(lib, pre + "all_sad_calculation_8x8_16x16" + suf)(p(src), C.c_uint32(stride_s), p(refp), C.c_uint32(stride_r), C.c_uint32(mv), p(bs[21:]), p(bs[5:]),
                                                                     p(bm[21:]), p(bm[5:]), p(e16), p(e8), C.c_bool(bool(sub_sad)))
            getattr(lib, pre + "eight_sad_calculation_32x32_64x64" + suf)(p(e16), p(bs[1:]), p(bs), p(bm[1:]), p(bm), C.c_uint32(mv), p(e32))
            s16, s8 = np.zeros(16, np.uint32), np.zeros(64, np.uint32)
            bs2, bm2 = init.copy(), np.zeros(85, np.uint32)
            getattr(lib, pre + "sad_calculation_8x8_16x16" + suf)(p(src), C.c_uint32(stride_s), p(refp), C.c_uint32(stride_r), p(bs2[21:]), p(bs2[5:]), p(bm2[21:]),
                                                                 p(bm2[5:]), C.c_uint32(mv), p(s16), p(s8), C.c_bool(bool(sub_sad)))
            s32 = np.zeros(4, np.uint32)
            s16full = e16[:, 0].copy()
            getattr(lib, pre + "sad_calculation_32x32_64x64" + suf)(p(s16full), p(bs2[1:]), p(bs2), p(bm2[1:]), p(bm2), C.c_uint32(mv), p(s32))
            res.append((bs, bm, e16, e8, e32, bs2, bm2, s16[:1], s8[:4], s32))
        for i, (a, b) in enumerate(zip(*res)):
            assert np.array_equal(a, b), (it, i)


def _quant_rows(rng, dc, ac):
    """int16[8] MacroblockPlane-style rows ([0] = DC, [1..7] = AC) of an rd.quant_row_from_step quantizer."""
    from svt_av1_psyex_amd import rd
    r = rd.quant_row_from_step(dc, ac)
    row = lambda k: np.array([r[k][0]] + [r[k][1]] * 7, np.int16)
    return {k: row(k) for k in ("zbin", "round", "quant", "quant_shift", "round_fp", "quant_fp", "dequant")}


@pytest.mark.parametrize("tx_size,log_scale", [(0, 0), (2, 0), (8, 0), (3, 1), (9, 1), (4, 2), (11, 2)])
def test_leaf_quantizers(leaf, oracle, tx_size, log_scale):
    """The ten quantizer pointers (aom_dsp_rtcd.h:244-263) against oracle/dsp_oracle.c (itself pinned on the reference's `_c` bodies):
    real scan orders, flat and random matrices, zero / DC-only / extreme / random coefficients (test/quantize_func_test.cc patterns)."""
    rng = np.random.default_rng(100 + tx_size)
    n = min(api.lib().svt_hip_tx_size_wide(tx_size), 32) * min(api.lib().svt_hip_tx_size_high(tx_size), 32)
    scan, iscan = np.zeros(n, np.int16), np.zeros(n, np.int16)
    assert api.lib().svt_hip_scan_order(tx_size, 0, p(scan), p(iscan)) == n
    for case, (dc, ac) in enumerate([(8, 9), (60, 75), (300, 410), (1336, 1828)]):
        t = _quant_rows(rng, dc, ac)
        for pat in ("random", "zero", "dc", "extreme", "small"):
            co = {"random": rng.integers(-40000, 40000, n), "zero": np.zeros(n), "dc": np.r_[rng.integers(-90000, 90000), np.zeros(n - 1)],
                  "extreme": rng.choice([-(1 << 20), (1 << 20) - 1], n), "small": rng.integers(-ac, ac + 1, n)}[pat].astype(np.int32)
            qm, iqm = rng.integers(16, 256, n).astype(np.uint8), rng.integers(16, 256, n).astype(np.uint8)
            for hbd in (0, 1):
                for use_qm in (0, 1):
                    m, im = (p(qm), p(iqm)) if use_qm else (None, None)
                    # "b": svt_aom_{highbd_,}quantize_b and the _qm pointers share the prototype
                    for name in (("svt_aom_highbd_quantize_b_hip", "svt_av1_highbd_quantize_b_qm_hip") if hbd else ("svt_aom_quantize_b_hip", "svt_av1_quantize_b_qm_hip")):
                        q1, d1, q2, d2 = (np.full(n, 7, np.int32) for _ in range(4))
                        e1, e2 = C.c_uint16(99), C.c_uint16(99)
                        getattr(leaf, name)(p(co), C.c_ssize_t(n), p(t["zbin"]), p(t["round"]), p(t["quant"]), p(t["quant_shift"]), p(q1), p(d1), p(t["dequant"]),
                                            C.byref(e1), p(scan), p(iscan), m, im, C.c_int32(log_scale))
                        oracle.orc_quantize_b(p(co), C.c_ssize_t(n), p(t["zbin"]), p(t["round"]), p(t["quant"]), p(t["quant_shift"]), p(q2), p(d2), p(t["dequant"]),
                                              C.byref(e2), p(scan), m, im, log_scale, hbd)
                        assert e1.value == e2.value and np.array_equal(q1, q2) and np.array_equal(d1, d2), (name, case, pat, use_qm)
                    # "fp"
                    q1, d1, q2, d2 = (np.full(n, 7, np.int32) for _ in range(4))
                    e1, e2 = C.c_uint16(99), C.c_uint16(99)
                    common = (p(co), C.c_ssize_t(n), p(t["zbin"]), p(t["round_fp"]), p(t["quant_fp"]), p(t["quant_shift"]), p(q1), p(d1), p(t["dequant"]), C.byref(e1), p(scan), p(iscan))
                    if use_qm:
                        getattr(leaf, "svt_av1_highbd_quantize_fp_qm_hip" if hbd else "svt_av1_quantize_fp_qm_hip")(*common, m, im, C.c_int16(log_scale))
                    elif hbd:
                        leaf.svt_av1_highbd_quantize_fp_hip(*common, C.c_int16(log_scale))
                    else:
                        getattr(leaf, ("svt_av1_quantize_fp_hip", "svt_av1_quantize_fp_32x32_hip", "svt_av1_quantize_fp_64x64_hip")[log_scale])(*common)
                    oracle.orc_quantize_fp(p(co), C.c_ssize_t(n), p(t["round_fp"]), p(t["quant_fp"]), p(q2), p(d2), p(t["dequant"]), C.byref(e2), p(scan), m, im, log_scale, hbd)
                    assert e1.value == e2.value and np.array_equal(q1, q2) and np.array_equal(d1, d2), ("fp", hbd, case, pat, use_qm)


def test_leaf_cul_level_and_fwht4x4(leaf, oracle):
    """svt_av1_compute_cul_level_hip / svt_av1_fwht4x4_hip with the reference's prototypes (aom_dsp_rtcd.h:904,208) against the oracle."""
    rng = np.random.default_rng(78)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    leaf.svt_av1_compute_cul_level_hip.restype = oracle.orc_compute_cul_level.restype = C.c_uint8
    for ts in (0, 2, 3, 9):
        n = min(abi.TX_W[ts], 32) * min(abi.TX_H[ts], 32)
        scan, iscan = np.zeros(n, np.int16), np.zeros(n, np.int16)
        assert oracle.orc_scan_order(ts, 0, p(scan), p(iscan)) == n
        for trial in range(12):
            q = np.zeros(n, np.int32)
            k = int(rng.integers(0, n + 1)) if trial else 0
            mag = int(rng.choice([1, 3, 70, 5000]))
            q[scan[:k]] = rng.integers(-mag, mag + 1, k)
            nz = np.flatnonzero(q[scan])
            eob = np.array([nz[-1] + 1 if len(nz) else 0], np.uint16)
            assert leaf.svt_av1_compute_cul_level_hip(p(scan), p(q), p(eob)) == oracle.orc_compute_cul_level(p(scan), p(q), p(eob)), (ts, trial)
    for trial in range(24):
        stride = int(rng.integers(4, 12))
        lim = int(rng.choice([255, 1023, 32767]))
        src = rng.integers(-lim, lim + 1, 4 * stride).astype(np.int16)
        a, b = np.zeros(16, np.int32), np.zeros(16, np.int32)
        leaf.svt_av1_fwht4x4_hip(p(src), p(a), C.c_uint32(stride))
        oracle.orc_fwht4x4(p(src), p(b), C.c_uint32(stride))
        assert np.array_equal(a, b), trial


# ---- hierarchical jobs: a 64x64 region's 85 nested blocks from one read of its samples (SvtHipBlockStatsDesc.pyramids) ----
def _regions(rng, w, h, n):
    reg = np.zeros(n, dtype=abi.BLOCK_JOB_DTYPE)
    for i in range(n):
        x0, y0, x1, y1 = rng.integers(0, w - 63), rng.integers(0, h - 63), rng.integers(0, w - 63), rng.integers(0, h - 63)
        reg[i] = (y0 * w + x0, y1 * w + x1, 64, 64, 0, 0)
    return reg


def test_pyramid_jobs_8bit_equal_plain_jobs_and_oracle(hip_ctx, oracle):
    rng = np.random.default_rng(77)
    w, h = 416, 240
    src = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ref = np.clip(src.astype(np.int32) + rng.integers(-30, 31, src.shape), 0, 255).astype(np.uint8)
    ref[:64, :64] = 255 - src[:64, :64]  # a region of large residuals
    reg = _regions(rng, w, h, 23)
    reg[0] = (0, 0, 64, 64, 0, 0)
    plain = stats.random_jobs(rng, w, h, 50, square_only=True)
    expanded = np.concatenate([plain] + [stats.expand_pyramid(r, w, w) for r in reg])
    want = pyoracle.block_stats(oracle, src, ref, expanded, 8, psy_rd=0.75)
    flat = stats.run_hip(hip_ctx, src, ref, expanded, 8, psy_rd=0.75)
    got = stats.run_hip(hip_ctx, src, ref, plain, 8, psy_rd=0.75, pyramids=reg)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=f"{k} vs oracle")
        np.testing.assert_array_equal(got[k], flat[k], err_msg=f"{k} vs plain jobs")
    nosatd = stats.run_hip(hip_ctx, src, ref, plain, 8, satd=False, psy_rd=0.75, pyramids=reg)  # without hadamard_path: the one-wave form
    for k in nosatd:
        np.testing.assert_array_equal(nosatd[k], want[k], err_msg=f"{k} (no SATD) vs oracle")


def test_pyramid_jobs_10bit_psy_facade(hip_ctx, oracle):
    rng = np.random.default_rng(78)
    w, h = 320, 192
    src = rng.integers(0, 1024, (h, w)).astype(np.uint16)
    ref = np.clip(src.astype(np.int32) + rng.integers(-90, 91, src.shape), 0, 1023).astype(np.uint16)
    ref[64:128, 64:128] = 1023 - src[64:128, 64:128]
    reg = _regions(rng, w, h, 17)
    reg[0] = (64 * w + 64, 64 * w + 64, 64, 64, 0, 0)
    expanded = np.concatenate([stats.expand_pyramid(r, w, w) for r in reg])
    n = len(expanded)
    facade = dict(pred_mode=rng.integers(0, 25, n).astype(np.uint8), compound_type=rng.integers(0, 4, n).astype(np.uint8), temporal_layer_index=3, spy_rd=1)
    want = pyoracle.block_stats(oracle, src, ref, expanded, 10, satd=False, psy_rd=1.35, facade=facade)
    got = stats.run_hip(hip_ctx, src, ref, np.zeros(0, abi.BLOCK_JOB_DTYPE), 10, satd=False, psy_rd=1.35, facade=facade, pyramids=reg)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
