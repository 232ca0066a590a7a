"""CPU, build container only: the oracle against the reference's own functions (oracle/_ref/libsvtref.so),
on fresh random inputs.  Parameter grids follow the reference's tests (test/SadTest.cc:57-112,161-207,433-444)."""
import ctypes as C

import numpy as np
import pytest

from me_cases import MCTF_GRID, MCTF_OUTPUTS, MeCase, compare

P = C.c_void_p


def _u8(a):
    return np.ascontiguousarray(a).ctypes.data_as(P)


@pytest.mark.parametrize("pattern", ["random", "ref_max", "src_max", "unaligned"])
def test_sad_loop_and_nxm(ref, oracle, pattern):
    rng = np.random.default_rng(7)
    for (bw, bh) in [(16, 8), (16, 16), (32, 16), (64, 32), (8, 4), (12, 6), (64, 64), (128, 64)]:
        for (sw, sh) in [(1, 1), (8, 3), (24, 7), (48, 32), (7, 5)]:
            for skip in (0, 1):
                stride = bw + sw + (3 if pattern == "unaligned" else 16)
                rows = bh * 2 + sh + 1
                if pattern == "ref_max":
                    src = np.zeros((bh * 2, bw), np.uint8); refp = np.full((rows, stride), 255, np.uint8)
                elif pattern == "src_max":
                    src = np.full((bh * 2, bw), 255, np.uint8); refp = np.zeros((rows, stride), np.uint8)
                else:
                    src = rng.integers(0, 256, (bh * 2, bw), dtype=np.uint8); refp = rng.integers(0, 256, (rows, stride), dtype=np.uint8)
                for sub in (1, 2):  # block rows every `sub` plane rows, as SUB_SAD_SEARCH does
                    out = []
                    for lib, fn in ((ref, "svt_sad_loop_kernel_c"), (oracle, "orc_sad_loop_kernel")):
                        b, x, y = C.c_uint64(0), C.c_int16(-3), C.c_int16(-3)
                        getattr(lib, fn)(_u8(src), C.c_uint32(bw * sub), _u8(refp), C.c_uint32(stride * sub), C.c_uint32(bh), C.c_uint32(bw),
                                         C.byref(b), C.byref(x), C.byref(y), C.c_uint32(stride), C.c_uint8(skip), C.c_int16(sw), C.c_int16(sh))
                        out.append((b.value, x.value, y.value))
                    assert out[0] == out[1], (bw, bh, sw, sh, skip, sub)
                a = ref.svt_nxm_sad_kernel_helper_c(_u8(src), C.c_uint32(bw), _u8(refp), C.c_uint32(stride), C.c_uint32(bh), C.c_uint32(bw))
                b = oracle.orc_nxm_sad(_u8(src), C.c_uint32(bw), _u8(refp), C.c_uint32(stride), C.c_uint32(bh), C.c_uint32(bw))
                assert a == b


@pytest.mark.parametrize("sub_sad", [0, 1])
def test_ext_sad_family(ref, oracle, sub_sad):
    rng = np.random.default_rng(11)
    for it in range(40):
        stride_s, stride_r = 64 + int(rng.integers(0, 9)), 80 + int(rng.integers(0, 9))
        src = rng.integers(0, 256, (64, stride_s), dtype=np.uint8)
        refp = rng.integers(0, 256, (64, stride_r), dtype=np.uint8)
        if it % 5 == 0:
            refp[:, :64] = src[:, :64]  # ties / zero SADs
        mv = int(rng.integers(0, 1 << 32))
        init = rng.integers(0, 20000, 85).astype(np.uint32) if it % 2 else np.full(85, 128 * 128 * 255, np.uint32)
        res = []
        for lib, pre in ((ref, "svt_ext_"), (oracle, "orc_ext_")):
            suf = "_c" if lib is ref else ""
            bs, bm = init.copy(), np.zeros(85, np.uint32)
            e16, e8, e32 = np.zeros((16, 8), np.uint32), np.zeros((64, 8), np.uint32), np.zeros((4, 8), np.uint32)
            getattr(lib, pre + "all_sad_calculation_8x8_16x16" + suf)(_u8(src), C.c_uint32(stride_s), _u8(refp), C.c_uint32(stride_r), C.c_uint32(mv),
                                                                     bs[21:].ctypes.data_as(P), bs[5:].ctypes.data_as(P), bm[21:].ctypes.data_as(P),
                                                                     bm[5:].ctypes.data_as(P), e16.ctypes.data_as(P), e8.ctypes.data_as(P), C.c_bool(bool(sub_sad)))
            getattr(lib, pre + "eight_sad_calculation_32x32_64x64" + suf)(e16.ctypes.data_as(P), bs[1:].ctypes.data_as(P), bs.ctypes.data_as(P),
                                                                         bm[1:].ctypes.data_as(P), bm.ctypes.data_as(P), C.c_uint32(mv), e32.ctypes.data_as(P))
            # single-point family on the first 16x16
            s16, s8 = np.zeros(16, np.uint32), np.zeros(64, np.uint32)
            bs2, bm2 = init.copy(), np.zeros(85, np.uint32)
            getattr(lib, pre + "sad_calculation_8x8_16x16" + suf)(_u8(src), C.c_uint32(stride_s), _u8(refp), C.c_uint32(stride_r), bs2[21:].ctypes.data_as(P),
                                                                 bs2[5:].ctypes.data_as(P), bm2[21:].ctypes.data_as(P), bm2[5:].ctypes.data_as(P), C.c_uint32(mv),
                                                                 s16.ctypes.data_as(P), s8.ctypes.data_as(P), C.c_bool(bool(sub_sad)))
            s32 = np.zeros(4, np.uint32)
            s16full = e16[:, 0].copy()
            getattr(lib, pre + "sad_calculation_32x32_64x64" + suf)(s16full.ctypes.data_as(P), bs2[1:].ctypes.data_as(P), bs2.ctypes.data_as(P),
                                                                   bm2[1:].ctypes.data_as(P), bm2.ctypes.data_as(P), C.c_uint32(mv), s32.ctypes.data_as(P))
            res.append((bs, bm, e16, e32, bs2, bm2, s16[:1], s8[:4], s32))
        for a, b in zip(*res):
            assert np.array_equal(a, b)


ME_GRID = [
    dict(width=352, height=288, enc_mode=em, temporal_layer_index=tl, seed=seed, kind=kind)
    for em, tl, seed, kind in [(-1, 1, 3, "pan"), (2, 2, 4, "pan"), (5, 1, 5, "noise"), (7, 3, 6, "pan"), (9, 1, 7, "fastpan"),
                               (11, 2, 8, "pan"), (13, 4, 9, "flat"), (6, 4, 10, "extremes")]
] + [
    dict(width=640, height=360, enc_mode=6, cur=4, refs={(0, 0): 3, (0, 1): 2, (0, 2): 1, (0, 3): 0, (1, 0): 5, (1, 1): 6, (1, 2): 7}, n_frames=8),
    dict(width=640, height=360, enc_mode=3, cur=4, refs={(0, 0): 0, (1, 0): 8}, n_frames=9, gm_enabled=1),
    dict(width=352, height=288, enc_mode=9, rtc_tune=1, sc_class1=1, refs={(0, 0): 1, (0, 1): 0}, temporal_layer_index=0),
    dict(width=352, height=288, enc_mode=10, rtc_tune=1, refs={(0, 0): 1, (1, 0): 3}),
    dict(width=1280, height=720, enc_mode=6, cur=2, refs={(0, 0): 1, (1, 0): 3}, seed=21),
]


@pytest.mark.parametrize("kw", ME_GRID, ids=lambda k: f"{k['width']}x{k['height']}_m{k['enc_mode']}_{k.get('kind', 'pan')}")
def test_me_picture_oracle_equals_reference(ref, kw):
    case = MeCase(**kw)
    assert not compare(case.run_cpu("ref"), case.run_cpu("oracle"))


@pytest.mark.parametrize("kw", MCTF_GRID, ids=lambda k: f"{k['width']}x{k['height']}_m{k['enc_mode']}_th{k['mctf_exit_th']}")
def test_mctf_oracle_equals_reference(ref, kw):
    """me_type == ME_MCTF (motion_estimation.c:1299-1302,3103-3126): unscaled distance, no pruning, HME-SAD early exit."""
    case = MeCase(**kw)
    a, b = case.run_cpu("ref"), case.run_cpu("oracle")
    assert not compare(a, b, MCTF_OUTPUTS)
    exits = (a["hme_sad"].reshape(-1, 8)[:, 0] < kw["mctf_exit_th"]).mean()
    if kw["mctf_exit_th"] == 6940:
        assert 0.02 < exits < 0.98, exits  # the mixed case really mixes


def test_full_sad_search_method(ref):
    def full(cfg):
        cfg.hme_search_method = 1
        cfg.me_search_method = 1
    case = MeCase(352, 288, enc_mode=6, cfg_edit=full)
    assert not compare(case.run_cpu("ref"), case.run_cpu("oracle"))


@pytest.mark.parametrize("w,h,enc_mode", [(64, 64, 6), (72, 80, 2), (128, 64, 11), (64, 200, 6)])
def test_smallest_pictures_oracle_vs_reference(ref, w, h, enc_mode):
    """The cases of tests/test_me_gpu.py::test_smallest_pictures: oracle == reference."""
    case = MeCase(w, h, enc_mode=enc_mode, refs={(0, 0): 0, (1, 0): 3, (0, 1): 1}, seed=w + h + enc_mode)
    assert not compare(case.run_cpu("ref"), case.run_cpu("oracle"))
