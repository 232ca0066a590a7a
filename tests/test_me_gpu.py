"""GPU: the HIP open-loop ME (libsvthip.so, through the C-ABI) against the golden fixtures and the oracle."""
import numpy as np
import pytest

from golden_io import GoldenMeCase, me_fixture_names
from me_cases import MCTF_GRID, MCTF_OUTPUTS, MeCase, compare, fill_unsearched
from test_oracle_vs_ref import ME_GRID

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True, params=[0, 2], ids=["one-kernel", "staged"])
def me_form(request, hip_ctx):
    """Every test of this module runs with the per-block pipeline in both forms (the default picks by launch size: small test pictures
    would never reach the staged kernels)."""
    hip_ctx.set_me_staged(request.param)
    yield request.param
    hip_ctx.set_me_staged(1)


@pytest.mark.parametrize("name", me_fixture_names())
def test_hip_matches_reference_fixture(hip_ctx, name):
    case = GoldenMeCase(name)
    assert not compare(case.expected, case.run_hip(hip_ctx)), name


@pytest.mark.parametrize("kw", ME_GRID, ids=lambda k: f"{k['width']}x{k['height']}_m{k['enc_mode']}_{k.get('kind', 'pan')}")
def test_hip_matches_oracle(hip_ctx, kw):
    case = MeCase(**kw)
    assert not compare(case.run_cpu("oracle"), case.run_hip(hip_ctx))


def test_full_sad_search_method(hip_ctx):
    def full(cfg):
        cfg.hme_search_method = 1
        cfg.me_search_method = 1
    case = MeCase(352, 288, enc_mode=6, cfg_edit=full)
    assert not compare(case.run_cpu("oracle"), case.run_hip(hip_ctx))


def test_large_search_areas_are_tiled(hip_ctx):
    """MR-class search areas do not fit the LDS arena in one piece: exercises the tile planner."""
    def big(cfg):
        cfg.me_sa.sa_min.width = cfg.me_sa.sa_min.height = 96
        cfg.me_sa.sa_max.width = cfg.me_sa.sa_max.height = 192
        cfg.hme_l0_sa.sa_min.width = cfg.hme_l0_sa.sa_min.height = 240
        cfg.hme_l0_sa.sa_max.width = cfg.hme_l0_sa.sa_max.height = 480
        cfg.me_early_exit_th = 0
        cfg.me_8x8_var_enabled = 0
    case = MeCase(640, 360, enc_mode=0, cfg_edit=big, kind="noise", seed=3)
    assert not compare(case.run_cpu("oracle"), case.run_hip(hip_ctx))


def test_device_pyramid_equals_host_pyramid(hip_ctx):
    case = MeCase(360, 296, enc_mode=6, seed=5)
    pic = hip_ctx.upload(case.cur, device_pyramid=True)
    try:
        for level in (0, 1, 2):
            buf, stride, p, w, h = case.cur.planes[level]
            assert np.array_equal(pic.download(level), buf[:, :w + 2 * p]), level
    finally:
        pic.free()
    assert not compare(case.run_hip(hip_ctx, device_pyramid=False), case.run_hip(hip_ctx, device_pyramid=True))


def test_row_bands_equal_whole_picture(hip_ctx):
    """The multi-GPU partition: any split into b64 row bands reproduces the whole-picture results."""
    case = MeCase(640, 360, enc_mode=6, seed=9)
    whole = case.run_hip(hip_ctx)
    cur = hip_ctx.upload(case.cur)
    refs = {k: hip_ctx.upload(v) for k, v in case.refs.items()}
    try:
        merged = None
        for start, count in ((0, 2), (2, 1), (3, 3)):
            case.desc.b64_row_start, case.desc.b64_row_count = start, count
            part = hip_ctx.me_picture(case.cfg, case.desc, cur, refs)
            w64 = 10
            if merged is None:
                merged = {k: np.zeros_like(v) for k, v in part.items()}
            for k, v in part.items():
                merged[k][start * w64:(start + count) * w64] = v[start * w64:(start + count) * w64]
        assert not compare(whole, merged)
    finally:
        case.desc.b64_row_start = case.desc.b64_row_count = 0
        cur.free()
        for r in refs.values():
            r.free()


def test_invalid_descriptors_fail_loudly(hip_ctx):
    from svt_av1_psyex_amd import api
    case = MeCase(352, 288, enc_mode=6)
    cur = hip_ctx.upload(case.cur)
    try:
        with pytest.raises(api.SvtHipError):  # reference [1][0] missing
            hip_ctx.me_picture(case.cfg, case.desc, cur, {(0, 0): cur})
        case.cfg.num_hme_sa_w = 3
        with pytest.raises(api.SvtHipError):
            hip_ctx.me_picture(case.cfg, case.desc, cur, {(0, 0): cur, (1, 0): cur})
    finally:
        cur.free()


@pytest.mark.parametrize("size,dist", [((1920, 1080), 2), ((3840, 2160), 1), ((3840, 2160), 8)])
def test_full_size_pictures_match_oracle(hip_ctx, size, dist):
    """BASELINE configs 1-3 at full size: the oracle finishes a 2160p picture in about a second."""
    layer = {1: 4, 2: 3, 4: 2, 8: 1}[dist]
    case = MeCase(size[0], size[1], enc_mode=6, cur=8, refs={(0, 0): 8 - dist, (1, 0): 8 + dist}, n_frames=17, seed=11, temporal_layer_index=layer)
    got = case.run_hip(hip_ctx)
    assert not compare(case.run_cpu("oracle"), got)
    # domain property: the sequence pans by (5,3) px per frame, so the dominant list-0 64x64 MV is (5,3)*dist
    mv = got["sb_best_mv"].reshape(-1, 2, 4, 85)[:, 0, 0, 0]
    x, y = (mv & 0xFFFF).astype(np.int16), (mv >> 16).astype(np.int16)
    assert np.median(x) == 5 * dist and np.median(y) == 3 * dist


def _fuzz_cfg(rng):
    def edit(cfg):
        cfg.me_sa.sa_min.width = int(rng.choice([8, 16, 24, 40, 104]))
        cfg.me_sa.sa_min.height = int(rng.choice([3, 8, 16, 37, 104]))
        cfg.me_sa.sa_max.width = max(cfg.me_sa.sa_min.width, int(rng.choice([8, 32, 64, 136])))
        cfg.me_sa.sa_max.height = max(cfg.me_sa.sa_min.height, int(rng.choice([3, 16, 32, 120])))
        cfg.hme_l0_sa.sa_min.width = int(rng.choice([8, 16, 32, 64]))
        cfg.hme_l0_sa.sa_min.height = int(rng.choice([8, 16, 32, 64]))
        cfg.hme_l0_sa.sa_max.width = int(rng.choice([96, 192, 320]))
        cfg.hme_l0_sa.sa_max.height = int(rng.choice([96, 192, 320]))
        cfg.hme_l1_sa.width, cfg.hme_l1_sa.height = int(rng.choice([8, 16])), int(rng.choice([3, 5, 16]))
        cfg.hme_l2_sa.width, cfg.hme_l2_sa.height = int(rng.choice([8, 16])), int(rng.choice([3, 7, 16]))
        cfg.me_early_exit_th = int(rng.choice([0, 64 * 64, 64 * 64 * 8]))
        cfg.me_8x8_var_enabled = int(rng.integers(0, 2))
        cfg.hme_search_method = int(rng.integers(0, 2))
        cfg.me_search_method = int(rng.integers(0, 2))
        cfg.prehme_enable = int(rng.integers(0, 2)) if cfg.prehme_sa_cfg[0].sa_max.width else 0
        cfg.prehme_skip_search_line = int(rng.integers(0, 2))
        cfg.enable_me_sr_adjustment = int(rng.choice([0, 1, 2]))
        cfg.enable_hme_level2_flag = int(rng.integers(0, 2))
        cfg.prune_me_candidates_th = int(rng.choice([0, 30, 65]))
    return edit


@pytest.mark.parametrize("seed", range(16))
def test_fuzzed_search_controls_match_oracle(hip_ctx, seed):
    """Random search-control combinations (areas, methods, early exits, tiling) on random content."""
    rng = np.random.default_rng(1000 + seed)
    kind = ["pan", "noise", "fastpan", "pan"][seed % 4]
    refs = [{(0, 0): 0, (1, 0): 4}, {(0, 0): 1, (0, 1): 0, (1, 0): 3}, {(0, 0): 0}, {(0, 0): 1, (1, 0): 3, (1, 1): 4}][seed % 4]
    size = [(352, 288), (640, 360), (360, 296), (704, 576)][(seed // 4) % 4]
    case = MeCase(size[0], size[1], enc_mode=int(rng.choice([2, 6, 9])), cur=2, refs=refs, n_frames=5, seed=seed, kind=kind,
                  temporal_layer_index=int(rng.integers(0, 3)) if (1, 0) not in refs else 1 + int(rng.integers(0, 3)),
                  cfg_edit=_fuzz_cfg(rng), gm_enabled=seed % 2)
    assert not compare(case.run_cpu("oracle"), case.run_hip(hip_ctx))


def test_several_pictures_in_one_launch(hip_ctx, me_form):
    """svt_hip_me_pictures_async: three different pictures (sizes, presets, reference counts, a row band) share one launch;
    each must equal its own single-picture call.  Also svt_hip_context_set_me_timing / svt_hip_me_launch_times: the kernels the launch
    used, each with a positive duration."""
    import ctypes as C
    import torch
    from svt_av1_psyex_amd import abi
    cases = [MeCase(352, 288, enc_mode=6), MeCase(640, 360, enc_mode=3, seed=3),
             MeCase(352, 288, enc_mode=0, cur=2, refs={(0, 0): 1, (0, 1): 0, (1, 0): 3, (1, 1): 4}, n_frames=5)]
    cases[1].desc.b64_row_start, cases[1].desc.b64_row_count = 2, 3
    jobs, keep, expect = [], [], []
    for c in cases:
        cur = hip_ctx.upload(c.cur)
        refs = {k: hip_ctx.upload(v) for k, v in c.refs.items()}
        expect.append(hip_ctx.me_picture(c.cfg, c.desc, cur, refs))
        n = abi.n_pu(c.desc.enable_me_16x16, c.desc.enable_me_8x8)
        nb = ((c.desc.aligned_width + 63) // 64) * ((c.desc.aligned_height + 63) // 64)
        res, bufs = abi.MeResults(), {}
        for name, dt, cnt in abi.RESULT_FIELDS:
            bufs[name] = torch.zeros(nb * cnt(n, c.desc.max_refs, c.desc.max_cand) * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda")
            setattr(res, name, bufs[name].data_ptr())
        jobs.append((c.cfg, c.desc, cur, refs, res))
        keep.append((bufs, nb, n))
    torch.cuda.synchronize()
    hip_ctx.set_me_timing(True)
    try:
        hip_ctx.me_pictures_async(jobs)
        times = hip_ctx.me_launch_times()
    finally:
        hip_ctx.set_me_timing(False)
    hip_ctx.sync()
    chain = {"svt_hip_me_dense_kernel", "svt_hip_me_b64_kernel"} | ({f"svt_hip_me_{k}_kernel" for k in ("mid1", "s1", "mid2", "s2", "tail")} if me_form == 2 else set())
    assert set(times) == chain, times
    assert all(0 < t < 50 for t in times.values()), times
    for c, (bufs, nb, n), exp in zip(cases, keep, expect):
        got = fill_unsearched(c.desc, {name: bufs[name].cpu().numpy().view(dt).reshape(nb, -1) for name, dt, _ in abi.RESULT_FIELDS})
        w64 = (c.desc.aligned_width + 63) // 64
        r0 = c.desc.b64_row_start
        r1 = r0 + (c.desc.b64_row_count or ((c.desc.aligned_height + 63) // 64 - r0))
        rows = slice(r0 * w64, r1 * w64)  # the synchronous entry returns only the band's rows as well
        bad = [k for k in exp if not np.array_equal(np.asarray(exp[k]).reshape(nb, -1)[rows], got[k][rows])]
        assert not bad, bad
    # more than 16 pictures in one call is refused
    assert api_rc_too_many(hip_ctx, jobs[0]) == 2


def api_rc_too_many(ctx, job):
    import ctypes as C
    from svt_av1_psyex_amd import abi, api
    arr = (abi.MeJob * 17)()
    return api.lib().svt_hip_me_pictures_async(ctx._h, 17, arr)


@pytest.mark.parametrize("kw", MCTF_GRID, ids=lambda k: f"{k['width']}x{k['height']}_m{k['enc_mode']}_th{k['mctf_exit_th']}")
def test_mctf_matches_oracle(hip_ctx, kw):
    case = MeCase(**kw)
    assert not compare(case.run_cpu("oracle"), case.run_hip(hip_ctx), MCTF_OUTPUTS)


def test_search_centre_outside_the_padded_plane(hip_ctx):
    """me_cases.probe_outside_case: the reference's 1-point probe would read far below its padded plane (undefined there); the
    oracle and the kernel both take the plane's nearest edge instead, so they agree, run after run."""
    from me_cases import probe_outside_case
    case = probe_outside_case()
    want = case.run_cpu("oracle")
    sc = want["hme_sc"].reshape(-1, 2, 4, 2)
    assert tuple(sc[22, 1, 0]) == (16, 184)  # b64 (2, 2): rows 128..143 of a 144-row picture, centre 184 rows further down
    for _ in range(3):
        assert not compare(want, case.run_hip(hip_ctx))


@pytest.mark.parametrize("w,h,enc_mode", [(64, 64, 6), (72, 80, 2), (128, 64, 11), (64, 200, 6)])
def test_smallest_pictures(hip_ctx, w, h, enc_mode):
    """One b64 (or one row / column of them), partial blocks: windows of every HME level hang over all four picture edges."""
    case = MeCase(w, h, enc_mode=enc_mode, refs={(0, 0): 0, (1, 0): 3, (0, 1): 1}, seed=w + h + enc_mode)
    assert not compare(case.run_cpu("oracle"), case.run_hip(hip_ctx))


def test_8k_picture_matches_oracle(hip_ctx):
    """The largest resolution class (input_resolution 6): 120 x 68 = 8,100 b64, the widest search areas of the preset tables."""
    case = MeCase(7680, 4320, enc_mode=6, refs={(0, 0): 0, (1, 0): 2}, cur=1, n_frames=3, seed=3, temporal_layer_index=2)
    assert case.desc.input_resolution == 6
    assert not compare(case.run_cpu("oracle"), case.run_hip(hip_ctx))


# ---- the dense pre-pass (pre-HME strips + HME level-0 regions ahead of the per-block kernel: csrc/me_dense.inl) ----

DENSE_CASES = [
    dict(width=640, height=360, enc_mode=6, seed=21),                                         # bottom row 40 samples high: 5 source rows
    dict(width=1280, height=720, enc_mode=6, refs={(0, 0): 0, (1, 0): 10}, cur=5, seed=22),    # distance 5: large strips, several segments
    dict(width=1280, height=720, enc_mode=2, refs={(0, 0): 3, (0, 1): 1, (1, 0): 5, (1, 1): 7}, cur=4, seed=23, temporal_layer_index=2),
    dict(width=360, height=296, enc_mode=4, seed=24, kind="noise"),                             # right column 40 samples wide: left to the per-block kernel
    dict(width=1920, height=1080, enc_mode=8, seed=25),
    dict(width=704, height=576, enc_mode=0, refs={(0, 0): 0, (1, 0): 9}, cur=8, seed=26, kind="fastpan"),
]


@pytest.mark.parametrize("kw", DENSE_CASES, ids=lambda k: f"{k['width']}x{k['height']}_m{k['enc_mode']}")
def test_dense_prepass_is_used_and_changes_nothing(hip_ctx, me_form, kw):
    """With the pre-pass on, the per-block kernel takes pre-HME / level-0 results from it (counter > 0) and every output -- the
    search-level arrays included -- equals the run without it and the oracle."""
    case = MeCase(**kw)
    want = case.run_cpu("oracle")
    try:
        hip_ctx.set_me_counting(True)
        hip_ctx.set_me_dense(False)
        hip_ctx.me_dense_counters()
        off = case.run_hip(hip_ctx)
        assert hip_ctx.me_dense_counters() == (0, 0)
        hip_ctx.set_me_dense(True)
        hip_ctx.set_me_staged(0)  # pre-pass + the one-kernel pipeline
        on = case.run_hip(hip_ctx)
        taken, own = hip_ctx.me_dense_counters()
        hip_ctx.set_me_staged(2)   # pre-pass + the chain of small kernels (+ the one-kernel form for deferred blocks)
        staged = case.run_hip(hip_ctx)
        taken_s, own_s = hip_ctx.me_dense_counters()
    finally:
        hip_ctx.set_me_dense(True)
        hip_ctx.set_me_staged(me_form)
        hip_ctx.set_me_counting(False)
    assert not compare(want, off)
    assert not compare(want, on)
    assert not compare(want, staged)
    assert taken > 0 and taken > 4 * own, (taken, own)
    assert taken_s >= taken, (taken_s, taken)  # what the one-kernel form took from the pre-pass the staged form takes too (a deferred block takes its slots twice)


def test_dense_prepass_skip_search_line_and_partial_octets(hip_ctx):
    """prehme_skip_search_line (odd search rows only) and search widths that are not multiples of 8 after clipping."""
    def edit(cfg):
        cfg.prehme_skip_search_line = 1
        cfg.prehme_sa_cfg[1].sa_min.width = 44
        cfg.prehme_sa_cfg[1].sa_max.width = 100
        cfg.prehme_sa_cfg[0].sa_min.height = 37
        cfg.me_early_exit_th = 0
    case = MeCase(416, 240, enc_mode=5, cfg_edit=edit, seed=31, kind="noise")
    want = case.run_cpu("oracle")
    hip_ctx.set_me_counting(True)
    try:
        hip_ctx.me_dense_counters()
        got = case.run_hip(hip_ctx)
        taken, own = hip_ctx.me_dense_counters()
    finally:
        hip_ctx.set_me_counting(False)
    assert not compare(want, got)
    assert taken > 0

