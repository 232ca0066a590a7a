"""GPU: parity at the sizes BASELINE.json's configs name -- whole 1080p / 2160p pictures through every batched entry, checked
against the oracle (which finishes them in seconds):
  config 2: 1920x1080  8-bit preset 8   -> ME of the whole picture + SAD / SSE / variance / Hadamard batch tiling the picture
  config 3: 1920x1080 10-bit preset 6   -> RD chain (fwd / quantize / block error / inverse / SSE) on the whole picture, incl. a tx-type mix
  config 4: 3840x2160 10-bit preset 6   -> the bench's own RD job lists (64x64, 32x32, 16x16) and its 16-picture ME launch
  config 5: 3840x2160 10-bit preset 2   -> ME with the preset-2 search areas + 10-bit psy-RD / facade batch over the whole picture
These are the launches bench.py times; test_rd_gpu.py / test_stats_gpu.py cover the grids of shapes and types on small planes."""
import numpy as np
import pytest

import pyoracle
from me_cases import MeCase, compare
from svt_av1_psyex_amd import abi, rd, stats, synth

pytestmark = pytest.mark.gpu

_planes = {}


def planes10(w, h, seed):
    """10-bit frames 0..3 of the SURVEY 8d pan sequence (cached)."""
    if (w, h, seed) not in _planes:
        _planes[(w, h, seed)] = synth.synth_sequence(w, h, 4, seed)
    return _planes[(w, h, seed)]


def check_rd(hip_ctx, fields, src, pred, jobs, rows, coeffs, **kw):
    want = pyoracle.rd_batch(fields, src, pred, jobs, rows, want_coeffs=coeffs, **kw)
    got = rd.run_hip(hip_ctx, fields, src, pred, jobs, rows, want_coeffs=coeffs, **kw)
    bad = [k for k in want if not np.array_equal(want[k], got[k])]
    assert not bad, (fields, bad)
    return want


@pytest.mark.parametrize("size,seed", [((1920, 1080), 7), ((3840, 2160), 11)], ids=["1080p", "2160p"])
@pytest.mark.parametrize("tx_size", [4, 3, 2], ids=["64x64", "32x32", "16x16"])
def test_rd_batch_whole_picture(hip_ctx, size, seed, tx_size):
    """bench.py's RD launches: every transform block of a whole 10-bit picture, source = picture 2, prediction = picture 1
    (offsets up to 8.3 M samples), "b" quantizer, all outputs incl. the reconstruction plane."""
    w, h = size
    y = planes10(w, h, seed)
    rows = np.stack([rd.quant_row_from_step(140, 176)])
    jobs = rd.grid_jobs(w, h, w, tx_size)
    f = dict(bit_depth=10, quant_kind=0, tx_size=tx_size, src_stride=w, pred_stride=w)
    want = check_rd(hip_ctx, f, y[2], y[1], jobs, rows, coeffs=(h == 1080))
    assert want["eob"].max() > 0 and want["sse"].sum() > 0  # the comparison is not between two empty results


def test_rd_batch_1080p_tx_type_mix(hip_ctx):
    """Config 3 with the transform-type mix of a preset-6 tx_type_search (product_coding_loop.c:4764-4934): every 16x16 block of
    the picture evaluated with DCT_DCT plus a rotating second type of the size's ext-tx set, two quantizer rows, fp quantizer on
    the 8-bit twin of the picture."""
    w, h = 1920, 1080
    y = planes10(w, h, 7)
    rows = np.stack([rd.quant_row_from_step(60, 75), rd.quant_row_from_step(140, 176)])
    base = rd.grid_jobs(w, h, w, 2)
    mix = base.copy()
    mix["tx_type"] = (1 + np.arange(len(mix)) % 15).astype(np.uint8)  # ADST / flip / identity / 1-D types
    mix["quant_row"] = (np.arange(len(mix)) // 7 % 2).astype(np.uint8)
    mix["pred_offset"] += w * h  # the second evaluation of a block reconstructs into a plane of its own (two copies of the prediction)
    jobs = np.concatenate([base, mix])
    pred = np.concatenate([y[1], y[1]])
    check_rd(hip_ctx, dict(bit_depth=10, quant_kind=0, tx_size=2, src_stride=w, pred_stride=w), y[2], pred, jobs, rows, coeffs=True)
    y8 = synth.to_8bit(y)
    check_rd(hip_ctx, dict(bit_depth=8, quant_kind=1, tx_size=2, src_stride=w, pred_stride=w), y8[2], np.concatenate([y8[1], y8[1]]), jobs, rows, coeffs=False)


def tiling_jobs(w, h, sizes):
    """block jobs tiling a w x h plane with each (bw, bh) of `sizes`; ref block co-located"""
    out = []
    for bw, bh in sizes:
        ys, xs = np.meshgrid(np.arange(0, h - bh + 1, bh), np.arange(0, w - bw + 1, bw), indexing="ij")
        j = np.zeros(ys.size, dtype=abi.BLOCK_JOB_DTYPE)
        j["src_offset"] = j["ref_offset"] = (ys.ravel() * w + xs.ravel()).astype(np.uint32)
        j["width"], j["height"] = bw, bh
        out.append(j)
    return np.concatenate(out)


def test_config2_1080p_preset8_me_and_block_stats(hip_ctx, oracle):
    """BASELINE config 2 at its size: preset-8 ME of the whole 1080p picture (2 references) and the SAD / SSE / variance /
    Hadamard-SATD batch over every 8x8 ... 64x64 block of the picture (hadamard_path_c, enc_mode_config.c:2151-2217)."""
    w, h = 1920, 1080
    case = MeCase(w, h, enc_mode=8, cur=2, refs={(0, 0): 1, (1, 0): 3}, seed=7, n_frames=4, temporal_layer_index=3)
    got = case.run_hip(hip_ctx)
    assert not compare(case.run_cpu("oracle"), got)
    y8 = synth.to_8bit(planes10(w, h, 7))
    jobs = tiling_jobs(w, h, [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (16, 32), (8, 16)])
    want = pyoracle.block_stats(oracle, y8[2], y8[1], jobs, 8, satd=True)
    have = stats.run_hip(hip_ctx, y8[2], y8[1], jobs, 8, satd=True)
    bad = [k for k in want if not np.array_equal(want[k], have[k])]
    assert not bad, bad
    assert want["satd"][:510].min() > 0


@pytest.mark.parametrize("kind,dist", [("pan", 8), ("noise", 2)])
def test_config5_2160p_preset2_me(hip_ctx, kind, dist):
    """BASELINE config 5's search: preset 2 at 3840x2160.  Far references (distance 8: the distance-scaled areas reach their
    preset-2 maxima, 128x128 ME / 192x192 HME level 0 / 8x400 pre-HME, which the LDS arena holds only in tiles) on the pan
    sequence, and i.i.d. noise (no early exit, no search-area reduction: every block runs the full-size searches)."""
    w, h = 3840, 2160
    layer = {2: 3, 8: 1}[dist]
    case = MeCase(w, h, enc_mode=2, cur=8, refs={(0, 0): 8 - dist, (1, 0): 8 + dist}, n_frames=17, seed=11, kind=kind, temporal_layer_index=layer)
    if kind == "noise":  # the oracle needs ~1 s per b64 row of full-size preset-2 searches: a band of rows, on both sides
        case.desc.b64_row_start, case.desc.b64_row_count = 15, 4
    want, got = case.run_cpu_banded("oracle"), case.run_hip(hip_ctx)
    if kind == "noise":
        lo, hi = 15 * 60, 19 * 60
        want, got = {k: v[lo:hi] for k, v in want.items()}, {k: v[lo:hi] for k, v in got.items()}
    assert not compare(want, got)


def test_config5_2160p_psy_rd_and_facades(hip_ctx, oracle):
    """BASELINE config 5's distortions on the whole 10-bit 2160p picture: SSE, highbd variance, the PSYEX psy-RD energy /
    get_svt_psy_full_dist (psy_rd.c:135-293) and the spy-rd facade (picture_operators_c.c:85-174) for every 64x64, 32x32,
    16x16 and 8x8 block, prediction = the previous picture."""
    w, h = 3840, 2160
    y = planes10(w, h, 11)
    jobs = tiling_jobs(w, h, [(64, 64), (32, 32), (16, 16), (8, 8)])
    n = len(jobs)
    rng = np.random.default_rng(5)
    fac = dict(pred_mode=rng.integers(0, 25, n).astype(np.uint8), compound_type=rng.integers(0, 4, n).astype(np.uint8), temporal_layer_index=2, spy_rd=1)
    want = pyoracle.block_stats(oracle, y[2], y[1], jobs, 10, satd=False, psy_rd=1.35, facade=fac)
    have = stats.run_hip(hip_ctx, y[2], y[1], jobs, 10, satd=False, psy_rd=1.35, facade=fac)
    bad = [k for k in want if not np.array_equal(want[k], have[k])]
    assert not bad, bad
    assert want["psy_dist"].max() > 0


def test_config4_bench_launch_of_sixteen_pictures(hip_ctx):
    """bench.py's ME step: 16 preset-6 2160p pictures (4 current pictures x distance 8 / 1 / 4 / 2) in ONE launch of
    svt_hip_me_pictures_async; every picture's MeSbResults against the oracle's whole-picture run."""
    import torch
    w, h = 3840, 2160
    dists, layer, curs = (8, 1, 4, 2), {1: 4, 2: 3, 4: 2, 8: 1}, (8, 9, 10, 11)
    cases = []
    for cur in curs:
        for d in dists:
            cases.append(MeCase(w, h, enc_mode=6, cur=cur, refs={(0, 0): cur - d, (1, 0): cur + d}, n_frames=20, seed=11, temporal_layer_index=layer[d]))
    dev = {}
    def up(pyr):
        if pyr.picture_number not in dev:
            dev[pyr.picture_number] = hip_ctx.upload(pyr)
        return dev[pyr.picture_number]
    jobs, bufs = [], []
    nb = 60 * 34
    for c in cases:
        n = abi.n_pu(c.desc.enable_me_16x16, c.desc.enable_me_8x8)
        res, keep = abi.MeResults(), {}
        for name, dt, cnt in abi.RESULT_FIELDS:
            if name in ("hme_sc", "hme_sad", "do_ref", "sb_best_sad"):
                continue
            t = torch.zeros(nb * cnt(n, c.desc.max_refs, c.desc.max_cand) * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda")
            keep[name] = (t, dt)
            setattr(res, name, t.data_ptr())
        bufs.append((res, keep))
        jobs.append((c.cfg, c.desc, up(c.cur), {k: up(v) for k, v in c.refs.items()}, res))
    torch.cuda.synchronize()
    hip_ctx.me_pictures_async(jobs)
    hip_ctx.sync()
    for c, (_, keep) in zip(cases, bufs):
        want = c.run_cpu_banded("oracle")
        got = {name: t.cpu().numpy().view(dt).reshape(want[name].shape) for name, (t, dt) in keep.items()}
        assert not compare(want, got, names=list(got)), (c.desc.picture_number,)
    for p in dev.values():
        p.free()
