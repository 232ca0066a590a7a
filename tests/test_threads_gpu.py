"""GPU: one SvtHipContext shared by several host threads, as the reference shares its kernels between its ME threads with
several pictures in flight (Globals/enc_handle.c:2265, Codec/me_process.c:140-172).  Every thread's results must equal the
results of the same call made alone."""
import threading

import numpy as np
import pytest

from dg_cases import DgCase
from me_cases import MeCase, compare, fill_unsearched

pytestmark = pytest.mark.gpu

CASES = [
    dict(width=352, height=288, enc_mode=6),
    dict(width=640, height=360, enc_mode=4, seed=3, kind="fastpan"),
    dict(width=352, height=288, enc_mode=12, seed=5, kind="noise"),
    dict(width=360, height=296, enc_mode=8, seed=7),
    dict(width=640, height=480, enc_mode=2, cur=2, refs={(0, 0): 1, (0, 1): 0, (1, 0): 3, (1, 1): 4}, n_frames=5),
    dict(width=1280, height=720, enc_mode=6, seed=11),
]


def test_me_picture_from_six_threads(hip_ctx):
    """svt_hip_me_picture (host pointers, synchronous) from 6 threads x 4 calls on ONE context: different pictures, sizes and
    presets in flight together; each thread checks every call against the single-threaded result, and the first case against
    the oracle."""
    cases = [MeCase(**kw) for kw in CASES]
    dev = []
    for c in cases:
        dev.append((hip_ctx.upload(c.cur), {k: hip_ctx.upload(v) for k, v in c.refs.items()}))
    alone = [hip_ctx.me_picture(c.cfg, c.desc, cur, refs) for c, (cur, refs) in zip(cases, dev)]
    assert not compare(cases[0].run_cpu("oracle"), alone[0])
    errors = []
    start = threading.Barrier(len(cases))

    def worker(i):
        try:
            c, (cur, refs) = cases[i], dev[i]
            start.wait()
            for rep in range(4):
                got = hip_ctx.me_picture(c.cfg, c.desc, cur, refs)
                bad = compare(alone[i], got)
                if bad:
                    errors.append((i, rep, bad[:2]))
        except Exception as e:  # noqa: BLE001
            errors.append((i, "exception", repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for cur, refs in dev:
        cur.free()
        for r in refs.values():
            r.free()
    assert not errors, errors


def test_mixed_entries_from_threads(hip_ctx):
    """The synchronous dynamic-GOP detector entry and svt_hip_me_picture interleaved from 4 threads, with pictures uploaded by the
    threads themselves (svt_hip_pa_picture_create enqueues on the context stream; the borrowed lanes wait for it)."""
    import pyoracle
    me = MeCase(640, 360, enc_mode=6, seed=9)
    dg = DgCase(640, 480, "fast")
    want_me = me.run_cpu("oracle")
    want_dg = pyoracle.dg_detector("oracle", dg.src, dg.ref, *dg.args())
    errors = []

    def me_worker():
        try:
            for _ in range(3):
                bad = compare(want_me, me.run_hip(hip_ctx))  # uploads + ME + frees inside
                if bad:
                    errors.append(("me", bad[:2]))
        except Exception as e:  # noqa: BLE001
            errors.append(("me", repr(e)))

    def dg_worker():
        try:
            for _ in range(3):
                s, r = hip_ctx.upload(dg.src), hip_ctx.upload(dg.ref)
                got = hip_ctx.dg_detector_hme_level0(s, r, *dg.args())
                s.free(); r.free()
                for k in want_dg:
                    if not np.array_equal(want_dg[k], got[k]):
                        errors.append(("dg", k))
        except Exception as e:  # noqa: BLE001
            errors.append(("dg", repr(e)))

    threads = [threading.Thread(target=f) for f in (me_worker, dg_worker, me_worker, dg_worker)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_async_launch_ahead(hip_ctx):
    """More asynchronous ME launches enqueued back to back than the parameter ring holds (SVT_HIP_PARAM_RING = 4), each with
    different descriptors: every launch must have read ITS parameter block."""
    import torch
    from svt_av1_psyex_amd import abi
    cases = [MeCase(352, 288, enc_mode=m, seed=s) for m, s in ((6, 1), (8, 2), (4, 3), (10, 4), (12, 5), (2, 6), (6, 7), (9, 8))]
    alone, dev, bufs = [], [], []
    for c in cases:
        cur, refs = hip_ctx.upload(c.cur), {k: hip_ctx.upload(v) for k, v in c.refs.items()}
        dev.append((cur, refs))
        alone.append(hip_ctx.me_picture(c.cfg, c.desc, cur, refs))
    for c in cases:
        n = abi.n_pu(c.desc.enable_me_16x16, c.desc.enable_me_8x8)
        nb = ((c.width + 63) // 64) * ((c.height + 63) // 64)
        res, keep = abi.MeResults(), {}
        for name, dt, cnt in abi.RESULT_FIELDS:
            t = torch.zeros(nb * cnt(n, c.desc.max_refs, c.desc.max_cand) * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda")
            keep[name] = (t, dt)
            setattr(res, name, t.data_ptr())
        bufs.append((res, keep))
    torch.cuda.synchronize()
    for c, (cur, refs), (res, _) in zip(cases, dev, bufs):
        hip_ctx.me_picture_async(c.cfg, c.desc, cur, refs, res)
    hip_ctx.sync()
    for i, (_, keep) in enumerate(bufs):
        got = fill_unsearched(cases[i].desc, {name: t.cpu().numpy().view(dt).reshape(alone[i][name].shape) for name, (t, dt) in keep.items()})
        assert not compare(alone[i], got), i
    for cur, refs in dev:
        cur.free()
        for r in refs.values():
            r.free()
