"""GPU: svt_hip_fullpel_pred -- every 16x16 PU of every b64 is the reference plane displaced by the PU's full-pel MV,
coordinates clamped into the picture (edge replication)."""
import ctypes as C

import numpy as np
import pytest

from svt_av1_psyex_amd import api

pytestmark = pytest.mark.gpu


def _expected(ref, mv, list_, ref_idx, row0, nrow):
    h, w = ref.shape
    w64 = (w + 63) // 64
    out = np.zeros_like(ref)
    for by16 in range(row0 * 4, min((row0 + nrow) * 4, (h + 15) // 16)):
        for bx16 in range((w + 15) // 16):
            x0, y0 = bx16 * 16, by16 * 16
            b = (x0 >> 6) + (y0 >> 6) * w64
            qx, qy = (x0 >> 4) & 3, (y0 >> 4) & 3
            z = (qx & 1) | ((qy & 1) << 1) | ((qx >> 1) << 2) | ((qy >> 1) << 3)
            m = int(mv[b, list_ * 4 + ref_idx, 5 + z])
            mvx, mvy = ((m & 0xFFFF) ^ 0x8000) - 0x8000, (((m >> 16) & 0xFFFF) ^ 0x8000) - 0x8000
            ys = np.clip(np.arange(y0, min(y0 + 16, h)) + mvy, 0, h - 1)
            xs = np.clip(np.arange(x0, min(x0 + 16, w)) + mvx, 0, w - 1)
            out[y0:y0 + len(ys), x0:x0 + len(xs)] = ref[np.ix_(ys, xs)]
    return out


@pytest.mark.parametrize("bd,w,h,stride_pad", [(10, 640, 360, 0), (8, 640, 360, 0), (10, 200, 136, 0), (8, 352, 288, 3), (10, 1920, 1080, 0)])
def test_fullpel_pred(hip_ctx, bd, w, h, stride_pad):
    import torch
    rng = np.random.default_rng(w + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    ref = rng.integers(0, 1 << bd, (h, w)).astype(dt)
    nb = ((w + 63) // 64) * ((h + 63) // 64)
    mv = np.zeros((nb, 8, 85), np.uint32)
    mvx = rng.integers(-70, 71, (nb, 8, 85)); mvy = rng.integers(-70, 71, (nb, 8, 85))
    mvx[::7] = rng.integers(-2000, 2000, mvx[::7].shape)  # far outside the picture: pure edge replication
    mv[:] = ((mvy.astype(np.int64) & 0xFFFF) << 16 | (mvx.astype(np.int64) & 0xFFFF)).astype(np.uint32)
    ps = w + stride_pad  # an odd pitch disables the vector stores
    t_ref = torch.from_numpy(ref.view(np.uint8).reshape(-1).copy()).cuda()
    t_mv = torch.from_numpy(mv.view(np.uint8).reshape(-1).copy()).cuda()
    t_pred = torch.zeros(h * ps * ref.itemsize, dtype=torch.uint8, device="cuda")
    h64 = (h + 63) // 64
    for (lst, ri, row0, nrow) in [(0, 0, 0, 0), (1, 2, 1, max(1, h64 - 2))]:
        t_pred.zero_()
        torch.cuda.synchronize()
        rc = api.lib().svt_hip_fullpel_pred(hip_ctx._h, C.c_void_p(t_ref.data_ptr()), w, w, h, bd, C.c_void_p(t_mv.data_ptr()), lst, ri, row0, nrow,
                                            C.c_void_p(t_pred.data_ptr()), ps)
        assert rc == 0
        hip_ctx.sync()
        got = t_pred.cpu().numpy().view(dt).reshape(h, ps)[:, :w]
        rows = nrow if nrow else h64 - row0
        want = _expected(ref, mv, lst, ri, row0, rows)
        y0, y1 = row0 * 64, min((row0 + rows) * 64, h)
        assert np.array_equal(got[y0:y1], want[y0:y1])
        assert not got[:y0].any() and not got[y1:].any()  # rows outside the band are untouched


def test_fullpel_pred_rejects_bad_arguments(hip_ctx):
    assert api.lib().svt_hip_fullpel_pred(hip_ctx._h, None, 64, 64, 64, 10, None, 0, 0, 0, 0, None, 64) == 2


@pytest.mark.parametrize("bd", [8, 10])
def test_fullpel_pred_batch_equals_single_launches(hip_ctx, bd):
    """svt_hip_fullpel_pred_batch: several pictures (own reference, MVs, prediction plane, row band) in one launch."""
    import torch
    from svt_av1_psyex_amd import abi
    w, h = 712, 400
    h64, nb = (h + 63) // 64, ((w + 63) // 64) * ((h + 63) // 64)
    dt = np.uint8 if bd == 8 else np.uint16
    rng = np.random.default_rng(5 + bd)
    bands = [(0, 0), (2, 3), (6, 1), (1, 4), (0, 7)]  # (row_start, row_count): whole picture, inner bands, the last (partial) row
    refs, mvs, preds, keep = [], [], [], []
    jobs = (abi.PredJob * len(bands))()
    for i, (r0, nr) in enumerate(bands):
        ref = rng.integers(0, 1 << bd, (h, w)).astype(dt)
        mv = ((rng.integers(-40, 41, (nb, 8, 85)).astype(np.int64) & 0xFFFF) << 16 | (rng.integers(-90, 91, (nb, 8, 85)).astype(np.int64) & 0xFFFF)).astype(np.uint32)
        t_ref = torch.from_numpy(ref.view(np.uint8).reshape(-1).copy()).cuda()
        t_mv = torch.from_numpy(mv.view(np.uint8).reshape(-1).copy()).cuda()
        t_pred = torch.zeros(h * w * ref.itemsize, dtype=torch.uint8, device="cuda")
        keep += [t_ref, t_mv, t_pred]
        refs.append(ref); mvs.append(mv); preds.append(t_pred)
        jobs[i].ref, jobs[i].sb_best_mv, jobs[i].pred = t_ref.data_ptr(), t_mv.data_ptr(), t_pred.data_ptr()
        jobs[i].b64_row_start, jobs[i].b64_row_count, jobs[i].list, jobs[i].ref_idx = r0, nr, i & 1, i % 4
    torch.cuda.synchronize()
    hip_ctx.check(api.lib().svt_hip_fullpel_pred_batch(hip_ctx._h, w, w, h, bd, w, len(bands), jobs), "svt_hip_fullpel_pred_batch")
    hip_ctx.sync()
    for i, (r0, nr) in enumerate(bands):
        rows = nr if nr else h64 - r0
        got = preds[i].cpu().numpy().view(dt).reshape(h, w)
        want = _expected(refs[i], mvs[i], i & 1, i % 4, r0, rows)
        y0, y1 = r0 * 64, min((r0 + rows) * 64, h)
        assert np.array_equal(got[y0:y1], want[y0:y1]), i
        assert not got[:y0].any() and not got[y1:].any(), i
    assert api.lib().svt_hip_fullpel_pred_batch(hip_ctx._h, w, w, h, bd, w, 17, jobs) == 2  # more than SVT_HIP_PRED_MAX_JOBS
    jobs[1].b64_row_start = h64
    assert api.lib().svt_hip_fullpel_pred_batch(hip_ctx._h, w, w, h, bd, w, len(bands), jobs) == 2
