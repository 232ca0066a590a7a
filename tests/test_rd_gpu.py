"""GPU: svt_hip_rd_batch (residual -> fwd txfm -> quant -> dist -> inv txfm -> SSE) against the oracle chain,
for all 19 transform sizes x the transform types AV1 allows per size, 8- and 10-bit, both quantizers."""
import numpy as np
import pyoracle
import pytest

from svt_av1_psyex_amd import abi, rd
from txfm_cases import valid_types

pytestmark = pytest.mark.gpu


def _planes(rng, bd, pattern, height=128, width=192):
    dt = np.uint8 if bd == 8 else np.uint16
    hi = (1 << bd) - 1
    if pattern == "smooth":
        base = rng.integers(0, hi + 1, (height // 8 + 2, width // 8 + 2)).astype(np.float64)
        up = np.kron(base, np.ones((8, 8)))[:height, :width]
        src = np.clip(up + rng.normal(0, 6 * (1 << (bd - 8)), (height, width)), 0, hi).astype(dt)
        pred = np.clip(up + rng.normal(0, 3 * (1 << (bd - 8)), (height, width)), 0, hi).astype(dt)
    elif pattern == "extreme":
        src = np.where(rng.integers(0, 2, (height, width)) == 0, 0, hi).astype(dt)
        pred = (hi - src).astype(dt)
    else:
        src = rng.integers(0, hi + 1, (height, width)).astype(dt)
        pred = rng.integers(0, hi + 1, (height, width)).astype(dt)
    return np.ascontiguousarray(src), np.ascontiguousarray(pred)


def _check(a, b, what):
    for k in a:
        assert np.array_equal(a[k], b[k]), (what, k, np.argwhere(a[k] != b[k])[:3].tolist())


@pytest.mark.parametrize("tx_size", range(19))
def test_rd_batch_matches_oracle(hip_ctx, tx_size):
    rng = np.random.default_rng(50 + tx_size)
    rows = np.stack([rd.quant_row_from_step(8, 10), rd.quant_row_from_step(60, 75), rd.quant_row_from_step(500, 640)])
    for bd in (8, 10):
        for pattern in ("smooth", "random", "extreme"):
            src, pred = _planes(rng, bd, pattern)
            for quant_kind in (0, 1):
                jobs = rd.grid_jobs(192, 128, 192, tx_size)
                types = valid_types(tx_size)
                jobs["tx_type"] = rng.choice(types, len(jobs))
                jobs["quant_row"] = rng.integers(0, 3, len(jobs))
                jobs["pred_offset"] = jobs["src_offset"][rng.permutation(len(jobs))]  # pred block elsewhere than src block
                f = dict(bit_depth=bd, quant_kind=quant_kind, tx_size=tx_size, src_stride=192, pred_stride=192)
                want = pyoracle.rd_batch(f, src, pred, jobs, rows)
                got = rd.run_hip(hip_ctx, f, src, pred, jobs, rows)
                _check(want, got, (tx_size, bd, pattern, quant_kind))


def test_every_transform_type_individually(hip_ctx):
    """Each (size, type) pair on its own, so a failure names the pair."""
    rng = np.random.default_rng(9)
    rows = np.stack([rd.quant_row_from_step(20, 24)])
    src, pred = _planes(rng, 10, "smooth")
    for ts in range(19):
        for tt in valid_types(ts):
            jobs = rd.grid_jobs(192, 128, 192, ts, tx_type=tt)[:6]
            f = dict(bit_depth=10, quant_kind=0, tx_size=ts, src_stride=192, pred_stride=192)
            _check(pyoracle.rd_batch(f, src, pred, jobs, rows), rd.run_hip(hip_ctx, f, src, pred, jobs, rows), (ts, tt))


def test_rd_batch_rejects_bad_descriptors(hip_ctx):
    import ctypes as C
    from svt_av1_psyex_amd import api
    d = abi.RdBatchDesc(n_jobs=4, bit_depth=12, tx_size=2)
    assert api.lib().svt_hip_rd_batch(hip_ctx._h, C.byref(d)) == 2
    d = abi.RdBatchDesc(n_jobs=4, bit_depth=8, tx_size=40)
    assert api.lib().svt_hip_rd_batch(hip_ctx._h, C.byref(d)) == 2


@pytest.mark.parametrize("tx_size", [0, 1, 2, 3, 4, 6, 9, 12, 13, 17, 18])
def test_partial_frequency_shapes_and_quantization_matrices(hip_ctx, tx_size):
    """pf_shape (N2 / N4 / ONLY_DC) per job and a quantization matrix per batch, both quantizers, 8 / 10 bit."""
    rng = np.random.default_rng(300 + tx_size)
    rows = np.stack([rd.quant_row_from_step(8, 10), rd.quant_row_from_step(60, 75)])
    npk = min(abi.TX_W[tx_size], 32) * min(abi.TX_H[tx_size], 32)
    for bd in (8, 10):
        src, pred = _planes(rng, bd, "smooth")
        for quant_kind in (0, 1):
            for use_qm in (False, True):
                jobs = rd.grid_jobs(192, 128, 192, tx_size)
                jobs["tx_type"] = rng.choice(valid_types(tx_size), len(jobs))
                jobs["quant_row"] = rng.integers(0, 2, len(jobs))
                jobs["pf_shape"] = rng.integers(0, 4, len(jobs))
                qm = rng.integers(16, 256, npk).astype(np.uint8) if use_qm else None   # AV1 matrices: 32 = unit weight
                iqm = rng.integers(16, 256, npk).astype(np.uint8) if use_qm else None
                f = dict(bit_depth=bd, quant_kind=quant_kind, tx_size=tx_size, src_stride=192, pred_stride=192)
                want = pyoracle.rd_batch(f, src, pred, jobs, rows, qmatrix=qm, iqmatrix=iqm)
                got = rd.run_hip(hip_ctx, f, src, pred, jobs, rows, qmatrix=qm, iqmatrix=iqm)
                _check(want, got, (tx_size, bd, quant_kind, use_qm))


@pytest.mark.parametrize("tx_size", [1, 3, 4, 10, 17])
def test_out_of_range_16bit_samples_stay_exact(hip_ctx, tx_size):
    """uint16 planes holding values beyond 10 bits: the forward transform's 24-bit-multiply fast path must not be taken
    (residuals up to +-32767 overflow 24 bits inside the 64-point passes); results still equal the oracle's wrapping arithmetic."""
    rng = np.random.default_rng(900 + tx_size)
    rows = np.stack([rd.quant_row_from_step(60, 75)])
    src = rng.integers(0, 65536, (128, 192)).astype(np.uint16)
    pred = rng.integers(0, 65536, (128, 192)).astype(np.uint16)
    src[:64] = np.clip(src[:64], 0, 1023); pred[:64] = np.clip(pred[:64], 0, 1023)  # valid and invalid blocks in one batch
    jobs = rd.grid_jobs(192, 128, 192, tx_size)
    jobs["tx_type"] = rng.choice(valid_types(tx_size), len(jobs))
    f = dict(bit_depth=10, quant_kind=0, tx_size=tx_size, src_stride=192, pred_stride=192)
    _check(pyoracle.rd_batch(f, src, pred, jobs, rows), rd.run_hip(hip_ctx, f, src, pred, jobs, rows), tx_size)


# ---- inverse transform alone: the 19 pointer-level entries and the batched entry ------------------------------------------------
def _hip_inv_leaf(L, ts, tt, co, pred, stride, out_stride, bd):
    import ctypes as C
    from txfm_cases import TX_H, TX_W
    w, h = TX_W[ts], TX_H[ts]
    out = np.full((h, out_stride), 0x5555, np.uint16)
    a = [co.ctypes.data_as(C.c_void_p), pred.ctypes.data_as(C.c_void_p), C.c_int32(stride), out.ctypes.data_as(C.c_void_p), C.c_int32(out_stride), C.c_uint8(tt)]
    if w == h:
        a += [C.c_int32(bd)]
    elif (w, h) in ((4, 8), (8, 4), (4, 16), (16, 4)):
        a += [C.c_uint8(ts), C.c_int32(bd)]
    else:
        a += [C.c_uint8(ts), C.c_int32(w * h), C.c_int32(bd)]
    getattr(L, f"svt_av1_inv_txfm2d_add_{w}x{h}_hip")(*a)
    return out


@pytest.mark.parametrize("tx_size", range(19))
def test_leaf_inverse_transforms(hip_ctx, oracle, tx_size):
    """svt_av1_inv_txfm2d_add_{W}x{H}_hip: every allowed type, 8 / 10 bit, random and extreme coefficients, separate read / write planes."""
    import ctypes as C
    from svt_av1_psyex_amd import api
    from txfm_cases import TX_H, TX_W, valid_types
    L = api.lib()
    assert L.svt_hip_leaf_bind(hip_ctx._h) == 0
    try:
        rng = np.random.default_rng(300 + tx_size)
        w, h = TX_W[tx_size], TX_H[tx_size]
        n = min(w, 32) * min(h, 32)
        for tt in valid_types(tx_size):
            for bd, pat in [(8, "random"), (10, "random"), (10, "big"), (8, "sparse")]:
                stride, ostride = w + int(rng.integers(0, 5)), w + int(rng.integers(0, 3))
                co = {"random": rng.integers(-(1 << (bd + 7)), 1 << (bd + 7), n), "big": rng.choice([-(1 << 17), (1 << 17) - 1, 0], n),
                      "sparse": np.where(rng.random(n) < 0.05, rng.integers(-3000, 3000, n), 0)}[pat].astype(np.int32)
                pred = rng.integers(0, 1 << bd, (h, stride)).astype(np.uint16)
                want = np.full((h, ostride), 0x5555, np.uint16)
                oracle.orc_inv_txfm2d_add(co.ctypes.data_as(C.c_void_p), pred.ctypes.data_as(C.c_void_p), C.c_int32(stride), want.ctypes.data_as(C.c_void_p),
                                          C.c_int32(ostride), tt, tx_size, bd)
                got = _hip_inv_leaf(L, tx_size, tt, co, pred, stride, ostride, bd)
                assert np.array_equal(got, want), (tt, bd, pat)
    finally:
        L.svt_hip_leaf_bind(None)


@pytest.mark.parametrize("tx_size,bd,sample_bytes", [(2, 10, 2), (3, 8, 1), (4, 10, 2), (9, 8, 2), (12, 10, 2), (0, 8, 1)])
def test_inverse_batch_matches_oracle(hip_ctx, oracle, tx_size, bd, sample_bytes):
    """svt_hip_inv_txfm_batch: a plane's worth of blocks in one launch, read plane != write plane, uint8 and uint16 storage."""
    import ctypes as C
    import torch
    from svt_av1_psyex_amd import abi, api, rd
    from txfm_cases import TX_H, TX_W, valid_types
    rng = np.random.default_rng(900 + tx_size)
    w, h = TX_W[tx_size], TX_H[tx_size]
    n = min(w, 32) * min(h, 32)
    PW, PH = 192, 128
    jobs = rd.grid_jobs(PW, PH, PW, tx_size)
    types = valid_types(tx_size)
    jobs["tx_type"] = rng.choice(types, len(jobs))
    dt = np.uint8 if sample_bytes == 1 else np.uint16
    pred = rng.integers(0, 1 << bd, (PH, PW)).astype(dt)
    co = np.where(rng.random((len(jobs), n)) < 0.3, rng.integers(-(1 << (bd + 6)), 1 << (bd + 6), (len(jobs), n)), 0).astype(np.int32)
    want = np.zeros((PH, PW), np.uint16)
    p16 = pred.astype(np.uint16)
    for j, jb in enumerate(jobs):
        off = int(jb["pred_offset"])
        oracle.orc_inv_txfm2d_add(C.c_void_p(co[j].ctypes.data), C.c_void_p(p16.ctypes.data + 2 * off), C.c_int32(PW), C.c_void_p(want.ctypes.data + 2 * off),
                                  C.c_int32(PW), int(jb["tx_type"]), tx_size, bd)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    t_pred, t_co, t_jobs = dev(pred), dev(co), dev(jobs)
    t_rec = torch.zeros(PH * PW * sample_bytes, dtype=torch.uint8, device="cuda")
    d = abi.InvTxBatchDesc(bit_depth=bd, sample_bytes=sample_bytes, tx_size=tx_size, n_jobs=len(jobs), pred_stride=PW, recon_stride=PW, pred=t_pred.data_ptr(),
                           recon=t_rec.data_ptr(), jobs=t_jobs.data_ptr(), dqcoeff=t_co.data_ptr())
    torch.cuda.synchronize()
    hip_ctx.check(api.lib().svt_hip_inv_txfm_batch(hip_ctx._h, C.byref(d)), "svt_hip_inv_txfm_batch")
    hip_ctx.sync()
    got = t_rec.cpu().numpy().view(dt).reshape(PH, PW)
    assert np.array_equal(got.astype(np.uint16), want)
    d.sample_bytes = 1 if bd == 10 else 3
    assert api.lib().svt_hip_inv_txfm_batch(hip_ctx._h, C.byref(d)) == 2


@pytest.mark.parametrize("tx_size", range(19))
def test_leaf_forward_transforms(hip_ctx, oracle, tx_size):
    """svt_av1_fwd_txfm2d_{W}x{H}{,_N2,_N4}_hip: the full W x H array (64-point sizes included), every allowed type, the
    reference's residual patterns (FwdTxfm2dAsmTest.cc: random / extremes)."""
    import ctypes as C
    from svt_av1_psyex_amd import api
    from txfm_cases import TX_H, TX_W, residual_block, valid_types
    L = api.lib()
    assert L.svt_hip_leaf_bind(hip_ctx._h) == 0
    try:
        rng = np.random.default_rng(700 + tx_size)
        w, h = TX_W[tx_size], TX_H[tx_size]
        for tt in valid_types(tx_size):
            for bd, pat in [(8, "random"), (10, "max"), (10, "checker"), (10, "laplace")]:
                stride = w + int(rng.integers(0, 5))
                r = residual_block(rng, w, h, stride, bd, pat)
                full = np.zeros(w * h, np.int32)
                oracle.orc_fwd_txfm2d(r.ctypes.data_as(C.c_void_p), full.ctypes.data_as(C.c_void_p), C.c_uint32(stride), tt, tx_size)
                for suf, sh in (("", 0), ("_N2", 1), ("_N4", 2)):
                    got = np.full(w * h, 12345, np.int32)
                    getattr(L, f"svt_av1_fwd_txfm2d_{w}x{h}{suf}_hip")(r.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_uint32(stride), C.c_uint8(tt), C.c_uint8(bd))
                    want = full.reshape(h, w).copy()
                    want[h >> sh:, :] = 0
                    want[:, w >> sh:] = 0
                    assert np.array_equal(got, want.reshape(-1)), (tt, bd, pat, suf)
    finally:
        L.svt_hip_leaf_bind(None)


def test_leaf_handle_transform(hip_ctx):
    """svt_handle_transform{16x64,32x64,64x16,64x32,64x64}{,_N2_N4}_hip (transforms.c:2374-2543): the energy of the discarded
    frequencies, the kept rows packed in place, the rest of the array untouched."""
    import ctypes as C
    from svt_av1_psyex_amd import api
    L = api.lib()
    assert L.svt_hip_leaf_bind(hip_ctx._h) == 0
    try:
        rng = np.random.default_rng(55)
        for (w, h) in [(16, 64), (32, 64), (64, 16), (64, 32), (64, 64)]:
            for suf in ("", "_N2_N4"):
                fn = getattr(L, f"svt_handle_transform{w}x{h}{suf}_hip")
                fn.restype = C.c_uint64
                a = rng.integers(-(1 << 22), 1 << 22, w * h).astype(np.int32)
                want = a.copy()
                m = a.reshape(h, w).astype(np.int64)
                wp, hp = min(w, 32), min(h, 32)
                energy = int((m ** 2).sum() - (m[:hp, :wp] ** 2).sum()) if suf == "" else 0
                if w == 64:
                    want[:wp * hp] = m[:hp, :wp].reshape(-1)
                got = a.copy()
                e = fn(got.ctypes.data_as(C.c_void_p))
                assert e == energy and np.array_equal(got, want), (w, h, suf)
    finally:
        L.svt_hip_leaf_bind(None)
