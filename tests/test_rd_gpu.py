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
