"""Shared definition of the RD-chain fixture (tests/golden/rd_chain.npz): the cases oracle/gen_golden.py runs through the reference's own
`_c` chain (residual -> svt_av1_fwd_txfm2d_* / svt_handle_transform* -> svt_aom_satd -> quantizer -> distortions -> svt_av1_inv_txfm2d_add_*
-> SSE, oracle/ref_harness.c:ref_rd_batch) and that the tests run through the oracle (CPU) and svt_hip_rd_batch (GPU).  A case = one of the
19 transform sizes x bit depth x quantizer ("b" / "fp") on a seeded 192 x 128 plane pair; its jobs tile the planes and cycle through the
transform types the size allows and three quantizer rows.  (Quantization matrices and the partial-frequency shapes are pinned function by
function in tests/test_dsp_oracle_vs_ref.py, which needs the reference build.)  Kept per case: the per-block scalars in full and a CRC of
every coefficient / reconstruction array (the fixture stays small)."""
import zlib

import numpy as np

from svt_av1_psyex_amd import abi, rd
from txfm_cases import valid_types

W, H = 192, 128
CASES = [(ts, bd, qk) for ts in range(19) for bd, qk in ((8, ts & 1), (10, 1 - (ts & 1)))] + [(ts, 10, 0) for ts in (4, 12, 18)] + [(ts, 8, 1) for ts in (4, 11, 17)]
SCALARS = [name for name, _, _ in abi.RD_OUT_FIELDS]
ARRAYS = ("coeff", "qcoeff", "dqcoeff", "recon")


def quant_rows():
    return np.stack([rd.quant_row_from_step(8, 10), rd.quant_row_from_step(60, 75), rd.quant_row_from_step(500, 640)])


def inputs(case_index):
    ts, bd, qk = CASES[case_index]
    rng = np.random.default_rng(7000 + case_index)
    hi = (1 << bd) - 1
    dt = np.uint8 if bd == 8 else np.uint16
    base = rng.integers(0, hi + 1, (H // 8 + 2, W // 8 + 2)).astype(np.float64)
    up = np.kron(base, np.ones((8, 8)))[:H, :W]
    src = np.clip(up + rng.normal(0, 9 * (1 << (bd - 8)), (H, W)), 0, hi).astype(dt)
    pred = np.clip(up + rng.normal(0, 4 * (1 << (bd - 8)), (H, W)), 0, hi).astype(dt)
    src[:32, :64] = np.where(rng.integers(0, 2, (32, 64)) == 0, 0, hi)  # a corner of extreme residuals
    pred[:32, :64] = hi - src[:32, :64]
    jobs = rd.grid_jobs(W, H, W, ts)
    types = valid_types(ts)
    jobs["tx_type"] = [types[i % len(types)] for i in range(len(jobs))]
    jobs["quant_row"] = [(i // 3) % 3 for i in range(len(jobs))]
    f = dict(bit_depth=bd, quant_kind=qk, tx_size=ts, src_stride=W, pred_stride=W)
    return f, np.ascontiguousarray(src), np.ascontiguousarray(pred), jobs


def digest(out):
    """what the fixture keeps of one case's outputs"""
    d = {name: np.ascontiguousarray(out[name]) for name in SCALARS}
    d["crc"] = np.array([zlib.crc32(np.ascontiguousarray(out[name]).tobytes()) for name in ARRAYS], np.uint32)
    return d


def compare(want, got_digest):
    bad = [name for name in SCALARS if not np.array_equal(want[name], got_digest[name])]
    bad += [ARRAYS[i] for i in range(len(ARRAYS)) if int(want["crc"][i]) != int(got_digest["crc"][i])]
    return bad
