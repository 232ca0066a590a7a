"""CPU: the TPL dispenser's transform / SATD / quantize_fp / block_error chain, run by the REFERENCE's own functions
(oracle/ref_harness.c:ref_tpl_chain, build container), against the oracle's RD batch on the equivalent jobs."""
import ctypes as C

import numpy as np
import pytest

import pyoracle
from tpl_cases import GRID, W, batch, planes, seed_of, tpl_outputs


@pytest.mark.parametrize("level,sub,pf,amp", GRID)
def test_rd_batch_reproduces_the_tpl_chain(ref, level, sub, pf, amp):
    src, pred = planes(seed_of(level, sub, pf), amp)
    fields, jobs, rows = batch(level, sub, pf)
    out = pyoracle.rd_batch(fields, src, pred, jobs, rows, want_recon=False)
    got = tpl_outputs(out, level, sub)
    size = 16 << level
    n = size * (size >> sub)
    co, q, dq = (np.full(1024, 12345, np.int32) for _ in range(3))  # dirty buffers, as the dispenser's stack arrays are
    for j, jb in enumerate(jobs):
        o4 = np.zeros(4, np.int64)
        off = int(jb["src_offset"])
        rc = ref.ref_tpl_chain(C.c_void_p(src.ctypes.data + off), W, C.c_void_p(pred.ctypes.data + off), W, level, sub, pf,
                               C.c_void_p(rows.ctypes.data + int(jb["quant_row"]) * rows.dtype.itemsize), co.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p),
                               dq.ctypes.data_as(C.c_void_p), o4.ctypes.data_as(C.c_void_p))
        assert rc == 0
        assert [got[k][j] for k in ("inter_cost", "eob", "recon_error", "sse")] == list(o4), (level, sub, pf, amp, j)
        np.testing.assert_array_equal(out["coeff"][j], co[:n])
        np.testing.assert_array_equal(out["qcoeff"][j], q[:n])
        np.testing.assert_array_equal(out["dqcoeff"][j], dq[:n])


def test_oracle_vs_golden(oracle):
    """Everywhere (no reference needed): the committed outputs of the reference's chain (oracle/gen_golden.py tpl)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tpl_chain.npz"))
    assert len(z["grid"]) == len(GRID)
    for i, (level, sub, pf, amp) in enumerate(z["grid"]):
        src, pred = planes(seed_of(int(level), int(sub), int(pf)), int(amp))
        fields, jobs, rows = batch(int(level), int(sub), int(pf))
        got = tpl_outputs(pyoracle.rd_batch(fields, src, pred, jobs, rows, want_recon=False), int(level), int(sub))
        np.testing.assert_array_equal(np.stack([got[k] for k in ("inter_cost", "eob", "recon_error", "sse")], axis=1), z[f"results_{i}"])
