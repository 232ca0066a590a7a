"""CPU: the TPL dispenser's transform / SATD / quantize_fp / block_error chain, run by the REFERENCE's own functions
(oracle/ref_harness.c:ref_tpl_chain, build container), against the oracle's RD batch on the equivalent jobs."""
import ctypes as C

import numpy as np
import pytest

import pyoracle
from tpl_cases import GRID, W, batch, planes, tpl_outputs


@pytest.mark.parametrize("sub,pf,amp", GRID)
def test_rd_batch_reproduces_the_tpl_chain(ref, sub, pf, amp):
    src, pred = planes(100 + sub * 10 + pf, amp)
    fields, jobs, rows = batch(sub, pf)
    out = pyoracle.rd_batch(fields, src, pred, jobs, rows, want_recon=False)
    got = tpl_outputs(out, sub)
    n = 16 * (16 >> sub)
    co, q, dq = (np.full(256, 12345, np.int32) for _ in range(3))  # dirty buffers, as the dispenser's stack arrays are
    for j, jb in enumerate(jobs):
        o4 = np.zeros(4, np.int64)
        off = int(jb["src_offset"])
        rc = ref.ref_tpl_chain(C.c_void_p(src.ctypes.data + off), W, C.c_void_p(pred.ctypes.data + off), W, sub, pf,
                               C.c_void_p(rows.ctypes.data + int(jb["quant_row"]) * rows.dtype.itemsize), co.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p),
                               dq.ctypes.data_as(C.c_void_p), o4.ctypes.data_as(C.c_void_p))
        assert rc == 0
        assert [got[k][j] for k in ("inter_cost", "eob", "recon_error", "sse")] == list(o4), (sub, pf, amp, j)
        np.testing.assert_array_equal(out["coeff"][j], co[:n])
        np.testing.assert_array_equal(out["qcoeff"][j], q[:n])
        np.testing.assert_array_equal(out["dqcoeff"][j], dq[:n])


def test_oracle_vs_golden(oracle):
    """Everywhere (no reference needed): the committed outputs of the reference's chain (oracle/gen_golden.py tpl)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tpl_chain.npz"))
    for (sub, pf, amp), want in zip(z["grid"], z["results"]):
        src, pred = planes(100 + int(sub) * 10 + int(pf), int(amp))
        fields, jobs, rows = batch(int(sub), int(pf))
        got = tpl_outputs(pyoracle.rd_batch(fields, src, pred, jobs, rows, want_recon=False), int(sub))
        np.testing.assert_array_equal(np.stack([got[k] for k in ("inter_cost", "eob", "recon_error", "sse")], axis=1), want)
