"""GPU parity: the TPL dispenser's kernel chain as svt_hip_rd_batch jobs (tests/tpl_cases.py) against the oracle and against
the committed outputs of the reference's own chain (tests/golden/tpl_chain.npz)."""
import os

import numpy as np
import pytest

import pyoracle
from svt_av1_psyex_amd import rd
from tpl_cases import GRID, batch, planes, seed_of, tpl_outputs

pytestmark = pytest.mark.gpu
KEYS = ("inter_cost", "eob", "recon_error", "sse")


@pytest.mark.parametrize("level,sub,pf,amp", GRID)
def test_tpl_chain_vs_oracle(hip_ctx, level, sub, pf, amp):
    src, pred = planes(seed_of(level, sub, pf), amp)
    fields, jobs, rows = batch(level, sub, pf)
    want = pyoracle.rd_batch(fields, src, pred, jobs, rows, want_recon=False)
    got = rd.run_hip(hip_ctx, fields, src, pred, jobs, rows, want_recon=False)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def test_tpl_chain_vs_reference_fixture(hip_ctx):
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tpl_chain.npz"))
    for i, (level, sub, pf, amp) in enumerate(z["grid"]):
        src, pred = planes(seed_of(int(level), int(sub), int(pf)), int(amp))
        fields, jobs, rows = batch(int(level), int(sub), int(pf))
        got = tpl_outputs(rd.run_hip(hip_ctx, fields, src, pred, jobs, rows, want_recon=False), int(level), int(sub))
        np.testing.assert_array_equal(np.stack([got[k] for k in KEYS], axis=1), z[f"results_{i}"])
