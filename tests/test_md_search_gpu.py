"""GPU parity: svt_hip_md_fullpel_batch / svt_hip_md_subpel_batch (csrc/md_search_kernel.hip) against the oracle and against the committed
outputs of the reference's own md_full_pel_search chains and svt_av1_find_best_sub_pixel_tree_pruned (tests/golden/md_search.npz)."""
import os

import numpy as np
import pytest

import md_search_cases as mc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "md_search.npz")


@pytest.mark.parametrize("gi", range(len(mc.FULLPEL_GRID)))
def test_fullpel_chain(hip_ctx, oracle, gi):
    dist, psad, ctype = mc.FULLPEL_GRID[gi]
    rng = np.random.default_rng(100 + dist * 10 + psad * 3 + ctype)
    src, refp = mc.planes(7 + dist)
    tables = mc.cost_tables(rng)
    rounds = mc.fullpel_chain(rng, 40, dist, psad)
    want = mc.run_fullpel_cpu(oracle.orc_md_fullpel_batch, src, refp, rounds, ctype, 37, tables)
    got = mc.run_fullpel_hip(hip_ctx, src, refp, rounds, ctype, 37, tables)
    z = np.load(GOLDEN)
    for r, ((cw, mw), (cg, mg)) in enumerate(zip(want, got)):
        np.testing.assert_array_equal(cg, cw, err_msg=f"round {r} cost vs oracle")
        np.testing.assert_array_equal(mg, mw, err_msg=f"round {r} mv vs oracle")
        np.testing.assert_array_equal(cg, z[f"fp_cost_{gi}"][r], err_msg=f"round {r} cost vs reference fixture")
        np.testing.assert_array_equal(mg, z[f"fp_mv_{gi}"][r], err_msg=f"round {r} mv vs reference fixture")


@pytest.mark.parametrize("si", range(len(mc.SUBPEL_SETTINGS)))
def test_subpel_tree_searches(hip_ctx, oracle, si):
    rng = np.random.default_rng(300 + si)
    src, refp = mc.planes(11 + si)
    tables = mc.cost_tables(rng)
    jobs = mc.subpel_jobs(rng, 60)
    want = mc.run_subpel_cpu(oracle.orc_md_subpel_batch, src, refp, jobs, mc.SUBPEL_SETTINGS[si], 41, 36, tables)
    got = mc.run_subpel_hip(hip_ctx, src, refp, jobs, mc.SUBPEL_SETTINGS[si], 41, 36, tables)
    z = np.load(GOLDEN)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=f"{k} vs oracle")
        np.testing.assert_array_equal(got[k], z[f"sp_{k}_{si}"], err_msg=f"{k} vs reference fixture")


def test_many_blocks_random_settings(hip_ctx, oracle):
    """a larger randomized sweep: 600 sub-pel jobs per setting with other seeds, full-pel chains of 300 blocks"""
    for si in range(len(mc.SUBPEL_SETTINGS)):
        rng = np.random.default_rng(900 + si)
        src, refp = mc.planes(40 + si)
        tables = mc.cost_tables(rng)
        jobs = mc.subpel_jobs(rng, 600)
        want = mc.run_subpel_cpu(oracle.orc_md_subpel_batch, src, refp, jobs, mc.SUBPEL_SETTINGS[si], 23 + si, 20 + 7 * si, tables)
        got = mc.run_subpel_hip(hip_ctx, src, refp, jobs, mc.SUBPEL_SETTINGS[si], 23 + si, 20 + 7 * si, tables)
        for k in want:
            np.testing.assert_array_equal(got[k], want[k], err_msg=f"{k} setting {si}")
    for gi, (dist, psad, ctype) in enumerate(mc.FULLPEL_GRID):
        rng = np.random.default_rng(1200 + gi)
        src, refp = mc.planes(60 + gi)
        tables = mc.cost_tables(rng)
        rounds = mc.fullpel_chain(rng, 300, dist, psad)
        want = mc.run_fullpel_cpu(oracle.orc_md_fullpel_batch, src, refp, rounds, ctype, 19 + gi, tables)
        got = mc.run_fullpel_hip(hip_ctx, src, refp, rounds, ctype, 19 + gi, tables)
        for r, ((cw, mw), (cg, mg)) in enumerate(zip(want, got)):
            np.testing.assert_array_equal(cg, cw, err_msg=f"grid {gi} round {r} cost")
            np.testing.assert_array_equal(mg, mw, err_msg=f"grid {gi} round {r} mv")
