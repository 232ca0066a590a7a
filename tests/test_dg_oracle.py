"""CPU: the dynamic-GOP detector restatement (oracle/dg_oracle.c) against the reference's own dg_detector_hme_level0
(oracle/_ref, build container) and against the committed fixture (tests/golden/dg_detector.npz, everywhere)."""
import os

import numpy as np
import pytest

import pyoracle
from dg_cases import GRID, METRICS, DgCase

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dg_detector.npz")


@pytest.mark.parametrize("w,h,kind", GRID)
def test_oracle_vs_reference(ref, w, h, kind):
    c = DgCase(w, h, kind)
    got = pyoracle.dg_detector("oracle", c.src, c.ref, *c.args())
    # the reference accumulates segment by segment (me_process.c:326-331): any split gives the same sums
    for segs in ((1, 1), (3, 2)):
        want = pyoracle.dg_detector("ref", c.src, c.ref, *c.args(), segments=segs)
        assert {k: got[k] for k in METRICS} == want, (c, segs)
    # per-block results are consistent with the sums they feed
    assert int(got["b64_sad"].astype(np.uint64).sum()) == got["tot_dist"]
    assert int((got["b64_mv"] != 0).any(axis=1).sum()) == got["tot_active"]


def test_flat_content_takes_the_first_position(oracle):
    c = DgCase(712, 400, "flat")
    got = pyoracle.dg_detector("oracle", c.src, c.ref, *c.args())
    assert (got["b64_sad"] == 16 * 16 * 41).all()
    # first position of the window: minus half the search side of this resolution class (16 here: the 15 px of usable
    # padding hold it even at the picture's top-left corner), in full-resolution pixels
    assert c.input_resolution == 1 and (got["b64_mv"] == -8 * 4).all()


def test_oracle_vs_golden(oracle):
    z = np.load(GOLDEN)
    for i, (w, h, kind) in enumerate(zip(z["width"], z["height"], z["kind"])):
        c = DgCase(int(w), int(h), str(kind))
        got = pyoracle.dg_detector("oracle", c.src, c.ref, *c.args())
        assert [got[k] for k in METRICS] == [int(v) for v in z["metrics"][i]], c
