"""Random picture sizes / content through the dynamic-GOP detector kernel against the oracle.  usage: python tools/dg_fuzz.py [count]"""
import os
import sys

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", d) for d in ("tests", "oracle", "")]
import numpy as np

import pyoracle
from dg_cases import METRICS, DgCase
from svt_av1_psyex_amd import api

ctx = api.Context(0)
rng = np.random.default_rng(77)
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for i in range(n):
    w, h = 8 * int(rng.integers(8, 180)), 8 * int(rng.integers(8, 110))
    kind = str(rng.choice(["pan", "fast", "random", "flat", "extreme", "zoom"]))
    c = DgCase(w, h, kind, distance=int(rng.integers(1, 6)), seed=int(rng.integers(0, 1000)))
    want = pyoracle.dg_detector("oracle", c.src, c.ref, *c.args())
    src, ref = ctx.upload(c.src, bool(i & 1)), ctx.upload(c.ref, bool(i & 2))
    got = ctx.dg_detector_hme_level0(src, ref, *c.args())
    src.free(); ref.free()
    ok = all(got[k] == want[k] for k in METRICS) and np.array_equal(got["b64_sad"], want["b64_sad"]) and np.array_equal(got["b64_mv"], want["b64_mv"])
    if not ok:
        bad += 1
        print("MISMATCH", w, h, kind, {k: (got[k], want[k]) for k in METRICS}, flush=True)
print(f"done: {n} cases, {bad} bad")
ctx.close()
sys.exit(1 if bad else 0)
