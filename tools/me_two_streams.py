"""Experiment: the bench's ME launch (16 pictures) as ONE launch against TWO half launches on two contexts' streams (8 pictures each, enqueued
back to back, running side by side).  usage (GPU box): python tools/me_two_streams.py"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "tests"), os.path.join(HERE, "..")]
import torch  # noqa: E402

import bench  # noqa: E402
from svt_av1_psyex_amd import api  # noqa: E402

ctx = api.Context()
ctx2 = api.Context()
wl = bench.Workload(ctx, 0, 1)


def run(split, n=12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        jobs = wl.me_jobs[i % bench.N_SETS][i % 2]
        if split:
            h = len(jobs) // 2 if split == 2 else split
            ctx.me_pictures_async(jobs[:h])
            ctx2.me_pictures_async(jobs[h:])
        else:
            ctx.me_pictures_async(jobs)
    ctx.sync(); ctx2.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(0, 3); run(2, 3)
for rep in range(3):
    print(f"one launch {run(0):.3f} ms   two halves on two streams {run(2):.3f} ms   4 + 12 {run(4):.3f} ms", flush=True)
