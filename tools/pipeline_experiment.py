"""Experiment: the step's ME + prediction on one stream and the RD launches of the PREVIOUS step on a second stream (a second context),
for several limits of the persistent ME waves per CU.  Timing only: the planes of consecutive steps are not double-buffered here."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SVT_BENCH_SETS", "3")
import torch
import bench
from svt_av1_psyex_amd import api

ctxA = api.Context(0)
ctxB = api.Context(0)
extA, extB = torch.cuda.ExternalStream(ctxA.stream), torch.cuda.ExternalStream(ctxB.stream)
wl = bench.Workload(ctxA, 0, 1)
L = api.lib()

def run(steps, pipelined):
    evA = None
    for i in range(steps):
        s, k = i % bench.N_SETS, i % 2
        with torch.cuda.stream(extA):
            ctxA.me_pictures_async(wl.me_jobs[s][k])
            ctxA.check(L.svt_hip_fullpel_pred_batch(ctxA._h, bench.W, bench.W, bench.H, 10, bench.W, len(wl.live), wl.pred_jobs[s]), "pred")
            e = torch.cuda.Event(); e.record(extA)
        ctx, ext = (ctxB, extB) if pipelined else (ctxA, extA)
        with torch.cuda.stream(ext):
            if pipelined:
                ext.wait_event(e)
            for ts, descs, _, _, n, _ in wl.rd:
                ctx.check(L.svt_hip_rd_batch(ctx._h, C.byref(descs[s])), "rd")
    ctxA.sync(); ctxB.sync(); torch.cuda.synchronize()

for waves in (0, 8, 6, 5, 4):
    L.svt_hip_context_set_me_waves_per_cu(ctxA._h, waves)
    for pipelined in (False, True):
        run(3, pipelined)
        t0 = time.perf_counter(); run(12, pipelined); dt = time.perf_counter() - t0
        print(f"ME waves/CU limit {waves or 'auto'}  {'two streams' if pipelined else 'one stream '}: {dt / 12 * 1e3:.3f} ms per step", flush=True)
