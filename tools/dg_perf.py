"""Times dg_hme_level0_kernel at 2160p (HIP events on the context stream via torch's external-stream wrapper)."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")]
import torch
from svt_av1_psyex_amd import api
from dg_cases import DgCase

ctx = api.Context(0)
for (w, h) in ((3840, 2160), (1920, 1080), (640, 480)):
    c = DgCase(w, h, "pan", distance=4, seed=11)
    src, ref = ctx.upload(c.src), ctx.upload(c.ref)
    n = ((c.aligned_width + 63) // 64) * ((c.aligned_height + 63) // 64)
    m = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    sad = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    mv = torch.zeros(n * 2, dtype=torch.int16, device="cuda:0")
    ext = torch.cuda.ExternalStream(ctx.stream, device="cuda:0")
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.dg_detector_hme_level0_async(src, ref, *c.args(), m.data_ptr(), sad.data_ptr(), mv.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    with torch.cuda.stream(ext):
        e0.record()
        for _ in range(reps):
            ctx.dg_detector_hme_level0_async(src, ref, *c.args(), m.data_ptr(), sad.data_ptr(), mv.data_ptr())
        e1.record()
    ctx.sync()
    t = e0.elapsed_time(e1) / reps
    side = 16 if c.input_resolution <= 1 else 64 if c.input_resolution <= 2 else 128
    print(f"{w}x{h}: {t * 1000:.1f} us per picture (memset + kernel), {n} b64, {n * side * side * 256 / t / 1e6:.0f} G |a-b|/s")
