#!/usr/bin/env python3
"""bench.py -- ME + RD-cost throughput of the MI355X backend on synthetic 2160p 10-bit pictures, preset 6.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

A *step* = one mini-GOP worth of 16 pictures of the sequence (4 current pictures x reference distances 1, 2, 4, 8;
temporal layers 4..1; R = 2 references per picture, one per list), each going through
  1. open-loop ME for every 64x64 block (== N x svt_aom_motion_estimation_b64); the pictures of the step are in
     flight together, as in the reference's ME threads, and share ONE launch (svt_hip_me_pictures_async),
  2. full-pel motion-compensated 10-bit prediction from the ME winners (svt_hip_fullpel_pred_batch, one launch for the step),
  3. the RD kernels on the 10-bit luma at three transform depths (64x64, 32x32, 16x16, DCT_DCT, "b" quantizer):
     residual -> fwd txfm -> SATD -> quantize -> coeff distortion -> inv txfm -> SSE (svt_hip_rd_batch; one batch per
     depth holds the blocks of all pictures of the step).
With N GPUs the b64 rows of every picture are sharded across the ranks in contiguous bands (all planes are replicated; no
halo exchange; the rows that do not divide by N rotate over the ranks from picture to picture, so every rank owns the same
number of rows per step) and the per-b64 ME results (MeSbResults arrays + the per-b64 scalars, ~1 KB per block) are all-gathered over
RCCL once per step, on a side stream behind the ME launch so that it overlaps the prediction / RD kernels.
`value` = luma pixels of the pictures fully processed per second, whole job (strong scaling: the pictures per step
are fixed, each rank handles 1/N of the b64 rows).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from svt_av1_psyex_amd import abi, api, rd, shard, synth  # noqa: E402

W, H = 3840, 2160
DISTS = (8, 1, 4, 2)  # long and short searches alternate: the per-XCD job queues of the ME launch (2 pictures each at N = 1) stay balanced
LAYER = {1: 4, 2: 3, 4: 2, 8: 1}
CURS = (8, 9, 10, 11)  # current pictures of one step; picture p = (CURS[p // 4], DISTS[p % 4])
CUR = CURS[0]
PICS = [(cur, d) for cur in CURS for d in DISTS]
N_FRAMES = max(CURS) + max(DISTS) + 1
RD_SIZES = (4, 3, 2)  # TX_64X64, TX_32X32, TX_16X16
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Workload:
    """Device-resident inputs and per-step launch plan for one rank."""

    def __init__(self, ctx, rank, world, seed=11):
        self.ctx, self.rank, self.world = ctx, rank, world
        t0 = time.time()
        y10 = synth.synth_sequence(W, H, N_FRAMES, seed)
        y8 = synth.to_8bit(y10)
        self.host8 = {i: synth.HostPyramid(y8[i], i) for i in range(N_FRAMES)}
        self.y10_host = y10
        self.pics = {i: ctx.upload(self.host8[i]) for i in range(N_FRAMES)}
        # 10-bit planes: the current pictures contiguous (one RD batch addresses all of them), the references one by one
        self.src10 = torch.from_numpy(np.stack([y10[c] for c in CURS]).astype(np.int16)).cuda().view(torch.int16).reshape(-1)
        self.y10 = {i: torch.from_numpy(y10[i].astype(np.int16)).cuda().view(torch.int16) for i in sorted({c - d for c, d in PICS})}
        self.w64, self.h64 = (W + 63) // 64, (H + 63) // 64
        # contiguous b64 row band of this rank in every picture of the step; the rows that do not divide evenly rotate over the
        # ranks from picture to picture (shard.rotation), so that every rank owns the same number of rows per step
        self.bands = [shard.band(self.h64, rank, world, shard.rotation(pi, self.h64, world)) for pi in range(len(PICS))]
        self.rows_per_step = sum(r1 - r0 for r0, r1 in self.bands)
        self.cfgs, self.descs = {}, {}
        for pi, (cur, d) in enumerate(PICS):
            self.cfgs[(cur, d)] = api.config_from_preset(6, W, H, qp=35, temporal_layer_index=LAYER[d], hierarchical_levels=4)
            desc = api.picture_desc(W, H, cur, {(0, 0): cur - d, (1, 0): cur + d}, enc_mode=6, temporal_layer_index=LAYER[d], hierarchical_levels=4)
            desc.b64_row_start, desc.b64_row_count = self.bands[pi][0], self.bands[pi][1] - self.bands[pi][0]
            self.descs[(cur, d)] = desc
        self.n_pu = abi.n_pu(desc.enable_me_16x16, desc.enable_me_8x8)
        # ME results of the pictures of a step: ONE compact device buffer holding only this rank's b64 rows (padded to the
        # largest band so that every rank contributes the same byte count to the all-gather): shard.BandLayout.  The
        # search-level MVs that feed this rank's prediction stay in a local buffer.
        self.layout = shard.BandLayout(self.w64, self.h64, world, self.n_pu, desc.max_refs, desc.max_cand, n_pictures=len(PICS))
        # Two result buffers, used by alternate steps: the exchange of step k may still be reading its buffer while the ME launch
        # of step k + 1 fills the other one.
        self.me_bufs = [torch.zeros(self.layout.nbytes, dtype=torch.uint8, device="cuda") for _ in range(2 if world > 1 else 1)]
        self.me_buf = self.me_bufs[0]
        nb = self.w64 * self.h64
        self.mv_buf = torch.zeros(len(PICS) * nb * 680, dtype=torch.int32, device="cuda")
        self.me_res, self.mv_ptr = [{} for _ in self.me_bufs], {}
        for pi, pic in enumerate(PICS):
            self.mv_ptr[pic] = self.mv_buf.data_ptr() + pi * nb * 680 * 4
            for k, buf in enumerate(self.me_bufs):
                self.me_res[k][pic] = self.layout.results_struct(buf.data_ptr(), pi, rank)
                self.me_res[k][pic].sb_best_mv = self.mv_ptr[pic]
        # RD: one prediction / recon plane per picture of the step (contiguous, so that one batch addresses all of them) and
        # job lists restricted to this rank's rows
        NP = len(PICS)
        self.pred = torch.zeros(NP * H * W, dtype=torch.int16, device="cuda")
        self.recon = torch.zeros(NP * H * W, dtype=torch.int16, device="cuda")
        self.rows = torch.from_numpy(np.stack([rd.quant_row_from_step(140, 176)]).view(np.uint8).reshape(-1)).cuda()
        self.rd = []
        self.rd_pixels = 0
        for ts in RD_SIZES:
            jobs = rd.grid_jobs(W, H, W, ts)
            ys = (jobs["src_offset"] // W).astype(np.int64)
            allp = []
            for pi in range(NP):
                y_lo, y_hi = self.bands[pi][0] * 64, min(self.bands[pi][1] * 64, H)
                keep = (ys >= y_lo) & (ys < y_hi)  # bands are whole b64 rows, so a block never straddles two ranks
                j = np.ascontiguousarray(jobs[keep]).copy()
                j["src_offset"] += (pi // len(DISTS)) * H * W  # its current picture
                j["pred_offset"] += pi * H * W                 # its prediction / recon plane
                allp.append(j)
            jobs = np.concatenate(allp)
            n = len(jobs)
            t_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).cuda()
            outs = {name: torch.zeros(max(n, 1) * k * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda") for name, dt, k in abi.RD_OUT_FIELDS}
            d = abi.RdBatchDesc(bit_depth=10, quant_kind=0, tx_size=ts, n_jobs=n, src_stride=W, pred_stride=W, src=self.src10.data_ptr(),
                                pred=self.pred.data_ptr(), recon=self.recon.data_ptr(), jobs=t_jobs.data_ptr(), quant_rows=self.rows.data_ptr(), n_quant_rows=1)
            for name, t in outs.items():
                setattr(d, name, t.data_ptr())
            self.rd.append((ts, d, t_jobs, outs, n))
            self.rd_pixels += n * abi.TX_W[ts] * abi.TX_H[ts]  # per step (all pictures)
        self.me_jobs = [[(self.cfgs[pic], self.descs[pic], self.pics[pic[0]], self.refs(pic), res[pic]) for pic in PICS] for res in self.me_res]
        # full-pel prediction of all pictures of the step: one launch
        self.pred_jobs = (abi.PredJob * NP)()
        for pi, (cur, d) in enumerate(PICS):
            pj = self.pred_jobs[pi]
            pj.ref, pj.sb_best_mv, pj.pred = self.y10[cur - d].data_ptr(), self.mv_ptr[(cur, d)], self.pred.data_ptr() + 2 * pi * H * W
            pj.b64_row_start, pj.b64_row_count, pj.list, pj.ref_idx = self.bands[pi][0], self.bands[pi][1] - self.bands[pi][0], 0, 0
        torch.cuda.synchronize()
        log(f"[rank {rank}] setup {time.time() - t0:.1f}s: {self.rows_per_step} of {self.h64 * NP} b64 rows per step, RD jobs {[r[4] for r in self.rd]}")

    def refs(self, pic):
        cur, d = pic
        return {(0, 0): self.pics[cur - d], (1, 0): self.pics[cur + d]}

    def step(self, ev=None, after_me=None, k=0):
        """Enqueue one step on the context stream.  `ev`: optional dict collecting (start, end) event pairs per kernel family;
        `after_me`: callback run right behind the ME launch (the multi-GPU exchange hooks in there); `k`: result buffer."""
        L = api.lib()
        mark = lambda: None
        if ev is not None:
            def mark():
                e = torch.cuda.Event(enable_timing=True); e.record(); return e
        e0 = mark()
        self.ctx.me_pictures_async(self.me_jobs[k])
        e1 = mark()
        if after_me is not None:
            after_me()
        self.ctx.check(L.svt_hip_fullpel_pred_batch(self.ctx._h, W, W, H, 10, W, len(self.pred_jobs), self.pred_jobs), "svt_hip_fullpel_pred_batch")
        e2 = mark()
        for ts, desc, _, _, n in self.rd:
            if n:
                self.ctx.check(L.svt_hip_rd_batch(self.ctx._h, C.byref(desc)), "svt_hip_rd_batch")
        e3 = mark()
        if ev is not None:
            ev["me"].append((e0, e1)); ev["pred"].append((e1, e2)); ev["rd"].append((e2, e3))


def cpu_baseline(wl, seconds_target=12.0):
    """The same work on the host cores for a bounded band of b64 rows (one row per thread; ctypes releases the GIL):
    ME of a distance-2 picture (R = 2) + the RD chain at the three depths.
      kind "reference": the reference's own kernels, compiled from its sources into oracle/_ref/libsvtref.so and driven by
        oracle/ref_harness.c -- svt_aom_motion_estimation_b64 with the AVX2 / SSE4.1 SAD kernels, and the RD chain through
        the AVX2 forward transforms / quantizer / distortions and the SSE4.1 inverse transforms (the dav1d .asm inverse
        needs nasm, which this image lacks);
      kind "port": the oracle (C restatement, bit-exact to the reference `_c` path) when that library is absent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import concurrent.futures as cf
    import pyoracle
    cores = min(os.cpu_count() or 1, 16)
    rows_total = cores  # one b64 row per thread
    row_start = wl.h64 // 2 - rows_total // 2
    d = 2
    cfg = wl.cfgs[(CUR, d)]
    src10 = wl.y10_host[CUR]
    pred10 = wl.y10_host[CUR - d]
    qrows = np.stack([rd.quant_row_from_step(140, 176)])
    refs8 = {(0, 0): wl.host8[CUR - d], (1, 0): wl.host8[CUR + d]}

    row_jobs = {}  # (row, tx_size) -> job list, built outside the timed region
    for ts in RD_SIZES:
        alljobs = rd.grid_jobs(W, H, W, ts)
        ys = alljobs["src_offset"] // W
        for row in range(row_start, row_start + rows_total):
            row_jobs[(row, ts)] = np.ascontiguousarray(alljobs[(ys >= row * 64) & (ys < row * 64 + 64)])

    def make_work(me_impl, rd_impl):
        def work(row):
            desc = abi.MePictureDesc.from_buffer_copy(bytes(wl.descs[(CUR, d)]))
            desc.b64_row_start, desc.b64_row_count = row, 1
            pyoracle.me_picture(me_impl, cfg, desc, wl.host8[CUR], refs8, search_level=False)
            for ts in RD_SIZES:
                jobs = row_jobs[(row, ts)]
                pyoracle.rd_batch(dict(bit_depth=10, quant_kind=0, tx_size=ts, src_stride=W, pred_stride=W), src10, pred10, jobs, qrows, want_coeffs=False,
                              want_recon=False, impl=rd_impl)
        return work

    def timed(work, budget):
        t0 = time.time()
        reps = 0
        while True:
            with cf.ThreadPoolExecutor(cores) as ex:
                list(ex.map(work, range(row_start, row_start + rows_total)))
            reps += 1
            if time.time() - t0 > budget or reps >= 8:
                break
        return reps * rows_total * 64 * W / (time.time() - t0) / 1e6, reps

    pyoracle.load_oracle()
    sample = "{reps} x {rows} b64 rows ({h}x{w} px) of the 2160p distance-2 picture: ME (R=2) + RD chain at 3 depths"
    have_ref = False
    if pyoracle.ref_available():
        try:
            ref = pyoracle.load_ref()
            have_ref = bool(ref.ref_has_avx2())
        except Exception:
            have_ref = False
    if have_ref:
        ref.ref_set_simd(1)
        ref.ref_set_simd_rd(1)
        v, reps = timed(make_work("ref", "ref_simd"), seconds_target)
        ref.ref_set_simd(0)
        out = {"value": round(v, 2), "unit": "Mpixels/s", "cores": cores, "kind": "reference",
               "sample": sample.format(reps=reps, rows=rows_total, h=rows_total * 64, w=W) + "; reference AVX2/SSE4.1 kernels (oracle/_ref)"}
        vp, _ = timed(make_work("oracle", "oracle"), seconds_target / 2)
        out["port_value"] = round(vp, 2)  # informational: the oracle's plain-C restatement on the same rows
    else:
        v, reps = timed(make_work("oracle", "oracle"), seconds_target)
        out = {"value": round(v, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
               "sample": sample.format(reps=reps, rows=rows_total, h=rows_total * 64, w=W) + "; oracle C restatement"}
    # The checker's other use ("verify parity on every timed run"): the rows just computed on the CPU, once more as one
    # band, against what the timed GPU steps left in the result buffer for the same picture.
    pi = PICS.index((CUR, d))
    desc = abi.MePictureDesc.from_buffer_copy(bytes(wl.descs[(CUR, d)]))
    desc.b64_row_start, desc.b64_row_count = row_start, rows_total
    cpu = pyoracle.me_picture("ref" if have_ref else "oracle", cfg, desc, wl.host8[CUR], refs8, search_level=False)
    gpu = wl.layout.unpack(wl.me_bufs[0].cpu().numpy(), pi)
    lo, hi = row_start * wl.w64, (row_start + rows_total) * wl.w64
    bad = [k for k in gpu if not np.array_equal(np.asarray(cpu[k]).reshape(gpu[k].shape)[lo:hi], gpu[k][lo:hi])]
    out["parity"] = f"ME results of the sample's {rows_total} b64 rows: GPU == {'reference build' if have_ref else 'oracle'}" if not bad else f"MISMATCH in {bad}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        log(f"--gpus {a.gpus} but WORLD_SIZE {world}: running with world size {world}")
    # SVT_BENCH_REHEARSAL=1: every rank uses GPU 0 and the exchange goes through gloo on host copies -- a functional dress
    # rehearsal of the N > 1 code path on a one-GPU box (band sharding, job lists, buffers, timing reductions); its `value`
    # means nothing.  Never set by the driver.
    rehearsal = os.environ.get("SVT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = api.Context(local_rank)
    ext = torch.cuda.ExternalStream(ctx.stream)
    # SVT_BENCH_EMULATE_RANK="r/N" (development aid, never set by the driver): a single process does the share of rank r of N
    # -- its row band, its job lists -- without any exchange, to read a rank's compute time at N GPUs off a one-GPU box
    emu = os.environ.get("SVT_BENCH_EMULATE_RANK")
    wl = Workload(ctx, *(map(int, emu.split("/")) if emu and world == 1 else (rank, world)))
    gather_outs = [torch.zeros(world * b.numel(), dtype=torch.uint8, device="cuda") for b in wl.me_bufs] if world > 1 else None
    landed = [None, None]  # event of the last exchange that read result buffer k
    step_no = [0]

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    comm = torch.cuda.Stream() if world > 1 else None

    def exchange(k):
        # per-b64 best-cost / MV / candidate results of this rank's rows -> every rank (RCCL over xGMI), on a side stream
        # ordered behind the ME launch; the prediction / RD kernels of the step -- and the next step's ME launch, which
        # writes the other result buffer -- overlap it
        done = torch.cuda.Event()
        done.record(ext)
        with torch.cuda.stream(comm):
            comm.wait_event(done)
            if rehearsal:
                host = wl.me_bufs[k].cpu()
                parts = [torch.zeros_like(host) for _ in range(world)]
                dist.all_gather(parts, host)
                gather_outs[k].copy_(torch.cat(parts), non_blocking=True)
            else:
                dist.all_gather_into_tensor(gather_outs[k], wl.me_bufs[k])
            landed[k] = torch.cuda.Event()
            landed[k].record(comm)

    def run(steps, ev=None):
        with torch.cuda.stream(ext):
            for _ in range(steps):
                k = step_no[0] % len(wl.me_bufs)
                step_no[0] += 1
                if world > 1 and landed[k] is not None:
                    ext.wait_event(landed[k])  # the exchange two steps back has read the result buffer this step's ME overwrites
                wl.step(ev, (lambda k=k: exchange(k)) if world > 1 else None, k)
        if world > 1:
            ext.wait_stream(comm)              # the timed region ends when the last exchange has landed

    run(a.warmup)
    barrier()
    t0 = time.perf_counter()
    run(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    # kernel-level durations (HIP events on the launch stream), separate short pass so events do not perturb the timed loop
    ev = {"me": [], "pred": [], "rd": []}
    run(min(a.steps, 3), ev)
    barrier()
    kms = {k: float(np.mean([s.elapsed_time(e) for s, e in v])) for k, v in ev.items()}
    if rehearsal and world > 1 and rank == 0:
        # the gathered buffers of all ranks, unpacked, must equal a whole-picture run (two pictures with different band rotations)
        for pi in (0, 5):
            pic = PICS[pi]
            whole_desc = abi.MePictureDesc.from_buffer_copy(bytes(wl.descs[pic]))
            whole_desc.b64_row_start, whole_desc.b64_row_count = 0, 0
            whole = ctx.me_picture(wl.cfgs[pic], whole_desc, wl.pics[pic[0]], wl.refs(pic), search_level=False)
            merged = wl.layout.unpack(gather_outs[(step_no[0] - 1) % 2].cpu().numpy(), pi)
            bad = [k for k in merged if not np.array_equal(np.asarray(whole[k]).reshape(merged[k].shape), merged[k])]
            log(f"rehearsal: gathered results of picture {pi} " + ("MATCH a whole-picture run" if not bad else f"DIFFER in {bad}"))
            assert not bad, bad
    t_all = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        if rehearsal:
            t_all = t_all.cpu()
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    dt = float(t_all.item())
    pictures = a.steps * len(PICS)
    value = pictures * W * H / dt / 1e6
    if rank == 0:
        R = 2
        frac_rows = wl.rows_per_step / (wl.h64 * len(PICS))
        me_bytes = (1.3125 * (1 + R) + 0.166 * R) * W * H * frac_rows * len(PICS)  # B_ME bytes per pixel x the pictures of one launch
        rd_bytes = wl.rd_pixels * (2 * 2 + 4 + 2)  # B_RD = 2*bpp + 4 (+bpp recon), bpp = 2; the step's three launches
        dom = "me" if kms["me"] >= kms["rd"] / len(RD_SIZES) else "rd"  # the single kernel with the longest launch
        # HBM bytes per launch from the PMC passes committed under profiles/ (same command, N = 1): rocprofv3 cannot run
        # inside this process, so the figure is the recorded one; null when it does not describe this run
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if world == 1 and dom == "me" and os.path.exists(tf):
            traffic = json.load(open(tf))["kernels"].get("svt_hip_me_b64_kernel", {}).get("hbm_bytes_per_launch")
        ach = (me_bytes if dom == "me" else rd_bytes) / (kms[dom] * 1e-3) / 1e9  # rd: the three launches together
        out = {
            "metric": "ME+RD-cost Mpixels/s (2160p10 preset-6)", "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8 SAD / i32 transforms", "data": "synthetic",
            "config": {"workload": "3840x2160 10-bit synthetic pan sequence, preset 6 (M6) search controls at qp 35; step = 16 pictures (4 current pictures x ref distance 1,2,4,8; R=2): "
                                   "open-loop ME of all 2040 b64 + full-pel pred + RD chain (64x64,32x32,16x16 DCT_DCT, 10-bit, b quantizer)",
                       "pictures_per_step": len(PICS), "b64_rows_per_rank_per_step": wl.rows_per_step, "parallelism": f"b64-row bands x{world} (left-over rows rotating over the ranks) + all-gather of ME results"},
            "roofline": {"bound": "hbm", "kernel": "svt_hip_me_b64_kernel" if dom == "me" else "rd_tx_kernel (3 sizes)", "achieved": round(ach, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(me_bytes if dom == "me" else rd_bytes), "avg_launch_ms": round(kms[dom], 4)},
            "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
