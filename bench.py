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
     depth holds the blocks of all pictures of the step); the quantized coefficients (what the host's rate estimator reads), the
     reconstruction and the per-block scalars are written.
Consecutive steps work on DIFFERENT picture sets (the sequence, its mirror image, its upside-down image): the ~0.7 GB of
planes a step touches are not the ones the previous step left in the 256 MB Infinity Cache.
With N GPUs the b64 rows of every picture are sharded across the ranks in contiguous bands (all planes are replicated; no
halo exchange; the rows that do not divide by N rotate over the ranks from picture to picture, so every rank owns the same
number of rows per step) and the per-b64 ME results (MeSbResults arrays + the per-b64 scalars, ~1 KB per block, live rows only) are
all-gathered over RCCL once per step through libsvthip.so's own C entry (svt_hip_me_results_all_gather), on a side stream behind the ME
launch so that it overlaps the prediction / RD kernels.
`value` = luma pixels of the pictures fully processed per second, whole job (strong scaling: the pictures per step
are fixed, each rank handles 1/N of the b64 rows), inputs resident in HBM.  `end_to_end` repeats the measurement with the
PCIe legs inside the step: 4 new input pictures uploaded (H2D, the 1/4 and 1/16 planes built on the device) and the gathered ME results
copied back (D2H) per step, one step ahead / behind on the context's transfer stream.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the end-to-end leg and the other kernels' timings")
    return ap.parse_args(argv)


def spawn_ranks_if_needed():
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher (no WORLD_SIZE in the environment): this process -- which has
    not imported torch, loaded libsvthip.so or touched the GPU in any way yet -- starts the N ranks itself through
    torch.distributed.run as a CHILD process (one rank per GPU, rendezvous on 127.0.0.1) and exits with its status.  Never an exec:
    replacing a process that has initialised the GPU is forbidden on this pool, and a child is just as good."""
    if __name__ != "__main__" or "WORLD_SIZE" in os.environ:
        return
    a = parse_args()
    if a.gpus <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as so:  # a free rendezvous port
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {a.gpus} without a launcher: starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


spawn_ranks_if_needed()

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from svt_av1_psyex_amd import abi, api, rd, shard, stats, synth  # noqa: E402

W, H = 3840, 2160
DISTS = (8, 1, 4, 2)  # long and short searches alternate: the per-XCD job queues of the ME launch (2 pictures each at N = 1) stay balanced
LAYER = {1: 4, 2: 3, 4: 2, 8: 1}
CURS = (8, 9, 10, 11)  # current pictures of one step; picture p = (CURS[p // 4], DISTS[p % 4])
CUR = CURS[0]
PICS = [(cur, d) for cur in CURS for d in DISTS]
N_FRAMES = max(CURS) + max(DISTS) + 1
RD_SIZES = (4, 3, 2)  # TX_64X64, TX_32X32, TX_16X16
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
# issue rate of v_qsad_pk_u16_u8 over the whole chip, measured with tools/ubench/qsad_peak.hip on MI355X (profiles/r02_qsad_peak.txt):
# 16 |a-b| per lane-instruction; the "packed-SAD VALU peak" of SURVEY 8d
QSAD_PEAK_TOPS = float(os.environ.get("SVT_QSAD_PEAK_TOPS", "114.3"))
N_SETS = int(os.environ.get("SVT_BENCH_SETS", "3"))
R = 2


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def flip_set(y10, k):
    """Picture set k of the bench: the synthetic sequence (0), its mirror image (1), its upside-down image (2): same statistics and
    motion magnitudes, different bytes."""
    if k == 0:
        return y10
    return np.ascontiguousarray(y10[:, :, ::-1] if k == 1 else y10[:, ::-1, :])


class PictureSet:
    """Device-resident inputs of one picture set: luma pyramids of every frame, the 10-bit current and reference planes."""

    def __init__(self, ctx, y10, keep_host):
        y8 = synth.to_8bit(y10)
        host8 = {i: synth.HostPyramid(y8[i], i) for i in range(N_FRAMES)}
        self.pics = {i: ctx.upload(host8[i]) for i in range(N_FRAMES)}
        # 10-bit planes: the current pictures contiguous (one RD batch addresses all of them), the references one by one
        self.src10 = torch.from_numpy(np.stack([y10[c] for c in CURS]).astype(np.int16)).cuda().view(torch.int16).reshape(-1)
        self.y10 = {i: torch.from_numpy(y10[i].astype(np.int16)).cuda().view(torch.int16) for i in sorted({c - d for c, d in PICS})}
        self.host8 = host8 if keep_host else None
        self.y10_host = y10 if keep_host else None

    def refs(self, pic):
        cur, d = pic
        return {(0, 0): self.pics[cur - d], (1, 0): self.pics[cur + d]}


class Workload:
    """Device-resident inputs and per-step launch plan for one rank."""

    def __init__(self, ctx, rank, world, seed=11):
        self.ctx, self.rank, self.world = ctx, rank, world
        t0 = time.time()
        y10 = synth.synth_sequence(W, H, N_FRAMES, seed)
        self.sets = [PictureSet(ctx, flip_set(y10, k), keep_host=(k == 0)) for k in range(N_SETS)]
        self.w64, self.h64 = (W + 63) // 64, (H + 63) // 64
        NP = len(PICS)
        self.cfgs, self.descs = {}, {}
        # ME results of the pictures of a step: ONE compact device buffer holding only this rank's live b64 rows: shard.BandLayout.
        # The search-level MVs that feed this rank's prediction stay in a local buffer.
        probe = api.picture_desc(W, H, CUR, {(0, 0): CUR - 1, (1, 0): CUR + 1}, enc_mode=6)
        self.n_pu = abi.n_pu(probe.enable_me_16x16, probe.enable_me_8x8)
        self.layout = shard.BandLayout(self.w64, self.h64, world, self.n_pu, probe.max_refs, probe.max_cand, n_pictures=NP)
        # contiguous b64 row band of this rank in every picture of the step; the rows that do not divide evenly rotate over the
        # ranks from picture to picture (shard.rotation), so that every rank owns the same number of rows per step
        self.bands = [self.layout.band(pi, rank) for pi in range(NP)]
        self.live = [pi for pi in range(NP) if self.bands[pi][1] > self.bands[pi][0]]  # an empty band (more ranks than rows) launches nothing
        self.rows_per_step = sum(r1 - r0 for r0, r1 in self.bands)
        for pi, (cur, d) in enumerate(PICS):
            self.cfgs[(cur, d)] = api.config_from_preset(6, W, H, qp=35, temporal_layer_index=LAYER[d], hierarchical_levels=4)
            desc = api.picture_desc(W, H, cur, {(0, 0): cur - d, (1, 0): cur + d}, enc_mode=6, temporal_layer_index=LAYER[d], hierarchical_levels=4)
            desc.b64_row_start, desc.b64_row_count = self.bands[pi][0], self.bands[pi][1] - self.bands[pi][0]
            self.descs[(cur, d)] = desc
        # Two result buffers, used by alternate steps: the exchange of step k may still be reading its buffer while the ME launch
        # of step k + 1 fills the other one.
        self.me_bufs = [torch.zeros(max(self.layout.nbytes, 16), dtype=torch.uint8, device="cuda") for _ in range(2)]
        nb = self.w64 * self.h64
        self.mv_buf = torch.zeros(NP * nb * 680, dtype=torch.int32, device="cuda")
        self.me_res, self.mv_ptr = [{} for _ in self.me_bufs], {}
        for pi, pic in enumerate(PICS):
            self.mv_ptr[pic] = self.mv_buf.data_ptr() + pi * nb * 680 * 4
            for k, buf in enumerate(self.me_bufs):
                self.me_res[k][pic] = self.layout.results_struct(buf.data_ptr(), pi, rank)
                self.me_res[k][pic].sb_best_mv = self.mv_ptr[pic]
        # RD: one prediction / recon plane per picture of the step (contiguous, so that one batch addresses all of them) and
        # job lists restricted to this rank's rows; outputs incl. the quantized coefficients
        self.pred = torch.zeros(NP * H * W, dtype=torch.int16, device="cuda")
        self.recon = torch.zeros(NP * H * W, dtype=torch.int16, device="cuda")
        self.rows = torch.from_numpy(np.stack([rd.quant_row_from_step(140, 176)]).view(np.uint8).reshape(-1)).cuda()
        self.rd = []
        self.rd_pixels, self.rd_bytes = 0, 0
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
            npk = min(abi.TX_W[ts], 32) * min(abi.TX_H[ts], 32)
            t_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).cuda()
            outs = {name: torch.zeros(max(n, 1) * k * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda") for name, dt, k in abi.RD_OUT_FIELDS}
            outs["qcoeff"] = torch.zeros(max(n, 1) * npk * 4, dtype=torch.uint8, device="cuda")
            descs = []
            for s in self.sets:  # same jobs and outputs, the set's source planes
                d = abi.RdBatchDesc(bit_depth=10, quant_kind=0, tx_size=ts, n_jobs=n, src_stride=W, pred_stride=W, src=s.src10.data_ptr(),
                                    pred=self.pred.data_ptr(), recon=self.recon.data_ptr(), jobs=t_jobs.data_ptr(), quant_rows=self.rows.data_ptr(), n_quant_rows=1)
                for name, t in outs.items():
                    setattr(d, name, t.data_ptr())
                descs.append(d)
            self.rd.append((ts, descs, t_jobs, outs, n, jobs))
            px = n * abi.TX_W[ts] * abi.TX_H[ts]
            self.rd_pixels += px
            # B_RD = source + prediction read, reconstruction written (2 bytes each at 10 bit) + the packed quantized coefficients (4 bytes)
            self.rd_bytes += px * 6 + n * npk * 4
        self.me_jobs = [[[(self.cfgs[PICS[pi]], self.descs[PICS[pi]], s.pics[PICS[pi][0]], s.refs(PICS[pi]), res[PICS[pi]]) for pi in self.live] for res in self.me_res]
                        for s in self.sets]
        # full-pel prediction of all pictures of the step: one launch
        self.pred_jobs = []
        for s in self.sets:
            pj_arr = (abi.PredJob * max(len(self.live), 1))()
            for j, pi in enumerate(self.live):
                cur, d = PICS[pi]
                pj = pj_arr[j]
                pj.ref, pj.sb_best_mv, pj.pred = s.y10[cur - d].data_ptr(), self.mv_ptr[(cur, d)], self.pred.data_ptr() + 2 * pi * H * W
                pj.b64_row_start, pj.b64_row_count, pj.list, pj.ref_idx = self.bands[pi][0], self.bands[pi][1] - self.bands[pi][0], 0, 0
            self.pred_jobs.append(pj_arr)
        torch.cuda.synchronize()
        log(f"[rank {rank}] setup {time.time() - t0:.1f}s: {N_SETS} picture sets, {self.rows_per_step} of {self.h64 * NP} b64 rows per step, RD jobs {[r[4] for r in self.rd]}")

    def step(self, s=0, ev=None, after_me=None, k=0):
        """Enqueue one step on picture set `s` on the context stream.  `ev`: optional dict collecting (start, end) event pairs per kernel
        family; `after_me`: callback run right behind the ME launch (the multi-GPU exchange hooks in there); `k`: result buffer."""
        L = api.lib()
        mark = lambda: None
        if ev is not None:
            def mark():
                e = torch.cuda.Event(enable_timing=True); e.record(); return e
        e0 = mark()
        if self.live:
            self.ctx.me_pictures_async(self.me_jobs[s][k])
        e1 = mark()
        if after_me is not None:
            after_me()
        if self.live:
            self.ctx.check(L.svt_hip_fullpel_pred_batch(self.ctx._h, W, W, H, 10, W, len(self.live), self.pred_jobs[s]), "svt_hip_fullpel_pred_batch")
        e2 = mark()
        marks = [e2]
        for ts, descs, _, _, n, _ in self.rd:
            if n:
                self.ctx.check(L.svt_hip_rd_batch(self.ctx._h, C.byref(descs[s])), "svt_hip_rd_batch")
            marks.append(mark())
        if ev is not None:
            ev["me"].append((e0, e1)); ev["pred"].append((e1, e2)); ev["rd"].append((e2, marks[-1]))
            for (ts, *_), a, b in zip(self.rd, marks[:-1], marks[1:]):
                ev[f"rd{abi.TX_W[ts]}"].append((a, b))


class Exchange:
    """The per-step all-gather of the ME results: libsvthip.so's own RCCL entry (svt_hip_comm.h); torch.distributed only carries the
    128-byte id and the host barriers (gloo).  Falls back to torch.distributed's RCCL all-gather -- loudly -- when the C entry cannot
    start (e.g. no librccl.so on the box), and to gloo on host copies in the one-GPU rehearsal."""

    def __init__(self, ctx, wl, rank, world, dist, rehearsal, ext):
        self.ctx, self.wl, self.world, self.dist, self.rehearsal, self.ext = ctx, wl, world, dist, rehearsal, ext
        self.out = [torch.zeros(world * wl.me_bufs[0].numel(), dtype=torch.uint8, device="cuda") for _ in wl.me_bufs]
        self.kind, self.comm = "gloo rehearsal (host copies)", None
        self.landed = [None, None]
        if rehearsal:
            self.side = torch.cuda.Stream()
            return
        L = api.lib()
        ident = (C.c_uint8 * 128)()
        ok = torch.zeros(1, dtype=torch.int32)
        if rank == 0:
            ok[0] = 1 if L.svt_hip_comm_unique_id(ctx._h, ident) == 0 else 0
        t = torch.from_numpy(np.frombuffer(ident, np.uint8).copy())
        dist.broadcast(ok, 0); dist.broadcast(t, 0)
        comm = C.c_void_p()
        good = torch.zeros(1, dtype=torch.int32)
        if int(ok[0]):
            ident = (C.c_uint8 * 128)(*t.tolist())
            good[0] = 1 if L.svt_hip_comm_create(ctx._h, ident, rank, world, C.byref(comm)) == 0 else 0
        dist.all_reduce(good, op=dist.ReduceOp.MIN)
        if int(good[0]):
            self.kind, self.comm = "libsvthip.so svt_hip_me_results_all_gather (RCCL)", comm
            self.off = (C.c_size_t * world)(*wl.layout.offsets())
            self.cnt = (C.c_size_t * world)(*wl.layout.rank_bytes)
        else:
            log(f"[rank {rank}] svt_hip_comm_create failed ({L.svt_hip_last_error(ctx._h).decode()}): falling back to torch.distributed's RCCL all-gather")
            if comm:
                L.svt_hip_comm_destroy(comm)
            self.kind = "torch.distributed all_gather_into_tensor (RCCL)"
            self.side = torch.cuda.Stream()
            self.nccl = dist.new_group(backend="nccl")

    def before_step(self, k):
        """the exchange two steps back has read the result buffer this step's ME overwrites"""
        if self.comm:
            self.ctx.check(api.lib().svt_hip_comm_stream_wait(self.comm, k), "svt_hip_comm_stream_wait")
        elif self.landed[k] is not None:
            self.ext.wait_event(self.landed[k])

    def run(self, k):
        """per-b64 best-cost / MV / candidate results of this rank's rows -> every rank, on a side stream ordered behind the ME launch; the
        prediction / RD kernels of the step -- and the next step's ME launch, which writes the other result buffer -- overlap it"""
        wl = self.wl
        if self.comm:
            L = api.lib()
            if wl.layout.uniform:
                rc = L.svt_hip_me_results_all_gather(self.comm, k, C.c_void_p(wl.me_bufs[k].data_ptr()), C.c_void_p(self.out[k].data_ptr()), C.c_size_t(wl.layout.nbytes))
            else:
                rc = L.svt_hip_me_results_all_gather_v(self.comm, k, C.c_void_p(wl.me_bufs[k].data_ptr()), C.c_void_p(self.out[k].data_ptr()), self.off, self.cnt)
            self.ctx.check(rc, "svt_hip_me_results_all_gather")
            return
        done = torch.cuda.Event()
        done.record(self.ext)
        with torch.cuda.stream(self.side):
            self.side.wait_event(done)
            if self.rehearsal:
                host = wl.me_bufs[k].cpu()
                parts = [torch.zeros_like(host) for _ in range(self.world)]
                self.dist.all_gather(parts, host)
                self.out[k].copy_(torch.cat(parts), non_blocking=True)
            else:
                self.dist.all_gather_into_tensor(self.out[k], wl.me_bufs[k], group=self.nccl)
            self.landed[k] = torch.cuda.Event()
            self.landed[k].record(self.side)

    def finish(self):
        """the timed region ends when the last exchange has landed"""
        if self.comm:
            self.ctx.check(api.lib().svt_hip_comm_sync(self.comm), "svt_hip_comm_sync")
        else:
            self.ext.wait_stream(self.side)

    def close(self):
        if self.comm:
            api.lib().svt_hip_comm_destroy(self.comm)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs the cgroup of this process may use at once (cpu.max = "quota period"; the GPU boxes give a job a share of the host's cores), or
    None when unlimited / unreadable.  sched_getaffinity is the other limit."""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return aff, quota


def cpu_baseline(wl, seconds=3.0, repeats=3):
    """The same work on the host's cores for a bounded sample of the step: every b64 row of four 2160p pictures (one per reference
    distance): ME (R = 2) + the RD chain at the three depths; one b64 row per work item.
      kind "reference": the reference's own kernels, compiled from its sources into oracle/_ref/libsvtref.so and driven by
        oracle/ref_harness.c:ref_bench_rows -- NATIVE pthreads, each with its own MeContext / pcs / transform scratch created before
        the clock starts (no Python, no allocation and no lock in the timed loop; round 2's Python thread pool measured the GIL and the
        allocator, not the kernels): svt_aom_motion_estimation_b64 with the AVX2 / SSE4.1 SAD kernels, and the RD chain through the AVX2
        forward transforms / quantizer / distortions and the SSE4.1 inverse transforms (the dav1d .asm inverse needs nasm, which this
        image lacks).  Timed on 1 thread, on half and on all of the host's hardware threads (`repeats` runs of `seconds` each for
        the latter two, median); `value` is the better of the two, `cores` the threads it used.
      kind "port": the oracle (C restatement, bit-exact to the reference `_c` path) through a Python thread pool when that library is absent.
    The checker's other uses ("verify parity on every timed run"): the sample's ME rows and its RD blocks -- recomputed on the
    prediction planes the GPU made -- against what the timed GPU steps left behind; and the |a-b| count of the sample's searches."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import concurrent.futures as cf
    import pyoracle
    s0 = wl.sets[0]
    hw_threads = os.cpu_count() or 1
    qrows = np.stack([rd.quant_row_from_step(140, 176)])

    def refs8(d):
        return {(0, 0): s0.host8[CUR - d], (1, 0): s0.host8[CUR + d]}

    o = pyoracle.load_oracle()
    o.orc_sad_ops.restype = C.c_uint64
    have_ref = False
    if pyoracle.ref_available():
        try:
            ref = pyoracle.load_ref()
            have_ref = bool(ref.ref_has_avx2())
        except Exception:
            have_ref = False
    if have_ref:
        pictures = []
        for d in DISTS:
            desc = abi.MePictureDesc.from_buffer_copy(bytes(wl.descs[(CUR, d)]))
            desc.b64_row_start, desc.b64_row_count = 0, 0
            pictures.append((wl.cfgs[(CUR, d)], desc, s0.host8[CUR], refs8(d), s0.y10_host[CUR], s0.y10_host[CUR - d]))

        def rate(nt, secs):
            return pyoracle.ref_bench_rows(pictures, W, H, 0, wl.h64, qrows, RD_SIZES, nt, secs, simd=True)[0]

        one = rate(1, seconds)
        affinity, quota = cpu_quota()
        # scan the thread count (one short run each), then repeat at the best one: where the rate stops growing is the CPU share the job
        # really has (the GPU boxes hand a job a slice of the host's cores; cpu.max / the affinity mask say so when they are readable)
        counts, nt = [], 2
        while nt < min(hw_threads, affinity):
            counts.append(nt)
            nt *= 2
        counts.append(min(hw_threads, affinity))
        if quota and int(round(quota)) not in counts:
            counts.append(max(1, int(round(quota))))
        scan = {nt: rate(nt, min(seconds, 2.0)) for nt in sorted(set(counts))}
        best = max(scan, key=lambda nt: scan[nt])
        runs = [rate(best, seconds) for _ in range(repeats)]
        v = statistics.median(runs)
        legs = dict(scan)
        legs[best] = v
        out = {"value": round(v, 1), "unit": "Mpixels/s", "cores": best, "hw_threads": hw_threads, "cpu": cpu_model(), "kind": "reference",
               "runs": [round(r, 1) for r in runs], "one_thread": round(one, 2), "scaling_efficiency": round(v / (best * one), 3),
               "by_threads": {str(nt): round(legs[nt], 1) for nt in sorted(legs)},
               "efficiency_by_threads": {str(nt): round(legs[nt] / (nt * one), 3) for nt in sorted(legs)}, "cgroup_cpu_quota": quota, "affinity_cpus": affinity,
               "linear_extrapolation_all_physical_cores": {"value": round(one * max(1, hw_threads // 2), 1), "unit": "Mpixels/s",
                                                           "what": "one_thread x physical cores (hw_threads / 2): an UPPER bound for the whole host (no SMT gain, no all-core clock drop, "
                                                                   "no memory contention); NOT measured: a job on the GPU box owns only a share of the host's cores (see cgroup_cpu_quota, affinity_cpus and where efficiency_by_threads collapses)"},
               "sample": f"all {wl.h64} b64 rows of 4 pictures (reference distance 8, 1, 4, 2; R=2) of the bench's 2160p sequence: ME + RD chain at 3 depths per row; "
                         f"native pthreads (oracle/ref_harness.c:ref_bench_rows), per-thread state created before the clock starts; 1 thread {seconds:g} s, "
                         f"a scan over {sorted(scan)} threads (one run each), then median of {repeats} runs of {seconds:g} s at the best count; reference AVX2/SSE4.1 kernels (oracle/_ref)"}
    else:
        cores = hw_threads
        rows_per_pic = max(1, min(wl.h64, -(-cores // len(DISTS))))
        row0 = wl.h64 // 2 - rows_per_pic // 2
        items = [(d, row) for d in DISTS for row in range(row0, row0 + rows_per_pic)]
        row_jobs = {}
        for ts in RD_SIZES:
            alljobs = rd.grid_jobs(W, H, W, ts)
            ys = alljobs["src_offset"] // W
            for row in range(row0, row0 + rows_per_pic):
                row_jobs[(row, ts)] = np.ascontiguousarray(alljobs[(ys >= row * 64) & (ys < row * 64 + 64)])

        def work(item):
            d, row = item
            desc = abi.MePictureDesc.from_buffer_copy(bytes(wl.descs[(CUR, d)]))
            desc.b64_row_start, desc.b64_row_count = row, 1
            pyoracle.me_picture("oracle", wl.cfgs[(CUR, d)], desc, s0.host8[CUR], refs8(d), search_level=False)
            for ts in RD_SIZES:
                pyoracle.rd_batch(dict(bit_depth=10, quant_kind=0, tx_size=ts, src_stride=W, pred_stride=W), s0.y10_host[CUR], s0.y10_host[CUR - d], row_jobs[(row, ts)],
                                  qrows, want_coeffs=False, want_recon=False, impl="oracle")

        rates = []
        with cf.ThreadPoolExecutor(cores) as ex:
            list(ex.map(work, items))
            for _ in range(repeats):
                t0, reps = time.time(), 0
                while time.time() - t0 < seconds:
                    list(ex.map(work, items))
                    reps += 1
                rates.append(reps * len(items) * 64 * W / (time.time() - t0) / 1e6)
        out = {"value": round(statistics.median(rates), 2), "unit": "Mpixels/s", "cores": cores, "cpu": cpu_model(), "kind": "port", "runs": [round(r, 1) for r in rates],
               "sample": f"{len(items)} b64 rows; oracle C restatement through a Python thread pool (oracle/_ref absent): a lower bound"}
    # the rows the parity leg recomputes: every row of the four pictures
    rows_per_pic = wl.h64
    row_start = 0
    items = [(d, row) for d in DISTS for row in range(row_start, row_start + rows_per_pic)]
    # ---- parity of the timed GPU run on the sample (ME rows + RD blocks), and the |a-b| count of its searches ----
    # The last timed step on picture set 0 left its results in result buffer `wl.last_k0` (main() records it).
    bad = []
    o.orc_sad_ops(1)
    for d in DISTS:
        pi = PICS.index((CUR, d))
        desc = abi.MePictureDesc.from_buffer_copy(bytes(wl.descs[(CUR, d)]))
        desc.b64_row_start, desc.b64_row_count = row_start, rows_per_pic
        cpu = pyoracle.me_picture("oracle", wl.cfgs[(CUR, d)], desc, s0.host8[CUR], refs8(d), search_level=False)
        if wl.world == 1:
            gpu = wl.layout.unpack(wl.me_bufs[wl.last_k0].cpu().numpy()[None], pi)
            lo, hi = row_start * wl.w64, (row_start + rows_per_pic) * wl.w64
            bad += [f"ME d{d} {k}" for k in gpu if not np.array_equal(np.asarray(cpu[k]).reshape(gpu[k].shape)[lo:hi], gpu[k][lo:hi])]
    sad_ops_sample = int(o.orc_sad_ops(1))
    out["sad_ops_per_b64"] = round(sad_ops_sample / (len(DISTS) * rows_per_pic * wl.w64), 1)
    if wl.world == 1:
        # RD: the GPU's prediction planes (full-pel MC from its own ME winners) come back to the host; the checker runs the chain on them
        for d in DISTS:
            pi = PICS.index((CUR, d))
            y_lo, y_hi = row_start * 64, min((row_start + rows_per_pic) * 64, H)
            pred_host = wl.pred[pi * H * W:(pi + 1) * H * W].cpu().numpy().view(np.uint16).reshape(H, W)
            recon_host = wl.recon[pi * H * W:(pi + 1) * H * W].cpu().numpy().view(np.uint16).reshape(H, W)
            for ts, descs, _, outs, n, jobs in wl.rd:
                sel = np.flatnonzero((jobs["pred_offset"] >= pi * H * W + y_lo * W) & (jobs["pred_offset"] < pi * H * W + y_hi * W))
                jb = np.ascontiguousarray(jobs[sel]).copy()
                jb["src_offset"] -= (pi // len(DISTS)) * H * W
                jb["pred_offset"] -= pi * H * W
                want = pyoracle.rd_batch(dict(bit_depth=10, quant_kind=0, tx_size=ts, src_stride=W, pred_stride=W), s0.y10_host[CUR], pred_host, jb, qrows, want_coeffs=True,
                                         want_recon=True, impl="ref" if (have_ref and ts in (2, 3, 4)) else "oracle")
                npk = min(abi.TX_W[ts], 32) * min(abi.TX_H[ts], 32)
                for name, dt, kk in abi.RD_OUT_FIELDS:
                    got = outs[name].cpu().numpy().view(dt).reshape(-1, kk)[sel]
                    if not np.array_equal(want[name], got):
                        bad.append(f"RD d{d} {abi.TX_W[ts]}x{abi.TX_H[ts]} {name}")
                gq = outs["qcoeff"].cpu().numpy().view(np.int32).reshape(-1, npk)[sel]
                if not np.array_equal(want["qcoeff"], gq):
                    bad.append(f"RD d{d} {abi.TX_W[ts]}x{abi.TX_H[ts]} qcoeff")
                if ts == RD_SIZES[-1] and not np.array_equal(want["recon"][y_lo:y_hi], recon_host[y_lo:y_hi]):  # the last launch's reconstruction is what the plane holds
                    bad.append(f"RD d{d} recon")
        checker = "reference build (RD) / oracle (ME)" if have_ref else "oracle"
        out["parity"] = (f"ME results and RD outputs (eob, satd, distortions, cul_level, qcoeff, recon) of the sample's {len(items)} b64 rows: GPU == {checker}"
                         if not bad else f"MISMATCH in {bad[:8]}")
    return out


def other_kernels(ctx, wl, ext):
    """Timings (HIP events on the launch stream, mean of 5 launches) of the kernels BASELINE configs 2 / 3 / 5 name besides the step's own:
    block statistics (SAD / SSE / variance / Hadamard SATD) over every 8x8 ... 64x64 block of a 1080p 8-bit picture pair, the 10-bit
    psy-RD / facade batch over a 2160p pair, the dynamic-GOP detector's HME, and an RD launch with the transform-type mix of a preset-6
    tx_type_search (16x16: all 16 types, 32x32: the 4 of its set).  Each with its algorithmic bytes -> GB/s."""
    L = api.lib()
    out = {}
    s0 = wl.sets[0]

    def timeit(fn, n=5):
        with torch.cuda.stream(ext):
            fn()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
        ctx.sync(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def tiling(w, h, sizes):
        js = []
        for bw, bh in sizes:
            ys, xs = np.meshgrid(np.arange(0, h - bh + 1, bh), np.arange(0, w - bw + 1, bw), indexing="ij")
            j = np.zeros(ys.size, dtype=abi.BLOCK_JOB_DTYPE)
            j["src_offset"] = j["ref_offset"] = (ys.ravel() * w + xs.ravel()).astype(np.uint32)
            j["width"], j["height"] = bw, bh
            js.append(j)
        return np.concatenate(js)

    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()

    def hier(w, h, jobs):
        """the same tiling as hierarchical work: the full 64x64 regions as pyramid jobs (85 nested blocks each from one read of the region),
        the blocks outside them as plain jobs; slot of every flat job in the hierarchical output arrays (for the equality check)"""
        full_w, full_h = (w // 64) * 64, (h // 64) * 64
        ys, xs = (jobs["src_offset"] // w).astype(np.int64), (jobs["src_offset"] % w).astype(np.int64)
        inside = (xs + jobs["width"] <= full_w) & (ys + jobs["height"] <= full_h)
        plain = np.ascontiguousarray(jobs[~inside])
        ry, rx = np.meshgrid(np.arange(0, full_h, 64), np.arange(0, full_w, 64), indexing="ij")
        reg = np.zeros(ry.size, dtype=abi.BLOCK_JOB_DTYPE)
        reg["src_offset"] = reg["ref_offset"] = (ry.ravel() * w + rx.ravel()).astype(np.uint32)
        reg["width"] = reg["height"] = 64
        slot = np.zeros(len(jobs), np.int64)
        slot[~inside] = np.arange(len(plain))
        n = jobs["width"].astype(np.int64)
        level_base = {64: 0, 32: 1, 16: 5, 8: 21}
        k = (ys // 64) * (full_w // 64) + xs // 64
        per = 64 // np.maximum(n, 1)
        z = np.array([level_base.get(int(v), 0) for v in n]) + ((ys % 64) // np.maximum(n, 1)) * per + (xs % 64) // np.maximum(n, 1)
        slot[inside] = (len(plain) + abi.PYRAMID_BLOCKS * k + z)[inside]
        return plain, reg, slot

    def both(tag, w, h, jobs, desc_kw, fields, bytes_per_px, extra=None):
        """flat job list vs hierarchical jobs of the same tiling: timings, and the outputs must be identical"""
        n = len(jobs)
        t_j = dev(jobs)
        o = {name: torch.zeros(n * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda") for name, dt in fields}
        d = abi.BlockStatsDesc(n_jobs=n, jobs=t_j.data_ptr(), **desc_kw)
        if extra:
            d.pred_mode, d.compound_type = extra[0].data_ptr(), extra[1].data_ptr()
        for name, _ in fields:
            setattr(d, name, o[name].data_ptr())
        ms_flat = timeit(lambda: ctx.check(L.svt_hip_block_stats_batch(ctx._h, C.byref(d)), "block_stats"))
        plain, reg, slot = hier(w, h, jobs)
        nh = len(plain) + abi.PYRAMID_BLOCKS * len(reg)
        t_p, t_r = dev(plain if len(plain) else np.zeros(1, abi.BLOCK_JOB_DTYPE)), dev(reg)
        oh = {name: torch.zeros(nh * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda") for name, dt in fields}
        dh = abi.BlockStatsDesc(n_jobs=len(plain), jobs=t_p.data_ptr(), n_pyramids=len(reg), pyramid_out_base=len(plain), pyramids=t_r.data_ptr(), **desc_kw)
        if extra:  # the per-candidate arrays follow the outputs' slot order
            sl = torch.from_numpy(slot).cuda()
            em = [torch.zeros(nh, dtype=torch.uint8, device="cuda") for _ in extra]
            for dst, src_t in zip(em, extra):
                dst[sl] = src_t
            dh.pred_mode, dh.compound_type = em[0].data_ptr(), em[1].data_ptr()
        for name, _ in fields:
            setattr(dh, name, oh[name].data_ptr())
        ms = timeit(lambda: ctx.check(L.svt_hip_block_stats_batch(ctx._h, C.byref(dh)), "block_stats"))
        same = all(np.array_equal(oh[name].cpu().numpy().view(dt)[slot], o[name].cpu().numpy().view(dt)) for name, dt in fields)
        px = int((jobs["width"].astype(np.int64) * jobs["height"]).sum())
        uniq = w * h * bytes_per_px // 4  # every sample of the two planes once
        out[tag] = {"ms": round(ms, 4), "flat_ms": round(ms_flat, 4), "speedup_vs_flat_jobs": round(ms_flat / ms, 2), "outputs_identical_to_flat_jobs": bool(same),
                    "gb_s": round(uniq / ms / 1e6, 1), "bytes": int(uniq), "flat_bytes": px * bytes_per_px // 4 * 1, "jobs": n, "pyramids": len(reg), "plain_jobs": len(plain),
                    "what": "hierarchical jobs: a wave per 64x64 region emits its 85 nested blocks from one read; `bytes` = the two planes once; flat = one job per block (every size re-reads its samples)"}

    # -- config 2: 1080p 8-bit block statistics incl. Hadamard
    y8 = synth.to_8bit(s0.y10_host[CUR:CUR + 2, :1080, :1920])
    jobs = tiling(1920, 1080, [(64, 64), (32, 32), (16, 16), (8, 8)])
    t_s, t_r = dev(y8[1]), dev(y8[0])
    both("block_stats_1080p8 (sad+sse+var+hadamard, 8x8..64x64)", 1920, 1080, jobs, dict(bit_depth=8, src_stride=1920, ref_stride=1920, src=t_s.data_ptr(), ref=t_r.data_ptr()),
         list(abi.STATS_OUT_FIELDS), 8)
    # -- config 5: 2160p 10-bit psy-RD + facade
    jobs = tiling(W, H, [(64, 64), (32, 32), (16, 16), (8, 8)])
    n = len(jobs)
    rng = np.random.default_rng(5)
    t_mode, t_comp = dev(rng.integers(0, 25, n).astype(np.uint8)), dev(rng.integers(0, 4, n).astype(np.uint8))
    fields = [f for f in list(abi.STATS_OUT_FIELDS) + list(abi.PSY_OUT_FIELDS) + list(abi.FACADE_OUT_FIELDS) + list(abi.VAR10_OUT_FIELDS) if f[0] != "satd"]
    both("psy_rd_2160p10 (sse+var10+psy+facade, 8x8..64x64)", W, H, jobs,
         dict(bit_depth=10, src_stride=W, ref_stride=W, src=s0.src10.data_ptr(), ref=s0.y10[CUR - 1].data_ptr(), psy_rd=1.35, temporal_layer_index=2, spy_rd=1), fields, 16,
         extra=(t_mode, t_comp))
    # -- dynamic-GOP detector HME on a 2160p pair
    m = torch.zeros(64, dtype=torch.uint8, device="cuda")
    a, b = s0.pics[CUR], s0.pics[CUR - 1]
    ms = timeit(lambda: ctx.dg_detector_hme_level0_async(a, b, W, H, 5, m.data_ptr()))
    out["dg_detector_hme_2160p"] = {"ms": round(ms, 4), "gb_s": round(2 * (W // 4) * (H // 4) / ms / 1e6, 2), "bytes": 2 * (W // 4) * (H // 4),
                                    "t_absdiff_per_s": round(wl.w64 * wl.h64 * 128 * 128 * 256 / ms / 1e9, 1)}
    # -- RD with a preset-6 transform-type mix on one 2160p picture
    mix = []
    for ts, types in ((2, list(range(16))), (3, [0, 9, 10, 11])):
        base = rd.grid_jobs(W, H, W, ts)
        js = []
        for t in types:
            j = base.copy(); j["tx_type"] = t
            js.append(j)
        jobs = np.concatenate(js)
        n = len(jobs)
        npk = min(abi.TX_W[ts], 32) * min(abi.TX_H[ts], 32)
        t_j = dev(jobs)
        o = {name: torch.zeros(n * k * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda") for name, dt, k in abi.RD_OUT_FIELDS}
        o["qcoeff"] = torch.zeros(n * npk * 4, dtype=torch.uint8, device="cuda")
        dd = abi.RdBatchDesc(bit_depth=10, quant_kind=0, tx_size=ts, n_jobs=n, src_stride=W, pred_stride=W, src=s0.src10.data_ptr(), pred=s0.y10[CUR - 1].data_ptr(),
                             recon=None, jobs=t_j.data_ptr(), quant_rows=wl.rows.data_ptr(), n_quant_rows=1)
        for name, t in o.items():
            setattr(dd, name, t.data_ptr())
        ms = timeit(lambda: ctx.check(L.svt_hip_rd_batch(ctx._h, C.byref(dd)), "rd_batch"))
        px = n * abi.TX_W[ts] * abi.TX_H[ts]
        by = px * 4 + n * npk * 4  # source + prediction read, quantized coefficients written (no reconstruction plane: the types of a block share it)
        out[f"rd_tx_type_mix_{abi.TX_W[ts]}x{abi.TX_H[ts]} ({len(types)} types x one 2160p picture)"] = {"ms": round(ms, 4), "gb_s": round(by / ms / 1e6, 1), "bytes": by, "jobs": n,
                                                                                                   "mpixels_per_s": round(px / ms / 1e3, 1)}
        mix.append(o)
    return out


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:  # the launcher's world size wins (the parent of spawn_ranks_if_needed starts exactly --gpus ranks)
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
        dist.init_process_group("gloo")  # bootstrap and host barriers only: the data path is RCCL (Exchange)
    ctx = api.Context(local_rank)
    ext = torch.cuda.ExternalStream(ctx.stream)
    # SVT_BENCH_EMULATE_RANK="r/N" (development aid, never set by the driver): a single process does the share of rank r of N
    # -- its row band, its job lists -- without any exchange, to read a rank's compute time at N GPUs off a one-GPU box
    emu = os.environ.get("SVT_BENCH_EMULATE_RANK")
    wl = Workload(ctx, *(map(int, emu.split("/")) if emu and world == 1 else (rank, world)))
    xch = Exchange(ctx, wl, rank, world, dist, rehearsal, ext) if world > 1 else None
    step_no = [0]
    wl.last_k0 = 0

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    def run(steps, ev=None, before=None, after=None):
        with torch.cuda.stream(ext):
            for _ in range(steps):
                k, s = step_no[0] % 2, step_no[0] % N_SETS
                step_no[0] += 1
                if s == 0:
                    wl.last_k0 = k
                if xch:
                    xch.before_step(k)
                if before:
                    before(s, k)
                wl.step(s, ev, (lambda k=k: xch.run(k)) if xch else None, k)
                if after:
                    after(s, k)
        if xch:
            xch.finish()

    run(a.warmup)
    barrier()
    t0 = time.perf_counter()
    run(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    # kernel-level durations (HIP events on the launch stream), separate short pass so events do not perturb the timed loop
    ev = {"me": [], "pred": [], "rd": [], **{f"rd{abi.TX_W[ts]}": [] for ts in RD_SIZES}}
    run(max(N_SETS, min(a.steps, 6)), ev)
    barrier()
    # per kernel family: the median over the pass's launches (one disturbed launch -- a clock ramp, a neighbour on the PCIe switch -- would
    # otherwise move a mean of six by a multiple)
    kms = {k: float(np.median([s.elapsed_time(e) for s, e in v])) for k, v in ev.items()}
    # the kernels of the ME launch one by one: the library brackets them with events of its own on the launch stream when asked to
    # (svt_hip_context_set_me_timing; off in the timed loop)
    chain = []
    if wl.live:
        ctx.set_me_timing(True)
        for _ in range(max(N_SETS, 3)):
            run(1)
            chain.append(ctx.me_launch_times())
        ctx.set_me_timing(False)
        barrier()
    chain_ms = {k: float(np.median([c.get(k, 0.0) for c in chain])) for k in (chain[0] if chain else {})}
    gather_check = None
    if world > 1:
        while (step_no[0] - 1) % N_SETS != 0:
            run(1)
        barrier()
    if world > 1 and rank == 0:
        # outside the timed region, on the real RCCL path as in the rehearsal: the gathered buffers of all ranks, unpacked, must equal a
        # whole-picture run of this rank (two pictures with different band rotations)
        k = wl.last_k0
        for pi in (0, 5):
            pic = PICS[pi]
            whole_desc = abi.MePictureDesc.from_buffer_copy(bytes(wl.descs[pic]))
            whole_desc.b64_row_start, whole_desc.b64_row_count = 0, 0
            whole = ctx.me_picture(wl.cfgs[pic], whole_desc, wl.sets[0].pics[pic[0]], wl.sets[0].refs(pic), search_level=False)
            merged = wl.layout.unpack(xch.out[k].cpu().numpy(), pi)
            bad = [kk for kk in merged if not np.array_equal(np.asarray(whole[kk]).reshape(merged[kk].shape), merged[kk])]
            log(f"{'rehearsal' if rehearsal else xch.kind}: gathered results of picture {pi} " + ("MATCH a whole-picture run" if not bad else f"DIFFER in {bad}"))
            assert not bad, bad
        gather_check = "gathered results of pictures 0 and 5 (all ranks' bands, unpacked) == a whole-picture run on rank 0"
    rccl_ranks = None
    if xch and xch.comm:
        n = C.c_int(0)
        ctx.check(api.lib().svt_hip_comm_count(xch.comm, C.byref(n)), "svt_hip_comm_count")
        rccl_ranks = int(n.value)
    # ---- end-to-end leg: the PCIe copies inside the step (4 new input pictures in, the gathered results out) ----
    e2e = None
    if not a.no_extras:
        NP = len(PICS)
        pool = {s: [wl.sets[s].pics[c] for c in CURS] for s in range(N_SETS)}
        pinned = []
        for s in range(N_SETS):
            row = []
            for c in CURS:
                buf, stride, p, w, h = (wl.sets[s].pics[c].download(2), None, 68, W, H)
                t = torch.from_numpy(buf).pin_memory()
                row.append((t, abi.PlaneDesc(t.data_ptr(), t.shape[1], 68, 68, W, H)))
            pinned.append(row)
        host_out = [torch.zeros(((world if xch else 1) * wl.me_bufs[0].numel()), dtype=torch.uint8).pin_memory() for _ in range(2)]
        L = api.lib()

        L.svt_hip_context_transfer_stream.restype = C.c_void_p
        io = torch.cuda.ExternalStream(L.svt_hip_context_transfer_stream(ctx._h))
        copied = [None, None]  # the D2H copy out of result buffer k has finished

        def before(s, k):
            # H2D, one step ahead, on the context's transfer stream: while this step computes, the NEXT step's four current pictures arrive from
            # the host and their 1/4 and 1/16 planes are rebuilt (svt_hip_pa_picture_update_ahead orders the refill behind the steps enqueued so
            # far; the next step's ME launch waits for the pictures' events)
            nxt = (s + 1) % N_SETS
            for pic, (t, desc) in zip(pool[nxt], pinned[nxt]):
                ctx.check(L.svt_hip_pa_picture_update_ahead(ctx._h, pic._h, C.byref(desc), 0), "svt_hip_pa_picture_update_ahead")
            if copied[k] is not None:  # this step's ME launch overwrites result buffer k: its last copy to the host must be through
                ext.wait_event(copied[k])

        def after(s, k):  # D2H on the transfer stream: the (gathered) per-b64 results go back to the host's entropy coder / mode decision
            src = xch.out[k] if xch else wl.me_bufs[k]
            if xch:
                xch.before_step(k)  # the copy reads what the exchange delivers
            done = torch.cuda.Event()
            done.record(ext)
            with torch.cuda.stream(io):
                io.wait_event(done)
                host_out[k].copy_(src, non_blocking=True)
                copied[k] = torch.cuda.Event()
                copied[k].record(io)

        def barrier_io():
            io.synchronize()
            barrier()

        run(3, before=before, after=after)
        barrier_io()
        t1 = time.perf_counter()
        run(a.steps, before=before, after=after)
        barrier_io()
        dt2 = time.perf_counter() - t1
        t_all2 = torch.tensor([dt2], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t_all2, op=dist.ReduceOp.MAX)
        dt2 = float(t_all2.item())
        h2d = sum(t.numel() for t, _ in pinned[0])
        e2e = {"value": round(a.steps * NP * W * H / dt2 / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(dt2 / a.steps * 1e3, 4), "h2d_bytes_per_step": int(h2d),
               "d2h_bytes_per_step": int(host_out[0].numel()),
               "what": "the same step with 4 new padded 8-bit input planes uploaded from page-locked host memory (pyramid levels rebuilt on the device) and the gathered ME results copied back, per step; "
                       "the copies run one step ahead / behind on the context's transfer stream (svt_hip_pa_picture_update_ahead)"}
    extras = other_kernels(ctx, wl, ext) if (not a.no_extras and world == 1 and rank == 0) else None
    # leave the results of a step on picture set 0 behind for the parity leg (its ME results, prediction / reconstruction planes, RD outputs)
    while (step_no[0] - 1) % N_SETS != 0:
        run(1)
    barrier()
    t_all = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    dt = float(t_all.item())
    pictures = a.steps * len(PICS)
    value = pictures * W * H / dt / 1e6
    if rank == 0:
        frac_rows = wl.rows_per_step / (wl.h64 * len(PICS))
        me_bytes = (1.3125 * (1 + R) + 0.166 * R) * W * H * frac_rows * len(PICS)  # B_ME bytes per pixel x the pictures of one launch
        dom = "me" if kms["me"] >= max(kms[f"rd{abi.TX_W[ts]}"] for ts in RD_SIZES) else "rd"  # the single kernel with the longest launch
        # HBM bytes per launch from the PMC passes committed under profiles/ (same command, N = 1): rocprofv3 cannot run
        # inside this process, so the figure is the recorded one; null when it does not describe this run
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "r03_hbm_traffic.json")
        if world == 1 and os.path.exists(tf):
            traffic = json.load(open(tf))["kernels"].get("me_launch", {}).get("hbm_bytes_per_launch")
            traffic_src = "profiles/r03_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; the kernels of the ME launch summed)"
        ach = me_bytes / (kms["me"] * 1e-3) / 1e9
        rd_roof = {}
        for ts, descs, _, _, n, _ in wl.rd:
            px, npk = n * abi.TX_W[ts] * abi.TX_H[ts], min(abi.TX_W[ts], 32) * min(abi.TX_H[ts], 32)
            by = px * 6 + n * npk * 4
            ms = kms[f"rd{abi.TX_W[ts]}"]
            rd_roof[f"rd_tx_kernel<{abi.TX_W[ts]}x{abi.TX_H[ts]}, 10>"] = {"avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(by), "achieved": round(by / ms / 1e6, 1),
                                                                      "frac": round(by / ms / 1e6 / HBM_PEAK_GBS, 4)}
        out = {
            "metric": "ME+RD-cost Mpixels/s (2160p10 preset-6)", "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8 SAD / i32 transforms", "data": "synthetic",
            "config": {"workload": "3840x2160 10-bit synthetic pan sequence, preset 6 (M6) search controls at qp 35; step = 16 pictures (4 current pictures x ref distance 1,2,4,8; R=2): "
                                   "open-loop ME of all 2040 b64 + full-pel pred + RD chain (64x64,32x32,16x16 DCT_DCT, 10-bit, b quantizer; qcoeff + recon + scalars written); "
                                   f"{N_SETS} picture sets alternate from step to step",
                       "pictures_per_step": len(PICS), "b64_rows_per_rank_per_step": wl.rows_per_step,
                       "parallelism": f"b64-row bands x{world} (left-over rows rotating over the ranks) + all-gather of ME results" + (f" via {xch.kind}" if xch else "")},
            "roofline": {"bound": "hbm", "kernel": "ME launch = " + " -> ".join(k.replace("svt_hip_me_", "").replace("_kernel", "") for k in chain_ms) + " (svt_hip_me_*_kernel, one stream)",
                         "kernel_chain_ms": {k: round(v, 4) for k, v in chain_ms.items()}, "achieved": round(ach, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(me_bytes), "avg_launch_ms": round(kms["me"], 4),
                         "note": "the ME of a step is ONE launch of the C entry = a chain of kernels on one stream (dense pre-pass, then the per-block pipeline staged at its searches); "
                                 "avg_launch_ms brackets the chain (HIP events on the launch stream), kernel_chain_ms each kernel (events the library records between them in a "
                                 "separate pass; profiles/r03_kernel_stats_bench.csv has rocprofv3's averages of the same kernels).  By the task's rule the bound is HBM; the search "
                                 "is issue-bound (SURVEY 8d): see valu_sad below", "longest_kernel": dom},
            "rd_roofline": rd_roof,
            # the longest SINGLE kernel of the step, whatever launch it belongs to (the ME launch is a chain of kernels)
            "dominant_single_kernel": max([{"name": k, "avg_launch_ms": v["avg_launch_ms"], "bound": "hbm", "achieved": v["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": v["frac"], "algorithmic_bytes_per_launch": v["algorithmic_bytes_per_launch"]} for k, v in rd_roof.items()]
                                          + [{"name": k, "avg_launch_ms": round(v, 4), "bound": "valu (v_qsad_pk_u16_u8 issue)", "note": "one kernel of the ME chain: see roofline / roofline.valu_sad for the launch"}
                                             for k, v in chain_ms.items()], key=lambda e: e["avg_launch_ms"]) if (rd_roof or chain_ms) else None,
            "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
        }
        if xch:
            out["exchange"] = {"kind": xch.kind, "rccl_ranks": rccl_ranks, "bytes_per_step_all_ranks": int(sum(wl.layout.rank_bytes)), "uniform_bands": bool(wl.layout.uniform),
                               "check": gather_check}
        if e2e:
            out["end_to_end"] = e2e
        if extras:
            out["other_kernels"] = extras
        if not a.no_cpu_baseline and world == 1:
            cb = cpu_baseline(wl)
            out["cpu_baseline"] = cb
            # SURVEY 8d's diagnostic for the search kernel: |a-b| evaluated per second against the chip's packed-SAD issue rate
            ops = cb["sad_ops_per_b64"] * wl.w64 * wl.h64 * len(PICS) * frac_rows
            out["roofline"]["valu_sad"] = {"absdiff_per_launch": int(ops), "achieved_tops": round(ops / (kms["me"] * 1e-3) / 1e12, 2), "peak_tops": QSAD_PEAK_TOPS,
                                           "frac": round(ops / (kms["me"] * 1e-3) / 1e12 / QSAD_PEAK_TOPS, 4),
                                           "what": "|a-b| evaluations of the reference algorithm's searches (counted by the oracle on the sample rows, scaled to the launch) / ME launch time, "
                                                   "against the measured chip-wide v_qsad_pk_u16_u8 issue rate x 16"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if xch:
            xch.close()
        dist.destroy_process_group()
    # No svt_hip_context_destroy here: torch's page-locked allocator still holds events recorded on the context's stream (the end-to-end
    # leg's copies) and releases them at interpreter exit; the process ends right away and the HIP runtime tears the context down.
    ctx.sync()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
