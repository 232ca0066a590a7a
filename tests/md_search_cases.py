"""Shared pieces of the mode-decision side motion-search tests (include/svt_hip_md_search.h): planes with real motion, job grids shaped like
the reference's call sites (md_nsq_motion_search's chain of md_full_pel_search calls, product_coding_loop.c:2260-2375; md_subpel_search's
parameters, :2637-2750 and the MdSubPelSearchCtrls levels of enc_mode_config.c) and the three runners (reference build, oracle, HIP)."""
import ctypes as C

import numpy as np

from svt_av1_psyex_amd import abi

MV_CENTRE = 1 << 14
PAD = 80          # EbPictureBufferDesc padding of the reference planes
W, H = 320, 192   # picture size (max_width / max_height)
SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 16), (16, 8), (32, 16), (16, 64), (64, 32), (128, 128), (4, 16), (64, 16), (8, 32)]


def planes(seed):
    """source picture (W x H) and a padded reference plane: the source displaced by a few samples + noise, so that searches have a minimum to find"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H + 2 * PAD + 16, W + 2 * PAD + 16)).astype(np.float32)
    k = np.ones(5, np.float32) / 5
    base = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, base)
    base = np.apply_along_axis(lambda c: np.convolve(c, k, "same"), 0, base)
    base = np.clip((base - 128) * 3 + 128, 0, 255)
    ref = np.ascontiguousarray(base[:H + 2 * PAD, :W + 2 * PAD]).astype(np.uint8)
    dy, dx = 3, -5
    src = np.clip(base[PAD + dy:PAD + dy + H, PAD + dx:PAD + dx + W] + rng.normal(0, 4, (H, W)), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(src), ref


def cost_tables(rng):
    j = np.array([11, 54, 5437, 342], np.int32)
    ramp = (np.abs(np.arange(-MV_CENTRE, MV_CENTRE + 1)) * 3 + 200).astype(np.int32)  # rate grows with the component, like a real table
    return j, (ramp + rng.integers(0, 64, ramp.size)).astype(np.int32), (ramp + rng.integers(0, 64, ramp.size)).astype(np.int32)


def fullpel_chain(rng, n_blocks, dist_type, psad):
    """md_nsq_motion_search's chain for n_blocks blocks, as ROUNDS of jobs (one batch per round, job i of every round = block i, results in place):
    two candidate centres (one position each, the second keeps the running best), the step-4 area around the winner (fresh best), +-2 step 2,
    +-1 step 1.  Returns [jobs_round0, jobs_round1, ...]."""
    rounds = []
    geo = []
    for i in range(n_blocks):
        bw, bh = SIZES[rng.integers(len(SIZES))]
        x, y = int(rng.integers(0, (W - bw) // 4 + 1)) * 4, int(rng.integers(0, (H - bh) // 4 + 1)) * 4
        if i % 7 == 0:  # blocks on the picture edge: the search-area adjustment bites
            x, y = [(0, 0), (W - bw, H - bh), (0, H - bh), (W - bw, 0)][(i // 7) % 4]
        geo.append((bw, bh, x, y, rng.integers(-60, 61, 2)))

    def base(i):
        bw, bh, x, y, rmv = geo[i]
        j = np.zeros(1, abi.FULLPEL_JOB_DTYPE)[0]
        j["src_offset"], j["blk_org_x"], j["blk_org_y"], j["width"], j["height"] = y * W + x, x, y, bw, bh
        j["dist_type"], j["flags"], j["ref_mv"], j["chain_from"] = dist_type, (abi.FP_ENABLE_PSAD if psad else 0), rmv, -1
        j["best_cost"], j["best_mvx"], j["best_mvy"] = 0xFFFFFFFF, -1, -1
        return j

    def make(fn):
        jobs = np.zeros(n_blocks, abi.FULLPEL_JOB_DTYPE)
        for i in range(n_blocks):
            j = base(i)
            fn(i, j)
            jobs[i] = j
        return jobs

    def cand0(i, j):
        j["mvx"], j["mvy"] = (rng.integers(-12, 13, 2) * 8)
        j["step"] = 1

    def cand1(i, j):
        j["mvx"], j["mvy"] = (rng.integers(-12, 13, 2) * 8)
        j["step"] = 1
        j["flags"] |= abi.FP_BEST_FROM_CHAIN
        j["chain_from"] = i

    def area(i, j):
        wdt, hgt = int(rng.choice([7, 15, 31])), int(rng.choice([5, 7, 15]))
        j["flags"] |= abi.FP_CENTRE_FROM_CHAIN
        j["chain_from"] = i
        j["start_x"], j["end_x"], j["start_y"], j["end_y"], j["step"] = -(wdt >> 1), wdt >> 1, -(hgt >> 1), hgt >> 1, 4

    def refine(r, st):
        def f(i, j):
            j["flags"] |= abi.FP_CENTRE_FROM_CHAIN | abi.FP_BEST_FROM_CHAIN
            j["chain_from"] = i
            j["start_x"], j["end_x"], j["start_y"], j["end_y"], j["step"] = -r, r, -r, r, st
            if st == 2:
                j["flags"] |= abi.FP_SPRS_LEV0_DONE
                j["sprs_lev0_start_x"], j["sprs_lev0_end_x"], j["sprs_lev0_start_y"], j["sprs_lev0_end_y"] = -3, 4, -2, 3
        return f

    return [make(cand0), make(cand1), make(area), make(refine(2, 2)), make(refine(1, 1))]


def fullpel_desc(src, ref, jobs, cost_type, epb, tables, cost, mv):
    jc, tr, tc = tables
    d = abi.FullpelBatchDesc(n_jobs=len(jobs), src_stride=src.shape[1], ref_stride=ref.shape[1], src=src.ctypes.data, ref=ref.ctypes.data, ref_org_x=PAD, ref_org_y=PAD,
                             ref_max_width=W, ref_max_height=H, jobs=jobs.ctypes.data, mv_cost_type=cost_type, error_per_bit=epb, mvjcost=jc.ctypes.data,
                             best_cost=cost.ctypes.data, best_mv=mv.ctypes.data)
    d.mvcost[0], d.mvcost[1] = tr.ctypes.data + 4 * MV_CENTRE, tc.ctypes.data + 4 * MV_CENTRE
    return d


def run_fullpel_cpu(fn, src, ref, rounds, cost_type, epb, tables):
    """fn = oracle.orc_md_fullpel_batch or ref.ref_md_fullpel_batch; the rounds share the output arrays (in-place chains)"""
    n = len(rounds[0])
    cost, mv = np.zeros(n, np.uint32), np.zeros((n, 2), np.int16)
    trace = []
    for jobs in rounds:
        d = fullpel_desc(src, ref, jobs, cost_type, epb, tables, cost, mv)
        assert fn(C.byref(d)) == 0
        trace.append((cost.copy(), mv.copy()))
    return trace


def run_fullpel_hip(ctx, src, ref, rounds, cost_type, epb, tables):
    import torch
    from svt_av1_psyex_amd import api
    L = api.lib()
    ext = torch.cuda.ExternalStream(ctx.stream)
    jc, tr, tc = tables
    n = len(rounds[0])
    trace = []
    with torch.cuda.stream(ext):
        t = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.uint8).reshape(-1)).cuda() for k, v in dict(src=src, ref=ref, jc=jc, tr=tr, tc=tc).items()}
        cost, mv = torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(2 * n, dtype=torch.int16, device="cuda")
        for jobs in rounds:
            tj = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).cuda()
            d = abi.FullpelBatchDesc(n_jobs=n, src_stride=src.shape[1], ref_stride=ref.shape[1], src=t["src"].data_ptr(), ref=t["ref"].data_ptr(), ref_org_x=PAD, ref_org_y=PAD,
                                     ref_max_width=W, ref_max_height=H, jobs=tj.data_ptr(), mv_cost_type=cost_type, error_per_bit=epb, mvjcost=t["jc"].data_ptr(),
                                     best_cost=cost.data_ptr(), best_mv=mv.data_ptr())
            d.mvcost[0], d.mvcost[1] = t["tr"].data_ptr() + 4 * MV_CENTRE, t["tc"].data_ptr() + 4 * MV_CENTRE
            ctx.check(L.svt_hip_md_fullpel_batch(ctx._h, C.byref(d)), "svt_hip_md_fullpel_batch")
            ctx.sync()
            trace.append((cost.cpu().numpy().view(np.uint32).copy(), mv.cpu().numpy().reshape(n, 2).copy()))
    return trace


# MdSubPelSearchCtrls-like settings: (allow_hp, forced_stop, iters_per_step, pred_variance_th, abs_th_mult, round_dev_th, skip_diag_refinement, bias_fp, mv_cost_type
#                                       [, search_method, subpel_search_type, mvp_th, hp_mv_th])
SUBPEL_SETTINGS = [
    (0, 0, 2, 0, 0, 2147483647, 0, 0, 0),        # everything on: quarter-pel (no hp), two levels per step, entropy MV cost
    (1, 0, 2, 0, 0, 2147483647, 1, 0, 0),        # eighth-pel, diagonal refinement only when the cardinal points improved
    (0, 1, 1, 50, 2, -25, 2, 0, 0),              # quarter-pel, one level, variance / absolute thresholds, early round exit
    (0, 2, 2, 0, 0, 2147483647, 3, 100, 4),      # half-pel only, MV_COST_OPT with its early exit, full-pel bias
    (1, 0, 2, 0, 0, 2147483647, 4, 110, 4),      # org_error 0: never the diagonal / second level
    (0, 3, 2, 0, 0, 2147483647, 0, 0, 0),        # forced_stop FULL_PEL: the centre error only
    # svt_av1_find_best_sub_pixel_tree (search_method 1): the accurate search on svt_aom_upsampled_pred
    (1, 0, 2, 0, 0, 2147483647, 0, 0, 0, 1, 3, 0, 0),    # eighth-pel, 8 taps, two levels per step
    (0, 0, 2, 0, 0, 2147483647, 0, 0, 0, 1, 2, 0, 0),    # quarter-pel, 4 taps
    (1, 0, 1, 30, 1, 2147483647, 0, 105, 4, 1, 1, 0, 0),  # 2 taps (bilinear table), one level, variance / absolute thresholds, full-pel bias, MV_COST_OPT
    (1, 0, 2, 0, 0, 2147483647, 0, 0, 0, 1, 3, 20, 16),   # the PD_PASS_1 branch: round limited by the distance to the best MVP (mvp_th, hp_mv_th)
    (0, 2, 2, 0, 0, 2147483647, 0, 0, 0, 1, 3, 0, 0),     # half-pel only
]


def full_setting(setting):
    return tuple(setting) + (0, 0, 0, 0)[len(setting) - 9:] if len(setting) < 13 else tuple(setting)


def subpel_jobs(rng, n):
    jobs = np.zeros(n, abi.SUBPEL_JOB_DTYPE)
    for i in range(n):
        bw, bh = SIZES[rng.integers(len(SIZES))]
        x, y = int(rng.integers(8, (W - bw - 8) // 4)) * 4, int(rng.integers(8, (H - bh - 8) // 4)) * 4
        j = jobs[i]
        j["src_offset"], j["ref_offset"] = y * W + x, (y + PAD) * (W + 2 * PAD) + x + PAD
        j["width"], j["height"], j["log2_pels"] = bw, bh, int(np.log2(bw * bh))
        j["early_neigh_check_exit"] = 1 if i % 11 == 5 else 0
        smv = rng.integers(-6, 7, 2) * 8
        j["start_mv"] = smv
        j["ref_mv"] = smv + rng.integers(-30, 31, 2)
        lim = int(rng.choice([9, 40, 2000]))  # tight limits cut the tree at the range test
        j["col_min"], j["col_max"], j["row_min"], j["row_max"] = smv[1] - lim, smv[1] + lim, smv[0] - lim, smv[0] + lim
        j["early_exit_th"] = 1020 - (max(bw, bh) >> 2)
        j["best_mvp_dist"] = int(rng.integers(0, 40000))  # (read by the tree search's PD_PASS_1 branch only)
        j["best_mvp"] = smv + rng.integers(-40, 41, 2)
    return jobs


def subpel_desc(src, ref, jobs, setting, epb, qp, tables, out):
    jc, tr, tc = tables
    hp, stop, iters, pvt, atm, rdt, sdr, bias, ctype, method, taps, mvp_th, hp_mv_th = full_setting(setting)
    d = abi.SubpelBatchDesc(n_jobs=len(jobs), src_stride=src.shape[1], ref_stride=ref.shape[1], src=src.ctypes.data, ref=ref.ctypes.data, jobs=jobs.ctypes.data, allow_hp=hp,
                            forced_stop=stop, iters_per_step=iters, pred_variance_th=pvt, abs_th_mult=atm, round_dev_th=rdt, skip_diag_refinement=sdr, bias_fp=bias, qp=qp,
                            search_method=method, subpel_search_type=taps, mvp_th=mvp_th, hp_mv_th=hp_mv_th,
                            mv_cost_type=ctype, error_per_bit=epb, mvjcost=jc.ctypes.data, best_mv=out["best_mv"].ctypes.data, besterr=out["besterr"].ctypes.data,
                            distortion=out["distortion"].ctypes.data, sse=out["sse"].ctypes.data, center_err=out["center_err"].ctypes.data)
    d.mvcost[0], d.mvcost[1] = tr.ctypes.data + 4 * MV_CENTRE, tc.ctypes.data + 4 * MV_CENTRE
    return d


def subpel_out(n):
    return {"best_mv": np.zeros((n, 2), np.int16), "besterr": np.zeros(n, np.uint32), "distortion": np.zeros(n, np.int32), "sse": np.zeros(n, np.uint32),
            "center_err": np.zeros(n, np.uint32)}


def run_subpel_cpu(fn, src, ref, jobs, setting, epb, qp, tables):
    out = subpel_out(len(jobs))
    assert fn(C.byref(subpel_desc(src, ref, jobs, setting, epb, qp, tables, out))) == 0
    return out


def run_subpel_hip(ctx, src, ref, jobs, setting, epb, qp, tables):
    import torch
    from svt_av1_psyex_amd import api
    L = api.lib()
    ext = torch.cuda.ExternalStream(ctx.stream)
    jc, tr, tc = tables
    hp, stop, iters, pvt, atm, rdt, sdr, bias, ctype, method, taps, mvp_th, hp_mv_th = full_setting(setting)
    n = len(jobs)
    with torch.cuda.stream(ext):
        t = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.uint8).reshape(-1)).cuda() for k, v in dict(src=src, ref=ref, jc=jc, tr=tr, tc=tc, jobs=jobs).items()}
        o = {"best_mv": torch.zeros(2 * n, dtype=torch.int16, device="cuda"), "besterr": torch.zeros(n, dtype=torch.int32, device="cuda"),
             "distortion": torch.zeros(n, dtype=torch.int32, device="cuda"), "sse": torch.zeros(n, dtype=torch.int32, device="cuda"),
             "center_err": torch.zeros(n, dtype=torch.int32, device="cuda")}
        d = abi.SubpelBatchDesc(n_jobs=n, src_stride=src.shape[1], ref_stride=ref.shape[1], src=t["src"].data_ptr(), ref=t["ref"].data_ptr(), jobs=t["jobs"].data_ptr(), allow_hp=hp,
                                forced_stop=stop, iters_per_step=iters, pred_variance_th=pvt, abs_th_mult=atm, round_dev_th=rdt, skip_diag_refinement=sdr, bias_fp=bias, qp=qp,
                                search_method=method, subpel_search_type=taps, mvp_th=mvp_th, hp_mv_th=hp_mv_th,
                                mv_cost_type=ctype, error_per_bit=epb, mvjcost=t["jc"].data_ptr(), best_mv=o["best_mv"].data_ptr(), besterr=o["besterr"].data_ptr(),
                                distortion=o["distortion"].data_ptr(), sse=o["sse"].data_ptr(), center_err=o["center_err"].data_ptr())
        d.mvcost[0], d.mvcost[1] = t["tr"].data_ptr() + 4 * MV_CENTRE, t["tc"].data_ptr() + 4 * MV_CENTRE
        ctx.check(L.svt_hip_md_subpel_batch(ctx._h, C.byref(d)), "svt_hip_md_subpel_batch")
        ctx.sync()
    return {"best_mv": o["best_mv"].cpu().numpy().reshape(n, 2), "besterr": o["besterr"].cpu().numpy().view(np.uint32), "distortion": o["distortion"].cpu().numpy(),
            "sse": o["sse"].cpu().numpy().view(np.uint32), "center_err": o["center_err"].cpu().numpy().view(np.uint32)}


FULLPEL_GRID = [(dist, psad, ctype) for dist in (0, 1) for psad in (0, 1) for ctype in (0, 4)]
