"""Shared pieces of the full-pel refinement search tests (svt_pme_sad_loop_kernel): job grids after test/SadTest.cc:1580-1640 (large and
small blocks, search areas, sparse steps, random 16-bit component costs) and the three runners (reference build, oracle, HIP)."""
import ctypes as C

import numpy as np

from svt_av1_psyex_amd import abi

MV_CENTRE = 1 << 14  # tables of 2 * MV_CENTRE + 1 entries: every index the reference's clamp (MV_LOW .. MV_UPP) can produce is valid


def cost_tables(rng):
    """(mvjcost[4], row table, column table) with the centre at index MV_CENTRE"""
    j = np.array([11, 54, 5437, 342], np.int32)
    return j, rng.integers(0, 1 << 16, 2 * MV_CENTRE + 1).astype(np.int32), rng.integers(0, 1 << 16, 2 * MV_CENTRE + 1).astype(np.int32)


def random_jobs(rng, plane_w, plane_h, n, wild=False):
    """n searches inside a plane_w x plane_h reference plane; wild: base vectors and start positions over the whole int16 range (the
    reference's own unit test), otherwise values a mode-decision call site produces (small refinements around the candidate)"""
    sizes = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 16), (16, 8), (32, 16), (16, 64), (64, 128), (128, 128), (128, 64), (4, 16), (64, 16)]
    jobs = np.zeros(n, abi.PME_JOB_DTYPE)
    for i in range(n):
        bw, bh = sizes[rng.integers(len(sizes))]
        step = int(rng.choice([1, 1, 2, 3, 8]))
        sa_w = int(rng.choice([8, 16, 24, 40])); sa_h = int(rng.integers(1, 24))
        sa_w = min(sa_w, (plane_w - bw - 8) // 8 * 8); sa_h = max(1, min(sa_h, plane_h - bh - 1))
        x0, y0 = rng.integers(0, plane_w - bw - sa_w - 3), rng.integers(0, plane_h - bh - sa_h + 1)
        sx, sy = rng.integers(0, plane_w - bw + 1), rng.integers(0, plane_h - bh + 1)
        j = jobs[i]
        j["src_offset"], j["ref_offset"], j["width"], j["height"] = sy * plane_w + sx, y0 * plane_w + x0, bw, bh
        j["sa_w"], j["sa_h"], j["step"] = sa_w, sa_h, step
        if wild:
            j["mvx"], j["mvy"], j["start_x"], j["start_y"] = rng.integers(-32768, 32768, 4)
            j["ref_mv"] = (23, 76)
        else:
            j["mvx"], j["mvy"] = rng.integers(-400, 401, 2) * 8
            j["start_x"], j["start_y"] = -(sa_w // 2), -(sa_h // 2)
            j["ref_mv"] = rng.integers(-300, 301, 2)
        j["best_cost"] = int(rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 5000, 200000, 0]))
        j["best_mvx"], j["best_mvy"] = rng.integers(-100, 100, 2)
    return jobs


def run_oracle(oracle, src, ref, jobs, cost_type, epb, tables):
    jc, tr, tc = tables
    n = len(jobs)
    cost, mv = np.zeros(n, np.uint32), np.zeros((n, 2), np.int16)
    d = abi.PmeBatchDesc(n_jobs=n, src_stride=src.shape[1], ref_stride=ref.shape[1], src=src.ctypes.data, ref=ref.ctypes.data, jobs=jobs.ctypes.data, mv_cost_type=cost_type,
                         error_per_bit=epb, mvjcost=jc.ctypes.data, best_cost=cost.ctypes.data, best_mv=mv.ctypes.data)
    d.mvcost[0], d.mvcost[1] = tr.ctypes.data + 4 * MV_CENTRE, tc.ctypes.data + 4 * MV_CENTRE
    assert oracle.orc_pme_sad_batch(C.byref(d)) == 0
    return cost, mv


def run_ref(ref_lib, src, ref, jobs, cost_type, epb, tables, garbage=None, fn="svt_pme_sad_loop_kernel_c"):
    """the reference's svt_pme_sad_loop_kernel_c (or another function with its prototype), call by call; garbage: byte the parameter struct
    is filled with before its fields are assigned (the padding bytes keep it)"""
    jc, tr, tc = tables
    n = len(jobs)
    cost, mv = np.zeros(n, np.uint32), np.zeros((n, 2), np.int16)
    for i, j in enumerate(jobs):
        rmv = abi.Mv(int(j["ref_mv"][0]), int(j["ref_mv"][1]))
        p = abi.MvCostParam()
        if garbage is not None:
            C.memset(C.byref(p), garbage, C.sizeof(p))
        p.ref_mv, p.mv_cost_type, p.mvjcost, p.error_per_bit = C.pointer(rmv), cost_type, jc.ctypes.data, epb
        p.mvcost[0], p.mvcost[1] = tr.ctypes.data + 4 * MV_CENTRE, tc.ctypes.data + 4 * MV_CENTRE
        bc, bx, by = C.c_uint32(int(j["best_cost"])), C.c_int16(int(j["best_mvx"])), C.c_int16(int(j["best_mvy"]))
        getattr(ref_lib, fn)(C.byref(p), C.c_void_p(src.ctypes.data + int(j["src_offset"])), C.c_uint32(src.shape[1]),
                                          C.c_void_p(ref.ctypes.data + int(j["ref_offset"])), C.c_uint32(ref.shape[1]), C.c_uint32(int(j["height"])), C.c_uint32(int(j["width"])),
                                          C.byref(bc), C.byref(bx), C.byref(by), C.c_int16(int(j["start_x"])), C.c_int16(int(j["start_y"])), C.c_int16(int(j["sa_w"])),
                                          C.c_int16(int(j["sa_h"])), C.c_int16(int(j["step"])), C.c_int16(int(j["mvx"])), C.c_int16(int(j["mvy"])))
        cost[i], mv[i] = bc.value, (bx.value, by.value)
    return cost, mv


def run_hip(ctx, src, ref, jobs, cost_type, epb, tables):
    import torch
    from svt_av1_psyex_amd import api
    jc, tr, tc = tables
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
    n = len(jobs)
    t = [dev(src), dev(np.concatenate([ref.reshape(-1), np.zeros(64, np.uint8)])), dev(jobs), dev(jc), dev(tr), dev(tc)]
    cost, mv = torch.zeros(n * 4, dtype=torch.uint8, device="cuda"), torch.zeros(n * 4, dtype=torch.uint8, device="cuda")
    d = abi.PmeBatchDesc(n_jobs=n, src_stride=src.shape[1], ref_stride=ref.shape[1], src=t[0].data_ptr(), ref=t[1].data_ptr(), jobs=t[2].data_ptr(), mv_cost_type=cost_type,
                         error_per_bit=epb, mvjcost=t[3].data_ptr(), best_cost=cost.data_ptr(), best_mv=mv.data_ptr())
    d.mvcost[0], d.mvcost[1] = t[4].data_ptr() + 4 * MV_CENTRE, t[5].data_ptr() + 4 * MV_CENTRE
    torch.cuda.synchronize()
    ctx.check(api.lib().svt_hip_pme_sad_batch(ctx._h, C.byref(d)), "svt_hip_pme_sad_batch")
    ctx.sync()
    return cost.cpu().numpy().view(np.uint32), mv.cpu().numpy().view(np.int16).reshape(n, 2)
