"""Host-side helpers of the batched block-statistics entry (svt_hip_block_stats_batch)."""
import ctypes as C

import numpy as np

from . import abi


def random_jobs(rng, plane_w, plane_h, n, sizes=None, square_only=False, subpel=False):
    """n block jobs with AV1 block shapes at random positions inside a plane_w x plane_h plane."""
    sizes = sizes or [(w, h) for w in (4, 8, 16, 32, 64, 128) for h in (4, 8, 16, 32, 64, 128) if max(w, h) <= 4 * min(w, h)]
    if square_only:
        sizes = [s for s in sizes if s[0] == s[1]]
    jobs = np.zeros(n, dtype=abi.BLOCK_JOB_DTYPE)
    for i in range(n):
        w, h = sizes[rng.integers(len(sizes))]
        w, h = min(w, plane_w), min(h, plane_h)
        x0, y0 = rng.integers(0, plane_w - w + 1 - int(subpel)), rng.integers(0, plane_h - h + 1 - int(subpel))  # sub-pel reads one more row / column
        x1, y1 = rng.integers(0, plane_w - w + 1), rng.integers(0, plane_h - h + 1)
        jobs[i] = (y0 * plane_w + x0, y1 * plane_w + x1, w, h, 0, 0)
    if subpel:
        jobs["subpel_x"] = rng.integers(0, 8, n)
        jobs["subpel_y"] = rng.integers(0, 8, n)
    return jobs


def expand_pyramid(region, src_stride, ref_stride):
    """the 85 plain jobs a hierarchical 64x64 region stands for, in its output-slot order: 64x64, 4 x 32x32, 16 x 16x16, 64 x 8x8, each level
    in raster order (SvtHipBlockStatsDesc.pyramids)"""
    out = np.zeros(abi.PYRAMID_BLOCKS, dtype=abi.BLOCK_JOB_DTYPE)
    k = 0
    for n in (64, 32, 16, 8):
        for y in range(0, 64, n):
            for x in range(0, 64, n):
                out[k] = (int(region["src_offset"]) + y * src_stride + x, int(region["ref_offset"]) + y * ref_stride + x, n, n, 0, 0)
                k += 1
    return out


def run_hip(ctx, src, ref, jobs, bit_depth, satd=True, psy_rd=None, facade=None, pyramids=None):
    """facade: dict(pred_mode=u8[n], compound_type=u8[n], temporal_layer_index=int, spy_rd=int) -> also `facade_dist`.
    pyramids: optional array of 64x64 region jobs; their 85 outputs each follow the plain jobs' (slots len(jobs) + 85 k ...; the facade
    arrays then cover those slots too)."""
    import torch
    from . import api
    L = api.lib()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
    n_plain = len(jobs)
    n = n_plain + (abi.PYRAMID_BLOCKS * len(pyramids) if pyramids is not None else 0)
    t_src, t_ref, t_jobs = dev(src), dev(ref), dev(jobs if n_plain else np.zeros(1, abi.BLOCK_JOB_DTYPE))
    fields = list(abi.STATS_OUT_FIELDS) + (list(abi.PSY_OUT_FIELDS) if psy_rd is not None else []) + (list(abi.FACADE_OUT_FIELDS) if facade else []) + (list(abi.VAR10_OUT_FIELDS) if bit_depth == 10 else [])
    outs = {name: torch.zeros(n * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda") for name, dt in fields}
    d = abi.BlockStatsDesc(bit_depth=bit_depth, n_jobs=n_plain, src_stride=src.shape[1], ref_stride=ref.shape[1])
    if pyramids is not None and len(pyramids):
        t_pyr = dev(pyramids)
        d.n_pyramids, d.pyramid_out_base, d.pyramids = len(pyramids), n_plain, t_pyr.data_ptr()
    if psy_rd is not None:
        d.psy_rd = psy_rd
        for name, _ in abi.PSY_OUT_FIELDS:
            setattr(d, name, outs[name].data_ptr())
    if bit_depth == 10:
        d.variance10, d.var_sse10 = outs["variance10"].data_ptr(), outs["var_sse10"].data_ptr()
    if facade:
        t_mode, t_comp = dev(np.asarray(facade["pred_mode"], np.uint8)), dev(np.asarray(facade["compound_type"], np.uint8))
        d.pred_mode, d.compound_type, d.facade_dist = t_mode.data_ptr(), t_comp.data_ptr(), outs["facade_dist"].data_ptr()
        d.temporal_layer_index, d.spy_rd = facade["temporal_layer_index"], facade["spy_rd"]
    d.src, d.ref, d.jobs = t_src.data_ptr(), t_ref.data_ptr(), t_jobs.data_ptr()
    for name, _ in abi.STATS_OUT_FIELDS:
        if name == "satd" and not satd:
            continue
        setattr(d, name, outs[name].data_ptr())
    torch.cuda.synchronize()
    rc = L.svt_hip_block_stats_batch(ctx._h, C.byref(d))
    ctx.check(rc, "svt_hip_block_stats_batch")
    ctx.sync()
    res = {name: outs[name].cpu().numpy().view(dt) for name, dt in fields}
    if not satd:
        res.pop("satd")
    return res
