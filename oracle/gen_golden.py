"""Generates tests/golden/*.npz from the REFERENCE build (oracle/_ref/libsvtref.so).

Run in the build container (needs /root/reference to have been compiled by oracle/Makefile):
    python oracle/gen_golden.py
Each fixture stores the inputs (frames, descriptor bytes) and the outputs of the reference's own
svt_aom_motion_estimation_b64 / `_c` kernels, so that the GPU box -- which has no /root/reference --
can pin both the oracle and the HIP path.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import pyoracle  # noqa: E402
from me_cases import MeCase  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

ME_CASES = {
    # name: MeCase kwargs
    "me_cif_m6_b": dict(width=352, height=288, enc_mode=6),
    "me_cif_m12_b": dict(width=352, height=288, enc_mode=12, temporal_layer_index=3),
    "me_cif_m8_noise": dict(width=352, height=288, enc_mode=8, kind="noise"),
    "me_cif_m0_mrp": dict(width=352, height=288, enc_mode=0, cur=2, refs={(0, 0): 1, (0, 1): 0, (1, 0): 3, (1, 1): 4}, n_frames=5),
    "me_cif_m6_p_base": dict(width=352, height=288, enc_mode=6, refs={(0, 0): 0}, temporal_layer_index=0),
    "me_odd_m6_b": dict(width=360, height=296, enc_mode=6, seed=5),  # partial b64 columns/rows (40x40 edge blocks)
    "me_cif_m4_gm": dict(width=352, height=288, enc_mode=4, gm_enabled=1, kind="fastpan"),
    "me_cif_m10_sc": dict(width=352, height=288, enc_mode=10, sc_class1=1, kind="extremes"),
    # temporal-filter ME (ME_MCTF): one reference, early exit on part of the blocks; search-level outputs only
    "me_vga_m4_mctf": dict(width=640, height=360, enc_mode=4, refs={(0, 0): 1}, mctf_exit_th=6940, seed=3),
}


def gen_me(only=None):
    from me_cases import MCTF_OUTPUTS
    for name, kw in ME_CASES.items():
        if only and name not in only:
            continue
        c = MeCase(**kw)
        out = c.run_cpu("ref")
        chk = c.run_cpu("oracle")
        if "mctf_exit_th" in kw:
            out = {k: out[k] for k in MCTF_OUTPUTS}
        for k in out:
            assert np.array_equal(out[k], chk[k]), (name, k)
        frames = {"cur": c.cur.inner(2)}
        for (li, ri), p in c.refs.items():
            frames[f"ref_{li}_{ri}"] = p.inner(2)
        meta = dict(cfg=np.frombuffer(bytes(c.cfg), np.uint8), desc=np.frombuffer(bytes(c.desc), np.uint8))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **frames, **meta, **{"out_" + k: v for k, v in out.items()})
        print("wrote", name, {k: v.shape for k, v in list(out.items())[:3]})


def gen_sad_kernels():
    """Known-answer vectors for the SAD leaf kernels, produced by the reference `_c` functions
    (grid modelled on test/SadTest.cc:57-112,433-444: block sizes x search areas x patterns)."""
    ref = pyoracle.load_ref()
    rng = np.random.default_rng(2024)
    recs = []
    blocks = [(16, 8), (16, 16), (32, 16), (64, 32), (8, 8), (24, 12), (48, 24), (64, 64), (128, 128), (6, 4), (14, 7)]
    areas = [(1, 1), (8, 3), (16, 9), (24, 7), (48, 48), (5, 11), (96, 1)]
    for pattern in ("random", "ref_max", "src_max", "equal"):
        for (bw, bh) in blocks:
            for (sw, sh) in areas:
                for skip in (0, 1):
                    if skip and not (bw == 16 and bh <= 16):
                        continue
                    stride = bw + sw + 13
                    rows = bh + sh + 2
                    if pattern == "random":
                        src = rng.integers(0, 256, (bh, bw), dtype=np.uint8)
                        refp = rng.integers(0, 256, (rows, stride), dtype=np.uint8)
                    elif pattern == "ref_max":
                        src = np.zeros((bh, bw), np.uint8); refp = np.full((rows, stride), 255, np.uint8)
                    elif pattern == "src_max":
                        src = np.full((bh, bw), 255, np.uint8); refp = np.zeros((rows, stride), np.uint8)
                    else:
                        src = np.full((bh, bw), 77, np.uint8); refp = np.full((rows, stride), 77, np.uint8)
                    src = np.ascontiguousarray(src)
                    best = C.c_uint64(0); xs = C.c_int16(-7); ys = C.c_int16(-7)
                    ref.svt_sad_loop_kernel_c(src.ctypes.data_as(C.c_void_p), C.c_uint32(bw), refp.ctypes.data_as(C.c_void_p), C.c_uint32(stride),
                                              C.c_uint32(bh), C.c_uint32(bw), C.byref(best), C.byref(xs), C.byref(ys), C.c_uint32(stride),
                                              C.c_uint8(skip), C.c_int16(sw), C.c_int16(sh))
                    recs.append(dict(src=src, ref=refp, bw=bw, bh=bh, sw=sw, sh=sh, skip=skip, best=best.value, x=xs.value, y=ys.value))
    np.savez_compressed(os.path.join(OUT, "sad_loop_kat.npz"),
                        meta=np.array([[r["bw"], r["bh"], r["sw"], r["sh"], r["skip"], r["best"], r["x"], r["y"]] for r in recs], np.int64),
                        src=np.concatenate([r["src"].ravel() for r in recs]), ref=np.concatenate([r["ref"].ravel() for r in recs]),
                        ref_stride=np.array([r["ref"].shape[1] for r in recs], np.int64), ref_rows=np.array([r["ref"].shape[0] for r in recs], np.int64))
    print("wrote sad_loop_kat", len(recs))


def gen_presets():
    """Every (preset, resolution class, sc, rtc, layer, levels, qp, fps) -> the reference's MeContext controls."""
    import itertools
    from svt_av1_psyex_amd import abi
    keys, cfgs = [], []
    for em, res, sc, rtc, tl, hl, qp, fr in itertools.product(range(-3, 14), range(7), (0, 1), (0, 1), (0, 2), (3, 4), (20, 35, 63), (0, 30 << 16)):
        pd = abi.MePresetDesc(enc_mode=em, input_resolution=res, sc_class1=sc, rtc_tune=rtc, temporal_layer_index=tl, hierarchical_levels=hl,
                              qp=qp, frame_rate_q16=fr, safe_limit_nref=tl & 1, safe_limit_zz_th=5000)
        keys.append([em, res, sc, rtc, tl, hl, qp, fr])
        cfgs.append(np.frombuffer(bytes(pyoracle.config_from_preset_ref(pd)), np.uint8))
    np.savez_compressed(os.path.join(OUT, "me_presets.npz"), keys=np.array(keys, np.int64), cfg=np.stack(cfgs))
    print("wrote me_presets", len(keys))


PSY_RD = 1.35  # strength used for the psy_dist column of the block-statistics fixture


# svt_spatial_full_distortion_kernel_facade fixture columns: name -> (temporal_layer_index, spy_rd, psy_rd)
FACADE_SETTINGS = {"facade_a": (3, 1, PSY_RD), "facade_b": (5, 1, 0.0), "facade_c": (0, 1, 0.0), "facade_d": (4, 2, 0.0)}


def gen_block_stats():
    """Block statistics (SAD / SSE / variance / hadamard_path SATD) of the reference's `_c` kernels on two small planes."""
    from svt_av1_psyex_amd import abi, stats
    ref = pyoracle.load_ref()
    P = C.c_void_p
    ptr = lambda a: a.ctypes.data_as(P)
    ref.svt_spatial_full_distortion_kernel_c.restype = C.c_uint64
    ref.svt_full_distortion_kernel16_bits_c.restype = C.c_uint64
    ref.ref_hadamard_path.restype = C.c_uint32
    out = {}
    for bd in (8, 10):
        rng = np.random.default_rng(100 + bd)
        W, H = 192, 160
        dt = np.uint8 if bd == 8 else np.uint16
        src = rng.integers(0, 1 << bd, (H, W)).astype(dt)
        refp = np.clip(src.astype(np.int32) + rng.integers(-24, 25, (H, W)), 0, (1 << bd) - 1).astype(dt)
        src[:64, :64] = (1 << bd) - 1; refp[:64, :64] = 0          # extremes (VarianceTest.cc / SadTest.cc patterns)
        src[64:128, :64] = 0; refp[64:128, :64] = (1 << bd) - 1
        jobs = stats.random_jobs(rng, W, H, 160)
        jobs[0] = (0, 0, 64, 64, 0, 0); jobs[1] = (64 * W, 64 * W, 64, 64, 0, 0); jobs[2] = (0, 64 * W, 32, 32, 0, 0)
        exp = {name: np.zeros(len(jobs), dtype=d) for name, d in abi.STATS_OUT_FIELDS + abi.PSY_OUT_FIELDS}
        ref.svt_psy_distortion.restype = ref.svt_psy_distortion_hbd.restype = ref.get_svt_psy_full_dist.restype = C.c_uint64
        for j, jb in enumerate(jobs):
            w, h = int(jb["width"]), int(jb["height"])
            s = src.reshape(-1)[int(jb["src_offset"]):]
            r = refp.reshape(-1)[int(jb["ref_offset"]):]
            vs = C.c_uint32()
            fpsy = ref.svt_psy_distortion if bd == 8 else ref.svt_psy_distortion_hbd
            exp["psy_energy"][j] = fpsy(ptr(s), C.c_uint32(W), ptr(r), C.c_uint32(W), C.c_uint32(w), C.c_uint32(h))
            exp["psy_dist"][j] = ref.get_svt_psy_full_dist(ptr(s), C.c_uint32(0), C.c_uint32(W), ptr(r), C.c_uint32(0), C.c_uint32(W), C.c_uint32(w), C.c_uint32(h),
                                                           C.c_uint8(bd != 8), C.c_double(PSY_RD))
            if bd == 8:
                exp["sad"][j] = ref.svt_nxm_sad_kernel_helper_c(ptr(s), C.c_uint32(W), ptr(r), C.c_uint32(W), C.c_uint32(h), C.c_uint32(w))
                exp["sse"][j] = ref.svt_spatial_full_distortion_kernel_c(ptr(s), C.c_uint32(0), C.c_uint32(W), ptr(r), C.c_int32(0), C.c_uint32(W), C.c_uint32(w), C.c_uint32(h))
                if (w, h) in abi.VARIANCE_SIZES:
                    exp["variance"][j] = getattr(ref, f"svt_aom_variance{w}x{h}_c")(ptr(s), W, ptr(r), W, C.byref(vs)) & 0xFFFFFFFF
                else:  # no svt_aom_variance{W}x{H} of this shape: the generic 16-bit restatement on widened samples defines it
                    s16, r16 = s.astype(np.uint16), r.astype(np.uint16)
                    exp["variance"][j] = ref.svt_aom_variance_highbd_c(ptr(s16), W, ptr(r16), W, w, h, C.byref(vs)) & 0xFFFFFFFF
                exp["var_sse"][j] = vs.value
                if w == h:
                    exp["satd"][j] = ref.ref_hadamard_path(ptr(s), C.c_uint32(W), ptr(r), C.c_uint32(W), C.c_uint32(w))
            else:
                exp["sad"][j] = ref.svt_aom_sad_16b_kernel_c(ptr(s), C.c_uint32(W), ptr(r), C.c_uint32(W), C.c_uint32(h), C.c_uint32(w))
                exp["sse"][j] = ref.svt_full_distortion_kernel16_bits_c(ptr(s), C.c_uint32(0), C.c_uint32(W), ptr(r), C.c_int32(0), C.c_uint32(W), C.c_uint32(w), C.c_uint32(h))
                exp["variance"][j] = ref.svt_aom_variance_highbd_c(ptr(s), W, ptr(r), W, w, h, C.byref(vs)) & 0xFFFFFFFF
                exp["var_sse"][j] = vs.value
        # PSYEX facades (picture_operators_c.c:85-174).  svt_spatial_psy_distortion_kernel_c exists for 8-bit input only: the 10-bit
        # psy_sse column is the sum of the two reference results it would add (sse + get_svt_psy_full_dist).
        ref.ref_set_simd_rd(0)  # the facade dispatches through svt_spatial_full_distortion_kernel / svt_full_distortion_kernel16_bits
        ref.svt_spatial_psy_distortion_kernel_c.restype = ref.svt_spatial_full_distortion_kernel_facade.restype = C.c_uint64
        frng = np.random.default_rng(500 + bd)
        modes, comps = frng.integers(0, 25, len(jobs)).astype(np.uint8), frng.integers(0, 4, len(jobs)).astype(np.uint8)
        modes[:13] = np.arange(13)  # every intra mode at least once
        fac = {k: np.zeros(len(jobs), np.uint64) for k in FACADE_SETTINGS}
        for j, jb in enumerate(jobs):
            w, h = int(jb["width"]), int(jb["height"])
            s = src.reshape(-1)[int(jb["src_offset"]):]
            r = refp.reshape(-1)[int(jb["ref_offset"]):]
            if bd == 8:
                exp["psy_sse"][j] = ref.svt_spatial_psy_distortion_kernel_c(ptr(s), C.c_uint32(0), C.c_uint32(W), ptr(r), C.c_int32(0), C.c_uint32(W), C.c_uint32(w),
                                                                            C.c_uint32(h), C.c_double(PSY_RD))
            else:
                exp["psy_sse"][j] = exp["sse"][j] + exp["psy_dist"][j]
            for k, (tli, spy, prd) in FACADE_SETTINGS.items():
                fac[k][j] = ref.svt_spatial_full_distortion_kernel_facade(ptr(s), C.c_uint32(0), C.c_uint32(W), ptr(r), C.c_int32(0), C.c_uint32(W), C.c_uint32(w),
                                                                          C.c_uint32(h), C.c_bool(bd != 8), C.c_uint8(int(modes[j])), C.c_uint8(int(comps[j])),
                                                                          C.c_uint8(tli), C.c_double(prd), C.c_uint8(spy))
        out.update({f"src{bd}": src, f"ref{bd}": refp, f"jobs{bd}": jobs.view(np.uint8).reshape(len(jobs), -1)})
        out.update({f"{k}{bd}": v for k, v in exp.items()})
        if bd == 10:  # svt_aom_highbd_10_variance{W}x{H}_c on the AV1 shapes (pointers CONVERT_TO_BYTEPTR'd: address >> 1)
            v10, s10, ok = np.zeros(len(jobs), np.uint32), np.zeros(len(jobs), np.uint32), np.zeros(len(jobs), np.uint8)
            for j, jb in enumerate(jobs):
                w, h = int(jb["width"]), int(jb["height"])
                if (w, h) not in abi.VARIANCE_SIZES:
                    continue
                vs = C.c_uint32()
                sa = src.ctypes.data + 2 * int(jb["src_offset"])
                ra = refp.ctypes.data + 2 * int(jb["ref_offset"])
                v10[j] = getattr(ref, f"svt_aom_highbd_10_variance{w}x{h}_c")(C.c_void_p(sa >> 1), W, C.c_void_p(ra >> 1), W, C.byref(vs)) & 0xFFFFFFFF
                s10[j], ok[j] = vs.value, 1
            out.update(hbd10_variance=v10, hbd10_var_sse=s10, hbd10_valid=ok)
        out.update({f"fac_mode{bd}": modes, f"fac_comp{bd}": comps})
        out.update({f"{k}{bd}": v for k, v in fac.items()})
    # sub-pixel variance (8-bit planes of above): AV1 variance shapes x all 64 phases, from svt_aom_sub_pixel_variance{W}x{H}_c
    rng = np.random.default_rng(208)
    src, refp = out["src8"], out["ref8"]
    W, H = src.shape[1], src.shape[0]
    sizes = [sz for sz in abi.VARIANCE_SIZES if sz[0] < W and sz[1] < H]
    sp = stats.random_jobs(rng, W, H, 320, sizes=sizes, subpel=True)
    sp["subpel_x"][:64] = np.arange(64) % 8; sp["subpel_y"][:64] = np.arange(64) // 8
    sp_var, sp_sse = np.zeros(len(sp), np.uint32), np.zeros(len(sp), np.uint32)
    for j, jb in enumerate(sp):
        w, h = int(jb["width"]), int(jb["height"])
        s = src.reshape(-1)[int(jb["src_offset"]):]
        r = refp.reshape(-1)[int(jb["ref_offset"]):]
        vs = C.c_uint32()
        sp_var[j] = getattr(ref, f"svt_aom_sub_pixel_variance{w}x{h}_c")(ptr(s), W, int(jb["subpel_x"]), int(jb["subpel_y"]), ptr(r), W, C.byref(vs)) & 0xFFFFFFFF
        sp_sse[j] = vs.value
    out.update(sp_jobs=sp.view(np.uint8).reshape(len(sp), -1), sp_variance=sp_var, sp_var_sse=sp_sse)
    np.savez_compressed(os.path.join(OUT, "block_stats.npz"), **out)
    print("wrote block_stats")


def gen_dg_detector():
    """Metrics of the reference's dg_detector_hme_level0 (2 x 2 segments) on tests/dg_cases.py's grid."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dg_cases import GRID, METRICS, DgCase
    rows = []
    for w, h, kind in GRID:
        c = DgCase(w, h, kind)
        m = pyoracle.dg_detector("ref", c.src, c.ref, *c.args(), segments=(2, 2))
        rows.append([m[k] for k in METRICS])
    np.savez_compressed(os.path.join(OUT, "dg_detector.npz"), width=np.array([g[0] for g in GRID]), height=np.array([g[1] for g in GRID]),
                        kind=np.array([g[2] for g in GRID]), metrics=np.array(rows, np.int64))
    print("dg_detector.npz", rows)


def gen_tpl_chain():
    """The reference's TPL kernel chain (ref_harness.c:ref_tpl_chain) on tests/tpl_cases.py's grid (dispenser levels 0 and 1: 16x16 and
    32x32 blocks, each with its row-sub-sampled transforms): per job {inter_cost, eob, recon_error, sse}."""
    from tpl_cases import GRID, W, batch, planes, seed_of
    ref = pyoracle.load_ref()
    res = {}
    for i, (level, sub, pf, amp) in enumerate(GRID):
        src, pred = planes(seed_of(level, sub, pf), amp)
        _, jobs, rows = batch(level, sub, pf)
        co, q, dq = (np.zeros(1024, np.int32) for _ in range(3))
        per = np.zeros((len(jobs), 4), np.int64)
        for j, jb in enumerate(jobs):
            off = int(jb["src_offset"])
            assert ref.ref_tpl_chain(C.c_void_p(src.ctypes.data + off), W, C.c_void_p(pred.ctypes.data + off), W, level, sub, pf,
                                     C.c_void_p(rows.ctypes.data + int(jb["quant_row"]) * rows.dtype.itemsize), co.ctypes.data_as(C.c_void_p),
                                     q.ctypes.data_as(C.c_void_p), dq.ctypes.data_as(C.c_void_p), per[j].ctypes.data_as(C.c_void_p)) == 0
        res[f"results_{i}"] = per
    np.savez_compressed(os.path.join(OUT, "tpl_chain.npz"), grid=np.array(GRID), **res)
    print("tpl_chain.npz", len(res), "configurations")


def gen_rd_chain():
    """The reference's RD chain (ref_harness.c:ref_rd_batch on the `_c` kernels) on tests/rd_cases.py's cases."""
    import rd_cases
    out = {}
    for ci in range(len(rd_cases.CASES)):
        f, src, pred, jobs = rd_cases.inputs(ci)
        d = rd_cases.digest(pyoracle.rd_batch(f, src, pred, jobs, rd_cases.quant_rows(), impl="ref"))
        for k, v in d.items():
            out[f"c{ci}_{k}"] = v
    np.savez_compressed(os.path.join(OUT, "rd_chain.npz"), **out)
    print("rd_chain.npz", len(rd_cases.CASES), "cases")


def gen_pyramid():
    """1/4 and 1/16 luma planes with their padding as the reference's picture analysis makes them (ref_harness.c:ref_pyramid ->
    svt_aom_downsample_filtering_input_picture -> svt_aom_downsample_2d_c + svt_aom_generate_padding) for tests/pyramid_cases.py's
    pictures.  Stored: the padded quarter / sixteenth planes (tight stride); the inputs are regenerated from the case list."""
    from pyramid_cases import CASES, luma, ref_pyramid
    out = {}
    for name, (w, h, kind, seed) in CASES.items():
        q, s = ref_pyramid(luma(w, h, kind, seed))
        out[name + "_q"], out[name + "_s"] = q, s
        print("pyramid", name, q.shape, s.shape)
    np.savez_compressed(os.path.join(OUT, "pyramid.npz"), **out)


def gen_md_search():
    """The reference's md_full_pel_search chains and svt_av1_find_best_sub_pixel_tree_pruned (oracle/ref_harness_md.c) on tests/md_search_cases.py's grids."""
    import md_search_cases as mc
    ref = pyoracle.load_ref()
    out = {}
    for gi, (dist, psad, ctype) in enumerate(mc.FULLPEL_GRID):
        rng = np.random.default_rng(100 + dist * 10 + psad * 3 + ctype)
        src, refp = mc.planes(7 + dist)
        tables = mc.cost_tables(rng)
        rounds = mc.fullpel_chain(rng, 40, dist, psad)
        got = mc.run_fullpel_cpu(ref.ref_md_fullpel_batch, src, refp, rounds, ctype, 37, tables)
        out[f"fp_cost_{gi}"] = np.stack([c for c, _ in got])
        out[f"fp_mv_{gi}"] = np.stack([m for _, m in got])
    for si in range(len(mc.SUBPEL_SETTINGS)):
        rng = np.random.default_rng(300 + si)
        src, refp = mc.planes(11 + si)
        tables = mc.cost_tables(rng)
        jobs = mc.subpel_jobs(rng, 60)
        got = mc.run_subpel_cpu(ref.ref_md_subpel_batch, src, refp, jobs, mc.SUBPEL_SETTINGS[si], 41, 36, tables)
        for k, v in got.items():
            out[f"sp_{k}_{si}"] = v
    np.savez_compressed(os.path.join(OUT, "md_search.npz"), **out)
    print("md_search.npz", len(out), "arrays")


GENERATORS = {"md": gen_md_search, "pyramid": gen_pyramid, "me": gen_me, "me_mctf": lambda: gen_me(only=["me_vga_m4_mctf"]), "sad": gen_sad_kernels, "presets": gen_presets, "stats": gen_block_stats, "dg": gen_dg_detector, "tpl": gen_tpl_chain, "rd": gen_rd_chain}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    assert pyoracle.ref_available(), "build oracle/_ref first (make -C oracle ref)"
    for name in (sys.argv[1:] or list(GENERATORS)):
        GENERATORS[name]()
