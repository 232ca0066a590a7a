"""TEST INFRASTRUCTURE: ctypes access to oracle/liboracle.so (this repo's CPU restatement) and, when it has
been built, oracle/_ref/libsvtref.so (the reference's own sources compiled by oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

import sys
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from svt_av1_psyex_amd import abi  # noqa: E402

_oracle = None
_ref = None


def build(verbose=False):
    """(Re)build liboracle.so and, where /root/reference exists, _ref/libsvtref.so."""
    r = subprocess.run(["make", "-C", HERE, "-j8", "all"], capture_output=True, text=True)
    if r.returncode != 0 or verbose:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("oracle build failed")


def load_oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _oracle = C.CDLL(path)
        _oracle.orc_sizeof.restype = C.c_size_t
        _oracle.orc_nxm_sad.restype = C.c_uint32
        _oracle.orc_sad_16b.restype = C.c_uint32
    return _oracle


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libsvtref.so"))


def load_ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(HERE, "_ref", "libsvtref.so"))
        _ref.svt_nxm_sad_kernel_helper_c.restype = C.c_uint32
        _ref.svt_aom_sad_16b_kernel_c.restype = C.c_uint32
        _ref.ref_set_simd(0)
    return _ref


def alloc_results(n_b64, desc, search_level=True):
    """numpy-backed SvtHipMeResults.  Returns (MeResults struct, dict name -> array)."""
    n = abi.n_pu(desc.enable_me_16x16, desc.enable_me_8x8)
    arrs = {}
    res = abi.MeResults()
    for name, dt, cnt in abi.RESULT_FIELDS:
        if not search_level and name in ("sb_best_sad", "sb_best_mv", "hme_sc", "hme_sad", "do_ref"):
            continue
        a = np.zeros((n_b64, cnt(n, desc.max_refs, desc.max_cand)), dtype=dt)
        arrs[name] = a
        setattr(res, name, a.ctypes.data)
    return res, arrs


def ref_plane_array(refs):
    """refs: dict (list, idx) -> HostPyramid.  Returns ctypes SvtHipPlaneDesc[2][4][3]."""
    arr = ((abi.PlaneDesc * 3) * abi.MAX_REFS * abi.MAX_LISTS)()
    for (li, ri), pyr in refs.items():
        for l in range(3):
            arr[li][ri][l] = pyr.desc(l)
    return arr


def me_picture(which, cfg, desc, cur, refs, search_level=True):
    """which: 'oracle' or 'ref'.  cur: HostPyramid; refs: dict (list, idx) -> HostPyramid."""
    lib, fn = (load_oracle(), "orc_me_picture") if which == "oracle" else (load_ref(), "ref_me_picture")
    w64 = (desc.aligned_width + 63) // 64
    h64 = (desc.aligned_height + 63) // 64
    res, arrs = alloc_results(w64 * h64, desc, search_level)
    rc = getattr(lib, fn)(C.byref(cfg), C.byref(desc), cur.descs(), ref_plane_array(refs), C.byref(res))
    if rc != 0:
        raise RuntimeError(f"{fn} failed: {rc}")
    return arrs


def dg_detector(which, src, ref, aligned_width, aligned_height, input_resolution, segments=(1, 1)):
    """dg_detector_hme_level0 on HostPyramids.  which: 'oracle' (metrics + per-b64 SAD / vector) or 'ref' (metrics only,
    run segment by segment over a segments = (columns, rows) split as the reference's ME threads do)."""
    m = abi.DgMetrics()
    s16, r16 = src.desc(0), ref.desc(0)
    if which == "ref":
        rc = load_ref().ref_dg_detector(C.byref(s16), C.byref(r16), aligned_width, aligned_height, input_resolution, segments[0],
                                        segments[1], C.byref(m))
        assert rc == 0 and m.reserved == segments[0] * segments[1]  # seg_completed
        return m.as_dict()
    n = ((aligned_width + 63) // 64) * ((aligned_height + 63) // 64)
    sad, mv = np.zeros(n, np.uint32), np.zeros((n, 2), np.int16)
    rc = load_oracle().orc_dg_detector_hme_level0(C.byref(s16), C.byref(r16), aligned_width, aligned_height, input_resolution, C.byref(m),
                                                  sad.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return dict(m.as_dict(), b64_sad=sad, b64_mv=mv)


def config_from_preset_ref(preset_desc):
    cfg = abi.MeConfig()
    rc = load_ref().ref_me_config_from_preset(C.byref(preset_desc), C.byref(cfg))
    assert rc == 0
    return cfg


# ---- CPU mirrors of the batched device entries (same descriptors, host pointers): checkers for tests / smoke / bench baseline ----
def rd_batch(desc_fields, src, pred, jobs, quant_rows, want_coeffs=True, want_recon=True, qmatrix=None, iqmatrix=None, impl="oracle"):
    """CPU mirror of svt_hip_rd_batch on host numpy arrays (test infrastructure; imports oracle/).  impl: "oracle" = the C
    restatement (oracle/rd_oracle.c); "ref" / "ref_simd" = the reference's own `_c` / AVX2+SSE4.1 kernels chained by
    oracle/ref_harness.c:ref_rd_batch (build container, or wherever oracle/_ref/libsvtref.so travelled to)."""
    if impl == "oracle":
        fn = load_oracle().orc_rd_batch
    else:
        r = load_ref()
        r.ref_set_simd_rd(1 if impl == "ref_simd" else 0)
        fn = r.ref_rd_batch
    ts = desc_fields["tx_size"]
    npk = min(abi.TX_W[ts], 32) * min(abi.TX_H[ts], 32)
    n = len(jobs)
    out = {name: np.zeros((n, k), dtype=dt) for name, dt, k in abi.RD_OUT_FIELDS}
    if want_coeffs:
        for name in ("coeff", "qcoeff", "dqcoeff"):
            out[name] = np.zeros((n, npk), np.int32)
    recon = pred.copy() if want_recon else None
    d = abi.RdBatchDesc(n_jobs=n, src=src.ctypes.data, pred=pred.ctypes.data, recon=recon.ctypes.data if want_recon else None,
                        jobs=jobs.ctypes.data, quant_rows=quant_rows.ctypes.data, n_quant_rows=len(quant_rows), **desc_fields)
    for name in out:
        setattr(d, name, out[name].ctypes.data)
    if qmatrix is not None:
        qmatrix, iqmatrix = np.ascontiguousarray(qmatrix, np.uint8), np.ascontiguousarray(iqmatrix, np.uint8)
        d.qmatrix, d.iqmatrix = qmatrix.ctypes.data, iqmatrix.ctypes.data
    assert fn(C.byref(d)) == 0
    if want_recon:
        out["recon"] = recon
    return out



def block_stats(oracle, src, ref, jobs, bit_depth, satd=True, psy_rd=None, facade=None):
    """src / ref: 2-D numpy planes (uint8 or uint16); returns a dict of per-job arrays from oracle/stats_oracle.c.
    psy_rd: also return the psy-RD terms (jobs must then have widths / heights that are multiples of 4)."""
    n = len(jobs)
    out = {name: np.zeros(n, dtype=dt) for name, dt in abi.STATS_OUT_FIELDS}
    d = abi.BlockStatsDesc(bit_depth=bit_depth, n_jobs=n, src_stride=src.shape[1], ref_stride=ref.shape[1])
    if psy_rd is not None:
        d.psy_rd = psy_rd
        for name, dt in abi.PSY_OUT_FIELDS:
            out[name] = np.zeros(n, dtype=dt)
            setattr(d, name, out[name].ctypes.data)
    if bit_depth == 10:  # svt_aom_highbd_10_variance{W}x{H}
        for name, dt in abi.VAR10_OUT_FIELDS:
            out[name] = np.zeros(n, dtype=dt)
            setattr(d, name, out[name].ctypes.data)
    if facade:  # dict(pred_mode, compound_type, temporal_layer_index, spy_rd): svt_spatial_full_distortion_kernel_facade
        modes, comps = np.ascontiguousarray(facade["pred_mode"], np.uint8), np.ascontiguousarray(facade["compound_type"], np.uint8)
        out["facade_dist"] = np.zeros(n, np.uint64)
        d.pred_mode, d.compound_type, d.facade_dist = modes.ctypes.data, comps.ctypes.data, out["facade_dist"].ctypes.data
        d.temporal_layer_index, d.spy_rd = facade["temporal_layer_index"], facade["spy_rd"]
    src, ref, jobs = np.ascontiguousarray(src), np.ascontiguousarray(ref), np.ascontiguousarray(jobs)
    d.src, d.ref, d.jobs = src.ctypes.data, ref.ctypes.data, jobs.ctypes.data
    for name, _ in abi.STATS_OUT_FIELDS:
        if name == "satd" and not satd:
            continue
        setattr(d, name, out[name].ctypes.data)
    oracle.orc_block_stats_batch.restype = C.c_int
    rc = oracle.orc_block_stats_batch(C.byref(d))
    assert rc == 0, rc
    if not satd:
        out.pop("satd")
    return out




# ---- bench.py's CPU baseline: the reference's kernels on native threads (oracle/ref_harness.c:ref_bench_rows) ----
class RefBenchPicture(C.Structure):
    _fields_ = [("cfg", C.c_void_p), ("desc", C.c_void_p), ("cur_planes", C.c_void_p), ("ref_planes", C.c_void_p), ("src10", C.c_void_p), ("pred10", C.c_void_p)]


class RefBenchDesc(C.Structure):
    _fields_ = [("n_pictures", C.c_uint32), ("pictures", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("row_start", C.c_uint32), ("n_rows", C.c_uint32),
                ("quant_row", C.c_void_p), ("n_tx_sizes", C.c_uint32), ("tx_sizes", C.c_void_p), ("n_threads", C.c_int32), ("seconds", C.c_double),
                ("items_done", C.c_uint64), ("elapsed", C.c_double), ("checksum", C.c_uint64)]


def ref_bench_rows(pictures, width, height, row_start, n_rows, quant_rows, tx_sizes, n_threads, seconds, simd=True):
    """pictures: list of (cfg, desc, cur HostPyramid, refs dict (list, idx) -> HostPyramid, src10 uint16 [H, W], pred10 uint16 [H, W]).
    Runs ME (svt_aom_motion_estimation_b64 over the sample's b64 rows) + the RD chain at `tx_sizes` on `n_threads` pthreads for about
    `seconds`; returns (Mpixels/s, items done, elapsed seconds, checksum).  One item = one b64 row = 64 x width luma pixels."""
    r = load_ref()
    r.ref_sizeof_bench.restype = C.c_size_t
    assert r.ref_sizeof_bench(0) == C.sizeof(RefBenchDesc) and r.ref_sizeof_bench(1) == C.sizeof(RefBenchPicture)
    r.ref_set_simd(1 if simd else 0)
    r.ref_set_simd_rd(1 if simd else 0)
    keep = []
    arr = (RefBenchPicture * len(pictures))()
    for i, (cfg, desc, cur, refs, src10, pred10) in enumerate(pictures):
        cp, rp = cur.descs(), ref_plane_array(refs)
        src10, pred10 = np.ascontiguousarray(src10, np.uint16), np.ascontiguousarray(pred10, np.uint16)
        assert src10.shape == (height, width) and pred10.shape == (height, width)
        keep += [cfg, desc, cp, rp, src10, pred10]
        arr[i] = RefBenchPicture(C.addressof(cfg), C.addressof(desc), C.addressof(cp), C.addressof(rp), src10.ctypes.data, pred10.ctypes.data)
    qr = np.ascontiguousarray(quant_rows)
    ts = np.ascontiguousarray(tx_sizes, np.int32)
    d = RefBenchDesc(n_pictures=len(pictures), pictures=C.addressof(arr), width=width, height=height, row_start=row_start, n_rows=n_rows, quant_row=qr.ctypes.data,
                     n_tx_sizes=len(ts), tx_sizes=ts.ctypes.data, n_threads=n_threads, seconds=seconds)
    rc = r.ref_bench_rows(C.byref(d))
    r.ref_set_simd(0)
    r.ref_set_simd_rd(0)
    if rc != 0:
        raise RuntimeError(f"ref_bench_rows failed: {rc}")
    return d.items_done * 64 * width / d.elapsed / 1e6, int(d.items_done), float(d.elapsed), int(d.checksum)
