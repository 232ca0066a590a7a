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


def config_from_preset_ref(preset_desc):
    cfg = abi.MeConfig()
    rc = load_ref().ref_me_config_from_preset(C.byref(preset_desc), C.byref(cfg))
    assert rc == 0
    return cfg
