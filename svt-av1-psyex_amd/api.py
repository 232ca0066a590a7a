"""Host-side Python binding of libsvthip.so (ctypes over the C-ABI of include/svt_hip_me.h).

This is plumbing for tests, bench.py and Python callers; the product is the shared library.  It fails loudly
when the library (or a gfx950 GPU, for anything that computes) is missing -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVT_HIP_LIBRARY") or os.path.join(_HERE, "libsvthip.so")  # override: a tuning build of the same ABI
_lib = None

ERRORS = {1: "no usable gfx950 device", 2: "bad parameter", 3: "out of device memory", 4: "launch/stream error"}


class SvtHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SvtHipError(f"{LIB_PATH} is missing: build it with `make -C svt-av1-psyex_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        # PyTorch-ROCm bundles its own libamdhip64.so.7; whichever copy is loaded first serves the whole process.
        # Load torch's first so that torch tensors / streams and this library share ONE HIP runtime.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional plumbing; the library itself only needs the ROCm runtime
            pass
        L = C.CDLL(LIB_PATH)
        L.svt_hip_last_error.restype = C.c_char_p
        L.svt_hip_last_error.argtypes = [C.c_void_p]
        L.svt_hip_context_stream.restype = C.c_void_p
        L.svt_hip_context_stream.argtypes = [C.c_void_p]
        L.svt_hip_sizeof.restype = C.c_size_t
        L.svt_hip_input_resolution.restype = C.c_uint8
        L.svt_hip_enable_me_8x8.restype = C.c_uint8
        for name in ("svt_hip_context_destroy", "svt_hip_context_sync"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.svt_hip_pa_picture_destroy.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def config_from_preset(enc_mode, width, height, qp=35, temporal_layer_index=1, hierarchical_levels=4, sc_class1=0,
                       rtc_tune=0, fps=30):
    """SvtHipMeConfig of a CLI preset (restates svt_aom_sig_deriv_me, Codec/enc_mode_config.c:681-833)."""
    L = lib()
    pd = abi.MePresetDesc(enc_mode=enc_mode, input_resolution=L.svt_hip_input_resolution(width, height), sc_class1=sc_class1,
                          rtc_tune=rtc_tune, temporal_layer_index=temporal_layer_index, hierarchical_levels=hierarchical_levels,
                          qp=qp, frame_rate_q16=fps << 16)
    cfg = abi.MeConfig()
    rc = L.svt_hip_me_config_from_preset(C.byref(pd), C.byref(cfg))
    if rc:
        raise SvtHipError(f"svt_hip_me_config_from_preset: {ERRORS.get(rc, rc)}")
    return cfg


def picture_desc(width, height, picture_number, refs, enc_mode=6, temporal_layer_index=1, hierarchical_levels=4, is_ref=1,
                 gm_enabled=0, rtc_tune=0):
    """SvtHipMePictureDesc for a picture with `refs` = {(list, idx): picture_number}."""
    L = lib()
    res = L.svt_hip_input_resolution(width, height)
    r0 = sum(1 for (li, _) in refs if li == 0)
    r1 = sum(1 for (li, _) in refs if li == 1)
    d = abi.MePictureDesc(picture_number=picture_number, aligned_width=width, aligned_height=height,
                          num_of_list_to_search=2 if r1 else 1, temporal_layer_index=temporal_layer_index,
                          hierarchical_levels=hierarchical_levels, is_ref=is_ref, similar_brightness_refs=0,
                          enable_me_8x8=L.svt_hip_enable_me_8x8(enc_mode, rtc_tune, res), enable_me_16x16=1,
                          max_number_of_pus_per_sb=85, input_resolution=res, gm_enabled=gm_enabled)
    d.num_of_ref_pic_to_search[0] = r0
    d.num_of_ref_pic_to_search[1] = r1
    # svt_aom_get_max_allocated_me_refs (Codec/pcs.c:91-96)
    d.max_refs = r0 + r1
    d.max_cand = r0 + r1 + r0 * r1 + (r0 - 1) + (1 if r1 == 3 else 0)
    d.max_l0 = r0
    for (li, ri), num in refs.items():
        d.ref_picture_number[li][ri] = num
    return d


class Context:
    """One per process / GPU (svt_hip_context_create)."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        rc = lib().svt_hip_context_create(C.byref(self._h), device)
        if rc:
            raise SvtHipError(f"svt_hip_context_create failed: {ERRORS.get(rc, rc)} -- the HIP path cannot run here")

    def check(self, rc, what):
        if rc:
            raise SvtHipError(f"{what}: {ERRORS.get(rc, rc)}: {lib().svt_hip_last_error(self._h).decode()}")

    @property
    def stream(self):
        return lib().svt_hip_context_stream(self._h)

    def sync(self):
        self.check(lib().svt_hip_context_sync(self._h), "sync")

    def close(self):
        if self._h:
            lib().svt_hip_context_destroy(self._h)
            self._h = C.c_void_p()

    def set_me_dense(self, on):
        """the dense pre-HME / HME level-0 pre-pass of the ME launches (svt_hip_context_set_me_dense); results are identical either way"""
        self.check(lib().svt_hip_context_set_me_dense(self._h, 1 if on else 0), "svt_hip_context_set_me_dense")

    def set_me_staged(self, on):
        """with the pre-pass on: the per-block pipeline as a chain of small kernels (svt_hip_context_set_me_staged): 0 never, 1 launches of many blocks (default), 2 always; results are identical either way"""
        self.check(lib().svt_hip_context_set_me_staged(self._h, int(on)), "svt_hip_context_set_me_staged")

    def set_me_timing(self, on):
        """ME launches record events around the kernels of their chain (svt_hip_context_set_me_timing)"""
        self.check(lib().svt_hip_context_set_me_timing(self._h, 1 if on else 0), "svt_hip_context_set_me_timing")

    def me_launch_times(self):
        """{kernel name: ms} of the last ME launch on the context stream (svt_hip_me_launch_times; waits for the launch); kernels it did not use are left out"""
        L = lib()
        ms = (C.c_float * 7)()
        self.check(L.svt_hip_me_launch_times(self._h, ms), "svt_hip_me_launch_times")
        L.svt_hip_me_chain_kernel_name.restype = C.c_char_p
        return {L.svt_hip_me_chain_kernel_name(i).decode(): float(ms[i]) for i in range(7) if ms[i] > 0}

    def set_me_counting(self, on):
        """ME waves report what they took from the dense pre-pass (svt_hip_context_set_me_counting); off by default"""
        self.check(lib().svt_hip_context_set_me_counting(self._h, 1 if on else 0), "svt_hip_context_set_me_counting")

    def me_dense_counters(self):
        """(searches taken from the dense pre-pass, searches the per-block kernel made itself) since the last call"""
        v = (C.c_ulonglong * 2)()
        self.check(lib().svt_hip_me_dense_counters(self._h, v), "svt_hip_me_dense_counters")
        return int(v[0]), int(v[1])

    # --- pictures -----------------------------------------------------------------------------------
    def upload(self, host_pyramid, device_pyramid=True):
        """HostPyramid -> HBM-resident pyramid.  device_pyramid=True builds 1/4 and 1/16 on the GPU."""
        pic = C.c_void_p()
        full = host_pyramid.desc(2)
        if device_pyramid:
            rc = lib().svt_hip_pa_picture_create(self._h, C.byref(full), None, None, C.byref(pic))
        else:
            q, s = host_pyramid.desc(1), host_pyramid.desc(0)
            rc = lib().svt_hip_pa_picture_create(self._h, C.byref(full), C.byref(q), C.byref(s), C.byref(pic))
        self.check(rc, "svt_hip_pa_picture_create")
        return DevicePicture(self, pic, host_pyramid.picture_number)

    def upload_dev(self, dev_ptr, stride, width, height, pad, picture_number=0):
        """Full-resolution padded plane already in device memory -> pyramid (no H2D copy)."""
        pic = C.c_void_p()
        full = abi.PlaneDesc(dev_ptr, stride, pad, pad, width, height)
        self.check(lib().svt_hip_pa_picture_create_dev(self._h, C.byref(full), C.byref(pic)), "svt_hip_pa_picture_create_dev")
        return DevicePicture(self, pic, picture_number)

    # --- ME ------------------------------------------------------------------------------------------
    def _ref_array(self, refs):
        arr = ((C.c_void_p * abi.MAX_REFS) * abi.MAX_LISTS)()
        for (li, ri), pic in refs.items():
            arr[li][ri] = pic._h
        return arr

    def me_picture(self, cfg, desc, cur, refs, search_level=True):
        """Synchronous form: results come back as numpy arrays (dict keyed like SvtHipMeResults)."""
        n = abi.n_pu(desc.enable_me_16x16, desc.enable_me_8x8)
        nb = ((desc.aligned_width + 63) // 64) * ((desc.aligned_height + 63) // 64)
        res, arrs = abi.MeResults(), {}
        for name, dt, cnt in abi.RESULT_FIELDS:
            if not search_level and name in ("sb_best_sad", "sb_best_mv", "hme_sc", "hme_sad", "do_ref"):
                continue
            arrs[name] = np.zeros((nb, cnt(n, desc.max_refs, desc.max_cand)), dtype=dt)
            setattr(res, name, arrs[name].ctypes.data)
        rc = lib().svt_hip_me_picture(self._h, C.byref(cfg), C.byref(desc), cur._h, self._ref_array(refs), C.byref(res))
        self.check(rc, "svt_hip_me_picture")
        return arrs

    def me_picture_async(self, cfg, desc, cur, refs, res_dev):
        """Asynchronous form: `res_dev` is an abi.MeResults of DEVICE pointers; enqueues on the context stream."""
        rc = lib().svt_hip_me_picture_async(self._h, C.byref(cfg), C.byref(desc), cur._h, self._ref_array(refs), C.byref(res_dev))
        self.check(rc, "svt_hip_me_picture_async")


    def me_pictures_async(self, jobs):
        """Several pictures in one launch: `jobs` = [(cfg, desc, cur, refs, res_dev), ...] as for me_picture_async."""
        arr = (abi.MeJob * len(jobs))()
        for j, (cfg, desc, cur, refs, res) in zip(arr, jobs):
            j.cfg, j.desc, j.results = C.addressof(cfg), C.addressof(desc), C.addressof(res)
            j.cur = cur._h if isinstance(cur._h, int) else cur._h.value
            for (li, ri), pic in refs.items():
                j.refs[li][ri] = pic._h if isinstance(pic._h, int) else pic._h.value
        self.check(lib().svt_hip_me_pictures_async(self._h, len(jobs), arr), "svt_hip_me_pictures_async")


    def dg_detector_hme_level0(self, src, ref, aligned_width, aligned_height, input_resolution, per_block=True):
        """dg_detector_hme_level0 over the whole picture (pd_process.c:492-588): metrics dict, plus the per-b64
        SAD [n_b64] and vector [n_b64, 2] (col, row) when per_block."""
        n = ((aligned_width + 63) // 64) * ((aligned_height + 63) // 64)
        m = abi.DgMetrics()
        sad = np.zeros(n, np.uint32) if per_block else None
        mv = np.zeros((n, 2), np.int16) if per_block else None
        rc = lib().svt_hip_dg_detector_hme_level0(self._h, src._h, ref._h, aligned_width, aligned_height, input_resolution, C.byref(m),
                                                  sad.ctypes.data_as(C.c_void_p) if per_block else None,
                                                  mv.ctypes.data_as(C.c_void_p) if per_block else None)
        self.check(rc, "svt_hip_dg_detector_hme_level0")
        out = m.as_dict()
        if per_block:
            out["b64_sad"], out["b64_mv"] = sad, mv
        return out


    def dg_detector_hme_level0_async(self, src, ref, aligned_width, aligned_height, input_resolution, metrics_ptr, sad_ptr=None, mv_ptr=None):
        """Asynchronous form on DEVICE pointers (ints); enqueues on the context stream."""
        rc = lib().svt_hip_dg_detector_hme_level0_async(self._h, src._h, ref._h, aligned_width, aligned_height, input_resolution,
                                                        C.c_void_p(metrics_ptr), C.c_void_p(sad_ptr), C.c_void_p(mv_ptr))
        self.check(rc, "svt_hip_dg_detector_hme_level0_async")


class DevicePicture:
    def __init__(self, ctx, handle, picture_number):
        self.ctx, self._h, self.picture_number = ctx, handle, picture_number

    def geometry(self, level):
        d = abi.PlaneDesc()
        self.ctx.check(lib().svt_hip_pa_picture_geometry(self._h, level, C.byref(d)), "geometry")
        return d

    def download(self, level):
        """Padded plane of `level` as a numpy array [height + 2*org_y, width + 2*org_x]."""
        g = self.geometry(level)
        out = np.zeros((g.height + 2 * g.org_y, g.width + 2 * g.org_x), np.uint8)
        self.ctx.check(lib().svt_hip_pa_picture_download(self.ctx._h, self._h, level, out.ctypes.data_as(C.c_void_p), out.shape[1]), "download")
        return out

    def free(self):
        if self._h:
            lib().svt_hip_pa_picture_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()
