"""Shared builders for ME parity cases (synthetic pictures, descriptors).  Test infrastructure."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from svt_av1_psyex_amd import abi, api, synth  # noqa: E402

_seq_cache = {}


def sequence(width, height, n_frames, seed, kind="pan"):
    """uint8 luma frames [n, h, w].  kinds: pan, noise (i.i.d.), flat, extremes (0 vs 255)."""
    key = (width, height, n_frames, seed, kind)
    if key not in _seq_cache:
        if kind == "pan":
            s = synth.to_8bit(synth.synth_sequence(width, height, n_frames, seed))
        elif kind == "fastpan":
            s = synth.to_8bit(synth.synth_sequence(width, height, n_frames, seed, pan=(23, -11 + 11)))
        elif kind == "noise":
            s = np.random.default_rng(seed).integers(0, 256, (n_frames, height, width), dtype=np.uint8)
        elif kind == "flat":
            s = np.full((n_frames, height, width), 128, np.uint8)
        elif kind == "extremes":
            s = np.zeros((n_frames, height, width), np.uint8)
            s[1::2] = 255
        else:
            raise ValueError(kind)
        _seq_cache[key] = s
    return _seq_cache[key]


class MeCase:
    """One picture + its references + descriptors, in the shape every implementation consumes."""

    def __init__(self, width, height, enc_mode=6, cur=2, refs=None, seed=1234, kind="pan", temporal_layer_index=1,
                 hierarchical_levels=4, is_ref=1, gm_enabled=0, qp=35, sc_class1=0, rtc_tune=0, n_frames=None, cfg_edit=None,
                 pad=68, mctf_exit_th=None):
        refs = refs if refs is not None else {(0, 0): 0, (1, 0): 3}
        n_frames = n_frames or (max([cur] + list(refs.values())) + 1)
        frames = sequence(width, height, n_frames, seed, kind)
        self.width, self.height = width, height
        self.cur = synth.HostPyramid(frames[cur], cur, pad=pad)
        self.refs = {k: synth.HostPyramid(frames[v], v, pad=pad) for k, v in refs.items()}
        self.cfg = api.config_from_preset(enc_mode, width, height, qp=qp, temporal_layer_index=temporal_layer_index,
                                          hierarchical_levels=hierarchical_levels, sc_class1=sc_class1, rtc_tune=rtc_tune)
        if cfg_edit:
            cfg_edit(self.cfg)
        self.desc = api.picture_desc(width, height, cur, refs, enc_mode=enc_mode, temporal_layer_index=temporal_layer_index,
                                     hierarchical_levels=hierarchical_levels, is_ref=is_ref, gm_enabled=gm_enabled, rtc_tune=rtc_tune)
        if mctf_exit_th is not None:  # temporal-filter ME (ME_MCTF): search-level results only
            self.cfg.me_type = 1
            self.desc.tf_me_exit_th = mctf_exit_th

    def run_cpu(self, which="oracle"):
        import pyoracle
        return pyoracle.me_picture(which, self.cfg, self.desc, self.cur, self.refs)

    def run_cpu_banded(self, which="oracle", threads=8):
        """run_cpu with the b64 rows of this case's band split over host threads (ctypes releases the GIL): the slow presets at
        2160p.  Rows outside the case's band stay zero, as in a band call."""
        import concurrent.futures as cf
        import pyoracle
        h64 = (self.height + 63) // 64
        r0 = self.desc.b64_row_start
        r1 = r0 + self.desc.b64_row_count if self.desc.b64_row_count else h64
        w64 = (self.width + 63) // 64

        def band(row):
            d = abi.MePictureDesc.from_buffer_copy(bytes(self.desc))
            d.b64_row_start, d.b64_row_count = row, 1
            return row, pyoracle.me_picture(which, self.cfg, d, self.cur, self.refs)

        out = None
        with cf.ThreadPoolExecutor(threads) as ex:
            for row, res in ex.map(band, range(r0, r1)):
                if out is None:
                    out = {k: np.zeros_like(v) for k, v in res.items()}
                for k, v in res.items():
                    out[k][row * w64:(row + 1) * w64] = v[row * w64:(row + 1) * w64]
        return out

    def run_hip(self, ctx, device_pyramid=True):
        cur = ctx.upload(self.cur, device_pyramid)
        refs = {k: ctx.upload(v, device_pyramid) for k, v in self.refs.items()}
        try:
            return ctx.me_picture(self.cfg, self.desc, cur, refs)
        finally:
            cur.free()
            for r in refs.values():
                r.free()


def compare(a, b, names=None):
    """Returns list of mismatch descriptions between two result dicts (bit-exact)."""
    bad = []
    for k in (names or a.keys()):
        if not np.array_equal(a[k], b[k]):
            idx = np.argwhere(a[k] != b[k])
            bad.append(f"{k}: {len(idx)} mismatches, first at {tuple(idx[0])}: {a[k][tuple(idx[0])]} vs {b[k][tuple(idx[0])]}")
    return bad


def fill_unsearched(desc, got, max_sad=128 * 128 * 255):
    """The asynchronous ME entries leave the search-level slots of (list, reference) pairs the picture does not search untouched
    (include/svt_hip_me.h); svt_hip_me_picture and the oracle hold SVT_HIP_MAX_SAD_VALUE / 0 there: bring a result read back from a device
    buffer to that form before comparing."""
    for name, val in (("sb_best_sad", max_sad), ("sb_best_mv", 0)):
        if name not in got:
            continue
        a = got[name].reshape(-1, 2, 4, 85)
        for li in range(2):
            for ri in range(4):
                if not (li < desc.num_of_list_to_search and ri < desc.num_of_ref_pic_to_search[li]):
                    a[:, li, ri, :] = val
    return got


MCTF_OUTPUTS = ("sb_best_sad", "sb_best_mv", "hme_sc", "hme_sad", "do_ref")  # what ME_MCTF produces (motion_estimation.c:3126)

MCTF_GRID = [
    dict(width=352, height=288, enc_mode=6, refs={(0, 0): 1}, mctf_exit_th=0),                      # no early exit
    dict(width=352, height=288, enc_mode=6, refs={(0, 0): 0}, mctf_exit_th=40000, seed=2),          # every block exits
    dict(width=640, height=360, enc_mode=4, refs={(0, 0): 1}, mctf_exit_th=6940, seed=3),           # mixed
    dict(width=352, height=288, enc_mode=9, refs={(0, 0): 1, (1, 0): 3}, mctf_exit_th=25000, kind="fastpan"),
    dict(width=352, height=288, enc_mode=2, cur=2, refs={(0, 0): 1, (0, 1): 0, (1, 0): 3, (1, 1): 4}, n_frames=5, mctf_exit_th=100),
]


def probe_outside_case():
    """A search centre far outside the picture: list 1 mirrors list 0's pre-HME vector (check_prehme_early_exit,
    motion_estimation.c:1693-1720), HME levels 1 / 2 are off, so (16, 184) reaches the integer search of the last, 16-px-high b64
    row of a 144-px-high picture unrefined and the 1-point probe (:1391-1406) addresses rows far below the padded plane.  The
    reference reads past its buffer there (undefined); oracle and HIP kernel read the plane's nearest edge instead.  Found by
    tools/me_fuzz_campaign.py (seed 5099)."""
    def edit(cfg):
        cfg.hme_search_method = cfg.me_search_method = 1
        cfg.enable_hme_level1_flag = cfg.enable_hme_level2_flag = 0
        cfg.prehme_enable = cfg.prehme_skip_search_line = cfg.prehme_l1_early_exit = 1
        cfg.me_sa.sa_min.width, cfg.me_sa.sa_min.height, cfg.me_sa.sa_max.width, cfg.me_sa.sa_max.height = 8, 37, 136, 37
        cfg.hme_l0_sa.sa_min.width, cfg.hme_l0_sa.sa_min.height, cfg.hme_l0_sa.sa_max.width, cfg.hme_l0_sa.sa_max.height = 64, 8, 96, 192
        cfg.me_early_exit_th, cfg.me_8x8_var_enabled, cfg.enable_me_sr_adjustment, cfg.distance_based_hme_resizing = 32768, 1, 1, 0
        cfg.mv_sa_adj_enabled, cfg.mv_sa_adj_nearest_ref_only, cfg.mv_sa_adj_mv_size_th, cfg.mv_sa_adj_sa_multiplier = 1, 0, 16, 2
        cfg.prune_ref_if_me_sad_dev_bigger_than_th, cfg.use_best_unipred_cand_only = 0xFFFF, 0
    return MeCase(640, 144, enc_mode=11, cur=4, refs={(0, 0): 2, (0, 1): 3, (0, 2): 1, (1, 0): 0, (1, 1): 7}, n_frames=9, seed=5099, kind="noise",
                  temporal_layer_index=1, cfg_edit=edit)
