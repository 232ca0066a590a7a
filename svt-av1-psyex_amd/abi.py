"""ctypes mirror of include/svt_hip_me.h (the C-ABI).  Pure declarations; no compute.

The structures follow the field order of the header exactly; tests/test_abi.py checks the sizes against
the compiled libraries (``svt_hip_sizeof`` / ``orc_sizeof``).
"""
import ctypes as C

import numpy as np

MAX_LISTS = 2
MAX_REFS = 4
SQUARE_PU_COUNT = 85
MAX_SAD_VALUE = 128 * 128 * 255


class SearchArea(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16)]


class SearchAreaMinMax(C.Structure):
    _fields_ = [("sa_min", SearchArea), ("sa_max", SearchArea)]


class MeConfig(C.Structure):
    _fields_ = [
        ("hme_search_method", C.c_uint8), ("me_search_method", C.c_uint8),
        ("enable_hme_flag", C.c_uint8), ("enable_hme_level0_flag", C.c_uint8),
        ("enable_hme_level1_flag", C.c_uint8), ("enable_hme_level2_flag", C.c_uint8),
        ("num_hme_sa_w", C.c_uint16), ("num_hme_sa_h", C.c_uint16),
        ("hme_l0_sa", SearchAreaMinMax), ("hme_l1_sa", SearchArea), ("hme_l2_sa", SearchArea),
        ("me_sa", SearchAreaMinMax),
        ("prehme_enable", C.c_uint8), ("prehme_skip_search_line", C.c_uint8), ("prehme_l1_early_exit", C.c_uint8), ("me_type", C.c_uint8),
        ("prehme_sa_cfg", SearchAreaMinMax * 2),
        ("enable_me_hme_ref_pruning", C.c_uint8),
        ("prune_ref_if_hme_sad_dev_bigger_than_th", C.c_uint16), ("prune_ref_if_me_sad_dev_bigger_than_th", C.c_uint16),
        ("zz_sad_th", C.c_uint32), ("zz_sad_pct", C.c_uint16), ("phme_sad_th", C.c_uint32), ("phme_sad_pct", C.c_uint16),
        ("enable_me_sr_adjustment", C.c_uint8),
        ("reduce_me_sr_based_on_mv_length_th", C.c_uint16), ("stationary_hme_sad_abs_th", C.c_uint16),
        ("stationary_me_sr_divisor", C.c_uint16), ("reduce_me_sr_based_on_hme_sad_abs_th", C.c_uint16),
        ("me_sr_divisor_for_low_hme_sad", C.c_uint16), ("distance_based_hme_resizing", C.c_uint8),
        ("me_8x8_var_enabled", C.c_uint8),
        ("me_sr_div4_th", C.c_uint32), ("me_sr_div2_th", C.c_uint32), ("me_sr_mult2_th", C.c_uint32),
        ("mv_sa_adj_enabled", C.c_uint8), ("mv_sa_adj_nearest_ref_only", C.c_uint8),
        ("mv_sa_adj_mv_size_th", C.c_uint16), ("mv_sa_adj_sa_multiplier", C.c_uint16),
        ("prune_me_candidates_th", C.c_int32), ("use_best_unipred_cand_only", C.c_uint8),
        ("reduce_hme_l0_sr_th_min", C.c_uint8), ("reduce_hme_l0_sr_th_max", C.c_uint8),
        ("me_early_exit_th", C.c_uint32), ("me_safe_limit_zz_th", C.c_uint32), ("prev_me_stage_based_exit_th", C.c_uint32),
    ]

    def as_dict(self):
        def conv(v):
            if isinstance(v, C.Structure):
                return {n: conv(getattr(v, n)) for n, _ in v._fields_}
            if isinstance(v, C.Array):
                return [conv(x) for x in v]
            return int(v)
        return conv(self)


class MePresetDesc(C.Structure):
    _fields_ = [
        ("enc_mode", C.c_int8), ("input_resolution", C.c_uint8), ("sc_class1", C.c_uint8), ("rtc_tune", C.c_uint8),
        ("temporal_layer_index", C.c_uint8), ("hierarchical_levels", C.c_uint8),
        ("qp", C.c_uint32), ("frame_rate_q16", C.c_uint32), ("safe_limit_nref", C.c_uint8), ("safe_limit_zz_th", C.c_uint32),
    ]


class PlaneDesc(C.Structure):
    _fields_ = [
        ("buffer_y", C.c_void_p), ("stride_y", C.c_uint32), ("org_x", C.c_uint16), ("org_y", C.c_uint16),
        ("width", C.c_uint16), ("height", C.c_uint16),
    ]


class MePictureDesc(C.Structure):
    _fields_ = [
        ("picture_number", C.c_uint64), ("aligned_width", C.c_uint16), ("aligned_height", C.c_uint16),
        ("num_of_list_to_search", C.c_uint8), ("num_of_ref_pic_to_search", C.c_uint8 * MAX_LISTS),
        ("temporal_layer_index", C.c_uint8), ("hierarchical_levels", C.c_uint8), ("is_ref", C.c_uint8),
        ("similar_brightness_refs", C.c_uint8), ("enable_me_8x8", C.c_uint8), ("enable_me_16x16", C.c_uint8),
        ("max_number_of_pus_per_sb", C.c_uint8), ("max_cand", C.c_uint8), ("max_refs", C.c_uint8), ("max_l0", C.c_uint8),
        ("input_resolution", C.c_uint8), ("only_l_bwd", C.c_uint8), ("gm_enabled", C.c_uint8),
        ("gm_use_distance_based_active_th", C.c_uint8), ("b64_row_start", C.c_uint16), ("b64_row_count", C.c_uint16), ("tf_me_exit_th", C.c_uint32),
        ("ref_picture_number", (C.c_uint64 * MAX_REFS) * MAX_LISTS),
    ]


class MeResults(C.Structure):
    _fields_ = [
        ("total_me_candidate_index", C.c_void_p), ("me_mv_array", C.c_void_p), ("me_candidate_array", C.c_void_p),
        ("me_64x64_distortion", C.c_void_p), ("me_32x32_distortion", C.c_void_p), ("me_16x16_distortion", C.c_void_p),
        ("me_8x8_distortion", C.c_void_p), ("rc_me_distortion", C.c_void_p), ("me_8x8_cost_variance", C.c_void_p),
        ("stationary_block_present_sb", C.c_void_p), ("rc_me_allow_gm", C.c_void_p),
        ("sb_best_sad", C.c_void_p), ("sb_best_mv", C.c_void_p), ("hme_sc", C.c_void_p), ("hme_sad", C.c_void_p),
        ("do_ref", C.c_void_p),
    ]


class InvTxBatchDesc(C.Structure):
    """SvtHipInvTxBatchDesc (include/svt_hip_dsp.h)."""
    _fields_ = [("bit_depth", C.c_uint8), ("sample_bytes", C.c_uint8), ("tx_size", C.c_uint8), ("reserved", C.c_uint8), ("n_jobs", C.c_uint32),
                ("pred_stride", C.c_uint32), ("recon_stride", C.c_uint32), ("pred", C.c_void_p), ("recon", C.c_void_p), ("jobs", C.c_void_p),
                ("dqcoeff", C.c_void_p)]


class PredJob(C.Structure):
    """SvtHipPredJob (include/svt_hip_dsp.h)."""
    _fields_ = [("ref", C.c_void_p), ("sb_best_mv", C.c_void_p), ("pred", C.c_void_p), ("b64_row_start", C.c_uint32), ("b64_row_count", C.c_uint32),
                ("list", C.c_uint8), ("ref_idx", C.c_uint8), ("reserved", C.c_uint8 * 6)]


class DgMetrics(C.Structure):
    """SvtHipDgMetrics (include/svt_hip_me.h)."""
    _fields_ = [("tot_dist", C.c_uint64), ("tot_cplx", C.c_uint32), ("tot_active", C.c_uint32), ("sum_in_vectors", C.c_int32),
                ("reserved", C.c_uint32)]

    def as_dict(self):
        return {"tot_dist": self.tot_dist, "tot_cplx": self.tot_cplx, "tot_active": self.tot_active, "sum_in_vectors": self.sum_in_vectors}


class MeJob(C.Structure):
    _fields_ = [("cfg", C.c_void_p), ("desc", C.c_void_p), ("cur", C.c_void_p), ("refs", (C.c_void_p * 4) * 2), ("results", C.c_void_p)]


# (field, numpy dtype, per-b64 element count as a function of (n_pu, max_refs, max_cand))
RESULT_FIELDS = [
    ("total_me_candidate_index", "u1", lambda n, r, c: n),
    ("me_mv_array", "u4", lambda n, r, c: n * r),
    ("me_candidate_array", "u1", lambda n, r, c: n * c),
    ("me_64x64_distortion", "u4", lambda n, r, c: 1), ("me_32x32_distortion", "u4", lambda n, r, c: 1),
    ("me_16x16_distortion", "u4", lambda n, r, c: 1), ("me_8x8_distortion", "u4", lambda n, r, c: 1),
    ("rc_me_distortion", "u4", lambda n, r, c: 1), ("me_8x8_cost_variance", "u4", lambda n, r, c: 1),
    ("stationary_block_present_sb", "u1", lambda n, r, c: 1), ("rc_me_allow_gm", "u1", lambda n, r, c: 1),
    ("sb_best_sad", "u4", lambda n, r, c: 2 * 4 * 85), ("sb_best_mv", "u4", lambda n, r, c: 2 * 4 * 85),
    ("hme_sc", "i2", lambda n, r, c: 2 * 4 * 2), ("hme_sad", "u4", lambda n, r, c: 2 * 4), ("do_ref", "u1", lambda n, r, c: 2 * 4),
]


def n_pu(enable_me_16x16, enable_me_8x8):
    return (85 if enable_me_8x8 else 21) if enable_me_16x16 else 5


# ---- include/svt_hip_dsp.h ------------------------------------------------------------------------------
class QuantRow(C.Structure):
    _fields_ = [("zbin", C.c_int16 * 2), ("round", C.c_int16 * 2), ("quant", C.c_int16 * 2), ("quant_shift", C.c_int16 * 2),
                ("round_fp", C.c_int16 * 2), ("quant_fp", C.c_int16 * 2), ("dequant", C.c_int16 * 2)]


class TxJob(C.Structure):
    _fields_ = [("src_offset", C.c_uint32), ("pred_offset", C.c_uint32), ("tx_type", C.c_uint8), ("quant_row", C.c_uint8),
                ("pf_shape", C.c_uint8), ("reserved", C.c_uint8)]


class RdBatchDesc(C.Structure):
    _fields_ = [("bit_depth", C.c_uint8), ("quant_kind", C.c_uint8), ("tx_size", C.c_uint8), ("reserved", C.c_uint8),
                ("n_jobs", C.c_uint32), ("src_stride", C.c_uint32), ("pred_stride", C.c_uint32),
                ("src", C.c_void_p), ("pred", C.c_void_p), ("recon", C.c_void_p), ("jobs", C.c_void_p), ("quant_rows", C.c_void_p),
                ("n_quant_rows", C.c_uint32),
                ("eob", C.c_void_p), ("satd", C.c_void_p), ("dist_coeff", C.c_void_p), ("three_quad_energy", C.c_void_p), ("sse", C.c_void_p),
                ("coeff", C.c_void_p), ("qcoeff", C.c_void_p), ("dqcoeff", C.c_void_p), ("qmatrix", C.c_void_p), ("iqmatrix", C.c_void_p),
                ("cul_level", C.c_void_p)]


TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
JOB_DTYPE = [("src_offset", "<u4"), ("pred_offset", "<u4"), ("tx_type", "u1"), ("quant_row", "u1"), ("pf_shape", "u1"), ("reserved", "u1")]
QUANT_ROW_DTYPE = [(n, "<i2", (2,)) for n in ("zbin", "round", "quant", "quant_shift", "round_fp", "quant_fp", "dequant")]
RD_OUT_FIELDS = [("eob", "<u2", 1), ("satd", "<u4", 1), ("dist_coeff", "<u8", 2), ("three_quad_energy", "<u8", 1), ("sse", "<u8", 1), ("cul_level", "u1", 1)]


class BlockJob(C.Structure):
    _fields_ = [("src_offset", C.c_uint32), ("ref_offset", C.c_uint32), ("width", C.c_uint8), ("height", C.c_uint8), ("subpel_x", C.c_uint8), ("subpel_y", C.c_uint8)]


class BlockStatsDesc(C.Structure):
    _fields_ = [("bit_depth", C.c_uint8), ("temporal_layer_index", C.c_uint8), ("spy_rd", C.c_uint8), ("reserved", C.c_uint8), ("n_jobs", C.c_uint32),
                ("src_stride", C.c_uint32), ("ref_stride", C.c_uint32),
                ("src", C.c_void_p), ("ref", C.c_void_p), ("jobs", C.c_void_p),
                ("sad", C.c_void_p), ("sse", C.c_void_p), ("variance", C.c_void_p), ("var_sse", C.c_void_p), ("satd", C.c_void_p),
                ("psy_rd", C.c_double), ("psy_energy", C.c_void_p), ("psy_dist", C.c_void_p),
                ("psy_sse", C.c_void_p), ("pred_mode", C.c_void_p), ("compound_type", C.c_void_p), ("facade_dist", C.c_void_p),
                ("variance10", C.c_void_p), ("var_sse10", C.c_void_p),
                ("n_pyramids", C.c_uint32), ("pyramid_out_base", C.c_uint32), ("pyramids", C.c_void_p)]


PYRAMID_BLOCKS = 85  # nested square blocks of a 64x64 region: 1 + 4 + 16 + 64, each level in raster order
BLOCK_JOB_DTYPE = [("src_offset", "<u4"), ("ref_offset", "<u4"), ("width", "u1"), ("height", "u1"), ("subpel_x", "u1"), ("subpel_y", "u1")]
STATS_OUT_FIELDS = [("sad", "<u4"), ("sse", "<u8"), ("variance", "<u4"), ("var_sse", "<u4"), ("satd", "<u4")]
PSY_OUT_FIELDS = [("psy_energy", "<u8"), ("psy_dist", "<u8"), ("psy_sse", "<u8")]
FACADE_OUT_FIELDS = [("facade_dist", "<u8")]
VAR10_OUT_FIELDS = [("variance10", "<u4"), ("var_sse10", "<u4")]  # 10-bit planes only
VARIANCE_SIZES = [(4, 4), (4, 8), (4, 16), (8, 4), (8, 8), (8, 16), (8, 32), (16, 4), (16, 8), (16, 16), (16, 32), (16, 64), (32, 8), (32, 16),
                  (32, 32), (32, 64), (64, 16), (64, 32), (64, 64), (64, 128), (128, 64), (128, 128)]


# ---- include/svt_hip_pme.h ----
class Mv(C.Structure):
    _fields_ = [("row", C.c_int16), ("col", C.c_int16)]


class MvCostParam(C.Structure):  # MV_COST_PARAMS, Codec/mcomp.h:37-48
    _fields_ = [("ref_mv", C.POINTER(Mv)), ("full_ref_mv", Mv), ("mv_cost_type", C.c_uint8), ("mvjcost", C.c_void_p), ("mvcost", C.c_void_p * 2),
                ("error_per_bit", C.c_int), ("early_exit_th", C.c_int), ("sad_per_bit", C.c_int)]


PME_JOB_DTYPE = np.dtype([("src_offset", "<u4"), ("ref_offset", "<u4"), ("width", "u1"), ("height", "u1"), ("start_x", "<i2"), ("start_y", "<i2"), ("sa_w", "<i2"),
                          ("sa_h", "<i2"), ("step", "<i2"), ("mvx", "<i2"), ("mvy", "<i2"), ("ref_mv", "<i2", (2,)), ("best_cost", "<u4"), ("best_mvx", "<i2"),
                          ("best_mvy", "<i2")], align=True)


class PmeBatchDesc(C.Structure):
    _fields_ = [("n_jobs", C.c_uint32), ("src_stride", C.c_uint32), ("ref_stride", C.c_uint32), ("src", C.c_void_p), ("ref", C.c_void_p), ("jobs", C.c_void_p),
                ("mv_cost_type", C.c_int32), ("error_per_bit", C.c_int32), ("mvjcost", C.c_void_p), ("mvcost", C.c_void_p * 2), ("best_cost", C.c_void_p),
                ("best_mv", C.c_void_p)]


# ---- include/svt_hip_md_search.h ----
FULLPEL_JOB_DTYPE = np.dtype([("src_offset", "<u4"), ("blk_org_x", "<i4"), ("blk_org_y", "<i4"), ("width", "u1"), ("height", "u1"), ("dist_type", "u1"), ("flags", "u1"),
                              ("mvx", "<i2"), ("mvy", "<i2"), ("start_x", "<i2"), ("end_x", "<i2"), ("start_y", "<i2"), ("end_y", "<i2"), ("step", "<i2"),
                              ("sprs_lev0_start_x", "<i2"), ("sprs_lev0_end_x", "<i2"), ("sprs_lev0_start_y", "<i2"), ("sprs_lev0_end_y", "<i2"), ("ref_mv", "<i2", (2,)),
                              ("best_cost", "<u4"), ("best_mvx", "<i2"), ("best_mvy", "<i2"), ("chain_from", "<i4")], align=True)
FP_CENTRE_FROM_CHAIN, FP_BEST_FROM_CHAIN, FP_SPRS_LEV0_DONE, FP_ENABLE_PSAD = 1, 2, 4, 8


class FullpelBatchDesc(C.Structure):
    _fields_ = [("n_jobs", C.c_uint32), ("src_stride", C.c_uint32), ("ref_stride", C.c_uint32), ("src", C.c_void_p), ("ref", C.c_void_p), ("ref_org_x", C.c_int32),
                ("ref_org_y", C.c_int32), ("ref_max_width", C.c_int32), ("ref_max_height", C.c_int32), ("jobs", C.c_void_p), ("mv_cost_type", C.c_int32),
                ("error_per_bit", C.c_int32), ("mvjcost", C.c_void_p), ("mvcost", C.c_void_p * 2), ("best_cost", C.c_void_p), ("best_mv", C.c_void_p)]


SUBPEL_JOB_DTYPE = np.dtype([("src_offset", "<u4"), ("ref_offset", "<u4"), ("width", "u1"), ("height", "u1"), ("log2_pels", "u1"), ("early_neigh_check_exit", "u1"),
                             ("start_mv", "<i2", (2,)), ("ref_mv", "<i2", (2,)), ("col_min", "<i2"), ("col_max", "<i2"), ("row_min", "<i2"), ("row_max", "<i2"),
                             ("early_exit_th", "<i4"), ("best_mvp_dist", "<u4"), ("best_mvp", "<i2", (2,))], align=True)
USE_2_TAPS, USE_4_TAPS, USE_8_TAPS = 1, 2, 3


class SubpelBatchDesc(C.Structure):
    _fields_ = [("n_jobs", C.c_uint32), ("src_stride", C.c_uint32), ("ref_stride", C.c_uint32), ("src", C.c_void_p), ("ref", C.c_void_p), ("jobs", C.c_void_p),
                ("allow_hp", C.c_int32), ("forced_stop", C.c_int32), ("iters_per_step", C.c_int32), ("pred_variance_th", C.c_int32), ("abs_th_mult", C.c_int32),
                ("round_dev_th", C.c_int32), ("skip_diag_refinement", C.c_int32), ("bias_fp", C.c_int32), ("qp", C.c_int32), ("search_method", C.c_int32),
                ("subpel_search_type", C.c_int32), ("mvp_th", C.c_int32), ("hp_mv_th", C.c_int32), ("mv_cost_type", C.c_int32),
                ("error_per_bit", C.c_int32), ("mvjcost", C.c_void_p), ("mvcost", C.c_void_p * 2), ("best_mv", C.c_void_p), ("besterr", C.c_void_p),
                ("distortion", C.c_void_p), ("sse", C.c_void_p), ("center_err", C.c_void_p)]
