/*
 * svt_hip_md_search.h -- C-ABI of the mode-decision side motion search: the consumer of the open-loop ME results.
 *
 *   svt_hip_md_fullpel_batch  = md_full_pel_search (Codec/product_coding_loop.c:2042-2180; wide 8-bit SAD searches through
 *                               md_full_pel_search_large_lbd :1958-2027 = svt_pme_sad_loop_kernel :1905-1950), one job per call of it,
 *                               jobs chained ON THE DEVICE the way md_nsq_motion_search / md_sq_motion_search chain their calls (:2260-2375:
 *                               the candidate centres, the step-4 area, the +-2 and +-1 refinements each start from the previous best)
 *   svt_hip_md_subpel_batch   = the fractional_mv_step_fp md_subpel_search calls (product_coding_loop.c:2637-2750, :2723-2725):
 *                               search_method 0: svt_av1_find_best_sub_pixel_tree_pruned (Codec/mcomp.c:606-687), the tree search on the bilinear
 *                               sub-pixel variance (vfp->vf / vfp->svf, Codec/av1me.c:29-172, C_DEFAULT/variance.c:28-75,256-318);
 *                               search_method 1: svt_av1_find_best_sub_pixel_tree (:688-777), the accurate search -- every candidate's error
 *                               is vfp->vf of svt_aom_upsampled_pred (C_DEFAULT/variance.c:204-254: separable 2 / 4 / 8-tap interpolation,
 *                               svt_aom_convolve8_horiz / _vert, Codec/convolve.c:244-301) -- with the round limits it derives from the MVP
 *                               distance when its context says so (:702-722); both with the MV-rate cost of Codec/mcomp.c:44-78.
 *
 * Out of scope (said so in DESIGN.md): the SSD distortion type of md_full_pel_search (PSYEX adds get_svt_psy_full_dist with a double factor
 * per position), the 16-bit (hbd_md) forms.
 *
 * Asynchronous like the other batched entries: device pointers, enqueued on the context stream, one wave per job.
 */
#ifndef SVT_HIP_MD_SEARCH_H
#define SVT_HIP_MD_SEARCH_H

#include <stddef.h>
#include <stdint.h>
#include "svt_hip_me.h"
#include "svt_hip_pme.h"

#ifdef __cplusplus
extern "C" {
#endif

/* DistortionType of md_full_pel_search (Codec/definitions.h): SAD and VAR are built */
#define SVT_HIP_DIST_SAD 0
#define SVT_HIP_DIST_VAR 1

#define SVT_HIP_FP_CENTRE_FROM_CHAIN 1 /* (mvx, mvy) = the best MV job `chain_from` left (its outputs, a job of an EARLIER batch or an earlier index of this one) */
#define SVT_HIP_FP_BEST_FROM_CHAIN 2   /* *best_cost / *best_mvx / *best_mvy on entry = job `chain_from`'s outputs */
#define SVT_HIP_FP_SPRS_LEV0_DONE 4    /* is_sprs_lev0_performed */
#define SVT_HIP_FP_ENABLE_PSAD 8       /* ctx->enable_psad */

/* One md_full_pel_search call */
typedef struct SvtHipFullpelJob {
    uint32_t src_offset;           /* input_origin_index: the block's top-left sample in the source plane */
    int32_t  blk_org_x, blk_org_y; /* ctx->blk_org_x / _y */
    uint8_t  width, height;        /* blk_geom->bwidth / bheight */
    uint8_t  dist_type;            /* SVT_HIP_DIST_* */
    uint8_t  flags;                /* SVT_HIP_FP_* */
    int16_t  mvx, mvy;             /* the search centre in 1/8 sample (unless CENTRE_FROM_CHAIN) */
    int16_t  start_x, end_x, start_y, end_y; /* search_position_start / end (full samples, inclusive) */
    int16_t  step;                 /* sparse_search_step >= 1 */
    int16_t  sprs_lev0_start_x, sprs_lev0_end_x, sprs_lev0_start_y, sprs_lev0_end_y; /* ctx->sprs_lev0_* (read when SPRS_LEV0_DONE and step == 2) */
    SvtHipMv ref_mv;               /* ctx->ref_mv: what the MV-rate is measured against */
    uint32_t best_cost;            /* *best_cost on entry (unless BEST_FROM_CHAIN) */
    int16_t  best_mvx, best_mvy;   /* *best_mvx / *best_mvy on entry */
    int32_t  chain_from;           /* job index whose outputs feed this one (flags), or -1 */
} SvtHipFullpelJob;

typedef struct SvtHipFullpelBatchDesc {
    uint32_t n_jobs;
    uint32_t src_stride, ref_stride;       /* in samples */
    const uint8_t *src;                    /* device pointer: input_pic->buffer_y (8-bit) */
    const uint8_t *ref;                    /* device pointer: ref_pic->buffer_y, the padded plane's first byte */
    int32_t  ref_org_x, ref_org_y, ref_max_width, ref_max_height; /* EbPictureBufferDesc fields of ref_pic */
    const SvtHipFullpelJob *jobs;          /* device pointer */
    int32_t  mv_cost_type, error_per_bit;  /* mv_cost_params (svt_init_mv_cost_params, product_coding_loop.c:2029-2041) */
    const int32_t *mvjcost;                /* device pointer, 4 entries (MV_COST_ENTROPY only) */
    const int32_t *mvcost[2];              /* device pointers to the CENTRE entries of the row / column tables, indices [-16384, 16384] */
    uint32_t *best_cost;                   /* in / out, device pointers: [n_jobs] (chains read earlier jobs' entries) */
    int16_t  *best_mv;                     /* [n_jobs][2] = (best_mvx, best_mvy) */
} SvtHipFullpelBatchDesc;

/* Jobs of one batch run concurrently: a job may only chain from a job of a batch enqueued earlier on the stream (same output arrays). */
int svt_hip_md_fullpel_batch(SvtHipContext *ctx, const SvtHipFullpelBatchDesc *d);

#define SVT_HIP_USE_2_TAPS 1
#define SVT_HIP_USE_4_TAPS 2
#define SVT_HIP_USE_8_TAPS 3

/* One call of the sub-pel search function */
typedef struct SvtHipSubpelJob {
    uint32_t src_offset;     /* ms_buffers->src->buf: the block's top-left sample in the source plane */
    uint32_t ref_offset;     /* ms_buffers->ref->buf: the co-located sample in the reference plane (MV (0, 0)) */
    uint8_t  width, height;  /* var_params.w / h: one of the svt_aom_mefn_ptr block sizes */
    uint8_t  log2_pels;      /* num_pels_log2_lookup[bsize] */
    uint8_t  early_neigh_check_exit;
    SvtHipMv start_mv;       /* full-sample precision, 1/8 units */
    SvtHipMv ref_mv;
    int16_t  col_min, col_max, row_min, row_max; /* SubpelMvLimits */
    int32_t  early_exit_th;  /* mv_cost_params.early_exit_th (1020 - (sq_size >> 2)) */
    /* search_method 1 with mvp_th > 0 (the function's PD_PASS_1 / SPEL_ME branch, mcomp.c:702-722): ctx->best_fp_mvp_dist[list][ref] and
     * ctx->mvp_array[list][ref][best_fp_mvp_idx] of the searched reference */
    uint32_t best_mvp_dist;
    SvtHipMv best_mvp;
} SvtHipSubpelJob;

typedef struct SvtHipSubpelBatchDesc {
    uint32_t n_jobs;
    uint32_t src_stride, ref_stride;
    const uint8_t *src, *ref;             /* device pointers, 8-bit planes */
    const SvtHipSubpelJob *jobs;          /* device pointer */
    /* SUBPEL_MOTION_SEARCH_PARAMS (Codec/mcomp.h:85-104) */
    int32_t  allow_hp, forced_stop, iters_per_step, pred_variance_th, abs_th_mult, round_dev_th, skip_diag_refinement, bias_fp;
    int32_t  qp;                          /* pcs->picture_qp */
    int32_t  search_method;               /* 0: svt_av1_find_best_sub_pixel_tree_pruned, 1: svt_av1_find_best_sub_pixel_tree (md_subpel_ctrls.subpel_search_method == SUBPEL_TREE) */
    int32_t  subpel_search_type;          /* search_method 1: var_params.subpel_search_type -- SVT_HIP_USE_2_TAPS / _4_TAPS / _8_TAPS (definitions.h:728-733) */
    int32_t  mvp_th, hp_mv_th;            /* search_method 1: ctx->md_subpel_me_ctrls.mvp_th / hp_mv_th when ctx->pd_pass == PD_PASS_1 and the stage is SPEL_ME, else 0 (branch off) */
    int32_t  mv_cost_type, error_per_bit;
    const int32_t *mvjcost;
    const int32_t *mvcost[2];
    /* outputs, device pointers */
    int16_t  *best_mv;                    /* [n_jobs][2] = (row, col) */
    uint32_t *besterr;                    /* [n_jobs]: the function's return value */
    int32_t  *distortion;                 /* [n_jobs] */
    uint32_t *sse;                        /* [n_jobs]: *sse1 (0 when no candidate improved on the start) */
    uint32_t *center_err;                 /* optional, [n_jobs]: the error at the start MV = what the functions store into ctx->fp_me_dist[list][ref] */
} SvtHipSubpelBatchDesc;

int svt_hip_md_subpel_batch(SvtHipContext *ctx, const SvtHipSubpelBatchDesc *d);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_MD_SEARCH_H */
