/*
 * ref_harness_md.c -- TEST INFRASTRUCTURE.  Drives the REFERENCE's own mode-decision side motion search for tests/test_md_search.py and
 * oracle/gen_golden.py:
 *   md_full_pel_search (Codec/product_coding_loop.c:2042-2180) is `static`: this translation unit therefore compiles product_coding_loop.c
 *   where it lies by including it (oracle/Makefile leaves it out of the list of separately compiled files), and adds a driver that fills the
 *   reference's own structs and calls it;
 *   svt_av1_find_best_sub_pixel_tree_pruned (Codec/mcomp.c:606-687) is called as md_subpel_search calls it (product_coding_loop.c:2637-2750),
 *   with vfp = &svt_aom_mefn_ptr[bsize] as init_fn_ptr() (Codec/av1me.c:31) fills it from the `_c` variance kernels.
 * Nothing of the reference is copied; the product never links this.
 */
#include "product_coding_loop.c"
#include "../include/svt_hip_md_search.h"

void init_fn_ptr(void);

static BlockSize bsize_of(int w, int h) {
    for (int b = 0; b < BlockSizeS_ALL; b++)
        if (block_size_wide[b] == w && block_size_high[b] == h) return (BlockSize)b;
    return BLOCK_INVALID;
}

static void setup_variance_pointers(void) {
    static int done = 0;
    if (done) return;
    /* what svt_aom_setup_rtcd_internal (Codec/aom_dsp_rtcd.c:188) stores for flags == 0, for the pointers init_fn_ptr() copies into the
     * vf / svf members this path calls (the function itself also names every SIMD body, most of which this build does not hold) */
#define VARP(W, H) svt_aom_variance##W##x##H = svt_aom_variance##W##x##H##_c; svt_aom_sub_pixel_variance##W##x##H = svt_aom_sub_pixel_variance##W##x##H##_c;
    VARP(4, 4) VARP(4, 8) VARP(4, 16) VARP(8, 4) VARP(8, 8) VARP(8, 16) VARP(8, 32) VARP(16, 4) VARP(16, 8) VARP(16, 16) VARP(16, 32) VARP(16, 64) VARP(32, 8)
    VARP(32, 16) VARP(32, 32) VARP(32, 64) VARP(64, 16) VARP(64, 32) VARP(64, 64) VARP(64, 128) VARP(128, 64) VARP(128, 128)
#undef VARP
    svt_aom_upsampled_pred = svt_aom_upsampled_pred_c; /* the accurate sub-pel search's predictor (C_DEFAULT/variance.c:204) ... */
    svt_memcpy = svt_memcpy_c;                         /* ... which copies full-pel positions through this pointer (:216) */
    init_fn_ptr();
    done = 1;
}

/* One md_full_pel_search call with the arguments / context fields a SvtHipFullpelJob carries.  mv_cost_type reaches the reference through
 * md_subpel_me_ctrls.skip_diag_refinement (svt_init_mv_cost_params, :2029-2041: MV_COST_OPT when >= 3, else MV_COST_ENTROPY). */
int ref_md_full_pel_search(const SvtHipFullpelBatchDesc *d, const SvtHipFullpelJob *jb, int16_t mvx, int16_t mvy, uint32_t *best_cost, int16_t *best_mvx, int16_t *best_mvy) {
    setup_variance_pointers();
    svt_pme_sad_loop_kernel = svt_pme_sad_loop_kernel_c;
    if (d->mv_cost_type != MV_COST_ENTROPY && d->mv_cost_type != MV_COST_OPT) return 1;
    PictureControlSet       *pcs  = (PictureControlSet *)calloc(1, sizeof(*pcs));
    PictureParentControlSet *ppcs = (PictureParentControlSet *)calloc(1, sizeof(*ppcs));
    SequenceControlSet      *scs  = (SequenceControlSet *)calloc(1, sizeof(*scs));
    ModeDecisionContext     *ctx  = (ModeDecisionContext *)calloc(1, sizeof(*ctx));
    MdRateEstimationContext *rate = (MdRateEstimationContext *)calloc(1, sizeof(*rate));
    BlockGeom                geom;
    EbPictureBufferDesc      in, ref;
    if (!pcs || !ppcs || !scs || !ctx || !rate) return 2;
    memset(&geom, 0, sizeof(geom)); memset(&in, 0, sizeof(in)); memset(&ref, 0, sizeof(ref));
    pcs->ppcs = ppcs; pcs->scs = scs; pcs->slice_type = B_SLICE; pcs->temporal_layer_index = 1;
    geom.bwidth = jb->width; geom.bheight = jb->height; geom.sq_size = jb->width > jb->height ? jb->width : jb->height; geom.bsize = bsize_of(jb->width, jb->height);
    if (geom.bsize == BLOCK_INVALID) return 3;
    ctx->blk_geom = &geom;
    ctx->blk_org_x = (uint16_t)jb->blk_org_x; ctx->blk_org_y = (uint16_t)jb->blk_org_y;
    ctx->ref_mv.row = jb->ref_mv.row; ctx->ref_mv.col = jb->ref_mv.col;
    ctx->enable_psad = (jb->flags & SVT_HIP_FP_ENABLE_PSAD) ? 1 : 0;
    ctx->sprs_lev0_start_x = jb->sprs_lev0_start_x; ctx->sprs_lev0_end_x = jb->sprs_lev0_end_x;
    ctx->sprs_lev0_start_y = jb->sprs_lev0_start_y; ctx->sprs_lev0_end_y = jb->sprs_lev0_end_y;
    ctx->md_subpel_me_ctrls.skip_diag_refinement = d->mv_cost_type == MV_COST_OPT ? 3 : 0;
    /* error_per_bit = AOMMAX(rdmult >> RD_EPB_SHIFT, 1) */
    for (int k = 0; k < 2; k++) ctx->full_lambda_md[k] = ctx->fast_lambda_md[k] = (uint32_t)d->error_per_bit << RD_EPB_SHIFT;
    ctx->md_rate_est_ctx = rate;
    if (d->mvjcost)
        for (int k = 0; k < MV_JOINTS; k++) rate->nmv_vec_cost[k] = d->mvjcost[k];
    rate->nmvcoststack[0] = (int32_t *)d->mvcost[0];
    rate->nmvcoststack[1] = (int32_t *)d->mvcost[1];
    in.buffer_y = (uint8_t *)d->src; in.stride_y = (uint16_t)d->src_stride;
    ref.buffer_y = (uint8_t *)d->ref; ref.stride_y = (uint16_t)d->ref_stride; ref.org_x = (uint16_t)d->ref_org_x; ref.org_y = (uint16_t)d->ref_org_y;
    ref.max_width = (uint16_t)d->ref_max_width; ref.max_height = (uint16_t)d->ref_max_height;
    md_full_pel_search(pcs, ctx, &in, &ref, jb->src_offset, (DistortionType)jb->dist_type, mvx, mvy, jb->start_x, jb->end_x, jb->start_y, jb->end_y, jb->step,
                       (jb->flags & SVT_HIP_FP_SPRS_LEV0_DONE) ? 1 : 0, best_mvx, best_mvy, best_cost, 0);
    free(pcs); free(ppcs); free(scs); free(ctx); free(rate);
    return 0;
}

/* host-memory mirror of svt_hip_md_fullpel_batch through the reference (jobs in order: a job may chain from an earlier one) */
int ref_md_fullpel_batch(const SvtHipFullpelBatchDesc *d) {
    for (uint32_t j = 0; j < d->n_jobs; j++) {
        const SvtHipFullpelJob *jb = &d->jobs[j];
        int16_t  mvx = jb->mvx, mvy = jb->mvy, bx = jb->best_mvx, by = jb->best_mvy;
        uint32_t best = jb->best_cost;
        if (jb->flags & SVT_HIP_FP_CENTRE_FROM_CHAIN) { mvx = d->best_mv[2 * jb->chain_from]; mvy = d->best_mv[2 * jb->chain_from + 1]; }
        if (jb->flags & SVT_HIP_FP_BEST_FROM_CHAIN) { best = d->best_cost[jb->chain_from]; bx = d->best_mv[2 * jb->chain_from]; by = d->best_mv[2 * jb->chain_from + 1]; }
        const int rc = ref_md_full_pel_search(d, jb, mvx, mvy, &best, &bx, &by);
        if (rc) return rc;
        d->best_cost[j] = best; d->best_mv[2 * j] = bx; d->best_mv[2 * j + 1] = by;
    }
    return 0;
}

/* svt_av1_find_best_sub_pixel_tree_pruned / svt_av1_find_best_sub_pixel_tree (search_method) on the jobs of a SvtHipSubpelBatchDesc.  The
 * context argument is the reference's own ModeDecisionContext with the fields the functions read (the tree search's PD_PASS_1 branch,
 * mcomp.c:702-722) and write (fp_me_dist) */
int ref_md_subpel_batch(const SvtHipSubpelBatchDesc *d) {
    setup_variance_pointers();
    ModeDecisionContext *ctx = (ModeDecisionContext *)calloc(1, sizeof(*ctx));
    MacroBlockD          xd;
    if (!ctx) return 2;
    memset(&xd, 0, sizeof(xd));
    ctx->pd_pass = d->mvp_th > 0 ? PD_PASS_1 : PD_PASS_0;
    ctx->md_subpel_me_ctrls.mvp_th = (uint8_t)d->mvp_th; ctx->md_subpel_me_ctrls.hp_mv_th = d->hp_mv_th;
    for (uint32_t j = 0; j < d->n_jobs; j++) {
        const SvtHipSubpelJob      *jb = &d->jobs[j];
        SUBPEL_MOTION_SEARCH_PARAMS ms;
        memset(&ms, 0, sizeof(ms));
        const BlockSize bsize = bsize_of(jb->width, jb->height);
        if (bsize == BLOCK_INVALID) return 3;
        MV ref_mv = {jb->ref_mv.row, jb->ref_mv.col};
        ms.allow_hp = d->allow_hp; ms.forced_stop = (SUBPEL_FORCE_STOP)d->forced_stop; ms.iters_per_step = d->iters_per_step; ms.pred_variance_th = d->pred_variance_th;
        ms.abs_th_mult = (uint8_t)d->abs_th_mult; ms.round_dev_th = d->round_dev_th; ms.skip_diag_refinement = (uint8_t)d->skip_diag_refinement;
        ms.search_stage = SPEL_ME;
        ms.mv_limits.col_min = jb->col_min; ms.mv_limits.col_max = jb->col_max; ms.mv_limits.row_min = jb->row_min; ms.mv_limits.row_max = jb->row_max;
        ms.mv_cost_params.ref_mv = &ref_mv; ms.mv_cost_params.mv_cost_type = (MV_COST_TYPE)d->mv_cost_type; ms.mv_cost_params.mvjcost = (const int *)d->mvjcost;
        ms.mv_cost_params.mvcost[0] = (const int *)d->mvcost[0]; ms.mv_cost_params.mvcost[1] = (const int *)d->mvcost[1];
        ms.mv_cost_params.error_per_bit = d->error_per_bit; ms.mv_cost_params.early_exit_th = jb->early_exit_th;
        ms.var_params.vfp = &svt_aom_mefn_ptr[bsize]; ms.var_params.w = jb->width; ms.var_params.h = jb->height; ms.var_params.bias_fp = d->bias_fp;
        ms.var_params.subpel_search_type = (SUBPEL_SEARCH_TYPE)d->subpel_search_type;
        ms.list_idx = 0; ms.ref_idx = 0;
        ctx->best_fp_mvp_dist[0][0] = jb->best_mvp_dist; ctx->best_fp_mvp_idx[0][0] = 0;
        ctx->mvp_array[0][0][0].row = jb->best_mvp.row; ctx->mvp_array[0][0][0].col = jb->best_mvp.col;
        struct svt_buf_2d src_b, ref_b;
        src_b.buf = (uint8_t *)d->src + jb->src_offset; src_b.stride = (int)d->src_stride; src_b.width = src_b.height = 0;
        ref_b.buf = (uint8_t *)d->ref + jb->ref_offset; ref_b.stride = (int)d->ref_stride; ref_b.width = ref_b.height = 0;
        ms.var_params.ms_buffers.src = &src_b; ms.var_params.ms_buffers.ref = &ref_b;
        MV           start = {jb->start_mv.row, jb->start_mv.col}, best = {0, 0};
        int          dist = 0;
        unsigned int sse1 = 0;
        fractional_mv_step_fp *search = d->search_method == 1 ? svt_av1_find_best_sub_pixel_tree : svt_av1_find_best_sub_pixel_tree_pruned; /* product_coding_loop.c:2723-2725 */
        d->besterr[j] = (uint32_t)search(ctx, &xd, NULL, &ms, start, &best, &dist, &sse1, d->qp, bsize, jb->early_neigh_check_exit);
        d->best_mv[2 * j] = best.row; d->best_mv[2 * j + 1] = best.col;
        d->distortion[j] = dist; d->sse[j] = sse1;
        if (d->center_err) d->center_err[j] = ctx->fp_me_dist[0][0];
    }
    free(ctx);
    return 0;
}
