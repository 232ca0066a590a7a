/*
 * md_search_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the mode-decision side motion search (include/svt_hip_md_search.h):
 *   md_full_pel_search + md_full_pel_search_large_lbd            Codec/product_coding_loop.c:2042-2180, :1958-2027
 *   svt_av1_find_best_sub_pixel_tree_pruned and its helpers      Codec/mcomp.c:94-687 (the is_scaled == 0 path: svt_estimated_pref_error)
 *   svt_aom_variance{W}x{H}_c / svt_aom_sub_pixel_variance{W}x{H}_c   C_DEFAULT/variance.c:28-75,256-318
 * Pinned on the reference's own functions in tests/test_md_search.py (oracle/ref_harness_md.c drives them).
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/svt_hip_md_search.h"

int  orc_mv_err_cost(int16_t mv_row, int16_t mv_col, const SvtHipMvCostParam *p); /* pme_oracle.c */
void orc_pme_sad_loop_kernel(const SvtHipMvCostParam *p, const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                             uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx, int16_t *best_mvy, int16_t start_x, int16_t start_y, int16_t sa_w,
                             int16_t sa_h, int16_t step, int16_t mvx, int16_t mvy);

/* variance(), variance.c:256-276: a = the first operand of vf / the filtered block, b = the second */
static uint32_t variance_wxh(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int w, int h, uint32_t *sse) {
    int      sum = 0;
    uint32_t s2  = 0;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const int diff = a[i * a_stride + j] - b[i * b_stride + j];
            sum += diff;
            s2 += (uint32_t)(diff * diff);
        }
    *sse = s2;
    return s2 - (uint32_t)(((int64_t)sum * sum) / (w * h));
}

/* svt_aom_sub_pixel_variance{W}x{H}_c, variance.c:28-75,308-318: two bilinear passes (taps 128 - 16 k, 16 k; ROUND_POWER_OF_TWO(.., 7)) */
static uint32_t sub_pixel_variance_wxh(const uint8_t *a, int a_stride, int xoffset, int yoffset, const uint8_t *b, int b_stride, int w, int h, uint32_t *sse) {
    uint16_t fdata3[(128 + 1) * 128];
    uint8_t  temp2[128 * 128];
    const int fx0 = 128 - 16 * xoffset, fx1 = 16 * xoffset, fy0 = 128 - 16 * yoffset, fy1 = 16 * yoffset;
    for (int i = 0; i < h + 1; i++)
        for (int j = 0; j < w; j++) fdata3[i * w + j] = (uint16_t)(((int)a[i * a_stride + j] * fx0 + (int)a[i * a_stride + j + 1] * fx1 + 64) >> 7);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) temp2[i * w + j] = (uint8_t)(((int)fdata3[i * w + j] * fy0 + (int)fdata3[(i + 1) * w + j] * fy1 + 64) >> 7);
    return variance_wxh(temp2, w, b, b_stride, w, h, sse);
}

/* ---------------------------------------------------------------------------------------------------------------------------------- */
/* md_full_pel_search (product_coding_loop.c:2042-2180), dist_type SAD / VAR on 8-bit planes                                           */
/* ---------------------------------------------------------------------------------------------------------------------------------- */
static void full_pel_search(const SvtHipFullpelBatchDesc *d, const SvtHipFullpelJob *jb, const SvtHipMvCostParam *mp, int16_t mvx, int16_t mvy, uint32_t *best_cost,
                            int16_t *best_mvx, int16_t *best_mvy) {
    int16_t sx = jb->start_x, ex = jb->end_x, sy = jb->start_y, ey = jb->end_y;
    const int bx = jb->blk_org_x, by = jb->blk_org_y, bw = jb->width, bh = jb->height, step = jb->step;
    /* search area adjustment (:2060-2076) */
    if ((bx + (mvx >> 3) + sx) < (-d->ref_org_x + 1)) sx = (int16_t)((-d->ref_org_x + 1) - (bx + (mvx >> 3)));
    if ((bx + bw + (mvx >> 3) + ex) > (d->ref_org_x + d->ref_max_width - 1)) ex = (int16_t)((d->ref_org_x + d->ref_max_width - 1) - (bx + bw + (mvx >> 3)));
    if ((by + (mvy >> 3) + sy) < (-d->ref_org_y + 1)) sy = (int16_t)((-d->ref_org_y + 1) - (by + (mvy >> 3)));
    if ((by + bh + (mvy >> 3) + ey) > (d->ref_org_y + d->ref_max_height - 1)) ey = (int16_t)((d->ref_org_y + d->ref_max_height - 1) - (by + bh + (mvy >> 3)));
    const uint8_t *src = d->src + jb->src_offset;
    if (jb->dist_type == SVT_HIP_DIST_SAD && (jb->flags & SVT_HIP_FP_ENABLE_PSAD) && (ex - sx) >= 7) {
        /* md_full_pel_search_large_lbd (:1958-2027): the area's width rounded up to a multiple of 8, one svt_pme_sad_loop_kernel call */
        const int32_t ref_origin_index = d->ref_org_x + (bx + (mvx >> 3) + sx) + (by + (mvy >> 3) + d->ref_org_y + sy) * (int32_t)d->ref_stride;
        int16_t remain = (int16_t)(8 - ((ex - sx) % 8));
        remain         = remain == 8 ? 0 : remain;
        ex             = (int16_t)(ex > ex + remain ? ex : ex + remain);
        const uint32_t sa_w = (uint32_t)(ex - sx), sa_h = (uint32_t)(ey - sy + 1);
        if (sa_w & 0xfffffff8u)
            orc_pme_sad_loop_kernel(mp, src, d->src_stride, d->ref + ref_origin_index, d->ref_stride, (uint32_t)bh, (uint32_t)bw, best_cost, best_mvx, best_mvy, sx, sy,
                                    (int16_t)(sa_w & 0xfffffff8u), (int16_t)sa_h, (int16_t)step, mvx, mvy);
        return;
    }
    for (int32_t px = sx; px <= ex; px += step)
        for (int32_t py = sy; py <= ey; py += step) {
            if (step == 2 && (jb->flags & SVT_HIP_FP_SPRS_LEV0_DONE)) /* sparse level 1 skips what level 0 visited (:2099-2109) */
                if ((px + (mvx >> 3)) >= jb->sprs_lev0_start_x && (px + (mvx >> 3)) <= jb->sprs_lev0_end_x && (py + (mvy >> 3)) >= jb->sprs_lev0_start_y &&
                    (py + (mvy >> 3)) <= jb->sprs_lev0_end_y)
                    if (px % 4 == 0 && py % 4 == 0) continue;
            const int32_t  ref_origin_index = d->ref_org_x + (bx + (mvx >> 3) + px) + (by + (mvy >> 3) + d->ref_org_y + py) * (int32_t)d->ref_stride;
            const uint8_t *pred             = d->ref + ref_origin_index;
            uint64_t       cost;
            if (jb->dist_type == SVT_HIP_DIST_VAR) {
                uint32_t sse;
                cost = variance_wxh(pred, (int)d->ref_stride, src, (int)d->src_stride, bw, bh, &sse); /* fn_ptr->vf(pred, .., src, .., &sse) */
            } else {
                uint32_t sad = 0;
                for (int y = 0; y < bh; y++)
                    for (int x = 0; x < bw; x++) sad += (uint32_t)abs((int)src[y * d->src_stride + x] - (int)pred[y * d->ref_stride + x]);
                cost = sad;
            }
            const int16_t col = (int16_t)(mvx + (px * 8)), row = (int16_t)(mvy + (py * 8));
            cost += (uint64_t)(int64_t)orc_mv_err_cost(row, col, mp);
            if (cost < *best_cost) { *best_mvx = col; *best_mvy = row; *best_cost = (uint32_t)cost; }
        }
}

int orc_md_fullpel_batch(const SvtHipFullpelBatchDesc *d) {
    for (uint32_t j = 0; j < d->n_jobs; j++) {
        const SvtHipFullpelJob *jb = &d->jobs[j];
        SvtHipMvCostParam       p;
        memset(&p, 0, sizeof(p));
        p.ref_mv = &jb->ref_mv; p.mv_cost_type = (uint8_t)d->mv_cost_type; p.mvjcost = d->mvjcost; p.mvcost[0] = d->mvcost[0]; p.mvcost[1] = d->mvcost[1];
        p.error_per_bit = d->error_per_bit;
        int16_t  mvx = jb->mvx, mvy = jb->mvy, bx = jb->best_mvx, by = jb->best_mvy;
        uint32_t best = jb->best_cost;
        if (jb->flags & SVT_HIP_FP_CENTRE_FROM_CHAIN) { mvx = d->best_mv[2 * jb->chain_from]; mvy = d->best_mv[2 * jb->chain_from + 1]; }
        if (jb->flags & SVT_HIP_FP_BEST_FROM_CHAIN) { best = d->best_cost[jb->chain_from]; bx = d->best_mv[2 * jb->chain_from]; by = d->best_mv[2 * jb->chain_from + 1]; }
        full_pel_search(d, jb, &p, mvx, mvy, &best, &bx, &by);
        d->best_cost[j] = best; d->best_mv[2 * j] = bx; d->best_mv[2 * j + 1] = by;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------------------- */
/* svt_av1_find_best_sub_pixel_tree_pruned (mcomp.c:606-687)                                                                           */
/* ---------------------------------------------------------------------------------------------------------------------------------- */
typedef struct {
    const SvtHipSubpelBatchDesc *d;
    const SvtHipSubpelJob       *jb;
    SvtHipMvCostParam            mp;
    const uint8_t               *src, *ref;
} Sp;

static int in_range(const Sp *s, SvtHipMv mv) { return mv.col >= s->jb->col_min && mv.col <= s->jb->col_max && mv.row >= s->jb->row_min && mv.row <= s->jb->row_max; }

/* svt_check_better_fast (mcomp.c:170-206) with is_scaled == 0 */
static unsigned check_better_fast(const Sp *s, SvtHipMv this_mv, SvtHipMv *best_mv, unsigned *besterr, unsigned *sse1, int *distortion) {
    unsigned cost;
    if (in_range(s, this_mv)) {
        unsigned sse;
        cost = (unsigned)orc_mv_err_cost(this_mv.row, this_mv.col, &s->mp);
        if (s->mp.mv_cost_type == SVT_HIP_MV_COST_OPT) {
            const int64_t bestcost = (int64_t)*distortion + cost;
            if (bestcost > (((int64_t)*besterr * (int64_t)s->jb->early_exit_th) / 1000)) return (uint32_t)bestcost;
        }
        const uint8_t *ref = s->ref + (this_mv.row >> 3) * (int)s->d->ref_stride + (this_mv.col >> 3); /* svt_get_buf_from_mv: floor */
        const int thismse = (int)sub_pixel_variance_wxh(ref, (int)s->d->ref_stride, this_mv.col & 7, this_mv.row & 7, s->src, (int)s->d->src_stride, s->jb->width, s->jb->height, &sse);
        cost += (unsigned)thismse;
        int weight = 100;
        if (s->d->bias_fp && best_mv->col % 8 == 0 && best_mv->row % 8 == 0) weight = s->d->bias_fp;
        if ((((uint64_t)cost * (uint64_t)(int64_t)weight) / 100) < *besterr) {
            *besterr = cost; *best_mv = this_mv; *distortion = thismse; *sse1 = sse;
        }
    } else
        cost = INT_MAX;
    return cost;
}

static void two_level_checks_fast(const Sp *s, SvtHipMv this_mv, SvtHipMv *best_mv, int hstep, unsigned *besterr, unsigned orgerr, unsigned *sse1, int *distortion, int iters) {
    /* first_level_check_fast (mcomp.c:371-417) */
    const SvtHipMv left_mv = {this_mv.row, (int16_t)(this_mv.col - hstep)}, right_mv = {this_mv.row, (int16_t)(this_mv.col + hstep)};
    const SvtHipMv top_mv = {(int16_t)(this_mv.row - hstep), this_mv.col}, bottom_mv = {(int16_t)(this_mv.row + hstep), this_mv.col};
    const unsigned left  = check_better_fast(s, left_mv, best_mv, besterr, sse1, distortion);
    const unsigned right = check_better_fast(s, right_mv, best_mv, besterr, sse1, distortion);
    const unsigned up    = check_better_fast(s, top_mv, best_mv, besterr, sse1, distortion);
    const unsigned down  = check_better_fast(s, bottom_mv, best_mv, besterr, sse1, distortion);
    const SvtHipMv diag_step = {(int16_t)(up <= down ? -hstep : hstep), (int16_t)(left <= right ? -hstep : hstep)};
    if (!(*besterr >= orgerr)) {
        const SvtHipMv diag_mv = {(int16_t)(this_mv.row + diag_step.row), (int16_t)(this_mv.col + diag_step.col)};
        check_better_fast(s, diag_mv, best_mv, besterr, sse1, distortion);
    }
    if (!(*besterr < orgerr) || iters <= 1) return;
    /* second_level_check_fast (mcomp.c:421-539) */
    const int tr = this_mv.row, tc = this_mv.col, br = best_mv->row, bc = best_mv->col;
    if (tr != br && tc != bc) {
        const SvtHipMv chess_mv_1 = {(int16_t)br, (int16_t)(bc + diag_step.col)}, chess_mv_2 = {(int16_t)(br + diag_step.row), (int16_t)bc};
        check_better_fast(s, chess_mv_1, best_mv, besterr, sse1, distortion);
        check_better_fast(s, chess_mv_2, best_mv, besterr, sse1, distortion);
    } else if (tr == br && tc != bc) {
        const SvtHipMv bottom_long_mv = {(int16_t)(br + hstep), (int16_t)(bc + diag_step.col)}, top_long_mv = {(int16_t)(br - hstep), (int16_t)(bc + diag_step.col)};
        check_better_fast(s, bottom_long_mv, best_mv, besterr, sse1, distortion);
        check_better_fast(s, top_long_mv, best_mv, besterr, sse1, distortion);
        const SvtHipMv rev_mv = {(int16_t)(br - diag_step.row), (int16_t)bc};
        check_better_fast(s, rev_mv, best_mv, besterr, sse1, distortion);
    } else if (tr != br && tc == bc) {
        const SvtHipMv right_long_mv = {(int16_t)(br + diag_step.row), (int16_t)(bc + hstep)}, left_long_mv = {(int16_t)(br + diag_step.row), (int16_t)(bc - hstep)};
        check_better_fast(s, right_long_mv, best_mv, besterr, sse1, distortion);
        check_better_fast(s, left_long_mv, best_mv, besterr, sse1, distortion);
        const SvtHipMv rev_mv = {(int16_t)br, (int16_t)(bc - diag_step.col)};
        check_better_fast(s, rev_mv, best_mv, besterr, sse1, distortion);
    }
}

static unsigned sub_pixel_tree_pruned(const Sp *s, SvtHipMv start_mv, SvtHipMv *bestmv, int *distortion, unsigned *sse1, unsigned *center_err) {
    const SvtHipSubpelBatchDesc *d = s->d;
    int      hstep = 4; /* INIT_SUBPEL_STEP_SIZE */
    unsigned besterr, org_error;
    *bestmv = start_mv;
    { /* svt_upsampled_setup_center_error (mcomp.c:353-360) */
        const uint8_t *ref = s->ref + (bestmv->row >> 3) * (int)d->ref_stride + (bestmv->col >> 3);
        uint32_t       sse;
        *distortion = (int)variance_wxh(ref, (int)d->ref_stride, s->src, (int)d->src_stride, s->jb->width, s->jb->height, &sse);
        besterr     = (unsigned)*distortion + (unsigned)orc_mv_err_cost(bestmv->row, bestmv->col, &s->mp);
    }
    *center_err = besterr;
    if (s->jb->early_neigh_check_exit) return besterr;
    const uint64_t th_normalizer = (uint64_t)(int64_t)(((s->jb->width * s->jb->height) >> 3) * (int)(uint8_t)d->abs_th_mult * (d->qp >> 1));
    if (besterr < th_normalizer) return besterr;
    const int round = (3 /* FULL_PEL */ - d->forced_stop) < (3 - !d->allow_hp) ? (3 - d->forced_stop) : (3 - !d->allow_hp);
    if (!round) return besterr;
    { /* variance of the full-pel prediction against the constant 128 (svt_aom_eb_av1_var_offs, stride 0) */
        const uint8_t *ref = s->ref + (bestmv->row >> 3) * (int)d->ref_stride + (bestmv->col >> 3);
        static const uint8_t offs[128] = {128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
                                          128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
                                          128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
                                          128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
                                          128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};
        uint32_t sse;
        const unsigned var = variance_wxh(ref, (int)d->ref_stride, offs, 0, s->jb->width, s->jb->height, &sse);
        const int block_var = (int)((var + ((1u << s->jb->log2_pels) >> 1)) >> s->jb->log2_pels); /* ROUND_POWER_OF_TWO */
        if (block_var < d->pred_variance_th) return besterr;
    }
    if ((uint8_t)d->skip_diag_refinement >= 4)
        org_error = 0;
    else {
        const unsigned demo = (uint8_t)d->skip_diag_refinement >= 2 ? ((s->jb->width >= 64 || s->jb->height >= 64) ? 2 : 1) : 1;
        org_error           = (uint8_t)d->skip_diag_refinement ? besterr / demo : INT_MAX;
    }
    for (int iter = 0; iter < round; ++iter) {
        const unsigned prev_besterr = besterr;
        two_level_checks_fast(s, start_mv, bestmv, hstep, &besterr, org_error, sse1, distortion, d->iters_per_step);
        hstep >>= 1;
        start_mv = *bestmv;
        if ((uint8_t)d->skip_diag_refinement && iter < 1 /* QUARTER_PEL */) org_error = org_error < besterr ? org_error : besterr;
        const int64_t a = besterr > 1 ? besterr : 1, b = prev_besterr > 1 ? prev_besterr : 1;
        const int32_t deviation = (int32_t)(((a - b) * 100) / b);
        if (deviation >= d->round_dev_th) return besterr;
    }
    return besterr;
}

/* ---------------------------------------------------------------------------------------------------------------------------------- */
/* svt_av1_find_best_sub_pixel_tree (mcomp.c:688-777): the accurate search                                                             */
/* ---------------------------------------------------------------------------------------------------------------------------------- */
/* rows 0, 2, .. 14 of av1_bilinear_filters / av1_sub_pel_filters_4 / av1_sub_pel_filters_8 (C_DEFAULT/variance.c:72-135): the kernels
 * av1_get_interp_filter_subpel_kernel(filter, subpel_q3 << 1) selects for USE_2_TAPS / USE_4_TAPS / USE_8_TAPS (av1_get_filter, :191-201) */
static const int16_t k_subpel_filters[3][8][8] = {
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, 0, 112, 16, 0, 0, 0}, {0, 0, 0, 96, 32, 0, 0, 0}, {0, 0, 0, 80, 48, 0, 0, 0}, {0, 0, 0, 64, 64, 0, 0, 0}, {0, 0, 0, 48, 80, 0, 0, 0},
     {0, 0, 0, 32, 96, 0, 0, 0}, {0, 0, 0, 16, 112, 0, 0, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, -8, 122, 18, -4, 0, 0}, {0, 0, -12, 110, 38, -8, 0, 0}, {0, 0, -14, 94, 58, -10, 0, 0}, {0, 0, -12, 76, 76, -12, 0, 0},
     {0, 0, -10, 58, 94, -14, 0, 0}, {0, 0, -8, 38, 110, -12, 0, 0}, {0, 0, -4, 18, 122, -8, 0, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 2, -10, 122, 18, -4, 0, 0}, {0, 2, -14, 110, 38, -10, 2, 0}, {0, 2, -16, 94, 58, -12, 2, 0}, {0, 2, -14, 76, 76, -14, 2, 0},
     {0, 2, -12, 58, 94, -16, 2, 0}, {0, 2, -10, 38, 110, -14, 2, 0}, {0, 0, -4, 18, 122, -10, 2, 0}}};
static uint8_t clip_pixel8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* svt_aom_upsampled_pred_c (C_DEFAULT/variance.c:204-254) = svt_aom_convolve8_horiz_c / _vert_c (Codec/convolve.c:244-301) at a fixed phase:
 * 8 taps around the sample (offsets -3 .. +4), (sum + 64) >> 7, clipped to 8 bits after EACH pass (the intermediate is an 8-bit array) */
static void upsampled_pred(uint8_t *pred, int w, int h, int sx, int sy, const uint8_t *ref, int ref_stride, int search_type) {
    const int16_t *fx = k_subpel_filters[search_type - 1][sx], *fy = k_subpel_filters[search_type - 1][sy];
    if (!sx && !sy) {
        for (int y = 0; y < h; y++) memcpy(pred + y * w, ref + y * ref_stride, (size_t)w);
    } else if (!sy) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int sum = 0;
                for (int k = 0; k < 8; k++) sum += ref[y * ref_stride + x - 3 + k] * fx[k];
                pred[y * w + x] = clip_pixel8((sum + 64) >> 7);
            }
    } else if (!sx) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int sum = 0;
                for (int k = 0; k < 8; k++) sum += ref[(y - 3 + k) * ref_stride + x] * fy[k];
                pred[y * w + x] = clip_pixel8((sum + 64) >> 7);
            }
    } else {
        static uint8_t temp[(128 + 7) * 128]; /* rows -3 .. h + 3 of the horizontally filtered block (single-threaded test code) */
        for (int y = 0; y < h + 7; y++)
            for (int x = 0; x < w; x++) {
                int sum = 0;
                for (int k = 0; k < 8; k++) sum += ref[(y - 3) * ref_stride + x - 3 + k] * fx[k];
                temp[y * 128 + x] = clip_pixel8((sum + 64) >> 7);
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int sum = 0;
                for (int k = 0; k < 8; k++) sum += temp[(y + k) * 128 + x] * fy[k];
                pred[y * w + x] = clip_pixel8((sum + 64) >> 7);
            }
    }
}

/* svt_upsampled_pref_error (mcomp.c:106-143) */
static int upsampled_pref_error(const Sp *s, SvtHipMv this_mv, unsigned *sse) {
    static uint8_t pred[128 * 128];
    const uint8_t *ref = s->ref + (this_mv.row >> 3) * (int)s->d->ref_stride + (this_mv.col >> 3);
    upsampled_pred(pred, s->jb->width, s->jb->height, this_mv.col & 7, this_mv.row & 7, ref, (int)s->d->ref_stride, s->d->subpel_search_type);
    return (int)variance_wxh(pred, s->jb->width, s->src, (int)s->d->src_stride, s->jb->width, s->jb->height, sse);
}

/* svt_check_better (mcomp.c:210-238) */
static unsigned check_better(const Sp *s, SvtHipMv this_mv, SvtHipMv *best_mv, unsigned *besterr, unsigned *sse1, int *distortion, int *is_better) {
    unsigned cost;
    if (in_range(s, this_mv)) {
        unsigned  sse;
        const int thismse = upsampled_pref_error(s, this_mv, &sse);
        cost = (unsigned)orc_mv_err_cost(this_mv.row, this_mv.col, &s->mp) + (unsigned)thismse;
        int weight = 100;
        if (s->d->bias_fp && best_mv->col % 8 == 0 && best_mv->row % 8 == 0) weight = s->d->bias_fp;
        if ((((uint64_t)cost * (uint64_t)(int64_t)weight) / 100) < *besterr) {
            *besterr = cost; *best_mv = this_mv; *distortion = thismse; *sse1 = sse; *is_better |= 1;
        }
    } else
        cost = INT_MAX;
    return cost;
}

static unsigned sub_pixel_tree(const Sp *s, SvtHipMv start_mv, SvtHipMv *bestmv, int *distortion, unsigned *sse1, unsigned *center_err) {
    const SvtHipSubpelBatchDesc *d = s->d;
    int      round = (3 /* FULL_PEL */ - d->forced_stop) < (3 - !d->allow_hp) ? (3 - d->forced_stop) : (3 - !d->allow_hp);
    int      hstep = 4;
    unsigned besterr;
    *bestmv = start_mv;
    { /* svt_upsampled_setup_center_error (mcomp.c:353-360) */
        const uint8_t *ref = s->ref + (bestmv->row >> 3) * (int)d->ref_stride + (bestmv->col >> 3);
        uint32_t       sse;
        *distortion = (int)variance_wxh(ref, (int)d->ref_stride, s->src, (int)d->src_stride, s->jb->width, s->jb->height, &sse);
        besterr     = (unsigned)*distortion + (unsigned)orc_mv_err_cost(bestmv->row, bestmv->col, &s->mp);
    }
    *center_err = besterr;
    if (d->mvp_th > 0) { /* ctx != NULL, search_stage == SPEL_ME, pd_pass == PD_PASS_1 (:702-722) */
        const int     mvp_err = (int)s->jb->best_mvp_dist + 1, me_err = (int)besterr + 1;
        const int32_t deviation = ((me_err - mvp_err) * 100) / me_err;
        if (deviation >= d->mvp_th)
            round = 1;
        else if (abs(bestmv->col - s->jb->best_mvp.col) > d->hp_mv_th || abs(bestmv->row - s->jb->best_mvp.row) > d->hp_mv_th)
            round = round < 2 ? round : 2;
    }
    if (s->jb->early_neigh_check_exit) return besterr;
    { /* variance of the full-pel prediction against the constant 128 (svt_aom_eb_av1_var_offs, stride 0) */
        const uint8_t *ref = s->ref + (bestmv->row >> 3) * (int)d->ref_stride + (bestmv->col >> 3);
        uint8_t        offs[128];
        uint32_t       sse;
        memset(offs, 128, sizeof(offs));
        const unsigned var = variance_wxh(ref, (int)d->ref_stride, offs, 0, s->jb->width, s->jb->height, &sse);
        const int block_var = (int)((var + ((1u << s->jb->log2_pels) >> 1)) >> s->jb->log2_pels);
        if (block_var < d->pred_variance_th) return besterr;
    }
    const uint64_t th_normalizer = (uint64_t)(int64_t)(((s->jb->width * s->jb->height) >> 2) * (int)(uint8_t)d->abs_th_mult * (d->qp >> 1));
    if (besterr < th_normalizer) return besterr;
    if (!round) return besterr;
    for (int iter = 0; iter < round; ++iter) {
        const SvtHipMv c = *bestmv; /* iter_center_mv */
        int dummy = 0;
        /* svt_first_level_check (:260-287) */
        const SvtHipMv left_mv = {c.row, (int16_t)(c.col - hstep)}, right_mv = {c.row, (int16_t)(c.col + hstep)}, top_mv = {(int16_t)(c.row - hstep), c.col},
                       bottom_mv = {(int16_t)(c.row + hstep), c.col};
        const unsigned left  = check_better(s, left_mv, bestmv, &besterr, sse1, distortion, &dummy);
        const unsigned right = check_better(s, right_mv, bestmv, &besterr, sse1, distortion, &dummy);
        const unsigned up    = check_better(s, top_mv, bestmv, &besterr, sse1, distortion, &dummy);
        const unsigned down  = check_better(s, bottom_mv, bestmv, &besterr, sse1, distortion, &dummy);
        SvtHipMv diag_step = {(int16_t)(up <= down ? -hstep : hstep), (int16_t)(left <= right ? -hstep : hstep)};
        const SvtHipMv diag_mv = {(int16_t)(c.row + diag_step.row), (int16_t)(c.col + diag_step.col)};
        check_better(s, diag_mv, bestmv, &besterr, sse1, distortion, &dummy);
        if (!(c.row == bestmv->row && c.col == bestmv->col) && d->iters_per_step > 1) { /* svt_second_level_check_v2 (:292-350) */
            if (c.row == bestmv->row) diag_step.row = (int16_t)-diag_step.row;
            else if (c.col == bestmv->col) diag_step.col = (int16_t)-diag_step.col;
            const SvtHipMv row_bias_mv = {(int16_t)(bestmv->row + diag_step.row), bestmv->col}, col_bias_mv = {bestmv->row, (int16_t)(bestmv->col + diag_step.col)},
                           diag_bias_mv = {(int16_t)(bestmv->row + diag_step.row), (int16_t)(bestmv->col + diag_step.col)};
            int has_better_mv = 0;
            check_better(s, row_bias_mv, bestmv, &besterr, sse1, distortion, &has_better_mv);
            check_better(s, col_bias_mv, bestmv, &besterr, sse1, distortion, &has_better_mv);
            if (has_better_mv) check_better(s, diag_bias_mv, bestmv, &besterr, sse1, distortion, &has_better_mv);
        }
        hstep >>= 1;
    }
    return besterr;
}

int orc_md_subpel_batch(const SvtHipSubpelBatchDesc *d) {
    if (d->search_method < 0 || d->search_method > 1) return 1;
    if (d->search_method == 1 && (d->subpel_search_type < SVT_HIP_USE_2_TAPS || d->subpel_search_type > SVT_HIP_USE_8_TAPS)) return 1;
    for (uint32_t j = 0; j < d->n_jobs; j++) {
        Sp s;
        memset(&s, 0, sizeof(s));
        s.d = d; s.jb = &d->jobs[j];
        s.mp.ref_mv = &s.jb->ref_mv; s.mp.mv_cost_type = (uint8_t)d->mv_cost_type; s.mp.mvjcost = d->mvjcost; s.mp.mvcost[0] = d->mvcost[0]; s.mp.mvcost[1] = d->mvcost[1];
        s.mp.error_per_bit = d->error_per_bit; s.mp.early_exit_th = s.jb->early_exit_th;
        s.src = d->src + s.jb->src_offset; s.ref = d->ref + s.jb->ref_offset;
        SvtHipMv best;
        int      dist = 0;
        unsigned sse1 = 0;
        unsigned center = 0;
        if (d->search_method == 1)
            d->besterr[j] = sub_pixel_tree(&s, s.jb->start_mv, &best, &dist, &sse1, &center);
        else
            d->besterr[j] = sub_pixel_tree_pruned(&s, s.jb->start_mv, &best, &dist, &sse1, &center);
        if (d->center_err) d->center_err[j] = center;
        d->best_mv[2 * j]     = best.row;
        d->best_mv[2 * j + 1] = best.col;
        d->distortion[j]      = dist;
        d->sse[j]             = sse1;
    }
    return 0;
}
size_t orc_sizeof_md_search(int what) {
    return what == 0 ? sizeof(SvtHipFullpelJob) : what == 1 ? sizeof(SvtHipFullpelBatchDesc) : what == 2 ? sizeof(SvtHipSubpelJob) : sizeof(SvtHipSubpelBatchDesc);
}
