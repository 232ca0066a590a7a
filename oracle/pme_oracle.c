/*
 * pme_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the mode-decision full-pel refinement search: svt_pme_sad_loop_kernel_c
 * (Codec/product_coding_loop.c:1905-1950) with svt_aom_fp_mv_err_cost = svt_mv_err_cost (Codec/mcomp.c:44-78,776), svt_mv_cost (mcomp.h:135-138)
 * and svt_av1_get_mv_joint (rd_cost.c:55-60).  Pinned on the reference's own function in tests/test_pme.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include "../include/svt_hip_pme.h"

#define MV_LOW (-(1 << 14)) /* cabac_context_model.h:523-525 */
#define MV_UPP (1 << 14)

/* RDDIV_BITS 7 + AV1_PROB_COST_SHIFT 9 - RD_EPB_SHIFT 6 + PIXEL_TRANSFORM_ERROR_SCALE 4 (rd_cost.h:35, definitions.h:323, restoration.h:344, mcomp.c:43) */
#define MV_ERR_SHIFT 14

static int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

int orc_mv_err_cost(int16_t mv_row, int16_t mv_col, const SvtHipMvCostParam *p) {
    const int16_t dr = (int16_t)(mv_row - p->ref_mv->row), dc = (int16_t)(mv_col - p->ref_mv->col); /* MV diff: int16 fields */
    const int16_t ar = (int16_t)abs(dr), ac = (int16_t)abs(dc);
    switch (p->mv_cost_type) {
    case SVT_HIP_MV_COST_ENTROPY: {
        if (!p->mvcost[0]) return 0; /* `if (mvcost)` tests the array, always true in the reference; a null table is the caller's error */
        const int joint = dr == 0 ? (dc == 0 ? 0 : 1) : (dc == 0 ? 2 : 3);
        const int bits  = p->mvjcost[joint] + p->mvcost[0][clip3(MV_LOW, MV_UPP, dr)] + p->mvcost[1][clip3(MV_LOW, MV_UPP, dc)];
        return (int)((((int64_t)bits * p->error_per_bit) + ((int64_t)1 << (MV_ERR_SHIFT - 1))) >> MV_ERR_SHIFT);
    }
    case SVT_HIP_MV_COST_L1_LOWRES: return (2 * (ar + ac)) >> 3;
    case SVT_HIP_MV_COST_L1_MIDRES: return (0 * (ar + ac)) >> 3;
    case SVT_HIP_MV_COST_L1_HDRES: return (1 * (ar + ac)) >> 3;
    case SVT_HIP_MV_COST_OPT:
        return (int)((((int64_t)((ar + ac) << 8) * p->error_per_bit) + ((int64_t)1 << (MV_ERR_SHIFT - 1))) >> MV_ERR_SHIFT);
    default: return 0;
    }
}

void orc_pme_sad_loop_kernel(const SvtHipMvCostParam *p, const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                             uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx, int16_t *best_mvy, int16_t start_x, int16_t start_y, int16_t sa_w,
                             int16_t sa_h, int16_t step, int16_t mvx, int16_t mvy) {
    int16_t col_num = 0, step_x = 1;
    for (int16_t ys = 0; ys < sa_h; ys += step) {
        for (int16_t xs = 0; xs < sa_w; xs += step_x) {
            if ((sa_w - xs) < 8 && col_num == 0) continue; /* no room for another group of 8 */
            if (col_num == 7) { col_num = 0; step_x = step; } else { col_num++; step_x = 1; }
            uint32_t cost = 0;
            for (uint32_t y = 0; y < block_height; y++)
                for (uint32_t x = 0; x < block_width; x++) cost += (uint32_t)abs((int)src[y * src_stride + x] - (int)ref[xs + y * ref_stride + x]);
            const uint32_t px = (uint32_t)(start_x + xs), py = (uint32_t)(start_y + ys);
            const int16_t  col = (int16_t)(mvx + (px * 8)), row = (int16_t)(mvy + (py * 8));
            cost += (uint32_t)orc_mv_err_cost(row, col, p);
            if (cost < *best_cost) { *best_mvx = col; *best_mvy = row; *best_cost = cost; }
        }
        ref += step * ref_stride;
    }
}

/* host-memory mirror of svt_hip_pme_sad_batch */
int orc_pme_sad_batch(const SvtHipPmeBatchDesc *d) {
    for (uint32_t j = 0; j < d->n_jobs; j++) {
        const SvtHipPmeJob *jb = &d->jobs[j];
        SvtHipMvCostParam   p  = {0};
        p.ref_mv = &jb->ref_mv; p.mv_cost_type = d->mv_cost_type; p.mvjcost = d->mvjcost; p.mvcost[0] = d->mvcost[0]; p.mvcost[1] = d->mvcost[1];
        p.error_per_bit = d->error_per_bit;
        uint32_t best = jb->best_cost;
        int16_t  bx = jb->best_mvx, by = jb->best_mvy;
        orc_pme_sad_loop_kernel(&p, d->src + jb->src_offset, d->src_stride, d->ref + jb->ref_offset, d->ref_stride, jb->height, jb->width, &best, &bx, &by, jb->start_x,
                                jb->start_y, jb->sa_w, jb->sa_h, jb->step, jb->mvx, jb->mvy);
        d->best_cost[j] = best; d->best_mv[2 * j] = bx; d->best_mv[2 * j + 1] = by;
    }
    return 0;
}
size_t orc_sizeof_pme(int what) { return what == 0 ? sizeof(SvtHipPmeJob) : what == 1 ? sizeof(SvtHipPmeBatchDesc) : sizeof(SvtHipMvCostParam); }
#include <stddef.h>
void orc_mv_cost_param_layout(size_t out[9]) {
    out[0] = sizeof(SvtHipMvCostParam);
    out[1] = offsetof(SvtHipMvCostParam, ref_mv); out[2] = offsetof(SvtHipMvCostParam, full_ref_mv); out[3] = offsetof(SvtHipMvCostParam, mv_cost_type);
    out[4] = offsetof(SvtHipMvCostParam, mvjcost); out[5] = offsetof(SvtHipMvCostParam, mvcost); out[6] = offsetof(SvtHipMvCostParam, error_per_bit);
    out[7] = offsetof(SvtHipMvCostParam, early_exit_th); out[8] = offsetof(SvtHipMvCostParam, sad_per_bit);
    { SvtHipMvCostParam p; out[3] |= (size_t)sizeof(p.mv_cost_type) << 16; }
}
