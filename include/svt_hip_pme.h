/*
 * svt_hip_pme.h -- C-ABI of the mode-decision full-pel refinement search: svt_pme_sad_loop_kernel, the SAD + MV-rate
 * search md_full_pel_search runs around every ME / predictive-ME candidate (Codec/product_coding_loop.c:1905-1950, driver :1952-2040,
 * callers :2292-2607).  Per search position: SAD of the block against the reference window + svt_aom_fp_mv_err_cost of that position's
 * motion vector (Codec/mcomp.c:44-78,776); the first strict minimum in visiting order replaces the incoming best.
 *
 * Two levels, as everywhere in this backend: a batched entry over (block, candidate centre) jobs on device-resident planes, and the
 * pointer-level entry with the reference's prototype (aom_dsp_rtcd.h:865-868) for installation into the rtcd pointer.
 */
#ifndef SVT_HIP_PME_H
#define SVT_HIP_PME_H

#include <stdint.h>
#include "svt_hip_me.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SvtHipMv { int16_t row, col; } SvtHipMv; /* MV / FULLPEL_MV, Codec/mv.h */

/* MV_COST_TYPE (Codec/mcomp.h:29-36) */
#define SVT_HIP_MV_COST_ENTROPY 0
#define SVT_HIP_MV_COST_L1_LOWRES 1
#define SVT_HIP_MV_COST_L1_MIDRES 2
#define SVT_HIP_MV_COST_L1_HDRES 3
#define SVT_HIP_MV_COST_OPT 4
#define SVT_HIP_MV_COST_NONE 5

/* MV_COST_PARAMS (`struct svt_mv_cost_param`, Codec/mcomp.h:37-48), same layout: the pointer-level entry takes the reference's struct as it is */
typedef struct SvtHipMvCostParam {
    const SvtHipMv *ref_mv;
    SvtHipMv        full_ref_mv;
    uint8_t         mv_cost_type; /* MV_COST_TYPE is UENUM1BYTE (Codec/mcomp.h:29-36, definitions.h:268): ONE byte; the three bytes after it are
                                   * padding that md_full_pel_search leaves uninitialised on its stack (product_coding_loop.c:2030-2049) */
    const int      *mvjcost;   /* [MV_JOINTS = 4] */
    const int      *mvcost[2]; /* row / column component costs, pointers to the CENTRE (index 0) of MV_VALS-entry tables */
    int             error_per_bit;
    int             early_exit_th;
    int             sad_per_bit;
} SvtHipMvCostParam;

/* One search: the arguments of one svt_pme_sad_loop_kernel call */
typedef struct SvtHipPmeJob {
    uint32_t src_offset;         /* the block's top-left sample in the source plane */
    uint32_t ref_offset;         /* the reference sample of search index (0, 0), block row 0: `ref` of the call */
    uint8_t  width, height;      /* block_width (a multiple of 4), block_height: 4 .. 128 */
    int16_t  start_x, start_y;   /* search_position_start_x / _y */
    int16_t  sa_w, sa_h;         /* search_area_width (the reference passes a multiple of 8), search_area_height */
    int16_t  step;               /* search_step >= 1 (sparse search: rows `step` apart, groups of 8 columns 7 + step apart) */
    int16_t  mvx, mvy;           /* the candidate's MV in 1/8 sample */
    SvtHipMv ref_mv;             /* *mv_cost_params->ref_mv */
    uint32_t best_cost;          /* *best_cost on entry */
    int16_t  best_mvx, best_mvy; /* *best_mvx / *best_mvy on entry (kept when no position beats best_cost) */
} SvtHipPmeJob;

typedef struct SvtHipPmeBatchDesc {
    uint32_t n_jobs;
    uint32_t src_stride, ref_stride; /* in samples */
    const uint8_t      *src, *ref;   /* device pointers, 8-bit planes */
    const SvtHipPmeJob *jobs;        /* device pointer */
    int32_t  mv_cost_type, error_per_bit; /* mv_cost_params->mv_cost_type / error_per_bit */
    const int32_t *mvjcost;          /* device pointer, 4 entries */
    const int32_t *mvcost[2];        /* device pointers to the CENTRE entries of the row / column tables; entries [-16384, 16384] (MV_LOW .. MV_UPP,
                                      * the reference's clamp) may be read.  Read for SVT_HIP_MV_COST_ENTROPY only. */
    uint32_t *best_cost;             /* out, device pointers: [n_jobs] */
    int16_t  *best_mv;               /* out: [n_jobs][2] = (best_mvx, best_mvy) */
} SvtHipPmeBatchDesc;

/* Enqueues the batch on the context stream (asynchronous); one wave per job. */
int svt_hip_pme_sad_batch(SvtHipContext *ctx, const SvtHipPmeBatchDesc *d);

/* svt_pme_sad_loop_kernel (aom_dsp_rtcd.h:868): host pointers, synchronous (see svt_hip_leaf.h for the contract of the pointer-level entries) */
void svt_pme_sad_loop_kernel_hip(const SvtHipMvCostParam *mv_cost_params, uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                                 uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx, int16_t *best_mvy, int16_t search_position_start_x,
                                 int16_t search_position_start_y, int16_t search_area_width, int16_t search_area_height, int16_t search_step, int16_t mvx, int16_t mvy);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_PME_H */
