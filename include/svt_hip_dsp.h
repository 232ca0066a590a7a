/*
 * svt_hip_dsp.h -- C-ABI of the MI355X mode-decision RD kernels (residual, forward / inverse integer AV1
 * transforms, quantize + dequantize, coefficient-domain and pixel-domain distortion, SATD).
 *
 * The batched entry svt_hip_rd_batch evaluates, for every job, one iteration of the reference's tx_type_search
 * loop body (Source/Lib/Codec/product_coding_loop.c:4764-4934):
 *     svt_residual_kernel8bit/16bit  (Codec/pic_operators.c:101-148)
 *  -> svt_aom_estimate_transform / svt_av1_fwd_txfm2d_{WxH} (+ svt_handle_transform{64..}) (Codec/transforms.c:2259-2631,3158)
 *  -> svt_aom_satd                                          (Codec/common_dsp_rtcd.c:70-77)
 *  -> svt_aom_quantize_b / svt_aom_highbd_quantize_b / svt_av1_quantize_fp / svt_av1_highbd_quantize_fp
 *                                                           (Codec/full_loop.c:29-198,282-474)
 *  -> svt_full_distortion_kernel32_bits                     (Codec/pic_operators.c:150-172)
 *  -> svt_av1_inv_txfm2d_add_{WxH}                          (Codec/inv_transforms.c:2459-2716)
 *  -> svt_spatial_full_distortion_kernel / svt_full_distortion_kernel16_bits (picture_operators_c.c:65-83, pic_operators.c:174-197)
 * Decision logic (which tx_type wins, rate estimation, RDOQ) stays on the host.
 *
 * Also here: the forward and the inverse transform as batches of their own (svt_hip_fwd_txfm_batch, svt_hip_inv_txfm_batch), the
 * batched block statistics incl. the PSYEX psy-RD term and distortion facades (svt_hip_block_stats_batch, svt_hip_spy_rd_bias),
 * full-pel prediction from ME results (svt_hip_fullpel_pred{,_batch}) and the scan-order / size helpers.
 */
#ifndef SVT_HIP_DSP_H
#define SVT_HIP_DSP_H

#include <stdint.h>
#include "svt_hip_me.h"

#ifdef __cplusplus
extern "C" {
#endif

/* TxSize / TxType use the reference's enum values (Codec/definitions.h): TX_4X4=0 ... TX_64X16=18; DCT_DCT=0 ... H_FLIPADST=15 */
#define SVT_HIP_TX_SIZES_ALL 19
#define SVT_HIP_TX_TYPES 16

/* Per-qindex quantizer rows as MacroblockPlane / Dequants hold them (Codec/full_loop.c:1627-1685): [0] = DC, [1] = AC */
typedef struct SvtHipQuantRow {
    int16_t zbin[2], round[2], quant[2], quant_shift[2]; /* "b" quantizer */
    int16_t round_fp[2], quant_fp[2];                    /* "fp" quantizer */
    int16_t dequant[2];
} SvtHipQuantRow;

typedef struct SvtHipTxJob {
    uint32_t src_offset;  /* sample offset of the block's top-left sample in the source plane   */
    uint32_t pred_offset; /* sample offset in the prediction plane (and in the recon plane)       */
    uint8_t  tx_type;
    uint8_t  quant_row;   /* index into SvtHipRdBatchDesc.quant_rows */
    uint8_t  pf_shape;    /* EB_TRANS_COEFF_SHAPE (Codec/definitions.h): 0 DEFAULT, 1 N2, 2 N4, 3 ONLY_DC -- the partial-frequency
                           * forward transforms av1_estimate_transform_{N2,N4,ONLY_DC} (Codec/transforms.c:2633-2948) */
    uint8_t  reserved;
} SvtHipTxJob;

typedef struct SvtHipRdBatchDesc {
    uint8_t  bit_depth;     /* 8: planes are uint8; 10: planes are uint16 */
    uint8_t  quant_kind;    /* 0 = "b" (zbin / quant_shift), 1 = "fp", 2 = "fp" with the log-scale of the size forced to 0: the TPL
                             * dispenser calls plain svt_av1_quantize_fp for every transform size (get_quantize_error, Codec/src_ops_process.c:225-249) */
    uint8_t  tx_size;       /* TxSize shared by every job of this call (one kernel instantiation per size) */
    uint8_t  reserved;
    uint32_t n_jobs;
    uint32_t src_stride, pred_stride; /* in samples */
    const void           *src, *pred; /* device pointers */
    void                 *recon;      /* device pointer or NULL; same geometry as pred */
    const SvtHipTxJob    *jobs;       /* device pointer, n_jobs entries */
    const SvtHipQuantRow *quant_rows; /* device pointer */
    uint32_t              n_quant_rows;
    /* per-job outputs (device pointers; eob .. sse mandatory) */
    uint16_t *eob;                    /* [n_jobs] */
    uint32_t *satd;                   /* [n_jobs] svt_aom_satd of the kept coefficients */
    uint64_t *dist_coeff;             /* [n_jobs][2] {sum (coeff - dqcoeff)^2, sum coeff^2} */
    uint64_t *three_quad_energy;      /* [n_jobs] energy of the frequencies a 64-point size discards (0 otherwise) */
    uint64_t *sse;                    /* [n_jobs] sum (src - recon)^2 */
    int32_t  *coeff, *qcoeff, *dqcoeff; /* optional: [n_jobs][min(W,32)*min(H,32)] packed like the reference */
    /* optional quantization matrices of this tx_size (device pointers, min(W,32)*min(H,32) bytes each, AOM_QM_BITS = 5 fixed
     * point: pcs->ppcs->gqmatrix / giqmatrix[level][plane][adjusted_tx_size], full_loop.c:1606-1613); applied to the jobs with
     * a 2-D tx_type (tx_type < IDTX), like the reference; NULL = flat */
    const uint8_t *qmatrix, *iqmatrix;
    /* optional, [n_jobs] (device pointer): svt_av1_compute_cul_level(scan, qcoeff, &eob) (Codec/full_loop.c:1449-1466) -- the value
     * svt_aom_quantize_inv_quantize returns when rate_est_ctrls.update_skip_ctx_dc_sign_ctx is set (:1832-1836): min(63, sum |qcoeff|)
     * with the DC sign in bits 6-7 (negative: | 64, positive: + 128); feeds the entropy contexts of the next block's rate estimate */
    uint8_t *cul_level;
} SvtHipRdBatchDesc;

/* Enqueues one batch on the context stream (asynchronous).  Every pointer in `d` is a DEVICE pointer.
 * Returns non-zero (and leaves nothing enqueued) when the descriptor fails validation. */
int svt_hip_rd_batch(SvtHipContext *ctx, const SvtHipRdBatchDesc *d);

/* ---- forward transform alone ------------------------------------------------------------------------------------
 * svt_av1_fwd_txfm2d_{W}x{H}{,_N2,_N4} (Codec/transforms.c:2259-2631,5202-5425,6769-6990) on an int16 residual plane: the FULL W x H
 * coefficient array per job, row-major (what the per-size pointers return; packing the 64-point sizes is svt_handle_transform*). */
typedef struct SvtHipFwdTxBatchDesc {
    uint8_t  tx_size, reserved[3];
    uint32_t n_jobs;
    uint32_t residual_stride;    /* in samples */
    const int16_t     *residual; /* device pointers */
    const SvtHipTxJob *jobs;     /* src_offset: the block in `residual`; tx_type; pf_shape 0 / 1 / 2 = full / _N2 / _N4 */
    int32_t           *coeff;    /* [n_jobs][W * H] */
} SvtHipFwdTxBatchDesc;
int svt_hip_fwd_txfm_batch(SvtHipContext *ctx, const SvtHipFwdTxBatchDesc *d);

/* ---- inverse transform + reconstruction alone ------------------------------------------------------------------
 * The tail of the RD chain on caller-supplied dequantized coefficients: svt_av1_inv_txfm2d_add_{W}x{H} (Codec/inv_transforms.c:
 * 2459-2716) = recon = clip(pred + inverse(dqcoeff)), read and write planes separate (may alias), as svt_aom_inv_transform_recon /
 * svt_aom_inv_transform_recon8bit drive it in the encode pass (inv_transforms.c:3087-3192). */
typedef struct SvtHipInvTxBatchDesc {
    uint8_t  bit_depth;    /* 8 or 10: clamp ranges of the stages and of the output */
    uint8_t  sample_bytes; /* 1: uint8 planes (bit_depth 8 only); 2: uint16 planes (the `_c` entries' layout for either depth) */
    uint8_t  tx_size;      /* shared by the batch */
    uint8_t  reserved;
    uint32_t n_jobs;
    uint32_t pred_stride, recon_stride; /* in samples */
    const void        *pred;            /* device pointers */
    void              *recon;
    const SvtHipTxJob *jobs;            /* pred_offset: the block in `pred`; src_offset: the block in `recon`; tx_type */
    const int32_t     *dqcoeff;         /* [n_jobs][min(W,32) * min(H,32)] packed like the reference's inverse entries take them */
} SvtHipInvTxBatchDesc;
int svt_hip_inv_txfm_batch(SvtHipContext *ctx, const SvtHipInvTxBatchDesc *d);

/* ---- batched block statistics ---------------------------------------------------------------------------------
 * Per job: SAD, SSE, variance and Hadamard SATD of (src block - ref block).  Restates
 *   svt_nxm_sad_kernel_helper_c / svt_aom_sad_16b_kernel_c          (C_DEFAULT/compute_sad_c.c:20-56,209)
 *   svt_spatial_full_distortion_kernel_c / svt_full_distortion_kernel16_bits_c / svt_aom_sse_c
 *                                                                   (C_DEFAULT/picture_operators_c.c:65-83, Codec/pic_operators.c:174-197)
 *   svt_aom_variance{W}x{H}_c, svt_aom_sub_pixel_variance{W}x{H}_c  (C_DEFAULT/variance.c:256-318)
 *   hadamard_path_c = residual -> svt_aom_hadamard_NxN -> svt_aom_satd over <= 32x32 tiles
 *                                                                   (Codec/enc_mode_config.c:2151-2217)
 *   svt_psy_distortion / svt_psy_distortion_hbd / get_svt_psy_full_dist  (Codec/psy_rd.c:135-293)                           */
typedef struct SvtHipBlockJob {
    uint32_t src_offset, ref_offset; /* sample offsets of the block's top-left sample in the two planes */
    uint8_t  width, height;          /* 1..128 */
    uint8_t  subpel_x, subpel_y;     /* 0..7: the src block is first interpolated at this 1/8-sample phase with the 2-tap bilinear
                                      * filters of svt_aom_sub_pixel_variance{W}x{H}_c (C_DEFAULT/variance.c:28-75,308-318;
                                      * filter.h:39-48): horizontal pass on height+1 rows, then vertical.  (0,0) = src as is */
} SvtHipBlockJob;

typedef struct SvtHipBlockStatsDesc {
    uint8_t  bit_depth;   /* 8: planes are uint8; 10: planes are uint16 */
    uint8_t  temporal_layer_index; /* pcs->temporal_layer_index (0..5), read for facade_dist only */
    uint8_t  spy_rd;               /* EbSvtAv1EncConfiguration.spy_rd (API/EbSvtAv1Enc.h:1020), read for facade_dist only */
    uint8_t  reserved;
    uint32_t n_jobs;
    uint32_t src_stride, ref_stride; /* in samples */
    const void           *src, *ref; /* device pointers */
    const SvtHipBlockJob *jobs;      /* device pointer, n_jobs entries */
    /* per-job outputs, device pointers; any of them may be NULL (not computed) */
    uint32_t *sad;      /* sum |src - ref| */
    uint64_t *sse;      /* sum (src - ref)^2, 64-bit */
    uint32_t *variance; /* sse32 - sum^2 / (w*h), the svt_aom_variance* return value (32-bit wrap like the reference) */
    uint32_t *var_sse;  /* the `sse` out-parameter of svt_aom_variance* (32-bit) */
    uint32_t *satd;     /* hadamard_path_c of a square block (4..128); 0 for other shapes; 8-bit planes only */
    /* PSYEX psy-RD term (Codec/psy_rd.c:135-293): src = input, ref = reconstruction; width and height multiples of 4 */
    double    psy_rd;     /* strength; only used for psy_dist */
    uint64_t *psy_energy; /* svt_psy_distortion / svt_psy_distortion_hbd */
    uint64_t *psy_dist;   /* get_svt_psy_full_dist: (uint64_t)(psy_energy * psy_rd), one fp64 multiply */
    /* PSYEX distortion facades (C_DEFAULT/picture_operators_c.c:85-174), src = input, ref = prediction or reconstruction */
    uint64_t      *psy_sse;       /* svt_spatial_psy_distortion_kernel_c: sse + (uint64_t)(psy_energy * psy_rd) when psy_rd > 0 */
    const uint8_t *pred_mode;     /* [n_jobs] PredictionMode of the candidate, device pointer; mandatory with facade_dist */
    const uint8_t *compound_type; /* [n_jobs] CompoundType, device pointer; mandatory with facade_dist */
    uint64_t      *facade_dist;   /* svt_spatial_full_distortion_kernel_facade: the SSE with the spy-rd mode biases */
    /* svt_aom_highbd_10_variance{W}x{H}_c (Codec/svt_psnr.c:139-177), 10-bit planes only: sse and sum are brought back to the
     * 8-bit scale with rounding ((sse + 8) >> 4, (sum + 2) >> 2) before sse - sum^2 / (w*h), clamped at 0 */
    uint32_t *variance10, *var_sse10;
    /* Optional hierarchical jobs: `n_pyramids` 64x64 regions (device array `pyramids`; width = height = 64, no sub-pixel view).  One wave
     * reads a region's samples ONCE and produces the outputs of its 85 nested square blocks -- the 64x64, its 4 32x32, 16 16x16 and 64 8x8
     * blocks, each level in raster order: SAD / sum / SSE (hence every variance and facade output) and the PSYEX psy energy (a sum over
     * 8x8 tiles for every block size, psy_rd.c:135-274) add up the tree; hadamard_path's SATD is evaluated per size from one staged
     * residual (a 64x64 SATD is the sum of its four 32x32 tiles', enc_mode_config.c:2151-2217).  Region k writes output slots
     * pyramid_out_base + 85 k .. + 84 of the same output arrays (pred_mode / compound_type are read at those slots); plain jobs keep
     * slots 0 .. n_jobs - 1.  Results are those of 85 plain jobs. */
    uint32_t              n_pyramids, pyramid_out_base;
    const SvtHipBlockJob *pyramids;
} SvtHipBlockStatsDesc;
#define SVT_HIP_PYRAMID_BLOCKS 85

/* The integer biases svt_spatial_full_distortion_kernel_facade applies to an SSE (picture_operators_c.c:130-171): host-only
 * arithmetic, for callers that already hold the SSE (e.g. SvtHipRdBatchDesc.sse).  mode / compound_type are the reference's
 * PredictionMode / CompoundType enumerators; temporal_layer_index <= 5. */
uint64_t svt_hip_spy_rd_bias(uint64_t sse, uint32_t area_width, uint32_t area_height, uint8_t mode, uint8_t compound_type,
                             uint8_t temporal_layer_index, double psy_rd, uint8_t spy_rd);

/* Enqueues one batch on the context stream (asynchronous); one wave per job. */
int svt_hip_block_stats_batch(SvtHipContext *ctx, const SvtHipBlockStatsDesc *d);

/* Full-pel motion-compensated prediction from ME results: every 16x16 PU copies the block of `ref` displaced by its
 * best integer MV (sb_best_mv = SvtHipMeResults.sb_best_mv, device pointer; list / ref_idx select the reference).
 * ref / pred are device planes of `bit_depth` 8 (uint8) or 10 (uint16), strides in samples, no padding needed
 * (coordinates are clamped to the picture).  The integer-MV case of inter prediction; feeds svt_hip_rd_batch. */
int svt_hip_fullpel_pred(SvtHipContext *ctx, const void *ref, uint32_t ref_stride, uint32_t width, uint32_t height, uint8_t bit_depth,
                         const uint32_t *sb_best_mv, uint8_t list, uint8_t ref_idx, uint32_t b64_row_start, uint32_t b64_row_count,
                         void *pred, uint32_t pred_stride); /* b64_row_count == 0: all rows from b64_row_start */

/* Several pictures in one launch (at most SVT_HIP_PRED_MAX_JOBS): all share the plane geometry; per job the reference plane, the
 * MV array, the prediction plane and the row band.  Host array of jobs holding device pointers. */
#define SVT_HIP_PRED_MAX_JOBS 16
typedef struct SvtHipPredJob {
    const void     *ref;
    const uint32_t *sb_best_mv;
    void           *pred;
    uint32_t        b64_row_start, b64_row_count; /* count 0: all rows from b64_row_start */
    uint8_t         list, ref_idx, reserved[6];
} SvtHipPredJob;
int svt_hip_fullpel_pred_batch(SvtHipContext *ctx, uint32_t ref_stride, uint32_t width, uint32_t height, uint8_t bit_depth, uint32_t pred_stride,
                               uint32_t n_jobs, const SvtHipPredJob *jobs);

/* get_hvs_modulation_factor (Codec/psy_rd.c:295-307): the psy-rd strength every psy call site passes on (e.g. product_coding_loop.c:972,
 * 4618): x0.4 on intra (I-slice) pictures, x0.75 / x0.9 / x0.95 on temporal layers 0 / 1 / 2, unchanged above.  Host arithmetic (one fp64
 * multiply, same operand order as the reference). */
double svt_hip_hvs_modulation_factor(double psy_rd, int is_islice, uint8_t temporal_layer_index);

/* Scan order of (tx_size, tx_type) as av1_scan_orders holds it (Codec/coefficients.h:2197); returns the length. */
int svt_hip_scan_order(int tx_size, int tx_type, int16_t *scan, int16_t *iscan);
int svt_hip_tx_size_wide(int tx_size);
int svt_hip_tx_size_high(int tx_size);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_DSP_H */
