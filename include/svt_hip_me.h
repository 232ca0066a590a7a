/*
 * svt_hip_me.h -- C-ABI of the MI355X open-loop motion-estimation backend.
 *
 * Plain C, plain pointers and sizes.  Everything in this header is the boundary a maintainer of
 * SVT-AV1-PSYEX binds to (see INTEGRATION.md for the reference-side stub).  Reference citations are
 * relative to the reference tree (Source/Lib/...).
 *
 * Two levels are exported:
 *   1. the batched, per-picture entry (svt_hip_me_picture*) that replaces the body of the b64 loop in
 *      svt_aom_motion_estimation_kernel (Codec/me_process.c:174-290), i.e. N calls of
 *      svt_aom_motion_estimation_b64 (Codec/motion_estimation.c:3076-3153);
 *   2. kernel-level entries with the reference's rtcd prototypes (svt_hip_dsp.h).
 */
#ifndef SVT_HIP_ME_H
#define SVT_HIP_ME_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVT_HIP_OK 0
#define SVT_HIP_ERR_NO_DEVICE 1 /* no gfx950 device / HIP runtime error at init            */
#define SVT_HIP_ERR_BAD_PARAM 2 /* a descriptor failed host-side shape validation            */
#define SVT_HIP_ERR_NO_MEMORY 3 /* hipMalloc failed                                          */
#define SVT_HIP_ERR_LAUNCH 4    /* kernel launch / stream error                              */

#define SVT_HIP_MAX_LISTS 2      /* MAX_NUM_OF_REF_PIC_LIST, Codec/definitions.h:2337 */
#define SVT_HIP_MAX_REFS 4       /* MAX_REF_IDX / REF_LIST_MAX_DEPTH, definitions.h:2352    */
#define SVT_HIP_SQUARE_PU_COUNT 85 /* SQUARE_PU_COUNT, Codec/me_sb_results.h:24             */
#define SVT_HIP_MAX_SAD_VALUE (128 * 128 * 255) /* MAX_SAD_VALUE, Codec/motion_estimation.h:85 */

/* ---- search-control subset of MeContext (Codec/me_context.h:366-509), plain data only ---- */
typedef struct SvtHipSearchArea {
    uint16_t width, height; /* SearchArea, me_context.h */
} SvtHipSearchArea;

typedef struct SvtHipSearchAreaMinMax {
    SvtHipSearchArea sa_min, sa_max; /* SearchAreaMinMax */
} SvtHipSearchAreaMinMax;

typedef struct SvtHipMeConfig {
    /* search methods: 0 = SUB_SAD_SEARCH (every other row, x2), 1 = FULL_SAD_SEARCH (definitions.h:2071-2072) */
    uint8_t hme_search_method, me_search_method;
    uint8_t enable_hme_flag, enable_hme_level0_flag, enable_hme_level1_flag, enable_hme_level2_flag;
    uint16_t num_hme_sa_w, num_hme_sa_h; /* always 2x2 in the reference (enc_mode_config.c:141-142) */
    SvtHipSearchAreaMinMax hme_l0_sa;    /* total HME-L0 area */
    SvtHipSearchArea       hme_l1_sa, hme_l2_sa;
    SvtHipSearchAreaMinMax me_sa;
    /* PreHmeCtrls */
    uint8_t                prehme_enable, prehme_skip_search_line, prehme_l1_early_exit;
    uint8_t                me_type; /* 0: ME_OPEN_LOOP and friends; 1: ME_MCTF (temporal filter, me_context.h:44-51) -- see svt_hip_me_picture */
    SvtHipSearchAreaMinMax prehme_sa_cfg[2];
    /* MeHmeRefPruneCtrls */
    uint8_t  enable_me_hme_ref_pruning;
    uint16_t prune_ref_if_hme_sad_dev_bigger_than_th, prune_ref_if_me_sad_dev_bigger_than_th;
    uint32_t zz_sad_th;
    uint16_t zz_sad_pct;
    uint32_t phme_sad_th;
    uint16_t phme_sad_pct;
    /* MeSrCtrls */
    uint8_t  enable_me_sr_adjustment;
    uint16_t reduce_me_sr_based_on_mv_length_th, stationary_hme_sad_abs_th, stationary_me_sr_divisor;
    uint16_t reduce_me_sr_based_on_hme_sad_abs_th, me_sr_divisor_for_low_hme_sad;
    uint8_t  distance_based_hme_resizing;
    /* Me8x8VarCtrls */
    uint8_t  me_8x8_var_enabled;
    uint32_t me_sr_div4_th, me_sr_div2_th, me_sr_mult2_th;
    /* MvBasedSearchAdj */
    uint8_t  mv_sa_adj_enabled, mv_sa_adj_nearest_ref_only;
    uint16_t mv_sa_adj_mv_size_th, mv_sa_adj_sa_multiplier;
    /* misc */
    int32_t  prune_me_candidates_th;
    uint8_t  use_best_unipred_cand_only;
    uint8_t  reduce_hme_l0_sr_th_min, reduce_hme_l0_sr_th_max;
    uint32_t me_early_exit_th, me_safe_limit_zz_th, prev_me_stage_based_exit_th;
} SvtHipMeConfig;

/* Inputs of svt_aom_sig_deriv_me (Codec/enc_mode_config.c:681-833) that select a preset's search
 * controls.  svt_hip_me_config_from_preset restates that derivation for the open-loop (TASK_PAME) case. */
typedef struct SvtHipMePresetDesc {
    int8_t   enc_mode;          /* EncMode: -3 (MRS) .. 13; presets 0..13 as on the CLI */
    uint8_t  input_resolution;  /* EbInputResolution 0..6 (definitions.h:2079-2085); see svt_hip_input_resolution */
    uint8_t  sc_class1;         /* screen-content class */
    uint8_t  rtc_tune;          /* pred_structure == SVT_AV1_PRED_LOW_DELAY_B */
    uint8_t  temporal_layer_index;
    uint8_t  hierarchical_levels;
    uint32_t qp;                /* static_config.qp (CRF value) */
    uint32_t frame_rate_q16;    /* scs->frame_rate (Q16) */
    uint8_t  safe_limit_nref;   /* scs->mrp_ctrls.safe_limit_nref */
    uint32_t safe_limit_zz_th;
} SvtHipMePresetDesc;

/* ---- pictures ---- */
/* One padded 8-bit luma plane as the reference describes it with EbPictureBufferDesc:
 * buffer_y, stride_y, org_x/org_y (padding), width/height (unpadded). */
typedef struct SvtHipPlaneDesc {
    const uint8_t *buffer_y;
    uint32_t       stride_y;
    uint16_t       org_x, org_y;
    uint16_t       width, height;
} SvtHipPlaneDesc;

typedef struct SvtHipContext   SvtHipContext;   /* one per GPU / process */
typedef struct SvtHipPaPicture SvtHipPaPicture; /* device-resident luma pyramid (EbPaReferenceObject) */

/* Per-picture descriptor: the fields of PictureParentControlSet / MeContext that
 * svt_aom_motion_estimation_b64 reads (me_process.c:183-262, motion_estimation.c passim). */
typedef struct SvtHipMePictureDesc {
    uint64_t picture_number;
    uint16_t aligned_width, aligned_height; /* pcs->aligned_width/height (multiple of 8) */
    uint8_t  num_of_list_to_search;         /* 1 (P) or 2 (B) */
    uint8_t  num_of_ref_pic_to_search[SVT_HIP_MAX_LISTS];
    uint8_t  temporal_layer_index, hierarchical_levels, is_ref, similar_brightness_refs;
    uint8_t  enable_me_8x8, enable_me_16x16; /* pcs.c:1389-1392 */
    uint8_t  max_number_of_pus_per_sb;       /* 85 */
    uint8_t  max_cand, max_refs, max_l0;     /* pcs->pa_me_data->*, pd_process.c:3513-3519 */
    uint8_t  input_resolution;               /* scs->input_resolution */
    uint8_t  only_l_bwd;                     /* scs->mrp_ctrls.only_l_bwd */
    uint8_t  gm_enabled, gm_use_distance_based_active_th; /* pcs->gm_ctrls */
    /* first b64 row handled by this call and number of rows (row-band sharding across GPUs);
     * b64_row_count == 0 means "all rows". */
    uint16_t b64_row_start, b64_row_count;
    uint32_t tf_me_exit_th; /* MeContext.tf_me_exit_th (me_context.h:495), read when cfg.me_type == 1 */
    uint64_t ref_picture_number[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
} SvtHipMePictureDesc;

/* Output arrays (host or device pointers depending on the entry point), all indexed by b64_index
 * in raster order over the whole picture; a row-band call writes only its own rows.
 * n_pu = number of PUs kept per b64 (85, 21 or 5: pcs.c:106-111). */
typedef struct SvtHipMeResults {
    /* MeSbResults (me_sb_results.h:44-51) */
    uint8_t  *total_me_candidate_index; /* [n_b64][n_pu]                                    */
    uint32_t *me_mv_array;              /* [n_b64][n_pu*max_refs]  MvCandidate.as_int       */
    uint8_t  *me_candidate_array;       /* [n_b64][n_pu*max_cand]  MeCandidate bitfield     */
    /* per-b64 pcs scalars written by compute_distortion / gm detection (motion_estimation.c:2964-3008) */
    uint32_t *me_64x64_distortion, *me_32x32_distortion, *me_16x16_distortion, *me_8x8_distortion;
    uint32_t *rc_me_distortion, *me_8x8_cost_variance;
    uint8_t  *stationary_block_present_sb, *rc_me_allow_gm;
    /* optional search-level results (may be NULL): MeContext.p_sb_best_sad / p_sb_best_mv and search_results.
     * Layout [n_b64][2][4][85] in the reference's n_idx order; entries of refs that were pruned (do_ref == 0) hold SVT_HIP_MAX_SAD_VALUE / 0.
     * The slots of (list, reference) pairs the picture does not search at all (list >= num_of_list_to_search, ref >= num_of_ref_pic_to_search[list])
     * hold the same values after svt_hip_me_picture; the asynchronous entries do not touch them (the reference's MeContext arrays hold stale
     * data there: nothing reads them) -- a quarter of the result bytes of a two-reference picture instead of all of them. */
    uint32_t *sb_best_sad, *sb_best_mv;
    int16_t  *hme_sc;  /* [n_b64][2][4][2] (x,y) */
    uint32_t *hme_sad; /* [n_b64][2][4] low 32 bits of SearchResults.hme_sad after me_prune_ref */
    uint8_t  *do_ref;  /* [n_b64][2][4] */
} SvtHipMeResults;

/* number of PUs per b64 kept in MeSbResults (pcs.c:106-111) */
static inline uint32_t svt_hip_me_n_pu(uint8_t enable_me_16x16, uint8_t enable_me_8x8) {
    return enable_me_16x16 ? (enable_me_8x8 ? 85u : 21u) : 5u;
}

/* ---- context ---- */
/* device < 0 selects hipGetDevice's current device.  Fails (non-zero) when no gfx950 GPU is usable:
 * the host then keeps its CPU dispatch (fail closed).
 *
 * THREADING CONTRACT.  One context per GPU is shared by all host threads, like the reference's kernel table is shared by its
 * ME and mode-decision threads (Globals/enc_handle.c:2265,2293; several pictures in flight, Codec/me_process.c:140-172):
 *   - every SYNCHRONOUS entry (host pointers in, results complete on return: svt_hip_me_picture,
 *     svt_hip_dg_detector_hme_level0) may be called from any number of threads at once; each call runs on a stream, parameter
 *     block and result buffer of its own (at most 8 such calls execute concurrently, further callers wait).  The pointer-level
 *     *_hip entries of svt_hip_leaf.h are synchronous too but share ONE stream and lock (svt_hip_leaf.h), and
 *     svt_hip_pa_picture_download copies on -- and waits for -- the context stream: it returns after everything enqueued there;
 *   - every ASYNCHRONOUS entry (device pointers: *_async, svt_hip_rd_batch, svt_hip_*_txfm_batch, svt_hip_block_stats_batch,
 *     svt_hip_fullpel_pred*, svt_hip_pa_picture_create*) enqueues on the ONE stream svt_hip_context_stream() returns.  They
 *     may be called from several threads too (enqueueing is serialised internally), but stream order is call order, and the
 *     caller owns the device buffers until svt_hip_context_sync() or an event of its own says the work is done;
 *   - a picture may be used by any entry as soon as svt_hip_pa_picture_create* has returned (other streams wait for its
 *     planes through an event); destroy it only after the calls that use it have completed;
 *   - svt_hip_last_error() returns the last message of the CALLING thread. */
int  svt_hip_context_create(SvtHipContext **ctx, int device);
void svt_hip_context_destroy(SvtHipContext *ctx);
const char *svt_hip_last_error(const SvtHipContext *ctx);
/* stream used by this context (hipStream_t as void*); callers that own HIP events time on it */
void *svt_hip_context_stream(SvtHipContext *ctx);
int   svt_hip_context_sync(SvtHipContext *ctx);
/* Upper limit of the persistent ME waves per CU of this context's launches (0 = as many as fit, the default).  A pipeline that runs the
 * mode-decision kernels of earlier pictures on another context's stream beside the ME launch -- as the reference runs its ME and
 * mode-decision processes side by side -- lowers it so that both find LDS and registers on every CU. */
int   svt_hip_context_set_me_waves_per_cu(SvtHipContext *ctx, uint32_t waves);
/* The dense pre-pass of the ME launches of this context (on by default; the environment variable SVT_HIP_ME_DENSE=0 turns it off at
 * context creation): the pre-HME strips (prehme_core, Codec/motion_estimation.c:1568-1666) and the HME level-0 regions (hme_level_0,
 * :820-920) of every block -- searches whose windows depend on the block position and the picture distance only -- are made by a kernel of
 * their own ahead of the per-block kernel, which takes their results instead of searching.  Results are identical either way. */
int   svt_hip_context_set_me_dense(SvtHipContext *ctx, int on);
/* With the pre-pass on, the per-block pipeline can run STAGED: a chain of small kernels cut at its searches (control / level-1 searches / control /
 * level-2 searches / integer search and outputs), a block's state travelling through HBM between them -- each kernel has the register and
 * LDS budget of its own part only.  Blocks whose pre-HME / level-0 searches the pre-pass did not make, or whose level-1 / level-2 searches
 * are too large for the search kernels' direct form, go through the one-kernel form at the end of the launch.  on = 0: never; 1 (default): launches of 2048 blocks and more (a small launch is latency-bound: one kernel serves it
 * better than nine); 2: every launch.  SVT_HIP_ME_STAGED=0/1/2 sets it at context creation.  Results are identical either way. */
int   svt_hip_context_set_me_staged(SvtHipContext *ctx, int on);
/* Measurement aid: with timing on, an ME launch records events around each kernel of its chain, and svt_hip_me_launch_times returns the
 * durations (ms) of the LAST launch enqueued on the context stream (it waits for that launch): ms[i] for kernel i of
 * svt_hip_me_chain_kernel_name(i) -- 0 for kernels the launch did not use (the one-kernel form uses the pre-pass and svt_hip_me_b64_kernel
 * only).  The events cost a few microseconds per kernel: off by default. */
#define SVT_HIP_ME_CHAIN_KERNELS 7
int   svt_hip_context_set_me_timing(SvtHipContext *ctx, int on);
int   svt_hip_me_launch_times(SvtHipContext *ctx, float ms[SVT_HIP_ME_CHAIN_KERNELS]);
const char *svt_hip_me_chain_kernel_name(int i);
/* Diagnostics (off by default: the counters cost two device atomics per wave): with counting on, svt_hip_me_dense_counters returns what the
 * launches since its last call did. */
int   svt_hip_context_set_me_counting(SvtHipContext *ctx, int on);
/* out[0] = searches the per-block kernel took from the pre-pass, out[1] = searches it made itself although the pre-pass was
 * on (edge blocks, configurations the pre-pass does not cover), since the last call; waits for the context's streams. */
int   svt_hip_me_dense_counters(SvtHipContext *ctx, unsigned long long out[2]);

/* ---- preset derivation (host only) ---- */
/* Restates svt_aom_sig_deriv_me + svt_aom_sig_deriv_multi_processes' HME flags (enc_mode_config.c:138-833,1632-1642). */
int     svt_hip_me_config_from_preset(const SvtHipMePresetDesc *p, SvtHipMeConfig *cfg);
uint8_t svt_hip_input_resolution(uint32_t width, uint32_t height); /* svt_aom_derive_input_resolution */
uint8_t svt_hip_enable_me_8x8(int8_t enc_mode, uint8_t rtc_tune, uint8_t input_resolution); /* enc_mode_config.c:76-95 */

/* ---- pictures (EbPaReferenceObject equivalents, resident in HBM) ---- */
/* Uploads the full-resolution padded 8-bit luma plane.  quarter / sixteenth may be NULL: the 1/4 and 1/16
 * planes are then produced on the device exactly as svt_aom_downsample_filtering_input_picture does
 * (pic_analysis_process.c:2139-2196: 2x2 box (sum+2)>>2, then svt_aom_generate_padding with 32 / 16 px). */
int  svt_hip_pa_picture_create(SvtHipContext *ctx, const SvtHipPlaneDesc *full, const SvtHipPlaneDesc *quarter,
                               const SvtHipPlaneDesc *sixteenth, SvtHipPaPicture **pic);
/* same, but `full_dev` points to device memory already laid out as `full` describes (no H2D copy) */
int  svt_hip_pa_picture_create_dev(SvtHipContext *ctx, const SvtHipPlaneDesc *full_dev, SvtHipPaPicture **pic);
/* Refill an existing picture (a pooled buffer, as the reference recycles its EbPaReferenceObject buffers) with a new full-resolution plane of
 * the same geometry and rebuild the 1/4 and 1/16 planes on the device; enqueued on the context stream (asynchronous when `full` is page-locked
 * host memory or, with full_on_device != 0, device memory). */
int  svt_hip_pa_picture_update(SvtHipContext *ctx, SvtHipPaPicture *pic, const SvtHipPlaneDesc *full, int full_on_device);
/* The same refill on the context's TRANSFER stream, so that the copy (and the two decimation launches) run beside the kernels of the
 * context stream: ordered behind everything enqueued on the context stream before this call -- whatever may still read the picture's old
 * content -- and beside everything enqueued after it; every entry that reads the picture waits for it through the picture's event.  The
 * usual pipeline: enqueue step k, refill the pictures of step k + 1, enqueue step k + 1, ...  `full` must be page-locked host memory (or device
 * memory) for the copy to be asynchronous.  svt_hip_context_transfer_stream() hands the stream out for the caller's own copies (e.g. results
 * back to the host behind an event recorded on the context stream). */
int  svt_hip_pa_picture_update_ahead(SvtHipContext *ctx, SvtHipPaPicture *pic, const SvtHipPlaneDesc *full, int full_on_device);
void *svt_hip_context_transfer_stream(SvtHipContext *ctx);
void svt_hip_pa_picture_destroy(SvtHipContext *ctx, SvtHipPaPicture *pic);
/* copies level (0 = sixteenth, 1 = quarter, 2 = full) back into a host plane of identical geometry */
int  svt_hip_pa_picture_download(SvtHipContext *ctx, const SvtHipPaPicture *pic, int level, uint8_t *dst,
                                 uint32_t dst_stride);
int  svt_hip_pa_picture_geometry(const SvtHipPaPicture *pic, int level, SvtHipPlaneDesc *out);

/* ---- the batched entry ---- */
/* One call = svt_aom_motion_estimation_b64 for every b64 of `cur` (or of a row band).  `res` holds HOST
 * pointers; results are complete when the call returns.  Returns 0, or an error after which the host runs
 * its CPU loop instead. */
int svt_hip_me_picture(SvtHipContext *ctx, const SvtHipMeConfig *cfg, const SvtHipMePictureDesc *desc,
                       const SvtHipPaPicture *cur, const SvtHipPaPicture *const refs[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS],
                       SvtHipMeResults *res);
/* Asynchronous form: `res` holds DEVICE pointers, work is enqueued on the context stream and the call
 * returns without waiting for the GPU (used by bench.py and by the multi-GPU path, which all-gathers device buffers): the
 * descriptors are copied into a ring of pinned parameter blocks, so up to 4 launches may be enqueued ahead before the call
 * waits for the oldest one's parameter copy. */
int svt_hip_me_picture_async(SvtHipContext *ctx, const SvtHipMeConfig *cfg, const SvtHipMePictureDesc *desc,
                             const SvtHipPaPicture *cur,
                             const SvtHipPaPicture *const refs[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS],
                             const SvtHipMeResults *res_dev);

/* Several pictures in ONE launch (the reference keeps several pictures in flight in its ME threads,
 * me_process.c:140-172 picks them up segment by segment): the b64 jobs of all of them feed the same persistent
 * workgroups, which matters when one picture -- or one GPU's row band of it -- has fewer blocks than the GPU has
 * resident workgroups.  Every job is what one svt_hip_me_picture_async call takes; at most 16 per call. */
typedef struct SvtHipMeJob {
    const SvtHipMeConfig      *cfg;
    const SvtHipMePictureDesc *desc;
    const SvtHipPaPicture     *cur;
    const SvtHipPaPicture     *refs[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
    const SvtHipMeResults     *results; /* device pointers */
} SvtHipMeJob;
int svt_hip_me_pictures_async(SvtHipContext *ctx, uint32_t n_pictures, const SvtHipMeJob *jobs);

/* ---- dynamic-GOP detector HME (me_type ME_DG_DETECTOR: me_process.c:113-118,326-331) ---- */
/* The fields of DGDetectorMetrics (pcs.h:731-737) that dg_detector_hme_level0 accumulates (pd_process.c:541-581). */
typedef struct SvtHipDgMetrics {
    uint64_t tot_dist;       /* sum of the level-0 SADs */
    uint32_t tot_cplx;       /* blocks with SAD > 16*16*30 */
    uint32_t tot_active;     /* blocks with a non-zero vector */
    int32_t  sum_in_vectors; /* inward (+) / outward (-) vector components */
    uint32_t reserved;
} SvtHipDgMetrics;

/* One call = dg_detector_hme_level0(ppcs, seg_idx) for EVERY segment of the picture (pd_process.c:492-588):
 * per b64, a full-SAD svt_sad_loop_kernel search of the 16x16 block of `src`'s sixteenth plane in `ref`'s
 * sixteenth plane (early_hme_b64, pd_process.c:393-490; search area 16 / 64 / 128 squared by input_resolution),
 * then the four metric sums.  `metrics` is a HOST pointer; `b64_sad` ([n_b64], hme_level0_sad) and `b64_mv`
 * ([n_b64][2] = sr_center col,row in full-resolution pixels) are optional HOST pointers to the per-block
 * results the reference only keeps in locals.  Complete on return. */
int svt_hip_dg_detector_hme_level0(SvtHipContext *ctx, const SvtHipPaPicture *src, const SvtHipPaPicture *ref,
                                   uint16_t aligned_width, uint16_t aligned_height, uint8_t input_resolution,
                                   SvtHipDgMetrics *metrics, uint32_t *b64_sad, int16_t *b64_mv);
/* Asynchronous form: all three are DEVICE pointers (b64_sad / b64_mv may be NULL); enqueued on the context stream. */
int svt_hip_dg_detector_hme_level0_async(SvtHipContext *ctx, const SvtHipPaPicture *src, const SvtHipPaPicture *ref,
                                         uint16_t aligned_width, uint16_t aligned_height, uint8_t input_resolution,
                                         SvtHipDgMetrics *metrics_dev, uint32_t *b64_sad_dev, int16_t *b64_mv_dev);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_ME_H */
