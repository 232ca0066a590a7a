/*
 * ref_harness.c -- TEST INFRASTRUCTURE, built only where /root/reference exists.
 *
 * Glue that drives the REFERENCE's own functions, compiled unmodified from /root/reference by
 * oracle/Makefile into oracle/_ref/libsvtref.so.  It contains no algorithm: it fills the reference's own
 * structs (MeContext, PictureParentControlSet, EbPictureBufferDesc ...) from the plain descriptors of
 * include/svt_hip_me.h, installs the reference's `_c` (or AVX2) kernels into its rtcd pointers and calls
 * svt_aom_sig_deriv_me / svt_aom_motion_estimation_b64.  Used to pin oracle/me_oracle.c and to generate
 * tests/golden fixtures; optionally as the "reference" CPU baseline of bench.py.
 */
#include <stdlib.h>
#include <string.h>

#include "definitions.h"
#include "pcs.h"
#include "sequence_control_set.h"
#include "me_context.h"
#include "me_sb_results.h"
#include "motion_estimation.h"
#include "aom_dsp_rtcd.h"
#include "common_dsp_rtcd.h"
#include "compute_sad_c.h"
#include "me_sad_calculation.h"
#include "enc_mode_config.h"
#include "coefficients.h"
#include "inv_transforms.h"
#include "transforms.h"
#include "full_loop.h"

#include "../include/svt_hip_me.h"

/* AVX2 / SSE4.1 / SSE2 variants (declared in aom_dsp_rtcd.h when ARCH_X86_64) */
static int g_simd = 0;

/* Installs what svt_aom_setup_rtcd_internal (Codec/aom_dsp_rtcd.c:188, :505-515) would install for the
 * pointers this path calls: flags == 0 -> `_c`, otherwise the AVX2/SSE4.1/SSE2 picks of that table. */
void ref_set_simd(int use_avx2) {
    g_simd = use_avx2;
    if (use_avx2 == 2) return; /* 2: leave the pointers as an external installer set them (ref_rtcd_slot) */
#ifdef REF_WITH_AVX2
    if (use_avx2) {
        svt_sad_loop_kernel                        = svt_sad_loop_kernel_avx2_intrin;
        svt_nxm_sad_kernel                         = svt_nxm_sad_kernel_helper_avx2;
        svt_ext_all_sad_calculation_8x8_16x16      = svt_ext_all_sad_calculation_8x8_16x16_avx2;
        svt_ext_eight_sad_calculation_32x32_64x64  = svt_ext_eight_sad_calculation_32x32_64x64_avx2;
        svt_ext_sad_calculation_8x8_16x16          = svt_ext_sad_calculation_8x8_16x16_avx2_intrin;
        svt_ext_sad_calculation_32x32_64x64        = svt_ext_sad_calculation_32x32_64x64_sse4_intrin;
        svt_initialize_buffer_32bits               = svt_initialize_buffer_32bits_sse2_intrin;
        return;
    }
#endif
    svt_sad_loop_kernel                       = svt_sad_loop_kernel_c;
    svt_nxm_sad_kernel                        = svt_nxm_sad_kernel_helper_c;
    svt_ext_all_sad_calculation_8x8_16x16     = svt_ext_all_sad_calculation_8x8_16x16_c;
    svt_ext_eight_sad_calculation_32x32_64x64 = svt_ext_eight_sad_calculation_32x32_64x64_c;
    svt_ext_sad_calculation_8x8_16x16         = svt_ext_sad_calculation_8x8_16x16_c;
    svt_ext_sad_calculation_32x32_64x64       = svt_ext_sad_calculation_32x32_64x64_c;
    svt_initialize_buffer_32bits              = svt_initialize_buffer_32bits_c;
}

int ref_has_avx2(void) {
#ifdef REF_WITH_AVX2
    return 1;
#else
    return 0;
#endif
}

static void cfg_from_ctx(const MeContext *m, SvtHipMeConfig *c) {
    memset(c, 0, sizeof(*c));
    c->hme_search_method      = m->hme_search_method;
    c->me_search_method       = m->me_search_method;
    c->enable_hme_flag        = m->enable_hme_flag;
    c->enable_hme_level0_flag = m->enable_hme_level0_flag;
    c->enable_hme_level1_flag = m->enable_hme_level1_flag;
    c->enable_hme_level2_flag = m->enable_hme_level2_flag;
    c->num_hme_sa_w           = m->num_hme_sa_w;
    c->num_hme_sa_h           = m->num_hme_sa_h;
    memcpy(&c->hme_l0_sa, &m->hme_l0_sa, sizeof(c->hme_l0_sa));
    memcpy(&c->hme_l1_sa, &m->hme_l1_sa, sizeof(c->hme_l1_sa));
    memcpy(&c->hme_l2_sa, &m->hme_l2_sa, sizeof(c->hme_l2_sa));
    memcpy(&c->me_sa, &m->me_sa, sizeof(c->me_sa));
    c->prehme_enable           = m->prehme_ctrl.enable;
    c->prehme_skip_search_line = m->prehme_ctrl.skip_search_line;
    c->prehme_l1_early_exit    = m->prehme_ctrl.l1_early_exit;
    memcpy(c->prehme_sa_cfg, m->prehme_ctrl.prehme_sa_cfg, sizeof(c->prehme_sa_cfg));
    c->enable_me_hme_ref_pruning               = m->me_hme_prune_ctrls.enable_me_hme_ref_pruning;
    c->prune_ref_if_hme_sad_dev_bigger_than_th = m->me_hme_prune_ctrls.prune_ref_if_hme_sad_dev_bigger_than_th;
    c->prune_ref_if_me_sad_dev_bigger_than_th  = m->me_hme_prune_ctrls.prune_ref_if_me_sad_dev_bigger_than_th;
    c->zz_sad_th                               = m->me_hme_prune_ctrls.zz_sad_th;
    c->zz_sad_pct                              = m->me_hme_prune_ctrls.zz_sad_pct;
    c->phme_sad_th                             = m->me_hme_prune_ctrls.phme_sad_th;
    c->phme_sad_pct                            = m->me_hme_prune_ctrls.phme_sad_pct;
    c->enable_me_sr_adjustment                 = m->me_sr_adjustment_ctrls.enable_me_sr_adjustment;
    c->reduce_me_sr_based_on_mv_length_th      = m->me_sr_adjustment_ctrls.reduce_me_sr_based_on_mv_length_th;
    c->stationary_hme_sad_abs_th               = m->me_sr_adjustment_ctrls.stationary_hme_sad_abs_th;
    c->stationary_me_sr_divisor                = m->me_sr_adjustment_ctrls.stationary_me_sr_divisor;
    c->reduce_me_sr_based_on_hme_sad_abs_th    = m->me_sr_adjustment_ctrls.reduce_me_sr_based_on_hme_sad_abs_th;
    c->me_sr_divisor_for_low_hme_sad           = m->me_sr_adjustment_ctrls.me_sr_divisor_for_low_hme_sad;
    c->distance_based_hme_resizing             = m->me_sr_adjustment_ctrls.distance_based_hme_resizing;
    c->me_8x8_var_enabled                      = m->me_8x8_var_ctrls.enabled;
    c->me_sr_div4_th                           = m->me_8x8_var_ctrls.me_sr_div4_th;
    c->me_sr_div2_th                           = m->me_8x8_var_ctrls.me_sr_div2_th;
    c->me_sr_mult2_th                          = m->me_8x8_var_ctrls.me_sr_mult2_th;
    c->mv_sa_adj_enabled                       = m->mv_based_sa_adj.enabled;
    c->mv_sa_adj_nearest_ref_only              = m->mv_based_sa_adj.nearest_ref_only;
    c->mv_sa_adj_mv_size_th                    = m->mv_based_sa_adj.mv_size_th;
    c->mv_sa_adj_sa_multiplier                 = m->mv_based_sa_adj.sa_multiplier;
    c->prune_me_candidates_th                  = m->prune_me_candidates_th;
    c->use_best_unipred_cand_only              = m->use_best_unipred_cand_only;
    c->reduce_hme_l0_sr_th_min                 = m->reduce_hme_l0_sr_th_min;
    c->reduce_hme_l0_sr_th_max                 = m->reduce_hme_l0_sr_th_max;
    c->me_early_exit_th                        = m->me_early_exit_th;
    c->me_safe_limit_zz_th                     = m->me_safe_limit_zz_th;
    c->prev_me_stage_based_exit_th             = m->prev_me_stage_based_exit_th;
}

static void ctx_from_cfg(const SvtHipMeConfig *c, MeContext *m) {
    m->hme_search_method      = c->hme_search_method;
    m->me_search_method       = c->me_search_method;
    m->enable_hme_flag        = c->enable_hme_flag;
    m->enable_hme_level0_flag = c->enable_hme_level0_flag;
    m->enable_hme_level1_flag = c->enable_hme_level1_flag;
    m->enable_hme_level2_flag = c->enable_hme_level2_flag;
    m->num_hme_sa_w           = c->num_hme_sa_w;
    m->num_hme_sa_h           = c->num_hme_sa_h;
    memcpy(&m->hme_l0_sa, &c->hme_l0_sa, sizeof(c->hme_l0_sa));
    memcpy(&m->hme_l1_sa, &c->hme_l1_sa, sizeof(c->hme_l1_sa));
    memcpy(&m->hme_l2_sa, &c->hme_l2_sa, sizeof(c->hme_l2_sa));
    memcpy(&m->me_sa, &c->me_sa, sizeof(c->me_sa));
    m->prehme_ctrl.enable           = c->prehme_enable;
    m->prehme_ctrl.skip_search_line = c->prehme_skip_search_line;
    m->prehme_ctrl.l1_early_exit    = c->prehme_l1_early_exit;
    memcpy(m->prehme_ctrl.prehme_sa_cfg, c->prehme_sa_cfg, sizeof(c->prehme_sa_cfg));
    m->me_hme_prune_ctrls.enable_me_hme_ref_pruning               = c->enable_me_hme_ref_pruning;
    m->me_hme_prune_ctrls.prune_ref_if_hme_sad_dev_bigger_than_th = c->prune_ref_if_hme_sad_dev_bigger_than_th;
    m->me_hme_prune_ctrls.prune_ref_if_me_sad_dev_bigger_than_th  = c->prune_ref_if_me_sad_dev_bigger_than_th;
    m->me_hme_prune_ctrls.zz_sad_th                               = c->zz_sad_th;
    m->me_hme_prune_ctrls.zz_sad_pct                              = c->zz_sad_pct;
    m->me_hme_prune_ctrls.phme_sad_th                             = c->phme_sad_th;
    m->me_hme_prune_ctrls.phme_sad_pct                            = c->phme_sad_pct;
    m->me_sr_adjustment_ctrls.enable_me_sr_adjustment             = c->enable_me_sr_adjustment;
    m->me_sr_adjustment_ctrls.reduce_me_sr_based_on_mv_length_th  = c->reduce_me_sr_based_on_mv_length_th;
    m->me_sr_adjustment_ctrls.stationary_hme_sad_abs_th           = c->stationary_hme_sad_abs_th;
    m->me_sr_adjustment_ctrls.stationary_me_sr_divisor            = c->stationary_me_sr_divisor;
    m->me_sr_adjustment_ctrls.reduce_me_sr_based_on_hme_sad_abs_th = c->reduce_me_sr_based_on_hme_sad_abs_th;
    m->me_sr_adjustment_ctrls.me_sr_divisor_for_low_hme_sad       = c->me_sr_divisor_for_low_hme_sad;
    m->me_sr_adjustment_ctrls.distance_based_hme_resizing         = c->distance_based_hme_resizing;
    m->me_8x8_var_ctrls.enabled                                   = c->me_8x8_var_enabled;
    m->me_8x8_var_ctrls.me_sr_div4_th                             = c->me_sr_div4_th;
    m->me_8x8_var_ctrls.me_sr_div2_th                             = c->me_sr_div2_th;
    m->me_8x8_var_ctrls.me_sr_mult2_th                            = c->me_sr_mult2_th;
    m->mv_based_sa_adj.enabled                                    = c->mv_sa_adj_enabled;
    m->mv_based_sa_adj.nearest_ref_only                           = c->mv_sa_adj_nearest_ref_only;
    m->mv_based_sa_adj.mv_size_th                                 = c->mv_sa_adj_mv_size_th;
    m->mv_based_sa_adj.sa_multiplier                              = c->mv_sa_adj_sa_multiplier;
    m->prune_me_candidates_th                                     = c->prune_me_candidates_th;
    m->use_best_unipred_cand_only                                 = c->use_best_unipred_cand_only;
    m->reduce_hme_l0_sr_th_min                                    = c->reduce_hme_l0_sr_th_min;
    m->reduce_hme_l0_sr_th_max                                    = c->reduce_hme_l0_sr_th_max;
    m->me_early_exit_th                                           = c->me_early_exit_th;
    m->me_safe_limit_zz_th                                        = c->me_safe_limit_zz_th;
    m->prev_me_stage_based_exit_th                                = c->prev_me_stage_based_exit_th;
}

/* Runs the reference's svt_aom_sig_deriv_me (enc_mode_config.c:681) on a pcs/scs that carry only the
 * inputs that function reads.  The four HME enable flags are set as svt_aom_sig_deriv_multi_processes
 * sets them (enc_mode_config.c:1634-1645) -- that function is not callable on a bare pcs. */
int ref_me_config_from_preset(const SvtHipMePresetDesc *p, SvtHipMeConfig *cfg) {
    SequenceControlSet      *scs = calloc(1, sizeof(*scs));
    PictureParentControlSet *pcs = calloc(1, sizeof(*pcs));
    MeContext               *m   = calloc(1, sizeof(*m));
    if (!scs || !pcs || !m)
        return 1;
    scs->static_config.pred_structure = p->rtc_tune ? SVT_AV1_PRED_LOW_DELAY_B : SVT_AV1_PRED_RANDOM_ACCESS;
    scs->static_config.qp             = p->qp;
    scs->input_resolution             = (EbInputResolution)p->input_resolution;
    scs->frame_rate                   = p->frame_rate_q16;
    scs->mrp_ctrls.safe_limit_nref    = p->safe_limit_nref;
    scs->mrp_ctrls.safe_limit_zz_th   = p->safe_limit_zz_th;
    pcs->scs                          = scs;
    pcs->enc_mode                     = (EncMode)p->enc_mode;
    pcs->sc_class1                    = p->sc_class1;
    pcs->temporal_layer_index         = p->temporal_layer_index;
    pcs->hierarchical_levels          = p->hierarchical_levels;
    pcs->enable_hme_flag              = 1;
    pcs->enable_hme_level0_flag       = 1;
    pcs->enable_hme_level1_flag       = 1;
    pcs->enable_hme_level2_flag       = (p->sc_class1 || p->enc_mode <= ENC_M6) ? 1 : 0;
    pcs->use_best_me_unipred_cand_only = p->enc_mode <= ENC_M3 ? 0 : 1; /* enc_mode_config.c:1845-1848 */
    svt_aom_sig_deriv_me(scs, pcs, m);
    cfg_from_ctx(m, cfg);
    free(m);
    free(pcs);
    free(scs);
    return 0;
}

static EbPictureBufferDesc *make_desc(const SvtHipPlaneDesc *p) {
    EbPictureBufferDesc *d = calloc(1, sizeof(*d));
    d->buffer_y            = (EbByte)p->buffer_y;
    d->stride_y            = p->stride_y;
    d->org_x               = p->org_x;
    d->org_y               = p->org_y;
    d->width               = p->width;
    d->height              = p->height;
    d->max_width           = p->width;
    d->max_height          = p->height;
    return d;
}

/* The b64 loop of svt_aom_motion_estimation_kernel (me_process.c:174-290) around the reference's own
 * svt_aom_motion_estimation_b64.  Same argument convention as orc_me_picture. */
int ref_me_picture(const SvtHipMeConfig *cfg, const SvtHipMePictureDesc *desc, const SvtHipPlaneDesc cur_planes[3],
                   const SvtHipPlaneDesc ref_planes[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][3], SvtHipMeResults *res) {
    ref_set_simd(g_simd);
    SequenceControlSet      *scs = calloc(1, sizeof(*scs));
    PictureParentControlSet *pcs = calloc(1, sizeof(*pcs));
    MeContext               *m   = calloc(1, sizeof(*m));
    MotionEstimationData    *med = calloc(1, sizeof(*med));
    const uint32_t w64 = (desc->aligned_width + 63) / 64, h64 = (desc->aligned_height + 63) / 64, nb = w64 * h64;
    const uint32_t n_pu = svt_hip_me_n_pu(desc->enable_me_16x16, desc->enable_me_8x8);
    scs->input_resolution     = (EbInputResolution)desc->input_resolution;
    scs->mrp_ctrls.only_l_bwd = desc->only_l_bwd;
    scs->b64_size             = 64;
    pcs->scs                  = scs;
    pcs->picture_number       = desc->picture_number;
    pcs->aligned_width        = desc->aligned_width;
    pcs->aligned_height       = desc->aligned_height;
    pcs->hierarchical_levels  = desc->hierarchical_levels;
    pcs->temporal_layer_index = desc->temporal_layer_index;
    pcs->similar_brightness_refs = desc->similar_brightness_refs;
    pcs->enable_me_8x8        = desc->enable_me_8x8;
    pcs->enable_me_16x16      = desc->enable_me_16x16;
    pcs->max_number_of_pus_per_sb = desc->max_number_of_pus_per_sb;
    pcs->gm_ctrls.enabled                      = desc->gm_enabled;
    pcs->gm_ctrls.use_distance_based_active_th = desc->gm_use_distance_based_active_th;
    pcs->pa_me_data           = med;
    med->max_cand             = desc->max_cand;
    med->max_refs             = desc->max_refs;
    med->max_l0               = desc->max_l0;
    med->me_results           = calloc(nb, sizeof(MeSbResults *));
    pcs->b64_geom             = calloc(nb, sizeof(B64Geom));
    pcs->me_64x64_distortion  = res->me_64x64_distortion;
    pcs->me_32x32_distortion  = res->me_32x32_distortion;
    pcs->me_16x16_distortion  = res->me_16x16_distortion;
    pcs->me_8x8_distortion    = res->me_8x8_distortion;
    pcs->rc_me_distortion     = res->rc_me_distortion;
    pcs->me_8x8_cost_variance = res->me_8x8_cost_variance;
    pcs->stationary_block_present_sb = res->stationary_block_present_sb;
    pcs->rc_me_allow_gm              = res->rc_me_allow_gm;
    for (uint32_t b = 0; b < nb; b++) {
        MeSbResults *r              = calloc(1, sizeof(*r));
        r->total_me_candidate_index = res->total_me_candidate_index + (size_t)b * n_pu;
        r->me_mv_array              = (MvCandidate *)(res->me_mv_array + (size_t)b * n_pu * desc->max_refs);
        r->me_candidate_array       = (MeCandidate *)(res->me_candidate_array + (size_t)b * n_pu * desc->max_cand);
        med->me_results[b]          = r;
        B64Geom *g                  = &pcs->b64_geom[b];
        g->org_x                    = (uint16_t)((b % w64) * 64);
        g->org_y                    = (uint16_t)((b / w64) * 64);
        g->width                    = (uint8_t)((desc->aligned_width - g->org_x) < 64 ? desc->aligned_width - g->org_x : 64);
        g->height                   = (uint8_t)((desc->aligned_height - g->org_y) < 64 ? desc->aligned_height - g->org_y : 64);
    }
    ctx_from_cfg(cfg, m);
    m->me_type               = cfg->me_type == 1 ? ME_MCTF : ME_OPEN_LOOP;
    m->tf_me_exit_th         = (uint16_t)desc->tf_me_exit_th;
    m->num_of_list_to_search = desc->num_of_list_to_search;
    m->num_of_ref_pic_to_search[0] = desc->num_of_ref_pic_to_search[0];
    m->num_of_ref_pic_to_search[1] = desc->num_of_ref_pic_to_search[1];
    m->temporal_layer_index  = desc->temporal_layer_index;
    m->is_ref                = desc->is_ref;
    EbPictureBufferDesc *cur[3], *refs[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][3];
    memset(refs, 0, sizeof(refs));
    for (int l = 0; l < 3; l++) cur[l] = make_desc(&cur_planes[l]);
    for (int li = 0; li < desc->num_of_list_to_search; li++)
        for (int ri = 0; ri < desc->num_of_ref_pic_to_search[li]; ri++) {
            for (int l = 0; l < 3; l++) refs[li][ri][l] = make_desc(&ref_planes[li][ri][l]);
            m->me_ds_ref_array[li][ri].sixteenth_picture_ptr = refs[li][ri][0];
            m->me_ds_ref_array[li][ri].quarter_picture_ptr   = refs[li][ri][1];
            m->me_ds_ref_array[li][ri].picture_ptr           = refs[li][ri][2];
            m->me_ds_ref_array[li][ri].picture_number        = desc->ref_picture_number[li][ri];
        }
    const uint32_t row0 = desc->b64_row_start, nrow = desc->b64_row_count ? desc->b64_row_count : h64 - row0;
    for (uint32_t by = row0; by < row0 + nrow && by < h64; by++)
        for (uint32_t bx = 0; bx < w64; bx++) {
            const uint32_t b = bx + by * w64, ox = bx * 64, oy = by * 64;
            /* me_process.c:183-214 */
            m->b64_src_ptr    = &cur[2]->buffer_y[(cur[2]->org_y + oy) * cur[2]->stride_y + cur[2]->org_x + ox];
            m->b64_src_stride = cur[2]->stride_y;
            m->quarter_b64_buffer = &cur[1]->buffer_y[(cur[1]->org_y + (oy >> 1)) * cur[1]->stride_y + cur[1]->org_x + (ox >> 1)];
            m->quarter_b64_buffer_stride = cur[1]->stride_y;
            m->sixteenth_b64_buffer = &cur[0]->buffer_y[(cur[0]->org_y + (oy >> 2)) * cur[0]->stride_y + cur[0]->org_x + (ox >> 2)];
            m->sixteenth_b64_buffer_stride = cur[0]->stride_y;
            svt_aom_motion_estimation_b64(pcs, b, ox, oy, m, cur[2]);
            /* canonical search-level outputs (see SvtHipMeResults) */
            for (int li = 0; li < 2; li++)
                for (int ri = 0; ri < 4; ri++) {
                    const int    live = li < desc->num_of_list_to_search && ri < desc->num_of_ref_pic_to_search[li];
                    const size_t k    = ((size_t)b * 2 + li) * 4 + ri;
                    if (res->do_ref) res->do_ref[k] = live ? m->search_results[li][ri].do_ref : 0;
                    if (res->hme_sc) {
                        res->hme_sc[2 * k]     = live ? m->search_results[li][ri].hme_sc_x : 0;
                        res->hme_sc[2 * k + 1] = live ? m->search_results[li][ri].hme_sc_y : 0;
                    }
                    if (res->hme_sad) res->hme_sad[k] = live ? (uint32_t)m->search_results[li][ri].hme_sad : 0;
                    /* ME_MCTF early exit (motion_estimation.c:3109-3113) leaves p_sb_best_sad of the previous block: canonical MAX */
                    const int tf_exit = cfg->me_type == 1 && m->search_results[0][0].hme_sad < m->tf_me_exit_th;
                    for (int n = 0; n < 85; n++) {
                        const int ok = live && m->search_results[li][ri].do_ref;
                        if (res->sb_best_sad) res->sb_best_sad[k * 85 + n] = (ok && !tf_exit) ? m->p_sb_best_sad[li][ri][n] : MAX_SAD_VALUE;
                        if (res->sb_best_mv) res->sb_best_mv[k * 85 + n] = ok ? m->p_sb_best_mv[li][ri][n] : 0;
                    }
                }
        }
    for (uint32_t b = 0; b < nb; b++) free(med->me_results[b]);
    for (int l = 0; l < 3; l++) free(cur[l]);
    for (int li = 0; li < 2; li++)
        for (int ri = 0; ri < 4; ri++)
            for (int l = 0; l < 3; l++) free(refs[li][ri][l]);
    free(med->me_results);
    free(pcs->b64_geom);
    free(med);
    free(m);
    free(pcs);
    free(scs);
    return 0;
}

/* dg_detector_hme_level0 (pd_process.c:492-588) run for every segment of a seg_cols x seg_rows split, exactly as
 * svt_aom_motion_estimation_kernel does for ME_DG_DETECTOR tasks (me_process.c:326-331).  Returns the metrics. */
#include "svt_threads.h"
#include "reference_object.h"
void dg_detector_hme_level0(struct PictureParentControlSet *ppcs, uint32_t seg_idx);

int ref_dg_detector(const SvtHipPlaneDesc *src16, const SvtHipPlaneDesc *ref16, uint16_t aligned_width, uint16_t aligned_height,
                    uint8_t input_resolution, uint32_t seg_cols, uint32_t seg_rows, SvtHipDgMetrics *out) {
    ref_set_simd(g_simd);
    SequenceControlSet      *scs = calloc(1, sizeof(*scs));
    PictureParentControlSet *pcs = calloc(1, sizeof(*pcs)), *rpcs = calloc(1, sizeof(*rpcs));
    EbObjectWrapper          wrap[2];
    EbPaReferenceObject      pa[2];
    DGDetectorSeg            dg;
    memset(wrap, 0, sizeof(wrap)); memset(pa, 0, sizeof(pa)); memset(&dg, 0, sizeof(dg));
    EbPictureBufferDesc *s = make_desc(src16), *r = make_desc(ref16);
    pa[0].sixteenth_downsampled_picture_ptr = s;
    pa[1].sixteenth_downsampled_picture_ptr = r;
    wrap[0].object_ptr = &pa[0];
    wrap[1].object_ptr = &pa[1];
    scs->b64_size = 64;
    pcs->scs = scs; rpcs->scs = scs;
    pcs->pa_ref_pic_wrapper  = &wrap[0];
    rpcs->pa_ref_pic_wrapper = &wrap[1];
    pcs->input_resolution = (EbInputResolution)input_resolution;
    pcs->aligned_width    = aligned_width;
    pcs->aligned_height   = aligned_height;
    pcs->me_segments_column_count = (uint8_t)seg_cols;
    pcs->me_segments_row_count    = (uint8_t)seg_rows;
    pcs->me_segments_total_count  = (uint16_t)(seg_cols * seg_rows);
    pcs->dg_detector = &dg;
    dg.ref_pic       = rpcs;
    dg.metrics_mutex  = svt_create_mutex();
    dg.frame_done_sem = svt_create_semaphore(0, 1);
    for (uint32_t seg = 0; seg < seg_cols * seg_rows; seg++) dg_detector_hme_level0(pcs, seg);
    out->tot_dist = dg.metrics.tot_dist; out->tot_cplx = dg.metrics.tot_cplx; out->tot_active = dg.metrics.tot_active;
    out->sum_in_vectors = dg.metrics.sum_in_vectors; out->reserved = dg.metrics.seg_completed;
    svt_destroy_mutex(dg.metrics_mutex);
    svt_destroy_semaphore(dg.frame_done_sem);
    free(s); free(r); free(pcs); free(rpcs); free(scs);
    return 0;
}

size_t ref_sizeof_me_context(void) { return sizeof(MeContext); }

/* ---- accessors for static / header-only data of the reference ---- */
/* av1_scan_orders[tx_size][tx_type] (Codec/coefficients.h:2197): copies scan / iscan, returns the length */
int ref_scan_order(int tx_size, int tx_type, int16_t *scan, int16_t *iscan) {
    const int w = tx_size_wide[tx_size] > 32 ? 32 : tx_size_wide[tx_size], h = tx_size_high[tx_size] > 32 ? 32 : tx_size_high[tx_size];
    const ScanOrder *so = &av1_scan_orders[tx_size][tx_type];
    memcpy(scan, so->scan, sizeof(int16_t) * w * h);
    memcpy(iscan, so->iscan, sizeof(int16_t) * w * h);
    return w * h;
}
const int32_t *ref_cospi(int bit) { return cospi_arr(bit); }
const int32_t *ref_sinpi(int bit) { return sinpi_arr(bit); }

/* hadamard_path_c (Codec/enc_mode_config.c:2151-2217) on a square 8-bit block: fills the Buf2D arguments the way
 * svt_aom_check_high_freq does (:2271-2300) and installs the `_c` kernels it dispatches through. */
uint32_t ref_hadamard_path(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride, uint32_t bsize_wide) {
    static int16_t res[32 * 32];
    static int32_t coeff[128 * 128];
    svt_residual_kernel8bit = svt_residual_kernel8bit_c;

    svt_aom_hadamard_8x8    = svt_aom_hadamard_8x8_c;
    svt_aom_hadamard_16x16  = svt_aom_hadamard_16x16_c;
    svt_aom_hadamard_32x32  = svt_aom_hadamard_32x32_c;
    svt_aom_satd            = svt_aom_satd_c;
    BlockSize bsize;
    switch (bsize_wide) {
    case 4: bsize = BLOCK_4X4; break;
    case 8: bsize = BLOCK_8X8; break;
    case 16: bsize = BLOCK_16X16; break;
    case 32: bsize = BLOCK_32X32; break;
    case 64: bsize = BLOCK_64X64; break;
    default: bsize = BLOCK_128X128; break;
    }
    Buf2D r = {(uint8_t *)res, NULL, 0, 0, 32}, c = {(uint8_t *)coeff, NULL, 0, 0, (int)bsize_wide};
    Buf2D i = {input, NULL, 0, 0, (int)input_stride}, p = {pred, NULL, 0, 0, (int)pred_stride};
    return hadamard_path_c(r, c, i, p, bsize);
}

/* =====================================================================================================================
 * The RD chain of tx_type_search (Codec/product_coding_loop.c:4764-4934) driven through the reference's OWN kernels -- its
 * rtcd function pointers, loaded with the `_c` bodies or with the AVX2 / SSE4.1 intrinsics the x86 dispatch would pick
 * (Codec/common_dsp_rtcd.c:466-560, Codec/aom_dsp_rtcd.c:188-300; the dav1d .asm inverse transforms need nasm, so the inverse
 * takes the SSE4.1 intrinsics).  Same descriptor and outputs as svt_hip_rd_batch (host pointers).  Pins oracle/rd_oracle.c
 * as a whole chain and serves as bench.py's "reference" CPU baseline.  "b" quantizer only (what the bench runs).
 * ===================================================================================================================== */
#include "../include/svt_hip_dsp.h"

typedef void (*FwdFn)(int16_t *, int32_t *, uint32_t, TxType, uint8_t);
typedef void (*InvFn)(const int32_t *, uint16_t *, int32_t, uint16_t *, int32_t, TxType, int32_t);

static FwdFn g_fwd[19];
static InvFn g_inv_sq[5]; /* 4x4 .. 64x64 take the 7-argument form; the rectangular ones add tx_size and eob */

void ref_set_simd_rd(int use_simd) {
#define FWD_C(i, n) g_fwd[i] = n##_c
    g_fwd[0] = svt_av1_transform_two_d_4x4_c; g_fwd[1] = svt_av1_transform_two_d_8x8_c; g_fwd[2] = svt_av1_transform_two_d_16x16_c;
    g_fwd[3] = svt_av1_transform_two_d_32x32_c; g_fwd[4] = svt_av1_transform_two_d_64x64_c;
    FWD_C(5, svt_av1_fwd_txfm2d_4x8); FWD_C(6, svt_av1_fwd_txfm2d_8x4); FWD_C(7, svt_av1_fwd_txfm2d_8x16); FWD_C(8, svt_av1_fwd_txfm2d_16x8);
    FWD_C(9, svt_av1_fwd_txfm2d_16x32); FWD_C(10, svt_av1_fwd_txfm2d_32x16); FWD_C(11, svt_av1_fwd_txfm2d_32x64); FWD_C(12, svt_av1_fwd_txfm2d_64x32);
    FWD_C(13, svt_av1_fwd_txfm2d_4x16); FWD_C(14, svt_av1_fwd_txfm2d_16x4); FWD_C(15, svt_av1_fwd_txfm2d_8x32); FWD_C(16, svt_av1_fwd_txfm2d_32x8);
    FWD_C(17, svt_av1_fwd_txfm2d_16x64); FWD_C(18, svt_av1_fwd_txfm2d_64x16);
#undef FWD_C
    g_inv_sq[0] = svt_av1_inv_txfm2d_add_4x4_c; g_inv_sq[1] = svt_av1_inv_txfm2d_add_8x8_c; g_inv_sq[2] = svt_av1_inv_txfm2d_add_16x16_c;
    g_inv_sq[3] = svt_av1_inv_txfm2d_add_32x32_c; g_inv_sq[4] = svt_av1_inv_txfm2d_add_64x64_c;
    svt_residual_kernel8bit           = svt_residual_kernel8bit_c;
    svt_residual_kernel16bit          = svt_residual_kernel16bit_c;
    svt_aom_satd                      = svt_aom_satd_c;
    svt_aom_highbd_quantize_b         = svt_aom_highbd_quantize_b_c;
    svt_aom_quantize_b                = svt_aom_quantize_b_c_ii;
    svt_full_distortion_kernel32_bits = svt_full_distortion_kernel32_bits_c;
    svt_full_distortion_kernel16_bits = svt_full_distortion_kernel16_bits_c;
    svt_spatial_full_distortion_kernel = svt_spatial_full_distortion_kernel_c;
    svt_handle_transform64x64 = svt_handle_transform64x64_c; svt_handle_transform64x32 = svt_handle_transform64x32_c;
    svt_handle_transform32x64 = svt_handle_transform32x64_c; svt_handle_transform64x16 = svt_handle_transform64x16_c;
    svt_handle_transform16x64 = svt_handle_transform16x64_c;
#ifdef REF_WITH_AVX2
    if (use_simd) { /* the picks of svt_aom_setup_common_rtcd_internal / svt_aom_setup_rtcd_internal with AVX2 available, minus .asm */
        g_fwd[0] = svt_av1_fwd_txfm2d_4x4_sse4_1; g_fwd[1] = svt_av1_fwd_txfm2d_8x8_avx2; g_fwd[2] = svt_av1_fwd_txfm2d_16x16_avx2;
        g_fwd[3] = svt_av1_fwd_txfm2d_32x32_avx2; g_fwd[4] = svt_av1_fwd_txfm2d_64x64_avx2;
        g_inv_sq[0] = svt_av1_inv_txfm2d_add_4x4_sse4_1; g_inv_sq[1] = svt_av1_inv_txfm2d_add_8x8_sse4_1; g_inv_sq[2] = svt_av1_inv_txfm2d_add_16x16_sse4_1;
        g_inv_sq[3] = svt_av1_inv_txfm2d_add_32x32_sse4_1; g_inv_sq[4] = svt_av1_inv_txfm2d_add_64x64_sse4_1;
        svt_residual_kernel16bit          = svt_residual_kernel16bit_avx2;
        svt_aom_highbd_quantize_b         = svt_aom_highbd_quantize_b_avx2;
        svt_full_distortion_kernel32_bits = svt_full_distortion_kernel32_bits_avx2;
        svt_full_distortion_kernel16_bits = svt_full_distortion_kernel16_bits_avx2;
        svt_handle_transform64x64         = svt_handle_transform64x64_avx2;
    }
#else
    (void)use_simd;
#endif
}

/* working buffers of one thread of the chain: the largest transform block (64 x 64) */
typedef struct RdScratch {
    int16_t  *res;
    int32_t  *co, *q, *dq;
    uint16_t *p16, *r16;
} RdScratch;
static int rd_scratch_alloc(RdScratch *s) {
    const size_t n = 64 * 64;
    s->res = aligned_alloc(64, sizeof(int16_t) * n);
    s->co = aligned_alloc(64, sizeof(int32_t) * n); s->q = aligned_alloc(64, sizeof(int32_t) * n); s->dq = aligned_alloc(64, sizeof(int32_t) * n);
    s->p16 = aligned_alloc(64, sizeof(uint16_t) * n); s->r16 = aligned_alloc(64, sizeof(uint16_t) * n);
    return !(s->res && s->co && s->q && s->dq && s->p16 && s->r16);
}
static void rd_scratch_free(RdScratch *s) { free(s->res); free(s->co); free(s->q); free(s->dq); free(s->p16); free(s->r16); }

static int rd_batch_core(const SvtHipRdBatchDesc *d, const RdScratch *sc) {
    const int ts = d->tx_size;
    if (ts < 0 || ts > 18 || d->quant_kind > 1 || d->qmatrix) return 2; /* every size, "b" and "fp" quantizers, flat matrices, full transforms */
    const int W = tx_size_wide[ts], H = tx_size_high[ts], WP = W > 32 ? 32 : W, HP = H > 32 ? 32 : H, NP = WP * HP;
    const int bd = d->bit_depth, hbd = bd != 8, log_scale = av1_get_tx_scale_tab[ts];
    int16_t  *res = sc->res;
    int32_t  *co = sc->co, *q = sc->q, *dq = sc->dq;
    uint16_t *p16 = sc->p16, *r16 = sc->r16;
    for (uint32_t j = 0; j < d->n_jobs; j++) {
        const SvtHipTxJob    *jb = &d->jobs[j];
        const SvtHipQuantRow *qr = &d->quant_rows[jb->quant_row];
        const TxType          tt = (TxType)(jb->tx_type & 15);
        /* MacroblockPlane rows are int16[8], 16-byte aligned: [0] = DC, [1..7] = AC (full_loop.c:1627-1685) */
        DECLARE_ALIGNED(16, int16_t, zbin[8]); DECLARE_ALIGNED(16, int16_t, rnd[8]); DECLARE_ALIGNED(16, int16_t, qnt[8]);
        DECLARE_ALIGNED(16, int16_t, qsh[8]); DECLARE_ALIGNED(16, int16_t, deq[8]);
        for (int k = 0; k < 8; k++) { zbin[k] = qr->zbin[k != 0]; rnd[k] = qr->round[k != 0]; qnt[k] = qr->quant[k != 0]; qsh[k] = qr->quant_shift[k != 0]; deq[k] = qr->dequant[k != 0]; }
        if (hbd) svt_residual_kernel16bit((uint16_t *)d->src + jb->src_offset, d->src_stride, (uint16_t *)d->pred + jb->pred_offset, d->pred_stride, res, W, W, H);
        else svt_residual_kernel8bit((uint8_t *)d->src + jb->src_offset, d->src_stride, (uint8_t *)d->pred + jb->pred_offset, d->pred_stride, res, W, W, H);
        if (jb->pf_shape) return 2;
        g_fwd[ts](res, co, W, tt, (uint8_t)bd);
        /* 64-point dimensions: the energy outside the top-left 32 x 32 and the packing of what is kept (transforms.c:2374-2505) */
        d->three_quad_energy[j] = ts == 4 ? svt_handle_transform64x64(co) : ts == 11 ? svt_handle_transform32x64(co) : ts == 12 ? svt_handle_transform64x32(co)
            : ts == 17 ? svt_handle_transform16x64(co) : ts == 18 ? svt_handle_transform64x16(co) : 0;
        d->satd[j]              = (uint32_t)svt_aom_satd(co, NP);
        const ScanOrder *so = &av1_scan_orders[ts][tt];
        if (d->quant_kind == 0) {
            if (hbd) svt_aom_highbd_quantize_b(co, NP, zbin, rnd, qnt, qsh, q, dq, deq, &d->eob[j], so->scan, so->iscan, NULL, NULL, log_scale);
            else svt_aom_quantize_b(co, NP, zbin, rnd, qnt, qsh, q, dq, deq, &d->eob[j], so->scan, so->iscan, NULL, NULL, log_scale);
        } else { /* svt_av1_quantize_fp_facade / svt_av1_highbd_quantize_fp_facade (full_loop.c:344-516): the fp rows, one function per log scale */
            DECLARE_ALIGNED(16, int16_t, rfp[8]); DECLARE_ALIGNED(16, int16_t, qfp[8]);
            for (int k = 0; k < 8; k++) { rfp[k] = qr->round_fp[k != 0]; qfp[k] = qr->quant_fp[k != 0]; }
            if (hbd) svt_av1_highbd_quantize_fp_c(co, NP, zbin, rfp, qfp, qsh, q, dq, deq, &d->eob[j], so->scan, so->iscan, (int16_t)log_scale);
            else if (log_scale == 0) svt_av1_quantize_fp_c(co, NP, zbin, rfp, qfp, qsh, q, dq, deq, &d->eob[j], so->scan, so->iscan);
            else if (log_scale == 1) svt_av1_quantize_fp_32x32_c(co, NP, zbin, rfp, qfp, qsh, q, dq, deq, &d->eob[j], so->scan, so->iscan);
            else svt_av1_quantize_fp_64x64_c(co, NP, zbin, rfp, qfp, qsh, q, dq, deq, &d->eob[j], so->scan, so->iscan);
        }
        if (d->cul_level) d->cul_level[j] = svt_av1_compute_cul_level_c(so->scan, q, &d->eob[j]); /* the wrapper's return value, full_loop.c:1836 */
        uint64_t dist[DIST_CALC_TOTAL];
        svt_full_distortion_kernel32_bits(co, WP, dq, WP, dist, WP, HP);
        d->dist_coeff[2 * (size_t)j] = dist[DIST_CALC_RESIDUAL]; d->dist_coeff[2 * (size_t)j + 1] = dist[DIST_CALC_PREDICTION];
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++)
                p16[r * W + c] = hbd ? ((const uint16_t *)d->pred)[jb->pred_offset + (size_t)r * d->pred_stride + c]
                                     : ((const uint8_t *)d->pred)[jb->pred_offset + (size_t)r * d->pred_stride + c];
        if (ts <= 4) g_inv_sq[ts](dq, p16, W, r16, W, tt, bd);
        else {
            typedef void (*InvRect4)(const int32_t *, uint16_t *, int32_t, uint16_t *, int32_t, TxType, TxSize, int32_t);
            typedef void (*InvRect)(const int32_t *, uint16_t *, int32_t, uint16_t *, int32_t, TxType, TxSize, int32_t, int32_t);
            static const InvRect4 k4[19] = {[5] = svt_av1_inv_txfm2d_add_4x8_c, [6] = svt_av1_inv_txfm2d_add_8x4_c, [13] = svt_av1_inv_txfm2d_add_4x16_c,
                                            [14] = svt_av1_inv_txfm2d_add_16x4_c};
            static const InvRect  kr[19] = {[7] = svt_av1_inv_txfm2d_add_8x16_c, [8] = svt_av1_inv_txfm2d_add_16x8_c, [9] = svt_av1_inv_txfm2d_add_16x32_c,
                                            [10] = svt_av1_inv_txfm2d_add_32x16_c, [11] = svt_av1_inv_txfm2d_add_32x64_c, [12] = svt_av1_inv_txfm2d_add_64x32_c,
                                            [15] = svt_av1_inv_txfm2d_add_8x32_c, [16] = svt_av1_inv_txfm2d_add_32x8_c, [17] = svt_av1_inv_txfm2d_add_16x64_c,
                                            [18] = svt_av1_inv_txfm2d_add_64x16_c};
            if (k4[ts]) k4[ts](dq, p16, W, r16, W, tt, (TxSize)ts, bd);
            else kr[ts](dq, p16, W, r16, W, tt, (TxSize)ts, W * H, bd);
        }
        uint64_t sse;
        if (hbd) sse = svt_full_distortion_kernel16_bits((uint8_t *)((uint16_t *)d->src + jb->src_offset), 0, d->src_stride, (uint8_t *)r16, 0, W, W, H);
        else {
            sse = 0;
            for (int r = 0; r < H; r++)
                for (int c = 0; c < W; c++) { const int64_t e = (int64_t)((const uint8_t *)d->src)[jb->src_offset + (size_t)r * d->src_stride + c] - r16[r * W + c]; sse += (uint64_t)(e * e); }
        }
        d->sse[j] = sse;
        if (d->recon)
            for (int r = 0; r < H; r++)
                for (int c = 0; c < W; c++) {
                    if (hbd) ((uint16_t *)d->recon)[jb->pred_offset + (size_t)r * d->pred_stride + c] = r16[r * W + c];
                    else ((uint8_t *)d->recon)[jb->pred_offset + (size_t)r * d->pred_stride + c] = (uint8_t)r16[r * W + c];
                }
        if (d->coeff) memcpy(d->coeff + (size_t)j * NP, co, sizeof(int32_t) * NP);
        if (d->qcoeff) memcpy(d->qcoeff + (size_t)j * NP, q, sizeof(int32_t) * NP);
        if (d->dqcoeff) memcpy(d->dqcoeff + (size_t)j * NP, dq, sizeof(int32_t) * NP);
    }
    return 0;
}

int ref_rd_batch(const SvtHipRdBatchDesc *d) {
    if (!g_fwd[0]) ref_set_simd_rd(0);
    RdScratch sc;
    if (rd_scratch_alloc(&sc)) { rd_scratch_free(&sc); return 1; }
    const int rc = rd_batch_core(d, &sc);
    rd_scratch_free(&sc);
    return rc;
}

/* =====================================================================================================================
 * The kernel chain of the TPL dispenser's inter / recon evaluation (Codec/src_ops_process.c:857-872 and get_quantize_error
 * :225-249) through the reference's own functions: svt_aom_subtract_block on every (1 << sub)-th row -> svt_av1_wht_fwd_txfm
 * (DCT_DCT, partial-frequency shape) -> svt_aom_satd -> svt_av1_quantize_fp with the DCT_DCT scan -> svt_av1_block_error.
 * 16-pixel blocks (dispenser_search_level 0): tx_size TX_16X16 / TX_16X8 / TX_16X4 for subsample_tx 0 / 1 / 2 (:380-382,531).
 * coeff is deliberately left dirty between blocks, as the dispenser's stack buffer is.
 * out8 = {inter_cost (satd << sub), eob, recon_error, sse}
 * ===================================================================================================================== */
void svt_av1_wht_fwd_txfm(int16_t *src_diff, int bw, int32_t *coeff, TxSize tx_size, EB_TRANS_COEFF_SHAPE pf_shape, int bit_depth, int is_hbd);

int ref_tpl_chain(const uint8_t *src, uint32_t src_stride, const uint8_t *pred, uint32_t pred_stride, int level, int sub, int pf_shape,
                  const SvtHipQuantRow *qr, int32_t *coeff, int32_t *qcoeff, int32_t *dqcoeff, int64_t *out4) {
    /* tx_size_array / sub2_tx_size_array / sub4_tx_size_array of dispenser levels 0 and 1 (src_ops_process.c:377-382; level 2 is never
     * selected -- initial_rc_process.c:308-368 -- and its 64x64 transform would not fit the dispenser's MAX_TPL_SIZE = 32 buffers) */
    static const TxSize sizes[2][3] = {{TX_16X16, TX_16X8, TX_16X4}, {TX_32X32, TX_32X16, TX_32X8}};
    if (level < 0 || level > 1 || sub < 0 || sub > 2) return 1;
    const TxSize tx_size = sizes[level][sub];
    const int    size    = 16 << level;
    svt_aom_subtract_block = svt_aom_subtract_block_c;
    svt_aom_satd           = svt_aom_satd_c;
    svt_av1_quantize_fp    = svt_av1_quantize_fp_c;
    svt_av1_block_error    = svt_av1_block_error_c;
    svt_av1_fwd_txfm2d_16x16 = svt_av1_transform_two_d_16x16_c; svt_av1_fwd_txfm2d_16x8 = svt_av1_fwd_txfm2d_16x8_c; svt_av1_fwd_txfm2d_16x4 = svt_av1_fwd_txfm2d_16x4_c;
    svt_av1_fwd_txfm2d_16x16_N2 = svt_aom_transform_two_d_16x16_N2_c; svt_av1_fwd_txfm2d_16x8_N2 = svt_av1_fwd_txfm2d_16x8_N2_c; svt_av1_fwd_txfm2d_16x4_N2 = svt_av1_fwd_txfm2d_16x4_N2_c;
    svt_av1_fwd_txfm2d_16x16_N4 = svt_aom_transform_two_d_16x16_N4_c; svt_av1_fwd_txfm2d_16x8_N4 = svt_av1_fwd_txfm2d_16x8_N4_c; svt_av1_fwd_txfm2d_16x4_N4 = svt_av1_fwd_txfm2d_16x4_N4_c;
    svt_av1_fwd_txfm2d_32x32 = svt_av1_transform_two_d_32x32_c; svt_av1_fwd_txfm2d_32x16 = svt_av1_fwd_txfm2d_32x16_c; svt_av1_fwd_txfm2d_32x8 = svt_av1_fwd_txfm2d_32x8_c;
    svt_av1_fwd_txfm2d_32x32_N2 = svt_aom_transform_two_d_32x32_N2_c; svt_av1_fwd_txfm2d_32x16_N2 = svt_av1_fwd_txfm2d_32x16_N2_c; svt_av1_fwd_txfm2d_32x8_N2 = svt_av1_fwd_txfm2d_32x8_N2_c;
    svt_av1_fwd_txfm2d_32x32_N4 = svt_aom_transform_two_d_32x32_N4_c; svt_av1_fwd_txfm2d_32x16_N4 = svt_av1_fwd_txfm2d_32x16_N4_c; svt_av1_fwd_txfm2d_32x8_N4 = svt_av1_fwd_txfm2d_32x8_N4_c;
    DECLARE_ALIGNED(16, int16_t, src_diff[32 * 32]);
    DECLARE_ALIGNED(16, int16_t, zbin[8]); DECLARE_ALIGNED(16, int16_t, rnd[8]); DECLARE_ALIGNED(16, int16_t, qnt[8]);
    DECLARE_ALIGNED(16, int16_t, qsh[8]); DECLARE_ALIGNED(16, int16_t, deq[8]);
    for (int k = 0; k < 8; k++) { zbin[k] = qr->zbin[k != 0]; rnd[k] = qr->round_fp[k != 0]; qnt[k] = qr->quant_fp[k != 0]; qsh[k] = qr->quant_shift[k != 0]; deq[k] = qr->dequant[k != 0]; }
    svt_aom_subtract_block(size >> sub, size, src_diff, size << sub, src, (ptrdiff_t)src_stride << sub, pred, (ptrdiff_t)pred_stride << sub);
    svt_av1_wht_fwd_txfm(src_diff, size << sub, coeff, tx_size, (EB_TRANS_COEFF_SHAPE)pf_shape, 8, 0);
    out4[0] = (int64_t)svt_aom_satd(coeff, (size * size) >> sub) << sub;
    /* get_quantize_error */
    const ScanOrder *const so      = &av1_scan_orders[tx_size][DCT_DCT];
    const int              pix_num = 1 << num_pels_log2_lookup[txsize_to_bsize[tx_size]];
    const int              shift   = tx_size == TX_32X32 ? 0 : 2;
    uint16_t               eob     = 0;
    svt_av1_quantize_fp(coeff, pix_num, zbin, rnd, qnt, qsh, qcoeff, dqcoeff, deq, &eob, so->scan, so->iscan);
    int64_t sse, err = svt_av1_block_error(coeff, dqcoeff, pix_num, &sse) >> shift;
    out4[1] = eob;
    out4[2] = err > 1 ? err : 1;
    sse >>= shift;
    out4[3] = sse > 1 ? sse : 1;
    return 0;
}

/* The luma pyramid exactly as the reference's picture-analysis stage makes it: svt_aom_downsample_filtering_input_picture
 * (Codec/pic_analysis_process.c:2139-2196) = downsample_2d (`_c`, :130-158) + svt_aom_generate_padding (Codec/pic_operators.c:397-443),
 * driven with the reference's own PictureParentControlSet flags and EbPictureBufferDesc geometry.  `full` is read, the padded
 * `quarter` / `sixteenth` buffers (padding 32 / 16, Globals/enc_handle.c:1260-1279) are written.  level1 == 0 takes the
 * reference's other branch: the sixteenth plane straight from the full plane with decimation step 4. */
#include "pic_analysis_process.h"
int ref_pyramid(const SvtHipPlaneDesc *full, const SvtHipPlaneDesc *quarter, const SvtHipPlaneDesc *sixteenth, int level1) {
    PictureParentControlSet *pcs = (PictureParentControlSet *)calloc(1, sizeof(*pcs));
    EbPictureBufferDesc      in, q, s;
    if (!pcs) return 1;
    memset(&in, 0, sizeof(in)); memset(&q, 0, sizeof(q)); memset(&s, 0, sizeof(s));
    pcs->enable_hme_flag        = 1;
    pcs->enable_hme_level0_flag = 1;
    pcs->enable_hme_level1_flag = level1 ? 1 : 0;
    downsample_2d = svt_aom_downsample_2d_c;
    svt_memcpy    = svt_memcpy_c; /* rtcd pointer read by svt_aom_generate_padding */
    in.buffer_y = (uint8_t *)full->buffer_y; in.stride_y = (uint16_t)full->stride_y; in.org_x = full->org_x; in.org_y = full->org_y;
    in.width = full->width; in.height = full->height;
    q.buffer_y = (uint8_t *)quarter->buffer_y; q.stride_y = (uint16_t)quarter->stride_y; q.org_x = quarter->org_x; q.org_y = quarter->org_y;
    q.width = quarter->width; q.height = quarter->height;
    s.buffer_y = (uint8_t *)sixteenth->buffer_y; s.stride_y = (uint16_t)sixteenth->stride_y; s.org_x = sixteenth->org_x; s.org_y = sixteenth->org_y;
    s.width = sixteenth->width; s.height = sixteenth->height;
    svt_aom_downsample_filtering_input_picture(pcs, &in, &q, &s);
    free(pcs);
    return 0;
}

/* get_hvs_modulation_factor is not declared in a header every caller includes */
double get_hvs_modulation_factor(double psy_rd, bool is_islice, uint8_t temporal_layer_index);
double ref_hvs_modulation_factor(double psy_rd, int is_islice, uint8_t temporal_layer_index) { return get_hvs_modulation_factor(psy_rd, is_islice != 0, temporal_layer_index); }

/* Address of one of the reference's rtcd function-pointer variables (the globals svt_aom_setup_rtcd_internal fills): lets a test install
 * another backend's kernels into the reference exactly where its own SIMD kernels go.  NULL: not a pointer this harness knows. */
void **ref_rtcd_slot(const char *name) {
#define SLOT(n) if (!strcmp(name, #n)) return (void **)&n;
    SLOT(svt_sad_loop_kernel) SLOT(svt_nxm_sad_kernel) SLOT(svt_ext_all_sad_calculation_8x8_16x16) SLOT(svt_ext_eight_sad_calculation_32x32_64x64)
    SLOT(svt_ext_sad_calculation_8x8_16x16) SLOT(svt_ext_sad_calculation_32x32_64x64) SLOT(svt_initialize_buffer_32bits)
#undef SLOT
    return NULL;
}

/* =====================================================================================================================
 * bench.py's CPU baseline, natively threaded: what the reference's ME threads and mode-decision threads do for the bench's
 * work -- the b64 loop of svt_aom_motion_estimation_kernel (Codec/me_process.c:174-290) around svt_aom_motion_estimation_b64
 * with the kernels ref_set_simd() installed, followed by the RD chain (rd_batch_core above, the kernels ref_set_simd_rd()
 * installed) over the same b64 row at the three transform depths.  One work item = one b64 row of one picture; n_threads
 * pthreads pull items from a shared counter (cyclically over the sample) for `seconds` (seconds <= 0: every item exactly once).  Everything a thread needs -- its
 * MeContext, pcs, MeSbResults of one row, transform scratch, job lists, output arrays -- is created ONCE before the clock starts;
 * the timed loop allocates nothing and shares only read-only planes (as the encoder's threads do).
 * ===================================================================================================================== */
#include <pthread.h>
#include <stdatomic.h>
#include <time.h>

typedef struct RefBenchPicture {
    const SvtHipMeConfig      *cfg;
    const SvtHipMePictureDesc *desc;
    const SvtHipPlaneDesc     *cur_planes;                /* [3] */
    const SvtHipPlaneDesc     *ref_planes;                /* [SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][3] */
    const uint16_t            *src10, *pred10;            /* W x H 10-bit planes, stride = width */
} RefBenchPicture;

typedef struct RefBenchDesc {
    uint32_t               n_pictures;
    const RefBenchPicture *pictures;
    uint32_t               width, height;                 /* luma size; the 10-bit planes have stride == width */
    uint32_t               row_start, n_rows;             /* the sample: b64 rows [row_start, row_start + n_rows) of every picture */
    const SvtHipQuantRow  *quant_row;
    uint32_t               n_tx_sizes;
    const int32_t         *tx_sizes;                      /* RD depths, e.g. {TX_64X64, TX_32X32, TX_16X16} */
    int32_t                n_threads;
    double                 seconds;
    /* out */
    uint64_t               items_done;
    double                 elapsed;
    uint64_t               checksum;                      /* keeps the work observable (sum of 64x64 distortions and eobs) */
} RefBenchDesc;

typedef struct BenchThread {
    RefBenchDesc *d;
    atomic_ullong *next;
    volatile int  *stop;
    /* ME state */
    SequenceControlSet *scs; PictureParentControlSet *pcs; MeContext *m; MotionEstimationData *med;
    MeSbResults *row_res;
    uint32_t    *dist[6];
    uint8_t     *flags[2];
    EbPictureBufferDesc cur[3], refs[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][3];
    int          cur_pic;
    /* RD state */
    RdScratch    sc;
    SvtHipTxJob *jobs;
    uint16_t    *eob; uint32_t *satd; uint64_t *dc, *sse, *tqe;
    uint64_t     done, checksum;
    int          failed;
} BenchThread;

static void fill_desc(EbPictureBufferDesc *d, const SvtHipPlaneDesc *p) {
    memset(d, 0, sizeof(*d));
    d->buffer_y = (EbByte)p->buffer_y; d->stride_y = p->stride_y; d->org_x = p->org_x; d->org_y = p->org_y;
    d->width = p->width; d->height = p->height; d->max_width = p->width; d->max_height = p->height;
}

static int bench_thread_init(BenchThread *t) {
    const RefBenchDesc *d = t->d;
    const SvtHipMePictureDesc *d0 = d->pictures[0].desc;
    const uint32_t w64 = (d0->aligned_width + 63) / 64, h64 = (d0->aligned_height + 63) / 64, nb = w64 * h64;
    const uint32_t n_pu = svt_hip_me_n_pu(d0->enable_me_16x16, d0->enable_me_8x8);
    t->scs = calloc(1, sizeof(*t->scs)); t->pcs = calloc(1, sizeof(*t->pcs)); t->m = calloc(1, sizeof(*t->m)); t->med = calloc(1, sizeof(*t->med));
    t->row_res = calloc(w64, sizeof(MeSbResults));
    t->med->me_results = calloc(nb, sizeof(MeSbResults *));
    t->pcs->b64_geom   = calloc(nb, sizeof(B64Geom));
    if (!t->scs || !t->pcs || !t->m || !t->med || !t->row_res || !t->med->me_results || !t->pcs->b64_geom) return 1;
    for (int k = 0; k < 6; k++) if (!(t->dist[k] = calloc(nb, sizeof(uint32_t)))) return 1;
    for (int k = 0; k < 2; k++) if (!(t->flags[k] = calloc(nb, 1))) return 1;
    for (uint32_t x = 0; x < w64; x++) {
        MeSbResults *r = &t->row_res[x];
        r->total_me_candidate_index = calloc(n_pu, 1);
        r->me_mv_array              = calloc((size_t)n_pu * SVT_HIP_MAX_REFS * SVT_HIP_MAX_LISTS, sizeof(MvCandidate));
        r->me_candidate_array       = calloc((size_t)n_pu * 32, sizeof(MeCandidate));
        if (!r->total_me_candidate_index || !r->me_mv_array || !r->me_candidate_array) return 1;
    }
    for (uint32_t b = 0; b < nb; b++) { /* every row of the picture lands in the thread's one row of result storage */
        t->med->me_results[b] = &t->row_res[b % w64];
        B64Geom *g = &t->pcs->b64_geom[b];
        g->org_x = (uint16_t)((b % w64) * 64); g->org_y = (uint16_t)((b / w64) * 64);
        g->width  = (uint8_t)((d0->aligned_width - g->org_x) < 64 ? d0->aligned_width - g->org_x : 64);
        g->height = (uint8_t)((d0->aligned_height - g->org_y) < 64 ? d0->aligned_height - g->org_y : 64);
    }
    t->scs->b64_size = 64;
    t->pcs->scs = t->scs; t->pcs->pa_me_data = t->med;
    t->pcs->me_64x64_distortion = t->dist[0]; t->pcs->me_32x32_distortion = t->dist[1]; t->pcs->me_16x16_distortion = t->dist[2];
    t->pcs->me_8x8_distortion = t->dist[3]; t->pcs->rc_me_distortion = t->dist[4]; t->pcs->me_8x8_cost_variance = t->dist[5];
    t->pcs->stationary_block_present_sb = t->flags[0]; t->pcs->rc_me_allow_gm = t->flags[1];
    t->cur_pic = -1;
    if (rd_scratch_alloc(&t->sc)) return 1;
    const size_t max_jobs = (size_t)((d->width + 3) / 4) * 16; /* 4x4 blocks of one b64 row: more than any depth needs */
    t->jobs = calloc(max_jobs, sizeof(SvtHipTxJob));
    t->eob = calloc(max_jobs, sizeof(uint16_t)); t->satd = calloc(max_jobs, sizeof(uint32_t)); t->dc = calloc(max_jobs * 2, sizeof(uint64_t));
    t->sse = calloc(max_jobs, sizeof(uint64_t)); t->tqe = calloc(max_jobs, sizeof(uint64_t));
    return !(t->jobs && t->eob && t->satd && t->dc && t->sse && t->tqe);
}

static void bench_thread_free(BenchThread *t) {
    if (t->row_res && t->d) {
        const uint32_t w64 = (t->d->pictures[0].desc->aligned_width + 63) / 64;
        for (uint32_t x = 0; x < w64; x++) { free(t->row_res[x].total_me_candidate_index); free(t->row_res[x].me_mv_array); free(t->row_res[x].me_candidate_array); }
    }
    if (t->med) free(t->med->me_results);
    if (t->pcs) free(t->pcs->b64_geom);
    for (int k = 0; k < 6; k++) free(t->dist[k]);
    for (int k = 0; k < 2; k++) free(t->flags[k]);
    free(t->row_res); free(t->med); free(t->m); free(t->pcs); free(t->scs);
    rd_scratch_free(&t->sc);
    free(t->jobs); free(t->eob); free(t->satd); free(t->dc); free(t->sse); free(t->tqe);
}

/* what the ME kernel does when it takes a new picture (me_process.c:140-172: signal derivation and picture pointers) */
static void bench_set_picture(BenchThread *t, int pi) {
    const RefBenchPicture     *p    = &t->d->pictures[pi];
    const SvtHipMePictureDesc *desc = p->desc;
    SequenceControlSet *scs = t->scs; PictureParentControlSet *pcs = t->pcs; MeContext *m = t->m;
    scs->input_resolution = (EbInputResolution)desc->input_resolution;
    scs->mrp_ctrls.only_l_bwd = desc->only_l_bwd;
    pcs->picture_number = desc->picture_number; pcs->aligned_width = desc->aligned_width; pcs->aligned_height = desc->aligned_height;
    pcs->hierarchical_levels = desc->hierarchical_levels; pcs->temporal_layer_index = desc->temporal_layer_index;
    pcs->similar_brightness_refs = desc->similar_brightness_refs; pcs->enable_me_8x8 = desc->enable_me_8x8; pcs->enable_me_16x16 = desc->enable_me_16x16;
    pcs->max_number_of_pus_per_sb = desc->max_number_of_pus_per_sb;
    pcs->gm_ctrls.enabled = desc->gm_enabled; pcs->gm_ctrls.use_distance_based_active_th = desc->gm_use_distance_based_active_th;
    t->med->max_cand = desc->max_cand; t->med->max_refs = desc->max_refs; t->med->max_l0 = desc->max_l0;
    ctx_from_cfg(p->cfg, m);
    m->me_type = p->cfg->me_type == 1 ? ME_MCTF : ME_OPEN_LOOP;
    m->tf_me_exit_th = (uint16_t)desc->tf_me_exit_th;
    m->num_of_list_to_search = desc->num_of_list_to_search;
    m->num_of_ref_pic_to_search[0] = desc->num_of_ref_pic_to_search[0]; m->num_of_ref_pic_to_search[1] = desc->num_of_ref_pic_to_search[1];
    m->temporal_layer_index = desc->temporal_layer_index; m->is_ref = desc->is_ref;
    for (int l = 0; l < 3; l++) fill_desc(&t->cur[l], &p->cur_planes[l]);
    for (int li = 0; li < desc->num_of_list_to_search; li++)
        for (int ri = 0; ri < desc->num_of_ref_pic_to_search[li]; ri++) {
            for (int l = 0; l < 3; l++) fill_desc(&t->refs[li][ri][l], &p->ref_planes[(li * SVT_HIP_MAX_REFS + ri) * 3 + l]);
            m->me_ds_ref_array[li][ri].sixteenth_picture_ptr = &t->refs[li][ri][0];
            m->me_ds_ref_array[li][ri].quarter_picture_ptr   = &t->refs[li][ri][1];
            m->me_ds_ref_array[li][ri].picture_ptr           = &t->refs[li][ri][2];
            m->me_ds_ref_array[li][ri].picture_number        = desc->ref_picture_number[li][ri];
        }
    t->cur_pic = pi;
}

static void bench_item(BenchThread *t, int pi, uint32_t by) {
    const RefBenchDesc *d = t->d;
    if (t->cur_pic != pi) bench_set_picture(t, pi);
    const RefBenchPicture *p = &d->pictures[pi];
    MeContext *m = t->m;
    const uint32_t w64 = (p->desc->aligned_width + 63) / 64;
    EbPictureBufferDesc *cur = t->cur;
    for (uint32_t bx = 0; bx < w64; bx++) { /* me_process.c:183-214 */
        const uint32_t b = bx + by * w64, ox = bx * 64, oy = by * 64;
        m->b64_src_ptr    = &cur[2].buffer_y[(cur[2].org_y + oy) * cur[2].stride_y + cur[2].org_x + ox];
        m->b64_src_stride = cur[2].stride_y;
        m->quarter_b64_buffer = &cur[1].buffer_y[(cur[1].org_y + (oy >> 1)) * cur[1].stride_y + cur[1].org_x + (ox >> 1)];
        m->quarter_b64_buffer_stride = cur[1].stride_y;
        m->sixteenth_b64_buffer = &cur[0].buffer_y[(cur[0].org_y + (oy >> 2)) * cur[0].stride_y + cur[0].org_x + (ox >> 2)];
        m->sixteenth_b64_buffer_stride = cur[0].stride_y;
        svt_aom_motion_estimation_b64(t->pcs, b, ox, oy, m, &cur[2]);
        t->checksum += t->dist[0][b];
    }
    /* the RD chain over the row's transform blocks, depth by depth */
    const uint32_t y_lo = by * 64, y_hi = (y_lo + 64) < d->height ? y_lo + 64 : d->height;
    for (uint32_t k = 0; k < d->n_tx_sizes; k++) {
        const int ts = d->tx_sizes[k], bw = tx_size_wide[ts], bh = tx_size_high[ts];
        uint32_t  n = 0;
        for (uint32_t y = y_lo; y + bh <= y_hi; y += bh)
            for (uint32_t x = 0; x + bw <= d->width; x += bw) {
                SvtHipTxJob *j = &t->jobs[n++];
                memset(j, 0, sizeof(*j));
                j->src_offset = j->pred_offset = y * d->width + x;
            }
        SvtHipRdBatchDesc rd;
        memset(&rd, 0, sizeof(rd));
        rd.bit_depth = 10; rd.quant_kind = 0; rd.tx_size = (uint8_t)ts; rd.n_jobs = n; rd.src_stride = rd.pred_stride = d->width;
        rd.src = p->src10; rd.pred = p->pred10; rd.jobs = t->jobs; rd.quant_rows = d->quant_row; rd.n_quant_rows = 1;
        rd.eob = t->eob; rd.satd = t->satd; rd.dist_coeff = t->dc; rd.sse = t->sse; rd.three_quad_energy = t->tqe;
        if (rd_batch_core(&rd, &t->sc)) { t->failed = 1; return; }
        for (uint32_t j = 0; j < n; j++) t->checksum += t->eob[j];
    }
}

static void *bench_thread_main(void *arg) {
    BenchThread        *t = arg;
    const RefBenchDesc *d = t->d;
    const uint64_t      n_items = (uint64_t)d->n_pictures * d->n_rows;
    const int           once = d->seconds <= 0; /* every item exactly once: a deterministic checksum for the tests */
    while (!*t->stop && !t->failed) {
        uint64_t i = atomic_fetch_add(t->next, 1);
        if (once && i >= n_items) break;
        i %= n_items;
        bench_item(t, (int)(i / d->n_rows), d->row_start + (uint32_t)(i % d->n_rows));
        t->done++;
    }
    return NULL;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* Runs the sample on d->n_threads threads for about d->seconds; fills items_done / elapsed / checksum.  The kernels are the ones
 * ref_set_simd / ref_set_simd_rd installed last.  0 on success. */
int ref_bench_rows(RefBenchDesc *d) {
    if (!d || !d->n_pictures || !d->n_rows || d->n_threads < 1 || d->n_tx_sizes > 8) return 2;
    ref_set_simd(g_simd);
    if (!g_fwd[0]) ref_set_simd_rd(0);
    const int    nt = d->n_threads;
    BenchThread *th = calloc(nt, sizeof(*th));
    pthread_t   *id = calloc(nt, sizeof(*id));
    atomic_ullong next = 0;
    volatile int  stop = 0;
    int rc = (!th || !id);
    for (int i = 0; i < nt && !rc; i++) { th[i].d = d; th[i].next = &next; th[i].stop = &stop; rc = bench_thread_init(&th[i]); }
    if (!rc) { /* one untimed item per thread: pages the planes in, warms the caches */
        for (int i = 0; i < nt; i++) { bench_item(&th[i], i % d->n_pictures, d->row_start + (i / d->n_pictures) % d->n_rows); th[i].checksum = 0; }
        int started = 0;
        const double t0 = now_s();
        for (int i = 0; i < nt; i++, started++) if (pthread_create(&id[i], NULL, bench_thread_main, &th[i])) { rc = 3; break; }
        struct timespec nap = {0, 2000000};
        while (!rc && now_s() - t0 < d->seconds) nanosleep(&nap, NULL);
        if (d->seconds > 0 || rc) stop = 1;
        for (int i = 0; i < started; i++) pthread_join(id[i], NULL);
        d->elapsed = now_s() - t0; /* up to the last thread's last item */
        d->items_done = 0; d->checksum = 0;
        for (int i = 0; i < nt; i++) { d->items_done += th[i].done; d->checksum += th[i].checksum; rc |= th[i].failed ? 4 : 0; }
    }
    for (int i = 0; th && i < nt; i++) bench_thread_free(&th[i]);
    free(th); free(id);
    return rc;
}
size_t ref_sizeof_bench(int what) { return what == 0 ? sizeof(RefBenchDesc) : sizeof(RefBenchPicture); }

/* layout of the reference's MV_COST_PARAMS (Codec/mcomp.h:37-48): sizeof, then offsetof of each field in declaration order */
#include <stddef.h>
#include "mcomp.h"
void ref_mv_cost_param_layout(size_t out[9]) {
    out[0] = sizeof(MV_COST_PARAMS);
    out[1] = offsetof(MV_COST_PARAMS, ref_mv); out[2] = offsetof(MV_COST_PARAMS, full_ref_mv); out[3] = offsetof(MV_COST_PARAMS, mv_cost_type);
    out[4] = offsetof(MV_COST_PARAMS, mvjcost); out[5] = offsetof(MV_COST_PARAMS, mvcost); out[6] = offsetof(MV_COST_PARAMS, error_per_bit);
    out[7] = offsetof(MV_COST_PARAMS, early_exit_th); out[8] = offsetof(MV_COST_PARAMS, sad_per_bit);
    { MV_COST_PARAMS p; out[3] |= (size_t)sizeof(p.mv_cost_type) << 16; } /* width of the enum field in the high half */
}
