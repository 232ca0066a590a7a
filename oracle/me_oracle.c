/*
 * me_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement (plain C) of the reference's open-loop
 * motion-estimation path, used only as the parity checker by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing in the product path (svt-av1-psyex_amd/) links or calls this.
 *
 * Pinning: every function here is checked bit-exactly against the reference's own C functions
 * compiled from /root/reference into oracle/_ref/libsvtref.so (tests/test_oracle_vs_ref.py) and against
 * fixtures generated from that build (tests/golden/, generator: oracle/gen_golden.py).
 *
 * Each function cites the reference file:line (relative to Source/Lib/) whose behaviour it restates.
 */
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/svt_hip_me.h"

#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))
#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_ABS(a) ((a) < 0 ? -(a) : (a))
#define MVX(mv) ((int16_t)((mv) & 0xFFFF)) /* _MVXT, Codec/definitions.h:2360 */
#define MVY(mv) ((int16_t)((mv) >> 16))    /* _MVYT */

/* =====================================================================================
 * Leaf SAD kernels
 * ===================================================================================== */

/* svt_fast_loop_nxm_sad_kernel / svt_nxm_sad_kernel_helper_c: C_DEFAULT/compute_sad_c.c:20-37,209 */
/* Diagnostic: the number of |a - b| evaluations the SAD kernels made (every SAD of the path goes through orc_nxm_sad): the work measure
 * bench.py relates to the packed-SAD issue rate of the GPU ("fraction of packed-SAD VALU peak").  Counted per thread, added to
 * the global sum when the thread's orc_me_picture call returns. */
static __thread uint64_t t_sad_ops;
static uint64_t          g_sad_ops;
/* the same count per stage of the block pipeline: 0 zero-MV SADs, 1 pre-HME, 2 / 3 / 4 HME level 0 / 1 / 2, 5 everything behind (centre checks,
 * the 8x8-variance probe, the integer search) */
enum { ORC_N_STAGES = 6 };
static __thread int      t_stage;
static __thread uint64_t t_stage_ops[ORC_N_STAGES];
static uint64_t          g_stage_ops[ORC_N_STAGES];
uint64_t orc_sad_ops(int reset) {
    const uint64_t v = __atomic_load_n(&g_sad_ops, __ATOMIC_RELAXED);
    if (reset) __atomic_store_n(&g_sad_ops, 0, __ATOMIC_RELAXED);
    return v;
}
uint64_t orc_sad_ops_stage(int stage, int reset) {
    if (stage < 0 || stage >= ORC_N_STAGES) return 0;
    const uint64_t v = __atomic_load_n(&g_stage_ops[stage], __ATOMIC_RELAXED);
    if (reset) __atomic_store_n(&g_stage_ops[stage], 0, __ATOMIC_RELAXED);
    return v;
}

uint32_t orc_nxm_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height,
                     uint32_t width) {
    uint32_t acc = 0;
    t_sad_ops += (uint64_t)height * width;
    t_stage_ops[t_stage] += (uint64_t)height * width;
    for (uint32_t r = 0; r < height; r++)
        for (uint32_t c = 0; c < width; c++) {
            int d = (int)src[r * src_stride + c] - (int)ref[r * ref_stride + c];
            acc += (uint32_t)(d < 0 ? -d : d);
        }
    return acc;
}

/* svt_aom_sad_16b_kernel_c: C_DEFAULT/compute_sad_c.c:39-56 */
uint32_t orc_sad_16b(const uint16_t *src, uint32_t src_stride, const uint16_t *ref, uint32_t ref_stride, uint32_t height,
                     uint32_t width) {
    uint32_t acc = 0;
    for (uint32_t r = 0; r < height; r++)
        for (uint32_t c = 0; c < width; c++) {
            int d = (int)src[r * src_stride + c] - (int)ref[r * ref_stride + c];
            acc += (uint32_t)(d < 0 ? -d : d);
        }
    return acc;
}

/* svt_sad_loop_kernel_c: C_DEFAULT/compute_sad_c.c:58-101.
 * Exhaustive search; first minimum in raster order wins (strict <, initial 0xffffff); when the block is
 * 16 wide and at most 16 high and skip_search_line is set, even search rows are skipped.  The search rows
 * advance by src_stride_raw while block rows advance by ref_stride. Outputs untouched if nothing beats 0xffffff. */
void orc_sad_loop_kernel(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                         uint32_t block_height, uint32_t block_width, uint64_t *best_sad, int16_t *x_search_center,
                         int16_t *y_search_center, uint32_t src_stride_raw, uint8_t skip_search_line,
                         int16_t search_area_width, int16_t search_area_height) {
    const int skip_even = (block_width == 16 && block_height <= 16 && skip_search_line);
    *best_sad           = 0xffffff;
    for (int16_t ys = 0; ys < search_area_height; ys++) {
        const uint8_t *row = ref + (size_t)ys * src_stride_raw;
        if (skip_even && !(ys & 1))
            continue;
        for (int16_t xs = 0; xs < search_area_width; xs++) {
            uint32_t s = orc_nxm_sad(src, src_stride, row + xs, ref_stride, block_height, block_width);
            if (s < *best_sad) {
                *best_sad        = s;
                *x_search_center = xs;
                *y_search_center = ys;
            }
        }
    }
}

/* Row-subsampled or full 8x8 SAD: svt_aom_compute8x4_sad_kernel_c (doubled stride, <<1) /
 * compute8x8_sad_kernel_c, Codec/motion_estimation.c:42-91 as used at :105-136 */
static uint32_t sad8x8_me(const uint8_t *src, uint32_t ss, const uint8_t *ref, uint32_t rs, int sub_sad) {
    if (sub_sad)
        return orc_nxm_sad(src, 2 * ss, ref, 2 * rs, 4, 8) << 1;
    return orc_nxm_sad(src, ss, ref, rs, 8, 8);
}

/* svt_ext_sad_calculation_8x8_16x16_c: Codec/motion_estimation.c:98-164 */
void orc_ext_sad_calculation_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                                       uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
                                       uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16, uint32_t *p_sad8x8,
                                       bool sub_sad) {
    uint32_t total = 0;
    for (int q = 0; q < 4; q++) {
        const uint32_t off_s = (q >> 1) * 8 * src_stride + (q & 1) * 8;
        const uint32_t off_r = (q >> 1) * 8 * ref_stride + (q & 1) * 8;
        p_sad8x8[q]          = sad8x8_me(src + off_s, src_stride, ref + off_r, ref_stride, sub_sad);
        if (p_sad8x8[q] < p_best_sad_8x8[q]) {
            p_best_sad_8x8[q] = p_sad8x8[q];
            p_best_mv8x8[q]   = mv;
        }
        total += p_sad8x8[q];
    }
    if (total < p_best_sad_16x16[0]) {
        p_best_sad_16x16[0] = total;
        p_best_mv16x16[0]   = mv;
    }
    *p_sad16x16 = total;
}

/* svt_ext_sad_calculation_32x32_64x64_c: Codec/motion_estimation.c:171-205 */
void orc_ext_sad_calculation_32x32_64x64(const uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32,
                                         uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64,
                                         uint32_t mv, uint32_t *p_sad32x32) {
    uint32_t total = 0;
    for (int q = 0; q < 4; q++) {
        uint32_t s = p_sad16x16[4 * q] + p_sad16x16[4 * q + 1] + p_sad16x16[4 * q + 2] + p_sad16x16[4 * q + 3];
        p_sad32x32[q] = s;
        if (s < p_best_sad_32x32[q]) {
            p_best_sad_32x32[q] = s;
            p_best_mv32x32[q]   = mv;
        }
        total += s;
    }
    if (total < p_best_sad_64x64[0]) {
        p_best_sad_64x64[0] = total;
        p_best_mv64x64[0]   = mv;
    }
}

/* 16x16 raster position -> index in the reference's PU order: Codec/motion_estimation.c:341 */
static const uint8_t k_raster16_to_z[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};

static uint32_t mv_add_x(uint32_t mv, int k) {
    int16_t x = (int16_t)(MVX(mv) + (int16_t)k);
    int16_t y = MVY(mv);
    return ((uint32_t)y << 16) | (uint16_t)x;
}

/* svt_ext_all_sad_calculation_8x8_16x16_c (+ static svt_ext_eight_sad_calculation_8x8_16x16):
 * Codec/motion_estimation.c:210-362.  For each 16x16 of the 64x64 (raster), for each of 8 consecutive
 * x positions: four 8x8 SADs, update 8x8 / 16x16 bests, emit p_eight_sad16x16[z16][k]. */
void orc_ext_all_sad_calculation_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                           uint32_t ref_stride, uint32_t mv, uint32_t *p_best_sad_8x8,
                                           uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16,
                                           uint32_t p_eight_sad16x16[16][8], uint32_t p_eight_sad8x8[64][8], bool sub_sad) {
    (void)p_eight_sad8x8;
    for (int b = 0; b < 16; b++) {
        const int      z16 = k_raster16_to_z[b];
        const uint8_t *s   = src + (size_t)(b >> 2) * 16 * src_stride + (b & 3) * 16;
        const uint8_t *r   = ref + (size_t)(b >> 2) * 16 * ref_stride + (b & 3) * 16;
        for (int k = 0; k < 8; k++) {
            uint32_t total = 0;
            for (int q = 0; q < 4; q++) {
                uint32_t v = sad8x8_me(s + (q >> 1) * 8 * src_stride + (q & 1) * 8, src_stride,
                                       r + (q >> 1) * 8 * ref_stride + (q & 1) * 8 + k, ref_stride, sub_sad);
                if (v < p_best_sad_8x8[4 * z16 + q]) {
                    p_best_sad_8x8[4 * z16 + q] = v;
                    p_best_mv8x8[4 * z16 + q]   = mv_add_x(mv, k);
                }
                total += v;
            }
            p_eight_sad16x16[z16][k] = total;
            if (total < p_best_sad_16x16[z16]) {
                p_best_sad_16x16[z16] = total;
                p_best_mv16x16[z16]   = mv_add_x(mv, k);
            }
        }
    }
}

/* svt_ext_eight_sad_calculation_32x32_64x64_c: Codec/motion_estimation.c:369-425 */
void orc_ext_eight_sad_calculation_32x32_64x64(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32,
                                               uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                               uint32_t *p_best_mv64x64, uint32_t mv, uint32_t p_sad32x32[4][8]) {
    for (int k = 0; k < 8; k++) {
        uint32_t total = 0;
        for (int q = 0; q < 4; q++) {
            uint32_t s = p_sad16x16[4 * q][k] + p_sad16x16[4 * q + 1][k] + p_sad16x16[4 * q + 2][k] +
                p_sad16x16[4 * q + 3][k];
            p_sad32x32[q][k] = s;
            if (s < p_best_sad_32x32[q]) {
                p_best_sad_32x32[q] = s;
                p_best_mv32x32[q]   = mv_add_x(mv, k);
            }
            total += s;
        }
        if (total < p_best_sad_64x64[0]) {
            p_best_sad_64x64[0] = total;
            p_best_mv64x64[0]   = mv_add_x(mv, k);
        }
    }
}

/* svt_initialize_buffer_32bits_c: Codec/me_sad_calculation.c:14-17 */
void orc_initialize_buffer_32bits(uint32_t *p, uint32_t count128, uint32_t count32, uint32_t value) {
    for (uint32_t i = 0; i < count128 * 4 + count32; i++) p[i] = value;
}

/* svt_aom_downsample_2d_c: Codec/pic_analysis_process.c:130-158 (2x2 box on the pixel pairs
 * (step/2-1, step/2) of every step-th row/column, rounded) */
void orc_downsample_2d(const uint8_t *in, uint32_t in_stride, uint32_t in_w, uint32_t in_h, uint8_t *out,
                       uint32_t out_stride, uint32_t step) {
    const uint32_t half = step >> 1;
    uint32_t       oy   = 0;
    for (uint32_t y = half; y < in_h; y += step, oy++) {
        const uint8_t *r1 = in + (size_t)y * in_stride, *r0 = r1 - in_stride;
        uint32_t       ox = 0;
        for (uint32_t x = half; x < in_w; x += step, ox++)
            out[(size_t)oy * out_stride + ox] = (uint8_t)(((uint32_t)r0[x - 1] + r0[x] + r1[x - 1] + r1[x] + 2) >> 2);
    }
}

/* svt_aom_generate_padding: Codec/pic_operators.c:397-443 (replicate edge columns, then rows) */
void orc_generate_padding(uint8_t *buf, uint32_t stride, uint32_t w, uint32_t h, uint32_t pad_w, uint32_t pad_h) {
    uint8_t *p = buf + pad_w + (size_t)pad_h * stride;
    for (uint32_t y = 0; y < h; y++, p += stride) {
        memset(p - pad_w, p[0], pad_w);
        memset(p + w, p[w - 1], pad_w);
    }
    uint8_t *top = buf + (size_t)pad_h * stride, *bot = buf + (size_t)(pad_h + h - 1) * stride;
    for (uint32_t y = 1; y <= pad_h; y++) {
        memcpy(top - (size_t)y * stride, top, stride);
        memcpy(bot + (size_t)y * stride, bot, stride);
    }
}

/* =====================================================================================
 * Per-b64 open-loop ME pipeline: svt_aom_motion_estimation_b64, Codec/motion_estimation.c:3076-3153
 * ===================================================================================== */

typedef struct OrcPyramid {
    SvtHipPlaneDesc lvl[3]; /* 0 = sixteenth, 1 = quarter, 2 = full */
} OrcPyramid;

typedef struct PreHme {
    int      sa_w, sa_h;
    int16_t  col, row;
    uint64_t sad;
    uint8_t  valid;
} PreHme;

typedef struct SearchRes {
    int16_t  hme_sc_x, hme_sc_y;
    uint64_t hme_sad;
    uint8_t  do_ref;
} SearchRes;

typedef struct MeState {
    const SvtHipMeConfig      *cfg;
    const SvtHipMePictureDesc *pic;
    const OrcPyramid          *cur;
    const OrcPyramid          *ref[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
    SvtHipSearchAreaMinMax     hme_l0_sa; /* mutable copy: get_hme_l0_search_area rescales it per ref */
    uint32_t                   org_x, org_y, b64_w, b64_h;
    const uint8_t             *src[3];
    uint32_t                   src_stride[3];
    uint32_t                   best_sad[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][85];
    uint32_t                   best_mv[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][85];
    uint32_t                   me_distortion[85];
    SearchRes                  sr[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
    uint32_t                   sr_divisor[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
    uint32_t                   zz_sad[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
    PreHme                     prehme[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][2];
    uint8_t                    performed_phme[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][2];
    /* [level][list][ref][sr_w][sr_h] */
    int16_t  hme_x[3][SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][2][2], hme_y[3][SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][2][2];
    uint64_t hme_sad[3][SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][2][2];
} MeState;

/* svt_aom_get_scaled_picture_distance: Codec/motion_estimation.c:1239-1243 */
static uint16_t scaled_distance(uint16_t dist) { return (uint16_t)((dist * 5) / 8 + ((dist % 8) ? 1 : 0)); }

static uint16_t ref_distance(const MeState *s, int li, int ri) {
    /* get_me_reference: Codec/motion_estimation.c:1232-1235 */
    int64_t d = (int64_t)s->pic->picture_number - (int64_t)s->pic->ref_picture_number[li][ri];
    return (uint16_t)(int16_t)(d < 0 ? -d : d);
}

/* One axis of the "correct the search area if it is not on the reference picture" sequence that
 * hme_level_0/1/2 (:838-866 etc.), prehme_core (:1585-1627) and integer_search_b64 (:1455-1499) share:
 * the low edge only moves the origin, the high edge moves the origin then crops the size. */
static void clip_axis(int org, int *origin, int *size, int pad, int dim) {
    if (org + *origin < -pad)
        *origin = -pad - org;
    if (org + *origin > dim - 1)
        *origin -= (org + *origin) - (dim - 1);
    if (org + *origin + *size > dim)
        *size = ORC_MAX(1, *size - ((org + *origin + *size) - dim));
}

/* The svt_sad_loop_kernel call the three HME levels and pre-HME make (e.g. :891-909): level picks
 * the source view and plane; rows are sub-sampled when hme_search_method is SUB_SAD. */
static void hme_sad_search(const MeState *s, int level, const SvtHipPlaneDesc *rp, int org_x, int org_y, int bw, int bh,
                           int origin_x, int origin_y, int sa_w, int sa_h, uint8_t skip_line, uint64_t *best_sad,
                           int16_t *bx, int16_t *by) {
    const int      full  = (s->cfg->hme_search_method == 1);
    const int64_t  index = ((int64_t)rp->org_x + org_x + origin_x) + ((int64_t)rp->org_y + org_y + origin_y) * rp->stride_y;
    orc_sad_loop_kernel(s->src[level], full ? s->src_stride[level] : s->src_stride[level] * 2, rp->buffer_y + index,
                        full ? rp->stride_y : rp->stride_y * 2, full ? (uint32_t)bh : (uint32_t)bh >> 1, (uint32_t)bw,
                        best_sad, bx, by, rp->stride_y, skip_line, (int16_t)sa_w, (int16_t)sa_h);
    if (!full)
        *best_sad *= 2;
}

/* hme_level_0 / hme_level_1 / hme_level_2: Codec/motion_estimation.c:820-920, 923-1022, 1025-1113 */
static void hme_level(const MeState *s, int level, const SvtHipPlaneDesc *rp, int org_x, int org_y, int bw, int bh,
                      int sa_w, int sa_h, int centre_x, int centre_y, int sr_w, int sr_h, uint64_t *best_sad, int16_t *out_x,
                      int16_t *out_y) {
    sa_w = (int16_t)((sa_w + 7) & ~7);
    int pad_w, pad_h, ox, oy;
    if (level == 2) {
        pad_w = pad_h = 63;
    } else {
        pad_w = (int16_t)rp->org_x - 1;
        pad_h = (int16_t)rp->org_y - 1;
    }
    if (level == 0) {
        ox = -(int16_t)((sa_w * s->cfg->num_hme_sa_w) >> 1) + (int16_t)(sa_w * sr_w);
        oy = -(int16_t)((sa_h * s->cfg->num_hme_sa_h) >> 1) + (int16_t)(sa_h * sr_h);
    } else {
        ox = -(sa_w >> 1) + centre_x;
        oy = -(sa_h >> 1) + centre_y;
    }
    clip_axis(org_x, &ox, &sa_w, pad_w, rp->width);
    sa_w = (sa_w < 8) ? sa_w : (sa_w & ~7);
    clip_axis(org_y, &oy, &sa_h, pad_h, rp->height);
    hme_sad_search(s, level, rp, org_x, org_y, bw, bh, ox, oy, sa_w, sa_h, 0, best_sad, out_x, out_y);
    const int scale = (level == 0) ? 4 : (level == 1) ? 2 : 1;
    *out_x          = (int16_t)((int16_t)(*out_x + ox) * scale);
    *out_y          = (int16_t)((int16_t)(*out_y + oy) * scale);
}

/* get_zz_sad: Codec/motion_estimation.c:1667-1689 (rows sub-sampled by 2, <<1) */
static uint32_t zero_mv_sad(const MeState *s, const SvtHipPlaneDesc *rp, int dx, int dy) {
    const int64_t index = ((int64_t)rp->org_x + (int)s->org_x + dx) + ((int64_t)rp->org_y + (int)s->org_y + dy) * rp->stride_y;
    return orc_nxm_sad(s->src[2], s->src_stride[2] << 1, rp->buffer_y + index, rp->stride_y << 1, s->b64_h >> 1, s->b64_w) << 1;
}

/* init_zz_sad: Codec/motion_estimation.c:2382-2437 */
static void init_zz_sad(MeState *s) {
    const SvtHipMeConfig *c    = s->cfg;
    uint32_t              best = 0xFFFFFFFFu;
    for (int li = 0; li < s->pic->num_of_list_to_search; li++)
        for (int ri = 0; ri < s->pic->num_of_ref_pic_to_search[li]; ri++)
            if (s->pic->temporal_layer_index > 0 || li == 0) {
                uint32_t z        = zero_mv_sad(s, &s->ref[li][ri]->lvl[2], 0, 0);
                z                 = (z * 64 * 64) / (s->b64_w * s->b64_h);
                s->zz_sad[li][ri] = z;
                best              = ORC_MIN(best, z);
            }
    if (s->pic->temporal_layer_index > 0 && best < c->zz_sad_th)
        for (int li = 0; li < s->pic->num_of_list_to_search; li++)
            for (int ri = 1; ri < s->pic->num_of_ref_pic_to_search[li]; ri++)
                if ((uint32_t)((s->zz_sad[li][ri] - best) * 100) > (uint32_t)(c->zz_sad_pct * best))
                    s->sr[li][ri].do_ref = 0;
    if (c->me_safe_limit_zz_th) {
        int limit = s->pic->hierarchical_levels > 0 && s->pic->num_of_list_to_search == 2 &&
            s->pic->temporal_layer_index >= s->pic->hierarchical_levels && s->pic->similar_brightness_refs &&
            s->zz_sad[0][0] < c->me_safe_limit_zz_th && s->zz_sad[1][0] < c->me_safe_limit_zz_th;
        if (limit)
            for (int li = 0; li < s->pic->num_of_list_to_search; li++)
                for (int ri = 1; ri < s->pic->num_of_ref_pic_to_search[li]; ri++) s->sr[li][ri].do_ref = 0;
    }
}

/* check_prehme_early_exit: Codec/motion_estimation.c:1693-1720 */
static int prehme_early_exit(MeState *s, int li, int ri, int sri) {
    const SvtHipMeConfig *c = s->cfg;
    PreHme               *p = &s->prehme[li][ri][sri];
    if (c->me_early_exit_th && s->zz_sad[li][ri] < c->me_early_exit_th) {
        p->col = p->row = 0;
        p->sad          = 0;
        p->valid        = 1;
        return 1;
    }
    if (c->prehme_l1_early_exit) {
        const PreHme *q = &s->prehme[0][ri][sri];
        if (li == 1 && q->valid && (q->sad < 32 * 32 || (ORC_ABS(q->col) < 16 && ORC_ABS(q->row) < 16))) {
            p->col   = (int16_t)-q->col;
            p->row   = (int16_t)-q->row;
            p->sad   = q->sad;
            p->valid = 1;
            return 1;
        }
    }
    return 0;
}

/* prehme_core: Codec/motion_estimation.c:1568-1666 */
static void prehme_core(MeState *s, const SvtHipPlaneDesc *rp, PreHme *p) {
    const int org_x = (int16_t)s->org_x >> 2, org_y = (int16_t)s->org_y >> 2;
    int       sa_w = (int16_t)p->sa_w, sa_h = (int16_t)p->sa_h;
    int       ox = -(int16_t)(sa_w >> 1), oy = -(int16_t)(sa_h >> 1);
    clip_axis(org_x, &ox, &sa_w, (int16_t)rp->org_x - 1, rp->width);
    clip_axis(org_y, &oy, &sa_h, (int16_t)rp->org_y - 1, rp->height);
    hme_sad_search(s, 0, rp, org_x, org_y, s->b64_w >> 2, s->b64_h >> 2, ox, oy, sa_w, sa_h,
                   s->cfg->prehme_skip_search_line, &p->sad, &p->col, &p->row);
    p->col   = (int16_t)((int16_t)(p->col + ox) * 4);
    p->row   = (int16_t)((int16_t)(p->row + oy) * 4);
    p->valid = 1;
}

/* prehme_b64: Codec/motion_estimation.c:1722-1796 */
static void prehme_b64(MeState *s) {
    const SvtHipMeConfig *c        = s->cfg;
    uint32_t              best_sad = 0xFFFFFFFFu;
    for (int li = 0; li < s->pic->num_of_list_to_search; li++)
        for (int ri = 0; ri < s->pic->num_of_ref_pic_to_search[li]; ri++) {
            if (s->pic->temporal_layer_index > 0 || li == 0) {
                const uint32_t f = scaled_distance(ref_distance(s, li, ri));
                for (int sri = 0; sri < 2; sri++) {
                    if (prehme_early_exit(s, li, ri, sri))
                        continue;
                    PreHme *p = &s->prehme[li][ri][sri];
                    if (!s->sr[li][ri].do_ref) {
                        p->col = p->row = 0;
                        p->sad          = 0xFFFFFFFFu;
                        continue;
                    }
                    p->sa_w = (uint16_t)ORC_MIN(c->prehme_sa_cfg[sri].sa_min.width * f, c->prehme_sa_cfg[sri].sa_max.width);
                    p->sa_h = (uint16_t)ORC_MIN(c->prehme_sa_cfg[sri].sa_min.height * f, c->prehme_sa_cfg[sri].sa_max.height);
                    prehme_core(s, &s->ref[li][ri]->lvl[0], p);
                    s->performed_phme[li][ri][sri] = 1;
                }
                uint32_t m = (uint32_t)ORC_MIN(s->prehme[li][ri][0].sad, s->prehme[li][ri][1].sad);
                best_sad   = ORC_MIN(best_sad, m);
            } else {
                for (int sri = 0; sri < 2; sri++) {
                    s->prehme[1][ri][sri].col = (int16_t)-s->prehme[0][ri][sri].col;
                    s->prehme[1][ri][sri].row = (int16_t)-s->prehme[0][ri][sri].row;
                    s->prehme[1][ri][sri].sad = s->prehme[0][ri][sri].sad;
                }
            }
        }
    if (s->pic->temporal_layer_index > 0 && best_sad < c->phme_sad_th)
        for (int li = 0; li < s->pic->num_of_list_to_search; li++)
            for (int ri = 1; ri < s->pic->num_of_ref_pic_to_search[li]; ri++) {
                if (!s->sr[li][ri].do_ref)
                    continue;
                uint32_t m = (uint32_t)ORC_MIN(s->prehme[li][ri][0].sad, s->prehme[li][ri][1].sad);
                if ((uint32_t)((m - best_sad) * 100) > (uint32_t)(c->phme_sad_pct * best_sad))
                    s->sr[li][ri].do_ref = 0;
            }
}

/* get_hme_l0_search_area: Codec/motion_estimation.c:1800-1868 */
static void hme_l0_search_area(MeState *s, int li, int ri, uint16_t dist, int16_t *sa_w, int16_t *sa_h) {
    const SvtHipMeConfig *c = s->cfg;
    if (c->enable_me_sr_adjustment && c->distance_based_hme_resizing) {
        int is_hor = 1, is_ver = 1, is_still = 0;
        if (c->reduce_hme_l0_sr_th_min && c->reduce_hme_l0_sr_th_max && (li || ri)) {
            const int mvx = s->hme_x[0][0][0][0][0], mvy = s->hme_y[0][0][0][0][0];
            is_ver   = ORC_ABS(mvx) < c->reduce_hme_l0_sr_th_min && ORC_ABS(mvy) > c->reduce_hme_l0_sr_th_max;
            is_hor   = ORC_ABS(mvx) > c->reduce_hme_l0_sr_th_max && ORC_ABS(mvy) < c->reduce_hme_l0_sr_th_min;
            is_still = ORC_ABS(mvx) < c->reduce_hme_l0_sr_th_min * 3 && ORC_ABS(mvy) < c->reduce_hme_l0_sr_th_min * 3;
        }
        uint8_t xo = is_hor ? 1 : 2, yo = is_ver ? 1 : 2;
        if (c->enable_me_sr_adjustment == 2 && is_still)
            xo = yo = 4;
        s->hme_l0_sa.sa_min.width  = (uint16_t)(s->hme_l0_sa.sa_min.width / (xo + ri));
        s->hme_l0_sa.sa_min.height = (uint16_t)(s->hme_l0_sa.sa_min.height / (yo + ri));
        s->hme_l0_sa.sa_max.width  = (uint16_t)(s->hme_l0_sa.sa_max.width / (xo + ri));
        s->hme_l0_sa.sa_max.height = (uint16_t)(s->hme_l0_sa.sa_max.height / (yo + ri));
    }
    const int32_t f = scaled_distance(dist);
    int16_t       w = (int16_t)(s->hme_l0_sa.sa_min.width / c->num_hme_sa_w);
    w               = (int16_t)ORC_MIN(((w * f) + 15) & ~15, ((s->hme_l0_sa.sa_max.width / c->num_hme_sa_w) + 15) & ~15);
    int16_t h       = (int16_t)(s->hme_l0_sa.sa_min.height / c->num_hme_sa_h);
    h               = (int16_t)ORC_MIN(h * f, s->hme_l0_sa.sa_max.height / c->num_hme_sa_h);
    *sa_w           = w;
    *sa_h           = h;
}

static void set_hme_all(MeState *s, int lvl, int li, int ri, int16_t x, int16_t y, uint64_t sad) {
    for (int h = 0; h < s->cfg->num_hme_sa_h; h++)
        for (int w = 0; w < s->cfg->num_hme_sa_w; w++) {
            s->hme_x[lvl][li][ri][w][h]   = x;
            s->hme_y[lvl][li][ri][w][h]   = y;
            s->hme_sad[lvl][li][ri][w][h] = sad;
        }
}

/* hme_level0_b64: Codec/motion_estimation.c:1906-2036 (incl. get_worst_quadrant :1872-1901) */
static void hme_level0_b64(MeState *s) {
    const SvtHipMeConfig        *c    = s->cfg;
    const SvtHipSearchAreaMinMax base = s->hme_l0_sa;
    for (int li = 0; li < s->pic->num_of_list_to_search; li++)
        for (int ri = 0; ri < s->pic->num_of_ref_pic_to_search[li]; ri++) {
            if (c->me_early_exit_th && s->zz_sad[li][ri] < (c->me_early_exit_th >> 2)) {
                set_hme_all(s, 0, li, ri, 0, 0, 0);
                continue;
            }
            if (c->prev_me_stage_based_exit_th) {
                const int sri = s->prehme[li][ri][0].sad <= s->prehme[li][ri][1].sad ? 0 : 1;
                if (s->performed_phme[li][ri][sri] && s->prehme[li][ri][sri].sad < (c->prev_me_stage_based_exit_th >> 4)) {
                    set_hme_all(s, 0, li, ri, s->prehme[li][ri][sri].col, s->prehme[li][ri][sri].row, s->prehme[li][ri][sri].sad);
                    continue;
                }
            }
            if (!s->sr[li][ri].do_ref) {
                set_hme_all(s, 0, li, ri, 0, 0, 0xFFFFFFFFu);
                continue;
            }
            if (!(s->pic->temporal_layer_index > 0 || li == 0))
                continue;
            int16_t sa_w = 0, sa_h = 0;
            hme_l0_search_area(s, li, ri, ref_distance(s, li, ri), &sa_w, &sa_h);
            for (int h = 0; h < c->num_hme_sa_h; h++)
                for (int w = 0; w < c->num_hme_sa_w; w++)
                    hme_level(s, 0, &s->ref[li][ri]->lvl[0], (int16_t)s->org_x >> 2, (int16_t)s->org_y >> 2, s->b64_w >> 2,
                              s->b64_h >> 2, sa_w, sa_h, 0, 0, w, h, &s->hme_sad[0][li][ri][w][h], &s->hme_x[0][li][ri][w][h],
                              &s->hme_y[0][li][ri][w][h]);
            if (c->enable_me_sr_adjustment && c->distance_based_hme_resizing)
                s->hme_l0_sa = base;
            if (c->prehme_enable) {
                /* worst quadrant (2x2 only), order (0,0),(1,0),(0,1),(1,1); the last compare does not update max */
                int      ww = 0, wh = 0;
                uint64_t mx = 0;
                if (s->hme_sad[0][li][ri][0][0] > mx) { mx = s->hme_sad[0][li][ri][0][0]; ww = 0; wh = 0; }
                if (s->hme_sad[0][li][ri][1][0] > mx) { mx = s->hme_sad[0][li][ri][1][0]; ww = 1; wh = 0; }
                if (s->hme_sad[0][li][ri][0][1] > mx) { mx = s->hme_sad[0][li][ri][0][1]; ww = 0; wh = 1; }
                if (s->hme_sad[0][li][ri][1][1] > mx) { ww = 1; wh = 1; }
                const int sri = s->prehme[li][ri][0].sad <= s->prehme[li][ri][1].sad ? 0 : 1;
                if (s->prehme[li][ri][sri].sad < s->hme_sad[0][li][ri][ww][wh]) {
                    s->hme_sad[0][li][ri][ww][wh] = s->prehme[li][ri][sri].sad;
                    s->hme_x[0][li][ri][ww][wh]   = s->prehme[li][ri][sri].col;
                    s->hme_y[0][li][ri][ww][wh]   = s->prehme[li][ri][sri].row;
                }
            }
        }
}

/* hme_level1_b64 / hme_level2_b64: Codec/motion_estimation.c:2041-2122, 2127-2177 */
static void hme_level12_b64(MeState *s, int lvl) {
    const SvtHipMeConfig *c = s->cfg;
    for (int li = 0; li < s->pic->num_of_list_to_search; li++)
        for (int ri = 0; ri < s->pic->num_of_ref_pic_to_search[li]; ri++) {
            if (!(s->pic->temporal_layer_index > 0 || li == 0))
                continue;
            if (lvl == 1) {
                if (c->me_early_exit_th && s->zz_sad[li][ri] < (c->me_early_exit_th >> 2)) {
                    set_hme_all(s, 1, li, ri, 0, 0, 0);
                    continue;
                }
                if (!s->sr[li][ri].do_ref) {
                    set_hme_all(s, 1, li, ri, 0, 0, 0xFFFFFFFFu);
                    continue;
                }
            }
            for (int h = 0; h < c->num_hme_sa_h; h++)
                for (int w = 0; w < c->num_hme_sa_w; w++) {
                    const uint32_t exit_th = c->prev_me_stage_based_exit_th >> (lvl == 1 ? 5 : 2);
                    if (c->prev_me_stage_based_exit_th && s->hme_sad[lvl - 1][li][ri][w][h] < exit_th) {
                        s->hme_x[lvl][li][ri][w][h]   = s->hme_x[lvl - 1][li][ri][w][h];
                        s->hme_y[lvl][li][ri][w][h]   = s->hme_y[lvl - 1][li][ri][w][h];
                        s->hme_sad[lvl][li][ri][w][h] = s->hme_sad[lvl - 1][li][ri][w][h];
                        continue;
                    }
                    if (lvl == 1)
                        hme_level(s, 1, &s->ref[li][ri]->lvl[1], (int16_t)s->org_x >> 1, (int16_t)s->org_y >> 1, s->b64_w >> 1,
                                  s->b64_h >> 1, (int16_t)c->hme_l1_sa.width, (int16_t)c->hme_l1_sa.height,
                                  s->hme_x[0][li][ri][w][h] >> 1, s->hme_y[0][li][ri][w][h] >> 1, 0, 0,
                                  &s->hme_sad[1][li][ri][w][h], &s->hme_x[1][li][ri][w][h], &s->hme_y[1][li][ri][w][h]);
                    else
                        hme_level(s, 2, &s->ref[li][ri]->lvl[2], (int16_t)s->org_x, (int16_t)s->org_y, s->b64_w, s->b64_h,
                                  (int16_t)c->hme_l2_sa.width, (int16_t)c->hme_l2_sa.height, s->hme_x[1][li][ri][w][h],
                                  s->hme_y[1][li][ri][w][h], 0, 0, &s->hme_sad[2][li][ri][w][h], &s->hme_x[2][li][ri][w][h],
                                  &s->hme_y[2][li][ri][w][h]);
                }
        }
}

/* set_final_seach_centre_sb: Codec/motion_estimation.c:2182-2380.  The last enabled level decides;
 * regions are scanned w-inner / h-outer starting at (1,0) with strict <.  hme_sad is a variable that
 * survives across refs (a ref that is not searched inherits the previous ref's value). */
static void set_final_search_centre(MeState *s) {
    const SvtHipMeConfig *c      = s->cfg;
    int16_t               cx = 0, cy = 0, sx = 0, sy = 0;
    uint64_t              hme_sad = 0;
    for (int li = 0; li < s->pic->num_of_list_to_search; li++)
        for (int ri = 0; ri < s->pic->num_of_ref_pic_to_search[li]; ri++) {
            if (s->pic->temporal_layer_index > 0 || li == 0) {
                if (c->enable_hme_flag) {
                    int lvl = -1;
                    if (c->enable_hme_level0_flag && !c->enable_hme_level1_flag && !c->enable_hme_level2_flag)
                        lvl = 0;
                    if (c->enable_hme_level1_flag && !c->enable_hme_level2_flag)
                        lvl = 1;
                    if (c->enable_hme_level2_flag)
                        lvl = 2;
                    if (lvl >= 0) {
                        cx      = s->hme_x[lvl][li][ri][0][0];
                        cy      = s->hme_y[lvl][li][ri][0][0];
                        hme_sad = s->hme_sad[lvl][li][ri][0][0];
                        int w   = 1;
                        for (int h = 0; h < c->num_hme_sa_h; h++) {
                            for (; w < c->num_hme_sa_w; w++)
                                if (s->hme_sad[lvl][li][ri][w][h] < hme_sad) {
                                    cx      = s->hme_x[lvl][li][ri][w][h];
                                    cy      = s->hme_y[lvl][li][ri][w][h];
                                    hme_sad = s->hme_sad[lvl][li][ri][w][h];
                                }
                            w = 0;
                        }
                    }
                    sx = cx;
                    sy = cy;
                }
            } else {
                sx = sy = 0;
            }
            s->sr[li][ri].hme_sc_x = sx;
            s->sr[li][ri].hme_sc_y = sy;
            s->sr[li][ri].hme_sad  = hme_sad;
        }
}

/* hme_prune_ref_and_adjust_sr: Codec/motion_estimation.c:2477-2518 */
static void hme_prune_and_adjust(MeState *s) {
    const SvtHipMeConfig *c  = s->cfg;
    const uint16_t        th = c->prune_ref_if_hme_sad_dev_bigger_than_th;
    if (c->enable_me_hme_ref_pruning && th != 0xFFFF) {
        uint64_t best = ~(uint64_t)0;
        for (int li = 0; li < 2; li++)
            for (int ri = 0; ri < 4; ri++) best = ORC_MIN(best, s->sr[li][ri].hme_sad);
        for (int li = 0; li < 2; li++)
            for (int ri = 1; ri < 4; ri++)
                if ((s->sr[li][ri].hme_sad - best) * 100 > (uint64_t)th * best)
                    s->sr[li][ri].do_ref = 0;
    }
    if (c->enable_me_sr_adjustment)
        for (int li = 0; li < 2; li++)
            for (int ri = 0; ri < 4; ri++) {
                const SearchRes *r = &s->sr[li][ri];
                if (ORC_ABS(r->hme_sc_x) <= c->reduce_me_sr_based_on_mv_length_th &&
                    ORC_ABS(r->hme_sc_y) <= c->reduce_me_sr_based_on_mv_length_th && r->hme_sad < c->stationary_hme_sad_abs_th)
                    s->sr_divisor[li][ri] = c->stationary_me_sr_divisor;
                else if (r->hme_sad < c->reduce_me_sr_based_on_hme_sad_abs_th)
                    s->sr_divisor[li][ri] = c->me_sr_divisor_for_low_hme_sad;
            }
}

/* check_00_center: Codec/motion_estimation.c:1139-1206 */
static uint32_t check_00_center(MeState *s, const SvtHipPlaneDesc *rp, int16_t *cx, int16_t *cy, uint32_t zz) {
    const int org_x = (int16_t)s->org_x, org_y = (int16_t)s->org_y;
    uint32_t  zero  = s->cfg->me_early_exit_th ? zz : (zero_mv_sad(s, rp, 0, 0) >> 1);
    zero <<= 1;
    if (org_x + *cx < -63) *cx = (int16_t)(-63 - org_x);
    if (org_x + *cx > (int16_t)rp->width - 1) *cx = (int16_t)(*cx - ((org_x + *cx) - ((int16_t)rp->width - 1)));
    if (org_y + *cy < -63) *cy = (int16_t)(-63 - org_y);
    if (org_y + *cy > (int16_t)rp->height - 1) *cy = (int16_t)(*cy - ((org_y + *cy) - ((int16_t)rp->height - 1)));
    const uint32_t hme = zero_mv_sad(s, rp, *cx, *cy);
    const uint64_t zc = (uint64_t)zero << 8, hc = (uint64_t)hme << 8; /* COST_PRECISION = 8 */
    if (ORC_MIN(zc, hc) == zc)
        *cx = *cy = 0;
    return hme;
}

/* open_loop_me_fullpel_search_sblock + the two per-point helpers: Codec/motion_estimation.c:429-817.
 * `win` points at the reference sample matching search index (0,0). */
static void fullpel_search(MeState *s, int li, int ri, const uint8_t *win, uint32_t stride, int16_t origin_x, int16_t origin_y,
                           uint32_t sa_w, uint32_t sa_h) {
    const int sub     = (s->cfg->me_search_method == 0);
    uint32_t *bs      = s->best_sad[li][ri];
    uint32_t *bm      = s->best_mv[li][ri];
    const uint32_t w8 = sa_w & ~7u;
    uint32_t       e16[16][8], e32[4][8], s16[16], s8[64], s32[4];
    for (uint32_t y = 0; y < sa_h; y++) {
        for (uint32_t x = 0; x < w8; x += 8) {
            const uint32_t mv = ((uint32_t)(int32_t)((int32_t)y + origin_y) << 16) | (uint16_t)((int32_t)x + origin_x);
            orc_ext_all_sad_calculation_8x8_16x16(s->src[2], s->src_stride[2], win + x + (size_t)y * stride, stride, mv, bs + 21,
                                                  bs + 5, bm + 21, bm + 5, e16, NULL, sub);
            orc_ext_eight_sad_calculation_32x32_64x64(e16, bs + 1, bs, bm + 1, bm, mv, e32);
        }
        for (uint32_t x = w8; x < sa_w; x++) {
            const uint32_t mv = ((uint32_t)(int32_t)((int32_t)y + origin_y) << 16) | (uint16_t)((int32_t)x + origin_x);
            const uint8_t *r  = win + x + (size_t)y * stride;
            for (int b = 0; b < 16; b++) {
                const int z = k_raster16_to_z[b];
                orc_ext_sad_calculation_8x8_16x16(s->src[2] + (size_t)(b >> 2) * 16 * s->src_stride[2] + (b & 3) * 16,
                                                  s->src_stride[2], r + (size_t)(b >> 2) * 16 * stride + (b & 3) * 16, stride,
                                                  bs + 21 + 4 * z, bs + 5 + z, bm + 21 + 4 * z, bm + 5 + z, mv, &s16[z],
                                                  &s8[4 * z], sub);
            }
            orc_ext_sad_calculation_32x32_64x64(s16, bs + 1, bs, bm + 1, bm, mv, s32);
        }
    }
}

/* integer_search_b64: Codec/motion_estimation.c:1249-1516 */
/* The 64x64 block displaced to (x .. x + w - 1 + 63, y .. y + h - 1 + 63) in picture coordinates.  The reference forms
 * this address without a bounds check, and its 1-point probe (:1391-1406) uses the UNCLIPPED search centre: a centre far outside
 * the picture (e.g. the mirrored list-1 pre-HME vector of check_prehme_early_exit, :1693-1720, with HME levels 1 / 2 off) makes
 * the reference read past its padded plane -- undefined there.  This restatement (and the HIP kernel) define that case: samples
 * outside the padded plane are those of its nearest edge.  Returns `direct` when the window lies inside the plane, else fills
 * *tmp (caller frees) and sets *stride. */
static const uint8_t *window_or_clamped(const SvtHipPlaneDesc *rp, int x, int y, int w, int h, uint8_t **tmp, uint32_t *stride) {
    const int x0 = -(int)rp->org_x, x1 = (int)rp->width + rp->org_x - 1, y0 = -(int)rp->org_y, y1 = (int)rp->height + rp->org_y - 1;
    const int ww = w - 1 + 64, hh = h - 1 + 64;
    *tmp    = NULL;
    *stride = rp->stride_y;
    if (x >= x0 && x + ww - 1 <= x1 && y >= y0 && y + hh - 1 <= y1)
        return rp->buffer_y + ((int64_t)rp->org_x + x) + ((int64_t)rp->org_y + y) * rp->stride_y;
    *tmp = (uint8_t *)malloc((size_t)ww * hh);
    for (int r = 0; r < hh; r++)
        for (int c = 0; c < ww; c++) {
            const int cx = ORC_MIN(ORC_MAX(x + c, x0), x1), cy = ORC_MIN(ORC_MAX(y + r, y0), y1);
            (*tmp)[(size_t)r * ww + c] = rp->buffer_y[((int64_t)rp->org_x + cx) + ((int64_t)rp->org_y + cy) * rp->stride_y];
        }
    *stride = (uint32_t)ww;
    return *tmp;
}

static void integer_search(MeState *s) {
    const SvtHipMeConfig *c = s->cfg;
    const int pic_w = (int16_t)s->pic->aligned_width, pic_h = (int16_t)s->pic->aligned_height;
    const int org_x = (int16_t)s->org_x, org_y = (int16_t)s->org_y;
    for (int li = 0; li < s->pic->num_of_list_to_search; li++)
        for (int ri = 0; ri < s->pic->num_of_ref_pic_to_search[li]; ri++) {
            const SvtHipPlaneDesc *rp = &s->ref[li][ri]->lvl[2];
            if (!s->sr[li][ri].do_ref)
                continue;
            int16_t  cx = s->sr[li][ri].hme_sc_x, cy = s->sr[li][ri].hme_sc_y;
            uint16_t dist = (uint16_t)ref_distance(s, li, ri);
            if (c->me_type != 1) dist = scaled_distance(dist); /* me_type != ME_MCTF, :1299-1302 */
            int16_t  sa_w = (int16_t)ORC_MIN(c->me_sa.sa_min.width * dist, c->me_sa.sa_max.width);
            int16_t  sa_h = (int16_t)ORC_MIN(c->me_sa.sa_min.height * dist, c->me_sa.sa_max.height);
            if (c->mv_sa_adj_enabled && (!c->mv_sa_adj_nearest_ref_only || ri == 0)) {
                if (ORC_ABS(cx) > c->mv_sa_adj_mv_size_th) sa_w = (int16_t)(sa_w * c->mv_sa_adj_sa_multiplier);
                if (ORC_ABS(cy) > c->mv_sa_adj_mv_size_th) sa_h = (int16_t)(sa_h * c->mv_sa_adj_sa_multiplier);
            }
            sa_w = (int16_t)((ORC_MAX(1u, (uint32_t)sa_w / s->sr_divisor[li][ri]) + 7) & ~7u);
            sa_h = (int16_t)ORC_MAX(3u, (uint32_t)sa_h / s->sr_divisor[li][ri]);
            const int16_t h0 = sa_h, w0 = sa_w;
            uint64_t      best_hme_sad = ~(uint64_t)0;
            if (c->me_early_exit_th) {
                if (s->zz_sad[li][ri] < c->me_early_exit_th / 6)
                    sa_w = sa_h = 1;
            } else {
                int accurate = 1;
                if ((cx != 0 || cy != 0) && s->pic->is_ref) {
                    best_hme_sad = check_00_center(s, rp, &cx, &cy, s->zz_sad[li][ri]);
                    if (cx == 0 && cy == 0)
                        accurate = 0;
                }
                if (c->enable_me_sr_adjustment == 2) {
                    if ((accurate && best_hme_sad < 24 * 24) || (s->pic->is_ref && s->sr[li][ri].hme_sad < 24 * 24))
                        sa_h = (int16_t)(sa_h / 2);
                    if ((li || ri) && s->best_sad[0][0][0] < 5000 && sa_h == h0 && sa_w == w0) {
                        sa_h = (int16_t)(sa_h >> 1);
                        sa_w = (int16_t)(sa_w >> 1);
                    }
                }
            }
            orc_initialize_buffer_32bits(s->best_sad[li][ri], 21, 1, SVT_HIP_MAX_SAD_VALUE);
            uint8_t *tmp;
            uint32_t wstride;
            if (c->me_8x8_var_enabled && sa_w * sa_h > 24) {
                const uint8_t *win = window_or_clamped(rp, org_x + cx, org_y + cy, 1, 1, &tmp, &wstride);
                fullpel_search(s, li, ri, win, wstride, cx, cy, 1, 1);
                free(tmp);
                const uint32_t *b8   = s->best_sad[li][ri] + 21;
                const uint32_t  mean = s->best_sad[li][ri][0] / 64;
                uint32_t        ssq  = 0;
                for (int i = 0; i < 64; i++) {
                    const int32_t d = (int32_t)b8[i] - (int32_t)mean;
                    ssq += (uint32_t)(d * d);
                }
                const uint32_t var = ssq / 64;
                if (var > c->me_sr_mult2_th) {
                    sa_w = (int16_t)((ORC_MAX(1, sa_w * 3 / 2) + 7) & ~7);
                    sa_h = (int16_t)ORC_MAX(1, sa_h * 3 / 2);
                }
                if (var < c->me_sr_div4_th) {
                    sa_w = (int16_t)((ORC_MAX(1, sa_w >> 2) + 7) & ~7);
                    sa_h = (int16_t)ORC_MAX(3, ORC_MAX(1, sa_h >> 2));
                } else if (var < c->me_sr_div2_th) {
                    sa_w = (int16_t)((ORC_MIN(sa_w, sa_w >> 1) + 7) & ~7);
                    sa_h = (int16_t)ORC_MAX(3, ORC_MIN(sa_h, sa_h >> 1));
                }
            }
            int ox = cx - (sa_w >> 1), oy = cy - (sa_h >> 1), w = sa_w, h = sa_h;
            ox = (int16_t)ox;
            oy = (int16_t)oy;
            clip_axis(org_x, &ox, &w, 63, pic_w);
            w = (w < 8) ? w : (w & ~7);
            clip_axis(org_y, &oy, &h, 63, pic_h);
            const uint8_t *win = window_or_clamped(rp, org_x + ox, org_y + oy, w, h, &tmp, &wstride);
            fullpel_search(s, li, ri, win, wstride, (int16_t)ox, (int16_t)oy, (uint32_t)w, (uint32_t)h);
            free(tmp);
        }
}

/* me_prune_ref: Codec/motion_estimation.c:1522-1565 */
static void me_prune_ref(MeState *s) {
    const SvtHipMeConfig *c = s->cfg;
    for (int li = 0; li < s->pic->num_of_list_to_search; li++)
        for (int ri = 0; ri < s->pic->num_of_ref_pic_to_search[li]; ri++) {
            if (!s->sr[li][ri].do_ref) {
                s->sr[li][ri].hme_sad = (uint64_t)SVT_HIP_MAX_SAD_VALUE * 64;
                continue;
            }
            uint64_t t = 0;
            for (int i = 0; i < 64; i++) t += s->best_sad[li][ri][21 + i];
            s->sr[li][ri].hme_sad = t;
        }
    const uint16_t th = c->prune_ref_if_me_sad_dev_bigger_than_th;
    if (c->enable_me_hme_ref_pruning && th != 0xFFFF) {
        uint64_t best = ~(uint64_t)0;
        for (int li = 0; li < 2; li++)
            for (int ri = 0; ri < 4; ri++) best = ORC_MIN(best, s->sr[li][ri].hme_sad);
        for (int li = 0; li < 2; li++)
            for (int ri = 1; ri < 4; ri++)
                if ((s->sr[li][ri].hme_sad - best) * 100 > (uint64_t)th * best)
                    s->sr[li][ri].do_ref = 0;
    }
}

/* z_to_raster: Codec/motion_estimation.c:2520-2531 -- n_idx (quad-tree order) -> raster-within-depth PU index */
static const uint8_t k_z_to_raster[85] = {
    0,  1,  2,  3,  4,  5,  6,  9,  10, 7,  8,  11, 12, 13, 14, 17, 18, 15, 16, 19, 20, 21, 22, 29, 30, 23, 24, 31, 32,
    37, 38, 45, 46, 39, 40, 47, 48, 25, 26, 33, 34, 27, 28, 35, 36, 41, 42, 49, 50, 43, 44, 51, 52, 53, 54, 61, 62, 55,
    56, 63, 64, 69, 70, 77, 78, 71, 72, 79, 80, 57, 58, 65, 66, 59, 60, 67, 68, 73, 74, 81, 82, 75, 76, 83, 84};

/* MeCandidate bitfield (Codec/me_sb_results.h:28-34): direction:2 ref_idx_l0:2 ref_idx_l1:2 ref0_list:1 ref1_list:1 */
static uint8_t pack_cand(unsigned dir, unsigned i0, unsigned i1, unsigned l0, unsigned l1) {
    return (uint8_t)((dir & 3) | ((i0 & 3) << 2) | ((i1 & 3) << 4) | ((l0 & 1) << 6) | ((l1 & 1) << 7));
}

typedef struct SbOut {
    uint8_t  *total;
    uint32_t *mv;
    uint8_t  *cand;
} SbOut;

static int use_me_pu(const SvtHipMePictureDesc *p, int n_idx) {
    return p->enable_me_16x16 ? (p->enable_me_8x8 || n_idx < 21) : n_idx < 5;
}

/* construct_me_candidate_array_single_ref / _mrp_off / generic: Codec/motion_estimation.c:2646-2836 */
static void construct_candidates(MeState *s, SbOut *o) {
    const SvtHipMePictureDesc *p   = s->pic;
    const SvtHipMeConfig      *c   = s->cfg;
    uint32_t                   nls = p->num_of_list_to_search;
    const uint32_t n_pu_kept       = svt_hip_me_n_pu(p->enable_me_16x16, p->enable_me_8x8);
    const int      r0 = p->num_of_ref_pic_to_search[0], r1 = p->num_of_ref_pic_to_search[1];
    if (r0 == 1 && r1 == 0) {
        const uint8_t do_ref = s->sr[0][0].do_ref;
        memset(o->total, 1, n_pu_kept);
        for (int n = 0; n < p->max_number_of_pus_per_sb; n++) {
            const int pu         = k_z_to_raster[n];
            s->me_distortion[pu] = s->best_sad[0][0][n];
            if (!do_ref || !use_me_pu(p, n))
                continue;
            o->cand[pu * p->max_cand] = pack_cand(0, 0, 0, 0, 0);
            o->mv[pu * p->max_refs]   = s->best_mv[0][0][n];
        }
    } else if (r0 == 1 && r1 == 1) {
        const uint8_t d0 = s->sr[0][0].do_ref, d1 = (nls == 1) ? 0 : s->sr[1][0].do_ref;
        if (nls < 2 || !s->sr[1][0].do_ref)
            nls = 1;
        const uint32_t prune_th = (d0 && d1) ? (uint32_t)c->prune_me_candidates_th : 0;
        memset(o->total, 1, n_pu_kept);
        for (int n = 0; n < p->max_number_of_pus_per_sb; n++) {
            const int      pu  = k_z_to_raster[n];
            const int      use = use_me_pu(p, n);
            uint8_t        blk[2] = {d0, d1};
            uint8_t        off = 0;
            const uint32_t best = (d0 && d1) ? ORC_MIN(s->best_sad[0][0][n], s->best_sad[1][0][n])
                : d0                         ? s->best_sad[0][0][n]
                                             : s->best_sad[1][0][n];
            s->me_distortion[pu] = best;
            int min_list         = -1;
            if (c->use_best_unipred_cand_only && blk[0] && blk[1])
                min_list = s->best_sad[0][0][n] < s->best_sad[1][0][n] ? 0 : 1;
            for (uint32_t li = 0; li < nls && (use || off == 0); li++) {
                if (!blk[li])
                    continue;
                if (prune_th > 0) {
                    const uint32_t dev = (s->best_sad[li][0][n] - best) * 100;
                    if (dev > best * prune_th) {
                        blk[li] = 0;
                        continue;
                    }
                }
                if (min_list != -1 && min_list != (int)li) {
                    if (use)
                        o->mv[pu * p->max_refs + (li ? p->max_l0 : 0)] = s->best_mv[li][0][n];
                    continue;
                }
                if (use) {
                    o->cand[pu * p->max_cand + off]                = pack_cand(li, 0, 0, li == 0 ? li : 24, li == 1 ? li : 24);
                    o->mv[pu * p->max_refs + (li ? p->max_l0 : 0)] = s->best_mv[li][0][n];
                }
                off++;
            }
            if (blk[0] && blk[1] && use) {
                o->cand[pu * p->max_cand + off] = pack_cand(2 /* BI_PRED */, 0, 0, 0, 1);
                o->total[pu]                    = (uint8_t)(off + 1);
            }
        }
    } else {
        for (int n = 0; n < p->max_number_of_pus_per_sb; n++) {
            const int      pu  = (n > 4) ? k_z_to_raster[n] : n;
            const int      use = use_me_pu(p, n);
            uint8_t        off = 0;
            uint8_t        blk[2][4];
            const uint32_t prune_th = (uint32_t)c->prune_me_candidates_th;
            uint32_t       best     = ~0u;
            memset(blk, 0, sizeof(blk));
            for (uint32_t li = 0; li < nls; li++)
                for (int ri = 0; ri < p->num_of_ref_pic_to_search[li]; ri++) {
                    blk[li][ri] = s->sr[li][ri].do_ref;
                    if (blk[li][ri])
                        best = ORC_MIN(best, s->best_sad[li][ri][n]);
                }
            s->me_distortion[pu] = best;
            for (uint32_t li = 0; li < nls && (use || off == 0); li++)
                for (int ri = 0; ri < p->num_of_ref_pic_to_search[li] && (use || off == 0); ri++) {
                    if (!blk[li][ri])
                        continue;
                    if (prune_th > 0) {
                        const uint32_t dev = (s->best_sad[li][ri][n] - best) * 100;
                        if (dev > best * prune_th) {
                            blk[li][ri] = 0;
                            continue;
                        }
                    }
                    if (use) {
                        o->cand[pu * p->max_cand + off] = pack_cand(li, ri, ri, li == 0 ? li : 24, li == 1 ? li : 24);
                        o->mv[pu * p->max_refs + (li ? p->max_l0 : 0) + ri] = s->best_mv[li][ri][n];
                    }
                    off++;
                }
            if (nls == 2 && use) {
                for (int a = 0; a < r0; a++)
                    for (int b = 0; b < r1; b++) {
                        if (p->only_l_bwd && (a > 0 || b > 0))
                            continue;
                        if (blk[0][a] && blk[1][b])
                            o->cand[pu * p->max_cand + off++] = pack_cand(2, a, b, 0, 1);
                    }
                if (!p->only_l_bwd) {
                    for (int a = 1; a < r0; a++)
                        if (blk[0][0] && blk[0][a])
                            o->cand[pu * p->max_cand + off++] = pack_cand(2, 0, a, 0, 0);
                    if (r1 == 3 && blk[1][0] && blk[1][2])
                        o->cand[pu * p->max_cand + off++] = pack_cand(2, 0, 2, 1, 1);
                }
            }
            if (use)
                o->total[pu] = off;
        }
    }
}

static const uint8_t k_8x8_to_16x16[64] = {5,  5,  6,  6,  7,  7,  8,  8,  5,  5,  6,  6,  7,  7,  8,  8,  9,  9,  10, 10, 11, 11,
                                           12, 12, 9,  9,  10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 13, 13, 14, 14,
                                           15, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 17, 17, 18, 18, 19, 19, 20, 20};
static const uint8_t k_16x16_to_32x32[16] = {1, 1, 2, 2, 1, 1, 2, 2, 3, 3, 4, 4, 3, 3, 4, 4};

/* perform_gm_detection: Codec/motion_estimation.c:2838-2961 */
static void gm_detection(MeState *s, const SbOut *o, uint8_t *stationary, uint8_t *allow_gm) {
    const SvtHipMePictureDesc *p = s->pic;
    uint64_t cnt[2][4][2][2], tot = 0, still = 0;
    memset(cnt, 0, sizeof(cnt));
    const int low = p->input_resolution <= 2; /* INPUT_SIZE_480p_RANGE */
    const int n   = low ? 64 : 16;
    for (int i = 0; i < n; i++) {
        uint8_t idx = (uint8_t)((low ? 21 : 5) + i);
        if (low && !p->enable_me_8x8) {
            if (idx >= 21) idx = k_8x8_to_16x16[idx - 21];
            if (!p->enable_me_16x16 && idx >= 5) idx = k_16x16_to_32x32[idx - 5];
        } else if (!low && !p->enable_me_16x16 && idx >= 5)
            idx = k_16x16_to_32x32[idx - 5];
        const uint8_t  cb  = o->cand[idx * p->max_cand];
        const unsigned dir = cb & 3;
        const unsigned li  = (dir == 0 || dir == 2) ? ((cb >> 6) & 1) : ((cb >> 7) & 1);
        const unsigned ri  = (dir == 0 || dir == 2) ? ((cb >> 2) & 3) : ((cb >> 4) & 3);
        const uint64_t a = p->picture_number, b = p->ref_picture_number[li][ri];
        uint16_t       dist;
        int            th;
        if (low) {
            dist = (uint16_t)ORC_ABS((int16_t)(ORC_MAX(a, b) - ORC_MIN(a, b)));
            th   = p->gm_use_distance_based_active_th ? ORC_MAX(dist >> 1, 4) : 4;
        } else {
            dist = (uint16_t)ORC_ABS((int16_t)(a - b));
            th   = p->gm_use_distance_based_active_th ? ORC_MAX(dist * 16, 32) : 32;
        }
        const int mx = MVX(s->best_mv[li][ri][idx]) << 2, my = MVY(s->best_mv[li][ri][idx]) << 2;
        if (mx < -th) cnt[li][ri][0][0]++; else if (mx > th) cnt[li][ri][0][1]++;
        if (my < -th) cnt[li][ri][1][0]++; else if (my > th) cnt[li][ri][1][1]++;
        const int sth = low ? 0 : 4;
        if (abs(mx) <= sth && abs(my) <= sth) still++;
        tot++;
    }
    if (still > (tot * 5) / 100) *stationary = 1;
    for (int li = 0; li < 2; li++)
        for (int ri = 0; ri < 4; ri++)
            for (int cpt = 0; cpt < 2; cpt++)
                for (int sg = 0; sg < 2; sg++)
                    if (cnt[li][ri][cpt][sg] > tot / 2) *allow_gm = 1;
}

/* Whole-picture driver: the b64 loop of svt_aom_motion_estimation_kernel (Codec/me_process.c:174-290)
 * around svt_aom_motion_estimation_b64 (Codec/motion_estimation.c:3076-3153); cfg->me_type 1 = ME_MCTF. */
int orc_me_picture(const SvtHipMeConfig *cfg, const SvtHipMePictureDesc *desc, const SvtHipPlaneDesc cur_planes[3],
                   const SvtHipPlaneDesc ref_planes[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS][3], SvtHipMeResults *res) {
    OrcPyramid cur, refs[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
    MeState   *s = (MeState *)calloc(1, sizeof(MeState));
    if (!s)
        return SVT_HIP_ERR_NO_MEMORY;
    memcpy(cur.lvl, cur_planes, sizeof(cur.lvl));
    s->cfg = cfg;
    s->pic = desc;
    s->cur = &cur;
    for (int li = 0; li < SVT_HIP_MAX_LISTS; li++)
        for (int ri = 0; ri < SVT_HIP_MAX_REFS; ri++) {
            memcpy(refs[li][ri].lvl, ref_planes[li][ri], sizeof(cur.lvl));
            s->ref[li][ri] = &refs[li][ri];
        }
    const uint32_t w64 = (desc->aligned_width + 63) / 64, h64 = (desc->aligned_height + 63) / 64;
    const uint32_t row0 = desc->b64_row_start, nrow = desc->b64_row_count ? desc->b64_row_count : h64 - row0;
    const uint32_t n_pu = svt_hip_me_n_pu(desc->enable_me_16x16, desc->enable_me_8x8);
    t_sad_ops = 0;
    for (int i = 0; i < ORC_N_STAGES; i++) t_stage_ops[i] = 0;
    /* desc->aligned_* == ALIGN_POWER_OF_TWO(input width/height, 3) (motion_estimation.c:3093-3094, pcs.c:1496-1497) */
    const uint16_t aw = desc->aligned_width, ah = desc->aligned_height;
    for (uint32_t by = row0; by < row0 + nrow && by < h64; by++)
        for (uint32_t bx = 0; bx < w64; bx++) {
            const uint32_t b = bx + by * w64;
            s->org_x         = bx * 64;
            s->org_y         = by * 64;
            s->b64_w         = (uint32_t)(aw - s->org_x) < 64 ? aw - s->org_x : 64;
            s->b64_h         = (uint32_t)(ah - s->org_y) < 64 ? ah - s->org_y : 64;
            for (int l = 0; l < 3; l++) {
                const SvtHipPlaneDesc *pl = &cur.lvl[l];
                const int              sh = 2 - l;
                s->src[l]        = pl->buffer_y + (size_t)(pl->org_y + (s->org_y >> sh)) * pl->stride_y + pl->org_x + (s->org_x >> sh);
                s->src_stride[l] = pl->stride_y;
            }
            /* init_me_hme_data: Codec/motion_estimation.c:3010-3071 */
            memset(s->hme_x, 0, sizeof(s->hme_x));
            memset(s->hme_y, 0, sizeof(s->hme_y));
            memset(s->best_mv, 0, sizeof(s->best_mv));
            memset(s->performed_phme, 0, sizeof(s->performed_phme));
            for (int li = 0; li < 2; li++)
                for (int ri = 0; ri < 4; ri++) {
                    s->sr[li][ri].do_ref    = 1;
                    s->sr[li][ri].hme_sad   = 0xFFFFFFFFu;
                    s->sr_divisor[li][ri]   = 1;
                    s->zz_sad[li][ri]       = ~0u;
                    s->prehme[li][ri][0].valid = s->prehme[li][ri][1].valid = 0;
                }
            s->hme_l0_sa = cfg->hme_l0_sa;
            /* hme_b64: Codec/motion_estimation.c:2441-2475 */
            t_stage = 0;
            if (cfg->me_early_exit_th || cfg->me_safe_limit_zz_th)
                init_zz_sad(s);
            t_stage = 1;
            if (cfg->prehme_enable)
                prehme_b64(s);
            if (cfg->enable_hme_flag) {
                t_stage = 2;
                if (cfg->enable_hme_level0_flag) hme_level0_b64(s);
                t_stage = 3;
                if (cfg->enable_hme_level1_flag) hme_level12_b64(s, 1);
                t_stage = 4;
                if (cfg->enable_hme_level2_flag) hme_level12_b64(s, 2);
            }
            t_stage = 5;
            set_final_search_centre(s);
            const int mctf    = cfg->me_type == 1;
            const int tf_exit = mctf && s->sr[0][0].hme_sad < desc->tf_me_exit_th; /* :3109-3113 */
            if (!tf_exit) {
                if (cfg->enable_hme_flag && !mctf) /* prune_ref, :3103,3115 */
                    hme_prune_and_adjust(s);
                integer_search(s);
                if (cfg->enable_hme_flag && !mctf && cfg->enable_me_hme_ref_pruning)
                    me_prune_ref(s);
            } else /* the reference leaves p_sb_best_sad stale here; canonical value = never searched */
                for (int li = 0; li < 2; li++)
                    for (int ri = 0; ri < 4; ri++)
                        for (int n = 0; n < 85; n++) s->best_sad[li][ri][n] = SVT_HIP_MAX_SAD_VALUE;
            if (mctf) goto search_level_results; /* :3126 */
            SbOut o = {res->total_me_candidate_index + (size_t)b * n_pu, res->me_mv_array + (size_t)b * n_pu * desc->max_refs,
                       res->me_candidate_array + (size_t)b * n_pu * desc->max_cand};
            construct_candidates(s, &o);
            /* compute_distortion: Codec/motion_estimation.c:2964-3008 */
            uint32_t d32 = 0, d16 = 0, d8 = 0;
            for (int i = 0; i < 4; i++) d32 += s->me_distortion[1 + i];
            for (int i = 0; i < 16; i++) d16 += s->me_distortion[5 + i];
            for (int i = 0; i < 64; i++) d8 += s->me_distortion[21 + i];
            const uint64_t mean = d8 / 64;
            uint64_t       ssq  = 0;
            for (int i = 0; i < 64; i++) {
                const int64_t d = (int64_t)s->me_distortion[21 + i] - (int64_t)mean;
                ssq += (uint64_t)(d * d);
            }
            /* B64Geom width/height come from the aligned picture size: pcs.c:1513-1518 */
            const uint32_t pix = s->b64_w * s->b64_h;
            res->me_8x8_cost_variance[b] = (uint32_t)(ssq / 64);
            res->rc_me_distortion[b]     = desc->input_resolution <= 2 ? d8 : d16;
            res->me_64x64_distortion[b]  = (s->me_distortion[0] * 4096u) / pix;
            res->me_32x32_distortion[b]  = (d32 * 4096u) / pix;
            res->me_16x16_distortion[b]  = (d16 * 4096u) / pix;
            res->me_8x8_distortion[b]    = (d8 * 4096u) / pix;
            res->stationary_block_present_sb[b] = 0;
            res->rc_me_allow_gm[b]              = 0;
            if (desc->gm_enabled)
                gm_detection(s, &o, &res->stationary_block_present_sb[b], &res->rc_me_allow_gm[b]);
        search_level_results:
            /* optional search-level results, canonicalised for refs that were not searched */
            for (int li = 0; li < 2; li++)
                for (int ri = 0; ri < 4; ri++) {
                    const int    live = li < desc->num_of_list_to_search && ri < desc->num_of_ref_pic_to_search[li];
                    const size_t k    = ((size_t)b * 2 + li) * 4 + ri;
                    if (res->do_ref) res->do_ref[k] = live ? s->sr[li][ri].do_ref : 0;
                    if (res->hme_sc) {
                        res->hme_sc[2 * k]     = live ? s->sr[li][ri].hme_sc_x : 0;
                        res->hme_sc[2 * k + 1] = live ? s->sr[li][ri].hme_sc_y : 0;
                    }
                    if (res->hme_sad) res->hme_sad[k] = live ? (uint32_t)s->sr[li][ri].hme_sad : 0;
                    for (int n = 0; n < 85; n++) {
                        const int ok = live && s->sr[li][ri].do_ref;
                        if (res->sb_best_sad) res->sb_best_sad[k * 85 + n] = ok ? s->best_sad[li][ri][n] : SVT_HIP_MAX_SAD_VALUE;
                        if (res->sb_best_mv) res->sb_best_mv[k * 85 + n] = ok ? s->best_mv[li][ri][n] : 0;
                    }
                }
        }
    free(s);
    __atomic_fetch_add(&g_sad_ops, t_sad_ops, __ATOMIC_RELAXED);
    for (int i = 0; i < ORC_N_STAGES; i++) __atomic_fetch_add(&g_stage_ops[i], t_stage_ops[i], __ATOMIC_RELAXED);
    t_stage = 0;
    return SVT_HIP_OK;
}

/* ABI self-description so the ctypes mirror can be checked against the compiled layout */
size_t orc_sizeof(int what) {
    switch (what) {
    case 0: return sizeof(SvtHipMeConfig);
    case 1: return sizeof(SvtHipMePictureDesc);
    case 2: return sizeof(SvtHipPlaneDesc);
    case 3: return sizeof(SvtHipMeResults);
    case 4: return sizeof(SvtHipMePresetDesc);
    case 5: return sizeof(SvtHipDgMetrics);
    default: return 0;
    }
}
