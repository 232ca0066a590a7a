// me_config.cpp -- host-only: derives the open-loop ME search controls of a preset.
//
// Restates, for TASK_PAME pictures, what the reference computes in svt_aom_sig_deriv_me
// (Source/Lib/Codec/enc_mode_config.c:681-833) and the helpers it calls (set_hme_search_params :138-218,
// set_me_search_params :223-350, *_ctrls setters :351-596), plus the four HME enable flags of
// svt_aom_sig_deriv_multi_processes (:1634-1645).  Checked field-by-field against the reference build for every
// preset / resolution / class combination in tests/test_me_config.py.
#include <string.h>
#include "../../include/svt_hip_me.h"

namespace {

enum { MRS = -3, MRP = -2, MR = -1 };
enum { RES_240 = 0, RES_360, RES_480, RES_720, RES_1080, RES_4K, RES_8K };

struct Area { uint16_t w, h; };

inline SvtHipSearchArea sa(uint16_t w, uint16_t h) { SvtHipSearchArea a = {w, h}; return a; }
inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
inline uint16_t umax(uint16_t a, uint16_t b) { return a > b ? a : b; }

void hme_areas(const SvtHipMePresetDesc &p, SvtHipMeConfig &c) {
    const int  m = p.enc_mode, res = p.input_resolution;
    const bool rtc = p.rtc_tune, sc = p.sc_class1;
    c.num_hme_sa_w = c.num_hme_sa_h = 2;
    int  q_mult = 0;
    Area lo = {32, 32}, hi = {192, 192};
    if (m <= MRS) {
        if (res < RES_4K) { lo = {128, 128}; hi = {256, 256}; } else { lo = {240, 240}; hi = {480, 480}; }
    } else if (m <= 1) {
        if (res >= RES_4K) { lo = {240, 240}; hi = {480, 480}; }
    } else if (m <= 3) {
    } else if (m <= 6) {
        q_mult = 3;
    } else if (!rtc && m <= 7) {
        if (!(sc || res >= RES_4K)) lo = {16, 16};
        q_mult = 3;
    } else if ((!rtc && m <= 9) || (rtc && m <= 7)) {
        if (!sc) lo = {16, 16};
        q_mult = 3;
    } else {
        if (!sc) {
            if (res < RES_4K) { lo = {8, 8}; hi = {96, 96}; } else { lo = {16, 16}; hi = {96, 96}; }
        }
        q_mult = 3;
    }
    if (q_mult) {
        const int qw = clip3(500, 1000, q_mult * ((8 * (int)p.qp) - 125));
        lo.w = umax(8, (uint16_t)((lo.w * qw) / 1000));   lo.h = umax(8, (uint16_t)((lo.h * qw) / 1000));
        hi.w = umax(96, (uint16_t)((hi.w * qw) / 1000));  hi.h = umax(96, (uint16_t)((hi.h * qw) / 1000));
    }
    c.hme_l0_sa.sa_min = sa(lo.w, lo.h);
    c.hme_l0_sa.sa_max = sa(hi.w, hi.h);
    c.hme_l1_sa = c.hme_l2_sa = (m <= MR) ? sa(16, 16) : sa(8, 3);
}

void me_areas(const SvtHipMePresetDesc &p, SvtHipMeConfig &c) {
    const int  m = p.enc_mode, res = p.input_resolution;
    const bool rtc = p.rtc_tune, sc = p.sc_class1;
    int  q_mult = 0;
    Area lo, hi;
    if (rtc) {
        if (sc) {
            if (m <= 7) { lo = {32, 32}; hi = {96, 96}; }
            else if (m <= 9) { if (res < RES_1080) { lo = {16, 16}; hi = {32, 16}; } else { lo = {24, 24}; hi = {24, 24}; } }
            else { if (res < RES_1080) { lo = {16, 16}; hi = {32, 16}; } else { lo = {16, 6}; hi = {16, 9}; } }
        } else if (m <= 8) {
            if (res < RES_1080) { lo = {16, 16}; hi = {32, 16}; } else { lo = {16, 6}; hi = {16, 9}; }
        } else {
            if (res < RES_720) { lo = {8, 3}; hi = {16, 9}; }
            else if (res < RES_1080) { lo = {8, 1}; hi = {16, 7}; }
            else if (res < RES_4K) { lo = {8, 1}; hi = {8, 7}; }
            else { lo = {8, 1}; hi = {8, 1}; }
        }
    } else if (sc) {
        if (m <= 1) { lo = {175, 175}; hi = {750, 750}; }
        else if (m <= 6) { lo = {48, 48}; hi = {224, 224}; }
        else if (m <= 7) { lo = {32, 32}; hi = {164, 164}; }
        else if (m <= 8) { lo = {32, 32}; hi = {96, 96}; }
        else if (m <= 9) { lo = {16, 16}; hi = {96, 96}; }
        else { lo = {8, 8}; hi = {32, 32}; }
    } else if (m <= 1) { lo = {64, 64}; hi = {256, 256}; }
    else if (m <= 2) { lo = {32, 32}; hi = {128, 128}; }
    else if (m <= 4) { lo = {24, 24}; hi = {104, 104}; }
    else if (m <= 6) { lo = {16, 16}; hi = {64, 32}; q_mult = 7; }
    else if (m <= 9) {
        if (p.hierarchical_levels <= 3) {
            if (res < RES_4K) { lo = {8, 5}; hi = {16, 9}; } else { lo = {8, 1}; hi = {8, 1}; }
        } else if (res < RES_1080) { lo = {16, 16}; hi = {32, 16}; }
        else { lo = {16, 6}; hi = {16, 9}; }
        q_mult = 7;
    } else { lo = {16, 6}; hi = {16, 6}; q_mult = 6; }
    if (q_mult) {
        const int qw = clip3(500, 1000, (q_mult * ((31 * (int)p.qp) - 700)) >> 3);
        lo.w = umax(8, (uint16_t)((lo.w * qw) / 1000)); lo.h = umax(3, (uint16_t)((lo.h * qw) / 1000));
        hi.w = umax(8, (uint16_t)((hi.w * qw) / 1000)); hi.h = umax(3, (uint16_t)((hi.h * qw) / 1000));
    }
    if (p.frame_rate_q16 >> 16) { // "low_frame_rate_flag" as written in the reference (true for >= 1 fps)
        lo.w = (uint16_t)((lo.w * 3) >> 1);
        lo.h = (uint16_t)((lo.h * 3) >> 1);
    }
    c.me_sa.sa_min = sa(lo.w, lo.h);
    c.me_sa.sa_max = sa(hi.w, hi.h);
}

void prehme_level(int level, SvtHipMeConfig &c) {
    static const uint16_t tab[5][10] = {
        // v.min(w,h) v.max(w,h) h.min(w,h) h.max(w,h) skip_line l1_exit
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
        {8, 144, 8, 496, 144, 3, 496, 3, 0, 0},
        {8, 100, 8, 400, 96, 3, 384, 3, 0, 0},
        {8, 100, 8, 350, 32, 7, 200, 7, 1, 0},
        {8, 100, 8, 350, 32, 7, 128, 7, 1, 1},
    };
    c.prehme_enable = level ? 1 : 0;
    if (!level) return;
    const uint16_t *t = tab[level];
    c.prehme_sa_cfg[0].sa_min = sa(t[0], t[1]); c.prehme_sa_cfg[0].sa_max = sa(t[2], t[3]);
    c.prehme_sa_cfg[1].sa_min = sa(t[4], t[5]); c.prehme_sa_cfg[1].sa_max = sa(t[6], t[7]);
    c.prehme_skip_search_line = (uint8_t)t[8];
    c.prehme_l1_early_exit    = (uint8_t)t[9];
}

void prune_level(int level, SvtHipMeConfig &c) {
    static const uint16_t hme_th[7] = {0xFFFF, 80, 50, 30, 15, 5, 5};
    c.enable_me_hme_ref_pruning               = level ? 1 : 0;
    c.prune_ref_if_hme_sad_dev_bigger_than_th = hme_th[level];
    c.prune_ref_if_me_sad_dev_bigger_than_th  = level >= 4 ? 60 : 0xFFFF;
    c.zz_sad_th = c.phme_sad_th = 0;
    c.zz_sad_pct = c.phme_sad_pct = 0;
    if (level == 6) { c.zz_sad_th = 20 * 64 * 64; c.zz_sad_pct = 5; c.phme_sad_th = 10 * 64 * 64; c.phme_sad_pct = 5; }
}

void sr_adjust_level(int level, SvtHipMeConfig &c) {
    static const uint16_t tab[6][5] = { // enable, mv_len_th, stationary_sad_th, low_sad_th, dist_resize
        {0, 0, 0, 0, 0}, {1, 4, 12000, 6000, 0}, {1, 4, 12000, 6000, 1}, {1, 4, 12000, 12000, 1}, {2, 16, 20000, 20000, 1}, {2, 20, 24000, 24000, 1}};
    c.enable_me_sr_adjustment = (uint8_t)tab[level][0];
    if (level) {
        c.reduce_me_sr_based_on_mv_length_th   = tab[level][1];
        c.stationary_hme_sad_abs_th            = tab[level][2];
        c.stationary_me_sr_divisor             = 8;
        c.reduce_me_sr_based_on_hme_sad_abs_th = tab[level][3];
        c.me_sr_divisor_for_low_hme_sad        = 8;
        c.distance_based_hme_resizing          = (uint8_t)tab[level][4];
    }
    if (!c.enable_hme_level2_flag) { // thresholds are in full-resolution SAD units
        const int div = c.enable_hme_level1_flag ? 4 : 16;
        c.stationary_hme_sad_abs_th            = (uint16_t)(c.stationary_hme_sad_abs_th / div);
        c.reduce_me_sr_based_on_hme_sad_abs_th = (uint16_t)(c.reduce_me_sr_based_on_hme_sad_abs_th / div);
    }
}

} // namespace

extern "C" {

uint8_t svt_hip_input_resolution(uint32_t width, uint32_t height) {
    // svt_aom_derive_input_resolution, Codec/sequence_control_set.c:113-131 (thresholds definitions.h:2051-2057)
    static const uint32_t th[6] = {0x28500, 0x4CE00, 0xA1400, 0x16DA00, 0x535200, 0x140A000};
    const uint32_t        sz    = width * height;
    uint8_t               r     = 0;
    while (r < 6 && sz >= th[r]) r++;
    return r;
}

uint8_t svt_hip_enable_me_8x8(int8_t enc_mode, uint8_t rtc_tune, uint8_t input_resolution) {
    if (rtc_tune) return enc_mode <= 7;
    if (enc_mode <= 5) return 1;
    if (enc_mode <= 8) return input_resolution <= RES_720;
    return 0;
}

int svt_hip_me_config_from_preset(const SvtHipMePresetDesc *pp, SvtHipMeConfig *cfg) {
    if (!pp || !cfg || pp->enc_mode < MRS || pp->enc_mode > 13 || pp->input_resolution > RES_8K) return SVT_HIP_ERR_BAD_PARAM;
    const SvtHipMePresetDesc &p = *pp;
    SvtHipMeConfig           &c = *cfg;
    memset(&c, 0, sizeof(c));
    const int  m = p.enc_mode;
    const bool rtc = p.rtc_tune, sc = p.sc_class1, is_base = p.temporal_layer_index == 0;
    me_areas(p, c);
    hme_areas(p, c);
    c.enable_hme_flag = c.enable_hme_level0_flag = c.enable_hme_level1_flag = 1;
    c.enable_hme_level2_flag = (sc || m <= 6) ? 1 : 0;
    c.hme_search_method = c.me_search_method = 0; // SUB_SAD_SEARCH
    if (rtc) {
        if (sc) { c.reduce_hme_l0_sr_th_min = 8; c.reduce_hme_l0_sr_th_max = 100; }
        else if (m > 7) { c.reduce_hme_l0_sr_th_min = 8; c.reduce_hme_l0_sr_th_max = 200; }
    }
    int ph = 0;
    if (m <= MRS) ph = 1;
    else if (sc) ph = rtc ? 1 : 2;
    else if (rtc) ph = m <= 9 ? 4 : 0;
    else ph = m <= 7 ? 2 : 4;
    if (!c.enable_hme_level1_flag) ph = 0;
    prehme_level(ph, c);
    int pr;
    if (sc) pr = m <= MRS ? 0 : m <= 2 ? 1 : m <= 7 ? 3 : 6;
    else if (m <= MRS) pr = 0;
    else if (m <= MR) pr = 1;
    else if (m <= 0) pr = is_base ? 1 : 2;
    else if (m <= 1) pr = is_base ? 1 : 4;
    else if (m <= 3) pr = is_base ? 1 : 5;
    else if (m <= 9) pr = is_base ? 1 : 6;
    else pr = 6;
    prune_level(pr, c);
    sr_adjust_level(sc ? (m <= 7 ? 4 : 5) : (m <= MR ? 0 : m <= 0 ? 1 : 3), c);
    const int mv_adj = m <= MRS ? 1 : m <= 3 ? 2 : 0;
    c.mv_sa_adj_enabled = mv_adj ? 1 : 0;
    if (mv_adj) { c.mv_sa_adj_nearest_ref_only = mv_adj == 2; c.mv_sa_adj_mv_size_th = 25; c.mv_sa_adj_sa_multiplier = 2; }
    c.me_8x8_var_enabled = 1; // level 2
    c.me_sr_div4_th = 80000; c.me_sr_div2_th = 150000; c.me_sr_mult2_th = 0xFFFFFFFFu;
    c.prune_me_candidates_th     = m <= 6 ? 0 : 65;
    c.use_best_unipred_cand_only = m <= 3 ? 0 : 1; // pcs->use_best_me_unipred_cand_only, enc_mode_config.c:1845-1848
    if (rtc) c.me_early_exit_th = sc ? 64 * 64 : (m <= 9 ? 64 * 64 * 8 : 64 * 64 * 9);
    else c.me_early_exit_th = m <= 4 ? 0 : 64 * 64 * 8;
    c.me_safe_limit_zz_th         = p.safe_limit_nref == 1 ? p.safe_limit_zz_th : 0;
    c.prev_me_stage_based_exit_th = (rtc && sc) ? 64 * 64 * 4 : 0;
    return SVT_HIP_OK;
}

} // extern "C"
