// me_picture.hip -- host entry points of the batched open-loop ME (include/svt_hip_me.h): descriptor validation,
// parameter block assembly, launch, and the synchronous host-pointer form.
#include <hip/hip_runtime.h>
#include <string.h>
#include <mutex>
#include "svt_hip_internal.h"

#include <stddef.h>
// the two MCTF fields sit in what used to be padding: the layout of both descriptors (and every stored fixture) is unchanged
static_assert(offsetof(SvtHipMeConfig, me_type) == 37 && offsetof(SvtHipMeConfig, prehme_sa_cfg) == 38 && sizeof(SvtHipMeConfig) == 128, "SvtHipMeConfig layout");
static_assert(offsetof(SvtHipMePictureDesc, tf_me_exit_th) == 36 && offsetof(SvtHipMePictureDesc, ref_picture_number) == 40 && sizeof(SvtHipMePictureDesc) == 104,
              "SvtHipMePictureDesc layout");

namespace {

struct Geometry {
    uint32_t w64, h64, row0, nrow, n_pu, n_b64;
};

int validate(SvtHipContext *ctx, const SvtHipMeConfig *cfg, const SvtHipMePictureDesc *d, const SvtHipPaPicture *cur,
             const SvtHipPaPicture *const refs[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS], const SvtHipMeResults *res, Geometry *g) {
    if (!ctx || !cfg || !d || !cur || !refs || !res) return SVT_HIP_ERR_BAD_PARAM;
    if (cfg->num_hme_sa_w != 2 || cfg->num_hme_sa_h != 2)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "num_hme_sa_w/h must be 2x2 (got %ux%u)", cfg->num_hme_sa_w, cfg->num_hme_sa_h);
    if (cfg->hme_search_method > 1 || cfg->me_search_method > 1) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "search method must be 0 or 1");
    if (cfg->me_type > 1) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "me_type %u (0 = open loop, 1 = ME_MCTF)", cfg->me_type);
    if (cfg->me_type == 1 && d->tf_me_exit_th > 0xFFFF) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "tf_me_exit_th %u does not fit the reference's uint16", d->tf_me_exit_th);
    if (cfg->me_type == 1 && (!res->sb_best_sad || !res->sb_best_mv || !res->hme_sad || !res->hme_sc))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "ME_MCTF returns its results through sb_best_sad / sb_best_mv / hme_sad / hme_sc");
    const uint16_t lim = 2048; // keeps every int16 search-area product of the reference in range
    if (cfg->me_sa.sa_max.width > lim || cfg->me_sa.sa_max.height > lim || cfg->hme_l0_sa.sa_max.width > lim ||
        cfg->hme_l0_sa.sa_max.height > lim || cfg->prehme_sa_cfg[0].sa_max.height > lim || cfg->prehme_sa_cfg[1].sa_max.width > lim ||
        cfg->hme_l1_sa.width > lim || cfg->hme_l1_sa.height > lim || cfg->hme_l2_sa.width > lim || cfg->hme_l2_sa.height > lim)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "search area larger than %u", lim);
    if (d->num_of_list_to_search < 1 || d->num_of_list_to_search > SVT_HIP_MAX_LISTS)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "num_of_list_to_search %u", d->num_of_list_to_search);
    const DevPlane &f = cur->pyr.lvl[2];
    if (d->aligned_width != f.width || d->aligned_height != f.height || (f.width & 7) || (f.height & 7))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "aligned size %ux%u does not match the picture %dx%d", d->aligned_width,
                            d->aligned_height, f.width, f.height);
    int r[2] = {0, 0};
    for (int li = 0; li < d->num_of_list_to_search; li++) {
        r[li] = d->num_of_ref_pic_to_search[li];
        if (r[li] > SVT_HIP_MAX_REFS || (li == 0 && r[li] < 1)) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "list %d: %d references", li, r[li]);
        for (int ri = 0; ri < r[li]; ri++) {
            const SvtHipPaPicture *rp = refs[li][ri];
            if (!rp) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "reference [%d][%d] is null", li, ri);
            for (int l = 0; l < 3; l++)
                if (rp->pyr.lvl[l].width != cur->pyr.lvl[l].width || rp->pyr.lvl[l].height != cur->pyr.lvl[l].height)
                    return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "reference [%d][%d] level %d is %dx%d, picture is %dx%d", li, ri, l,
                                        rp->pyr.lvl[l].width, rp->pyr.lvl[l].height, cur->pyr.lvl[l].width, cur->pyr.lvl[l].height);
        }
    }
    if (d->max_number_of_pus_per_sb == 0 || d->max_number_of_pus_per_sb > SVT_HIP_SQUARE_PU_COUNT)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "max_number_of_pus_per_sb %u", d->max_number_of_pus_per_sb);
    // capacity of MeSbResults rows as allocated by svt_aom_me_sb_results_ctor (pcs.c:91-117)
    const int r1     = d->num_of_list_to_search == 2 ? d->num_of_ref_pic_to_search[1] : 0;
    const int ncand  = (r[0] == 1 && r1 <= 1) ? (r1 ? 3 : 1) : r[0] + r1 + r[0] * r1 + (r[0] - 1) + (r1 == 3 ? 1 : 0);
    if (d->max_cand > 32 || d->max_refs > SVT_HIP_MAX_LISTS * SVT_HIP_MAX_REFS) // MAX_PA_ME_CAND is 23, MAX_PA_ME_MV 7 (me_sb_results.h:22-23)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "max_cand %u / max_refs %u exceed what one block's result rows may take", d->max_cand, d->max_refs);
    if (d->max_cand < ncand || d->max_l0 < r[0] || d->max_refs < d->max_l0 + r1 || d->max_refs < 1)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "max_cand %u / max_refs %u / max_l0 %u too small for %d+%d references", d->max_cand,
                            d->max_refs, d->max_l0, r[0], r1);
    g->w64  = (d->aligned_width + 63u) / 64u;
    g->h64  = (d->aligned_height + 63u) / 64u;
    g->row0 = d->b64_row_start;
    if (g->row0 >= g->h64) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "b64_row_start %u >= %u rows", g->row0, g->h64);
    g->nrow = d->b64_row_count ? d->b64_row_count : g->h64 - g->row0;
    if (g->row0 + g->nrow > g->h64) g->nrow = g->h64 - g->row0;
    g->n_pu  = svt_hip_me_n_pu(d->enable_me_16x16, d->enable_me_8x8);
    g->n_b64 = g->w64 * g->h64;
    if (!res->total_me_candidate_index || !res->me_mv_array || !res->me_candidate_array || !res->me_64x64_distortion ||
        !res->me_32x32_distortion || !res->me_16x16_distortion || !res->me_8x8_distortion || !res->rc_me_distortion ||
        !res->me_8x8_cost_variance || !res->stationary_block_present_sb || !res->rc_me_allow_gm)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "a mandatory result pointer is null");
    return SVT_HIP_OK;
}

struct Field { size_t off, elem, count; bool optional; };

// layout of one result set inside a single device allocation (used by the synchronous entry)
size_t layout(const Geometry &g, const SvtHipMePictureDesc *d, const SvtHipMeResults *host, SvtHipMeResults *dev, uint8_t *base,
              Field f[16]) {
    const size_t nb = g.n_b64;
    const size_t counts[16] = {g.n_pu, (size_t)g.n_pu * d->max_refs, (size_t)g.n_pu * d->max_cand, 1, 1, 1, 1, 1, 1, 1, 1, 680, 680, 16, 8, 8};
    const size_t elems[16]  = {1, 4, 1, 4, 4, 4, 4, 4, 4, 1, 1, 4, 4, 2, 4, 1};
    void *const *hp = reinterpret_cast<void *const *>(host);
    void       **dp = reinterpret_cast<void **>(dev);
    size_t       off = 0;
    for (int i = 0; i < 16; i++) {
        f[i].elem = elems[i]; f[i].count = counts[i] * nb; f[i].optional = i >= 11; f[i].off = off;
        if (hp[i]) { dp[i] = base ? base + off : nullptr; off += (f[i].count * f[i].elem + 255) & ~(size_t)255; }
        else dp[i] = nullptr;
    }
    return off;
}

} // namespace

// The launch of `n_pictures` pictures on an explicit lane, whose holder is the caller.
int svt_hip_me_pictures_on_lane(SvtHipContext *ctx, SvtHipLane *lane, uint32_t n_pictures, const SvtHipMeJob *jobs) {
    if (!ctx || !lane || !jobs || n_pictures == 0) return SVT_HIP_ERR_BAD_PARAM;
    if (n_pictures > SVT_HIP_ME_MAX_PICTURES) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "%u pictures in one call (at most %d)", n_pictures, SVT_HIP_ME_MAX_PICTURES);
    static thread_local MeKernelParams params[SVT_HIP_ME_MAX_PICTURES];
    uint32_t n_jobs[SVT_HIP_ME_MAX_PICTURES];
    for (uint32_t i = 0; i < n_pictures; i++) {
        const SvtHipMeJob &j = jobs[i];
        Geometry g;
        int rc = validate(ctx, j.cfg, j.desc, j.cur, j.refs, j.results, &g);
        if (rc) return rc;
        MeKernelParams &p = params[i];
        memset(&p, 0, sizeof(p));
        dev_me_config(p.cfg, *j.cfg);
        dev_me_desc(p.desc, *j.desc);
        p.cur = j.cur->pyr;
        for (int li = 0; li < j.desc->num_of_list_to_search; li++)
            for (int ri = 0; ri < j.desc->num_of_ref_pic_to_search[li]; ri++) p.ref[li][ri] = j.refs[li][ri]->pyr;
        p.res  = *j.results;
        p.w64  = g.w64;
        p.row0 = g.row0;
        p.n_pu = g.n_pu;
        n_jobs[i] = g.nrow * g.w64;
    }
    hipSetDevice(ctx->device);
    // pictures whose planes were filled on another stream (svt_hip_pa_picture_create* enqueue on the context stream)
    for (uint32_t i = 0; i < n_pictures; i++) {
        const SvtHipMeJob &j = jobs[i];
        if (int rc = svt_hip_wait_picture(ctx, lane->stream, j.cur)) return rc;
        for (int li = 0; li < j.desc->num_of_list_to_search; li++)
            for (int ri = 0; ri < j.desc->num_of_ref_pic_to_search[li]; ri++)
                if (int rc = svt_hip_wait_picture(ctx, lane->stream, j.refs[li][ri])) return rc;
    }
    return svt_hip_me_launch(ctx, lane, params, n_jobs, n_pictures);
}

extern "C" {

int svt_hip_me_pictures_async(SvtHipContext *ctx, uint32_t n_pictures, const SvtHipMeJob *jobs) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    std::lock_guard<std::mutex> lk(ctx->async_mu); // lane 0's queue counters and parameter ring
    return svt_hip_me_pictures_on_lane(ctx, &ctx->lane[0], n_pictures, jobs);
}

int svt_hip_me_picture_async(SvtHipContext *ctx, const SvtHipMeConfig *cfg, const SvtHipMePictureDesc *desc, const SvtHipPaPicture *cur,
                             const SvtHipPaPicture *const refs[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS], const SvtHipMeResults *res_dev) {
    if (!refs) return SVT_HIP_ERR_BAD_PARAM;
    SvtHipMeJob job;
    job.cfg = cfg; job.desc = desc; job.cur = cur; job.results = res_dev;
    for (int li = 0; li < SVT_HIP_MAX_LISTS; li++)
        for (int ri = 0; ri < SVT_HIP_MAX_REFS; ri++) job.refs[li][ri] = refs[li][ri];
    return svt_hip_me_pictures_async(ctx, 1, &job);
}

int svt_hip_me_picture(SvtHipContext *ctx, const SvtHipMeConfig *cfg, const SvtHipMePictureDesc *desc, const SvtHipPaPicture *cur,
                       const SvtHipPaPicture *const refs[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS], SvtHipMeResults *res) {
    Geometry g;
    int rc = validate(ctx, cfg, desc, cur, refs, res, &g);
    if (rc) return rc;
    hipSetDevice(ctx->device);
    // a lane of its own for the duration of the call: stream, queue counters, parameter block and result buffer are not shared
    // with calls made from other host threads (Globals/enc_handle.c:2265: several ME threads, several pictures in flight)
    SvtHipLaneGuard guard(ctx);
    SvtHipLane     *lane = guard.lane();
    if (!lane) return SVT_HIP_ERR_NO_MEMORY;
    Field           f[16];
    SvtHipMeResults dev;
    const size_t    bytes = layout(g, desc, res, &dev, nullptr, f);
    void           *scratch;
    if ((rc = svt_hip_scratch(ctx, lane, bytes, &scratch))) return rc;
    layout(g, desc, res, &dev, static_cast<uint8_t *>(scratch), f);
    SvtHipMeJob job;
    job.cfg = cfg; job.desc = desc; job.cur = cur; job.results = &dev;
    for (int li = 0; li < SVT_HIP_MAX_LISTS; li++)
        for (int ri = 0; ri < SVT_HIP_MAX_REFS; ri++) job.refs[li][ri] = refs[li][ri];
    if ((rc = svt_hip_me_pictures_on_lane(ctx, lane, 1, &job))) return rc;
    // copy back only the rows of this call's band
    void *const *hp = reinterpret_cast<void *const *>(res);
    void *const *dp = reinterpret_cast<void *const *>(&dev);
    for (int i = 0; i < 16; i++) {
        if (!hp[i]) continue;
        const size_t per_b64 = f[i].count / g.n_b64 * f[i].elem;
        const size_t lo = (size_t)g.row0 * g.w64 * per_b64, n = (size_t)g.nrow * g.w64 * per_b64;
        SVT_HIP_CHECK(ctx, hipMemcpyAsync(static_cast<uint8_t *>(hp[i]) + lo, static_cast<const uint8_t *>(dp[i]) + lo, n, hipMemcpyDeviceToHost,
                                          lane->stream));
    }
    SVT_HIP_CHECK(ctx, hipStreamSynchronize(lane->stream));
    // The kernel writes the search-level arrays for the (list, reference) pairs the picture searches only; this entry hands back whole
    // arrays with the canonical values in the other slots
    if (res->sb_best_sad || res->sb_best_mv)
        for (size_t b = (size_t)g.row0 * g.w64; b < (size_t)(g.row0 + g.nrow) * g.w64; b++)
            for (int li = 0; li < SVT_HIP_MAX_LISTS; li++)
                for (int ri = 0; ri < SVT_HIP_MAX_REFS; ri++) {
                    if (li < desc->num_of_list_to_search && ri < desc->num_of_ref_pic_to_search[li]) continue;
                    const size_t o = b * 680 + (size_t)(li * 4 + ri) * 85;
                    for (int n = 0; n < 85; n++) {
                        if (res->sb_best_sad) res->sb_best_sad[o + n] = SVT_HIP_MAX_SAD_VALUE;
                        if (res->sb_best_mv) res->sb_best_mv[o + n] = 0;
                    }
                }
    return SVT_HIP_OK;
}

size_t svt_hip_sizeof(int what) {
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

} // extern "C"
