// me_kernel.h -- launch-time parameter block of svt_hip_me_b64_kernel (host and device view).
#ifndef SVT_HIP_ME_KERNEL_H
#define SVT_HIP_ME_KERNEL_H
#include <stdint.h>
#include "../../include/svt_hip_me.h"

// Shape of the ME kernel (tunable at build time): one wave per 64x64 block; waves per SIMD the launch bounds plan for
// (2 -> 256 VGPRs, 8 blocks in flight per CU, no register spills; 3 -> 168 VGPRs, 12 blocks, 35 spilled registers and 1.5 % more ME time)
// and the LDS window arena each wave owns (the integer search needs 64 + 2 rows of 96 .. 112 bytes at least: 8 KiB; measured at 2 waves
// per SIMD: 9 / 10 KiB 2.94 ms per launch of the bench, 12 / 14 KiB 3.25 ms -- the eighth wave of a CU no longer fits its LDS).
#ifndef SVT_HIP_ME_WAVES_PER_SIMD
#define SVT_HIP_ME_WAVES_PER_SIMD 2
#endif
#ifndef SVT_HIP_ME_WIN_BYTES
#define SVT_HIP_ME_WIN_BYTES 10240
#endif
#define SVT_HIP_ME_QUEUES 8 /* one b64 band queue per XCD */

// A padded 8-bit luma plane resident in HBM.  `base` is the padded buffer's first byte (buffer_y); it and
// `stride` are multiples of 16 so every row keeps the same 16-byte phase; 256 bytes of slack follow the last row.
struct DevPlane {
    const uint8_t *base;
    uint32_t       stride;
    int32_t        org_x, org_y, width, height;
};

struct DevPyramid {
    DevPlane lvl[3]; // 0 = sixteenth, 1 = quarter, 2 = full
};

// 32-bit mirrors of SvtHipMeConfig / SvtHipMePictureDesc (same field names): sub-dword fields cannot be fetched with
// scalar loads on gfx950, so the launcher widens them once.  u8 / u16 fields become int32_t -- exactly the value C's
// integer promotion gives them in an expression -- wider fields keep their type.
struct DevSearchArea { int32_t width, height; };
struct DevSearchAreaMinMax { DevSearchArea sa_min, sa_max; };

#define SVT_ME_CFG_SMALL(X) X(hme_search_method) X(me_search_method) X(enable_hme_flag) X(enable_hme_level0_flag) \
    X(enable_hme_level1_flag) X(enable_hme_level2_flag) X(num_hme_sa_w) X(num_hme_sa_h) X(prehme_enable) \
    X(prehme_skip_search_line) X(prehme_l1_early_exit) X(me_type) X(enable_me_hme_ref_pruning) \
    X(prune_ref_if_hme_sad_dev_bigger_than_th) X(prune_ref_if_me_sad_dev_bigger_than_th) X(zz_sad_pct) X(phme_sad_pct) \
    X(enable_me_sr_adjustment) X(reduce_me_sr_based_on_mv_length_th) X(stationary_hme_sad_abs_th) \
    X(stationary_me_sr_divisor) X(reduce_me_sr_based_on_hme_sad_abs_th) X(me_sr_divisor_for_low_hme_sad) \
    X(distance_based_hme_resizing) X(me_8x8_var_enabled) X(mv_sa_adj_enabled) X(mv_sa_adj_nearest_ref_only) \
    X(mv_sa_adj_mv_size_th) X(mv_sa_adj_sa_multiplier) X(prune_me_candidates_th) X(use_best_unipred_cand_only) \
    X(reduce_hme_l0_sr_th_min) X(reduce_hme_l0_sr_th_max)
#define SVT_ME_CFG_U32(X) X(zz_sad_th) X(phme_sad_th) X(me_sr_div4_th) X(me_sr_div2_th) X(me_sr_mult2_th) X(me_early_exit_th) \
    X(me_safe_limit_zz_th) X(prev_me_stage_based_exit_th)
#define SVT_ME_DESC_SMALL(X) X(aligned_width) X(aligned_height) X(num_of_list_to_search) X(temporal_layer_index) \
    X(hierarchical_levels) X(is_ref) X(similar_brightness_refs) X(enable_me_8x8) X(enable_me_16x16) \
    X(max_number_of_pus_per_sb) X(max_cand) X(max_refs) X(max_l0) X(input_resolution) X(only_l_bwd) X(gm_enabled) \
    X(gm_use_distance_based_active_th) X(b64_row_start) X(b64_row_count)

struct DevMeConfig {
#define SVT_X(n) int32_t n;
    SVT_ME_CFG_SMALL(SVT_X)
#undef SVT_X
#define SVT_X(n) uint32_t n;
    SVT_ME_CFG_U32(SVT_X)
#undef SVT_X
    DevSearchAreaMinMax hme_l0_sa, me_sa, prehme_sa_cfg[2];
    DevSearchArea       hme_l1_sa, hme_l2_sa;
};

struct DevMeDesc {
    uint64_t picture_number;
    uint64_t ref_picture_number[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
#define SVT_X(n) int32_t n;
    SVT_ME_DESC_SMALL(SVT_X)
#undef SVT_X
    int32_t num_of_ref_pic_to_search[SVT_HIP_MAX_LISTS];
    uint32_t tf_me_exit_th;
    uint32_t pad;
};

#ifdef __cplusplus
static inline DevSearchArea dev_sa(const SvtHipSearchArea &a) { DevSearchArea r = {a.width, a.height}; return r; }
static inline DevSearchAreaMinMax dev_sa(const SvtHipSearchAreaMinMax &a) { DevSearchAreaMinMax r = {dev_sa(a.sa_min), dev_sa(a.sa_max)}; return r; }
static inline void dev_me_config(DevMeConfig &o, const SvtHipMeConfig &c) {
#define SVT_X(n) o.n = c.n;
    SVT_ME_CFG_SMALL(SVT_X)
    SVT_ME_CFG_U32(SVT_X)
#undef SVT_X
    o.hme_l0_sa = dev_sa(c.hme_l0_sa); o.me_sa = dev_sa(c.me_sa);
    o.prehme_sa_cfg[0] = dev_sa(c.prehme_sa_cfg[0]); o.prehme_sa_cfg[1] = dev_sa(c.prehme_sa_cfg[1]);
    o.hme_l1_sa = dev_sa(c.hme_l1_sa); o.hme_l2_sa = dev_sa(c.hme_l2_sa);
}
static inline void dev_me_desc(DevMeDesc &o, const SvtHipMePictureDesc &d) {
    o.picture_number = d.picture_number;
    o.tf_me_exit_th = d.tf_me_exit_th;
    o.pad = 0;
    for (int l = 0; l < SVT_HIP_MAX_LISTS; l++) {
        o.num_of_ref_pic_to_search[l] = d.num_of_ref_pic_to_search[l];
        for (int r = 0; r < SVT_HIP_MAX_REFS; r++) o.ref_picture_number[l][r] = d.ref_picture_number[l][r];
    }
#define SVT_X(n) o.n = d.n;
    SVT_ME_DESC_SMALL(SVT_X)
#undef SVT_X
}
#endif

struct MeKernelParams {
    DevMeConfig         cfg;
    DevMeDesc           desc;
    DevPyramid          cur;
    DevPyramid          ref[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
    SvtHipMeResults     res;   // device pointers
    uint32_t            w64, row0, n_pu, pad;
};

// One launch covers the b64 jobs of up to SVT_HIP_ME_MAX_PICTURES pictures.  Global job index = job_base[pic] + the
// picture's band-local b64 raster index; eight queues (one per XCD) each own a contiguous range of global indices.
#define SVT_HIP_ME_MAX_PICTURES 16
#define SVT_HIP_ME_HEADER_BYTES 256 /* sizeof(MeBatchHeader) rounded up: the parameter blocks follow at this offset */
// Dense pre-pass (me_dense.inl): one slot per (global job index, searched (list, reference) pair, kind)
#define SVT_HIP_ME_DENSE_KINDS 6   /* two pre-HME strips + 2 x 2 level-0 regions */
#define SVT_HIP_ME_COUNTER_WORD 112 /* u32 index into a lane's queue_head block: two u64 counters (dense slots taken / not found) */
struct MeDenseSlot {
    unsigned long long key;  // sad << 32 | y << 16 | x (search indices), ~0 when no position was evaluated / not computed
    uint32_t           org;  // (uint16)ox | (uint16)oy << 16: displacement of search index (0, 0)
    uint32_t           size; // sa_w | sa_h << 16
};
// One (picture, searched reference, kind) of a launch's dense pre-pass: units = band rows x dy segments x lane chunks
struct MeDenseEntry {
    uint32_t unit_base;         // first unit of the entry in the launch's unit numbering
    uint16_t pic;
    uint8_t  li, ri, k, kind;   // k: row of (li, ri) among the picture's searched pairs
    uint16_t noct, nos;         // octets of the unclipped search width; octet slots per block (a lane walks the octets slot, slot + nos, ...)
    uint16_t n_chunk, n_seg;    // chunks of 64 lanes per block row; dy segments
    uint16_t seg_len, n_rows;   // search rows per segment; b64 rows of the picture's band
    uint32_t pad;
};
#define SVT_HIP_ME_DENSE_MAX_ENTRIES (SVT_HIP_ME_MAX_PICTURES * SVT_HIP_MAX_LISTS * SVT_HIP_MAX_REFS * SVT_HIP_ME_DENSE_KINDS)

// Staged launches (me_kernel.hip): per job a record in HBM -- the head of the block's state, its search requests, their results -- and a flag word
#define SVT_HIP_ME_STAGE_BYTES 2560
#define SVT_HIP_ME_JOB_DEFERRED 1u /* a search the pre-pass did not make: the whole-pipeline kernel makes the block (list 0) */
/* u32 indices into a lane's queue_head block (2 KiB): [0, 8) the one-kernel form's band-queue counters, [16, 112) the profiling build's phase
 * sums, [112, 116) the dense counters, [200, 206) the three lists' cursors / counts */
#define SVT_HIP_ME_LIST_CURSOR(l) (200 + 2 * (l))
#define SVT_HIP_ME_LIST_COUNT(l) (201 + 2 * (l))
#define SVT_HIP_ME_QUEUE_BLOCK_BYTES 2048
#ifndef SVT_HIP_ME_MID_WAVES_PER_SIMD
#define SVT_HIP_ME_MID_WAVES_PER_SIMD 4
#endif
#ifndef SVT_HIP_ME_SEARCH_DEPTH
#define SVT_HIP_ME_SEARCH_DEPTH 2 /* steps of window rows in flight in the search kernels' direct loop */
#endif
#ifndef SVT_HIP_ME_SEARCH_WAVES_PER_SIMD
#define SVT_HIP_ME_SEARCH_WAVES_PER_SIMD 3
#endif
#ifndef SVT_HIP_ME_TAIL_WAVES_PER_SIMD
#define SVT_HIP_ME_TAIL_WAVES_PER_SIMD 2
#endif

struct MeBatchHeader {
    uint32_t  n_pictures, n_slot; // n_slot: the largest number of (list, reference) pairs a picture of the launch searches
    uint32_t  job_base[SVT_HIP_ME_MAX_PICTURES + 1];
    uint32_t  queue_begin[SVT_HIP_ME_QUEUES + 1];
    uint32_t  cshift, pad; // 1: the source views in LDS keep their even rows only (every search of the launch is row-subsampled)
    uint32_t *queue_head; // SVT_HIP_ME_QUEUES counters, zeroed before launch
    MeDenseSlot *dense;   // results of the dense pre-pass, [job_base[n_pictures]][n_slot][SVT_HIP_ME_DENSE_KINDS]; null: no pre-pass
    uint32_t  n_dense_entries, n_dense_units;
    uint32_t  count_dense; // the waves add their taken / own-search counts to the lane's counters (svt_hip_context_set_me_counting; two atomics per wave, serialised at the memory side: ~0.1 ms of a launch of 4096 waves)
    // staged launches (null otherwise): per job SVT_HIP_ME_STAGE_BYTES of travelling state, a flag word; three job lists (0: deferred to the
    // whole-pipeline kernel, 1 / 2: level-1 / level-2 searches in the staged form)
    uint8_t   *stage;
    uint32_t  *job_flags;
    uint32_t  *lists[1]; // [0]: the deferred blocks
};
#ifdef __cplusplus
static_assert(sizeof(MeBatchHeader) <= SVT_HIP_ME_HEADER_BYTES, "header size");
#endif
// A lane's parameter block (device copy and the pinned ring): header, SVT_HIP_ME_MAX_PICTURES parameter blocks, the dense entry table
#define SVT_HIP_ME_ENTRIES_OFFSET (SVT_HIP_ME_HEADER_BYTES + sizeof(MeKernelParams) * SVT_HIP_ME_MAX_PICTURES)
#define SVT_HIP_ME_PARAM_BYTES (SVT_HIP_ME_ENTRIES_OFFSET + sizeof(MeDenseEntry) * SVT_HIP_ME_DENSE_MAX_ENTRIES)

#endif
