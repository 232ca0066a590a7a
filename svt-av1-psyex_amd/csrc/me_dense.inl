// me_dense.inl -- the dense pre-pass of the open-loop ME (included by me_kernel.hip, whose helpers it shares).
//
// The pre-HME strips (prehme_core, Codec/motion_estimation.c:1568-1666) and the four HME level-0 regions (hme_level_0, :820-920) are 70 % of
// the |a-b| evaluations of a preset-6 block, and their search windows depend on nothing but the block's position and the picture distance
// (when the reduce_hme_l0_sr thresholds are off, or for list 0 / reference 0: :1800-1866).  They therefore leave the per-block wave, whose
// time goes into the latencies of its dependent stages, for a kernel in which every resident wave does nothing but packed SADs:
//
//   * a lane owns one 16 x 16 block of the sixteenth-resolution picture and one OCTET of horizontal search positions; the block's 8
//     (row-subsampled) source rows stay in its registers for the whole unit.
//   * the wave walks DOWN the reference rows of the search window once.  Reference row R meets source row r at the search row
//     dy = R - r, so one 24-byte row fetched from global memory (L1 / L2 hits: neighbouring lanes overlap) feeds 8 source rows x 8
//     v_qsad_pk_u16_u8 -- 64 packed-SAD instructions per load -- into a ring of 16 x 2 packed accumulators (search rows dy .. dy + 15 in
//     flight, ring slot = dy mod 16, compile-time register numbers through a loop unrolled 16 x 8).  The u16 lanes cannot overflow: 8 rows x
//     16 pixels x 255 = 32,640.  A search row is complete when its last source row has been added; the lane then folds the 8 sums into its
//     running best with the reference's order (strict `<` over search rows visited in ascending order, first position of the octet on ties).
//   * no LDS staging, no tile plan, no serial control code: the only LDS traffic is one 64-bit atomic min per lane at the end of a unit
//     (lanes of one block meet there), followed by one global atomic min per block: key = sad << 32 | y << 16 | x, the reference's "first
//     minimum in raster order" (C_DEFAULT/compute_sad_c.c:58-101).
//   * work = units: (entry = picture x searched reference x kind, b64 row, segment of search rows, chunk of 64 lanes); tall strips are
//     cut into segments of about 48 search rows, wide ones give a lane several octets, so that units cost about the same.
//
// Exactly the searches svt_sad_loop_kernel would make: geometry from prehme_geometry / hme_level_geometry (shared with the per-block
// kernel), blocks that are not a full 64 samples wide and searches with all 16 rows (hme_search_method 1) are left to the per-block kernel.

namespace {

#ifndef SVT_HIP_ME_DENSE_GROUPS
#define SVT_HIP_ME_DENSE_GROUPS 384
#endif
constexpr int kDenseTargetGroups = SVT_HIP_ME_DENSE_GROUPS; // (octets x search rows x source rows) a lane aims for per unit (measured on the bench launch: 256 -> 0.553 ms, 384 -> 0.527, 768 -> 0.547)

__device__ __forceinline__ u64 dense_group(const uint32_t (&wv)[6], const uint32_t (&s)[4], u64 acc, int half) {
    // 4 pixels x 4 dwords of one source row against the lane's octet: positions 0..3 (half 0) or 4..7 (half 1)
#pragma unroll
    for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_qsad_pk_u16_u8(((u64)wv[j + 1 + half] << 32) | wv[j + half], s[j], acc);
    return acc;
}

} // namespace

#ifndef SVT_HIP_ME_DENSE_WAVES
#define SVT_HIP_ME_DENSE_WAVES 3
#endif
extern "C" __global__ void __launch_bounds__(64, SVT_HIP_ME_DENSE_WAVES)
svt_hip_me_dense_kernel(const MeBatchHeader *__restrict__ ghdr, const MeKernelParams *__restrict__ gparams, const MeDenseEntry *__restrict__ gent) {
    typedef const SVT_CONST_AS MeBatchHeader CHeader;
    typedef const SVT_CONST_AS MeDenseEntry  CEntry;
    CHeader  &hdr  = *(CHeader *)ghdr;
    const int lane = threadIdx.x;
    __shared__ u64 smin[64];

    // ---- which unit: binary search of the entry whose unit range holds blockIdx.x (uniform, scalar loads) ----
    const uint32_t unit = blockIdx.x;
    int lo = 0, hi = (int)hdr.n_dense_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (((CEntry *)gent)[mid].unit_base <= unit) lo = mid; else hi = mid - 1;
    }
    CEntry &e = ((CEntry *)gent)[lo];
    CParams &p = ((CParams *)gparams)[e.pic];
    auto &c = p.cfg;
    auto &d = p.desc;
    const int li = e.li, ri = e.ri, kind = e.kind, nos = e.nos, n_chunk = e.n_chunk, n_seg = e.n_seg, seg_len = e.seg_len;
    const uint32_t u = unit - e.unit_base;
    const uint32_t t = u / (uint32_t)n_chunk, chunk = u - t * (uint32_t)n_chunk;
    const uint32_t by = t / (uint32_t)n_seg, seg = t - by * (uint32_t)n_seg;
    if (by >= e.n_rows) return;
    const int w64 = (int)p.w64;
    const uint32_t byi = p.row0 + by;

    // ---- the lane's block and octet slot ----
    const int  item   = (int)chunk * 64 + lane;
    int        bx     = (int)div_by_rcp((uint32_t)item, rcp_of((uint32_t)nos));
    const int  oslot  = item - bx * nos;
    const bool in_row = bx < w64;
    if (!in_row) bx = w64 - 1; // keeps every address below inside the planes; nothing of such a lane is used
    const uint32_t org_x = (uint32_t)bx * 64u, org_y = byi * 64u;
    const int b64_w = imin(64, (int)d.aligned_width - (int)org_x), b64_h = imin(64, (int)d.aligned_height - (int)org_y);
    const int nrows = (b64_h >> 2) >> 1; // source rows of the row-subsampled 16 x 16 block (uniform); the last one always sits at k = 7
    if (nrows < 1) return;
    const int kfirst = 8 - nrows;
    SearchGeo g;
    if (kind < 2)
        g = prehme_geometry(p, li, ri, kind, org_x, org_y);
    else {
        int sa_w, sa_h;
        hme_l0_area(p, li, ri, ref_distance(p, li, ri), 0, 0, sa_w, sa_h); // (entries exist only where the area does not depend on earlier results)
        g = hme_level_geometry(p, 0, p.ref[li][ri].lvl[0], (int16_t)org_x >> 2, (int16_t)org_y >> 2, sa_w, sa_h, 0, 0, (kind - 2) & 1, (kind - 2) >> 1);
    }
    // vertical geometry is the same for the whole block row: lane 0's copy, in scalar registers
    const int oy = __builtin_amdgcn_readfirstlane(g.oy), sa_h = __builtin_amdgcn_readfirstlane(g.sa_h);
    const int noct_l = (g.sa_w + 7) >> 3; // octets of this lane's (clipped) search width
    // a search the entry's partition does not cover (cannot happen: clipping only shrinks a search) is left to the per-block kernel
    const bool covered = g.sa_w > 0 && sa_h > 0 && noct_l <= (int)e.noct && sa_h <= n_seg * seg_len;
    const bool active  = in_row && b64_w == 64 && covered;
    const int y_lo = (int)seg * seg_len, y_hi = imin(sa_h, y_lo + seg_len);
    if (y_lo >= y_hi) return; // uniform
    const int n = y_hi - y_lo;
    const bool skip_even = kind < 2 && c.prehme_skip_search_line; // push_req: skip && bw == 16 && rows <= 16 -- always true for the blocks taken here

    const size_t slot_idx = (((size_t)hdr.job_base[e.pic] + (size_t)by * (size_t)w64 + (size_t)bx) * hdr.n_slot + e.k) * SVT_HIP_ME_DENSE_KINDS + (size_t)kind;
    if (active && oslot == 0 && seg == 0) { // the geometry the key is valid for; doubles as the slot's "filled" mark
        hdr.dense[slot_idx].org  = dense_pack(g.ox, g.oy);
        hdr.dense[slot_idx].size = dense_pack(g.sa_w, sa_h);
    }

    // ---- the block's source rows (sixteenth plane, every other row), last row at k = 7 ----
    uint32_t s[8][4];
    {
        CPlane &cp = p.cur.lvl[0];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int row = k >= kfirst ? 2 * (k - kfirst) : 0;
            const V4U q = *reinterpret_cast<GV4U *>(reinterpret_cast<uintptr_t>(plane_at(cp, (int)(org_x >> 2), (int)(org_y >> 2) + row)));
            s[k][0] = q.x; s[k][1] = q.y; s[k][2] = q.z; s[k][3] = q.w;
        }
    }

    // ---- reference rows: uniform row base + the lane's byte offset.  Loop index Rp (0 .. n - 1 + 14) <-> plane row (block row + oy + y_lo + Rp - 2 kfirst) ----
    CPlane &rp = p.ref[li][ri].lvl[0];
    const uint32_t stride = rp.stride;
    const uint8_t *const row0 = rp.base + (long long)(rp.org_y + (int)(org_y >> 2) + oy + y_lo - 2 * kfirst) * stride; // uniform (rows below 2 kfirst are never read)
    const int x_first = rp.org_x + (int)(org_x >> 2) + (active ? g.ox : 0); // >= 1: the windows are clipped to the padded plane
    const int Rp_begin = 2 * kfirst, Rp_end = n - 1 + 14;

    u64 lane_best = ~0ull;
    u64 acc0[16], acc1[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0; acc1[i] = 0; }

    const int n_oct_iter = (int)(((uint32_t)e.noct + (uint32_t)nos - 1u) / (uint32_t)nos); // uniform
    typedef uint32_t V2U __attribute__((ext_vector_type(2), aligned(1)));
    typedef const __attribute__((address_space(1))) V2U GV2U;
    // byte offset of octet iteration oi's window in a reference row (lanes past their clipped width read their first octet again: unused)
    auto octet_off = [&](int oi) { const int o = oslot + oi * nos; return (uint32_t)(x_first + 8 * ((active && o < noct_l) ? o : 0)); };
    // the first row of an octet's walk is fetched while the previous octet is still being evaluated
    V4U  nfa;
    V2U  nfb;
    {
        const uintptr_t a = reinterpret_cast<uintptr_t>(row0 + (long long)Rp_begin * stride) + octet_off(0);
        nfa = *reinterpret_cast<GV4U *>(a);
        nfb = *reinterpret_cast<GV2U *>(a + 16);
    }
    for (int oi = 0; oi < n_oct_iter; oi++) {
        const int  o     = oslot + oi * nos;
        const bool live  = active && o < noct_l;
        const int  nvalid = live ? imin(8, g.sa_w - 8 * o) : 0; // positions of the octet inside the search width
        const bool all_full = __all(nvalid == 8);
        const uint32_t loff = octet_off(oi);
        uint32_t best = 0xFFFFFFFFu; // sad << 16 | position in the octet
        int      bestd = 0;
        V4U  bufa[2];
        uint2 bufb[2];
        auto fetch = [&](int which, int Rp) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(row0 + (long long)Rp * stride) + loff;
            bufa[which] = *reinterpret_cast<GV4U *>(a);
            const V2U q = *reinterpret_cast<GV2U *>(a + 16);
            bufb[which] = make_uint2(q.x, q.y);
        };
        bufa[0] = nfa; bufb[0] = make_uint2(nfb.x, nfb.y); // Rp_begin = 2 kfirst is even: step Rp_begin reads slot 0
        if (oi + 1 < n_oct_iter) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(row0 + (long long)Rp_begin * stride) + octet_off(oi + 1);
            nfa = *reinterpret_cast<GV4U *>(a);
            nfb = *reinterpret_cast<GV2U *>(a + 16);
        }
        for (int it = 0; it * 16 <= Rp_end; it++) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int Rp = it * 16 + j;
                if (Rp < Rp_begin || Rp > Rp_end) continue; // uniform
                if (Rp < Rp_end) fetch((j + 1) & 1, Rp + 1); // the next row is on its way while this one is evaluated
                if (skip_even && !((y_lo + Rp) & 1)) continue; // search rows of the parity that is skipped (dy and Rp have the same parity)
                const uint32_t wv[6] = {bufa[j & 1].x, bufa[j & 1].y, bufa[j & 1].z, bufa[j & 1].w, bufb[j & 1].x, bufb[j & 1].y};
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int dd = Rp - 2 * k; // search row (segment-relative) this reference row serves through source row k
                    if (k < kfirst || dd < 0 || dd >= n) continue; // uniform
                    constexpr int kRing = 16;
                    const int sl = (j - 2 * k) & (kRing - 1); // == dd mod 16: a compile-time number
                    acc0[sl] = dense_group(wv, s[k], acc0[sl], 0);
                    acc1[sl] = dense_group(wv, s[k], acc1[sl], 1);
                    if (k == 7) { // the search row is complete
                        const uint32_t a = (uint32_t)acc0[sl], b = (uint32_t)(acc0[sl] >> 32), cc = (uint32_t)acc1[sl], e2 = (uint32_t)(acc1[sl] >> 32);
                        uint32_t key[8] = {a << 16,        (a & 0xFFFF0000u) | 1u, (b << 16) | 2u,  (b & 0xFFFF0000u) | 3u,
                                           (cc << 16) | 4u, (cc & 0xFFFF0000u) | 5u, (e2 << 16) | 6u, (e2 & 0xFFFF0000u) | 7u};
                        if (!all_full) {
#pragma unroll
                            for (int i = 0; i < 8; i++) key[i] = i < nvalid ? key[i] : 0xFFFFFFFFu;
                        }
                        const uint32_t m = umin(umin(umin(key[0], key[1]), key[2]), umin(umin(umin(key[3], key[4]), key[5]), umin(key[6], key[7])));
                        if (m < (best & 0xFFFF0000u)) { best = m; bestd = dd; } // strict `<` on the SAD: the first search row wins ties
                        acc0[sl] = 0; acc1[sl] = 0;
                    }
                }
            }
        }
        if (best != 0xFFFFFFFFu) {
            const u64 key = ((u64)(best >> 16) << 32) | ((u64)(uint32_t)(y_lo + bestd) << 16) | (u64)(uint32_t)(8 * o + (int)(best & 7u));
            lane_best = key < lane_best ? key : lane_best;
        }
    }

    // ---- the lanes of a block meet in LDS; one global atomic min per block and unit ----
    const int bx0 = __builtin_amdgcn_readfirstlane(bx);
    smin[lane] = ~0ull;
    wave_sync();
    if (active && lane_best != ~0ull) atomicMin(&smin[bx - bx0], lane_best);
    wave_sync();
    const u64 kmin = smin[lane];
    if (kmin != ~0ull) { // lane <-> block bx0 + lane of this wave
        const size_t idx = (((size_t)hdr.job_base[e.pic] + (size_t)by * (size_t)w64 + (size_t)(bx0 + lane)) * hdr.n_slot + e.k) * SVT_HIP_ME_DENSE_KINDS + (size_t)kind;
        atomicMin(&hdr.dense[idx].key, kmin);
    }
}

namespace {

// The launch's dense entries: every (picture, searched (list, reference) pair, kind) whose search window is known before any search
// result.  Returns the number of entries; units = the total number of units (0: nothing to do).
// `target_groups`: (octets x search rows x source rows) a lane aims for per unit; `seg_rows`: search rows of a unit's dy segment when an area is
// taller than `seg_max` -- the knobs that trade unit length against unit count.
uint32_t dense_plan_with(const MeKernelParams *params, const uint32_t *n_jobs, uint32_t n_pictures, MeDenseEntry *ent, uint32_t *units, uint32_t target_groups, int seg_max, int seg_rows) {
    uint32_t n = 0;
    struct Cost { uint32_t cost; };
    Cost cost[SVT_HIP_ME_DENSE_MAX_ENTRIES];
    for (uint32_t pi = 0; pi < n_pictures; pi++) {
        const MeKernelParams &p = params[pi];
        const auto &c = p.cfg;
        const auto &d = p.desc;
        if (c.hme_search_method != 0 || n_jobs[pi] == 0) continue; // all 16 rows of a block: the per-block kernel's searches
        const int nl = d.num_of_list_to_search, r0n = d.num_of_ref_pic_to_search[0];
        const bool l0_dep = c.enable_me_sr_adjustment && c.distance_based_hme_resizing && c.reduce_hme_l0_sr_th_min && c.reduce_hme_l0_sr_th_max;
        for (int li = 0; li < nl; li++) {
            if (!(d.temporal_layer_index > 0 || li == 0)) continue; // searched()
            for (int ri = 0; ri < d.num_of_ref_pic_to_search[li]; ri++)
                for (int kind = 0; kind < SVT_HIP_ME_DENSE_KINDS; kind++) {
                    int sa_w, sa_h;
                    if (kind < 2) {
                        if (!c.prehme_enable) continue;
                        prehme_size(p, li, ri, kind, sa_w, sa_h);
                    } else {
                        if (!(c.enable_hme_flag && c.enable_hme_level0_flag) || (l0_dep && (li || ri))) continue;
                        hme_l0_area(p, li, ri, ref_distance(p, li, ri), 0, 0, sa_w, sa_h);
                        sa_w = (int16_t)((sa_w + 7) & ~7);
                    }
                    if (sa_w <= 0 || sa_h <= 0) continue;
                    MeDenseEntry &e = ent[n];
                    memset(&e, 0, sizeof(e));
                    e.pic = (uint16_t)pi; e.li = (uint8_t)li; e.ri = (uint8_t)ri; e.k = (uint8_t)((li ? r0n : 0) + ri); e.kind = (uint8_t)kind;
                    e.noct = (uint16_t)((sa_w + 7) >> 3);
                    e.n_seg = (uint16_t)(sa_h <= seg_max ? 1 : (sa_h + seg_rows - 1) / seg_rows);
                    e.seg_len = (uint16_t)((sa_h + e.n_seg - 1) / e.n_seg);
                    const uint32_t groups = (uint32_t)e.noct * e.seg_len * 8u; // of one block
                    uint32_t nos = (groups + target_groups - 1) / target_groups;
                    nos = nos < 1 ? 1 : (nos > e.noct ? e.noct : nos);
                    while (e.noct % nos) nos++; // a divisor of the octet count: every lane of a block walks the same number of octets
                    e.nos = (uint16_t)nos;
                    e.n_chunk = (uint16_t)((p.w64 * e.nos + 63u) / 64u);
                    e.n_rows = (uint16_t)(n_jobs[pi] / p.w64);
                    cost[n].cost = ((uint32_t)e.noct + e.nos - 1u) / e.nos * e.seg_len;
                    n++;
                }
        }
    }
    // long units first: the launch's tail is made of short ones (insertion sort: at most a few hundred entries)
    for (uint32_t i = 1; i < n; i++) {
        const MeDenseEntry e = ent[i];
        const Cost         k = cost[i];
        uint32_t           j = i;
        for (; j > 0 && cost[j - 1].cost < k.cost; j--) { ent[j] = ent[j - 1]; cost[j] = cost[j - 1]; }
        ent[j] = e; cost[j] = k;
    }
    uint32_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        ent[i].unit_base = total;
        total += (uint32_t)ent[i].n_rows * ent[i].n_seg * ent[i].n_chunk;
    }
    *units = total;
    return n;
}

// A launch of many blocks gets long units (384 groups per lane: few waves' worth of set-up per |a-b|); a launch of few blocks -- a rank's band
// at 8 GPUs, a single small picture -- would leave most of the chip's resident waves without a unit that way, so it is cut finer.
uint32_t dense_plan(const MeKernelParams *params, const uint32_t *n_jobs, uint32_t n_pictures, MeDenseEntry *ent, uint32_t *units, uint32_t resident_waves) {
    uint32_t n = dense_plan_with(params, n_jobs, n_pictures, ent, units, kDenseTargetGroups, 64, 48);
    if (*units && *units < 3u * resident_waves) n = dense_plan_with(params, n_jobs, n_pictures, ent, units, kDenseTargetGroups / 2, 32, 24);
    if (*units && *units < 3u * resident_waves) n = dense_plan_with(params, n_jobs, n_pictures, ent, units, kDenseTargetGroups / 4, 16, 16);
    return n;
}

} // namespace
