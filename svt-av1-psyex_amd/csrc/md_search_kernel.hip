// md_search_kernel.hip -- the mode-decision side motion search (include/svt_hip_md_search.h), one wave per job:
//
//   md_fullpel_kernel  = md_full_pel_search (Codec/product_coding_loop.c:2042-2180): the search-area adjustment, then either the wide 8-bit SAD
//     form (md_full_pel_search_large_lbd :1958-2027 = one svt_pme_sad_loop_kernel call :1905-1950: rows `step` apart, groups of 8 columns
//     7 + step apart) or the position loop (columns outer, rows inner, sparse-level skip rule) with SAD or variance as the distortion; every
//     position's cost = distortion + MV rate (svt_aom_fp_mv_err_cost, Codec/mcomp.c:44-78,776); lane <-> search position, the block's rows
//     through v_sad_u8 / dot products straight from global memory (L1 / L2: the windows of a job are a few KB); the winner is the wave minimum
//     of (cost << 32 | visiting order) against the incoming best with the reference's strict `<`.  Jobs chain on the device (centre and / or
//     incoming best = an earlier job's outputs): the rounds of md_nsq_motion_search (:2260-2375) never come back to the host.
//   md_subpel_kernel   = svt_av1_find_best_sub_pixel_tree (Codec/mcomp.c:688-777; search_method 1: every candidate's error is the variance of
//     svt_aom_upsampled_pred -- separable 2 / 4 / 8-tap interpolation, recomputed per pixel) or
//     svt_av1_find_best_sub_pixel_tree_pruned (Codec/mcomp.c:606-687) with svt_estimated_pref_error (:147-167): the tree's
//     control flow is wave-uniform scalar code; each candidate's error = svt_aom_sub_pixel_variance{W}x{H}_c (C_DEFAULT/variance.c:28-75,
//     308-318: two bilinear passes with 7-bit rounding, then sse - sum^2 / (w h)) with the block's pixels spread over the lanes.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <string.h>
#include "svt_hip_internal.h"
#include "../../include/svt_hip_md_search.h"

namespace {
typedef unsigned long long u64;
typedef long long          i64;

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

struct MvCost {
    SvtHipMv       ref_mv;
    int            type, error_per_bit;
    const int32_t *mvjcost, *row, *col;
};

// svt_mv_err_cost (mcomp.c:44-69)
__device__ __forceinline__ int mv_err_cost(int16_t row, int16_t col, const MvCost &m) {
    const int16_t dr = (int16_t)(row - m.ref_mv.row), dc = (int16_t)(col - m.ref_mv.col); // MV fields are int16
    const int16_t ar = (int16_t)(dr < 0 ? -dr : dr), ac = (int16_t)(dc < 0 ? -dc : dc);
    switch (m.type) {
    case SVT_HIP_MV_COST_ENTROPY: {
        const int joint = dr == 0 ? (dc == 0 ? 0 : 1) : (dc == 0 ? 2 : 3);
        const int bits  = m.mvjcost[joint] + m.row[clip3(-(1 << 14), 1 << 14, dr)] + m.col[clip3(-(1 << 14), 1 << 14, dc)];
        return (int)((((i64)bits * m.error_per_bit) + ((i64)1 << 13)) >> 14);
    }
    case SVT_HIP_MV_COST_L1_LOWRES: return (2 * (ar + ac)) >> 3;
    case SVT_HIP_MV_COST_L1_MIDRES: return 0;
    case SVT_HIP_MV_COST_L1_HDRES: return (ar + ac) >> 3;
    case SVT_HIP_MV_COST_OPT: return (int)((((i64)((ar + ac) << 8) * m.error_per_bit) + ((i64)1 << 13)) >> 14);
    default: return 0;
    }
}

__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u64 t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}

// one lane: SAD or (sum of differences, sum of squares) of a w x h block, byte by byte in dwords where the width allows
__device__ __forceinline__ uint32_t lane_sad(const uint8_t *a, uint32_t as, const uint8_t *b, uint32_t bs, int w, int h) {
    uint32_t sad = 0;
    for (int y = 0; y < h; y++) {
        const uint8_t *pa = a + (size_t)y * as, *pb = b + (size_t)y * bs;
        for (int x = 0; x < w; x += 4) {
            uint32_t va, vb;
            memcpy(&va, pa + x, 4);
            memcpy(&vb, pb + x, 4);
            sad = __builtin_amdgcn_sad_u8(va, vb, sad);
        }
    }
    return sad;
}
__device__ __forceinline__ void lane_var(const uint8_t *a, uint32_t as, const uint8_t *b, uint32_t bs, int w, int h, int &sum, uint32_t &sse) {
    sum = 0; sse = 0;
    for (int y = 0; y < h; y++) {
        const uint8_t *pa = a + (size_t)y * as, *pb = b + (size_t)y * bs;
        for (int x = 0; x < w; x++) {
            const int d = (int)pa[x] - (int)pb[x];
            sum += d;
            sse += (uint32_t)(d * d);
        }
    }
}

struct FullpelParams { SvtHipFullpelBatchDesc d; };

__global__ void __launch_bounds__(64) md_fullpel_kernel(const FullpelParams p) {
    const uint32_t job = blockIdx.x;
    const int      lane = threadIdx.x;
    const SvtHipFullpelBatchDesc &d = p.d;
    const SvtHipFullpelJob jb = d.jobs[job];
    int16_t  mvx = jb.mvx, mvy = jb.mvy, in_x = jb.best_mvx, in_y = jb.best_mvy;
    uint32_t in_cost = jb.best_cost;
    if (jb.flags & SVT_HIP_FP_CENTRE_FROM_CHAIN) { mvx = d.best_mv[2 * jb.chain_from]; mvy = d.best_mv[2 * jb.chain_from + 1]; }
    if (jb.flags & SVT_HIP_FP_BEST_FROM_CHAIN) { in_cost = d.best_cost[jb.chain_from]; in_x = d.best_mv[2 * jb.chain_from]; in_y = d.best_mv[2 * jb.chain_from + 1]; }
    // every lane has read the chain inputs before any lane of this wave overwrites them (a job may chain from itself: results in place)
    __builtin_amdgcn_wave_barrier();
    int sx = jb.start_x, ex = jb.end_x, sy = jb.start_y, ey = jb.end_y;
    const int bx = jb.blk_org_x, by = jb.blk_org_y, bw = jb.width, bh = jb.height, step = jb.step < 1 ? 1 : jb.step;
    // search area adjustment (:2060-2076); the reference keeps the positions in int16
    if ((bx + (mvx >> 3) + sx) < (-d.ref_org_x + 1)) sx = (int16_t)((-d.ref_org_x + 1) - (bx + (mvx >> 3)));
    if ((bx + bw + (mvx >> 3) + ex) > (d.ref_org_x + d.ref_max_width - 1)) ex = (int16_t)((d.ref_org_x + d.ref_max_width - 1) - (bx + bw + (mvx >> 3)));
    if ((by + (mvy >> 3) + sy) < (-d.ref_org_y + 1)) sy = (int16_t)((-d.ref_org_y + 1) - (by + (mvy >> 3)));
    if ((by + bh + (mvy >> 3) + ey) > (d.ref_org_y + d.ref_max_height - 1)) ey = (int16_t)((d.ref_org_y + d.ref_max_height - 1) - (by + bh + (mvy >> 3)));
    const uint8_t *src = d.src + jb.src_offset;
    MvCost mc = {jb.ref_mv, d.mv_cost_type, d.error_per_bit, d.mvjcost, d.mvcost[0], d.mvcost[1]};
    u64 best = ~0ull; // cost << 32 | visiting order
    const bool wide = jb.dist_type == SVT_HIP_DIST_SAD && (jb.flags & SVT_HIP_FP_ENABLE_PSAD) && (ex - sx) >= 7;
    int n_groups = 0, ny = 0;
    if (wide) { // md_full_pel_search_large_lbd: width rounded up to a multiple of 8, svt_pme_sad_loop_kernel's visiting pattern
        int remain = 8 - ((ex - sx) % 8);
        remain     = remain == 8 ? 0 : remain;
        const int ex2 = (int16_t)(ex + remain);
        const int sa_w = (ex2 - sx) & ~7, sa_h = ey - sy + 1;
        n_groups = sa_w >= 8 ? (sa_w - 8) / (7 + step) + 1 : 0;
        ny       = sa_h > 0 ? (sa_h - 1) / step + 1 : 0;
        const int n_pos = n_groups * 8 * ny;
        for (int it = lane; it < n_pos; it += 64) {
            const int yi = it / (n_groups * 8), rem = it - yi * (n_groups * 8), g = rem >> 3, i = rem & 7;
            const int px = sx + g * (7 + step) + i, py = sy + yi * step;
            const uint8_t *ref = d.ref + (ptrdiff_t)(d.ref_org_x + (bx + (mvx >> 3) + px)) + (ptrdiff_t)(by + (mvy >> 3) + d.ref_org_y + py) * (ptrdiff_t)d.ref_stride;
            const uint32_t sad = lane_sad(src, d.src_stride, ref, d.ref_stride, bw, bh);
            const int16_t  col = (int16_t)(mvx + ((uint32_t)px * 8)), row = (int16_t)(mvy + ((uint32_t)py * 8));
            const uint32_t cost = sad + (uint32_t)mv_err_cost(row, col, mc);
            const u64      key  = ((u64)cost << 32) | (uint32_t)it;
            best = key < best ? key : best;
        }
    } else if (ex >= sx && ey >= sy) { // the position loop: columns outer, rows inner
        const int nx = (ex - sx) / step + 1;
        ny = (ey - sy) / step + 1;
        for (int it = lane; it < nx * ny; it += 64) {
            const int xi = it / ny, yi = it - xi * ny;
            const int px = sx + xi * step, py = sy + yi * step;
            if (step == 2 && (jb.flags & SVT_HIP_FP_SPRS_LEV0_DONE)) // sparse level 1 skips what level 0 visited (:2099-2109)
                if ((px + (mvx >> 3)) >= jb.sprs_lev0_start_x && (px + (mvx >> 3)) <= jb.sprs_lev0_end_x && (py + (mvy >> 3)) >= jb.sprs_lev0_start_y &&
                    (py + (mvy >> 3)) <= jb.sprs_lev0_end_y && px % 4 == 0 && py % 4 == 0)
                    continue;
            const uint8_t *ref = d.ref + (ptrdiff_t)(d.ref_org_x + (bx + (mvx >> 3) + px)) + (ptrdiff_t)(by + (mvy >> 3) + d.ref_org_y + py) * (ptrdiff_t)d.ref_stride;
            u64 cost;
            if (jb.dist_type == SVT_HIP_DIST_VAR) {
                int      sum;
                uint32_t sse;
                lane_var(ref, d.ref_stride, src, d.src_stride, bw, bh, sum, sse);
                cost = sse - (uint32_t)(((i64)sum * sum) / (bw * bh)); // fn_ptr->vf: uint32 arithmetic (variance.c:256-296)
            } else
                cost = lane_sad(src, d.src_stride, ref, d.ref_stride, bw, bh);
            const int16_t col = (int16_t)(mvx + (px * 8)), row = (int16_t)(mvy + (py * 8));
            cost += (u64)(i64)mv_err_cost(row, col, mc);
            // the reference compares the 64-bit cost with *best_cost; a cost beyond 32 bits can never win
            const u64 key = cost > 0xFFFFFFFFull ? ~0ull : ((cost << 32) | (uint32_t)it);
            best = key < best ? key : best;
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) {
        uint32_t out_cost = in_cost;
        int16_t  out_x = in_x, out_y = in_y;
        if (best != ~0ull && (uint32_t)(best >> 32) < in_cost) { // strict `<` against what the caller brought
            const int it = (int)(uint32_t)best;
            int px, py;
            if (wide) {
                const int yi = it / (n_groups * 8), rem = it - yi * (n_groups * 8);
                px = sx + (rem >> 3) * (7 + step) + (rem & 7); py = sy + yi * step;
            } else {
                const int xi = it / ny, yi = it - xi * ny;
                px = sx + xi * step; py = sy + yi * step;
            }
            out_cost = (uint32_t)(best >> 32);
            out_x    = (int16_t)(mvx + ((uint32_t)px * 8));
            out_y    = (int16_t)(mvy + ((uint32_t)py * 8));
        }
        d.best_cost[job]       = out_cost;
        d.best_mv[2 * job]     = out_x;
        d.best_mv[2 * job + 1] = out_y;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
struct SubpelParams { SvtHipSubpelBatchDesc d; };

struct SpCtx {
    const SvtHipSubpelBatchDesc *d;
    SvtHipSubpelJob              jb;
    MvCost                       mc;
    const uint8_t               *src, *ref;
};

// the wave: variance of the block at `a` (bilinear-filtered at the 1/8 offsets xo, yo; C_DEFAULT/variance.c:28-75) against `b`; lane <-> pixel
__device__ __forceinline__ uint32_t wave_sub_pixel_variance(const uint8_t *a, uint32_t as, int xo, int yo, const uint8_t *b, uint32_t bs, int w, int h, uint32_t &sse_out) {
    const int fx0 = 128 - 16 * xo, fx1 = 16 * xo, fy0 = 128 - 16 * yo, fy1 = 16 * yo;
    int      sum = 0;
    uint32_t sse = 0;
    const int n = w * h, lw = 31 - __clz(w); // widths are powers of two
    for (int i = threadIdx.x; i < n; i += 64) {
        const int y = i >> lw, x = i & (w - 1);
        const uint8_t *r0 = a + (size_t)y * as + x, *r1 = r0 + as;
        int v;
        if (xo | yo) { // taps with a zero weight still read their sample in the reference; the product is zero either way
            const int h0 = ((int)r0[0] * fx0 + (int)r0[1] * fx1 + 64) >> 7, h1 = ((int)r1[0] * fx0 + (int)r1[1] * fx1 + 64) >> 7;
            v = (h0 * fy0 + h1 * fy1 + 64) >> 7;
        } else
            v = r0[0];
        const int dd = v - (int)b[(size_t)y * bs + x];
        sum += dd;
        sse += (uint32_t)(dd * dd);
    }
    sum = wave_sum_i32(sum);
    sse = wave_sum_u32(sse);
    sse_out = sse;
    return sse - (uint32_t)(((i64)sum * sum) / n);
}

// rows 0, 2, .. 14 of av1_bilinear_filters / av1_sub_pel_filters_4 / av1_sub_pel_filters_8 (C_DEFAULT/variance.c:72-135): what
// av1_get_interp_filter_subpel_kernel(av1_get_filter(subpel_search_type), subpel_q3 << 1) selects (:191-201, :219-222)
__device__ const int8_t c_subpel_filters[3][8][8] = {
    {{0, 0, 0, 127, 0, 0, 0, 0}, {0, 0, 0, 112, 16, 0, 0, 0}, {0, 0, 0, 96, 32, 0, 0, 0}, {0, 0, 0, 80, 48, 0, 0, 0}, {0, 0, 0, 64, 64, 0, 0, 0}, {0, 0, 0, 48, 80, 0, 0, 0},
     {0, 0, 0, 32, 96, 0, 0, 0}, {0, 0, 0, 16, 112, 0, 0, 0}},
    {{0, 0, 0, 127, 0, 0, 0, 0}, {0, 0, -8, 122, 18, -4, 0, 0}, {0, 0, -12, 110, 38, -8, 0, 0}, {0, 0, -14, 94, 58, -10, 0, 0}, {0, 0, -12, 76, 76, -12, 0, 0},
     {0, 0, -10, 58, 94, -14, 0, 0}, {0, 0, -8, 38, 110, -12, 0, 0}, {0, 0, -4, 18, 122, -8, 0, 0}},
    {{0, 0, 0, 127, 0, 0, 0, 0}, {0, 2, -10, 122, 18, -4, 0, 0}, {0, 2, -14, 110, 38, -10, 2, 0}, {0, 2, -16, 94, 58, -12, 2, 0}, {0, 2, -14, 76, 76, -14, 2, 0},
     {0, 2, -12, 58, 94, -16, 2, 0}, {0, 2, -10, 38, 110, -14, 2, 0}, {0, 0, -4, 18, 122, -10, 2, 0}}};
// (phase 0 -- weight 128, which an int8 cannot hold -- is never read: a pass with phase 0 is skipped, as in the reference)

// the wave: variance of svt_aom_upsampled_pred (C_DEFAULT/variance.c:204-254; svt_aom_convolve8_horiz_c / _vert_c, Codec/convolve.c:244-301) of the
// block at `a` (MV floor) at the 1/8 offsets (xo, yo) against `b`; lane <-> pixel.  8 taps at offsets -3 .. +4, (sum + 64) >> 7 and a clip to 8 bits
// after EACH pass (the reference's intermediate is an 8-bit array): a 2-D position recomputes the eight horizontally filtered samples above and
// below itself (64 multiply-adds per pixel; these searches are a few positions per block, nobody stages anything)
__device__ __forceinline__ int clip_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ uint32_t wave_upsampled_variance(const uint8_t *a, uint32_t as, int xo, int yo, int type, const uint8_t *b, uint32_t bs, int w, int h, uint32_t &sse_out) {
    int fx[8], fy[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { fx[k] = c_subpel_filters[type - 1][xo][k]; fy[k] = c_subpel_filters[type - 1][yo][k]; }
    int      sum = 0;
    uint32_t sse = 0;
    const int n = w * h, lw = 31 - __clz(w); // widths are powers of two
    for (int i = threadIdx.x; i < n; i += 64) {
        const int y = i >> lw, x = i & (w - 1);
        const uint8_t *r = a + (ptrdiff_t)y * (ptrdiff_t)as + x;
        int v;
        if (!xo && !yo)
            v = r[0];
        else if (!yo) {
            int t = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) t += (int)r[k - 3] * fx[k];
            v = clip_u8((t + 64) >> 7);
        } else if (!xo) {
            int t = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) t += (int)r[(ptrdiff_t)(k - 3) * (ptrdiff_t)as] * fy[k];
            v = clip_u8((t + 64) >> 7);
        } else {
            int t2 = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint8_t *rr = r + (ptrdiff_t)(j - 3) * (ptrdiff_t)as;
                int t = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) t += (int)rr[k - 3] * fx[k];
                t2 += clip_u8((t + 64) >> 7) * fy[j];
            }
            v = clip_u8((t2 + 64) >> 7);
        }
        const int dd = v - (int)b[(size_t)y * bs + x];
        sum += dd;
        sse += (uint32_t)(dd * dd);
    }
    sum = wave_sum_i32(sum);
    sse = wave_sum_u32(sse);
    sse_out = sse;
    return sse - (uint32_t)(((i64)sum * sum) / n);
}

__device__ __forceinline__ bool sp_in_range(const SpCtx &s, SvtHipMv mv) {
    return mv.col >= s.jb.col_min && mv.col <= s.jb.col_max && mv.row >= s.jb.row_min && mv.row <= s.jb.row_max;
}

// svt_check_better_fast (mcomp.c:170-206), is_scaled == 0; every value is wave-uniform
__device__ __forceinline__ uint32_t sp_check(const SpCtx &s, SvtHipMv this_mv, SvtHipMv &best_mv, uint32_t &besterr, uint32_t &sse1, int &distortion) {
    uint32_t cost;
    if (sp_in_range(s, this_mv)) {
        cost = (uint32_t)mv_err_cost(this_mv.row, this_mv.col, s.mc);
        if (s.mc.type == SVT_HIP_MV_COST_OPT) {
            const i64 bestcost = (i64)distortion + cost;
            if (bestcost > (((i64)besterr * (i64)s.jb.early_exit_th) / 1000)) return (uint32_t)bestcost;
        }
        uint32_t sse;
        const uint8_t *ref = s.ref + (ptrdiff_t)(this_mv.row >> 3) * (ptrdiff_t)s.d->ref_stride + (this_mv.col >> 3); // svt_get_buf_from_mv: floor
        const int thismse = (int)wave_sub_pixel_variance(ref, s.d->ref_stride, this_mv.col & 7, this_mv.row & 7, s.src, s.d->src_stride, s.jb.width, s.jb.height, sse);
        cost += (uint32_t)thismse;
        int weight = 100;
        if (s.d->bias_fp && best_mv.col % 8 == 0 && best_mv.row % 8 == 0) weight = s.d->bias_fp;
        if ((((u64)cost * (u64)(i64)weight) / 100) < besterr) { besterr = cost; best_mv = this_mv; distortion = thismse; sse1 = sse; }
    } else
        cost = INT_MAX;
    return cost;
}

// svt_check_better (mcomp.c:210-238): the accurate search's test of one candidate
__device__ __forceinline__ uint32_t sp_check_accurate(const SpCtx &s, SvtHipMv this_mv, SvtHipMv &best_mv, uint32_t &besterr, uint32_t &sse1, int &distortion, int &is_better) {
    uint32_t cost;
    if (sp_in_range(s, this_mv)) {
        uint32_t sse;
        const uint8_t *ref = s.ref + (ptrdiff_t)(this_mv.row >> 3) * (ptrdiff_t)s.d->ref_stride + (this_mv.col >> 3);
        const int thismse = (int)wave_upsampled_variance(ref, s.d->ref_stride, this_mv.col & 7, this_mv.row & 7, s.d->subpel_search_type, s.src, s.d->src_stride, s.jb.width, s.jb.height, sse);
        cost = (uint32_t)mv_err_cost(this_mv.row, this_mv.col, s.mc) + (uint32_t)thismse;
        int weight = 100;
        if (s.d->bias_fp && best_mv.col % 8 == 0 && best_mv.row % 8 == 0) weight = s.d->bias_fp;
        if ((((u64)cost * (u64)(i64)weight) / 100) < besterr) { besterr = cost; best_mv = this_mv; distortion = thismse; sse1 = sse; is_better |= 1; }
    } else
        cost = INT_MAX;
    return cost;
}

__device__ __forceinline__ SvtHipMv mv_of(int row, int col) { SvtHipMv m; m.row = (int16_t)row; m.col = (int16_t)col; return m; }

__global__ void __launch_bounds__(64) md_subpel_kernel(const SubpelParams p) {
    const uint32_t job = blockIdx.x;
    const SvtHipSubpelBatchDesc &d = p.d;
    SpCtx s;
    s.d = &d; s.jb = d.jobs[job];
    s.mc.ref_mv = s.jb.ref_mv; s.mc.type = d.mv_cost_type; s.mc.error_per_bit = d.error_per_bit; s.mc.mvjcost = d.mvjcost; s.mc.row = d.mvcost[0]; s.mc.col = d.mvcost[1];
    s.src = d.src + s.jb.src_offset; s.ref = d.ref + s.jb.ref_offset;
    const int w = s.jb.width, h = s.jb.height;
    SvtHipMv start_mv = s.jb.start_mv, bestmv = start_mv;
    int      distortion = 0, hstep = 4; // INIT_SUBPEL_STEP_SIZE
    uint32_t sse1 = 0, besterr, org_error;
    bool     done = false;
    const uint8_t *ref0 = s.ref + (ptrdiff_t)(bestmv.row >> 3) * (ptrdiff_t)d.ref_stride + (bestmv.col >> 3);
    { // svt_upsampled_setup_center_error (mcomp.c:353-360)
        uint32_t sse;
        distortion = (int)wave_sub_pixel_variance(ref0, d.ref_stride, 0, 0, s.src, d.src_stride, w, h, sse);
        besterr    = (uint32_t)distortion + (uint32_t)mv_err_cost(bestmv.row, bestmv.col, s.mc);
    }
    if (d.center_err && threadIdx.x == 0) d.center_err[job] = besterr; // what the functions leave in ctx->fp_me_dist
    const bool accurate = d.search_method == 1; // svt_av1_find_best_sub_pixel_tree (mcomp.c:688-777)
    int round = (3 - d.forced_stop) < (3 - !d.allow_hp) ? (3 - d.forced_stop) : (3 - !d.allow_hp); // FULL_PEL = 3
    if (accurate && d.mvp_th > 0) { // its PD_PASS_1 / SPEL_ME branch (:702-722)
        const int     mvp_err = (int)s.jb.best_mvp_dist + 1, me_err = (int)besterr + 1;
        const int32_t deviation = ((me_err - mvp_err) * 100) / me_err;
        const int     dc = bestmv.col - s.jb.best_mvp.col, dr = bestmv.row - s.jb.best_mvp.row;
        if (deviation >= d.mvp_th) round = 1;
        else if ((dc < 0 ? -dc : dc) > d.hp_mv_th || (dr < 0 ? -dr : dr) > d.hp_mv_th) round = round < 2 ? round : 2;
    }
    if (s.jb.early_neigh_check_exit) done = true;
    if (!done && !accurate) { // (the accurate search tests the prediction's variance first, and normalises by w h / 4)
        const u64 th_normalizer = (u64)(i64)(((w * h) >> 3) * (int)(uint8_t)d.abs_th_mult * (d.qp >> 1));
        if (besterr < th_normalizer) done = true;
    }
    if (!round && !accurate) done = true;
    if (!done) { // variance of the full-pel prediction itself (against the constant 128: svt_aom_eb_av1_var_offs with stride 0)
        int      sum = 0;
        uint32_t sse = 0;
        const int n = w * h, lw = 31 - __clz(w);
        for (int i = threadIdx.x; i < n; i += 64) {
            const int dd = (int)ref0[(size_t)(i >> lw) * d.ref_stride + (i & (w - 1))] - 128;
            sum += dd;
            sse += (uint32_t)(dd * dd);
        }
        sum = wave_sum_i32(sum);
        sse = wave_sum_u32(sse);
        const uint32_t var = sse - (uint32_t)(((i64)sum * sum) / n);
        const int block_var = (int)((var + ((1u << s.jb.log2_pels) >> 1)) >> s.jb.log2_pels);
        if (block_var < d.pred_variance_th) done = true;
    }
    if (!done && accurate) {
        const u64 th_normalizer = (u64)(i64)(((w * h) >> 2) * (int)(uint8_t)d.abs_th_mult * (d.qp >> 1));
        if (besterr < th_normalizer || !round) done = true;
        for (int iter = 0; iter < round && !done; ++iter) {
            const SvtHipMv c = bestmv; // iter_center_mv
            int dummy = 0;
            // svt_first_level_check (:260-287)
            const uint32_t left  = sp_check_accurate(s, mv_of(c.row, c.col - hstep), bestmv, besterr, sse1, distortion, dummy);
            const uint32_t right = sp_check_accurate(s, mv_of(c.row, c.col + hstep), bestmv, besterr, sse1, distortion, dummy);
            const uint32_t up    = sp_check_accurate(s, mv_of(c.row - hstep, c.col), bestmv, besterr, sse1, distortion, dummy);
            const uint32_t down  = sp_check_accurate(s, mv_of(c.row + hstep, c.col), bestmv, besterr, sse1, distortion, dummy);
            SvtHipMv diag_step = mv_of(up <= down ? -hstep : hstep, left <= right ? -hstep : hstep);
            sp_check_accurate(s, mv_of(c.row + diag_step.row, c.col + diag_step.col), bestmv, besterr, sse1, distortion, dummy);
            if (!(c.row == bestmv.row && c.col == bestmv.col) && d.iters_per_step > 1) { // svt_second_level_check_v2 (:292-350)
                if (c.row == bestmv.row) diag_step.row = (int16_t)-diag_step.row;
                else if (c.col == bestmv.col) diag_step.col = (int16_t)-diag_step.col;
                const SvtHipMv row_bias = mv_of(bestmv.row + diag_step.row, bestmv.col), col_bias = mv_of(bestmv.row, bestmv.col + diag_step.col),
                               diag_bias = mv_of(bestmv.row + diag_step.row, bestmv.col + diag_step.col);
                int has_better = 0;
                sp_check_accurate(s, row_bias, bestmv, besterr, sse1, distortion, has_better);
                sp_check_accurate(s, col_bias, bestmv, besterr, sse1, distortion, has_better);
                if (has_better) sp_check_accurate(s, diag_bias, bestmv, besterr, sse1, distortion, has_better);
            }
            hstep >>= 1;
        }
        done = true;
    }
    if (!done) {
        const int sdr = (uint8_t)d.skip_diag_refinement;
        if (sdr >= 4)
            org_error = 0;
        else {
            const uint32_t demo = sdr >= 2 ? ((w >= 64 || h >= 64) ? 2 : 1) : 1;
            org_error           = sdr ? besterr / demo : (uint32_t)INT_MAX;
        }
        for (int iter = 0; iter < round; ++iter) {
            const uint32_t prev_besterr = besterr;
            { // two_level_checks_fast (mcomp.c:542-585)
                const SvtHipMv this_mv = start_mv;
                // first_level_check_fast (:371-417)
                const uint32_t left  = sp_check(s, mv_of(this_mv.row, this_mv.col - hstep), bestmv, besterr, sse1, distortion);
                const uint32_t right = sp_check(s, mv_of(this_mv.row, this_mv.col + hstep), bestmv, besterr, sse1, distortion);
                const uint32_t up    = sp_check(s, mv_of(this_mv.row - hstep, this_mv.col), bestmv, besterr, sse1, distortion);
                const uint32_t down  = sp_check(s, mv_of(this_mv.row + hstep, this_mv.col), bestmv, besterr, sse1, distortion);
                const SvtHipMv diag_step = mv_of(up <= down ? -hstep : hstep, left <= right ? -hstep : hstep);
                if (!(besterr >= org_error)) sp_check(s, mv_of(this_mv.row + diag_step.row, this_mv.col + diag_step.col), bestmv, besterr, sse1, distortion);
                if (besterr < org_error && d.iters_per_step > 1) { // second_level_check_fast (:421-539)
                    const int tr = this_mv.row, tc = this_mv.col, br = bestmv.row, bc = bestmv.col;
                    if (tr != br && tc != bc) {
                        sp_check(s, mv_of(br, bc + diag_step.col), bestmv, besterr, sse1, distortion);
                        sp_check(s, mv_of(br + diag_step.row, bc), bestmv, besterr, sse1, distortion);
                    } else if (tr == br && tc != bc) {
                        sp_check(s, mv_of(br + hstep, bc + diag_step.col), bestmv, besterr, sse1, distortion);
                        sp_check(s, mv_of(br - hstep, bc + diag_step.col), bestmv, besterr, sse1, distortion);
                        sp_check(s, mv_of(br - diag_step.row, bc), bestmv, besterr, sse1, distortion);
                    } else if (tr != br && tc == bc) {
                        sp_check(s, mv_of(br + diag_step.row, bc + hstep), bestmv, besterr, sse1, distortion);
                        sp_check(s, mv_of(br + diag_step.row, bc - hstep), bestmv, besterr, sse1, distortion);
                        sp_check(s, mv_of(br, bc - diag_step.col), bestmv, besterr, sse1, distortion);
                    }
                }
            }
            hstep >>= 1;
            start_mv = bestmv;
            if (sdr && iter < 1 /* QUARTER_PEL */) org_error = org_error < besterr ? org_error : besterr;
            const i64 a = besterr > 1 ? besterr : 1, b = prev_besterr > 1 ? prev_besterr : 1;
            const int32_t deviation = (int32_t)(((a - b) * 100) / b);
            if (deviation >= d.round_dev_th) break;
        }
    }
    if (threadIdx.x == 0) {
        d.best_mv[2 * job]     = bestmv.row;
        d.best_mv[2 * job + 1] = bestmv.col;
        d.besterr[job]         = besterr;
        d.distortion[job]      = distortion;
        d.sse[job]             = sse1;
    }
}

} // namespace

extern "C" int svt_hip_md_fullpel_batch(SvtHipContext *ctx, const SvtHipFullpelBatchDesc *d) {
    if (!ctx || !d) return SVT_HIP_ERR_BAD_PARAM;
    if (d->n_jobs == 0) return SVT_HIP_OK;
    if (!d->src || !d->ref || !d->jobs || !d->best_cost || !d->best_mv) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "md full-pel batch: a mandatory pointer is null");
    if (d->mv_cost_type < 0 || d->mv_cost_type > SVT_HIP_MV_COST_NONE) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "md full-pel batch: mv_cost_type %d", d->mv_cost_type);
    if (d->mv_cost_type == SVT_HIP_MV_COST_ENTROPY && (!d->mvjcost || !d->mvcost[0] || !d->mvcost[1]))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "md full-pel batch: MV_COST_ENTROPY needs the joint and component cost tables");
    hipSetDevice(ctx->device);
    FullpelParams p;
    p.d = *d;
    hipLaunchKernelGGL(md_fullpel_kernel, dim3(d->n_jobs), dim3(64), 0, ctx->stream, p);
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int svt_hip_md_subpel_batch(SvtHipContext *ctx, const SvtHipSubpelBatchDesc *d) {
    if (!ctx || !d) return SVT_HIP_ERR_BAD_PARAM;
    if (d->n_jobs == 0) return SVT_HIP_OK;
    if (!d->src || !d->ref || !d->jobs || !d->best_mv || !d->besterr || !d->distortion || !d->sse)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "md sub-pel batch: a mandatory pointer is null");
    if (d->mv_cost_type < 0 || d->mv_cost_type > SVT_HIP_MV_COST_NONE || d->forced_stop < 0 || d->forced_stop > 3)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "md sub-pel batch: mv_cost_type %d / forced_stop %d", d->mv_cost_type, d->forced_stop);
    if (d->search_method < 0 || d->search_method > 1 || (d->search_method == 1 && (d->subpel_search_type < SVT_HIP_USE_2_TAPS || d->subpel_search_type > SVT_HIP_USE_8_TAPS)))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "md sub-pel batch: search_method %d / subpel_search_type %d", d->search_method, d->subpel_search_type);
    if (d->mv_cost_type == SVT_HIP_MV_COST_ENTROPY && (!d->mvjcost || !d->mvcost[0] || !d->mvcost[1]))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "md sub-pel batch: MV_COST_ENTROPY needs the joint and component cost tables");
    hipSetDevice(ctx->device);
    SubpelParams p;
    p.d = *d;
    hipLaunchKernelGGL(md_subpel_kernel, dim3(d->n_jobs), dim3(64), 0, ctx->stream, p);
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}
