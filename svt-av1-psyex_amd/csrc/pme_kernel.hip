// pme_kernel.hip -- the mode-decision full-pel refinement search (include/svt_hip_pme.h): svt_pme_sad_loop_kernel
// (reference: Codec/product_coding_loop.c:1905-1950; MV rate: Codec/mcomp.c:44-78, mcomp.h:135-138, rd_cost.c:55-60).
//
// One wave per job.  The positions the reference visits (rows `step` apart; along a row groups of 8 consecutive columns, the groups
// 7 + step apart) are cut into quads; lane <-> quad: v_qsad_pk_u16_u8 evaluates its 4 positions x 4 pixels per instruction over the
// block (source and reference dwords straight from global memory / L1: the windows of one job are small, many jobs are in flight).
// Then each position's cost = SAD + MV rate; the job's first strict minimum in visiting order -- a 64-bit key (cost, visit number) reduced
// over the wave -- replaces the incoming best when it beats it.
#include <hip/hip_runtime.h>
#include <mutex>
#include <string.h>
#include "svt_hip_internal.h"
#include "leaf_guard.h"
#include "../../include/svt_hip_pme.h"

namespace {
typedef unsigned long long u64;
typedef long long          i64;

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

// svt_mv_err_cost (mcomp.c:44-69) of the int16 vector (row, col) against ref_mv
__device__ __forceinline__ int mv_err_cost(int16_t row, int16_t col, SvtHipMv ref_mv, int type, int error_per_bit, const int32_t *mvjcost, const int32_t *cost_row,
                                           const int32_t *cost_col) {
    const int16_t dr = (int16_t)(row - ref_mv.row), dc = (int16_t)(col - ref_mv.col); // MV fields are int16
    const int16_t ar = (int16_t)(dr < 0 ? -dr : dr), ac = (int16_t)(dc < 0 ? -dc : dc);
    switch (type) {
    case SVT_HIP_MV_COST_ENTROPY: {
        const int joint = dr == 0 ? (dc == 0 ? 0 : 1) : (dc == 0 ? 2 : 3); // svt_av1_get_mv_joint
        const int bits  = mvjcost[joint] + cost_row[clip3(-(1 << 14), 1 << 14, dr)] + cost_col[clip3(-(1 << 14), 1 << 14, dc)];
        return (int)((((i64)bits * error_per_bit) + ((i64)1 << 13)) >> 14); // ROUND_POWER_OF_TWO_64(.., RDDIV_BITS + AV1_PROB_COST_SHIFT - RD_EPB_SHIFT + 4)
    }
    case SVT_HIP_MV_COST_L1_LOWRES: return (2 * (ar + ac)) >> 3;
    case SVT_HIP_MV_COST_L1_MIDRES: return 0;
    case SVT_HIP_MV_COST_L1_HDRES: return (ar + ac) >> 3;
    case SVT_HIP_MV_COST_OPT: return (int)((((i64)((ar + ac) << 8) * error_per_bit) + ((i64)1 << 13)) >> 14);
    default: return 0;
    }
}

struct PmeParams { SvtHipPmeBatchDesc d; };

__global__ void __launch_bounds__(64) pme_sad_kernel(const PmeParams p) {
    const uint32_t job = blockIdx.x;
    const int      lane = threadIdx.x;
    const SvtHipPmeJob jb = p.d.jobs[job];
    const int bw = jb.width, bh = jb.height, step = jb.step < 1 ? 1 : jb.step;
    const int n_groups = jb.sa_w >= 8 ? (jb.sa_w - 8) / (7 + step) + 1 : 0; // group k starts at column k * (7 + step) while 8 columns remain
    const int n_rows   = jb.sa_h > 0 ? (jb.sa_h - 1) / step + 1 : 0;
    const int n_items  = n_groups * 2 * n_rows;                               // quads
    const uint8_t *src = p.d.src + jb.src_offset, *ref = p.d.ref + jb.ref_offset;
    u64 best = ~0ull;
    for (int it = lane; it < n_items; it += 64) {
        const int yi = it / (n_groups * 2), q = it - yi * (n_groups * 2), g = q >> 1, half = q & 1;
        const int xs = g * (7 + step) + 4 * half, ys = yi * step;
        const uint8_t *r0 = ref + (size_t)ys * p.d.ref_stride + xs;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int r = 0; r < bh; r++) {
            const uint8_t *sp = src + (size_t)r * p.d.src_stride, *rp = r0 + (size_t)r * p.d.ref_stride;
            u64      acc = 0; // at most 32 qsads per row (128 pixels): 32 x 4 x 255 < 65536
            uint32_t lo;
            memcpy(&lo, rp, 4);
            for (int j = 0; j < (bw >> 2); j++) {
                uint32_t hi, s;
                memcpy(&hi, rp + 4 * j + 4, 4);
                memcpy(&s, sp + 4 * j, 4);
                acc = __builtin_amdgcn_qsad_pk_u16_u8(((u64)hi << 32) | lo, s, acc);
                lo  = hi;
            }
            a0 += (uint32_t)(acc & 0xFFFF); a1 += (uint32_t)((acc >> 16) & 0xFFFF); a2 += (uint32_t)((acc >> 32) & 0xFFFF); a3 += (uint32_t)(acc >> 48);
        }
        const uint32_t sad[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t px = (uint32_t)(jb.start_x + xs + i), py = (uint32_t)(jb.start_y + ys); // refinement_pos_x / _y (uint32 in the reference)
            const int16_t  col = (int16_t)(jb.mvx + (px * 8)), row = (int16_t)(jb.mvy + (py * 8));
            const uint32_t cost = sad[i] + (uint32_t)mv_err_cost(row, col, jb.ref_mv, p.d.mv_cost_type, p.d.error_per_bit, p.d.mvjcost, p.d.mvcost[0], p.d.mvcost[1]);
            const u64      key  = ((u64)cost << 32) | (uint32_t)((yi * n_groups + g) * 8 + 4 * half + i); // visiting order: rows, then groups, then columns
            best = key < best ? key : best;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u64 t = __shfl_xor(best, o, 64); best = t < best ? t : best; }
    if (lane == 0) {
        uint32_t out_cost = jb.best_cost;
        int16_t  out_x = jb.best_mvx, out_y = jb.best_mvy;
        if (best != ~0ull && (uint32_t)(best >> 32) < jb.best_cost) { // strict `<` against what the caller brought
            const int ord = (int)(uint32_t)best, yi = ord / (n_groups * 8), rem = ord - yi * (n_groups * 8), g = rem >> 3, i = rem & 7;
            const uint32_t px = (uint32_t)(jb.start_x + g * (7 + step) + i), py = (uint32_t)(jb.start_y + yi * step);
            out_cost = (uint32_t)(best >> 32);
            out_x    = (int16_t)(jb.mvx + (px * 8));
            out_y    = (int16_t)(jb.mvy + (py * 8));
        }
        p.d.best_cost[job]       = out_cost;
        p.d.best_mv[2 * job]     = out_x;
        p.d.best_mv[2 * job + 1] = out_y;
    }
}

} // namespace

extern "C" int svt_hip_pme_sad_batch(SvtHipContext *ctx, const SvtHipPmeBatchDesc *d) {
    if (!ctx || !d) return SVT_HIP_ERR_BAD_PARAM;
    if (d->n_jobs == 0) return SVT_HIP_OK;
    if (!d->src || !d->ref || !d->jobs || !d->best_cost || !d->best_mv) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "pme batch: a mandatory pointer is null");
    if (d->mv_cost_type < 0 || d->mv_cost_type > SVT_HIP_MV_COST_NONE) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "pme batch: mv_cost_type %d", d->mv_cost_type);
    if (d->mv_cost_type == SVT_HIP_MV_COST_ENTROPY && (!d->mvjcost || !d->mvcost[0] || !d->mvcost[1]))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "pme batch: MV_COST_ENTROPY needs the joint and component cost tables");
    hipSetDevice(ctx->device);
    PmeParams p;
    p.d = *d;
    hipLaunchKernelGGL(pme_sad_kernel, dim3(d->n_jobs), dim3(64), 0, ctx->stream, p);
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

// ---- pointer-level entry (the reference's prototype, host pointers, synchronous) ----
// Like every pointer-level entry it fails closed (leaf_guard.h): without a bound context or on a device error the call goes to the kernel
// the encoder had in its svt_pme_sad_loop_kernel slot.  Runs under the entries' common lock on the context stream, staging through lane 0's
// buffer.  Only the part of the MV-rate tables the reference defines travels: nmv_costs[c][MV_VALS] centred on MV_MAX (Codec/mcomp.h) --
// indices -MV_MAX .. MV_MAX; the two entries a clamp could still reach beyond them read as zero on the device.
extern "C" void svt_pme_sad_loop_kernel_hip(const SvtHipMvCostParam *mp, uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                                            uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx, int16_t *best_mvy, int16_t start_x, int16_t start_y, int16_t sa_w,
                                            int16_t sa_h, int16_t step, int16_t mvx, int16_t mvy) LEAF_TRY
    std::lock_guard<std::mutex> lock(leaf_mutex());
    SvtHipContext *ctx = leaf_ctx();
    if (sa_w < 8 || sa_h < 1 || block_height == 0 || block_width == 0) return; // no position is visited
    hipSetDevice(ctx->device);
    const int    st = step < 1 ? 1 : step;
    const size_t src_bytes = ((size_t)block_height - 1) * src_stride + block_width;
    const size_t ref_bytes = ((size_t)((sa_h - 1) / st) * st + block_height - 1) * ref_stride + block_width + sa_w + 4;
    const int    mv_max = (1 << 14) - 1, r_lo = -(1 << 14), r_hi = 1 << 14; // the table's index range on the device: [r_lo, r_hi]; the host's: [-mv_max, mv_max]
    const bool   entropy = mp->mv_cost_type == SVT_HIP_MV_COST_ENTROPY;
    const size_t tab = entropy ? (size_t)(r_hi - r_lo + 1) * sizeof(int32_t) : 0;
    const size_t a_src = 0, a_ref = (src_bytes + 255) & ~(size_t)255, a_job = a_ref + ((ref_bytes + 255) & ~(size_t)255), a_out = a_job + 256, a_j = a_out + 256,
                 a_t0 = a_j + 256, a_t1 = a_t0 + ((tab + 255) & ~(size_t)255), total = a_t1 + ((tab + 255) & ~(size_t)255);
    uint8_t *d_all = leaf_scratch(ctx, total);
    leaf_check(ctx, hipMemcpyAsync(d_all + a_src, src, src_bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipMemcpyAsync(d_all + a_ref, ref, ref_bytes - 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync"); // the last 4 bytes are the kernel's dword over-read
    SvtHipPmeJob jb;
    memset(&jb, 0, sizeof(jb));
    jb.width = (uint8_t)block_width; jb.height = (uint8_t)block_height; jb.start_x = start_x; jb.start_y = start_y; jb.sa_w = sa_w; jb.sa_h = sa_h; jb.step = step;
    jb.mvx = mvx; jb.mvy = mvy; jb.ref_mv = *mp->ref_mv; jb.best_cost = *best_cost; jb.best_mvx = *best_mvx; jb.best_mvy = *best_mvy;
    leaf_check(ctx, hipMemcpyAsync(d_all + a_job, &jb, sizeof(jb), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    SvtHipPmeBatchDesc d;
    memset(&d, 0, sizeof(d));
    d.n_jobs = 1; d.src_stride = src_stride; d.ref_stride = ref_stride; d.src = d_all + a_src; d.ref = d_all + a_ref;
    d.jobs = reinterpret_cast<const SvtHipPmeJob *>(d_all + a_job);
    d.mv_cost_type = mp->mv_cost_type; d.error_per_bit = mp->error_per_bit;
    if (entropy) {
        leaf_check(ctx, hipMemcpyAsync(d_all + a_j, mp->mvjcost, 4 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
        for (int c = 0; c < 2; c++) {
            uint8_t *t = d_all + (c ? a_t1 : a_t0);
            leaf_check(ctx, hipMemsetAsync(t, 0, tab, ctx->stream), "hipMemsetAsync");
            leaf_check(ctx, hipMemcpyAsync(t + (size_t)(-mv_max - r_lo) * sizeof(int32_t), mp->mvcost[c] - mv_max, (size_t)(2 * mv_max + 1) * sizeof(int32_t), hipMemcpyHostToDevice,
                                           ctx->stream), "hipMemcpyAsync");
        }
        d.mvjcost = reinterpret_cast<const int32_t *>(d_all + a_j);
        d.mvcost[0] = reinterpret_cast<const int32_t *>(d_all + a_t0) - r_lo;
        d.mvcost[1] = reinterpret_cast<const int32_t *>(d_all + a_t1) - r_lo;
    }
    d.best_cost = reinterpret_cast<uint32_t *>(d_all + a_out);
    d.best_mv   = reinterpret_cast<int16_t *>(d_all + a_out + 16);
    if (svt_hip_pme_sad_batch(ctx, &d) != SVT_HIP_OK) leaf_fail("%s", svt_hip_err_buf());
    struct { uint32_t cost; uint32_t pad[3]; int16_t mv[2]; } out;
    leaf_check(ctx, hipMemcpyAsync(&out, d_all + a_out, sizeof(out), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    leaf_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    *best_cost = out.cost; *best_mvx = out.mv[0]; *best_mvy = out.mv[1];
LEAF_CATCH(svt_pme_sad_loop_kernel_hip, mp, src, src_stride, ref, ref_stride, block_height, block_width, best_cost, best_mvx, best_mvy, start_x, start_y, sa_w, sa_h, step, mvx, mvy)
