// dg_kernel.hip -- dynamic-GOP detector level-0 HME (the ME kernel's ME_DG_DETECTOR flavour) for gfx950.
//
// Replaces dg_detector_hme_level0 + early_hme_b64 (reference: Codec/pd_process.c:393-588, dispatched from
// Codec/me_process.c:326-331): per b64, an exhaustive svt_sad_loop_kernel search (C_DEFAULT/compute_sad_c.c:58-101) of the
// 16x16 block of the source's sixteenth plane over a 16 / 64 / 128-squared window of the reference's sixteenth plane, then
// four picture-level sums.  One workgroup per b64: the window is staged once in LDS (<= 143 rows of 160 bytes), every lane
// evaluates 4 neighbouring positions of 4 consecutive rows per step with v_qsad_pk_u16_u8, the running best is a (sad, y, x) key so that the first
// minimum in raster order wins exactly as the reference's strict `<` does; the sums are integer atomics (order-free).
#include <hip/hip_runtime.h>
#include "svt_hip_internal.h"
#include "../../include/svt_hip_me.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPitch   = 176;      // bytes per staged window row: 11 x 16 (odd multiple of 16 -> rows spread over the LDS banks)
constexpr int kMaxSide = 128;      // largest search side (pd_process.c:497-498)
constexpr int kYs      = 4;             // search rows per item
constexpr int kRows    = kMaxSide + 15 + kYs; // + the rows a partial last group reads past the staged window

struct DgParams {
    DevPlane         src, ref;     // sixteenth planes
    int              w64, h64, side;
    SvtHipDgMetrics *metrics;
    uint32_t        *b64_sad;      // may be null
    int16_t         *b64_mv;       // may be null
};

// early_hme_b64's clipping of one axis (pd_process.c:413-449): the low edge moves the origin only, the high edge moves
// the origin and then crops the size (never below 1)
__device__ __forceinline__ void clip_axis(int org, int &origin, int &size, int pad, int dim) {
    if (org + origin < -pad) origin = -pad - org;
    if (org + origin > dim - 1) origin -= (org + origin) - (dim - 1);
    if (org + origin + size > dim) {
        const int cropped = size - ((org + origin + size) - dim);
        size              = cropped > 1 ? cropped : 1;
    }
}

// Search of the staged window: item = (group of YS consecutive search rows, quad of columns 4q .. 4q+3).  The YS + 15 window rows of a
// group are read once and feed all its rows (95 LDS reads per 256 qsads at YS = 4, instead of 320).  Returns this lane's best key
// (sad << 16) | (y << 8) | x : sad <= 16*16*255 < 2^16, x and y < 128.
template <int YS> __device__ __forceinline__ uint32_t dg_search(const uint8_t *win, const uint32_t (&s)[16][4], int sa_w, int sa_h, int tid) {
    const int qpr = (sa_w + 3) >> 2, ngrp = (sa_h + YS - 1) / YS;
    const float rq = __builtin_amdgcn_rcpf((float)qpr);
    uint32_t  best = 0xFFFFFFFFu;
    for (int i = tid; i < ngrp * qpr; i += kThreads) {
        const int g = (int)(((float)i + 0.5f) * rq), q = i - g * qpr, y0 = g * YS; // exact: i < 2^21
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&win[y0 * kPitch + q * 4]);
        unsigned long long acc[YS]; // 4 x u16 each: 64 qsads x 4 x 255 = 65280 cannot overflow a lane
#pragma unroll
        for (int yy = 0; yy < YS; yy++) acc[yy] = 0;
#pragma unroll
        for (int wr = 0; wr < YS + 15; wr++) {
            uint32_t wv[5];
#pragma unroll
            for (int j = 0; j < 5; j++) wv[j] = w[wr * (kPitch / 4) + j];
#pragma unroll
            for (int yy = 0; yy < YS; yy++) {
                const int r = wr - yy; // block row that window row wr is for search row y0 + yy
                if (r >= 0 && r < 16) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[yy] = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)wv[j + 1] << 32) | wv[j], s[r][j], acc[yy]);
                }
            }
        }
#pragma unroll
        for (int yy = 0; yy < YS; yy++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = q * 4 + k, y = y0 + yy;
                const uint32_t key = ((uint32_t)((acc[yy] >> (16 * k)) & 0xFFFF) << 16) | ((uint32_t)y << 8) | (uint32_t)x;
                if (x < sa_w && y < sa_h && key < best) best = key; // rows / columns past the area read staged-or-stale LDS: masked here
            }
    }
    return best;
}

__global__ __launch_bounds__(kThreads) void dg_hme_level0_kernel(const DgParams p) {
    __shared__ __attribute__((aligned(16))) uint8_t  win[kRows * kPitch];
    __shared__ __attribute__((aligned(16))) uint32_t blk[16 * 4];
    __shared__ uint32_t                              wave_key[kThreads / 64];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x % p.w64, by = blockIdx.x / p.w64;
    const int org_x = bx * 16, org_y = by * 16; // b64 origin >> 2 (:531-532)

    int sa_w = (p.side + 7) & ~7, sa_h = p.side; // :410
    int ox = -(sa_w >> 1), oy = -(sa_h >> 1);
    clip_axis(org_x, ox, sa_w, p.ref.org_x - 1, p.ref.width);
    sa_w = sa_w < 8 ? sa_w : sa_w & ~7; // :432
    clip_axis(org_y, oy, sa_h, p.ref.org_y - 1, p.ref.height);

    // ---- stage the source block and the window (unaligned 16-byte global reads; the planes carry slack behind each row) ----
    typedef uint32_t V4 __attribute__((ext_vector_type(4)));
    if (tid < 16) {
        const uint8_t *s = p.src.base + (long long)(p.src.org_y + org_y + tid) * p.src.stride + (p.src.org_x + org_x);
        V4 v; __builtin_memcpy(&v, s, 16);
        *reinterpret_cast<V4 *>(&blk[tid * 4]) = v;
    }
    const int rows = sa_h + 15, vpr = (sa_w + 19 + 15) >> 4; // a quad of positions reads 19 bytes past its first one
    const uint8_t *w0 = p.ref.base + (long long)(p.ref.org_y + org_y + oy) * p.ref.stride + (p.ref.org_x + org_x + ox);
    for (int i = tid; i < rows * vpr; i += kThreads) {
        const int r = i / vpr, c = i - r * vpr;
        V4 v; __builtin_memcpy(&v, w0 + (long long)r * p.ref.stride + c * 16, 16);
        *reinterpret_cast<V4 *>(&win[r * kPitch + c * 16]) = v;
    }
    __syncthreads();

    // ---- search ----
    uint32_t s[16][4];
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) s[r][j] = blk[r * 4 + j]; // same address in every lane: broadcast
    // small areas (the 16-squared one of low resolutions) have too few row groups to occupy the workgroup: one row per item there
    uint32_t best = (sa_h * ((sa_w + 3) >> 2) >= kYs * kThreads) ? dg_search<kYs>(win, s, sa_w, sa_h, tid) : dg_search<1>(win, s, sa_w, sa_h, tid);
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(best, o, 64); best = t < best ? t : best; }
    if ((tid & 63) == 0) wave_key[tid >> 6] = best;
    __syncthreads();

    // ---- results and metrics (pd_process.c:479-486, 541-581) ----
    if (tid == 0) {
        for (int wv = 1; wv < kThreads / 64; wv++) best = wave_key[wv] < best ? wave_key[wv] : best;
        const uint32_t sad = best >> 16;
        const int col = ((int)(best & 0xFF) + ox) * 4, row = ((int)((best >> 8) & 0xFF) + oy) * 4;
        atomicAdd(reinterpret_cast<unsigned long long *>(&p.metrics->tot_dist), (unsigned long long)sad);
        if (sad > 16 * 16 * 30) atomicAdd(&p.metrics->tot_cplx, 1u);
        if (col != 0 || row != 0) atomicAdd(&p.metrics->tot_active, 1u);
        const int sr = (row > 0) - (row < 0), sc = (col > 0) - (col < 0);
        int in = 0;
        if (by < p.h64 / 2) in -= sr; else if (by > p.h64 / 2) in += sr;
        if (bx < p.w64 / 2) in -= sc; else if (bx > p.w64 / 2) in += sc;
        if (in) atomicAdd(&p.metrics->sum_in_vectors, in);
        if (p.b64_sad) p.b64_sad[blockIdx.x] = sad;
        if (p.b64_mv) { p.b64_mv[2 * blockIdx.x] = (int16_t)col; p.b64_mv[2 * blockIdx.x + 1] = (int16_t)row; }
    }
}

} // namespace

// the launch on an explicit stream (the asynchronous entry: the context stream; the synchronous one: its borrowed lane)
static int dg_launch(SvtHipContext *ctx, hipStream_t stream, const SvtHipPaPicture *src, const SvtHipPaPicture *ref, uint16_t aligned_width,
                     uint16_t aligned_height, uint8_t input_resolution, SvtHipDgMetrics *metrics_dev, uint32_t *b64_sad_dev, int16_t *b64_mv_dev) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    if (!src || !ref || !metrics_dev) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "dg detector: null picture or metrics pointer");
    if (aligned_width == 0 || aligned_height == 0) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "dg detector: empty picture");
    DgParams p;
    p.src = src->pyr.lvl[0];
    p.ref = ref->pyr.lvl[0];
    p.w64 = (aligned_width + 63) / 64;
    p.h64 = (aligned_height + 63) / 64;
    // search side by resolution class (pd_process.c:497-498; INPUT_SIZE_360p_RANGE = 1, INPUT_SIZE_480p_RANGE = 2)
    p.side = input_resolution <= 1 ? 16 : input_resolution <= 2 ? 64 : kMaxSide;
    // every 16x16 block (the last column / row may hang over the picture) must lie inside the padded source plane, and the
    // clipped windows inside the padded reference plane: both hold when the planes describe this picture with >= 16 px padding
    if ((p.w64 * 64 + 3) / 4 > p.src.width + p.src.org_x || (p.h64 * 64 + 3) / 4 > p.src.height + p.src.org_y || p.ref.org_x < 16 ||
        p.ref.org_y < 16 || p.ref.width != p.src.width || p.ref.height != p.src.height)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "dg detector: sixteenth planes %dx%d / %dx%d (padding %d, %d) do not fit a %ux%u picture",
                            p.src.width, p.src.height, p.ref.width, p.ref.height, p.src.org_x, p.ref.org_x, aligned_width, aligned_height);
    p.metrics = metrics_dev;
    p.b64_sad = b64_sad_dev;
    p.b64_mv  = b64_mv_dev;
    SVT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (int rc = svt_hip_wait_picture(ctx, stream, src)) return rc;
    if (int rc = svt_hip_wait_picture(ctx, stream, ref)) return rc;
    SVT_HIP_CHECK(ctx, hipMemsetAsync(metrics_dev, 0, sizeof(SvtHipDgMetrics), stream));
    hipLaunchKernelGGL(dg_hme_level0_kernel, dim3(p.w64 * p.h64), dim3(kThreads), 0, stream, p);
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int svt_hip_dg_detector_hme_level0_async(SvtHipContext *ctx, const SvtHipPaPicture *src, const SvtHipPaPicture *ref,
                                                    uint16_t aligned_width, uint16_t aligned_height, uint8_t input_resolution,
                                                    SvtHipDgMetrics *metrics_dev, uint32_t *b64_sad_dev, int16_t *b64_mv_dev) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    return dg_launch(ctx, ctx->stream, src, ref, aligned_width, aligned_height, input_resolution, metrics_dev, b64_sad_dev, b64_mv_dev);
}

extern "C" int svt_hip_dg_detector_hme_level0(SvtHipContext *ctx, const SvtHipPaPicture *src, const SvtHipPaPicture *ref, uint16_t aligned_width,
                                              uint16_t aligned_height, uint8_t input_resolution, SvtHipDgMetrics *metrics, uint32_t *b64_sad,
                                              int16_t *b64_mv) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    if (!metrics) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "dg detector: null metrics pointer");
    const size_t n = (size_t)((aligned_width + 63) / 64) * ((aligned_height + 63) / 64);
    SvtHipLaneGuard guard(ctx); // own stream and result buffer: callable from several host threads at once
    SvtHipLane     *lane = guard.lane();
    if (!lane) return SVT_HIP_ERR_NO_MEMORY;
    void *scratch = nullptr;
    const size_t off_sad = 64, off_mv = off_sad + n * sizeof(uint32_t);
    if (int rc = svt_hip_scratch(ctx, lane, off_mv + n * 2 * sizeof(int16_t), &scratch)) return rc;
    uint8_t *d = static_cast<uint8_t *>(scratch);
    if (int rc = dg_launch(ctx, lane->stream, src, ref, aligned_width, aligned_height, input_resolution, reinterpret_cast<SvtHipDgMetrics *>(d),
                           reinterpret_cast<uint32_t *>(d + off_sad), reinterpret_cast<int16_t *>(d + off_mv)))
        return rc;
    SVT_HIP_CHECK(ctx, hipMemcpyAsync(metrics, d, sizeof(SvtHipDgMetrics), hipMemcpyDeviceToHost, lane->stream));
    if (b64_sad) SVT_HIP_CHECK(ctx, hipMemcpyAsync(b64_sad, d + off_sad, n * sizeof(uint32_t), hipMemcpyDeviceToHost, lane->stream));
    if (b64_mv) SVT_HIP_CHECK(ctx, hipMemcpyAsync(b64_mv, d + off_mv, n * 2 * sizeof(int16_t), hipMemcpyDeviceToHost, lane->stream));
    SVT_HIP_CHECK(ctx, hipStreamSynchronize(lane->stream));
    return SVT_HIP_OK;
}
