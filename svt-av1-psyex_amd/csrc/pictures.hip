// pictures.hip -- context, HBM-resident luma pyramids (EbPaReferenceObject equivalents) and the 2x2
// decimation + edge padding kernels (reference: Codec/pic_analysis_process.c:130-158,2139-2196,
// Codec/pic_operators.c:397-443).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "svt_hip_internal.h"

namespace {

constexpr size_t kFrontSlack = 256, kBackSlack = 512;

// One thread per output byte of the PADDED destination plane: coordinates are clamped into the picture, so the
// padding replicates the decimated edge samples exactly as svt_aom_generate_padding does after downsample_2d.
__global__ void downsample2x_pad_kernel(DevPlane src, DevPlane dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x; // padded coordinates
    const int y = blockIdx.y;
    const int pw = dst.width + 2 * dst.org_x, ph = dst.height + 2 * dst.org_y;
    if (x >= pw || y >= ph) return;
    int ix = x - dst.org_x, iy = y - dst.org_y;
    ix = ix < 0 ? 0 : (ix > dst.width - 1 ? dst.width - 1 : ix);
    iy = iy < 0 ? 0 : (iy > dst.height - 1 ? dst.height - 1 : iy);
    const uint8_t *s = src.base + (size_t)(src.org_y + 2 * iy) * src.stride + src.org_x + 2 * ix;
    const uint32_t sum = (uint32_t)s[0] + s[1] + s[src.stride] + s[src.stride + 1];
    const_cast<uint8_t *>(dst.base)[(size_t)y * dst.stride + x] = (uint8_t)((sum + 2) >> 2);
}

int alloc_plane(SvtHipContext *ctx, int width, int height, int org_x, int org_y, DevPlane *pl, void **mem, size_t *bytes) {
    const uint32_t stride = (uint32_t)(((width + 2 * org_x + 63) & ~63) + 64);
    const size_t   rows   = (size_t)height + 2 * (size_t)org_y;
    *bytes                = kFrontSlack + (size_t)stride * rows + kBackSlack;
    if (hipMalloc(mem, *bytes) != hipSuccess)
        return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "hipMalloc(%zu) failed", *bytes);
    pl->base   = static_cast<uint8_t *>(*mem) + kFrontSlack;
    pl->stride = stride;
    pl->org_x  = org_x;
    pl->org_y  = org_y;
    pl->width  = width;
    pl->height = height;
    return SVT_HIP_OK;
}

int check_plane(SvtHipContext *ctx, const SvtHipPlaneDesc *p, int min_pad, const char *what) {
    if (!p || !p->buffer_y) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "%s plane: null", what);
    if (p->width == 0 || p->height == 0 || p->org_x < min_pad || p->org_y < min_pad ||
        p->stride_y < (uint32_t)p->width + 2u * p->org_x)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "%s plane: %ux%u org %u,%u stride %u (needs padding >= %d on every side)", what,
                            p->width, p->height, p->org_x, p->org_y, p->stride_y, min_pad);
    return SVT_HIP_OK;
}

int upload_plane(SvtHipContext *ctx, const SvtHipPlaneDesc *h, DevPlane *d, bool from_device) {
    const size_t w = (size_t)h->width + 2 * (size_t)h->org_x, rows = (size_t)h->height + 2 * (size_t)h->org_y;
    SVT_HIP_CHECK(ctx, hipMemcpy2DAsync(const_cast<uint8_t *>(d->base), d->stride, h->buffer_y, h->stride_y, w, rows,
                                        from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    return SVT_HIP_OK;
}

int make_level(SvtHipContext *ctx, SvtHipPaPicture *pic, int level, const SvtHipPlaneDesc *host) {
    const DevPlane &src = pic->pyr.lvl[level + 1];
    const int       pad = (level == 1) ? 32 : 16; // b64_size >> 1, b64_size >> 2 (Globals/enc_handle.c:1260-1279)
    int rc = alloc_plane(ctx, src.width >> 1, src.height >> 1, pad, pad, &pic->pyr.lvl[level], &pic->mem[level], &pic->bytes[level]);
    if (rc) return rc;
    DevPlane &dst = pic->pyr.lvl[level];
    if (host) {
        if (host->width != dst.width || host->height != dst.height || host->org_x != pad || host->org_y != pad)
            return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "level %d plane geometry %ux%u pad %u,%u does not match %dx%d pad %d", level,
                                host->width, host->height, host->org_x, host->org_y, dst.width, dst.height, pad);
        return upload_plane(ctx, host, &dst, false);
    }
    const int pw = dst.width + 2 * pad, ph = dst.height + 2 * pad;
    hipLaunchKernelGGL(downsample2x_pad_kernel, dim3((pw + 255) / 256, ph), dim3(256), 0, ctx->stream, src, dst);
    SVT_HIP_CHECK(ctx, hipGetLastError());
    return SVT_HIP_OK;
}

int create_common(SvtHipContext *ctx, const SvtHipPlaneDesc *full, const SvtHipPlaneDesc *quarter, const SvtHipPlaneDesc *sixteenth,
                  bool full_on_device, SvtHipPaPicture **out) {
    if (!ctx || !out) return SVT_HIP_ERR_BAD_PARAM;
    int rc = check_plane(ctx, full, 64, "full");
    if (rc) return rc;
    if (quarter && (rc = check_plane(ctx, quarter, 32, "quarter"))) return rc;
    if (sixteenth && (rc = check_plane(ctx, sixteenth, 16, "sixteenth"))) return rc;
    if ((full->width & 7) || (full->height & 7))
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "full plane %ux%u: width and height must be multiples of 8", full->width, full->height);
    SvtHipPaPicture *pic = static_cast<SvtHipPaPicture *>(calloc(1, sizeof(SvtHipPaPicture)));
    if (!pic) return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "calloc");
    hipSetDevice(ctx->device);
    rc = alloc_plane(ctx, full->width, full->height, full->org_x, full->org_y, &pic->pyr.lvl[2], &pic->mem[2], &pic->bytes[2]);
    if (!rc) rc = upload_plane(ctx, full, &pic->pyr.lvl[2], full_on_device);
    if (!rc) rc = make_level(ctx, pic, 1, quarter);
    if (!rc) rc = make_level(ctx, pic, 0, sixteenth);
    if (rc) {
        svt_hip_pa_picture_destroy(ctx, pic);
        return rc;
    }
    *out = pic;
    return SVT_HIP_OK;
}

} // namespace

extern "C" {

int svt_hip_context_create(SvtHipContext **out, int device) {
    if (!out) return SVT_HIP_ERR_BAD_PARAM;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SVT_HIP_ERR_NO_DEVICE;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return SVT_HIP_ERR_NO_DEVICE;
    if (device >= n) return SVT_HIP_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SVT_HIP_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SVT_HIP_ERR_NO_DEVICE; // code objects are gfx950 only
    if (hipSetDevice(device) != hipSuccess) return SVT_HIP_ERR_NO_DEVICE;
    SvtHipContext *ctx = static_cast<SvtHipContext *>(calloc(1, sizeof(SvtHipContext)));
    if (!ctx) return SVT_HIP_ERR_NO_MEMORY;
    ctx->device  = device;
    ctx->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&ctx->queue_head), 512) != hipSuccess || hipMemset(ctx->queue_head, 0, 512) != hipSuccess ||
        hipMalloc(&ctx->me_params, SVT_HIP_ME_HEADER_BYTES + sizeof(MeKernelParams) * SVT_HIP_ME_MAX_PICTURES) != hipSuccess) {
        free(ctx);
        return SVT_HIP_ERR_NO_DEVICE;
    }
    *out = ctx;
    return SVT_HIP_OK;
}

void svt_hip_context_destroy(SvtHipContext *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) hipFree(ctx->scratch);
    hipFree(ctx->queue_head);
    hipFree(ctx->me_params);
    hipStreamDestroy(ctx->stream);
    free(ctx);
}

const char *svt_hip_last_error(const SvtHipContext *ctx) { return ctx ? ctx->err : "no context"; }
void       *svt_hip_context_stream(SvtHipContext *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int svt_hip_context_sync(SvtHipContext *ctx) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    SVT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SVT_HIP_OK;
}

int svt_hip_pa_picture_create(SvtHipContext *ctx, const SvtHipPlaneDesc *full, const SvtHipPlaneDesc *quarter,
                              const SvtHipPlaneDesc *sixteenth, SvtHipPaPicture **pic) {
    return create_common(ctx, full, quarter, sixteenth, false, pic);
}

int svt_hip_pa_picture_create_dev(SvtHipContext *ctx, const SvtHipPlaneDesc *full_dev, SvtHipPaPicture **pic) {
    return create_common(ctx, full_dev, nullptr, nullptr, true, pic);
}

void svt_hip_pa_picture_destroy(SvtHipContext *ctx, SvtHipPaPicture *pic) {
    if (!pic) return;
    if (ctx) {
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
    }
    for (int l = 0; l < 3; l++)
        if (pic->mem[l]) hipFree(pic->mem[l]);
    free(pic);
}

int svt_hip_pa_picture_geometry(const SvtHipPaPicture *pic, int level, SvtHipPlaneDesc *out) {
    if (!pic || !out || level < 0 || level > 2) return SVT_HIP_ERR_BAD_PARAM;
    const DevPlane &p = pic->pyr.lvl[level];
    out->buffer_y = p.base;
    out->stride_y = p.stride;
    out->org_x    = (uint16_t)p.org_x;
    out->org_y    = (uint16_t)p.org_y;
    out->width    = (uint16_t)p.width;
    out->height   = (uint16_t)p.height;
    return SVT_HIP_OK;
}

int svt_hip_pa_picture_download(SvtHipContext *ctx, const SvtHipPaPicture *pic, int level, uint8_t *dst, uint32_t dst_stride) {
    if (!ctx || !pic || !dst || level < 0 || level > 2) return SVT_HIP_ERR_BAD_PARAM;
    const DevPlane &p = pic->pyr.lvl[level];
    const size_t    w = (size_t)p.width + 2 * (size_t)p.org_x, rows = (size_t)p.height + 2 * (size_t)p.org_y;
    if (dst_stride < w) return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "dst_stride %u < %zu", dst_stride, w);
    SVT_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dst_stride, p.base, p.stride, w, rows, hipMemcpyDeviceToHost, ctx->stream));
    SVT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SVT_HIP_OK;
}

} // extern "C"

int svt_hip_scratch(SvtHipContext *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            SVT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(ctx->scratch);
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        if (hipMalloc(&ctx->scratch, bytes) != hipSuccess) return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "hipMalloc(%zu) failed", bytes);
        ctx->scratch_bytes = bytes;
    }
    *out = ctx->scratch;
    return SVT_HIP_OK;
}

// Diagnostic: per-phase shader-clock sums of the ME kernel (non-zero only in a -DSVT_HIP_ME_PROFILE build); clears them.
extern "C" int svt_hip_me_profile_read(SvtHipContext *ctx, unsigned long long out[24]) {
    if (!ctx || !out) return SVT_HIP_ERR_BAD_PARAM;
    SVT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    SVT_HIP_CHECK(ctx, hipMemcpy(out, reinterpret_cast<char *>(ctx->queue_head) + 64, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    SVT_HIP_CHECK(ctx, hipMemset(reinterpret_cast<char *>(ctx->queue_head) + 64, 0, 24 * sizeof(unsigned long long)));
    return SVT_HIP_OK;
}
