// pictures.hip -- context, HBM-resident luma pyramids (EbPaReferenceObject equivalents) and the 2x2
// decimation + edge padding kernels (reference: Codec/pic_analysis_process.c:130-158,2139-2196,
// Codec/pic_operators.c:397-443).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <new>
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

int upload_plane(SvtHipContext *ctx, const SvtHipPlaneDesc *h, DevPlane *d, bool from_device, hipStream_t stream = nullptr) {
    const size_t w = (size_t)h->width + 2 * (size_t)h->org_x, rows = (size_t)h->height + 2 * (size_t)h->org_y;
    SVT_HIP_CHECK(ctx, hipMemcpy2DAsync(const_cast<uint8_t *>(d->base), d->stride, h->buffer_y, h->stride_y, w, rows,
                                        from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream ? stream : ctx->stream));
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
    // whoever reads the planes on another stream (a synchronous entry on a borrowed lane) waits for this event first
    if (!rc && hipEventCreateWithFlags(&pic->ready, hipEventDisableTiming) != hipSuccess) rc = svt_hip_fail(ctx, SVT_HIP_ERR_LAUNCH, "hipEventCreate failed");
    if (!rc && hipEventRecord(pic->ready, ctx->stream) != hipSuccess) rc = svt_hip_fail(ctx, SVT_HIP_ERR_LAUNCH, "hipEventRecord failed");
    pic->ready_stream = ctx->stream;
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
    SvtHipContext *ctx = new (std::nothrow) SvtHipContext();
    if (!ctx) return SVT_HIP_ERR_NO_MEMORY;
    ctx->device  = device;
    ctx->num_cus = prop.multiProcessorCount;
    { const char *e = getenv("SVT_HIP_ME_DENSE"); ctx->me_dense = !(e && e[0] == '0'); }
    { const char *e = getenv("SVT_HIP_ME_STAGED"); ctx->me_staged = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }
    // lane 0 (the asynchronous entries' stream) and this device's transform tables exist from the start; the borrowed
    // lanes are made when a synchronous entry first needs one
    if (svt_hip_lane_setup(ctx, &ctx->lane[0], true) != SVT_HIP_OK || svt_hip_rd_tables_init(ctx) != SVT_HIP_OK) {
        svt_hip_context_destroy(ctx);
        return SVT_HIP_ERR_NO_DEVICE;
    }
    ctx->stream    = ctx->lane[0].stream;
    ctx->lane_busy = 1u; // lane 0 is never borrowed
    *out = ctx;
    return SVT_HIP_OK;
}

void svt_hip_context_destroy(SvtHipContext *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    for (int i = 0; i < SVT_HIP_LANES; i++) {
        SvtHipLane &l = ctx->lane[i];
        if (l.stream) hipStreamSynchronize(l.stream);
        if (l.scratch) hipFree(l.scratch);
        if (l.dense) hipFree(l.dense);
        if (l.stage) hipFree(l.stage);
        for (hipEvent_t &e : l.me_mark) if (e) hipEventDestroy(e);
        if (l.queue_head) hipFree(l.queue_head);
        if (l.params_dev) hipFree(l.params_dev);
        for (int k = 0; k < SVT_HIP_PARAM_RING; k++) {
            if (l.params_host[k]) hipHostFree(l.params_host[k]);
            if (l.params_copied[k]) hipEventDestroy(l.params_copied[k]);
        }
        if (l.stream) hipStreamDestroy(l.stream);
    }
    if (ctx->io_stream) { hipStreamSynchronize(ctx->io_stream); hipStreamDestroy(ctx->io_stream); }
    if (ctx->io_fence) hipEventDestroy(ctx->io_fence);
    svt_hip_rd_tables_free(ctx);
    delete ctx;
}

const char *svt_hip_last_error(const SvtHipContext *) { return svt_hip_err_buf(); }
void       *svt_hip_context_stream(SvtHipContext *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int svt_hip_context_sync(SvtHipContext *ctx) {
    if (!ctx) return SVT_HIP_ERR_BAD_PARAM;
    SVT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->io_stream) SVT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->io_stream));
    return SVT_HIP_OK;
}

int svt_hip_pa_picture_create(SvtHipContext *ctx, const SvtHipPlaneDesc *full, const SvtHipPlaneDesc *quarter,
                              const SvtHipPlaneDesc *sixteenth, SvtHipPaPicture **pic) {
    return create_common(ctx, full, quarter, sixteenth, false, pic);
}

int svt_hip_pa_picture_create_dev(SvtHipContext *ctx, const SvtHipPlaneDesc *full_dev, SvtHipPaPicture **pic) {
    return create_common(ctx, full_dev, nullptr, nullptr, true, pic);
}

// A picture buffer taken from a pool and filled with the next input picture (the reference recycles its EbPaReferenceObject buffers the
// same way, Codec/reference_object.c): the full plane is copied in again (from page-locked host memory the copy is asynchronous), the
// 1/4 and 1/16 planes are rebuilt on the device.  Enqueued on the context stream.
static int picture_update_on(SvtHipContext *ctx, SvtHipPaPicture *pic, const SvtHipPlaneDesc *full, int full_on_device, hipStream_t stream) {
    int rc = check_plane(ctx, full, 64, "full");
    if (rc) return rc;
    DevPlane &f = pic->pyr.lvl[2];
    if (full->width != f.width || full->height != f.height || full->org_x != f.org_x || full->org_y != f.org_y)
        return svt_hip_fail(ctx, SVT_HIP_ERR_BAD_PARAM, "update: plane %ux%u pad %u,%u does not match the picture's %dx%d pad %d,%d", full->width, full->height,
                            full->org_x, full->org_y, f.width, f.height, f.org_x, f.org_y);
    if ((rc = upload_plane(ctx, full, &f, full_on_device != 0, stream))) return rc;
    for (int level = 1; level >= 0; level--) {
        const DevPlane &src = pic->pyr.lvl[level + 1], &dst = pic->pyr.lvl[level];
        const int pw = dst.width + 2 * dst.org_x, ph = dst.height + 2 * dst.org_y;
        hipLaunchKernelGGL(downsample2x_pad_kernel, dim3((pw + 255) / 256, ph), dim3(256), 0, stream, src, dst);
    }
    SVT_HIP_CHECK(ctx, hipGetLastError());
    SVT_HIP_CHECK(ctx, hipEventRecord(pic->ready, stream));
    pic->ready_stream = stream;
    return SVT_HIP_OK;
}

int svt_hip_pa_picture_update(SvtHipContext *ctx, SvtHipPaPicture *pic, const SvtHipPlaneDesc *full, int full_on_device) {
    if (!ctx || !pic) return SVT_HIP_ERR_BAD_PARAM;
    hipSetDevice(ctx->device);
    return picture_update_on(ctx, pic, full, full_on_device, ctx->stream);
}

// The refill on the context's transfer stream: ordered behind everything enqueued on the context stream so far (whatever still reads the
// picture's old content), beside whatever is enqueued afterwards; readers wait through the picture's `ready` event.
int svt_hip_pa_picture_update_ahead(SvtHipContext *ctx, SvtHipPaPicture *pic, const SvtHipPlaneDesc *full, int full_on_device) {
    if (!ctx || !pic) return SVT_HIP_ERR_BAD_PARAM;
    hipSetDevice(ctx->device);
    std::lock_guard<std::mutex> lock(ctx->async_mu);
    if (!ctx->io_stream) {
        SVT_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->io_stream, hipStreamNonBlocking));
        SVT_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->io_fence, hipEventDisableTiming));
    }
    SVT_HIP_CHECK(ctx, hipEventRecord(ctx->io_fence, ctx->stream));
    SVT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->io_stream, ctx->io_fence, 0));
    return picture_update_on(ctx, pic, full, full_on_device, ctx->io_stream);
}

void *svt_hip_context_transfer_stream(SvtHipContext *ctx) {
    if (!ctx) return nullptr;
    std::lock_guard<std::mutex> lock(ctx->async_mu);
    if (!ctx->io_stream) {
        hipSetDevice(ctx->device);
        if (hipStreamCreateWithFlags(&ctx->io_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&ctx->io_fence, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    return (void *)ctx->io_stream;
}

void svt_hip_pa_picture_destroy(SvtHipContext *ctx, SvtHipPaPicture *pic) {
    if (!pic) return;
    if (ctx) {
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
    }
    for (int l = 0; l < 3; l++)
        if (pic->mem[l]) hipFree(pic->mem[l]);
    if (pic->ready) hipEventDestroy(pic->ready);
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
    if (int rc = svt_hip_wait_picture(ctx, ctx->stream, pic)) return rc; // filled on another stream (a refill ahead): wait for it
    SVT_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dst_stride, p.base, p.stride, w, rows, hipMemcpyDeviceToHost, ctx->stream));
    SVT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SVT_HIP_OK;
}

} // extern "C"

char *svt_hip_err_buf(void) {
    static thread_local char buf[SVT_HIP_ERR_BYTES] = "";
    return buf;
}

int svt_hip_lane_setup(SvtHipContext *ctx, SvtHipLane *l, bool make_stream) {
    if (l->ready) return SVT_HIP_OK;
    // a set-up that failed part-way is retried by the lane's next holder: only the objects that do not exist yet are made
    if (make_stream && !l->stream) SVT_HIP_CHECK(ctx, hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking));
    if (!l->queue_head) {
        if (hipMalloc(reinterpret_cast<void **>(&l->queue_head), SVT_HIP_ME_QUEUE_BLOCK_BYTES) != hipSuccess) return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "lane set-up: hipMalloc failed");
        if (hipMemset(l->queue_head, 0, SVT_HIP_ME_QUEUE_BLOCK_BYTES) != hipSuccess) { hipFree(l->queue_head); l->queue_head = nullptr; return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "lane set-up: hipMemset failed"); }
    }
    if (!l->params_dev && hipMalloc(reinterpret_cast<void **>(&l->params_dev), SVT_HIP_ME_PARAM_BYTES) != hipSuccess)
        return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "lane set-up: hipMalloc failed");
    for (int k = 0; k < SVT_HIP_PARAM_RING; k++) {
        if (!l->params_host[k] && hipHostMalloc(reinterpret_cast<void **>(&l->params_host[k]), SVT_HIP_ME_PARAM_BYTES, hipHostMallocDefault) != hipSuccess)
            return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "lane set-up: hipHostMalloc failed");
        if (!l->params_copied[k]) SVT_HIP_CHECK(ctx, hipEventCreateWithFlags(&l->params_copied[k], hipEventDisableTiming));
    }
    l->ring_next = 0;
    l->ready     = true;
    return SVT_HIP_OK;
}

SvtHipLaneGuard::SvtHipLaneGuard(SvtHipContext *ctx) : ctx_(ctx), lane_(nullptr), index_(-1) {
    std::unique_lock<std::mutex> lk(ctx->pool_mu);
    ctx->pool_cv.wait(lk, [&] { return ctx->lane_busy != (1u << SVT_HIP_LANES) - 1u; });
    for (int i = 1; i < SVT_HIP_LANES; i++)
        if (!(ctx->lane_busy & (1u << i))) { index_ = i; break; }
    ctx->lane_busy |= 1u << index_;
    lk.unlock();
    hipSetDevice(ctx->device);
    // only the holder touches a borrowed lane, so its one-time set-up needs no lock
    if (svt_hip_lane_setup(ctx, &ctx->lane[index_], true) == SVT_HIP_OK) lane_ = &ctx->lane[index_];
}

SvtHipLaneGuard::~SvtHipLaneGuard() {
    // A holder that leaves early (an error return between two enqueues) may leave work queued on the lane's stream: the next holder must not
    // reuse -- or grow, i.e. free -- the lane's buffers under it.  After a normal call the stream is already idle and this returns at once.
    if (lane_ && lane_->stream) hipStreamSynchronize(lane_->stream);
    {
        std::lock_guard<std::mutex> lk(ctx_->pool_mu);
        ctx_->lane_busy &= ~(1u << index_);
    }
    ctx_->pool_cv.notify_one();
}

int svt_hip_wait_picture(SvtHipContext *ctx, hipStream_t stream, const SvtHipPaPicture *pic) {
    if (pic && pic->ready && pic->ready_stream != stream) SVT_HIP_CHECK(ctx, hipStreamWaitEvent(stream, pic->ready, 0));
    return SVT_HIP_OK;
}

int svt_hip_scratch(SvtHipContext *ctx, SvtHipLane *l, size_t bytes, void **out) {
    if (bytes > l->scratch_bytes) {
        if (l->scratch) {
            SVT_HIP_CHECK(ctx, hipStreamSynchronize(l->stream));
            hipFree(l->scratch);
            l->scratch = nullptr;
            l->scratch_bytes = 0;
        }
        if (hipMalloc(&l->scratch, bytes) != hipSuccess) return svt_hip_fail(ctx, SVT_HIP_ERR_NO_MEMORY, "hipMalloc(%zu) failed", bytes);
        l->scratch_bytes = bytes;
    }
    *out = l->scratch;
    return SVT_HIP_OK;
}

// Diagnostic: per-phase shader-clock sums of the ME kernel (non-zero only in a -DSVT_HIP_ME_PROFILE build), summed over the
// lanes; clears them.
extern "C" int svt_hip_me_profile_read(SvtHipContext *ctx, unsigned long long out[48]) {
    if (!ctx || !out) return SVT_HIP_ERR_BAD_PARAM;
    for (int k = 0; k < 48; k++) out[k] = 0;
    for (int i = 0; i < SVT_HIP_LANES; i++) {
        SvtHipLane &l = ctx->lane[i];
        if (!l.ready) continue;
        unsigned long long v[48];
        SVT_HIP_CHECK(ctx, hipStreamSynchronize(l.stream));
        SVT_HIP_CHECK(ctx, hipMemcpy(v, reinterpret_cast<char *>(l.queue_head) + 64, sizeof(v), hipMemcpyDeviceToHost));
        SVT_HIP_CHECK(ctx, hipMemset(reinterpret_cast<char *>(l.queue_head) + 64, 0, sizeof(v)));
        for (int k = 0; k < 48; k++) out[k] += v[k];
    }
    return SVT_HIP_OK;
}
