// svt_hip_internal.h -- private host-side declarations of libsvthip.so
#ifndef SVT_HIP_INTERNAL_H
#define SVT_HIP_INTERNAL_H
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdio.h>
#include "me_kernel.h"

struct SvtHipContext {
    int         device;
    int         num_cus;
    hipStream_t stream;
    uint32_t   *queue_head; // SVT_HIP_ME_QUEUES counters in HBM
    void       *me_params;  // device copy of the ME launch's MeBatchHeader (SVT_HIP_ME_HEADER_BYTES) + MeKernelParams[SVT_HIP_ME_MAX_PICTURES]
    // scratch result buffers of the synchronous (host-pointer) entry points, grown on demand
    void  *scratch;
    size_t scratch_bytes;
    char   err[512];
};

struct SvtHipPaPicture {
    DevPyramid pyr;
    void      *mem[3];
    size_t     bytes[3];
};

static inline int svt_hip_fail(SvtHipContext *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define SVT_HIP_CHECK(ctx, call)                                                                                  \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return svt_hip_fail(ctx, SVT_HIP_ERR_LAUNCH, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                                              \
    } while (0)

// me_kernel.hip
size_t svt_hip_me_kernel_lds_bytes(void);
int    svt_hip_me_launch(SvtHipContext *ctx, const MeKernelParams *params, const uint32_t *n_jobs, uint32_t n_pictures);
// pictures.hip
int    svt_hip_scratch(SvtHipContext *ctx, size_t bytes, void **out);
#endif
