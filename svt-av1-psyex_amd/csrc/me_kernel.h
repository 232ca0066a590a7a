// me_kernel.h -- launch-time parameter block of svt_hip_me_b64_kernel (host and device view).
#ifndef SVT_HIP_ME_KERNEL_H
#define SVT_HIP_ME_KERNEL_H
#include <stdint.h>
#include "../../include/svt_hip_me.h"

// Workgroup shape of the ME kernel (tunable at build time): threads per 64x64 block, resident workgroups per CU the
// grid and the launch bounds are sized for, and the LDS window arena each workgroup owns.
#ifndef SVT_HIP_ME_THREADS
#define SVT_HIP_ME_THREADS 256
#endif
#ifndef SVT_HIP_ME_WG_PER_CU
#define SVT_HIP_ME_WG_PER_CU 4
#endif
#ifndef SVT_HIP_ME_WIN_BYTES
#define SVT_HIP_ME_WIN_BYTES 16384
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

struct MeKernelParams {
    SvtHipMeConfig      cfg;
    SvtHipMePictureDesc desc;
    DevPyramid          cur;
    DevPyramid          ref[SVT_HIP_MAX_LISTS][SVT_HIP_MAX_REFS];
    SvtHipMeResults     res;   // device pointers
    uint32_t            w64, row0, n_pu;
    uint32_t            queue_begin[SVT_HIP_ME_QUEUES + 1]; // job index ranges (jobs are band-local b64 raster indices)
    uint32_t           *queue_head;                         // SVT_HIP_ME_QUEUES counters, zeroed before launch
};

#endif
