// svt_hip_internal.h -- private host-side declarations of libsvthip.so
#ifndef SVT_HIP_INTERNAL_H
#define SVT_HIP_INTERNAL_H
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdio.h>
#include <condition_variable>
#include <mutex>
#include "me_kernel.h"

// Threading model (the reference calls the kernels from several ME / mode-decision threads with several pictures in
// flight: Globals/enc_handle.c:2265,2293, Codec/me_process.c:140-172):
//   * a LANE owns everything one host call in flight needs on the device: a stream, the ME job-queue counters, the ME
//     parameter block (a ring of pinned host copies + one device copy) and a grow-on-demand result buffer;
//   * lane 0 belongs to the asynchronous entries (they enqueue on svt_hip_context_stream(), whose order the caller
//     owns); its enqueues are serialised by `async_mu`;
//   * every synchronous (host-pointer) entry borrows one of the other lanes for the duration of the call, so calls
//     from different host threads neither share result memory nor wait for each other's kernels.
#define SVT_HIP_LANES 9       /* lane 0 + 8 borrowed lanes (more callers than that wait for a free lane) */
#define SVT_HIP_PARAM_RING 4  /* ME launches that may be enqueued ahead on one lane before the host waits */

struct SvtHipLane {
    hipStream_t stream;
    uint32_t   *queue_head;  // SVT_HIP_ME_QUEUE_BLOCK_BYTES: job-queue counters of the ME kernels, the profiling build's phase sums, list cursors (me_kernel.h)
    uint8_t    *params_dev;  // MeBatchHeader (SVT_HIP_ME_HEADER_BYTES) + MeKernelParams[SVT_HIP_ME_MAX_PICTURES]
    uint8_t    *params_host[SVT_HIP_PARAM_RING];   // pinned staging copies of the same block
    hipEvent_t  params_copied[SVT_HIP_PARAM_RING]; // recorded behind the H2D copy that reads params_host[i]
    int         ring_next;
    void       *scratch;     // result buffer of the synchronous entries, grown on demand (only touched by the lane's holder)
    size_t      scratch_bytes;
    void       *dense;       // slots of the ME dense pre-pass (me_dense.inl), grown on demand
    size_t      dense_bytes;
    void       *stage;       // staged ME launches: the blocks' travelling records, flag words and the three job lists
    size_t      stage_bytes;
    hipEvent_t  me_mark[SVT_HIP_ME_CHAIN_KERNELS + 1]; // svt_hip_context_set_me_timing: events around the kernels of the last ME launch (created on first use)
    uint32_t    me_marked;   // bit i: kernel i of the chain ran in the last launch (me_mark[i] .. me_mark[i + 1] bracket it)
    bool        ready;       // device objects exist (lanes are set up on first use)
};

struct SvtHipContext {
    int         device;
    int         num_cus;
    hipStream_t stream; // == lane[0].stream
    SvtHipLane  lane[SVT_HIP_LANES];
    std::mutex  async_mu;              // serialises enqueues through lane 0
    std::mutex  pool_mu;               // guards lane_busy / lane set-up
    std::condition_variable pool_cv;
    uint32_t    lane_busy;             // bit i: lane i is borrowed
    int16_t    *iscan_dev;             // [19][3][1024] inverse scan orders (rd_kernel.hip), this device's copy
    bool        me_attr_set;           // hipFuncSetAttribute done for the ME kernel on this device
    int         me_staged;             // with a pre-pass, the per-block pipeline runs as a chain of small kernels: 0 never, 1 launches of many blocks, 2 always (svt_hip_context_set_me_staged)
    bool        me_counting;           // ME waves report what they took from the dense pre-pass (svt_hip_context_set_me_counting)
    bool        me_timing;             // ME launches record events around their kernels (svt_hip_context_set_me_timing)
    bool        me_dense;              // the dense pre-HME / level-0 pre-pass runs ahead of the per-block ME kernel (svt_hip_context_set_me_dense)
    uint32_t    me_waves_per_cu;       // 0: as many persistent ME waves per CU as fit; else an upper limit (svt_hip_context_set_me_waves_per_cu)
    hipStream_t io_stream;             // transfer stream of svt_hip_pa_picture_update_ahead (created on first use)
    hipEvent_t  io_fence;              // orders the transfer stream behind the context stream
};

struct SvtHipPaPicture {
    DevPyramid  pyr;
    void       *mem[3];
    size_t      bytes[3];
    hipEvent_t  ready;        // recorded behind the uploads / decimation kernels that fill the planes
    hipStream_t ready_stream; // the stream that work was enqueued on: other streams wait for `ready` before reading
};

// The last error message of the CALLING thread (svt_hip_last_error): contexts are shared between threads
char *svt_hip_err_buf(void);
#define SVT_HIP_ERR_BYTES 512

static inline int svt_hip_fail(SvtHipContext *, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(svt_hip_err_buf(), SVT_HIP_ERR_BYTES, fmt, ap);
    va_end(ap);
    return code;
}

#define SVT_HIP_CHECK(ctx, call)                                                                                  \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return svt_hip_fail(ctx, SVT_HIP_ERR_LAUNCH, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                                              \
    } while (0)

// Borrow / return a lane for a synchronous entry (RAII).  lane() is null when the lane's device objects could not be made.
class SvtHipLaneGuard {
public:
    explicit SvtHipLaneGuard(SvtHipContext *ctx);
    ~SvtHipLaneGuard();
    SvtHipLane *lane() const { return lane_; }
private:
    SvtHipContext *ctx_;
    SvtHipLane    *lane_;
    int            index_;
};
int svt_hip_lane_setup(SvtHipContext *ctx, SvtHipLane *l, bool make_stream);
// makes `stream` wait for the work that fills `pic` when that was enqueued on another stream
int svt_hip_wait_picture(SvtHipContext *ctx, hipStream_t stream, const SvtHipPaPicture *pic);

// me_kernel.hip
size_t svt_hip_me_kernel_lds_bytes(void);
int    svt_hip_me_launch(SvtHipContext *ctx, SvtHipLane *lane, const MeKernelParams *params, const uint32_t *n_jobs, uint32_t n_pictures);
// me_picture.hip: the asynchronous ME entry on an explicit lane
int    svt_hip_me_pictures_on_lane(SvtHipContext *ctx, SvtHipLane *lane, uint32_t n_pictures, const SvtHipMeJob *jobs);
// pictures.hip
int    svt_hip_scratch(SvtHipContext *ctx, SvtHipLane *lane, size_t bytes, void **out);
// rd_kernel.hip: cosine / inverse-scan tables of the context's device
int    svt_hip_rd_tables_init(SvtHipContext *ctx);
void   svt_hip_rd_tables_free(SvtHipContext *ctx);
#endif
