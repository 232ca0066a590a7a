/*
 * svt_hip_comm.h -- C-ABI of the multi-GPU exchange of the open-loop ME results (RCCL over xGMI).
 *
 * Open-loop ME has no dependency between 64x64 blocks, so the b64 rows of a picture are sharded across the GPUs of a node in
 * contiguous bands (SvtHipMePictureDesc.b64_row_start / b64_row_count; reference pyramids are replicated); the per-b64 results of
 * every band -- the MeSbResults arrays (Codec/me_sb_results.h:44-51) and the per-b64 scalars -- are then gathered on every rank so that
 * the host side that consumes them (the reference posts a picture once all its b64 are done, Codec/me_process.c:174-313) finds the
 * whole picture.  One process per GPU; one communicator per process.
 *
 * Bootstrap like any NCCL / RCCL program: rank 0 makes an id (svt_hip_comm_unique_id), the host distributes those 128 bytes by its
 * own means, every rank calls svt_hip_comm_create.  librccl.so is opened at run time: libsvthip.so itself does not link it.
 */
#ifndef SVT_HIP_COMM_H
#define SVT_HIP_COMM_H

#include <stddef.h>
#include <stdint.h>
#include "svt_hip_me.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SVT_HIP_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
#define SVT_HIP_COMM_SLOTS 2      /* exchanges in flight (double-buffered result sets) */

typedef struct SvtHipComm SvtHipComm;

int  svt_hip_comm_unique_id(SvtHipContext *ctx, uint8_t id[SVT_HIP_COMM_ID_BYTES]);
int  svt_hip_comm_create(SvtHipContext *ctx, const uint8_t id[SVT_HIP_COMM_ID_BYTES], int rank, int world, SvtHipComm **comm);
void svt_hip_comm_destroy(SvtHipComm *comm);
int  svt_hip_comm_rank(const SvtHipComm *comm);
int  svt_hip_comm_world(const SvtHipComm *comm);
/* The rank count RCCL reports for the communicator (ncclCommCount): recorded next to multi-GPU measurements. */
int  svt_hip_comm_count(SvtHipComm *comm, int *ranks);

/* Enqueue the exchange of result set `slot` (0 / 1): `send_dev` = this rank's compact result buffer (device, bytes_per_rank bytes, written by
 * work already enqueued on the context stream, e.g. svt_hip_me_pictures_async), `recv_dev` = world * bytes_per_rank bytes, rank r's buffer
 * at r * bytes_per_rank.  The exchange runs on a stream of its own behind that work; whatever the caller enqueues on the context stream
 * afterwards overlaps it.  Returns without waiting. */
int svt_hip_me_results_all_gather(SvtHipComm *comm, int slot, const void *send_dev, void *recv_dev, size_t bytes_per_rank);
/* Same with a different byte count per rank (bands of different height): rank r's bytes[r] bytes land at recv_dev + offsets[r]; this rank
 * sends bytes[rank] bytes from send_dev.  offsets / bytes are host arrays of `world` entries, identical on every rank. */
int svt_hip_me_results_all_gather_v(SvtHipComm *comm, int slot, const void *send_dev, void *recv_dev, const size_t *offsets, const size_t *bytes);
/* Make the context stream wait for slot's last exchange (call it before enqueueing work that overwrites that slot's send buffer or
 * reads its receive buffer); no-op when the slot was never used. */
int svt_hip_comm_stream_wait(SvtHipComm *comm, int slot);
/* Block the host until every enqueued exchange has completed. */
int svt_hip_comm_sync(SvtHipComm *comm);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_COMM_H */
