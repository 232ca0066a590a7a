"""GPU: the RCCL exchange behind the C-ABI (include/svt_hip_comm.h).  A one-GPU box can only form a communicator of one rank: the
test drives the real entry points end to end at world size 1 (librccl.so opened at run time, communicator, the exchange stream ordered behind
the context stream, both gather forms, slot events); the band layout across ranks is covered on the CPU over gloo (test_shard_gloo.py) and
by bench.py's rehearsal mode."""
import ctypes as C

import numpy as np
import pytest

from svt_av1_psyex_amd import api

pytestmark = pytest.mark.gpu


def test_all_gather_world_one(hip_ctx):
    import torch
    L = api.lib()
    ident = (C.c_uint8 * 128)()
    hip_ctx.check(L.svt_hip_comm_unique_id(hip_ctx._h, ident), "svt_hip_comm_unique_id")
    comm = C.c_void_p()
    hip_ctx.check(L.svt_hip_comm_create(hip_ctx._h, ident, 0, 1, C.byref(comm)), "svt_hip_comm_create")
    try:
        assert L.svt_hip_comm_rank(comm) == 0 and L.svt_hip_comm_world(comm) == 1
        n_ranks = C.c_int(-1)
        hip_ctx.check(L.svt_hip_comm_count(comm, C.byref(n_ranks)), "svt_hip_comm_count")  # what RCCL itself says (ncclCommCount)
        assert n_ranks.value == 1
        ext = torch.cuda.ExternalStream(hip_ctx.stream)
        n = 1 << 20
        for slot, form in ((0, "plain"), (1, "v"), (0, "plain")):
            with torch.cuda.stream(ext):
                send = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")  # produced on the context stream: the exchange must order behind it
                recv = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
            if form == "plain":
                rc = L.svt_hip_me_results_all_gather(comm, slot, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), C.c_size_t(n))
            else:
                off, cnt = (C.c_size_t * 1)(32), (C.c_size_t * 1)(n - 100)
                rc = L.svt_hip_me_results_all_gather_v(comm, slot, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), off, cnt)
            hip_ctx.check(rc, "all_gather")
            hip_ctx.check(L.svt_hip_comm_stream_wait(comm, slot), "svt_hip_comm_stream_wait")  # context stream waits for the exchange
            with torch.cuda.stream(ext):
                got = recv.clone()
            hip_ctx.sync()
            hip_ctx.check(L.svt_hip_comm_sync(comm), "svt_hip_comm_sync")
            a, b = send.cpu().numpy(), got.cpu().numpy()
            if form == "plain":
                assert np.array_equal(a, b[:n]) and not b[n:].any()
            else:
                assert np.array_equal(a[:n - 100], b[32:32 + n - 100]) and not b[:32].any() and not b[32 + n - 100:].any()
        assert L.svt_hip_me_results_all_gather(comm, 5, None, None, C.c_size_t(0)) == 2  # bad slot
        # arguments are validated before any stream is touched: a null buffer leaves the slot's events alone
        assert L.svt_hip_me_results_all_gather_v(comm, 0, None, None, None, None) == 2
        hip_ctx.check(L.svt_hip_comm_stream_wait(comm, 0), "svt_hip_comm_stream_wait")
        hip_ctx.sync()
    finally:
        L.svt_hip_comm_destroy(comm)
