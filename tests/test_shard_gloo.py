"""CPU, world_size 2 over gloo: the row-band sharding + all-gather of per-b64 ME results reproduces the single-rank
results.  The compute stand-in on CPU is the oracle; on GPUs bench.py runs the same layout code over RCCL."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from me_cases import MeCase, compare
from svt_av1_psyex_amd import abi, shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import pyoracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = MeCase(640, 360, enc_mode=6, seed=9)
        d = case.desc
        w64, h64 = 10, 6
        lay = shard.BandLayout(w64, h64, world, abi.n_pu(d.enable_me_16x16, d.enable_me_8x8), d.max_refs, d.max_cand, n_pictures=2)
        buf = np.zeros(lay.nbytes, np.uint8)
        for pic in range(2):  # two pictures in one exchange (bench.py: sixteen); the left-over rows rotate between them
            r0, r1 = lay.band(pic, rank)
            if r1 == r0:
                continue  # an empty band (more ranks than rows): b64_row_count 0 would mean "all rows"
            d.b64_row_start, d.b64_row_count = r0, r1 - r0
            res = lay.results_struct(buf.ctypes.data, pic, rank)
            rc = pyoracle.load_oracle().orc_me_picture(C.byref(case.cfg), C.byref(d), case.cur.descs(), pyoracle.ref_plane_array(case.refs), C.byref(res))
            assert rc == 0
        mine = torch.from_numpy(buf)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        if rank == 0:
            q.put(np.stack([g.numpy() for g in gathered]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_band_sharding_and_all_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case = MeCase(640, 360, enc_mode=6, seed=9)
    whole = case.run_cpu("oracle")
    d = case.desc
    lay = shard.BandLayout(10, 6, world, abi.n_pu(d.enable_me_16x16, d.enable_me_8x8), d.max_refs, d.max_cand, n_pictures=2)
    for pic in range(2):
        merged = lay.unpack(gathered, pic)
        assert not compare({k: whole[k] for k in merged}, merged)


def test_bands_cover_every_row_once():
    for h64 in (5, 17, 34):
        for world in (1, 2, 3, 4, 8):
            for rot in range(world):
                rows = [r for k in range(world) for r in range(*shard.band(h64, k, world, rot))]
                assert rows == list(range(h64))
                sizes = [shard.band(h64, k, world, rot)[1] - shard.band(h64, k, world, rot)[0] for k in range(world)]
                assert max(sizes) == shard.rows_max(h64, world) and max(sizes) - min(sizes) <= 1
            # over `world` consecutive pictures every rank owns the same number of rows
            tot = [sum(shard.band(h64, k, world, shard.rotation(p, h64, world))[1] - shard.band(h64, k, world, shard.rotation(p, h64, world))[0]
                       for p in range(world)) for k in range(world)]
            assert len(set(tot)) == 1 and tot[0] == h64


def test_layout_holds_live_rows_only_and_survives_empty_bands():
    """The per-rank buffers hold the live rows only (no padding to the tallest band); with more ranks than b64 rows some bands are
    empty (a caller then skips the launch: b64_row_count == 0 means "all rows" in the C-ABI)."""
    lay = shard.BandLayout(60, 34, 8, 85, 2, 3, n_pictures=16)
    live = 68 * 60 * lay.bytes_per_b64
    assert lay.uniform and lay.rank_rows == [68] * 8 and live <= lay.nbytes < live + 16 * 16 * len(lay.fields)  # at most one 16-byte pad per field band
    for r in range(8):  # every field band starts on a 16-byte boundary (32-bit fields are stored with dword stores)
        for p in range(16):
            assert all(off % 16 == 0 for off, _ in lay.field_offsets(p, r).values())
    odd = shard.BandLayout(30, 17, 4, 85, 2, 3, n_pictures=3)  # 1080p: 30 blocks per row, bands of 4 / 5 rows -> u8 bands that are no multiple of 4
    assert all(off % 16 == 0 for r in range(4) for p in range(3) for off, _ in odd.field_offsets(p, r).values())
    lay = shard.BandLayout(60, 34, 3, 85, 2, 3, n_pictures=16)
    assert sum(lay.rank_rows) == 16 * 34 and max(lay.rank_rows) - min(lay.rank_rows) <= 1
    small = shard.BandLayout(6, 5, 8, 85, 2, 3, n_pictures=2)  # 5 rows on 8 ranks
    empty = [(p, r) for p in range(2) for r in range(8) if small.band(p, r)[0] == small.band(p, r)[1]]
    assert len(empty) == 6 and sum(small.rank_rows) == 10
    # pack / unpack round trip through the layout with a fake "result" = the b64 index
    bufs = np.zeros((8, small.nbytes), np.uint8)
    for r in range(8):
        for p in range(2):
            r0, r1 = small.band(p, r)
            for n, (off, per) in small.field_offsets(p, r).items():
                blk = bufs[r, off:off + (r1 - r0) * 6 * per].reshape(-1, per)
                blk[:] = (np.arange(r0 * 6, r1 * 6) % 251 + p)[:, None]
    for p in range(2):
        full = small.unpack(bufs, p)
        for n, a in full.items():
            assert np.array_equal(a.view(np.uint8).reshape(30, -1)[:, 0], (np.arange(30) % 251 + p).astype(np.uint8)), n
