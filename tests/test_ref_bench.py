"""bench.py's natively threaded CPU baseline (oracle/ref_harness.c:ref_bench_rows) does the work it claims: in its run-once mode the
checksum (sum of the 64x64 ME distortions of every b64 + sum of the RD chain's eobs over the sample's rows) equals the one computed
from ref_me_picture + ref_rd_batch call by call, for one thread and for several."""
import numpy as np
import pytest

import pyoracle
from svt_av1_psyex_amd import abi, api, rd, synth

W, H = 448, 256


@pytest.mark.parametrize("simd", [False, True])
def test_ref_bench_rows_checksum(ref, simd):
    y10 = synth.synth_sequence(W, H, 5, 3)
    y8 = synth.to_8bit(y10)
    pyr = {i: synth.HostPyramid(y8[i], i) for i in range(5)}
    qr = np.stack([rd.quant_row_from_step(140, 176)])
    pics, want = [], 0
    sizes = (4, 3, 2)
    for d in (1, 2):
        cfg = api.config_from_preset(6, W, H, qp=35, temporal_layer_index=3, hierarchical_levels=4)
        desc = api.picture_desc(W, H, 2, {(0, 0): 2 - d, (1, 0): 2 + d}, enc_mode=6, temporal_layer_index=3, hierarchical_levels=4)
        refs = {(0, 0): pyr[2 - d], (1, 0): pyr[2 + d]}
        pics.append((cfg, desc, pyr[2], refs, y10[2], y10[2 - d]))
        ref.ref_set_simd(1 if simd else 0)
        me = pyoracle.me_picture("ref", cfg, desc, pyr[2], refs, search_level=False)
        ref.ref_set_simd(0)
        w64 = (W + 63) // 64
        want += int(me["me_64x64_distortion"].reshape(-1)[w64:3 * w64].astype(np.int64).sum())  # rows 1 and 2
        for ts in sizes:
            jobs = rd.grid_jobs(W, H, W, ts)
            ys = jobs["src_offset"] // W
            jobs = np.ascontiguousarray(jobs[(ys >= 64) & (ys < 192)])
            out = pyoracle.rd_batch(dict(bit_depth=10, quant_kind=0, tx_size=ts, src_stride=W, pred_stride=W), y10[2], y10[2 - d], jobs, qr, want_coeffs=False,
                                    want_recon=False, impl="ref_simd" if simd else "ref")
            want += int(out["eob"].astype(np.int64).sum())
    for nt in (1, 3):
        rate, items, elapsed, checksum = pyoracle.ref_bench_rows(pics, W, H, 1, 2, qr, sizes, nt, 0.0, simd=simd)
        assert items == 4 and checksum == want, (nt, items, checksum, want)
    rate, items, elapsed, _ = pyoracle.ref_bench_rows(pics, W, H, 1, 2, qr, sizes, 2, 0.2, simd=simd)
    assert items > 4 and 0.2 <= elapsed < 5 and rate > 0
