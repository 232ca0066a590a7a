"""The TPL dispenser's kernel chain (Codec/src_ops_process.c:857-872 + get_quantize_error :225-249) expressed as svt_hip_rd_batch
jobs: 8-bit planes, the dispenser's quantizer call (plain svt_av1_quantize_fp -- log-scale 0 -- for EVERY size: quant_kind 2), DCT_DCT,
the dispenser's partial-frequency shape, and row sub-sampling (subsample_tx) as a doubled / quadrupled plane stride with the half / quarter
height transform (:380-382,531).  Dispenser level 0 works on 16x16 blocks, level 1 on 32x32 (:377-382); level 2 (64x64) is never
selected by the reference (initial_rc_process.c:308-368) and would not fit its MAX_TPL_SIZE = 32 buffers.  Shared by the CPU and GPU tests."""
import numpy as np

from svt_av1_psyex_amd import abi, rd

# [level][subsample_tx] -> TX_16X16, TX_16X8, TX_16X4 / TX_32X32, TX_32X16, TX_32X8
TPL_TX_SIZE = {0: {0: 2, 1: 8, 2: 14}, 1: {0: 3, 1: 10, 2: 16}}
W, H = 96, 64


def planes(seed, amp):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, 256, (H, W), dtype=np.uint8)
    pred = np.clip(src.astype(np.int32) + rng.integers(-amp, amp + 1, src.shape), 0, 255).astype(np.uint8)
    return src, pred


def batch(level, sub, pf_shape, n_rows=3):
    """desc fields + jobs: every (16 << level)-square block of the W x H plane, quantizer rows cycling over the jobs."""
    size = 16 << level
    fields = dict(bit_depth=8, quant_kind=2, tx_size=TPL_TX_SIZE[level][sub], src_stride=W << sub, pred_stride=W << sub)
    ys, xs = np.meshgrid(np.arange(0, H, size), np.arange(0, W, size), indexing="ij")
    jobs = np.zeros(ys.size, dtype=abi.JOB_DTYPE)
    jobs["src_offset"] = jobs["pred_offset"] = (ys.ravel() * W + xs.ravel()).astype(np.uint32)
    jobs["tx_type"] = 0  # DCT_DCT (svt_av1_wht_fwd_txfm, transforms.c:3640-3655)
    jobs["pf_shape"] = pf_shape
    jobs["quant_row"] = np.arange(ys.size) % n_rows
    rows = np.stack([rd.quant_row_from_step(*s) for s in ((8, 9), (40, 52), (220, 305))][:n_rows])
    return fields, jobs, rows


def tpl_outputs(out, level, sub):
    """What the dispenser derives from the chain: inter_cost (:871), eob, recon_error and sse (get_quantize_error :244-248: the error
    sums are shifted by 2 unless tx_size == TX_32X32)."""
    d = out["dist_coeff"].astype(np.int64).reshape(-1, 2)
    shift = 0 if (level == 1 and sub == 0) else 2
    return {"inter_cost": out["satd"].astype(np.int64).ravel() << sub, "eob": out["eob"].astype(np.int64).ravel(),
            "recon_error": np.maximum(d[:, 0] >> shift, 1), "sse": np.maximum(d[:, 1] >> shift, 1)}


def seed_of(level, sub, pf):
    return 100 + level * 1000 + sub * 10 + pf


GRID = [(level, sub, pf, amp) for level in (0, 1) for sub in (0, 1, 2) for pf in (0, 1, 2) for amp in (3, 60)]
