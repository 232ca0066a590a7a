"""Host-side helpers for the batched RD entry (svt_hip_rd_batch): job / quantizer-row construction and a
torch-backed runner.  torch is plumbing here (device buffers on the context's stream); the compute is in libsvthip.so."""
import ctypes as C

import numpy as np

from . import abi, api


def quant_row_from_step(dc_step, ac_step):
    """A plausible SvtHipQuantRow for dequantizer steps (dc, ac), built the way libaom's av1_build_quantizer fills
    its rows (invert_quant, zbin = 84/128 step, round = 48/128 step, fp round = 64/128 step).  Synthetic stand-in for
    the encoder's per-qindex tables, which the real caller passes in (Codec/full_loop.c:1627-1685)."""
    row = np.zeros((), dtype=abi.QUANT_ROW_DTYPE)
    for i, d in enumerate((int(dc_step), int(ac_step))):
        # invert_quant(): t = 1 + (1 << 16) * ((1 << l) - d) / d, shift = 1 << (16 - l), l = floor(log2(d))
        l = d.bit_length() - 1
        m = 1 + (1 << (16 + l)) // d
        row["quant"][i] = np.array(m - (1 << 16), np.int64).astype(np.int16)  # (int16_t) cast, wraps like the C code
        row["quant_shift"][i] = np.array(1 << (16 - l), np.int64).astype(np.int16)
        row["zbin"][i] = (84 * d + 64) >> 7
        row["round"][i] = (48 * d) >> 7
        row["quant_fp"][i] = min((1 << 16) // d, 32767)
        row["round_fp"][i] = (64 * d) >> 7
        row["dequant"][i] = d
    return row


def grid_jobs(width, height, stride, tx_size, tx_type=0, quant_row=0, org=0):
    """One job per tx block tiling a width x height picture (offsets in samples, row-major)."""
    w, h = abi.TX_W[tx_size], abi.TX_H[tx_size]
    ys, xs = np.meshgrid(np.arange(0, height - h + 1, h), np.arange(0, width - w + 1, w), indexing="ij")
    jobs = np.zeros(ys.size, dtype=abi.JOB_DTYPE)
    jobs["src_offset"] = (org + ys.ravel() * stride + xs.ravel()).astype(np.uint32)
    jobs["pred_offset"] = jobs["src_offset"]
    jobs["tx_type"] = tx_type
    jobs["quant_row"] = quant_row
    return jobs


def run_hip(ctx, desc_fields, src, pred, jobs, quant_rows, want_coeffs=True, want_recon=True, qmatrix=None, iqmatrix=None):
    """Runs svt_hip_rd_batch on device copies of the inputs; returns numpy results."""
    import torch
    ts = desc_fields["tx_size"]
    npk = min(abi.TX_W[ts], 32) * min(abi.TX_H[ts], 32)
    n = len(jobs)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
    t_src, t_pred, t_jobs, t_q = dev(src), dev(pred), dev(jobs), dev(quant_rows)
    outs = {name: torch.zeros(n * k * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda") for name, dt, k in abi.RD_OUT_FIELDS}
    if want_coeffs:
        for name in ("coeff", "qcoeff", "dqcoeff"):
            outs[name] = torch.zeros(n * npk * 4, dtype=torch.uint8, device="cuda")
    t_rec = t_pred.clone() if want_recon else None
    d = abi.RdBatchDesc(n_jobs=n, src=t_src.data_ptr(), pred=t_pred.data_ptr(), recon=t_rec.data_ptr() if want_recon else None,
                        jobs=t_jobs.data_ptr(), quant_rows=t_q.data_ptr(), n_quant_rows=len(quant_rows), **desc_fields)
    for name, t in outs.items():
        setattr(d, name, t.data_ptr())
    if qmatrix is not None:
        t_qm, t_iqm = dev(np.asarray(qmatrix, np.uint8)), dev(np.asarray(iqmatrix, np.uint8))
        d.qmatrix, d.iqmatrix = t_qm.data_ptr(), t_iqm.data_ptr()
    torch.cuda.synchronize()
    ctx.check(api.lib().svt_hip_rd_batch(ctx._h, C.byref(d)), "svt_hip_rd_batch")
    ctx.sync()
    res = {}
    for name, dt, k in abi.RD_OUT_FIELDS:
        res[name] = outs[name].cpu().numpy().view(dt).reshape(n, k)
    if want_coeffs:
        for name in ("coeff", "qcoeff", "dqcoeff"):
            res[name] = outs[name].cpu().numpy().view(np.int32).reshape(n, npk)
    if want_recon:
        res["recon"] = t_rec.cpu().numpy().view(pred.dtype).reshape(pred.shape)
    return res
