"""Shared helpers for transform parity tests: size / type grids and reference-function dispatch."""
import ctypes as C

import numpy as np

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
DCT_DCT, IDTX, V_DCT, H_DCT = 0, 9, 10, 11


def valid_types(tx_size):
    """Transform types the AV1 syntax allows per size class (ext-tx sets): all 16 up to 16x16, DCT/identity
    combinations at 32, DCT_DCT only at 64."""
    m = max(TX_W[tx_size], TX_H[tx_size])
    if m <= 16:
        return list(range(16))
    if m == 32:
        return [DCT_DCT, IDTX, V_DCT, H_DCT]
    return [DCT_DCT]


def ref_fwd_name(tx_size):
    w, h = TX_W[tx_size], TX_H[tx_size]
    return f"svt_av1_transform_two_d_{w}x{h}_c" if w == h else f"svt_av1_fwd_txfm2d_{w}x{h}_c"


def ref_fwd(ref, tx_size, tx_type, resid, stride, bd):
    w, h = TX_W[tx_size], TX_H[tx_size]
    out = np.zeros(w * h, np.int32)
    getattr(ref, ref_fwd_name(tx_size))(resid.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint32(stride), C.c_int(tx_type), C.c_uint8(bd))
    return out


def ref_inv(ref, tx_size, tx_type, coeff, pred, stride, bd):
    """pred: uint16 [h, stride]; returns recon uint16 [h, stride] (columns >= w untouched)."""
    w, h = TX_W[tx_size], TX_H[tx_size]
    out = pred.copy()
    fn = getattr(ref, f"svt_av1_inv_txfm2d_add_{w}x{h}_c")
    a = [coeff.ctypes.data_as(C.c_void_p), pred.ctypes.data_as(C.c_void_p), C.c_int32(stride), out.ctypes.data_as(C.c_void_p), C.c_int32(stride), C.c_int(tx_type)]
    if w == h:
        a += [C.c_int32(bd)]
    elif (w, h) in ((4, 8), (8, 4), (4, 16), (16, 4)):
        a += [C.c_int(tx_size), C.c_int32(bd)]
    else:
        a += [C.c_int(tx_size), C.c_int32(w * h), C.c_int32(bd)]
    fn(*a)
    return out


def residual_block(rng, w, h, stride, bd, pattern):
    lim = (1 << bd) - 1
    if pattern == "max":
        r = np.full((h, stride), lim, np.int16)
    elif pattern == "min":
        r = np.full((h, stride), -lim, np.int16)
    elif pattern == "checker":
        r = np.where((np.add.outer(np.arange(h), np.arange(stride)) & 1) == 0, lim, -lim).astype(np.int16)
    elif pattern == "laplace":
        r = np.clip(np.rint(rng.laplace(0, 12, (h, stride))), -lim, lim).astype(np.int16)
    else:
        r = rng.integers(-lim, lim + 1, (h, stride)).astype(np.int16)
    return np.ascontiguousarray(r)
