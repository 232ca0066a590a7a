"""Synthetic luma sequences and host-side luma pyramids (numpy only; no GPU code here).

The sequence generator follows SURVEY.md §8(d): a box-filtered noise field, panned by (5,3) px per frame
plus Gaussian noise, in 10 bit; the 8-bit ME planes are (y10 + 2) >> 2.  The pyramid mirrors the geometry
the reference allocates for EbPaReferenceObject (Globals/enc_handle.c:1244-1279): full plane with
`pad` px of padding (scs->left_padding = BLOCK_SIZE_64 + 4, enc_handle.c:4110), quarter with 32, sixteenth
with 16, padding by edge replication (Codec/pic_operators.c:397-443).
"""
import numpy as np

from . import abi


def box_filter(a, k=9):
    """k x k box mean via an integral image (float64)."""
    pad = k // 2
    ap = np.pad(a, pad, mode="edge").astype(np.float64)
    ii = np.zeros((ap.shape[0] + 1, ap.shape[1] + 1))
    ii[1:, 1:] = ap.cumsum(0).cumsum(1)
    h, w = a.shape
    return (ii[k:k + h, k:k + w] - ii[:h, k:k + w] - ii[k:k + h, :w] + ii[:h, :w]) / (k * k)


def synth_sequence(width, height, n_frames, seed, pan=(5, 3), noise_sigma=6.0):
    """Returns uint16 array [n_frames, height, width] of 10-bit luma."""
    rng = np.random.default_rng(seed)
    bw, bh = width + pan[0] * n_frames + 16, height + pan[1] * n_frames + 16
    base = box_filter(rng.random((bh, bw)), 9)
    lo, hi = base.min(), base.max()
    base = 60.0 + (base - lo) * (900.0 / max(hi - lo, 1e-9))
    out = np.empty((n_frames, height, width), np.uint16)
    for i in range(n_frames):
        x0, y0 = pan[0] * i, pan[1] * i
        f = base[y0:y0 + height, x0:x0 + width] + rng.normal(0.0, noise_sigma, (height, width))
        out[i] = np.clip(np.rint(f), 0, 1023).astype(np.uint16)
    return out


def to_8bit(y10):
    return ((y10.astype(np.uint32) + 2) >> 2).clip(0, 255).astype(np.uint8)


def pad_plane(img, pad_x, pad_y, stride_align=1):
    """Edge-replicated padded copy; returns (buffer [H+2*pad_y, stride], stride)."""
    h, w = img.shape
    stride = w + 2 * pad_x
    if stride_align > 1:
        stride = (stride + stride_align - 1) // stride_align * stride_align
    buf = np.zeros((h + 2 * pad_y, stride), img.dtype)
    buf[:, :w + 2 * pad_x] = np.pad(img, ((pad_y, pad_y), (pad_x, pad_x)), mode="edge")
    return buf, stride


def downsample_2x(img):
    """svt_aom_downsample_2d_c with decim_step 2 (Codec/pic_analysis_process.c:130-158)."""
    h, w = img.shape
    a = img[: h // 2 * 2, : w // 2 * 2].astype(np.uint32)
    return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)


class HostPyramid:
    """Full / quarter / sixteenth padded 8-bit luma planes on the host (numpy) + their PlaneDesc."""

    def __init__(self, luma8, picture_number=0, pad=68):
        assert luma8.dtype == np.uint8 and luma8.ndim == 2
        self.picture_number = int(picture_number)
        q = downsample_2x(luma8)
        s = downsample_2x(q)
        self.height, self.width = luma8.shape
        self.planes = []  # index 0 = sixteenth, 1 = quarter, 2 = full
        for img, p in ((s, 16), (q, 32), (luma8, pad)):
            buf, stride = pad_plane(img, p, p)
            self.planes.append((np.ascontiguousarray(buf), stride, p, img.shape[1], img.shape[0]))

    def desc(self, level):
        buf, stride, p, w, h = self.planes[level]
        return abi.PlaneDesc(buf.ctypes.data, stride, p, p, w, h)

    def descs(self):
        return (abi.PlaneDesc * 3)(*[self.desc(l) for l in range(3)])

    def inner(self, level):
        buf, stride, p, w, h = self.planes[level]
        return buf[p:p + h, p:p + w]
