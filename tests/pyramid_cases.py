"""Pictures of the luma-pyramid parity tests (test infrastructure) and the call into the reference build that makes the
fixture tests/golden/pyramid.npz (oracle/gen_golden.py pyramid)."""
import ctypes as C

import numpy as np

from svt_av1_psyex_amd import abi, synth

# name: (width, height, content, seed) -- widths / heights are multiples of 8 like pcs->aligned_width / aligned_height; 72x80 is
# smaller than one 64x64 block row + column of padding, 360x296 has partial b64 columns / rows
CASES = {
    "p72x80": (72, 80, "noise", 1),
    "p360x296": (360, 296, "noise", 2),
    "cif": (352, 288, "pan", 1234),
    "p1080": (1920, 1080, "ramp", 0),
}
PAD_FULL, PAD_Q, PAD_S = 68, 32, 16


def luma(w, h, kind, seed):
    if kind == "noise":
        return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "pan":
        return synth.to_8bit(synth.synth_sequence(w, h, 1, seed))[0]
    y, x = np.mgrid[0:h, 0:w]  # "ramp": compressible, every 2x2 sum phase occurs, extremes on the edges
    img = ((x // 3) * 5 + (y // 2) * 3 + ((x ^ y) & 1)) & 255
    img[0, :], img[-1, :], img[:, 0], img[:, -1] = 255, 0, 0, 255
    return img.astype(np.uint8)


def padded(img, pad):
    buf, stride = synth.pad_plane(img, pad, pad)
    return np.ascontiguousarray(buf), stride


def ref_pyramid(img, level1=1):
    """(quarter, sixteenth) padded planes from the reference's svt_aom_downsample_filtering_input_picture (needs oracle/_ref)."""
    import pyoracle
    ref = pyoracle.load_ref()
    h, w = img.shape
    full, fs = padded(img, PAD_FULL)
    q = np.full((h // 2 + 2 * PAD_Q, w // 2 + 2 * PAD_Q), 0xA5, np.uint8)   # poisoned: every byte must be written
    s = np.full((h // 4 + 2 * PAD_S, w // 4 + 2 * PAD_S), 0xA5, np.uint8)
    fd = abi.PlaneDesc(full.ctypes.data, fs, PAD_FULL, PAD_FULL, w, h)
    qd = abi.PlaneDesc(q.ctypes.data, q.shape[1], PAD_Q, PAD_Q, w // 2, h // 2)
    sd = abi.PlaneDesc(s.ctypes.data, s.shape[1], PAD_S, PAD_S, w // 4, h // 4)
    assert ref.ref_pyramid(C.byref(fd), C.byref(qd), C.byref(sd), level1) == 0
    return q, s
