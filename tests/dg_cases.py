"""Dynamic-GOP detector HME cases shared by the CPU (oracle vs reference / golden) and GPU parity tests.

A case is a (source, reference) pair of luma pictures: the detector compares the picture at the end of a
mini-GOP with an earlier one (pd_process.c:590-640), so the reference is the same synthetic pan a few frames
back; the adversarial kinds follow the reference's SAD tests (test/SadTest.cc:161-207: flat, extreme, random)."""
import numpy as np

from svt_av1_psyex_amd import api, synth


class DgCase:
    def __init__(self, width, height, kind="pan", distance=4, seed=3):
        self.width, self.height, self.kind = width, height, kind
        rng = np.random.default_rng(seed)
        if kind == "pan":  # SURVEY 8(d) sequence
            seq = synth.to_8bit(synth.synth_sequence(width, height, distance + 1, seed))
            src, ref = seq[distance], seq[0]
        elif kind == "fast":  # motion beyond the small search areas, mixed directions
            seq = synth.to_8bit(synth.synth_sequence(width, height, 2, seed, pan=(45, 27)))
            src, ref = seq[0], seq[1]
        elif kind == "random":  # no structure: best matches land anywhere in the window
            src = rng.integers(0, 256, (height, width), dtype=np.uint8)
            ref = rng.integers(0, 256, (height, width), dtype=np.uint8)
        elif kind == "flat":  # every position ties: the first one in raster order must win
            src = np.full((height, width), 90, np.uint8)
            ref = np.full((height, width), 131, np.uint8)
        elif kind == "extreme":  # maximum SAD everywhere
            src = np.zeros((height, width), np.uint8)
            ref = np.full((height, width), 255, np.uint8)
        elif kind == "zoom":  # content moving away from the centre: sum_in_vectors takes one sign
            seq = synth.to_8bit(synth.synth_sequence(width + 64, height + 64, 1, seed))[0]
            ref = seq[32:32 + height, 32:32 + width]
            ys, xs = np.mgrid[0:height, 0:width]
            yy = np.clip(32 + ys - (ys - height // 2) // 12, 0, height + 63)
            xx = np.clip(32 + xs - (xs - width // 2) // 12, 0, width + 63)
            src = np.ascontiguousarray(seq[yy, xx])
        else:
            raise ValueError(kind)
        self.src, self.ref = synth.HostPyramid(np.ascontiguousarray(src), distance), synth.HostPyramid(np.ascontiguousarray(ref), 0)
        self.aligned_width, self.aligned_height = (width + 7) & ~7, (height + 7) & ~7
        self.input_resolution = input_resolution(width, height)

    def args(self):
        return self.aligned_width, self.aligned_height, self.input_resolution

    def __repr__(self):
        return f"DgCase({self.width}x{self.height},{self.kind})"


def input_resolution(width, height):
    """svt_aom_derive_input_resolution (Codec/utility.c): thresholds on the luma sample count, restated on the host side of
    the library (svt_hip_input_resolution, no GPU involved)."""
    return int(api.lib().svt_hip_input_resolution(width, height))


# (width, height, kind): the three search-area sides (16 / 64 / 128), partial b64 columns and rows, every content kind
GRID = [(352, 288, "pan"), (352, 288, "random"), (640, 480, "pan"), (640, 480, "fast"), (856, 480, "random"), (1280, 720, "pan"),
        (1280, 720, "zoom"), (1000, 568, "random"), (1920, 1080, "fast"), (712, 400, "flat"), (712, 400, "extreme"), (200, 136, "random")]

METRICS = ("tot_dist", "tot_cplx", "tot_active", "sum_in_vectors")
