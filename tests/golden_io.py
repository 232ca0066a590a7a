"""Loads tests/golden/me_*.npz fixtures back into the shapes the implementations consume."""
import ctypes as C
import glob
import os

import numpy as np

from svt_av1_psyex_amd import abi, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def me_fixture_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "me_*.npz")) if "presets" not in p)


class GoldenMeCase:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.cfg = abi.MeConfig.from_buffer_copy(z["cfg"].tobytes())
        self.desc = abi.MePictureDesc.from_buffer_copy(z["desc"].tobytes())
        self.cur = synth.HostPyramid(z["cur"], self.desc.picture_number)
        self.refs = {}
        for k in z.files:
            if k.startswith("ref_"):
                _, li, ri = k.split("_")
                self.refs[(int(li), int(ri))] = synth.HostPyramid(z[k], self.desc.ref_picture_number[int(li)][int(ri)])
        self.expected = {k[4:]: z[k] for k in z.files if k.startswith("out_")}

    def run_cpu(self, which="oracle"):
        import pyoracle
        return pyoracle.me_picture(which, self.cfg, self.desc, self.cur, self.refs)

    def run_hip(self, ctx, device_pyramid=True):
        cur = ctx.upload(self.cur, device_pyramid)
        refs = {k: ctx.upload(v, device_pyramid) for k, v in self.refs.items()}
        try:
            return ctx.me_picture(self.cfg, self.desc, cur, refs)
        finally:
            cur.free()
            for r in refs.values():
                r.free()
