"""CPU: svt_hip_me_config_from_preset (host code of the product) against the reference's svt_aom_sig_deriv_me
outputs stored in tests/golden/me_presets.npz (11,424 preset / resolution / class / layer / qp combinations)."""
import ctypes as C
import os

import numpy as np

from svt_av1_psyex_amd import abi, api

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "me_presets.npz")


def test_presets_match_reference_table():
    z = np.load(GOLDEN)
    L = api.lib()
    bad = []
    for key, want in zip(z["keys"], z["cfg"]):
        em, res, sc, rtc, tl, hl, qp, fr = (int(v) for v in key)
        pd = abi.MePresetDesc(enc_mode=em, input_resolution=res, sc_class1=sc, rtc_tune=rtc, temporal_layer_index=tl,
                              hierarchical_levels=hl, qp=qp, frame_rate_q16=fr, safe_limit_nref=tl & 1, safe_limit_zz_th=5000)
        cfg = abi.MeConfig()
        assert L.svt_hip_me_config_from_preset(C.byref(pd), C.byref(cfg)) == 0
        if bytes(cfg) != want.tobytes():
            bad.append(tuple(key))
    assert not bad, f"{len(bad)} presets differ, e.g. {bad[:3]}"
    assert len(z["keys"]) == 11424


def test_presets_match_live_reference(ref):
    import pyoracle
    L = api.lib()
    for em in (-3, 0, 6, 8, 12, 13):
        for res in (0, 4, 5):
            pd = abi.MePresetDesc(enc_mode=em, input_resolution=res, qp=35, frame_rate_q16=60 << 16, temporal_layer_index=2, hierarchical_levels=5)
            a = pyoracle.config_from_preset_ref(pd)
            b = abi.MeConfig()
            assert L.svt_hip_me_config_from_preset(C.byref(pd), C.byref(b)) == 0
            assert bytes(a) == bytes(b), (em, res)
