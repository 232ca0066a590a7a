"""Shared helpers of the block-statistics tests: the committed fixture (reference `_c` outputs) and its comparison."""
import os

import numpy as np

from svt_av1_psyex_amd import abi

PSY_RD = 1.35  # the strength the fixture's psy_dist column was generated with (oracle/gen_golden.py)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "block_stats.npz")


def load_fixture(bd):
    z = np.load(GOLDEN)
    jobs = np.ascontiguousarray(z[f"jobs{bd}"]).view(abi.BLOCK_JOB_DTYPE).reshape(-1)
    exp = {name: z[f"{name}{bd}"] for name, _ in abi.STATS_OUT_FIELDS + abi.PSY_OUT_FIELDS}
    return z[f"src{bd}"], z[f"ref{bd}"], jobs, exp


def mismatches(exp, got, bd):
    bad = []
    for name, _ in abi.STATS_OUT_FIELDS + abi.PSY_OUT_FIELDS:
        if name == "satd" and bd != 8:
            continue
        if name not in got:
            bad.append(f"{name}: missing")
        elif not np.array_equal(exp[name], got[name]):
            i = int(np.flatnonzero(exp[name] != got[name])[0])
            bad.append(f"{name}: first mismatch at job {i}: {exp[name][i]} vs {got[name][i]}")
    return bad


def load_subpel_fixture():
    """8-bit planes + jobs with sub-pel phases; expected variance / var_sse from svt_aom_sub_pixel_variance{W}x{H}_c."""
    z = np.load(GOLDEN)
    jobs = np.ascontiguousarray(z["sp_jobs"]).view(abi.BLOCK_JOB_DTYPE).reshape(-1)
    return z["src8"], z["ref8"], jobs, {"variance": z["sp_variance"], "var_sse": z["sp_var_sse"]}


# svt_spatial_full_distortion_kernel_facade columns of the fixture: name -> (temporal_layer_index, spy_rd, psy_rd); same table as oracle/gen_golden.py
FACADE_SETTINGS = {"facade_a": (3, 1, PSY_RD), "facade_b": (5, 1, 0.0), "facade_c": (0, 1, 0.0), "facade_d": (4, 2, 0.0)}


def load_facade_fixture(bd):
    """(pred_mode, compound_type, {column: expected facade_dist}) of the jobs of load_fixture(bd)."""
    z = np.load(GOLDEN)
    return z[f"fac_mode{bd}"], z[f"fac_comp{bd}"], {k: z[f"{k}{bd}"] for k in FACADE_SETTINGS}


def facade_arg(modes, comps, setting):
    tli, spy, _ = FACADE_SETTINGS[setting]
    return dict(pred_mode=modes, compound_type=comps, temporal_layer_index=tli, spy_rd=spy)


def load_var10_fixture():
    """10-bit jobs of load_fixture(10): svt_aom_highbd_10_variance{W}x{H}_c results and the mask of jobs with an AV1 variance shape."""
    z = np.load(GOLDEN)
    return z["hbd10_variance"], z["hbd10_var_sse"], z["hbd10_valid"].astype(bool)
