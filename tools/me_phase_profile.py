"""Phase tables of the kernels of the ME chain from the diagnostic builds (tools/build_me_profile_lib.sh: -DSVT_HIP_ME_PROFILE, where lane 0 of
each wave accumulates s_memtime deltas per phase of a block; one library per kernel of the chain) on bench.py's own ME launch; prints the share
of a block's time per phase and writes gpurun_out/me_phase_table.txt.   usage (GPU box): python tools/me_phase_profile.py [mode ...]
(modes as in me_kernel.hip: 0 one-kernel form -- run with SVT_HIP_ME_STAGED=0 --, 1 mid1, 5 s2, 7 tail; default 7 1 5).  One process per
mode (the library is chosen at import): the script re-runs itself as a child per mode before anything touches the GPU."""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "tests"), os.path.join(HERE, "..")]
import numpy as np

from svt_av1_psyex_amd import api

KERNEL = {0: "svt_hip_me_b64_kernel (one-kernel form)", 1: "svt_hip_me_mid1_kernel", 2: "svt_hip_me_s1_kernel", 4: "svt_hip_me_mid2_kernel", 5: "svt_hip_me_s2_kernel", 7: "svt_hip_me_tail_kernel"}
if len(sys.argv) != 3 or sys.argv[1] != "--one":
    import subprocess
    modes = [int(m) for m in sys.argv[1:]] or [7, 1, 5]
    os.makedirs("gpurun_out", exist_ok=True)
    texts = []
    for m in modes:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(m)], capture_output=True, text=True, env={**os.environ, **({"SVT_HIP_ME_STAGED": "0"} if m == 0 else {})})
        if r.returncode:
            print(r.stderr[-2000:])
            sys.exit(r.returncode)
        texts.append(r.stdout)
    open("gpurun_out/me_phase_table.txt", "w").write("\n\n".join(texts))
    print("\n\n".join(texts))
    sys.exit(0)
MODE = int(sys.argv[2])
api.LIB_PATH = os.path.join(HERE, f"_libprof{MODE}.so")

MODE = int(sys.argv[2]) if len(sys.argv) == 3 and sys.argv[1] == "--one" else -1
# index -> phase (PROF(i) in csrc/me_kernel.hip)
STAGES = ("zero-MV SADs", "pre-HME", "HME level 0", "HME level 1", "HME level 2", "check-00")
PHASES = [
    (0, "job fetch (queue atomic)" if MODE == 0 else "end of the previous job (results out) + job fetch" if MODE != 1 else "everything behind the set-up (zero-MV / pre-HME / level-0 results folded, level-1 searches pushed, state out) + job fetch"), (1, "block set-up: source 64x64 / 32x32 / 16x16 views into LDS"), (2, "stage preamble"),
    (6, "control before zero-MV SADs"), (7, "control before pre-HME"), (8, "control before HME level 0"), (9, "control before HME level 1"),
    (10, "control before HME level 2"), (11, "control before check-00"), (12, "control before the 8x8-variance probe"), (20, "control before the integer search"),
] + [(24 + i, f"{n}: plan a round + issue its window loads" if MODE not in (2, 5) else ("search kernel: job flags read", "search kernel: requests + source view into LDS", "direct search: qualification, lane set-up", "direct search: row loop")[i] if i < 4 else "-") for i, n in enumerate(STAGES)] + [(32 + i, f"{n}: window registers -> LDS arena") for i, n in enumerate(STAGES)] + [
    (40 + i, f"{n}: rest of the evaluation (tile entry -> registers, arg-min across the wave, result)") for i, n in enumerate(STAGES)] + [
    (22, "all stages: plan the next round + issue its window loads (inside the evaluation phase)"), (30, "all stages: wide tiles, item loop (8 positions x whole block per lane)"),
    (23, "all stages: small searches, item loop (8 positions x row slice per lane, LDS atomics)"), (19, "searches: tail"), (3, "8x8-variance probe: folding the bests, merge into best_sad / best_mv"), (4, "integer search: folding the bests, merge into best_sad / best_mv"),
    (46, "integer search / probe: window staged (global -> registers -> LDS arena)"), (47, "integer search / probe: positions evaluated"),
    (5, "control after a stage (fold results, centres, early exits)"),
    (13, "reference pruning"), (14, "candidate lists"), (15, "distortions / variance outputs"), (16, "result rows stored"),
]
NSLOT = 48

import torch  # noqa: E402

import bench  # noqa: E402  (the repo's bench.py: its Workload is the launch that is profiled)

ctx = api.Context()
ext = torch.cuda.ExternalStream(ctx.stream)
wl = bench.Workload(ctx, 0, 1)
n_b64 = wl.w64 * wl.h64 * len(bench.PICS)
out = (C.c_ulonglong * NSLOT)()


def launches(n):
    with torch.cuda.stream(ext):
        for i in range(n):
            ctx.me_pictures_async(wl.me_jobs[i % bench.N_SETS][i % 2])
    ctx.sync()


launches(2)
api.lib().svt_hip_me_profile_read(ctx._h, out)  # reads and clears
N = 6
launches(N)
api.lib().svt_hip_me_profile_read(ctx._h, out)
col = np.array(out[:NSLOT], dtype=np.float64) / N / n_b64
tot = col.sum()
lines = [f"{KERNEL[MODE]}, diagnostic build -DSVT_HIP_ME_PROFILE -DSVT_HIP_ME_PROFILE_MODE={MODE} (tools/build_me_profile_lib.sh): where a block's wall time in this kernel goes.  One wave per block; lane 0 accumulates",
         f"s_memtime deltas per phase; averaged over the {n_b64} blocks of bench.py's launch (16 pictures 3840x2160, preset 6, R = 2, reference distances 8 / 1 / 4 / 2) and {N} launches",
         "on the three picture sets.  The waves of a CU share its issue slots, so a phase's share of wall time includes the cycles its wave waited for the others to issue.", "",
         f"{'phase':86s} {'clocks/block':>12s} {'%':>5s}"]
for i, name in PHASES:
    if col[i] >= 0.5:
        lines.append(f"{name:86s} {col[i]:12.0f} {100 * col[i] / tot:5.1f}")
lines.append(f"{'total clocks per block (s_memtime)':86s} {tot:12.0f}")
grp = {"control (serial, one lane)": (2, 6, 7, 8, 9, 10, 11, 12, 20, 5, 13), "set-up, outputs, fetch": (0, 1, 14, 15, 16), "probe + integer search": (3, 4, 46, 47)}
if MODE in (2, 5):
    grp = {"per-job prologue / epilogue (flags, requests, source view, lane set-up, arg-min, keys out; slots 0, 24, 25, 26, 23)": (0, 24, 25, 26, 23), "row loop (27)": (27,)}
elif MODE == 1:
    grp = {}  # (the kernel's straight-line stage flow has no marks behind the set-up: its time lands in the next job's fetch slot)
else:
    grp.update({n + " (without the item loops)": (24 + i, 32 + i, 40 + i) for i, n in enumerate(STAGES)})
    grp["item loops (22, 30, 23)"] = (22, 30, 23)
lines.append("")
for g, idx in grp.items():
    v = sum(col[i] for i in idx)
    if v >= 0.5:
        lines.append(f"{g:86s} {v:12.0f} {100 * v / tot:5.1f}")
print("\n".join(lines))
