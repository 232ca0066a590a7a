"""Re-runs one seed of tools/me_fuzz_campaign.py and prints where HIP and the oracle differ (search-level arrays included)."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", d) for d in ("tests", "oracle", "", "tools")]
import numpy as np
from me_cases import MeCase
from svt_av1_psyex_amd import api
import me_fuzz_campaign as F

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
w, h = int(rng.choice([176, 352, 360, 640, 712, 856])), int(rng.choice([144, 200, 288, 360, 488]))
if os.environ.get("FUZZ_LARGE"):
    w, h = [(1280, 720), (1920, 1080), (1000, 568), (1928, 1088), (2560, 1440)][int(rng.integers(0, 5))]
n0, n1 = int(rng.integers(1, 5)), int(rng.integers(0, 4))
cur = 4
pick = rng.permutation([f for f in range(9) if f != cur])
refs = {(0, i): int(pick[i]) for i in range(n0)}
refs.update({(1, i): int(pick[n0 + i]) for i in range(n1)})
tl = int(rng.integers(0, 5)) if n1 == 0 else int(rng.integers(1, 5))
kw = dict(enc_mode=int(rng.integers(-1, 14)), cur=cur, refs=refs, n_frames=9, seed=seed, kind=str(rng.choice(["pan", "noise", "fastpan", "flat", "extremes"])),
          temporal_layer_index=tl, cfg_edit=F.fuzz(rng), gm_enabled=int(rng.integers(0, 2)), is_ref=int(rng.integers(0, 2)))
case = MeCase(w, h, **kw)
cfg = case.cfg
for f, _ in cfg._fields_:
    v = getattr(cfg, f)
    try:
        print(f, int(v), end="; ")
    except Exception:
        pass
print()
print("me_sa", cfg.me_sa.sa_min.width, cfg.me_sa.sa_min.height, cfg.me_sa.sa_max.width, cfg.me_sa.sa_max.height, "l0", cfg.hme_l0_sa.sa_min.width, cfg.hme_l0_sa.sa_min.height,
      cfg.hme_l0_sa.sa_max.width, cfg.hme_l0_sa.sa_max.height, "l1", cfg.hme_l1_sa.width, cfg.hme_l1_sa.height, "l2", cfg.hme_l2_sa.width, cfg.hme_l2_sa.height)
print("size", w, h, "refs", refs, "tl", tl, kw["kind"], "M", kw["enc_mode"])
ctx = api.Context(0)
a, b = case.run_cpu("oracle"), case.run_hip(ctx)
for k in a:
    if not np.array_equal(a[k], b[k]):
        idx = np.argwhere(np.asarray(a[k]).reshape(b[k].shape) != b[k])
        print(k, len(idx), "first", [tuple(int(x) for x in i) for i in idx[:6]], "oracle", [int(np.asarray(a[k]).reshape(b[k].shape)[tuple(i)]) for i in idx[:6]], "hip", [int(b[k][tuple(i)]) for i in idx[:6]])
ctx.close()
