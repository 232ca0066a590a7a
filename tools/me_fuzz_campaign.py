"""One-off robustness campaign: random presets, reference sets, picture sizes, content kinds and search controls through the
HIP ME kernel against the oracle (a wider net than tests/test_me_gpu.py::test_fuzzed_search_controls_match_oracle).
usage: python tools/me_fuzz_campaign.py [first_seed] [count]          HIP vs oracle (GPU box)
       python tools/me_fuzz_campaign.py [first_seed] [count] ref      oracle vs the reference build oracle/_ref (CPU, build container)"""
import os
import sys

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", d) for d in ("tests", "oracle", "")]
import numpy as np

from me_cases import MCTF_OUTPUTS, MeCase, compare
from svt_av1_psyex_amd import api


def fuzz(rng):
    def edit(cfg):
        pick = lambda *v: int(rng.choice(v))
        cfg.me_sa.sa_min.width, cfg.me_sa.sa_min.height = pick(8, 16, 24, 40, 104), pick(3, 8, 16, 37, 104)
        cfg.me_sa.sa_max.width = max(cfg.me_sa.sa_min.width, pick(8, 32, 64, 136))
        cfg.me_sa.sa_max.height = max(cfg.me_sa.sa_min.height, pick(3, 16, 32, 120))
        cfg.hme_l0_sa.sa_min.width, cfg.hme_l0_sa.sa_min.height = pick(8, 16, 32, 64), pick(8, 16, 32, 64)
        cfg.hme_l0_sa.sa_max.width, cfg.hme_l0_sa.sa_max.height = pick(96, 192, 320), pick(96, 192, 320)
        cfg.hme_l1_sa.width, cfg.hme_l1_sa.height = pick(8, 16), pick(3, 5, 16)
        cfg.hme_l2_sa.width, cfg.hme_l2_sa.height = pick(8, 16), pick(3, 7, 16)
        cfg.me_early_exit_th = pick(0, 0, 64 * 64, 64 * 64 * 8)
        cfg.me_safe_limit_zz_th = pick(0, 0, 64 * 64 * 2)
        cfg.prev_me_stage_based_exit_th = pick(0, 0, 16 * 16 * 8, 64 * 64)
        cfg.me_8x8_var_enabled = pick(0, 1)
        cfg.hme_search_method, cfg.me_search_method = pick(0, 1), pick(0, 1)
        if cfg.prehme_sa_cfg[0].sa_max.width:
            cfg.prehme_enable, cfg.prehme_l1_early_exit, cfg.prehme_skip_search_line = pick(0, 1), pick(0, 1), pick(0, 1)
        cfg.enable_me_sr_adjustment = pick(0, 1, 2)
        cfg.distance_based_hme_resizing = pick(0, 1)
        if pick(0, 1):
            cfg.reduce_hme_l0_sr_th_min, cfg.reduce_hme_l0_sr_th_max = pick(2000, 8000), pick(20000, 60000)
        cfg.enable_hme_level1_flag = pick(0, 1, 1)
        cfg.enable_hme_level2_flag = pick(0, 1) if cfg.enable_hme_level1_flag else 0
        cfg.enable_me_hme_ref_pruning = pick(0, 1)
        cfg.prune_ref_if_hme_sad_dev_bigger_than_th, cfg.prune_ref_if_me_sad_dev_bigger_than_th = pick(5, 30, 0xFFFF), pick(10, 60, 0xFFFF)
        cfg.prune_me_candidates_th = pick(0, 30, 65)
        cfg.use_best_unipred_cand_only = pick(0, 1)
        cfg.mv_sa_adj_enabled = pick(0, 1)
        if cfg.mv_sa_adj_enabled:
            cfg.mv_sa_adj_nearest_ref_only, cfg.mv_sa_adj_mv_size_th, cfg.mv_sa_adj_sa_multiplier = pick(0, 1), pick(4, 16), pick(2, 3)
    return edit


def main():
    first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 5000), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
    vs_ref = len(sys.argv) > 3 and sys.argv[3] == "ref"
    ctx = None if vs_ref else api.Context(0)
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        w, h = int(rng.choice([176, 352, 360, 640, 712, 856])), int(rng.choice([144, 200, 288, 360, 488]))
        if os.environ.get("FUZZ_LARGE"):  # a second draw keeps the small-size seeds reproducible
            w, h = [(1280, 720), (1920, 1080), (1000, 568), (1928, 1088), (2560, 1440)][int(rng.integers(0, 5))]
        n0, n1 = int(rng.integers(1, 5)), int(rng.integers(0, 4))
        frames = list(range(9))
        cur = 4
        others = [f for f in frames if f != cur]
        pick = rng.permutation(others)
        refs = {(0, i): int(pick[i]) for i in range(n0)}
        refs.update({(1, i): int(pick[n0 + i]) for i in range(n1)})
        tl = int(rng.integers(0, 5)) if n1 == 0 else int(rng.integers(1, 5))
        kw = dict(enc_mode=int(rng.integers(-1, 14)), cur=cur, refs=refs, n_frames=9, seed=seed, kind=str(rng.choice(["pan", "noise", "fastpan", "flat", "extremes"])),
                  temporal_layer_index=tl, cfg_edit=fuzz(rng), gm_enabled=int(rng.integers(0, 2)), is_ref=int(rng.integers(0, 2)))
        mctf = rng.random() < 0.2
        try:
            case = MeCase(w, h, mctf_exit_th=int(rng.choice([0, 400, 64 * 64 * 4])) if mctf else None, **kw)
            want = case.run_cpu("ref" if vs_ref else "oracle")
            got = case.run_cpu("oracle") if vs_ref else case.run_hip(ctx)
            d = compare({k: want[k] for k in MCTF_OUTPUTS} if mctf else want, got)
        except Exception as e:  # noqa: BLE001
            d = [f"exception {type(e).__name__}: {e}"]
        if d:
            bad += 1
            print(f"seed {seed} {w}x{h} refs {refs} {kw['kind']} M{kw['enc_mode']} mctf {mctf}: {d[:3]}", flush=True)
        elif (seed - first) % 10 == 9:
            print(f"... {seed - first + 1} cases, {bad} bad", flush=True)
    print(f"done: {count} cases, {bad} bad")
    if ctx:
        ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
