"""Turns the rocprofv3 outputs merged into gpurun_out/ by tools/final_verify.sh (<R>s = --stats, <R>f / <R>w = --pmc FETCH_SIZE / WRITE_SIZE,
<R>q = SQ counters, bench_final.log, me_phase_table.txt) into the committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
CMD = "python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras"


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    assert f, pattern
    return f[0]


def short(k):
    if 'svt_hip_me_' in k:
        return 'svt_hip_me_' + k.split('svt_hip_me_')[1].split('_kernel')[0] + '_kernel'
    if 'rd_tx_kernel' in k:
        return 'rd_tx_kernel<%s>' % k.split('rd_tx_kernel<')[1].split('>')[0]
    if 'fullpel' in k:
        return 'fullpel_pred_batch_kernel'
    return None


shutil.copy(one(f'gpurun_out/{R}s/**/*kernel_stats.csv'), f'profiles/{R}_kernel_stats_bench.csv')
b = json.loads(open('gpurun_out/bench_final.log').read().strip().split('\n')[-1])
out = {"command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) -- {CMD}",
       "unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB, averaged here over the launches of the run.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE "
                    "tallies the 128-byte requests of 16-byte-per-lane loads at 64 bytes, so it is doubled for the ME kernel (whose window staging is such loads); the RD kernels' figure "
                    "matches their known input bytes undoubled (calibrated in round 1: source + prediction planes read once).  Both counters sit on the memory side of L2 and include "
                    "Infinity-Cache hits.", "kernels": {}}
for d, cn in ((R + 'f', 'FETCH_SIZE'), (R + 'w', 'WRITE_SIZE')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(one(f'gpurun_out/{d}/**/*counter_collection.csv'))):
        acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        if short(k):
            out["kernels"].setdefault(short(k), {})[cn + "_KiB"] = round(sum(v) / len(v), 1)
            out["kernels"][short(k)]["launches"] = len(v)
for k, v in out["kernels"].items():
    f = 2 if k.startswith('svt_hip_me_') else 1
    v["hbm_bytes_per_launch"] = int((f * v.get("FETCH_SIZE_KiB", 0) + v.get("WRITE_SIZE_KiB", 0)) * 1024)
# the ME of a step is one launch of the C entry = a chain of kernels: their bytes summed (every kernel runs once per launch)
chain = [k for k in out["kernels"] if k.startswith('svt_hip_me_')]
out["kernels"]["me_launch"] = {"kernels": chain, "hbm_bytes_per_launch": sum(out["kernels"][k]["hbm_bytes_per_launch"] for k in chain)}
alg = {"me_launch": b["roofline"]["algorithmic_bytes_per_launch"], **{k: v["algorithmic_bytes_per_launch"] for k, v in b["rd_roofline"].items()}}
TX = {"4": "64x64", "3": "32x32", "2": "16x16"}  # TxSize enumerators of the instantiations bench.py launches
for k, v in out["kernels"].items():
    a = k
    if k.startswith('rd_tx_kernel<'):
        ts = k.split('<')[1].split(',')[0].replace('(TxSize)', '').strip()
        a = f"rd_tx_kernel<{TX.get(ts, ts)}, 10>"
        v["bench_name"] = a
    if a in alg:
        v["algorithmic_bytes_per_launch"] = alg[a]
        v["traffic_over_algorithmic"] = round(v["hbm_bytes_per_launch"] / alg[a], 2)
json.dump(out, open(f'profiles/{R}_hbm_traffic.json', 'w'), indent=1)
b['roofline']['traffic'] = out['kernels']['me_launch']['hbm_bytes_per_launch']  # the PMC passes of this very verification run
json.dump(b, open(f'profiles/{R}_bench_final.json', 'w'), indent=1)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(one(f'gpurun_out/{R}q/**/*counter_collection.csv'))):
    if short(r['Kernel_Name']) and 'fullpel' not in r['Kernel_Name']:
        acc[(short(r['Kernel_Name']), r['Counter_Name'])].append(float(r['Counter_Value']))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
kern = sorted({kk[0] for kk in avg})
busy = {k: 4 * avg[(k, 'SQ_ACTIVE_INST_VALU')] / 1024 / (avg[(k, 'SQ_BUSY_CYCLES')] / 32) for k in kern if avg[(k, 'SQ_BUSY_CYCLES')] > 0}
wait = {k: avg[(k, 'SQ_WAIT_ANY')] / avg[(k, 'SQ_WAVE_CYCLES')] for k in kern}
with open(f'profiles/{R}_sq_counters.txt', 'w') as f:
    f.write(f"rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD -- {CMD}\n")
    f.write("average per launch (one launch = the 16 pictures of a step = 32,640 blocks of 64x64 for the ME kernel); *_CYCLES / ACTIVE / WAIT counters are in quad-cycles per wave (x4 = clocks); SQ_BUSY_CYCLES summed over 32 shader engines\n")
    f.write("derived: VALU busy per SIMD = 4*SQ_ACTIVE_INST_VALU/1024 / (SQ_BUSY_CYCLES/32): " + ", ".join(f"{k} {100 * v:.0f} %" for k, v in sorted(busy.items())) + "\n")
    f.write("derived: share of wave-cycles waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES): " + ", ".join(f"{k} {100 * v:.0f} %" for k, v in sorted(wait.items())) + "\n")
    nb = 32640.0
    mek = [k for k in kern if k.startswith('svt_hip_me_')]
    f.write("derived: ME launch (its kernels summed) wave-instructions per 64x64 block: " + ", ".join(f"{c[9:]} {sum(avg[(k, c)] for k in mek) / nb:.0f}" for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD')) + "\n")
    f.write("\n")
    for k in sorted(avg):
        f.write(f"{k[0]:28s} {k[1]:22s} {avg[k]:.4g}\n")
# MFMA evidence: the block-statistics launches of bench.py's `other_kernels` leg (hadamard_path's 16x16 / 32x32 tiles on the matrix cores)
try:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(one(f'gpurun_out/{R}m/**/*counter_collection.csv'))):
        if 'block_stats_kernel' in r['Kernel_Name']:
            acc[('block_stats_kernel<' + r['Kernel_Name'].split('block_stats_kernel<')[1].split('>')[0] + '>', r['Counter_Name'])].append(float(r['Counter_Value']))
    with open(f'profiles/{R}_mfma_counters.txt', 'w') as f:
        f.write("rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline\n")
        f.write("block_stats_kernel launches (average per launch); the 8-bit instantiation with `satd` requested runs hadamard_path's 16x16 / 32x32 tiles as v_mfma_f32_16x16x16_f16 pairs\n\n")
        for k in sorted(acc):
            f.write(f"{k[0]:62s} {k[1]:28s} {sum(acc[k]) / len(acc[k]):.4g}  ({len(acc[k])} launches)\n")
        if os.path.exists('gpurun_out/hadamard_mfma.txt'):
            f.write("\ntools/ubench/hadamard_mfma (the 8x8 core + SATD alone, int16 residual blocks, VALU butterflies vs two f16 MFMAs per four blocks):\n")
            f.write(open('gpurun_out/hadamard_mfma.txt').read())
except AssertionError:
    pass
if os.path.exists('gpurun_out/me_phase_table.txt'):
    shutil.copy('gpurun_out/me_phase_table.txt', f'profiles/{R}_me_phase_table.txt')
print(open(f'profiles/{R}_kernel_stats_bench.csv').read()[:900])
print({k: (v["hbm_bytes_per_launch"], v.get("traffic_over_algorithmic")) for k, v in out["kernels"].items()})
print(b["value"], b["kernel_ms"], b["roofline"], b.get("cpu_baseline"))
print({k: round(v, 2) for k, v in busy.items()})
