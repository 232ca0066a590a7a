"""Turns the rocprofv3 outputs merged into gpurun_out/ (r01s = --stats, r01f / r01w = --pmc FETCH_SIZE / WRITE_SIZE, pmc7 = SQ
counters, bench_final.log) into the committed summaries under profiles/."""
import collections
import csv
import json
import shutil

shutil.copy('gpurun_out/r01s/bench_kernel_stats.csv', 'profiles/r01_kernel_stats_bench_steps5.csv')
out = json.load(open('profiles/r01_hbm_traffic.json'))
out["kernels"] = {}


def short(k):
    if 'me_b64' in k:
        return 'svt_hip_me_b64_kernel'
    if 'rd_tx_kernel' in k:
        return 'rd_tx_kernel<%s>' % k.split('rd_tx_kernel<')[1].split('>')[0]
    if 'fullpel' in k:
        return 'fullpel_pred_batch_kernel'
    return None


for d, cn in (('r01f', 'FETCH_SIZE'), ('r01w', 'WRITE_SIZE')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f'gpurun_out/{d}/bench_counter_collection.csv')):
        acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        if short(k):
            out["kernels"].setdefault(short(k), {})[cn + "_KiB"] = round(sum(v) / len(v), 1)
            out["kernels"][short(k)]["launches"] = len(v)
for k, v in out["kernels"].items():
    f = 2 if k == 'svt_hip_me_b64_kernel' else 1
    v["hbm_bytes_per_launch"] = int((f * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024)
json.dump(out, open('profiles/r01_hbm_traffic.json', 'w'), indent=1)
b = json.loads(open('gpurun_out/bench_final.log').read().strip().split('\n')[-1])
json.dump(b, open('profiles/r01_bench_final.json', 'w'), indent=1)
acc = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmc7/a_counter_collection.csv')):
    if short(r['Kernel_Name']) and 'fullpel' not in r['Kernel_Name']:
        acc[(short(r['Kernel_Name']), r['Counter_Name'])].append(float(r['Counter_Value']))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
busy = {k: 4 * avg[(k, 'SQ_ACTIVE_INST_VALU')] / 1024 / (avg[(k, 'SQ_BUSY_CYCLES')] / 32) for k in {kk[0] for kk in avg}}
with open('profiles/r01_sq_counters.txt', 'w') as f:
    f.write("rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline\n")
    f.write("average per launch (one launch = the 16 pictures of a step); *_CYCLES / ACTIVE / WAIT counters are in quad-cycles per wave (x4 = clocks); SQ_BUSY_CYCLES summed over 32 shader engines\n")
    f.write("derived: VALU busy per SIMD = 4*SQ_ACTIVE_INST_VALU/1024 / (SQ_BUSY_CYCLES/32): " + ", ".join(f"{k} {100 * v:.0f} %" for k, v in sorted(busy.items())) + "\n\n")
    for k in sorted(avg):
        f.write(f"{k[0]:28s} {k[1]:22s} {avg[k]:.4g}\n")
print(open('profiles/r01_kernel_stats_bench_steps5.csv').read()[:720])
print({k: v["hbm_bytes_per_launch"] for k, v in out["kernels"].items()})
print(b["value"], b["kernel_ms"], b["roofline"], b["cpu_baseline"])
print({k: round(v, 2) for k, v in busy.items()})
