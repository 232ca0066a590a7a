#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/pmc_ic
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU -d gpurun_out/pmc_ic -o v --output-format csv -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/pmc_ic.log 2>&1
tail -3 gpurun_out/pmc_ic.log
python3 - <<PY
import csv,glob,collections,re
f=glob.glob("gpurun_out/pmc_ic/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    if "rd_tx_kernel" in k or "me_b64" in k:
        k=re.sub(r".*(rd_tx_kernel<[^>]*>|svt_hip_me_b64_kernel).*",r"\1",k)
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    d={c: v/len(n[k]) for c,v in acc[k].items()}
    print(k, {c: "%.3e"%v for c,v in d.items()})
PY
