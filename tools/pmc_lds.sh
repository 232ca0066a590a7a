#!/bin/bash
# LDS counters of the ME kernel on bench.py's launch: bank conflicts, unaligned stalls, LDS-array activity against the kernel's busy cycles
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/pmc_lds
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d gpurun_out/pmc_lds -o v --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_lds.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(float); n=set()
for r in csv.DictReader(open(f[0])):
    if "me_b64" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print("ME kernel, per launch:", {c: round(v/len(n)) for c,v in sorted(acc.items())})
PY
