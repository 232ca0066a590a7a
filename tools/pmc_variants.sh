#!/bin/bash
# SQ instruction counters of the ME kernel per build variant (diagnostic / ablation builds): one bench.py launch set each
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in "$@"; do
  if [ "$v" = "default" ]; then lib=$PWD/svt-av1-psyex_amd/libsvthip.so; else lib=$PWD/svt-av1-psyex_amd/variants/lib_$v.so; fi
  rm -rf gpurun_out/pmc_v
  SVT_HIP_LIBRARY=$lib timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD -d gpurun_out/pmc_v -o v --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_v.log 2>&1
  python3 - "$v" <<PY
import csv,glob,collections,sys
f=glob.glob("gpurun_out/pmc_v/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(float); n=set()
for r in csv.DictReader(open(f[0])):
    if "me_b64" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
nb=32640.0*len(n)
print(sys.argv[1], "per b64:", {c: round(v/nb,1) for c,v in sorted(acc.items())})
PY
done
