#!/bin/bash
# SQ counters of the RD kernels (per launch, per transform size) for one bench.py run
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/pmc_rd
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/pmc_rd -o v --output-format csv -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/pmc_rd.log 2>&1
python3 - <<PY
import csv,glob,collections,re
f=glob.glob("gpurun_out/pmc_rd/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    if "rd_tx_kernel" in k or "me_b64" in k:
        k=re.sub(r".*(rd_tx_kernel<[^>]*>|svt_hip_me_b64_kernel).*",r"\1",k)
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
px=16*3840*2160
for k in sorted(acc):
    d={c: v/len(n[k]) for c,v in acc[k].items()}
    busy=4*d["SQ_ACTIVE_INST_VALU"]/1024/(d["SQ_BUSY_CYCLES"]/32)
    print(k, "VALU lane-instr/px %.1f"%(d["SQ_INSTS_VALU"]*64/px), "VALU busy %.0f%%"%(100*busy), "wait_any %.0f%%"%(100*d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"]), "wait_inst %.0f%%"%(100*d["SQ_WAIT_INST_ANY"]/d["SQ_WAVE_CYCLES"]), "LDS instr/px %.2f"%(d["SQ_INSTS_LDS"]*64/px), "SALU/VALU %.2f"%(d["SQ_INSTS_SALU"]/d["SQ_INSTS_VALU"]))
PY
