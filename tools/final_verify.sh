#!/bin/bash
# Round-end verification on the GPU box: full GPU suite, smoke, the bench line, and the rocprofv3 passes profiles/ is built from.
# Run through gpurun from the repo root; outputs land in gpurun_out/ (then: python3 tools/refresh_profiles.py).
R=${SVT_ROUND:-r03}
set -e -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/bench_final.log 2> gpurun_out/bench_final.err || { tail -20 gpurun_out/bench_final.err; exit 1; }
tail -1 gpurun_out/bench_final.log | cut -c1-400
rm -rf gpurun_out/${R}s gpurun_out/${R}f gpurun_out/${R}w gpurun_out/${R}q
BENCH="bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}s -o bench --output-format csv -- python3 $BENCH > gpurun_out/${R}s.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${R}f -o bench --output-format csv -- python3 $BENCH > gpurun_out/${R}f.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${R}w -o bench --output-format csv -- python3 $BENCH > gpurun_out/${R}w.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD -d gpurun_out/${R}q -o bench --output-format csv -- python3 $BENCH > gpurun_out/${R}q.log 2>&1
rm -rf gpurun_out/${R}m
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES -d gpurun_out/${R}m -o bench --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/${R}m.log 2>&1
if [ -x tools/ubench/hadamard_mfma ]; then ./tools/ubench/hadamard_mfma > gpurun_out/hadamard_mfma.txt 2>&1 || true; fi
if [ -f tools/_libprof7.so ]; then timeout -k 10 300 python tools/me_phase_profile.py 1 2 5 7 > gpurun_out/me_phase.log 2>&1 || tail -5 gpurun_out/me_phase.log; fi
find gpurun_out/${R}s gpurun_out/${R}f gpurun_out/${R}w gpurun_out/${R}q -name "*.csv" | head -20
echo verify-done
