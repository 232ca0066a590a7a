#!/bin/bash
# Round-end verification on the GPU box: full GPU suite, smoke, the bench line, and the rocprofv3 passes profiles/ is built from.
# Run through gpurun from the repo root; outputs land in gpurun_out/ (then: python3 tools/refresh_profiles.py).
set -e -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/bench_final.log 2> gpurun_out/bench_final.err || { tail -20 gpurun_out/bench_final.err; exit 1; }
tail -1 gpurun_out/bench_final.log | cut -c1-400
rm -rf gpurun_out/r01s gpurun_out/r01f gpurun_out/r01w gpurun_out/pmc7
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01s -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01s.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r01f -o bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01f.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r01w -o bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01w.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d gpurun_out/pmc7 -o a --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc7.log 2>&1
find gpurun_out/r01s gpurun_out/r01f gpurun_out/r01w gpurun_out/pmc7 -name "*.csv" | head -20
echo verify-done
