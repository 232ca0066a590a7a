#!/bin/bash
# bench.py (ME + pred + RD step) against build variants of libsvthip.so: prints kernel_ms per variant
for v in "$@"; do
  if [ "$v" = "default" ]; then lib=$PWD/svt-av1-psyex_amd/libsvthip.so; else lib=$PWD/svt-av1-psyex_amd/variants/lib_$v.so; fi
  SVT_HIP_LIBRARY=$lib timeout -k 5 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernel_ms'])" || exit 1
done
