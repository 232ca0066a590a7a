#!/bin/bash
for v in "$@"; do
  echo -n "$v: "
  SVT_HIP_LIBRARY=$PWD/svt-av1-psyex_amd/variants/lib_$v.so timeout -k 5 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'])" || exit 1
done
