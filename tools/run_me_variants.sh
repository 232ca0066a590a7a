#!/bin/bash
for v in "$@"; do
  SVT_HIP_LIBRARY=$PWD/svt-av1-psyex_amd/variants/lib_$v.so timeout -k 5 120 python tools/me_variant_perf.py 2>&1 | tail -1 || exit 1
done
