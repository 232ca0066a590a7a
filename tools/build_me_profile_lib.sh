#!/bin/bash
# Diagnostic builds of libsvthip.so with -DSVT_HIP_ME_PROFILE: one library per kernel of the ME chain whose phases are wanted
# (tools/_libprof<mode>.so; modes as in me_kernel.hip: 0 one-kernel form, 1 mid1, 2 s1, 4 mid2, 5 s2, 7 tail).  usage: tools/build_me_profile_lib.sh [mode ...]
set -e
cd /root/repo/svt-av1-psyex_amd/csrc
for M in ${@:-7 1 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -Wno-pass-failed -DSVT_HIP_ME_PROFILE -DSVT_HIP_ME_PROFILE_MODE=$M -x hip -c me_kernel.hip -o /tmp/me_kernel_prof$M.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/_libprof$M.so /tmp/me_kernel_prof$M.o $(ls build/*.hip.o build/*.cpp.o | grep -v me_kernel.hip.o)
done
