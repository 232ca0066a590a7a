#!/bin/bash
set -e
cd /root/repo/svt-av1-psyex_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -DSVT_HIP_ME_PROFILE -x hip -c me_kernel.hip -o build/me_kernel_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/_libprof.so build/me_kernel_prof.o $(ls build/*.o | grep -v me_kernel)
