#!/bin/bash
set -e
cd /root/repo/svt-av1-psyex_amd/csrc
build() {
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -x hip -c me_kernel.hip -o build/me_kernel_$name.o "$@" -Rpass-analysis=kernel-resource-usage 2> build/$name.log
  objs=$(ls build/*.o | grep -v me_kernel)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/lib_$name.so build/me_kernel_$name.o $objs
  grep -E "VGPRs:|ScratchSize" build/$name.log | head -5 | tr '\n' ' '; echo " <- $name"
}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  build $name $flags
done
