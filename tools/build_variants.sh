#!/bin/bash
# Builds experimental variants of the device library into build_exp/: tools/build_variants.sh name "-DFLAG=..." [name flags]...
cd "$(dirname "$0")/.."
mkdir -p build_exp
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function $f -shared -o build_exp/lib_$n.so raytracing_amd/csrc/device/rt_device.hip 2>&1 | grep -E "error" ; echo "built $n" ) &
done
wait
