#!/bin/bash
# Runs on the GPU box: bench.py (cornellBox 1080p, main line only) once per experimental build in build_exp/.
# usage: tools/variants_bench.sh [steps] name [name...]
set -o pipefail
STEPS=$1; shift
R=$GRAFT_REPO_ROOT
for V in "$@"; do
  LIB=$R/build_exp/lib_$V.so
  [ "$V" = "default" ] && LIB=$R/raytracing_amd/lib/libraytracing_amd.so
  RT_AMD_DEVICE_LIB=$LIB timeout -k 10 200 python3 -u $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-secondary --no-strong > $R/gpurun_out/vb_$V.json 2> $R/gpurun_out/vb_$V.err || { echo "$V FAILED"; tail -3 $R/gpurun_out/vb_$V.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/vb_$V.json'))
print('%-8s %7.0f Mrays/s  ms/step %.2f  extend %.3f ms  shade %.3f ms' % ('$V', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_shade']['avg_launch_ms']))"
done
