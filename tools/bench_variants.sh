#!/bin/bash
# Runs on the GPU box: bench.py (cornellBox 1080p, timed region only) once per configuration and prints Mrays/s, ms per step
# and the average extend / shade launch times. A configuration is  name[,VAR=value...]  where name picks the build of the
# device library (build_exp/lib_<name>.so, "default" = the shipped one) and the VAR=value pairs go into the environment.
# usage: tools/bench_variants.sh <tag> config [config...]      env: RT_BV_STEPS (default 16)
set -o pipefail
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
STEPS=${RT_BV_STEPS:-16}
for CFG in "$@"; do
  IFS=',' read -ra PARTS <<< "$CFG"
  V=${PARTS[0]}
  LIB=$R/build_exp/lib_$V.so
  [ "$V" = "default" ] && LIB=$R/raytracing_amd/lib/libraytracing_amd.so
  ENVS=("RT_AMD_DEVICE_LIB=$LIB")
  for P in "${PARTS[@]:1}"; do ENVS+=("$P"); done
  F=$OUT/$(echo "$CFG" | tr ',=' '__').json
  env "${ENVS[@]}" timeout -k 10 200 python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-secondary --no-strong --no-c1 > $F 2> $F.err || { echo "$CFG FAILED"; tail -3 $F.err; exit 1; }
  python3 - "$CFG" $F <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r, s = d["roofline"], d["roofline_shade"]
print(f"{sys.argv[1]:60s} {d['value']:8.0f} Mrays/s  {d['ms_per_step']:6.2f} ms/step  extend {r['avg_launch_ms']:.3f} ms (alone {r.get('exclusive', {}).get('avg_launch_ms', 0):.3f})  "
      f"shade {s['avg_launch_ms']:.3f} ms (alone {s.get('exclusive', {}).get('avg_launch_ms', 0):.3f})")
PY
done
