#!/bin/bash
# Runs on the GPU box: tools/<soup|grid>_bench.py once per experimental build of the device library in build_exp/
# (tools/build_variants.sh), streaming schedule only. usage: tools/variants_run.sh <soup|grid> <what> name [name...]
set -o pipefail
WHICH=$1; WHAT=$2; shift 2
R=$GRAFT_REPO_ROOT
for V in "$@"; do
  LIB=$R/build_exp/lib_$V.so
  [ "$V" = "default" ] && LIB=$R/raytracing_amd/lib/libraytracing_amd.so
  RT_AMD_DEVICE_LIB=$LIB RT_HBM_WHAT=$WHAT RT_HBM_SCHEDULES=${RT_VR_SCHEDULE:-stream} timeout -k 10 240 python3 -u $R/tools/${WHICH}_bench.py > $R/gpurun_out/var_${WHICH}_$V.txt 2>&1 || { echo "$V FAILED"; tail -3 $R/gpurun_out/var_${WHICH}_$V.txt; exit 1; }
  python3 - "$V" $R/gpurun_out/var_${WHICH}_$V.txt <<'PY'
import json, sys
out = []
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l)
        if any(k in d for k in ("stream", "refill")):
            out.append("%s %.0f" % (d["workload"].split(",")[-1].strip().split()[0], (d.get("refill") or d.get("stream"))["Mrays_per_s"]))
        else:
            out.append("path %.0f Msamples/s" % d["Msamples_per_s"])
print("%-10s %s" % (sys.argv[1], " | ".join(out)))
PY
done
