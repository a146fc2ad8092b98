#!/bin/bash
# Runs on the GPU box: for every named build of the device library (build_exp/lib_<name>.so, "default" = the shipped one)
# tools/trace_bench.py on the path tracer's bounce rays (cornellBox), once plain for the times and once under one
# rocprofv3 --pmc pass for the wave-level instruction counts; prints, per build and schedule, rays per second and vector
# instructions per ray. usage: tools/tb_variants.sh <tag> name [name...]     env: RT_TB_ONLY (default cosine), RT_TB_SCHEDULES
set -o pipefail
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export RT_TB_ONLY=${RT_TB_ONLY:-cosine}
N=${RT_TB_N:-4194304}
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  LIB=$R/build_exp/lib_$V.so
  [ "$V" = "default" ] && LIB=$R/raytracing_amd/lib/libraytracing_amd.so
  export RT_AMD_DEVICE_LIB=$LIB
  timeout -k 10 120 python3 -u $R/tools/trace_bench.py $N 5 > $OUT/$V.txt 2>&1 || { echo "$V FAILED"; tail -3 $OUT/$V.txt; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU \
      --output-format csv -d $OUT/pmc_$V -o pmc -- python3 $R/tools/trace_bench.py $N 1 > $OUT/pmc_$V.log 2>&1 || { echo "$V pmc FAILED"; tail -3 $OUT/pmc_$V.log; exit 1; }
  python3 - "$V" $OUT/$V.txt $OUT/pmc_$V $N <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
v, txt, pmc, n = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
times = [l.rstrip() for l in open(txt) if "Grays/s" in l and "for hits" not in l]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(pmc, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"== {v}")
for t in times:
    print("  " + t.strip())
for k in sorted(acc):
    if not k.startswith("k_trace"):
        continue
    c = acc[k]
    # the last dispatches of a kernel are the timed batch's (the first closest-hit lockstep ones trace camera rays)
    last = lambda name: c[name][-1]
    print(f"  {k:45s} VALU/ray {last('SQ_INSTS_VALU') / n:6.1f}  SALU/ray {last('SQ_INSTS_SALU') / n:6.1f}  LDS/ray {last('SQ_INSTS_LDS') / n:5.1f}  "
          f"waves {last('SQ_WAVES'):.0f}  resident {last('SQ_WAVE_CYCLES') / last('SQ_BUSY_CYCLES'):5.1f}  waiting-on-inst {last('SQ_WAIT_INST_ANY') / last('SQ_WAVE_CYCLES'):.2f}")
PY
done
