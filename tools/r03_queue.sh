#!/bin/bash
# Runs on the GPU box: the work-queue schedule beside the other two on the batch tracers: cornellBox (tools/trace_bench.py),
# 1 M-triangle soup and 4 M-triangle mesh (camera and incoherent rays).
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
RT_TB_QUICK=1 timeout -k 10 300 python3 -u tools/trace_bench.py > gpurun_out/r03_queue_cornell.txt 2>&1 || { echo cornell FAILED; tail -5 gpurun_out/r03_queue_cornell.txt; exit 1; }
cat gpurun_out/r03_queue_cornell.txt
for W in soup grid; do
RT_HBM_WHAT=primary,incoherent RT_HBM_SCHEDULES=lockstep,stream,queue timeout -k 10 300 python3 -u tools/${W}_bench.py > gpurun_out/r03_queue_$W.txt 2>&1 || { echo $W FAILED; tail -5 gpurun_out/r03_queue_$W.txt; exit 1; }
python3 - gpurun_out/r03_queue_$W.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["workload"], {k: round(d[k]["Mrays_per_s"]) for k in ("lockstep", "stream", "queue") if k in d})
PY
done
