#!/bin/bash
# Runs on the GPU box: smoke parity, bench (short), rocprofv3 kernel trace of bench, per-kernel summary.
# usage: tools/gpu_profile.sh <tag> [steps]
set -o pipefail
TAG=${1:-run}
STEPS=${2:-16}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 200 python -u $R/__graft_entry__.py smoke > $R/gpurun_out/${TAG}_smoke.log 2>&1 || { echo SMOKE FAILED; tail -5 $R/gpurun_out/${TAG}_smoke.log; exit 1; }
tail -1 $R/gpurun_out/${TAG}_smoke.log
timeout -k 10 300 python -u $R/bench.py --steps 32 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || { echo BENCH FAILED; tail -5 $R/gpurun_out/${TAG}_bench.err; exit 1; }
python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/${TAG}_bench.json'))
print('Mrays/s %.1f  Msamples/s %.1f  ms/step %.3f  extend avg ms %.4f' % (d['value'], d['samples_per_sec']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms']))"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o prof -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-secondary --no-strong --no-c1 > $R/gpurun_out/${TAG}_prof.log 2>&1 || { echo PROF FAILED; tail -5 $R/gpurun_out/${TAG}_prof.log; exit 1; }
python3 $R/tools/prof_summary.py $R/gpurun_out/${TAG}_prof/prof_results.db | head -14
python3 - <<PY
import sqlite3
db = sqlite3.connect('$R/gpurun_out/${TAG}_prof/prof_results.db')
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i,r in enumerate(rows) if r[0].startswith('k_raygen')]
i0, i1 = idx[-3], idx[-2]
line = []
for n,s,e in rows[i0:i1]:
    line.append('%s:%.0f' % (n.replace('void ','')[:8], (e-s)/1e3))
print('one pass (us): ' + ' '.join(line[:40]) + ' ... pass total %.2f ms' % ((rows[i1][1]-rows[i0][1])/1e6))
PY
