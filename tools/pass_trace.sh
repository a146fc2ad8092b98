#!/bin/bash
# Runs on the GPU box: kernel trace of tools/pass_trace.py; prints the launches of the last pass in order with their durations.
# usage: tools/pass_trace.sh <tag> [lib]
set -o pipefail
TAG=$1; LIB=$2
R=$GRAFT_REPO_ROOT
[ -n "$LIB" ] && export RT_AMD_DEVICE_LIB=$R/$LIB
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT -o prof -- python3 $R/tools/pass_trace.py > $OUT/run.log 2>&1 || { echo "trace failed"; tail -n 5 $OUT/run.log; exit 1; }
python3 - $OUT/prof_results.db <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith('k_raygen')]
i0 = idx[-1]
tot = {}
line = []
for n, s, e in rows[i0:]:
    k = re.sub(r"[<(].*", "", n.replace('void ', ''))
    tot[k] = tot.get(k, 0) + (e - s) / 1e3
    line.append('%s %.0f' % (k.replace('k_', ''), (e - s) / 1e3))
print(' | '.join(line[:70]))
print('totals (us): ' + ', '.join('%s %.0f' % kv for kv in sorted(tot.items(), key=lambda kv: -kv[1])) + ' ; pass %.2f ms' % ((rows[-1][2] - rows[i0][1]) / 1e6))
PY
