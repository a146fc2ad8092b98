#!/bin/bash
# Runs on the GPU box: the scenes whose tables live in HBM (tools/soup_bench.py = 1 M-triangle soup, tools/grid_bench.py =
# 4.19 M-triangle mesh). Plain run (all workloads, both schedules), then a rocprofv3 kernel trace and one --pmc pass per
# counter group of the streaming closest-hit kernel on incoherent rays, then the same for the path tracer.
# usage: tools/gpu_hbm_profile.sh <tag> <soup|grid> [pmc=1]
set -o pipefail
TAG=${1:-hbm}; WHICH=${2:-soup}; PMC=${3:-1}
R=$GRAFT_REPO_ROOT
PROG=$R/tools/${WHICH}_bench.py
OUT=$R/gpurun_out/${TAG}_${WHICH}
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 python3 -u $PROG > $OUT/plain.log 2>&1 || { echo PLAIN FAILED; tail -5 $OUT/plain.log; exit 1; }
cat $OUT/plain.log
cd /tmp && export TMPDIR=/tmp
for W in incoherent path; do
  export RT_HBM_WHAT=$W RT_HBM_SCHEDULES=stream
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$W -o prof -- python3 $PROG > $OUT/kt_$W.log 2>&1 || { echo KT $W FAILED; tail -5 $OUT/kt_$W.log; exit 1; }
  python3 $R/tools/prof_summary.py $OUT/kt_$W/prof_results.db $OUT/kt_$W.md | head -12
  [ "$PMC" = "1" ] || continue
  i=0
  for GROUP in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    mkdir -p $OUT/pmc_$W
    timeout -s KILL 300 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/pmc_$W/g$i -o pmc -- python3 $PROG > $OUT/pmc_$W/g$i.log 2>&1 || { echo "pmc pass $i ($GROUP) failed"; tail -3 $OUT/pmc_$W/g$i.log; exit 1; }
  done
  python3 $R/tools/pmc_summary.py $OUT/pmc_$W $OUT/pmc_$W.md > /dev/null
  grep -A40 "k_trace_closest_stream\|k_extend_stream" $OUT/pmc_$W.md | head -60
done
