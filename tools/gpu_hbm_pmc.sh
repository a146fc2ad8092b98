#!/bin/bash
# Runs on the GPU box: hardware counters of one batch-tracer schedule on one scene in HBM, incoherent rays only, one
# rocprofv3 --pmc pass per counter group. usage: tools/gpu_hbm_pmc.sh <tag> <soup|grid> <schedule> [lib]
set -o pipefail
TAG=$1; WHICH=$2; SCHED=$3; LIB=$4
R=$GRAFT_REPO_ROOT
[ -n "$LIB" ] && export RT_AMD_DEVICE_LIB=$R/$LIB
export RT_HBM_WHAT=incoherent RT_HBM_SCHEDULES=$SCHED
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/g$i -o pmc -- python3 $R/tools/${WHICH}_bench.py > $OUT/g$i.log 2>&1 || { echo "pmc pass $i ($GROUP) failed"; tail -3 $OUT/g$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $OUT $OUT.md > /dev/null
grep -B1 -A32 "^### k_trace_closest" $OUT.md | head -80
