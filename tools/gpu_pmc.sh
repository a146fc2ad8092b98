#!/bin/bash
# Runs on the GPU box: hardware counters for the bench workload, one rocprofv3 --pmc pass per counter group
# (FETCH_SIZE and WRITE_SIZE cannot share a pass). usage: [PROG=tools/trace_bench.py] tools/gpu_pmc.sh <tag> [program args...]
set -o pipefail
TAG=${1:-pmc}; shift
ARGS=${@:---steps 4 --warmup 0 --no-cpu-baseline --no-secondary --no-strong --no-c1}
R=$GRAFT_REPO_ROOT
PROG=${PROG:-bench.py}
OUT=$R/gpurun_out/${TAG}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/g$i -o pmc -- python3 $R/$PROG $ARGS > $OUT/g$i.log 2>&1 || { echo "pmc pass $i ($GROUP) failed"; tail -3 $OUT/g$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $OUT
