#!/bin/bash
# Runs on the GPU box: two extra rocprofv3 --pmc passes over a program (latency levels of the memory paths and instruction
# fetch; instruction classes). usage: [PROG=tools/pass_trace.py] tools/gpu_pmc_extra.sh <tag>
set -o pipefail
TAG=$1
R=$GRAFT_REPO_ROOT
PROG=${PROG:-tools/pass_trace.py}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/g$i -o pmc -- python3 $R/$PROG > $OUT/g$i.log 2>&1 || { echo "pmc pass $i failed"; tail -n 3 $OUT/g$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $OUT
