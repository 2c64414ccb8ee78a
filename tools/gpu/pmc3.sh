#!/bin/bash
# usage: pmc3.sh <tag> <scene> <warm> <steps> <mode>   -> three SQ counter passes under gpurun_out/<tag>/
set -o pipefail
TAG=$1; SCENE=$2; WARM=$3; STEPS=$4; MODE=$5
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
B="SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
C="SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
for p in A B C; do
  eval "ctr=\$$p"
  rocprofv3 --pmc $ctr -d $O/pmc_$p -o p --output-format csv -- python3 tools/pmc_run.py $SCENE $WARM $STEPS $MODE > $O/pmc_$p.log 2>&1 || { tail -5 $O/pmc_$p.log; exit 1; }
done
echo "pmc $TAG done"
