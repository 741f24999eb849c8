#!/bin/bash
# usage: tools/gpu_iter2.sh TAG [bench args]  -- K1p GPU tests, a bench line, a kernel trace and two SQ counter passes (run on the GPU box)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
bash $R/tools/gpu_iter.sh $TAG "$@" || exit 1
bash $R/tools/gpu_pmc.sh ${TAG}_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "$@"
bash $R/tools/gpu_pmc.sh ${TAG}_sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "$@"
