#!/bin/bash
# usage: tools/gpu_prof_all.sh TAG [bench args] -- tests + bench + kernel trace + SQ counters + FETCH/WRITE sizes (run on the GPU box)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
bash $R/tools/gpu_iter2.sh $TAG "$@" > $R/gpurun_out/${TAG}_iter2.txt 2>&1 || { tail -5 $R/gpurun_out/${TAG}_iter2.txt; exit 1; }
bash $R/tools/gpu_pmc.sh ${TAG}_fetch "FETCH_SIZE" "$@" > $R/gpurun_out/${TAG}_fetch.txt 2>&1
bash $R/tools/gpu_pmc.sh ${TAG}_write "WRITE_SIZE" "$@" > $R/gpurun_out/${TAG}_write.txt 2>&1
grep -v "^avr::k_synth\|at::" $R/gpurun_out/${TAG}_iter2.txt | head -40
