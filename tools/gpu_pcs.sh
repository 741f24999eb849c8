#!/bin/bash
# usage: tools/gpu_pcs.sh TAG METHOD INTERVAL UNIT "<bench args>"  -- PC sampling of one bench run (rocprofv3 beta); raw csv under gpurun_out/TAG
TAG=$1; METHOD=$2; INTERVAL=$3; UNIT=$4; ARGS=$5
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INTERVAL --kernel-trace --output-format csv -d $O/pcs -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 $ARGS > $O/pcs.log 2>&1
echo "rc $?"; tail -5 $O/pcs.log; find $O/pcs -type f | head; 
