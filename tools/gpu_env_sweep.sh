#!/bin/bash
# usage: tools/gpu_env_sweep.sh VAR "v1 v2 ..." [bench args]  -- ms per step of bench.py under each value of an environment switch
VAR=$1; shift; VALS=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  export $VAR=$v
  python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" > $R/gpurun_out/sweep_${VAR}_$v.json 2>/dev/null || { echo "$VAR=$v FAILED"; continue; }
  python3 -c "import json; d=json.load(open('$R/gpurun_out/sweep_${VAR}_$v.json')); print('$VAR=$v', round(d['ms_per_step'],4), 'ms  status_errors', d['slice_status_errors'])"
done
