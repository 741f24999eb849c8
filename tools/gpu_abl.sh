#!/bin/bash
# usage: tools/gpu_abl.sh TAG KERNEL_SUBSTRING "<bench args>" "hook=value hook=value;hook=value;..."  -- the average duration of one kernel under sets of test hooks (timing only)
TAG=$1; KERN=$2; ARGS=$3; SETS=$4
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra LIST <<< "$SETS"
i=0
for S in "${LIST[@]}"; do
  i=$((i+1))
  H=""; for kv in $S; do H="$H --test-hook $kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$i -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 4 --warmup 1 $ARGS $H > $O/s$i.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('$O/s$i/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if '$KERN' in r['Name']: print('$S'.ljust(40), r['Name'].replace('void ','').replace('avr::','').split('(')[0][:36].ljust(36), round(float(r['AverageNs'])/1e6,4))
PY
done
