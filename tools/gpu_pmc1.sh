#!/bin/bash
# usage: tools/gpu_pmc1.sh TAG "COUNTERS" KERNEL_SUBSTRING "<bench args>" "hooks;hooks;..."  -- per-launch maximum of PMC counters of one kernel under sets of test hooks
TAG=$1; CTR=$2; KERN=$3; ARGS=$4; SETS=$5
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra LIST <<< "$SETS"
i=0
for S in "${LIST[@]}"; do
  i=$((i+1))
  H=""; for kv in $S; do H="$H --test-hook $kv"; done
  rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $O/s$i -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 $ARGS $H > $O/s$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob('$O/s$i/**/*counter_collection.csv',recursive=True)[0]
m=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if '$KERN' in r['Kernel_Name']: m[r['Counter_Name']]=max(m[r['Counter_Name']], float(r['Counter_Value']))
print('$S'.ljust(36), ' '.join('%s=%.4g' % kv for kv in sorted(m.items())))
PY
done
