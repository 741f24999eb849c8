#!/bin/bash
# usage: tools/gpu_exp.sh TAG ENV=VAL [bench args] -- kernel times of one bench run under an experiment switch (no parity check)
TAG=$1; shift; E=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export $E
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/${TAG}_stats/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k1p' in r['Name'] or 'k_cabac' in r['Name'] or 'k_range' in r['Name']: print(r['Name'][:40].ljust(40), r['Calls'], round(float(r['AverageNs'])/1e6,4))
PY
