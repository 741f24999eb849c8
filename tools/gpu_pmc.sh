#!/bin/bash
# usage: tools/gpu_pmc.sh TAG "COUNTERS..." [bench args]  -- one rocprofv3 --pmc pass over a 1-step bench run (run on the GPU box)
TAG=$1; shift; CTRS=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $R/gpurun_out/${TAG} -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $R/gpurun_out/${TAG}.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/${TAG}/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=(r['Kernel_Name'].split('(')[0][-24:],r['Counter_Name']); agg[k]+=float(r['Counter_Value']); cnt[k]+=1
for k in sorted(agg):
    if 'k1p' in k[0] or 'cabac' in k[0] or 'range' in k[0]: print(k[0].ljust(26),k[1].ljust(24),'%.4g'%(agg[k]/cnt[k]))
PY
