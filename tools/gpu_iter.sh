#!/bin/bash
# usage: tools/gpu_iter.sh TAG [bench args]   -- K1p GPU tests, a bench line and a kernel trace (run on the GPU box)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_gpu_k1p.py -x -q -m gpu > $R/gpurun_out/${TAG}_tests.log 2>&1; tail -3 $R/gpurun_out/${TAG}_tests.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" > $R/gpurun_out/${TAG}_bench.json || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,json
print(json.load(open('$R/gpurun_out/${TAG}_bench.json'))['ms_per_step'])
f=glob.glob('$R/gpurun_out/${TAG}_stats/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k1p' in r['Name'] or 'k_cabac' in r['Name'] or 'k_range' in r['Name']: print(r['Name'][:40].ljust(40), r['Calls'], round(float(r['AverageNs'])/1e6,4))
PY
