#!/bin/bash
# usage: tools/gpu_prof.sh TAG [bench args]   -- a bench line (no CPU baseline, no e2e) and a kernel trace of the same command (run on the GPU box)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || { tail -5 $R/gpurun_out/${TAG}_bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 5 --warmup 1 "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,json
d=json.load(open('$R/gpurun_out/${TAG}_bench.json'))
print('$TAG', '$*', '| ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4), 'errors', d['slice_status_errors'])
f=glob.glob('$R/gpurun_out/${TAG}_stats/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if ('k1p' in r['Name'] or 'k_cabac' in r['Name'] or 'k_range' in r['Name'] or 'k2p' in r['Name'] or 'k1_' in r['Name']) and 'synth' not in r['Name']:
        n=r['Name'].replace('void ','').replace('avr::','').split('(')[0]
        print('  ',n[:44].ljust(44), r['Calls'].rjust(3), round(float(r['AverageNs'])/1e6,4))
        tot+=float(r['TotalDurationNs'])/1e6
print('   sum of kernels per step (6 steps):', round(tot/6,4))
PY
