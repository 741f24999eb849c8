#!/bin/bash
# usage: tools/gpu_e2e_trace.sh TAG -- kernel + memory-copy timeline of the pipelined batch API (tools/e2e_batch.py)
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
E2E_CODES=0 E2E_ROUNDS=9 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/tools/e2e_batch.py > $O/e2e.log 2>&1
tail -2 $O/e2e.log
python3 - <<PY
import csv,glob
k=glob.glob('$O/trace/**/*kernel_trace.csv',recursive=True)[0]
m=glob.glob('$O/trace/**/*memory_copy_trace.csv',recursive=True)[0]
ev=[]
for r in csv.DictReader(open(k)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'K '+r['Kernel_Name'].split('(')[0][-28:], ''))
for r in csv.DictReader(open(m)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'C '+r.get('Direction',r.get('Name','')), r.get('Bytes', r.get('Size',''))))
ev.sort()
t0=ev[0][0]
tail=[e for e in ev if e[0] > ev[-1][0]-40_000_000]
with open('$O/timeline.txt','w') as f:
    for s,e,n,b in tail:
        if e-s > 20000 or n.startswith('C'):
            f.write('%10.3f %10.3f %8.3f  %s %s\n'%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n,b))
PY
