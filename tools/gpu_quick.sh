#!/bin/bash
# usage: tools/gpu_quick.sh TAG "<pytest -k expression>" "<bench args>;<bench args>;..."   -- a subset of the GPU tests and a few bench lines
TAG=$1; KEXPR=$2; BENCHES=$3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q -k "$KEXPR" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
  tail -3 $O/tests.log
fi
IFS=';' read -ra LIST <<< "$BENCHES"
i=0
for B in "${LIST[@]}"; do
  i=$((i+1))
  timeout -k 10 400 python3 $R/bench.py $B > $O/bench_$i.json 2> $O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_$i.json'));c=d.get('cpu_baseline');print('$B |', round(d['ms_per_step'],3),'ms', round(d['value']/1e9,3),'GB/s frac', round(d['roofline']['frac'],4), '| parity', c['parity_vs_gpu'] if c else None, '| errors', d['slice_status_errors'])"
done
