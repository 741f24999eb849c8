#!/bin/bash
# usage: tools/gpu_r3_k2.sh TAG  -- GPU tests, then the compress direction (K2 / K2p) at full size with its CPU baseline
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for W in 2 4 5; do
  timeout -k 10 300 python3 $R/bench.py --workload $W --kind range --steps 3 --warmup 1 > $O/bench_w${W}_k2.json 2> $O/bench_w${W}_k2.err || { tail -5 $O/bench_w${W}_k2.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_w${W}_k2.json'));print('w$W K2', round(d['ms_per_step'],3),'ms', round(d['value']/1e9,3),'GB/s gpu/cpu', d.get('gpu_over_cpu'), d['cpu_baseline']['parity_vs_gpu'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/w2_k2_stats -- python3 $R/bench.py --no-cpu-baseline --workload 2 --kind range --steps 2 --warmup 1 > /dev/null 2>&1
find $O/w2_k2_stats -name "*kernel_stats.csv" | head -1 | xargs head -8
