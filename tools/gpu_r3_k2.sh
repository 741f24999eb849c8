#!/bin/bash
# usage: tools/gpu_r3_k2.sh TAG  -- GPU tests, then the compress direction (K2 / K2p) at full size with its CPU baselines
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -x -q --durations=12 > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -16 $O/tests.log
for W in 2 4 5; do
  timeout -k 10 400 python3 $R/bench.py --workload $W --kind range --steps 3 --warmup 1 > $O/bench_w${W}_k2.json 2> $O/bench_w${W}_k2.err || { tail -5 $O/bench_w${W}_k2.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_w${W}_k2.json'));c=d['cpu_baseline'];print('w$W K2', round(d['ms_per_step'],3),'ms', round(d['value']/1e9,3),'GB/s | cpu', c['kind'], round(c['value']/1e6,1),'MB/s on',c['cores'],'| port',round(c['port_value']/1e6,1),'| gpu/cpu', round(d.get('gpu_over_cpu'),2), c['parity_vs_gpu'], '| e2e', round(d['e2e']['value']/1e9,3) if 'e2e' in d else None)"
done
timeout -k 10 400 python3 $R/bench.py > $O/bench_w2.json 2> $O/bench_w2.err || { tail -5 $O/bench_w2.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_w2.json'));c=d['cpu_baseline'];print('w2 K1', round(d['ms_per_step'],3),'ms', round(d['value']/1e9,3),'GB/s | cpu', c['kind'], round(c['value']/1e6,1),'MB/s on',c['cores'],'| port',round(c['port_value']/1e6,1),'| gpu/cpu', round(d.get('gpu_over_cpu'),2), c['parity_vs_gpu'], '| e2e', round(d['e2e']['value']/1e9,3), d['e2e']['ms_per_batch'])"
