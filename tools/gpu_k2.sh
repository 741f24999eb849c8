#!/bin/bash
# usage: tools/gpu_k2.sh TAG  -- K2 on configs 5, 2, 4: the shipped kernel, then the measurement variants
# (AVR_K2_VARIANT bit 0 = integer long division, bit 1 = range recurrence alone; AVR_K2_DEPTH=1 = shallow read-ahead)
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python3 -m pytest tests -x -q -m gpu -k "range or k2 or K2 or golden or host or roundtrip" > $O/tests.log 2>&1 || { tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
run() {  # name, env..., -- bench args
  local name=$1; shift
  env "$@" > /dev/null 2>&1
}
for CASE in "v0:" "v0_shallow:AVR_K2_DEPTH=1" "v0_deep:AVR_K2_DEPTH=4" "v1:AVR_K2_VARIANT=1" "v2:AVR_K2_VARIANT=2" ; do
  NAME=${CASE%%:*}; ENVS=${CASE#*:}
  for W in 5 2 4; do
    S=20; [ $W != 5 ] && S=3
    env $ENVS python3 bench.py --kind range --workload $W --steps $S --warmup 1 --no-cpu-baseline > $O/k2_${NAME}_w$W.json 2> $O/k2_${NAME}_w$W.err
    python3 -c "
import json,sys
j=json.loads(open('$O/k2_${NAME}_w$W.json').read().strip().splitlines()[-1])
print('$NAME w$W', '%.3f ms'%j['ms_per_step'], '%.2f GB/s'%(j['value']/1e9), 'frac %.4f'%j['roofline']['frac'])
" || true
  done
done
