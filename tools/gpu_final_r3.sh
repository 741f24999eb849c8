#!/bin/bash
# usage: tools/gpu_final_r3.sh TAG PART   -- what the round's profiles/ are made from (run on the GPU box; each part fits one gpurun call)
#   a: smoke, all GPU tests, the K1 bench lines (configs 2-5)       b: the K2 bench lines, stage 2 alone, a small batch
#   c: rocprofv3 kernel stats + PMC passes (config 2 K1p, config 5 K1 and its LDS-row emitter variant, config 2 / 5 K2)
TAG=$1; PART=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bench() { # name args...
  local name=$1; shift
  timeout -k 10 500 python3 $R/bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_$name.json'));c=d.get('cpu_baseline') or {};e=d.get('e2e') or {};print('$name', round(d['ms_per_step'],3),'ms', round(d['value']/1e9,3),'GB/s frac', round(d['roofline']['frac'],4), '| cpu', c.get('kind'), round(c.get('value',0)/1e6,1), 'MB/s x', round(d.get('gpu_over_cpu',0),1), c.get('parity_vs_gpu'), '| e2e', round(e.get('value',0)/1e9,2))"
}
if [ "$PART" = a ]; then
  python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
  tail -1 $O/smoke.log
  timeout -k 10 600 python3 -m pytest $R/tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
  tail -1 $O/tests.log
  bench w2; bench w3 --workload 3; bench w4 --workload 4; bench w5 --workload 5
  bench w2_resolved --records resolved --no-cpu-baseline --no-e2e
  bench w2_s128 --slices 128 --no-cpu-baseline --no-e2e
  bench w2_s128_whole_chains --slices 128 --no-cpu-baseline --no-e2e --test-hook chain_whole=1
  $R/tools/ubench/chain_latency > $O/chain_latency.txt 2>&1; tail -2 $O/chain_latency.txt
elif [ "$PART" = b ]; then
  bench w2_k2 --workload 2 --kind range --steps 5 --warmup 1
  bench w4_k2 --workload 4 --kind range --steps 5 --warmup 1
  bench w5_k2 --workload 5 --kind range
  bench w3_k2 --workload 3 --kind range --steps 2 --warmup 1 --no-cpu-baseline --no-e2e
  bench w5_lds_rows --workload 5 --no-cpu-baseline --no-e2e --test-hook k1_emit_lds=1
  bench w5_ref_form --workload 5 --no-cpu-baseline --no-e2e --test-hook k1_form_ref=1
  bench w5_norm_lds_rows --workload 5 --no-cpu-baseline --no-e2e --test-hook k1_emit_lds=2
else
  SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
  prof() { # dir steps args...
    local dir=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${dir}_stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 5 --warmup 1 "$@" > /dev/null 2>&1 || exit 1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${dir}_fetch -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 3 --warmup 1 "$@" > /dev/null 2>&1 || exit 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${dir}_write -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 3 --warmup 1 "$@" > /dev/null 2>&1 || exit 1
    rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${dir}_sq -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 1 "$@" > /dev/null 2>&1 || exit 1
    echo profiles $dir done
  }
  prof w2
  prof w5 --workload 5
  prof w5lds --workload 5 --test-hook k1_emit_lds=1
  prof w5ref --workload 5 --test-hook k1_form_ref=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/w2k2_stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e --workload 2 --kind range --steps 3 --warmup 1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/w4k2_stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e --workload 4 --kind range --steps 3 --warmup 1 > /dev/null 2>&1
  echo all done
fi
