#!/bin/bash
# usage: tools/gpu_final.sh TAG   -- everything the round's profiles/ are made from (run on the GPU box, ~4 min)
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python3 $R/bench.py > $O/bench_w2.json 2> $O/bench_w2.err || exit 1
python3 $R/bench.py --workload 3 > $O/bench_w3.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --workload 4 > $O/bench_w4.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --workload 5 > $O/bench_w5.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --workload 5 --kind range > $O/bench_w5_k2.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --records resolved --no-cpu-baseline > $O/bench_w2_resolved.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --workload 2 --kind range --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_w2_k2.json 2>> $O/bench.err || exit 1
python3 $R/bench.py --workload 4 --kind range --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_w4_k2.json 2>> $O/bench.err || exit 1
echo benches done
for W in 2 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/w${W}_stats -- python3 $R/bench.py --no-cpu-baseline --workload $W --steps 5 --warmup 1 > /dev/null 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/w${W}_fetch -- python3 $R/bench.py --no-cpu-baseline --workload $W --steps 3 --warmup 1 > /dev/null 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w${W}_write -- python3 $R/bench.py --no-cpu-baseline --workload $W --steps 3 --warmup 1 > /dev/null 2>&1 || exit 1
  echo profiles w$W done
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/w2_sq -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/w5_sq -- python3 $R/bench.py --no-cpu-baseline --workload 5 --steps 1 --warmup 1 > /dev/null 2>&1
echo all done
