#!/bin/bash
# usage: tools/gpu_final_r4.sh TAG PART   -- what the round's profiles/ are made from (run on the GPU box; each part fits one gpurun call)
#   a: smoke, all GPU tests, the K1 bench lines (configs 2-5)       b: the K2 bench lines, stage 2 alone, a small batch, the micro-benchmarks
#   c: rocprofv3 kernel stats + PMC passes for configs 2, 3, 4 (K1p) and 5 (serial K1)           d: the CLI over a directory
TAG=$1; PART=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bench() { # name args...
  local name=$1; shift
  timeout -k 10 500 python3 $R/bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench_$name.json'));c=d.get('cpu_baseline') or {};e=d.get('e2e') or {};e8=d.get('e2e_cabac8') or {};print('$name', round(d['ms_per_step'],3),'ms', round(d['value']/1e9,3),'GB/s frac', round(d['roofline']['frac'],4), '| cpu', c.get('kind'), round(c.get('value',0)/1e6,1), 'MB/s x', round(d.get('gpu_over_cpu',0),1), c.get('parity_vs_gpu'), '| e2e', round(e.get('value',0)/1e9,2), 'cabac8', round(e8.get('value',0)/1e9,2))"
}
if [ "$PART" = a ]; then
  python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
  tail -1 $O/smoke.log
  timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
  tail -1 $O/tests.log
  bench w2; bench w3 --workload 3; bench w4 --workload 4; bench w5 --workload 5
elif [ "$PART" = b ]; then
  bench w2_resolved --records resolved --no-cpu-baseline --no-e2e
  bench w2_s128 --slices 128 --no-cpu-baseline --no-e2e
  bench w2_k2 --workload 2 --kind range --steps 5 --warmup 1
  bench w4_k2 --workload 4 --kind range --steps 5 --warmup 1
  bench w5_k2 --workload 5 --kind range
  bench w3_k2 --workload 3 --kind range --steps 2 --warmup 1 --no-cpu-baseline --no-e2e
  $R/tools/ubench/issue_rate > $O/ubench_issue_rate.txt 2>&1; tail -1 $O/ubench_issue_rate.txt
  $R/tools/ubench/lds_chain > $O/ubench_lds_chain.txt 2>&1; tail -1 $O/ubench_lds_chain.txt
  $R/tools/ubench/read_patterns > $O/ubench_read_patterns.txt 2>&1; tail -1 $O/ubench_read_patterns.txt
elif [ "$PART" = c ]; then
  SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
  prof() { # dir args...
    local dir=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${dir}_stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 5 --warmup 1 "$@" > /dev/null 2>&1 || exit 1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${dir}_fetch -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 3 --warmup 1 "$@" > /dev/null 2>&1 || exit 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${dir}_write -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 3 --warmup 1 "$@" > /dev/null 2>&1 || exit 1
    rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${dir}_sq -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 1 "$@" > /dev/null 2>&1 || exit 1
    echo profiles $dir done
  }
  prof w2
  prof w5 --workload 5
  prof w3 --workload 3
  prof w4 --workload 4
  echo all done
else
  # `recode test <dir>` over 16 copies of each of the image's two clips: the directory's files together against a file at a time
  IMG=/opt/conda/lib/python3.9/site-packages/imageio/resources/images
  for MODE in together sequential; do
    D=/tmp/cli_$MODE; rm -rf $D; mkdir -p $D
    for k in 00 01 02 03 04 05 06 07 08 09 10 11 12 13 14 15; do cp $IMG/realshort.mp4 $D/${k}_realshort.mp4; cp $IMG/cockatoo.mp4 $D/${k}_cockatoo.mp4; done
  done
  S0=$(date +%s.%N); AVR_TIMING=1 $R/avrecode-ms_amd/recode test /tmp/cli_together > $O/cli_together.out 2> $O/cli_together.err; S1=$(date +%s.%N)
  AVR_TEST_SEQUENTIAL=1 $R/avrecode-ms_amd/recode test /tmp/cli_sequential > $O/cli_sequential.out 2> $O/cli_sequential.err; S2=$(date +%s.%N)
  python3 - <<PY > $O/cli_timing.txt
import filecmp, os
a, b = "/tmp/cli_together/output", "/tmp/cli_sequential/output"
names = sorted(n for n in os.listdir(a) if n.endswith(".mp4"))
same = all(filecmp.cmp(os.path.join(a, n), os.path.join(b, n), shallow=False) for n in names)
size = sum(os.path.getsize(os.path.join("/tmp/cli_together", n)) for n in names)
t1, t2 = $S1 - $S0, $S2 - $S1
print("recode test <dir> on the GPU box: %d files (16 x realshort.mp4 + 16 x cockatoo.mp4, %.1f MB)" % (len(names), size / 1e6))
print("  the directory's files together (one file per host thread, one GPU batch per direction for all of them): %.2f s = %.1f MB/s" % (t1, size / 1e6 / t1))
print("  a file at a time, a batch per file and direction (AVR_TEST_SEQUENTIAL=1: rounds 1-3):                  %.2f s = %.1f MB/s" % (t2, size / 1e6 / t2))
print("  ratio %.1f; the %d output files are %s; failures: %s / %s" % (t2 / t1, len(names), "identical" if same else "DIFFERENT",
      open("$O/cli_together.out").read().count("failed on"), open("$O/cli_sequential.out").read().count("failed on")))
print("  phases of the batched run (AVR_TIMING=1), summed over the 32 files' threads where they are per file:")
import collections
agg = collections.defaultdict(float)
for line in open("$O/cli_together.err"):
    if line.startswith("[timing]"):
        agg[line[9:38].strip()] += float(line[38:].split()[0])
for k, v in agg.items(): print("    %-30s %9.1f ms" % (k, v))
PY
  cat $O/cli_timing.txt
fi
