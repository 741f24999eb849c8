#!/bin/bash
# usage: tools/gpu_stats.sh TAG "<bench args>" -- rocprofv3 kernel stats of one bench.py run, the library's kernels listed
TAG=$1; ARGS=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 5 --warmup 1 $ARGS > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "avr::" in r["Name"] and "synth" not in r["Name"]:
        print(r["Name"].split("(")[0][:44].ljust(44), r["Calls"].rjust(4), "%9.1f us" % (float(r["AverageNs"]) / 1e3))
PY
