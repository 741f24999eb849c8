#!/bin/bash
# usage: tools/gpu_calib.sh TAG -- FETCH_SIZE / WRITE_SIZE of tools/ubench/hbm_patterns (known byte counts per access pattern)
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $R/tools/ubench/hbm_patterns > $O/calib.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- $R/tools/ubench/hbm_patterns >> $O/calib.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/calib_stats -- $R/tools/ubench/hbm_patterns >> $O/calib.txt 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for c in ("fetch", "write"):
    f = glob.glob("$O/calib_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(c, k, "KB per launch:", [round(x) for x in v])
PY
