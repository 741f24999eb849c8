#!/bin/bash
# usage: tools/gpu_sq.sh TAG "<bench args>" -- SQ counters per kernel of one bench.py step
TAG=$1; ARGS=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 1 $ARGS > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/sq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if "k1p" not in k and "k2p" not in k and "cabac" not in k and "range" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in agg.items():
    n = cnt[k]
    print(k[:40].ljust(40), "launches", n, " ".join("%s=%.3g" % (c.replace("SQ_", ""), x / n) for c, x in v.items()))
PY
