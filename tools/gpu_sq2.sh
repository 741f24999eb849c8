#!/bin/bash
# usage: tools/gpu_sq2.sh TAG "<bench args>" -- SQ counters (three passes: waits, pipes, LDS / vector memory) per kernel of one bench.py step
TAG=$1; ARGS=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
G3="SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/sq$i -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 1 $ARGS > $O/sq$i.log 2>&1 || { tail -5 $O/sq$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for i in (1, 2, 3):
    f = glob.glob("$O/sq%d/**/*counter_collection.csv" % i, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("avr::", "")
        if "k1p" not in k and "k2p" not in k and "cabac" not in k and "range" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open("$O/sq_table.csv", "w") as out:
    names = sorted({c for v in agg.values() for c in v})
    out.write("kernel,launches," + ",".join(names) + "\n")
    for k, v in sorted(agg.items()):
        n = max(cnt[k].values())
        out.write('"%s",%d,' % (k, n) + ",".join("%.6g" % (v.get(c, 0) / max(cnt[k].get(c, 1), 1)) for c in names) + "\n")
        w = v["SQ_WAVE_CYCLES"] / cnt[k]["SQ_WAVE_CYCLES"]
        def pct(c): return 100.0 * v.get(c, 0) / max(cnt[k].get(c, 1), 1) / max(w, 1)
        print(k[:34].ljust(34), "waves %.0f" % (v["SQ_WAVES"] / cnt[k]["SQ_WAVES"]), "| of wave-cycles: wait %.0f%% stall %.0f%% (lds %.0f%%) active %.0f%% | active valu %.0f%% lds %.0f%% vmem %.0f%% sca %.0f%%"
              % (pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"), pct("SQ_WAIT_INST_LDS"), pct("SQ_ACTIVE_INST_ANY"), pct("SQ_ACTIVE_INST_VALU"), pct("SQ_ACTIVE_INST_LDS"), pct("SQ_ACTIVE_INST_VMEM"), pct("SQ_ACTIVE_INST_SCA")))
PY
