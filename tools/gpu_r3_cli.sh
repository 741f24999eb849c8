#!/bin/bash
# usage: tools/gpu_r3_cli.sh TAG -- the phases of `recode roundtrip` on the image's clips (AVR_TIMING=1), twice each (cold, warm)
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
IM=/opt/conda/lib/python3.9/site-packages/imageio/resources/images
TIMEFORMAT='wall %R s'
for H in 0 1; do for F in realshort.mp4 cockatoo.mp4; do
  for rep in 1 2; do
    echo "== $F hooks=$H rep=$rep"
    time ( AVR_TIMING=1 AVR_MODEL_HOOKS=$H $R/avrecode-ms_amd/recode roundtrip $IM/$F /tmp/out.recode 2>&1 | grep -v "^Input\|Duration\|Stream" )
  done
done; done > $O/cli_timing.txt 2>&1
cat $O/cli_timing.txt
