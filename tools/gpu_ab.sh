#!/bin/bash
# usage: tools/gpu_ab.sh TAG KERNEL_SUBSTRING "<bench args>" "hooks;hooks;..."  -- tools/gpu_abl.sh on this build's test library, with the
# same command on a reference build (tools/ab/base_hooks.so: the test library of an earlier commit) before and after, on the same box
TAG=$1; KERN=$2; ARGS=$3; SETS=$4
R=$GRAFT_REPO_ROOT
L=$R/avrecode-ms_amd/libavrecode_hip_hooks.so
cp $L /tmp/new_hooks.so
cp $R/tools/ab/base_hooks.so $L && echo base && bash $R/tools/gpu_abl.sh ${TAG}_b0 "$KERN" "$ARGS" "census_stride=0" || exit 1
cp /tmp/new_hooks.so $L && echo new && bash $R/tools/gpu_abl.sh ${TAG}_n "$KERN" "$ARGS" "$SETS" || exit 1
cp $R/tools/ab/base_hooks.so $L && echo base && bash $R/tools/gpu_abl.sh ${TAG}_b1 "$KERN" "$ARGS" "census_stride=0" || exit 1
cp /tmp/new_hooks.so $L
