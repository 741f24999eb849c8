#!/bin/bash
# usage: tools/gpu_e2e_env.sh -- the pipelined batch API under the runtime's copy-engine switches
R=$GRAFT_REPO_ROOT
cd $R
for E in "X=1" "GPU_BLIT_ENGINE_TYPE=2" "GPU_FORCE_BLIT_COPY_SIZE=0" "DEBUG_CLR_LIMIT_BLIT_WG=32" "DEBUG_CLR_LIMIT_BLIT_WG=8" "HSA_ENABLE_SDMA=0"; do
  echo "== $E"
  env $E E2E_ROUNDS=15 python3 tools/e2e_batch.py 2>&1 | grep 'in turn' | sed 's/| last run.*//'
done
