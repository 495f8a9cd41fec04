#!/bin/bash
# round 5, GPU call V: the final profile (tools/r05/profile.sh + the ts_hybhol counters) on the final sources, then the whole GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
bash tools/r05/profile.sh > gpurun_out/r05/v_profile.log 2>&1
bash tools/r05/hybhol_pmc.sh > gpurun_out/r05/v_hybhol_pmc.log 2>&1
timeout 1200 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05/v_suite.log 2>&1
tail -3 gpurun_out/r05/v_suite.log
