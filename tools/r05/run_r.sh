#!/bin/bash
# round 5, GPU call R: final profile (tools/r05/profile.sh) on the sources with the gamma step's scalar pairs, then the whole GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
bash tools/r05/profile.sh > gpurun_out/r05/r_profile.log 2>&1
timeout 1200 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05/r_suite.log 2>&1
tail -3 gpurun_out/r05/r_suite.log
