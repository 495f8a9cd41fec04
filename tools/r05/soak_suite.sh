#!/bin/bash
# the GPU suite twice in a row (flakiness check before the round ends)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for i in 1 2; do
  timeout 1500 python3 -m pytest tests -m gpu -q --durations=8 > $O/h_suite_$i.log 2>&1
  grep -n "passed\|failed" $O/h_suite_$i.log | tail -2
done
