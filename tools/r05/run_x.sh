#!/bin/bash
# round 5, GPU call X (end-of-round check on the final build): 400 random parity cases (new seed), then the GPU suite with durations
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python3 tools/stress_parity.py 400 91 > $O/x_stress.log 2>&1
tail -2 $O/x_stress.log
timeout 1200 python3 -m pytest tests -m gpu -q --durations=12 > $O/x_suite.log 2>&1
grep -n "passed\|failed" $O/x_suite.log | tail -2
