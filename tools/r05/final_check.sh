cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05/final_suite.log 2>&1
grep -E "passed|failed" gpurun_out/r05/final_suite.log | tail -1
