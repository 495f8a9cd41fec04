# Round 6, call b: the hybrid / hybhol tests after their mirrors followed the new item counts, then the CLI end to end (cli_e2e.sh)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
(timeout 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_hybhol.py -q -m gpu -x --durations=8 > $O/b_tests.log 2>&1; echo "exit $?" >> $O/b_tests.log)
tail -14 $O/b_tests.log
bash tools/r06/cli_e2e.sh 2>&1 | tee $O/b_cli_e2e.log
