cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c8; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_local_shards.py -x -q -m gpu --durations=5 > $O/t_multi.log 2>&1; echo "multirank rc=$?" | tee -a $O/summary.txt; tail -8 $O/t_multi.log
bash tools/rehearse_multi.sh 2 400000 8 > $O/rehearse_2x_k8.log 2>&1; grep -E "exchange self-test:|^\{|exit code" $O/rehearse_2x_k8.log | cut -c1-1200 | tee -a $O/summary.txt
bash tools/rehearse_multi.sh 4 400000 16 > $O/rehearse_4x_k16.log 2>&1; grep -E "exchange self-test:|^\{|exit code" $O/rehearse_4x_k16.log | cut -c1-1200 | tee -a $O/summary.txt
